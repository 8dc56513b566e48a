#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of bench.py's step (5 steps after 2 warm-up): the default engine on two streams
# and on one (VQW_OVERLAP=0), and the fp32-MFMA engine.  Summaries land in gpurun_out/r3_prof_<name>_kernel_stats.csv.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, env assignments...
  name=$1; shift
  rm -rf /tmp/prof_$name
  env "$@" true
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o $name -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gen --no-other-engine --no-config4 > $OUT/r3_prof_$name.json 2> $OUT/r3_prof_$name.log )
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/r3_prof_${name}_kernel_stats.csv; else
    db=$(find /tmp/prof_$name -name "*.db" | head -1); [ -n "$db" ] && python3 $REPO/tools/rocpd_stats.py "$db" $OUT/r3_prof_${name}_kernel_stats.csv; fi
  t=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && [ "$name" = single_stream ] && python3 $REPO/tools/step_trace.py "$t" > $OUT/r3_prof_${name}_launches.txt 2>/dev/null
  echo "== $name"; head -12 $OUT/r3_prof_${name}_kernel_stats.csv
}
run two_streams VQW_DUMMY=1
run single_stream VQW_OVERLAP=0
run fp32_engine VQW_ENGINE=fp32 VQW_OVERLAP=0
