#!/bin/bash
# rocprofv3 kernel stats of BASELINE.json configs[4] on one GPU: '2019' encoder, T = 6400, bf16 storage + fp32 accumulate
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bf16
VQW_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf16 -o bf16 -- python3 $REPO/bench.py --encoder 2019 --length 6400 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-gen --no-other-engine --no-config4 > $OUT/r3_prof_bf16.json 2> $OUT/r3_prof_bf16.log
f=$(find /tmp/prof_bf16 -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_prof_bf16_kernel_stats.csv
head -25 $OUT/r3_prof_bf16_kernel_stats.csv | cut -c1-200
