#!/bin/bash
# SQ counters (MFMA busy, CU busy, clock) of every kernel of bench.py's training step, one --pmc pass: gpurun_out/r3_pmc_step.txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_step
VQW_OVERLAP=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_step -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gen --no-other-engine --no-config4 > $OUT/r3_pmc_step.json 2> $OUT/r3_pmc_step.log
python3 $REPO/tools/pmc_step.py /tmp/pmc_step $OUT/r3_pmc_step.txt
cat $OUT/r3_pmc_step.txt
