#!/bin/bash
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters; one pass each, kernel trace only beside them) over single
# kernels of the default engine as round 3 runs them (tools/x3_one.py).  Results: gpurun_out/r3_pmc_*.{json,txt}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
one() {  # name, x3_one mode, kernel substring, algorithmic bytes, note
  name=$1; mode=$2; kn=$3; alg=$4; note=$5
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${name}_$c -- python3 $REPO/tools/x3_one.py $mode 8 > /dev/null 2>&1
  done
  python3 $REPO/tools/pmc_traffic.py /tmp/pmc_${name}_FETCH_SIZE /tmp/pmc_${name}_WRITE_SIZE "$kn" $OUT/r3_pmc_${name}_traffic.json $alg "$note"
  rm -rf /tmp/pmc_${name}_sq
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_${name}_sq -- python3 $REPO/tools/x3_one.py $mode 8 > /dev/null 2>&1
  python3 $REPO/tools/pmc_sq.py /tmp/pmc_${name}_sq "$kn" $OUT/r3_pmc_${name}_sq.txt; cat $OUT/r3_pmc_${name}_sq.txt
}
one gate gate_r3 gate_f16x3_kernel 165150720 "gate_f16x3_kernel<fp16x3, 128-row blocks> as round 3 runs it (no fp32 gated output, tanh not stored): B=8 T=6656 256->512 k=3 d=8; algorithmic = input planes 54.5 MB + sigmoid fp32 54.5 MB + gated planes 54.5 MB + weight planes 1.6 MB"
one gate_bwd gate_bwd_r3 gate_bwd_f16x3_kernel 381681664 "gate_bwd_f16x3_kernel<fp16x3, 128-row blocks, gated from planes>: dpre as planes only; algorithmic = [dskip|dnet] planes 163.6 MB + gated planes 54.5 MB + sigmoid 54.5 MB read, dpre planes 109.1 MB written"
one wgrad_batch wgrad_batch wgrad_f16x3_kernel 1054867456 "wgrad_f16x3_kernel<p, q from planes> over SIX layers' gate kernels in one launch (36 tiles x 7 K splits): algorithmic = 6 x (input planes 54.5 MB + dpre planes 109.1 MB) read + slab 252 x 256 KB written (re-read by the reduction launch, not counted here)"
