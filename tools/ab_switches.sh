#!/bin/bash
# Whole-step time of bench.py under the block-height switches of the fp16x3 engine, on ONE box (boxes differ by several %).
# usage (GPU box, from the repo root): bash tools/ab_switches.sh
run() { python bench.py --steps 20 --warmup 5 --no-other-engine --no-gen --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3))"; }
run "default                       "
VQW_X3_HALF_BWD=1 run "VQW_X3_HALF_BWD=1   (128 rows)"
VQW_X3_HALF_BWD=0 run "VQW_X3_HALF_BWD=0   (256 rows)"
VQW_X3_HALF_SKIP=1 run "VQW_X3_HALF_SKIP=1  (128 rows)"
VQW_X3_HALF_DGRAD=1 run "VQW_X3_HALF_DGRAD=1 (128 rows)"
VQW_X3_HALF=0 run "VQW_X3_HALF=0       (256 rows)"
VQW_X3_HALF_BWD=1 run "VQW_X3_HALF_BWD=1   (128 rows)"
VQW_X3_HALF_BWD=0 run "VQW_X3_HALF_BWD=0   (256 rows)"
run "default again                 "
