#!/bin/bash
# whole-step time against the batch sizes of the weight-gradient launches (model.py backward)
# usage: wg_batch_sweep.sh ["4 6 8 12 24"] ["10 15 29"]
GS=${1:-"4 6 8 12 24"}; RS=${2:-"10 15 29"}
for g in $GS; do for r in $RS; do
  VQW_WG_GATE_BATCH=$g VQW_WG_RES_BATCH=$r python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-gen --no-other-engine --no-config4 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('gate $g res $r: %.2f ms/step, wgrad single-stream %.2f ms' % (r['ms_per_step'], r['roofline_wgrad']['single_stream']['ms_per_step']))"
done; done
