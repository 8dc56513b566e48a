#!/bin/bash
# whole-step time under the block-height switches of the conv kernels (model.py: VQW_X3_HALF*), two runs each
run() { for i in 1 2; do env "$@" python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-gen --no-other-engine --no-config4 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$*: %.2f ms/step' % r['ms_per_step'])"; done; }
run VQW_DUMMY=1
run VQW_X3_HALF_DGRAD=1
run VQW_X3_HALF_BWD=0
run VQW_X3_HALF=0
run VQW_X3_HALF_SKIP=1
run VQW_OVERLAP=0
