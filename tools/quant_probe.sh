#!/bin/bash
# Tile-quantisation probe: the engine's kernels at B=8 with T = 6656 (208 column tiles), 4096 (128: whole rounds on
# 256 CUs) and 2560 (80: the tail), both block heights.
for half in 0 1; do for t in 6656 4096 2560; do
  echo "=== VQW_X3_HALF=$half T=$t"; VQW_X3_HALF=$half XT=$t python tools/x3_bench.py 15 2>&1 | grep -v "^VQW\|amdgpu.ids"
done; done
