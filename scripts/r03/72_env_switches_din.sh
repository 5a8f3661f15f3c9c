#!/bin/bash
# switches that select other code around the round's late DIN / DIEN kernels, through the DIN / DIEN tests and the goldens
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_env
O=gpurun_out/r03_env/env_switches_din.log
: > $O
for sw in SPRK_DYN_F16=0 SPRK_DIN_TAIL=0 SPRK_FORCE_INTERPRETER=1 SPRK_TILE_FOLD=0; do
  a=$(env $sw timeout 600 python -m pytest tests/test_reference_blocks.py tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py -q -m gpu -k "din or dien or golden or hip_matches or tail" -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -6 | tr "\n" ";")
  echo "$sw: $a" | tee -a $O
done
