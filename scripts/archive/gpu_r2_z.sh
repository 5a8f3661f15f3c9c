#!/bin/bash
# run Z: every A/B environment switch once through the golden / parity tests
set -u
mkdir -p gpurun_out/r02z
for e in SPRK_V2_HALF=0 SPRK_V2_ROWS=1 SPRK_V2_JOINT=0 SPRK_DYN_F16=0 SPRK_DIN_HALF=0 SPRK_DIN_WPB=4 SPRK_DIN_WPB=16 SPRK_DIN_ATTN_MB=0 SPRK_DIN_LEGACY=1 \
         SPRK_V1_ROWTAB=0 SPRK_V1_STATIC_SCALE=0 SPRK_V1_CHAIN=0 SPRK_MLP_ROWS=0 SPRK_TILE_FOLD=0 SPRK_NCF_CHAIN=0 SPRK_CSV_TWO_PASS=1; do
  r=$(env $e timeout 600 python -m pytest tests -m gpu -q -x -k "golden or sweep or several_batches or ingest or smoke" 2>&1 | grep -E "passed|failed" | tail -1)
  echo "$e: $r" | tee -a gpurun_out/r02z/switches.log
done
