#!/bin/bash
# Every A/B switch of SprkTuning (host_common.h) through the reference-lines goldens, the randomised shape sweep, the executed-graph
# pins and the several-batches-per-launch tests: whatever kernel a switch selects must give the same answers.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_env
O=gpurun_out/r03_env/env_switches_pytest.log
: > $O
SEL="tests/test_reference_blocks.py tests/test_gpu_shape_sweep.py tests/test_savedmodel_pins.py tests/test_gpu_ingest.py"
KSEL="several_batches_per_launch or golden"
for sw in "" SPRK_V2_HALF=0 SPRK_V2_ROWS=1 SPRK_V2_JOINT=0 SPRK_V2_FOLD=0 SPRK_V2J_ONE=0 SPRK_ROWS_ONE=0 SPRK_ROWS_UNF=0 SPRK_TAIL_UNF=0 SPRK_TAIL_POOLED_F16=0 SPRK_DIEN_MFMA=0 SPRK_DYN_F16=0 SPRK_DIN_HALF=0 \
          SPRK_DIN_COLS=0 SPRK_DIN_COLS_TS=1 SPRK_DIN_COLS_TS=2 SPRK_DIN_COLS_TS=4 SPRK_DIN_WPB=4 SPRK_DIN_WPB=16 SPRK_DIN_ATTN_MB=0 SPRK_DIN_LEGACY=1 \
          SPRK_DIN_TAIL=0 SPRK_V1_ROWTAB=0 SPRK_V1_STATIC_SCALE=0 SPRK_V1_CHAIN=0 SPRK_V1_ONE=0 SPRK_MLP_CHAIN=0 SPRK_TILE_FOLD=0 SPRK_NCF_CHAIN=0 \
          SPRK_CSV_TWO_PASS=1 SPRK_HALF_RANGE_GUARD=0 SPRK_FORCE_INTERPRETER=1 SPRK_ROCTX=1; do
  a=$(env $sw timeout 600 python -m pytest $SEL -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1)
  b=$(env $sw timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_chain.py -q -m gpu -k "$KSEL" -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "${sw:-(defaults)}: $a | $b" | tee -a $O
done
