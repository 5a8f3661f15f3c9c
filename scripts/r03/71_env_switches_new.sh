#!/bin/bash
# the three switches added after scripts/r03/70_env_switches.sh ran, through the same tests
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_env
O=gpurun_out/r03_env/env_switches_new.log
: > $O
SEL="tests/test_reference_blocks.py tests/test_gpu_shape_sweep.py tests/test_savedmodel_pins.py tests/test_gpu_ingest.py"
for sw in SPRK_TAIL_UNF=0 SPRK_TAIL_POOLED_F16=0 SPRK_DIEN_MFMA=0; do
  a=$(env $sw timeout 600 python -m pytest $SEL -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1)
  b=$(env $sw timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_chain.py -q -m gpu -k "several_batches_per_launch or golden or dien_vs_oracle or din" -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "$sw: $a | $b" | tee -a $O
done
