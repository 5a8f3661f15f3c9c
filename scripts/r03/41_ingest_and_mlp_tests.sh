#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1200 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_shape_sweep.py -x -q -m gpu 2>&1 | grep -v "NCCL\|RCCL" | tail -8
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mlp or wide or embedding" 2>&1 | grep -v "NCCL\|RCCL" | tail -5
