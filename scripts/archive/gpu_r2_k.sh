#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "pairs or stated or deepfm" 2>&1 | tail -8 | tee gpurun_out/r2_pytest_k.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-160; tail -3 gpurun_out/$out.err; }
b r2k_c4_pairs python bench.py --workload deepfm_c4 --steps 200 --warmup 20 --cpu-seconds 0
b r2k_c2_pairs python bench.py --workload deepfm_c2 --steps 200 --warmup 20 --cpu-seconds 0
