#!/bin/bash
# Functional run of the N>1 bench path on the 1-GPU box: two gloo ranks sharing cuda:0 (RCCL refuses two
# ranks per device; the 8-GPU RCCL run is the driver's).  Also N=1 sanity.
set -u
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 200 --warmup 40 --cpu-seconds 0 2>&1 | tail -3 | tee gpurun_out/bench_c2_gloo2.json | cut -c1-600
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --backend gloo --steps 40 --warmup 10 --cpu-seconds 0 --workload din_c3 2>&1 | tail -2 | tee gpurun_out/bench_c3_gloo2.json | cut -c1-600
timeout 300 python bench.py --steps 500 --warmup 50 --cpu-seconds 0 2>&1 | tail -1 | cut -c1-300
