#!/bin/bash
# Functional + host-cost check of bench.py's N>1 code path on ONE GPU: WORLD_SIZE 1, RCCL process group, the grouped
# all-gather issued for real on the side stream (SPRK_BENCH_FORCE_DIST / SPRK_FORCE_COLLECTIVE).
set -u
mkdir -p gpurun_out
export SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1
for g in 16 32 64; do
  timeout 300 python bench.py --steps 2048 --warmup 256 --cpu-seconds 0 --gather-group $g > gpurun_out/force_dist_g$g.log 2>&1
  echo "G=$g rc=$? last line is json: $(tail -1 gpurun_out/force_dist_g$g.log | cut -c1-120)"
done
timeout 300 python bench.py --steps 1003 --warmup 37 --cpu-seconds 0 > gpurun_out/force_dist_odd.log 2>&1; echo "odd rc=$? $(tail -1 gpurun_out/force_dist_odd.log | cut -c1-120)"
timeout 300 python bench.py --steps 400 --warmup 40 --cpu-seconds 0 --workload din_c3 > gpurun_out/force_dist_din.log 2>&1; echo "din rc=$? $(tail -1 gpurun_out/force_dist_din.log | cut -c1-120)"
unset SPRK_BENCH_FORCE_DIST SPRK_FORCE_COLLECTIVE
timeout 300 python bench.py --steps 2048 --warmup 256 --cpu-seconds 0 2>&1 | tail -1 | cut -c1-120
