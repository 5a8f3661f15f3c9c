#!/bin/bash
# A/B of the dynamic-scale f16 fc1 in k_din_tail (SPRK_DYN_F16=1/0): parity, then kernel stats.
set -u
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "=== pytest din tail"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "din_tail" 2>&1 | tail -15 | tee gpurun_out/pytest_dyn.log
cd /tmp && export TMPDIR=/tmp
for d in 1 0; do
  SPRK_DYN_F16=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dyn$d -o c3 -- python $R/bench.py --steps 100 --warmup 10 --workload din_c3 --cpu-seconds 0 --overlap-streams 0 > $R/gpurun_out/prof_dyn$d.log 2>&1
  tail -1 $R/gpurun_out/prof_dyn$d.log | cut -c1-400
  for f in $(find $R/gpurun_out/prof_dyn$d -name "*kernel_stats.csv"); do head -4 $f | cut -c1-160; done
done
