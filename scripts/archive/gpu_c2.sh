#!/bin/bash
# config-2 round: DeepFM_v2 parity tests, bench A/B of the execution paths, kernel trace.
set -u
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "=== pytest deepfm"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "deepfm or config2 or config4 or forward_many or concurrent or ragged or missing" 2>&1 | tail -15 | tee gpurun_out/pytest_c2.log
echo "=== bench c2"
timeout 300 python bench.py --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/bench_c2_joint.json
SPRK_V2_HALF=0 timeout 300 python bench.py --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/bench_c2_joint_f32.json
timeout 300 python bench.py --cpu-seconds 0 --batch 1048576 --steps 400 --warmup 40 2>&1 | tail -1 | tee gpurun_out/bench_c2_b1m_joint.json
timeout 300 python bench.py --cpu-seconds 0 --dist zipf 2>&1 | tail -1 | tee gpurun_out/bench_c2_zipf_joint.json
