#!/bin/bash
# Round 5, call 10: the whole GPU suite on the seven-unit build (kernels defined in tu_1 .. tu_6.hip, launched from sparrow_hip.hip through
# `extern template`), smoke(), and the strict figures of the four BASELINE configs as a regression check against the one-unit build.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_10}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.3f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))"; }
echo "c2 $(timeout 200 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)" | tee -a $O/strict.txt
echo "c2_pairs $(timeout 200 python bench.py --workload deepfm_c2 --steps 400 --warmup 40 $STRICT 2>/dev/null | tail -1 | get)" | tee -a $O/strict.txt
echo "c3 $(timeout 200 python bench.py --workload din_c3 --steps 60 --warmup 6 $STRICT 2>/dev/null | tail -1 | get)" | tee -a $O/strict.txt
echo "c5 $(timeout 300 python bench.py --workload widedeep_c5 --steps 100 --warmup 10 $STRICT 2>/dev/null | tail -1 | get)" | tee -a $O/strict.txt
echo "v2_ref $(timeout 300 python bench.py --workload deepfm_v2_ref --steps 400 --warmup 40 $STRICT 2>/dev/null | tail -1 | get)" | tee -a $O/strict.txt
