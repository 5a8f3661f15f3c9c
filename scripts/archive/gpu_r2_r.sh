#!/bin/bash
# run R: dress rehearsal of what the driver runs at round end
set -u
mkdir -p gpurun_out/r02r
O=gpurun_out/r02r
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 ) 2>&1 | grep -v "^NCCL\|^$" | tail -16 | tee $O/pytest_gpu.log
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -6 | tee $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -1 $O/bench_c2_driver.json | cut -c1-300; tail -5 $O/bench_c2_driver.err
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-200; tail -4 $O/bench_default.err
