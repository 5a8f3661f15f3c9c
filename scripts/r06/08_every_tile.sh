#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_11}
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_stated_sizes.py -m gpu -x -q -k "every_tile" --durations=10 > $O/pytest_every_tile.log 2>&1; tail -25 $O/pytest_every_tile.log
