#!/bin/bash
# Round 5, call 5: (a) DIN.py's literal shape (hist_len 5, emb_dim 10) through k_din_fused<1, ..., TAIL> with the tail's four embedding columns as
# raw 64-byte rows (new: k_din_fused.h KC = 1 UNF) -- SPRK_DIN_FUSED_MIN_T=1 -- against the two-launch path (k_din_attn_cols -> k_din_tail):
# parity tests under both, then strict and several-batches-per-launch timings; (b) config 4's DeepFM_v2 on the 16-wave HOIST form; (c) the new
# sharded-table tests (27 M rows over two processes, the peer-denied exit, close / collect); (d) the REST shim behind 8 worker processes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_05}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());print('strict %.3f us frac %.3f | value %.4g samples/s (%.3f us/step)' % (l['roofline'].get('step_us_all_kernels', l['roofline']['avg_launch_us']), l['roofline']['frac'], l['value'], l['ms_per_step']*1e3))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py -m gpu -x -q -k "din or dien" > $O/pytest_din_default.log 2>&1; tail -1 $O/pytest_din_default.log
SPRK_DIN_FUSED_MIN_T=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py -m gpu -x -q -k "din" > $O/pytest_din_min1.log 2>&1; tail -1 $O/pytest_din_min1.log
for rep in 1 2; do
  a=$(timeout 300 python bench.py --workload din_ref --steps 200 --warmup 20 --cpu-seconds 0 --side-workloads= --no-hardware-probe --hbm-resident 0 2>$O/din_ref_default.err | tail -1 | get)
  b=$(SPRK_DIN_FUSED_MIN_T=1 timeout 300 python bench.py --workload din_ref --steps 200 --warmup 20 --cpu-seconds 0 --side-workloads= --no-hardware-probe --hbm-resident 0 2>$O/din_ref_min1.err | tail -1 | get)
  echo "din_ref two launches: $a" | tee -a $O/din_ref.txt
  echo "din_ref k_din_fused : $b" | tee -a $O/din_ref.txt
done
g2() { python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.3f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))"; }
for rep in 1 2; do
  echo "c4_v2 $(timeout 300 python bench.py --workload deepfm_v2_c4 --steps 200 --warmup 20 $STRICT 2>/dev/null | tail -1 | g2)" | tee -a $O/c4.txt
done
timeout 900 python -m pytest tests/test_gpu_sharded_table.py -m gpu -x -q > $O/pytest_sharded.log 2>&1; tail -3 $O/pytest_sharded.log
for w in 1 8; do
  timeout 300 python scripts/bench_serving.py --clients 8 --seconds 4 --workers $w > $O/serving_w$w.json 2>$O/serving_w$w.err; cat $O/serving_w$w.json | cut -c1-260
done
timeout 200 python scripts/bench_serving.py --clients 1 --seconds 3 --workers 8 > $O/serving_w8_c1.json 2>/dev/null; cut -c1-260 $O/serving_w8_c1.json
