#!/bin/bash
# Round 5, call 3: what bounds k_mlp_rows?  Ablated builds (-DMR_XP=bits, WRONG results, timing only): 1 no small-column LDS reads,
# 2 one weight-fragment pair for the whole second layer (no fragment traffic), 4 big rows not loaded, 8 no wide part (hash + cross row),
# 16 no second-layer MFMAs, 31 all of them (the skeleton).  Config 5 and EmbeddingMLP.py's shape, strict launches.
# Also: the product's joint1 with 16 waves + the HOIST form keeping the early wait.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_03}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.3f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))"; }
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
use() { if [ $1 = product ]; then cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r05/libsparrow_hip_$1.so sparrowrecsys_amd/libsparrow_hip.so; fi; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mlp or joint or v2" > $O/pytest.log 2>&1
tail -1 $O/pytest.log
for lib in r04 product mrxp1 mrxp2 mrxp4 mrxp8 mrxp16 mrxp31; do
  use $lib
  a=$(timeout 300 python bench.py --workload widedeep_c5 --steps 100 --warmup 10 $STRICT 2>$O/c5_$lib.err | tail -1 | get)
  b=$(timeout 300 python bench.py --workload embedding_mlp_ref --steps 200 --warmup 20 $STRICT 2>$O/emb_$lib.err | tail -1 | get)
  echo "$lib: widedeep_c5 $a | embedding_mlp_ref $b" | tee -a $O/mlp_ablation.txt
done
for rep in 1 2; do
for lib in r04 product; do
  use $lib
  a=$(timeout 200 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)
  b=$(timeout 300 python bench.py --steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)
  c=$(timeout 300 python bench.py --workload deepfm_v2_c4 --steps 200 --warmup 20 $STRICT 2>/dev/null | tail -1 | get)
  echo "$lib: config 2 $a | HBM-resident $b | c4_v2 $c" | tee -a $O/joint1.txt
done
done
use product
