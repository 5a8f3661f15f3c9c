#!/bin/bash
# Round 5, call 2: (a) k_mlp_rows re-written from its ISA (k_mlp_rows.h header: NS as a template parameter, padded instead of swizzled LDS
# rows, single-instruction ReLU, the cross hash's modulo as a multiply, fragments a group ahead, the gather's registers coalesced) against
# round 4's library on config 5 and EmbeddingMLP.py's shape, parity tests first; (b) k_deepfm_v2_joint1 with the HOIST form's copies behind
# the gathers (product) against 16 waves per workgroup (w16) and round 4; (c) the stamped timeline of the new joint1.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_02}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stated_sizes.py -m gpu -x -q -k "mlp or wide or config5 or joint or v2" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.3f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))"; }
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
use() { if [ $1 = product ]; then cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r05/libsparrow_hip_$1.so sparrowrecsys_amd/libsparrow_hip.so; fi; }
for rep in 1 2; do
for lib in r04 product; do
  use $lib
  a=$(timeout 300 python bench.py --workload widedeep_c5 --steps 100 --warmup 10 $STRICT 2>$O/c5_$lib.err | tail -1 | get)
  b=$(timeout 300 python bench.py --workload embedding_mlp_ref --steps 200 --warmup 20 $STRICT 2>$O/emb_$lib.err | tail -1 | get)
  echo "$lib: widedeep_c5 $a | embedding_mlp_ref $b" | tee -a $O/mlp.txt
done
done
for rep in 1 2; do
for lib in r04 product w16; do
  use $lib
  a=$(timeout 200 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)
  b=$(timeout 300 python bench.py --steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)
  c=$(timeout 300 python bench.py --workload deepfm_v2_c4 --steps 200 --warmup 20 $STRICT 2>/dev/null | tail -1 | get)
  echo "$lib: config 2 $a | HBM-resident $b | c4_v2 $c" | tee -a $O/joint1.txt
done
done
use xp
SPRK_V2J1_TS_FILE=$O/ts_c2.bin timeout 200 python bench.py --steps 60 --warmup 10 --input-batches 32 $STRICT 2>$O/c2.err | tail -1 > $O/c2_xp.json
SPRK_V2J1_TS_FILE=$O/ts_c2_hbm.bin timeout 300 python bench.py --steps 60 --warmup 10 --big-vocab 8388608 --input-batches 32 $STRICT 2>$O/c2_hbm.err | tail -1 > $O/c2_hbm_xp.json
use product
for w in c2 c2_hbm; do
  python scripts/r04/v2j1_timeline.py $O/ts_$w.bin | tee $O/timeline_$w.txt
done
rm -f $O/ts_*.bin
