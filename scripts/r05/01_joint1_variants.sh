#!/bin/bash
# Round 5: k_deepfm_v2_joint1 A/B of two changes read off the ISA (no GPU needed to find them):
#   zzlate: the rows' first-order scalar (the LAST load a wave issues) consumed behind the row fence -- hipcc had put its vmcnt(0) behind the
#           first numerics MFMA, so five f32 MFMAs + ten LDS reads of "phase A" ran only after every row had landed;
#   dedup : A fragments stored once per (field, n-block) as {hi4 | lo4} (one ds_read_b128 where there were two; the {h, h} operand is two
#           register copies), the 0/1 selection fragment built in registers: 28 -> 21 ds_read_b128 per wave, image 17 -> 10 KB per workgroup.
# Libraries: r04 (HEAD of round 4), zzlate, dedup, product (both), w16 (both + 16 waves per workgroup).  Parity of the DeepFM_v2 kernels first.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_01}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v2 or joint" > $O/pytest_v2.log 2>&1
tail -1 $O/pytest_v2.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.3f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))"; }
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for rep in 1 2; do
for lib in r04 zzlate dedup product w16; do
  if [ $lib = product ]; then cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r05/libsparrow_hip_$lib.so sparrowrecsys_amd/libsparrow_hip.so; fi
  a=$(timeout 200 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)
  b=$(timeout 300 python bench.py --steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)
  c=$(timeout 300 python bench.py --workload deepfm_v2_c4 --steps 200 --warmup 20 $STRICT 2>/dev/null | tail -1 | get)
  echo "$lib: config 2 $a | HBM-resident $b | c4_v2 $c" | tee -a $O/variants.txt
done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
