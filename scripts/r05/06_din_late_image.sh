#!/bin/bash
# Round 5, call 6: k_din_fused with the tail's image and raw rows requested BEHIND the first trip of the slot loop (product) against in
# front of it (libsparrow_hip_early.so = -DDF_LATE_IMAGE=0, round 4's order): config 3 strict + several batches per launch; DIN parity first.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_06}
mkdir -p $O
echo "(parity: first run of this script)"
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.3f us frac %.3f (attention only %.3f us) | value %.4g samples/s (%.3f us/step)' % (r['avg_launch_us'], r['frac'], r.get('attention_only',{}).get('avg_launch_us',0), l['value'], l['ms_per_step']*1e3))"; }
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for rep in 1 2; do
for lib in product early product early; do
  if [ $lib = product ]; then cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r05/libsparrow_hip_$lib.so sparrowrecsys_amd/libsparrow_hip.so; fi
  echo "$lib: $(timeout 300 python bench.py --workload din_c3 --steps 60 --warmup 6 --cpu-seconds 0 --no-check --side-workloads= --no-hardware-probe --hbm-resident 0 2>$O/c3_$lib.err | tail -1 | get)" | tee -a $O/din_c3.txt
done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
