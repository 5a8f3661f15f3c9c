#!/bin/bash
# Round 5, call 8: k_din_fused with fc0's pooled-independent part computed inside the slot loop, at a different trip per wave (product)
# against in the epilogue (libsparrow_hip_noearly.so = -DDF_EARLY_FC0=0); parity (incl. launch-shape invariance) first; the stamped timeline last.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_08}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py tests/test_gpu_stated_sizes.py -m gpu -x -q -k "din" > $O/pytest_din.log 2>&1; tail -1 $O/pytest_din.log
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.3f us frac %.3f (attention only %.3f us) | value %.4g samples/s (%.3f us/step)' % (r['avg_launch_us'], r['frac'], r.get('attention_only',{}).get('avg_launch_us',0), l['value'], l['ms_per_step']*1e3))"; }
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
use() { if [ $1 = product ]; then cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r05/libsparrow_hip_$1.so sparrowrecsys_amd/libsparrow_hip.so; fi; }
for rep in 1 2; do
for lib in product noearly early product noearly early; do
  use $lib
  echo "$lib: $(timeout 300 python bench.py --workload din_c3 --steps 60 --warmup 6 --cpu-seconds 0 --no-check --side-workloads= --no-hardware-probe --hbm-resident 0 2>$O/c3_$lib.err | tail -1 | get)" | tee -a $O/din_c3.txt
done
done
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
use xp
SPRK_DF_XP=1024 SPRK_DF_TS_FILE=$O/ts.bin timeout 200 python bench.py --workload din_c3 --steps 40 --warmup 8 $STRICT 2>$O/ts.err | tail -1 > $O/ts.json
use product
python scripts/r04/din_fused_timeline.py $O/ts.bin $O/ts.json | tee $O/timeline.txt
rm -f $O/ts.bin
