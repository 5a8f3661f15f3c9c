#!/bin/bash
# Round 4: the two workloads whose tables are larger than the Infinity Cache (config 2 HBM-resident, config 4's DeepFM_v2) again on
# the final tree (they run k_deepfm_v2_joint1<HOIST> now): strict traces + untraced twins; then the GPU suite and the driver's command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_prof
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
declare -A WL
WL[c2_hbm]="--steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32"
WL[c4_v2]="--steps 200 --warmup 20 --workload deepfm_v2_c4"
cd /tmp && export TMPDIR=/tmp
for w in c2_hbm c4_v2; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o t -- python $R/bench.py ${WL[$w]} $STRICT > $O/${w}_strict.log 2>&1
  grep '^{"metric"' $O/${w}_strict.log | tail -1 > $O/${w}_strict_bench.json
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_strict_kernel_stats.csv
  rm -rf $O/trace_$w
  echo "$w: $(head -2 $O/${w}_strict_kernel_stats.csv | tail -1 | cut -c1-150)"
done
cd $R
for w in c2_hbm c4_v2; do
  timeout 400 python bench.py ${WL[$w]} $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${w}_strict_untraced.json
done
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -2 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_driver_command.json
python -c "
import json
l=json.loads(open('$O/bench_driver_command.json').read())
print('driver: value %.4g one-batch %.4g frac %.4f hbm %.4f' % (l['value'], l['value_one_batch_per_launch'], l['roofline']['frac'], l['roofline_hbm_resident']['frac']))
for k,w in l['workloads'].items(): print(k, ('%.4g' % w['value'], '%.4f' % w['roofline']['frac']) if 'value' in w else w.get('latency_ms'))"
