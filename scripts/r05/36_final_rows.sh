#!/bin/bash
# Round 5, last evidence call: after k_dien_fused, the fence in k_dien_seq_mfma and k_din_tail's task as shared functions (its ISA moved by a few
# instructions) -- the strict traces + untraced twins of the two workloads those touch (din_ref, dien_ref) into gpurun_out/r05_prof next to the
# others of scripts/r05/20_profiles.sh, the whole GPU suite and the driver's command on the final tree.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_prof
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
declare -A WL
WL[din_ref]="--steps 100 --warmup 10 --workload din_ref"
WL[dien_ref]="--steps 100 --warmup 10 --workload dien_ref"
cd /tmp && export TMPDIR=/tmp
for w in din_ref dien_ref; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o t -- python $R/bench.py ${WL[$w]} $STRICT > $O/${w}_strict.log 2>&1
  grep '^{"metric"' $O/${w}_strict.log | tail -1 > $O/${w}_strict_bench.json
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_strict_kernel_stats.csv
  rm -rf $O/trace_$w $O/${w}_strict.log
  head -4 $O/${w}_strict_kernel_stats.csv | cut -c1-150
done
cd $R
for w in din_ref dien_ref; do
  timeout 400 python bench.py ${WL[$w]} $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${w}_strict_untraced.json
done
SPRK_DIEN_FUSED=0 timeout 400 python bench.py ${WL[dien_ref]} $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/dien_ref_two_launches_strict_untraced.json
python -c "
import json
for f in ('din_ref_strict_untraced', 'dien_ref_strict_untraced', 'dien_ref_two_launches_strict_untraced'):
    r = json.loads(open('$O/' + f + '.json').read())['roofline']
    print(f, r['kernel'][:40], 'launch %.3f us' % r['avg_launch_us'], 'step %.3f us' % r.get('step_us_all_kernels', r['avg_launch_us']))"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -6 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_driver_command.json
python -c "
import json
l=json.loads(open('$O/bench_driver_command.json').read())
print('driver: value %.4g one-batch %.4g frac %.4f hbm %.4f' % (l['value'], l['value_one_batch_per_launch'], l['roofline']['frac'], l['roofline_hbm_resident']['frac']))"
