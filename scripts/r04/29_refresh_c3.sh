#!/bin/bash
# Round 4: config 3's evidence again on the final tree (the strict trace, its untraced twin, the four PMC passes), the GPU suite and the
# driver's command -- the rest of profiles/r04 is from the tree two commits earlier, whose other kernels are unchanged.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_prof
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
cd /tmp && export TMPDIR=/tmp
w=c3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o t -- python $R/bench.py --steps 60 --warmup 6 --workload din_c3 $STRICT > $O/${w}_strict.log 2>&1
grep '^{"metric"' $O/${w}_strict.log | tail -1 > $O/${w}_strict_bench.json
f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_strict_kernel_stats.csv
rm -rf $O/trace_$w
head -3 $O/${w}_strict_kernel_stats.csv | cut -c1-160
cd $R
timeout 400 python bench.py --steps 60 --warmup 6 --workload din_c3 $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/c3_strict_untraced.json
bash scripts/r04/26_c3_pmc.sh > $O/c3_pmc_again.log 2>&1; tail -3 $O/c3_pmc_again.log | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -4 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_driver_command.json
python -c "
import json
l=json.loads(open('$O/bench_driver_command.json').read())
print('driver: value %.4g one-batch %.4g frac %.4f hbm %.4f' % (l['value'], l['value_one_batch_per_launch'], l['roofline']['frac'], l['roofline_hbm_resident']['frac']))
for k,w in l['workloads'].items(): print(k, ('%.4g' % w['value'], '%.4f' % w['roofline']['frac']) if 'value' in w else w.get('latency_ms'))"
