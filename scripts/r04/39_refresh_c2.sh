#!/bin/bash
# Round 4: config 2's strict trace and untraced twin on the final tree (the kernel's symbol gained a template parameter).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_prof
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
cd /tmp && export TMPDIR=/tmp
w=c2
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o t -- python $R/bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT > $O/${w}_strict.log 2>&1
grep '^{"metric"' $O/${w}_strict.log | tail -1 > $O/${w}_strict_bench.json
f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_strict_kernel_stats.csv
rm -rf $O/trace_$w
head -2 $O/${w}_strict_kernel_stats.csv | tail -1 | cut -c1-170
cd $R
timeout 400 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/c2_strict_untraced.json
python -c "
import json;l=json.loads(open('$O/c2_strict_untraced.json').read());print('untraced %.3f us frac %.4f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))"
