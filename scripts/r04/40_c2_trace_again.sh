#!/bin/bash
# Round 4: one more strict trace of config 2 on the final tree (another box: the first one's tracer read 10 % above the untraced time)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_c2b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 400 --warmup 40 --input-batches 32 --cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= > $O/log.txt 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c2_strict_kernel_stats.csv
grep '^{"metric"' $O/log.txt | tail -1 > $O/c2_strict_bench.json
rm -rf $O/trace
head -2 $O/c2_strict_kernel_stats.csv | tail -1 | cut -c100-260
