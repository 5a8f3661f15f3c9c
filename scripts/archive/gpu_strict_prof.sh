set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2s -o c2_strict -- python $R/bench.py --steps 400 --warmup 40 --cpu-seconds 0 --no-check --overlap-streams 0 --launch-batches 1 > $R/gpurun_out/prof_c2s.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3s -o c3_strict -- python $R/bench.py --steps 50 --warmup 5 --workload din_c3 --cpu-seconds 0 --no-check --overlap-streams 0 > $R/gpurun_out/prof_c3s.log 2>&1
cd $R
tail -1 gpurun_out/prof_c2s.log | cut -c1-1500 | grep -o '"avg_launch_us": [0-9.]*'; tail -1 gpurun_out/prof_c3s.log | grep -o '"avg_launch_us": [0-9.]*'
for f in $(find gpurun_out/prof_c2s gpurun_out/prof_c3s -name "*kernel_stats.csv"); do echo "--- $f"; head -3 $f | cut -c1-200; done
