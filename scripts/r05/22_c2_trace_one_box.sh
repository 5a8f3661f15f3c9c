#!/bin/bash
# Round 5: config 2's strict trace on ONE more box ("quote the median over boxes, not the minimum") -- run as its own gpurun
# call, three times: gpurun_out/r05_prof/c2_box<tag>_kernel_stats.csv + the untraced HIP-event figure of the same box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_prof
mkdir -p $O
TAG=${1:-b}
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
cd /tmp && export TMPDIR=/tmp
for w in c2 c2_hbm; do
  X="--steps 400 --warmup 40 --input-batches 32"; [ $w = c2_hbm ] && X="$X --big-vocab 8388608"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${w}_$TAG -o t -- python $R/bench.py $X $STRICT > $O/${w}_box${TAG}_strict.log 2>&1
  f=$(find $O/trace_${w}_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_box${TAG}_kernel_stats.csv
  rm -rf $O/trace_${w}_$TAG
  timeout 300 python $R/bench.py $X $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${w}_box${TAG}_untraced.json
  python - $O/${w}_box${TAG}_kernel_stats.csv $O/${w}_box${TAG}_untraced.json $w $TAG <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_deepfm_v2_joint1" in r["Name"]]
r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
u = json.loads(open(sys.argv[2]).read())["roofline"]
print("%s box %s: rocprof avg %.3f us (stddev %s, %s launches) | HIP events untraced %.3f us" % (sys.argv[3], sys.argv[4], float(r["AverageNs"]) / 1e3, r.get("StdDev", "?"), r["Calls"], u["avg_launch_us"]))
PY
done
rocminfo | grep -E "Marketing Name|Uuid" | head -4 > $O/box${TAG}_rocminfo.txt 2>&1
