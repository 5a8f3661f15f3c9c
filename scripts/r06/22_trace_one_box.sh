#!/bin/bash
# Round 6: config 2 (cache-resident and HBM-resident tables) and config 3 strict traces on ONE more box (a gpurun call each: a fresh box) --
# the medians DESIGN quotes.  gpurun_out/r06_prof/<w>_box<tag>_kernel_stats.csv + the untraced HIP-event figure of the same box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_prof
mkdir -p $O
TAG=${1:-b}
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-c2 c2_hbm c3}; do          # (WORKLOADS="c5": config 5 after round 6's k_mlp_rows change)
  X="--steps 400 --warmup 40 --input-batches 32"; [ $w = c2_hbm ] && X="$X --big-vocab 8388608"; [ $w = c3 ] && X="--steps 60 --warmup 6 --workload din_c3"
  [ $w = c5 ] && X="--steps 100 --warmup 10 --workload widedeep_c5"
  K=k_deepfm_v2_joint1; [ $w = c3 ] && K="k_din_fused<2, false, true"; [ $w = c5 ] && K="k_mlp_rows<"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${w}_$TAG -o t -- python $R/bench.py $X $STRICT > $O/${w}_box${TAG}_strict.log 2>&1
  f=$(find $O/trace_${w}_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_box${TAG}_kernel_stats.csv
  rm -rf $O/trace_${w}_$TAG
  timeout 300 python $R/bench.py $X $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${w}_box${TAG}_untraced.json
  python - $O/${w}_box${TAG}_kernel_stats.csv $O/${w}_box${TAG}_untraced.json $w $TAG "$K" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[5] in r["Name"]]
r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
u = json.loads(open(sys.argv[2]).read())["roofline"]
print("%s box %s: rocprof avg %.3f us (stddev %s, %s launches) | HIP events untraced %.3f us" % (sys.argv[3], sys.argv[4], float(r["AverageNs"]) / 1e3, r.get("StdDev", "?"), r["Calls"], u["avg_launch_us"]))
PY
done
rocminfo | grep -E "Marketing Name|Uuid" | head -4 > $O/box${TAG}_rocminfo.txt 2>&1
