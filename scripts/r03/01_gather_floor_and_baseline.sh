#!/bin/bash
# Round 3, GPU call 1: (a) the row-gather floor (scripts/ubench/row_gather: HBM vs Infinity Cache vs translation footprint),
# (b) which translation counters this rocprofv3 exposes, (c) the GPU suite on the round-2 build, (d) the default driver line
# with the new `workloads` block and the 32-batch HBM-resident loop, (e) translation counters on the HBM-resident kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_01
mkdir -p $O
echo "=== row_gather"; timeout 300 scripts/ubench/row_gather 2>&1 | tee $O/ubench_row_gather.log
echo "=== counters"; (rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) > $O/counters_all.txt 2>&1
grep -i -o "\b[A-Z0-9_]*\(UTCL\|TLB\|XNACK\)[A-Za-z0-9_]*" $O/counters_all.txt | sort -u | tr '\n' ' ' | tee $O/counters_tlb.txt; echo
echo "=== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
echo "=== bench default"; /usr/bin/time -v timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 600 $O/bench_driver.err | grep -E "Elapsed|Exit"; tail -1 $O/bench_driver.json | python -c "
import sys, json
l = json.loads(sys.stdin.readline())
print('value', l['value'], 'one-batch', l['value_one_batch_per_launch'], 'frac', l['roofline']['frac'])
print('hbm_resident', {k: l['roofline_hbm_resident'][k] for k in ('frac','avg_launch_us','working_set_mb','frac_16_batches_per_launch')})
for k, w in l.get('workloads', {}).items():
    print(k, w['value'], w['ms_per_step'], w['roofline']['frac'], w['roofline']['avg_launch_us'], w['oracle_check_max_abs_err'])
"
cd /tmp && export TMPDIR=/tmp
HB="python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-check --big-vocab 8388608 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --input-batches 32"
C2="python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
echo "=== strict kernel trace, HBM-resident (32 batches cycled)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hbm -o hbm -- python $R/bench.py --steps 400 --warmup 40 --cpu-seconds 0 --no-check --big-vocab 8388608 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --input-batches 32 > $O/prof_hbm.log 2>&1
for f in $(find $O/prof_hbm -name "*kernel_stats.csv"); do head -4 $f | cut -c1-220; done
pass() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"; }
pass hbm_utcl1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -- $HB
pass c2_utcl1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -- $C2
pass hbm_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- $HB
pass hbm_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -- $HB
pass c2_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -- $C2
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/r03_01/pmc_*/')):
    tag = os.path.basename(d.rstrip('/'))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if 'joint' not in k: continue
            print(tag, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, 'launches', len(next(iter(cs.values()))))
PY
