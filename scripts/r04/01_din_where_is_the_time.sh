#!/bin/bash
# Round 4, GPU call 1: what bounds k_din_attn_cols?  (a) ids from a 1 024-row window (--dist hot: every row an L1/L2 hit) against
# uniform ids, at 1 / 2 waves per task, strict launches; (b) PMC: memory latency seen by the waves (SQ_INST_LEVEL_VMEM /
# SQ_INSTS_VMEM), LDS, and the issue counters, on the attention kernel alone.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_01
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    r = l['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.4g' % l['value'], 'us/step %.2f' % (l['ms_per_step'] * 1e3), 'attention us %.2f frac %.3f' % (r['avg_launch_us'], r['frac']),
          'strict step us %.2f' % r.get('step_us_all_kernels', 0))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for ts in 1 2; do
  b c3_uniform_ts$ts SPRK_DIN_COLS_TS=$ts python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT
  b c3_hot_ts$ts SPRK_DIN_COLS_TS=$ts python bench.py --workload din_c3 --steps 120 --warmup 12 --dist hot $STRICT
done
b c3_mb python bench.py --workload din_c3 --steps 320 --warmup 32 --cpu-seconds 0 --no-check --side-workloads= --no-hardware-probe
b c3_mb_hot python bench.py --workload din_c3 --steps 320 --warmup 32 --dist hot --cpu-seconds 0 --no-check --side-workloads= --no-hardware-probe
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
pass() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1; }
CMD="python $R/bench.py --workload din_c3 --steps 20 --warmup 5 $STRICT"
pass lat SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $CMD
pass act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU -- $CMD
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES -- $CMD
pass lat_hot SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $CMD --dist hot
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum -- $CMD
cd $R
python - <<'PY'
import csv, glob, collections, os, json
summary = {}
for d in sorted(glob.glob('gpurun_out/r04_01/pmc_*/')):
    tag = os.path.basename(d.rstrip('/'))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if 'din' not in k: continue
            summary.setdefault(tag, {})[k[:40]] = {c: sum(v) / len(v) for c, v in cs.items()}
json.dump(summary, open('gpurun_out/r04_01/pmc_summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1))
PY
