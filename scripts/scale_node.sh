#!/bin/bash
# ONE command for the day a multi-GPU MI355X node appears.  No scaling curve has ever been measured: every
# round's boxes had one GPU.  This runs, on N in {1, 2, 4, 8} (or $SCALE_GPUS):
#   * the headline (DeepFM_v2, config 2) and DIN (config 3), weak AND strong scaling, over the three score exchanges
#       torch  torch.distributed.all_gather_into_tensor (RCCL)      sprk  ncclAllGather behind the C ABI (sprk_comm_*)
#       peer   direct peer writes over xGMI (sprk_peer_*, k_peer_gather.h)
#   * config 4 with its 27 M-row tables ROW-SHARDED over the ranks (sprk_vtable_*: peers' rows loaded over xGMI by the fused kernel)
#     against the replicated tables;
# and writes gpurun_out/scale/scale_table.{json,md}: samples/s per (workload, scaling, collective, N), the efficiency against N = 1,
# and every bench line.  bench.py starts its own ranks (one process per GPU, 127.0.0.1 rendezvous) when no launcher did.
#   usage: bash scripts/scale_node.sh [out_dir]      env: SCALE_GPUS="1 2 4 8"  SCALE_STEPS=200  SCALE_WARMUP=20  SCALE_DRY=1 (CPU plumbing run)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=${1:-$R/gpurun_out/scale}
mkdir -p "$O"
GPUS=${SCALE_GPUS:-"1 2 4 8"}
STEPS=${SCALE_STEPS:-200}
WARM=${SCALE_WARMUP:-20}
export HSA_ENABLE_IPC_MODE_LEGACY=0
have=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "devices visible: $have" | tee "$O/devices.txt"
rocm-smi --showtopo > "$O/topology.txt" 2>&1 || true
COMMON="--steps $STEPS --warmup $WARM --cpu-seconds 0 --side-workloads= --no-hardware-probe --hbm-resident 0 --variants 0"
# SCALE_DRY=1: the same matrix of runs over gloo with a stand-in forward on CPU tensors (bench.py --dry-run): what tests/test_bench_cpu.py
# runs to keep this script alive without a node
if [ "${SCALE_DRY:-0}" = 1 ]; then COMMON="$COMMON --dry-run --backend gloo --batch 256"; have=8; fi
run() { # tag, n, args...
  local tag=$1 n=$2; shift 2
  if [ "$n" -gt "$have" ]; then echo "skip $tag (needs $n GPUs, $have visible)"; return; fi
  timeout 900 python "$R/bench.py" --gpus "$n" $COMMON "$@" > "$O/$tag.log" 2>&1
  grep '^{"metric"' "$O/$tag.log" | tail -1 > "$O/$tag.json"
  python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read())
    print("%-44s n=%d  %.4g samples/s  %.3f us/step" % (sys.argv[2], l["n_gpus"], l["value"], l["ms_per_step"] * 1e3))
except Exception as e:
    print("%-44s FAILED (%s): see the .log" % (sys.argv[2], e))
PY
}
for n in $GPUS; do
  for coll in torch sprk peer; do
    [ "$n" = 1 ] && [ "$coll" != torch ] && continue
    for sc in weak strong; do
      run "c2_${sc}_${coll}_n$n" "$n" --workload deepfm_v2_c2 --scaling $sc --collective $coll
      run "c3_${sc}_${coll}_n$n" "$n" --workload din_c3 --scaling $sc --collective $coll
    done
  done
  run "c4_replicated_n$n" "$n" --workload deepfm_c4 --scaling weak --collective torch
  run "c4_sharded_n$n" "$n" --workload deepfm_c4 --scaling weak --collective torch --shard-tables
  run "c5_weak_n$n" "$n" --workload widedeep_c5 --scaling weak --collective torch
done
python - "$O" <<'PY'
import glob, json, os, re, sys
O = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(O, "*_n*.json"))):
    tag = os.path.basename(f)[:-5]
    m = re.match(r"(.*)_n(\d+)$", tag)
    try:
        l = json.loads(open(f).read())
    except Exception:
        continue
    rows.setdefault(m.group(1), {})[int(m.group(2))] = l
table = {}
md = ["| run | " + " | ".join("N=%d samples/s (eff.)" % n for n in (1, 2, 4, 8)) + " |", "|---|---|---|---|---|"]
for tag, by_n in sorted(rows.items()):
    base = by_n.get(1) or rows.get(re.sub(r"_(sprk|peer)$", "_torch", tag), {}).get(1)
    cells = []
    for n in (1, 2, 4, 8):
        l = by_n.get(n)
        if not l:
            cells.append("-")
            continue
        # weak: value = all ranks' samples/s, strong: global rows/s -- N times the one-GPU figure is the ideal of either
        eff = l["value"] / base["value"] / n if base else None
        table.setdefault(tag, {})[n] = {"value": l["value"], "ms_per_step": l["ms_per_step"], "efficiency_vs_n1": eff, "scaling": l.get("scaling")}
        cells.append("%.4g (%s)" % (l["value"], "%.0f %%" % (100 * eff) if eff is not None else "-"))
    md.append("| %s | %s |" % (tag, " | ".join(cells)))
json.dump(table, open(os.path.join(O, "scale_table.json"), "w"), indent=1, sort_keys=True)
open(os.path.join(O, "scale_table.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
PY
