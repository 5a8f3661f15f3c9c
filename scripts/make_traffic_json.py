"""profiles/traffic.json from a round's PMC summary (gpurun_out/r02/pmc_summary.json written by scripts/gpu_r2_prof.sh):
read = 2 x FETCH_SIZE x 1024 (the calibration of scripts/pmc_calib.py), write = WRITE_SIZE x 1024, per launch of each workload's
dominant kernel.    python scripts/make_traffic_json.py profiles/r02/pmc_summary.json r02"""
import json
import sys

src, rnd = sys.argv[1], sys.argv[2]
p = json.load(open(src))
out = json.load(open("profiles/traffic.json"))
MAP = {"deepfm_v2_c2": ("c2", "k_deepfm_v2_joint", 65536), "deepfm_v2_c2_hbm_resident": ("c2hbm", "k_deepfm_v2_joint", 65536),
       "deepfm_c2": ("pairs", "k_deepfm_pairs", 65536), "din_c3": ("c3", "k_din_attn", 32768), "deepfm_c4": ("c4pairs", "k_deepfm_pairs", 65536),
       "widedeep_c5": ("c5", "k_mlp_rows", 131072), "deepfm_v2_ref": ("v2ref", "k_rows_chain", 65536)}
if rnd >= "r06":       # round 6's tags (scripts/r06/20_profiles.sh) and the kernels those workloads dispatch since rounds 3-5
    MAP = {"deepfm_v2_c2": ("c2", "k_deepfm_v2_joint1", 65536), "deepfm_v2_c2_hbm_resident": ("c2_hbm", "k_deepfm_v2_joint1", 65536),
           "deepfm_c2": ("c2_pairs", "k_deepfm_pairs1", 65536), "din_c3": ("c3", "k_din_fused<2, false, true, false, 0>", 32768),
           "deepfm_c4": ("c4_pairs", "k_deepfm_pairs", 65536), "widedeep_c5": ("c5", "k_mlp_rows", 131072), "deepfm_v2_ref": ("v2_ref", "k_rows_chain1", 65536)}
for wl, (tag, kern, batch) in MAP.items():
    try:
        rd = int(2 * p["pmc_%s_fetch" % tag][kern]["FETCH_SIZE"] * 1024)
        wr = int(p["pmc_%s_write" % tag][kern]["WRITE_SIZE"] * 1024)
    except KeyError as e:
        print("skip", wl, e)
        continue
    keep = {k: v for k, v in out.get(wl, {}).items() if k in ("kernel", "kernel_launched", "note")}      # (bench.py matches on "kernel": the family name)
    out[wl] = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "batch": batch, "kernel": kern, "round": rnd}
    if keep:
        out[wl].update({"kernel": keep.get("kernel", kern), "kernel_launched": kern if kern != keep.get("kernel") else keep.get("kernel_launched", kern)})
        if "note" in keep: out[wl]["note"] = keep["note"]
    print(wl, out[wl])
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
