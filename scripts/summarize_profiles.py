"""One line per bench_*.json of a profiles directory: value, ms per step, the strict one-batch launch and its roofline fraction.
    python scripts/summarize_profiles.py profiles/r02"""
import glob
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02"
for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
        continue
    if "value" not in j:
        print("%-46s %s" % (os.path.basename(f)[6:-5], json.dumps(j)[:150]))
        continue
    r = j.get("roofline") or {}
    hb = j.get("roofline_hbm_resident") or {}
    print("%-46s %8.3f G/s  %8.3f us/step  strict %7.3f us  frac %.3f  two-streams %s  hbm-resident %s / %s"
          % (os.path.basename(f)[6:-5], j["value"] / 1e9, j["ms_per_step"] * 1e3, r.get("avg_launch_us") or 0, r.get("frac") or 0,
             ("%.2f G" % (j["value_one_batch_per_launch_two_streams"] / 1e9)) if j.get("value_one_batch_per_launch_two_streams") else "-",
             ("%.2f us" % hb["avg_launch_us"]) if hb.get("avg_launch_us") else "-", ("%.3f" % hb["frac"]) if hb.get("frac") else "-"))
