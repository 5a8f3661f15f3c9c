"""The committed evidence is self-consistent: every fraction of profiles/r06/roofline_table.json (the table DESIGN.md section 7.2 quotes)
follows from the rocprofv3 csv next to it -- algorithmic bytes per launch / the kernel's average duration / 8.0e12 --, the bench lines
kept beside the csvs describe the same kernel and batch, the PMC columns follow from pmc_summary.json, and the driver's line
(bench_driver_command.json) carries the blocks VERDICT r03 asked for with the bound of its 64-batch region labelled as what it is."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r06")
HBM_PEAK = 8.0e12


def _table():
    return json.load(open(os.path.join(P, "roofline_table.json")))


def _kernel_rows(path):
    with open(path, newline="") as f:
        return [(r["Name"], int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(f)]


@pytest.mark.parametrize("row", _table(), ids=lambda r: r["workload"])
def test_fraction_follows_from_the_kept_trace(row):
    w = row["workload"].replace("c3_attn", "c3").replace("dien_seq", "dien_ref")       # (config 3's two kernels share one trace; so do DIEN's)
    rows = _kernel_rows(os.path.join(P, w + "_strict_kernel_stats.csv"))
    hit = [r for r in rows if "::" + row["kernel"] in r[0].replace("(anonymous namespace)::", "::", 1).replace("sprk_dev::", "::", 1)]
    assert len(hit) == 1, (row["kernel"], [r[0][:60] for r in rows[:4]])
    name, calls, avg_ns = hit[0]
    assert calls == row["launches"] and abs(avg_ns / 1e3 - row["rocprof_avg_us"]) < 1e-6
    alg = row["algorithmic_bytes_per_sample"] * row["batch"]
    assert abs(alg / (avg_ns * 1e-9) / HBM_PEAK - row["frac"]) < 1e-9
    assert abs(alg / 1e6 - row["algorithmic_mb"]) < 1e-6
    # the untraced twin of the same command: same workload and batch, its HIP-event time is the table's
    b = json.load(open(os.path.join(P, "bench_" + w + "_strict.json")))
    assert b["config"]["workload"].split(":")[0] == row["bench_workload"] and b["config"]["batch_per_gpu"] == row["batch"]
    rl = b["roofline"]["attention_only"] if row["workload"] == "c3_attn" else b["roofline"]     # (round 5: `roofline` IS the fused launch)
    if row["workload"] == "dien_seq":
        rl = b["roofline"]["sequence_only"]
    assert abs(rl["avg_launch_us"] - row["hip_event_us"]) < 1e-6
    # HIP events and the tracer agree on every kernel of 7 us and more (the tracer inflates shorter ones)
    if row["rocprof_avg_us"] >= 7.0 and row["hip_event_us"] >= 7.0:
        assert 0.94 <= row["events_vs_rocprof"] <= 1.06, row["events_vs_rocprof"]


def test_pmc_columns_follow_from_the_summary():
    pm = json.load(open(os.path.join(P, "pmc_summary.json")))
    seen = 0
    for row in _table():
        if "pmc_traffic_mb" not in row:
            continue
        w = row["workload"].replace("c3_attn", "c3")
        k = row["kernel"] if row["kernel"] in pm["pmc_%s_fetch" % w] else row["kernel"].split("<")[0]
        t = 2 * pm["pmc_%s_fetch" % w][k]["FETCH_SIZE"] * 1024 + pm["pmc_%s_write" % w][k]["WRITE_SIZE"] * 1024
        assert abs(t / 1e6 - row["pmc_traffic_mb"]) < 1e-6
        seen += 1
        if "valu_per_sample" in row:
            assert abs(pm["pmc_%s_sq2" % w][k]["SQ_INSTS_VALU"] / row["batch"] - row["valu_per_sample"]) < 1e-9
    assert seen >= 8
    # config 3: the one-launch kernel and its attention-only form are reported apart
    assert {"k_din_fused<2, false, true, false, 0>", "k_din_fused<2, false, false, false, 0>"} <= set(pm["pmc_c3_sq2"])


def test_driver_line_blocks():
    l = json.load(open(os.path.join(P, "bench_driver_command.json")))
    assert l["metric"] == "ctr_samples_per_sec" and l["n_gpus"] == 1 and l["steps"] == 20 and l["warmup"] == 5
    assert l["roofline_timed_region"]["bound"] == "infinity_cache" and "NOT an HBM utilisation" in l["roofline_timed_region"]["frac_is"]
    assert set(l["workloads"]) >= {"din_c3", "deepfm_c2", "deepfm_c4", "widedeep_c5", "neuralcf_serving", "predict_csv"}
    assert l["workloads"]["neuralcf_serving"]["latency_ms"]["p50"] < 0.5
    assert l["roofline"]["bound"] == "hbm" and abs(l["roofline"]["achieved"] / l["roofline"]["peak"] - l["roofline"]["frac"]) < 1e-9
    assert l["cpu_baseline"]["kind"] == "port" and l["cpu_baseline"]["cores"] >= 1
    # value = samples per second of the timed region; the one-batch figure is the strict roofline's
    assert abs(l["value"] * l["ms_per_step"] * 1e-3 / l["config"]["batch_per_gpu"] - 1.0) < 1e-6
    assert l["workloads"]["din_c3"]["roofline"]["kernel"].startswith("k_din_fused<TAIL>")          # the PRODUCT kernel, the attention-only loop a sub-field
    assert l["workloads"]["din_c3"]["roofline"]["attention_only"]["avg_launch_us"] < l["workloads"]["din_c3"]["roofline"]["avg_launch_us"]
    # [r5] the scalars of the blocks the driver's record drops, repeated inside the two it keeps
    rl, cf = l["roofline"], l["config"]
    assert abs(rl["hbm_resident_frac"] - l["roofline_hbm_resident"]["frac"]) < 1e-12 and rl["strict_samples_per_s"] > 8e9
    assert abs(rl["zipf_strict_us"] - l["roofline_variants"]["zipf"]["avg_launch_us"]) < 1e-9 and rl["zipf_oracle_err"] <= 1e-4
    assert rl["f32_mfma_strict_us"] > rl["avg_launch_us"] and rl["f32_mfma_oracle_err"] <= 1e-4
    assert "f32" in l["roofline_variants"]["f32_mfma"]["kernel"] and "split-f16" in l["roofline_variants"]["zipf"]["kernel"]
    for w in ("din_c3", "deepfm_c2", "deepfm_c4", "widedeep_c5"):
        assert abs(cf[w + "_strict_us"] - l["workloads"][w]["roofline"]["avg_launch_us"]) < 1e-9 and 0 < cf[w + "_strict_frac"] < 1
    assert cf["neuralcf_serving_p50_ms"] < 0.5 and cf["neuralcf_serving_requests_per_s_12_workers_8_clients"] > 8000
    # [r6, VERDICT r05 item 2 / weak 7] the steady-state HBM figure and configs 4 / 5's working sets in kept keys; the working sets are 4 x the
    # Infinity Cache, and nothing labelled an HBM rate exceeds what HBM delivers (6.3 TB/s)
    assert abs(rl["hbm_resident_frac_16_batches"] - l["roofline_hbm_resident"]["frac_16_batches_per_launch"]) < 1e-12
    assert rl["hbm_resident_working_set_mb"] > 512 and rl["hbm_resident_frac_16_batches"] * 8.0 <= 6.3
    for w in ("deepfm_c4", "widedeep_c5"):
        blk = l["workloads"][w]
        assert cf[w + "_working_set_mb"] >= 4 * 256 and cf[w + "_input_batches_cycled"] == blk["input_batches_cycled"] > 8
        assert blk["roofline"]["hbm_side_GBps"] <= 6300 and blk["roofline"]["hbm_side_GBps_many"] <= 6300
    assert cf["predict_csv_rows_per_s"] > 1e7 and l["workloads"]["predict_csv"]["repeats_equal_first_block"]


def test_rows_that_cycle_a_working_set_say_so():
    """[r6] configs 4 / 5 of the table: the strict loop walked a working set of at least 4 x the Infinity Cache, and the row says how large."""
    rows = {r["workload"]: r for r in _table()}
    for w in ("c4_v2", "c4_pairs", "c5"):
        assert rows[w]["working_set_mb"] >= 1024 and rows[w]["input_batches_cycled"] > 8, rows[w]
    # the small reference shapes carry their several-batches-per-launch figure next to the strict one
    for w in ("ncf_ref", "deepfm_ref", "v2_ref"):
        assert rows[w]["many_us_per_step"] < rows[w]["rocprof_avg_us"] and rows[w]["many_batches_per_launch"] >= 16
