#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: parity tests, smoke, bench A/B lines, rocprof kernel trace.
set -u
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt 2>&1
echo "=== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "=== bench"
b() { out=$1; shift; timeout 600 env "$@" 2>&1 | tail -1 | tee gpurun_out/$out.json | cut -c1-200; }
b bench_c2 python bench.py
b bench_c2_lb32 python bench.py --cpu-seconds 0 --launch-batches 32
b bench_c2_lb64 python bench.py --cpu-seconds 0 --launch-batches 64
b bench_c2_lb1 python bench.py --cpu-seconds 0 --launch-batches 1
b bench_c2_joint_f32 SPRK_V2_HALF=0 python bench.py --cpu-seconds 0 --launch-batches 1
b bench_c2_perfield SPRK_V2_JOINT=0 python bench.py --cpu-seconds 0
b bench_c2_unfolded SPRK_V2_FOLD=0 python bench.py --cpu-seconds 0
b bench_c2_interp SPRK_FORCE_INTERPRETER=1 python bench.py --cpu-seconds 0
b bench_c2_zipf python bench.py --cpu-seconds 0 --dist zipf --launch-batches 1
b bench_c2_zipf_lb16 python bench.py --cpu-seconds 0 --dist zipf
b bench_c2_b1m python bench.py --cpu-seconds 0 --batch 1048576 --steps 400 --warmup 40 --launch-batches 1
b bench_c2_strict python bench.py --cpu-seconds 0 --overlap-streams 0 --launch-batches 1
b bench_c2_pairs python bench.py --workload deepfm_c2 --cpu-seconds 0
b bench_c2_pairs_lb1 python bench.py --workload deepfm_c2 --cpu-seconds 0 --launch-batches 1
b bench_c2_pairs_interp SPRK_V1_CHAIN=0 python bench.py --workload deepfm_c2 --cpu-seconds 0
b bench_c3 python bench.py --steps 320 --warmup 32 --workload din_c3 --cpu-seconds 6
b bench_c3_lb1 python bench.py --steps 320 --warmup 32 --workload din_c3 --cpu-seconds 0 --launch-batches 1
b bench_c3_f32 SPRK_DIN_HALF=0 python bench.py --steps 300 --warmup 30 --workload din_c3 --cpu-seconds 0
b bench_c3_interp_tail SPRK_DIN_TAIL=0 python bench.py --steps 300 --warmup 30 --workload din_c3 --cpu-seconds 0
b bench_c3_legacy SPRK_DIN_LEGACY=1 SPRK_DIN_TAIL=0 SPRK_TILE_FOLD=0 python bench.py --steps 100 --warmup 10 --workload din_c3 --cpu-seconds 0
b bench_c5 python bench.py --workload widedeep_c5 --steps 200 --warmup 20 --cpu-seconds 0
b bench_c5_interp SPRK_MLP_CHAIN=0 python bench.py --workload widedeep_c5 --steps 200 --warmup 20 --cpu-seconds 0
b bench_c2_f32_hidden SPRK_DYN_F16=0 python bench.py --workload deepfm_c2 --cpu-seconds 0
b bench_c3_f32_hidden SPRK_DYN_F16=0 python bench.py --steps 300 --warmup 30 --workload din_c3 --cpu-seconds 0
b bench_c2_forced_collective SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0
b bench_emb_rank python scripts/bench_emb_rank.py
b bench_emb_rank_scores_only python scripts/bench_emb_rank.py --no-rank --cpu-seconds 1
echo "=== DIEN timing"
timeout 300 python - <<'PY' 2>&1 | grep DIEN | tee gpurun_out/dien_time.log
import torch
from sparrowrecsys_amd import models as M, synthetic as SY
for T in (5, 50):
    B = 32768
    f = SY.synth_din(B, T, 1001, 30001, seed=1)
    m = M.DIEN(seed=2, emb_dim=10, hist_len=T)
    ids, dense = m.pack(f)
    ids, dense = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    eng = m.engine
    aux = torch.empty((B, eng.n_aux), device="cuda")
    for name, fn in (("stage", lambda: eng.din_pool(ids, aux, None)), ("forward", lambda: m.predict_device(ids, dense))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print("DIEN T=%d B=%d %s: %.1f us" % (T, B, name, e0.elapsed_time(e1) * 20))
PY
echo "=== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -o c2 -- python $R/bench.py --steps 400 --warmup 40 --cpu-seconds 0 --no-check > $R/gpurun_out/prof_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o c3 -- python $R/bench.py --steps 50 --warmup 5 --workload din_c3 --cpu-seconds 0 --no-check > $R/gpurun_out/prof_c3.log 2>&1
# the same commands with launches in strict stream order: per-kernel durations comparable with bench.py's roofline block
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2s -o c2_strict -- python $R/bench.py --steps 400 --warmup 40 --cpu-seconds 0 --no-check --overlap-streams 0 --launch-batches 1 > $R/gpurun_out/prof_c2s.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3s -o c3_strict -- python $R/bench.py --steps 50 --warmup 5 --workload din_c3 --cpu-seconds 0 --no-check --overlap-streams 0 > $R/gpurun_out/prof_c3s.log 2>&1
cd $R
tail -1 gpurun_out/prof_c2s.log | cut -c1-200; tail -1 gpurun_out/prof_c3s.log | cut -c1-200
for f in $(find gpurun_out/prof_c2 gpurun_out/prof_c3 gpurun_out/prof_c2s gpurun_out/prof_c3s -name "*kernel_stats.csv"); do echo "--- $f"; head -6 $f | cut -c1-200; done
