"""PCIe-inclusive rate of the headline workload: packed ids / dense start in PINNED HOST memory,
scores end in pinned host memory.  Never bench.py's `value` (that is HBM-resident); this is the number a caller that
hands over host buffers sees.  Two variants: one stream (copy in, forward, copy out, serial) and two streams
(batch i+1's upload overlaps batch i's forward / download).

    python scripts/bench_pcie.py [--workload deepfm_v2_c2] [--steps 200]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="deepfm_v2_c2")
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    import torch
    import bench
    B = {"din_c3": 32768}.get(a.workload, 65536)
    model, feats, desc, roof = bench.build_workload(a.workload, B, "uniform")
    eng = model.engine
    host = []
    for f in feats:
        ids, dense = model.pack(f)
        host.append((torch.from_numpy(ids).pin_memory(), torch.from_numpy(dense).pin_memory()))
    NB = len(host)
    bytes_in = host[0][0].numel() * 4 + host[0][1].numel() * 4
    res = {"workload": a.workload, "batch": B, "bytes_in_per_batch": bytes_in, "bytes_out_per_batch": B * 4}
    for nstreams in (1, 2):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        dev = [(torch.empty_like(host[0][0], device="cuda"), torch.empty_like(host[0][1], device="cuda"),
                torch.empty(B, dtype=torch.float32, device="cuda"), torch.empty(B, dtype=torch.float32).pin_memory(),
                torch.empty(max(eng.workspace_bytes(B) // 4, 1), dtype=torch.float32, device="cuda")) for _ in range(nstreams)]

        def run(k):
            for i in range(k):
                s = streams[i % nstreams]
                d_ids, d_dense, d_out, h_out, ws = dev[i % nstreams]
                with torch.cuda.stream(s):
                    d_ids.copy_(host[i % NB][0], non_blocking=True)
                    d_dense.copy_(host[i % NB][1], non_blocking=True)
                    eng.forward(d_ids, d_dense, d_out, ws, s.cuda_stream)
                    h_out.copy_(d_out, non_blocking=True)
        run(20)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        run(a.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["streams_%d" % nstreams] = {"us_per_batch": dt / a.steps * 1e6, "samples_per_sec": B * a.steps / dt,
                                        "host_to_device_GBps": bytes_in * a.steps / dt / 1e9}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
