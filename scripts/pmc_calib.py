#!/usr/bin/env python
"""Known-byte-count access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU
(MI355X_MICROARCH.md, HBM section: 'calibrate on a known byte count in your own access pattern').

Runs sprk_embedding_gather (k_embedding_gather) over a table much larger than the 256 MB Infinity Cache
with unique random row ids:
    pattern A: 128-byte rows (D = 32 floats): every id touches one whole 128-B line
    pattern B:  64-byte rows (D = 16 floats): every id touches half a line (two rows share a line)
Known traffic per launch: ids N*4 B + rows N*D*4 B read, N*D*4 B written.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sparrowrecsys_amd import _lib as L
    lib = L.load_library()
    N = 1 << 20
    for D, V in ((32, 1 << 23), (16, 1 << 24)):          # 1 GiB tables
        table = torch.empty((V, D), dtype=torch.float32, device="cuda").uniform_(-1, 1)
        g = torch.Generator(device="cuda").manual_seed(D)
        ids = torch.randperm(V, device="cuda", generator=g)[:N].to(torch.int32).contiguous()
        out = torch.empty((N, D), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            L.check(lib.sprk_embedding_gather(C.c_void_p(table.data_ptr()), V, D, D, C.c_void_p(ids.data_ptr()), N,
                                              C.c_void_p(out.data_ptr()), C.c_void_p(st)))
        torch.cuda.synchronize()
        print("calib D=%d: known read %d B (ids %d + rows %d), known write %d B per launch"
              % (D, N * 4 + N * D * 4, N * 4, N * D * 4, N * D * 4), flush=True)
        del table, out, ids


if __name__ == "__main__":
    main()
