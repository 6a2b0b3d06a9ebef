#!/usr/bin/env python3
"""Developer probe: HBM-resident SpMM time against the size / alignment of the allocation the
gathered table is carved from (fresh allocations, one process)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, e = 10_000_000, 200_000_000
if len(sys.argv) > 1 and sys.argv[1] == "warm":   # some allocator history first, as in bench.py
    junk = [torch.randn(40_000_000, device=dev) for _ in range(12)]
    del junk
src, dst, _ = synth.power_law_coo_device(n, e, 64, dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
del src, dst, eid
w = torch.rand(e, device=dev)
out = torch.empty((n, 64), device=dev)
ws = ops.spmm_workspace(e, 64, dev)
row_bytes = 256
need = n * row_bytes


def tz(x):
    return (x & -x).bit_length() - 1


def t(X, label):
    for _ in range(3):
        ops.spmm(indptr, col, row_of, X, w, out=out, workspace=ws)
    ev = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.spmm(indptr, col, row_of, X, w, out=out, workspace=ws)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    print("%-44s ptr %x (2^%d aligned): median %.3f ms" % (label, X.data_ptr(), tz(X.data_ptr()),
                                                           np.median([a.elapsed_time(b) for a, b in ev])))


for gib in (0, 4, 8, 16, 32):
    for rep in range(2):
        size = need if gib == 0 else gib << 30
        arena = torch.empty(size, dtype=torch.uint8, device=dev)
        base = arena.data_ptr()
        # the table at the start of the arena, and at the first 1 GiB-aligned address inside it
        offs = [0]
        al = (-base) % (1 << 30)
        if al + need <= size and al:
            offs.append(al)
        for off in offs:
            X = arena[off:off + need].view(torch.float32).view(n, 64)
            X.normal_()
            t(X, "%s arena, offset %.2f GiB%s" % ("exact-size" if gib == 0 else "%d GiB" % gib, off / 2 ** 30,
                                                  " (1 GiB-aligned VA)" if off else ""))
        del arena, X
        torch.cuda.empty_cache()
