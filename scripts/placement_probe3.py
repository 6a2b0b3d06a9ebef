#!/usr/bin/env python3
"""Developer probe: for a series of fresh allocations of the gathered table, the HBM-resident SpMM
time (random 256-byte row gathers) next to a sequential read of the same bytes - does the slow mode
hit streaming too (physical channel mapping) or only random access (translation reach)?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, e = 10_000_000, 200_000_000
src, dst, _ = synth.power_law_coo_device(n, e, 64, dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
del src, dst, eid
w = torch.rand(e, device=dev)
out = torch.empty((n, 64), device=dev)
ws = ops.spmm_workspace(e, 64, dev)
# a gather with 4 KiB-local structure: sources grouped so that consecutive edges hit the same 2 MiB region
col_local = (torch.arange(e, device=dev, dtype=torch.int64) // 4096 * 7919 % (n // 8192) * 8192
             + torch.randint(0, 8192, (e,), device=dev)).clamp_(max=n - 1).to(torch.int32)


def med(fn, reps=6):
    for _ in range(2):
        fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


keep = []
for i in range(10):
    X = torch.empty((n, 64), device=dev)
    X.normal_()
    t_rand = med(lambda: ops.spmm(indptr, col, row_of, X, w, out=out, workspace=ws))
    t_local = med(lambda: ops.spmm(indptr, col_local, row_of, X, w, out=out, workspace=ws))
    t_seq = med(lambda: X.sum())
    t_copy = med(lambda: out.copy_(X))
    print("alloc %d ptr %x: spmm random rows %.3f ms | spmm with 2 MiB-local rows %.3f ms | sum(X) %.3f ms | copy %.3f ms"
          % (i, X.data_ptr(), t_rand, t_local, t_seq, t_copy))
    keep.append(X)
    if len(keep) > 3:
        keep.pop(0)
