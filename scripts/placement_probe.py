#!/usr/bin/env python3
"""Developer probe: does WHERE / HOW the gathered table was allocated change the HBM-resident SpMM?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, e = 10_000_000, 200_000_000
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
early = None
if mode == "early":           # the table is the process's first device allocation
    early = torch.randn((n, 64), device=dev)
if mode == "reserve":         # an early 12 GB segment, returned to torch's cache: later tensors are carved from it
    r = torch.empty(12 << 30, dtype=torch.uint8, device=dev)
    del r
src, dst, _ = synth.power_law_coo_device(n, e, 64, dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
del src, dst, eid
w = torch.rand(e, device=dev)
out = torch.empty((n, 64), device=dev)
ws = ops.spmm_workspace(e, 64, dev)


def t(X, label):
    for _ in range(3):
        ops.spmm(indptr, col, row_of, X, w, out=out, workspace=ws)
    ev = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.spmm(indptr, col, row_of, X, w, out=out, workspace=ws)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    print("%-46s ptr %x : median %.3f ms" % (label, X.data_ptr(), np.median([a.elapsed_time(b) for a, b in ev])))


if early is not None:
    t(early, "first allocation of the process (kernel-filled)")
Xk = torch.randn((n, 64), device=dev)
t(Xk, "late, torch.randn on the device")
Xh = torch.randn((n, 64)).to(dev)
t(Xh, "late, filled by a host-to-device copy")
Xc = torch.empty_like(Xk)
Xc.copy_(Xk)
t(Xc, "late, device-to-device copy")
torch.cuda.empty_cache()
Xf = torch.randn((n, 64), device=dev)
t(Xf, "late, after empty_cache (fresh hipMalloc)")
big = torch.empty((4 * n, 64), device=dev)
Xs = big[n:2 * n]
Xs.normal_()
t(Xs, "late, slice of a 10 GB allocation")
print(torch.cuda.memory_summary(abbreviated=True)[:0])
