#!/usr/bin/env python3
"""Developer probe: what the h * h_N epilogue (KGAT_SPMM_MUL_SELF: a dependent X[v] load when a row ends) costs the
aggregation, and what the same product costs when the bi-interaction kernel forms it while loading its rows."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg()
indptr, col, eid, row_of = ops.csr_from_coo(n, torch.as_tensor(trip[:, 2].copy(), device=dev),
                                            torch.as_tensor(trip[:, 0].copy(), device=dev))
E = col.numel()
w = torch.rand(E, device=dev)


def ev(fn, k=60):
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        out.append((a, b))
    torch.cuda.synchronize()
    return 1e3 * float(np.median([a.elapsed_time(b) for a, b in out][10:]))


for d_in, d_out in ((64, 64), (64, 32), (32, 16)):
    X = torch.randn(n, d_in, device=dev)
    W2 = torch.randn(d_out, d_in, device=dev) / d_in ** 0.5
    ws = ops.spmm_workspace(E, d_in, dev)
    out = torch.empty((n, d_in), device=dev)
    wide = torch.empty((n, 176), device=dev)
    h = torch.empty((n, d_out), device=dev)
    t_plain = ev(lambda: ops.spmm(indptr, col, row_of, X, w, out=out, workspace=ws))
    t_mul = ev(lambda: ops.spmm(indptr, col, row_of, X, w, out=out, workspace=ws, mul_self=True))
    t_bi = ev(lambda: ops.bi_interaction(out, W2, 0.01, h_out=h, norm_out=wide[:, 64:64 + d_out]))
    t_bi_mul = ev(lambda: ops.bi_interaction_train(X, out, W2, 0.01, 0.0, 0, norm_out=wide[:, 64:64 + d_out]))
    print("%d -> %d: spmm plain %.1f us, with h*h_N epilogue %.1f us | bi-interaction %.1f us, forming h*h_N while loading "
          "(training kernel, p = 0) %.1f us" % (d_in, d_out, t_plain, t_mul, t_bi, t_bi_mul))
