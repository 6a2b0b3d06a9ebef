#!/usr/bin/env python3
"""Developer tool: what forming the aggregation's left rows costs the dense kernel - kgat_bi_interaction_mul_f32 on a
complete h_N against kgat_bi_interaction_mul_deferred_f32 on the h_N of a KGAT_SPMM_DEFER_FINISH aggregation (same
graph, same rows), launches alternated; and the aggregation with and without its second launch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg()
src = torch.as_tensor(trip[:, 2].copy(), device=dev)
dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
w = torch.rand(len(trip), device=dev)


def ev(fns, k=80):
    ts = {name: [] for name in fns}
    for it in range(k):
        for name, fn in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            ts[name].append((a, b))
    torch.cuda.synchronize()
    return {name: 1e3 * float(np.median([a.elapsed_time(b) for a, b in v][10:])) for name, v in ts.items()}


for d_in, d_out in ((64, 64), (64, 32), (32, 16)):
    X = torch.randn(n, d_in, device=dev)
    W2 = torch.randn(d_out, d_in, device=dev) / d_in ** 0.5
    wide = torch.empty((n, 176), device=dev)
    h = torch.empty((n, d_out), device=dev)
    nrm = wide[:, 64:64 + d_out]
    hn = ops.spmm(indptr, col, row_of, X, w)
    hn_d, left = ops.spmm(indptr, col, row_of, X, w, defer_finish=True)
    ws = left.workspace
    r = ev({"dense, complete h_N": lambda: ops.bi_interaction_mul(X, hn, W2, 0.01, h_out=h, norm_out=nrm),
            "dense, left rows formed on the way": lambda: ops.bi_interaction_mul(X, hn_d, W2, 0.01, h_out=h, norm_out=nrm, deferred=left),
            "aggregation, two launches": lambda: ops.spmm(indptr, col, row_of, X, w, out=hn, workspace=ws),
            "aggregation, second launch deferred": lambda: ops.spmm(indptr, col, row_of, X, w, out=hn_d, workspace=ws, defer_finish=True)})
    print("%d -> %d: " % (d_in, d_out) + " | ".join("%s %.1f us" % kv for kv in r.items()))
