#!/usr/bin/env python3
"""Developer tool: how much of a bi-interaction launch hides under an aggregation launch when the two run on
different HIP streams (independent buffers) - the measurement behind the "halves of a layer on two streams" idea
of DESIGN 7.  Prints sequential vs concurrent time of the pair on the amazon-book graph."""
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
E = len(trip)
for D, DO in ((64, 64), (32, 16)):
    X = torch.randn(n, D, device=dev)
    w = torch.rand(E, device=dev)
    out = torch.empty(n, D, device=dev)
    ws = ops.spmm_workspace(E, D, dev)
    P = torch.randn(n, D, device=dev)
    W2 = torch.randn(DO, D, device=dev) / 8
    h = torch.empty(n, DO, device=dev)
    wide = torch.empty(n, 176, device=dev)
    s1 = torch.cuda.Stream()

    def spmm():
        ops.spmm(indptr, col, row_of, X, w, out=out, mul_self=True, workspace=ws)

    def bi():
        ops.bi_interaction(P, W2, 0.01, h_out=h, norm_out=wide[:, 64:64 + DO])

    def timed(fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        return float(np.median(ts))

    def both_seq():
        spmm()
        bi()

    def both_par():
        e = torch.cuda.Event()
        e.record()
        with torch.cuda.stream(s1):
            s1.wait_event(e)
            bi()
            f = torch.cuda.Event()
            f.record()
        spmm()
        torch.cuda.current_stream().wait_event(f)

    t_s, t_b = timed(spmm), timed(bi)
    print("D=%d -> %d: aggregation %.1f us, bi-interaction %.1f us, one after the other %.1f us, on two streams %.1f us"
          % (D, DO, t_s, t_b, timed(both_seq), timed(both_par)))
