#!/usr/bin/env python3
"""Developer probe: partition.GraphedShardForward (HIP-graph replay of a rank's local launches) against the eager
step - unsharded graph and every rank of an 8-way partition of the benchmark graph, exchange stubbed; checks that
the replayed readout equals the eager one bit for bit."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dgl_kgat_amd as K  # noqa: E402
from dgl_kgat_amd import partition, synth  # noqa: E402

dev = torch.device("cuda:0")
K.enable_lazy_edge_weights()
n, trip, R = synth.amazon_book_ckg()
torch.manual_seed(1234)
model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
g = synth.build_graph(n, trip, dev)


def timeit(fn, k=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for P in (1, 8):
    for r in range(P):
        if P == 1:
            sg = g
        else:
            sg, _ = partition.shard_graph(g, r, P)
            sg.partition.exchange_enabled = False

        def step():
            with torch.no_grad():
                sg.edata["w"] = model.compute_attention(sg)
                return model.gnn(sg)
        ref = step().clone()
        gs = partition.GraphedShardForward(model, sg)
        out = gs()
        same = torch.equal(out, ref)
        print("P=%d rank %d: eager %.4f ms  graphed %.4f ms  same bits: %s" % (P, r, timeit(step), timeit(gs), same), flush=True)
        del gs
