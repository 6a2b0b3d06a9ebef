#!/usr/bin/env python3
"""Developer probe: what one rank of a P-way destination partition does locally per
step - the exchange stubbed out - timed on one GPU, rank by rank, with per-op HIP events."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgl_kgat_amd as K  # noqa: E402
from dgl_kgat_amd import ops, partition, synth  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg()
torch.manual_seed(1234)
model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
g = synth.build_graph(n, trip, dev)
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(world))
for r in only:
    sg, keep = partition.shard_graph(g, r, world)
    sg.partition.exchange_enabled = False

    def step():
        with torch.no_grad():
            sg.edata["w"] = model.compute_attention(sg)
            return model.gnn(sg)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    with ops.KernelTimer() as kt:
        for _ in range(20):
            step()
    torch.cuda.synchronize()
    parts = {k: (len(v) // 20, float(np.sum([ms for _, ms in v])) / 20) for k, v in kt.summary().items()}
    print("rank %d/%d rows %6d edges %7d: %.4f ms per step (exchange stubbed: world-size-1 group semantics) | %s"
          % (r, world, sg.partition.hi - sg.partition.lo, len(keep), dt * 1e3,
             "  ".join("%s x%d %.4f" % (k, c, ms) for k, (c, ms) in parts.items())))
