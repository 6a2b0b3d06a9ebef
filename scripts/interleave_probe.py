#!/usr/bin/env python3
"""Developer probe: the fused attention launch timed alone in a loop and inside the full step
(interleaved with softmax / SpMM / bi-interaction), same process, same graph."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgl_kgat_amd as K  # noqa: E402
from dgl_kgat_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg()
torch.manual_seed(1234)
model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
g = synth.build_graph(n, trip, dev)


def att():
    with torch.no_grad():
        return model.compute_attention(g)


def step():
    with torch.no_grad():
        a = model.compute_attention(g)
        g.edata["w"] = a
        return model.gnn(g)


def timed(fn, reps):
    with ops.KernelTimer() as kt:
        for _ in range(reps):
            fn()
    torch.cuda.synchronize()
    return {k: float(np.median([ms for _, ms in v])) for k, v in kt.summary().items()}


step(); step()
print("attention alone      :", timed(att, 50))
print("full step            :", timed(step, 50))
print("attention alone again:", timed(att, 50))
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def att_cold():
    flush.fill_(1.0)
    return att()


print("attention, caches flushed before each:", timed(att_cold, 20))


def step_sync():
    with torch.no_grad():
        a = model.compute_attention(g)
        torch.cuda.synchronize()
        g.edata["w"] = a
        out = model.gnn(g)
        torch.cuda.synchronize()
        return out


print("full step, host sync around attention:", timed(step_sync, 30))

# which of the attention's inputs does the step evict?  Touch them (a read pass) before the launch.
groups = g._st.rel_groups(g.edata["type"], R)
tiles = groups.g_tab["tiles"]
idx_arrays = [groups.src_g, groups.gid, groups.pos_g, groups.g_node, tiles[0]]
ent = model.entity_embed.weight


def step_touch(which):
    with torch.no_grad():
        out = model.gnn(g)
        if which in ("indices", "both"):
            for t in idx_arrays:
                t.sum()
        if which in ("table", "both"):
            ent.sum()
        a = model.compute_attention(g)
        g.edata["w"] = a
        return out


for which in ("none", "indices", "table", "both"):
    step_touch(which)
    print("gnn, touch %-8s, attention:" % which, {k: round(v, 4) for k, v in timed(lambda: step_touch(which), 30).items() if k == "att_score"})
