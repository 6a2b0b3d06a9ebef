#!/usr/bin/env python3
"""Temporary probe: why do long back-to-back loops of gnn / eager steps take 2-3 x the per-step time?"""
import gc, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgl_kgat_amd as K
from dgl_kgat_amd import synth
dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg()
torch.manual_seed(1234)
model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
g = synth.build_graph(n, trip, dev)

def step():
    with torch.no_grad():
        a = model.compute_attention(g)
        g.edata["w"] = a
        return model.gnn(g)

def eager():
    with torch.no_grad():
        a = g.kgat_attention(model.entity_embed.weight, model.W_R, model.relation_embed.weight, lazy=False)
        g.edata["w"] = a
        return model.gnn(g)

def gnn_only():
    with torch.no_grad():
        return model.gnn(g)

def run(name, fn, k):
    torch.cuda.synchronize()
    r0 = torch.cuda.memory_reserved()
    t0 = time.perf_counter()
    ts = []
    for _ in range(k):
        fn()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-10s k=%3d  %.4f ms/iter | host enqueue done at %.1f ms of %.1f | reserved %+.0f MB | gc counts %s" % (
        name, k, dt / k * 1e3, ts[-1] * 1e3, dt * 1e3, (torch.cuda.memory_reserved() - r0) / 1e6, gc.get_count()))

step(); step()
for rep in range(2):
    for k in (20, 50, 100):
        run("step", step, k)
        run("eager", eager, k)
        step()
        run("gnn_only", gnn_only, k)
gc.disable()
print("gc disabled")
for k in (50, 100):
    run("step", step, k)
    run("eager", eager, k)
    step()
    run("gnn_only", gnn_only, k)
