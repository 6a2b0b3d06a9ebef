#!/usr/bin/env python3
"""Developer tool: the whole benchmark step (compute_attention + edge_softmax + 3 x (aggregation + bi-interaction),
amazon-book-shaped CKG, d = 64) on the shipped library against builds with extra compiler flags
(AB_FLAGS="-DA=1;-DB=2 -DC=3": one variant per ';'), alternating the builds block by block on one box.
Prints the median / minimum time per step of each build and whether the step's output has the shipped build's bits."""
import gc
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dgl_kgat_amd as K  # noqa: E402
from dgl_kgat_amd import _lib, synth  # noqa: E402

base = _lib.load()
variants = {"shipped": base}
for vi, flags in enumerate(f for f in os.environ.get("AB_FLAGS", "").split(";") if f.strip()):
    so = "/tmp/libkgat_hip_sv%d.so" % vi
    tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
    objs, procs = [], []
    for src_, extra in _lib.SOURCES.items():
        obj = "/tmp/sv%d_%s.o" % (vi, src_.replace(".hip", ""))
        objs.append(obj)
        procs.append(subprocess.Popen([_lib._hipcc()] + _lib.BASE_FLAGS + extra + flags.split() + [tag, "-c",
                                       os.path.join(_lib.CSRC, src_), "-o", obj]))
    for p_ in procs:
        assert p_.wait() == 0
    subprocess.check_call([_lib._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so] + objs)
    _lib.SO_PATH, _lib._lib = so, None
    variants[flags.strip()] = _lib.load()
_lib._lib = base

dev = torch.device("cuda:0")
if os.environ.get("AB_EAGER", "") in ("", "0"):   # AB_EAGER=1: the library default (edge-id-ordered weights written every step)
    K.enable_lazy_edge_weights()
n, trip, n_rel = synth.amazon_book_ckg()
g = synth.build_graph(n, trip, device=dev)
E = g.number_of_edges()
D = int(os.environ.get("AB_DIM", "64"))
torch.manual_seed(1234)
model = K.KGATPropagation(n, n_rel, input_node_dim=D, relation_dim=D, num_gnn_layers=3, n_hidden=D, dropout=0.0).to(dev)


def step():
    with torch.no_grad():
        a = model.compute_attention(g)
        g.edata["w"] = a
        return model.gnn(g)


ref = step().clone()
torch.cuda.synchronize()
gc.collect()
gc.freeze()
STEPS, ROUNDS = int(os.environ.get("AB_STEPS", "100")), int(os.environ.get("AB_ROUNDS", "7"))
times = {k: [] for k in variants}
same = {}
for r in range(ROUNDS + 1):
    for name, lib in variants.items():
        _lib._lib = lib
        for _ in range(10):
            out = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            out = step()
        torch.cuda.synchronize()
        if r > 0:
            times[name].append((time.perf_counter() - t0) / STEPS * 1e3)
        same[name] = bool(torch.equal(out, ref))
_lib._lib = base
for name, t in times.items():
    print("%-60s median %.4f  min %.4f ms per step (%d rounds x %d steps) | shipped build's bits: %s" % (
        name, float(np.median(t)), min(t), ROUNDS, STEPS, same[name]))
