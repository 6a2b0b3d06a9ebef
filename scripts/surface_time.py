"""The drop-in surface (the reference's own call sequence over DGLGraph) against the fused path, same GPU.

Round 5: every call is timed with its own pair of HIP events and the MEDIAN is reported next to the mean.  Round 4's
figure for `gnn` over the surface (6.2 ms; 4.2 ms on a later box) was one stall of 36-57 ms inside ten back-to-back
calls - the interpreter's generation-2 garbage collection, which walks every container alive in the process (the same
stall bench.py freezes the collector against) - spread over the ten: rocprofv3 --kernel-trace shows nine of ten calls
at 0.54 ms with no gap between their launches (profiles/r05_surface_trace_summary.txt)."""
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgl_kgat_amd as K
from dgl_kgat_amd import synth

dev = torch.device('cuda:0')
n, trip, R = synth.amazon_book_ckg()
torch.manual_seed(0)
m = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
g = synth.build_graph(n, trip, dev)


def timed(fn, reps):
    out = fn()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()          # as bench.py: later collections only look at objects created from here on
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) for a, b in ev])
    return out, float(np.median(ts)), float(ts.mean()), float(ts.max())


with torch.no_grad():
    for name, fn, reps in (("surface (reference call sequence: 41x filter_edges/apply_edges + edge_softmax)",
                            lambda: m.compute_attention_surface(g), 3),
                           ("fused (kgat_attention)", lambda: m.compute_attention(g), 10)):
        a, med, mean, mx = timed(fn, reps)
        print("%-90s median %.3f ms (mean %.3f, max %.3f)" % (name, med, mean, mx))
    a1, a2 = m.compute_attention_surface(g), m.compute_attention(g)
    print("max abs diff surface vs fused: %.3e" % float((a1 - a2).abs().max()))
    g.edata['w'] = a2
    for name, fused in (("gnn surface (update_all + torch dense)", False), ("gnn fused", True)):
        o, med, mean, mx = timed(lambda: m.gnn(g, fused=fused), 20)
        print("%-90s median %.3f ms (mean %.3f, max %.3f)" % (name, med, mean, mx))
