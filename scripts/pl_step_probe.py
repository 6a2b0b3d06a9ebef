#!/usr/bin/env python3
"""Developer probe: per-launch times of the D = 64 SpMM inside the step on the 10 M / 200 M graph,
then the very same launches (same tensors) repeated back to back."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgl_kgat_amd as K  # noqa: E402
from dgl_kgat_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, e = 10_000_000, 200_000_000
src, dst, et = synth.power_law_coo_device(n, e, 64, dev)
torch.manual_seed(1234)
model = K.KGATPropagation(n, 64, 64, 64, 3, 64, dropout=0.0).to(dev)
g = K.DGLGraph()
g.add_nodes(n)
g.add_edges(src.cpu().numpy(), dst.cpu().numpy())
g.readonly()
g.ndata["id"] = torch.arange(n, device=dev)
g.edata["type"] = et.long()
del src, dst


def step():
    with torch.no_grad():
        a = model.compute_attention(g)
        g.edata["w"] = a
        return model.gnn(g)


step()
with ops.KernelTimer() as kt:
    for _ in range(3):
        step()
torch.cuda.synchronize()
print("in step :", ["%.3f (D=%d)" % (ms, info[2]) for info, ms in kt.summary()["spmm"]])
csr = g._st.csr(dev)
w_csr = g._st.csr_weights(g.edata["w"])
X = model.entity_embed.weight.detach()
out = torch.empty((n, 64), device=dev)
ws = ops.spmm_workspace(e, 64, dev)
for mul in (True, False):
    ev = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.spmm(csr.indptr, csr.col, csr.row_of, X, w_csr, out=out, mul_self=mul, workspace=ws)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    print("back to back, same tensors, mul_self=%d:" % mul, ["%.3f" % a.elapsed_time(b) for a, b in ev])
X2 = torch.randn((n, 64), device=dev)
ev = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.spmm(csr.indptr, csr.col, csr.row_of, X2, w_csr, out=out, mul_self=True, workspace=ws)
    b.record()
    ev.append((a, b))
torch.cuda.synchronize()
print("back to back, freshly allocated X, mul_self=1:", ["%.3f" % a.elapsed_time(b) for a, b in ev])
print("X ptr %x  X2 ptr %x  out ptr %x" % (X.data_ptr(), X2.data_ptr(), out.data_ptr()))
