#!/usr/bin/env python3
"""Developer tool: one KGATConv forward as two launches (kgat_spmm_umule_sum_f32 MUL_SELF + kgat_bi_interaction_f32)
against the one-launch form (kgat_spmm_bi_fused_f32), alternating, on the amazon-book-shaped CKG; plus the
distribution of rows per 1,024-edge tile (what the LDS row buffer of the fused form has to hold)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

# AB_FLAGS="-DKGAT_FUSED_CAP16=56;-DKGAT_FUSED_CAP16=56 -DKGAT_FUSED_HALF_RUNS=1": also time the fused launch of
# variant builds (each linked -Bsymbolic so that its kernels are its own)
import subprocess
base = _lib.load()
variants = {"shipped": base}
for vi, flags in enumerate(f for f in os.environ.get("AB_FLAGS", "").split(";") if f.strip()):
    so = "/tmp/libkgat_hip_fv%d.so" % vi
    tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
    objs, procs = [], []
    for src_, extra in _lib.SOURCES.items():
        obj = "/tmp/fv%d_%s.o" % (vi, src_.replace(".hip", ""))
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
n, trip, R = synth.amazon_book_ckg()
src = torch.as_tensor(trip[:, 2].copy(), device=dev)
dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
E = col.numel()
w = torch.rand(E, device=dev)
ro = row_of.cpu().numpy()
for te in (1024,):
    t0 = np.arange(0, E, te)
    t1 = np.minimum(t0 + te, E) - 1
    rows = ro[t1] - ro[t0] + 1
    print("rows per %d-edge tile: mean %.1f median %d p90 %d p99 %d max %d; tiles over 56 rows: %.1f %%, rows beyond slot 56: %.1f %% of all"
          % (te, rows.mean(), np.median(rows), np.percentile(rows, 90), np.percentile(rows, 99), rows.max(),
             100 * (rows > 56).mean(), 100 * np.maximum(rows - 56, 0).sum() / n))


def ev(fn, k=40):
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        out.append((a, b))
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in out])[5:]


for d_in, d_out in ((64, 64), (64, 32), (32, 16)):
    X = torch.randn(n, d_in, device=dev)
    W2 = torch.randn(d_out, d_in, device=dev) / d_in ** 0.5
    wide = torch.empty((n, 176), device=dev)
    ws = ops.spmm_workspace(E, d_in, dev)
    prod = torch.empty((n, d_in), device=dev)
    h = torch.empty((n, d_out), device=dev)
    ego = wide[:, :64] if (d_in == 64 and d_out == 64) else None   # layer 0 of the readout also copies its input rows
    # round 4's split: the plain operator, then the dense kernel forming h * h_N while it loads its rows
    two_a = lambda: ops.spmm(indptr, col, row_of, X, w, out=prod, workspace=ws)
    two_b = lambda: ops.bi_interaction_mul(X, prod, W2, 0.01, h_out=h, norm_out=wide[:, 64:64 + d_out], self_out=ego)
    one = lambda: ops.spmm_bi_fused(indptr, col, row_of, X, w, W2, 0.01, h_out=h, norm_out=wide[:, 64:64 + d_out],
                                    scratch=prod, workspace=ws, self_out=ego)
    res = {"spmm": [], "bi": [], "two": []}
    for rep in range(3):
        _lib._lib = base
        res["spmm"].append(np.median(ev(two_a)))
        res["bi"].append(np.median(ev(two_b)))
        res["two"].append(np.median(ev(lambda: (two_a(), two_b()))))
        for name, lib in variants.items():
            _lib._lib = lib
            res.setdefault("fused[%s]" % name, []).append(np.median(ev(one)))
    _lib._lib = base
    print("%d -> %d: " % (d_in, d_out) + "  ".join("%s %.1f us" % (k, 1e3 * min(v)) for k, v in res.items()))
