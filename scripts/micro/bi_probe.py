#!/usr/bin/env python3
"""Developer tool: where kgat_bi_interaction_f32's time goes - the launch with both outputs, with one of them, and
(for scale) torch's copy of the same bytes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops  # noqa: E402
import subprocess
base = _lib.load()
variants = {"shipped": base}
for vi, flags in enumerate(f for f in os.environ.get("AB_FLAGS", "").split(";") if f.strip()):
    so = "/tmp/libkgat_hip_bv%d.so" % vi
    tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
    objs, procs = [], []
    for src_, extra in _lib.SOURCES.items():
        obj = "/tmp/bv%d_%s.o" % (vi, src_.replace(".hip", ""))
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
n = 159251


def ev(fn, k=60):
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        out.append((a, b))
    torch.cuda.synchronize()
    return 1e3 * float(np.median([a.elapsed_time(b) for a, b in out][10:]))


for d_in, d_out in ((64, 64), (64, 32), (32, 16)):
    P = torch.randn(n, d_in, device=dev)
    W2 = torch.randn(d_out, d_in, device=dev) / d_in ** 0.5
    wide = torch.empty((n, 176), device=dev)
    h = torch.empty((n, d_out), device=dev)
    nrm = wide[:, 64:64 + d_out]
    dense = torch.empty((n, d_out), device=dev)
    HN = torch.randn(n, d_in, device=dev)
    ego = wide[:, :d_in] if d_in == 64 else None
    for name, lib in variants.items():
        _lib._lib = lib
        print("   %s: product formed on the way (kgat_bi_interaction_mul_f32%s) %.1f us%s" % (
            name, ", ego block copied" if ego is not None else "",
            ev(lambda: ops.bi_interaction_mul(P, HN, W2, 0.01, h_out=h, norm_out=nrm, self_out=ego)),
            "" if name == "shipped" else " | plain form, both outputs %.1f us" % ev(
                lambda: ops.bi_interaction(P, W2, 0.01, h_out=h, norm_out=nrm))))
    _lib._lib = base
    print("%d -> %d: both %.1f us | h only %.1f | norm (strided 176) only %.1f | norm (dense) only %.1f | torch copy of P %.1f | "
          "torch mm %.1f" % (
              d_in, d_out,
              ev(lambda: ops.bi_interaction(P, W2, 0.01, h_out=h, norm_out=nrm)),
              ev(lambda: ops.bi_interaction(P, W2, 0.01, h_out=h)),
              ev(lambda: ops.bi_interaction(P, W2, 0.01, norm_out=nrm, want_h=False)),
              ev(lambda: ops.bi_interaction(P, W2, 0.01, norm_out=dense, want_h=False)),
              ev(lambda: h.copy_(P[:, :d_out]) if d_out != d_in else h.copy_(P)),
              ev(lambda: torch.mm(P, W2.t(), out=h))))
