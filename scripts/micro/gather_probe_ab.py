#!/usr/bin/env python3
"""Developer tool: kgat_gather_probe_f32 (D = 64, the benchmark graph's own col array) at different numbers of rows in
flight: U rows per lane group (-DKGAT_PROBE_U) x resident workgroups per CU (-DKGAT_PROBE_LDS ballast)."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

base = _lib.load()
variants = {"shipped (U=8, no ballast)": base}
for vi, flags in enumerate(f for f in os.environ.get("AB_FLAGS", "").split(";") if f.strip()):
    so = "/tmp/libkgat_hip_gp%d.so" % vi
    tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
    objs, procs = [], []
    for src_, extra in _lib.SOURCES.items():
        obj = "/tmp/gp%d_%s.o" % (vi, src_.replace(".hip", ""))
        objs.append(obj)
        if src_ != "kgat_spmm.hip" and vi > 0:
            objs[-1] = "/tmp/gp0_%s.o" % src_.replace(".hip", "")
            continue
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
indptr, col, eid, row_of = ops.csr_from_coo(n, torch.as_tensor(trip[:, 2].copy(), device=dev),
                                            torch.as_tensor(trip[:, 0].copy(), device=dev))
X = torch.randn(n, 64, device=dev)
sink = ops.gather_probe(col, X)
for name, lib in variants.items():
    _lib._lib = lib
    ts = []
    for _ in range(60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gather_probe(col, X, sink); b.record(); ts.append((a, b))
    torch.cuda.synchronize()
    t = np.median([a.elapsed_time(b) for a, b in ts][10:])
    print("%-50s %.1f us  %.2f TB/s of gathered rows" % (name, t * 1e3, col.numel() * 256 / (t * 1e-3) / 1e12))
