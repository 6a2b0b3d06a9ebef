#!/usr/bin/env python3
"""Developer tool: build-time variants of the merge SpMM (-D flags of csrc/kgat_spmm.hip) on the
HBM-resident 10 M / 200 M graph, alternating launch by launch on ONE allocation of the gathered
table (so the placement mode of NOTEBOOK.md 3.1 is common to all arms), plus the cache-served
amazon-book graph.  Only kgat_spmm.hip is recompiled per variant.

  python scripts/micro/spmm_hbm_ab.py "-DKGAT_SPMM_GROUP=8" "-DKGAT_SPMM_GROUP=2" ...
"""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

variants = sys.argv[1:] or ["-DKGAT_SPMM_GROUP=8"]
base = _lib.load()
tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
libs = {"shipped": base}
objs = [os.path.join(_lib.OBJ_DIR, s.replace(".hip", ".o")) for s in _lib.SOURCES]
for vi, flag in enumerate(variants):
    obj = "/tmp/spmm_var%d.o" % vi
    subprocess.check_call([_lib._hipcc()] + _lib.BASE_FLAGS + _lib.SOURCES["kgat_spmm.hip"] + flag.split() +
                          ["-c", os.path.join(_lib.CSRC, "kgat_spmm.hip"), "-o", obj])
    so = "/tmp/libkgat_hip_var%d.so" % vi
    these = [obj if o.endswith("kgat_spmm.o") else o for o in objs]
    if not all(os.path.exists(o) for o in these):   # the shipped objects did not travel: rebuild them once
        _lib.build(force=True)
    subprocess.check_call([_lib._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so] + these)
    _lib.SO_PATH, _lib._lib = so, None
    libs[flag] = _lib.load()

dev = torch.device("cuda:0")


def ab(label, indptr, col, row_of, X, w, mul_self, rounds):
    res = {k: [] for k in libs}
    outs = {}
    e = col.numel()
    ws = ops.spmm_workspace(e, X.shape[1], dev)
    out = torch.empty((indptr.numel() - 1, X.shape[1]), device=dev)
    for it in range(rounds + 3):
        for name, lib in libs.items():
            _lib._lib = lib
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.spmm(indptr, col, row_of, X, w, out=out, algo="merge", mul_self=mul_self, workspace=ws)
            b.record()
            torch.cuda.synchronize()
            if it >= 3:
                res[name].append(a.elapsed_time(b))
            if it == 0:
                outs[name] = out.clone() if e < 50_000_000 else out[:100000].clone()
    ref = outs["shipped"]
    for k, v in res.items():
        print("%-22s %-28s median %.4f min %.4f ms | same bits as shipped: %s" % (label, k, np.median(v), np.min(v),
                                                                               torch.equal(outs[k], ref)))


n, trip, R = synth.amazon_book_ckg()
src = torch.as_tensor(trip[:, 2].copy(), device=dev)
dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
ab("amazon-book D=64", indptr, col, row_of, torch.randn(n, 64, device=dev), torch.rand(len(trip), device=dev), True, 40)
ab("amazon-book D=32", indptr, col, row_of, torch.randn(n, 32, device=dev), torch.rand(len(trip), device=dev), True, 40)
del indptr, col, eid, row_of
n, e = 10_000_000, 200_000_000
src, dst, _ = synth.power_law_coo_device(n, e, 64, dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
del src, dst, eid, _
w = torch.rand(e, device=dev)
for trial in range(3):   # three allocations of the table: the arms share each one
    X = torch.randn(n, 64, device=dev)
    ab("10M/200M table %d" % trial, indptr, col, row_of, X, w, False, 8)
    keep = X  # hold it so the next trial gets another block
    X = None
