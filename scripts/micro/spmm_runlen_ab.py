#!/usr/bin/env python3
"""Developer tool: the shipped merge SpMM against a build with one more compiler flag (AB_FLAG; default
-DKGAT_SPMM_MID_LIMIT=0: no half-length runs for mid-size launches; -DKGAT_SPMM_XCD_REMAP=1: every XCD a
contiguous eighth of the tiles), alternating the two builds launch by launch on one box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

base = _lib.load()
so = "/tmp/libkgat_hip_nomid.so"
tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
objs, procs = [], []
for src, extra in _lib.SOURCES.items():
    obj = "/tmp/nomid_%s.o" % src.replace(".hip", "")
    objs.append(obj)
    procs.append(subprocess.Popen([_lib._hipcc()] + _lib.BASE_FLAGS + extra + os.environ.get("AB_FLAG", "-DKGAT_SPMM_MID_LIMIT=0").split() + [tag, "-c",
                                   os.path.join(_lib.CSRC, src), "-o", obj]))
for p in procs:
    assert p.wait() == 0
subprocess.check_call([_lib._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so] + objs)
_lib.SO_PATH = so
_lib._lib = None
nomid = _lib.load()
AB = os.environ.get("AB_FLAG", "-DKGAT_SPMM_MID_LIMIT=0")
libs = {"shipped": base, AB: nomid}

dev = torch.device("cuda:0")
WLS = os.environ.get("AB_WORKLOADS", "amazon-book,last-fm").split(",")
DIMS = [int(x) for x in os.environ.get("AB_DIMS", "8,16,32,64,128").split(",")]
for wl, mk in (("amazon-book", synth.amazon_book_ckg), ("last-fm", synth.last_fm_ckg)):
    if wl not in WLS:
        continue
    n, trip, R = mk()
    src = torch.as_tensor(trip[:, 2].copy(), device=dev)
    dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
    indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
    w = torch.rand(len(trip), device=dev)
    for D in DIMS:
        X = torch.randn(n, D, device=dev)
        res, outs = {}, {}
        for name in libs:
            res[name] = []
        for it in range(43):
            for name, lib in libs.items():
                _lib._lib = lib
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                o = ops.spmm(indptr, col, row_of, X, w, algo="merge", mul_self=os.environ.get("AB_MUL_SELF", "1") == "1")
                b.record()
                torch.cuda.synchronize()
                if it >= 3:
                    res[name].append(a.elapsed_time(b))
                outs[name] = o
        same = torch.equal(outs["shipped"], outs[AB])
        print("%-12s D=%3d  " % (wl, D) + "  ".join("%s: median %.4f min %.4f ms" % (k, np.median(v), np.min(v))
                                                    for k, v in res.items()) + "  | same bits: %s" % same)
