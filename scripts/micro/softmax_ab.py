#!/usr/bin/env python3
"""Developer tool: build-time variants of csrc/kgat_softmax.hip (-D flags) on the benchmark graph, the
sweep fed as the propagation path feeds it - grouped-order logits read through the CSR-position ->
grouped-position map - and with CSR-ordered logits; variants alternate launch by launch.

  python scripts/micro/softmax_ab.py "-DKGAT_SM_SMALL_EDGES=0" "src=scripts/micro/build/kgat_softmax_old.hip"

A variant is a set of compiler flags for the shipped source, or src=<another source file> (e.g. an earlier
revision saved with `git show <rev>:dgl-kgat_amd/csrc/kgat_softmax.hip`).  A torch elementwise pass over the
same bytes (three E-sized reads, one write) is timed beside them as the stream floor of this size.
"""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

variants = sys.argv[1:]
base = _lib.load()
libs = {"shipped": base}
objs = [os.path.join(_lib.OBJ_DIR, s.replace(".hip", ".o")) for s in _lib.SOURCES]
if not all(os.path.exists(o) for o in objs):
    _lib.build(force=True)
for vi, flag in enumerate(variants):
    obj = "/tmp/sm_var%d.o" % vi
    src_file = os.path.join(_lib.CSRC, "kgat_softmax.hip")
    extra = flag.split()
    if flag.startswith("src="):
        src_file, extra = os.path.join(ROOT, flag[4:]), []
    subprocess.check_call([_lib._hipcc()] + _lib.BASE_FLAGS + _lib.SOURCES["kgat_softmax.hip"] + extra +
                          ["-c", src_file, "-o", obj])
    so = "/tmp/libkgat_hip_smvar%d.so" % vi
    subprocess.check_call([_lib._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so] +
                          [obj if o.endswith("kgat_softmax.o") else o for o in objs])
    _lib.SO_PATH, _lib._lib = so, None
    libs[flag] = _lib.load()
_lib._lib = base

dev = torch.device("cuda:0")


def _power_law():  # SOFTMAX_AB_WORKLOAD=power-law: the 10 M / 200 M graph drawn on the device (16 positions per lane by default)
    n = 10_000_000
    s_, d_, t_ = synth.power_law_coo_device(n, 200_000_000, 64, dev)
    return n, (s_, d_, t_), 64


workloads = ((("power-law", _power_law),) if os.environ.get("SOFTMAX_AB_WORKLOAD") == "power-law"
             else (("amazon-book", synth.amazon_book_ckg), ("last-fm", synth.last_fm_ckg)))
for wl, mk in workloads:
    n, trip, R = mk()
    if isinstance(trip, tuple):
        src, dst, et = trip
        trip = src  # (only its length is used below)
    else:
        src = torch.as_tensor(trip[:, 2].copy(), device=dev)
        dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
        et = torch.as_tensor(trip[:, 1].copy(), device=dev)
    indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
    rp, idx = ops.group_by_relation(ops.gather(eid, et), R)
    gpos = ops.invert_permutation(idx)           # grouped position of every CSR position
    logits_csr = torch.randn(len(trip), device=dev) * 2
    logits_g = ops.gather(idx, logits_csr)       # the same logits in grouped order
    res = {}
    outs = {}
    for it in range(23):
        for name, lib in libs.items():
            _lib._lib = lib
            for mode in ("indexed", "csr"):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if mode == "indexed":
                    o = ops.edge_softmax(indptr, row_of, gpos, logits_g, in_csr_order=False, want_out=False, want_csr=True)[1]
                else:
                    o = ops.edge_softmax(indptr, row_of, eid, logits_csr, in_csr_order=True, want_out=False, want_csr=True)[1]
                b.record()
                torch.cuda.synchronize()
                if it >= 3:
                    res.setdefault((name, mode), []).append(a.elapsed_time(b))
                outs[(name, mode)] = o
    fa, fb, fc = torch.randn(3, len(trip), device=dev).unbind(0)
    fo = torch.empty_like(fa)
    fl = []
    for it in range(23):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        torch.addcmul(fa, fb, fc, out=fo)
        b.record()
        torch.cuda.synchronize()
        if it >= 3:
            fl.append(a.elapsed_time(b))
    print("%-12s %-36s %-8s median %.4f min %.4f ms" % (wl, "torch.addcmul (3 reads, 1 write)", "stream", np.median(fl), np.min(fl)))
    ref = outs[("shipped", "csr")]
    for (name, mode), v in res.items():
        o = outs[(name, mode)]
        print("%-12s %-36s %-8s median %.4f min %.4f ms | max |diff| vs shipped csr %.2e" % (
            wl, name, mode, np.median(v), np.min(v), float((o - ref).abs().max())))
