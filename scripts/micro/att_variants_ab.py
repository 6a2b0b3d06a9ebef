#!/usr/bin/env python3
"""Developer tool: build-time variants of the fused attention kernels (-D flags of
csrc/kgat_att_persistent.hip) on the benchmark graph, alternating launch by launch, for a list of
tile-split cost triples.  Only kgat_att_persistent.hip is recompiled per variant.  The variant libraries are
linked with -Bsymbolic: the loader opens libraries RTLD_GLOBAL, and without it a variant's calls from one
translation unit into another (the entry point in kgat_att.hip -> the launchers in kgat_att_persistent.hip)
bind to the FIRST library loaded, i.e. every variant silently runs the shipped kernels.

  python scripts/micro/att_variants_ab.py --dim 128 --costs 64,20,600:64,12,700 -- "-DKGAT_F128_PASSES=2"
"""
import argparse
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--costs", default="64,20,600")
ap.add_argument("--caps", default="256")
ap.add_argument("--rounds", type=int, default=15)
ap.add_argument("--workload", default="amazon-book")
ap.add_argument("variants", nargs="*")
args = ap.parse_args()

base = _lib.load()
tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
libs = {"shipped": base}
objs = [os.path.join(_lib.OBJ_DIR, s.replace(".hip", ".o")) for s in _lib.SOURCES]
if not all(os.path.exists(o) for o in objs):
    _lib.build(force=True)
for vi, flag in enumerate(args.variants):
    obj = "/tmp/att_var%d.o" % vi
    subprocess.check_call([_lib._hipcc()] + _lib.BASE_FLAGS + _lib.SOURCES["kgat_att_persistent.hip"] + flag.split() +
                          ["-c", os.path.join(_lib.CSRC, "kgat_att_persistent.hip"), "-o", obj])
    so = "/tmp/libkgat_hip_attvar%d.so" % vi
    subprocess.check_call([_lib._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so] +
                          [obj if o.endswith("kgat_att_persistent.o") else o for o in objs])
    _lib.SO_PATH, _lib._lib = so, None
    libs[flag] = _lib.load()
_lib._lib = base

dev = torch.device("cuda:0")
n, trip, R = (synth.amazon_book_ckg if args.workload == "amazon-book" else synth.last_fm_ckg)()
D = args.dim
src = torch.as_tensor(trip[:, 2].copy(), device=dev)
dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
et = torch.as_tensor(trip[:, 1].copy(), device=dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
rp, idx = ops.group_by_relation(ops.gather(eid, et), R)
perm, sg, dg = ops.gather(idx, eid), ops.gather(idx, col), ops.gather(idx, row_of)
gid, gptr, g_node, n_groups = ops.head_groups(rp, dg)
rec = ops.att_pack_records(rp, gptr, gid, sg)
g = torch.Generator().manual_seed(0)
ent = torch.randn(n, D, generator=g).to(dev)
W = ((torch.rand(R, D, D, generator=g) - 0.5) * (2 * 1.414 * (6 / (D * D + R * D)) ** 0.5)).to(dev)
rel = torch.randn(R, D, generator=g).to(dev)
print("N=%d E=%d R=%d D=%d groups %d" % (n, len(trip), R, D, n_groups))
fns = {}
for cap in [int(c) for c in args.caps.split(",")]:
    for cost in [tuple(int(x) for x in c.split(",")) for c in args.costs.split(":")]:
        tl, tp, pp = ops.fold_tiles(rp, gid, gptr, n_groups, cap=cap, cost=cost)
        for name, lib in libs.items():
            def fn(tl=tl, tp=tp, pp=pp, lib=lib):
                _lib._lib = lib
                return ops.att_score_fused(n, rp, perm, sg, idx, gid, gptr, g_node, tl, tp, ent, W, rel, want_eid=False,
                                           want_csr=False, want_grouped=True, part_tptr=pp, rec_g=rec)[2]
            fns["cap %d cost %s %s" % (cap, cost, name)] = fn
ref = None
res = {k: [] for k in fns}
for it in range(args.rounds + 3):
    for k, fn in fns.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        torch.cuda.synchronize()
        if it >= 3:
            res[k].append(a.elapsed_time(b))
        if it == 0:
            ref = out.clone() if ref is None else ref
            if not torch.equal(out, ref):
                print("   (differs from the first variant's output: max |diff| %.3e) %s" % (float((out - ref).abs().max()), k))
for k, v in res.items():
    print("%-60s median %.4f min %.4f ms" % (k, np.median(v), np.min(v)))
