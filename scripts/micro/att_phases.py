#!/usr/bin/env python3
"""Diagnostic (developer tool): where a wavefront of the fused attention kernel spends its time.
Per wave, s_memtime totals of the four sections of a tile step - issuing the next loads, the MFMA
phase (two products + tanh + parking V in LDS), the first 64 positions, the later chunks - for both
product forms.  Private -DKGAT_ATT_STAMPS build (the stamps themselves cost about a tenth)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

so = "/tmp/libkgat_hip_att_stamps.so"
tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
objs, procs = [], []
for src, extra in _lib.SOURCES.items():
    obj = "/tmp/att_stamps_%s.o" % src.replace(".hip", "")
    objs.append(obj)
    procs.append(subprocess.Popen([_lib._hipcc()] + _lib.BASE_FLAGS + extra + ["-DKGAT_ATT_STAMPS", tag, "-c",
                                   os.path.join(_lib.CSRC, src), "-o", obj]))
for p in procs:
    assert p.wait() == 0
subprocess.check_call([_lib._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so] + objs)
_lib.SO_PATH = so
_lib._lib = None
lib = _lib.load()
lib.kgat_debug_set_att_phases.restype = C.c_int
lib.kgat_debug_set_att_phases.argtypes = [C.c_void_p]

dev = torch.device("cuda:0")
workload = sys.argv[1] if len(sys.argv) > 1 else "amazon-book"
n, trip, R = synth.amazon_book_ckg() if workload == "amazon-book" else synth.last_fm_ckg()
E, D = len(trip), 64
src = torch.as_tensor(trip[:, 2].copy(), device=dev)
dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
et = torch.as_tensor(trip[:, 1].copy(), device=dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
rp, idx = ops.group_by_relation(ops.gather(eid, et), R)
perm, sg, dg = ops.gather(idx, eid), ops.gather(idx, col), ops.gather(idx, row_of)
gid, gptr, g_node, n_groups = ops.head_groups(rp, dg)
g = torch.Generator().manual_seed(0)
ent = torch.randn(n, D, generator=g).to(dev)
W = ((torch.rand(R, D, D, generator=g) - 0.5) * (2 * 1.414 * (6 / (D * D + R * D)) ** 0.5)).to(dev)
rel = torch.randn(R, D, generator=g).to(dev)
n_wg = torch.cuda.get_device_properties(dev).multi_processor_count
tiles, tptr, parts = ops.fold_tiles(rp, gid, gptr, n_groups)
for name, f32p in (("bf16-piece products", False), ("fp32 products", True)):
    fn = lambda: ops.att_score_fused(n, rp, perm, sg, idx, gid, gptr, g_node, tiles, tptr, ent, W, rel,  # noqa: E731
                                     want_eid=False, part_tptr=parts, f32_products=f32p)
    for _ in range(3):
        fn()
    ph = torch.zeros(n_wg * 8 * 5, dtype=torch.int64, device=dev)
    assert lib.kgat_debug_set_att_phases(ph.data_ptr()) == 0
    fn()
    torch.cuda.synchronize()
    assert lib.kgat_debug_set_att_phases(None) == 0
    a = ph.cpu().numpy().reshape(n_wg * 8, 5).astype(np.float64)
    tl = a[:, 4].sum()
    tot = a[:, :4].sum(1)
    print("%-20s tiles %d | per tile, cycles of wave time: issue %.0f  mfma phase %.0f  first chunk %.0f  later chunks %.0f  "
          "sum %.0f | per wave total: median %.0f max %.0f"
          % (name, tl, a[:, 0].sum() / tl, a[:, 1].sum() / tl, a[:, 2].sum() / tl, a[:, 3].sum() / tl,
             a[:, :4].sum() / tl, np.median(tot), tot.max()))
