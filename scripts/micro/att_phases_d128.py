#!/usr/bin/env python3
"""Diagnostic (developer tool): per-wave s_memtime totals of the d = 128 folded head kernel's
sections (first product, tanh, second product, store) for both product forms.  Private
-DKGAT_ATT_STAMPS build."""
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
lib.kgat_debug_set_att_stamps.restype = C.c_int
lib.kgat_debug_set_att_stamps.argtypes = [C.c_void_p]

dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg()
E, D = len(trip), 128
src = torch.as_tensor(trip[:, 2].copy(), device=dev)
dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
et = torch.as_tensor(trip[:, 1].copy(), device=dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
rp, idx = ops.group_by_relation(ops.gather(eid, et), R)
perm, sg, dg = ops.gather(idx, eid), ops.gather(idx, col), ops.gather(idx, row_of)
gid, gptr, g_node, n_groups = ops.head_groups(rp, dg)
g = torch.Generator().manual_seed(0)
ent = torch.randn(n, D, generator=g).to(dev)
W = ((torch.rand(R, D, D, generator=g) - 0.5) * (1.414 * (6 / (D * D + R * D)) ** 0.5)).to(dev)
rel = torch.randn(R, D, generator=g).to(dev)
n_wg = torch.cuda.get_device_properties(dev).multi_processor_count
g_tab = torch.empty((n_groups, D), device=dev)
for name, f32p in (("bf16-piece products", False), ("fp32 products", True)):
    fn = lambda: ops.att_score_split(n, rp, perm, sg, idx, gid, gptr, g_node, n_groups, ent, W, rel,  # noqa: E731
                                     want_eid=False, g_tab=g_tab, folded=True, f32_products=f32p)
    for _ in range(3):
        fn()
    ph = torch.zeros(n_wg * 8 * 5, dtype=torch.int64, device=dev)
    st = torch.zeros(n_wg * 2, dtype=torch.int64, device=dev)
    assert lib.kgat_debug_set_att_phases(ph.data_ptr()) == 0
    assert lib.kgat_debug_set_att_stamps(st.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    assert lib.kgat_debug_set_att_phases(None) == 0
    assert lib.kgat_debug_set_att_stamps(None) == 0
    sn = st.cpu().numpy().reshape(n_wg, 2).astype(np.float64)
    busy = sn[:, 1] - sn[:, 0]
    print("both launches %.3f ms | workgroup busy ticks: median %.0f max %.0f ; last end - first start %.0f"
          % (e0.elapsed_time(e1), np.median(busy), busy.max(), sn[:, 1].max() - sn[:, 0].min()))
    a = ph.cpu().numpy().reshape(n_wg * 8, 5).astype(np.float64)
    tl = a[:, 4].sum()
    tot = a[:, :4].sum(1)
    print("%-20s tiles %d | per tile, cycles of wave time: product 1 %.0f  tanh %.0f  product 2 %.0f  store %.0f  sum %.0f "
          "| per wave total: median %.0f max %.0f"
          % (name, tl, a[:, 0].sum() / tl, a[:, 1].sum() / tl, a[:, 2].sum() / tl, a[:, 3].sum() / tl,
             a[:, :4].sum() / tl, np.median(tot), tot.max()))
