#!/usr/bin/env python3
"""Diagnostic (developer tool): start / end s_memtime of every workgroup of the fused attention
kernel, for the equal-count split of the tiles and for cost-balanced splits.  Builds a private copy
of the library with -DKGAT_ATT_STAMPS (the shipped library never contains the stamps)."""
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
objs = []
procs = []
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
lib.kgat_debug_set_att_stamps.restype = C.c_int
lib.kgat_debug_set_att_stamps.argtypes = [C.c_void_p]

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
variants = [("equal tile counts", None)] + [("cost %s" % (c,), c) for c in (ops.FOLD_TILE_COST, ops.FOLD_TILE_COST_F32, (64, 12, 0), (64, 38, 0), (64, 25, 1051), (64, 50, 1400))]
fit_rows, fit_y = [], []
for cap in (256, 512):
    for name, cost in variants:
        tiles, tptr, parts = ops.fold_tiles(rp, gid, gptr, n_groups, cap=cap, cost=cost or ops.FOLD_TILE_COST)
        pt = parts if cost else None
        fn = lambda: ops.att_score_fused(n, rp, perm, sg, idx, gid, gptr, g_node, tiles, tptr, ent, W, rel,  # noqa: E731
                                         want_eid=False, part_tptr=pt)
        for _ in range(3):
            fn()
        stamps = torch.zeros(n_wg * 2, dtype=torch.int64, device=dev)
        assert lib.kgat_debug_set_att_stamps(stamps.data_ptr()) == 0
        fn()
        torch.cuda.synchronize()
        assert lib.kgat_debug_set_att_stamps(None) == 0
        st = stamps.cpu().numpy().reshape(n_wg, 2).astype(np.float64)
        dur = st[:, 1] - st[:, 0]
        t0 = st[:, 0].min()
        ev = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        print("cap %d %-22s kernel %.4f ms | workgroup busy time (s_memtime ticks): min %.0f p10 %.0f median %.0f p90 %.0f "
              "max %.0f  max/mean %.3f | last end - first start %.0f"
              % (cap, name, ms, dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(),
                 dur.max() / dur.mean(), st[:, 1].max() - t0))
        # regressors of the workgroup's busy time: tiles, first-chunk positions, later positions, later chunks, relations
        tl = tiles.cpu().numpy()[:int(tptr[-1])].astype(np.int64)
        pp = (pt if pt is not None else torch.as_tensor([len(tl) * b // n_wg for b in range(n_wg + 1)])).cpu().numpy()
        P = tl[:, 3] - tl[:, 2]
        feats = np.stack([np.ones_like(P), np.minimum(P, 64), np.maximum(P - 64, 0), (np.maximum(P - 64, 0) + 63) // 64], 1).astype(np.float64)
        cs = np.concatenate([np.zeros((1, 4)), np.cumsum(feats, 0)])
        rows = cs[pp[1:]] - cs[pp[:-1]]
        nrel = np.array([len(np.unique(tl[pp[b]:pp[b + 1], 0])) if pp[b + 1] > pp[b] else 0 for b in range(n_wg)], dtype=np.float64)
        fit_rows.append(np.concatenate([rows, nrel[:, None], np.ones((n_wg, 1))], 1))
        fit_y.append(dur)
A, y = np.concatenate(fit_rows), np.concatenate(fit_y)
coef, res, _, _ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ coef
print("least squares over %d workgroup samples: ticks = %.0f*tiles + %.1f*first_pos + %.1f*later_pos + %.0f*later_chunks "
      "+ %.0f*relations + %.0f ; rms residual %.0f (%.1f%% of mean)"
      % (len(y), *coef, np.sqrt(np.mean((pred - y) ** 2)), 100 * np.sqrt(np.mean((pred - y) ** 2)) / y.mean()))
