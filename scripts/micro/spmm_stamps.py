#!/usr/bin/env python3
"""Diagnostic (developer tool): phase times inside spmm_merge2_kernel from s_memtime stamps.
Builds a private copy of the library with -DKGAT_SPMM_STAMPS (the shipped library never
contains the stamps), runs the D=64 aggregation on the benchmark graph and prints the median
cycles per phase per tile and the tile start/end spread."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

so = "/tmp/libkgat_hip_stamps.so"
srcs = [os.path.join(_lib.CSRC, s) for s in _lib.SOURCES]
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.check_call([_lib._hipcc()] + _lib.BASE_FLAGS + ["-DKGAT_SPMM_STAMPS", '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()] + extra + ["-shared", "-o", so] + srcs)
print("build flags:", extra)
_lib.SO_PATH = so
_lib._lib = None
lib = _lib.load()
lib.kgat_debug_set_spmm_stamps.restype = C.c_int
lib.kgat_debug_set_spmm_stamps.argtypes = [C.c_void_p]
lib.kgat_debug_set_spmm_bi_stamps.restype = C.c_int
lib.kgat_debug_set_spmm_bi_stamps.argtypes = [C.c_void_p]

dev = torch.device("cuda:0")
MUL_SELF = os.environ.get("STAMPS_MUL_SELF", "0") == "1"   # (round 4: the step calls the plain operator)
same_rows = "--same-rows" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("-")]
D = int(args[0]) if args else 64
n, trip, R = synth.last_fm_ckg() if (len(args) > 1 and args[1] == "last-fm") else synth.amazon_book_ckg()
E = len(trip)
src = torch.as_tensor(trip[:, 2].copy(), device=dev)
dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
if same_rows:
    col = col % 16
X = torch.randn(n, D, device=dev)
w = torch.rand(E, device=dev)
te = (256 // (D // 4)) * (64 if D >= 32 else (32 if D == 16 else (16 if D == 8 else 8)))
tiles = (E + te - 1) // te
stamps = torch.zeros(tiles * 32, dtype=torch.int64, device=dev)
out = torch.empty(n, D, device=dev)
for _ in range(3):
    ops.spmm(indptr, col, row_of, X, w, out=out, mul_self=MUL_SELF)
assert lib.kgat_debug_set_spmm_stamps(stamps.data_ptr()) == 0
ops.spmm(indptr, col, row_of, X, w, out=out, mul_self=MUL_SELF)
torch.cuda.synchronize()
st = stamps.cpu().numpy()[:tiles * 16].reshape(tiles, 16).astype(np.float64)
names = ["stage (loads -> LDS records, barrier)", "edge loop", "partials -> LDS, barrier", "combine walk, barrier", "emit"]
print("same_rows =", same_rows, " tiles =", tiles, " (s_memtime ticks = shader cycles)")
for i, nm in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print("  %-40s median %8.0f   p10 %8.0f   p90 %8.0f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
import time
fn = lambda: ops.spmm(indptr, col, row_of, X, w, out=out, mul_self=MUL_SELF)
assert lib.kgat_debug_set_spmm_stamps(None) == 0
for _ in range(5): fn()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); ts.append((a, b))
torch.cuda.synchronize()
print("  op time (merge + finish): median %.4f ms" % np.median([a.elapsed_time(b) for a, b in ts]))
if "--fused" in sys.argv:
    # the one-launch layer (kgat_spmm_bi_fused_f32, D -> D_OUT): the same phases + W2 staging + the dense blocks
    d_out = int(os.environ.get("D_OUT", D))
    W2 = torch.randn(d_out, D, device=dev) / D ** 0.5
    h = torch.empty(n, d_out, device=dev)
    fz = lambda: ops.spmm_bi_fused(indptr, col, row_of, X, w, W2, 0.01, h_out=h, scratch=out)
    for _ in range(3): fz()
    stamps.zero_()
    assert lib.kgat_debug_set_spmm_bi_stamps(stamps.data_ptr()) == 0
    fz()
    torch.cuda.synchronize()
    assert lib.kgat_debug_set_spmm_bi_stamps(None) == 0
    sf = stamps.cpu().numpy().astype(np.float64)
    print("fused %d -> %d:" % (D, d_out))
    ro = row_of.cpu().numpy()
    te_f = te // 2 if (D >= 64 and any('KGAT_FUSED_HALF_RUNS=1' in a for a in extra)) else te
    tiles_f = (E + te_f - 1) // te_f
    sf = sf.reshape(-1)[:tiles_f * 16].reshape(tiles_f, 16)
    rows_t = ro[np.minimum(np.arange(tiles_f) * te_f + te_f, E) - 1] - ro[np.arange(tiles_f) * te_f] + 1
    for i, nm in enumerate(names[:4] + ["barrier (P rows visible)", "dense blocks"]):
        d = sf[:, i + 1] - sf[:, i]
        print("  %-40s median %8.0f   p10 %8.0f   p90 %8.0f   | tiles over 56 rows: median %8.0f" % (
            nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), np.median(d[rows_t > 56])))
    print("  %-40s median %8.0f" % ("whole tile", np.median(sf[:, 6] - sf[:, 0])))
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fz(); b.record(); ts.append((a, b))
    torch.cuda.synchronize()
    print("  op time (fused merge + finish): median %.4f ms" % np.median([a.elapsed_time(b) for a, b in ts]))
    t0 = sf[:, 0].min()
    print("  kernel span: %.0f cycles; tile starts: p50 %.0f p90 %.0f max %.0f" % (
        sf[:, 6].max() - t0, np.median(sf[:, 0] - t0), np.percentile(sf[:, 0] - t0, 90), (sf[:, 0] - t0).max()))
tot = st[:, 5] - st[:, 0]
print("  %-40s median %8.0f" % ("whole tile", np.median(tot)))
t0 = st[:, 0].min()
print("  kernel span (first start -> last end): %.0f cycles; tile starts: p50 %.0f p90 %.0f max %.0f" % (
    st[:, 5].max() - t0, np.median(st[:, 0] - t0), np.percentile(st[:, 0] - t0, 90), (st[:, 0] - t0).max()))
