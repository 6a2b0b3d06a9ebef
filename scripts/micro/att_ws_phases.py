#!/usr/bin/env python3
"""Diagnostic (developer tool): where the producer and consumer waves of att_fold_ws_kernel spend their time
(s_memtime totals per wave, private -DKGAT_ATT_STAMPS build; extra -D flags may follow on the command line).
Producer sections: issue (descriptor, head index, next head rows) | head rows arrive + first cut | product 1 |
tanh + cut | product 2 | wait for a free slot | park V + publish.  Consumer: issue records | wait for V | first
chunk (incl. its rows' arrival) | later chunks."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import _lib, ops, synth  # noqa: E402

D = 64
if len(sys.argv) > 1 and sys.argv[1].isdigit():
    D = int(sys.argv.pop(1))
extra_flags = sys.argv[1:]
so = "/tmp/libkgat_hip_att_ws_stamps.so"
tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % _lib.source_hash()
objs, procs = [], []
for src, extra in _lib.SOURCES.items():
    obj = "/tmp/att_ws_stamps_%s.o" % src.replace(".hip", "")
    objs.append(obj)
    procs.append(subprocess.Popen([_lib._hipcc()] + _lib.BASE_FLAGS + extra + ["-DKGAT_ATT_STAMPS", "-DKGAT_ATT_WAVE_ROLES", tag] + extra_flags +
                                  ["-c", os.path.join(_lib.CSRC, src), "-o", obj]))
for p in procs:
    assert p.wait() == 0
subprocess.check_call([_lib._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so] + objs)
_lib.SO_PATH = so
_lib._lib = None
lib = _lib.load()
lib.kgat_debug_set_att_phases.restype = C.c_int
lib.kgat_debug_set_att_phases.argtypes = [C.c_void_p]

dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg()
E = len(trip)
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
n_wg = torch.cuda.get_device_properties(dev).multi_processor_count
tiles, tptr, parts = ops.fold_tiles(rp, gid, gptr, n_groups, cost=ops.fold_tile_cost(D))
fn = lambda: ops.att_score_fused(n, rp, perm, sg, idx, gid, gptr, g_node, tiles, tptr, ent, W, rel, want_eid=False,  # noqa: E731
                                 want_csr=False, want_grouped=True, part_tptr=parts, rec_g=rec)
for _ in range(3):
    fn()
NWAVE = 16 if D == 64 else 8
ph = torch.zeros(n_wg * NWAVE * 8, dtype=torch.int64, device=dev)
assert lib.kgat_debug_set_att_phases(ph.data_ptr()) == 0
a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a0.record()
fn()
b0.record()
torch.cuda.synchronize()
assert lib.kgat_debug_set_att_phases(None) == 0
a = ph.cpu().numpy().reshape(n_wg, NWAVE, 8).astype(np.float64)
tiles_of = a[:, :, 7]
prod = tiles_of > 0
# producers are the waves with sections 4..6 populated
is_p = (a[:, :, 4].sum(0) > 0) if D == 64 else (np.arange(NWAVE) < 4)
print("launch %.1f us (stamped build) flags %s" % (a0.elapsed_time(b0) * 1e3, extra_flags))
for name, sel, labels in (("producer", is_p, ["issue", "rows+cut", "product 1", "tanh+cut", "product 2", "wait free", "park+publish"]),
                          ("consumer", ~is_p, ["issue", "wait V", "first chunk", "later chunks"])):
    x = a[:, sel, :]
    tl = x[:, :, 7].sum()
    if tl == 0:
        continue
    per = [x[:, :, k].sum() / tl for k in range(len(labels))]
    tot = x[:, :, :len(labels)].sum(2)
    print("%-9s waves %d tiles %d | per tile, ticks of wave time: %s | sum %.0f | per wave total: median %.0f max %.0f" % (
        name, sel.sum(), tl, "  ".join("%s %.0f" % (l, v) for l, v in zip(labels, per)), sum(per), np.median(tot), tot.max()))
