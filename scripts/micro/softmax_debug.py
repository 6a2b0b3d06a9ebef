#!/usr/bin/env python3
"""Developer tool: where does the softmax sweep differ from an fp64 segment softmax?  Prints the wrong
positions with their lane / slot / wavefront and the row extents around them."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
EPL = 8


def check(name, dst, seed=0):
    e = len(dst)
    n = int(dst.max()) + 2
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e).astype(np.int32)
    s = (rng.standard_normal(e) * 3).astype(np.float32)
    indptr, col, eid, row_of = ops.csr_from_coo(n, torch.as_tensor(src, device=dev), torch.as_tensor(dst, device=dev))
    s_csr = torch.as_tensor(s, device=dev)[eid.long()]
    _, o = ops.edge_softmax(indptr, row_of, eid, s_csr, in_csr_order=True, want_out=False, want_csr=True)
    o = o.cpu().numpy().astype(np.float64)
    r = row_of.cpu().numpy()
    x = s_csr.cpu().numpy().astype(np.float64)
    m = np.full(n, -np.inf)
    np.maximum.at(m, r, x)
    ex = np.exp(x - m[r])
    z = np.zeros(n)
    np.add.at(z, r, ex)
    ref = ex / z[r]
    bad = np.nonzero(np.abs(o - ref) > 1e-5 * np.maximum(ref, 1e-3))[0]
    print("%-28s E=%d rows=%d bad=%d" % (name, e, len(np.unique(r)), len(bad)))
    ip = indptr.cpu().numpy()
    shown = set()
    for p in bad[:400]:
        row = r[p]
        if row in shown:
            continue
        shown.add(row)
        b, en = ip[row], ip[row + 1]
        w = p // (64 * EPL)
        q = p % (64 * EPL)
        print("   row %d [%d,%d) len %d: first bad p=%d wave %d lane %d slot %d | row spans lanes %d.%d - %d.%d (wave %d-%d) got %.5f ref %.5f ratio %.4f" % (
            row, b, en, en - b, p, w, q // EPL, q % EPL, (b % 512) // EPL, b % EPL, ((en - 1) % 512) // EPL, (en - 1) % EPL,
            b // 512, (en - 1) // 512, o[p], ref[p], o[p] / ref[p]))
        if len(shown) >= 12:
            break


rng = np.random.default_rng(1)
check("uniform deg~17", np.sort(rng.integers(0, 300, 5000)).astype(np.int32))
check("one lane rows (8 each)", np.repeat(np.arange(64 * 3), 8).astype(np.int32))
check("rows of 3", np.repeat(np.arange(700), 3).astype(np.int32))
check("rows of 16", np.repeat(np.arange(100), 16).astype(np.int32))
check("rows of 20", np.repeat(np.arange(100), 20).astype(np.int32))
check("rows of 100", np.repeat(np.arange(30), 100).astype(np.int32))
check("rows of 200", np.repeat(np.arange(30), 200).astype(np.int32))
check("rows of 512", np.repeat(np.arange(5), 512).astype(np.int32))
check("rows of 700", np.repeat(np.arange(5), 700).astype(np.int32))
check("one row 5000", np.zeros(5000, np.int32))
check("mixed", np.sort(np.concatenate([rng.integers(0, 50, 300), np.full(1500, 60), rng.integers(61, 400, 3000)])).astype(np.int32))
