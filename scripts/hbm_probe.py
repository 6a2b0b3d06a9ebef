#!/usr/bin/env python3
"""Developer probe: the D = 64 SpMM on device-drawn power-law graphs (HBM-resident X), one cap
variant per invocation so that a rocprofv3 --kernel-trace --stats run separates the kernels.

  python scripts/hbm_probe.py redraw|shift [nodes] [edges] [launches]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from dgl_kgat_amd import ops, synth
    cap = sys.argv[1] if len(sys.argv) > 1 else "redraw"
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    e = int(float(sys.argv[3])) if len(sys.argv) > 3 else 200_000_000
    launches = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    mul_self = os.environ.get("PROBE_MUL_SELF", "1") == "1"
    dev = torch.device("cuda:0")
    if cap == "numpy":
        _, trip, _ = synth.power_law_ckg(n, e, 64)
        src = torch.as_tensor(trip[:, 2].copy(), device=dev)
        dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
        del trip
    else:
        src, dst, _ = synth.power_law_coo_device(n, e, 64, dev, cap=cap)
    indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
    deg = (indptr[1:] - indptr[:-1])
    print("cap=%s N=%d E=%d max deg %d zero rows %d rows>1024 %d edges in rows>1024 %.3f" % (
        cap, n, e, int(deg.max()), int((deg == 0).sum()), int((deg > 1024).sum()),
        float(deg[deg > 1024].sum()) / e))
    del src, dst, eid
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    X = torch.randn((n, 64), generator=gen, device=dev)
    w = torch.rand(e, generator=gen, device=dev)
    if os.environ.get("PROBE_SOFTMAX_W"):
        w = ops.edge_softmax(indptr, row_of, torch.arange(e, dtype=torch.int32, device=dev), w * 6, in_csr_order=True,
                             want_out=False, want_csr=True)[1]
    xv = os.environ.get("PROBE_X", "randn")
    if xv == "zeros":
        X.zero_()
    elif xv == "small":
        X.mul_(1e-3)
    elif xv == "relu":
        X.clamp_(min=0)
    elif xv == "const":
        X.fill_(0.37)
    out = torch.empty((n, 64), device=dev)
    ws = ops.spmm_workspace(e, 64, dev)
    for _ in range(5):
        ops.spmm(indptr, col, row_of, X, w, out=out, mul_self=mul_self, workspace=ws)
    evs = []
    for _ in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.spmm(indptr, col, row_of, X, w, out=out, mul_self=mul_self, workspace=ws)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    bts = e * (4 * 64 + 8) + n * (4 * 64 + 4)
    print("spmm D=64 mul_self=%d: median %.4f ms min %.4f -> %.1f GB/s algorithmic = %.4f of 8 TB/s"
          % (mul_self, np.median(t), t.min(), bts / np.median(t) / 1e6, bts / np.median(t) / 1e6 / 8000))


if __name__ == "__main__":
    main()
