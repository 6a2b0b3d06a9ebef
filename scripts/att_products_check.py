"""Developer tool: the fused attention kernel with its two product forms (three-bf16-piece
products on v_mfma_f32_16x16x32_bf16, fp32 products on v_mfma_f32_16x16x4_f32) against an fp64
evaluation of reference models.py:135-144 on the device, and their kernel times.

  python scripts/att_products_check.py [--workload amazon|lastfm] [--dim 64] [--scale 1.0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgl_kgat_amd  # noqa: E402,F401
from dgl_kgat_amd import ops, synth  # noqa: E402


def rel_err_8c(x, y):
    return float(((x - y).abs() / torch.maximum(y.abs(), 1e-3 * y.abs().max())).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="amazon")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=50)
    ap.add_argument("--ent-scale", type=float, default=1.0)
    ap.add_argument("--costs", default="", help="tile split costs to time, e.g. '64,38,1051;64,30,800'")
    ap.add_argument("--cap", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, trip, R = (synth.amazon_book_ckg if args.workload == "amazon" else synth.last_fm_ckg)(scale=args.scale)
    D = args.dim
    src = torch.as_tensor(trip[:, 2].astype(np.int32)).to(dev)
    dst = torch.as_tensor(trip[:, 0].astype(np.int32)).to(dev)
    et = torch.as_tensor(trip[:, 1].astype(np.int32)).to(dev)
    E = len(trip)
    indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
    g = torch.Generator().manual_seed(0)
    ent = (torch.randn(n, D, generator=g) * args.ent_scale).to(dev)
    W = ((torch.rand(R, D, D, generator=g) - 0.5) * (2 * 1.414 * (6 / (D * D + R * D)) ** 0.5)).to(dev)
    rel = torch.randn(R, D, generator=g).to(dev)
    et_csr = ops.gather(eid, et)
    rp2, idx2 = ops.group_by_relation(et_csr, R)
    perm2, sg2, dg2 = ops.gather(idx2, eid), ops.gather(idx2, col), ops.gather(idx2, row_of)
    gid, gptr, g_node, n_groups = ops.head_groups(rp2, dg2)
    tl, tp, pp = ops.fold_tiles(rp2, gid, gptr, n_groups, **({"cap": args.cap} if args.cap else {}))
    print("N=%d E=%d R=%d D=%d groups=%d tiles=%d" % (n, E, R, D, n_groups, int(tp[-1])))
    if D == 128:
        W = W * 0.5  # keep the logits O(1) at this width

    # fp64 on the device, relation by relation
    ref = torch.zeros(E, dtype=torch.float64, device=dev)
    e64, W64, r64 = ent.double(), W.double(), rel.double()
    for r in range(R):
        idx = torch.nonzero(et == r).squeeze(1)
        if idx.numel() == 0:
            continue
        t = e64[src[idx].long()] @ W64[r]
        h = e64[dst[idx].long()] @ W64[r]
        ref[idx] = (t * torch.tanh(h + r64[r])).sum(1)
    out = {}
    fused = ops.att_score_fused_supported(n, D, D, R)
    g_tab = torch.empty((max(n_groups, 1), D), device=dev) if (not fused or D == 128) else None
    for name, f32p in (("bf16x3 pieces", False), ("fp32 products", True)):
        if fused and not (D == 128 and f32p):   # (the one-launch form at d = 128 has the piece products only)
            fn = lambda: ops.att_score_fused(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, tl, tp, ent, W, rel,  # noqa: E731
                                             want_csr=False, part_tptr=pp, f32_products=f32p)[0]
        else:  # two-launch folded form (d = 128)
            fn = lambda: ops.att_score_split(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, n_groups, ent, W, rel,  # noqa: E731
                                             want_csr=False, g_tab=g_tab, folded=True, f32_products=f32p)[0]
        y = fn()
        torch.cuda.synchronize()
        err = (y.double() - ref).abs()
        out[name] = y
        for _ in range(5):
            fn()
        ts = []
        for _ in range(args.rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print("%-14s max|err| %.3e  mean|err| %.3e  rel_err[8c] %.3e  (max|ref| %.3f)   median %.4f ms  min %.4f ms"
              % (name, float(err.max()), float(err.mean()), rel_err_8c(y.double(), ref), float(ref.abs().max()),
                 float(np.median(ts)), float(np.min(ts))))
    for cs in [c for c in args.costs.split(";") if c]:
        cost = tuple(int(v) for v in cs.split(","))
        pc = ops.fold_tiles(rp2, gid, gptr, n_groups, cost=cost, **({"cap": args.cap} if args.cap else {}))[2]
        for name, f32p in (("bf16x3", False), ("fp32", True)):
            fn = lambda: ops.att_score_fused(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, tl, tp, ent, W, rel,  # noqa: E731
                                             want_eid=False, part_tptr=pc, f32_products=f32p)
            for _ in range(5):
                fn()
            ts = []
            for _ in range(args.rounds):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            print("cost %-16s %-7s median %.4f ms  min %.4f ms" % (cost, name, float(np.median(ts)), float(np.min(ts))))
    d = (out["bf16x3 pieces"] - out["fp32 products"]).abs()
    print("between the two forms: max %.3e  mean %.3e" % (float(d.max()), float(d.mean())))


if __name__ == "__main__":
    main()
