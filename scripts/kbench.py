#!/usr/bin/env python3
"""Kernel micro-benchmarks on the benchmark graph (developer tool; not part of the product
path or of bench.py's contract).  Interleaved rounds in one process, HIP events per launch.

  python scripts/kbench.py att   [--algos mfma,mfma_chunk,generic] [--dim 64]
  python scripts/kbench.py spmm  [--algos merge,rows,rows_ordered] [--dim 64]
  python scripts/kbench.py softmax
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fns, rounds, warm=3):
    names = list(fns)
    for _ in range(warm):
        for n in names:
            fns[n]()
    ts = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fns[n]()
            b.record()
            ts[n].append((a, b))
    torch.cuda.synchronize()
    return {n: np.array([a.elapsed_time(b) for a, b in v]) for n, v in ts.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", choices=["att", "spmm", "softmax", "train", "kg"])
    ap.add_argument("--algos", default=None)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--workload", default="amazon-book")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--same-rows", action="store_true")
    args = ap.parse_args()
    from dgl_kgat_amd import ops, synth
    dev = torch.device("cuda:0")
    if args.workload == "amazon-book":
        n, trip, R = synth.amazon_book_ckg(scale=args.scale)
    elif args.workload == "last-fm":
        n, trip, R = synth.last_fm_ckg(scale=args.scale)
    else:
        n, trip, R = synth.power_law_ckg(int(1e7 * args.scale), int(2e8 * args.scale), 64)
    E, D = len(trip), args.dim
    src = torch.as_tensor(trip[:, 2].copy(), device=dev)
    dst = torch.as_tensor(trip[:, 0].copy(), device=dev)
    et = torch.as_tensor(trip[:, 1].copy(), device=dev)
    indptr, col, eid, row_of = ops.csr_from_coo(n, src, dst)
    g = torch.Generator().manual_seed(0)
    print("N=%d E=%d R=%d D=%d" % (n, E, R, D))
    if args.kernel == "att":
        rel_ptr, perm = ops.group_by_relation(et, R)
        sg, dg = ops.gather(perm, src), ops.gather(perm, dst)
        pos = ops.gather(perm, ops.invert_permutation(eid))
        ent = torch.randn(n, D, generator=g).to(dev)
        W = ((torch.rand(R, D, D, generator=g) - 0.5) * (2 * 1.414 * (6 / (D * D + R * D)) ** 0.5)).to(dev)
        rel = torch.randn(R, D, generator=g).to(dev)
        algos = (args.algos or "mfma,mfma_chunk").split(",")
        fns = {a: (lambda a=a: ops.att_score(n, rel_ptr, perm, sg, dg, ent, W, rel, pos_g=pos, algo=a)) for a in algos}
        # split form: head projections once per (head, relation) group
        et_csr = ops.gather(eid, et)
        rp2, idx2 = ops.group_by_relation(et_csr, R)
        perm2, sg2, dg2 = ops.gather(idx2, eid), ops.gather(idx2, col), ops.gather(idx2, row_of)
        gid, gptr, g_node, n_groups = ops.head_groups(rp2, dg2)
        g_tab = torch.empty((max(n_groups, 1), D), device=dev)
        print("head groups: %d (%.2f edges per group)" % (n_groups, E / max(n_groups, 1)))
        if ops.att_score_folded_supported(n, D, D, R) and not ops.att_score_split_supported(n, D, D, R):
            fns["folded"] = lambda: ops.att_score_split(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, n_groups, ent, W, rel, g_tab=g_tab, folded=True)
            fns["folded_csr"] = lambda: ops.att_score_split(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, n_groups, ent, W, rel, g_tab=g_tab, folded=True, want_eid=False)
            algos = algos + ["folded"]
        if ops.att_score_fused_supported(n, D, D, R):
            for cap in (128, 256, 512):
                tl, tp, pp = ops.fold_tiles(rp2, gid, gptr, n_groups, cap=cap)
                print("fold tiles cap %d: %d (base %d)" % (cap, int(tp[-1]), (n_groups + 15) // 16))
                fns["fused%d_csr" % cap] = lambda tl=tl, tp=tp, pp=pp: ops.att_score_fused(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, tl, tp, ent, W, rel, want_eid=False, part_tptr=pp)
                fns["fused%d_csr_even" % cap] = lambda tl=tl, tp=tp: ops.att_score_fused(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, tl, tp, ent, W, rel, want_eid=False)
                if cap == 256:
                    for cost in ((64, 12, 0), (64, 12, 233), (64, 12, 700), (64, 8, 466), (64, 16, 466)):
                        pc = ops.fold_tiles(rp2, gid, gptr, n_groups, cap=cap, cost=cost)[2]
                        fns["fused256_cost%d_%d_%d" % cost] = lambda tl=tl, tp=tp, pc=pc: ops.att_score_fused(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, tl, tp, ent, W, rel, want_eid=False, part_tptr=pc)
            fns["fused"] = lambda tl=tl, tp=tp: ops.att_score_fused(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, tl, tp, ent, W, rel)
            algos = algos + ["fused"]
        if ops.att_score_split_supported(n, D, D, R):
            fns["split"] = lambda: ops.att_score_split(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, n_groups, ent, W, rel, g_tab=g_tab)
            fns["split_csr"] = lambda: ops.att_score_split(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, n_groups, ent, W, rel, g_tab=g_tab, want_eid=False)
            fns["folded"] = lambda: ops.att_score_split(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, n_groups, ent, W, rel, g_tab=g_tab, folded=True)
            fns["folded_csr"] = lambda: ops.att_score_split(n, rp2, perm2, sg2, idx2, gid, gptr, g_node, n_groups, ent, W, rel, g_tab=g_tab, folded=True, want_eid=False)
            algos = algos + ["split", "folded"]
        if args.same_rows:  # diagnostic: every edge reads rows 0..15 (cache resident): isolates gather latency
            sz, dz = sg % 16, dg % 16
            for a in algos:
                fns[a + "_samerows"] = (lambda a=a: ops.att_score(n, rel_ptr, perm, sz, dz, ent, W, rel, pos_g=pos, algo=a))
        res = timeit(fns, args.rounds)
        flops = E * (4 * D * D + 3 * D)
        for a, t in res.items():
            print("att %-10s median %.4f ms  min %.4f ms  -> %.1f TFLOP/s (%.1f%% of 157.3), %.2f G edges/s"
                  % (a, np.median(t), t.min(), flops / np.median(t) / 1e9, flops / np.median(t) / 1e9 / 157.3 * 100,
                     E / np.median(t) / 1e6))
        if args.check:
            from oracle import kgat_oracle as orc
            idx = np.random.default_rng(0).choice(E, 20000, replace=False)
            ref = orc.att_score(ent.cpu().numpy(), W.cpu().numpy(), rel.cpu().numpy(), trip[idx, 2], trip[idx, 0], trip[idx, 1])
            for a in algos:
                out = fns[a]()[0].cpu().numpy()[idx]
                print("  %-10s max abs err %.3e (max |ref| %.3f), rel-to-max %.2e" % (a, np.abs(out - ref).max(), np.abs(ref).max(),
                                                                                 np.abs(out - ref).max() / np.abs(ref).max()))
    elif args.kernel == "spmm":
        X = torch.randn(n, D, generator=g).to(dev)
        logits = torch.randn(E, generator=g).to(dev)
        _, w_csr = ops.edge_softmax(indptr, row_of, eid, logits, want_out=False, want_csr=True)
        order = ops.row_order_by_degree(indptr)
        out = torch.empty(n, D, device=dev)
        ws = ops.spmm_workspace(E, D, dev)
        algos = (args.algos or "merge,rows,rows_ordered").split(",")
        fns = {}
        for a in algos:
            if a == "rows_ordered":
                fns[a] = lambda: ops.spmm(indptr, col, row_of, X, w_csr, out=out, order=order, algo="rows", workspace=ws)
            else:
                fns[a] = lambda a=a: ops.spmm(indptr, col, row_of, X, w_csr, out=out, algo=a, workspace=ws,
                                              mul_self=os.environ.get("KBENCH_MUL_SELF", "0") == "1")  # (round 4: the step calls the plain operator)
        if args.same_rows:  # diagnostic: every edge gathers one of 16 rows (cache resident)
            col16 = col % 16
            fns["merge_samerows"] = lambda: ops.spmm(indptr, col16, row_of, X, w_csr, out=out, algo="merge", workspace=ws, mul_self=True)
        res = timeit(fns, args.rounds)
        b = E * (4 * D + 8) + n * (4 * D + 4)
        for a, t in res.items():
            print("spmm %-13s median %.4f ms  min %.4f ms -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s), %.2f G edges/s"
                  % (a, np.median(t), t.min(), b / np.median(t) / 1e6, b / np.median(t) / 1e6 / 80, E / np.median(t) / 1e6))
        # cold-cache variant: a 1 GiB write between launches evicts X, the indices and the
        # weights from L2 and the 256 MiB Infinity Cache, so the launch starts from HBM
        flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        cold = []
        for _ in range(args.rounds):
            flush.fill_(1.0)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            fns[algos[0]]()
            eb.record()
            cold.append((ea, eb))
        torch.cuda.synchronize()
        tc = np.array([x.elapsed_time(y) for x, y in cold])
        print("spmm %-13s cold caches: median %.4f ms  min %.4f ms -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)"
              % (algos[0], np.median(tc), tc.min(), b / np.median(tc) / 1e6, b / np.median(tc) / 1e6 / 80))
    elif args.kernel == "train":
        # the CF step of kgat.py:146-168: gnn (all layers, full graph) -> BPR loss -> backward -> Adam
        import time
        import dgl_kgat_amd as K
        torch.manual_seed(0)
        model = K.KGATPropagation(n, R, D, D, 3, D, dropout=0.1).to(dev)
        graph = synth.build_graph(n, trip, dev)
        opt = K.FusedAdam(model.parameters(), lr=0.01)
        with torch.no_grad():
            graph.edata["w"] = model.compute_attention(graph)
        B = 10240
        u = torch.randint(0, 70679, (B,), device=dev).int()
        pi = torch.randint(70679, 95594, (B,), device=dev).int()
        ni = torch.randint(70679, 95594, (B,), device=dev).int()

        def step():
            loss = model.get_loss(model.gnn(graph), u, pi, ni)
            loss.backward()
            opt.step()
            opt.zero_grad()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        with ops.KernelTimer() as kt:
            t0 = time.perf_counter()
            for _ in range(args.rounds):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.rounds
        print("train step (fwd+bwd+Adam, 3 layers, batch %d): %.3f ms -> %.2f G edge traversals/s (fwd+bwd = 6E)"
              % (B, dt * 1e3, 6 * E / dt / 1e9))
        for name, v in kt.summary().items():
            print("   %-14s %4d launches  avg %.4f ms" % (name, len(v), float(np.mean([m for _, m in v]))))
    elif args.kernel == "kg":
        # the KG step of kgat.py:116-136: TransR loss on a batch of 2048 triplets -> backward -> Adam over all parameters
        import time
        import dgl_kgat_amd as K
        torch.manual_seed(0)
        model = K.KGATPropagation(n, R, D, D, 3, D, dropout=0.1).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        B = 2048
        idx = torch.randint(0, E, (B,), device=dev)
        h, r, pt = dst[idx].long(), et[idx].long(), src[idx].long()
        nt = torch.randint(0, n, (B,), device=dev)

        h32, r32, pt32, nt32 = h.int(), r.int(), pt.int(), nt.int()
        for fused in (True, False, "fused+fusedAdam", "fused+FusedAdam(ours)+int32"):
            if fused == "fused+fusedAdam":
                opt = torch.optim.Adam(model.parameters(), lr=0.01, fused=True)
            if fused == "fused+FusedAdam(ours)+int32":
                opt = K.FusedAdam(model.parameters(), lr=0.01)
                h, r, pt, nt = h32, r32, pt32, nt32
            def step():
                loss = model.transR(h, r, pt, nt, fused=bool(fused))
                loss.backward()
                opt.step()
                opt.zero_grad()
                return loss
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.rounds):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.rounds
            print("KG step (TransR fwd+bwd+Adam, batch %d, %s): %.3f ms" % (B, {True: "fused kernels", False: "torch ops", "fused+fusedAdam": "fused kernels + torch's fused Adam"}.get(fused, "fused kernels + kgat FusedAdam + int32 ids"), dt * 1e3))
    else:
        logits = torch.randn(E, generator=g).to(dev)
        fns = {"softmax_eid": lambda: ops.edge_softmax(indptr, row_of, eid, logits, want_out=True, want_csr=True),
               "softmax_csr": lambda: ops.edge_softmax(indptr, row_of, eid, logits, in_csr_order=True, want_out=False, want_csr=True),
               "3pass_eid": lambda: ops.edge_softmax(indptr, row_of, eid, logits, want_out=True, want_csr=True, three_pass=True),
               "3pass_csr": lambda: ops.edge_softmax(indptr, row_of, eid, logits, in_csr_order=True, want_out=False, want_csr=True, three_pass=True)}
        res = timeit(fns, args.rounds)
        b = 12 * E + 4 * n
        for a, t in res.items():
            print("%-12s median %.4f ms min %.4f ms -> %.0f GB/s algorithmic" % (a, np.median(t), t.min(), b / np.median(t) / 1e6))


if __name__ == "__main__":
    main()
