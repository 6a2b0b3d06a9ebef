#!/usr/bin/env python3
"""Minimal KGAT training / evaluation harness over this package: the epoch
structure of the reference's ``kgat.py:114-196`` - KG phase (TransR), attention refresh under
no_grad, CF phase (full-graph ``gnn`` + BPR loss per batch), evaluation (recall@20 / ndcg@20
on the validation and test interactions) - with the propagation path running on the HIP
kernels.  Samplers are the reference's "uniform" modes (uniform positive edge, uniformly random
negative) drawn with torch on the device instead of DGL's C++ EdgeSampler.

  python examples/train_kgat.py --data_dir datasets/amazon-book/data      # reference file format
  python examples/train_kgat.py --synthetic 0.01 --epochs 2              # amazon-book-shaped toy
  python examples/train_kgat.py --synthetic 0.01 --gpus 2                # CF phase on destination shards

``--gpus N``: one process per GPU (started here as a child ``torch.distributed.run``),
parameters replicated, the training graph sharded by destination range.  Every rank draws the same
batches (same seed); the attention refresh and the CF forward run on the local shard with one
layer-output exchange per layer, the CF backward sums the per-rank gradients of the replicated
operands (all-reduce), so all ranks apply the one-GPU run's update.  The KG phase (dense TransR
batches, no graph) runs replicated.
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgl_kgat_amd as K  # noqa: E402
from dgl_kgat_amd import ckg_io, metrics, synth  # noqa: E402


def synthetic_data_dir(scale, seed=1234):
    """An amazon-book-shaped CKG written in the reference's file format (90/5/5 split)."""
    n, trip, n_rel = synth.amazon_book_ckg(seed=seed, scale=scale)
    n_users = max(int(round(70679 * scale)), 4)
    kg = trip[trip[:, 1] < n_rel - 2].copy()
    uv = trip[trip[:, 1] == n_rel - 2][:, [0, 2]].copy()
    kg[:, 0] -= n_users
    kg[:, 2] -= n_users
    uv[:, 1] -= n_users
    uv = np.unique(uv, axis=0)
    rng = np.random.default_rng(seed)
    rng.shuffle(uv)
    # every user / item must appear in the training split (the reference assumes it, dataset.py:31)
    first = np.unique(np.concatenate([np.unique(uv[:, 0], return_index=True)[1], np.unique(uv[:, 1], return_index=True)[1]]))
    rest = np.setdiff1d(np.arange(len(uv)), first)
    n_hold = len(rest) // 10
    train = uv[np.concatenate([first, rest[2 * n_hold:]])]
    val, test = uv[rest[:n_hold]], uv[rest[n_hold:2 * n_hold]]
    remap_u = {u: i for i, u in enumerate(np.unique(train[:, 0]))}
    remap_v = {v: i for i, v in enumerate(np.unique(train[:, 1]))}
    fix = lambda a: np.array([[remap_u[u], remap_v[v]] for u, v in a if u in remap_u and v in remap_v], np.int32).reshape(-1, 2)  # noqa: E731
    n_items_old = int(kg[:, [0, 2]].max()) + 1
    ent_map = np.full(n_items_old, -1, np.int64)
    for v, i in remap_v.items():
        ent_map[v] = i
    nxt = len(remap_v)
    for e in np.unique(kg[:, [0, 2]]):
        if ent_map[e] < 0:
            ent_map[e] = nxt
            nxt += 1
    kg[:, 0], kg[:, 2] = ent_map[kg[:, 0]], ent_map[kg[:, 2]]
    kg[:, 1] = np.unique(kg[:, 1], return_inverse=True)[1]
    d = os.path.join(tempfile.mkdtemp(prefix="kgat_synth_"), "data")
    ckg_io.save_ckg_files(d, len(remap_u), fix(train), fix(val), fix(test), np.unique(kg, axis=0))
    return d


def user_dict(pairs, item_offset):
    order = np.argsort(pairs[:, 0], kind="stable")
    p = pairs[order]
    users, start = np.unique(p[:, 0], return_index=True)
    return {int(u): p[s:e, 1] - item_offset for u, s, e in zip(users, start, list(start[1:]) + [len(p)])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_dir", default=None)
    ap.add_argument("--synthetic", type=float, default=0.01, help="scale of the synthetic amazon-book-shaped CKG")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--entity_embed_dim", type=int, default=64)
    ap.add_argument("--relation_embed_dim", type=int, default=64)
    ap.add_argument("--gnn_num_layer", type=int, default=3)
    ap.add_argument("--gnn_hidden_size", type=int, default=64)
    ap.add_argument("--dropout_rate", type=float, default=0.1)
    ap.add_argument("--lr", type=float, default=0.0001)
    ap.add_argument("--batch_size", type=int, default=10240)
    ap.add_argument("--batch_size_kg", type=int, default=2048)
    ap.add_argument("--max_iters", type=int, default=0, help="cap on iterations per phase (0 = full epoch)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--grad_digest", action="store_true",
                    help="print |grad| sums of the first CF step (to compare an N-GPU run with the one-GPU run)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        if args.data_dir is None:   # every rank must read the same files
            args.data_dir = synthetic_data_dir(args.synthetic, args.seed)
            sys.argv += ["--data_dir", args.data_dir]
        sys.exit(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                 str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                                 os.path.abspath(__file__)] + sys.argv[1:]).returncode)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.manual_seed(args.seed)
    dev = torch.device("cuda", int(os.environ.get("KGAT_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("KGAT_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    say = print if rank == 0 else (lambda *a, **k: None)
    ds = ckg_io.CKGDataset(args.data_dir or synthetic_data_dir(args.synthetic, args.seed))
    say("users %d items %d | CKG: %d entities, %d relations, %d train triplets" % (
        ds.n_users, ds.n_items, ds.n_KG_entity, ds.n_KG_relation, len(ds.train_KG_triplet)))
    model = K.KGATPropagation(ds.n_KG_entity, ds.n_KG_relation, args.entity_embed_dim, args.relation_embed_dim,
                              args.gnn_num_layer, args.gnn_hidden_size, args.dropout_rate).to(dev)
    K.enable_lazy_edge_weights()   # the attention refresh hands back its edge-id-ordered copy unwritten (nothing here reads it)

    def replicas_agree(tag):
        """Every rank holds a replica of the parameters and runs the same optimiser on the same gradients
        (only the shard layers' partial gradients are summed across ranks): a nondeterministic kernel or a
        diverging per-rank RNG would make the replicas drift silently.  Compare a checksum of all
        parameters across ranks (MAX - MIN of the per-rank sums) and stop when they differ."""
        if world == 1:
            return
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        if float(hi - lo) > 1e-9 * max(abs(float(hi)), 1.0):
            raise RuntimeError("%s: parameter replicas differ across ranks (checksum spread %.3e)" % (tag, float(hi - lo)))
    if world > 1:
        for p in model.parameters():   # one source of truth for the initial parameters, whatever the ranks' RNG drew
            dist.broadcast(p.data, src=0)
        replicas_agree("after the initial broadcast")
    # one Adam over all parameters (kgat.py:85): the same update, bit for bit, as one launch per step (optim.FusedAdam)
    opt = K.FusedAdam(model.parameters(), lr=args.lr)
    train_g, test_g = ds.train_graph(dev), ds.test_graph(dev)
    if world > 1:
        from dgl_kgat_amd import partition
        train_g, test_g = partition.shard_graph(train_g, rank, world)[0], partition.shard_graph(test_g, rank, world)[0]
    trip = torch.as_tensor(ds.train_KG_triplet.astype(np.int32), device=dev)   # int32 ids go to the kernels as they are
    pairs = torch.as_tensor(ds.train_pairs.astype(np.int32), device=dev)
    off = ds.n_users
    train_dict = user_dict(ds.train_pairs, off)
    valid_dict, test_dict = user_dict(ds.valid_pairs, off), user_dict(ds.test_pairs, off)
    train_valid_dict = user_dict(np.vstack([ds.train_pairs, ds.valid_pairs]), off)

    def cap(n):
        return n if args.max_iters <= 0 else min(n, args.max_iters)

    for epoch in range(1, args.epochs + 1):
        # ---- KG phase (kgat.py:116-136)
        t0 = time.time()
        model.train()
        total, n_it = 0.0, cap(len(trip) // args.batch_size_kg + 1)
        for _ in range(n_it):
            idx = torch.randint(0, len(trip), (min(args.batch_size_kg, len(trip)),), device=dev)
            h, r, pos_t = trip[idx, 0], trip[idx, 1], trip[idx, 2]
            neg_t = torch.randint(0, ds.n_KG_entity, h.shape, device=dev, dtype=torch.int32)
            # transR -> backward -> step -> zero_grad of kgat.py:127-131 as two library calls (same bits)
            loss = model.kg_step(h.contiguous(), r.contiguous(), pos_t.contiguous(), neg_t, opt)
            total += loss.item()
        say("Epoch %04d | KGE %.1fs loss %.4f" % (epoch, time.time() - t0, total / n_it))
        # ---- attention refresh (kgat.py:139-145)
        t0 = time.time()
        with torch.no_grad():
            train_g.edata["w"] = model.compute_attention(train_g)
        torch.cuda.synchronize()
        say("           | attention %.4fs" % (time.time() - t0))
        # ---- CF phase (kgat.py:146-168): full-graph gnn for every batch
        t0 = time.time()
        total, n_it = 0.0, cap(len(pairs) // args.batch_size + 1)
        for _ in range(n_it):
            idx = torch.randint(0, len(pairs), (min(args.batch_size, len(pairs)),), device=dev)
            users, pos_items = pairs[idx, 0], pairs[idx, 1]
            neg_items = torch.randint(off, off + ds.n_items, users.shape, device=dev, dtype=torch.int32)
            emb = model.gnn(train_g, train_g.ndata["id"])
            loss = model.get_loss(emb, users, pos_items, neg_items)
            loss.backward()
            if args.grad_digest and epoch == 1 and _ == 0:
                say("           | grad digest: loss %.9g  " % loss.item() + "  ".join(
                    "%s %.9g" % (k, p.grad.double().abs().sum().item()) for k, p in model.named_parameters()
                    if p.grad is not None))
            opt.step()
            opt.zero_grad()
            total += loss.item()
        say("           | GNN %.1fs loss %.4f" % (time.time() - t0, total / n_it))
        replicas_agree("epoch %d" % epoch)
        # ---- evaluation (kgat.py:53-62, 171-196)
        t0 = time.time()
        with torch.no_grad():
            for name, g, seen, held in (("valid", train_g, train_dict, valid_dict), ("test", test_g, train_valid_dict, test_dict)):
                g.edata["w"] = model.compute_attention(g)
                emb = model.gnn(g, g.ndata["id"])
                rec, ndcg = metrics.calc_recall_ndcg(emb, seen, held, ds.item_id_range, K=20)
                say("           | %s recall@20 %.5f ndcg@20 %.5f" % (name, rec, ndcg))
            train_g.edata["w"] = model.compute_attention(train_g)
        say("           | eval %.2fs" % (time.time() - t0))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
