#!/usr/bin/env python3
"""Minimal KGAT training / evaluation harness over this package: the epoch
structure of the reference's ``kgat.py:114-196`` - KG phase (TransR), attention refresh under
no_grad, CF phase (full-graph ``gnn`` + BPR loss per batch), evaluation (recall@20 / ndcg@20
on the validation and test interactions) - with the propagation path running on the HIP
kernels.  Samplers are the reference's "uniform" modes (uniform positive edge, uniformly random
negative) drawn with torch on the device instead of DGL's C++ EdgeSampler: a phase's batches are drawn
up front (one gather per id column), the KG phase runs as ``KGATPropagation.kg_phase`` (one sort
launch for all batches + three launches per iteration), losses are summed on the device and read
once per phase (the reference reads ``loss.item()`` every step), and every phase is timed with a
host clock between two synchronisations (``--log_json`` keeps the per-epoch records).  Evaluation runs
in eval mode (the reference never leaves training mode, so its evaluation passes through dropout).

  python examples/train_kgat.py --data_dir datasets/amazon-book/data      # reference file format
  python examples/train_kgat.py --synthetic 1.0 --epochs 3               # amazon-book shape: 0.17 s per epoch
  python examples/train_kgat.py --planted --epochs 12 --lr 0.03 --batch_size 512 --batch_size_kg 512 --eval_before
                                                                          # planted structure: recall@20 must rise
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


def planted_data_dir(n_users=2000, n_items=1000, n_attrs=40, per_user=24, seed=1234):
    """A small CKG with PLANTED structure, written in the reference's file format: every item carries two
    attributes (KG triplets ``item -has-> attribute``), every user has one preferred attribute and draws 90 % of
    its interactions among the items that carry it (10 % uniformly).  A user's held-out items therefore share a KG
    neighbour with its training items: a model whose gradients, optimiser and attention refresh compose must push
    recall@20 well above the 20 / n_items of a random ranking within a few short epochs - the end-to-end check the
    kernels' stand-alone parity tests cannot give (tests/test_gpu_train_ops.py::test_planted_structure_recall_rises)."""
    rng = np.random.default_rng(seed)
    attr_of = np.stack([rng.integers(0, n_attrs, n_items), rng.integers(0, n_attrs, n_items)], 1)
    by_attr = [np.nonzero((attr_of == a).any(1))[0] for a in range(n_attrs)]
    taste = rng.integers(0, n_attrs, n_users)
    pairs = []
    for u in range(n_users):
        own = by_attr[taste[u]]
        k_own = min(int(round(0.9 * per_user)), len(own))
        items = np.concatenate([rng.choice(own, k_own, replace=False), rng.integers(0, n_items, per_user - k_own)])
        pairs.append(np.stack([np.full(len(items), u), items], 1))
    uv = np.unique(np.vstack(pairs), axis=0)
    rng.shuffle(uv)
    first = np.unique(np.concatenate([np.unique(uv[:, 0], return_index=True)[1], np.unique(uv[:, 1], return_index=True)[1]]))
    rest = np.setdiff1d(np.arange(len(uv)), first)
    n_hold = len(rest) // 8
    train, val, test = uv[np.concatenate([first, rest[2 * n_hold:]])], uv[rest[:n_hold]], uv[rest[n_hold:2 * n_hold]]
    items_seen = np.unique(train[:, 1])                 # item ids must be 0..n-1 over the training split
    remap = np.full(n_items, -1, np.int64)
    remap[items_seen] = np.arange(len(items_seen))
    keep = lambda a: a[remap[a[:, 1]] >= 0]             # noqa: E731
    fix = lambda a: np.stack([a[:, 0], remap[a[:, 1]]], 1)   # noqa: E731
    n_it = len(items_seen)
    kg = np.vstack([np.stack([remap[items_seen], np.zeros(n_it, np.int64), n_it + attr_of[items_seen, 0]], 1),
                    np.stack([remap[items_seen], np.ones(n_it, np.int64), n_it + attr_of[items_seen, 1]], 1)])
    ents = np.unique(kg[:, 2])                          # attribute entities numbered densely after the items
    kg[:, 2] = n_it + np.searchsorted(ents, kg[:, 2])
    d = os.path.join(tempfile.mkdtemp(prefix="kgat_planted_"), "data")
    ckg_io.save_ckg_files(d, n_users, fix(train), fix(keep(val)), fix(keep(test)), np.unique(kg, axis=0))
    return d


def user_dict(pairs, item_offset):
    order = np.argsort(pairs[:, 0], kind="stable")
    p = pairs[order]
    users, start = np.unique(p[:, 0], return_index=True)
    return {int(u): p[s:e, 1] - item_offset for u, s, e in zip(users, start, list(start[1:]) + [len(p)])}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_dir", default=None)
    ap.add_argument("--synthetic", type=float, default=0.01, help="scale of the synthetic amazon-book-shaped CKG")
    ap.add_argument("--planted", action="store_true", help="the small planted-structure CKG (planted_data_dir)")
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
    ap.add_argument("--eval_before", action="store_true", help="evaluate the untrained model first (epoch 0)")
    ap.add_argument("--log_json", default=None, help="write the per-epoch records (phase wall-clock, losses, metrics) here")
    ap.add_argument("--grad_digest", action="store_true",
                    help="print |grad| sums of the first CF step (to compare an N-GPU run with the one-GPU run)")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        argv = list(sys.argv[1:] if argv is None else argv)
        if args.data_dir is None:   # every rank must read the same files
            args.data_dir = planted_data_dir(seed=args.seed) if args.planted else synthetic_data_dir(args.synthetic, args.seed)
            argv += ["--data_dir", args.data_dir]
        sys.exit(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                 str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                                 os.path.abspath(__file__)] + argv).returncode)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.manual_seed(args.seed)
    dev = torch.device("cuda", int(os.environ.get("KGAT_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("KGAT_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    say = print if rank == 0 else (lambda *a, **k: None)
    if args.data_dir is None:
        args.data_dir = planted_data_dir(seed=args.seed) if args.planted else synthetic_data_dir(args.synthetic, args.seed)
    ds = ckg_io.CKGDataset(args.data_dir)
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
    # the sampled tables column by column (int32 ids go to the kernels as they are): a phase's batches are rows of
    # ONE gather per column, so a step's host work is the two library calls, not five indexing launches
    trip_cols = torch.as_tensor(np.ascontiguousarray(ds.train_KG_triplet.T.astype(np.int32)), device=dev)
    pair_cols = torch.as_tensor(np.ascontiguousarray(ds.train_pairs[:, :2].T.astype(np.int32)), device=dev)
    n_trip, n_pairs = trip_cols.shape[1], pair_cols.shape[1]
    off = ds.n_users
    train_dict = user_dict(ds.train_pairs, off)
    valid_dict, test_dict = user_dict(ds.valid_pairs, off), user_dict(ds.test_pairs, off)
    train_valid_dict = user_dict(np.vstack([ds.train_pairs, ds.valid_pairs]), off)
    # the evaluation's static side (users, item lists as CSR arrays) once per split, not once per call
    plans = {"valid": metrics.EvalPlan(train_dict, valid_dict, ds.item_id_range, dev),
             "test": metrics.EvalPlan(train_valid_dict, test_dict, ds.item_id_range, dev)}

    def cap(n):
        return n if args.max_iters <= 0 else min(n, args.max_iters)

    def clock():
        torch.cuda.synchronize()
        return time.perf_counter()

    def evaluate(rec):
        t0 = clock()
        with torch.no_grad():
            for name, g, seen, held in (("valid", train_g, train_dict, valid_dict), ("test", test_g, train_valid_dict, test_dict)):
                g.edata["w"] = model.compute_attention(g)
                emb = model.gnn(g, g.ndata["id"])
                rec[name + "_recall"], rec[name + "_ndcg"] = metrics.calc_recall_ndcg(
                    emb, seen, held, ds.item_id_range, K=20, plan=plans[name])
                say("           | %s recall@20 %.5f ndcg@20 %.5f" % (name, rec[name + "_recall"], rec[name + "_ndcg"]))
        rec["eval_s"] = clock() - t0
        say("           | eval %.4fs" % rec["eval_s"])

    history = []
    if args.eval_before:
        model.eval()
        history.append({"epoch": 0})
        say("Epoch 0000 | (untrained)")
        evaluate(history[-1])
    for epoch in range(1, args.epochs + 1):
        rec = {"epoch": epoch}
        t_epoch = clock()
        # ---- KG phase (kgat.py:116-136).  The reference reads loss.item() after every step (a host round trip per
        # iteration, kgat.py:132); here the running sum stays on the device and is read once per phase.
        t0 = clock()
        model.train()
        n_it, bs = cap(n_trip // args.batch_size_kg + 1), min(args.batch_size_kg, n_trip)
        idx = torch.randint(0, n_trip, (n_it, bs), device=dev)
        h_all, r_all, t_all = trip_cols[0][idx], trip_cols[1][idx], trip_cols[2][idx]
        neg_all = torch.randint(0, ds.n_KG_entity, (n_it, bs), device=dev, dtype=torch.int32)
        # every iteration = transR -> backward -> step -> zero_grad of kgat.py:127-131: one launch sorts all batches,
        # then one library call (three launches) per iteration; the losses stay on the device (kg_phase)
        total = model.kg_phase(h_all, r_all, t_all, neg_all, opt).sum()
        rec["kg_s"], rec["kg_iters"] = clock() - t0, n_it
        rec["kg_loss"] = float(total) / n_it
        say("Epoch %04d | KGE %.4fs (%d it, %.4f ms/it) loss %.4f" % (epoch, rec["kg_s"], n_it, 1e3 * rec["kg_s"] / n_it,
                                                                     rec["kg_loss"]))
        del idx, h_all, r_all, t_all, neg_all
        # ---- attention refresh (kgat.py:139-145)
        t0 = clock()
        with torch.no_grad():
            train_g.edata["w"] = model.compute_attention(train_g)
        rec["attention_s"] = clock() - t0
        say("           | attention %.4fs" % rec["attention_s"])
        # ---- CF phase (kgat.py:146-168): full-graph gnn for every batch
        t0 = clock()
        n_it, bs = cap(n_pairs // args.batch_size + 1), min(args.batch_size, n_pairs)
        idx = torch.randint(0, n_pairs, (n_it, bs), device=dev)
        u_all, p_all = pair_cols[0][idx], pair_cols[1][idx]
        n_all = torch.randint(off, off + ds.n_items, (n_it, bs), device=dev, dtype=torch.int32)
        total = torch.zeros((), dtype=torch.float32, device=dev)
        for i in range(n_it):
            emb = model.gnn(train_g, train_g.ndata["id"])
            loss = model.get_loss(emb, u_all[i], p_all[i], n_all[i])
            loss.backward()
            if args.grad_digest and epoch == 1 and i == 0:
                say("           | grad digest: loss %.9g  " % loss.item() + "  ".join(
                    "%s %.9g" % (k, p.grad.double().abs().sum().item()) for k, p in model.named_parameters()
                    if p.grad is not None))
            opt.step()
            opt.zero_grad()
            total += loss.detach()
        rec["cf_s"], rec["cf_iters"] = clock() - t0, n_it
        rec["cf_loss"] = float(total) / n_it
        say("           | GNN %.4fs (%d it, %.4f ms/it) loss %.4f" % (rec["cf_s"], n_it, 1e3 * rec["cf_s"] / n_it, rec["cf_loss"]))
        del idx, u_all, p_all, n_all
        replicas_agree("epoch %d" % epoch)
        # ---- evaluation (kgat.py:53-62, 171-196)
        model.eval()
        evaluate(rec)
        rec["epoch_s"] = clock() - t_epoch
        say("           | epoch %.4fs" % rec["epoch_s"])
        history.append(rec)
    if args.log_json and rank == 0:
        import json
        with open(args.log_json, "w") as f:
            json.dump({"args": vars(args), "n_users": ds.n_users, "n_items": ds.n_items, "n_entities": ds.n_KG_entity,
                       "n_relations": ds.n_KG_relation, "n_train_triplets": int(n_trip), "n_train_pairs": int(n_pairs),
                       "epochs": history}, f, indent=1)
    if world > 1:
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
