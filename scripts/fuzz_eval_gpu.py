#!/usr/bin/env python3
"""Developer tool: randomised exactness fuzzing of the evaluation kernel (kgat_eval_recall_ndcg_f32) - small-integer
embeddings (every fp32 score exact: the ranking is decided by the rule alone, ties included), random user / item counts,
readout widths (register form at 161..176 columns, LDS form otherwise), K, training / test lists (empty, everything
masked, long).  Checks the K ranked positions of every user against a stable descending sort and recall / ndcg against
the oracle.  Prints the first failure with its seed and exits non-zero.

  python scripts/fuzz_eval_gpu.py [seconds] [first_seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_kgat_amd import metrics, ops  # noqa: E402
from oracle import kgat_oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def one(seed):
    rng = np.random.default_rng(seed)
    F = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(161, 177)), 176, int(rng.integers(40, 361))]))
    K = int(rng.choice([1, 5, 20, 32, int(rng.integers(1, 33))]))
    n_i = int(rng.choice([int(rng.integers(K, K + 70)), int(rng.integers(200, 3000)), int(rng.integers(3000, 12000))]))
    n_i = max(n_i, K)
    n_u = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 400)), int(rng.integers(400, 1500))]))
    amp = int(rng.choice([1, 2, 3]))
    emb = rng.integers(-amp, amp + 1, (n_u + n_i, F)).astype(np.float64)
    emb[rng.random(n_u + n_i) < 0.05] = 0.0
    if n_i > 20:
        dup = rng.integers(n_u, n_u + n_i, 8)
        emb[dup] = emb[rng.integers(n_u, n_u + n_i, 8)]
    item_range = np.arange(n_u, n_u + n_i)
    train, test = {}, {}
    users = list(range(n_u))
    rng.shuffle(users)
    for u in users:
        mode = rng.random()
        n_tr = 0 if mode < 0.1 else (n_i if mode > 0.97 else int(rng.integers(0, min(n_i, 80))))
        train[u] = rng.choice(n_i, n_tr, replace=False)
        test[u] = rng.choice(n_i, int(rng.integers(0, min(n_i, 12))), replace=False)
    plan = metrics.EvalPlan(train, test, item_range, dev)
    e = torch.as_tensor(emb.astype(np.float32), device=dev)
    rec, ndcg, topk = ops.eval_recall_ndcg(e, plan.user_ids, plan.item_ids, plan.train_ptr, plan.train_items, plan.test_ptr,
                                           plan.test_items, K, want_topk=True)
    topk = topk.cpu().numpy().astype(np.int64)
    for r, u in enumerate(test.keys()):
        score = emb[item_range] @ emb[u]
        score[np.asarray(train[u], dtype=np.int64)] = 0.0
        exp = np.argsort(-score, kind="stable")[:K]
        if not np.array_equal(topk[r], exp):
            return "seed %d (n_u %d n_i %d F %d K %d): user %d ranks %s expected %s" % (seed, n_u, n_i, F, K, u, topk[r], exp)
    ref = orc.recall_ndcg_per_user(emb, train, test, item_range, K)
    if abs(float(rec.sum()) / n_u - ref[0]) > 1e-12 or abs(float(ndcg.sum()) / n_u - ref[1]) > 1e-12:
        return "seed %d: recall / ndcg %r %r vs oracle %r" % (seed, float(rec.sum()) / n_u, float(ndcg.sum()) / n_u, ref)
    return None


t0 = time.time()
seed = seed0
while time.time() - t0 < budget:
    msg = one(seed)
    if msg:
        print("FAIL", msg)
        sys.exit(1)
    seed += 1
print("eval fuzz ok: %d cases (seeds %d..%d) in %.0f s" % (seed - seed0, seed0, seed - 1, time.time() - t0))
