"""Evaluation kernel (kgat_eval_recall_ndcg_f32; reference metric.py:36-68) against the oracle's per-user
restatement of the reference's loop (oracle.recall_ndcg_per_user, pinned to the reference's own value for the toy
fixture in the CPU suite) and against exact integer cases where every tie is decided by the rule itself."""
import os

import numpy as np
import pytest
import torch

from oracle import kgat_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _as_dict(users, items):
    return {int(u): np.array([int(x) for x in str(s).split(";")]) for u, s in zip(users, items)}


def _expected_topk(emb, users, item_range, train, K):
    """The reference's ranking for exactly representable scores: fp32 matmul (exact on small integers), training
    items set to 0.0, stable descending sort."""
    e = np.asarray(emb, np.float64)
    out = np.zeros((len(users), K), np.int64)
    for r, u in enumerate(users):
        score = e[item_range] @ e[u]
        score[np.asarray(train.get(u, ()), dtype=np.int64)] = 0.0
        out[r] = np.argsort(-score, kind="stable")[:K]
    return out


def _run_kernel(emb, train, test, item_range, K, dev):
    from dgl_kgat_amd import metrics, ops
    plan = metrics.EvalPlan(train, test, item_range, dev)
    e = torch.as_tensor(np.asarray(emb, np.float32), device=dev)
    rec, ndcg, topk = ops.eval_recall_ndcg(e, plan.user_ids, plan.item_ids, plan.train_ptr, plan.train_items,
                                           plan.test_ptr, plan.test_items, K, want_topk=True)
    return rec.cpu().numpy(), ndcg.cpu().numpy(), topk.cpu().numpy(), plan


def test_eval_golden_fixture(dev):
    """The value the reference's own calc_recall_ndcg returned for the toy dataset (tests/golden/toy_dataset.npz)."""
    from conftest import GOLDEN_DIR
    from dgl_kgat_amd import metrics
    z = np.load(os.path.join(GOLDEN_DIR, "toy_dataset.npz"))
    train = _as_dict(z["train_users"], z["train_user_items"])
    test = _as_dict(z["test_users"], z["test_user_items"])
    emb = torch.as_tensor(z["metric_embedding"], device=dev)
    rec, ndcg = metrics.calc_recall_ndcg(emb, train, test, z["item_id_range"], K=5)
    assert abs(rec - z["metric_recall_ndcg_at5"][0]) < 1e-12 and abs(ndcg - z["metric_recall_ndcg_at5"][1]) < 1e-12
    rec2, ndcg2 = metrics.calc_recall_ndcg_sorted(emb, train, test, z["item_id_range"], K=5, batch_users=4)
    assert abs(rec2 - rec) < 1e-12 and abs(ndcg2 - ndcg) < 1e-12


def test_eval_vs_oracle_masked_items_and_empty_lists(dev):
    """300 users x 500 items, fp64 scores all distinct: the best raw scores are training items (masked to 0.0),
    some users have no test item (recall 0), hits that are not a prefix of the ranking (own-hit-list ideal DCG)."""
    from dgl_kgat_amd import metrics
    rng = np.random.default_rng(11)
    n_u, n_i, K = 300, 500, 20
    e = rng.standard_normal((n_u + n_i, 12))
    item_range = np.arange(n_u, n_u + n_i)
    train, test = {}, {}
    for u in range(n_u):
        score = e[item_range] @ e[u]
        top = np.argsort(-score)
        train[u] = top[:rng.integers(0, 8)]
        n_pos = 0 if u % 37 == 0 else int(rng.integers(1, 12))
        cand = np.concatenate([top[8:40], rng.integers(0, n_i, 20)])
        test[u] = np.unique(rng.choice(cand, n_pos, replace=False)) if n_pos else np.zeros(0, np.int64)
    ref = orc.recall_ndcg_per_user(e, train, test, item_range, K)
    got = metrics.calc_recall_ndcg(torch.as_tensor(e, device=dev), train, test, item_range, K=K)
    assert 0.05 < ref[0] < 0.95 and abs(got[0] - ref[0]) < 1e-12 and abs(got[1] - ref[1]) < 1e-12, (got, ref)
    got_s = metrics.calc_recall_ndcg_sorted(torch.as_tensor(e, device=dev), train, test, item_range, K=K, batch_users=64)
    assert abs(got_s[0] - ref[0]) < 1e-12 and abs(got_s[1] - ref[1]) < 1e-12


@pytest.mark.parametrize("n_u,n_i,F,K", [(70, 45, 13, 5), (33, 32, 8, 32), (5, 64, 7, 1), (260, 1500, 24, 20),
                                         (129, 2100, 176, 20), (64, 40, 200, 8), (40, 300, 360, 20),
                                         (150, 1900, 41, 20), (97, 2300, 96, 20), (200, 1500, 161, 9), (77, 2500, 256, 20),
                                         (60, 700, 257, 5), (70, 1800, 352, 20), (45, 2600, 300, 7)])
def test_eval_ties_and_masked_zeros_exact(dev, n_u, n_i, F, K):
    """Small-integer embeddings: every fp32 score is exact, so the ranking is decided by the rule alone - many equal
    scores (lower position first), negative K-th scores (the masked 0.0 of training items ranks), all-zero rows
    (every score 0.0), duplicate item rows; sizes off every tile (32 items, 32 users per wavefront), odd F, several
    item segments, K = 1 and K = 32 = n_items."""
    rng = np.random.default_rng(100 * n_u + n_i + F)
    emb = rng.integers(-2, 3, (n_u + n_i, F)).astype(np.float64)
    emb[rng.random(n_u + n_i) < 0.1] = 0.0                      # zero rows: every score of / with them is 0.0
    dup = rng.integers(n_u, n_u + n_i, 10)
    emb[dup] = emb[rng.integers(n_u, n_u + n_i, 10)]            # duplicate item rows: equal scores for every user
    neg = rng.random(n_u) < 0.3                                 # users whose scores are mostly negative
    item_range = np.arange(n_u, n_u + n_i)
    train, test = {}, {}
    users = list(range(n_u))
    rng.shuffle(users)                                          # the dict order is the evaluation order
    for u in users:
        if neg[u]:
            emb[u] = -np.sign(emb[item_range].sum(0))
        n_tr = int(rng.integers(0, min(n_i, 40)))
        train[u] = rng.choice(n_i, n_tr, replace=False)
        n_te = int(rng.integers(0, 9))
        test[u] = rng.choice(n_i, n_te, replace=False)
    if n_u > 3:
        train[users[0]] = np.arange(n_i)                        # every item masked: the K lowest positions rank
        train[users[1]] = np.zeros(0, np.int64)
    rec, ndcg, topk, plan = _run_kernel(emb, train, test, item_range, K, dev)
    exp = _expected_topk(emb, list(test.keys()), item_range, train, K)
    assert np.array_equal(topk.astype(np.int64), exp), np.argwhere(topk != exp)[:5]
    ref = orc.recall_ndcg_per_user(emb, train, test, item_range, K)
    assert abs(rec.sum() / n_u - ref[0]) < 1e-12 and abs(ndcg.sum() / n_u - ref[1]) < 1e-12
    # the same call again: bit for bit
    rec2, ndcg2, topk2, _ = _run_kernel(emb, train, test, item_range, K, dev)
    assert np.array_equal(rec, rec2) and np.array_equal(ndcg, ndcg2) and np.array_equal(topk, topk2)


@pytest.mark.parametrize("n_u,n_i,F,K", [(1100, 6000, 176, 20), (400, 9001, 176, 7), (500, 4000, 64, 20), (40, 30000, 176, 20),
                                         (600, 5000, 208, 20), (300, 7000, 120, 20), (300, 5000, 352, 20)])
def test_eval_real_valued_scores_and_racing_thresholds(dev, n_u, n_i, F, K):
    """Real-valued embeddings (fp32 scores, no exact ties) at sizes with a sample segment and several item segments per
    user block: the kernel's K best against an fp64 ranking of the same fp32 inputs - equal, or different only where two
    fp64 scores are closer than the fp32 rounding of a 176-term dot product -, and the same bits from call to call: the
    segments of a user race each other through the shared K-th best (tau_shared), which may change the work, never the
    result."""
    from dgl_kgat_amd import metrics, ops
    rng = np.random.default_rng(7 * n_u + n_i)
    emb = rng.standard_normal((n_u + n_i, F)).astype(np.float32)
    item_range = np.arange(n_u, n_u + n_i)
    train = {u: np.unique(rng.integers(0, n_i, int(rng.integers(0, 60)))) for u in range(n_u)}
    test = {u: np.unique(rng.integers(0, n_i, 1 + u % 7)) for u in range(n_u)}
    # a few users whose best items are all training items
    e64 = emb.astype(np.float64)
    for u in range(0, n_u, 97):
        train[u] = np.unique(np.argsort(-(e64[item_range] @ e64[u]))[:45])
    plan = metrics.EvalPlan(train, test, item_range, dev)
    e = torch.as_tensor(emb, device=dev)
    runs = [ops.eval_recall_ndcg(e, plan.user_ids, plan.item_ids, plan.train_ptr, plan.train_items, plan.test_ptr,
                                 plan.test_items, K, want_topk=True) for _ in range(4)]
    torch.cuda.synchronize()
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], r))
    rec, ndcg, topk = (x.cpu().numpy() for x in runs[0])
    users = list(test.keys())
    tol = 2e-4
    n_exact = 0
    for r, u in enumerate(users):
        score = e64[item_range] @ e64[u]
        score[train[u]] = 0.0
        order = np.argsort(-score, kind="stable")
        if np.array_equal(topk[r], order[:K]):
            n_exact += 1
            continue
        # not the fp64 order: every selected item within rounding of the K-th best, in descending order up to rounding
        got = score[topk[r]]
        assert np.all(got >= score[order[K - 1]] - tol), (u, got, score[order[:K]])
        assert np.all(np.diff(got) <= tol), (u, got)
        assert len(set(topk[r].tolist())) == K
    assert n_exact >= 0.995 * n_u, (n_exact, n_u)
    ref = orc.recall_ndcg_per_user(e64, train, test, item_range, K)
    assert abs(rec.mean() - ref[0]) < 2e-3 and abs(ndcg.mean() - ref[1]) < 2e-3, (rec.mean(), ndcg.mean(), ref)


def test_eval_rejects_what_the_reference_cannot_do(dev):
    from dgl_kgat_amd import metrics, ops
    from dgl_kgat_amd.ops import KGATLibraryError
    emb = torch.zeros((20, 8), device=dev)
    train, test = {0: np.array([1])}, {0: np.array([2])}
    with pytest.raises(KGATLibraryError):   # K > 32
        metrics.calc_recall_ndcg(emb, train, test, np.arange(4, 20), K=33)
    with pytest.raises(KGATLibraryError):   # fewer items than K (the reference reads rank_indices[K - 1])
        metrics.calc_recall_ndcg(emb, train, test, np.arange(4, 8), K=5)
    with pytest.raises(IndexError):         # an item id outside the item range
        metrics.calc_recall_ndcg(emb, {0: np.array([99])}, test, np.arange(4, 20), K=5)
    with pytest.raises(KGATLibraryError):   # CPU tensors: no CPU implementation
        metrics.calc_recall_ndcg(emb.cpu(), train, test, np.arange(4, 20), K=5)
    assert ops.eval_supported(176, 20) and not ops.eval_supported(176, 0)


def test_eval_full_size_amazon_book_shape(dev):
    """70,679 users x 24,915 items x 176 columns (the readout of the 64-64-32-16 model), K = 20: exact integer
    scores, sampled users checked rank for rank; the launch timed (task 5: <= 15 ms)."""
    from dgl_kgat_amd import metrics, ops
    n_u, n_i, F, K = 70679, 24915, 176, 20
    g = torch.Generator(device="cpu").manual_seed(5)
    emb = torch.randint(-3, 4, (n_u + n_i, F), generator=g).float()
    rng = np.random.default_rng(6)
    item_range = np.arange(n_u, n_u + n_i)
    deg = np.minimum(rng.zipf(1.6, n_u) + 1, 3000)
    train = {u: np.unique(rng.integers(0, n_i, deg[u])) for u in range(n_u)}
    test = {u: np.unique(rng.integers(0, n_i, 1 + (u % 5))) for u in range(n_u)}
    plan = metrics.EvalPlan(train, test, item_range, dev)
    e = emb.to(dev)
    rec, ndcg, topk = ops.eval_recall_ndcg(e, plan.user_ids, plan.item_ids, plan.train_ptr, plan.train_items,
                                           plan.test_ptr, plan.test_items, K, want_topk=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.eval_recall_ndcg(e, plan.user_ids, plan.item_ids, plan.train_ptr, plan.train_items, plan.test_ptr,
                             plan.test_items, K)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("[eval] 70,679 users x 24,915 items x 176, K = 20: %.2f ms (%.1f TFLOP/s fp32)"
          % (min(ts), 2.0 * n_u * n_i * F / min(ts) / 1e9))
    sample = rng.choice(n_u, 200, replace=False)
    sample[:3] = np.argsort(-deg)[:3]                            # the longest training lists too
    exp = _expected_topk(emb.numpy(), list(sample), item_range, train, K)
    assert np.array_equal(topk.cpu().numpy()[sample].astype(np.int64), exp)
    hits = np.array([[int(i in set(test[u].tolist())) for i in exp[r]] for r, u in enumerate(sample)], np.float64)
    disc = 1.0 / np.log2(np.arange(2, K + 2))
    r_exp = hits.sum(1) / np.array([len(test[u]) for u in sample])
    assert np.allclose(rec.cpu().numpy()[sample], r_exp, rtol=0, atol=1e-15)
    ideal = np.array([disc[:int(h.sum())].sum() for h in hits])
    n_exp = np.where(ideal > 0, (hits * disc).sum(1) / np.maximum(ideal, 1e-300), 0.0)
    assert np.allclose(ndcg.cpu().numpy()[sample], n_exp, rtol=0, atol=1e-14)
    assert min(ts) <= 15.0, ts
