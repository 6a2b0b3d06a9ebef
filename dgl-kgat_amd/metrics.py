"""recall@K / ndcg@K of the reference (``metric.py:36-68``) on the device.

Same definitions, including the reference's quirks: training items are masked by setting their
score to 0.0 (not -inf), and the ideal DCG of a user is the DCG of that user's *own* hit list
sorted (``one_ndcg_at_k``), not the DCG of min(|positives|, K) leading ones.

``calc_recall_ndcg`` runs ``kgat_eval_recall_ndcg_f32`` (csrc/kgat_eval.hip): the user x item
scores come out of the fp32 MFMA tile by tile and go straight into a running top-K per user -
no (users x items) score matrix, no sort of 24,915 scores per user.  Equal scores rank by item
position (what the reference's stable CPU ``th.sort`` gives: among the masked zeros the lower
index first).  ``calc_recall_ndcg_sorted`` is the batched matmul + full stable sort of rounds
1-4, kept as an independent check of the kernel (tests); nothing dispatches to it.
"""
import numpy as np
import torch


_EMPTY = np.zeros(0, dtype=np.int64)


class EvalPlan:
    """The static part of an evaluation: test users, item node ids and the users' train / test item lists as CSR
    arrays on the device (positions ascending per user).  Build once per (train, test) split and pass it as
    ``plan=``; the dicts are read when the plan is built, never again.

    Train lists are de-duplicated (the reference masks ``score[train_items] = 0.0`` - an item listed twice is masked
    once, ``metric.py:50``); test lists keep their duplicates (``len(pos_item_l)`` counts them, ``metric.py:24``).
    Built with array operations over the concatenated lists: one dict lookup per user is the only per-user Python
    step (amazon-book: 650 k pairs, ~0.3 s; the per-user ``np.sort`` loops of round 5 took seconds)."""

    def __init__(self, train_user_dict, test_user_dict, all_item_id_range, device):
        users = list(test_user_dict.keys())
        self.n_users = len(users)
        items = np.asarray(all_item_id_range, dtype=np.int64).reshape(-1)
        self.n_items = len(items)
        n_items = max(self.n_items, 1)

        def csr(d, name, unique):
            lists = [d.get(u, _EMPTY) for u in users]
            lens = np.fromiter((len(x) for x in lists), dtype=np.int64, count=len(lists))
            total = int(lens.sum())
            if total >= 2 ** 31:
                raise ValueError("more than 2^31 (user, item) pairs")
            if total == 0:
                return (torch.zeros(len(users) + 1, dtype=torch.int32, device=device),
                        torch.zeros(0, dtype=torch.int32, device=device))
            flat = np.concatenate([np.asarray(x).reshape(-1) for x in lists if len(x)]).astype(np.int64, copy=False)
            owner = np.repeat(np.arange(len(users), dtype=np.int64), lens)
            bad = (flat < 0) | (flat >= self.n_items)
            if bad.any():
                # (the reference would index `score` out of range, metric.py:50)
                raise IndexError("%s items of user %r outside [0, %d)" % (name, users[int(owner[np.argmax(bad)])],
                                                                          self.n_items))
            key = owner * n_items + flat                  # one sort orders users and, within a user, positions
            key = np.unique(key) if unique else np.sort(key)
            owner = key // n_items
            ptr = np.zeros(len(users) + 1, dtype=np.int64)
            np.cumsum(np.bincount(owner, minlength=len(users)), out=ptr[1:])
            return (torch.as_tensor(ptr.astype(np.int32), device=device),
                    torch.as_tensor((key - owner * n_items).astype(np.int32), device=device))

        u = np.asarray(users, dtype=np.int64).reshape(-1)
        if u.size and (u.min() < 0 or u.max() >= 2 ** 31 or (items.size and (items.min() < 0 or items.max() >= 2 ** 31))):
            raise IndexError("user / item node ids must lie in [0, 2^31)")
        self.max_node_id = int(max(u.max() if u.size else -1, items.max() if items.size else -1))
        self.user_ids = torch.as_tensor(u.astype(np.int32), device=device)
        self.item_ids = torch.as_tensor(items.astype(np.int32), device=device)
        self.train_ptr, self.train_items = csr(train_user_dict, "train", True)
        self.test_ptr, self.test_items = csr(test_user_dict, "test", False)


def calc_recall_ndcg(embedding, train_user_dict, test_user_dict, all_item_id_range, K=20, plan=None,
                     return_per_user=False):
    """``embedding`` (N, F) node embeddings on the HIP device; the dicts map a user id to the array of its (raw,
    un-shifted) item ids; ``all_item_id_range`` the node ids of the items.  As the reference does
    (``metric.py:36-68``), the dicts are read on EVERY call - nothing is cached between calls; a caller that
    evaluates the same split repeatedly builds an ``EvalPlan`` once and passes it as ``plan=`` (the dicts are then
    ignored).

    Restrictions (no CPU / PyTorch fallback, by design): ``embedding`` must live on the HIP device
    (``KGATLibraryError`` otherwise), K <= 32; a float64 embedding is ranked in float32 - the kernel's
    arithmetic, and what the reference's fp32 model produces - so near-ties of a float64 input can rank
    differently than a float64 sort would rank them.  ``calc_recall_ndcg_sorted`` (torch operators) takes any K,
    dtype and device."""
    from . import ops
    if plan is None:
        plan = EvalPlan(train_user_dict, test_user_dict, all_item_id_range, embedding.device)
    if plan.n_users == 0:
        raise ZeroDivisionError("no test users")   # (metric.py:65 divides by len(test_user_dict))
    if plan.max_node_id >= embedding.shape[0]:
        # (the reference's embedding[user_id] / embedding[all_item_id_range] raises IndexError, metric.py:44-46)
        raise IndexError("user / item node id %d outside the embedding's %d rows" % (plan.max_node_id,
                                                                                    embedding.shape[0]))
    emb = embedding.detach()
    if emb.dtype != torch.float32:
        emb = emb.float()
    if emb.stride(1) != 1:
        emb = emb.contiguous()
    with torch.no_grad():
        recall, ndcg = ops.eval_recall_ndcg(emb, plan.user_ids, plan.item_ids, plan.train_ptr, plan.train_items,
                                            plan.test_ptr, plan.test_items, K)
    if return_per_user:
        return recall, ndcg
    return float(recall.sum()) / plan.n_users, float(ndcg.sum()) / plan.n_users


def calc_recall_ndcg_sorted(embedding, train_user_dict, test_user_dict, all_item_id_range, K=20, batch_users=2048):
    """The same metric by one matmul + one stable descending sort per user batch (torch operators; rounds 1-4)."""
    dev = embedding.device
    items = torch.as_tensor(np.asarray(all_item_id_range), device=dev, dtype=torch.long)
    item_emb = embedding.index_select(0, items).t().contiguous()
    users = list(test_user_dict.keys())
    n_items = items.numel()
    disc = 1.0 / torch.log2(torch.arange(2, K + 2, device=dev, dtype=torch.float64))
    recall_sum = ndcg_sum = 0.0
    with torch.no_grad():
        for lo in range(0, len(users), batch_users):
            ub = users[lo:lo + batch_users]
            u_idx = torch.as_tensor(np.asarray(ub), device=dev, dtype=torch.long)
            score = embedding.index_select(0, u_idx) @ item_emb  # (B, n_items)
            rows, cols, prow, pcol = [], [], [], []
            for b, u in enumerate(ub):
                tr = np.asarray(train_user_dict.get(u, ()), dtype=np.int64)
                rows.append(np.full(len(tr), b)); cols.append(tr)
                ps = np.asarray(test_user_dict[u], dtype=np.int64)
                prow.append(np.full(len(ps), b)); pcol.append(ps)
            score[torch.as_tensor(np.concatenate(rows), device=dev), torch.as_tensor(np.concatenate(cols), device=dev)] = 0.0
            pos = torch.zeros((len(ub), n_items), dtype=torch.bool, device=dev)
            pos[torch.as_tensor(np.concatenate(prow), device=dev), torch.as_tensor(np.concatenate(pcol), device=dev)] = True
            top = torch.sort(score, dim=1, descending=True, stable=True).indices[:, :K]
            hits = pos.gather(1, top).to(torch.float64)  # (B, K) binary_rank_K
            n_pos = torch.as_tensor([len(test_user_dict[u]) for u in ub], device=dev, dtype=torch.float64)
            recall_sum += float((hits.sum(1) / n_pos.clamp(min=1)).where(n_pos > 0, torch.zeros_like(n_pos)).sum())
            dcg = (hits * disc).sum(1)
            ideal = (torch.sort(hits, dim=1, descending=True).values * disc).sum(1)
            ndcg_sum += float(torch.where(ideal > 0, dcg / ideal.clamp(min=1e-300), torch.zeros_like(dcg)).sum())
    return recall_sum / len(users), ndcg_sum / len(users)
