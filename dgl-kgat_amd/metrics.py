"""recall@K / ndcg@K of the reference (``metric.py:36-68``), batched on the device.

Same definitions, including the reference's quirks: training items are masked by setting their
score to 0.0 (not -inf), and the ideal DCG of a user is the DCG of that user's *own* hit list
sorted (``one_ndcg_at_k``), not the DCG of min(|positives|, K) leading ones.  Users are scored
in batches with one matmul + one stable descending sort per batch instead of one sort per user
(the reference's ``th.sort`` is stable on the CPU: among equal scores - the masked zeros - the
lower item index ranks first; a top-K primitive would break such ties its own way).
"""
import numpy as np
import torch


def calc_recall_ndcg(embedding, train_user_dict, test_user_dict, all_item_id_range, K=20, batch_users=2048):
    """``embedding`` (N, F) node embeddings; the dicts map a user id to the array of its (raw,
    un-shifted) item ids; ``all_item_id_range`` the node ids of the items."""
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
