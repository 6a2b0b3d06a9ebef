"""CPU oracle for KGAT's attentive embedding-propagation path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the algorithm the reference runs for the hot
path.  It is the checker for the HIP kernels, never the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package (``dgl-kgat_amd/``)
must not import anything from ``oracle/``.

Pinning status
--------------
* The dense arithmetic of the path is the reference's own torch code
  (``/root/reference/models.py``).  ``tests/golden/make_golden.py`` imports that
  file (with a stand-in ``dgl`` module, in the authoring container only) and
  stores its outputs as fixtures; ``tests/test_oracle_golden.py`` checks this
  oracle against them.  That part is pinned.
* The sparse arithmetic (``update_all(u_mul_e, sum)``, ``edge_softmax``,
  ``filter_edges``/``apply_edges`` frame semantics, COO->CSR) lives in the
  third-party ``dgl`` package (0.4.1-0.4.3 by API usage; not vendored, not
  pinned, not installable here).  It is restated from DGL's published
  semantics, cross-checked by an independent dense formulation
  (``dense_*`` below), but **parity for the DGL-internal ops is unpinned** by
  any reference-owned test or golden vector (the reference has none).

Conventions (reference ``dataset.py:112-120``): a triplet row ``[h, r, t]`` makes
edge ``e`` with ``src[e] = t``, ``dst[e] = h``, ``etype[e] = r``; edge id = row
index.  All per-edge arrays handed in or out are in edge-id order.
"""
import numpy as np

__all__ = [
    "csr_from_coo", "group_by_relation", "head_groups", "att_score", "edge_softmax",
    "spmm_u_mul_e_sum", "spmm_u_mul_e_sum_sparse", "spmm_backward_x", "sddmm_dot", "bi_interaction",
    "l2_normalize", "gnn_forward", "compute_attention",
    "dense_spmm", "dense_edge_softmax", "edge_softmax_backward", "recall_ndcg_per_user",
]


# --------------------------------------------------------------------------- graph
def csr_from_coo(n_nodes, src, dst):
    """COO -> CSR grouped by destination, stable in edge id.

    Follows reference ``dataset.py:112-120`` (``g.add_edges(t, h)``: edge id = row
    order) and DGL's in-CSR construction: row ``v`` lists the edges whose
    ``dst == v``; within a row edge ids ascend.  Returns ``(indptr[N+1], col[E],
    eid[E])`` as int32 where ``col`` holds source ids.
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    if src.shape != dst.shape or src.ndim != 1:
        raise ValueError("src/dst must be 1-D and equal length")
    if src.size and (src.min() < 0 or dst.min() < 0 or
                     src.max() >= n_nodes or dst.max() >= n_nodes):
        raise ValueError("node id out of range")
    eid = np.argsort(dst, kind="stable")
    counts = np.bincount(dst, minlength=n_nodes)
    indptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr.astype(np.int32), src[eid].astype(np.int32), eid.astype(np.int32)


def group_by_relation(etype, n_rel):
    """Edges grouped by relation id, stable in edge id.

    Replaces the ``R`` full-graph ``filter_edges`` sweeps of reference
    ``models.py:149-150``.  Edges whose type is outside ``[0, n_rel)`` are never
    visited by the reference loop; they are placed after ``rel_ptr[n_rel]``.
    Returns ``(rel_ptr[n_rel+1], perm[E])`` int32.
    """
    etype = np.asarray(etype, dtype=np.int64)
    key = np.where((etype >= 0) & (etype < n_rel), etype, n_rel)
    perm = np.argsort(key, kind="stable")
    counts = np.bincount(key, minlength=n_rel + 1)[:n_rel]
    rel_ptr = np.zeros(n_rel + 1, dtype=np.int64)
    np.cumsum(counts, out=rel_ptr[1:])
    return rel_ptr.astype(np.int32), perm.astype(np.int32)


def head_groups(rel_ptr, dst_g):
    """Runs of equal (relation, destination) in a relation-grouped edge list whose relations
    are sorted by destination: (gid[E], gptr[R+1], g_node[n_groups]).  Structure used by the
    split attention kernels to evaluate ``tanh(ent[h] W_r + e_r)`` (reference models.py:141-143)
    once per (head, relation) instead of once per edge; only positions below rel_ptr[R] count."""
    rel_ptr = np.asarray(rel_ptr, np.int64)
    dst_g = np.asarray(dst_g, np.int64)
    n_rel, scored = len(rel_ptr) - 1, int(rel_ptr[-1])
    flag = np.zeros(len(dst_g) + 1, np.int64)
    if scored:
        flag[0] = 1
        flag[1:scored] = dst_g[1:scored] != dst_g[:scored - 1]
        starts = rel_ptr[:-1][rel_ptr[:-1] < rel_ptr[1:]]
        flag[starts] = 1
    inc = np.cumsum(flag)
    gid = (inc[:len(dst_g)] - 1).astype(np.int32)
    ex = np.concatenate([[0], inc])
    gptr = ex[rel_ptr].astype(np.int32)
    heads = np.nonzero(flag[:scored])[0]
    return gid, gptr, dst_g[heads].astype(np.int32)


# ----------------------------------------------------------------------- attention
def att_score(ent, W_R, rel, src, dst, etype, dtype=np.float64):
    """TransR-style attention logits, reference ``models.py:135-154``.

    For every relation ``r`` in ``range(R)`` (``:149``) and every edge of that type
    (``:150``): ``t_r = ent[src] @ W_R[r]`` (``:140``), ``h_r = ent[dst] @ W_R[r]``
    (``:141``), ``att = sum_j t_r[j] * tanh(h_r[j] + rel[r][j])`` (``:142-143``).
    Edges whose type is outside ``[0, R)`` keep logit 0 (DGL zero-initialises the
    ``att_w`` column on the first partial ``apply_edges`` write).  Returns ``(E,)``.
    """
    ent = np.asarray(ent, dtype=dtype)
    W_R = np.asarray(W_R, dtype=dtype)
    rel = np.asarray(rel, dtype=dtype)
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    etype = np.asarray(etype, dtype=np.int64)
    out = np.zeros(src.shape[0], dtype=dtype)
    for r in range(W_R.shape[0]):
        idx = np.nonzero(etype == r)[0]
        if idx.size == 0:
            continue
        t_r = ent[src[idx]] @ W_R[r]
        h_r = ent[dst[idx]] @ W_R[r]
        out[idx] = np.sum(t_r * np.tanh(h_r + rel[r][None, :]), axis=1)
    return out


def edge_softmax(n_nodes, dst, logits, dtype=np.float64):
    """Softmax over the incoming edges of each destination node.

    Call site reference ``models.py:153``; semantics of
    ``dgl.nn.pytorch.softmax.edge_softmax`` (DGL 0.4.x): ``smax[v] = max_{e->v}
    s[e]``; ``x[e] = exp(s[e] - smax[dst e])``; ``z[v] = sum_{e->v} x[e]``;
    ``a[e] = x[e] / z[dst e]``.  All relations share one group per destination.
    Trailing feature dimensions are kept (DGL passes ``(E, 1)``).
    """
    s = np.asarray(logits, dtype=dtype)
    dst = np.asarray(dst, dtype=np.int64)
    flat = s.reshape(s.shape[0], -1)
    smax = np.full((n_nodes, flat.shape[1]), -np.inf, dtype=dtype)
    np.maximum.at(smax, dst, flat)
    x = np.exp(flat - smax[dst])
    z = np.zeros((n_nodes, flat.shape[1]), dtype=dtype)
    np.add.at(z, dst, x)
    return (x / z[dst]).reshape(s.shape)


def edge_softmax_backward(n_nodes, dst, a, grad_a, dtype=np.float64):
    """Backward of edge_softmax (DGL 0.4.x ``EdgeSoftmax.backward``):
    ``grad_s = a*grad_a - a * sum_{e'->dst(e)} a[e']*grad_a[e']``."""
    a = np.asarray(a, dtype=dtype).reshape(len(dst), -1)
    g = np.asarray(grad_a, dtype=dtype).reshape(len(dst), -1)
    dst = np.asarray(dst, dtype=np.int64)
    sds = a * g
    acc = np.zeros((n_nodes, a.shape[1]), dtype=dtype)
    np.add.at(acc, dst, sds)
    return sds - a * acc[dst]


def compute_attention(n_nodes, src, dst, etype, ent, W_R, rel, dtype=np.float64):
    """``Model.compute_attention`` (reference ``models.py:146-154``): logits per
    relation, then one edge_softmax over all edges.  Returns ``(E, 1)``."""
    s = att_score(ent, W_R, rel, src, dst, etype, dtype=dtype)
    return edge_softmax(n_nodes, dst, s[:, None], dtype=dtype)


# --------------------------------------------------------------------- aggregation
def spmm_u_mul_e_sum(n_nodes, src, dst, X, w, dtype=np.float64):
    """``update_all(fn.u_mul_e('h','w','m'), fn.sum('m','h_neighbor'))``,
    reference ``models.py:63``: ``out[v,:] = sum_{e:u->v} w[e] * X[u,:]``; ``w`` is
    ``(E,)`` or ``(E,1)`` broadcast over features; zero in-degree rows are 0."""
    X = np.asarray(X, dtype=dtype)
    w = np.asarray(w, dtype=dtype).reshape(-1)
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    out = np.zeros((n_nodes, X.shape[1]), dtype=dtype)
    np.add.at(out, dst, X[src] * w[:, None])
    return out


def spmm_u_mul_e_sum_sparse(n_nodes, src, dst, X, w, dtype=np.float64):
    """The same aggregation as ``spmm_u_mul_e_sum`` through scipy's sparse product (the
    (dst, src) weights assembled into a CSR matrix - parallel edges add up, the map is linear in
    ``w`` - then ``A @ X``): an independently written formulation that finishes in seconds at
    the benchmark graphs' sizes, where the ``np.add.at`` form above takes minutes."""
    import scipy.sparse as sp
    w = np.asarray(w, dtype=dtype).reshape(-1)
    A = sp.coo_matrix((w, (np.asarray(dst, np.int64), np.asarray(src, np.int64))),
                      shape=(n_nodes, n_nodes)).tocsr()
    return np.asarray(A @ np.asarray(X, dtype=dtype))


def spmm_backward_x(n_nodes, src, dst, grad_out, w, dtype=np.float64):
    """Autograd of ``models.py:63`` w.r.t. ``h``: ``grad_X[u,:] = sum_{e:u->v}
    w[e] * grad_out[v,:]`` (the same SpMM on the reversed graph)."""
    return spmm_u_mul_e_sum(n_nodes, dst, src, grad_out, w, dtype=dtype)


def sddmm_dot(src, dst, X, grad_out, dtype=np.float64):
    """Autograd of ``models.py:63`` w.r.t. ``w``: ``grad_w[e] = <X[src e],
    grad_out[dst e]>`` (unused by the reference loop: ``w`` is made under
    ``no_grad``, ``kgat.py:142-144``; provided for operator completeness)."""
    X = np.asarray(X, dtype=dtype)
    g = np.asarray(grad_out, dtype=dtype)
    return np.sum(X[np.asarray(src, np.int64)] * g[np.asarray(dst, np.int64)], axis=1)


def bi_interaction(h, h_neighbor, W2, negative_slope=0.01, dtype=np.float64):
    """``KGATConv.forward`` dense part, reference ``models.py:66``:
    ``leaky_relu(res_fc_2(h * h_neighbor))`` with ``res_fc_2.weight = W2``
    ``(D_out, D_in)``, no bias; dropout omitted (p = 0 for parity)."""
    z = (np.asarray(h, dtype) * np.asarray(h_neighbor, dtype)) @ np.asarray(W2, dtype).T
    return np.where(z >= 0, z, z * negative_slope)


def l2_normalize(x, eps=1e-12):
    """``F.normalize(h, p=2, dim=1)`` (reference ``models.py:165``)."""
    n = np.sqrt(np.sum(x * x, axis=1, keepdims=True))
    return x / np.maximum(n, eps)


def gnn_forward(n_nodes, src, dst, w, ent, W2_list, dtype=np.float64, spmm=None):
    """``Model.gnn`` (reference ``models.py:156-168``) with KGATConv layers
    (``:60-70``): ``h0 = ent``; per layer ``h = leaky_relu((h*h_N) W2^T)``;
    the cache gets ``normalize(h)`` while the un-normalised ``h`` feeds the next
    layer; output is the concatenation ``[h0, n(h1), ...]``."""
    h = np.asarray(ent, dtype=dtype)
    cache = [h]
    spmm = spmm or spmm_u_mul_e_sum
    for W2 in W2_list:
        h_n = spmm(n_nodes, src, dst, h, w, dtype=dtype)
        h = bi_interaction(h, h_n, W2, dtype=dtype)
        cache.append(l2_normalize(h))
    return np.concatenate(cache, axis=1)


# ------------------------------------------------ independent dense formulations
def dense_spmm(n_nodes, src, dst, X, w, dtype=np.float64):
    """Cross-check: ``out = S_dst^T (w * (S_src X))`` with one-hot incidence
    matrices.  O(E*N) memory - toy sizes only."""
    E = len(src)
    S_src = np.zeros((E, n_nodes), dtype=dtype)
    S_dst = np.zeros((E, n_nodes), dtype=dtype)
    S_src[np.arange(E), src] = 1
    S_dst[np.arange(E), dst] = 1
    return S_dst.T @ (np.asarray(w, dtype).reshape(-1, 1) * (S_src @ np.asarray(X, dtype)))


def dense_edge_softmax(n_nodes, dst, logits, dtype=np.float64):
    """Cross-check: per-destination python loop, no scatter primitives."""
    s = np.asarray(logits, dtype=dtype).reshape(-1)
    out = np.zeros_like(s)
    dst = np.asarray(dst)
    for v in range(n_nodes):
        idx = np.nonzero(dst == v)[0]
        if idx.size == 0:
            continue
        x = np.exp(s[idx] - s[idx].max())
        out[idx] = x / x.sum()
    return out


def fold_tiles(rel_ptr, gid, gptr, cap, groups_per_tile=16):
    """Work tiles of the fused attention kernel, restated (include/kgat_hip.h kgat_fold_tiles):
    per relation, blocks of 16 (or 32) consecutive head groups; a block spanning more than `cap` grouped
    positions is cut into consecutive position ranges of `cap`.  Returns (tiles (T,4) int32 =
    (relation, first group, first position, end position), rel_tptr (R+1,))."""
    rel_ptr, gid, gptr = (np.asarray(x, dtype=np.int64) for x in (rel_ptr, gid, gptr))
    n_rel, n_groups, n_scored = len(rel_ptr) - 1, int(gptr[-1]), int(rel_ptr[-1])
    gstart = np.full(n_groups + 1, n_scored, dtype=np.int64)
    if n_scored:
        first = np.nonzero(np.r_[True, gid[1:n_scored] != gid[:n_scored - 1]])[0]
        gstart[gid[first]] = first
    tiles, rel_tptr = [], [0]
    for r in range(n_rel):
        for g0 in range(int(gptr[r]), int(gptr[r + 1]), groups_per_tile):
            g1 = min(g0 + groups_per_tile, int(gptr[r + 1]))
            pb, pe = int(gstart[g0]), int(gstart[g1])
            p = pb
            while True:
                q = min(p + cap, pe)
                tiles.append((r, g0, p, q))
                p = q
                if p >= pe:
                    break
        rel_tptr.append(len(tiles))
    return np.asarray(tiles, dtype=np.int32).reshape(-1, 4), np.asarray(rel_tptr, dtype=np.int32)


def fold_tile_parts(tiles, rel_tptr, n_parts, cost):
    """Cost-balanced split of the fused attention kernel's tiles over its workgroups, restated
    (include/kgat_hip.h kgat_fold_tile_parts): tile cost ``c_tile + c_chunk * ceil(max(P - 64, 0) / 64)
    + (c_rel if the tile opens its relation)`` with P the tile's positions; part b starts at the
    first tile whose exclusive cost prefix reaches ``total * b // n_parts``.  Returns part_tptr
    (n_parts+1,)."""
    tiles = np.asarray(tiles, dtype=np.int64).reshape(-1, 4)
    rel_tptr = np.asarray(rel_tptr, dtype=np.int64)
    c_tile, c_chunk, c_rel = cost
    P = tiles[:, 3] - tiles[:, 2]
    later = np.where(P > 64, (P - 64 + 63) // 64, 0)
    opens = rel_tptr[tiles[:, 0]] == np.arange(len(tiles))
    c = c_tile + c_chunk * later + np.where(opens, c_rel, 0)
    prefix = np.concatenate([[0], np.cumsum(c)])
    total = int(prefix[-1])
    out = [int(np.searchsorted(prefix[:len(tiles) + 1], total * b // n_parts, side="left")) for b in range(n_parts)]
    return np.asarray(out + [len(tiles)], dtype=np.int32)


# ---------------------------------------------------------------------------- evaluation
def recall_ndcg_per_user(embedding, train_user_dict, test_user_dict, all_item_id_range, K):
    """``calc_recall_ndcg`` restated user by user (reference ``metric.py:36-68`` with
    ``one_recall_at_k`` ``:5-7``, ``one_dcg_at_k`` ``:8-22`` method 1, ``one_ndcg_at_k`` ``:23-34``):
    scores of a user against all items, training items set to 0.0 (``:50``), descending sort
    (``:51``), the first K ranks marked where they hit a test item (``:56-59``), recall = hits /
    |test items|, ndcg = dcg / dcg of the user's own sorted hit list; both averaged over the test
    users.  Ties in the score are broken by a stable sort here (the reference's ``th.sort`` leaves
    them unspecified)."""
    emb = np.asarray(embedding, dtype=np.float64)
    items = np.asarray(all_item_id_range)
    disc = 1.0 / np.log2(np.arange(2, K + 2))
    recall_all = ndcg_all = 0.0
    for u, pos in test_user_dict.items():
        score = emb[items] @ emb[u]
        score[np.asarray(train_user_dict[u], dtype=np.int64)] = 0.0
        rank = np.argsort(-score, kind="stable")[:K]
        pos_set = set(int(x) for x in pos)
        hit = np.array([1.0 if int(r) in pos_set else 0.0 for r in rank])
        if len(pos) > 0:
            recall_all += hit.sum() / len(pos)
        ideal = float((np.sort(hit)[::-1] * disc).sum())
        ndcg_all += float((hit * disc).sum()) / ideal if ideal else 0.0
    return recall_all / len(test_user_dict), ndcg_all / len(test_user_dict)
