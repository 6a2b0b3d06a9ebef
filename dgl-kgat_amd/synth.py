"""Synthetic collaborative knowledge graphs with the shapes of the benchmark configs
(the real amazon-book / last-fm files are not available offline).

Layout follows the reference's DataLoader (dataset.py:57-98): node ids are
``<users | items | attribute entities>``; triplets ``[h, r, t]`` are the item-KG triplets
followed by the user->item pairs (relation R_kg) and their reverses (relation R_kg + 1)
(``np.vstack((kg, uv, vu))``, dataset.py:89); the graph is then built with
``add_edges(t, h)`` (dataset.py:116).  Degrees are power-law distributed, so hub
destinations exist as they do in the real KGs.
"""
import numpy as np
import torch

from .graph import DGLGraph


def _zipf_choice(rng, n_items, size, exponent, shuffle=True):
    w = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), exponent)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    idx = np.searchsorted(cdf, rng.random(size), side="left").astype(np.int64)
    np.minimum(idx, n_items - 1, out=idx)
    if shuffle:  # popularity rank should not coincide with id order
        idx = rng.permutation(n_items)[idx]
    return idx


def collaborative_kg(n_users, n_items, n_attrs, n_kg_rel, n_kg, n_uv, seed=1234, inverse_frac=0.4):
    """Triplets (E,3) int32 ``[h, r, t]`` of a CKG with the given counts."""
    rng = np.random.default_rng(seed)
    n = n_users + n_items + n_attrs
    item0, attr0 = n_users, n_users + n_items
    n_inv = int(n_kg * inverse_frac)
    n_fwd = n_kg - n_inv
    rel = _zipf_choice(rng, n_kg_rel, n_kg, 0.8)
    fwd_h = item0 + _zipf_choice(rng, n_items, n_fwd, 0.5)
    fwd_t = attr0 + _zipf_choice(rng, n_attrs, n_fwd, 1.0)
    inv_h = attr0 + _zipf_choice(rng, n_attrs, n_inv, 1.0)
    inv_t = item0 + _zipf_choice(rng, n_items, n_inv, 0.5)
    kg = np.stack([np.concatenate([fwd_h, inv_h]), rel, np.concatenate([fwd_t, inv_t])], 1)
    kg = kg[rng.permutation(n_kg)]
    u = _zipf_choice(rng, n_users, n_uv, 0.6)
    v = item0 + _zipf_choice(rng, n_items, n_uv, 0.9)
    order = np.argsort(u, kind="stable")  # the reference sorts interactions by user (dataset.py:26)
    u, v = u[order], v[order]
    uv = np.stack([u, np.full(n_uv, n_kg_rel), v], 1)
    vu = np.stack([v, np.full(n_uv, n_kg_rel + 1), u], 1)
    trip = np.vstack([kg, uv, vu]).astype(np.int32)
    return n, trip, n_kg_rel + 2


def amazon_book_ckg(seed=1234, scale=1.0):
    """amazon-book-shaped CKG: N = 159,251, E = 3,663,302, R = 41 at scale 1."""
    s = lambda x: max(int(round(x * scale)), 4)  # noqa: E731
    return collaborative_kg(s(70679), s(24915), s(63657), 39, s(2557746), s(552778), seed)


def last_fm_ckg(seed=1234, scale=1.0):
    """last-fm-shaped CKG: N = 81,832, E ~ 4.83 M, R = 11."""
    s = lambda x: max(int(round(x * scale)), 4)  # noqa: E731
    return collaborative_kg(s(23566), s(48123), s(10143), 9, s(464567), s(2185000), seed)


def power_law_ckg(n_nodes, n_edges, n_rel, seed=1234, alpha=1.1, max_in_degree=1_000_000):
    """Power-law graph: destinations Zipf(alpha) with a degree cap, sources uniform."""
    rng = np.random.default_rng(seed)
    h = _zipf_choice(rng, n_nodes, n_edges, alpha)
    deg = np.bincount(h, minlength=n_nodes)
    over = np.nonzero(deg > max_in_degree)[0]
    for v in over:  # re-draw the excess of capped hubs uniformly
        idx = np.nonzero(h == v)[0][max_in_degree:]
        h[idx] = rng.integers(0, n_nodes, len(idx))
    t = rng.integers(0, n_nodes, n_edges)
    r = rng.integers(0, n_rel, n_edges)
    return n_nodes, np.stack([h, r, t], 1).astype(np.int32), n_rel


def power_law_coo_device(n_nodes, n_edges, n_rel, device, seed=1234, alpha=1.1, max_in_degree=1_000_000,
                         cap="redraw"):
    """The C5 graph drawn on the device (torch RNG) so that a 10 M / 200 M instance costs a
    fraction of a second instead of minutes of host sampling.  Destinations follow Zipf(alpha)
    over a random relabelling of the nodes; sources and relation types are uniform.  The heaviest
    destinations are capped near `max_in_degree` in one of two ways:

    * ``cap="redraw"`` (default; the shape of `power_law_ckg`): an edge that lands on a node whose
      expected in-degree exceeds the cap keeps it with probability cap / expected, otherwise its
      destination is re-drawn uniformly - the hubs end at ~cap edges (binomial spread) and their
      excess becomes a uniform background;
    * ``cap="shift"``: a shifted Zipf ``p_i ~ (i + i0)^-alpha`` (Zipf-Mandelbrot) with the smallest
      shift that keeps the heaviest expectation under the cap - no background, a flatter head.

    Returns int32 device tensors ``(src, dst, etype)`` in edge-id order (src = tail, dst = head,
    as dataset.py:116)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    rank = torch.arange(1, n_nodes + 1, dtype=torch.float64, device=device)
    shift = 0.0
    if cap == "shift":
        lo, hi = 0.0, float(n_nodes)
        for _ in range(40):  # bisection on the shift: p_max(i0) is decreasing in i0
            i0 = 0.5 * (lo + hi)
            w = (rank + i0).pow(-alpha)
            if float(w[0] / w.sum()) * n_edges > max_in_degree:
                lo = i0
            else:
                hi = i0
        shift = hi
    elif cap != "redraw":
        raise ValueError("cap must be 'redraw' or 'shift'")
    w = (rank + shift).pow(-alpha)
    w /= w.sum()
    keep = None
    if cap == "redraw":
        keep = (max_in_degree / (w * n_edges)).clamp_(max=1.0)  # per popularity rank
        if bool((keep >= 1.0).all()):
            keep = None
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    del w, rank
    dst = torch.empty(n_edges, dtype=torch.int32, device=device)
    relabel = torch.randperm(n_nodes, generator=gen, device=device).to(torch.int32)
    chunk = 1 << 26
    for lo_e in range(0, n_edges, chunk):
        m = min(chunk, n_edges - lo_e)
        u = torch.rand(m, generator=gen, device=device, dtype=torch.float64)
        idx = torch.searchsorted(cdf, u).clamp_(max=n_nodes - 1)
        if keep is not None:
            redraw = torch.rand(m, generator=gen, device=device, dtype=torch.float64) >= keep[idx]
            uniform = torch.randint(0, n_nodes, (m,), generator=gen, device=device)
            idx = torch.where(redraw, uniform, idx)
            del redraw, uniform
        dst[lo_e:lo_e + m] = relabel[idx]
        del u, idx
    src = torch.randint(0, n_nodes, (n_edges,), generator=gen, device=device, dtype=torch.int32)
    etype = torch.randint(0, n_rel, (n_edges,), generator=gen, device=device, dtype=torch.int32)
    return src, dst, etype


def build_graph(n_nodes, triplets, device=None):
    """The reference's graph construction (dataset.py:112-120) on this package's DGLGraph."""
    g = DGLGraph()
    g.add_nodes(n_nodes)
    g.add_edges(triplets[:, 2], triplets[:, 0])
    g.readonly()
    ids = torch.arange(n_nodes, dtype=torch.long)
    et = torch.as_tensor(triplets[:, 1].astype(np.int64))
    if device is not None:
        ids, et = ids.to(device), et.to(device)
    g.ndata["id"] = ids
    g.edata["type"] = et
    return g


def build_graph_device(n_nodes, src, dst, etype):
    """The same graph from device-resident int32 endpoint / type arrays (`power_law_coo_device`):
    the edge list stays on the device, no host copy is made."""
    g = DGLGraph()
    g.add_nodes(n_nodes)
    g.add_edges(src, dst)
    g.readonly()
    g.ndata["id"] = torch.arange(n_nodes, dtype=torch.long, device=src.device)
    g.edata["type"] = etype.long()
    return g
