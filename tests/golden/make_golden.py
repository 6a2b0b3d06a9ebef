"""Generate golden fixtures from the reference's own code.  Runs ONLY in the authoring
container (needs /root/reference); the .npz files it writes are committed and are
the only thing that travels.

What is pinned by these fixtures
--------------------------------
The reference (`/root/reference/models.py`) is imported unmodified.  Its `dgl`
dependency is not installable here, so a stand-in `dgl` module is injected into
`sys.modules` whose graph object implements exactly the calls the path makes
(`local_var`, `ndata`/`edata`, `filter_edges`, `apply_edges`, `update_all`,
`edge_softmax`); its sparse operators are the fp64 numpy oracle
(`oracle/kgat_oracle.py`).  Consequently:

* F1 (`att_score_*`): outputs of the reference's `Model._att_score`
  (`models.py:135-144`) per relation on explicit edge batches - reference torch
  code only, no oracle involved.  Pins the attention arithmetic.
* F2 (`attention`, `gnn_out`, `layer_out_*`): outputs of the reference's
  `Model.compute_attention` / `Model.gnn` / `KGATConv.forward` glue
  (`models.py:49-70,146-168`) over the stand-in graph.  Pins the glue (relation
  loop, zero-initialised partial column, un-normalised h feeding the next layer,
  concat order, leaky-relu slope, no bias); the sparse ops inside are the oracle's
  (DGL parity itself stays unpinned, see oracle header).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import kgat_oracle as orc  # noqa: E402

REF = "/root/reference"


# ------------------------------------------------------------------ stand-in dgl
class _Frame(dict):
    pass


class _EdgeBatch:
    def __init__(self, g, eids):
        self._g, self._e = g, eids

    @property
    def src(self):
        g, e = self._g, self._e
        return {k: v[g._src[e]] for k, v in g.ndata.items()}

    @property
    def dst(self):
        g, e = self._g, self._e
        return {k: v[g._dst[e]] for k, v in g.ndata.items()}

    @property
    def data(self):
        return {k: v[self._e] for k, v in self._g.edata.items()}


class StubGraph:
    """Just enough of DGLGraph 0.4.x for models.py; sparse ops = fp64 oracle."""

    def __init__(self, n, src, dst):
        self._n = n
        self._src = th.as_tensor(np.asarray(src), dtype=th.long)
        self._dst = th.as_tensor(np.asarray(dst), dtype=th.long)
        self.ndata, self.edata = _Frame(), _Frame()

    def local_var(self):
        g = StubGraph.__new__(StubGraph)
        g._n, g._src, g._dst = self._n, self._src, self._dst
        g.ndata, g.edata = _Frame(self.ndata), _Frame(self.edata)
        return g

    def filter_edges(self, pred):
        mask = pred(_EdgeBatch(self, th.arange(len(self._src))))
        return th.nonzero(mask, as_tuple=False).reshape(-1)

    def apply_edges(self, func, eids):
        out = func(_EdgeBatch(self, eids))
        for k, v in out.items():
            if k not in self.edata:  # DGL: new column, zero-initialised, partial write
                self.edata[k] = th.zeros((len(self._src),) + tuple(v.shape[1:]), dtype=v.dtype)
            col = self.edata[k].clone()
            col[eids] = v
            self.edata[k] = col

    def update_all(self, msg, red):
        assert msg[0] == "u_mul_e" and red[0] == "sum" and msg[3] == red[1]
        X = self.ndata[msg[1]].detach().numpy()
        w = self.edata[msg[2]].detach().numpy()
        out = orc.spmm_u_mul_e_sum(self._n, self._src.numpy(), self._dst.numpy(), X, w)
        self.ndata[red[2]] = th.as_tensor(out, dtype=self.ndata[msg[1]].dtype)


def _install_stub():
    dgl = types.ModuleType("dgl")
    fn = types.ModuleType("dgl.function")
    fn.u_mul_e = lambda a, b, m: ("u_mul_e", a, b, m)
    fn.sum = lambda m, o: ("sum", m, o)
    nn = types.ModuleType("dgl.nn")
    pt = types.ModuleType("dgl.nn.pytorch")
    sm = types.ModuleType("dgl.nn.pytorch.softmax")
    cv = types.ModuleType("dgl.nn.pytorch.conv")

    def edge_softmax(g, logits):
        a = orc.edge_softmax(g._n, g._dst.numpy(), logits.detach().numpy())
        return th.as_tensor(a, dtype=logits.dtype)

    sm.edge_softmax = edge_softmax
    cv.SAGEConv = object
    dgl.function, dgl.nn, nn.pytorch, pt.softmax, pt.conv = fn, nn, pt, sm, cv
    for name, mod in [("dgl", dgl), ("dgl.function", fn), ("dgl.nn", nn), ("dgl.nn.pytorch", pt),
                      ("dgl.nn.pytorch.softmax", sm), ("dgl.nn.pytorch.conv", cv)]:
        sys.modules[name] = mod


# ------------------------------------------------------------------ toy CKGs
def toy_ckg(seed, n_users, n_items, n_attr, n_kg_rel, n_kg, n_uv, hub_deg):
    """A miniature collaborative KG in the reference's layout (dataset.py:57-98):
    ids = <users | items | attribute entities>; triplets [h, r, t]; u->v pairs get
    relation n_kg_rel and v->u pairs n_kg_rel+1.  Has multi-edges, an isolated
    destination (last attribute entity never a head) and one hub head with
    >= hub_deg in-edges (> 2 wavefront chunks)."""
    rng = np.random.default_rng(seed)
    n = n_users + n_items + n_attr
    items = np.arange(n_users, n_users + n_items)
    attrs = np.arange(n_users + n_items, n - 1)  # last attr is never a head
    h = rng.choice(items, n_kg)
    t = rng.choice(np.arange(n_users + n_items, n), n_kg)
    r = rng.integers(0, n_kg_rel, n_kg)
    kg = np.stack([h, r, t], 1)
    # inverse direction for part of them so attribute entities are heads too
    inv = kg[: n_kg // 2][:, [2, 1, 0]].copy()
    inv = inv[inv[:, 0] != n - 1]
    # hub: one attribute head with hub_deg incoming edges, incl. exact duplicates
    hub = attrs[0]
    hub_t = rng.choice(items, hub_deg)
    hub_tr = np.stack([np.full(hub_deg, hub), rng.integers(0, n_kg_rel, hub_deg), hub_t], 1)
    u = rng.integers(0, n_users, n_uv)
    v = rng.choice(items, n_uv)
    uv = np.stack([u, np.full(n_uv, n_kg_rel), v], 1)
    vu = np.stack([v, np.full(n_uv, n_kg_rel + 1), u], 1)
    dup = kg[:5].copy()  # same (h, r, t) repeated: multi-edges
    dup2 = kg[5:10].copy()
    dup2[:, 1] = (dup2[:, 1] + 1) % n_kg_rel  # same (h, t), different relation
    trip = np.vstack([kg, inv, hub_tr, dup, dup2, uv, vu]).astype(np.int32)
    return n, trip, n_kg_rel + 2


def run_case(models, name, seed, d, k, hidden, n_layers, graph_kw):
    n, trip, n_rel = toy_ckg(seed, **graph_kw)
    src, dst, etype = trip[:, 2], trip[:, 0], trip[:, 1]
    th.manual_seed(seed)
    model = models.Model(use_KG=True, input_node_dim=d, gnn_model="kgat", num_gnn_layers=n_layers,
                         n_hidden=hidden, dropout=0.0, use_attention=True, n_entities=n,
                         n_relations=n_rel, relation_dim=k)
    # parameters are fp32-representable; run the reference in fp64 on them
    p32 = {kk: v.detach().clone().numpy() for kk, v in model.state_dict().items()}
    model = model.double()
    g = StubGraph(n, src, dst)
    g.ndata["id"] = th.arange(n, dtype=th.long)
    g.edata["type"] = th.as_tensor(etype.astype(np.int64))
    out = {"n_nodes": np.int64(n), "n_rel": np.int64(n_rel), "triplets": trip,
           "entity_embed": p32["entity_embed.weight"], "relation_embed": p32["relation_embed.weight"],
           "W_R": p32["W_R"]}
    for i in range(n_layers):
        out["W2_%d" % i] = p32["layers.%d.res_fc_2.weight" % i]
    with th.no_grad():
        # F1: reference _att_score on explicit per-relation edge batches
        for r in range(n_rel):
            eids = th.nonzero(g.edata["type"] == r, as_tuple=False).reshape(-1)
            model.W_r = model.W_R[r]
            out["att_score_%d" % r] = model._att_score(_EdgeBatch(g, eids))["att_w"].numpy()
            out["att_eids_%d" % r] = eids.numpy().astype(np.int32)
        # F2: reference glue over the stand-in graph
        a = model.compute_attention(g)
        out["attention"] = a.numpy()
        g.edata["w"] = a
        out["gnn_out"] = model.gnn(g, g.ndata["id"]).numpy()
        h = model.entity_embed(g.ndata["id"])
        for i, layer in enumerate(model.layers):
            h = layer(g, h)
            out["layer_out_%d" % i] = h.numpy()
    # dense losses of the reference Model (pure torch, no graph): transR (models.py:114-133) and
    # the BPR loss get_loss (models.py:170-178) on fixed index batches
    rng = np.random.default_rng(seed)
    B = 37
    hh = th.as_tensor(rng.integers(0, n, B)); rr = th.as_tensor(rng.integers(0, n_rel, B))
    pt = th.as_tensor(rng.integers(0, n, B)); nt = th.as_tensor(rng.integers(0, n, B))
    with th.no_grad():
        out["transR_idx"] = np.stack([hh.numpy(), rr.numpy(), pt.numpy(), nt.numpy()])
        out["transR_loss"] = model.transR(hh, rr, pt, nt).numpy()
        out["bpr_loss"] = model.get_loss(th.as_tensor(out["gnn_out"]), hh, pt, nt).numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: N=%d E=%d R=%d d=%d k=%d -> %s (%.1f KB)" % (
        name, n, len(trip), n_rel, d, k, path, os.path.getsize(path) / 1024))


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (authoring container only)")
    _install_stub()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import models  # the reference's models.py, unmodified
    small = dict(n_users=12, n_items=20, n_attr=18, n_kg_rel=3, n_kg=120, n_uv=60, hub_deg=140)
    mid = dict(n_users=60, n_items=80, n_attr=60, n_kg_rel=5, n_kg=900, n_uv=400, hub_deg=300)
    run_case(models, "toy_d8", 1234, d=8, k=8, hidden=8, n_layers=3, graph_kw=small)
    run_case(models, "toy_d16_k32", 1235, d=16, k=32, hidden=32, n_layers=2, graph_kw=small)
    run_case(models, "toy_d64", 1236, d=64, k=64, hidden=64, n_layers=3, graph_kw=mid)


if __name__ == "__main__":
    main()
