"""DGLGraph operator surface for the KGAT propagation path, backed by the gfx950 kernels.

Mirrors, with the same names, argument meaning and error behaviour, exactly the DGL 0.4.x
calls the reference makes:

* construction ``DGLGraph()``, ``add_nodes``, ``add_edges``, ``readonly``   (dataset.py:114-117)
* frames ``g.ndata[k]`` / ``g.edata[k]`` get / set / ``pop``            (models.py:62,64,153; kgat.py:58,100-101,144)
* ``g.local_var()``                                                     (models.py:61,148,157)
* ``g.filter_edges(pred)`` / ``g.apply_edges(func, eids)`` + EdgeBatch  (models.py:140-143,150,152)
* ``g.update_all(fn.u_mul_e(..), fn.sum(..))``                          (models.py:63)

Structure (COO, destination-major CSR, relation grouping) is built on the device by
``libkgat_hip.so`` the first time a kernel needs it and is shared by every ``local_var`` view.
The sparse arithmetic has no CPU implementation: calling ``update_all`` / ``edge_softmax`` on
CPU tensors raises.  On top of the surface the graph offers a fused fast path,
``kgat_attention`` (one launch instead of the per-relation ``filter_edges``/``apply_edges``
loop), which the generic surface does not require.
"""
from collections import namedtuple
from collections.abc import MutableMapping

import os

import numpy as np
import torch

from . import ops
from .options import options
from .function import BuiltinMessage, BuiltinReduce
from .lazy import LazyEdgeWeights, pending_csr_weights

ALL = "__ALL__"


class DGLError(Exception):
    """Same role as dgl.DGLError: misuse of the graph API."""


def _as_id_array(x, name):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    if x.ndim == 0:
        x = x.reshape(1)
    if x.ndim != 1:
        raise DGLError("%s must be a 1-D array of node ids" % name)
    if x.size and not np.issubdtype(x.dtype, np.integer):
        raise DGLError("%s must hold integers" % name)
    return x.astype(np.int64, copy=False)


class Frame(MutableMapping):
    """Column store for node or edge features (``g.ndata`` / ``g.edata``)."""

    def __init__(self, num_rows_fn, what, cols=None):
        self._n = num_rows_fn
        self._what = what
        self._cols = dict(cols) if cols else {}
        self._owned = set()  # columns this frame allocated itself (no other view can alias them)

    def __getitem__(self, key):
        # a column that has been handed out may be aliased by the caller: from here on a partial
        # write goes through a copy again (DGL's out-of-place update_rows semantics)
        self._owned.discard(key)
        return self._cols[key]

    def _own(self, key):
        """The column itself, for the in-place continuation of a partial write (apply_edges)."""
        return self._cols[key]

    def __setitem__(self, key, val):
        if not isinstance(val, torch.Tensor):
            raise DGLError("feature data must be a tensor")
        if val.dim() == 0 or val.shape[0] != self._n():
            raise DGLError("Expect number of features to match number of %s (len(%s)). Got %d and %d "
                           "instead." % (self._what, self._what,
                                         val.shape[0] if val.dim() else 0, self._n()))
        self._cols[key] = val
        self._owned.discard(key)

    def __delitem__(self, key):
        del self._cols[key]
        self._owned.discard(key)

    def __iter__(self):
        return iter(self._cols)

    def __len__(self):
        return len(self._cols)

    def __contains__(self, key):  # (the Mapping default goes through __getitem__, which hands the column out)
        return key in self._cols

    def clone(self):
        # the clone shares the column tensors: neither side may write them in place any more
        self._owned.clear()
        return Frame(self._n, self._what, self._cols)


CSR = namedtuple("CSR", "indptr col eid row_of")
RelGroups = namedtuple("RelGroups", "rel_ptr perm src_g dst_g pos_g gid gptr g_node n_groups g_tab")
# g_tab: per-graph scratch and lazily built statics of the attention forms, keyed by name:
#   "tiles" / "tiles_cost"  work tiles of the fused kernel and the cost they were split with
#   "rec_g"                 packed per-position records of the fused kernel (ops.att_pack_records)
#   "gpos_csr"              grouped position of every CSR position (inverse of pos_g): what the softmax
#                           reads grouped-order logits through
#   <width>                 per-group table of the two-launch forms; ("form", d, k): the form picked


class _Structure:
    """Edge list + per-device derived structure, shared by a graph and its local_var views.

    The edge list lives on the host (int64 numpy, what `add_edges` is given by the reference,
    dataset.py:116) or on a device (int32 tensors: `add_edges` with device tensors, e.g. a graph
    drawn or sharded on the GPU - no host copy is ever made of those).  Once a read-only graph has
    its device CSR, the host copy is dropped (3.2 GB at 200 M edges); `_src` / `_dst` rebuild it on
    demand for the few host-side queries (`edges()`, degrees)."""

    def __init__(self):
        self.n_nodes = 0
        self._host = (np.zeros(0, np.int64), np.zeros(0, np.int64))  # None: the device copy is the edge list
        self._dev_edges = None  # (src, dst) int32 device tensors
        self._n_edges = 0
        self.readonly = False
        self._dev = {}

    # ---- mutation
    def add_nodes(self, num):
        if self.readonly:
            raise DGLError("readonly graph. Mutation is not allowed.")
        self.n_nodes += int(num)
        self._dev.clear()

    def add_edges(self, u, v):
        if self.readonly:
            raise DGLError("readonly graph. Mutation is not allowed.")
        if (isinstance(u, torch.Tensor) and isinstance(v, torch.Tensor) and u.is_cuda and v.is_cuda
                and self._n_edges == 0 and u.dim() == 1 and u.shape == v.shape
                and not u.dtype.is_floating_point and not v.dtype.is_floating_point):
            # device-resident edge list: kept where it is
            if u.numel() and (int(torch.minimum(u.min(), v.min())) < 0 or
                              int(torch.maximum(u.max(), v.max())) >= self.n_nodes):
                raise DGLError("Node id out of range (number of nodes: %d)" % self.n_nodes)
            self._dev_edges = (u.to(torch.int32).contiguous(), v.to(torch.int32).contiguous())
            self._host = None
            self._n_edges = int(u.numel())
            self._dev.clear()
            return
        u, v = _as_id_array(u, "u"), _as_id_array(v, "v")
        if len(u) != len(v):
            if len(u) == 1:
                u = np.repeat(u, len(v))
            elif len(v) == 1:
                v = np.repeat(v, len(u))
            else:
                raise DGLError("Expect number of source and destination ids to match")
        if len(u) and (u.min() < 0 or v.min() < 0 or u.max() >= self.n_nodes or v.max() >= self.n_nodes):
            raise DGLError("Node id out of range (number of nodes: %d)" % self.n_nodes)
        src, dst = self._src, self._dst
        self._host = (np.concatenate([src, u]), np.concatenate([dst, v]))
        self._dev_edges = None
        self._n_edges = len(self._host[0])
        self._dev.clear()

    @property
    def n_edges(self):
        return self._n_edges

    def _host_edges(self):
        if self._host is not None:
            return self._host
        s, d = self._dev_edges
        return s.cpu().numpy().astype(np.int64), d.cpu().numpy().astype(np.int64)

    @property
    def _src(self):
        return self._host_edges()[0]

    @property
    def _dst(self):
        return self._host_edges()[1]

    def edge_device(self):
        """The device the edge list lives on, or None while it is (also) on the host."""
        return None if self._host is not None else self._dev_edges[0].device

    # ---- per-device caches
    def _cache(self, device):
        device = torch.device(device)
        c = self._dev.get(device)
        if c is None:
            c = self._dev[device] = {}
        return c

    def coo(self, device, dtype=torch.int32):
        c = self._cache(device)
        key = ("coo", dtype)
        if key not in c:
            if self._dev_edges is not None:
                c[key] = tuple(t.to(device=torch.device(device), dtype=dtype) for t in self._dev_edges)
            else:
                c[key] = (torch.as_tensor(self._host[0], dtype=dtype).to(device),
                          torch.as_tensor(self._host[1], dtype=dtype).to(device))
        return c[key]

    def csr(self, device):
        """Destination-major CSR (kgat_csr_from_coo)."""
        c = self._cache(device)
        if "csr" not in c:
            src, dst = self.coo(device)
            c["csr"] = CSR(*ops.csr_from_coo(self.n_nodes, src, dst))
            if self.readonly and self._host is not None and src.is_cuda:
                # a read-only graph with its structure on the device: the int32 device COO is the
                # edge list from here on, the int64 host arrays go
                self._dev_edges, self._host = (src, dst), None
        return c["csr"]

    def csr_pos(self, device):
        """CSR position of every edge id (inverse of csr.eid)."""
        c = self._cache(device)
        if "csr_pos" not in c:
            c["csr_pos"] = ops.invert_permutation(self.csr(device).eid)
        return c["csr_pos"]

    def csr_rev(self, device):
        """CSR of the reversed graph (rows = sources): used by the SpMM backward (S1b)."""
        c = self._cache(device)
        if "csr_rev" not in c:
            src, dst = self.coo(device)
            c["csr_rev"] = CSR(*ops.csr_from_coo(self.n_nodes, dst, src))
        return c["csr_rev"]

    def row_order(self, device):
        c = self._cache(device)
        if "order" not in c:
            c["order"] = ops.row_order_by_degree(self.csr(device).indptr)
        return c["order"]

    def rel_groups(self, etype, n_rel, device=None):
        """Edges grouped by relation and, inside a relation, sorted by destination (the CSR-ordered
        edge list grouped stably by relation, kgat_group_by_relation), with the grouped endpoint
        arrays, the CSR position of every grouped position, and the (head, relation) groups the
        split attention kernels share projections over (kgat_head_groups).  Cached per device on
        the identity + version of the `etype` tensor handed in (kept alive by the cache, so its
        identity cannot be recycled); a CPU-resident `etype` is copied to the device once."""
        dev = torch.device(device) if device is not None else etype.device
        c = self._cache(dev)
        hit = c.get("rel_groups")
        if hit is None or hit[2] is not etype or hit[0] != (etype._version, int(n_rel)):
            csr = self.csr(dev)
            et32 = etype.to(device=dev, dtype=torch.int32).contiguous()
            rel_ptr, idx = ops.group_by_relation(ops.gather(csr.eid, et32), int(n_rel))
            dst_g = ops.gather(idx, csr.row_of)
            gid, gptr, g_node, n_groups = ops.head_groups(rel_ptr, dst_g)
            hit = ((etype._version, int(n_rel)),
                   RelGroups(rel_ptr, ops.gather(idx, csr.eid), ops.gather(idx, csr.col), dst_g, idx,
                             gid, gptr, g_node, n_groups, {}), etype)
            c["rel_groups"] = hit
        return hit[1]

    def weight_in_csr_order(self, w_flat):
        """Per-edge weights permuted to CSR order once per distinct weight tensor; the SpMM
        then streams them coalesced (kgat.py:144 sets 'w' once per epoch, models.py:63 uses
        it in every layer of every batch)."""
        c = self._cache(w_flat.device)
        key = (w_flat.data_ptr(), w_flat._version, w_flat.numel())
        hit = c.get("w_csr")
        if hit is None or hit[0] != key:
            hit = (key, ops.gather(self.csr(w_flat.device).eid, w_flat), w_flat)
            c["w_csr"] = hit
        return hit[1]

    def weight_in_rev_order(self, w_flat):
        """The same weights in the reversed graph's CSR order (SpMM backward, S1b), cached the
        same way: one permutation per attention refresh, not one per backward call."""
        c = self._cache(w_flat.device)
        key = (w_flat.data_ptr(), w_flat._version, w_flat.numel())
        hit = c.get("w_rev")
        if hit is None or hit[0] != key:
            hit = (key, ops.gather(self.csr_rev(w_flat.device).eid, w_flat), w_flat)
            c["w_rev"] = hit
        return hit[1]

    @staticmethod
    def _flat(w):
        if w.dim() == 2 and w.shape[1] == 1:
            return w.reshape(-1)
        if w.dim() == 1:
            return w
        raise NotImplementedError("u_mul_e on the KGAT path takes an (E,1) or (E,) edge weight; got %s"
                                  % (tuple(w.shape),))

    def csr_weights(self, w):
        """The edge weights `w` (edge-id order, (E,) or (E,1)) in CSR order.  A still-pending
        LazyEdgeWeights of this graph is answered from the CSR copy it stands for, without touching
        (and thereby materialising) it."""
        hit = pending_csr_weights(w, self)
        if hit is not None:
            return hit
        return self.weight_in_csr_order(self._flat(w).detach().contiguous())

    def rev_weights(self, w):
        """The same weights in the reversed graph's CSR order (SpMM backward); a pending
        LazyEdgeWeights is served from its CSR copy through the composed position map."""
        w_csr = pending_csr_weights(w, self)
        if w_csr is None:
            return self.weight_in_rev_order(self._flat(w).detach().contiguous())
        c = self._cache(w_csr.device)
        hit = c.get("w_rev_of_csr")
        if hit is None or hit[0] is not w_csr:
            if "rev_from_csr" not in c:  # CSR position of the edge at every reversed-CSR position
                c["rev_from_csr"] = ops.gather(self.csr_rev(w_csr.device).eid, self.csr_pos(w_csr.device))
            hit = c["w_rev_of_csr"] = (w_csr, ops.gather(c["rev_from_csr"], w_csr))
        return hit[1]

    def remember_weight(self, w_flat, w_csr):
        self._cache(w_flat.device)["w_csr"] = ((w_flat.data_ptr(), w_flat._version, w_flat.numel()),
                                               w_csr, w_flat)


def _f32_products():
    """``KGAT_ATT_F32_PRODUCTS=1``: the attention kernels' two products on the fp32 MFMA (the C
    ABI's KGAT_ATT_F32_PRODUCTS flag) instead of the default bf16-piece products."""
    return options.att_f32_products


def _fused_gpt(n_nodes, d):
    """Head groups per work tile of the fused attention kernel: 32 at d = k = 64 with the piece products (the kernel on
    v_mfma_f32_32x32x16_f16, KGAT_ATT_TILES32; node ids below 2^27), else 16."""
    return 32 if (d == 64 and options.att_tiles32 and not _f32_products() and n_nodes < (1 << 27)) else 16


def _gpos_csr(groups):
    """The CSR-position -> grouped-position map the softmax reads grouped-order logits through (graph-static)."""
    if "gpos_csr" not in groups.g_tab:
        groups.g_tab["gpos_csr"] = ops.invert_permutation(groups.pos_g)
    return groups.g_tab["gpos_csr"]


def _fused_statics(groups, gpt=16):
    """Packed records of the fused path for `gpt` groups per tile + the position map (graph-static)."""
    rec = groups.g_tab.get(("rec_g", gpt))
    if rec is None:
        rec = groups.g_tab[("rec_g", gpt)] = ops.att_pack_records(groups.rel_ptr, groups.gptr, groups.gid, groups.src_g,
                                                                  groups_per_tile=gpt)
    return rec, _gpos_csr(groups)


def _fused_tiles(groups, d, gpt=16):
    """Work tiles of the fused attention kernel and their split over the workgroups, kept with the
    relation grouping (graph-static); the split cost goes with the product form taken at width d."""
    cost = ops.fold_tile_cost(d, _f32_products(), gpt)
    key = (cost, gpt)
    tiles = groups.g_tab.get("tiles")
    if tiles is None or groups.g_tab.get("tiles_cost") != key:
        tiles = groups.g_tab["tiles"] = ops.fold_tiles(groups.rel_ptr, groups.gid, groups.gptr, groups.n_groups,
                                                       cap=ops.FOLD_TILE_CAP32 if gpt == 32 else ops.FOLD_TILE_CAP,
                                                       cost=cost, groups_per_tile=gpt)
        groups.g_tab["tiles_cost"] = key
    return tiles


class EdgeBatch:
    """What a UDF passed to filter_edges / apply_edges sees: lazily gathered src / dst /
    edge features of the selected edges (reference models.py:140-143)."""

    class _Lazy(MutableMapping):
        def __init__(self, frame, index):
            self._frame, self._index, self._got = frame, index, {}

        def __getitem__(self, key):
            if key not in self._got:
                col = self._frame[key]
                idx = self._index(col.device)
                self._got[key] = col if idx is None else col.index_select(0, idx)
            return self._got[key]

        def __setitem__(self, key, val):
            self._got[key] = val

        def __delitem__(self, key):
            del self._got[key]

        def __iter__(self):
            return iter(self._frame)

        def __len__(self):
            return len(self._frame)

    def __init__(self, g, eids):
        self._g, self._eids = g, eids
        st = g._st

        def edge_index(device):
            return None if eids is None else eids.to(device)

        def end_index(which):
            def fn(device):
                ends = st.coo(device, torch.int64)[which]
                return ends if eids is None else ends.index_select(0, eids.to(device))
            return fn

        self.src = EdgeBatch._Lazy(g._node_frame, end_index(0))
        self.dst = EdgeBatch._Lazy(g._node_frame, end_index(1))
        self.data = EdgeBatch._Lazy(g._edge_frame, edge_index)

    def batch_size(self):
        return self._g.number_of_edges() if self._eids is None else len(self._eids)

    def __len__(self):
        return self.batch_size()


class DGLGraph:
    def __init__(self, _structure=None, _nframe=None, _eframe=None):
        self._st = _structure if _structure is not None else _Structure()
        self._node_frame = _nframe if _nframe is not None else Frame(self.number_of_nodes, "nodes")
        self._edge_frame = _eframe if _eframe is not None else Frame(self.number_of_edges, "edges")
        self.partition = None  # set by partition.shard_graph for multi-GPU runs

    # ---- construction (reference dataset.py:114-117)
    def add_nodes(self, num, data=None):
        if len(self._node_frame):
            raise DGLError("adding nodes after node features were set is not supported")
        self._st.add_nodes(num)
        if data:
            for k, v in data.items():
                self.ndata[k] = v

    def add_edges(self, u, v, data=None):
        if len(self._edge_frame):
            raise DGLError("adding edges after edge features were set is not supported")
        self._st.add_edges(u, v)
        if data:
            for k, val in data.items():
                self.edata[k] = val

    def readonly(self, readonly_state=True):
        self._st.readonly = bool(readonly_state)
        return self

    # ---- queries
    def number_of_nodes(self):
        return self._st.n_nodes

    def number_of_edges(self):
        return self._st.n_edges

    def __len__(self):
        return self.number_of_nodes()

    @property
    def is_readonly(self):
        return self._st.readonly

    def edges(self, form="uv", order=None):
        """DGLGraph.edges / all_edges: endpoints (and ids) of all edges, in edge-id order
        (order=None / 'eid') or sorted by (src, dst) (order='srcdst')."""
        src_h, dst_h = self._st._host_edges()
        src, dst = torch.as_tensor(src_h), torch.as_tensor(dst_h)
        eid = torch.arange(len(src))
        if order == "srcdst":
            perm = torch.as_tensor(np.lexsort((dst_h, src_h)))
            src, dst, eid = src[perm], dst[perm], eid[perm]
        elif order not in (None, "eid"):
            raise DGLError("unknown edge order %r" % (order,))
        return {"uv": (src, dst), "eid": eid, "all": (src, dst, eid)}[form]

    all_edges = edges

    def in_degrees(self):
        return torch.as_tensor(np.bincount(self._st._dst, minlength=self._st.n_nodes))

    def out_degrees(self):
        return torch.as_tensor(np.bincount(self._st._src, minlength=self._st.n_nodes))

    # ---- frames
    @property
    def ndata(self):
        return self._node_frame

    @property
    def edata(self):
        return self._edge_frame

    def local_var(self):
        """A view whose feature writes do not leak to this graph (models.py:61,148,157)."""
        g = DGLGraph(self._st, self._node_frame.clone(), self._edge_frame.clone())
        g.partition = self.partition
        return g

    class _LocalScope:
        def __init__(self, g):
            self.g = g

        def __enter__(self):
            g = self.g
            self.saved = (g._node_frame, g._edge_frame)
            g._node_frame, g._edge_frame = g._node_frame.clone(), g._edge_frame.clone()
            return g

        def __exit__(self, *exc):
            self.g._node_frame, self.g._edge_frame = self.saved
            return False

    def local_scope(self):
        return DGLGraph._LocalScope(self)

    # ---- UDF plumbing
    def _edge_ids(self, edges):
        if isinstance(edges, str) and edges == ALL:
            return None
        if isinstance(edges, torch.Tensor):
            eids = edges.to(torch.int64).reshape(-1)
        else:
            eids = torch.as_tensor(np.asarray(edges, dtype=np.int64)).reshape(-1)
        return eids

    def filter_edges(self, predicate, edges=ALL):
        """Ids (int64) of the edges for which ``predicate(EdgeBatch)`` is true (models.py:150)."""
        eids = self._edge_ids(edges)
        mask = predicate(EdgeBatch(self, eids))
        if mask.dtype != torch.bool:
            mask = mask != 0
        hit = torch.nonzero(mask.reshape(-1), as_tuple=False).reshape(-1)
        return hit if eids is None else eids.to(hit.device).index_select(0, hit)

    def apply_edges(self, func, edges=ALL):
        """Run a UDF on an edge batch and write its outputs into ``edata`` (models.py:152).  A
        column that does not exist yet is zero-initialised before a partial write."""
        if isinstance(func, BuiltinMessage):
            raise NotImplementedError("apply_edges with a builtin is outside the KGAT path")
        eids = self._edge_ids(edges)
        out = func(EdgeBatch(self, eids))
        if not isinstance(out, dict):
            raise DGLError("edge UDF must return a dict of tensors")
        n_sel = self.number_of_edges() if eids is None else len(eids)
        for key, val in out.items():
            if val.shape[0] != n_sel:
                raise DGLError("Expect number of features to match number of edges in the batch")
            if eids is None:
                self._edge_frame[key] = val
                continue
            tracked = torch.is_grad_enabled() and val.requires_grad
            if key in self._edge_frame._owned and not tracked:
                # the column was zero-initialised by an earlier partial write through this very
                # view (the per-relation loop of models.py:149-152): nothing else can see it, so
                # the next relation's slice goes in place instead of through an E-sized copy
                self._edge_frame._own(key).index_copy_(0, eids.to(val.device), val)
                continue
            if key in self._edge_frame:
                col = self._edge_frame[key].clone()
            else:
                col = torch.zeros((self.number_of_edges(),) + tuple(val.shape[1:]), dtype=val.dtype,
                                  device=val.device)
            col.index_copy_(0, eids.to(val.device), val)
            self._edge_frame[key] = col
            if not tracked:
                self._edge_frame._owned.add(key)

    # ---- the aggregation (models.py:63)
    def update_all(self, message_func, reduce_func, apply_node_func=None):
        if not (isinstance(message_func, BuiltinMessage) and isinstance(reduce_func, BuiltinReduce)):
            raise NotImplementedError("only builtin message/reduce pairs run on the HIP kernels; "
                                      "python UDF message passing is outside the KGAT path")
        if message_func.name != "u_mul_e" or reduce_func.name != "sum":
            raise NotImplementedError("the KGAT path uses update_all(fn.u_mul_e, fn.sum); got %s/%s"
                                      % (message_func.name, reduce_func.name))
        if message_func.out_field != reduce_func.msg_field:
            raise DGLError("reduce reads message field %r but the message function writes %r"
                           % (reduce_func.msg_field, message_func.out_field))
        if message_func.src_field not in self._node_frame:
            raise KeyError(message_func.src_field)
        if message_func.edge_field not in self._edge_frame:
            raise KeyError(message_func.edge_field)
        from .autograd import u_mul_e_sum
        x = self._node_frame[message_func.src_field]
        w = self._edge_frame[message_func.edge_field]
        self._node_frame[reduce_func.out_field] = u_mul_e_sum(self, x, w)
        if apply_node_func is not None:
            raise NotImplementedError("apply_node_func is outside the KGAT path")

    def _pick_attention_form(self, groups, n_rel, d, k, race_run=None):
        """Attention form as a function of the graph's statistics only (so that every process, and
        every rank of a sharded job, takes the same arithmetic path and produces the same bits).
        Head-group forms when the groups actually share work (measured on MI355X, d = 64: folded
        ~0.156 ms per 1e6 groups + 0.038 ms per 1e6 edges, one-kernel 0.153 ms per 1e6 edges -> the
        group forms win below ~0.74 groups per edge).  Between the two folded forms: the fused
        kernel recomputes a 16-group block's vectors once per `cap` positions, so it wins while its
        tile count stays near the minimum ceil(n_groups / 16) and loses when large groups inflate
        it.  With 256 positions per tile and the cost-balanced split it wins on both benchmark
        shapes (amazon-book: 1.04x the minimum, 0.192 vs 0.278 ms; last-fm: 1.87x, 0.126 vs
        0.137 ms); the threshold stays at 2x.  ``KGAT_ATT_FORM=race`` times the two instead
        (opt-in: a wall-clock race is not reproducible across processes)."""
        st = self._st
        shares = 4 * groups.n_groups <= 3 * st.n_edges
        fused_ok = shares and ops.att_score_fused_supported(st.n_nodes, d, k, n_rel) and not (d == 128 and _f32_products())
        folded_ok = shares and ops.att_score_folded_supported(st.n_nodes, d, k, n_rel)
        if fused_ok and folded_ok:
            if race_run is not None:
                best = None
                for f in ("fused", "folded"):
                    race_run(f)
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    for _ in range(3):
                        race_run(f)
                    t1.record()
                    t1.synchronize()
                    ms = t0.elapsed_time(t1)
                    if best is None or ms < best[0]:
                        best = (ms, f)
                form = best[1]
            else:
                gpt = _fused_gpt(st.n_nodes, d)
                tiles = _fused_tiles(groups, d, gpt)
                n_tiles = int(tiles[1][-1])
                form = "fused" if n_tiles <= 2 * ((groups.n_groups + gpt - 1) // gpt) else "folded"
            groups.g_tab.pop("tiles" if form == "folded" else d, None)  # the other form's scratch
            return form
        if fused_ok:
            return "fused"
        if folded_ok:
            return "folded"
        if 2 * groups.n_groups <= st.n_edges and ops.att_score_split_supported(st.n_nodes, d, k, n_rel):
            return "split"
        return "one"

    # ---- fused fast path (not part of the DGL surface)
    def kgat_attention(self, ent, W_R, rel, etype=None, algo="auto", lazy=None):
        """compute_attention (models.py:146-154): relation-grouped attention logits + destination
        softmax.  Returns the (E,1) weights in edge-id order; the graph keeps their CSR-ordered
        copy, so a following ``edata['w'] = result`` + ``update_all`` streams them without a
        permutation pass.  With ``lazy=True``, or ``lazy=None`` after the process opted in
        (``dgl_kgat_amd.enable_lazy_edge_weights()`` / ``KGAT_LAZY_EDGE_WEIGHTS=1``; see lazy.py),
        the result is a `lazy.LazyEdgeWeights`: real storage whose edge-id-ordered values are
        written by the first operation that looks at them - nothing on the path does."""
        if etype is None:
            etype = self._edge_frame["type"]
        st = self._st
        dev = ent.device
        csr = st.csr(dev)
        groups = st.rel_groups(etype, W_R.shape[0], dev)
        n_rel, d, k = W_R.shape
        ent_c, W_c, rel_c = ent.detach().contiguous(), W_R.detach().contiguous(), rel.detach().contiguous()
        # head-group forms (work shared by the edges of a (head, relation) group) when the groups
        # actually share; otherwise the one-kernel form.  "fused" (default, d <= 64) and "folded"
        # do the whole relation-space product per group and a d-length dot per edge (one launch
        # with the per-group vectors in LDS / two launches with a table); "split" keeps the
        # reference's contraction order (bit-identical to "one")
        grouped_out = not options.att_scatter_csr

        def run(form):
            if form == "fused":
                gpt = _fused_gpt(st.n_nodes, d)
                tiles = _fused_tiles(groups, d, gpt)  # graph-static work tiles of the fused kernel
                rec, _ = _fused_statics(groups, gpt)
                # the logits leave in grouped order (coalesced stores); the softmax reads them through
                # the inverse position map.  KGAT_ATT_SCATTER_CSR=1: the round-2 form, a 4-byte
                # scatter into CSR order (73 MB written for 14.6 MB of logits on the benchmark graph)
                res = ops.att_score_fused(st.n_nodes, groups.rel_ptr, groups.perm, groups.src_g, groups.pos_g,
                                          groups.gid, groups.gptr, groups.g_node, tiles[0], tiles[1],
                                          ent_c, W_c, rel_c, want_eid=False, want_csr=not grouped_out,
                                          want_grouped=grouped_out, part_tptr=tiles[2],
                                          f32_products=_f32_products(), rec_g=rec, groups_per_tile=gpt)
                return res[2] if grouped_out else res[1]
            if form in ("folded", "split"):
                folded = form == "folded"
                width = d if folded else k
                g_tab = groups.g_tab.get(width)  # per-group scratch table, kept with the graph
                if g_tab is None:
                    g_tab = groups.g_tab[width] = torch.empty((max(groups.n_groups, 1), width), dtype=torch.float32,
                                                              device=dev)
                return ops.att_score_split(st.n_nodes, groups.rel_ptr, groups.perm, groups.src_g, groups.pos_g,
                                           groups.gid, groups.gptr, groups.g_node, groups.n_groups,
                                           ent_c, W_c, rel_c, g_tab=g_tab, want_eid=False, folded=folded,
                                           f32_products=folded and _f32_products())[1]
            return ops.att_score(st.n_nodes, groups.rel_ptr, groups.perm, groups.src_g, groups.dst_g,
                                 ent_c, W_c, rel_c, pos_g=groups.pos_g, algo="auto" if form == "one" else form)[1]

        form = options.att_form if algo == "auto" else algo
        race = form == "race"
        if form in ("auto", "race"):
            form = groups.g_tab.get(("form", d, k))
        if form is None:
            form = self._pick_attention_form(groups, n_rel, d, k, run if race else None)
            groups.g_tab[("form", d, k)] = form
        logits = run(form)
        if form not in ("fused", "folded", "split"):
            form = "one"
        st.last_att_form = (form, groups.n_groups)
        if form == "fused" and grouped_out:
            # grouped-order logits: the sweep gathers logits[gpos_csr[q]] for CSR position q (the map
            # takes the place of `eid` on the input side; no edge-id-ordered output is asked for)
            _, a_csr = ops.edge_softmax(csr.indptr, csr.row_of, _gpos_csr(groups), logits,
                                        in_csr_order=False, want_out=False, want_csr=True)
        else:
            _, a_csr = ops.edge_softmax(csr.indptr, csr.row_of, csr.eid, logits, in_csr_order=True,
                                        want_out=False, want_csr=True)
        from . import lazy as lazy_mod
        if lazy is None:
            lazy = lazy_mod.enabled()
        elif lazy:
            lazy_mod._guard_legacy_to_dlpack()   # an explicit lazy=True: the export guard must be in place
        a = torch.empty((st.n_edges, 1), dtype=torch.float32, device=dev)
        a_flat = a.view(-1)

        def fill():  # edge-id order: coalesced writes, cached reads
            ops.gather(st.csr_pos(dev), a_csr, out=a_flat)
            st.remember_weight(a_flat, a_csr)
        if not lazy:
            fill()
            return a
        return LazyEdgeWeights(a, fill, st, a_csr)
