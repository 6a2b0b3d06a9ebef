"""Tensor-level wrappers around the C ABI (include/kgat_hip.h).

torch is used here for what it is good at on ROCm - device memory, the current HIP stream -
and nothing else: every arithmetic step of the path happens inside libkgat_hip.so.  All
functions validate device / dtype / contiguity before the call and turn a non-zero return
code into KGATLibraryError.  CPU tensors are rejected: this path has no CPU implementation.
"""
import torch

from . import _lib
from ._lib import KGATLibraryError, check

SPMM_MUL_SELF = 1
SPMM_DEFER_FINISH = 2
SPMM_ALGO = {"auto": 0, "merge": 1, "rows": 2, "generic": 3, "merge1": 4}
ATT_ALGO = {"auto": 0, "mfma": 1, "generic": 2, "mfma_chunk": 3}


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class KernelTimer:
    """Optional per-op HIP-event timing (bench.py): while active, every wrapped C-ABI op
    records an event pair on the stream it launches on.  Costs nothing when inactive."""
    active = None

    def __init__(self):
        self.records = []  # (name, info, start_event, end_event)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None
        return False

    def summary(self):
        """name -> list of (info, milliseconds); call after a device synchronize."""
        out = {}
        for name, info, a, b in self.records:
            out.setdefault(name, []).append((info, a.elapsed_time(b)))
        return out


class _timed:
    def __init__(self, name, info=None):
        self.t = KernelTimer.active
        self.name, self.info = name, info

    def __enter__(self):
        if self.t is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            self.b.record()
            self.t.records.append((self.name, self.info, self.a, self.b))
        return False


def _need(t, dtype, name, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if type(t) is not torch.Tensor and hasattr(t, "materialize"):
        t.materialize()  # lazy.LazyEdgeWeights: the kernels read through the raw pointer
    if not t.is_cuda:
        raise KGATLibraryError("%s is on %s: the KGAT propagation path only runs on a HIP device "
                               "(no CPU implementation exists in this package)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def csr_from_coo(n_nodes, src, dst):
    """(indptr, col, eid, row_of) of the destination-major CSR; stable in edge id."""
    src = _need(src, torch.int32, "src")
    dst = _need(dst, torch.int32, "dst", src.shape)
    lib = _lib.load()
    e, dev = src.numel(), src.device
    indptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(e, dtype=torch.int32, device=dev)
    eid = torch.empty(e, dtype=torch.int32, device=dev)
    row_of = torch.empty(e, dtype=torch.int32, device=dev)
    nb = lib.kgat_csr_from_coo_workspace_bytes(n_nodes, e)
    ws = _workspace(nb, dev)
    check(lib.kgat_csr_from_coo(n_nodes, e, _ptr(src), _ptr(dst), _ptr(indptr), _ptr(col), _ptr(eid),
                                _ptr(row_of), _ptr(ws), ws.numel(), _stream(src)), "kgat_csr_from_coo")
    return indptr, col, eid, row_of


def group_by_relation(etype, n_rel):
    etype = _need(etype, torch.int32, "etype")
    lib = _lib.load()
    e, dev = etype.numel(), etype.device
    rel_ptr = torch.empty(n_rel + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    ws = _workspace(lib.kgat_group_by_relation_workspace_bytes(e, n_rel), dev)
    check(lib.kgat_group_by_relation(e, n_rel, _ptr(etype), _ptr(rel_ptr), _ptr(perm), _ptr(ws),
                                     ws.numel(), _stream(etype)), "kgat_group_by_relation")
    return rel_ptr, perm


def invert_permutation(perm):
    perm = _need(perm, torch.int32, "perm")
    inv = torch.empty_like(perm)
    check(_lib.load().kgat_invert_permutation(perm.numel(), _ptr(perm), _ptr(inv), _stream(perm)),
          "kgat_invert_permutation")
    return inv


def row_order_by_degree(indptr):
    indptr = _need(indptr, torch.int32, "indptr")
    lib = _lib.load()
    n = indptr.numel() - 1
    order = torch.empty(n, dtype=torch.int32, device=indptr.device)
    ws = _workspace(lib.kgat_row_order_workspace_bytes(n), indptr.device)
    check(lib.kgat_row_order_by_degree(n, _ptr(indptr), _ptr(order), _ptr(ws), ws.numel(),
                                       _stream(indptr)), "kgat_row_order_by_degree")
    return order


def gather(index, values, out=None):
    """out[i] = values[index[i]] (float32 or int32 values); `out` may be a preallocated flat tensor."""
    index = _need(index, torch.int32, "index")
    dtype = torch.float32 if values.dtype == torch.float32 else torch.int32
    values = _need(values, dtype, "values")
    if out is None:
        out = torch.empty(index.numel(), dtype=dtype, device=index.device)
    else:
        out = _need(out, dtype, "out", (index.numel(),))
    fn = _lib.load().kgat_gather_f32 if dtype == torch.float32 else _lib.load().kgat_gather_i32
    with _timed("gather", (index.numel(),)):
        check(fn(index.numel(), _ptr(index), _ptr(values), _ptr(out), _stream(index)), "kgat_gather")
    return out


def att_score(n_nodes, rel_ptr, perm, src_g, dst_g, ent, W_R, rel, pos_g=None, algo="auto"):
    """Attention logits (E,) in edge-id order; with pos_g (CSR position of edge perm[i]) also
    in CSR order."""
    ent = _need(ent, torch.float32, "ent")
    n_rel, d, k = W_R.shape
    W_R = _need(W_R, torch.float32, "W_R")
    rel = _need(rel, torch.float32, "rel", (n_rel, k))
    if ent.shape != (n_nodes, d):
        raise ValueError("ent has shape %s, expected %s" % (tuple(ent.shape), (n_nodes, d)))
    perm = _need(perm, torch.int32, "perm")
    e = perm.numel()
    rel_ptr = _need(rel_ptr, torch.int32, "rel_ptr", (n_rel + 1,))
    src_g = _need(src_g, torch.int32, "src_g", (e,))
    dst_g = _need(dst_g, torch.int32, "dst_g", (e,))
    logits = torch.empty(e, dtype=torch.float32, device=ent.device)
    logits_csr = None
    if pos_g is not None:
        pos_g = _need(pos_g, torch.int32, "pos_g", (e,))
        logits_csr = torch.empty(e, dtype=torch.float32, device=ent.device)
    with _timed("att_score", (e, d, k)):
        check(_lib.load().kgat_att_score_f32(n_nodes, e, d, k, n_rel, _ptr(rel_ptr), _ptr(perm), _ptr(src_g),
                                             _ptr(dst_g), _ptr(ent), _ptr(W_R), _ptr(rel), _ptr(logits),
                                             _ptr(logits_csr), _ptr(pos_g), ATT_ALGO[algo], _stream(ent)),
              "kgat_att_score_f32")
    return logits, logits_csr


def head_groups(rel_ptr, dst_g):
    """(gid, gptr, g_node, n_groups) of a relation-grouped, destination-sorted edge list.
    Reads the group count back to the host (one sync, once per graph)."""
    rel_ptr = _need(rel_ptr, torch.int32, "rel_ptr")
    dst_g = _need(dst_g, torch.int32, "dst_g")
    lib = _lib.load()
    e, n_rel, dev = dst_g.numel(), rel_ptr.numel() - 1, dst_g.device
    gid_pad = torch.zeros(e + 16, dtype=torch.int32, device=dev)  # kernels read 16-byte groups: 16 ids of slack
    gid = gid_pad[:e]
    gptr = torch.empty(n_rel + 1, dtype=torch.int32, device=dev)
    g_node = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    ws = _workspace(lib.kgat_head_groups_workspace_bytes(e), dev)
    check(lib.kgat_head_groups(e, n_rel, _ptr(rel_ptr), _ptr(dst_g), _ptr(gid), _ptr(gptr), _ptr(g_node),
                               _ptr(ws), ws.numel(), _stream(dst_g)), "kgat_head_groups")
    n_groups = int(gptr[n_rel].item())
    return gid, gptr, g_node[:max(n_groups, 1)].clone(), n_groups


ATT_F32_PRODUCTS = 1  # include/kgat_hip.h: KGAT_ATT_F32_PRODUCTS
ATT_TILES32 = 2       # include/kgat_hip.h: KGAT_ATT_TILES32


def att_score_split_supported(n_nodes, d, k, n_rel):
    return bool(_lib.load().kgat_att_score_split_supported(int(n_nodes), int(d), int(k), int(n_rel)))


def att_score_folded_supported(n_nodes, d, k, n_rel):
    return bool(_lib.load().kgat_att_score_folded_supported(int(n_nodes), int(d), int(k), int(n_rel)))


def att_score_split(n_nodes, rel_ptr, perm, src_g, pos_g, gid, gptr, g_node, n_groups, ent, W_R, rel,
                    want_csr=True, g_tab=None, want_eid=True, folded=False, f32_products=False):
    """Attention logits via head groups (see kgat_att_score_split_f32; folded=True:
    kgat_att_score_folded_f32, whose scratch table is n_groups x d; f32_products: its
    KGAT_ATT_F32_PRODUCTS flag).  Returns (logits edge-id order, logits CSR order or None)."""
    ent = _need(ent, torch.float32, "ent")
    n_rel, d, k = W_R.shape
    W_R = _need(W_R, torch.float32, "W_R")
    rel = _need(rel, torch.float32, "rel", (n_rel, k))
    perm = _need(perm, torch.int32, "perm")
    e = perm.numel()
    for name, t in (("src_g", src_g), ("pos_g", pos_g), ("gid", gid)):
        _need(t, torch.int32, name, (e,))
    _need(rel_ptr, torch.int32, "rel_ptr", (n_rel + 1,))
    _need(gptr, torch.int32, "gptr", (n_rel + 1,))
    _need(g_node, torch.int32, "g_node")
    width = d if folded else k
    if g_tab is None:
        g_tab = torch.empty((max(n_groups, 1), width), dtype=torch.float32, device=ent.device)
    else:
        _need(g_tab, torch.float32, "g_tab")
        if g_tab.numel() < max(n_groups, 1) * width:
            raise ValueError("g_tab holds %d floats, needs %d" % (g_tab.numel(), max(n_groups, 1) * width))
    logits = torch.empty(e, dtype=torch.float32, device=ent.device) if want_eid else None
    logits_csr = torch.empty(e, dtype=torch.float32, device=ent.device) if want_csr else None
    name = "kgat_att_score_folded_f32" if folded else "kgat_att_score_split_f32"
    extra = ((ATT_F32_PRODUCTS if f32_products else 0),) if folded else ()
    with _timed("att_score", (e, d, k)):
        check(getattr(_lib.load(), name)(n_nodes, e, d, k, n_rel, _ptr(rel_ptr), _ptr(perm), _ptr(src_g),
                                         _ptr(pos_g), _ptr(gid), _ptr(gptr), _ptr(g_node), n_groups,
                                         _ptr(ent), _ptr(W_R), _ptr(rel), _ptr(g_tab), _ptr(logits),
                                         _ptr(logits_csr), *extra, _stream(ent)), name)
    return logits, logits_csr


# positions per tile of the fused kernel: a 16-group block with more positions is cut and every
# piece recomputes the block's V rows (1,459 ticks) where walking on would cost 267 ticks per 64
# positions, so pieces should be long - but one wave walks a piece alone, and a piece must stay a
# small part of its workgroup's time for the eight waves to balance.  Measured on MI355X with the
# cost-balanced split (amazon-book-shaped CKG): 128: 0.2135 ms, 256: 0.2019.
FOLD_TILE_CAP = 256
FOLD_TILE_CAP32 = 512   # 32-group tiles (d = 64): the same positions per head group
# per-tile cost model of the fused kernel (kgat_fold_tile_parts; from per-workgroup clock stamps,
# scripts/micro/att_stamps.py): a tile, a chunk of 64 positions past the first 64, a relation
# change inside a workgroup's range.  With the bf16-piece products (d % 32 == 0) the fit is 649
# ticks of workgroup time per tile, 389 per later chunk, 10,661 per relation change; with the fp32
# products 1,459 / 267 / 10,630.  Both in units of a 64th of a tile.  Round 3 re-scanned the triple
# for the kernel with packed records and coalesced stores.  Stand-alone launches (variants
# alternating, every launch with another tile split; the tool went with round 5's clean-up) preferred
# cheaper chunks - (64,24,800) 0.145 ms against 0.151 - but INSIDE the step, where the launch starts
# with the table and the index arrays partly evicted, the order is the opposite (KGAT_FOLD_TILE_COST
# A/B of bench.py on one box: (64,38,1051) 142.5-144.0 us, (64,30,1051) 145.2-146.4, (64,30,800)
# 146.4-146.5, (64,24,800) 154.1-154.6): later chunks, whose rows are not requested ahead, cost
# more there.  The step is what counts: the fitted triple stays.
FOLD_TILE_COST = (64, 38, 1051)
FOLD_TILE_COST_F32 = (64, 12, 466)
# d = 128 (att_fold_fused128_kernel): scanned inside the step on the amazon-book-shaped CKG
# (scripts/micro/att128_cost_scan.sh).  With round 5's products (two fp16 pieces per operand, three piece products:
# half the MFMAs of rounds 3-4) a later chunk weighs more against a tile's products than it did: (64,12,700) - round
# 3's choice - 0.385 ms, (64,18,700) 0.352, (64,24,700) 0.350, (64,32,700) 0.361, (64,24,400) 0.355, (64,24,1200) 0.351,
# (64,40,1000) 0.376.
FOLD_TILE_COST_128 = (64, 24, 700)


FOLD_TILE_COST32 = (110, 38, 1051)   # 32-group tiles: a tile's MFMA phase serves twice the groups (round 5)


def fold_tile_cost(d, f32_products=False, groups_per_tile=16):
    """The split cost that goes with the product form att_score_fused takes at width d
    (``KGAT_FOLD_TILE_COST="tile,chunk,relation"`` overrides it: A/B runs of the whole step)."""
    from .options import options
    if options.fold_tile_cost:
        return options.fold_tile_cost
    if d == 128:
        return FOLD_TILE_COST_128
    if groups_per_tile == 32:
        return FOLD_TILE_COST32
    return FOLD_TILE_COST if (d % 32 == 0 and not f32_products) else FOLD_TILE_COST_F32


def fold_tiles(rel_ptr, gid, gptr, n_groups, cap=FOLD_TILE_CAP, n_parts=None, cost=FOLD_TILE_COST, groups_per_tile=16):
    """Work tiles of the fused attention kernel (kgat_fold_tiles) and their cost-balanced split
    over `n_parts` workgroups (kgat_fold_tile_parts; default: one per compute unit).  Returns
    (tiles (T_max, 4) int32, rel_tptr (R+1,) int32, part_tptr (n_parts+1,) int32); rel_tptr[-1]
    is the number of tiles in use."""
    lib = _lib.load()
    rel_ptr = _need(rel_ptr, torch.int32, "rel_ptr")
    gptr = _need(gptr, torch.int32, "gptr", rel_ptr.shape)
    gid = _need(gid, torch.int32, "gid")
    n_rel = rel_ptr.numel() - 1
    e = gid.numel()
    dev = gid.device
    t_max = max(int(lib.kgat_fold_tiles_max(e, int(n_groups), n_rel, int(cap))), 1)
    tiles = torch.zeros((t_max, 4), dtype=torch.int32, device=dev)
    rel_tptr = torch.zeros(n_rel + 1, dtype=torch.int32, device=dev)
    ws = _workspace(lib.kgat_fold_tiles_workspace_bytes(int(n_groups), n_rel), dev)
    check(lib.kgat_fold_tiles(e, n_rel, int(n_groups), _ptr(rel_ptr), _ptr(gid), _ptr(gptr), int(cap),
                              int(groups_per_tile), _ptr(tiles), _ptr(rel_tptr), _ptr(ws), ws.numel(), _stream(gid)),
          "kgat_fold_tiles")
    if n_parts is None:
        n_parts = torch.cuda.get_device_properties(dev).multi_processor_count
    part_tptr = torch.zeros(int(n_parts) + 1, dtype=torch.int32, device=dev)
    ws2 = _workspace(lib.kgat_fold_tile_parts_workspace_bytes(t_max), dev)
    check(lib.kgat_fold_tile_parts(t_max, n_rel, _ptr(tiles), _ptr(rel_tptr), int(n_parts), int(cost[0]), int(cost[1]),
                                   int(cost[2]), _ptr(part_tptr), _ptr(ws2), ws2.numel(), _stream(gid)),
          "kgat_fold_tile_parts")
    return tiles, rel_tptr, part_tptr


def att_score_fused_supported(n_nodes, d, k, n_rel):
    return bool(_lib.load().kgat_att_score_fused_supported(int(n_nodes), int(d), int(k), int(n_rel)))


def att_pack_records(rel_ptr, gptr, gid, src_g, groups_per_tile=16):
    """rec_g[p] = src_g[p] | (slot of p's head group in its 16-group block << 28; 32-group block << 27): the one
    index the fused attention kernel reads per grouped position (kgat_att_pack_records; graph-static)."""
    rel_ptr = _need(rel_ptr, torch.int32, "rel_ptr")
    gptr = _need(gptr, torch.int32, "gptr", rel_ptr.shape)
    src_g = _need(src_g, torch.int32, "src_g")
    e = src_g.numel()
    gid = _need(gid, torch.int32, "gid", (e,))
    rec = torch.empty(e, dtype=torch.int32, device=src_g.device)
    check(_lib.load().kgat_att_pack_records(e, rel_ptr.numel() - 1, _ptr(rel_ptr), _ptr(gptr), _ptr(gid), _ptr(src_g),
                                            int(groups_per_tile), _ptr(rec), _stream(src_g)), "kgat_att_pack_records")
    return rec


def att_score_fused(n_nodes, rel_ptr, perm, src_g, pos_g, gid, gptr, g_node, tiles, rel_tptr, ent, W_R, rel,
                    want_csr=True, want_eid=True, part_tptr=None, f32_products=False, rec_g=None, want_grouped=False,
                    groups_per_tile=16, part_clocks=None):
    """Attention logits, fused folded form (kgat_att_score_fused_f32).  The kernel reads one packed
    record per grouped position (`rec_g`, att_pack_records; built here from `src_g` / `gid` when the
    caller does not keep one).  `part_tptr`: the tile range of every workgroup (fold_tiles); None:
    equal tile counts, one workgroup per compute unit.  `f32_products`: the two products on the fp32
    MFMA (KGAT_ATT_F32_PRODUCTS) instead of the three-bf16-piece products the kernel takes by
    default when d % 32 == 0.  `part_clocks` (int64, 2 x n_parts): measurement aid
    (kgat_att_score_fused_timed_f32) - every workgroup's start / end time in 100 MHz ticks.  Returns (logits edge-id
    order, logits CSR order) - unrequested ones None - and, with want_grouped, a third item: the logits in grouped order."""
    ent = _need(ent, torch.float32, "ent")
    n_rel, d, k = W_R.shape
    W_R = _need(W_R, torch.float32, "W_R")
    rel = _need(rel, torch.float32, "rel", (n_rel, k))
    _need(rel_ptr, torch.int32, "rel_ptr", (n_rel + 1,))
    _need(gptr, torch.int32, "gptr", (n_rel + 1,))
    if rec_g is None:
        rec_g = att_pack_records(rel_ptr, gptr, gid, src_g, groups_per_tile)
    rec_g = _need(rec_g, torch.int32, "rec_g")
    e = rec_g.numel()
    if want_eid:
        perm = _need(perm, torch.int32, "perm", (e,))
    if want_csr:
        pos_g = _need(pos_g, torch.int32, "pos_g", (e,))
    _need(rel_tptr, torch.int32, "rel_tptr", (n_rel + 1,))
    _need(g_node, torch.int32, "g_node")
    _need(tiles, torch.int32, "tiles")
    n_parts = 0
    if part_tptr is not None:
        part_tptr = _need(part_tptr, torch.int32, "part_tptr")
        n_parts = part_tptr.numel() - 1
    logits = torch.empty(e, dtype=torch.float32, device=ent.device) if want_eid else None
    logits_csr = torch.empty(e, dtype=torch.float32, device=ent.device) if want_csr else None
    logits_g = torch.empty(e, dtype=torch.float32, device=ent.device) if want_grouped else None
    flags = (ATT_F32_PRODUCTS if f32_products else 0) | (ATT_TILES32 if groups_per_tile == 32 else 0)
    args = (n_nodes, e, d, k, n_rel, _ptr(rel_ptr), _ptr(perm) if want_eid else None, _ptr(rec_g),
            _ptr(pos_g) if want_csr else None, _ptr(gptr), _ptr(g_node), _ptr(tiles), _ptr(rel_tptr), _ptr(part_tptr), n_parts,
            _ptr(ent), _ptr(W_R), _ptr(rel), _ptr(logits), _ptr(logits_csr), _ptr(logits_g), flags)
    with _timed("att_score", (e, d, k)):
        if part_clocks is None:
            check(_lib.load().kgat_att_score_fused_f32(*args, _stream(ent)), "kgat_att_score_fused_f32")
        else:
            part_clocks = _need(part_clocks, torch.int64, "part_clocks", (2 * n_parts,))
            check(_lib.load().kgat_att_score_fused_timed_f32(*args, _ptr(part_clocks), _stream(ent)),
                  "kgat_att_score_fused_timed_f32")
    return (logits, logits_csr, logits_g) if want_grouped else (logits, logits_csr)


def transr_supported(n_nodes, d, k, n_rel, batch):
    return bool(_lib.load().kgat_transr_supported(int(n_nodes), int(d), int(k), int(n_rel), int(batch)))


def transr_workspace(batch, d, k, n_rel, device):
    return _workspace(_lib.load().kgat_transr_workspace_bytes(batch, d, k, n_rel), device)


def transr_loss_grad(h, r, pos_t, neg_t, ent, W_R, rel, reg_lambda, want_grad=True, out=None, workspace=None):
    """TransR loss of a triplet batch and, if want_grad, its gradients with respect to the entity
    table (dense), W_R and the relation table (kgat_transr_loss_grad_f32).  Index tensors are
    int32.  Returns (loss 0-d tensor, grad_ent, grad_W, grad_rel) - gradients None without
    want_grad.  `out` = (loss, grad_ent, grad_W, grad_rel) and `workspace` reuse buffers of an earlier call."""
    ent = _need(ent, torch.float32, "ent")
    n_rel, d, k = W_R.shape
    W_R = _need(W_R, torch.float32, "W_R")
    rel = _need(rel, torch.float32, "rel", (n_rel, k))
    h = _need(h, torch.int32, "h")
    b = h.numel()
    for name, t in (("r", r), ("pos_t", pos_t), ("neg_t", neg_t)):
        _need(t, torch.int32, name, (b,))
    lib = _lib.load()
    dev = ent.device
    if out is not None:
        loss, g_ent, g_w, g_rel = out
        _need(g_ent, torch.float32, "grad_ent", ent.shape)
        _need(g_w, torch.float32, "grad_W", W_R.shape)
        _need(g_rel, torch.float32, "grad_rel", rel.shape)
    else:
        loss = torch.empty((), dtype=torch.float32, device=dev)
        g_ent = torch.empty_like(ent) if want_grad else None
        g_w = torch.empty_like(W_R) if want_grad else None
        g_rel = torch.empty_like(rel) if want_grad else None
    need = lib.kgat_transr_workspace_bytes(b, d, k, n_rel)
    ws = workspace if workspace is not None and workspace.numel() >= need else _workspace(need, dev)
    with _timed("transr", (b, d, k)):
        check(lib.kgat_transr_loss_grad_f32(ent.shape[0], n_rel, d, k, b, _ptr(h), _ptr(r), _ptr(pos_t), _ptr(neg_t),
                                            _ptr(ent), _ptr(W_R), _ptr(rel), float(reg_lambda), _ptr(loss),
                                            _ptr(g_ent), _ptr(g_w), _ptr(g_rel), _ptr(ws), ws.numel(), _stream(ent)),
              "kgat_transr_loss_grad_f32")
    return loss, g_ent, g_w, g_rel


def transr_presort(h, r, pos_t, neg_t, n_nodes, n_rel):
    """The sorts of a whole KG phase's batches in one launch (kgat_transr_presort_f32): h, r, pos_t, neg_t are
    (n_batches, batch) int32 tensors; returns (sorted, stride) - batch b's block is sorted[b * stride:(b + 1) * stride]."""
    h = _need(h, torch.int32, "h")
    if h.dim() != 2:
        raise ValueError("h must be (n_batches, batch)")
    for name, t in (("r", r), ("pos_t", pos_t), ("neg_t", neg_t)):
        _need(t, torch.int32, name, h.shape)
    lib = _lib.load()
    nb, b = h.shape
    stride = int(lib.kgat_transr_sorted_bytes(b, n_rel))
    out = torch.empty(max(nb * stride, 256), dtype=torch.uint8, device=h.device)
    with _timed("transr_presort", (nb, b)):
        check(lib.kgat_transr_presort_f32(int(n_nodes), int(n_rel), nb, b, _ptr(h), _ptr(r), _ptr(pos_t), _ptr(neg_t),
                                          _ptr(out), out.numel(), _stream(h)), "kgat_transr_presort_f32")
    return out, stride


class TransRAdamState:
    """What a sequence of kgat_transr_adam_step_f32 calls keeps between calls: the step workspace, the row_slot words
    (zero once, never cleared: a word is valid for the call whose tag it carries) and the tag counter."""

    def __init__(self, n_nodes, d, k, n_rel, batch, device):
        lib = _lib.load()
        self.key = (int(n_nodes), int(d), int(k), int(n_rel), int(batch), str(device))
        self.workspace = _workspace(lib.kgat_transr_step_workspace_bytes(batch, d, k, n_rel), device)
        self.row_slot = torch.zeros(int(n_nodes), dtype=torch.int64, device=device)
        self.tag = 0


def transr_adam_step(h, r, pos_t, neg_t, sorted_block, ent, W_R, rel, exp_avg, exp_avg_sq, steps, lr, beta1, beta2, eps,
                     reg_lambda, loss_out, state):
    """One KG iteration (kgat_transr_adam_step_f32): TransR loss of the batch into `loss_out` (a 1-element fp32 view)
    and the Adam step on ent / W_R / rel in place.  exp_avg / exp_avg_sq: ctypes arrays of the three moment pointers
    (built once per phase by the caller); steps: the three step counts after this step.  Shapes and dtypes are the
    caller's responsibility here (KGATPropagation.kg_phase validates them once per phase): this is the per-iteration
    call of a 1,641-iteration loop."""
    import ctypes as C
    n_rel, d, k = W_R.shape
    state.tag += 1
    arr_t = (C.c_int64 * 3)(*steps)
    check(_lib.load().kgat_transr_adam_step_f32(ent.shape[0], n_rel, d, k, h.numel(), h.data_ptr(), r.data_ptr(),
                                                pos_t.data_ptr(), neg_t.data_ptr(), sorted_block.data_ptr(), ent.data_ptr(),
                                                W_R.data_ptr(), rel.data_ptr(), exp_avg, exp_avg_sq, arr_t, float(lr),
                                                float(beta1), float(beta2), float(eps), float(reg_lambda),
                                                loss_out.data_ptr(), state.row_slot.data_ptr(), state.tag,
                                                state.workspace.data_ptr(), state.workspace.numel(), _stream(ent)),
          "kgat_transr_adam_step_f32")


def transr_forward(h, r, pos_t, neg_t, ent, W_R, rel, reg_lambda):
    """TransR loss with the per-sample rows left in the returned workspace for transr_backward
    (kgat_transr_forward_f32)."""
    ent = _need(ent, torch.float32, "ent")
    n_rel, d, k = W_R.shape
    W_R = _need(W_R, torch.float32, "W_R")
    rel = _need(rel, torch.float32, "rel", (n_rel, k))
    h = _need(h, torch.int32, "h")
    b = h.numel()
    for name, t in (("r", r), ("pos_t", pos_t), ("neg_t", neg_t)):
        _need(t, torch.int32, name, (b,))
    lib = _lib.load()
    loss = torch.empty((), dtype=torch.float32, device=ent.device)
    ws = _workspace(lib.kgat_transr_workspace_bytes(b, d, k, n_rel), ent.device)
    with _timed("transr_forward", (b, d, k)):
        check(lib.kgat_transr_forward_f32(ent.shape[0], n_rel, d, k, b, _ptr(h), _ptr(r), _ptr(pos_t), _ptr(neg_t),
                                          _ptr(ent), _ptr(W_R), _ptr(rel), float(reg_lambda), _ptr(loss), _ptr(ws),
                                          ws.numel(), _stream(ent)), "kgat_transr_forward_f32")
    return loss, ws


def transr_backward(h, r, pos_t, neg_t, shapes, ws, grad_scale=None):
    """The three gradients of transr_forward's loss (kgat_transr_backward_f32), times the device scalar grad_scale."""
    (n_nodes, d), (n_rel, _, k) = shapes
    dev = ws.device
    g_ent = torch.empty((n_nodes, d), dtype=torch.float32, device=dev)
    g_w = torch.empty((n_rel, d, k), dtype=torch.float32, device=dev)
    g_rel = torch.empty((n_rel, k), dtype=torch.float32, device=dev)
    if grad_scale is not None:
        grad_scale = _need(grad_scale.reshape(1), torch.float32, "grad_scale")
    with _timed("transr_backward", (h.numel(), d, k)):
        check(_lib.load().kgat_transr_backward_f32(n_nodes, n_rel, d, k, h.numel(), _ptr(h), _ptr(r), _ptr(pos_t),
                                                   _ptr(neg_t), _ptr(grad_scale), _ptr(g_ent), _ptr(g_w), _ptr(g_rel),
                                                   _ptr(ws), ws.numel(), _stream(ws)), "kgat_transr_backward_f32")
    return g_ent, g_w, g_rel


def edge_softmax(indptr, row_of, eid, logits, in_csr_order=False, e_range=None, want_out=True,
                 want_csr=False, three_pass=False):
    """Softmax over each destination's in-edges (`indptr`, `row_of`, `eid` of the destination-major
    CSR).  `logits` (E,) is in edge-id order, or in CSR order with in_csr_order=True.  Returns
    (out in edge-id order, out_csr in CSR order); unrequested ones are None.  three_pass=True runs
    the independent three-pass implementation (kgat_edge_softmax_3pass_f32)."""
    logits = _need(logits, torch.float32, "logits")
    indptr = _need(indptr, torch.int32, "indptr")
    row_of = _need(row_of, torch.int32, "row_of", logits.shape)
    eid = _need(eid, torch.int32, "eid", logits.shape)
    n_nodes = indptr.numel() - 1
    lib = _lib.load()
    e0, e1 = (0, logits.numel()) if e_range is None else e_range
    out = torch.empty_like(logits) if want_out else None
    out_csr = torch.empty_like(logits) if want_csr else None
    with _timed("edge_softmax", (e1 - e0,)):
        if three_pass:
            ws = _workspace(lib.kgat_edge_softmax_3pass_workspace_bytes(n_nodes), logits.device)
            check(lib.kgat_edge_softmax_3pass_f32(n_nodes, e0, e1, _ptr(row_of), _ptr(eid), _ptr(logits),
                                                  1 if in_csr_order else 0, _ptr(out), _ptr(out_csr), _ptr(ws),
                                                  ws.numel(), _stream(logits)), "kgat_edge_softmax_3pass_f32")
        else:
            ws = _workspace(lib.kgat_edge_softmax_workspace_bytes(n_nodes, e1 - e0), logits.device)
            check(lib.kgat_edge_softmax_f32(n_nodes, e0, e1, _ptr(indptr), _ptr(row_of), _ptr(eid), _ptr(logits),
                                            1 if in_csr_order else 0, _ptr(out), _ptr(out_csr), _ptr(ws),
                                            ws.numel(), _stream(logits)), "kgat_edge_softmax_f32")
    return out, out_csr


def edge_softmax_bwd(indptr, eid, a, grad_a, row_range=None):
    a = _need(a, torch.float32, "a")
    grad_a = _need(grad_a, torch.float32, "grad_a", a.shape)
    indptr = _need(indptr, torch.int32, "indptr")
    if eid is not None:
        eid = _need(eid, torch.int32, "eid", a.shape)
    row0, n_rows = (0, indptr.numel() - 1) if row_range is None else row_range
    out = torch.zeros_like(a)
    check(_lib.load().kgat_edge_softmax_bwd_f32(n_rows, row0, _ptr(indptr), _ptr(eid), _ptr(a),
                                                _ptr(grad_a), _ptr(out), _stream(a)),
          "kgat_edge_softmax_bwd_f32")
    return out


def spmm_workspace(n_edges, D, device):
    return _workspace(_lib.load().kgat_spmm_workspace_bytes(n_edges, D), device)


class DeferredRows:
    """What spmm(defer_finish=True) leaves to bi_interaction_mul(deferred=...): the row offsets of the call's rows, its
    CSR position range, the workspace holding the tiles' boundary partials and the tile size."""
    __slots__ = ("indptr_rows", "e_range", "workspace", "tile_edges", "n_rows", "D")

    def __init__(self, indptr_rows, e_range, workspace, tile_edges, n_rows, D):
        self.indptr_rows, self.e_range, self.workspace = indptr_rows, e_range, workspace
        self.tile_edges, self.n_rows, self.D = tile_edges, n_rows, D


def bi_interaction_deferral_supported(d_in, d_out):
    """The widths kgat_spmm_umule_sum_f32(KGAT_SPMM_DEFER_FINISH) + kgat_bi_interaction_mul_deferred_f32 cover."""
    return int(d_in) in (16, 32, 64, 128) and int(d_out) in (16, 32, 64, 128)


def spmm(indptr, col, row_of, X, w, eid=None, out=None, order=None, mul_self=False, algo="auto",
         rows=None, e_range=None, workspace=None, self_out=None, defer_finish=False):
    """out[v - row0] = sum_p w_p X[col[p]] over the CSR rows `rows` = (row0, n_rows) whose CSR
    positions are `e_range` (defaults: the whole graph).  w is in CSR order, or in edge-id
    order when `eid` is given.  `self_out` (with mul_self): an (n_rows, D) column slice of a wider
    row-major buffer that also receives X[v] (the ego block of the readout).
    defer_finish=True (KGAT_SPMM_DEFER_FINISH): returns (out, DeferredRows) - the rows the edge tiles cut and the
    rows without in-edges are NOT in `out`; bi_interaction_mul(deferred=...) forms them on the way."""
    X = _need(X, torch.float32, "X")
    if X.dim() != 2:
        raise ValueError("X must be (N, D)")
    D = X.shape[1]
    indptr = _need(indptr, torch.int32, "indptr")
    col = _need(col, torch.int32, "col")
    w = _need(w, torch.float32, "w", col.shape)
    if row_of is not None:
        row_of = _need(row_of, torch.int32, "row_of", col.shape)
    if eid is not None:
        eid = _need(eid, torch.int32, "eid", col.shape)
    if order is not None:
        order = _need(order, torch.int32, "order")
    row0, n_rows = (0, indptr.numel() - 1) if rows is None else rows
    e0, e1 = (0, col.numel()) if e_range is None else e_range
    if out is None:
        out = torch.empty((n_rows, D), dtype=torch.float32, device=X.device)
    else:
        out = _need(out, torch.float32, "out", (n_rows, D))
    if workspace is None:
        workspace = spmm_workspace(e1 - e0, D, X.device)
    self_stride = 0
    if self_out is not None:
        self_stride = _strided_rows(self_out, n_rows, D, "self_out")
    with _timed("spmm", (e1 - e0, n_rows, D)):
        check(_lib.load().kgat_spmm_umule_sum_f32(n_rows, row0, e0, e1, D, _ptr(indptr), _ptr(col), _ptr(row_of),
                                                  _ptr(eid), _ptr(X), _ptr(w), _ptr(out), _ptr(order),
                                                  _ptr(workspace), workspace.numel(),
                                                  (SPMM_MUL_SELF if mul_self else 0) | (SPMM_DEFER_FINISH if defer_finish else 0),
                                                  SPMM_ALGO[algo], _ptr(self_out), self_stride, _stream(X)),
              "kgat_spmm_umule_sum_f32")
    if defer_finish:
        te = int(_lib.load().kgat_spmm_tile_edges(e1 - e0, D))
        return out, DeferredRows(indptr[row0:row0 + n_rows + 1], (e0, e1), workspace, te, n_rows, D)
    return out


def spmm_bi_fused_supported(d_in, d_out):
    return bool(_lib.load().kgat_spmm_bi_fused_supported(int(d_in), int(d_out)))


def spmm_bi_fused(indptr, col, row_of, X, w, W2, negative_slope=0.01, h_out=None, norm_out=None, want_h=True,
                  rows=None, e_range=None, workspace=None, scratch=None, self_out=None):
    """One KGATConv forward in one pass (kgat_spmm_bi_fused_f32; reference models.py:63-66 + :165):
    Z = leaky_relu(((sum_p w_p X[col[p]]) * X[v]) @ W2^T) over the CSR rows `rows`; returns Z (or None with
    want_h=False) and writes Z / ||Z_row|| into `norm_out` (a column slice of a wider row-major buffer).  The
    same bits as spmm(mul_self=True) followed by bi_interaction."""
    X = _need(X, torch.float32, "X")
    W2 = _need(W2, torch.float32, "W2")
    if X.dim() != 2 or W2.dim() != 2 or W2.shape[1] != X.shape[1]:
        raise ValueError("X must be (N, d_in) and W2 (d_out, d_in)")
    d_in, d_out = X.shape[1], W2.shape[0]
    indptr = _need(indptr, torch.int32, "indptr")
    col = _need(col, torch.int32, "col")
    w = _need(w, torch.float32, "w", col.shape)
    row_of = _need(row_of, torch.int32, "row_of", col.shape)
    row0, n_rows = (0, indptr.numel() - 1) if rows is None else rows
    e0, e1 = (0, col.numel()) if e_range is None else e_range
    if want_h and h_out is None:
        h_out = torch.empty((n_rows, d_out), dtype=torch.float32, device=X.device)
    if h_out is not None:
        h_out = _need(h_out, torch.float32, "h_out", (n_rows, d_out))
    stride = 0
    if norm_out is not None:
        stride = _strided_rows(norm_out, n_rows, d_out, "norm_out")
    self_stride = 0
    if self_out is not None:
        self_stride = _strided_rows(self_out, n_rows, d_in, "self_out")
    if scratch is None:
        scratch = torch.empty((n_rows, d_in), dtype=torch.float32, device=X.device)
    else:
        scratch = _need(scratch, torch.float32, "scratch", (n_rows, d_in))
    if workspace is None:
        workspace = spmm_workspace(e1 - e0, d_in, X.device)
    with _timed("spmm_bi_fused", (e1 - e0, n_rows, d_in, d_out)):
        check(_lib.load().kgat_spmm_bi_fused_f32(n_rows, row0, e0, e1, d_in, d_out, _ptr(indptr), _ptr(col),
                                                 _ptr(row_of), _ptr(X), _ptr(w), _ptr(W2), float(negative_slope),
                                                 _ptr(h_out), _ptr(norm_out), stride, _ptr(scratch), _ptr(workspace),
                                                 workspace.numel(), _ptr(self_out), self_stride, _stream(X)),
              "kgat_spmm_bi_fused_f32")
    return h_out


def bi_interaction_supported(d_in, d_out):
    return bool(_lib.load().kgat_bi_interaction_supported(int(d_in), int(d_out)))


def bi_interaction(P, W2, negative_slope=0.01, h_out=None, norm_out=None, want_h=True):
    """Z = leaky_relu(P @ W2^T); returns Z (next layer's input) and writes Z / ||Z_row|| into
    `norm_out`, which may be a column slice of a wider row-major buffer."""
    P = _need(P, torch.float32, "P")
    W2 = _need(W2, torch.float32, "W2")
    n, d_in = P.shape
    d_out = W2.shape[0]
    if W2.shape[1] != d_in:
        raise ValueError("W2 has shape %s, expected (*, %d)" % (tuple(W2.shape), d_in))
    if want_h and h_out is None:
        h_out = torch.empty((n, d_out), dtype=torch.float32, device=P.device)
    if h_out is not None:
        h_out = _need(h_out, torch.float32, "h_out", (n, d_out))
    stride = 0
    if norm_out is not None:
        if (not norm_out.is_cuda or norm_out.dtype != torch.float32 or tuple(norm_out.shape) != (n, d_out)
                or norm_out.stride(1) != 1):
            raise ValueError("norm_out must be an (n, d_out) float32 device view with unit column stride")
        stride = norm_out.stride(0)
    with _timed("bi_interaction", (n, d_in, d_out)):
        check(_lib.load().kgat_bi_interaction_f32(n, d_in, d_out, _ptr(P), _ptr(W2), float(negative_slope),
                                                  _ptr(h_out), _ptr(norm_out), stride, _stream(P)),
              "kgat_bi_interaction_f32")
    return h_out


def bi_interaction_mul(H, HN, W2, negative_slope=0.01, h_out=None, norm_out=None, want_h=True, self_out=None,
                       deferred=None):
    """Z = leaky_relu((H * HN) @ W2^T) (kgat_bi_interaction_mul_f32): the layer input H and the plain aggregation HN,
    the product formed while the rows are loaded; `self_out`: an (n, d_in) column slice that also receives H (the
    ego block of the readout).  Otherwise as bi_interaction.  `deferred`: the DeferredRows of the
    spmm(defer_finish=True) call that produced HN (kgat_bi_interaction_mul_deferred_f32; same bits)."""
    H = _need(H, torch.float32, "H")
    HN = _need(HN, torch.float32, "HN", H.shape)
    W2 = _need(W2, torch.float32, "W2")
    n, d_in = H.shape
    d_out = W2.shape[0]
    if W2.shape[1] != d_in:
        raise ValueError("W2 has shape %s, expected (*, %d)" % (tuple(W2.shape), d_in))
    if want_h and h_out is None:
        h_out = torch.empty((n, d_out), dtype=torch.float32, device=H.device)
    if h_out is not None:
        h_out = _need(h_out, torch.float32, "h_out", (n, d_out))
    stride = _strided_rows(norm_out, n, d_out, "norm_out") if norm_out is not None else 0
    self_stride = _strided_rows(self_out, n, d_in, "self_out") if self_out is not None else 0
    with _timed("bi_interaction", (n, d_in, d_out)):
        if deferred is not None:
            if deferred.n_rows != n or deferred.D != d_in:
                raise ValueError("deferred rows of a (%d, %d) aggregation with a (%d, %d) input" % (deferred.n_rows, deferred.D, n, d_in))
            check(_lib.load().kgat_bi_interaction_mul_deferred_f32(
                n, d_in, d_out, _ptr(H), _ptr(HN), _ptr(W2), float(negative_slope), _ptr(h_out), _ptr(norm_out), stride,
                _ptr(self_out), self_stride, _ptr(deferred.indptr_rows), deferred.e_range[0], deferred.e_range[1],
                _ptr(deferred.workspace), deferred.tile_edges, _stream(H)), "kgat_bi_interaction_mul_deferred_f32")
        else:
            check(_lib.load().kgat_bi_interaction_mul_f32(n, d_in, d_out, _ptr(H), _ptr(HN), _ptr(W2), float(negative_slope),
                                                          _ptr(h_out), _ptr(norm_out), stride, _ptr(self_out), self_stride,
                                                          _stream(H)),
                  "kgat_bi_interaction_mul_f32")
    return h_out


def _strided_rows(t, n, d, name):
    if (not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (n, d) or t.stride(1) != 1):
        raise ValueError("%s must be an (n, %d) float32 device view with unit column stride" % (name, d))
    return t.stride(0)


def bi_interaction_train(H, HN, W2, negative_slope, drop_p, seed, norm_out=None, row0=0, self_out=None):
    """Training form: h_out = dropout_p(leaky_relu((H * HN) @ W2^T)) and its normalised copy into
    `norm_out` (kgat_bi_interaction_train_f32; the mask is a hash of (seed, element); `row0`: the
    global index of row 0 when H holds a row range of a larger matrix, so that a destination shard
    draws the mask the unsharded layer draws)."""
    H = _need(H, torch.float32, "H")
    HN = _need(HN, torch.float32, "HN", H.shape)
    W2 = _need(W2, torch.float32, "W2")
    n, d_in = H.shape
    d_out = W2.shape[0]
    if W2.shape[1] != d_in:
        raise ValueError("W2 has shape %s, expected (*, %d)" % (tuple(W2.shape), d_in))
    h_out = torch.empty((n, d_out), dtype=torch.float32, device=H.device)
    stride = _strided_rows(norm_out, n, d_out, "norm_out") if norm_out is not None else 0
    self_stride = _strided_rows(self_out, n, d_in, "self_out") if self_out is not None else 0   # the ego block (copy of H)
    with _timed("bi_interaction", (n, d_in, d_out)):
        check(_lib.load().kgat_bi_interaction_train_f32(n, d_in, d_out, _ptr(H), _ptr(HN), _ptr(W2),
                                                        float(negative_slope), float(drop_p), int(seed) & (2 ** 64 - 1),
                                                        int(row0), _ptr(h_out), _ptr(norm_out), stride, _ptr(self_out),
                                                        self_stride, _stream(H)),
              "kgat_bi_interaction_train_f32")
    return h_out


def add3_rows(a, b, c):
    """(a + b) + c where `a` is an (n, d) column slice of a wider fp32 matrix and b, c are contiguous (n, d)
    (kgat_add3_rows_f32): the three gradient paths into the embedding table in one pass."""
    b = _need(b, torch.float32, "b")
    c = _need(c, torch.float32, "c", b.shape)
    n, d = b.shape
    stride = _strided_rows(a, n, d, "a")
    out = torch.empty_like(b)
    check(_lib.load().kgat_add3_rows_f32(n, d, _ptr(a), stride, _ptr(b), _ptr(c), _ptr(out), _stream(b)), "kgat_add3_rows_f32")
    return out


def bi_interaction_bwd_pre(h_out, grad_a, grad_b, grad_norm, negative_slope, drop_p, seed, row0=0):
    """grad_z of the training layer (kgat_bi_interaction_bwd_pre_f32); grad_a / grad_b / grad_norm may be
    None; `row0` as in bi_interaction_train."""
    h_out = _need(h_out, torch.float32, "h_out")
    n, d = h_out.shape
    for name, t in (("grad_a", grad_a), ("grad_b", grad_b)):
        if t is not None:
            _need(t, torch.float32, name, (n, d))
    stride = _strided_rows(grad_norm, n, d, "grad_norm") if grad_norm is not None else 0
    gz = torch.empty_like(h_out)
    check(_lib.load().kgat_bi_interaction_bwd_pre_f32(n, d, _ptr(h_out), _ptr(grad_a), _ptr(grad_b), _ptr(grad_norm),
                                                      stride, float(negative_slope), float(drop_p),
                                                      int(seed) & (2 ** 64 - 1), int(row0), _ptr(gz), _stream(h_out)),
          "kgat_bi_interaction_bwd_pre_f32")
    return gz


def bi_interaction_bwd_input_supported(d_in, d_out):
    return bool(_lib.load().kgat_bi_interaction_bwd_input_supported(int(d_in), int(d_out)))


def bi_interaction_bwd_input(grad_z, W2, H, HN):
    """((grad_z @ W2) * H, (grad_z @ W2) * HN) in one pass (kgat_bi_interaction_bwd_input_f32): the gradient of the
    layer's dense part towards h_N (times h: what the reversed-CSR aggregation then sums) and towards h itself."""
    grad_z = _need(grad_z, torch.float32, "grad_z")
    W2 = _need(W2, torch.float32, "W2")
    n, d_out = grad_z.shape
    d_in = W2.shape[1]
    if W2.shape[0] != d_out:
        raise ValueError("W2 has shape %s, expected (%d, *)" % (tuple(W2.shape), d_out))
    H = _need(H, torch.float32, "H", (n, d_in))
    HN = _need(HN, torch.float32, "HN", (n, d_in))
    t, gb = torch.empty_like(H), torch.empty_like(H)
    check(_lib.load().kgat_bi_interaction_bwd_input_f32(n, d_in, d_out, _ptr(grad_z), _ptr(W2), _ptr(H), _ptr(HN), _ptr(t),
                                                        _ptr(gb), _stream(H)), "kgat_bi_interaction_bwd_input_f32")
    return t, gb


def sum_partials(partial_sets):
    """[(n_partials, ...) tensor, ...] -> [sum over dim 0, ...] for up to four sets in one launch
    (kgat_sum_partials_f32; a fixed order of additions)."""
    import ctypes as C
    outs = []
    for lo in range(0, len(partial_sets), 4):
        sets = [_need(t, torch.float32, "partials") for t in partial_sets[lo:lo + 4]]
        res = [torch.empty(t.shape[1:], dtype=torch.float32, device=t.device) for t in sets]
        n = len(sets)
        if any(t[0].numel() % 4 for t in sets):
            raise ValueError("sum_partials: partial sizes must be multiples of 4 floats")
        check(_lib.load().kgat_sum_partials_f32(n, (C.c_void_p * n)(*[t.data_ptr() for t in sets]),
                                                (C.c_void_p * n)(*[r.data_ptr() for r in res]),
                                                (C.c_int64 * n)(*[t.shape[0] for t in sets]),
                                                (C.c_int64 * n)(*[t[0].numel() for t in sets]), _stream(sets[0])),
              "kgat_sum_partials_f32")
        outs += res
    return outs


def bi_interaction_bwd_weight(grad_z, H, HN, want_partials=False):
    """grad_W2 = grad_z^T (H * HN) (kgat_bi_interaction_bwd_weight_f32: per-workgroup partials over 64-row slabs, the
    product formed on the way; the partials are added here in index order - or, want_partials=True, handed back for
    ops.sum_partials, which sums several layers' sets in one launch)."""
    grad_z = _need(grad_z, torch.float32, "grad_z")
    n, d_out = grad_z.shape
    H = _need(H, torch.float32, "H")
    d_in = H.shape[1]
    HN = _need(HN, torch.float32, "HN", (n, d_in))
    if H.shape[0] != n:
        raise ValueError("H has %d rows, grad_z %d" % (H.shape[0], n))
    lib = _lib.load()
    nb = int(lib.kgat_bi_interaction_bwd_weight_partials(n))
    partials = torch.empty((nb, d_out, d_in), dtype=torch.float32, device=H.device)
    check(lib.kgat_bi_interaction_bwd_weight_f32(n, d_in, d_out, _ptr(grad_z), _ptr(H), _ptr(HN), _ptr(partials), nb,
                                                 _stream(H)), "kgat_bi_interaction_bwd_weight_f32")
    return partials if want_partials else partials.sum(0)


def mul2(a, b, c):
    """(a * b, a * c) in one pass."""
    a = _need(a, torch.float32, "a")
    b = _need(b, torch.float32, "b", a.shape)
    c = _need(c, torch.float32, "c", a.shape)
    ab, ac = torch.empty_like(a), torch.empty_like(a)
    check(_lib.load().kgat_mul2_f32(a.numel(), _ptr(a), _ptr(b), _ptr(c), _ptr(ab), _ptr(ac), _stream(a)), "kgat_mul2_f32")
    return ab, ac


def dropout_keep_mask(seed, n_rows, d, drop_p, row0=0):
    """The mask kgat_bi_interaction_train_f32 applies, restated in numpy (tests): element (row, col)
    is kept iff murmur3-finalised (((row0 + row)*d + col) * 0x9E3779B1 ^ seed32) >= p * 2^32."""
    import numpy as np
    seed = int(seed) & (2 ** 64 - 1)
    seed32 = np.uint32((seed ^ (seed >> 32)) & 0xFFFFFFFF)
    x = ((np.arange(n_rows * d, dtype=np.uint64) + np.uint64(int(row0) * d)) * np.uint64(0x9E3779B1)).astype(np.uint32) ^ seed32
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x85EBCA6B)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x.astype(np.uint64) * np.uint64(0xC2B2AE35)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    t = min(int(float(np.float32(drop_p)) * 4294967296.0), 0xFFFFFFFF)
    return (x >= np.uint32(t)).reshape(n_rows, d)


def l2_normalize_rows(x, out):
    """out[:, :] = x / max(||x_row||, 1e-12); `out` may be a column slice of a wider buffer."""
    x = _need(x, torch.float32, "x")
    if (not out.is_cuda or out.dtype != torch.float32 or tuple(out.shape) != tuple(x.shape) or out.stride(1) != 1):
        raise ValueError("out must be a float32 device view of x's shape with unit column stride")
    check(_lib.load().kgat_l2_normalize_rows_f32(x.shape[0], x.shape[1], _ptr(x), _ptr(out), out.stride(0), _stream(x)),
          "kgat_l2_normalize_rows_f32")
    return out


def readout_concat(blocks, normalize, out=None):
    """[b0 | b1 | ...] with the blocks flagged in `normalize` L2-normalised per row
    (kgat_readout_concat_f32): the readout of Model.gnn from separately held layer outputs."""
    import ctypes as C
    n = blocks[0].shape[0]
    widths = []
    for i, b in enumerate(blocks):
        _need(b, torch.float32, "blocks[%d]" % i)
        if b.dim() != 2 or b.shape[0] != n:
            raise ValueError("blocks must be (n, w_i) matrices with the same n")
        widths.append(int(b.shape[1]))
    if out is None:
        out = torch.empty((n, sum(widths)), dtype=torch.float32, device=blocks[0].device)
    else:
        out = _need(out, torch.float32, "out", (n, sum(widths)))
    k = len(blocks)
    ptrs = (C.c_void_p * k)(*[b.data_ptr() for b in blocks])
    w_arr = (C.c_int * k)(*widths)
    f_arr = (C.c_int * k)(*[1 if f else 0 for f in normalize])
    check(_lib.load().kgat_readout_concat_f32(n, k, ptrs, w_arr, f_arr, _ptr(out), out.stride(0), _stream(out)),
          "kgat_readout_concat_f32")
    return out


def gather_probe(col, X, sink=None):
    """Measurement aid (kgat_gather_probe_f32): fetch the rows X[col[p]] with the aggregation's access pattern and
    do nothing else; returns the scratch so a caller timing many launches can pass it back in."""
    X = _need(X, torch.float32, "X")
    col = _need(col, torch.int32, "col")
    if sink is None:
        sink = torch.zeros(((col.numel() + 2047) // 2048 * 4 + 4) * 4, dtype=torch.float32, device=X.device)
    with _timed("gather_probe", (col.numel(), X.shape[1])):
        check(_lib.load().kgat_gather_probe_f32(col.numel(), X.shape[1], _ptr(col), _ptr(X), _ptr(sink), _stream(X)),
              "kgat_gather_probe_f32")
    return sink


def sddmm_dot(src, dst, X, G):
    X = _need(X, torch.float32, "X")
    G = _need(G, torch.float32, "grad_out")
    src = _need(src, torch.int32, "src")
    dst = _need(dst, torch.int32, "dst", src.shape)
    if X.shape[1] != G.shape[1]:
        raise ValueError("feature widths differ")
    out = torch.empty(src.numel(), dtype=torch.float32, device=X.device)
    check(_lib.load().kgat_sddmm_dot_f32(src.numel(), X.shape[1], _ptr(src), _ptr(dst), _ptr(X), _ptr(G),
                                         _ptr(out), _stream(X)), "kgat_sddmm_dot_f32")
    return out


__all__ = ["csr_from_coo", "group_by_relation", "invert_permutation", "row_order_by_degree", "gather",
           "att_score", "att_score_split", "att_score_split_supported", "att_score_folded_supported", "att_score_fused", "att_pack_records", "att_score_fused_supported", "fold_tiles", "fold_tile_cost", "bi_interaction_train", "add3_rows", "bi_interaction_bwd_pre", "bi_interaction_bwd_input", "bi_interaction_bwd_input_supported", "bi_interaction_bwd_weight", "sum_partials", "mul2", "dropout_keep_mask", "transr_loss_grad", "transr_supported", "transr_presort", "transr_adam_step", "TransRAdamState", "head_groups", "edge_softmax", "edge_softmax_bwd", "spmm", "spmm_workspace", "sddmm_dot",
           "bi_interaction", "bi_interaction_supported", "l2_normalize_rows", "readout_concat",
           "KGATLibraryError"]


def eval_supported(F, K):
    return bool(_lib.load().kgat_eval_supported(int(F), int(K)))


def eval_recall_ndcg(emb, user_ids, item_ids, train_ptr, train_items, test_ptr, test_items, K, want_topk=False):
    """recall@K / ndcg@K per test user (reference metric.py:36-68) on the device: kgat_eval_items_kmajor_f32 +
    kgat_eval_recall_ndcg_f32.  `emb` (N, F) fp32 rows (row stride = emb.stride(0)); user_ids / item_ids int32 node
    ids; the CSR lists hold item POSITIONS (ascending per user).  Returns (recall, ndcg[, topk]) - float64 (n_users,)
    and int32 (n_users, K)."""
    if emb.dtype != torch.float32 or not emb.is_cuda or emb.dim() != 2 or emb.stride(1) != 1:
        raise KGATLibraryError("eval_recall_ndcg: `emb` must be a float32 HIP matrix with unit column stride")
    F, stride = emb.shape[1], emb.stride(0)
    user_ids = _need(user_ids, torch.int32, "user_ids")
    item_ids = _need(item_ids, torch.int32, "item_ids")
    n_users, n_items = user_ids.numel(), item_ids.numel()
    train_ptr = _need(train_ptr, torch.int32, "train_ptr", (n_users + 1,))
    test_ptr = _need(test_ptr, torch.int32, "test_ptr", (n_users + 1,))
    train_items = _need(train_items, torch.int32, "train_items")
    test_items = _need(test_items, torch.int32, "test_items")
    lib = _lib.load()
    if not lib.kgat_eval_supported(F, K):
        raise KGATLibraryError("eval_recall_ndcg: K = %d / F = %d outside the kernel's range" % (K, F))
    dev = emb.device
    import numpy as np
    disc = torch.as_tensor(1.0 / np.log2(np.arange(2, K + 2)), dtype=torch.float64, device=dev)
    recall = torch.zeros(n_users, dtype=torch.float64, device=dev)
    ndcg = torch.zeros(n_users, dtype=torch.float64, device=dev)
    topk = torch.empty((n_users, K), dtype=torch.int32, device=dev) if want_topk else None
    if n_users == 0:
        return (recall, ndcg, topk) if want_topk else (recall, ndcg)
    itemT = torch.empty(max(int(lib.kgat_eval_items_elems(n_items, F)), 1), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.kgat_eval_workspace_bytes(n_users, n_items, F, K), dtype=torch.uint8, device=dev)
    with _timed("eval_items_kmajor", (n_items, F)):
        check(lib.kgat_eval_items_kmajor_f32(n_items, F, _ptr(emb), stride, _ptr(item_ids), _ptr(itemT), _stream(emb)),
              "kgat_eval_items_kmajor_f32")
    with _timed("eval_recall_ndcg", (n_users, n_items, F, K)):
        check(lib.kgat_eval_recall_ndcg_f32(n_users, _ptr(user_ids), n_items, F, _ptr(emb), stride, _ptr(itemT),
                                            _ptr(train_ptr), _ptr(train_items), _ptr(test_ptr), _ptr(test_items), K,
                                            _ptr(disc), _ptr(ws), ws.numel(), _ptr(recall), _ptr(ndcg),
                                            _ptr(topk) if want_topk else None, _stream(emb)),
              "kgat_eval_recall_ndcg_f32")
    return (recall, ndcg, topk) if want_topk else (recall, ndcg)


def bpr_loss(emb, u, p, n, reg_lambda):
    """BPR loss of reference models.py:170-178 on the readout (kgat_bpr_loss_f32).  Returns (loss (0-dim), coef (B,),
    workspace) - the last two feed bpr_grad."""
    if emb.dtype != torch.float32 or not emb.is_cuda or emb.dim() != 2 or emb.stride(1) != 1:
        raise KGATLibraryError("bpr_loss: `emb` must be a float32 HIP matrix with unit column stride")
    u, p, n = (_need(t, torch.int32, name) for t, name in ((u, "u"), (p, "p"), (n, "n")))
    b = u.numel()
    if p.numel() != b or n.numel() != b or b < 1:
        raise ValueError("bpr_loss: the three id lists must have one common, non-zero length")
    lib = _lib.load()
    loss = torch.empty((), dtype=torch.float32, device=emb.device)
    coef = torch.empty(b, dtype=torch.float32, device=emb.device)
    ws = _workspace(lib.kgat_bpr_workspace_bytes(b, emb.shape[1]), emb.device)
    with _timed("bpr_loss", (b, emb.shape[1])):
        check(lib.kgat_bpr_loss_f32(emb.shape[0], emb.shape[1], _ptr(emb), emb.stride(0), b, _ptr(u), _ptr(p), _ptr(n),
                                    float(reg_lambda), _ptr(loss), _ptr(coef), _ptr(ws), ws.numel(), _stream(emb)),
              "kgat_bpr_loss_f32")
    return loss, coef, ws


def bpr_grad(emb, u, p, n, coef, reg_lambda, grad_scale=None, workspace=None):
    """d loss / d emb, dense (N, F), scaled by the device scalar `grad_scale` (kgat_bpr_grad_f32)."""
    lib = _lib.load()
    b = u.numel()
    grad = torch.empty((emb.shape[0], emb.shape[1]), dtype=torch.float32, device=emb.device)
    if grad_scale is not None:
        grad_scale = _need(grad_scale.reshape(1), torch.float32, "grad_scale")
    ws = workspace if workspace is not None else _workspace(lib.kgat_bpr_workspace_bytes(b, emb.shape[1]), emb.device)
    with _timed("bpr_grad", (b, emb.shape[1])):
        check(lib.kgat_bpr_grad_f32(emb.shape[0], emb.shape[1], _ptr(emb), emb.stride(0), b, _ptr(u), _ptr(p), _ptr(n),
                                    _ptr(coef), float(reg_lambda), _ptr(grad_scale), _ptr(grad), _ptr(ws), ws.numel(),
                                    _stream(emb)), "kgat_bpr_grad_f32")
    return grad
