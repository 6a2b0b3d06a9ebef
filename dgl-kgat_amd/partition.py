"""Destination-range partition of the CKG across the GPUs of one node.

One process per GPU.  Rank r owns the destination rows [lo_r, hi_r), chosen on the in-degree
prefix sum so that every rank gets ~E/P edges (equal node counts would be badly unbalanced
on a power-law graph).  Softmax and aggregation only reduce over the in-edges of a
destination, and the attention logit is per edge, so a rank that holds all in-edges of its
rows computes attention -> softmax -> aggregation -> bi-interaction with no exchange; the one
exchange per layer is the layer OUTPUT: each rank writes its (hi-lo) x D_out rows into a
zeroed N x D_out buffer and the buffers are summed with an RCCL all-reduce over xGMI (every
row has exactly one non-zero contributor, so the sum is exact and order independent).  Two
cheaper equivalents of that sum (direct slice exchange, per-owner broadcast) are selectable,
see `Partition`.

A shard is an ordinary DGLGraph over the full node set holding only the local edges (in
global edge-id order), so every kernel runs on it unchanged.
"""
import os

import numpy as np
import torch

from .options import options
import torch.distributed as dist

from . import ops


# What a destination row costs its owner besides its in-edges, in units of one edge: the dense part
# of every layer and the row's share of the layer-output exchange (a slice exchange is bound by the
# largest slice; the all-reduce is indifferent).  Measured per-rank costs at P = 8 on the
# amazon-book-shaped CKG: ~0.37 ns per edge of attention + aggregation, ~0.4 ns per
# row of bi-interaction + ~2.9 ns per row of slice exchange over one xGMI link -> ~8 edges.
ROW_WEIGHT = 8


def _row_weight(row_weight):
    if row_weight is None:
        row_weight = options.partition_row_weight if options.partition_row_weight is not None else ROW_WEIGHT
    return max(int(row_weight), 0)


def balanced_row_bounds(in_degrees, world, row_weight=0):
    """Row boundaries b[0..world] with ~equal cost per range, cost(v) = in_degree(v) + row_weight
    (row_weight = 0: equal edge counts)."""
    deg = np.asarray(in_degrees, dtype=np.int64) + int(row_weight)
    prefix = np.concatenate([[0], np.cumsum(deg)])
    total = int(prefix[-1])
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(prefix, (total * r) // world, side="left")))
    bounds.append(len(deg))
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def balanced_row_bounds_device(dst, n_nodes, world, row_weight=0):
    """`balanced_row_bounds` from a device-resident destination array (no host copy of the edges)."""
    deg = torch.bincount(dst.long(), minlength=n_nodes) + int(row_weight)
    prefix = torch.cumsum(deg, 0)                       # prefix[v] = cost of rows 0..v (inclusive)
    total = int(prefix[-1]) if n_nodes else 0
    cuts = torch.tensor([(total * r) // world for r in range(1, world)], dtype=prefix.dtype, device=prefix.device)
    # first v with exclusive-prefix(v) >= cut  <=>  number of inclusive prefixes < cut
    idx = torch.searchsorted(prefix, cuts, right=False)
    # exclusive prefix of row v is prefix[v - 1]: row v starts a range when prefix[v-1] >= cut
    bounds = [0] + [min(int(i) + (1 if int(c) > 0 else 0), n_nodes) for i, c in zip(idx.tolist(), cuts.tolist())] + [n_nodes]
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


EXCHANGE_MODES = ("allreduce", "p2p", "broadcast", "allgather")


class Partition:
    """Row ownership of one rank + the exchange of a layer output.

    Exchange modes (``mode`` argument, default from ``KGAT_EXCHANGE``, else "allreduce"):

    * ``allreduce`` - the owned rows inside a zeroed N x width buffer, summed over ranks (every
      row has exactly one non-zero contributor: exact and order independent).  Moves
      2(P-1)/P x N x width x 4 bytes per rank.
    * ``p2p`` - every rank sends its own row slice to each peer and receives the peers' slices
      straight into their place in the buffer (``batch_isend_irecv``: one grouped RCCL
      send/recv set, a direct xGMI link per pair).  Half the bytes of the all-reduce.
    * ``broadcast`` - one broadcast per owner of its slice (what an all-gather of unequal
      slices amounts to), issued one after the other.
    * ``allgather`` - ``dist.all_gather`` straight into the row slices of the buffer; with unequal
      slices the RCCL backend runs the per-owner broadcasts as ONE group (concurrent, a direct
      link per pair).  Backends that insist on equal slice sizes (gloo) only take it for an even
      split of the rows.
    All of them leave the same bits in the buffer.

    ``force_collectives`` (default from ``KGAT_FORCE_COLLECTIVES``): issue every collective even in a
    one-rank group, where each is an identity - the way to push all the exchange forms (and the
    gradient all-reduces of `_ShardConv.backward`) through RCCL on a box with one GPU."""

    def __init__(self, rank, world, bounds, n_nodes, group=None, mode=None, force_collectives=None):
        self.rank, self.world, self.bounds, self.n_nodes, self.group = rank, world, list(bounds), n_nodes, group
        if force_collectives is None:
            force_collectives = options.force_collectives
        self.force_collectives = bool(force_collectives)
        self.lo, self.hi = bounds[rank], bounds[rank + 1]
        self.mode = mode or options.exchange
        if self.mode not in EXCHANGE_MODES:
            raise ValueError("exchange mode %r is not one of %s" % (self.mode, EXCHANGE_MODES))
        self.exchange_enabled = True  # False: the collective is skipped (local-time probes on one GPU)
        self._bufs = {}
        self._chunks = {}
        self.n_chunks = options.exchange_chunks  # > 1: propagate_fused overlaps the exchange

    @property
    def collectives_on(self):
        """Whether `assemble` / the backward's gradient sums issue collectives at all."""
        return (self.world > 1 or self.force_collectives) and self.exchange_enabled

    def _global_rank(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def new_buffer(self, width, device, dtype=torch.float32, slot=None):
        """The N x width exchange buffer, zeroed where the exchange mode sums: only the rows this
        rank does not own (the owned rows are about to be overwritten).  `slot` (an int) makes the
        buffer a persistent one of this partition, reused by every later call with the same width
        and slot - for callers that consume the assembled rows before they ask again."""
        if slot is None:
            full = torch.empty((self.n_nodes, width), dtype=dtype, device=device)
        else:
            key = (width, str(device), dtype, slot)
            full = self._bufs.get(key)
            if full is None:
                full = self._bufs[key] = torch.empty((self.n_nodes, width), dtype=dtype, device=device)
        if self.mode == "allreduce" and (self.world > 1 or self.force_collectives):
            full[:self.lo].zero_()
            full[self.hi:].zero_()
        return full

    def assemble(self, full):
        """Complete `full` (N x width, this rank's rows already in place) with the other ranks'
        rows, in place."""
        if not self.collectives_on:
            return full
        if self.mode == "allreduce":
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
            return full
        b = self.bounds
        return self._assemble_pieces(full, b[:-1], b[1:])

    def _assemble_pieces(self, full, starts, ends):
        """One row piece [starts[r], ends[r]) per rank, each written by its rank, completed on every rank."""
        world = self.world
        if self.mode == "allreduce":
            # the zero-padded sum piece by piece: every rank contributes zeros to the pieces it does not own
            for r in range(world):
                if ends[r] > starts[r]:
                    dist.all_reduce(full[starts[r]:ends[r]], op=dist.ReduceOp.SUM, group=self.group)
        elif self.mode == "p2p":
            ops_ = []
            mine = full[starts[self.rank]:ends[self.rank]]
            for peer in range(world):
                if peer == self.rank:
                    continue
                if ends[self.rank] > starts[self.rank]:
                    ops_.append(dist.P2POp(dist.isend, mine, self._global_rank(peer), group=self.group))
                if ends[peer] > starts[peer]:
                    ops_.append(dist.P2POp(dist.irecv, full[starts[peer]:ends[peer]], self._global_rank(peer),
                                           group=self.group))
            if ops_:
                for work in dist.batch_isend_irecv(ops_):
                    work.wait()
        elif self.mode == "allgather":
            dist.all_gather([full[starts[r]:ends[r]] for r in range(world)],
                            full[starts[self.rank]:ends[self.rank]].clone(), group=self.group)
        else:
            for owner in range(world):
                if ends[owner] > starts[owner]:
                    dist.broadcast(full[starts[owner]:ends[owner]], src=self._global_rank(owner), group=self.group)
        return full

    # -- chunked exchange overlapped with the next chunk's compute (task 4) ------------------
    def plan_chunks(self, n_chunks, indptr):
        """Cut every rank's owned row range into `n_chunks` blocks of about equal cost (in_degree + ROW_WEIGHT
        per row) and make the cuts known to all ranks.  `indptr`: the device row offsets of this rank's shard
        CSR (its positions are [0, E_local)).  One host read + one all_gather_object, once per (graph, n_chunks)."""
        key = int(n_chunks)
        hit = self._chunks.get(key)
        if hit is not None:
            return hit
        lo, hi = self.lo, self.hi
        if hi > lo and key > 1:
            ip = indptr[lo:hi + 1].to(torch.int64)
            cost = (ip[1:] - ip[:-1]) + _row_weight(None)
            prefix = torch.cumsum(cost, 0)
            total = int(prefix[-1])
            targets = torch.tensor([(total * k) // key for k in range(1, key)], dtype=prefix.dtype, device=prefix.device)
            cuts = torch.searchsorted(prefix, targets, right=False).tolist()
            rows = [lo] + [min(lo + int(c) + 1, hi) for c in cuts] + [hi]
            for i in range(1, len(rows)):
                rows[i] = max(rows[i], rows[i - 1])
        else:
            rows = [lo] + [hi] * key
        edges = [int(x) for x in indptr[torch.tensor(rows, device=indptr.device)].tolist()]
        everyone = [rows]
        if self.world > 1:
            everyone = [None] * self.world
            dist.all_gather_object(everyone, rows, group=self.group)
        hit = self._chunks[key] = {"rows": everyone, "edges": edges}
        return hit

    def propagate_overlapped(self, g, h, weight, n_chunks, slot=None, side_stream=None):
        """`propagate_fused` with the layer-output exchange hidden behind compute: the owned rows are processed
        in `n_chunks` blocks (aggregation over the block's CSR positions + bi-interaction into the exchange buffer),
        and block k's rows travel - on `side_stream` - while block k + 1 is computed; the main stream waits for the
        last transfer only.  Every exchange mode moves the same pieces (the all-reduce as one small all-reduce per
        (rank, block) piece), every rank issues the same collectives in the same order, and the assembled buffer
        carries the same bits in every mode.  (Against `propagate_fused` the aggregation's tiles start at the
        block boundaries, so sums of rows that straddle a tile are rounded in a different - equally fixed - order:
        equal to ~1e-7, not bit for bit.)"""
        st = g._st
        csr = st.csr(h.device)
        w_csr = st.csr_weights(g.edata["w"])
        plan = self.plan_chunks(n_chunks, csr.indptr)
        rows_all, edges = plan["rows"], plan["edges"]
        mine = rows_all[self.rank]
        full = self.new_buffer(weight.shape[0], h.device, slot=slot)
        main = torch.cuda.current_stream(h.device)
        side = side_stream if side_stream is not None else self._side_stream(h.device)
        h_c, w_c = h.detach().contiguous(), weight.detach().contiguous()
        if self.collectives_on:
            side.wait_stream(main)   # the buffer's zero fill / earlier readers are ordered before the first transfer
        for k in range(n_chunks):
            r0, r1 = mine[k], mine[k + 1]
            if r1 > r0:
                hn = ops.spmm(csr.indptr, csr.col, csr.row_of, h_c, w_csr, rows=(r0, r1 - r0),
                              e_range=(edges[k], edges[k + 1]))
                ops.bi_interaction_mul(h_c[r0:r1], hn, w_c, 0.01, h_out=full[r0:r1])
            if not self.collectives_on:
                continue
            done = torch.cuda.Event()
            done.record(main)
            with torch.cuda.stream(side):
                side.wait_event(done)
                self._assemble_pieces(full, [rows_all[r][k] for r in range(self.world)],
                                      [rows_all[r][k + 1] for r in range(self.world)])
        if self.collectives_on:
            main.wait_stream(side)
        return full

    def _side_stream(self, device):
        key = ("side", str(device))
        s_ = self._bufs.get(key)
        if s_ is None:
            s_ = self._bufs[key] = torch.cuda.Stream(device=device)
        return s_

    def pad(self, local_rows, width):
        """This rank's rows inside an N x width exchange buffer."""
        full = self.new_buffer(width, local_rows.device, local_rows.dtype)
        full[self.lo:self.hi] = local_rows
        return full

    def exchange(self, local_rows, width):
        """The assembled N x width layer output from this rank's rows."""
        return self.assemble(self.pad(local_rows, width))

    def propagate_local(self, g, h, weight):
        """The owned rows of one KGATConv output: aggregation (+ h*h_N epilogue) over the
        shard's edges restricted to the owned row range - every local edge ends in it, so the
        CSR positions are [0, E_local) - then the dense part on those rows only."""
        st = g._st
        csr = st.csr(h.device)
        w_csr = st.csr_weights(g.edata["w"])
        prod = ops.spmm(csr.indptr, csr.col, csr.row_of, h.detach().contiguous(), w_csr, mul_self=True,
                        rows=(self.lo, self.hi - self.lo), e_range=(0, st.n_edges))
        return torch.nn.functional.leaky_relu(torch.nn.functional.linear(prod, weight))

    def propagate(self, g, h, weight):
        """One KGATConv on a shard + the all-reduce of its D_out-wide result."""
        return self.exchange(self.propagate_local(g, h, weight), weight.shape[0])

    def local_into_buffer(self, g, h, weight, slot=None):
        """The rank-local half of `propagate_fused`: the owned rows of LeakyReLU((h*h_N) W2^T) written into the
        (zero-padded where the mode sums) exchange buffer; no collective."""
        st = g._st
        csr = st.csr(h.device)
        w_csr = st.csr_weights(g.edata["w"])
        h_c = h.detach().contiguous()
        # (the aggregation's second launch left to the dense kernel where the widths allow, as in the unsharded layer:
        # kgat_layer._gnn_fused; KGAT_GNN_DEFER_FINISH=0 restores it)
        defer = (options.gnn_defer_finish and self.hi > self.lo and
                 ops.bi_interaction_deferral_supported(h_c.shape[1], weight.shape[0]))
        hn = ops.spmm(csr.indptr, csr.col, csr.row_of, h_c, w_csr, rows=(self.lo, self.hi - self.lo),
                      e_range=(0, st.n_edges), defer_finish=defer)
        rows_left = None
        if defer:
            hn, rows_left = hn
        full = self.new_buffer(weight.shape[0], h.device, slot=slot)
        if self.hi > self.lo:   # h * h_N is formed by the dense kernel while it loads its rows
            ops.bi_interaction_mul(h_c[self.lo:self.hi], hn, weight.detach().contiguous(), 0.01, h_out=full[self.lo:self.hi],
                                   deferred=rows_left)
        return full

    def propagate_fused(self, g, h, weight, slot=None):
        """Forward-only form of `propagate` with fewer launches: the owned rows of
        LeakyReLU((h*h_N) W2^T) are written by the bi-interaction kernel straight into the
        exchange buffer, then exchanged.  `slot`: see new_buffer."""
        if self.n_chunks > 1 and h.is_cuda:
            return self.propagate_overlapped(g, h, weight, self.n_chunks, slot=slot)
        return self.assemble(self.local_into_buffer(g, h, weight, slot=slot))


class GraphedForward:
    """The no-grad step - compute_attention + gnn, the sequence of the reference's eval() (kgat.py:53-59) - on a
    graph or a destination shard, with the launches of every stretch between two layer-output exchanges (the whole
    step on an unsharded graph) captured ONCE as a HIP graph (torch.cuda.CUDAGraph over the C ABI's launches on
    the capture stream) and replayed: the same kernels on the same buffers, the same bits, without the per-launch
    host work (unsharded benchmark graph: 0.469 -> 0.457 ms per step).

    Why: at P = 8 a rank's share of the benchmark graph is ~20 launches of 5-20 us each; issued one by one from
    Python (~10 us per call through ctypes) the rank is host-bound - 0.20-0.22 ms per step against 0.166 ms of
    kernels (profiles/r03_shard_local_time_8way.txt).  A replay costs the host ~15 us per stretch.

    Stretches (world > 1): [attention + softmax + layer 0 on the owned rows] | exchange 0 | [layer 1] | exchange 1 |
    ... | [readout].  The exchanges stay ordinary (eager) collectives between the replays.  All tensors live in one
    graph memory pool, so a stretch reads what the previous one wrote; the readout returned by __call__ is a
    static buffer, overwritten by the next call.  Inference only; the graph structure, the model's parameter
    tensors (updated in place by an optimiser: fine) and the exchange mode must not change after construction."""

    def __init__(self, model, g, warmup=3):
        self.model, self.g, self.part = model, g, g.partition
        dev = model.entity_embed.weight.device
        self._stretches = self._plan()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(int(warmup), 1)):   # every lazily built structure exists before the capture
                self._run_eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._pool = torch.cuda.graph_pool_handle()
        self._graphs = []
        self._state = {}
        with torch.no_grad():
            for fn, exchange in self._stretches:
                gr = torch.cuda.CUDAGraph()
                # (thread_local: a process group's watchdog thread may touch the device while this thread captures)
                with torch.cuda.graph(gr, pool=self._pool, capture_error_mode="thread_local"):
                    fn(self._state)
                self._graphs.append(gr)
                if exchange is not None:
                    exchange(self._state)   # (eager, between captures: the next stretch reads the assembled rows)
        torch.cuda.synchronize(dev)

    def _plan(self):
        from . import ops as _ops
        model, g, part = self.model, self.g, self.part
        layers = list(model.layers)

        def attention(state):
            g.edata["w"] = model.compute_attention(g)
        if part is None:
            def whole(state):
                attention(state)
                state["out"] = model.gnn(g)
            return [(whole, None)]
        widths = [model.entity_embed.weight.shape[1]] + [layer.res_fc_2.out_features for layer in layers]
        stretches = []
        for li, layer in enumerate(layers):
            def local(state, li=li, layer=layer):
                if li == 0:
                    attention(state)
                    state["blocks"] = [model._node_embeddings(g).detach()] + [None] * len(layers)
                # (per-layer keys: at replay time the Python side of every stretch is frozen to what the capture left)
                state["full", li] = part.local_into_buffer(g, state["blocks"][li], layer.res_fc_2.weight, slot=li)

            def exchange(state, li=li):
                state["blocks"][li + 1] = part.assemble(state["full", li])   # (the partition's persistent buffer of slot li)
            stretches.append((local, exchange))

        def readout(state):
            state["out"] = _ops.readout_concat(state["blocks"][:len(widths)], [False] + [True] * len(layers))
        stretches.append((readout, None))
        return stretches

    def _run_eager(self):
        state = {}
        for fn, exchange in self._stretches:
            fn(state)
            if exchange is not None:
                exchange(state)
        return state["out"]

    def __call__(self):
        for gr, (fn, exchange) in zip(self._graphs, self._stretches):
            gr.replay()
            if exchange is not None:
                exchange(self._state)
        return self._state["out"]


GraphedShardForward = GraphedForward   # (the name it was introduced under)


def shard_graph(g, rank, world, group=None, bounds=None, row_weight=None, mode=None, force_collectives=None):
    """The rank's shard of `g`: all nodes, the edges whose destination lies in the rank's row
    range (global edge-id order kept), edge features sliced accordingly.  Returns the shard
    (with ``.partition`` set) and the global ids of its edges.  A graph whose edge list lives on a
    device is sharded there (bounds from a device bincount, the kept edges selected on the device,
    ids returned as a device tensor); a host-resident one with numpy, as the reference's loader
    builds it.  `row_weight`: see ROW_WEIGHT (None: the default / KGAT_PARTITION_ROW_WEIGHT)."""
    from .graph import DGLGraph
    st = g._st
    dev = st.edge_device()
    if dev is not None:
        src, dst = st.coo(dev)
        if bounds is None:
            bounds = balanced_row_bounds_device(dst, st.n_nodes, world, _row_weight(row_weight))
        part = Partition(rank, world, bounds, st.n_nodes, group, mode=mode, force_collectives=force_collectives)
        keep = torch.nonzero((dst >= part.lo) & (dst < part.hi)).reshape(-1)
        sg = DGLGraph()
        sg.add_nodes(st.n_nodes)
        if keep.numel():
            sg.add_edges(src.index_select(0, keep), dst.index_select(0, keep))
        keep_t = keep
    else:
        if bounds is None:
            bounds = balanced_row_bounds(np.bincount(st._dst, minlength=st.n_nodes), world, _row_weight(row_weight))
        part = Partition(rank, world, bounds, st.n_nodes, group, mode=mode, force_collectives=force_collectives)
        keep = np.nonzero((st._dst >= part.lo) & (st._dst < part.hi))[0]
        sg = DGLGraph()
        sg.add_nodes(st.n_nodes)
        sg.add_edges(st._src[keep], st._dst[keep])
        keep_t = torch.as_tensor(keep)
    sg.readonly()
    for k, v in g.ndata.items():
        sg.ndata[k] = v
    for k, v in g.edata.items():
        sg.edata[k] = v.index_select(0, keep_t.to(v.device))
    sg.partition = part
    return sg, keep


class _ShardConv(torch.autograd.Function):
    """One KGATConv (reference models.py:60-70) on a destination-range shard, differentiable
    (SURVEY 8e "backward"; the CF phase of kgat.py:146-168 on N GPUs).  Parameters and the layer
    input are replicated, every rank differentiates the same loss on the same batch.

    forward:  h_N and LeakyReLU((h * h_N) W2^T) (+ hash dropout) on the owned rows from the local
              edges, then the layer-output exchange: every rank ends with all N x D_out rows.
    backward: the incoming gradient is the same on every rank; each takes the slice of its own
              rows, runs the dense backward on them, the reversed-CSR SpMM over its local edges, and
              the per-rank partial gradients of the replicated operands are summed:
              all_reduce(SUM) of grad_h (N x D_in) and of grad_W2."""

    @staticmethod
    def forward(ctx, part, g, slope, drop_p, seed, owner_only, h, weight):
        st = g._st
        dev = h.device
        lo, hi = part.lo, part.hi
        h_c = h.detach().contiguous()
        w_c = weight.detach().contiguous()
        csr = st.csr(dev)
        ew = g.edata["w"]
        hn = ops.spmm(csr.indptr, csr.col, csr.row_of, h_c, st.csr_weights(ew), rows=(lo, hi - lo),
                      e_range=(0, st.n_edges))
        full = part.new_buffer(w_c.shape[0], dev)
        z = None
        if hi > lo:
            # the dropout mask is a hash of (seed, GLOBAL row, column): the shards of a layer draw the
            # mask the unsharded layer draws
            z = ops.bi_interaction_train(h_c[lo:hi], hn, w_c, slope, drop_p, seed, row0=lo)
            full[lo:hi] = z
        part.assemble(full)
        ctx.part, ctx.g, ctx.ew = part, g, ew
        ctx.slope, ctx.drop_p, ctx.seed, ctx.owner_only = slope, drop_p, seed, bool(owner_only)
        ctx.save_for_backward(h_c, hn, z if z is not None else h_c.new_empty((0, w_c.shape[0])), w_c)
        return full

    @staticmethod
    def backward(ctx, grad_full):
        from .autograd import tall_weight_grad
        h, hn, z, w = ctx.saved_tensors
        part, st = ctx.part, ctx.g._st
        lo, hi = part.lo, part.hi
        dev = grad_full.device
        d_in = h.shape[1]
        grad_h = grad_w = None
        need_h, need_w = ctx.needs_input_grad[6], ctx.needs_input_grad[7]
        gh = torch.zeros((h.shape[0], d_in), dtype=torch.float32, device=dev) if need_h else None
        gw = torch.zeros_like(w) if need_w else None
        if hi > lo:
            g_loc = grad_full[lo:hi].contiguous()
            gz = ops.bi_interaction_bwd_pre(z, g_loc, None, None, ctx.slope, ctx.drop_p, ctx.seed, row0=lo)
            h_loc = h[lo:hi]
            if need_w:
                if ops.bi_interaction_bwd_input_supported(d_in, gz.shape[1]):
                    gw = ops.bi_interaction_bwd_weight(gz, h_loc, hn)
                else:
                    gw = tall_weight_grad(gz, h_loc * hn)
            if need_h:
                # grad_P * h (aggregated back to the sources), grad_P * h_N
                if ops.bi_interaction_bwd_input_supported(w.shape[1], w.shape[0]):
                    t, g_b = ops.bi_interaction_bwd_input(gz, w.contiguous(), h_loc, hn)
                else:
                    t, g_b = ops.mul2(gz @ w, h_loc, hn)
                rev = st.csr_rev(dev)
                # the local edges' destinations all lie in [lo, hi): only those rows of the operand are read
                t_full = torch.empty((h.shape[0], d_in), dtype=torch.float32, device=dev)
                t_full[lo:hi] = t
                gh = ops.spmm(rev.indptr, rev.col, rev.row_of, t_full, st.rev_weights(ctx.ew))
                gh[lo:hi] += g_b
        if part.collectives_on:
            if need_h:
                if ctx.owner_only:
                    # the layer input is the previous shard layer's output: its backward reads only the rows it
                    # owns (g_loc = grad[lo:hi]), so every row's sum is needed on its owner alone - one reduce per
                    # owner slice (a reduce-scatter over unequal slices): half the bytes of the all-reduce.  Rows
                    # this rank does not own keep their partial sums; nothing reads them.
                    b = part.bounds
                    works = [dist.reduce(gh[b[o]:b[o + 1]], dst=part._global_rank(o), op=dist.ReduceOp.SUM,
                                         group=part.group, async_op=True)
                             for o in range(part.world) if b[o + 1] > b[o]]
                    for w_ in works:
                        w_.wait()
                else:
                    dist.all_reduce(gh, op=dist.ReduceOp.SUM, group=part.group)
            if need_w:
                dist.all_reduce(gw, op=dist.ReduceOp.SUM, group=part.group)
        if need_h:
            grad_h = gh
        if need_w:
            grad_w = gw
        return None, None, None, None, None, None, grad_h, grad_w


def shard_conv(part, g, h, weight, slope=0.01, drop_p=0.0, seed=0, owner_only_grad=False):
    """Differentiable KGATConv on the shard `g` of partition `part`; returns the assembled N x D_out
    layer output (every rank holds all rows).  `owner_only_grad`: the gradient w.r.t. `h` is only needed on the
    rows' owners (h is the output of another shard layer, whose backward reads its own rows only): the partial
    sums are reduced to the owners instead of all-reduced.  Never for a replicated parameter (the embedding
    table feeding layer 0), whose gradient every rank needs in full."""
    return _ShardConv.apply(part, g, float(slope), float(drop_p), int(seed), bool(owner_only_grad), h, weight)
