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
import torch.distributed as dist

from . import ops


def balanced_row_bounds(in_degrees, world):
    """Row boundaries b[0..world] with ~equal edge counts per range."""
    deg = np.asarray(in_degrees, dtype=np.int64)
    prefix = np.concatenate([[0], np.cumsum(deg)])
    total = int(prefix[-1])
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(prefix, (total * r) // world, side="left")))
    bounds.append(len(deg))
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
    All of them leave the same bits in the buffer."""

    def __init__(self, rank, world, bounds, n_nodes, group=None, mode=None):
        self.rank, self.world, self.bounds, self.n_nodes, self.group = rank, world, list(bounds), n_nodes, group
        self.lo, self.hi = bounds[rank], bounds[rank + 1]
        self.mode = mode or os.environ.get("KGAT_EXCHANGE", "allreduce")
        if self.mode not in EXCHANGE_MODES:
            raise ValueError("exchange mode %r is not one of %s" % (self.mode, EXCHANGE_MODES))
        self.exchange_enabled = True  # False: the collective is skipped (local-time probes on one GPU)
        self._bufs = {}

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
        if self.mode == "allreduce" and self.world > 1:
            full[:self.lo].zero_()
            full[self.hi:].zero_()
        return full

    def assemble(self, full):
        """Complete `full` (N x width, this rank's rows already in place) with the other ranks'
        rows, in place."""
        if self.world == 1 or not self.exchange_enabled:
            return full
        b = self.bounds
        if self.mode == "allreduce":
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
        elif self.mode == "p2p":
            ops_ = []
            mine = full[self.lo:self.hi]
            for peer in range(self.world):
                if peer == self.rank:
                    continue
                if self.hi > self.lo:
                    ops_.append(dist.P2POp(dist.isend, mine, self._global_rank(peer), group=self.group))
                if b[peer + 1] > b[peer]:
                    ops_.append(dist.P2POp(dist.irecv, full[b[peer]:b[peer + 1]], self._global_rank(peer),
                                           group=self.group))
            if ops_:
                for work in dist.batch_isend_irecv(ops_):
                    work.wait()
        elif self.mode == "allgather":
            dist.all_gather([full[b[r]:b[r + 1]] for r in range(self.world)], full[self.lo:self.hi].clone(),
                            group=self.group)
        else:
            for owner in range(self.world):
                if b[owner + 1] > b[owner]:
                    dist.broadcast(full[b[owner]:b[owner + 1]], src=self._global_rank(owner), group=self.group)
        return full

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

    def propagate_fused(self, g, h, weight, slot=None):
        """Forward-only form of `propagate` with fewer launches: the owned rows of
        LeakyReLU((h*h_N) W2^T) are written by the bi-interaction kernel straight into the
        exchange buffer, then exchanged.  `slot`: see new_buffer."""
        st = g._st
        csr = st.csr(h.device)
        w_csr = st.csr_weights(g.edata["w"])
        prod = ops.spmm(csr.indptr, csr.col, csr.row_of, h.detach().contiguous(), w_csr, mul_self=True,
                        rows=(self.lo, self.hi - self.lo), e_range=(0, st.n_edges))
        full = self.new_buffer(weight.shape[0], h.device, slot=slot)
        if self.hi > self.lo:
            ops.bi_interaction(prod, weight.detach().contiguous(), 0.01, h_out=full[self.lo:self.hi])
        return self.assemble(full)


def shard_graph(g, rank, world, group=None, bounds=None):
    """The rank's shard of `g`: all nodes, the edges whose destination lies in the rank's row
    range (global edge-id order kept), edge features sliced accordingly.  Returns the shard
    (with ``.partition`` set) and the global ids of its edges."""
    from .graph import DGLGraph
    st = g._st
    if bounds is None:
        bounds = balanced_row_bounds(np.bincount(st._dst, minlength=st.n_nodes), world)
    part = Partition(rank, world, bounds, st.n_nodes, group)
    keep = np.nonzero((st._dst >= part.lo) & (st._dst < part.hi))[0]
    sg = DGLGraph()
    sg.add_nodes(st.n_nodes)
    sg.add_edges(st._src[keep], st._dst[keep])
    sg.readonly()
    for k, v in g.ndata.items():
        sg.ndata[k] = v
    keep_t = torch.as_tensor(keep)
    for k, v in g.edata.items():
        sg.edata[k] = v.index_select(0, keep_t.to(v.device))
    sg.partition = part
    return sg, keep
