"""Destination-range partition of the CKG across the GPUs of one node.

One process per GPU.  Rank r owns the destination rows [lo_r, hi_r), chosen on the in-degree
prefix sum so that every rank gets ~E/P edges (equal node counts would be badly unbalanced
on a power-law graph).  Softmax and aggregation only reduce over the in-edges of a
destination, and the attention logit is per edge, so a rank that holds all in-edges of its
rows computes attention -> softmax -> aggregation -> bi-interaction with no exchange; the one
exchange per layer is the layer OUTPUT: each rank writes its (hi-lo) x D_out rows into a
zeroed N x D_out buffer and the buffers are summed with an RCCL all-reduce over xGMI (every
row has exactly one non-zero contributor, so the sum is exact and order independent).

A shard is an ordinary DGLGraph over the full node set holding only the local edges (in
global edge-id order), so every kernel runs on it unchanged.
"""
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


class Partition:
    def __init__(self, rank, world, bounds, n_nodes, group=None):
        self.rank, self.world, self.bounds, self.n_nodes, self.group = rank, world, list(bounds), n_nodes, group
        self.lo, self.hi = bounds[rank], bounds[rank + 1]

    def pad(self, local_rows, width):
        """This rank's rows inside a zeroed N x width buffer."""
        full = torch.zeros((self.n_nodes, width), dtype=local_rows.dtype, device=local_rows.device)
        full[self.lo:self.hi] = local_rows
        return full

    def exchange(self, local_rows, width):
        """Zero-padded N x width buffer holding this rank's rows, all-reduced (sum): every row
        has exactly one non-zero contributor, so the result is exact and order independent."""
        full = self.pad(local_rows, width)
        if self.world > 1:
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
        return full

    def propagate_local(self, g, h, weight):
        """The owned rows of one KGATConv output: aggregation (+ h*h_N epilogue) over the
        shard's edges restricted to the owned row range - every local edge ends in it, so the
        CSR positions are [0, E_local) - then the dense part on those rows only."""
        st = g._st
        csr = st.csr(h.device)
        w = g.edata["w"]
        w_csr = st.weight_in_csr_order(w.detach().reshape(-1).contiguous())
        prod = ops.spmm(csr.indptr, csr.col, csr.row_of, h.detach().contiguous(), w_csr, mul_self=True,
                        rows=(self.lo, self.hi - self.lo), e_range=(0, st.n_edges))
        return torch.nn.functional.leaky_relu(torch.nn.functional.linear(prod, weight))

    def propagate(self, g, h, weight):
        """One KGATConv on a shard + the all-reduce of its D_out-wide result."""
        return self.exchange(self.propagate_local(g, h, weight), weight.shape[0])

    def propagate_fused(self, g, h, weight):
        """Forward-only form of `propagate` with fewer launches: the owned rows of
        LeakyReLU((h*h_N) W2^T) are written by the bi-interaction kernel straight into the
        zeroed exchange buffer, then all-reduced."""
        st = g._st
        csr = st.csr(h.device)
        w_csr = st.weight_in_csr_order(g.edata["w"].detach().reshape(-1).contiguous())
        prod = ops.spmm(csr.indptr, csr.col, csr.row_of, h.detach().contiguous(), w_csr, mul_self=True,
                        rows=(self.lo, self.hi - self.lo), e_range=(0, st.n_edges))
        full = torch.zeros((self.n_nodes, weight.shape[0]), dtype=torch.float32, device=h.device)
        if self.hi > self.lo:
            ops.bi_interaction(prod, weight.detach().contiguous(), 0.01, h_out=full[self.lo:self.hi])
        if self.world > 1:
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
        return full


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
