"""Autograd glue for the two differentiable sparse operators of the path.

* ``u_mul_e_sum`` - update_all(u_mul_e, sum), reference models.py:63.  Backward w.r.t. the
  node features is the same SpMM kernel on the reversed graph's CSR;
  backward w.r.t. the edge weight is an SDDMM (only when the weight requires grad - in the
  reference it never does, kgat.py:142-144).
* ``edge_softmax`` - reference models.py:153; backward as DGL 0.4.x EdgeSoftmax.backward.
"""
import torch

from . import ops


def _flat_weight(w, n_edges):
    if w.dim() == 2 and w.shape[1] == 1:
        return w.reshape(-1)
    if w.dim() == 1:
        return w
    raise NotImplementedError("u_mul_e on the KGAT path takes an (E,1) or (E,) edge weight; got %s"
                              % (tuple(w.shape),))


class _UMulESum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, g, mul_self):
        st = g._st
        dev = x.device
        x2 = x if x.dim() == 2 else x.unsqueeze(1)
        x2 = x2.contiguous()
        if x2.dtype != torch.float32:
            raise TypeError("node features must be float32, got %s" % x2.dtype)
        w_flat = _flat_weight(w, st.n_edges).contiguous()
        csr = st.csr(dev)
        w_csr = st.weight_in_csr_order(w_flat.detach())
        # (a destination-range shard is an ordinary graph holding only its local edges: rows it
        # does not own come out as zeros and the caller exchanges them, partition.py)
        out = ops.spmm(csr.indptr, csr.col, csr.row_of, x2.detach(), w_csr, mul_self=mul_self)
        ctx.g, ctx.mul_self, ctx.squeeze = g, mul_self, x.dim() == 1
        ctx.w_shape = w.shape
        ctx.save_for_backward(x2, w_flat, out if mul_self else None)
        return out.squeeze(1) if x.dim() == 1 else out

    @staticmethod
    def backward(ctx, grad_out):
        x2, w_flat, _ = ctx.saved_tensors
        st = ctx.g._st
        dev = grad_out.device
        if ctx.mul_self:
            raise NotImplementedError("backward of the fused h*h_N epilogue: use the unfused op")
        go = (grad_out.unsqueeze(1) if ctx.squeeze else grad_out).contiguous()
        grad_x = grad_w = None
        if ctx.needs_input_grad[0]:
            rev = st.csr_rev(dev)
            grad_x = ops.spmm(rev.indptr, rev.col, rev.row_of, go, st.weight_in_rev_order(w_flat.detach()))
            if ctx.squeeze:
                grad_x = grad_x.squeeze(1)
        if ctx.needs_input_grad[1]:
            src, dst = st.coo(dev)
            grad_w = ops.sddmm_dot(src, dst, x2.detach(), go).reshape(ctx.w_shape)
        return grad_x, grad_w, None, None


def u_mul_e_sum(g, x, w, mul_self=False):
    """h_N[v] = sum_{e:u->v} w[e] * x[u]  (optionally * x[v], the bi-interaction product)."""
    if x.shape[0] != g.number_of_nodes():
        raise ValueError("node feature has %d rows, graph has %d nodes" % (x.shape[0], g.number_of_nodes()))
    if w.shape[0] != g.number_of_edges():
        raise ValueError("edge weight has %d rows, graph has %d edges" % (w.shape[0], g.number_of_edges()))
    return _UMulESum.apply(x, w, g, mul_self)


class _EdgeSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, g):
        st = g._st
        flat = logits.reshape(-1) if (logits.dim() == 1 or (logits.dim() == 2 and logits.shape[1] == 1)) else None
        if flat is None:
            raise NotImplementedError("edge_softmax on the KGAT path takes (E,1) or (E,) logits; got %s"
                                      % (tuple(logits.shape),))
        flat = flat.detach().contiguous()
        if flat.dtype != torch.float32:
            raise TypeError("logits must be float32, got %s" % flat.dtype)
        csr = st.csr(flat.device)
        _, a_csr = ops.edge_softmax(st.n_nodes, csr.row_of, csr.eid, flat, want_out=False, want_csr=True)
        a = ops.gather(st.csr_pos(flat.device), a_csr)  # back to edge-id order
        st.remember_weight(a, a_csr)
        ctx.g = g
        ctx.save_for_backward(a)
        return a.reshape(logits.shape)

    @staticmethod
    def backward(ctx, grad_a):
        (a,) = ctx.saved_tensors
        csr = ctx.g._st.csr(a.device)
        gs = ops.edge_softmax_bwd(csr.indptr, csr.eid, a, grad_a.reshape(-1).contiguous())
        return gs.reshape(grad_a.shape), None


def edge_softmax(graph, logits, eids=None):
    """dgl.nn.pytorch.softmax.edge_softmax: normalise `logits` over the incoming edges of each
    destination node.  Output has the shape of the input."""
    if eids is not None:
        raise NotImplementedError("edge_softmax on an edge subset is outside the KGAT path")
    if logits.shape[0] != graph.number_of_edges():
        raise ValueError("logits has %d rows, graph has %d edges" % (logits.shape[0], graph.number_of_edges()))
    return _EdgeSoftmax.apply(logits, graph)


class _TransRLoss(torch.autograd.Function):
    """TransR loss of a triplet batch (reference models.py:114-133) with its gradients computed in
    the same handful of launches (kgat_transr_loss_grad_f32); backward only scales them."""

    @staticmethod
    def forward(ctx, ent, W_R, rel, h, r, pos_t, neg_t, reg_lambda):
        want = any(ctx.needs_input_grad[:3])
        loss, g_ent, g_w, g_rel = ops.transr_loss_grad(h, r, pos_t, neg_t, ent.detach(), W_R.detach(), rel.detach(),
                                                       reg_lambda, want_grad=want)
        if want:
            ctx.save_for_backward(g_ent, g_w, g_rel)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        g_ent, g_w, g_rel = ctx.saved_tensors
        need = ctx.needs_input_grad
        return (g_ent * grad_out if need[0] else None, g_w * grad_out if need[1] else None,
                g_rel * grad_out if need[2] else None, None, None, None, None, None)


def transr_loss(ent, W_R, rel, h, r, pos_t, neg_t, reg_lambda):
    """Differentiable fused TransR loss; index tensors of any integer dtype."""
    i32 = [t.to(torch.int32).contiguous() for t in (h, r, pos_t, neg_t)]
    return _TransRLoss.apply(ent, W_R, rel, *i32, float(reg_lambda))
