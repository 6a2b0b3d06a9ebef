"""Autograd glue for the two differentiable sparse operators of the path.

* ``u_mul_e_sum`` - update_all(u_mul_e, sum), reference models.py:63.  Backward w.r.t. the
  node features is the same SpMM kernel on the reversed graph's CSR;
  backward w.r.t. the edge weight is an SDDMM (only when the weight requires grad - in the
  reference it never does, kgat.py:142-144).
* ``edge_softmax`` - reference models.py:153; backward as DGL 0.4.x EdgeSoftmax.backward.
"""
import torch

from . import ops


class _UMulESum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, g, mul_self):
        st = g._st
        dev = x.device
        x2 = x if x.dim() == 2 else x.unsqueeze(1)
        x2 = x2.contiguous()
        if x2.dtype != torch.float32:
            raise TypeError("node features must be float32, got %s" % x2.dtype)
        csr = st.csr(dev)
        w_csr = st.csr_weights(w)  # a pending lazy attention tensor is served from its CSR copy
        # (a destination-range shard is an ordinary graph holding only its local edges: rows it
        # does not own come out as zeros and the caller exchanges them, partition.py)
        out = ops.spmm(csr.indptr, csr.col, csr.row_of, x2.detach(), w_csr, mul_self=mul_self)
        ctx.g, ctx.mul_self, ctx.squeeze = g, mul_self, x.dim() == 1
        ctx.w_shape = w.shape
        ctx.save_for_backward(x2, w, out if mul_self else None)
        return out.squeeze(1) if x.dim() == 1 else out

    @staticmethod
    def backward(ctx, grad_out):
        x2, w, _ = ctx.saved_tensors
        st = ctx.g._st
        dev = grad_out.device
        if ctx.mul_self:
            raise NotImplementedError("backward of the fused h*h_N epilogue: use the unfused op")
        go = (grad_out.unsqueeze(1) if ctx.squeeze else grad_out).contiguous()
        grad_x = grad_w = None
        if ctx.needs_input_grad[0]:
            rev = st.csr_rev(dev)
            grad_x = ops.spmm(rev.indptr, rev.col, rev.row_of, go, st.rev_weights(w))
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
        _, a_csr = ops.edge_softmax(csr.indptr, csr.row_of, csr.eid, flat, want_out=False, want_csr=True)
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
    """TransR loss of a triplet batch (reference models.py:114-133).  Forward: the loss, the per-sample rows and the
    weight-gradient partials (kgat_transr_forward_f32); backward: the ordered reductions into the three gradients,
    which read the incoming gradient from device memory (kgat_transr_backward_f32) - no multiply passes."""

    @staticmethod
    def forward(ctx, ent, W_R, rel, h, r, pos_t, neg_t, reg_lambda):
        want = any(ctx.needs_input_grad[:3])
        if not want:
            return ops.transr_loss_grad(h, r, pos_t, neg_t, ent.detach(), W_R.detach(), rel.detach(), reg_lambda,
                                        want_grad=False)[0]
        loss, ws = ops.transr_forward(h, r, pos_t, neg_t, ent.detach(), W_R.detach(), rel.detach(), reg_lambda)
        ctx.ws, ctx.shapes = ws, (tuple(ent.shape), tuple(W_R.shape))
        ctx.save_for_backward(h, r, pos_t, neg_t)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        h, r, pos_t, neg_t = ctx.saved_tensors
        g_ent, g_w, g_rel = ops.transr_backward(h, r, pos_t, neg_t, ctx.shapes, ctx.ws,
                                                grad_scale=grad_out.detach().to(torch.float32))
        need = ctx.needs_input_grad
        return (g_ent if need[0] else None, g_w if need[1] else None, g_rel if need[2] else None,
                None, None, None, None, None)


def transr_loss(ent, W_R, rel, h, r, pos_t, neg_t, reg_lambda):
    """Differentiable fused TransR loss; index tensors of any integer dtype."""
    i32 = [t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous() for t in (h, r, pos_t, neg_t)]
    return _TransRLoss.apply(ent, W_R, rel, *i32, float(reg_lambda))


class _BPRLoss(torch.autograd.Function):
    """BPR loss of the CF phase (reference models.py:170-178) as two launches forward and one sort + one scatter
    backward (kgat_bpr_loss_f32 / kgat_bpr_grad_f32) instead of ~75 torch operator launches; the incoming gradient
    is read by the scatter kernel from device memory."""

    @staticmethod
    def forward(ctx, emb, u, p, n, reg_lambda):
        e = emb.detach()
        loss, coef, ws = ops.bpr_loss(e, u, p, n, reg_lambda)
        ctx.reg_lambda, ctx.ws = reg_lambda, ws
        ctx.save_for_backward(e, u, p, n, coef)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        e, u, p, n, coef = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        g = ops.bpr_grad(e, u, p, n, coef, ctx.reg_lambda, grad_scale=grad_out.detach().to(torch.float32), workspace=ctx.ws)
        return g, None, None, None, None


def bpr_loss(embedding, src_ids, pos_dst_ids, neg_dst_ids, reg_lambda):
    """Differentiable fused BPR loss; index tensors of any integer dtype (int32 is passed through as is)."""
    i32 = [t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()
           for t in (src_ids, pos_dst_ids, neg_dst_ids)]
    return _BPRLoss.apply(embedding, *i32, float(reg_lambda))


class _GNNTrain(torch.autograd.Function):
    """The whole propagation stack under autograd (reference models.py:156-168 with the KGATConv of
    :49-70 in training mode) as one differentiable unit: per layer the SpMM, one kernel for
    (h * h_N) W2^T + LeakyReLU + dropout + the normalised copy written into its slice of the
    readout; backward per layer one kernel for the normalise / dropout / LeakyReLU gradients, two
    dense GEMMs, one two-product pass and the SpMM on the reversed CSR.  Gradients: the input
    embeddings and every W2; the edge weights are constants (kgat.py:139-145)."""

    @staticmethod
    def forward(ctx, g, slope, drop_p, seed, h0, *weights):
        st = g._st
        dev = h0.device
        h = h0.detach().contiguous()
        csr = st.csr(dev)
        ew = g.edata["w"]
        w_csr = st.csr_weights(ew)
        widths = [h.shape[1]] + [w.shape[0] for w in weights]
        out = torch.empty((h.shape[0], sum(widths)), dtype=torch.float32, device=dev)
        # the ego block (out[:, :d] = h0, models.py:159,168) is written by layer 0's dense kernel from the rows it loads
        # anyway (self_out) where the slice allows 16-byte stores; otherwise by a copy here
        ego_in_kernel = len(weights) > 0 and widths[0] % 4 == 0 and out.shape[1] % 4 == 0 and h.shape[0] > 0
        if not ego_in_kernel:
            out[:, :widths[0]] = h
        off = widths[0]
        hs, hns = [h], []
        for li, w in enumerate(weights):
            hn = ops.spmm(csr.indptr, csr.col, csr.row_of, hs[-1], w_csr)
            hs.append(ops.bi_interaction_train(hs[-1], hn, w.detach().contiguous(), slope, drop_p, seed + li,
                                               norm_out=out[:, off:off + widths[li + 1]],
                                               self_out=out[:, :widths[0]] if (li == 0 and ego_in_kernel) else None))
            hns.append(hn)
            off += widths[li + 1]
        ctx.g, ctx.slope, ctx.drop_p, ctx.seed, ctx.widths, ctx.ew = g, slope, drop_p, seed, widths, ew
        ctx.save_for_backward(*hs, *hns, *weights)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        n_l = len(ctx.widths) - 1
        saved = ctx.saved_tensors
        hs, hns, weights = saved[:n_l + 1], saved[n_l + 1:2 * n_l + 1], saved[2 * n_l + 1:]
        st = ctx.g._st
        dev = grad_out.device
        rev = st.csr_rev(dev)
        w_rev = st.rev_weights(ctx.ew)
        grad_out = grad_out.contiguous()
        offs = [0]
        for wd in ctx.widths:
            offs.append(offs[-1] + wd)
        g_a = g_b = None  # the two addends of the gradient arriving at hs[li + 1] from the layer above
        grad_w = [None] * n_l
        pending = []
        for li in range(n_l - 1, -1, -1):
            gz = ops.bi_interaction_bwd_pre(hs[li + 1], g_a, g_b, grad_out[:, offs[li + 1]:offs[li + 2]], ctx.slope,
                                            ctx.drop_p, ctx.seed + li)
            if ctx.needs_input_grad[5 + li]:
                if ops.bi_interaction_bwd_input_supported(hs[li].shape[1], gz.shape[1]):
                    # grad_z^T (h * h_N) as per-workgroup partials; every layer's set is summed by ONE launch at the end
                    pending.append((li, ops.bi_interaction_bwd_weight(gz, hs[li], hns[li], want_partials=True)))
                else:
                    grad_w[li] = tall_weight_grad(gz, hs[li] * hns[li])
            w_l = weights[li].detach().contiguous()
            if ops.bi_interaction_bwd_input_supported(w_l.shape[1], w_l.shape[0]):
                # grad_P = grad_z W2 formed per tile and multiplied on the way: grad_P * h (to be aggregated), grad_P * h_N
                t, g_b = ops.bi_interaction_bwd_input(gz, w_l, hs[li], hns[li])
            else:
                t, g_b = ops.mul2(gz @ w_l, hs[li], hns[li])
            g_a = ops.spmm(rev.indptr, rev.col, rev.row_of, t, w_rev)
        if pending:
            for (li, _), summed in zip(pending, ops.sum_partials([p_ for _, p_ in pending])):
                grad_w[li] = summed
        grad_h0 = None
        if ctx.needs_input_grad[4]:
            g0 = grad_out[:, :ctx.widths[0]]
            if g_a is not None and ctx.widths[0] % 4 == 0 and grad_out.shape[1] % 4 == 0 and g0.data_ptr() % 16 == 0:
                grad_h0 = ops.add3_rows(g0, g_a, g_b)      # one pass: (g0 + g_a) + g_b, the same additions
            else:
                grad_h0 = g0 + g_a
                grad_h0 += g_b
        return (None, None, None, None, grad_h0, *grad_w)


def tall_weight_grad(grad, x, slabs=128):
    """grad^T @ x for tall operands (N ~ 10^5 rows, <= 128 columns) reducing over N into a tiny
    result: a batched GEMM over row slabs plus a sum (the library's single GEMM for this shape takes
    0.45-0.5 ms at N = 159k; this takes ~30 us)."""
    n = x.shape[0]
    m = (n // slabs) * slabs
    if m < 16 * slabs:
        return grad.t() @ x
    gw = torch.bmm(grad[:m].view(slabs, m // slabs, -1).transpose(1, 2), x[:m].view(slabs, m // slabs, -1)).sum(0)
    if m < n:
        gw = gw + grad[m:].t() @ x[m:]
    return gw


def gnn_train(g, h0, weights, slope=0.01, drop_p=0.0, seed=0):
    """Differentiable fused propagation stack; returns the (N, sum of widths) readout."""
    return _GNNTrain.apply(g, float(slope), float(drop_p), int(seed), h0, *weights)
