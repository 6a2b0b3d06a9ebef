"""KGAT's attentive embedding-propagation layer on the DGLGraph surface of this package.

A from-scratch restatement of the reference's model glue for the hot path (SURVEY.md 8a rows
A1, B1, B2): ``KGATConv`` follows reference models.py:49-70, ``KGATPropagation.compute_attention``
models.py:146-154 and ``KGATPropagation.gnn`` models.py:156-168.  Parameter names and shapes
match the reference's ``Model`` (``entity_embed.weight`` (N,d), ``relation_embed.weight`` (R,k),
``W_R`` (R,d,k), ``layers.i.res_fc_2.weight`` (D_out,D_in)) so its state_dict loads unchanged.

Two ways through every step:
* the *surface* way - the very call sequence of the reference (``filter_edges`` /
  ``apply_edges`` per relation, ``edge_softmax``, ``update_all``), exercising the drop-in
  boundary; and
* the *fused* way - ``g.kgat_attention`` (one attention launch over relation-grouped edges)
  and the SpMM with the ``h * h_N`` product in its epilogue.  Same results, fewer passes.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import function as fn
from .autograd import edge_softmax, u_mul_e_sum
from .options import options


def ops_transr_supported(model, h):
    from . import ops
    n_rel, d, k = model.W_R.shape
    return ops.transr_supported(model.entity_embed.weight.shape[0], d, k, n_rel, h.numel())


class _TallLinear(torch.autograd.Function):
    """``x @ W^T`` for a tall x (N ~ 10^5 rows, <= 128 columns).  Forward and grad_x are ordinary
    library GEMMs; the weight gradient ``grad^T @ x`` reduces over N into a tiny (D_out, D_in)
    result, a shape the library handles badly (0.45-0.5 ms at N = 159k against ~25 us for the
    other two GEMMs), so it is computed as a batched GEMM over row slabs plus a sum."""

    SLABS = 128

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight)

    @staticmethod
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        grad = grad.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = grad @ weight
        if ctx.needs_input_grad[1]:
            from .autograd import tall_weight_grad
            gw = tall_weight_grad(grad, x, _TallLinear.SLABS)
        return gx, gw


class KGATConv(nn.Module):
    """Bi-interaction propagation layer: LeakyReLU_{0.01}(W2 (h * h_N)), dropout
    (reference models.py:49-70; only the ``Bi`` branch with ``res_fc_2`` exists there)."""

    def __init__(self, entity_in_feats, out_feats, dropout, res_type="Bi"):
        super().__init__()
        if res_type != "Bi":
            raise NotImplementedError(res_type)
        self.mess_drop = nn.Dropout(dropout)
        self._res_type = res_type
        self.res_fc_2 = nn.Linear(entity_in_feats, out_feats, bias=False)

    def forward(self, g, nfeat, fused=None, seed=None):
        part = g.partition
        if fused is None:
            fused = not (torch.is_grad_enabled() and nfeat.requires_grad)
        if part is not None:
            if torch.is_grad_enabled() and (nfeat.requires_grad or self.res_fc_2.weight.requires_grad):
                # differentiable shard layer (partition._ShardConv): local backward + all-reduce of the
                # gradients of the replicated operands; dropout is its hash mask, drawn on global rows
                from . import ops
                from .graph import DGLError
                from .partition import shard_conv
                # the shard layer treats the edge weights as constants (kgat.py:142-144 computes them under
                # no_grad) and runs the fused bi-interaction kernels: refuse what it cannot differentiate
                # instead of dropping a gradient or failing inside autograd
                if g.edata["w"].requires_grad:
                    raise DGLError("KGATConv on a shard does not differentiate through the edge weights "
                                   "(g.edata['w'].requires_grad is True): detach them, or use the unsharded graph")
                lin = self.res_fc_2
                if not ops.bi_interaction_supported(lin.in_features, lin.out_features):
                    raise DGLError("KGATConv on a shard under autograd supports the bi-interaction kernel's widths "
                                   "only; got %d -> %d" % (lin.in_features, lin.out_features))
                p = self.mess_drop.p if self.training else 0.0
                if seed is None:
                    seed = int(torch.empty((), dtype=torch.int64).random_()) if p > 0 else 0
                return shard_conv(part, g, nfeat, self.res_fc_2.weight, 0.01, p, seed)
            out = part.propagate(g, nfeat, self.res_fc_2.weight)
        elif fused:
            # h * h_N formed in the SpMM epilogue (models.py:63 + the th.mul of :66)
            prod = u_mul_e_sum(g, nfeat, g.edata["w"], mul_self=True)
            out = F.leaky_relu(self.res_fc_2(prod))
        else:
            g = g.local_var()
            g.ndata["h"] = nfeat
            g.update_all(fn.u_mul_e("h", "w", "m"), fn.sum("m", "h_neighbor"))
            h_neighbor = g.ndata["h_neighbor"]
            out = F.leaky_relu(_TallLinear.apply(torch.mul(g.ndata["h"], h_neighbor), self.res_fc_2.weight))
        return self.mess_drop(out)


class KGATPropagation(nn.Module):
    """The hot-path part of the reference's ``Model`` (models.py:72-111,135-168): embeddings,
    W_R, the KGATConv stack, ``compute_attention`` and ``gnn``.  The TransR / BPR losses of the
    reference (models.py:114-133,170-178) are outside this path; ``get_loss`` is kept because the
    training-step test needs a scalar to differentiate."""

    def __init__(self, n_entities, n_relations, input_node_dim=64, relation_dim=64, num_gnn_layers=3,
                 n_hidden=64, dropout=0.1, reg_lambda_gnn=0.01):
        super().__init__()
        self._n_entities, self._n_relations = n_entities, n_relations
        self._reg_lambda_gnn = reg_lambda_gnn
        self.entity_embed = nn.Embedding(n_entities, input_node_dim)
        self.relation_embed = nn.Embedding(n_relations, relation_dim)
        self.W_R = nn.Parameter(torch.empty(n_relations, input_node_dim, relation_dim))
        nn.init.xavier_uniform_(self.W_R, gain=nn.init.calculate_gain("relu"))
        self.layers = nn.ModuleList()
        for i in range(num_gnn_layers):  # widths: models.py:91-111
            d_in = input_node_dim if i == 0 else n_hidden // int(math.pow(2, i - 1))
            self.layers.append(KGATConv(d_in, n_hidden // int(math.pow(2, i)), dropout))

    # -- attention (models.py:135-154)
    def _att_score(self, edges):
        t_r = torch.matmul(self.entity_embed(edges.src["id"]), self.W_r)
        h_r = torch.matmul(self.entity_embed(edges.dst["id"]), self.W_r)
        # the batched dot product of models.py:143 (a bmm of (B,1,k) x (B,k,1) there: 1.5 ms per relation in the
        # library's batched GEMM on this shape, 61 of the surface's 72 ms - profiles/r05_surface_trace_summary.txt)
        att_w = (t_r * torch.tanh(h_r + self.relation_embed(edges.data["type"]))).sum(-1, keepdim=True)
        return {"att_w": att_w}

    def compute_attention_surface(self, g):
        """The reference's own call sequence over the drop-in surface."""
        g = g.local_var()
        for i in range(self._n_relations):
            e_idxs = g.filter_edges(lambda edges: edges.data["type"] == i)
            self.W_r = self.W_R[i]
            g.apply_edges(self._att_score, e_idxs)
        return edge_softmax(g, g.edata.pop("att_w"))

    def compute_attention(self, g, algo="auto"):
        """Fused: one attention-logit launch over relation-grouped edges + destination softmax.
        The kernels index the table by node position; `_node_embeddings` is the table itself when
        ndata['id'] is arange(N) (dataset.py:118) and entity_embed(ids) otherwise (models.py:140-141)."""
        return g.kgat_attention(self._node_embeddings(g), self.W_R, self.relation_embed.weight, algo=algo)

    # -- propagation (models.py:156-168)
    def gnn(self, g, x=None, fused=None):
        auto = fused is None
        if auto:
            fused = not torch.is_grad_enabled()
        if fused and self._can_fuse_readout():
            return self._gnn_fused(g) if g.partition is None else self._gnn_fused_sharded(g)
        if auto and self._can_fuse_training(g):
            # training mode (kgat.py:146-168): the whole stack as one autograd unit; the dropout mask
            # seed comes from torch's CPU generator, so torch.manual_seed reproduces a run
            from .autograd import gnn_train
            p = self.layers[0].mess_drop.p if self.training else 0.0
            seed = int(torch.empty((), dtype=torch.int64).random_()) if p > 0 else 0
            return gnn_train(g, self._node_embeddings(g), [layer.res_fc_2.weight for layer in self.layers],
                             0.01, p, seed)
        if auto and g.partition is not None and torch.is_grad_enabled() and self._can_fuse_training(g, sharded=True):
            return self._gnn_train_sharded(g)
        g = g.local_var()
        h = self._node_embeddings(g)
        node_embed_cache = [h]
        for layer in self.layers:
            # (a reference-shaped model routed here by compat.accelerate has the reference's own layers)
            h = layer(g, h, fused=fused) if isinstance(layer, KGATConv) else layer(g, h)
            node_embed_cache.append(F.normalize(h, p=2, dim=1))
        return torch.cat(node_embed_cache, 1)

    def _gnn_train_sharded(self, g):
        """Training mode on a destination-range shard (kgat.py:146-168 on N GPUs): every layer is a
        differentiable shard layer (partition.shard_conv), the readout is assembled from the
        exchanged layer outputs on every rank.  The dropout seed is drawn as the unsharded training
        path draws it (one draw from torch's CPU generator per call, layer i uses seed + i), and the
        hash mask is indexed by global row: with the same torch.manual_seed the shards reproduce the
        one-GPU run's masks."""
        from .partition import shard_conv
        p = self.layers[0].mess_drop.p if self.training else 0.0
        seed = int(torch.empty((), dtype=torch.int64).random_()) if p > 0 else 0
        h = self._node_embeddings(g)
        cache = [h]
        for li, layer in enumerate(self.layers):
            # (the gradient of a layer's input is needed in full only where that input is a replicated parameter -
            # the embedding table under layer 0; deeper layers reduce it to the rows' owners)
            h = shard_conv(g.partition, g, h, layer.res_fc_2.weight, 0.01, p, seed + li,
                           owner_only_grad=li > 0 and not options.shard_grad_allreduce)
            cache.append(F.normalize(h, p=2, dim=1))
        return torch.cat(cache, 1)

    def _can_fuse_training(self, g, sharded=False):
        from . import ops
        w = self.entity_embed.weight
        return ((g.partition is None) != sharded and w.is_cuda and w.dtype == torch.float32 and "w" in g.edata and
                not g.edata["w"].requires_grad and
                all(ops.bi_interaction_supported(layer.res_fc_2.in_features, layer.res_fc_2.out_features) and
                    layer.mess_drop.p < 1.0 for layer in self.layers))

    def _node_embeddings(self, g):
        """entity_embed(g.ndata['id']) (models.py:159); the reference's ids are arange(N)
        (dataset.py:118), in which case the lookup is the table itself."""
        ids = g.ndata["id"]
        # cached on the tensor object itself (held here, so its identity cannot be recycled by the
        # allocator for another graph's ids) and its version counter
        hit = getattr(self, "_ids_hit", None)
        if hit is None or hit[0] is not ids or hit[1] != ids._version:
            same = (ids.numel() == self._n_entities and
                    bool(torch.equal(ids, torch.arange(ids.numel(), device=ids.device, dtype=ids.dtype))))
            hit = self._ids_hit = (ids, ids._version, same)
        return self.entity_embed.weight if hit[2] else self.entity_embed(ids)

    def _can_fuse_readout(self):
        from . import ops
        drop_off = all((not layer.training) or layer.mess_drop.p == 0 for layer in self.layers)
        return drop_off and all(ops.bi_interaction_supported(layer.res_fc_2.in_features, layer.res_fc_2.out_features)
                                for layer in self.layers)

    def _gnn_fused(self, g):
        """No-grad fast path: aggregation with the h*h_N epilogue, then one kernel per layer for
        Linear + LeakyReLU + the L2-normalised copy written into its slice of the output."""
        from . import ops
        h = self._node_embeddings(g).detach()
        widths = [h.shape[1]] + [layer.res_fc_2.out_features for layer in self.layers]
        out = torch.empty((h.shape[0], sum(widths)), dtype=torch.float32, device=h.device)
        h0 = h
        off = widths[0]
        w = g.edata["w"]
        # the first aggregation also writes the ego block (kgat_spmm_umule_sum_f32's self_out: X[v] is in a
        # register for the h * h_N product anyway) instead of an N x d copy pass at the end: step 0.491 ->
        # 0.481 ms on the benchmark graph (the copy launch was 18 us; the aggregation grows by 3.5 us, and
        # the attention launch of the next step by 3.5 us because the pass no longer ends on the embedding
        # table).  KGAT_GNN_COPY_SELF=0 restores the separate copy.
        copy_self = options.gnn_copy_self and widths[0] % 4 == 0
        # KGAT_FUSE_BI=1: aggregation and dense part of a layer in ONE launch where the widths allow
        # (kgat_spmm_bi_fused_f32: the rows h * h_N stay with the workgroup that completed them; same bits as the
        # two launches).  Off by default: measured 3-4 % SLOWER per layer than the two launches on the benchmark
        # graph (119.7 vs 116.4 us at 64 -> 64, profiles/r04_fused_bi_ab.txt) - the dense tail
        # keeps a workgroup's gather slots idle, which costs the latency-bound aggregation more than the 82 MB
        # round trip of h * h_N costs the separate launch.
        fuse_bi = options.fuse_bi
        # KGAT_GNN_MUL_IN_SPMM=1: rounds 1-3's split - h * h_N in the aggregation's epilogue (A/B)
        mul_in_spmm = options.gnn_mul_in_spmm
        defer = options.gnn_defer_finish
        st = g._st
        scratch = None
        for li, layer in enumerate(self.layers):
            last = li + 1 == len(self.layers)
            norm_out = out[:, off:off + widths[li + 1]]
            if (fuse_bi and ops.spmm_bi_fused_supported(widths[li], widths[li + 1]) and off % 4 == 0 and
                    out.shape[1] % 4 == 0 and h.shape[0] > 0):
                csr = st.csr(h.device)
                if scratch is None or scratch.shape[1] != widths[li]:
                    scratch = torch.empty((h.shape[0], widths[li]), dtype=torch.float32, device=h.device)
                h = ops.spmm_bi_fused(csr.indptr, csr.col, csr.row_of, h.contiguous(), st.csr_weights(w),
                                      layer.res_fc_2.weight.detach(), 0.01, norm_out=norm_out, want_h=not last,
                                      scratch=scratch, self_out=out[:, :widths[0]] if (li == 0 and copy_self) else None)
                off += widths[li + 1]
                continue
            if not mul_in_spmm:
                # the plain aggregation, and h * h_N formed by the dense kernel while it loads its rows (+ the ego
                # block of the readout from the rows of layer 0's input): round 4 - the aggregation's h * h_N
                # epilogue is a dependent X[v] load per finished row inside its edge loop, 91 vs 78 us per launch
                # ... and the aggregation's second launch (the sums of the rows its edge tiles cut, the zero rows)
                # left to that kernel too: the rows are formed there from the tiles' partials, in the same order of
                # additions (KGAT_SPMM_DEFER_FINISH; same bits, one dependent launch less per layer: step 0.4415 ->
                # 0.43 ms).  KGAT_GNN_DEFER_FINISH=0 restores the two launches.
                if defer and ops.bi_interaction_deferral_supported(widths[li], widths[li + 1]) and h.shape[0] > 0:
                    csr = st.csr(h.device)
                    hc = h.contiguous()
                    hn, rows_left = ops.spmm(csr.indptr, csr.col, csr.row_of, hc, st.csr_weights(w), defer_finish=True)
                    h = ops.bi_interaction_mul(hc, hn, layer.res_fc_2.weight.detach(), 0.01, norm_out=norm_out,
                                               want_h=not last, deferred=rows_left,
                                               self_out=out[:, :widths[0]] if (li == 0 and copy_self) else None)
                    off += widths[li + 1]
                    continue
                hn = u_mul_e_sum(g, h, w)
                h = ops.bi_interaction_mul(h.contiguous(), hn, layer.res_fc_2.weight.detach(), 0.01, norm_out=norm_out,
                                           want_h=not last,
                                           self_out=out[:, :widths[0]] if (li == 0 and copy_self) else None)
                off += widths[li + 1]
                continue
            if li == 0 and copy_self:
                csr = st.csr(h.device)
                prod = ops.spmm(csr.indptr, csr.col, csr.row_of, h.contiguous(), st.csr_weights(w), mul_self=True,
                                self_out=out[:, :widths[0]])
            else:
                prod = u_mul_e_sum(g, h, w, mul_self=True)
            h = ops.bi_interaction(prod, layer.res_fc_2.weight.detach(), 0.01, norm_out=norm_out, want_h=not last)
            off += widths[li + 1]
        # the ego-embedding block last: the pass ends having just touched the embedding table, which
        # is what the next attention refresh gathers from (a step's working set is about the size
        # of the 256 MiB Infinity Cache; written first, the table was the oldest resident by then:
        # the attention launch measured 0.231 ms inside the step against 0.19 ms on its own)
        if not copy_self:
            out[:, :widths[0]] = h0
        return out

    def transR(self, h, r, pos_t, neg_t, reg_lambda_kg=0.01, fused=None):
        """TransR pairwise ranking loss of the KG phase (reference models.py:114-133 with
        bmm_maybe_select :13-47).  On the GPU in fp32 it runs as the fused
        loss+gradient kernels (kgat_transr_loss_grad_f32, ~1,800 of these steps per amazon-book
        epoch); fused=False (and CPU / float64 modules, batches beyond the kernels' limits) takes
        the torch restatement below, which is what the parity tests compare the kernels with."""
        if fused is None:
            fused = self.entity_embed.weight.is_cuda and self.entity_embed.weight.dtype == torch.float32 and \
                ops_transr_supported(self, h)
        if fused:
            from .autograd import transr_loss
            return transr_loss(self.entity_embed.weight, self.W_R, self.relation_embed.weight, h, r, pos_t, neg_t,
                               reg_lambda_kg)
        W = self.W_R.index_select(0, r)  # (B, d, k)

        def proj(ids):
            return F.normalize(torch.bmm(self.entity_embed(ids).unsqueeze(1), W).squeeze(1), p=2, dim=1)

        h_vec, pos_vec, neg_vec = proj(h), proj(pos_t), proj(neg_t)
        r_vec = F.normalize(self.relation_embed(r), p=2, dim=1)
        pos_score = (h_vec + r_vec - pos_vec).pow(2).sum(1, keepdim=True)
        neg_score = (h_vec + r_vec - neg_vec).pow(2).sum(1, keepdim=True)
        loss = (-F.logsigmoid(neg_score - pos_score)).mean()
        reg = sum((v.pow(2).sum(1) / 2.0).mean() for v in (h_vec, r_vec, pos_vec, neg_vec))
        return loss + reg_lambda_kg * reg

    def kg_step(self, h, r, pos_t, neg_t, optimizer, reg_lambda_kg=0.01):
        """One iteration of the KG phase (reference kgat.py:116-136: transR -> backward -> optimizer.step ->
        zero_grad) without the autograd bookkeeping around it: kgat_transr_loss_grad_f32 writes the loss and the three
        dense gradients into buffers kept with the model, they become the parameters' ``.grad``, ``optimizer.step()``
        runs (any torch optimiser; FusedAdam makes the whole iteration two library calls), the gradients are dropped.
        The same kernels on the same data as ``transR(...).backward()``: the same bits.  int32 index tensors are used
        as they are.  Returns the loss (a 0-dim device tensor; reading it synchronises, as loss.item() does).
        NOTE: like the reference's step - which ends in optimizer.zero_grad() - this leaves EVERY parameter's ``.grad``
        None, including gradients another phase accumulated and has not applied yet: call it between complete steps."""
        from . import ops
        ent, W_R, rel = self.entity_embed.weight, self.W_R, self.relation_embed.weight
        if not (ent.is_cuda and ent.dtype == torch.float32 and ops_transr_supported(self, h)):
            loss = self.transR(h, r, pos_t, neg_t, reg_lambda_kg)
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            return loss.detach()
        ids = [t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()
               for t in (h, r, pos_t, neg_t)]
        buf = getattr(self, "_kg_buffers", None)
        if buf is None or buf[1].shape != ent.shape or buf[1].device != ent.device:
            buf = self._kg_buffers = [torch.empty((), dtype=torch.float32, device=ent.device), torch.empty_like(ent),
                                      torch.empty_like(W_R), torch.empty_like(rel), None, -1]
        if buf[4] is None or buf[5] != ids[0].numel():
            buf[4:] = [ops.transr_workspace(ids[0].numel(), W_R.shape[1], W_R.shape[2], W_R.shape[0], ent.device),
                       ids[0].numel()]
        with torch.no_grad():
            out = ops.transr_loss_grad(*ids, ent.detach(), W_R.detach(), rel.detach(), reg_lambda_kg, out=buf[:4],
                                       workspace=buf[4])
        ent.grad, W_R.grad, rel.grad = buf[1], buf[2], buf[3]
        for p_ in self.parameters():       # (parameters the KG loss does not reach have no gradient in this phase)
            if p_ is not ent and p_ is not W_R and p_ is not rel:
                p_.grad = None
        optimizer.step()
        ent.grad = W_R.grad = rel.grad = None
        return out[0]

    def kg_phase(self, h, r, pos_t, neg_t, optimizer, reg_lambda_kg=0.01):
        """The KG phase of an epoch (reference kgat.py:116-136) over batches drawn up front: h, r, pos_t, neg_t are
        (n_iterations, batch) id tensors; iteration i is transR(h[i], ...) -> backward -> optimizer.step -> zero_grad.
        Returns the n_iterations losses as a device tensor (no host round trip per iteration; the reference reads
        loss.item() every step).

        With ``FusedAdam`` on the GPU: one launch sorts every batch (kgat_transr_presort_f32), then an iteration is ONE
        library call of three launches (kgat_transr_adam_step_f32: per-sample kernel, weight-gradient partials +
        gradient rows + loss, Adam on ent / W_R / rel without a dense gradient) - the bits of ``kg_step`` and of the
        reference's autograd + torch.optim.Adam sequence.  Any other optimiser / size: a loop over ``kg_step``.
        As in ``kg_step``, parameters the KG loss does not reach get no update and their ``.grad`` is left alone."""
        from . import ops
        from .optim import FusedAdam
        ent, W_R, rel = self.entity_embed.weight, self.W_R, self.relation_embed.weight
        if h.dim() != 2 or r.shape != h.shape or pos_t.shape != h.shape or neg_t.shape != h.shape:
            raise ValueError("kg_phase: h, r, pos_t, neg_t must share one (n_iterations, batch) shape")
        n_it, b = h.shape
        hyper = optimizer.kg_state((ent, W_R, rel)) if isinstance(optimizer, FusedAdam) else None
        fused = (hyper is not None and n_it > 0 and ent.is_cuda and ent.dtype == torch.float32 and
                 ops.transr_supported(ent.shape[0], W_R.shape[1], W_R.shape[2], W_R.shape[0], b) and
                 all(t.data_ptr() % 16 == 0 for t in (ent, W_R, rel)))
        if not fused:
            if n_it == 0:
                return torch.zeros(0, dtype=torch.float32, device=ent.device)
            # (kg_step hands back its persistent loss buffer: copy it before the next iteration overwrites it)
            return torch.stack([self.kg_step(h[i], r[i], pos_t[i], neg_t[i], optimizer, reg_lambda_kg).clone() for i in range(n_it)])
        import ctypes as C
        ids = [t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous() for t in (h, r, pos_t, neg_t)]
        st = getattr(self, "_kg_phase_state", None)
        key = (ent.shape[0], W_R.shape[1], W_R.shape[2], W_R.shape[0], b, str(ent.device))
        if st is None or st.key != key:
            st = self._kg_phase_state = ops.TransRAdamState(*key[:5], ent.device)
        losses = torch.empty(n_it, dtype=torch.float32, device=ent.device)
        with torch.no_grad():
            sorted_all, stride = ops.transr_presort(*ids, ent.shape[0], W_R.shape[0])
            states = [optimizer.state[p] for p in (ent, W_R, rel)]
            arr_m = (C.c_void_p * 3)(*[s_["exp_avg"].data_ptr() for s_ in states])
            arr_v = (C.c_void_p * 3)(*[s_["exp_avg_sq"].data_ptr() for s_ in states])
            steps = [int(s_["step"]) for s_ in states]
            e_, w_, r_ = ent.detach(), W_R.detach(), rel.detach()
            lr, b1, b2, eps = hyper
            for i in range(n_it):
                steps = [t + 1 for t in steps]
                ops.transr_adam_step(ids[0][i], ids[1][i], ids[2][i], ids[3][i], sorted_all[i * stride:(i + 1) * stride],
                                     e_, w_, r_, arr_m, arr_v, steps, lr, b1, b2, eps, reg_lambda_kg, losses[i:i + 1], st)
            for s_, t in zip(states, steps):
                s_["step"].fill_(float(t))
        return losses

    def _gnn_fused_sharded(self, g):
        """No-grad path on a destination-range shard: per layer the local aggregation, the
        bi-interaction kernel on the owned rows, one all-reduce, and the row normalisation of the
        assembled layer output into its slice of the readout."""
        from . import ops
        part = g.partition
        h = self._node_embeddings(g).detach()
        widths = [h.shape[1]] + [layer.res_fc_2.out_features for layer in self.layers]
        # every layer's assembled rows stay in an exchange buffer of the partition (one per layer),
        # and the readout - ego block copied, layer blocks normalised - is written in one pass at the
        # end (one launch instead of a normalisation per layer plus the copy)
        blocks = [h]
        for li, layer in enumerate(self.layers):
            h = part.propagate_fused(g, h, layer.res_fc_2.weight, slot=li)
            blocks.append(h)
        if len(blocks) <= 8 and all(w % 4 == 0 and w <= 128 for w in widths):
            return ops.readout_concat(blocks, [False] + [True] * len(self.layers))
        out = torch.empty((h.shape[0], sum(widths)), dtype=torch.float32, device=h.device)
        off = 0
        for li, b in enumerate(blocks):
            if li == 0:
                out[:, :widths[0]] = b
            else:
                ops.l2_normalize_rows(b, out[:, off:off + widths[li]])
            off += widths[li]
        return out

    def get_loss(self, embedding, src_ids, pos_dst_ids, neg_dst_ids, fused=None):
        """BPR loss of reference models.py:170-178.  On the GPU in fp32 (readout width a multiple of 4): the fused
        loss / gradient kernels (kgat_bpr_loss_f32, kgat_bpr_grad_f32; ~15 launches forward + backward against ~75
        torch operator launches, 0.7 ms of the CF step); fused=False (and CPU / float64) takes the torch
        restatement below, which is what the parity tests compare the kernels with."""
        if fused is None:
            fused = (embedding.is_cuda and embedding.dtype == torch.float32 and embedding.dim() == 2 and
                     embedding.shape[1] % 4 == 0 and embedding.stride(1) == 1 and embedding.stride(0) % 4 == 0 and
                     src_ids.numel() > 0)
        if fused:
            from .autograd import bpr_loss
            return bpr_loss(embedding, src_ids, pos_dst_ids, neg_dst_ids, self._reg_lambda_gnn)
        # one gather of the three id lists (one dense zero-fill + one sorted scatter in backward
        # instead of three), then split: the same rows as embedding[src_ids] etc.
        b = src_ids.shape[0]
        rows = embedding[torch.cat([src_ids, pos_dst_ids, neg_dst_ids])]
        s, p, n = rows[:b], rows[b:2 * b], rows[2 * b:]
        pos = (s * p).sum(1)
        neg = (s * n).sum(1)
        cf = -F.logsigmoid(pos - neg).mean()
        reg = sum((v.pow(2).sum(1) / 2.0).mean() for v in (s, p, n))
        return cf + self._reg_lambda_gnn * reg
