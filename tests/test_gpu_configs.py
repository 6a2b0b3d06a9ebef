"""GPU parity at the sizes of BASELINE.json's configs, through the C ABI:

* configs[0]'s shape  - last-fm-shaped CKG, 1 layer, d = k = 8 (the reference's CPU-runnable case, on the HIP path)
* configs[1]          - last-fm-shaped CKG, 3 layers, d = 64
* configs[2]          - amazon-book-shaped CKG, 3 layers, d = 64 (the graded config)
* configs[3]          - amazon-book-shaped CKG, d = k = 128 (layers 128 -> 128 -> 64 -> 32) + its 8-way
                        destination partition, shard by shard on one GPU
* configs[4]          - power-law CKG beyond the Infinity Cache (X > 256 MiB), R = 64, drawn on the device

Every case compares the device result with the C/OpenMP fp32 oracle (full tensors) and the numpy
fp64 oracle (full tensors where it finishes in seconds, sampled rows / edges on the power-law
graph) under SURVEY 8c's metric, and bounds the device's distance from fp64 by what the CPU fp32
forward itself shows (conftest.parity_8c)."""
import numpy as np
import pytest
import torch

from conftest import blocks, parity_8c, parity_8c_mean, rel_err, rel_err_inf
from oracle import c_oracle as co
from oracle import kgat_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _model(n, n_rel, d, layers, hidden, dev, seed=1234):
    import dgl_kgat_amd as K
    torch.manual_seed(seed)
    m = K.KGATPropagation(n, n_rel, input_node_dim=d, relation_dim=d, num_gnn_layers=layers, n_hidden=hidden,
                          dropout=0.0)
    params = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    return m.to(dev), params


def _oracles(n, trip, params, n_layers):
    """(attention fp32-C, readout fp32-C, attention fp64, readout fp64, un-normalised layer outputs
    fp64) of one full step."""
    src, dst, et = trip[:, 2], trip[:, 0], trip[:, 1]
    ent, W_R, rel = params["entity_embed.weight"], params["W_R"], params["relation_embed.weight"]
    W2 = [params["layers.%d.res_fc_2.weight" % i] for i in range(n_layers)]
    indptr, col, eid = co.csr_from_coo(n, src, dst)
    a_c = co.edge_softmax(n, indptr, eid, co.att_score(ent, W_R, rel, src, dst, et))
    h, cache = ent, [ent]
    for W in W2:
        h = co.bi_interaction(h, co.spmm(n, indptr, col, eid, h, a_c), W)
        cache.append(co.l2_normalize(h))
    out_c = np.concatenate(cache, 1)
    a_64 = orc.compute_attention(n, src, dst, et, ent, W_R, rel).reshape(-1)
    h64, cache64, layers64 = np.asarray(ent, np.float64), [np.asarray(ent, np.float64)], []
    for W in W2:   # orc.gnn_forward (models.py:156-168), keeping the un-normalised layer outputs too
        h64 = orc.bi_interaction(h64, orc.spmm_u_mul_e_sum_sparse(n, src, dst, h64, a_64), W)
        layers64.append(h64)
        cache64.append(orc.l2_normalize(h64))
    return a_c, out_c, a_64, np.concatenate(cache64, 1), layers64


def _check_step(tag, dev, n, trip, n_rel, d, layers, hidden, expect_form=None):
    from dgl_kgat_amd import synth
    model, params = _model(n, n_rel, d, layers, hidden, dev)
    g = synth.build_graph(n, trip, dev)
    with torch.no_grad():
        a = model.compute_attention(g)
        g.edata["w"] = a
        out = model.gnn(g)
    torch.cuda.synchronize()
    if expect_form is not None:
        assert g._st.last_att_form[0] == expect_form, g._st.last_att_form
    a_c, out_c, a_64, out_64, layers_64 = _oracles(n, trip, params, layers)
    a_h, out_h = a.cpu().numpy().reshape(-1), out.cpu().numpy()
    assert a_h.shape == a_64.shape and out_h.shape == out_64.shape
    # Bars (round-2 review: hold the device to what it achieves, not to twice the CPU fp32 run):
    #  * the 8c elementwise metric: its maximum under 1e-4 or within 1.5 x of the CPU fp32 forward's (the
    #    maximum is one element of the smallest-norm row: on configs[0] the device has measured 0.94 x and
    #    1.09 x the CPU run's depending on the softmax's summation order), its MEAN within 1.25 x;
    #  * every tensor at its own scale (max|x - y| / max|y|): 1e-5 - the "<= 1e-4 relative fp32" of the
    #    north star with a decade to spare - or, for a normalised readout block whose small-norm rows
    #    amplify any fp32 forward (configs[0]: width 8, one layer: 1.9e-5 on the device, C fp32 alike),
    #    no further from fp64 than the CPU fp32 forward; the attention additionally under the raw 8c
    #    bar of 1e-4;
    #  * every layer's UN-normalised output (what the next layer consumes) at tensor scale: 1e-5.
    e_att, _ = parity_8c(tag + " attention", a_h, a_c, a_64, factor=1.0)
    assert e_att <= 1e-4, e_att
    assert rel_err_inf(a_h, a_64) <= 1e-5
    widths = [d] + [model.layers[i].res_fc_2.out_features for i in range(layers)]
    for bi, (x, c, y) in enumerate(zip(blocks(out_h, widths), blocks(out_c, widths), blocks(out_64, widths))):
        parity_8c("%s readout block %d" % (tag, bi), x, c, y, factor=1.5)
        parity_8c_mean("%s readout block %d" % (tag, bi), x, c, y)
        e_inf, c_inf = rel_err_inf(x, y), rel_err_inf(c, y)
        # which clause carries the block (say so in the log): the fixed 1e-5 of the tensor's scale,
        # or - a normalised block whose small-norm rows amplify ANY fp32 forward - "no further from fp64 than the
        # CPU fp32 forward"; either way inside the north star's 1e-4, asserted on its own
        clause = "1e-5 of the tensor's scale" if e_inf <= 1e-5 else "no worse than the CPU fp32 forward (%.3e)" % c_inf
        print("[scale] %s readout block %d gpu %.3e  c-fp32 %.3e (max|x-y| / max|y|): passes by the clause \"%s\""
              % (tag, bi, e_inf, c_inf, clause))
        assert e_inf <= max(1e-5, c_inf), (bi, e_inf, c_inf)
        assert e_inf <= 1e-4, (bi, e_inf)     # north star: <= 1e-4 relative fp32, unconditionally
    with torch.no_grad():
        h = model.entity_embed.weight.detach()
        for li, layer in enumerate(model.layers):
            h = layer(g, h, fused=True)
            e_inf = rel_err_inf(h.cpu().numpy(), layers_64[li])
            print("[scale] %s layer %d un-normalised output gpu %.3e (bar 1e-5)" % (tag, li, e_inf))
            assert e_inf <= 1e-5, (li, e_inf)
    # zero in-degree destinations: exact zeros after every layer (and their normalised copies)
    iso = np.bincount(trip[:, 0], minlength=n) == 0
    if iso.any():
        assert np.all(out_h[iso][:, d:] == 0)
    # the attention is a partition of unity over each destination's in-edges
    sums = np.zeros(n)
    np.add.at(sums, trip[:, 0], a_h.astype(np.float64))
    assert np.allclose(sums[~iso], 1.0, atol=1e-5)
    return model, g, params, (a_h, out_h)


def test_config0_shape_lastfm_dim8_one_layer(dev):
    """BASELINE configs[0]'s shape on the HIP path (its CPU anchor is the oracle itself)."""
    from dgl_kgat_amd import synth
    n, trip, R = synth.last_fm_ckg()
    assert (n, R) == (81832, 11) and 4.8e6 < len(trip) < 4.9e6
    _check_step("configs[0] last-fm d=8 L=1", dev, n, trip, R, 8, 1, 8)


def test_config1_lastfm_dim64_three_layers(dev):
    from dgl_kgat_amd import synth
    n, trip, R = synth.last_fm_ckg()
    _check_step("configs[1] last-fm d=64", dev, n, trip, R, 64, 3, 64)


def test_config2_amazon_book_dim64_three_layers(dev):
    from dgl_kgat_amd import synth
    n, trip, R = synth.amazon_book_ckg()
    assert (n, len(trip), R) == (159251, 3663302, 41)
    _check_step("configs[2] amazon-book d=64", dev, n, trip, R, 64, 3, 64)


def test_config3_amazon_book_dim128_and_8way_partition(dev):
    """d = k = 128 (folded attention with W_r in LDS, SpMM D = 128 / 128 / 64, bi-interaction
    128->128, 128->64, 64->32) on one GPU, then the same step shard by shard over the 8-way
    destination partition: attention needs no exchange, each layer's owned rows zero-padded and
    summed (what the RCCL all-reduce computes) must reproduce the unsharded layer output."""
    from dgl_kgat_amd import ops, partition, synth
    n, trip, R = synth.amazon_book_ckg()
    model, g, params, (a_h, out_h) = _check_step("configs[3] amazon-book d=128", dev, n, trip, R, 128, 3, 128)
    assert ops.bi_interaction_supported(128, 128) and ops.bi_interaction_supported(128, 64)
    world = 8
    a_full = torch.as_tensor(a_h, device=dev)
    with torch.no_grad():
        shards = [partition.shard_graph(g, r, world) for r in range(world)]
        bounds = shards[0][0].partition.bounds
        assert bounds[0] == 0 and bounds[-1] == n and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
        assert sum(len(keep) for _, keep in shards) == len(trip)
        a_parts = torch.zeros_like(a_full)
        for sg, keep in shards:
            a_loc = model.compute_attention(sg)
            a_parts[torch.as_tensor(keep, device=dev)] = a_loc.reshape(-1)
            sg.edata["w"] = a_loc
        assert float((a_parts - a_full).abs().max()) <= 2e-6  # the shard may take another attention form
        h = model.entity_embed.weight.detach()
        g.edata["w"] = a_full.unsqueeze(1)
        for li, layer in enumerate(model.layers):
            ref = layer(g, h, fused=True)
            acc = torch.zeros_like(ref)
            for sg, _ in shards:
                part = sg.partition
                loc = part.propagate_local(sg, h, layer.res_fc_2.weight)
                assert loc.shape == (part.hi - part.lo, ref.shape[1])
                acc += part.pad(loc, loc.shape[1])
            scale = float(ref.abs().max())
            assert float((acc - ref).abs().max()) <= 2e-6 * scale, li
            h = ref


def test_config4_power_law_beyond_infinity_cache(dev):
    """configs[4]'s regime on one GPU at a size the host oracle still finishes: 2.5 M nodes /
    40 M edges / R = 64, X = 640 MB (> 256 MiB Infinity Cache), in-degree shifted Zipf capped at
    200k.  CSR build bit-exact against the C oracle; attention logits, softmax and aggregation
    against fp64 on sampled destination rows that include the heaviest hubs and empty rows; the
    full tensors against the C fp32 oracle; layer readout blocks under the 8c metric."""
    from dgl_kgat_amd import ops, synth
    import dgl_kgat_amd as K
    n, e, R, d = 2_500_000, 40_000_000, 64, 64
    src_d, dst_d, et_d = synth.power_law_coo_device(n, e, R, dev, seed=7, max_in_degree=200_000)
    src, dst, et = src_d.cpu().numpy(), dst_d.cpu().numpy(), et_d.cpu().numpy()
    # --- structure: bit-exact
    indptr, col, eid, row_of = ops.csr_from_coo(n, src_d, dst_d)
    oi, oc, oe = co.csr_from_coo(n, src, dst)
    assert np.array_equal(indptr.cpu().numpy(), oi) and np.array_equal(col.cpu().numpy(), oc)
    assert np.array_equal(eid.cpu().numpy(), oe) and np.array_equal(row_of.cpu().numpy(), dst[oe])
    deg = np.diff(oi)
    assert 100_000 < deg.max() <= 260_000 and (deg == 0).sum() > 0
    del indptr, col, eid, row_of
    # --- one full step
    model, params = _model(n, R, d, 3, d, dev)
    g = K.DGLGraph()
    g.add_nodes(n)
    g.add_edges(src, dst)
    g.readonly()
    g.ndata["id"] = torch.arange(n, device=dev)
    g.edata["type"] = et_d.long()
    with torch.no_grad():
        a = model.compute_attention(g)
        g.edata["w"] = a
        out = model.gnn(g)
    torch.cuda.synchronize()
    a_h, out_h = a.cpu().numpy().reshape(-1), out.cpu().numpy()
    ent, W_R, rel = params["entity_embed.weight"], params["W_R"], params["relation_embed.weight"]
    W2 = [params["layers.%d.res_fc_2.weight" % i] for i in range(3)]
    # --- C fp32 oracle, full tensors
    logit_c = co.att_score(ent, W_R, rel, src, dst, et)
    a_c = co.edge_softmax(n, oi, oe, logit_c)
    h, cache = ent, [ent]
    for W in W2:
        h = co.bi_interaction(h, co.spmm(n, oi, oc, oe, h, a_c), W)
        cache.append(co.l2_normalize(h))
    out_c = np.concatenate(cache, 1)
    # --- fp64 on sampled destination rows: the 6 heaviest hubs, 3 empty rows, 2,000 random rows
    rng = np.random.default_rng(3)
    rows = np.unique(np.concatenate([np.argsort(deg)[-6:], np.nonzero(deg == 0)[0][:3], rng.integers(0, n, 2000)]))
    pos = np.concatenate([np.arange(oi[v], oi[v + 1]) for v in rows])
    ed = oe[pos]                                                    # edge ids of the sampled rows, row by row
    logit64 = orc.att_score(ent, W_R, rel, src[ed], dst[ed], et[ed])
    local = np.searchsorted(rows, dst[ed])                          # dense row index 0..len(rows)
    a64 = orc.edge_softmax(len(rows), local, logit64)
    parity_8c("configs[4] attention (sampled rows)", a_h[ed], a_c[ed], a64)
    hn64 = orc.spmm_u_mul_e_sum_sparse(n, src[ed], dst[ed], ent.astype(np.float64), a64)[rows]
    with torch.no_grad():
        hn_gpu = ops.spmm(*[getattr(g._st.csr(dev), k) for k in ("indptr", "col", "row_of")],
                          model.entity_embed.weight.detach(), g._st.csr_weights(a))
    hn_c = co.spmm(n, oi, oc, oe, ent, a_c)
    parity_8c("configs[4] h_N layer 0 (sampled rows)", hn_gpu.cpu().numpy()[rows], hn_c[rows], hn64)
    assert np.all(hn_gpu.cpu().numpy()[deg == 0] == 0)
    # --- full tensors, device vs C fp32: two fp32 forwards with different summation orders, whose
    # difference the normalisation of small-norm rows amplifies layer by layer (the C oracle itself
    # sits at 1e-3..1e-2 from fp64 under the 8c metric on blocks 2-3 of the smaller configs): a
    # coarse whole-tensor agreement check, the tight statements are the sampled fp64 ones above
    assert np.max(np.abs(a_h - a_c)) <= 2e-6
    for bi, (x, c) in enumerate(zip(blocks(out_h, [64, 64, 32, 16]), blocks(out_c, [64, 64, 32, 16]))):
        assert np.max(np.abs(x - c)) <= 2e-4 * np.abs(c).max(), bi
    sums = np.zeros(n)
    np.add.at(sums, dst, a_h.astype(np.float64))
    assert np.allclose(sums[deg > 0], 1.0, atol=2e-5)


def test_config4_full_size_on_device(dev):
    """configs[4] at its FULL size on one GPU - 10 M nodes / 200 M edges / R = 64, d = 64, X = 2.56 GB -, graph
    drawn and kept on the device, every check on the device in fp64 (no host oracle finishes at this size):
    CSR invariants; attention weights of 2,000 random destination rows against an fp64 evaluation of
    models.py:135-154; every non-empty row's weights summing to one; the first layer's aggregation on the six
    heaviest hubs (~10^6 in-edges each), rows without in-edges and the random rows against an fp64 gather;
    the readout finite with unit-norm layer blocks."""
    from dgl_kgat_amd import ops, synth
    import dgl_kgat_amd as K
    n, e, R, d = 10_000_000, 200_000_000, 64, 64
    src, dst, et = synth.power_law_coo_device(n, e, R, dev, seed=11)
    torch.manual_seed(1234)
    model = K.KGATPropagation(n, R, input_node_dim=d, relation_dim=d, num_gnn_layers=3, n_hidden=d, dropout=0.0).to(dev)
    g = K.DGLGraph()
    g.add_nodes(n)
    g.add_edges(src, dst)
    g.readonly()
    g.ndata["id"] = torch.arange(n, device=dev)
    g.edata["type"] = et.long()
    csr = g._st.csr(dev)
    indptr, col, eid, row_of = csr.indptr, csr.col, csr.eid, csr.row_of
    # --- structure
    deg = (indptr[1:] - indptr[:-1]).long()
    assert int(indptr[0]) == 0 and int(indptr[-1]) == e and int(deg.min()) >= 0
    assert int(deg.max()) > 500_000 and int((deg == 0).sum()) > 0
    assert torch.equal(dst.index_select(0, eid.long()), row_of) and torch.equal(src.index_select(0, eid.long()), col)
    assert int(eid.long().sum()) == e * (e - 1) // 2 and bool((row_of[1:] >= row_of[:-1]).all())
    with torch.no_grad():
        a = torch.as_tensor(model.compute_attention(g)).reshape(-1)
        g.edata["w"] = a.reshape(-1, 1)
        out = model.gnn(g)
    torch.cuda.synchronize()
    a_csr = a.index_select(0, eid.long())
    # --- every non-empty row's weights sum to one (fp64 prefix sums over the CSR order)
    cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), torch.cumsum(a_csr.double(), 0)])
    row_sum = cs[indptr[1:].long()] - cs[indptr[:-1].long()]
    assert float((row_sum[deg > 0] - 1.0).abs().max()) < 1e-4  # (a 2e8-term fp64 prefix sum: ~1e-9 per difference)
    del cs, row_sum
    # --- attention weights of random rows against fp64
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    rows = torch.unique(torch.randint(0, n, (2000,), generator=gen, device=dev))
    rows = rows[deg[rows] <= 4096]
    starts, lens = indptr[rows].long(), deg[rows]
    seg = torch.repeat_interleave(torch.arange(rows.numel(), device=dev), lens)
    pos = torch.repeat_interleave(starts - torch.cumsum(lens, 0) + lens, lens) + torch.arange(int(lens.sum()), device=dev)
    ed = eid[pos].long()
    ent, W_R, rel = (model.entity_embed.weight.detach().double(), model.W_R.detach().double(),
                     model.relation_embed.weight.detach().double())
    et_s, t_s, h_s = et[ed].long(), ent[src[ed].long()], ent[dst[ed].long()]
    logit = torch.zeros(ed.numel(), dtype=torch.float64, device=dev)
    for r in range(R):
        m = torch.nonzero(et_s == r).reshape(-1)
        if m.numel():
            logit[m] = ((t_s[m] @ W_R[r]) * torch.tanh(h_s[m] @ W_R[r] + rel[r])).sum(1)
    mx = torch.full((rows.numel(),), -float("inf"), dtype=torch.float64, device=dev).scatter_reduce(0, seg, logit, "amax")
    ex = torch.exp(logit - mx[seg])
    a64 = ex / torch.zeros(rows.numel(), dtype=torch.float64, device=dev).index_add_(0, seg, ex)[seg]
    err = (a_csr[pos].double() - a64).abs() / torch.clamp(a64, min=1e-3 * float(a64.max()))
    assert float(err.max()) < 1e-4, float(err.max())                       # SURVEY 8c, elementwise
    assert float((a_csr[pos].double() - a64).abs().max()) < 1e-5 * float(a64.max())
    # --- first layer's aggregation: hubs, empty rows, the random rows, against an fp64 gather with the device's weights
    X = model.entity_embed.weight.detach()
    hn = ops.spmm(indptr, col, row_of, X, a_csr)
    pick = torch.unique(torch.cat([torch.topk(deg, 6).indices, torch.nonzero(deg == 0).reshape(-1)[:3], rows]))
    worst = 0.0
    for v in pick.tolist():
        b, en = int(indptr[v]), int(indptr[v + 1])
        if en == b:
            assert float(hn[v].abs().max()) == 0.0
            continue
        terms = a_csr[b:en].double()[:, None] * X[col[b:en].long()].double()
        ref, mag = terms.sum(0), terms.abs().sum(0)
        worst = max(worst, float(((hn[v].double() - ref).abs() / torch.clamp(mag, min=1e-300)).max()))
    assert worst < 1e-5, worst   # (tests/conftest.py::sum_err: forward error of a sum of products)
    # --- readout
    assert out.shape == (n, 64 + 64 + 32 + 16) and bool(torch.isfinite(out).all())
    assert torch.equal(out[:, :64], X)
    for lo, hi in ((64, 128), (128, 160), (160, 176)):
        nrm = out[:, lo:hi].norm(dim=1)
        assert float(nrm.max()) < 1.0 + 1e-5 and float((nrm[nrm > 0.5] - 1.0).abs().max()) < 1e-5


def test_attention_form_is_deterministic_across_processes(dev):
    """Two fresh processes pick the same attention form for the same graph and produce equal bits
    (the form is a function of graph statistics, not of a timing race)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import sys, hashlib, torch; sys.path.insert(0, %r)\n"
        "import dgl_kgat_amd as K\n"
        "from dgl_kgat_amd import synth\n"
        "dev = torch.device('cuda:0')\n"
        "for name, (n, trip, R) in (('amazon', synth.amazon_book_ckg(scale=0.25)), ('lastfm', synth.last_fm_ckg(scale=0.25))):\n"
        "    torch.manual_seed(5)\n"
        "    m = K.KGATPropagation(n, R, 64, 64, 1, 64, dropout=0.0).to(dev)\n"
        "    g = synth.build_graph(n, trip, dev)\n"
        "    with torch.no_grad():\n"
        "        a = m.compute_attention(g)\n"
        "    print(name, g._st.last_att_form[0], hashlib.sha256(a.cpu().numpy().tobytes()).hexdigest())\n" % ROOT)
    env = dict(os.environ)
    env.pop("KGAT_ATT_FORM", None)
    outs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
            for _ in range(2)]
    for o in outs:
        assert o.returncode == 0, o.stderr[-2000:]
    assert outs[0].stdout == outs[1].stdout and len(outs[0].stdout.splitlines()) == 2, (outs[0].stdout, outs[1].stdout)


def test_lazy_edge_weights_on_device(dev):
    """compute_attention hands back a lazy (E,1) tensor: the aggregation (forward, and the
    training stack's backward on the reversed CSR) is served from the CSR-ordered copy without
    materialising it; the first value-level read runs the permutation and gives exactly what the
    eager path gives; an in-place edit afterwards is seen by the next aggregation."""
    import dgl_kgat_amd as K
    from dgl_kgat_amd import lazy, synth
    from dgl_kgat_amd.lazy import LazyEdgeWeights
    n, trip, R = synth.amazon_book_ckg(scale=0.05)
    torch.manual_seed(2)
    m0 = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
    g0 = synth.build_graph(n, trip, dev)
    with torch.no_grad():   # the deferred form is opt-in: by default an ordinary, fully written tensor comes back
        assert not lazy.enabled() and type(m0.compute_attention(g0)) is torch.Tensor
    del m0, g0
    K.enable_lazy_edge_weights()
    try:
        _lazy_edge_weights_body(K, synth, LazyEdgeWeights, n, trip, R, dev)
    finally:
        K.enable_lazy_edge_weights(False)
    import torch.utils.dlpack as dlpack
    assert not getattr(dlpack.to_dlpack, "_kgat_guarded", False)   # the wrapper is gone again


def _lazy_edge_weights_body(K, synth, LazyEdgeWeights, n, trip, R, dev):
    torch.manual_seed(2)
    m = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
    g = synth.build_graph(n, trip, dev)
    with torch.no_grad():
        eager = g.kgat_attention(m.entity_embed.weight, m.W_R, m.relation_embed.weight, lazy=False)
        assert type(eager) is torch.Tensor
        g.edata["w"] = eager
        ref = m.gnn(g)
        a = m.compute_attention(g)
        assert isinstance(a, LazyEdgeWeights) and a.pending and a.shape == (len(trip), 1) and a.is_cuda
        g.edata["w"] = a
        out = m.gnn(g)
        assert a.pending and torch.equal(out, ref)          # served from the CSR copy, same bits
    m.train()
    loss = (m.gnn(g) ** 2).sum()                             # fused training stack: forward + reversed-CSR backward
    loss.backward()
    assert a.pending and float(m.entity_embed.weight.grad.abs().sum()) > 0
    grad_lazy = m.entity_embed.weight.grad.clone()
    m.zero_grad()
    g.edata["w"] = eager
    (m.gnn(g) ** 2).sum().backward()
    assert torch.equal(m.entity_embed.weight.grad, grad_lazy)
    m.eval()
    assert torch.equal(a.cpu(), eager.cpu()) and not a.pending   # first read: the permutation runs
    with torch.no_grad():
        g.edata["w"] = a
        assert torch.equal(m.gnn(g), ref)                    # now through the ordinary cached path
        a.mul_(0.5)                                          # in-place edit bumps the version: cache miss, new CSR copy
        assert torch.allclose(m.layers[0](g, m.entity_embed.weight, fused=True),
                              m.layers[0](g.local_var(), m.entity_embed.weight, fused=False), atol=1e-6)
        g.edata["w"] = eager * 0.5
        half = m.layers[0](g, m.entity_embed.weight, fused=True)
        g.edata["w"] = a
        assert torch.allclose(m.layers[0](g, m.entity_embed.weight, fused=True), half, atol=1e-7)


def test_accelerate_reference_shaped_model(dev):
    """compat.accelerate on a model with the reference Model's attribute layout and call signatures
    (models.py:72-111,135-168; the reference's own file cannot travel to this box): the routed
    compute_attention / gnn give what the surface call sequence gives, parameters are shared, the
    training path produces gradients."""
    import torch.nn as nn
    import torch.nn.functional as F
    import dgl_kgat_amd as K
    from dgl_kgat_amd import function as fn, synth

    class RefConv(nn.Module):  # the layout of the reference's KGATConv (models.py:49-70), surface calls only
        def __init__(self, i, o, p):
            super().__init__()
            self.mess_drop = nn.Dropout(p)
            self._res_type = "Bi"
            self.res_fc_2 = nn.Linear(i, o, bias=False)

        def forward(self, g, nfeat):
            g = g.local_var()
            g.ndata["h"] = nfeat
            g.update_all(fn.u_mul_e("h", "w", "m"), fn.sum("m", "h_neighbor"))
            return self.mess_drop(F.leaky_relu(self.res_fc_2(torch.mul(g.ndata["h"], g.ndata["h_neighbor"]))))

    class RefModel(nn.Module):
        def __init__(self, n, R, d):
            super().__init__()
            self._use_KG, self._n_entities, self._n_relations = True, n, R
            self.entity_embed, self.relation_embed = nn.Embedding(n, d), nn.Embedding(R, d)
            self.W_R = nn.Parameter(torch.empty(R, d, d))
            nn.init.xavier_uniform_(self.W_R, gain=nn.init.calculate_gain("relu"))
            self.layers = nn.ModuleList([RefConv(d, d, 0.0), RefConv(d, d // 2, 0.0)])

        def _att_score(self, edges):
            t_r = torch.matmul(self.entity_embed(edges.src["id"]), self.W_r)
            h_r = torch.matmul(self.entity_embed(edges.dst["id"]), self.W_r)
            return {"att_w": torch.bmm(t_r.unsqueeze(1), torch.tanh(h_r + self.relation_embed(edges.data["type"])).unsqueeze(2)).squeeze(-1)}

        def compute_attention(self, g):
            g = g.local_var()
            for i in range(self._n_relations):
                e_idxs = g.filter_edges(lambda edges: edges.data["type"] == i)
                self.W_r = self.W_R[i]
                g.apply_edges(self._att_score, e_idxs)
            return K.edge_softmax(g, g.edata.pop("att_w"))

        def gnn(self, g, x):
            g = g.local_var()
            h = self.entity_embed(g.ndata["id"])
            cache = [h]
            for layer in self.layers:
                h = layer(g, h)
                cache.append(F.normalize(h, p=2, dim=1))
            return torch.cat(cache, 1)

    n, trip, R = synth.amazon_book_ckg(scale=0.02)
    torch.manual_seed(4)
    m = RefModel(n, R, 32).to(dev)
    g = synth.build_graph(n, trip, dev)
    with torch.no_grad():
        a_ref = m.compute_attention(g)
        g.edata["w"] = a_ref
        out_ref = m.gnn(g, g.ndata["id"])
    w_before = m.W_R
    assert K.accelerate(m) is m and m.W_R is w_before
    with torch.no_grad():
        a = m.compute_attention(g)
        assert rel_err(a.cpu().numpy(), a_ref.cpu().numpy()) < 1e-4
        g.edata["w"] = a
        out = m.gnn(g, g.ndata["id"])
        assert out.shape == out_ref.shape and float((out - out_ref).abs().max()) < 1e-5 * float(out_ref.abs().max())
    loss = m.gnn(g, g.ndata["id"]).pow(2).sum()   # gradients enabled: the fused training stack
    loss.backward()
    assert m.entity_embed.weight.grad is not None and float(m.layers[0].res_fc_2.weight.grad.abs().sum()) > 0



def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _run_ranks(script, world, tmp_path, extra_env=None, timeout=600):
    """Start `world` copies of `script` (RANK = 0..world-1) on a free rendezvous port, their output
    going to files (a rank blocked on a full pipe while its peer waits in a collective would hang the
    test), and require exit code 0 of each."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("KGAT_EXCHANGE", None)
    env.update(extra_env or {})
    logs = [tmp_path / ("rank%d.log" % r) for r in range(world)]
    procs = []
    for r in range(world):
        with open(logs[r], "wb") as fh:
            procs.append(subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=fh,
                                          stderr=subprocess.STDOUT))
    try:
        for p_ in procs:
            p_.wait(timeout=timeout)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    for p_, log in zip(procs, logs):
        assert p_.returncode == 0, log.read_text()[-3000:]
    return [log.read_text() for log in logs]

_RANK_WORKER = r"""
import os, sys, hashlib, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import dgl_kgat_amd as K
from dgl_kgat_amd import partition, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")                      # both ranks share the one GPU of the test box
n, trip, R = synth.amazon_book_ckg(scale=0.05)
torch.manual_seed(11)
model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
g = synth.build_graph(n, trip, dev)
with torch.no_grad():
    g.edata["w"] = model.compute_attention(g)
    ref = model.gnn(g)
    for mode in partition.EXCHANGE_MODES:
        if mode in ("p2p", "allgather"):
            continue                              # gloo: device send/recv staged through the host / equal slices only (CPU test)
        sg, keep = partition.shard_graph(g, rank, world, mode=mode)
        a_loc = model.compute_attention(sg)       # all in-edges of the owned rows are local: no exchange
        assert float((a_loc.reshape(-1) - g.edata["w"].reshape(-1)[torch.as_tensor(keep, device=dev)]).abs().max()) <= 2e-6
        sg.edata["w"] = a_loc
        out = model.gnn(sg)                       # per layer: local rows -> exchange -> assembled rows on every rank
        err = float((out - ref).abs().max()) / float(ref.abs().max())
        assert err <= 2e-6, (mode, err)
        digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()
        gathered = [None] * world
        dist.all_gather_object(gathered, digest)
        assert len(set(gathered)) == 1, "ranks hold different assembled outputs"
        out2 = model.gnn(sg)                      # persistent exchange buffers: a second pass gives the same bits
        assert torch.equal(out, out2)
        # the same exchange in two row blocks per rank, block k travelling while block k + 1 is computed
        sg.partition.n_chunks = 2
        out3 = model.gnn(sg)
        assert float((out3 - ref).abs().max()) / float(ref.abs().max()) <= 2e-6, (mode, "chunked")
        gathered = [None] * world
        dist.all_gather_object(gathered, hashlib.sha256(out3.cpu().numpy().tobytes()).hexdigest())
        assert len(set(gathered)) == 1, "ranks hold different assembled outputs (chunked exchange)"
        if mode != partition.EXCHANGE_MODES[0]:
            assert gathered[0] == first_chunked, "exchange modes disagree (chunked exchange)"
        first_chunked = gathered[0]
    # the stretches between the exchanges replayed as HIP graphs, the collectives staying eager calls between the
    # replays (round 4: the first form kept ONE "current buffer" on the Python side, which a replay finds frozen at
    # the last layer's - every exchange then assembled the wrong buffer)
    sg, keep = partition.shard_graph(g, rank, world, mode="allreduce")
    sg.edata["w"] = model.compute_attention(sg)
    eager = model.gnn(sg).clone()
    gs = K.GraphedForward(model, sg)
    assert torch.equal(gs(), eager) and torch.equal(gs(), eager), "graph replay differs from the eager sharded step"
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_ranks_sharded_forward_over_gloo_on_one_gpu(dev, tmp_path):
    """The N > 1 forward path with real kernels: two processes share cuda:0, exchange layer outputs
    over gloo (all-reduce and per-owner broadcast forms) and must both end with the unsharded
    readout, bit-identical across ranks and across repeated passes."""
    from conftest import ROOT
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER % ROOT)
    _run_ranks(script, 2, tmp_path)


_GRAD_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import dgl_kgat_amd as K
from dgl_kgat_amd import partition, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")                      # the ranks share the one GPU of the test box
n, trip, R = synth.amazon_book_ckg(scale=0.05)
for drop in (0.0, 0.1):
    torch.manual_seed(11)
    model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=drop).to(dev)
    model.train()
    g = synth.build_graph(n, trip, dev)
    with torch.no_grad():
        g.edata["w"] = model.compute_attention(g)
    users = torch.arange(0, 4000, device=dev) %% n
    pos, neg = (users * 7 + 3) %% n, (users * 13 + 5) %% n

    def grads(graph):
        model.zero_grad()
        torch.manual_seed(5)                      # the dropout seed is drawn from torch's CPU generator
        loss = model.get_loss(model.gnn(graph), users, pos, neg)
        loss.backward()
        return float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    loss1, g1 = grads(g)                          # the one-GPU training stack (_GNNTrain)
    sg, keep = partition.shard_graph(g, rank, world)
    assert sg.partition.hi > sg.partition.lo
    lossP, gP = grads(sg)                         # shard layers: local backward + all-reduce
    assert abs(loss1 - lossP) <= 1e-5 * abs(loss1), (drop, loss1, lossP)
    assert set(g1) == set(gP) and "entity_embed.weight" in g1 and "layers.0.res_fc_2.weight" in g1
    for k in g1:
        scale = float(g1[k].abs().max())
        err = float((g1[k] - gP[k]).abs().max())
        assert scale > 0 and err <= 1e-5 * scale, (drop, k, err, scale)
    digest = [float(gP[k].double().abs().sum()) for k in sorted(gP)]
    both = [None] * world
    dist.all_gather_object(both, digest)
    assert all(b == both[0] for b in both), "ranks hold different gradients"
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_ranks_sharded_backward_matches_one_gpu_gradients(dev, tmp_path):
    """SURVEY 8e 'backward': the CF step (kgat.py:146-168) on destination shards.  Two processes
    share cuda:0 over gloo; loss and every parameter gradient of the sharded training path
    (partition._ShardConv: local reversed-CSR SpMM + all_reduce of grad_h / grad_W2) equal the
    one-GPU fused training stack's to 1e-5 of each tensor's scale - with dropout off and with
    dropout 0.1 (the hash mask is drawn on global rows, so the shards reproduce the one-GPU mask) -
    and both ranks end with the same gradients."""
    from conftest import ROOT
    script = tmp_path / "grad_worker.py"
    script.write_text(_GRAD_WORKER % ROOT)
    _run_ranks(script, 2, tmp_path)


_RCCL_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import dgl_kgat_amd as K
from dgl_kgat_amd import partition, synth
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl"
n, trip, R = synth.amazon_book_ckg(scale=0.05)
torch.manual_seed(1234)
model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
g = synth.build_graph(n, trip, dev)
with torch.no_grad():
    g.edata["w"] = model.compute_attention(g)
    ref = model.gnn(g)

# 1. every exchange form of Partition.assemble as RCCL calls (a one-rank group: each is an identity,
#    but the call goes through the communicator: all_reduce of the padded buffer, all_gather into a
#    list of row-slice views, per-owner broadcast, grouped send/recv with no peers)
src = torch.randn(n, 64, device=dev)
for mode in partition.EXCHANGE_MODES:
    q = partition.Partition(0, 1, [0, n], n, None, mode=mode, force_collectives=True)
    assert q.collectives_on
    full = q.new_buffer(64, dev)
    full[q.lo:q.hi] = src
    got = q.assemble(full)
    torch.cuda.synchronize()
    assert torch.equal(got, src), mode
# ... and with an owned range that is a strict sub-range (the zeroed halves of the all-reduce buffer)
q = partition.Partition(0, 1, [n // 4, n // 2], n, None, mode="allreduce", force_collectives=True)
full = q.new_buffer(32, dev)
full[q.lo:q.hi] = src[q.lo:q.hi, :32]
q.assemble(full)
assert torch.equal(full[q.lo:q.hi], src[q.lo:q.hi, :32]) and not full[:q.lo].any() and not full[q.hi:].any()

# 2. the sharded forward through RCCL, every mode
with torch.no_grad():
    for mode in partition.EXCHANGE_MODES:
        sg, keep = partition.shard_graph(g, 0, 1, mode=mode, force_collectives=True)
        sg.edata["w"] = model.compute_attention(sg)
        assert torch.equal(sg.edata["w"].reshape(-1), g.edata["w"].reshape(-1))
        out = model.gnn(sg)
        # the shard path normalises in the readout launch instead of the bi-interaction's epilogue: the
        # same bar as the two-rank gloo test against the unsharded pass, the same BITS across the modes
        assert float((out - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), (mode, float((out - ref).abs().max()))
        first = out if mode == partition.EXCHANGE_MODES[0] else first
        assert torch.equal(out, first), mode
print("forward ok: modes", partition.EXCHANGE_MODES)

# 2b. the exchange overlapped with compute (three row blocks per layer, transfers on a side stream), every mode
with torch.no_grad():
    for mode in partition.EXCHANGE_MODES:
        sg, keep = partition.shard_graph(g, 0, 1, mode=mode, force_collectives=True)
        sg.partition.n_chunks = 3
        sg.edata["w"] = model.compute_attention(sg)
        out = model.gnn(sg)
        assert float((out - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), (mode, "chunked")
        first_c = out if mode == partition.EXCHANGE_MODES[0] else first_c
        assert torch.equal(out, first_c), mode
        assert torch.equal(out, model.gnn(sg))      # persistent buffers, side stream: a second pass gives the same bits
print("overlapped forward ok")

# 3. shard_conv backward (all_reduce of grad_h and grad_W2 through RCCL) against the one-GPU training stack
model.train()
users = torch.arange(0, 4000, device=dev) %% n
pos, neg = (users * 7 + 3) %% n, (users * 13 + 5) %% n
def grads(graph):
    model.zero_grad()
    torch.manual_seed(5)
    loss = model.get_loss(model.gnn(graph), users, pos, neg)
    loss.backward()
    return float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
loss1, g1 = grads(g)
sg, keep = partition.shard_graph(g, 0, 1, mode="allreduce", force_collectives=True)
with torch.no_grad():
    sg.edata["w"] = model.compute_attention(sg)
lossP, gP = grads(sg)
assert abs(loss1 - lossP) <= 1e-6 * abs(loss1), (loss1, lossP)
assert set(g1) == set(gP) and "entity_embed.weight" in g1
for k in g1:
    scale = float(g1[k].abs().max())
    err = float((g1[k] - gP[k]).abs().max())
    assert scale > 0 and err <= 1e-5 * scale, (k, err, scale)
print("backward ok")
# which RCCL is mapped into this process
with open("/proc/self/maps") as fh:
    libs = sorted({ln.split()[-1] for ln in fh if "rccl" in ln or "nccl" in ln})
print("rccl libraries mapped:", libs)
assert libs, "no RCCL library mapped into the process"
dist.barrier()
dist.destroy_process_group()
print("rank 0 ok")
"""


def test_one_rank_rccl_group_runs_every_collective_form(dev, tmp_path):
    """SURVEY 8e on the one GPU a build box has: a world-size-1 `nccl` (= RCCL) process group with
    `force_collectives`, so that every collective of the multi-GPU path is an actual RCCL call -
    all four `Partition.assemble` forms (all_reduce of the zero-padded buffer, all_gather into
    unequal-slice views, per-owner broadcast, batch_isend_irecv with no peers), the sharded forward of
    every mode (the same bits in every mode, within 2e-6 of the unsharded readout) and `_ShardConv.backward`'s gradient all-reduces
    (gradients equal to the one-GPU training stack's).  Runs in a child process: a stalled collective
    fails the test at the timeout instead of hanging the suite."""
    from conftest import ROOT
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER % ROOT)
    out = _run_ranks(script, 1, tmp_path, timeout=600)[0]
    print(out[-1500:])
    assert "forward ok" in out and "backward ok" in out and "rccl" in out.lower()


def test_smoke_gates_hold_on_three_parameter_seeds(dev, capsys):
    """VERDICT round 3, task 7: the robust gates of __graft_entry__.smoke() (99.9th percentile of the 8c metric within
    1.5 x of the CPU fp32 run's, mean within 1.25 x, tensor scale 1e-5, maximum as a 10 x tripwire) on three
    parameter seeds in one run."""
    import __graft_entry__ as ge
    ge.smoke(seeds=(7, 8, 9))
    out = capsys.readouterr().out
    print(out)
    assert out.count("smoke ok") == 3


def test_graphed_forward_replays_the_same_bits(dev):
    """partition.GraphedForward: the step's launches captured once as HIP graphs and replayed - unsharded graph (one
    graph) and a destination shard (one graph per stretch between two exchanges; world = 1, all exchange modes) -
    must give the eager step's readout bit for bit, also after the parameters changed in place (an optimiser step)
    and after a second replay."""
    import dgl_kgat_amd as K
    from dgl_kgat_amd import partition, synth
    n, trip, R = synth.amazon_book_ckg(scale=0.05)
    torch.manual_seed(21)
    model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
    g = synth.build_graph(n, trip, dev)

    def eager(graph):
        with torch.no_grad():
            graph.edata["w"] = model.compute_attention(graph)
            return model.gnn(graph).clone()
    for graph in (g, partition.shard_graph(g, 0, 1, mode="allreduce")[0], partition.shard_graph(g, 1, 3, mode="broadcast")[0]):
        if graph.partition is not None:
            graph.partition.exchange_enabled = graph.partition.world == 1
        ref = eager(graph)
        gs = K.GraphedForward(model, graph)
        assert torch.equal(gs(), ref) and torch.equal(gs(), ref)
        with torch.no_grad():
            model.entity_embed.weight.mul_(1.25)             # in-place parameter update between replays
            model.layers[0].res_fc_2.weight.add_(0.01)
        ref2 = eager(graph)
        assert not torch.equal(ref2, ref) and torch.equal(gs(), ref2)


_MULTI_GPU_WORKER = r"""
import os, sys, hashlib, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import dgl_kgat_amd as K
from dgl_kgat_amd import partition, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ["KGAT_TEST_BACKEND"]
one_gpu = os.environ.get("KGAT_TEST_ONE_GPU") == "1"          # rehearsal: every rank on cuda:0 over gloo
dev = torch.device("cuda", 0 if one_gpu else rank)
torch.cuda.set_device(dev)
dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
assert dist.get_world_size() == world and dist.get_backend() == backend
modes = [m for m in partition.EXCHANGE_MODES if backend == "nccl" or m not in ("p2p", "allgather")]
n, trip, R = synth.amazon_book_ckg(scale=0.05)
torch.manual_seed(11)
model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
for p_ in model.parameters():                                  # one source of truth, whatever each rank's RNG drew
    dist.broadcast(p_.data, src=0)
g = synth.build_graph(n, trip, dev)

def same_on_all_ranks(t, what):
    got = [None] * world
    dist.all_gather_object(got, hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest())
    assert len(set(got)) == 1, what + ": ranks disagree"
    return got[0]

with torch.no_grad():
    g.edata["w"] = model.compute_attention(g)
    ref = model.gnn(g)                                         # the one-GPU result, computed on every rank's own device
    same_on_all_ranks(ref, "unsharded readout")
    digests = {}
    for mode in modes:
        for chunks in (1, 3):                                  # 3: the exchange in row blocks, overlapped with compute
            sg, keep = partition.shard_graph(g, rank, world, mode=mode)
            assert sg.partition.collectives_on and sg.partition.hi > sg.partition.lo
            sg.partition.n_chunks = chunks
            a_loc = model.compute_attention(sg)
            assert float((a_loc.reshape(-1) - g.edata["w"].reshape(-1)[torch.as_tensor(keep, device=dev)]).abs().max()) <= 2e-6
            sg.edata["w"] = a_loc
            out = model.gnn(sg)
            err = float((out - ref).abs().max()) / float(ref.abs().max())
            assert err <= 2e-6, (mode, chunks, err)
            assert torch.equal(out, model.gnn(sg)), (mode, chunks, "second pass differs")
            digests[(mode, chunks)] = same_on_all_ranks(out, "sharded readout %%s x%%d" %% (mode, chunks))
    for chunks in (1, 3):                                      # every exchange form moves the same bits
        assert len({digests[(m, chunks)] for m in modes}) == 1, digests
print("rank", rank, "forward ok:", modes, flush=True)

# backward: the CF step on shards (local reversed-CSR aggregation + gradient reductions) against the one-GPU stack
for drop in (0.0, 0.1):
    for li, layer in enumerate(model.layers):
        layer.mess_drop.p = drop
    model.train()
    users = torch.arange(0, 4000, device=dev) %% n
    pos, neg = (users * 7 + 3) %% n, (users * 13 + 5) %% n
    def grads(graph):
        model.zero_grad()
        torch.manual_seed(5)
        loss = model.get_loss(model.gnn(graph), users, pos, neg)
        loss.backward()
        return float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    loss1, g1 = grads(g)
    sg, keep = partition.shard_graph(g, rank, world)
    with torch.no_grad():
        sg.edata["w"] = model.compute_attention(sg)
    lossP, gP = grads(sg)
    assert abs(loss1 - lossP) <= 1e-5 * abs(loss1), (drop, loss1, lossP)
    assert set(g1) == set(gP) and "entity_embed.weight" in g1 and "layers.0.res_fc_2.weight" in g1
    for k in g1:
        scale = float(g1[k].abs().max())
        err = float((g1[k] - gP[k]).abs().max())
        assert scale > 0 and err <= 1e-5 * scale, (drop, k, err, scale)
        same_on_all_ranks(gP[k], "gradient " + k)
print("rank", rank, "backward ok", flush=True)
if backend == "nccl":
    with open("/proc/self/maps") as fh:
        libs = sorted({ln.split()[-1] for ln in fh if "rccl" in ln or "nccl" in ln})
    assert libs, "no RCCL library mapped into the process"
    print("rank", rank, "device", torch.cuda.get_device_name(dev), "rccl:", libs, flush=True)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_multi_gpu_worker_rehearsal_two_ranks_gloo_one_gpu(dev, tmp_path):
    """The worker of the multi-GPU RCCL tests below, rehearsed where it can run today: two ranks on cuda:0 over
    gloo (all-reduce and broadcast forms, plain and in row blocks; backward with and without dropout)."""
    from conftest import ROOT
    script = tmp_path / "multi_gpu_worker.py"
    script.write_text(_MULTI_GPU_WORKER % ROOT)
    outs = _run_ranks(script, 2, tmp_path, extra_env={"KGAT_TEST_BACKEND": "gloo", "KGAT_TEST_ONE_GPU": "1"})
    assert all("forward ok" in o and "backward ok" in o for o in outs)


def _n_gpus():
    import torch
    return torch.cuda.device_count()     # (counting devices does not initialise the GPU runtime)


@pytest.mark.skipif(_n_gpus() < 2, reason="first contact with >= 2 RCCL ranks needs a box with >= 2 GPUs "
                                          "(VERDICT round 5, task 7: prepared, runs the day such a box appears)")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_sharded_forward_and_backward_match_one_gpu(world, tmp_path):
    """SURVEY 8e on real links: `world` processes, one per GPU, `nccl` (= RCCL over xGMI) process group - the sharded
    forward in EVERY exchange form (all-reduce of the zero-padded buffer = north star; all-gather of unequal slices;
    per-owner broadcast; grouped send / recv), plain and overlapped in three row blocks, must equal the one-GPU
    readout to 2e-6 and carry the same bits on every rank and in every form; the sharded CF backward must give the
    one-GPU gradients (1e-5 of each tensor's scale) on every rank, with and without dropout."""
    if _n_gpus() < world:
        pytest.skip("%d GPUs visible" % _n_gpus())
    from conftest import ROOT
    script = tmp_path / "multi_gpu_worker.py"
    script.write_text(_MULTI_GPU_WORKER % ROOT)
    outs = _run_ranks(script, world, tmp_path, extra_env={"KGAT_TEST_BACKEND": "nccl"}, timeout=900)
    for o in outs:
        assert "forward ok" in o and "backward ok" in o and "rccl" in o.lower(), o[-2000:]
    print(outs[0][-800:])
