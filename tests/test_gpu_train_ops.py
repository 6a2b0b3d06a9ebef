"""Training-step kernels of round 5 against their torch restatements: the fused BPR loss / gradient
(kgat_bpr_loss_f32, kgat_bpr_grad_f32; reference models.py:170-178) and the one-launch Adam
(kgat_adam_step_f32; reference kgat.py:85 optim.Adam)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _model(n, dev, F_dims=(64, 3, 64)):
    import dgl_kgat_amd as K
    return K.KGATPropagation(n, 4, F_dims[0], F_dims[0], F_dims[1], F_dims[2], dropout=0.0).to(dev)


# (batch sizes on both sides of the gradient's sort forms: 3 B ids in one 4,096-id slice, exactly one slice, one id
#  over, sixteen slices = the last size of the slice sort + rank merge, and the device radix sort beyond it)
@pytest.mark.parametrize("n,F,B", [(500, 176, 1000), (97, 8, 13), (3000, 48, 10240), (64, 260, 300), (700, 16, 1365),
                                   (700, 16, 1366), (5000, 8, 21845), (5000, 8, 21846)])
def test_bpr_loss_and_gradient_vs_torch(dev, n, F, B):
    """Loss and d loss / d readout of the fused kernels against the torch restatement of get_loss evaluated in
    float64 (rows repeated many times inside a batch, every role; the gradient scaled by what arrives at the loss),
    and bit for bit the same on a second call."""
    torch.manual_seed(n + F)
    m = _model(n, dev)
    emb = torch.randn(n, F, device=dev)
    u = torch.randint(0, max(n // 3, 1), (B,), device=dev)
    p = torch.randint(0, n, (B,), device=dev)
    q = torch.randint(0, n, (B,), device=dev)
    p[:5] = u[:5]                                   # a row that is source and positive of the same sample
    outs = []
    for fused, dt in ((True, torch.float32), (True, torch.float32), (False, torch.float64)):
        e = emb.to(dt).clone().requires_grad_(True)
        loss = m.get_loss(e, u, p, q, fused=fused)
        (loss * 2.5).backward()
        outs.append((loss.detach().double().cpu().numpy(), e.grad.double().cpu().numpy()))
    (l0, g0), (l1, g1), (lr, gr) = outs
    assert np.array_equal(l0, l1) and np.array_equal(g0, g1)          # fixed order of additions
    assert abs(l0 - lr) <= 2e-6 * abs(lr)
    scale = np.abs(gr).max()
    assert np.abs(g0 - gr).max() <= 2e-6 * scale, np.abs(g0 - gr).max() / scale
    assert np.array_equal(g0 == 0, gr == 0) or np.abs(g0[gr == 0]).max() == 0   # rows outside the batch: exact zeros
    # int32 ids are taken as they are
    e = emb.clone().requires_grad_(True)
    l2 = m.get_loss(e, u.int(), p.int(), q.int())
    assert float(l2.detach()) == float(l0)


@pytest.mark.parametrize("B,hot", [(4000, 1500), (300, 299), (10240, 40)])
def test_bpr_gradient_hot_rows(dev, B, hot):
    """Batches as an edge-uniform sampler draws them: one item is the positive of `hot` samples (its contributions span
    many 32-position windows of the sorted ids: window pieces + carry launch), one user the source of 70, one row
    appears in all three roles; against the float64 torch restatement, and bit for bit the same on a second call."""
    torch.manual_seed(B)
    n, F = 2000, 176
    m = _model(n, dev)
    emb = torch.randn(n, F, device=dev)
    u = torch.randint(0, n, (B,), device=dev)
    p = torch.randint(0, n, (B,), device=dev)
    q = torch.randint(0, n, (B,), device=dev)
    p[torch.randperm(B, device=dev)[:hot]] = 1234
    u[torch.randperm(B, device=dev)[:min(70, B)]] = 77
    q[:3] = 1234
    u[3:6] = 1234
    outs = []
    for fused, dt in ((True, torch.float32), (True, torch.float32), (False, torch.float64)):
        e = emb.to(dt).clone().requires_grad_(True)
        loss = m.get_loss(e, u, p, q, fused=fused)
        loss.backward()
        outs.append(e.grad.double().cpu().numpy())
    g0, g1, gr = outs
    assert np.array_equal(g0, g1)
    scale = np.abs(gr).max()
    assert np.abs(g0 - gr).max() <= 3e-6 * scale, np.abs(g0 - gr).max() / scale
    assert not (gr == 0).any() or np.abs(g0[gr == 0]).max() == 0


def test_transr_hub_batch(dev):
    """A KG batch in which one entity heads 300 of 1,024 samples and is the tail of 100 more (its gradient row is the
    sum of 400 rows: the scatter's looks of sixteen rows, beyond its first 64 positions): kg_step against the autograd
    path, same bits; gradients against the torch restatement."""
    import dgl_kgat_amd as K
    torch.manual_seed(11)
    n, R, B = 3000, 7, 1024
    m = K.KGATPropagation(n, R, 64, 64, 2, 32, dropout=0.0).to(dev)
    h = torch.randint(0, n, (B,), device=dev); r = torch.randint(0, R, (B,), device=dev)
    pt = torch.randint(0, n, (B,), device=dev); nt = torch.randint(0, n, (B,), device=dev)
    h[:300] = 42
    pt[300:400] = 42
    grads = []
    for fused in (True, False):
        m.zero_grad()
        m.transR(h, r, pt, nt, fused=fused).backward()
        grads.append([x.grad.clone() for x in (m.entity_embed.weight, m.W_R, m.relation_embed.weight)])
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12
    g_again = []
    m.zero_grad()
    m.transR(h, r, pt, nt).backward()
    g_again = [x.grad.clone() for x in (m.entity_embed.weight, m.W_R, m.relation_embed.weight)]
    for a, b in zip(grads[0], g_again):
        assert torch.equal(a, b)


def test_bpr_under_the_training_stack(dev):
    """get_loss(gnn(g)) -> backward through the fused propagation stack: parameter gradients equal to those of the
    torch-operator loss."""
    import dgl_kgat_amd as K
    from dgl_kgat_amd import synth
    n, trip, R = synth.amazon_book_ckg(scale=0.02)
    torch.manual_seed(5)
    m = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
    g = synth.build_graph(n, trip, dev)
    with torch.no_grad():
        g.edata["w"] = m.compute_attention(g)
    B = 2048
    u, p, q = (torch.randint(0, n, (B,), device=dev) for _ in range(3))
    grads = []
    for fused in (True, False):
        m.zero_grad()
        m.get_loss(m.gnn(g), u, p, q, fused=fused).backward()
        grads.append([x.grad.clone() for x in m.parameters() if x.grad is not None])
    assert len(grads[0]) == len(grads[1]) >= 4
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12


def _ulp_diff(a, b):
    a, b = a.detach().cpu().numpy().ravel(), b.detach().cpu().numpy().ravel()
    ai, bi = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return int(np.abs(ai - bi).max())


def test_fused_adam_matches_torch_adam(dev):
    """Step by step from the same state: parameters and both moments within 1 ulp of torch.optim.Adam's (the same
    fp32 operations in the same order), over tensors of every alignment / size class, parameters that get a gradient
    only in some steps (their step count lags), state_dict interchange, and the gradient-clearing variant."""
    import dgl_kgat_amd as K
    torch.manual_seed(0)
    shapes = [(1000, 64), (41, 64, 64), (41, 64), (64, 64), (7,), (1,), (3, 5), (4099,)]
    ref_p = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    our_p = [torch.nn.Parameter(x.detach().clone()) for x in ref_p]
    ref = torch.optim.Adam(ref_p, lr=0.01)
    ours = K.FusedAdam(our_p, lr=0.01)
    worst = 0
    for it in range(12):
        for k, (a, b) in enumerate(zip(ref_p, our_p)):
            if k == 3 and it % 3 != 0:          # a parameter without a gradient in most steps
                a.grad = b.grad = None
                continue
            gscale = 10.0 ** ((it % 5) - 3)
            gr = torch.randn_like(a) * gscale
            if k == 0:
                gr[torch.rand(gr.shape[0], device=dev) < 0.9] = 0     # mostly zero rows: the dense semantics
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step()
        ours.step()
        for a, b in zip(ref_p, our_p):
            worst = max(worst, _ulp_diff(a, b))
            assert _ulp_diff(a, b) <= 1, (it, a.shape, _ulp_diff(a, b))
            sa, sb = ref.state[a], ours.state[b]
            if sa:
                assert float(sa["step"]) == float(sb["step"])
                assert _ulp_diff(sa["exp_avg"], sb["exp_avg"]) <= 1 and _ulp_diff(sa["exp_avg_sq"], sb["exp_avg_sq"]) <= 1
            # re-synchronise: the next step starts from identical state (a 1-ulp difference must not accumulate into the bar)
            with torch.no_grad():
                b.copy_(a)
                if sa:
                    sb["exp_avg"].copy_(sa["exp_avg"]); sb["exp_avg_sq"].copy_(sa["exp_avg_sq"])
    print("[adam] worst difference over 12 steps x 8 tensors: %d ulp" % worst)
    # state_dict moves between the two classes
    sd = copy.deepcopy(ref.state_dict())
    ours2 = K.FusedAdam([torch.nn.Parameter(x.detach().clone()) for x in ref_p], lr=0.01)
    ours2.load_state_dict(sd)
    ref2 = torch.optim.Adam([torch.nn.Parameter(x.detach().clone()) for x in ref_p], lr=0.01)
    ref2.load_state_dict(copy.deepcopy(ours.state_dict()))
    # the gradient-clearing variant: same update, gradients zero afterwards
    z = K.FusedAdam([torch.nn.Parameter(torch.ones(1000, 64, device=dev))], lr=0.1, zero_grads=True)
    pz = z.param_groups[0]["params"][0]
    pz.grad = torch.full_like(pz, 2.0)
    z.step()
    assert float(pz.grad.abs().max()) == 0 and abs(float(pz[0, 0]) - 0.9) < 1e-6
    with pytest.raises(NotImplementedError):
        K.FusedAdam(our_p, lr=0.01, weight_decay=0.1)
    with pytest.raises(K.KGATLibraryError):
        cpu = K.FusedAdam([torch.nn.Parameter(torch.ones(3))], lr=0.1)
        cpu.param_groups[0]["params"][0].grad = torch.ones(3)
        cpu.step()


def test_training_steps_with_fused_loss_and_optimiser(dev):
    """A few CF and KG steps of the epoch structure (kgat.py:114-168) with FusedAdam and the fused losses: the loss
    falls, and the parameters stay within rounding of the same steps taken with torch.optim.Adam."""
    import dgl_kgat_amd as K
    from dgl_kgat_amd import synth
    n, trip, R = synth.amazon_book_ckg(scale=0.02)
    g = synth.build_graph(n, trip, dev)
    B = 1024
    gen = torch.Generator(device="cpu").manual_seed(3)
    one = tuple(torch.randint(0, n, (B,), generator=gen).to(dev) for _ in range(3))
    batches = [one] * 5          # the same batch every step: the loss has to fall
    finals, losses = [], []
    for fused in (True, False):
        torch.manual_seed(9)
        m = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
        opt = K.FusedAdam(m.parameters(), lr=0.01) if fused else torch.optim.Adam(m.parameters(), lr=0.01)
        ls = []
        for u, p, q in batches:
            r = (u % R)
            loss = m.transR(u, r, p, q)
            loss.backward(); opt.step(); opt.zero_grad()
            with torch.no_grad():
                g.edata["w"] = m.compute_attention(g)
            loss = m.get_loss(m.gnn(g), u, p, q, fused=fused)
            loss.backward(); opt.step(); opt.zero_grad()
            ls.append(float(loss))
        finals.append([x.detach().clone() for x in m.parameters()])
        losses.append(ls)
    assert losses[0][-1] < losses[0][0] and losses[1][-1] < losses[1][0], losses
    assert max(abs(a - b) for a, b in zip(*losses)) <= 1e-3 * abs(losses[1][0]), losses
    for a, b in zip(*finals):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7, float((a - b).abs().max())


def test_kg_step_same_bits_as_the_autograd_path(dev):
    """KGATPropagation.kg_step (loss + gradients written straight into .grad, then the optimiser) against
    transR(...).backward(); optimizer.step(): the same parameters bit for bit after several iterations, with
    FusedAdam and with torch.optim.Adam, int64 and int32 ids."""
    import dgl_kgat_amd as K
    n, R, B = 4000, 7, 1500
    gen = torch.Generator(device="cpu").manual_seed(1)
    batches = [tuple(torch.randint(0, hi, (B,), generator=gen).to(dev) for hi in (n, R, n, n)) for _ in range(4)]
    for opt_cls in (K.FusedAdam, torch.optim.Adam):
        finals = []
        for direct in (True, False):
            torch.manual_seed(3)
            m = K.KGATPropagation(n, R, 64, 64, 2, 64, dropout=0.0).to(dev)
            opt = opt_cls(m.parameters(), lr=0.01)
            losses = []
            for i, (h, r, pt, nt) in enumerate(batches):
                if direct:
                    ids = (h.int(), r.int(), pt.int(), nt.int()) if i % 2 else (h, r, pt, nt)
                    losses.append(float(m.kg_step(*ids, opt, reg_lambda_kg=1e-3)))
                else:
                    loss = m.transR(h, r, pt, nt, reg_lambda_kg=1e-3)
                    loss.backward(); opt.step(); opt.zero_grad()
                    losses.append(float(loss.detach()))
            finals.append(([p.detach().clone() for p in m.parameters()], losses))
            assert all(p.grad is None for p in m.parameters())
        for a, b in zip(finals[0][0], finals[1][0]):
            assert torch.equal(a, b)
        assert finals[0][1] == finals[1][1]


def test_planted_structure_recall_rises(dev, capsys):
    """End to end (task 2): the epoch structure of the reference's kgat.py:114-196 - KG phase,
    attention refresh, CF phase, evaluation - on a CKG with PLANTED structure (examples/train_kgat.py::planted_data_dir:
    a user's held-out items share a KG attribute with its training items).  If the gradients (fused BPR / TransR /
    aggregation backward), the optimiser (one-launch Adam) and the attention refresh compose, recall@20 on the held-out
    interactions must leave the 20 / n_items = 0.02 of a random ranking within three short epochs; a sign error, a
    dropped gradient or a stale attention leaves it there (measured round 6: 0.022 -> 0.040 / 0.076 / 0.130)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import train_kgat
    hist = train_kgat.main(["--planted", "--epochs", "3", "--lr", "0.03", "--batch_size", "512", "--batch_size_kg", "512",
                            "--eval_before", "--seed", "1234"])
    rec = [h["test_recall"] for h in hist]
    val = [h["valid_recall"] for h in hist]
    with capsys.disabled():
        print("\nplanted-structure run: test recall@20 by epoch %s, valid %s, CF loss %s" % (
            ["%.4f" % r for r in rec], ["%.4f" % r for r in val], ["%.3f" % h["cf_loss"] for h in hist[1:]]))
    assert 0.005 < rec[0] < 0.05                       # untrained: a random ranking
    assert rec[3] > 3.0 * rec[0] and rec[3] > rec[2] > rec[1]
    assert val[3] > 3.0 * val[0]
    assert hist[3]["cf_loss"] < hist[1]["cf_loss"] and hist[3]["kg_loss"] < hist[1]["kg_loss"]


@pytest.mark.parametrize("n,R,B,d,k", [(4000, 7, 1500, 64, 64), (900, 3, 700, 20, 12), (3000, 41, 2048, 32, 64),
                                       (600000, 5, 512, 8, 8)])
def test_kg_phase_same_bits_as_kg_step_and_the_autograd_path(dev, n, R, B, d, k):
    """KGATPropagation.kg_phase (every batch sorted by one launch, then one three-launch library call per iteration:
    no dense entity gradient, the weight-gradient partials summed inside the Adam launch) against (a) a loop over
    kg_step with FusedAdam and (b) the reference's sequence transR(...).backward(); torch.optim.Adam.step(): the same
    parameters, moments, step counts and losses BIT FOR BIT after several iterations - hub entities (one node heads 300
    samples of a batch), widths off the MFMA form (20 x 12), 41 relations, node ids beyond 2^19 (the 64-bit sort), and a
    CF-style step of the same optimiser over all parameters afterwards (ent has then stepped more often than W_R)."""
    import dgl_kgat_amd as K
    gen = torch.Generator(device="cpu").manual_seed(n + B)
    n_it = 5
    ids = [torch.randint(0, hi, (n_it, B), generator=gen) for hi in (n, R, n, n)]
    ids[0][2] = ids[0][1]                  # consecutive batches that share most of their rows
    ids[2][3, :B // 2] = ids[2][2, :B // 2]
    ids[0][1, :min(300, B // 2)] = 42
    ids[2][1, B // 2:B // 2 + 100] = 42
    ids = [t.to(dev) for t in ids]
    results = []
    for mode in ("phase", "steps", "autograd"):
        torch.manual_seed(3)
        m = K.KGATPropagation(n, R, d, k, 2, 16, dropout=0.0).to(dev)
        opt = (torch.optim.Adam if mode == "autograd" else K.FusedAdam)(m.parameters(), lr=0.01)
        if mode == "phase":
            losses = m.kg_phase(ids[0].int(), ids[1].int(), ids[2].int(), ids[3].int(), opt, reg_lambda_kg=1e-3)
            assert all(p.grad is None for p in m.parameters())
            losses = losses.tolist()
        elif mode == "steps":
            losses = [float(m.kg_step(ids[0][i], ids[1][i], ids[2][i], ids[3][i], opt, reg_lambda_kg=1e-3)) for i in range(n_it)]
        else:
            losses = []
            for i in range(n_it):
                loss = m.transR(ids[0][i], ids[1][i], ids[2][i], ids[3][i], reg_lambda_kg=1e-3)
                loss.backward(); opt.step(); opt.zero_grad()
                losses.append(float(loss.detach()))
        kg_params = (m.entity_embed.weight, m.W_R, m.relation_embed.weight)
        state = [(float(opt.state[p]["step"]), opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) for p in kg_params]
        # one more step of the SAME optimiser with a dense gradient on the table only (what a CF step does to the counts)
        m.entity_embed.weight.grad = torch.full_like(m.entity_embed.weight, 1e-3)
        opt.step(); opt.zero_grad()
        if mode != "autograd":
            more = m.kg_phase(ids[0][:1].int(), ids[1][:1].int(), ids[2][:1].int(), ids[3][:1].int(), opt, reg_lambda_kg=1e-3) \
                if mode == "phase" else m.kg_step(ids[0][0], ids[1][0], ids[2][0], ids[3][0], opt, reg_lambda_kg=1e-3).reshape(1)
        else:
            loss = m.transR(ids[0][0], ids[1][0], ids[2][0], ids[3][0], reg_lambda_kg=1e-3)
            loss.backward(); opt.step(); opt.zero_grad()
            more = loss.detach().reshape(1)
        results.append(([p.detach().clone() for p in m.parameters()], losses + more.tolist(), state,
                        [float(opt.state[p]["step"]) for p in kg_params]))
    ref = results[2]
    for got in results[:2]:
        assert got[1] == ref[1], (got[1], ref[1])
        for a, b in zip(got[0], ref[0]):
            assert torch.equal(a, b)
        for (sa, ma, va), (sb, mb, vb) in zip(got[2], ref[2]):
            assert sa == sb == n_it and torch.equal(ma, mb) and torch.equal(va, vb)
        assert got[3] == ref[3] == [n_it + 2, n_it + 1, n_it + 1]
    assert ref[1][-1] < ref[1][0]          # batch 0 again, after the updates


def test_kg_phase_falls_back_to_kg_step(dev):
    """torch.optim.Adam (not FusedAdam), an empty phase, mismatched shapes."""
    import dgl_kgat_amd as K
    torch.manual_seed(0)
    n, R, B = 500, 4, 256
    ids = [torch.randint(0, hi, (3, B), device=dev) for hi in (n, R, n, n)]
    outs = []
    for via_phase in (True, False):
        torch.manual_seed(1)
        m = K.KGATPropagation(n, R, 16, 16, 1, 16, dropout=0.0).to(dev)
        opt = torch.optim.Adam(m.parameters(), lr=0.01)
        if via_phase:
            ls = m.kg_phase(*ids, opt).tolist()
        else:
            ls = [float(m.kg_step(ids[0][i], ids[1][i], ids[2][i], ids[3][i], opt)) for i in range(3)]
        outs.append((ls, [p.detach().clone() for p in m.parameters()]))
    assert outs[0][0] == outs[1][0] and all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    m = K.KGATPropagation(n, R, 16, 16, 1, 16, dropout=0.0).to(dev)
    opt = K.FusedAdam(m.parameters(), lr=0.01)
    assert m.kg_phase(*[t[:0] for t in ids], opt).shape == (0,)
    with pytest.raises(ValueError):
        m.kg_phase(ids[0], ids[1][:2], ids[2], ids[3], opt)


@pytest.mark.parametrize("beta1", [0.3, 0.5, 0.0])
def test_fused_adam_small_beta1_matches_torch(dev, beta1):
    """torch's lerp switches formula at weight 0.5, i.e. for beta1 <= 0.5: the kernel follows it."""
    import dgl_kgat_amd as K
    torch.manual_seed(4)
    a = torch.nn.Parameter(torch.randn(3000, 16, device=dev))
    b = torch.nn.Parameter(a.detach().clone())
    ref, ours = torch.optim.Adam([a], lr=0.01, betas=(beta1, 0.99)), K.FusedAdam([b], lr=0.01, betas=(beta1, 0.99))
    for it in range(6):
        g = torch.randn_like(a) * 10.0 ** (it % 3 - 2)
        a.grad, b.grad = g.clone(), g.clone()
        ref.step(); ours.step()
        assert _ulp_diff(a, b) == 0 and _ulp_diff(ref.state[a]["exp_avg"], ours.state[b]["exp_avg"]) == 0, (beta1, it)
