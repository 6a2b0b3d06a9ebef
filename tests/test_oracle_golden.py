"""CPU: pin the oracle (numpy + C restatements) against the fixtures generated from the
reference's own models.py (tests/golden/make_golden.py) and against hand-computed
known answers."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import c_oracle as co
from oracle import kgat_oracle as orc


def test_att_score_matches_reference(golden):
    g = golden
    s = orc.att_score(g["entity_embed"], g["W_R"], g["relation_embed"], g["src"], g["dst"], g["etype"])
    for r in range(g["R"]):
        eids = g["att_eids_%d" % r]
        ref = g["att_score_%d" % r].reshape(-1)
        assert np.array_equal(eids, np.nonzero(g["etype"] == r)[0])
        assert rel_err(s[eids], ref) < 1e-12
    s32 = co.att_score(g["entity_embed"], g["W_R"], g["relation_embed"], g["src"], g["dst"], g["etype"])
    assert rel_err(s32, s) < 1e-4


def test_attention_matches_reference_glue(golden):
    g = golden
    a = orc.compute_attention(g["n"], g["src"], g["dst"], g["etype"], g["entity_embed"], g["W_R"],
                              g["relation_embed"])
    assert a.shape == g["attention"].shape == (len(g["src"]), 1)
    assert rel_err(a, g["attention"]) < 1e-12
    indptr, col, eid = co.csr_from_coo(g["n"], g["src"], g["dst"])
    s32 = co.att_score(g["entity_embed"], g["W_R"], g["relation_embed"], g["src"], g["dst"], g["etype"])
    a32 = co.edge_softmax(g["n"], indptr, eid, s32)
    assert rel_err(a32, g["attention"].reshape(-1)) < 1e-4


def test_gnn_matches_reference_glue(golden):
    g = golden
    out = orc.gnn_forward(g["n"], g["src"], g["dst"], g["attention"], g["entity_embed"], g["W2"])
    assert out.shape == g["gnn_out"].shape
    assert rel_err(out, g["gnn_out"]) < 1e-11
    # layer by layer, C fp32 restatement
    indptr, col, eid = co.csr_from_coo(g["n"], g["src"], g["dst"])
    h = g["entity_embed"].astype(np.float32)
    w = g["attention"].reshape(-1).astype(np.float32)
    for i, W2 in enumerate(g["W2"]):
        hn = co.spmm(g["n"], indptr, col, eid, h, w)
        h = co.bi_interaction(h, hn, W2)
        assert rel_err(h, g["layer_out_%d" % i]) < 1e-4
    fused = co.spmm(g["n"], indptr, col, eid, g["entity_embed"], w, mul_self=True)
    plain = co.spmm(g["n"], indptr, col, eid, g["entity_embed"], w)
    assert np.array_equal(fused, plain * g["entity_embed"].astype(np.float32))


def test_csr_and_relation_grouping(golden):
    g = golden
    indptr, col, eid = orc.csr_from_coo(g["n"], g["src"], g["dst"])
    ci, cc, ce = co.csr_from_coo(g["n"], g["src"], g["dst"])
    assert np.array_equal(indptr, ci) and np.array_equal(col, cc) and np.array_equal(eid, ce)
    assert indptr[0] == 0 and indptr[-1] == len(g["src"])
    for v in range(g["n"]):
        seg = eid[indptr[v]:indptr[v + 1]]
        assert np.all(g["dst"][seg] == v) and np.all(np.diff(seg) > 0)
        assert np.array_equal(col[indptr[v]:indptr[v + 1]], g["src"][seg])
    rp, perm = orc.group_by_relation(g["etype"], g["R"])
    rp2, perm2 = co.group_by_relation(g["etype"], g["R"])
    assert np.array_equal(rp, rp2) and np.array_equal(perm, perm2)
    for r in range(g["R"]):
        assert np.array_equal(perm[rp[r]:rp[r + 1]], np.nonzero(g["etype"] == r)[0])
    # types outside [0, R) are never scored
    et = g["etype"].copy()
    et[:7] = g["R"] + 3
    et[7:9] = -1
    rp3, perm3 = co.group_by_relation(et, g["R"])
    assert rp3[-1] == len(et) - 9 and sorted(perm3[rp3[-1]:]) == list(range(9))
    s = orc.att_score(g["entity_embed"], g["W_R"], g["relation_embed"], g["src"], g["dst"], et)
    assert np.all(s[:9] == 0)
    s32 = co.att_score(g["entity_embed"], g["W_R"], g["relation_embed"], g["src"], g["dst"], et)
    assert np.all(s32[:9] == 0)


def test_dense_cross_check(golden):
    g = golden
    if g["n"] > 64:
        pytest.skip("dense formulation only on the small graphs")
    rng = np.random.default_rng(0)
    X = rng.standard_normal((g["n"], 5))
    w = rng.random(len(g["src"]))
    a = orc.spmm_u_mul_e_sum(g["n"], g["src"], g["dst"], X, w)
    b = orc.dense_spmm(g["n"], g["src"], g["dst"], X, w)
    assert np.allclose(a, b, rtol=1e-12, atol=1e-13)
    s = rng.standard_normal(len(g["src"])) * 3
    assert np.allclose(orc.edge_softmax(g["n"], g["dst"], s), orc.dense_edge_softmax(g["n"], g["dst"], s),
                       rtol=1e-12, atol=0)
    gx = orc.spmm_backward_x(g["n"], g["src"], g["dst"], a, w)
    assert np.allclose(gx, orc.dense_spmm(g["n"], g["dst"], g["src"], a, w))


def test_known_answers():
    # single edge -> softmax 1; two equal logits -> .5/.5; isolated dst -> zero row
    src = np.array([1, 2, 3, 0]); dst = np.array([0, 0, 2, 3]); n = 5
    a = orc.edge_softmax(n, dst, np.array([0.3, 0.3, -7.0, 80.0]))
    assert np.allclose(a, [0.5, 0.5, 1.0, 1.0])
    X = np.arange(10.0).reshape(5, 2)
    out = orc.spmm_u_mul_e_sum(n, src, dst, X, a)
    assert np.allclose(out[0], 0.5 * X[1] + 0.5 * X[2]) and np.all(out[1] == 0) and np.all(out[4] == 0)
    W2 = np.eye(2)
    h1 = orc.bi_interaction(X, out, W2)
    assert np.all(h1[1] == 0) and np.all(orc.l2_normalize(h1)[1] == 0)
    # shift invariance per destination; large logits stay finite
    s = np.array([80.0, -80.0, 5.0, 88.0])
    shift = np.array([11.0, 11.0, -3.0, 100.0])
    assert np.allclose(orc.edge_softmax(n, dst, s), orc.edge_softmax(n, dst, s + shift), atol=1e-15)
    indptr, col, eid = co.csr_from_coo(n, src, dst)
    a32 = co.edge_softmax(n, indptr, eid, (s + shift).astype(np.float32))
    assert np.all(np.isfinite(a32)) and np.allclose(a32, orc.edge_softmax(n, dst, s), atol=1e-6)


def test_permutation_invariance(golden):
    g = golden
    rng = np.random.default_rng(3)
    p = rng.permutation(len(g["src"]))
    a = g["attention"].reshape(-1)
    X = g["entity_embed"]
    base = orc.spmm_u_mul_e_sum(g["n"], g["src"], g["dst"], X, a)
    perm = orc.spmm_u_mul_e_sum(g["n"], g["src"][p], g["dst"][p], X, a[p])
    assert np.allclose(base, perm, rtol=1e-12, atol=1e-14)
    s = rng.standard_normal(len(p))
    assert np.allclose(orc.edge_softmax(g["n"], g["dst"], s)[p], orc.edge_softmax(g["n"], g["dst"][p], s[p]))


def test_edge_softmax_backward_fd():
    rng = np.random.default_rng(5)
    n, e = 6, 20
    dst = rng.integers(0, n, e)
    s = rng.standard_normal(e)
    ga = rng.standard_normal(e)
    a = orc.edge_softmax(n, dst, s)
    gs = orc.edge_softmax_backward(n, dst, a, ga).reshape(-1)
    eps = 1e-6
    for i in range(e):
        sp = s.copy(); sp[i] += eps
        sm = s.copy(); sm[i] -= eps
        fd = np.dot(ga, orc.edge_softmax(n, dst, sp) - orc.edge_softmax(n, dst, sm)) / (2 * eps)
        assert abs(fd - gs[i]) < 1e-7


def test_c1_config_cpu_anchor():
    """BASELINE configs[0] - the reference's own CPU-runnable case (last-fm CKG, 1 propagation
    layer, embed_dim = 8) - on a 5 % last-fm-shaped graph: the two independently written CPU
    restatements (numpy fp64, C/OpenMP fp32) agree on the whole path."""
    from dgl_kgat_amd import synth
    n, trip, R = synth.last_fm_ckg(scale=0.05)
    src, dst, et = trip[:, 2], trip[:, 0], trip[:, 1]
    rng = np.random.default_rng(1234)
    d = 8
    ent = rng.standard_normal((n, d)).astype(np.float32)
    W_R = ((rng.random((R, d, d)) - 0.5) * 0.8).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    W2 = (rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
    a64 = orc.compute_attention(n, src, dst, et, ent, W_R, rel)
    out64 = orc.gnn_forward(n, src, dst, a64, ent, [W2])
    indptr, col, eid = co.csr_from_coo(n, src, dst)
    a32 = co.edge_softmax(n, indptr, eid, co.att_score(ent, W_R, rel, src, dst, et))
    h = co.bi_interaction(ent, co.spmm(n, indptr, col, eid, ent, a32), W2)
    out32 = np.concatenate([ent, co.l2_normalize(h)], 1)
    assert rel_err(a32, a64.reshape(-1)) < 1e-4
    assert np.max(np.abs(out32 - out64)) < 1e-4 * np.abs(out64).max()
