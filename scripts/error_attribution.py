#!/usr/bin/env python3
"""Which kernel moves a small-graph readout away from fp64 (developer tool; round-2 review: the
smoke graph's n(h2) sat 2.6 x further from fp64 on the device than in the C fp32 run).

The step is evaluated in fp64 by the oracle; then, one stage at a time, that stage alone is
replaced by the device kernel fed with the fp64 chain's (fp32-rounded) inputs, the rest of the
chain staying fp64.  Printed: SURVEY 8c metric of every readout block for each substitution, for
the all-device run and for the C fp32 run.

  python scripts/error_attribution.py [--seed 7] [--nodes 40,60,50]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel_err(x, y):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    return float(np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-3 * np.abs(y).max())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--nodes", default="40,60,50")
    ap.add_argument("--kg", type=int, default=1500)
    ap.add_argument("--uv", type=int, default=700)
    args = ap.parse_args()
    import dgl_kgat_amd as K
    from dgl_kgat_amd import ops, synth
    from oracle import c_oracle as co
    from oracle import kgat_oracle as orc
    dev = torch.device("cuda:0")
    nu, ni, na = (int(x) for x in args.nodes.split(","))
    n, trip, R = synth.collaborative_kg(nu, ni, na, 4, args.kg, args.uv, seed=args.seed)
    torch.manual_seed(args.seed)
    model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
    g = synth.build_graph(n, trip, dev)
    src, dst, et = trip[:, 2], trip[:, 0], trip[:, 1]
    p = {k: v.detach().cpu().double().numpy() for k, v in model.state_dict().items()}
    ent, W_R, rel = p["entity_embed.weight"], p["W_R"], p["relation_embed.weight"]
    W2 = [p["layers.%d.res_fc_2.weight" % i] for i in range(3)]
    widths = [64, 64, 32, 16]
    st = g._st
    csr = st.csr(dev)
    eid = csr.eid.cpu().numpy()

    def t32(x):
        return torch.as_tensor(np.asarray(x, np.float32), device=dev)

    def chain(sub=None):
        """fp64 chain with stage `sub` on the device: 'logits', 'softmax', ('spmm', l), ('bi', l)."""
        logits = orc.att_score(ent, W_R, rel, src, dst, et)
        if sub == "logits":
            with torch.no_grad():
                groups = st.rel_groups(g.edata["type"], R, dev)
                from dgl_kgat_amd.graph import _fused_tiles
                tiles = _fused_tiles(groups, 64)
                lc = ops.att_score_fused(n, groups.rel_ptr, groups.perm, groups.src_g, groups.pos_g, groups.gid,
                                         groups.gptr, groups.g_node, tiles[0], tiles[1], t32(ent), t32(W_R), t32(rel),
                                         want_eid=True, part_tptr=tiles[2])[0]
            logits = lc.cpu().double().numpy()
        a = orc.edge_softmax(n, dst, logits)
        if sub == "softmax":
            a = ops.edge_softmax(csr.indptr, csr.row_of, csr.eid, t32(logits))[0].cpu().double().numpy()
        h, cache = ent, [ent]
        for li, W in enumerate(W2):
            hn = orc.spmm_u_mul_e_sum_sparse(n, src, dst, h, a)
            if sub == ("spmm", li):
                hn = ops.spmm(csr.indptr, csr.col, csr.row_of, t32(h), t32(a[eid])).cpu().double().numpy()
            z = orc.bi_interaction(h, hn, W)
            nz = orc.l2_normalize(z)
            if sub == ("bi", li):
                no = torch.empty((n, W.shape[0]), device=dev)
                z = ops.bi_interaction(t32(h * hn), t32(W), 0.01, norm_out=no).cpu().double().numpy()
                nz = no.cpu().double().numpy()
            h = z
            cache.append(nz)
        return a, np.concatenate(cache, 1)

    a64, out64 = chain()
    with torch.no_grad():
        a_g = model.compute_attention(g)
        g.edata["w"] = a_g
        out_g = model.gnn(g).cpu().double().numpy()
    indptr, col, eid_c = co.csr_from_coo(n, src, dst)
    a_c = co.edge_softmax(n, indptr, eid_c, co.att_score(ent, W_R, rel, src, dst, et))
    h, cache = ent.astype(np.float32), [ent.astype(np.float32)]
    for W in W2:
        h = co.bi_interaction(h, co.spmm(n, indptr, col, eid_c, h, a_c), W)
        cache.append(co.l2_normalize(h))
    out_c = np.concatenate(cache, 1)

    def blocks(x):
        o, res = 0, []
        for w in widths:
            res.append(x[:, o:o + w])
            o += w
        return res

    def line(name, a, out):
        errs = [rel_err(x, y) for x, y in zip(blocks(out)[1:], blocks(out64)[1:])]
        print("%-22s attention %.2e | n(h1) %.2e  n(h2) %.2e  n(h3) %.2e" % (name, rel_err(a.reshape(-1), a64.reshape(-1)), *errs))

    print("N=%d E=%d  (8c metric vs the fp64 chain)" % (n, len(trip)))
    line("device, whole step", a_g.cpu().double().numpy(), out_g)
    line("C fp32, whole step", a_c, out_c)
    for sub in ["logits", "softmax"] + [(k, li) for li in range(3) for k in ("spmm", "bi")]:
        a, out = chain(sub)
        line("only %s on device" % (sub if isinstance(sub, str) else "%s layer %d" % sub), a, out)


if __name__ == "__main__":
    main()
