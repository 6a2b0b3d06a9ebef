"""Developer probe (round 5): are the fused attention kernel's per-workgroup times repeatable, and does re-cutting the
workgroups' tile ranges from ONE timed launch (kgat_att_score_fused_timed_f32) bring their end times together?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dgl_kgat_amd as K
from dgl_kgat_amd import ops, synth, graph as G   # (graph internals: the tiles and records of the fused form)

dev = torch.device("cuda:0")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
IN_STEP = len(sys.argv) > 2 and sys.argv[2] == "step"   # run the propagation layers between two attention launches
n, trip, R = synth.amazon_book_ckg(seed=1234, scale=1.0)
g = synth.build_graph(n, trip, dev)
st = g._st
groups = st.rel_groups(g.edata["type"], R, dev)
tiles, rel_tptr, part_tptr = G._fused_tiles(groups, d, 16)
rec_g, _ = G._fused_statics(groups, 16)
torch.manual_seed(0)
ent = torch.randn(n, d, device=dev) * 0.1
W = (torch.rand(R, d, d, device=dev) - 0.5) * (2 / d ** 0.5)
rel = torch.randn(R, d, device=dev) * 0.1
n_parts = part_tptr.numel() - 1
cost = ops.fold_tile_cost(d)

model = K.KGATPropagation(n, R, d, d, 3, d, dropout=0.0).to(dev)
with torch.no_grad():
    g.edata["w"] = torch.rand(st.n_edges, 1, device=dev)

def between():
    if IN_STEP:
        with torch.no_grad():
            model.gnn(g)

def launch(parts, clocks=None):
    between()
    return ops.att_score_fused(n, groups.rel_ptr, None, None, None, None, groups.gptr, groups.g_node, tiles, rel_tptr, ent, W, rel,
                               want_csr=False, want_eid=False, part_tptr=parts, rec_g=rec_g, want_grouped=True, part_clocks=clocks)

def timed(parts, reps=5):
    out = []
    for _ in range(reps):
        clk = torch.zeros(2 * n_parts, dtype=torch.int64, device=dev)
        launch(parts, clk)
        torch.cuda.synchronize()
        c = clk.cpu().numpy().reshape(-1, 2)
        out.append(((c[:, 1] - c[:, 0]) / 100.0, (c[:, 1].max() - c[:, 0].min()) / 100.0))   # us
    return out

def event_time(parts, reps=60):
    for _ in range(10): launch(parts)
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        between()
        a.record(); ops.att_score_fused(n, groups.rel_ptr, None, None, None, None, groups.gptr, groups.g_node, tiles, rel_tptr, ent, W, rel,
                                        want_csr=False, want_eid=False, part_tptr=parts, rec_g=rec_g, want_grouped=True)
        b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))

# model cost per tile on the host
T = int(rel_tptr[-1])
tl = tiles.cpu().numpy().reshape(-1, 4)[:T]
P = tl[:, 3] - tl[:, 2]
later = np.where(P > 64, (P - 64 + 63) // 64, 0)
rt = rel_tptr.cpu().numpy()
opens = np.zeros(T, dtype=np.int64); opens[rt[:-1][rt[:-1] < T]] = 1
c_model = cost[0] + cost[1] * later + cost[2] * opens

parts = part_tptr.clone()
ref0 = launch(parts)[2].clone()
for it in range(4):
    runs = timed(parts)
    durs = np.stack([r[0] for r in runs])
    span = np.median([r[1] for r in runs])
    med = np.median(durs, 0)
    cc = np.corrcoef(durs[0], durs[1])[0, 1]
    print("iteration %d: launch %.1f us (event-timed median), span of the timed launches %.1f; workgroups: mean %.1f max %.1f "
          "(max/mean %.3f); run-to-run correlation of the workgroups' times %.3f; std over runs / spread over workgroups %.2f"
          % (it, event_time(parts), span, med.mean(), med.max(), med.max() / med.mean(), cc, durs.std(0).mean() / med.std()))
    pb = parts.cpu().numpy().astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(c_model)])
    C_p = csum[pb[1:]] - csum[pb[:-1]]
    scale = med / np.maximum(C_p, 1)
    part_of_tile = np.searchsorted(pb[1:], np.arange(T), side="right")
    w = c_model * scale[np.minimum(part_of_tile, n_parts - 1)]
    ws = np.concatenate([[0], np.cumsum(w)])
    targets = ws[-1] * np.arange(n_parts + 1) / n_parts
    nb = np.searchsorted(ws, targets, side="left").astype(np.int32)
    nb[0], nb[-1] = 0, T
    nb = np.maximum.accumulate(nb)
    parts = torch.as_tensor(nb, device=dev)
    out = launch(parts)[2]
    assert torch.equal(out, ref0), "results changed with the split"
