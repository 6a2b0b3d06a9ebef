"""Developer probe (round 5, VERDICT r4 task 8b): does a tile -> XCD assignment by SOURCE affinity lift the gather
ceiling?  kgat_gather_probe_f32 reads positions in workgroups of 2,048, and workgroup b runs on XCD b mod 8; permuting
the 2,048-position blocks of the CSR source array therefore emulates any static tile -> XCD map without touching a
kernel.  Maps: identity (the aggregation's), a contiguous eighth per XCD, blocks dealt to XCDs by their dominant
source-id eighth (so that an XCD's 4 MiB L2 sees one eighth of X, 5 MB, most of the time), by node type of the sources."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dgl_kgat_amd import ops, synth

dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg(seed=1234, scale=1.0)
g = synth.build_graph(n, trip, dev)
csr = g._st.csr(dev)
col = csr.col.cpu().numpy()
E = col.shape[0]
x = torch.randn(n, 64, device=dev)
BLK = 2048
nb = E // BLK            # whole blocks only (the ragged tail stays in place)

def t(colv, launches=40):
    c = torch.as_tensor(colv, device=dev)
    sink = ops.gather_probe(c, x)
    for _ in range(5): ops.gather_probe(c, x, sink)
    ts = []
    for _ in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gather_probe(c, x, sink); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e3

def deal(xcd_of_block):
    """block order in which launch block b (XCD b mod 8) is a block assigned to that XCD, lists balanced by moving the
    excess of full XCDs to the emptiest ones"""
    lists = [list(np.nonzero(xcd_of_block == k)[0]) for k in range(8)]
    quota = (nb + 7) // 8
    spare = []
    for l in lists:
        while len(l) > quota: spare.append(l.pop())
    for l in lists:
        while len(l) < quota and spare: l.append(spare.pop())
    order = []
    for i in range(quota):
        for k in range(8):
            if i < len(lists[k]): order.append(lists[k][i])
    order = np.array(order[:nb])
    out = col.copy()
    out[:nb * BLK] = col[:nb * BLK].reshape(nb, BLK)[order].reshape(-1)
    return out

blocks = col[:nb * BLK].reshape(nb, BLK)
print("E = %d, %d blocks of %d positions; X = %.1f MB, eight L2s of 4 MiB" % (E, nb, BLK, n * 256 / 1e6))
print("identity (block b -> XCD b mod 8)            %.1f us" % t(col))
print("a contiguous eighth of the positions per XCD %.1f us" % t(deal(np.arange(nb) * 8 // nb)))
eighth = np.minimum(blocks.astype(np.int64) * 8 // n, 7)
dom = np.array([np.bincount(e, minlength=8).argmax() for e in eighth])
share = np.mean([np.bincount(e, minlength=8).max() / BLK for e in eighth])
print("by dominant source-id eighth (mean share of the dominant eighth inside a block %.2f)  %.1f us" % (share, t(deal(dom))))
n_users, n_items = 70679, 24915
typ = np.where(blocks < n_users, 0, np.where(blocks < n_users + n_items, 1, 2))
tdom = np.array([np.bincount(e, minlength=3).argmax() for e in typ])
# users' rows (sources = items) to XCDs 0-1, the rest spread: type 0 -> {0..2}, 1 -> {3..5}, 2 -> {6,7} by block index
xt = np.where(tdom == 0, np.arange(nb) % 3, np.where(tdom == 1, 3 + np.arange(nb) % 3, 6 + np.arange(nb) % 2))
print("by dominant source node type (users / items / entities: %s blocks)  %.1f us" % (np.bincount(tdom, minlength=3).tolist(), t(deal(xt))))
srt = np.sort(col)
print("sorted sources (every XCD sees the whole of X once, in order)   %.1f us" % t(srt))
