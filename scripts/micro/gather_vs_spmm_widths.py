import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from dgl_kgat_amd import ops, synth
dev = torch.device("cuda:0")
n, trip, R = synth.amazon_book_ckg()
indptr, col, eid, row_of = ops.csr_from_coo(n, torch.as_tensor(trip[:, 2].copy(), device=dev), torch.as_tensor(trip[:, 0].copy(), device=dev))
w = torch.rand(col.numel(), device=dev)
def ev(fn, k=60):
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); out.append((a, b))
    torch.cuda.synchronize()
    return 1e3 * float(np.median([a.elapsed_time(b) for a, b in out][10:]))
for D in (16, 32, 64, 128):
    X = torch.randn(n, D, device=dev)
    sink = ops.gather_probe(col, X)
    out = torch.empty((n, D), device=dev); ws = ops.spmm_workspace(col.numel(), D, dev)
    tp = ev(lambda: ops.gather_probe(col, X, sink)); ts = ev(lambda: ops.spmm(indptr, col, row_of, X, w, out=out, workspace=ws))
    print("D=%3d gather probe %.1f us  plain aggregation %.1f us  ratio %.2f" % (D, tp, ts, tp / ts))
