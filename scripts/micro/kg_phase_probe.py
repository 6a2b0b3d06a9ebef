"""Developer probe (round 6): the KG phase's iteration (KGATPropagation.kg_phase) on the amazon-book-shaped CKG -
wall clock per iteration over N iterations, for a kernel trace under rocprofv3 (scripts/r06_kg_trace.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dgl_kgat_amd as K
from dgl_kgat_amd import synth

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
n, trip, n_rel = synth.amazon_book_ckg(seed=1234, scale=1.0)
torch.manual_seed(0)
model = K.KGATPropagation(n, n_rel, 64, 64, 3, 64, 0.1).to(dev)
opt = K.FusedAdam(model.parameters(), lr=1e-3)
cols = torch.as_tensor(np.ascontiguousarray(trip.T.astype(np.int32)), device=dev)
for rep in range(3):
    idx = torch.randint(0, cols.shape[1], (n_it, 2048), device=dev)
    neg = torch.randint(0, n, (n_it, 2048), device=dev, dtype=torch.int32)
    h, r, t = cols[0][idx], cols[1][idx], cols[2][idx]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = model.kg_phase(h, r, t, neg, opt, reg_lambda_kg=1e-4)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("kg_phase: %d iterations, %.4f ms per iteration (host issue %.4f ms per iteration), loss %.4f -> %.4f" % (
        n_it, dt / n_it * 1e3, t_issue / n_it * 1e3, float(losses[0]), float(losses[-1])))
