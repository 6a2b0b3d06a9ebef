import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import dgl_kgat_amd as K
from dgl_kgat_amd import synth
dev = torch.device('cuda:0')
n, trip, R = synth.amazon_book_ckg()
torch.manual_seed(0)
model = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.1).to(dev)
opt = K.FusedAdam(model.parameters(), lr=0.001)
gen = torch.Generator(device="cpu").manual_seed(99)
B = 2048
idx = torch.randint(0, len(trip), (B,), generator=gen).numpy()
h = torch.as_tensor(trip[idx, 0].astype(np.int32), device=dev)
r = torch.as_tensor(trip[idx, 1].astype(np.int32), device=dev)
pt = torch.as_tensor(trip[idx, 2].astype(np.int32), device=dev)
nt = torch.randint(0, n, (B,), generator=gen).int().to(dev)
def kg_step():
    loss = model.transR(h, r, pt, nt, reg_lambda_kg=1e-4)
    loss.backward(); opt.step(); opt.zero_grad()
for _ in range(10): kg_step()
torch.cuda.synchronize()
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(100): kg_step()
    t1 = time.perf_counter()
    b.record(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("KG step: wall %.3f ms, host enqueue %.3f ms, device span %.3f ms" % ((t2 - t0) * 10, (t1 - t0) * 10, a.elapsed_time(b) / 100))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): kg_step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
