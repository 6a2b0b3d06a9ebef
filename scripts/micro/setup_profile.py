"""Developer probe (round 6): where the host time of a graph's FIRST step goes (structure builds: CSR, relation /
head groups, work tiles) - the device idles through it and the following steps run 4-9 % slower per kernel
(profiles/r06_first_steps_trace.txt: clock ramp)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dgl_kgat_amd as K
from dgl_kgat_amd import synth
dev = torch.device("cuda:0")
n, trip, n_rel = synth.amazon_book_ckg(seed=1234, scale=1.0)
torch.manual_seed(0)
model = K.KGATPropagation(n, n_rel, 64, 64, 3, 64, 0.0).to(dev)
for rep in range(2):
    g = synth.build_graph(n, trip, dev)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    with torch.no_grad():
        g.edata["w"] = model.compute_attention(g)
        out = model.gnn(g)
    torch.cuda.synchronize()
    pr.disable()
    print("first step of a fresh graph: %.2f ms wall" % ((time.perf_counter() - t0) * 1e3))
    if rep == 1:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
