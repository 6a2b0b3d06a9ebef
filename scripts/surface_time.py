import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgl_kgat_amd as K
from dgl_kgat_amd import synth
dev = torch.device('cuda:0')
n, trip, R = synth.amazon_book_ckg()
torch.manual_seed(0)
m = K.KGATPropagation(n, R, 64, 64, 3, 64, dropout=0.0).to(dev)
g = synth.build_graph(n, trip, dev)
with torch.no_grad():
    for name, fn in (("surface (reference call sequence: 41x filter_edges/apply_edges + edge_softmax)", m.compute_attention_surface),
                     ("fused (kgat_attention)", m.compute_attention)):
        a = fn(g); torch.cuda.synchronize()
        t = time.time()
        for _ in range(3): a = fn(g)
        torch.cuda.synchronize()
        print("%-90s %.2f ms" % (name, (time.time() - t) / 3 * 1e3))
    a1, a2 = m.compute_attention_surface(g), m.compute_attention(g)
    print("max abs diff surface vs fused: %.3e" % float((a1 - a2).abs().max()))
    g.edata['w'] = a2
    for name, fused in (("gnn surface (update_all + torch dense)", False), ("gnn fused", True)):
        o = m.gnn(g, fused=fused); torch.cuda.synchronize()
        t = time.time()
        for _ in range(10): o = m.gnn(g, fused=fused)
        torch.cuda.synchronize()
        print("%-90s %.3f ms" % (name, (time.time() - t) / 10 * 1e3))
