"""Developer probe (round 5): the pure row gather (kgat_gather_probe_f32, D = 64) over the amazon-book-shaped graph's
sources in the three orders the step reads them: CSR (destination-major; the aggregation), relation-grouped (the
attention's tail rows) and the head rows of the (head, relation) groups; plus a uniformly random index of each length."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dgl_kgat_amd as K
from dgl_kgat_amd import ops, synth

dev = torch.device("cuda:0")
n, trip, n_rel = synth.amazon_book_ckg(seed=1234, scale=1.0)
g = synth.build_graph(n, trip, dev)
st = g._st
et = g.edata["type"]
x = torch.randn(n, 64, device=dev)
csr = st.csr(dev)
groups = st.rel_groups(et, n_rel, dev)

def t(col, launches=40):
    sink = ops.gather_probe(col, x)
    for _ in range(5): ops.gather_probe(col, x, sink)
    ts = []
    for _ in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gather_probe(col, x, sink); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e3

E = csr.col.numel()
print("rows gathered: E = %d (938 MB at 256 B)" % E)
print("CSR order (aggregation)            %.1f us" % t(csr.col))
print("relation-grouped order (attention) %.1f us" % t(groups.src_g))
print("uniformly random, E indices        %.1f us" % t(torch.randint(0, n, (E,), device=dev, dtype=torch.int32)))
print("sorted sources (best case)         %.1f us" % t(torch.sort(csr.col).values.contiguous()))
gn = groups.g_node
print("head rows of %d groups              %.1f us" % (gn.numel(), t(gn)))
both = torch.cat([groups.src_g, gn])
print("tails + heads back to back (%d)     %.1f us" % (both.numel(), t(both)))
# interleaved the way the fused kernel reads them: per 16-group tile its ~60 tail rows, then its 16 head rows
gid = groups.gid.long()                      # group of every grouped position
tile_of_pos = gid // 16
tile_of_grp = torch.arange(gn.numel(), device=dev) // 16
key = torch.cat([tile_of_pos * 2, tile_of_grp * 2 + 1])
order = torch.sort(key, stable=True).indices
inter = both[order].contiguous()
print("tails + heads interleaved per tile          %.1f us" % t(inter))
# the same rows, but every XCD (workgroup id mod 8) reading a contiguous eighth: the kernel's tile ranges
print("(for scale) tails twice                      %.1f us" % t(torch.cat([groups.src_g, groups.src_g])))
