"""Developer probe (round 6): the edge-id-ordered copy of the attention weights - the shipped gather against a form in
which every XCD reads one L2-resident eighth of the table (scripts/micro/perm_probe.hip)."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
from dgl_kgat_amd import ops, synth
so, src_f = os.path.join(HERE, "build", "perm_probe.so"), os.path.join(HERE, "perm_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src_f):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, src_f])
lib = ctypes.CDLL(so)
P = ctypes.c_void_p
lib.perm_plain.argtypes = [ctypes.c_int64, P, P, P, P]
lib.perm_xcd.argtypes = [ctypes.c_int64, P, P, P, P, P, P]
dev = torch.device("cuda:0")
n, trip, n_rel = synth.amazon_book_ckg(seed=1234, scale=1.0)
g = synth.build_graph(n, trip, dev)
st = g._st
csr = st.csr(dev)
E = csr.col.numel()
pos = st.csr_pos(dev)                      # pos[i] = CSR position of edge i
w = torch.rand(E, device=dev)
out0, out1 = torch.empty(E, device=dev), torch.zeros(E, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for parts in (8, 16, 32):
    pass
eighth = (pos.long() * 8 // E).int()
order = torch.sort(eighth.long() * E + torch.arange(E, device=dev), stable=True).indices   # by (eighth, edge id)
counts = torch.bincount(eighth.long(), minlength=8)
padded = ((counts + 3) // 4 * 4)
seg_ptr = torch.zeros(9, dtype=torch.int64, device=dev)
seg_ptr[1:] = torch.cumsum(padded, 0)
pi = torch.empty(int(seg_ptr[-1]), dtype=torch.int32, device=dev)
pp = torch.empty_like(pi)
# pad every segment to a multiple of four pairs by repeating its last pair (a duplicate write of the same value)
start = 0
for x in range(8):
    c = int(counts[x]); lo = int(seg_ptr[x]); hi = int(seg_ptr[x + 1])
    ids = order[start:start + c].int()
    pi[lo:lo + c] = ids; pp[lo:lo + c] = pos[ids.long()]
    if hi > lo + c:
        pi[lo + c:hi] = ids[-1]; pp[lo + c:hi] = pos[ids[-1].long()]
    start += c
seg32 = seg_ptr.int()
max_seg = int(padded.max())


def timed(fn, k=60):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e3


f_lib = lambda: ops.gather(pos, w)
f0 = lambda: lib.perm_plain(E, pos.data_ptr(), w.data_ptr(), out0.data_ptr(), stream)
f1 = lambda: lib.perm_xcd(max_seg, seg32.data_ptr(), pi.data_ptr(), pp.data_ptr(), w.data_ptr(), out1.data_ptr(), stream)
f0(); f1(); torch.cuda.synchronize()
assert torch.equal(out0, w[pos.long()]) and torch.equal(out1, out0)
print("library gather (kgat_gather_f32)            %.1f us" % timed(f_lib))
print("probe: plain gather, 16 B of indices / lane  %.1f us" % timed(f0))
print("probe: one table eighth per XCD              %.1f us" % timed(f1))
