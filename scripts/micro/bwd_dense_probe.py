import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dgl_kgat_amd import ops
from dgl_kgat_amd.autograd import tall_weight_grad
dev = torch.device('cuda:0'); n = 159251
def ev(fn, k=60):
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); out.append((a, b))
    torch.cuda.synchronize()
    return 1e3 * float(np.median([a.elapsed_time(b) for a, b in out][10:]))
for d_in, d_out in ((64, 64), (64, 32), (32, 16)):
    gz = torch.randn(n, d_out, device=dev); H = torch.randn(n, d_in, device=dev); HN = torch.randn(n, d_in, device=dev)
    W = torch.randn(d_out, d_in, device=dev)
    print(d_in, d_out, 'weight: kernel+sum %.1f us | torch mul + slab bmm %.1f us || input: fused %.1f us | mm + mul2 %.1f us' % (
        ev(lambda: ops.bi_interaction_bwd_weight(gz, H, HN)), ev(lambda: tall_weight_grad(gz, H * HN)),
        ev(lambda: ops.bi_interaction_bwd_input(gz, W, H, HN)), ev(lambda: ops.mul2(gz @ W, H, HN))))
