"""Developer probe (round 6): what bounds the KG iteration's Adam pass (42.5 us for 245 MB = 5.8 TB/s)?  The tables
(p, m, v = 122 MB) are smaller than the 256 MiB Infinity Cache, but a cyclic sweep that loads and stores 245 MB per
iteration leaves an LRU-like cache nothing to hit.  Streams marked non-temporal (policy bits 1 = p, 2 = m, 4 = v) and
table sizes from 1/8 to 2 x the amazon-book table; back-to-back launches, as the KG phase issues them."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
import numpy as np, torch
so, src = os.path.join(HERE, "build", "adam_policy_probe.so"), os.path.join(HERE, "adam_policy_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, src])
lib = ctypes.CDLL(so)
lib.adam_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
other = torch.empty(6 * 1024 * 1024 // 4, device=dev)   # ~6 MB of other traffic between two passes (the step's own buffers)


def run(n_rows, policy, shape, reps=100, between=True):
    n = n_rows * 64
    p, m, v = (torch.randn(n, device=dev) for _ in range(3))
    v.abs_()
    for _ in range(10):
        lib.adam_probe_launch(policy, shape, p.data_ptr(), m.data_ptr(), v.data_ptr(), n, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        if between:
            other.add_(1.0)
        a.record()
        lib.adam_probe_launch(policy, shape, p.data_ptr(), m.data_ptr(), v.data_ptr(), n, st)
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


N = 159251
print("table rows x 64 floats; time in us per pass (median of 100), GB/s = 24 B/element / time")
for rows in (N // 8, N // 4, N // 2, N, 2 * N):
    line = "rows %7d (%6.1f MB of p+m+v):" % (rows, rows * 64 * 12 / 1e6)
    for pol in (0, 4, 6, 7, 1):
        t = run(rows, pol, 0)
        line += "  pol%d %6.1f (%5.2f TB/s)" % (pol, t, rows * 64 * 24 / t / 1e6)
    print(line)
print("load shape at the full table:")
for pol in (0, 4, 6):
    print("  policy %d: loop %.1f us, loads-first %.1f us" % (pol, run(N, pol, 0), run(N, pol, 1)))
