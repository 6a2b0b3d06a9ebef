import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["toy_d8", "toy_d16_k32", "toy_d64"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    trip = g["triplets"]
    # reference dataset.py:116: add_edges(triplet[:,2], triplet[:,0]) -> src = t, dst = h
    g["src"], g["dst"], g["etype"] = trip[:, 2].copy(), trip[:, 0].copy(), trip[:, 1].copy()
    g["n"] = int(g["n_nodes"])
    g["R"] = int(g["n_rel"])
    g["W2"] = [g[k] for k in sorted(k for k in g if k.startswith("W2_"))]
    return g


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return load_golden(request.param)


def rel_err(x, y):
    """SURVEY 8c tolerance form: max|x-y| / max(|y|, 1e-3*||y||inf), per tensor."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    scale = np.maximum(np.abs(y), 1e-3 * max(np.abs(y).max(), 1e-30))
    return float(np.max(np.abs(x - y) / scale)) if y.size else 0.0


def rel_err_inf(x, y):
    """max|x-y| / max|y| per tensor: the "<= 1e-4 relative fp32" bar of BASELINE.json's
    north_star for multi-layer outputs (rounding accumulated over three fp32 layers sits at a
    few 1e-7 of the tensor's scale, so near-zero elements cannot meet an elementwise bound)."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    return float(np.max(np.abs(x - y)) / max(np.abs(y).max(), 1e-30)) if y.size else 0.0


def blocks_rel_err_inf(x, y, widths):
    """rel_err_inf per concatenated block (Model.gnn output = [h0 | n(h1) | n(h2) | ...])."""
    errs, o = [], 0
    for w in widths:
        errs.append(rel_err_inf(x[:, o:o + w], y[:, o:o + w]))
        o += w
    assert o == x.shape[1]
    return max(errs)


def rel_err_rows(x, y):
    """Row-wise bound for the aggregation: max|x-y| in a row relative to that row's largest
    |y| (a row is one destination's sum; a hub row adds thousands of fp32 terms, so its
    near-cancelling columns carry the rounding of the row's scale).  Zero rows must be exact."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    if y.size == 0:
        return 0.0
    scale = np.abs(y).max(axis=1)
    err = np.abs(x - y).max(axis=1)
    if np.any(err[scale == 0] != 0):
        return float("inf")
    nz = scale > 0
    return float(np.max(err[nz] / scale[nz])) if nz.any() else 0.0


def sum_err(x, y, abs_sum):
    """Forward-error metric for a sum of products: max |x - y| / sum_p |w_p * X_p| per output
    element (`abs_sum` = the same aggregation on |w|, |X|).  This is the quantity fp32
    summation bounds (<= n*eps in the worst case, ~sqrt(n)*eps typically); an element whose
    terms cancel cannot be held to a tolerance relative to its own (tiny) value.  Elements
    with no terms must be exactly zero."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    a = np.asarray(abs_sum, np.float64)
    if y.size == 0:
        return 0.0
    if np.any(x[a == 0] != 0):
        return float("inf")
    nz = a > 0
    return float(np.max(np.abs(x - y)[nz] / a[nz])) if nz.any() else 0.0


def parity_8c(name, gpu, c32, ref64, tol=1e-4, factor=2.0):
    """The SURVEY 8c metric, measured instead of argued: ``rel_err`` (elementwise, floor 1e-3 of
    the tensor's max) of the device result AND of the C/OpenMP fp32 oracle, both against the fp64
    oracle.  The device passes when it is within `tol`, or no further from fp64 than `factor` x what
    a CPU fp32 forward shows under the same metric ("matches the DGL-CPU forward": fp32 sums of
    thousands of terms cannot meet an elementwise 1e-4 on elements that nearly cancel, on any
    machine).  Prints both numbers; returns them."""
    e_gpu, e_c = rel_err(gpu, ref64), rel_err(c32, ref64)
    print("[8c] %-34s gpu %.3e   c-fp32 %.3e   (rel_err vs fp64; bar max(%.0e, %gx c-fp32))"
          % (name, e_gpu, e_c, tol, factor))
    assert e_gpu <= max(tol, factor * e_c), (name, e_gpu, e_c)
    return e_gpu, e_c


def err_8c(x, y):
    """The elementwise quantity whose maximum is `rel_err`: |x - y| / max(|y|, 1e-3 * max|y|)."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    return np.abs(x - y) / np.maximum(np.abs(y), 1e-3 * max(np.abs(y).max(), 1e-30))


def parity_8c_mean(name, gpu, c32, ref64, factor=1.25, floor=1e-7):
    """The MEAN of the 8c metric, device vs CPU fp32 run.  The maximum (`parity_8c`) sits on one element of
    the row with the smallest norm and moves by tens of percent with any change of summation order on
    either machine (scripts/error_attribution.py); the mean over 10^6-10^7 elements does not, so it
    is the statistic a factor close to 1 can be asked of."""
    e_gpu, e_c = float(err_8c(gpu, ref64).mean()), float(err_8c(c32, ref64).mean())
    print("[8c mean] %-30s gpu %.3e   c-fp32 %.3e   (bar max(%.0e, %gx c-fp32))" % (name, e_gpu, e_c, floor, factor))
    assert e_gpu <= max(floor, factor * e_c), (name, e_gpu, e_c)
    return e_gpu, e_c


def robust_quantile(size):
    """The quantile the robust 8c gate looks at: the 99.9th percentile, or - on samples too small for that to be
    anything but the maximum - the level of the 50th-largest element (never below the 90th percentile)."""
    return float(min(99.9, max(90.0, 100.0 * (1.0 - 50.0 / max(size, 1)))))


def readout_abs_bar(block):
    """ABSOLUTE bar of a readout block, as a fraction of the block's largest magnitude (one bar that
    does not ride on another fp32 run's error).  Block 0 is a copy of the embedding table: exact.  Block l >= 1 is
    layer l's normalised output: l layers of fp32 sums, 1.5e-6 per layer = about a dozen fp32 ulps of the row maximum
    per layer (measured over three seeds and the golden cases: 4e-7 / 8e-7 / 1.3e-6 at l = 1 / 2 / 3,
    profiles/r04_robust_gates_three_seeds.txt).  The attention weights: 1e-6 (measured 1.9e-7 ... 2.9e-7)."""
    return 0.0 if block == 0 else 1.5e-6 * block


ATTENTION_ABS_BAR = 1e-6


def parity_8c_robust(name, gpu, c32, ref64, q_factor=1.5, mean_factor=1.25, scale_bar=1e-5, max_factor=10.0,
                     tol=1e-4, floor=1e-7, abs_bar=None):
    """SURVEY 8c with statistics that do not ride on one element (task 7).  The maximum of the
    elementwise metric over a normalised readout block sits on one element of the row with the smallest norm and
    moves 2-3 x between two fp32 summation orders on ANY pair of machines (scripts/error_attribution.py); so:
      * the high quantile (robust_quantile) of the device's 8c errors   <= q_factor    x the CPU fp32 run's,
      * their mean                                                      <= mean_factor x the CPU fp32 run's,
      * the whole tensor within scale_bar of its own scale (max |x - y| / max |y|),
      * and the maximum only as a tripwire: <= max(tol, max_factor x the CPU fp32 run's maximum)."""
    eg, ec = err_8c(gpu, ref64).reshape(-1), err_8c(c32, ref64).reshape(-1)
    q = robust_quantile(eg.size)
    qg, qc = float(np.percentile(eg, q)), float(np.percentile(ec, q))
    e_inf = rel_err_inf(gpu, ref64)
    print("[8c robust] %-34s q%.2f gpu %.3e c-fp32 %.3e | mean gpu %.3e c-fp32 %.3e | max gpu %.3e c-fp32 %.3e | at tensor "
          "scale %.2e" % (name, q, qg, qc, eg.mean(), ec.mean(), eg.max(), ec.max(), e_inf))
    assert qg <= max(floor, q_factor * qc), (name, "quantile", q, qg, qc)
    assert eg.mean() <= max(floor, mean_factor * ec.mean()), (name, "mean", eg.mean(), ec.mean())
    assert e_inf <= scale_bar, (name, "tensor scale", e_inf)
    if abs_bar is not None:   # max |x - y| <= abs_bar * max |y|: no reference to the CPU fp32 run
        assert e_inf <= abs_bar, (name, "absolute bar", e_inf, abs_bar)
    assert eg.max() <= max(tol, max_factor * ec.max()), (name, "max tripwire", eg.max(), ec.max())
    return qg, qc


def blocks(x, widths):
    """Column blocks of a Model.gnn output [h0 | n(h1) | n(h2) | ...]."""
    out, o = [], 0
    for w in widths:
        out.append(x[:, o:o + w])
        o += w
    assert o == x.shape[1]
    return out
