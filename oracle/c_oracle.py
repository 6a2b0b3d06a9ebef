"""ctypes loader for oracle/libkgat_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkgat_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "kgat_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.kgat_oracle_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def threads():
    return int(lib().kgat_oracle_threads())


def set_threads(n):
    lib().kgat_oracle_set_threads(C.c_int(int(n)))


def csr_from_coo(n, src, dst):
    src, dst = _i32(src), _i32(dst)
    e = src.shape[0]
    indptr = np.empty(n + 1, np.int32)
    col = np.empty(e, np.int32)
    eid = np.empty(e, np.int32)
    rc = lib().kgat_oracle_csr_from_coo(C.c_int64(n), C.c_int64(e), _p(src), _p(dst),
                                        _p(indptr), _p(col), _p(eid))
    if rc != 0:
        raise ValueError("kgat_oracle_csr_from_coo failed: %d" % rc)
    return indptr, col, eid


def group_by_relation(etype, n_rel):
    etype = _i32(etype)
    e = etype.shape[0]
    rel_ptr = np.empty(n_rel + 1, np.int32)
    perm = np.empty(e, np.int32)
    rc = lib().kgat_oracle_group_by_relation(C.c_int64(e), C.c_int(n_rel), _p(etype),
                                             _p(rel_ptr), _p(perm))
    if rc != 0:
        raise ValueError("kgat_oracle_group_by_relation failed: %d" % rc)
    return rel_ptr, perm


def att_score(ent, W_R, rel, src, dst, etype):
    ent, W_R, rel = _f32(ent), _f32(W_R), _f32(rel)
    src, dst, etype = _i32(src), _i32(dst), _i32(etype)
    e = src.shape[0]
    out = np.zeros(e, np.float32)
    lib().kgat_oracle_att_score_f32(C.c_int64(e), C.c_int(ent.shape[1]), C.c_int(W_R.shape[2]),
                                    C.c_int(W_R.shape[0]), _p(src), _p(dst), _p(etype), _p(ent),
                                    _p(W_R), _p(rel), _p(out))
    return out


def att_score_fast(ent, W_R, rel, src, dst, etype):
    """kgat_oracle_att_score_fast_f32: the vectorised twin of att_score (bench.py's timed leg)."""
    ent, W_R, rel = _f32(ent), _f32(W_R), _f32(rel)
    src, dst, etype = _i32(src), _i32(dst), _i32(etype)
    e = src.shape[0]
    out = np.zeros(e, np.float32)
    lib().kgat_oracle_att_score_fast_f32(C.c_int64(e), C.c_int(ent.shape[1]), C.c_int(W_R.shape[2]),
                                         C.c_int(W_R.shape[0]), _p(src), _p(dst), _p(etype), _p(ent),
                                         _p(W_R), _p(rel), _p(out))
    return out


def edge_softmax(n, indptr, eid, logits):
    logits = _f32(logits).reshape(-1)
    out = np.zeros_like(logits)
    lib().kgat_oracle_edge_softmax_f32(C.c_int64(n), _p(_i32(indptr)), _p(_i32(eid)), _p(logits),
                                       _p(out))
    return out


def spmm(n, indptr, col, eid, X, w, mul_self=False):
    X, w = _f32(X), _f32(w).reshape(-1)
    out = np.empty((n, X.shape[1]), np.float32)
    indptr, col = _i32(indptr), _i32(col)
    eid_p = _p(_i32(eid)) if eid is not None else None
    lib().kgat_oracle_spmm_f32(C.c_int64(n), C.c_int(X.shape[1]), _p(indptr), _p(col), eid_p,
                               _p(X), _p(w), _p(out), C.c_int(1 if mul_self else 0))
    return out


def bi_interaction(h, h_n, W2, slope=0.01):
    h, h_n, W2 = _f32(h), _f32(h_n), _f32(W2)
    out = np.empty((h.shape[0], W2.shape[0]), np.float32)
    lib().kgat_oracle_bi_interaction_f32(C.c_int64(h.shape[0]), C.c_int(h.shape[1]),
                                         C.c_int(W2.shape[0]), _p(h), _p(h_n), _p(W2),
                                         C.c_float(slope), _p(out))
    return out


def l2_normalize(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().kgat_oracle_l2_normalize_f32(C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(x), _p(out))
    return out
