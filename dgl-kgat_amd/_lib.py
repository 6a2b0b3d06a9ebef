"""Loader for libkgat_hip.so - the C-ABI library (include/kgat_hip.h) that holds every HIP
kernel of the path.  There is no fallback: if the library is missing, does not load, or
lacks a declared symbol, importing the op layer fails loudly."""
import ctypes as C
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "libkgat_hip.so")
# source -> extra hipcc flags.  -amdgpu-mfma-vgpr-form puts MFMA results in VGPRs so the VALU
# epilogue reads them without v_accvgpr_read copies (f32 MFMA and VALU share the SIMD's issue
# port - measured with scripts/micro/mfma_rate.hip - so fewer instructions is the lever); the
# LDS-staged chunk kernels lose occupancy with it and are built without.
SOURCES = {
    "kgat_graph.hip": [],
    "kgat_spmm.hip": [],
    "kgat_spmm_bi.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "kgat_softmax.hip": [],
    "kgat_att.hip": [],
    "kgat_att_persistent.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "kgat_dense.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "kgat_transr.hip": [],
    "kgat_eval.hip": [],
    "kgat_optim.hip": [],
    "kgat_bpr.hip": [],
}
HEADER = os.path.join(os.path.dirname(_HERE), "include", "kgat_hip.h")

_i64, _i32, _u32, _sz, _p = C.c_int64, C.c_int, C.c_uint, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); must list every function include/kgat_hip.h declares
SIGNATURES = {
    "kgat_version": (_i32, []),
    "kgat_last_error": (C.c_char_p, []),
    "kgat_build_hash": (C.c_char_p, []),
    "kgat_csr_from_coo_workspace_bytes": (_sz, [_i64, _i64]),
    "kgat_csr_from_coo": (_i32, [_i64, _i64, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "kgat_group_by_relation_workspace_bytes": (_sz, [_i64, _i32]),
    "kgat_group_by_relation": (_i32, [_i64, _i32, _p, _p, _p, _p, _sz, _p]),
    "kgat_invert_permutation": (_i32, [_i64, _p, _p, _p]),
    "kgat_row_order_workspace_bytes": (_sz, [_i64]),
    "kgat_row_order_by_degree": (_i32, [_i64, _p, _p, _p, _sz, _p]),
    "kgat_att_score_f32": (_i32, [_i64, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                  _p, _i32, _p]),
    "kgat_head_groups_workspace_bytes": (_sz, [_i64]),
    "kgat_head_groups": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "kgat_att_score_split_supported": (_i32, [_i64, _i32, _i32, _i32]),
    "kgat_att_score_split_f32": (_i32, [_i64, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p,
                                        _p, _p, _p, _p, _p]),
    "kgat_att_score_folded_supported": (_i32, [_i64, _i32, _i32, _i32]),
    "kgat_att_score_folded_f32": (_i32, [_i64, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p,
                                         _p, _p, _p, _p, _i32, _p]),
    "kgat_fold_tiles_max": (_i64, [_i64, _i64, _i32, _i32]),
    "kgat_fold_tiles_workspace_bytes": (_sz, [_i64, _i32]),
    "kgat_fold_tiles": (_i32, [_i64, _i32, _i64, _p, _p, _p, _i32, _i32, _p, _p, _p, _sz, _p]),
    "kgat_att_score_fused_supported": (_i32, [_i64, _i32, _i32, _i32]),
    "kgat_fold_tile_parts_workspace_bytes": (_sz, [_i64]),
    "kgat_fold_tile_parts": (_i32, [_i64, _i32, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "kgat_att_pack_records": (_i32, [_i64, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    "kgat_att_score_fused_f32": (_i32, [_i64, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32,
                                        _p, _p, _p, _p, _p, _p, _i32, _p]),
    "kgat_att_score_fused_timed_f32": (_i32, [_i64, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32,
                                              _p, _p, _p, _p, _p, _p, _i32, _p, _p]),
    "kgat_edge_softmax_workspace_bytes": (_sz, [_i64, _i64]),
    "kgat_edge_softmax_f32": (_i32, [_i64, _i64, _i64, _p, _p, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "kgat_edge_softmax_3pass_workspace_bytes": (_sz, [_i64]),
    "kgat_edge_softmax_3pass_f32": (_i32, [_i64, _i64, _i64, _p, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "kgat_edge_softmax_bwd_f32": (_i32, [_i64, _i64, _p, _p, _p, _p, _p, _p]),
    "kgat_spmm_workspace_bytes": (_sz, [_i64, _i32]),
    "kgat_spmm_umule_sum_f32": (_i32, [_i64, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p,
                                       _p, _sz, _u32, _i32, _p, _i64, _p]),
    "kgat_bi_interaction_supported": (_i32, [_i32, _i32]),
    "kgat_spmm_bi_fused_supported": (_i32, [_i32, _i32]),
    "kgat_spmm_bi_fused_f32": (_i32, [_i64, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _p, _p, C.c_float, _p, _p,
                                      _i64, _p, _p, _sz, _p, _i64, _p]),
    "kgat_bi_interaction_train_f32": (_i32, [_i64, _i32, _i32, _p, _p, _p, C.c_float, C.c_float, C.c_uint64, _i64, _p,
                                             _p, _i64, _p, _i64, _p]),
    "kgat_add3_rows_f32": (_i32, [_i64, _i32, _p, _i64, _p, _p, _p, _p]),
    "kgat_sum_partials_f32": (_i32, [_i32, _p, _p, _p, _p, _p]),
    "kgat_bi_interaction_bwd_pre_f32": (_i32, [_i64, _i32, _p, _p, _p, _p, _i64, C.c_float, C.c_float, C.c_uint64, _i64,
                                               _p, _p]),
    "kgat_mul2_f32": (_i32, [_i64, _p, _p, _p, _p, _p, _p]),
    "kgat_bi_interaction_bwd_input_supported": (_i32, [_i32, _i32]),
    "kgat_bi_interaction_bwd_input_f32": (_i32, [_i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "kgat_bi_interaction_bwd_weight_partials": (_i64, [_i64]),
    "kgat_bi_interaction_bwd_weight_f32": (_i32, [_i64, _i32, _i32, _p, _p, _p, _p, _i64, _p]),
    "kgat_transr_supported": (_i32, [_i64, _i32, _i32, _i32, _i64]),
    "kgat_transr_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "kgat_transr_loss_grad_f32": (_i32, [_i64, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, C.c_float, _p, _p,
                                         _p, _p, _p, _sz, _p]),
    "kgat_transr_forward_f32": (_i32, [_i64, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, C.c_float, _p, _p, _sz, _p]),
    "kgat_transr_backward_f32": (_i32, [_i64, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "kgat_bi_interaction_f32": (_i32, [_i64, _i32, _i32, _p, _p, C.c_float, _p, _p, _i64, _p]),
    "kgat_bi_interaction_mul_f32": (_i32, [_i64, _i32, _i32, _p, _p, _p, C.c_float, _p, _p, _i64, _p, _i64, _p]),
    "kgat_bi_interaction_mul_deferred_f32": (_i32, [_i64, _i32, _i32, _p, _p, _p, C.c_float, _p, _p, _i64, _p, _i64,
                                                    _p, _i64, _i64, _p, _i32, _p]),
    "kgat_spmm_tile_edges": (_i32, [_i64, _i32]),
    "kgat_l2_normalize_rows_f32": (_i32, [_i64, _i32, _p, _p, _i64, _p]),
    "kgat_readout_concat_f32": (_i32, [_i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "kgat_sddmm_dot_f32": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p]),
    "kgat_gather_probe_f32": (_i32, [_i64, _i32, _p, _p, _p, _p]),
    "kgat_gather_f32": (_i32, [_i64, _p, _p, _p, _p]),
    "kgat_gather_i32": (_i32, [_i64, _p, _p, _p, _p]),
    "kgat_eval_supported": (_i32, [_i32, _i32]),
    "kgat_eval_items_elems": (_i64, [_i64, _i32]),
    "kgat_eval_items_kmajor_f32": (_i32, [_i64, _i32, _p, _i64, _p, _p, _p]),
    "kgat_eval_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "kgat_bpr_workspace_bytes": (_sz, [_i64, _i32]),
    "kgat_bpr_loss_f32": (_i32, [_i64, _i32, _p, _i64, _i64, _p, _p, _p, C.c_float, _p, _p, _p, _sz, _p]),
    "kgat_bpr_grad_f32": (_i32, [_i64, _i32, _p, _i64, _i64, _p, _p, _p, _p, C.c_float, _p, _p, _p, _sz, _p]),
    "kgat_transr_sorted_bytes": (_sz, [_i64, _i32]),
    "kgat_transr_presort_f32": (_i32, [_i64, _i32, _i64, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "kgat_transr_step_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "kgat_transr_adam_step_f32": (_i32, [_i64, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                         C.c_double, C.c_double, C.c_double, C.c_double, C.c_float, _p, _p, C.c_uint64,
                                         _p, _sz, _p]),
    "kgat_adam_max_tensors": (_i32, []),
    "kgat_adam_step_f32": (_i32, [_i32, _p, _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_double, C.c_double, _i32, _p]),
    "kgat_eval_recall_ndcg_f32": (_i32, [_i64, _p, _i64, _i32, _p, _i64, _p, _p, _p, _p, _p, _i32, _p, _p, _sz, _p,
                                         _p, _p, _p]),
}

_lib = None


class KGATLibraryError(RuntimeError):
    pass


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
OBJ_DIR = os.path.join(_HERE, "build")


ABI_VERSION = 10


def source_hash():
    """sha256[:16] over everything the library is built from: csrc/*, the public header, the
    compiler flags.  Embedded in the library at build time (kgat_build_hash) and compared at load
    time - file times do not survive a copy to another machine, content does."""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    with open(HEADER, "rb") as fh:
        h.update(fh.read())
    h.update(repr((BASE_FLAGS, sorted(SOURCES.items()))).encode())
    return h.hexdigest()[:16]


def built_hash():
    """The hash embedded in the library on disk (read from the file, no dlopen), or None."""
    try:
        with open(SO_PATH, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    tag = b"kgat-src-hash:"
    i = blob.find(tag)
    return blob[i + len(tag):i + len(tag) + 16].decode("ascii", "replace") if i >= 0 else None


def needs_build():
    return built_hash() != source_hash()


def _unit_hash(src, flags):
    """Key of one object file: its source, every header it may include, its flags."""
    h = hashlib.sha256()
    for f in [src] + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(HEADER, "rb") as fh:
        h.update(fh.read())
    h.update(repr(flags).encode())
    return h.hexdigest()


def build(force=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU): one
    object per source (per-file flags; an object whose inputs did not change is kept), then one
    shared library.  The source hash is compiled into kgat_graph.hip (kgat_build_hash).

    Safe under a multi-rank launch after a source edit: the build holds an exclusive lock on a
    file in the object directory (the other ranks wait, then find the library fresh), and objects
    and the library are written under temporary names and moved into place atomically, so no
    process can dlopen a half-written file."""
    if not (force or needs_build()):
        return SO_PATH
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not (force or needs_build()):  # another process built it while this one waited
                return SO_PATH
            return _build_locked(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force):
    procs, objs = [], []
    tag = '-DKGAT_BUILD_HASH="kgat-src-hash:%s"' % source_hash()
    for src, extra in SOURCES.items():
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        flags = BASE_FLAGS + extra + ([tag] if src == "kgat_graph.hip" else [])
        key = _unit_hash(src, flags)
        try:
            with open(obj + ".key") as fh:
                fresh = os.path.exists(obj) and fh.read() == key
        except OSError:
            fresh = False
        if fresh and not force:
            continue
        tmp = "%s.%d.tmp" % (obj, os.getpid())
        procs.append((subprocess.Popen([_hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", tmp]), obj, tmp, key))
    failed = None
    for p, obj, tmp, key in procs:
        if p.wait() != 0:
            failed = failed or subprocess.CalledProcessError(p.returncode, p.args)
            continue
        os.replace(tmp, obj)
        with open(obj + ".key.tmp", "w") as fh:
            fh.write(key)
        os.replace(obj + ".key.tmp", obj + ".key")
    if failed is not None:
        raise failed
    so_tmp = "%s.%d.tmp" % (SO_PATH, os.getpid())
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so_tmp] + objs)
    os.replace(so_tmp, SO_PATH)
    return SO_PATH


def _under_profiler():
    """rocprofv3 / rocprof run their target with a tool library preloaded (and ROCPROF* / ROCP_* variables set)."""
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre.lower() or any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ)


def load():
    """dlopen libkgat_hip.so and bind every declared symbol.  `import torch` must have
    happened first so that libamdhip64 resolves to the runtime torch already loaded (one HIP
    runtime per process - streams and pointers are shared with torch).  A library that is
    missing or was built from other sources than the ones beside it is rebuilt (hipcc
    cross-compiles); if that fails, loading fails - there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64.so first)
    if needs_build():
        if _under_profiler():
            # a rebuild would spawn hipcc from inside the profiled process: its children inherit the profiler's
            # preload, which initialises the GPU before hipcc execs clang - the launcher hop this pool forbids
            raise KGATLibraryError(
                "libkgat_hip.so (%s) is missing or stale and this process runs under a profiler preload: build first "
                "(`python -c 'import __graft_entry__ as g; g.build()'`), then profile" % SO_PATH)
        try:
            build()
        except (OSError, subprocess.CalledProcessError) as e:
            raise KGATLibraryError(
                "libkgat_hip.so (%s) is %s and rebuilding it failed (%s). Run `python -c 'import "
                "__graft_entry__ as g; g.build()'`; there is no CPU or PyTorch fallback for this path."
                % (SO_PATH, "missing" if not os.path.exists(SO_PATH) else "stale (built from other sources)", e)) from e
    try:
        lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise KGATLibraryError("cannot load %s: %s" % (SO_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise KGATLibraryError("libkgat_hip.so lacks symbol %s" % name) from e
        fn.restype, fn.argtypes = res, args
    if lib.kgat_version() != ABI_VERSION:
        raise KGATLibraryError("libkgat_hip.so ABI version %d != %d" % (lib.kgat_version(), ABI_VERSION))
    got = lib.kgat_build_hash().decode()
    if got != "kgat-src-hash:" + source_hash():
        raise KGATLibraryError("libkgat_hip.so was built from other sources (%s) than the ones beside it (%s)"
                               % (got, source_hash()))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().kgat_last_error()
        raise KGATLibraryError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
