"""Loader for libkgat_hip.so - the C-ABI library (include/kgat_hip.h) that holds every HIP
kernel of the path.  There is no fallback: if the library is missing, does not load, or
lacks a declared symbol, importing the op layer fails loudly."""
import ctypes as C
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
    "kgat_softmax.hip": [],
    "kgat_att.hip": [],
    "kgat_att_persistent.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "kgat_dense.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "kgat_transr.hip": [],
}
HEADER = os.path.join(os.path.dirname(_HERE), "include", "kgat_hip.h")

_i64, _i32, _u32, _sz, _p = C.c_int64, C.c_int, C.c_uint, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); must list every function include/kgat_hip.h declares
SIGNATURES = {
    "kgat_version": (_i32, []),
    "kgat_last_error": (C.c_char_p, []),
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
                                         _p, _p, _p, _p, _p]),
    "kgat_fold_tiles_max": (_i64, [_i64, _i64, _i32, _i32]),
    "kgat_fold_tiles_workspace_bytes": (_sz, [_i64, _i32]),
    "kgat_fold_tiles": (_i32, [_i64, _i32, _i64, _p, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "kgat_att_score_fused_supported": (_i32, [_i64, _i32, _i32, _i32]),
    "kgat_att_score_fused_f32": (_i32, [_i64, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                        _p, _p, _p, _p, _p]),
    "kgat_edge_softmax_workspace_bytes": (_sz, [_i64, _i64]),
    "kgat_edge_softmax_f32": (_i32, [_i64, _i64, _i64, _p, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "kgat_edge_softmax_3pass_workspace_bytes": (_sz, [_i64]),
    "kgat_edge_softmax_3pass_f32": (_i32, [_i64, _i64, _i64, _p, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "kgat_edge_softmax_bwd_f32": (_i32, [_i64, _i64, _p, _p, _p, _p, _p, _p]),
    "kgat_spmm_workspace_bytes": (_sz, [_i64, _i32]),
    "kgat_spmm_umule_sum_f32": (_i32, [_i64, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p,
                                       _p, _sz, _u32, _i32, _p]),
    "kgat_bi_interaction_supported": (_i32, [_i32, _i32]),
    "kgat_bi_interaction_train_f32": (_i32, [_i64, _i32, _i32, _p, _p, _p, C.c_float, C.c_float, C.c_uint64, _p, _p,
                                             _i64, _p]),
    "kgat_bi_interaction_bwd_pre_f32": (_i32, [_i64, _i32, _p, _p, _p, _p, _i64, C.c_float, C.c_float, C.c_uint64, _p,
                                               _p]),
    "kgat_mul2_f32": (_i32, [_i64, _p, _p, _p, _p, _p, _p]),
    "kgat_transr_supported": (_i32, [_i64, _i32, _i32, _i32, _i64]),
    "kgat_transr_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "kgat_transr_loss_grad_f32": (_i32, [_i64, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, C.c_float, _p, _p,
                                         _p, _p, _p, _sz, _p]),
    "kgat_bi_interaction_f32": (_i32, [_i64, _i32, _i32, _p, _p, C.c_float, _p, _p, _i64, _p]),
    "kgat_l2_normalize_rows_f32": (_i32, [_i64, _i32, _p, _p, _i64, _p]),
    "kgat_sddmm_dot_f32": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p]),
    "kgat_gather_f32": (_i32, [_i64, _p, _p, _p, _p]),
    "kgat_gather_i32": (_i32, [_i64, _p, _p, _p, _p]),
}

_lib = None


class KGATLibraryError(RuntimeError):
    pass


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
OBJ_DIR = os.path.join(_HERE, "build")


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [HEADER, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > os.path.getmtime(SO_PATH) for d in deps)


def build(force=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU): one
    object per source (per-file flags), then one shared library."""
    if not (force or needs_build()):
        return SO_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    procs, objs = [], []
    for src, extra in SOURCES.items():
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([_hipcc()] + BASE_FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]))
    for p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, p.args)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH] + objs)
    return SO_PATH


def load():
    """dlopen libkgat_hip.so and bind every declared symbol.  `import torch` must have
    happened first so that libamdhip64 resolves to the runtime torch already loaded (one HIP
    runtime per process - streams and pointers are shared with torch)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64.so first)
    if not os.path.exists(SO_PATH):
        try:  # a source checkout without the built library: compile it (hipcc cross-compiles)
            build()
        except (OSError, subprocess.CalledProcessError):
            pass
    if not os.path.exists(SO_PATH):
        raise KGATLibraryError(
            "libkgat_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
            "g.build()'`; there is no CPU or PyTorch fallback for this path." % SO_PATH)
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
    if lib.kgat_version() != 1:
        raise KGATLibraryError("libkgat_hip.so ABI version %d != 1" % lib.kgat_version())
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().kgat_last_error()
        raise KGATLibraryError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
