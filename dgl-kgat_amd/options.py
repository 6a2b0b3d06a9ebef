"""Run-time switches of the package, read ONCE (at import) from the environment into one object.

Every switch has the shipped default below; the environment variables exist so that a test, a benchmark or a
multi-rank launcher can select another arithmetic-preserving path without editing code.  Nothing else in the package
reads ``os.environ`` for behaviour (``_lib`` reads ``HIPCC`` for the build).  ``options.override(name=value, ...)``
is a context manager for tests; ``options.reload()`` re-reads the environment (a launcher that sets variables after
import).

=============================  =======  =====================================================================
environment variable           default  meaning
=============================  =======  =====================================================================
KGAT_ATT_F32_PRODUCTS          0        attention products on the fp32 MFMA instead of the bf16/fp16-piece products
KGAT_ATT_TILES32               0        fused attention at d = 64 on 32-group tiles (v_mfma_f32_32x32x16_f16; same speed)
KGAT_ATT_SCATTER_CSR           0        fused attention writes its logits in CSR order (round-2 form) instead of grouped order
KGAT_ATT_FORM                  auto     attention form: auto / fused / folded / split / one / race
KGAT_FOLD_TILE_COST            -        "tile,chunk,relation" cost triple of the fused attention's tile split
KGAT_GNN_COPY_SELF             1        the ego block of the readout written by the first layer's dense kernel
KGAT_FUSE_BI                   0        aggregation + dense part of a layer in one launch (slower; same bits)
KGAT_GNN_MUL_IN_SPMM           0        h * h_N in the aggregation's epilogue (rounds 1-3) instead of in the dense kernel
KGAT_GNN_DEFER_FINISH          1        the aggregation's second launch folded into the dense kernel
KGAT_SHARD_GRAD_ALLREDUCE      0        shard layers all-reduce grad_h on every layer (instead of reducing to owners)
KGAT_LAZY_EDGE_WEIGHTS         0        compute_attention defers its edge-id-ordered copy (lazy.py)
KGAT_EAGER_EDGE_WEIGHTS        0        ... never, even after enable_lazy_edge_weights()
KGAT_PARTITION_ROW_WEIGHT      8        per-row weight of the destination-range split
KGAT_FORCE_COLLECTIVES         0        a one-rank group still runs its collectives (RCCL on a one-GPU box)
KGAT_EXCHANGE                  allreduce  layer-output exchange: allreduce / allgather / broadcast / p2p
KGAT_EXCHANGE_CHUNKS           1        row blocks per layer whose exchange overlaps the next block's compute
=============================  =======  =====================================================================
"""
import contextlib
import os


def _flag(name, default):
    v = os.environ.get(name)
    return default if v is None else v not in ("", "0")


class Options:
    __slots__ = ("att_f32_products", "att_tiles32", "att_scatter_csr", "att_form", "fold_tile_cost", "gnn_copy_self", "fuse_bi",
                 "gnn_mul_in_spmm", "gnn_defer_finish", "shard_grad_allreduce", "lazy_edge_weights",
                 "eager_edge_weights", "partition_row_weight", "force_collectives", "exchange", "exchange_chunks")

    def __init__(self):
        self.load()

    def load(self):
        e = os.environ
        self.att_f32_products = _flag("KGAT_ATT_F32_PRODUCTS", False)
        self.att_tiles32 = _flag("KGAT_ATT_TILES32", False)
        self.att_scatter_csr = _flag("KGAT_ATT_SCATTER_CSR", False)
        self.att_form = e.get("KGAT_ATT_FORM", "auto")
        cost = e.get("KGAT_FOLD_TILE_COST")
        self.fold_tile_cost = tuple(int(x) for x in cost.split(",")) if cost else None
        self.gnn_copy_self = _flag("KGAT_GNN_COPY_SELF", True)
        self.fuse_bi = _flag("KGAT_FUSE_BI", False)
        self.gnn_mul_in_spmm = _flag("KGAT_GNN_MUL_IN_SPMM", False)
        self.gnn_defer_finish = _flag("KGAT_GNN_DEFER_FINISH", True)
        self.shard_grad_allreduce = _flag("KGAT_SHARD_GRAD_ALLREDUCE", False)
        self.lazy_edge_weights = _flag("KGAT_LAZY_EDGE_WEIGHTS", False)
        self.eager_edge_weights = bool(e.get("KGAT_EAGER_EDGE_WEIGHTS"))
        rw = e.get("KGAT_PARTITION_ROW_WEIGHT")
        self.partition_row_weight = int(rw) if rw else None
        self.force_collectives = _flag("KGAT_FORCE_COLLECTIVES", False)
        self.exchange = e.get("KGAT_EXCHANGE", "allreduce")
        self.exchange_chunks = int(e.get("KGAT_EXCHANGE_CHUNKS", "1"))


options = Options()


def reload():
    """Re-read the environment (for launchers that set variables after the package was imported)."""
    options.load()
    return options


@contextlib.contextmanager
def override(**kw):
    """Temporarily set switches by attribute name (tests): ``with options.override(fuse_bi=True): ...``"""
    old = {k: getattr(options, k) for k in kw}
    try:
        for k, v in kw.items():
            setattr(options, k, v)
        yield options
    finally:
        for k, v in old.items():
            setattr(options, k, v)
