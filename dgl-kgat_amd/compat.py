"""Make this package answer to the module names the reference imports
(``import dgl``, ``import dgl.function as fn``, ``from dgl.nn.pytorch.softmax import
edge_softmax``, ``from dgl.nn.pytorch.conv import SAGEConv`` - models.py:4-6, dataset.py:3),
so reference-style model code runs over the HIP kernels without edits."""
import sys
import types


def install_as_dgl(force=False):
    if "dgl" in sys.modules and not force and not getattr(sys.modules["dgl"], "_kgat_amd", False):
        raise RuntimeError("a real `dgl` is already imported; pass force=True to shadow it")
    import dgl_kgat_amd as pkg
    from . import function, graph, softmax

    dgl = types.ModuleType("dgl")
    dgl._kgat_amd = True
    dgl.DGLGraph, dgl.DGLError, dgl.ALL = graph.DGLGraph, graph.DGLError, graph.ALL
    dgl.function = function
    nn = types.ModuleType("dgl.nn")
    pt = types.ModuleType("dgl.nn.pytorch")
    conv = types.ModuleType("dgl.nn.pytorch.conv")

    class SAGEConv:  # models.py:6 imports it; the graphsage branch is outside the KGAT path
        def __init__(self, *a, **k):
            raise NotImplementedError("SAGEConv (--gnn_model graphsage) is outside the KGAT path")

    conv.SAGEConv = SAGEConv
    nn.pytorch, pt.softmax, pt.conv = pt, softmax, conv
    dgl.nn = nn
    mods = {"dgl": dgl, "dgl.function": function, "dgl.nn": nn, "dgl.nn.pytorch": pt,
            "dgl.nn.pytorch.softmax": softmax, "dgl.nn.pytorch.conv": conv}
    sys.modules.update(mods)
    return pkg


def accelerate(model, lazy_edge_weights=None):
    """Route a reference-shaped model's two hot-path methods to the fused kernels, in place.

    `model` is an instance of the reference's ``models.Model`` (unmodified; constructed with
    ``gnn_model="kgat"``, ``use_KG=True``) or anything with the same attribute layout:
    ``entity_embed``, ``relation_embed``, ``W_R``, ``layers[i].res_fc_2`` / ``.mess_drop``.  After
    the call

    * ``model.compute_attention(g)`` (models.py:146-154) is one fused attention-logit launch + the
      destination softmax (``DGLGraph.kgat_attention``) instead of R rounds of ``filter_edges`` /
      ``apply_edges`` through Python (72 ms -> 0.26 ms on the amazon-book-shaped CKG), and
    * ``model.gnn(g, x)`` (models.py:156-168) is the aggregation with the ``h * h_N`` epilogue +
      the bi-interaction kernel per layer (the whole stack as one autograd unit when gradients
      are enabled)

    with the same parameters (shared, not copied), the same call signatures and the same results.
    This is the one line a maintainer adds after ``model = Model(...)`` in kgat.py:95-98; the
    reference's files stay as they are.  Returns the model.

    ``lazy_edge_weights=True`` also opts the process into the deferred edge-id-ordered attention tensor
    (``lazy.enable()``: ``compute_attention`` returns a tensor whose values are written on first read, the
    aggregation is served from the CSR-ordered copy; 6 % of a step on the benchmark graph); ``False`` turns it
    off; ``None`` (default) leaves the process setting as it is."""
    import types

    from .kgat_layer import KGATPropagation

    if lazy_edge_weights is not None:
        from . import lazy
        lazy.enable(bool(lazy_edge_weights))
    for attr in ("entity_embed", "relation_embed", "W_R", "layers"):
        if not hasattr(model, attr):
            raise TypeError("accelerate(): the model has no attribute %r (not a KGAT Model)" % attr)
    for layer in model.layers:
        if not (hasattr(layer, "res_fc_2") and hasattr(layer, "mess_drop")):
            raise TypeError("accelerate(): layer %s is not a bi-interaction KGATConv (gnn_model='kgat', res_type='Bi')"
                            % type(layer).__name__)
    if getattr(model, "_use_KG", True) is False:
        raise TypeError("accelerate(): use_KG=False models build their input from item/user projections; outside the path")
    if not hasattr(model, "_n_entities") or model._n_entities is None:
        model._n_entities = model.entity_embed.weight.shape[0]

    def gnn(self, g, x=None, fused=None):
        return KGATPropagation.gnn(self, g, x, fused)

    def compute_attention(self, g, algo="auto"):
        return KGATPropagation.compute_attention(self, g, algo)

    for name in ("_can_fuse_readout", "_can_fuse_training", "_gnn_fused", "_gnn_fused_sharded", "_gnn_train_sharded",
                 "_node_embeddings"):
        setattr(model, name, types.MethodType(getattr(KGATPropagation, name), model))
    model.gnn = types.MethodType(gnn, model)
    model.compute_attention = types.MethodType(compute_attention, model)
    model._kgat_accelerated = True
    return model
