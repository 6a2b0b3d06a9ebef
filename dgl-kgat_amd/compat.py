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
