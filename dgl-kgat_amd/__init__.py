"""dgl-kgat_amd: KGAT's attentive embedding-propagation layer, MI355X-native.

The package holds only what the hot path needs: ``csrc/`` (HIP kernels + the C ABI of
``include/kgat_hip.h``), the ctypes binding (``_lib``, ``ops``) and the host-side mirror of the
reference's operator surface (``DGLGraph``, ``function``, ``edge_softmax``, ``KGATConv``).
Import name: ``dgl_kgat_amd`` (a shim at the repository root maps it to this directory).
"""
from . import function  # noqa: F401
from .graph import ALL, DGLError, DGLGraph  # noqa: F401
from .softmax import edge_softmax  # noqa: F401
from .kgat_layer import KGATConv, KGATPropagation  # noqa: F401
from .compat import accelerate, install_as_dgl  # noqa: F401
from .ckg_io import CKGDataset  # noqa: F401
from ._lib import KGATLibraryError  # noqa: F401
from .lazy import enable as enable_lazy_edge_weights  # noqa: F401
from .partition import GraphedForward  # noqa: F401
from .optim import FusedAdam  # noqa: F401

__all__ = ["DGLGraph", "DGLError", "ALL", "function", "edge_softmax", "KGATConv", "KGATPropagation",
           "install_as_dgl", "accelerate", "KGATLibraryError", "CKGDataset", "enable_lazy_edge_weights", "GraphedForward",
           "FusedAdam"]
