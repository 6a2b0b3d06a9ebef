"""Import shim: the package directory is named ``dgl-kgat_amd`` (not a valid Python
identifier), so ``import dgl_kgat_amd`` resolves here and this module turns itself into
that package (its ``__path__`` points at the directory, its namespace is the package's
``__init__``)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "dgl-kgat_amd")]
__package__ = __name__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
