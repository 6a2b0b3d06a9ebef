"""Edge-id-ordered attention weights whose values are produced on first use.

``compute_attention`` (reference models.py:146-154) returns the (E,1) attention in edge-id order
and the training loop stores it as ``g.edata['w']`` (kgat.py:142-144); the only reader on the
path is ``update_all(u_mul_e('h','w','m'), sum)`` (models.py:63), whose kernel streams the
weights in destination-major CSR order - the order the softmax kernel produces them in.  The
edge-id-ordered copy is a permutation of 4-byte items (a full pass through the cache fabric at
sector granularity, ~30 us on the amazon-book CKG, 5 % of a step) that nothing on the path reads.

``LazyEdgeWeights`` is that tensor with the permutation deferred: it owns real, correctly shaped
device storage from the start; the graph structure keeps the CSR-ordered values next to it and
hands them to the aggregation without touching this tensor; the first torch operation that could
observe or alias its *values* (anything but shape / dtype / device style metadata) first runs the
permutation into the storage it already owns, on the current stream, and from then on it IS an
ordinary tensor: the object is re-classed to ``torch.Tensor`` (same object, same ``data_ptr``), so
nothing lazy-specific can misbehave afterwards (``copy.deepcopy``, pickling, subclass-unaware
code).  Value-reading paths that do not go through a torch operator are covered one by one:
``Tensor.type(dtype)`` (a cast, unlike the argument-less metadata query), ``copy.deepcopy`` /
pickling, and the legacy ``torch.utils.dlpack.to_dlpack`` (a C function that skips
``__torch_function__``: wrapped by ``enable()``).  What remains out of reach is a C++ extension that
takes the raw pointer of a tensor it was handed inside a container without calling any torch
operator or ``data_ptr()`` on the Python side.

**Opt-in.**  The deferred form is OFF unless the process asks for it: ``dgl_kgat_amd.enable_lazy_edge_weights()``
(= ``lazy.enable()``), ``compat.accelerate(model, lazy_edge_weights=True)``, ``KGAT_LAZY_EDGE_WEIGHTS=1`` in the
environment, or ``lazy=True`` on one ``kgat_attention`` call.  Importing this package changes nothing in torch;
only ``enable()`` wraps ``torch.utils.dlpack.to_dlpack`` (one process-wide wrapper, undone by ``enable(False)``),
because a library that rewrites a torch global at import is a liability inside someone else's training process.
By default ``compute_attention`` therefore returns an ordinary, fully written tensor (6 % of a step on the
benchmark graph); ``KGAT_EAGER_EDGE_WEIGHTS=1`` forces that even after ``enable()``.
"""
import copy

import torch
import torch.utils.dlpack as _dlpack

_T = torch.Tensor
# operations that only look at metadata: they must not trigger the fill
_META = {
    _T.shape.__get__, _T.dtype.__get__, _T.device.__get__, _T.is_cuda.__get__, _T.ndim.__get__,
    _T.requires_grad.__get__, _T.grad_fn.__get__, _T.layout.__get__, _T._version.__get__, _T.is_leaf.__get__,
    _T.grad.__get__, _T.names.__get__, _T.is_sparse.__get__, _T.is_quantized.__get__, _T.is_meta.__get__,
    _T.dim, _T.size, _T.numel, _T.nelement, _T.ndimension, _T.stride, _T.is_contiguous, _T.element_size,
    _T.storage_offset, _T.is_floating_point, _T.is_complex, _T.get_device, _T.__len__, _T.__hash__,
    _T.is_shared, _T.is_pinned, _T.has_names, _T.is_same_size, _T.is_set_to,
}


def _lazies(args, kwargs):
    stack = [args, kwargs]
    while stack:
        x = stack.pop()
        if isinstance(x, LazyEdgeWeights):
            yield x
        elif isinstance(x, (list, tuple)):
            stack.extend(x)
        elif isinstance(x, dict):
            stack.extend(x.values())


class LazyEdgeWeights(torch.Tensor):
    @staticmethod
    def __new__(cls, storage_tensor, fill, structure, w_csr):
        t = torch.Tensor._make_subclass(cls, storage_tensor)
        t._kgat_fill = fill              # writes the values into storage_tensor; None once done
        t._kgat_lazy = (structure, w_csr)  # what the aggregation reads instead, while pending
        return t

    @property
    def pending(self):
        return self._kgat_fill is not None

    def materialize(self):
        """Run the deferred permutation (once) and turn this object into a plain torch.Tensor."""
        fill = self.__dict__.get("_kgat_fill")
        if fill is not None:
            if self.__dict__.get("_kgat_filling"):
                return self                    # re-entered from inside the fill itself
            self.__dict__["_kgat_filling"] = True
            try:
                fill()                         # may raise (launch error, out of memory): the object then stays
            finally:                           # pending, with its fill intact, and the next read tries again
                self.__dict__.pop("_kgat_filling", None)
        self.__dict__.pop("_kgat_fill", None)
        self.__dict__.pop("_kgat_lazy", None)
        if type(self) is LazyEdgeWeights:
            self.__class__ = torch.Tensor
            self.__dict__["pending"] = False   # `w.pending` keeps answering on the plain tensor
        return self

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        # Tensor.type() without arguments names the type (metadata); with a dtype it is a cast
        meta = func in _META or (func is _T.type and len(args) <= 1 and not kwargs)
        if not meta:
            for t in list(_lazies(args, kwargs)):
                if isinstance(t, LazyEdgeWeights):  # (the same tensor may appear twice: w * w)
                    t.materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    # paths that copy or export the values without a torch operator
    def __deepcopy__(self, memo):
        self.materialize()
        return copy.deepcopy(self, memo)      # a plain tensor now

    def __reduce_ex__(self, proto):
        self.materialize()
        return self.__reduce_ex__(proto)      # torch.Tensor's


_enabled = None   # None: not decided yet (the environment is consulted on first use)
_dlpack_inner = None


def _guard_legacy_to_dlpack(install=True):
    """torch.utils.dlpack.to_dlpack is the C function torch._C._to_dlpack: it exports the storage
    without consulting __torch_function__.  Wrap it (once) so that a pending tensor is filled first;
    install=False puts the original back."""
    global _dlpack_inner
    cur = _dlpack.to_dlpack
    if not install:
        if getattr(cur, "_kgat_guarded", False) and _dlpack_inner is not None:
            if getattr(torch, "to_dlpack", None) is cur:
                torch.to_dlpack = _dlpack_inner
            _dlpack.to_dlpack = _dlpack_inner
        return
    if getattr(cur, "_kgat_guarded", False):
        return
    inner = _dlpack_inner = cur

    def to_dlpack(tensor):
        if isinstance(tensor, LazyEdgeWeights):
            tensor.materialize()
        return inner(tensor)
    to_dlpack.__doc__ = getattr(inner, "__doc__", None)
    to_dlpack._kgat_guarded = True
    _dlpack.to_dlpack = to_dlpack
    if getattr(torch, "to_dlpack", None) is inner:
        torch.to_dlpack = to_dlpack


def enable(flag=True):
    """Turn the deferred edge-id-ordered attention on (or off) for this process; turning it on wraps the legacy
    torch.utils.dlpack.to_dlpack so that a pending tensor is filled before it is exported.  ``flag=None`` returns to
    "not decided by the process" (the environment's KGAT_LAZY_EDGE_WEIGHTS decides again).  Returns the PREVIOUS
    setting (True / False / None), so a caller can restore it: ``prev = enable(True); ...; enable(prev)``."""
    global _enabled
    prev = _enabled
    _enabled = None if flag is None else bool(flag)
    _guard_legacy_to_dlpack(install=bool(_enabled))
    return prev


def enabled():
    """Whether kgat_attention hands back LazyEdgeWeights by default (see the module docstring)."""
    from .options import options
    if options.eager_edge_weights:
        return False
    if _enabled is None and options.lazy_edge_weights:
        enable(True)
    return bool(_enabled)


def pending_csr_weights(w, structure):
    """The CSR-ordered values standing behind `w` if it is a still-pending LazyEdgeWeights of this
    graph structure, else None.  Does not touch `w`'s values."""
    lz = w.__dict__.get("_kgat_lazy") if isinstance(w, LazyEdgeWeights) else None
    return lz[1] if lz is not None and lz[0] is structure else None
