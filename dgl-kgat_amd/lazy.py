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
permutation into the storage it already owns, on the current stream, and from then on it is an
ordinary tensor (same object, same ``data_ptr``).  What cannot be intercepted is code that takes
the raw pointer out of a C++ extension without going through a torch operator; set
``KGAT_EAGER_EDGE_WEIGHTS=1`` (or ``lazy=False``) to get the permutation up front.
"""
import torch

_T = torch.Tensor
# operations that only look at metadata: they must not trigger the fill
_META = {
    _T.shape.__get__, _T.dtype.__get__, _T.device.__get__, _T.is_cuda.__get__, _T.ndim.__get__,
    _T.requires_grad.__get__, _T.grad_fn.__get__, _T.layout.__get__, _T._version.__get__, _T.is_leaf.__get__,
    _T.grad.__get__, _T.names.__get__, _T.is_sparse.__get__, _T.is_quantized.__get__, _T.is_meta.__get__,
    _T.dim, _T.size, _T.numel, _T.nelement, _T.ndimension, _T.stride, _T.is_contiguous, _T.element_size,
    _T.storage_offset, _T.is_floating_point, _T.is_complex, _T.get_device, _T.__len__, _T.__hash__,
    _T.is_shared, _T.is_pinned, _T.has_names, _T.is_same_size, _T.is_set_to, _T.type,
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
        """Run the deferred permutation (once); afterwards this is an ordinary tensor."""
        fill = self._kgat_fill
        if fill is not None:
            self._kgat_fill = None
            self._kgat_lazy = None
            fill()
        return self

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func not in _META:
            for t in _lazies(args, kwargs):
                t.materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def pending_csr_weights(w, structure):
    """The CSR-ordered values standing behind `w` if it is a still-pending LazyEdgeWeights of this
    graph structure, else None.  Does not touch `w`'s values."""
    lz = getattr(w, "_kgat_lazy", None) if isinstance(w, LazyEdgeWeights) else None
    return lz[1] if lz is not None and lz[0] is structure else None
