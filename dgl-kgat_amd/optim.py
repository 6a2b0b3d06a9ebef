"""``FusedAdam``: torch.optim.Adam (the optimiser of reference kgat.py:85) with the whole step as ONE launch of
``kgat_adam_step_f32`` over every parameter that has a gradient.

Same hyper-parameters, same state (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter: a ``state_dict`` moves
between this class and ``torch.optim.Adam``), same dense semantics - the reference's embedding table moves in every
step, also in rows whose gradient is zero.  fp32 HIP parameters only (anything else raises: there is no fallback);
``amsgrad`` / ``weight_decay`` / ``maximize`` are not part of the reference's call and are refused."""
import ctypes as C

import torch

from . import _lib
from ._lib import KGATLibraryError, check


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, zero_grads=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam covers the reference's optim.Adam(parameters, lr): no weight decay, no amsgrad")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False))
        self._zero_grads = bool(zero_grads)   # also clear the gradients in the same pass (zero_grad(set_to_none=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        cap = lib.kgat_adam_max_tensors()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise KGATLibraryError("FusedAdam: parameters must be contiguous float32 HIP tensors (got %s on %s)"
                                           % (p.dtype, p.device))
                if p.grad.is_sparse:
                    raise KGATLibraryError("FusedAdam: sparse gradients are outside the reference's dense Adam")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # one launch per device and per `cap` tensors (a launch runs on ONE device's current stream: ADVICE round 5)
            by_dev = {}
            for p in ps:
                by_dev.setdefault(p.device, []).append(p)
            chunks = [(dv, lst[lo:lo + cap]) for dv, lst in by_dev.items() for lo in range(0, len(lst), cap)]
            for dv, chunk in chunks:
                n = len(chunk)
                steps = []
                for p in chunk:
                    st = self.state[p]
                    st["step"] += 1                      # (a CPU scalar tensor, as torch.optim.Adam keeps it)
                    steps.append(int(st["step"]))
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in chunk]
                arr_p = (C.c_void_p * n)(*[p.data_ptr() for p in chunk])
                arr_g = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
                arr_m = (C.c_void_p * n)(*[self.state[p]["exp_avg"].data_ptr() for p in chunk])
                arr_v = (C.c_void_p * n)(*[self.state[p]["exp_avg_sq"].data_ptr() for p in chunk])
                arr_n = (C.c_int64 * n)(*[p.numel() for p in chunk])
                arr_t = (C.c_int64 * n)(*steps)
                zero = self._zero_grads and all(g is p.grad for g, p in zip(grads, chunk))
                with torch.cuda.device(dv):
                    check(lib.kgat_adam_step_f32(n, arr_n, arr_p, arr_g, arr_m, arr_v, arr_t, float(group["lr"]),
                                                 float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                                                 int(zero), torch.cuda.current_stream(dv).cuda_stream),
                          "kgat_adam_step_f32")
        return loss

    def kg_state(self, params):
        """For KGATPropagation.kg_phase: the Adam state of `params` (created as step() creates it), and the hyper-parameters
        they share - None when they sit in groups with different hyper-parameters."""
        hyper = None
        for p in params:
            grp = next((g for g in self.param_groups if any(q is p for q in g["params"])), None)
            if grp is None:
                return None
            h = (float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]))
            if hyper is not None and h != hyper:
                return None
            hyper = h
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                return None
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return hyper
