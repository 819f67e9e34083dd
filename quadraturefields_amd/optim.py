"""``torch.optim.Adam`` for the fields of the path, one launch per parameter tensor.

The reference builds ``torch.optim.Adam([... radiance_field.parameters(), field_net.parameters() ...], eps=1e-15)``
(``examples/train_finetune.py:402-417``) and calls ``optimizer.step()`` once per iteration (:531-533).  torch's foreach
Adam issues eleven launches and seven passes over every tensor; on the deformation field's T = 2^24 hash table (203 M
parameters) that is 3.8 ms of a step.  ``Adam`` below is the same optimiser -- same constructor arguments, the same
``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so checkpoints written by either load into
the other), the same update element for element -- with ``qf_adam_step`` doing a tensor's update in one launch.
``amsgrad``, ``capturable``, ``differentiable`` and sparse gradients are not supported (the reference uses none).

    from quadraturefields_amd.optim import Adam      # instead of torch.optim.Adam
"""
import torch

from . import _C


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference and not implemented")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=maximize))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("sparse gradients are not supported")
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise RuntimeError("quadraturefields_amd.optim.Adam updates contiguous fp32 parameters on the HIP device")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)          # torch's layout (a host scalar tensor)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                with torch.cuda.device(p.device):
                    _C.check(_C.lib().qf_adam_step(
                        _C.ptr(p.data), _C.ptr(g), _C.ptr(state["exp_avg"]), _C.ptr(state["exp_avg_sq"]), p.numel(),
                        float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]),
                        1 if group["maximize"] else 0, int(state["step"]), _C.stream()), "qf_adam_step")
        return loss
