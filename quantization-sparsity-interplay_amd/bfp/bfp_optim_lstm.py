"""BFPAdam with the reference's name and module path (src/transformers/bfp/bfp_optim_lstm.py:12-93): Adam whose
freshly updated parameters are snapped to the wide-mantissa HBFP grid -- float_to_bfp_blocked(p, sgd_update=True),
i.e. `weight_mant_bits` instead of `mant_bits`, no sparsity (identifier '') -- so that weights are stored at HBFP16
precision between steps.  The moment math is plain Adam; only the snap runs in the BFP engine."""
import math

import torch

from . import bfp_ops
from .bfp_util import get_bfp_args

required = object()


class BFPAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                 bfp_args=None, quantize_fn=None):
        """bfp_args: the `hbfp` config dict (default: bfp_util.get_bfp_args(), like the reference, :15).
        quantize_fn: callable with float_to_bfp_blocked's signature (default: the HIP engine; tests may inject a CPU checker)."""
        self.bfp_args = dict(bfp_args) if bfp_args is not None else get_bfp_args()
        self._quantize = quantize_fn or bfp_ops.float_to_bfp_blocked
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        fmt = self.bfp_args['num_format']
        if fmt not in ('fp32', 'bfp'):
            raise NotImplementedError('NumFormat not implemented')
        known = bfp_ops.unpack_bfp_args(dict(self.bfp_args))            # float_to_bfp_blocked takes exactly the 20 known keys
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                grad = p.grad
                if grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                state = self.state[p]
                if len(state) == 0:                                     # reference :45-53
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                    if group['amsgrad']:
                        state['max_exp_avg_sq'] = torch.zeros_like(p)
                state['step'] += 1
                if group['weight_decay'] != 0:
                    grad = grad.add(p, alpha=group['weight_decay'])     # :62-63
                m, v = state['exp_avg'], state['exp_avg_sq']
                m.mul_(beta1).add_(grad, alpha=1 - beta1)               # :66
                v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)     # :67
                if group['amsgrad']:
                    torch.max(state['max_exp_avg_sq'], v, out=state['max_exp_avg_sq'])   # :70
                    denom = state['max_exp_avg_sq'].sqrt().add_(group['eps'])
                else:
                    denom = v.sqrt().add_(group['eps'])                 # :74
                bc1 = 1 - beta1 ** state['step']
                bc2 = 1 - beta2 ** state['step']
                step_size = group['lr'] * math.sqrt(bc2) / bc1          # :76-78
                p.addcdiv_(m, denom, value=-step_size)                  # :82 / :85
                if fmt == 'bfp':
                    p.copy_(self._quantize(p, sgd_update=True, **known))   # :85-87
        return loss
