import sys; sys.path.insert(0,'.')
import torch
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from quantization_sparsity_interplay_amd import native
x = (torch.linspace(-1,1,16)*60000).to(torch.float16).view(1,16)
print("in  ", [hex(v & 0xffff) for v in x.view(torch.int16).view(-1).tolist()])
for m in (3,7):
    y = bfp_ops._no_sparsity_float_to_bfp(x.cuda(), 16, m, 1e-8, 'determ', 'cuda')
    print("fused m",m, [hex(v & 0xffff) for v in y.cpu().view(torch.int16).view(-1).tolist()])
    buf = torch.empty(16+8, dtype=torch.float16, device='cuda'); xs = buf[1:17].view(1,16); xs.copy_(x)
    y = bfp_ops._no_sparsity_float_to_bfp(xs, 16, m, 1e-8, 'determ', 'cuda')
    print("rows  m",m, [hex(v & 0xffff) for v in y.cpu().view(torch.int16).view(-1).tolist()])
x2 = x.clone(); x2[0,0] = -30000; x2[0,15]=30000
y = bfp_ops._no_sparsity_float_to_bfp(x2.cuda(), 16, 3, 1e-8, 'determ', 'cuda')
print("fast path", [hex(v & 0xffff) for v in y.cpu().view(torch.int16).view(-1).tolist()])
