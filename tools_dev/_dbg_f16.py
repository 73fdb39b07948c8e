import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from util import from_bits
from oracle import oracle as O
dname="f16"; dt=torch.float16
hi = 0x7C00
pat = np.arange(0, hi, dtype=np.uint16); n = pat.size
rng = np.random.default_rng(9)
blk = np.zeros((n, 64), dtype=np.uint16); blk[:, 0] = pat
for j in range(1, 64):
    drop = rng.integers(0, 40, size=n).astype(np.int64) * (1 << 10)
    m_ = np.maximum(pat.astype(np.int64) - drop - rng.integers(0, 128, size=n), 0)
    blk[:, j] = (m_.astype(np.uint16)) | (rng.integers(0, 2, size=n).astype(np.uint16) << 15)
blk[:, 0] |= (rng.integers(0, 2, size=n).astype(np.uint16) << 15)
blk[:, [0, 37]] = blk[:, [37, 0]]
xc = from_bits(blk.reshape(-1), dt).view(n, 64); x = xc.cuda()
m=3
codes, exps = bfp_ops.float_to_bfp_packed(x, m, 64, code_bits=4)
c = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64, num_format='bfp', weight_mant_bits=15, in_sparsity=False, w_sparsity=False, grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp', first='s', sparsity_mode='structured')
want = O.float_to_bfp_blocked(xc, **c, identifier='w').to(torch.float64)
cc = codes.cpu()
q = torch.stack([(cc & 0xF).to(torch.int16), (cc >> 4).to(torch.int16)], dim=-1).view(n, 64)
q = torch.where(q > 7, q - 16, q).to(torch.float64)
e = exps.cpu().to(torch.float64).view(n, 1)
val = q * torch.pow(torch.tensor(2.0, dtype=torch.float64), e - m)
nanblk = exps.cpu().view(n) == -128
bad = (val != want) & ~nanblk.view(n,1) & ~torch.isnan(want)
idx = bad.nonzero()
print("mismatches", len(idx))
for r, cidx in idx[:12].tolist():
    print(f"row {r} (max bits 0x{int(pat[r]):04x}) col {cidx}: x bits 0x{int(blk[r,cidx]):04x} x={float(xc[r,cidx]):.6g} got q={q[r,cidx].item()} e={e[r,0].item()} val={val[r,cidx].item():.6g} want={want[r,cidx].item():.6g}")
rows = sorted(set(idx[:,0].tolist())); print("rows", rows[:20], "...", rows[-5:])
