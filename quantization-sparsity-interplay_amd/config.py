"""BFPConfig: typed view of the reference's `hbfp:` config dict.

The reference has no config class: a YAML dict (bfp/bfp_config.yaml:1-21) is read by
bfp_util.get_bfp_args() (bfp_util.py:8-16) and splatted as **kwargs into BFPLinear/BFPConv2d, where
unpack_bfp_args (bfp_ops.py:202-231) pops the 20 known keys with defaults and silently drops the
rest.  BFPConfig round-trips exactly that dict (same keys, same defaults, unknown keys tolerated).
"""
from dataclasses import dataclass, fields, asdict

import yaml


@dataclass
class BFPConfig:
    num_format: str = 'fp32'
    sparsity_num_format: str = 'fp32'
    rounding_mode: str = 'stoc'
    epsilon: float = 1e-8
    mant_bits: int = 0
    block_size: int = 0
    weight_mant_bits: int = 0
    in_sparsity: bool = False
    w_sparsity: bool = False
    grad_sparsity: bool = False
    N: int = 0
    M: int = 0
    first: str = 's'
    sparsity_mode: str = 'unstructured'
    sparsity_frac: float = 0
    mx_w_elem_format: str = ''
    mx_a_elem_format: str = ''
    bfloat: int = 16
    scale_bits: int = 8
    device: str = 'cpu'

    @classmethod
    def keys(cls):
        return [f.name for f in fields(cls)]

    @classmethod
    def from_dict(cls, d):
        """unknown keys (e.g. `bfp_tile_size`, `unconstrained`, written by some reference scripts) are ignored"""
        known = set(cls.keys())
        return cls(**{k: v for k, v in dict(d).items() if k in known})

    @classmethod
    def from_yaml(cls, path):
        with open(path) as f:
            return cls.from_dict(yaml.safe_load(f)['hbfp'])

    def to_kwargs(self):
        """the dict BFPLinear(**kwargs) / float_to_bfp_blocked(**kwargs) expect"""
        return asdict(self)

    def to_yaml(self, path):
        with open(path, 'w') as f:
            yaml.safe_dump({'hbfp': self.to_kwargs()}, f, sort_keys=False)

    # convenience constructors for the configurations BASELINE.json names
    @classmethod
    def hbfp(cls, bits, block_size, device='cuda', rounding_mode='determ', **kw):
        """HBFP<bits>: sign + (bits-1) magnitude bits (reference naming: hbfp8 -> mant_bits 7)"""
        return cls(num_format='bfp', sparsity_num_format='bfp', mant_bits=bits - 1, block_size=block_size,
                   rounding_mode=rounding_mode, device=device, **kw)
