"""Config helpers with the reference's names (src/transformers/bfp/bfp_util.py:8-35)."""
import os

import yaml

_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bfp_config.yaml')


def get_bfp_args(path=None):
    """reference: bfp_util.py:8-16 -- the `hbfp` dict of the YAML next to this module.
    (The reference also prints the dict on every call, i.e. once per constructed layer; this does not.)"""
    with open(path or os.environ.get('BFP_CONFIG', _CONFIG)) as f:
        return yaml.safe_load(f)['hbfp']


def extract_sparsity_args(bfp_args):
    """reference: bfp_util.py:18-26"""
    out = {"sparsity": True if bfp_args["w_sparsity"] else False}
    for name in ("device", "sparsity_mode", "sparsity_frac", "N", "M"):
        out[name] = bfp_args[name]
    return out


def extract_mx_args(bfp_args):
    """reference: bfp_util.py:28-35"""
    return {"w_elem_format": bfp_args["mx_w_elem_format"], "a_elem_format": bfp_args["mx_a_elem_format"],
            "block_size": bfp_args["block_size"], "bfloat": bfp_args["bfloat"], "scale_bits": bfp_args["scale_bits"]}
