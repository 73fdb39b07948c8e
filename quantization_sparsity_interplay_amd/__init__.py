"""Importable alias of the package directory `quantization-sparsity-interplay_amd/` (a hyphen cannot
appear in a Python module name).  `import quantization_sparsity_interplay_amd as bfpq` resolves every
submodule (`.bfp.bfp_ops`, `.native`, `.config`, `.dist`) from that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "quantization-sparsity-interplay_amd")
__path__.insert(0, _real)

from .native import lib_path, load_library, NativeUnavailable  # noqa: E402,F401
from .config import BFPConfig  # noqa: E402,F401
