"""The C ABI without Python/torch in the loop: tests/cabi/cabi_check.cpp is compiled with hipcc against
include/bfpq.h + libbfpq.so (+ the oracle library as the checker) and run on the GPU.
The CPU half checks argument validation through ctypes (no launch happens for a rejected call)."""
import ctypes
import os
import subprocess

import pytest

import quantization_sparsity_interplay_amd as pkg
from quantization_sparsity_interplay_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_argument_validation_without_gpu():
    L = pkg.load_library()
    n = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)               # a non-null, never dereferenced pointer: rejected before any launch
    assert L.bfpq_quantize_nm(n, one, n, n, 4, 64, native.BF16, 64, 3, 1e-8, 2, 4, 1, 0, 0, one, one, n, n) == -1      # null input
    assert L.bfpq_quantize_nm(one, n, n, n, 4, 64, native.BF16, 64, 3, 1e-8, 2, 4, 1, 0, 0, one, one, n, n) == -1      # no output
    assert L.bfpq_quantize_nm(one, one, n, n, 4, 64, 9, 64, 3, 1e-8, 2, 4, 1, 0, 0, one, one, n, n) == -1              # bad dtype
    assert L.bfpq_quantize_nm(one, one, n, n, 4, 64, native.BF16, 64, 3, 1e-8, 5, 4, 1, 0, 0, one, one, n, n) == -1    # N > M
    assert L.bfpq_quantize_nm(one, one, n, n, 4, 64, native.BF16, 64, 3, 1e-8, 2, 65, 1, 0, 0, one, one, n, n) == -2   # M > 64
    assert L.bfpq_quantize_nm(one, one, one, n, 4, 64, native.BF16, 64, 7, 1e-8, 0, 0, 1, 4, 0, one, one, n, n) == -1  # 4-bit codes, 7-bit mantissa
    assert L.bfpq_quantize_nm(one, one, n, n, 4, 64, native.BF16, 64, 3, 1e-8, 0, 0, 1, 0, 0, n, one, n, n) == -1      # no exponent table
    assert L.bfpq_quantize_nm(one, one, n, n, 0, 64, native.BF16, 64, 3, 1e-8, 2, 4, 1, 0, 0, one, one, n, n) == 0     # empty tensor: ok, no launch
    assert L.bfpq_nm_sparsify(one, one, 4, 64, native.F32, 0, 4, one, n) == -1
    assert L.bfpq_select_resolve(one, 64, native.BF16, 3, 5, n, 1, 0, one, n, n) == -1                                  # pass out of range
    assert L.bfpq_select_hist(one, 64, native.BF16, 0, 65, 64, one, n, n) == -1                                         # k > numel
    assert L.bfpq_select_hist(one, 64, native.BF16, 0, 5, 1 << 33, one, n, n) == -2                                     # 32-bit counters
    assert L.bfpq_select_resolve(one, 64, native.BF16, 0, 5, one, 2, 2, one, n, n) == -1                                # rank >= n_ranks
    assert L.bfpq_select_ws_bytes() == 80 + 4 * (12 + 8 * 512 + 3 * 8 * 33024 + 3 * 256 + 2 * 256 * 2048)
    assert L.bfpq_select(one, 64, native.BF16, 65, one, n) == -1 and L.bfpq_select(one, 0, native.BF16, 0, one, n) == 0            # k > numel; empty
    assert L.bfpq_threshold_apply(one, one, 64, native.BF16, one, n) == -1                                               # in place: refused
    assert L.bfpq_threshold_apply(one, one, 0, native.BF16, one, n) == 0                                                # empty: no launch
    assert L.bfpq_int_quantize(one, one, 1, 4, 4, 5, 8, n, n) == -1
    assert L.bfpq_tune(0, 0) == -1 and L.bfpq_tune(99, 5) == -1 and L.bfpq_tune(0, 1280) == 0
    assert L.bfpq_tune(3, -1) == -1 and L.bfpq_tune(3, 24) == 0                                                        # own-launch threshold of list calls
    # list entry points (ABI 4): rejected before anything is launched or any event is made
    plan = native._Plan(native.BF16, 64, 3, 2, 4, 1, 1e-8, 16, 16)
    pp = ctypes.c_void_p(ctypes.addressof(plan))
    assert L.bfpq_fake_quantize_list(n, one, 1, n, n, 0) == -1                                                          # no plan
    assert L.bfpq_fake_quantize_list(pp, n, 1, n, n, 0) == -1                                                           # descriptors missing
    assert L.bfpq_fake_quantize_list(pp, one, 1, n, n, 2) == -1                                                         # aux count without aux streams
    assert L.bfpq_fake_quantize_list(pp, one, 0, n, n, 0) == 0 and L.bfpq_fake_quantize_batched(pp, one, 0, n) == 0     # empty list: ok
    bad = native._TensorDesc(16, 16, -1, 64, 1, 0)
    assert L.bfpq_fake_quantize_list(pp, ctypes.c_void_p(ctypes.addressof(bad)), 1, n, n, 0) == -1                      # negative rows
    wsp = (ctypes.c_void_p * 2)(16, 16)
    assert L.bfpq_prune_quantize_list(one, 1, native.BF16, 64, 3, 1e-8, one, ctypes.c_void_p(ctypes.addressof(wsp)), 2, n, n, 0) == -1   # one workspace given twice
    assert L.bfpq_prune_quantize_list(one, 1, native.BF16, 64, 3, 1e-8, one, n, 1, n, n, 0) == -1                       # no workspaces
    assert L.bfpq_prune_quantize_list(n, 0, native.BF16, 64, 3, 1e-8, one, ctypes.c_void_p(ctypes.addressof(wsp)), 1, n, n, 0) == 0     # empty list: ok
    assert L.bfpq_is_fused(4, 64, 5, 64, 2, 4) == 0
    # packed-format entry points
    assert L.bfpq_compact24(one, one, one, 6, one, n) == -1                                     # code bytes not a multiple of 4
    assert L.bfpq_compact24(one, one, one, 8, n, n) == -1                                       # no status word
    assert L.bfpq_compact24(one, one, one, 0, one, n) == 0 and L.bfpq_expand24(one, one, one, 0, n) == 0
    assert L.bfpq_hbfp_linear_tiled_ok(4096, 11008) == 1 and L.bfpq_hbfp_linear_tiled_ok(4096, 128) == 0 and L.bfpq_hbfp_linear_tiled_ok(100, 1024) == 0
    assert L.bfpq_hbfp_linear_decode_tiled(one, one, one, one, one, 65, 4096, 11008, native.BF16, 3, 7, n) == -1     # more than 64 tokens
    assert L.bfpq_hbfp_linear_decode_tiled(one, one, one, one, one, 1, 4096, 11008, native.BF16, 4, 7, n) == -1      # weight mantissa > 3 bits
    assert L.bfpq_hbfp_linear_decode_tiled(one, one, one, one, one, 1, 4096, 1000, native.BF16, 3, 7, n) == -2       # shape the tiled layout does not take
    # prefill consumer (block-scaled matrix instruction) and its image builders
    assert L.bfpq_hbfp_linear_mx8_ok(2048, 11008, 4096) == 1 and L.bfpq_hbfp_linear_mx8_ok(2048, 11008, 4160) == 0 and L.bfpq_hbfp_linear_mx8_ok(0, 8, 256) == 0
    assert L.bfpq_hbfp_linear_mx8(one, one, one, one, n, one, 8, 8, 200, native.BF16, n) == -2                        # K % 256 != 0
    assert L.bfpq_hbfp_linear_mx8(one, one, one, one, n, one, 8, 8, 256, 7, n) == -1                                  # bad output dtype
    assert L.bfpq_hbfp_linear_mx8(n, one, one, one, n, one, 8, 8, 256, native.BF16, n) == -1                          # null operand
    assert L.bfpq_hbfp_linear_mx8(ctypes.c_void_p(8), one, one, one, n, one, 8, 8, 256, native.BF16, n) == -1         # image not 16-byte aligned
    assert L.bfpq_hbfp_linear_mx8(one, one, one, one, n, one, 0, 8, 256, native.BF16, n) == 0                         # no tokens: ok, no launch
    assert L.bfpq_mx8_from_hbfp(one, one, one, one, 4, 64, 4, 4, n) == -1                                             # 4-bit codes hold <= 3 mantissa bits
    assert L.bfpq_mx8_from_hbfp(one, one, one, one, 4, 64, 8, 5, n) == -1                                             # 5 mantissa bits do not fit e4m3
    assert L.bfpq_mx8_from_hbfp(one, one, one, one, 4, 100, 8, 3, n) == -1                                            # cols % 64 != 0
    assert L.bfpq_quantize_mx8(one, one, one, 4, 100, native.F32, 3, 1e-8, one, n) == -2                              # cols % 64 != 0: through the codes
    assert L.bfpq_quantize_mx8(one, one, one, 4, 64, native.BF16, 5, 1e-8, one, n) == -1
    assert L.bfpq_quantize_mx8(one, one, one, 0, 64, native.BF16, 3, 1e-8, one, n) == 0
    assert L.bfpq_hbfp_linear_mx8_parts(128, 4096, 11008) == 8 and L.bfpq_hbfp_linear_mx8_parts(2048, 4096, 11008) == 1 and L.bfpq_hbfp_linear_mx8_parts(128, 11008, 4096) == 2
    assert L.bfpq_hbfp_linear_mx8_splitk(one, one, one, one, n, one, n, 4, 8, 8, 256, native.BF16, n) == -1           # no slabs
    assert L.bfpq_hbfp_linear_mx8_splitk(one, one, one, one, n, one, one, 0, 8, 8, 256, native.BF16, n) == -1         # parts < 1
    assert L.bfpq_tune(2, 7) == -1 and L.bfpq_tune(2, 6) == 0 and L.bfpq_tune(2, -1) == 0
    assert L.bfpq_hbfp_linear_decode(one, one, one, one, one, one, 17, 4096, 11008, native.BF16, 3, 7, n) == -1      # row-major layout: <= 16 tokens
    assert L.bfpq_nm8_lut_host(0, one) == -1 and L.bfpq_nm8_lut_host(4, n) == -1
    assert L.bfpq_tune(1, 3) == -1 and L.bfpq_tune(1, 2) == 0 and L.bfpq_tune(1, 0) == 0


@pytest.mark.gpu
def test_cabi_check_program(tmp_path):
    exe = str(tmp_path / "cabi_check")
    pkg_dir = os.path.join(ROOT, "quantization-sparsity-interplay_amd")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "cabi_check.cpp"), "-o", exe,
                           "-L", pkg_dir, "-lbfpq", "-L", os.path.join(ROOT, "oracle"), "-lbfp_oracle",
                           "-Wl,-rpath," + pkg_dir, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "CABI CHECK PASSED" in out.stdout, out.stdout + out.stderr
