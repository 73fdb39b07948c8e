"""Integer-only synthetic input generator shared by make_golden.py (build container, reference side)
and the tests (both sides regenerate the same bits; no libm, no torch RNG involved)."""
import numpy as np


def _xorshift(n, seed):
    """n uint32 words from a 64-bit xorshift* stream (pure integer arithmetic, portable)."""
    out = np.empty(n, dtype=np.uint32)
    s = np.uint64(seed * 0x9E3779B97F4A7C15 % (1 << 64) or 1)
    # vectorised: run 1024 independent lanes to keep python overhead low
    lanes = 1024
    st = (np.arange(1, lanes + 1, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)) ^ s
    st[st == 0] = np.uint64(1)
    i = 0
    with np.errstate(over="ignore"):
        while i < n:
            st ^= st >> np.uint64(12)
            st ^= st << np.uint64(25)
            st ^= st >> np.uint64(27)
            w = ((st * np.uint64(0x2545F4914F6CDD1D)) >> np.uint64(32)).astype(np.uint32)
            m = min(lanes, n - i)
            out[i:i + m] = w[:m]
            i += m
    return out


def int_bits_tensor(shape, dname, seed, exp_span=12):
    """Random float bit patterns of dtype dname ('f32'|'f16'|'bf16'), shape `shape`, as a numpy
    unsigned array.  Sign random; exponent drawn from `exp_span` binades below 2^-2 with a
    triangular-ish bias to the upper ones; mantissa random but coarse (tie-heavy)."""
    n = int(np.prod(shape))
    w = _xorshift(2 * n, seed)
    a, b = w[:n], w[n:]
    sign = a & 1
    # min of two uniform draws skews toward small offsets below the top binade
    d1 = (a >> 1) % exp_span
    d2 = (a >> 9) % exp_span
    off = np.minimum(d1, d2)
    if dname == "f32":
        e = (125 - off).astype(np.uint32)
        man = (b & 0x7FFFFF) & ~np.uint32(0x3FFF)          # 9 significant mantissa bits -> ties
        return ((sign.astype(np.uint32) << 31) | (e << 23) | man).reshape(shape)
    if dname == "f16":
        e = np.maximum(13 - off.astype(np.int64), 0).astype(np.uint32)     # reaches fp16 subnormals
        man = b & 0x3FF
        return ((sign << 15) | (e << 10) | man).astype(np.uint16).reshape(shape)
    e = (125 - off).astype(np.uint32)
    man = b & 0x7F
    return ((sign << 15) | (e << 7) | man).astype(np.uint16).reshape(shape)
