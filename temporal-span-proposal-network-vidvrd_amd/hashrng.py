"""Build-owned counter-based RNG for synthetic inputs and weights.

Every synthetic tensor used by the parity tests, the golden-vector generator
(tests/golden/make_golden.py) and bench.py comes from this generator, so that
both sides of a comparison can regenerate inputs and weights bit-for-bit
without storing them and without depending on torch's RNG stream (which is not
stable across torch versions / devices).

Element ``i`` of stream ``(seed, tag)`` is ``splitmix64(key + i)`` where
``key = splitmix64(seed * GOLDEN + hash(tag))``.  Only integer arithmetic
mod 2**64 and exactly representable float operations are used, so the
values are identical on every platform / numpy version.
"""
import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = 0x9E3779B97F4A7C15
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(z):
    """splitmix64 finaliser on a uint64 ndarray (wrapping arithmetic)."""
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z += np.uint64(_GOLDEN)
        z ^= z >> np.uint64(30)
        z *= _M1
        z ^= z >> np.uint64(27)
        z *= _M2
        z ^= z >> np.uint64(31)
    return z


def _key(seed, tag):
    t = zlib.crc32(tag.encode("utf-8")) & 0xFFFFFFFF
    k = (int(seed) * _GOLDEN + t * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    return int(_splitmix64(np.array([k], dtype=np.uint64))[0])


def bits(seed, tag, n, offset=0):
    """``n`` raw uint64 words of stream (seed, tag), starting at ``offset``."""
    k = _key(seed, tag)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        idx = idx * np.uint64(_GOLDEN) + np.uint64(k)
    return _splitmix64(idx)


def uniform(seed, tag, shape, lo=0.0, hi=1.0, dtype=np.float32):
    """U[lo, hi) with a 24-bit mantissa (exactly representable in fp32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    b = bits(seed, tag, n)
    u = (b >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)
    if lo != 0.0 or hi != 1.0:
        u = lo + (hi - lo) * u
    return u.astype(dtype).reshape(shape)


def normal(seed, tag, shape, std=1.0, dtype=np.float32):
    """Approximately N(0, std^2): Irwin-Hall sum of four 16-bit uniforms.

    Exact in float64 (sums of 16-bit integers), one correctly rounded multiply,
    so it is reproducible everywhere.  Variance is exactly std^2; the tails are
    bounded (|x| <= 2*sqrt(3)*std), which is irrelevant for weight init.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    b = bits(seed, tag, n)
    m = np.uint64(0xFFFF)
    s = ((b & m).astype(np.float64) + ((b >> np.uint64(16)) & m).astype(np.float64)
         + ((b >> np.uint64(32)) & m).astype(np.float64) + ((b >> np.uint64(48)) & m).astype(np.float64))
    # each term U{0..65535}: mean 32767.5, var (65536^2-1)/12
    z = (s - 4 * 32767.5) / np.sqrt(4 * (65536.0 ** 2 - 1) / 12.0)
    return (z * std).astype(dtype).reshape(shape)


def integers(seed, tag, shape, lo, hi):
    """Uniform integers in [lo, hi) as int64."""
    n = int(np.prod(shape)) if len(shape) else 1
    b = bits(seed, tag, n)
    span = np.uint64(hi - lo)
    return ((b >> np.uint64(11)) % span).astype(np.int64).reshape(shape) + lo
