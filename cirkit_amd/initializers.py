"""Closed-form, seedable parameter values.

The reference draws parameters from the global torch RNG on every ``compile`` (reset_parameters,
cirkit/backend/torch/compiler.py:302), so fixtures cannot rely on it.  Instead every named tensor
of a plan is filled from a counter-based generator that needs only 64-bit integer arithmetic and
one float multiply -- bit-reproducible on any host (and cheap enough that 32 MB parameter blobs
never have to be committed).  The reference side of the parity fixtures loads exactly these values
through ``load_state_dict`` (see tests/golden/make_fixtures.py).

``theta = sqrt(3) * (u1 + u2 + u3 + u4 - 2)`` with ``u_i`` uniform 16-bit fractions taken from one
splitmix64 output: zero mean, unit variance, support [-3.46, 3.46] (Irwin-Hall, n = 4).
"""

from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def pseudo_normal(shape: tuple[int, ...], *, stream: int, seed: int = 0) -> np.ndarray:
    """float32 array of `shape`; element i depends only on (seed, stream, i)."""
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    with np.errstate(over="ignore"):
        base = _splitmix64(np.asarray([(seed << 32) ^ stream], dtype=np.uint64))[0]
        h = _splitmix64(np.arange(n, dtype=np.uint64) + base)
    acc = np.zeros(n, dtype=np.int64)
    for s in (0, 16, 32, 48):
        acc += ((h >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.int64)
    # (sum of four 16-bit fractions - 2) * sqrt(3), evaluated exactly in float64 then rounded once
    val = (acc.astype(np.float64) / 65536.0 - 2.0) * 1.7320508075688772
    return val.astype(np.float32).reshape(shape)


def stream_of(name: str) -> int:
    """Stable 32-bit stream id of a tensor name."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def init_plan_tensors(plan, *, seed: int = 0) -> dict[str, np.ndarray]:
    """Deterministic values for every named tensor of `plan` (float32, or complex64 with a zero
    imaginary part when the plan declares a complex tensor)."""
    out: dict[str, np.ndarray] = {}
    for name, (shape, dtype) in plan.tensors.items():
        v = pseudo_normal(tuple(shape), stream=stream_of(name), seed=seed)
        if "complex" in dtype:
            v = v.astype(np.complex64)
        elif dtype not in ("float32", "float"):
            v = v.astype(np.dtype(dtype))
        out[name] = v
    return out
