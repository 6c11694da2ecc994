"""Counter-based noise shared by the oracle and the HIP sampler (TEST INFRASTRUCTURE).

The reference draws its Exp(1) race noise with `torch.empty_like(p).exponential_(1)`
(`fam/llm/fast_inference_utils.py:61-65`), i.e. from torch's device Philox stream,
which no other implementation can reproduce bit-for-bit.  The engine therefore
defines its own stream -- Philox4x32-10 (Salmon et al., SC'11), one evaluation per
(token id, step, stream) -- and this file restates it in numpy so that a sampled
token stream can be checked exactly.  Parity with the *reference* for stochastic
sampling is distributional only (DESIGN.md "Sampling").
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Counters are uint32 arrays, key two python ints."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def uniform24(bits):
    """uint32 -> float32 in (0,1): ((x >> 8) + 0.5) * 2^-24, exact in fp32."""
    return ((bits >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def exp1_noise(seed: int, step: int, n: int, stream: int = 0) -> np.ndarray:
    """Exp(1) variate for token ids 0..n-1 at decode step `step` (fp32).

    counter = (token id, step, stream, 0), key = (seed lo, seed hi), word 0 is used.
    Mirrors `mv_exp1_noise` in metavoice-src_amd/csrc/sampling.hip.
    """
    i = np.arange(n, dtype=np.uint32)
    w0, _, _, _ = philox4x32_10(i, np.uint32(step & 0xFFFFFFFF), np.uint32(stream & 0xFFFFFFFF), np.uint32(0),
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = uniform24(w0)
    return (-np.log(u.astype(np.float32))).astype(np.float32)


def uniform_noise(seed: int, rows: int, head: int) -> np.ndarray:
    """U(0,1) variate of the second-stage sampler for flat row index 0..rows-1 (= b*T + t) and codebook `head`.

    counter = (row, head, 1, 0), key = (seed lo, seed hi), word 0.  Mirrors `s2_sample_kernel` in
    metavoice-src_amd/csrc/stage2.hip.
    """
    i = np.arange(rows, dtype=np.uint32)
    w0, _, _, _ = philox4x32_10(i, np.uint32(head), np.uint32(1), np.uint32(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return uniform24(w0)
