"""Deterministic, platform-independent test inputs (integer splitmix64; no libm, no numpy RNG).

Shared by tests/golden/make_golden.py (which feeds them to the genuine reference) and by the
tests (which feed the same inputs to the oracle / the HIP path and compare with the stored
outputs or digests).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n uint64 words of the splitmix64 stream started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rand_bytes(seed, shape):
    n = int(np.prod(shape))
    words = splitmix64(seed, (n + 7) // 8)
    return words.view(np.uint8)[:n].reshape(shape).copy()


def pm1_message(seed, K, frames):
    """iid +-1 symbols (testbench.cc:126-127: 1 - 2*bit), shape [K][frames]."""
    bits = rand_bytes(seed, (K, frames)) & 1
    return (1 - 2 * bits.astype(np.int8)).astype(np.int8)


# sigma = sqrt(1/(2*10^(snr/10))) (testbench.cc:114) as literals so no libm call is involved
_SIGMA = {"awgn-3": 0.9988493699365052, "awgn-1": 0.7934201284279122, "awgn+1": 0.6302344985034508}


def family_llr(family, seed, codeword):
    """int8 LLR inputs of the named family, same shape as `codeword` ([N][frames])."""
    shape = codeword.shape
    if family == "uniform":
        return rand_bytes(seed, shape).view(np.int8)
    if family == "small":
        return (rand_bytes(seed, shape) % 7).astype(np.int8) - np.int8(3)
    sigma = _SIGMA[family]
    # approximately normal noise: centred sum of 16 uniform bytes (variance 16*(256^2-1)/12)
    u = rand_bytes(seed, shape + (16,)).astype(np.int64).sum(axis=-1) - 2040
    g = u.astype(np.float64) / 295.60108254199625
    fact = 2.0 / (sigma * sigma)
    y = codeword.astype(np.float64) + sigma * g
    return np.clip(np.rint(fact * y), -128, 127).astype(np.int8)


def corner_frames(N):
    """Hand-made corner frames, shape [N][12]."""
    i = np.arange(N)
    cols = [
        np.zeros(N), np.full(N, -128), np.full(N, 127), np.full(N, -127), np.full(N, 1), np.full(N, -1),
        np.where(i % 2 == 0, 127, -128), np.where(i % 2 == 0, -128, 127),
        np.where(i % 4 == 1, -5, 5),                 # equal magnitudes: every SPC position ties
        np.where((i * 2654435761 >> 7) % 3 == 0, -9, 9),
        np.where(i % 3 == 0, 0, np.where(i % 3 == 1, 64, -64)),
        ((i * 37) % 256 - 128),
    ]
    return np.stack(cols, axis=1).astype(np.int8)
