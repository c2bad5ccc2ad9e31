"""R's default random number stream (Mersenne-Twister + inversion), enough of it to regenerate the INPUTS of the
reference tests that hold hard-coded expected values (tests/testthat/test_approx.R:37-52, test_basics.R:183-200):
`set.seed(s); rnorm(k)`.  Follows R's documented algorithm (src/main/RNG.c: Randomize / MT_sgenrand scrambling,
fixup; snorm.c INVERSION with BIG = 134217728) -- written from that description, no R needed.

Usage: python tests/golden/r_rng.py   -> prints set.seed(123); rnorm(5) and set.seed(1); rnorm(9)
"""
import numpy as np
from scipy.special import ndtri


def r_set_seed(seed: int) -> np.random.MT19937:
    """RNG_Init(MERSENNE_TWISTER, seed): 50 LCG scrambling rounds, then 625 LCG outputs fill dummy[0..624];
    FixupSeeds sets dummy[0] = 624 (mti = N: the state is regenerated on the first draw)."""
    s = np.uint32(seed)
    with np.errstate(over="ignore"):
        for _ in range(50):
            s = np.uint32(69069) * s + np.uint32(1)
        key = np.zeros(625, dtype=np.uint32)
        for j in range(625):
            s = np.uint32(69069) * s + np.uint32(1)
            key[j] = s
    bg = np.random.MT19937()
    st = bg.state
    st["state"]["key"] = key[1:]
    st["state"]["pos"] = 624
    bg.state = st
    return bg


def r_unif(bg) -> float:
    """MT_genrand() * 2.3283064365386963e-10, then fixup() into (0, 1)"""
    v = float(bg.random_raw()) * 2.3283064365386963e-10
    if v <= 0.0:
        return 0.5 * 2.328306437080797e-10
    if 1.0 - v <= 0.0:
        return 1.0 - 0.5 * 2.328306437080797e-10
    return v


def r_rnorm(seed: int, k: int) -> np.ndarray:
    bg = r_set_seed(seed)
    big = 134217728.0
    out = np.zeros(k)
    for i in range(k):
        u = r_unif(bg)
        u = int(big * u) + r_unif(bg)
        out[i] = ndtri(u / big)
    return out


if __name__ == "__main__":
    np.set_printoptions(precision=15)
    print("set.seed(123); rnorm(5):", r_rnorm(123, 5))
    print("set.seed(1); rnorm(9):", r_rnorm(1, 9))
