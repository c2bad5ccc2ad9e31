"""Long-running use of the library in one process: hundreds of calls of changing shapes must not grow device memory or the
host's resident set (the workspace arena caches device blocks between calls, `bssm_amd_trim_cache` returns them), and the
in-library multi-device path is exercised with as many logical shards as the 8-GPU node of the scaling bench has devices."""
import ctypes as C
import os

import numpy as np
import pytest

from bssm_amd import _lib
from bssm_amd import filters as F
from bssm_amd import models as M
from bssm_amd import post_correct as PC
from models_zoo import mng_zoo, negbin_ar1, negbin_m4_informative, poisson_local_level

pytestmark = pytest.mark.gpu


def _free_device_bytes():
    _lib.lib()
    hip = C.CDLL("libamdhip64.so.7")  # by SONAME: the runtime already in the process (bssm_amd/_lib.py), not a second one
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _rss_bytes():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def test_three_hundred_calls_leave_device_memory_and_host_rss_where_they_were():
    mng = dict(mng_zoo())["mixed_m4_p3"]
    rng = np.random.default_rng(1)

    def one_round(i):
        N = int(rng.choice([64, 300, 2048, 5000, 20000]))
        k = i % 5
        if k == 0:
            F.bootstrap_filter(negbin_ar1(25, seed=i), N, seed=i, return_indices=True)
        elif k == 1:
            F.particle_smoother(poisson_local_level(30, seed=i), N, method="psi", seed=i)
        elif k == 2:
            F.bootstrap_filter(negbin_m4_informative(20), N, seed=i)
        elif k == 3:
            F.bootstrap_filter(mng, min(N, 4096), seed=i)
        else:
            F.logLik_batch(negbin_ar1(20, seed=i), 256, 16, method="psi", seed=i)

    for i in range(40):       # warm-up: every kernel family loaded, arenas at their working size
        one_round(i)
    _lib.lib().bssm_amd_trim_cache()
    free0, rss0 = _free_device_bytes(), _rss_bytes()
    for i in range(40, 340):
        one_round(i)
    _lib.lib().bssm_amd_trim_cache()
    free1, rss1 = _free_device_bytes(), _rss_bytes()
    assert free0 - free1 < 64 << 20, (free0, free1)      # device: nothing kept beyond what trim returns
    assert rss1 - rss0 < 256 << 20, (rss0, rss1)         # host: numpy outputs come and go, no steady growth


def test_post_correction_over_eight_logical_shards_is_bit_identical():
    """eight shards inside one call (the device list of an 8-GPU node, here eight times the one GPU of the test box): eight host
    threads, eight streams, the padded exchange -- and a job whose draws do not divide by eight"""
    from oracle import oracle as O
    rng = np.random.default_rng(6)
    n, ns = 30, 67
    y = rng.poisson(4.0, n).astype(float); y[5] = np.nan
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.2, 1), sd_slope=M.halfnormal(0.02, 1), distribution="poisson", P1=np.diag([1.0, 0.1]))
    theta = np.log(np.column_stack([rng.uniform(0.1, 0.4, ns), rng.uniform(0.01, 0.05, ns)]))
    counts = rng.integers(1, 5, ns).astype(np.uint32)
    modes = np.stack([O.psi_setup(a)["mode"] for a in PC.expand_models(model, theta)], axis=1)   # (inputs only)
    for is_type, output_type in (("is2", 3), ("is1", 2), ("is3", 2)):
        mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(ns), output_type=output_type)
        one = PC.post_correct(model, mo, 64, threads=2, is_type=is_type, seed=5)
        many = PC.post_correct(model, mo, 64, threads=8, is_type=is_type, seed=5, devices=[0] * 8)
        sh = many["timing_ms"]["shards"]
        assert len(sh) == 8 and sum(s["count"] for s in sh) == len(one["weights"])
        for key in ("weights", "loglik", "posterior"):
            assert np.array_equal(one[key], many[key]), (is_type, key)
        if output_type == 2:
            assert np.array_equal(one["alphahat"], many["alphahat"]) and np.array_equal(one["Vt"], many["Vt"])
