"""One line of throughputs for an A/B of library builds on one box (scripts/ab2.sh): the headline batch (C2-m4, informative
series, 32 in flight, factored weights), C4 at 64 and 32 filters in flight, regime B (512 LDS-resident filters, m = k = 4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from bssm_amd import _lib  # noqa: E402
from bssm_amd.plan import FilterPlan  # noqa: E402
from models_zoo import mng_c4, negbin_ar1, negbin_m4_informative  # noqa: E402

st = torch.cuda.current_stream().cuda_stream
what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["head", "c4_64", "c4_32", "regb"]
out = []


def run(model, N, R, T, reps, **kw):
    p = FilterPlan(model, N, method="bsf", n_rep=R, **kw)
    p.run(3, hip_stream=st); torch.cuda.synchronize()
    ms = p.time(5, reps, hip_stream=st)
    k = FilterPlan.last_kernel()
    p.close()
    _lib.lib().bssm_amd_trim_cache()
    return float(N) * T * R / (ms * 1e-3), k


if "head" in what:
    v, k = run(negbin_m4_informative(1000), 65536, 32, 1000, 10, keep_history=True, force="grid", factored_weights=True)
    out.append("head %.4g" % v)
for tag, R in (("c4_64", 64), ("c4_32", 32), ("c4_8", 8), ("c4_1", 1)):
    if tag in what:
        v, k = run(mng_c4(2000), 16384, R, 2000, 3, keep_history=True, force="grid", factored_weights=True)
        out.append("%s %.4g" % (tag, v))
if "regb" in what:
    v, k = run(negbin_m4_informative(1000), 2048, 512, 1000, 5, keep_history=False, force="wg")
    out.append("regb %.4g" % v)
for tag, (Nb, Rb) in (("regb1k", (1024, 1024)), ("regb512", (512, 2048)), ("regb1536", (1536, 512))):
    if tag in what:
        v, k = run(negbin_m4_informative(1000), Nb, Rb, 1000, 5, keep_history=False, force="wg")
        out.append("%s %.4g" % (tag, v))
for tag, (Nb, Rb) in (("wg8_1k", (1024, 512)), ("wg8_256", (256, 2048)), ("wg8_2k", (2048, 256)), ("wg8_64", (64, 8192)), ("wg8_512", (512, 1024))):
    if tag in what:
        v, k = run(mng_c4(500), Nb, Rb, 500, 5, keep_history=False, force="wg")
        out.append("%s %.4g" % (tag, v))
for tag, (Nb, Rb) in (("m1_1k", (1024, 1024)), ("m1_1536", (1536, 512)), ("m1_2k", (2048, 512))):
    if tag in what:
        v, k = run(negbin_ar1(1000), Nb, Rb, 1000, 5, keep_history=False, force="wg")
        out.append("%s %.4g" % (tag, v))
if any(t.startswith(("wg16", "g16")) for t in what):
    import numpy as np
    from bssm_amd import models as M
    rng = np.random.default_rng(5)
    tt = np.arange(300)
    ys = rng.poisson(np.exp(1.5 + 0.5 * np.sin(2 * np.pi * tt / 12.0))).astype(float)
    seas = M.bsm_ng(ys, sd_level=0.05, sd_slope=0.005, sd_seasonal=0.02, distribution="poisson", period=12)   # m = 13
    for tag, (Nb, Rb) in (("g16_16k", (16384, 16)), ("g16_4k", (4096, 64))):
        if tag in what:
            v, k = run(seas, Nb, Rb, 300, 3, keep_history=True, force="grid")
            out.append("%s %.4g (%s)" % (tag, v, k))
    for tag, (Nb, Rb) in (("wg16_256", (256, 2048)), ("wg16_512", (512, 1024)), ("wg16_1k", (1024, 512))):
        if tag in what:
            v, k = run(seas, Nb, Rb, 300, 5, keep_history=False, force="wg")
            out.append("%s %.4g (%s)" % (tag, v, k))
print("  ".join(out))
