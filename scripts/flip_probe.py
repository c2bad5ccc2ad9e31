"""WHERE do the device's and the oracle's ancestor tables differ at the metric's full size?  (round-3 review, item 1)

For every checked (seed, slot) pair of the very plan bench.py times: number of differing entries, the first differing
(t, slot), the per-column counts of the first columns that differ, whether every difference sits in the last column, the
number of differing entries one step after the first one, |dlogLik|, and whether the states / weights in front of the first
difference agree.  Writes gpurun_out/flip_probe.json.

    python scripts/flip_probe.py [--pairs 8] [--reference-layout]
"""
import argparse
import ctypes as C
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--particles", type=int, default=65536)
    ap.add_argument("--replicas", type=int, default=32)
    ap.add_argument("--reference-layout", action="store_true")
    ap.add_argument("--slots", type=str, default="0,15,16,31")
    ap.add_argument("--seeds", type=str, default="20011,20012")
    ap.add_argument("--series", choices=["model", "informative"], default="model")
    ap.add_argument("--out", type=str, default="flip_probe.json")
    args = ap.parse_args()
    import torch
    from bssm_amd import models as M
    from bssm_amd.plan import FilterPlan
    from models_zoo import negbin_m4, negbin_m4_informative
    from oracle import oracle as O

    N, T, R = args.particles, args.T, args.replicas
    model = (negbin_m4 if args.series == "model" else negbin_m4_informative)(T, seed=103)
    arr = M.model_arrays(model)
    stream = torch.cuda.current_stream().cuda_stream
    plan = FilterPlan(model, N, method="bsf", keep_history=True, force="grid", n_rep=R, factored_weights=not args.reference_layout)
    slots = [int(s) for s in args.slots.split(",")]
    seeds = [int(s) for s in args.seeds.split(",")]
    plan.run(3, hip_stream=stream); torch.cuda.synchronize()
    ms_per_pass = plan.time(5, 5, hip_stream=stream)
    dev = {}
    for sd in seeds:
        plan.run(sd, stream0=0, hip_stream=stream)
        ll = plan.loglik(stream)
        for f in slots:
            got = plan.fetch(f, hip_stream=stream)
            dev[(sd, f)] = (float(ll[f]), got["indices"], got["alpha"], got["weights"])
    kernel = FilterPlan.last_kernel()
    plan.close()

    def one(key):
        sd, f = key
        ref = O.bsf_filter(arr, N, seed=sd, stream=f)
        gl, gix, ga, gw = dev[key]
        rix = ref["indices"].T                       # (n, N)
        neq = gix != rix
        per_col = neq.sum(axis=1)
        cols = np.flatnonzero(per_col)
        rec = {"seed": sd, "slot": f, "dloglik": gl - float(ref["logLik"]), "differing_entries": int(neq.sum()),
               "differing_columns": int(cols.size)}
        if cols.size:
            t0 = int(cols[0])
            j0 = int(np.flatnonzero(neq[t0])[0])
            rec.update(first_t=t0, first_slot=j0, first_dev=int(gix[t0, j0]), first_ref=int(rix[t0, j0]),
                       all_in_last_column=bool(cols[0] == T - 1), next_step_differing=int(per_col[t0 + 1]) if t0 + 1 < T else None,
                       first_columns=[[int(c), int(per_col[c])] for c in cols[:12]],
                       last_columns=[[int(c), int(per_col[c])] for c in cols[-4:]])
            # the two candidate ancestors: are their states the same particle (a duplicate)?
            a_dev = ga[t0, :, gix[t0, j0]]
            a_ref = ga[t0, :, rix[t0, j0]]
            rec["ancestor_states_equal"] = bool(np.array_equal(a_dev, a_ref))
            rec["ancestor_weights"] = [float(gw[t0, gix[t0, j0]]), float(gw[t0, rix[t0, j0]])]
            # how close the slot's target lies to the CDF entry between the two candidates, on the oracle's side, in units
            # of one stratum (1 / N): the reference's arithmetic (normalise, sequential cumsum, (j + u) / N)
            wcol = ref["weights"][:, t0]
            cdf = np.cumsum(wcol / wcol.sum())
            u = O.lib().orc_rng_uniform(C.byref(O.make_rng(sd, f)[0]), 1, t0, j0, 4)
            kk = min(int(gix[t0, j0]), int(rix[t0, j0]))
            rec["resampling_uniform"] = float(u)
            rec["margin_in_strata"] = float(((j0 + u) / N - cdf[kk]) * N)
            rec["column_weight_spread"] = [float(wcol.min()), float(wcol.max())]
            rec["differing_entries_by_step_after_first"] = [int(v) for v in per_col[t0:t0 + 6]]
            # how the differences are distributed over slots in the first differing column
            js = np.flatnonzero(neq[t0])
            rec["first_col_slots"] = [int(j) for j in js[:16]]
            rec["first_col_pairs"] = [[int(gix[t0, j]), int(rix[t0, j])] for j in js[:16]]
        ra = np.transpose(ref["alpha"], (1, 0, 2))
        T1 = (int(cols[0]) + 1) if cols.size else T + 1
        rec["alpha_max_abs_diff_before"] = float(np.max(np.abs(ga[:T1] - ra[:T1])))
        rec["alpha_max_abs_diff_all"] = float(np.max(np.abs(ga - ra)))
        rw = ref["weights"].T
        rec["weights_max_rel_diff_all"] = float(np.max(np.abs(gw - rw) / (1e-300 + np.abs(rw))))
        return rec

    keys = sorted(dev)
    with ThreadPoolExecutor(max_workers=min(len(keys), max(1, (os.cpu_count() or 1) // 2))) as ex:
        recs = list(ex.map(one, keys))
    out = {"kernel": kernel, "N": N, "T": T, "R": R, "factored": not args.reference_layout, "series": args.series,
           "ms_per_pass": ms_per_pass, "particle_steps_per_s": N * float(T) * R / (ms_per_pass * 1e-3), "pairs": recs}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", args.out), "w") as fh:
        json.dump(out, fh, indent=1)
    print("series", args.series, "ms_per_pass", ms_per_pass, "p-steps/s %.4g" % out["particle_steps_per_s"])
    for r in recs:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
