"""Decode the reference's shipped data sets (data/*.rda, R serialization RDX2/XDR inside gzip/bz2)
into plain-text fixtures.  These are DATA the reference's own tests use (e.g. `data("exchange")` in
tests/testthat/test_bootstrap_filter.R:95, negbin_series in test_mcmc.R:39, poisson_series in
benchmarks/replications.Rmd), not source.  Run in the build container (needs /root/reference):

    python tests/golden/make_datasets.py
"""
import bz2
import gzip
import os
import struct
import sys

import numpy as np

REF = "/root/reference/data"
OUT = os.path.dirname(os.path.abspath(__file__))


class Reader:
    def __init__(self, b):
        self.b, self.p = b, 0

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.p)[0]; self.p += 4; return v

    def f64s(self, n):
        v = np.frombuffer(self.b, dtype=">f8", count=n, offset=self.p).astype(float); self.p += 8 * n; return v

    def i32s(self, n):
        v = np.frombuffer(self.b, dtype=">i4", count=n, offset=self.p).astype(int); self.p += 4 * n; return v

    def item(self):
        flags = self.i32()
        typ, has_attr, has_tag = flags & 0xFF, bool(flags & 0x200), bool(flags & 0x400)
        if typ == 254: return None                       # NILVALUE
        if typ == 255: return self.syms[(flags >> 8) - 1]   # REFSXP (index packed in the flags word)
        if typ == 1:                                     # SYMSXP
            name = self.item(); self.syms.append(name); return name
        if typ == 9:                                     # CHARSXP
            n = self.i32()
            if n == -1: return None
            s = self.b[self.p:self.p + n].decode("latin1"); self.p += n; return s
        if typ == 2:                                     # LISTSXP (pairlist)
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                out.append((tag, self.item()))
                flags = self.i32()
                typ, has_attr, has_tag = flags & 0xFF, bool(flags & 0x200), bool(flags & 0x400)
                if typ == 254: return out
                if typ != 2: raise ValueError(f"unexpected type {typ} in pairlist")
        if typ in (10, 13): v = self.i32s(self.i32())    # LGLSXP / INTSXP
        elif typ == 14: v = self.f64s(self.i32())        # REALSXP
        elif typ == 16: v = [self.item() for _ in range(self.i32())]  # STRSXP
        elif typ == 19: v = [self.item() for _ in range(self.i32())]  # VECSXP
        else: raise ValueError(f"unsupported SEXP type {typ}")
        attrs = dict(self.item() or []) if has_attr else {}
        return dict(value=v, attrs=attrs)

    syms = []


def load_rda(path):
    raw = open(path, "rb").read()
    for op in (gzip.decompress, bz2.decompress):
        try:
            raw = op(raw); break
        except Exception:
            continue
    assert raw[:5] == b"RDX2\n" and raw[5:7] == b"X\n", raw[:8]
    r = Reader(raw[7:])
    r.syms = []
    r.i32(); r.i32(); r.i32()  # format version, R version, min version
    return dict(r.item())


def val(o):
    return o["value"], {k: (v["value"] if isinstance(v, dict) else v) for k, v in o["attrs"].items()}


def main():
    ex, a = val(load_rda(f"{REF}/exchange.rda")["exchange"])
    np.savetxt(f"{OUT}/exchange.txt", ex, fmt="%.17g", header="bssm data(exchange): n=%d tsp=%s" % (len(ex), a.get("tsp")))
    po, a = val(load_rda(f"{REF}/poisson_series.rda")["poisson_series"])
    np.savetxt(f"{OUT}/poisson_series.txt", po, fmt="%.17g", header="bssm data(poisson_series): n=%d" % len(po))
    nb, a = val(load_rda(f"{REF}/negbin_series.rda")["negbin_series"])
    dim = a.get("dim")
    nb = np.asarray(nb, dtype=float).reshape(tuple(dim), order="F") if dim is not None else np.asarray(nb, dtype=float)
    np.savetxt(f"{OUT}/negbin_series.txt", nb, fmt="%.17g",
               header="bssm data(negbin_series): dim=%s dimnames=%s" % (dim, a.get("dimnames")))
    dr, a = val(load_rda(f"{REF}/drownings.rda")["drownings"])
    dim = a.get("dim")
    dr = np.asarray(dr, dtype=float).reshape(tuple(dim), order="F")
    np.savetxt(f"{OUT}/drownings.txt", dr, fmt="%.17g", header="bssm data(drownings): dim=%s dimnames=%s" % (dim, a.get("dimnames")))
    for f in ("exchange", "poisson_series", "negbin_series", "drownings"):
        x = np.loadtxt(f"{OUT}/{f}.txt")
        print(f, x.shape, x.ravel()[:4])


if __name__ == "__main__":
    main()
