"""Minimal reader of R's serialization format (XDR, versions 2 and 3) for the RESULT FILES the reference ships
(vignettes/psi_pf_experiments/*.rds): numeric vectors, lists / data frames, factors, ALTREP compact sequences and
wrappers.  Data only -- used by make_reference_vectors.py to quote the reference's published Monte-Carlo study."""
import bz2
import gzip
import lzma

import numpy as np

from make_datasets import Reader


class RdsReader(Reader):
    def item(self):
        p0 = self.p
        flags = self.i32()
        typ = flags & 0xFF
        if typ != 238:                       # not ALTREP: the base reader handles it
            self.p = p0
            return Reader.item(self)
        info = self.item()                   # pairlist (class symbol, package symbol, base type)
        p1 = self.p
        sflags = self.i32()
        if (sflags & 0xFF) == 2 and not (sflags & 0x600):   # wrapper state: a dotted pair CONS(x, meta)
            car = self.item()
            self.item()                                     # meta
            state = dict(value=[car], attrs={})
        else:
            self.p = p1
            state = self.item()
        attrs = self.item()
        cls = info[0][1] if isinstance(info, list) else None
        if cls in ("compact_intseq", "compact_realseq"):
            n, start, inc = [float(x) for x in state["value"]]
            v = start + inc * np.arange(int(n))
            out = dict(value=v.astype(int) if cls == "compact_intseq" else v, attrs={})
        elif cls and cls.startswith("wrap_"):
            out = state["value"][0]          # list(x, meta)
        else:
            raise ValueError(f"unsupported ALTREP class {cls}")
        if attrs:
            out = dict(value=out["value"], attrs={**out.get("attrs", {}), **dict(attrs)})
        return out


def load_rds(path):
    raw = open(path, "rb").read()
    for op in (gzip.decompress, bz2.decompress, lzma.decompress):
        try:
            raw = op(raw)
            break
        except Exception:  # noqa: BLE001
            continue
    assert raw[:2] == b"X\n", raw[:4]
    r = RdsReader(raw[2:])
    r.syms = []
    version = r.i32(); r.i32(); r.i32()
    if version == 3:                         # native encoding string
        ln = r.i32()
        r.p += ln
    return r.item()


def data_frame(obj):
    """-> dict column name -> numpy array (factors as their level strings)"""
    names = obj["attrs"]["names"]["value"]
    out = {}
    for k, col in zip(names, obj["value"]):
        v = np.asarray(col["value"])
        lev = col.get("attrs", {}).get("levels")
        out[k] = np.asarray(lev["value"], dtype=object)[v - 1] if lev is not None else v
    return out
