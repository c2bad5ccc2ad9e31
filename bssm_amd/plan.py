"""Device-resident filter plan (bssm_plan_* in include/bssm_amd.h): fixed models, only the seed changes.
Used by benchmarks and by callers that filter the same model repeatedly (pseudo-marginal MCMC)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .models import model_arrays


class FilterPlan:
    def __init__(self, models, particles: int, method: str = "bsf", n_rep: int = 1, keep_history: bool = False,
                 force: str | None = None, factored_weights: bool = False):
        if not isinstance(models, (list, tuple)):
            models = [models]
        self._keep = []
        packed = []
        for mdl in models:
            s, keep = _lib.pack_model(model_arrays(mdl))
            packed.append(s)
            self._keep.append(keep)
        arr_t = _lib.UngModel * len(packed)
        self._models = arr_t(*packed)
        flags = (_lib.PLAN_KEEP_HISTORY if keep_history else 0)
        flags |= {"grid": _lib.PLAN_FORCE_GRID, "wg": _lib.PLAN_FORCE_WG, None: 0}[force]
        # opt-in: weights kept as block-scaled value x per-(time point, block) factor (BSSM_PLAN_FACTORED_WEIGHTS); fetch()
        # multiplies them out, weights_layout() says whether the last run produced that form
        flags |= _lib.PLAN_FACTORED_WEIGHTS if factored_weights else 0
        L = _lib.lib()
        self._h = L.bssm_plan_create(self._models, len(packed), n_rep, particles, 1 if method == "psi" else 2, flags)
        if not self._h:
            raise _lib.BssmAmdError(L.bssm_amd_last_error().decode())
        self.n_filters = len(packed) * n_rep
        self.particles = particles
        self.n = models[0].n
        self.m = models[0].m

    def update(self, models, hip_stream: int = 0):
        """New parameter values for the same model structure (the theta proposal of a PM / DA MCMC iteration): the
        tables are rebuilt and copied into the plan's existing device buffers (bssm_plan_update)."""
        if not isinstance(models, (list, tuple)):
            models = [models]
        keep, packed = [], []
        for mdl in models:
            s, k = _lib.pack_model(model_arrays(mdl))
            packed.append(s); keep.append(k)
        arr = (_lib.UngModel * len(packed))(*packed)
        L = _lib.lib()
        L.bssm_plan_update.argtypes = [C.c_void_p, C.POINTER(_lib.UngModel), C.c_void_p]
        _lib.check(L.bssm_plan_update(self._h, arr, C.c_void_p(hip_stream)))

    def sample_or_summarise(self, filter_index: int = 0, seed: int = 1, sample: bool = True, hip_stream: int = 0):
        """src/summary.cpp:82-103 on the kept particle system (needs keep_history): a uniformly drawn back-traced
        trajectory (n+1) x m, or (alphahat m x (n+1), Vt m x m x (n+1))."""
        L = _lib.lib()
        L.bssm_plan_sample_or_summarise.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, _lib.c_double_p,
                                                    _lib.c_double_p, _lib.c_double_p]
        n, m = self.n, self.m
        if sample:
            alpha = np.zeros((n + 1, m), order="F")
            _lib.check(L.bssm_plan_sample_or_summarise(self._h, filter_index, seed, C.c_void_p(hip_stream), _lib.dptr(alpha), None, None))
            return alpha
        ah, vt = np.zeros((m, n + 1), order="F"), np.zeros((m, m, n + 1), order="F")
        _lib.check(L.bssm_plan_sample_or_summarise(self._h, filter_index, seed, C.c_void_p(hip_stream), None, _lib.dptr(ah), _lib.dptr(vt)))
        return ah, vt

    def info(self):
        path, nf, blocks, threads = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        bps = C.c_double()
        _lib.check(_lib.lib().bssm_plan_info(self._h, C.byref(path), C.byref(nf), C.byref(blocks), C.byref(threads),
                                              C.byref(bps)))
        return dict(path="grid" if path.value else "wg", n_filters=nf.value, blocks=blocks.value,
                    threads=threads.value, bytes_per_particle_step=bps.value)

    def run(self, seed: int, stream0: int = 0, hip_stream: int = 0):
        _lib.check(_lib.lib().bssm_plan_run(self._h, seed, stream0, C.c_void_p(hip_stream)))

    def loglik(self, hip_stream: int = 0):
        out = np.zeros(self.n_filters)
        _lib.check(_lib.lib().bssm_plan_loglik(self._h, C.c_void_p(hip_stream), _lib.dptr(out)))
        return out

    def fetch(self, filter_index: int, hip_stream: int = 0, alpha: bool = True, indices: bool = True, weights: bool = True):
        """Kept outputs of one filter of the plan after run() (needs keep_history): alpha (n+1, m, N), indices (n, N) --
        row t = ancestors of the particles of slot t+1 --, weights (n+1, N), as numpy arrays (bssm_plan_fetch)."""
        L = _lib.lib()
        L.bssm_plan_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, _lib.c_double_p, _lib.c_u32_p, _lib.c_double_p]
        n, m, N = self.n, self.m, self.particles
        a = np.empty((n + 1, m, N)) if alpha else None
        ix = np.empty((n, N), dtype=np.uint32) if indices else None
        w = np.empty((n + 1, N)) if weights else None
        _lib.check(L.bssm_plan_fetch(self._h, filter_index, C.c_void_p(hip_stream), _lib.dptr(a),
                                     None if ix is None else ix.ctypes.data_as(_lib.c_u32_p), _lib.dptr(w)))
        return dict(alpha=a, indices=ix, weights=w)

    def weights_layout(self):
        """dict(factored, block_particles, n_blocks) of the weights the last run() left on the device"""
        L = _lib.lib()
        L.bssm_plan_weights_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_void_p)]
        fa, bp, nb, sc = C.c_int32(), C.c_int32(), C.c_int32(), C.c_void_p()
        _lib.check(L.bssm_plan_weights_layout(self._h, C.byref(fa), C.byref(bp), C.byref(nb), C.byref(sc)))
        return dict(factored=bool(fa.value), block_particles=bp.value, n_blocks=nb.value)

    def time(self, seed: int, reps: int, hip_stream: int = 0) -> float:
        ms = C.c_double()
        _lib.check(_lib.lib().bssm_plan_time(self._h, seed, reps, C.c_void_p(hip_stream), C.byref(ms)))
        return ms.value

    def time_kernel(self, seed: int, sample_every: int = 16, hip_stream: int = 0):
        """(average duration in ms of the dominant kernel's launches, number of sampled launches)"""
        ms, cnt = C.c_double(), C.c_int32()
        _lib.check(_lib.lib().bssm_plan_time_kernel(self._h, seed, sample_every, C.c_void_p(hip_stream), C.byref(ms),
                                                     C.byref(cnt)))
        return ms.value, cnt.value

    def launch_sequences(self) -> int:
        """concurrent step-launch sequences of the last time_kernel run (each timed launch covers n_filters / this)"""
        L = _lib.lib()
        L.bssm_plan_launch_sequences.argtypes = [C.c_void_p]
        return int(L.bssm_plan_launch_sequences(self._h))

    @staticmethod
    def last_kernel() -> str:
        """the filter kernel variant this thread launched last, as named by the launcher"""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().bssm_amd_last_kernel(buf, C.c_int32(128)))
        return buf.value.decode()

    @staticmethod
    def family_launches() -> dict:
        """filter batches launched so far by kernel family, process-wide (post-correction filters on a worker thread)"""
        c = [C.c_uint64(0) for _ in range(3)]
        _lib.check(_lib.lib().bssm_amd_family_launches(*[C.byref(x) for x in c]))
        return dict(wg=c[0].value, step=c[1].value, row=c[2].value)

    def close(self):
        if self._h:
            _lib.lib().bssm_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
