"""ctypes binding of libhtk_amd.so (the C ABI of include/htk_amd.h) plus thin Python holders.

This is the host-side mirror used by the tests, bench.py and the Python drivers.  It contains no
numerics: every score, trellis and statistic comes from the HIP kernels behind the C ABI, and the
import fails loudly when the shared object is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, "libhtk_amd.so")

LZERO = -1.0e10
LSMALL = -0.5e10
NOPRUNE = 1.0e20
UPMEANS, UPVARS, UPTRANS, UPMIXES = 1, 2, 4, 8
UPALL = 15
UPMAP = 32
UTT_OK, UTT_SKIPPED, UTT_ETEE, UTT_EALPHA = 1, 0, -7332, -7390


def _stream(s):
    """A raw hipStream_t given as a Python int (torch: stream.cuda_stream) must travel as a pointer, not as a C int."""
    return None if s is None else (s if isinstance(s, C.c_void_p) else C.c_void_p(int(s)))


SCORE_EXACT, SCORE_MFMA, SCORE_FASTLADD, SCORE_BF16, SCORE_SOUTP, SCORE_DIAGC, SCORE_F16 = 0, 1, 2, 4, 8, 16, 32
ERANGE = -7
COMPAT_STREAM_REVISIT = 1
COMPAT_SHARED_LOGWT = 2
ORDER_AUTO, ORDER_FAST, ORDER_EXACT = 0, 1, 2


class HtkAmdError(RuntimeError):
    rc = 0


class ModelDesc(C.Structure):
    _fields_ = [("vecSize", C.c_int), ("numStates", C.c_int), ("numComp", C.c_int), ("numGauss", C.c_int),
                ("numTrans", C.c_int), ("numPhys", C.c_int),
                ("stateCompOff", C.c_void_p), ("compWeight", C.c_void_p), ("compGauss", C.c_void_p),
                ("mean", C.c_void_p), ("var", C.c_void_p), ("gconst", C.c_void_p),
                ("transN", C.c_void_p), ("transOff", C.c_void_p), ("transP", C.c_void_p),
                ("hmmTrans", C.c_void_p), ("hmmStateOff", C.c_void_p), ("hmmState", C.c_void_p),
                ("numStreams", C.c_int), ("dimStream", C.c_void_p), ("hsKind", C.c_int), ("streamWeight", C.c_void_p)]


class AccsLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs",
                                         "totalPr", "totalT", "nUttDone", "nUttSkipped", "nEval", "total")]


class FbConfig(C.Structure):
    _fields_ = [("pruneInit", C.c_double), ("pruneInc", C.c_double), ("pruneLim", C.c_double),
                ("minFrwdP", C.c_float), ("uFlags", C.c_int), ("scoreMode", C.c_int)]


class UpdateConfig(C.Structure):
    _fields_ = [("minEgs", C.c_int), ("minVar", C.c_float), ("mixWeightFloor", C.c_float), ("uFlags", C.c_int),
                ("singleProcess", C.c_int), ("varFloor", C.POINTER(C.c_float)), ("rowNormalise", C.c_int), ("mapTau", C.c_float), ("mapMinObs", C.c_float)]


class UpdateStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nFloorVar", "nFloorVarMix", "nSkippedHmm", "nNoTransOut", "nNoMixUse", "nNoVarUse", "nWeightAboveOne", "nMapObserved")]


class BatchDesc(C.Structure):
    _fields_ = [("nUtt", C.c_int), ("dX", C.c_void_p), ("frameOff", C.c_void_p), ("labOff", C.c_void_p), ("labs", C.c_void_p)]


_lib = None


def lib():
    """Load the native library.  There is deliberately no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise HtkAmdError("%s is missing: run `python -m htk_amd.build` (hipcc --offload-arch=gfx950)" % LIBPATH)
        L = C.CDLL(LIBPATH)
        L.htkamd_last_error.restype = C.c_char_p
        L.htkamd_fb_frame_states.restype = C.c_longlong
        L.htkamd_fb_frame_states.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        e = HtkAmdError("%s failed (%d): %s" % (what, rc, lib().htkamd_last_error().decode()))
        e.rc = rc
        raise e


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DevArray:
    """A device buffer owned through the C ABI (hipMalloc)."""

    def __init__(self, host: np.ndarray | None = None, nbytes: int | None = None):
        self.ptr = C.c_void_p()
        self.nbytes = int(host.nbytes if host is not None else nbytes)
        check(lib().htkamd_dev_malloc(C.byref(self.ptr), C.c_size_t(self.nbytes)), "dev_malloc")
        if host is not None:
            h = np.ascontiguousarray(host)
            check(lib().htkamd_memcpy_h2d(self.ptr, _p(h), C.c_size_t(h.nbytes), None), "memcpy_h2d")

    def to_host(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().htkamd_memcpy_d2h(_p(out), self.ptr, C.c_size_t(out.nbytes), None), "memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib().htkamd_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Model:
    """htkamd_model holder.  `pk` is the packed layout dict (htk_amd.synth.SynthSet.packed() / the MMF loader)."""

    def __init__(self, pk: dict):
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self.pk = pk
        self._keep = dict(stateCompOff=i32(pk["stateCompOff"]), compWeight=f32(pk["compWeight"]), compGauss=i32(pk["compGauss"]),
                          mean=f32(pk["mean"]), var=f32(pk["var"]), gconst=f32(pk.get("gconst")),
                          transN=i32(pk["transN"]), transOff=i32(pk["transOff"]), transP=f32(pk["transP"]),
                          hmmTrans=i32(pk["hmmTrans"]), hmmStateOff=i32(pk["hmmStateOff"]), hmmState=i32(pk["hmmState"]),
                          dimStream=i32(pk["dimStream"]) if pk.get("dimStream") is not None else None,
                          streamWeight=f32(pk.get("streamWeight")))
        k = self._keep
        self.NS = int(pk.get("numStreams", 1) or 1)
        d = ModelDesc(int(pk["vecSize"]), int(pk["numStates"]), int(pk["numComp"]), int(pk["numGauss"]),
                      int(pk["numTrans"]), int(pk["numPhys"]),
                      _p(k["stateCompOff"]), _p(k["compWeight"]), _p(k["compGauss"]), _p(k["mean"]), _p(k["var"]), _p(k["gconst"]),
                      _p(k["transN"]), _p(k["transOff"]), _p(k["transP"]), _p(k["hmmTrans"]), _p(k["hmmStateOff"]), _p(k["hmmState"]),
                      self.NS, _p(k["dimStream"]), int(pk.get("hsKind", 0) or 0), _p(k["streamWeight"]))
        self.h = C.c_void_p()
        check(lib().htkamd_model_create(C.byref(d), C.byref(self.h)), "model_create")
        self.D, self.S, self.C, self.G = d.vecSize, d.numStates, d.numComp, d.numGauss
        self.nT, self.H = d.numTrans, d.numPhys
        self.maxN = int(np.max(k["transN"]))

    def set_sharing(self, meanShare, varShare):
        """Tied mean / variance vectors (htkamd_model_set_sharing): arrays [G] of share numbers, -1 = private."""
        ms = np.ascontiguousarray(meanShare, np.int32); vs = np.ascontiguousarray(varShare, np.int32)
        check(lib().htkamd_model_set_sharing(self.h, _p(ms), _p(vs)), "model_set_sharing")

    def set_compat(self, flags: int):
        """The reference's own arithmetic where the library computes something else (htkamd_model_set_compat; COMPAT_STREAM_REVISIT)."""
        check(lib().htkamd_model_set_compat(self.h, C.c_int(int(flags))), "model_set_compat")

    def set_scan_order(self, order):
        """The order in which UpdateModels visits the physical models (htkamd_model_set_scan_order; hmm_scan_order() gives HTK's)."""
        o = np.ascontiguousarray(order, np.int32)
        assert o.shape == (self.H,)
        check(lib().htkamd_model_set_scan_order(self.h, _p(o)), "model_set_scan_order")

    def set_params(self, mean=None, var=None, gconst=None, compWeight=None, transP=None):
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        args = [f32(mean), f32(var), f32(gconst), f32(compWeight), f32(transP)]
        check(lib().htkamd_model_set_params(self.h, *[_p(a) for a in args]), "model_set_params")

    def get_prepared(self):
        ivar = np.empty((self.G, self.D), np.float32); gc = np.empty(self.G, np.float32)
        lw = np.empty(self.C, np.float32); md = np.empty(self.nT, np.int32)
        check(lib().htkamd_model_get_prepared(self.h, _p(ivar), _p(gc), _p(lw), _p(md)), "model_get_prepared")
        return dict(ivar=ivar, gconst=gc, compLogWt=lw, minDur=md)

    def update(self, accs: "Accs", vec: np.ndarray, minEgs=3, minVar=0.0, mixWeightFloor=0.0, uFlags=UPALL, singleProcess=False, varFloor=None,
               rowNormalise=False, mapTau=20.0, mapMinObs=0.0):
        """UpdateModels (HERest.c:1326) from a host copy of the (summed) accumulator vector.  varFloor: the ~v "varFloor1" vector.
        uFlags with UPMAP: MAPUpdateModels (HMap.c:413) with prior weight mapTau."""
        vec = np.ascontiguousarray(vec, np.float64)
        vf = None
        if varFloor is not None:
            vf = np.ascontiguousarray(varFloor, np.float32)
            assert vf.shape == (self.D,)
        cfg = UpdateConfig(minEgs, minVar, mixWeightFloor, uFlags, int(singleProcess),
                           vf.ctypes.data_as(C.POINTER(C.c_float)) if vf is not None else None, int(rowNormalise), mapTau, mapMinObs)
        st = UpdateStats()
        check(lib().htkamd_model_update(self.h, accs.h, _p(vec), C.byref(cfg), C.byref(st)), "model_update")
        return {n: getattr(st, n) for n, _ in UpdateStats._fields_}

    def update_device(self, accs: "Accs", minEgs=3, minVar=0.0, mixWeightFloor=0.0, uFlags=UPALL, singleProcess=False, varFloor=None,
                      rowNormalise=False, stream=None):
        """The same update on the device, from the accumulator vector where it lies (htkamd_model_update_device)."""
        vf = None
        if varFloor is not None:
            vf = np.ascontiguousarray(varFloor, np.float32)
            assert vf.shape == (self.D,)
        cfg = UpdateConfig(minEgs, minVar, mixWeightFloor, uFlags, int(singleProcess),
                           vf.ctypes.data_as(C.POINTER(C.c_float)) if vf is not None else None, int(rowNormalise), 0.0, 0.0)
        st = UpdateStats()
        check(lib().htkamd_model_update_device(self.h, accs.h, C.byref(cfg), C.byref(st), _stream(stream)), "model_update_device")
        return {n: getattr(st, n) for n, _ in UpdateStats._fields_}

    def update_device_begin(self, accs: "Accs", minEgs=3, minVar=0.0, mixWeightFloor=0.0, uFlags=UPALL, singleProcess=False, varFloor=None,
                            rowNormalise=False, stream=None):
        """First half of update_device: every launch and the copy of what the host needs, no wait (htkamd_model_update_device_begin)."""
        vf = None
        if varFloor is not None:
            vf = np.ascontiguousarray(varFloor, np.float32)
            assert vf.shape == (self.D,)
        cfg = UpdateConfig(minEgs, minVar, mixWeightFloor, uFlags, int(singleProcess),
                           vf.ctypes.data_as(C.POINTER(C.c_float)) if vf is not None else None, int(rowNormalise), 0.0, 0.0)
        check(lib().htkamd_model_update_device_begin(self.h, accs.h, C.byref(cfg), _stream(stream)), "model_update_device_begin")

    def update_device_end(self):
        """Second half: waits for the update's copy only and returns the counters (htkamd_model_update_device_end)."""
        st = UpdateStats()
        check(lib().htkamd_model_update_device_end(self.h, C.byref(st)), "model_update_device_end")
        return {n: getattr(st, n) for n, _ in UpdateStats._fields_}

    def get_params(self) -> dict:
        k = self._keep
        out = dict(mean=np.empty((self.G, self.D), np.float32), var=np.empty((self.G, self.D), np.float32),
                   gconst=np.empty(self.G, np.float32), compWeight=np.empty(self.C, np.float32),
                   transP=np.empty(len(k["transP"]), np.float32))
        check(lib().htkamd_model_get_params(self.h, _p(out["mean"]), _p(out["var"]), _p(out["gconst"]),
                                            _p(out["compWeight"]), _p(out["transP"])), "model_get_params")
        return out

    def outp_block(self, X: np.ndarray, states: np.ndarray, mode: int = 0) -> np.ndarray:
        """Scores [T, ns] of the listed tied states (HIP kernel K1), returned frame-major for convenience."""
        X = np.ascontiguousarray(X, np.float32); states = np.ascontiguousarray(states, np.int32)
        T, ns = X.shape[0], len(states)
        if T == 0 or ns == 0:
            return np.empty((T, ns), np.float32)
        dX, dS = DevArray(X), DevArray(states)
        dO = DevArray(nbytes=4 * T * ns)
        check(lib().htkamd_outp_block_mode(self.h, dX.ptr, C.c_int(T), dS.ptr, C.c_int(ns), dO.ptr, C.c_int(T), C.c_int(mode), None), "outp_block")
        if mode & SCORE_F16:
            check(lib().htkamd_model_f16_check(self.h, None), "model_f16_check")      # HTKAMD_ERANGE: repeat with SCORE_BF16
        out = dO.to_host(np.float32, (ns, T))
        return np.ascontiguousarray(out.T)

    def close(self):
        if self.h:
            lib().htkamd_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Accs:
    def __init__(self, model: Model):
        self.model = model
        self.h = C.c_void_p()
        check(lib().htkamd_accs_create(model.h, C.byref(self.h)), "accs_create")
        self.lay = AccsLayout()
        check(lib().htkamd_accs_get_layout(self.h, C.byref(self.lay)), "accs_get_layout")

    def zero(self, stream=None):
        check(lib().htkamd_accs_zero(self.h, _stream(stream)), "accs_zero")

    def device_vector(self):
        p = C.c_void_p(); n = C.c_size_t()
        check(lib().htkamd_accs_device_vector(self.h, C.byref(p), C.byref(n)), "accs_device_vector")
        return p.value, n.value

    def download(self) -> dict:
        v = np.empty(self.lay.total, np.float64)
        check(lib().htkamd_accs_download(self.h, _p(v), None), "accs_download")
        return self.split(v)

    def upload_add(self, vec: np.ndarray):
        vec = np.ascontiguousarray(vec, np.float64)
        assert vec.size == self.lay.total
        check(lib().htkamd_accs_upload_add(self.h, _p(vec), None), "accs_upload_add")

    def wire_round(self, stream=None):
        """One rank's share of the fp32-on-the-wire exchange (htkamd_accs_allreduce_wire, HTKAMD_WIRE_F32): the statistics rounded to float once."""
        check(lib().htkamd_accs_wire_round(self.h, _stream(stream)), "accs_wire_round")

    def state_ranges(self, state0: int, state1: int, with_rest: bool = False):
        """htkamd_accs_state_ranges: [(offset, length)] of the vector's ranges that belong to tied states [state0, state1) (+ tr / trOcc with with_rest)"""
        off = (C.c_size_t * 7)(); ln = (C.c_size_t * 7)(); n = C.c_int(0)
        check(lib().htkamd_accs_state_ranges(self.h, C.c_int(state0), C.c_int(state1), C.c_int(int(with_rest)), off, ln, C.byref(n)), "accs_state_ranges")
        return [(int(off[k]), int(ln[k])) for k in range(n.value)]

    def _ranges(self, fn, ranges, wire, ptr, stream):
        n = len(ranges)
        off = (C.c_size_t * max(n, 1))(*[r[0] for r in ranges]); ln = (C.c_size_t * max(n, 1))(*[r[1] for r in ranges])
        check(fn(self.h, C.c_int(n), off, ln, C.c_int(wire), C.c_void_p(ptr), _stream(stream)), "accs ranges")

    def pack_ranges(self, ranges, wire: int, dst_ptr: int, stream=None):
        """htkamd_accs_pack_ranges: the ranges one behind the other into device memory at dst_ptr, as floats (wire = 1) or doubles (0)"""
        self._ranges(lib().htkamd_accs_pack_ranges, ranges, wire, dst_ptr, stream)

    def unpack_ranges(self, ranges, wire: int, src_ptr: int, stream=None):
        self._ranges(lib().htkamd_accs_unpack_ranges, ranges, wire, src_ptr, stream)

    def split(self, v: np.ndarray) -> dict:
        m, L = self.model, self.lay
        GD = m.G * m.D
        return dict(vec=v, mu=v[L.mu:L.mu + GD].reshape(m.G, m.D), muOcc=v[L.muOcc:L.muOcc + m.G],
                    va=v[L.va:L.va + GD].reshape(m.G, m.D), vaOcc=v[L.vaOcc:L.vaOcc + m.G],
                    wt=v[L.wt:L.wt + m.C], wtOcc=v[L.wtOcc:L.tr], tr=v[L.tr:L.trOcc], trOcc=v[L.trOcc:L.nEgs],
                    nEgs=v[L.nEgs:L.nEgs + m.H], totalPr=v[L.totalPr], totalT=v[L.totalT], nUttDone=v[L.nUttDone],
                    nUttSkipped=v[L.nUttSkipped], nEval=v[L.nEval])

    def close(self):
        if self.h:
            lib().htkamd_accs_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fb_config(pruneInit=NOPRUNE, pruneInc=0.0, pruneLim=NOPRUNE, minFrwdP=10.0, uFlags=UPALL, scoreMode=0):
    return FbConfig(pruneInit, pruneInc, pruneLim, minFrwdP, uFlags, scoreMode)


class ForwardBackward:
    """htkamd_fb holder: FBFile (HFB.c:1923) over a batch of utterances."""

    def __init__(self, model: Model, debug: bool = False, force_general: bool = False, no_state_path: bool = False, stats_list: str = "auto", no_lr_path: bool = False):
        """stats_list: how the mixture statistics reach the accumulators -- "auto": record list + per-Gaussian reduction,
        "tiny": a 128-record list (everything beyond it takes the direct-atomics fallback), "off": direct atomics only."""
        self.model = model
        self.h = C.c_void_p()
        check(lib().htkamd_fb_create(model.h, C.byref(self.h)), "fb_create")
        flags = (1 if debug else 0) | (2 if force_general else 0) | (4 if no_state_path else 0) | {"auto": 0, "tiny": 8, "off": 16}[stats_list] | (32 if no_lr_path else 0)
        if flags:
            check(lib().htkamd_fb_set_debug(self.h, flags), "fb_set_debug")
        self.nUtt = 0
        self._keep = None

    def prepare(self, dX_ptr: int, frameOff: np.ndarray, labOff: np.ndarray, labs: np.ndarray, stream=None):
        frameOff = np.ascontiguousarray(frameOff, np.int32); labOff = np.ascontiguousarray(labOff, np.int32)
        labs = np.ascontiguousarray(labs, np.int32)
        self.nUtt = len(frameOff) - 1
        self._keep = (frameOff, labOff, labs)
        b = BatchDesc(self.nUtt, C.c_void_p(dX_ptr), _p(frameOff), _p(labOff), _p(labs))
        check(lib().htkamd_fb_prepare(self.h, C.byref(b), _stream(stream)), "fb_prepare")

    def execute(self, cfg: FbConfig, accs: Accs, stream=None):
        check(lib().htkamd_fb_execute(self.h, C.byref(cfg), accs.h, _stream(stream)), "fb_execute")

    def execute_begin(self, cfg: FbConfig, accs: Accs, stream=None) -> bool:
        """htkamd_fb_execute_begin: the pass but for the state-bucketed mixture statistics; True when those wait for execute_mix."""
        d = C.c_int(0)
        check(lib().htkamd_fb_execute_begin(self.h, C.byref(cfg), accs.h, _stream(stream), C.byref(d)), "fb_execute_begin")
        return bool(d.value)

    def execute_mix(self, state0: int, state1: int, stream=None):
        """htkamd_fb_execute_mix: the mixture statistics of tied states [state0, state1)."""
        check(lib().htkamd_fb_execute_mix(self.h, C.c_int(state0), C.c_int(state1), _stream(stream)), "fb_execute_mix")

    def results_begin(self, stream=None):
        """Queue the copy of the results behind the pass on its stream (htkamd_fb_results_begin); results() then only waits for it."""
        check(lib().htkamd_fb_results_begin(self.h, _stream(stream)), "fb_results_begin")

    def results(self, stream=None):
        pr = np.empty(self.nUtt, np.float64); st = np.empty(self.nUtt, np.int32)
        check(lib().htkamd_fb_results(self.h, _p(pr), _p(st), _stream(stream)), "fb_results")
        return pr, st

    def prepared_current(self) -> bool:
        return bool(lib().htkamd_fb_prepared_current(self.h))

    def frame_states(self) -> int:
        return int(lib().htkamd_fb_frame_states(self.h))

    def kernel_times(self):
        t = (C.c_double * 4)()
        check(lib().htkamd_fb_kernel_times(self.h, t), "fb_kernel_times")
        return list(t)

    def set_event_mode(self, mode: int):
        """htkamd_fb_set_event_mode: 0 events between all kernels (default), 1 the scoring dispatch's own only (the other intervals read -1)."""
        check(lib().htkamd_fb_set_event_mode(self.h, C.c_int(mode)), "fb_set_event_mode")

    def kernel_times5(self):
        """seconds: scoring, beta, alpha, left-to-right statistics, mixture statistics (htkamd_fb_kernel_times5)"""
        t = (C.c_double * 5)()
        check(lib().htkamd_fb_kernel_times5(self.h, t), "fb_kernel_times5")
        return list(t)

    def mix_counts(self):
        """(pairs, triples) of the last pass's mixture statistics, -1 where uncounted (htkamd_fb_mix_counts)"""
        t = (C.c_longlong * 2)()
        check(lib().htkamd_fb_mix_counts(self.h, t), "fb_mix_counts")
        return int(t[0]), int(t[1])

    def score_work(self):
        """(issued, issued without the per-state frame ranges, needed) in (wavefront, pair) units of the pair kernels (htkamd_fb_score_work)"""
        t = (C.c_longlong * 3)()
        check(lib().htkamd_fb_score_work(self.h, t), "fb_score_work")
        return int(t[0]), int(t[1]), int(t[2])

    def trellis(self, u: int, want_alpha: bool = True):
        T = C.c_int(); Q = C.c_int(); mN = C.c_int()
        check(lib().htkamd_fb_get_trellis(self.h, C.c_int(u), None, None, None, None, None, None, None,
                                          C.byref(T), C.byref(Q), C.byref(mN), None), "fb_get_trellis")
        T, Q, mN = T.value, Q.value, mN.value
        d = dict(beta=np.empty((T, Q, mN)), alpha=np.empty((T, Q, mN)) if want_alpha else None,
                 outp=np.empty((T, Q, mN), np.float32), qLo=np.zeros(T, np.int32), qHi=np.zeros(T, np.int32),
                 aLo=np.zeros(T, np.int32), aHi=np.zeros(T, np.int32))
        check(lib().htkamd_fb_get_trellis(self.h, C.c_int(u), _p(d["beta"]), _p(d["alpha"]), _p(d["outp"]),
                                          _p(d["qLo"]), _p(d["qHi"]), _p(d["aLo"]), _p(d["aHi"]), None, None, None, None),
              "fb_get_trellis")
        return d

    def close(self):
        if self.h:
            lib().htkamd_fb_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Viterbi:
    """htkamd_viterbi holder: HVite -a forced alignment of a batch (HRec token passing on the label chain)."""

    def __init__(self, model: Model):
        self.model = model
        self.h = C.c_void_p()
        check(lib().htkamd_viterbi_create(model.h, C.byref(self.h)), "viterbi_create")
        self._keep = None

    def align(self, dX_ptr: int, frameOff, labOff, labs, genBeam: float = 1.0e10, stream=None, scoreMode: int = 0):
        frameOff = np.ascontiguousarray(frameOff, np.int32); labOff = np.ascontiguousarray(labOff, np.int32)
        labs = np.ascontiguousarray(labs, np.int32)
        self._keep = (frameOff, labOff, labs)
        nUtt = len(frameOff) - 1
        b = BatchDesc(nUtt, C.c_void_p(dX_ptr), _p(frameOff), _p(labOff), _p(labs))
        check(lib().htkamd_viterbi_align_mode(self.h, C.byref(b), C.c_float(genBeam), C.c_int(scoreMode), _stream(stream)), "viterbi_align")
        ns = C.c_size_t(); nm = C.c_size_t()
        check(lib().htkamd_viterbi_sizes(self.h, C.byref(ns), C.byref(nm)), "viterbi_sizes")
        r = dict(segStart=np.empty(ns.value, np.int32), segEnd=np.empty(ns.value, np.int32), segScore=np.empty(ns.value, np.float64),
                 modStart=np.empty(nm.value, np.int32), modEnd=np.empty(nm.value, np.int32), modScore=np.empty(nm.value, np.float64),
                 total=np.empty(nUtt, np.float64), status=np.empty(nUtt, np.int32))
        check(lib().htkamd_viterbi_results(self.h, _p(r["segStart"]), _p(r["segEnd"]), _p(r["segScore"]), _p(r["modStart"]),
                                           _p(r["modEnd"]), _p(r["modScore"]), _p(r["total"]), _p(r["status"]), _stream(stream)), "viterbi_results")
        # split per utterance
        m = self.model
        k = m._keep
        out = []
        so = mo = 0
        for u in range(nUtt):
            ql = labs[labOff[u]:labOff[u + 1]]
            nst = [int(k["transN"][k["hmmTrans"][h]]) - 2 for h in ql]
            n = sum(nst)
            out.append(dict(labs=ql, nStates=nst, segStart=r["segStart"][so:so + n], segEnd=r["segEnd"][so:so + n],
                            segScore=r["segScore"][so:so + n], modStart=r["modStart"][mo:mo + len(ql)],
                            modEnd=r["modEnd"][mo:mo + len(ql)], modScore=r["modScore"][mo:mo + len(ql)],
                            total=r["total"][u], status=int(r["status"][u])))
            so += n; mo += len(ql)
        return out

    def close(self):
        if self.h:
            lib().htkamd_viterbi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_rec(utt: dict, names, frame_dur: int = 100000):
    """Label lines as HVite -a -f -m writes them: start end s<j> score [model modelscore word] (HRec.c:2300-2335)."""
    lines = []
    k = 0
    for q, (h, ns) in enumerate(zip(utt["labs"], utt["nStates"])):
        first = True
        for j in range(ns):
            if utt["segStart"][k] >= 0:
                s = "%d %d s%d %f" % (utt["segStart"][k] * frame_dur, utt["segEnd"][k] * frame_dur, j + 2, np.float32(utt["segScore"][k]))
                if first:
                    s += " %s %f %s" % (names[h], np.float32(utt["modScore"][q]), names[h])
                    first = False
                lines.append(s)
            k += 1
    return lines


class MfccConfig(C.Structure):
    _fields_ = [("sampPeriod", C.c_double), ("winDur", C.c_double), ("frPeriod", C.c_double),
                ("numChans", C.c_int), ("numCeps", C.c_int), ("cepLifter", C.c_int), ("preEmph", C.c_float),
                ("useHam", C.c_int), ("usePower", C.c_int), ("zMeanSource", C.c_int), ("rawEnergy", C.c_int), ("eNormalise", C.c_int),
                ("loFreq", C.c_float), ("hiFreq", C.c_float), ("cepScale", C.c_float), ("silFloor", C.c_float), ("eScale", C.c_float),
                ("hasC0", C.c_int), ("hasE", C.c_int), ("hasD", C.c_int), ("hasA", C.c_int), ("hasZ", C.c_int),
                ("delWin", C.c_int), ("accWin", C.c_int)]


def mfcc_config(kind="MFCC_0_D_A", sampPeriod=625.0, winDur=250000.0, frPeriod=100000.0, numChans=26, numCeps=12, cepLifter=22,
                preEmph=0.97, useHam=True, usePower=False, zMeanSource=False, rawEnergy=True, eNormalise=True,
                loFreq=-1.0, hiFreq=-1.0, cepScale=1.0, silFloor=50.0, eScale=0.1, delWin=2, accWin=2):
    """HParm configuration variables with their defaults (HParm.c:337-367); `kind` is TARGETKIND."""
    q = kind.upper().split("_")
    if q[0] != "MFCC":
        raise HtkAmdError("only MFCC target kinds are on this path")
    return MfccConfig(sampPeriod, winDur, frPeriod, numChans, numCeps, cepLifter, preEmph, int(useHam), int(usePower), int(zMeanSource),
                      int(rawEnergy), int(eNormalise), loFreq, hiFreq, cepScale, silFloor, eScale,
                      int("0" in q[1:]), int("E" in q[1:]), int("D" in q[1:]), int("A" in q[1:]), int("Z" in q[1:]), delWin, accWin)


class Mfcc:
    """htkamd_mfcc holder: waveform -> MFCC feature matrix on the device (HParm/HSigP front end)."""

    def __init__(self, cfg: MfccConfig):
        self.cfg = cfg
        self.h = C.c_void_p()
        check(lib().htkamd_mfcc_create(C.byref(cfg), C.byref(self.h)), "mfcc_create")
        self.cols = lib().htkamd_mfcc_num_cols(C.byref(cfg))

    def compute(self, waves, stream=None):
        """waves: list of int16 arrays.  Returns (DevArray features [sumT, cols], frameOff)."""
        waves = [np.ascontiguousarray(w, np.int16) for w in waves]
        sampOff = np.concatenate([[0], np.cumsum([len(w) for w in waves])]).astype(np.int32)
        allw = np.concatenate(waves) if waves else np.zeros(0, np.int16)
        frames = [lib().htkamd_mfcc_num_frames(C.byref(self.cfg), C.c_int(len(w))) for w in waves]
        total = int(sum(frames))
        dW = DevArray(allw if len(allw) else np.zeros(1, np.int16))
        dO = DevArray(nbytes=4 * max(total, 1) * self.cols)
        frameOff = np.zeros(len(waves) + 1, np.int32)
        check(lib().htkamd_mfcc_compute(self.h, dW.ptr, _p(sampOff), C.c_int(len(waves)), _p(frameOff), dO.ptr, _stream(stream)), "mfcc_compute")
        assert frameOff[-1] == total
        return dO, frameOff

    def compute_host(self, waves):
        dO, frameOff = self.compute(waves)
        return dO.to_host(np.float32, (int(frameOff[-1]), self.cols)), frameOff

    def close(self):
        if self.h:
            lib().htkamd_mfcc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _desc_from_packed(pk: dict):
    """(ModelDesc, keepalive) from a packed dict -- for the pure-host entry points that take a description."""
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    k = dict(stateCompOff=i32(pk["stateCompOff"]), compWeight=f32(pk["compWeight"]), compGauss=i32(pk["compGauss"]),
             mean=f32(pk["mean"]), var=f32(pk["var"]), gconst=f32(pk.get("gconst")),
             transN=i32(pk["transN"]), transOff=i32(pk["transOff"]), transP=f32(pk["transP"]),
             hmmTrans=i32(pk["hmmTrans"]), hmmStateOff=i32(pk["hmmStateOff"]), hmmState=i32(pk["hmmState"]),
             dimStream=i32(pk["dimStream"]) if pk.get("dimStream") is not None else None, streamWeight=f32(pk.get("streamWeight")))
    d = ModelDesc(int(pk["vecSize"]), int(pk["numStates"]), int(pk["numComp"]), int(pk["numGauss"]), int(pk["numTrans"]), int(pk["numPhys"]),
                  _p(k["stateCompOff"]), _p(k["compWeight"]), _p(k["compGauss"]), _p(k["mean"]), _p(k["var"]), _p(k["gconst"]),
                  _p(k["transN"]), _p(k["transOff"]), _p(k["transP"]), _p(k["hmmTrans"]), _p(k["hmmStateOff"]), _p(k["hmmState"]),
                  int(pk.get("numStreams", 1) or 1), _p(k["dimStream"]), int(pk.get("hsKind", 0) or 0), _p(k["streamWeight"]))
    return d, k


def _names_array(names):
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    return arr


def accs_layout(pk: dict) -> AccsLayout:
    d, keep = _desc_from_packed(pk)
    lay = AccsLayout()
    check(lib().htkamd_accs_layout_from_desc(C.byref(d), C.byref(lay)), "accs_layout_from_desc")
    return lay


def hmm_scan_order(names) -> np.ndarray:
    order = np.zeros(len(names), np.int32)
    check(lib().htkamd_hmm_scan_order(_names_array(names), C.c_int(len(names)), _p(order)), "hmm_scan_order")
    return order


def accs_dump_file(pk: dict, vec: np.ndarray, names, path: str, uFlags: int = UPALL, sharing=None):
    """sharing: (meanShare, varShare) of Mmf.sharing() for a set with ~u / ~v vectors."""
    d, keep = _desc_from_packed(pk)
    vec = np.ascontiguousarray(vec, np.float64)
    ms, vs = (None, None) if sharing is None else (np.ascontiguousarray(sharing[0], np.int32), np.ascontiguousarray(sharing[1], np.int32))
    check(lib().htkamd_accs_dump_file_shared(C.byref(d), _p(vec), _names_array(names), C.c_int(uFlags), _p(ms), _p(vs), path.encode()), "accs_dump_file")


def accs_load_file(pk: dict, vec: np.ndarray, names, path: str, uFlags: int = UPALL, sharing=None):
    d, keep = _desc_from_packed(pk)
    assert vec.dtype == np.float64 and vec.flags.c_contiguous
    ms, vs = (None, None) if sharing is None else (np.ascontiguousarray(sharing[0], np.int32), np.ascontiguousarray(sharing[1], np.int32))
    check(lib().htkamd_accs_load_file_shared(C.byref(d), _p(vec), _names_array(names), C.c_int(uFlags), _p(ms), _p(vs), path.encode()), "accs_load_file")


def stats_write_file(pk: dict, vec: np.ndarray, names, path: str):
    """HERest -s: state occupation statistics file (htkamd_stats_write_file)."""
    d, keep = _desc_from_packed(pk)
    vec = np.ascontiguousarray(vec, np.float64)
    check(lib().htkamd_stats_write_file(C.byref(d), _p(vec), _names_array(names), path.encode()), "stats_write_file")


def parm_read(path: str):
    """HTK parameter file -> (float32 [T, cols], sampPeriod, kind) through the host C reader (handles _C and _K)."""
    data = C.c_void_p(); T = C.c_int(); cols = C.c_int(); per = C.c_int(); kind = C.c_int()
    check(lib().htkamd_parm_read(path.encode(), C.byref(data), C.byref(T), C.byref(cols), C.byref(per), C.byref(kind)), "parm_read")
    n = T.value * cols.value
    arr = np.ctypeslib.as_array((C.c_float * max(n, 1)).from_address(data.value))[:n].copy().reshape(T.value, cols.value)
    lib().htkamd_free(data)
    return arr, per.value, kind.value


WAVE_HTK, WAVE_WAV = 1, 2


def wave_read(path: str, fmt: int = WAVE_WAV):
    """Waveform file -> (int16 samples, sampPeriod in 100 ns units) through the host C reader."""
    data = C.c_void_p(); n = C.c_long(); per = C.c_double()
    check(lib().htkamd_wave_read(path.encode(), C.c_int(fmt), C.byref(data), C.byref(n), C.byref(per)), "wave_read")
    arr = np.ctypeslib.as_array((C.c_short * max(n.value, 1)).from_address(data.value))[:n.value].copy()
    lib().htkamd_free(data)
    return arr, per.value


def parm_write(path: str, X: np.ndarray, sampPeriod: int, kind: int, withCrc: bool = False):
    X = np.ascontiguousarray(X, np.float32)
    check(lib().htkamd_parm_write(path.encode(), _p(X), C.c_int(X.shape[0]), C.c_int(X.shape[1]), C.c_int(sampPeriod), C.c_int(kind),
                                  C.c_int(int(withCrc))), "parm_write")


def parm_add_qualifiers(stat_list, hasD=True, hasA=False, delWin=2, accWin=2) -> "DevArray":
    """Statics of several utterances -> device table with deltas/accelerations (AddQualifiers, HParm.c:1618).
    Returns (DevArray [sum T, cols], frameOff)."""
    stat = np.ascontiguousarray(np.concatenate(stat_list), np.float32)
    frameOff = np.concatenate([[0], np.cumsum([x.shape[0] for x in stat_list])]).astype(np.int32)
    n = stat.shape[1]
    cols = n * (1 + int(hasD) + int(hasA))
    dIn = DevArray(stat)
    dOut = DevArray(nbytes=4 * max(stat.shape[0] * cols, 1))
    check(lib().htkamd_parm_add_qualifiers(dIn.ptr, _p(frameOff), C.c_int(len(stat_list)), C.c_int(n), C.c_int(int(hasD)), C.c_int(int(hasA)),
                                           C.c_int(delWin), C.c_int(accWin), dOut.ptr, None), "parm_add_qualifiers")
    return dOut, frameOff, cols


def compv(dX_ptr, nFrames: int, D: int, minVar: float = 0.0):
    """htkamd_compv: HCompV's global mean and variance of a device table [nFrames x D]."""
    mean = np.zeros(D, np.float32); var = np.zeros(D, np.float32)
    check(lib().htkamd_compv(dX_ptr, C.c_longlong(nFrames), C.c_int(D), C.c_float(minVar), _p(mean), _p(var), None), "compv")
    return mean, var


def write_vfloors(path: str, var: np.ndarray, scale: float):
    var = np.ascontiguousarray(var, np.float32)
    check(lib().htkamd_mmf_write_vfloors(path.encode(), _p(var), C.c_int(len(var)), C.c_float(scale)), "mmf_write_vfloors")


class ParmQuals(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nStat", "nZeroMean", "hasD", "hasA", "hasT", "delWin", "accWin", "thirdWin", "nullECol",
                                       "v1Compat", "simpleDiffs")]


def parm_quals_from_kind(kind: str, nStat: int, delWin=2, accWin=2, thirdWin=2, v1Compat=False, simpleDiffs=False) -> ParmQuals:
    """Qualifier step for tables of base kind + _0/_E statics (`nStat` columns) read with TARGETKIND = `kind`, e.g. "MFCC_E_D_A_N"."""
    q = kind.upper().split("_")[1:]
    nE = int("E" in q) + int("0" in q)
    base = nStat - nE
    nZ = (base + int("0" in q and "N" not in q)) if "Z" in q else 0            # HParm.c:1712-1715
    null = base if ("N" in q and nE) else -1                                   # the column after the base coefficients
    return ParmQuals(nStat, nZ, int("D" in q), int("A" in q), int("T" in q), delWin, accWin, thirdWin, null, int(v1Compat), int(simpleDiffs))


def parm_qualify(stat_list, quals: ParmQuals):
    """htkamd_parm_qualify on the statics of several utterances. Returns (DevArray [sum T, cols], frameOff, cols)."""
    stat = np.ascontiguousarray(np.concatenate(stat_list), np.float32)
    frameOff = np.concatenate([[0], np.cumsum([x.shape[0] for x in stat_list])]).astype(np.int32)
    assert stat.shape[1] == quals.nStat
    cols = lib().htkamd_parm_quals_cols(C.byref(quals))
    dIn = DevArray(stat)
    dOut = DevArray(nbytes=4 * max(stat.shape[0] * cols, 1))
    check(lib().htkamd_parm_qualify(dIn.ptr, _p(frameOff), C.c_int(len(stat_list)), C.byref(quals), dOut.ptr, None), "parm_qualify")
    return dOut, frameOff, cols


class ParmStream:
    """htkamd_parm_stream holder: the qualifier step in HParm's buffer mode (rows in pushes, observations out with qwin rows of delay)."""

    def __init__(self, quals: ParmQuals, max_rows: int):
        self.h = C.c_void_p()
        self.q = quals
        self.max_rows = max_rows
        check(lib().htkamd_parm_stream_open(C.byref(quals), C.c_int(max_rows), C.byref(self.h)), "parm_stream_open")
        self.cols = lib().htkamd_parm_quals_cols(C.byref(quals))
        self.lookahead = lib().htkamd_parm_stream_lookahead(self.h)

    def push(self, rows: np.ndarray, last: bool = False) -> np.ndarray:
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, self.q.nStat)
        dIn = DevArray(rows) if rows.shape[0] else None
        dOut = DevArray(nbytes=4 * max((rows.shape[0] + self.lookahead) * self.cols, 1))
        n = C.c_int(0)
        check(lib().htkamd_parm_stream_push(self.h, dIn.ptr if dIn else None, C.c_int(rows.shape[0]), C.c_int(int(last)), dOut.ptr, C.byref(n), None), "parm_stream_push")
        return dOut.to_host(np.float32, (max(rows.shape[0] + self.lookahead, 1), self.cols))[:n.value].copy()

    def close(self):
        if self.h:
            lib().htkamd_parm_stream_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mmf:
    """htkamd_mmf holder: LoadHMMSet / SaveHMMSet for text model definitions (htk_amd/host/mmf.c)."""

    def __init__(self, files=(), hmm_list=None, hmm_dir=None, ext=None):
        L = lib()
        L.htkamd_mmf_desc.restype = C.POINTER(ModelDesc)
        for f in ("htkamd_mmf_logical_name", "htkamd_mmf_phys_name", "htkamd_mmf_parm_kind"):
            getattr(L, f).restype = C.c_char_p
        L.htkamd_mmf_var_floor.restype = C.POINTER(C.c_float)
        self.h = C.c_void_p()
        check(L.htkamd_mmf_create(C.byref(self.h)), "mmf_create")
        for f in files:
            check(L.htkamd_mmf_read(self.h, str(f).encode(), None), "mmf_read")
        check(L.htkamd_mmf_finish(self.h, hmm_list.encode() if hmm_list else None, hmm_dir.encode() if hmm_dir else None,
                                  ext.encode() if ext else None), "mmf_finish")
        self.desc = L.htkamd_mmf_desc(self.h).contents
        self.kind = L.htkamd_mmf_parm_kind(self.h).decode()
        H = self.desc.numPhys
        self.phys_names = [L.htkamd_mmf_phys_name(self.h, C.c_int(h)).decode() for h in range(H)]
        n = L.htkamd_mmf_num_logical(self.h)
        self.logical = {L.htkamd_mmf_logical_name(self.h, C.c_int(i)).decode(): L.htkamd_mmf_logical_phys(self.h, C.c_int(i)) for i in range(n)}
        vf = L.htkamd_mmf_var_floor(self.h)
        self.var_floor = np.ctypeslib.as_array(vf, (self.desc.vecSize,)).copy() if vf else None

    def packed(self) -> dict:
        """The flat description as the dict of numpy arrays Model() takes."""
        d = self.desc
        def arr(p, n, dt):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(dt)), (max(n, 1),))[:n].copy()
        S, Cn, G, nT, H, D = d.numStates, d.numComp, d.numGauss, d.numTrans, d.numPhys, d.vecSize
        transOff = arr(d.transOff, nT + 1, C.c_int); hmmStateOff = arr(d.hmmStateOff, H + 1, C.c_int)
        pk = dict(vecSize=D, numStates=S, numComp=Cn, numGauss=G, numTrans=nT, numPhys=H,
                  stateCompOff=arr(d.stateCompOff, S * max(d.numStreams, 1) + 1, C.c_int), compWeight=arr(d.compWeight, Cn, C.c_float),
                  compGauss=arr(d.compGauss, Cn, C.c_int), mean=arr(d.mean, G * D, C.c_float).reshape(G, D),
                  var=arr(d.var, G * D, C.c_float).reshape(G, D),
                  gconst=arr(d.gconst, G, C.c_float) if d.gconst else None,
                  transN=arr(d.transN, nT, C.c_int), transOff=transOff, transP=arr(d.transP, int(transOff[-1]), C.c_float),
                  hmmTrans=arr(d.hmmTrans, H, C.c_int), hmmStateOff=hmmStateOff, hmmState=arr(d.hmmState, int(hmmStateOff[-1]), C.c_int),
                  numStreams=max(d.numStreams, 1), dimStream=arr(d.dimStream, D, C.c_int) if d.numStreams > 1 else None, hsKind=int(d.hsKind),
                  streamWeight=arr(d.streamWeight, S * d.numStreams, C.c_float) if (d.numStreams > 1 and d.streamWeight) else None)
        return pk

    def sharing(self):
        """(meanShare, varShare): per Gaussian the number of the ~u / ~v macro its mean / variance is, -1 = private; None if nothing is shared."""
        G = self.desc.numGauss
        ms = np.full(max(G, 1), -1, np.int32); vs = np.full(max(G, 1), -1, np.int32)
        n = lib().htkamd_mmf_sharing(self.h, _p(ms), _p(vs))
        return (ms[:G], vs[:G]) if n > 0 else None

    def mixup(self, target: int, states=None):
        """HHEd's MU command on the loaded set: `target` > 0 components per state (or -target more if negative) for the states
        whose indices are in `states` (None = all).  packed() / write() reflect the new set afterwards."""
        sel = None
        if states is not None:
            sel = np.zeros(self.desc.numStates, np.uint8); sel[list(states)] = 1
        check(lib().htkamd_mmf_mixup(self.h, C.c_int(target), _p(sel) if sel is not None else None), "mmf_mixup")

    def write(self, params: dict, one_file=None, out_dir=None, binary=False):
        g = params.get("gconst")
        fn = lib().htkamd_mmf_write_binary if binary else lib().htkamd_mmf_write
        check(fn(self.h, _p(np.ascontiguousarray(params["mean"], np.float32)), _p(np.ascontiguousarray(params["var"], np.float32)),
                                     _p(np.ascontiguousarray(g, np.float32)) if g is not None else None,
                                     _p(np.ascontiguousarray(params["compWeight"], np.float32)), _p(np.ascontiguousarray(params["transP"], np.float32)),
                                     one_file.encode() if one_file else None, out_dir.encode() if out_dir else None), "mmf_write")

    def write_sources(self, params: dict, master_out, out_dir=None, binary=False):
        """SaveHMMSet for a set loaded from several master files: master_out[k] = path for the k-th file read (htkamd_mmf_write_sources)."""
        g = params.get("gconst")
        arr = (C.c_char_p * len(master_out))(*[str(x).encode() for x in master_out])
        check(lib().htkamd_mmf_write_sources(self.h, _p(np.ascontiguousarray(params["mean"], np.float32)), _p(np.ascontiguousarray(params["var"], np.float32)),
                                             _p(np.ascontiguousarray(g, np.float32)) if g is not None else None,
                                             _p(np.ascontiguousarray(params["compWeight"], np.float32)), _p(np.ascontiguousarray(params["transP"], np.float32)),
                                             arr, C.c_int(len(master_out)), out_dir.encode() if out_dir else None, C.c_int(1 if binary else 0)), "mmf_write_sources")

    def close(self):
        if self.h:
            lib().htkamd_mmf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _labels_to_list(h):
    L = lib()
    L.htkamd_labels_name.restype = C.c_char_p
    L.htkamd_labels_start.restype = C.c_longlong; L.htkamd_labels_end.restype = C.c_longlong; L.htkamd_labels_score.restype = C.c_float
    n = L.htkamd_labels_count(h)
    return [(L.htkamd_labels_name(h, C.c_int(i)).decode(), L.htkamd_labels_start(h, C.c_int(i)), L.htkamd_labels_end(h, C.c_int(i)),
             L.htkamd_labels_score(h, C.c_int(i))) for i in range(n)]


def labels_read(path: str):
    """HTK label file -> [(name, start, end, score)] (htk_amd/host/labio.c)."""
    h = C.c_void_p()
    check(lib().htkamd_labels_read(path.encode(), C.byref(h)), "labels_read")
    out = _labels_to_list(h)
    lib().htkamd_labels_free(h)
    return out


OUT_FLAGS = dict(S=1, W=2, T=4, N=8, X=16, C=32, M=64)       # the letters of HVite -o


class Trans:
    """htkamd_trans holder: a transcription being written (labels with up to two auxiliary labels), HVite's -o formatting."""

    def __init__(self, max_aux: int = 0):
        self.h = C.c_void_p()
        check(lib().htkamd_trans_create(C.c_int(max_aux), C.byref(self.h)), "trans_create")

    def add(self, start, end, name, score=0.0, aux1=None, aux1_score=0.0, aux2=None, aux2_score=0.0):
        check(lib().htkamd_trans_add(self.h, C.c_double(start), C.c_double(end), name.encode(), C.c_float(score),
                                     aux1.encode() if aux1 is not None else None, C.c_float(aux1_score),
                                     aux2.encode() if aux2 is not None else None, C.c_float(aux2_score)), "trans_add")

    def format(self, frame_dur=100000.0, states=False, models=False, flags=""):
        bits = sum(OUT_FLAGS[c] for c in flags)
        check(lib().htkamd_trans_format(self.h, C.c_double(frame_dur), C.c_int(states), C.c_int(models), C.c_int(bits)), "trans_format")

    def write(self, path: str):
        check(lib().htkamd_trans_write(self.h, path.encode()), "trans_write")

    def __del__(self):
        try:
            if self.h:
                lib().htkamd_trans_free(self.h); self.h = C.c_void_p()
        except Exception:
            pass


class MlfOut:
    def __init__(self, path: str):
        self.h = C.c_void_p()
        check(lib().htkamd_mlf_out_open(path.encode(), C.byref(self.h)), "mlf_out_open")

    def add(self, lab_file: str, trans: Trans):
        check(lib().htkamd_mlf_out_add(self.h, lab_file.encode(), trans.h), "mlf_out_add")

    def close(self):
        if self.h:
            lib().htkamd_mlf_out_close(self.h); self.h = C.c_void_p()


def scp_read(path: str):
    """Script file -> [(logical, physical, start, end)] (start/end -1 = whole file); extended file names are split."""
    L = lib()
    for f in ("htkamd_scp_logical", "htkamd_scp_physical"):
        getattr(L, f).restype = C.c_char_p
    L.htkamd_scp_start.restype = C.c_long; L.htkamd_scp_end.restype = C.c_long
    h = C.c_void_p()
    check(L.htkamd_scp_read(path.encode(), C.byref(h)), "scp_read")
    out = [(L.htkamd_scp_logical(h, i).decode(), L.htkamd_scp_physical(h, i).decode(), int(L.htkamd_scp_start(h, i)), int(L.htkamd_scp_end(h, i)))
           for i in range(L.htkamd_scp_count(h))]
    L.htkamd_scp_free(h)
    return out


class Mlf:
    def __init__(self, path: str):
        self.h = C.c_void_p()
        check(lib().htkamd_mlf_read(path.encode(), C.byref(self.h)), "mlf_read")
        lib().htkamd_mlf_find.restype = C.c_void_p

    def find(self, lab_file: str):
        p = lib().htkamd_mlf_find(self.h, lab_file.encode())
        return None if not p else _labels_to_list(C.c_void_p(p))

    def __del__(self):
        try:
            if self.h:
                lib().htkamd_mlf_free(self.h); self.h = C.c_void_p()
        except Exception:
            pass


NET_ALLOWXWRDEXP, NET_FORCECXTEXP, NET_FORCELEFTBI, NET_FORCERIGHTBI = 1, 2, 4, 8


class NetDesc(C.Structure):
    _fields_ = [("nNodes", C.c_int), ("nLinks", C.c_int), ("nProns", C.c_int), ("initial", C.c_int), ("final", C.c_int),
                ("kind", C.c_void_p), ("model", C.c_void_p), ("pronProb", C.c_void_p),
                ("linkOff", C.c_void_p), ("linkDest", C.c_void_p), ("linkLike", C.c_void_p)]


class Net:
    """htkamd_net holder: SLF word network + dictionary expanded over a model set (htk_amd/host/net.c)."""

    def __init__(self, slf: str | None, dictionary: str, mmf: "Mmf", words=None, boundary: str | None = None, flags: int = 0):
        """slf: word lattice file; or slf=None and words=[...]: the alignment network of HVite -a for that transcription.
        flags: NET_ALLOWXWRDEXP | NET_FORCECXTEXP | ... (HNet's configuration switches; cross-word context expansion)."""
        L = lib()
        self._mmf = mmf                                    # a cross-word network names models through the set it was built on
        L.htkamd_net_get.restype = C.POINTER(NetDesc)
        L.htkamd_net_out_sym.restype = C.c_char_p
        self.h = C.c_void_p()
        if slf is not None:
            check(L.htkamd_net_build_ex(slf.encode(), dictionary.encode(), mmf.h, C.c_int(flags), C.byref(self.h)), "net_build")
        else:
            arr = (C.c_char_p * len(words))(*[w.encode() for w in words])
            check(L.htkamd_net_build_words(arr, C.c_int(len(words)), boundary.encode() if boundary else None, dictionary.encode(),
                                           mmf.h, C.byref(self.h)), "net_build_words")
        self.desc = L.htkamd_net_get(self.h).contents
        self.out_syms = [L.htkamd_net_out_sym(self.h, C.c_int(k)).decode() for k in range(self.desc.nProns)]
        L.htkamd_net_word_name.restype = C.c_char_p
        self.word_names = [L.htkamd_net_word_name(self.h, C.c_int(k)).decode() for k in range(self.desc.nProns)]
        self.xwrd = bool(L.htkamd_net_is_xwrd(self.h))
        self.pron_models = []
        buf = (C.c_int * 256)()
        for k in range(self.desc.nProns):
            n = L.htkamd_net_pron_models(self.h, C.c_int(k), buf, C.c_int(256))
            self.pron_models.append([int(buf[i]) for i in range(min(n, 256))])

    def seq_models(self, prons):
        """Physical models of a recognised pronunciation sequence, word by word (cross-word contexts applied between neighbours)."""
        L = lib()
        buf = (C.c_int * 256)()
        real = [k for k in prons]
        out = []
        for i, k in enumerate(real):
            prev = next((real[j] for j in range(i - 1, -1, -1) if self.pron_models[real[j]]), -1)
            nxt = next((real[j] for j in range(i + 1, len(real)) if self.pron_models[real[j]]), -1)
            n = L.htkamd_net_seq_models(self.h, C.c_int(k), C.c_int(prev), C.c_int(nxt), buf, C.c_int(256))
            if n < 0:
                raise HtkAmdError("net_seq_models: " + L.htkamd_last_error().decode())
            out.append([int(buf[j]) for j in range(min(n, 256))])
        return out

    def arrays(self) -> dict:
        d = self.desc
        def arr(p, n, dt):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(dt)), (max(n, 1),))[:n].copy()
        return dict(kind=arr(d.kind, d.nNodes, C.c_int), model=arr(d.model, d.nNodes, C.c_int), pronProb=arr(d.pronProb, d.nNodes, C.c_float),
                    linkOff=arr(d.linkOff, d.nNodes + 1, C.c_int), linkDest=arr(d.linkDest, d.nLinks, C.c_int),
                    linkLike=arr(d.linkLike, d.nLinks, C.c_float), initial=d.initial, final=d.final)

    def __del__(self):
        try:
            if self.h:
                lib().htkamd_net_destroy(self.h); self.h = C.c_void_p()
        except Exception:
            pass


class DecodeConfig(C.Structure):
    _fields_ = [("genBeam", C.c_float), ("wordBeam", C.c_float), ("lmScale", C.c_float), ("wordPen", C.c_float), ("prScale", C.c_float),
                ("scoreMode", C.c_int), ("maxActive", C.c_int)]


class LatticeOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("nNodes", "nArcs", "nodeFrame", "nodePron", "nodeNet", "nodeLike", "arcStart", "arcEnd", "arcAc", "arcLm", "arcPr",
                                          "arcScore", "total")]


class LatticeAlignOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("arcAlignOff", "alState", "alModel", "alDur", "alLike")]


class LatticeAlign(C.Structure):
    _fields_ = [("arcAlignOff", C.c_void_p), ("alState", C.c_void_p), ("alModel", C.c_void_p), ("alDur", C.c_void_p), ("alLike", C.c_void_p), ("models", C.c_int)]


class Lattice(C.Structure):
    _fields_ = [("nNodes", C.c_int), ("nArcs", C.c_int), ("nodeFrame", C.c_void_p), ("nodePron", C.c_void_p), ("nodeLike", C.c_void_p),
                ("arcStart", C.c_void_p), ("arcEnd", C.c_void_p), ("arcAc", C.c_void_p), ("arcLm", C.c_void_p), ("arcPr", C.c_void_p),
                ("lmScale", C.c_float), ("wordPen", C.c_float), ("prScale", C.c_float), ("frameDur", C.c_double)]


def _lattice_struct(lat: dict, frame_dur=0.01):
    keep = [np.ascontiguousarray(lat[k], dt) for k, dt in (("nodeFrame", np.int32), ("nodePron", np.int32), ("nodeLike", np.float64), ("arcStart", np.int32),
                                                          ("arcEnd", np.int32), ("arcAc", np.float32), ("arcLm", np.float32), ("arcPr", np.float32))]
    st = Lattice(len(keep[0]), len(keep[3]), *[_p(x) for x in keep], lat["lmScale"], lat["wordPen"], lat["prScale"], frame_dur)
    return st, keep


def lattice_write(lat: dict, net: "Net", path: str, utterance=None, lm_name=None, vocab_name=None, fmt=0, frame_dur=0.01, mmf: "Mmf" = None):
    """htkamd_lattice_write: the SLF file WriteLattice gives for the lattice (fmt: HTKAMD_LAT_* bits, 0 = t v a l; with alignment records in
    `lat` -- Decoder.run_lattice(align=...) -- and the model set `mmf` for the names: 0 = HVite's default t v a l d with durations and likelihoods)."""
    st, keep = _lattice_struct(lat, frame_dur)
    enc = lambda x: x.encode() if x is not None else None
    if "arcAlignOff" in lat and mmf is not None:
        ka = [np.ascontiguousarray(lat[k], dt) for k, dt in (("arcAlignOff", np.int32), ("alState", np.int32), ("alModel", np.int32), ("alDur", np.int32), ("alLike", np.float32))]
        al = LatticeAlign(*[_p(x) for x in ka], 1 if lat.get("alignModels") else 0)
        check(lib().htkamd_lattice_write_align(C.byref(st), C.byref(al), mmf.h, net.h, path.encode(), enc(utterance), enc(lm_name), enc(vocab_name), C.c_int(fmt or 0x3f8)), "lattice_write_align")
        return
    check(lib().htkamd_lattice_write(C.byref(st), net.h, path.encode(), enc(utterance), enc(lm_name), enc(vocab_name), C.c_int(fmt or 0x78)), "lattice_write")


def lattice_nbest(lat: dict, net: "Net", N: int, frame_dur=0.01, max_len=4096):
    """htkamd_lattice_nbest: up to N alternatives, each a list of (pron, startFrame, endFrame, score) for its arcs."""
    st, keep = _lattice_struct(lat, frame_dur)
    nAlt = C.c_int(0); altLen = np.zeros(N, np.int32); arcs = np.zeros(N * max_len, np.int32)
    check(lib().htkamd_lattice_nbest(C.byref(st), net.h, C.c_int(N), C.c_int(max_len), C.byref(nAlt), _p(altLen), _p(arcs)), "lattice_nbest")
    L = lib(); L.htkamd_lattice_arc_score.restype = C.c_float
    out = []
    for i in range(nAlt.value):
        alt = []
        for j in range(int(altLen[i])):
            a_ = int(arcs[i * max_len + j])
            alt.append((int(lat["nodePron"][lat["arcEnd"][a_]]), int(lat["nodeFrame"][lat["arcStart"][a_]]), int(lat["nodeFrame"][lat["arcEnd"][a_]]),
                        float(L.htkamd_lattice_arc_score(C.byref(st), C.c_int(a_)))))
        out.append(alt)
    return out


def lattice_nbest_align(lat: dict, net: "Net", mmf: "Mmf", N: int, states=False, models=True, frame_dur=0.01, max_len=4096, flags=""):
    """The N most likely alternatives as the label lines HVite -n N M -m / -f writes: htkamd_lattice_nbest for the arcs,
    htkamd_lattice_align_trans for the labels of the arcs' alignment records, htkamd_trans_format (-o) and the label writer."""
    import tempfile
    st, keep = _lattice_struct(lat, frame_dur)
    ka = [np.ascontiguousarray(lat[k], dt) for k, dt in (("arcAlignOff", np.int32), ("alState", np.int32), ("alModel", np.int32), ("alDur", np.int32), ("alLike", np.float32))]
    al = LatticeAlign(*[_p(x) for x in ka], 1 if lat.get("alignModels") else 0)
    nAlt = C.c_int(0); altLen = np.zeros(N, np.int32); arcs = np.zeros(N * max_len, np.int32)
    check(lib().htkamd_lattice_nbest(C.byref(st), net.h, C.c_int(N), C.c_int(max_len), C.byref(nAlt), _p(altLen), _p(arcs)), "lattice_nbest")
    out = []
    L = lib()
    for i in range(nAlt.value):
        tr = C.c_void_p()
        a_i = np.ascontiguousarray(arcs[i * max_len:i * max_len + int(altLen[i])], np.int32)
        check(L.htkamd_lattice_align_trans(C.byref(st), C.byref(al), mmf.h, net.h, _p(a_i), C.c_int(len(a_i)), C.byref(tr)), "lattice_align_trans")
        if not tr.value:
            out.append(None); continue
        check(L.htkamd_trans_format(tr, C.c_double(frame_dur * 1.0e7), C.c_int(int(states)), C.c_int(int(models)), C.c_int(sum(OUT_FLAGS[c] for c in flags))), "trans_format")
        with tempfile.NamedTemporaryFile("r", suffix=".rec") as f:
            check(L.htkamd_trans_write(tr, f.name.encode()), "trans_write")
            out.append(open(f.name).read().splitlines())
        L.htkamd_trans_free(tr)
    return out


class Decoder:
    """htkamd_decoder holder: HVite -w (1-best word labels) for a batch of utterances on the device."""

    def __init__(self, model: "Model", net: "Net", lmScale: float = 1.0):
        self.h = C.c_void_p()
        self.model, self.net, self.lmScale = model, net, float(lmScale)
        check(lib().htkamd_decoder_create(model.h, C.byref(net.desc), C.c_float(lmScale), C.byref(self.h)), "decoder_create")

    def set_order(self, mode: int):
        """htkamd_decoder_set_order: ORDER_AUTO (exact-tie utterances again in HRec's instance order), ORDER_FAST, ORDER_EXACT."""
        check(lib().htkamd_decoder_set_order(self.h, C.c_int(mode)), "decoder_set_order")

    def last_tied(self) -> int:
        return int(lib().htkamd_decoder_last_tied(self.h))

    def last_times(self):
        """(scoring ms, token-kernel ms) of the last run, by device events."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        check(lib().htkamd_decoder_last_times(self.h, C.byref(a), C.byref(b)), "decoder_last_times")
        return a.value, b.value

    def last_live(self):
        """(model-instance steps with a live token, without) of the last run's token kernel: htkamd_decoder_last_live."""
        out = (C.c_longlong * 2)()
        check(lib().htkamd_decoder_last_live(self.h, out), "decoder_last_live")
        return int(out[0]), int(out[1])

    def run(self, feats, genBeam=1.0e10, wordBeam=1.0e10, lmScale=None, wordPen=0.0, prScale=1.0, maxWords=1024, scoreMode=0, maxActive=0):
        """feats: list of [T, D] arrays.  Returns per utterance (list of (pron, startFrame, endFrame, score) or None, total)."""
        lmScale = self.lmScale if lmScale is None else float(lmScale)
        if lmScale != self.lmScale:
            raise HtkAmdError("Decoder.run: the LM scale is fixed at creation (LikeToWord look-ahead)")
        nU = len(feats)
        X = np.ascontiguousarray(np.concatenate(feats) if nU else np.zeros((0, self.model.D)), np.float32)
        frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int32)
        dX = DevArray(X) if X.size else DevArray(nbytes=4)
        nW = np.zeros(max(nU, 1), np.int32); tot = np.zeros(max(nU, 1), np.float64)
        wp = np.zeros(max(nU, 1) * maxWords, np.int32); ws = np.zeros_like(wp); we = np.zeros_like(wp); sc = np.zeros(max(nU, 1) * maxWords, np.float32)
        lm = np.zeros_like(sc)
        cfg = DecodeConfig(genBeam, wordBeam, lmScale, wordPen, prScale, scoreMode, int(maxActive))
        check(lib().htkamd_decoder_run(self.h, C.byref(cfg), dX.ptr, _p(frameOff), C.c_int(nU), C.c_int(maxWords), _p(nW), _p(wp), _p(ws), _p(we),
                                       _p(sc), _p(lm), _p(tot), None), "decoder_run")
        self.last_lm = [[float(lm[u * maxWords + i]) for i in range(max(int(nW[u]), 0))] for u in range(nU)]
        out = []
        for u in range(nU):
            if nW[u] < 0:
                out.append((None, float(tot[u])))
            else:
                o = u * maxWords
                out.append(([(int(wp[o + i]), int(ws[o + i]), int(we[o + i]), float(sc[o + i])) for i in range(nW[u])], float(tot[u])))
        return out

    def run_lattice(self, feats, nToks, genBeam=1.0e10, wordBeam=1.0e10, nBeam=None, lmScale=None, wordPen=0.0, prScale=1.0, maxNodes=20000, maxArcs=80000, scoreMode=0, maxActive=0,
                    align=0, maxAlign=400000):
        """HVite -n nToks [-u maxActive]: per utterance the lattice as a dict of arrays (oracle.decode_nbest's fields + nodePron), or None."""
        lmScale = self.lmScale if lmScale is None else float(lmScale)
        if lmScale != self.lmScale:
            raise HtkAmdError("Decoder.run_lattice: the LM scale is fixed at creation (LikeToWord look-ahead)")
        nU = len(feats)
        X = np.ascontiguousarray(np.concatenate(feats) if nU else np.zeros((0, self.model.D)), np.float32)
        frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int32)
        dX = DevArray(X) if X.size else DevArray(nbytes=4)
        n1 = max(nU, 1)
        nn = np.zeros(n1, np.int32); na = np.zeros(n1, np.int32); tot = np.zeros(n1, np.float64)
        nF = np.zeros(n1 * maxNodes, np.int32); nP = np.zeros_like(nF); nNet = np.zeros_like(nF); nL = np.zeros(n1 * maxNodes, np.float64)
        aS = np.zeros(n1 * maxArcs, np.int32); aE = np.zeros_like(aS); aAc = np.zeros(n1 * maxArcs, np.float32); aLm = np.zeros_like(aAc); aPr = np.zeros_like(aAc)
        aSc = np.zeros(n1 * maxArcs, np.float64)
        out = LatticeOut(_p(nn), _p(na), _p(nF), _p(nP), _p(nNet), _p(nL), _p(aS), _p(aE), _p(aAc), _p(aLm), _p(aPr), _p(aSc), _p(tot))
        cfg = DecodeConfig(genBeam, wordBeam, lmScale, wordPen, prScale, scoreMode, int(maxActive))
        if align:                                            # HVite -n with -m (1) / -f (2): lAlign of every arc
            aOff = np.zeros(n1 * (maxArcs + 1), np.int32); alS = np.zeros(n1 * maxAlign, np.int32); alM = np.zeros_like(alS); alD = np.zeros_like(alS)
            alL = np.zeros(n1 * maxAlign, np.float32)
            alo = LatticeAlignOut(_p(aOff), _p(alS), _p(alM), _p(alD), _p(alL))
            check(lib().htkamd_decoder_run_lattice_align(self.h, C.byref(cfg), C.c_int(nToks), C.c_float(genBeam if nBeam is None else nBeam), C.c_int(int(align)), dX.ptr,
                                                         _p(frameOff), C.c_int(nU), C.c_int(maxNodes), C.c_int(maxArcs), C.c_int(maxAlign), C.byref(out), C.byref(alo), None),
                  "decoder_run_lattice_align")
        else:
            check(lib().htkamd_decoder_run_lattice(self.h, C.byref(cfg), C.c_int(nToks), C.c_float(genBeam if nBeam is None else nBeam), dX.ptr, _p(frameOff), C.c_int(nU),
                                                   C.c_int(maxNodes), C.c_int(maxArcs), C.byref(out), None), "decoder_run_lattice")
        res = []
        for u in range(nU):
            if nn[u] == -1:
                res.append(None); continue
            if nn[u] < 0:
                raise HtkAmdError("decoder_run_lattice: the lattice of utterance %d does not fit (%d)" % (u, nn[u]))
            n, a_, o, q = int(nn[u]), int(na[u]), u * maxNodes, u * maxArcs
            res.append(dict(nodeFrame=nF[o:o + n].copy(), nodePron=nP[o:o + n].copy(), nodeNet=nNet[o:o + n].copy(), nodeLike=nL[o:o + n].copy(),
                            arcStart=aS[q:q + a_].copy(), arcEnd=aE[q:q + a_].copy(), arcAc=aAc[q:q + a_].copy(), arcLm=aLm[q:q + a_].copy(), arcPr=aPr[q:q + a_].copy(),
                            arcScore=aSc[q:q + a_].copy(), total=float(tot[u]), lmScale=lmScale, wordPen=float(wordPen), prScale=float(prScale)))
            if align:
                off = aOff[u * (maxArcs + 1):u * (maxArcs + 1) + a_ + 1].copy(); k = int(off[-1]); b = u * maxAlign
                res[-1].update(arcAlignOff=off, alState=alS[b:b + k].copy(), alModel=alM[b:b + k].copy(), alDur=alD[b:b + k].copy(), alLike=alL[b:b + k].copy(),
                               alignModels=bool(align & 1))
        return res

    def __del__(self):
        try:
            if self.h:
                lib().htkamd_decoder_destroy(self.h); self.h = C.c_void_p()
        except Exception:
            pass
