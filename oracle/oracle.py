"""numpy/ctypes front end of oracle/liboracle.so (our CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  dsp_amd/ never imports this package.
"""
import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")

KIND_IIR, KIND_FIR = 0, 1
INIT_IIR_STABLE, INIT_FIR_STABLE, INIT_FIR_PARTIAL = 0, 1, 2

_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_zp = C.POINTER(C.c_size_t)


class _Chan(C.Structure):
    _fields_ = [("state", _dp), ("filtered_value", C.c_double), ("initialized", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        L = C.CDLL(_PATH)
        L.orc_updaten.argtypes = [C.c_int, C.c_int, _dp, C.c_size_t, _dp, C.c_size_t,
                                  C.POINTER(_Chan), _dp, _dp, C.c_size_t]
        L.orc_updaten.restype = C.c_int
        L.orc_bank_updaten.argtypes = [C.c_size_t, _ip, _ip, _ip, _zp, _zp, _zp, _dp,
                                       _dp, C.c_size_t, _dp, _ip,
                                       _dp, C.c_size_t, _dp, C.c_size_t,
                                       C.c_size_t, C.c_size_t, C.c_size_t]
        L.orc_bank_updaten.restype = C.c_int
        L.orc_iir_updaten_ld.argtypes = [_dp, C.c_size_t, _dp, C.c_size_t, C.c_void_p, _ip,
                                         _dp, _dp, C.c_size_t]
        L.orc_iir_updaten_ld.restype = None
        L.orc_synth.argtypes = [_dp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t]
        L.orc_synth.restype = None
        _lib = L
    return _lib


def _p(arr):
    return arr.ctypes.data_as(_dp)


class Filter:
    """One channel: coefficient bank + the three mutable fields of RTFilter."""

    def __init__(self, kind, b, a=None, init_mode=None):
        self.kind = kind
        self.b = np.ascontiguousarray(b, dtype=np.float64)
        self.a = np.ascontiguousarray(a if a is not None else [1.0], dtype=np.float64)
        if init_mode is None:
            init_mode = INIT_IIR_STABLE if kind == KIND_IIR else INIT_FIR_STABLE
        self.init_mode = init_mode
        n = max(len(self.a), len(self.b)) if kind == KIND_IIR else len(self.b)
        self.state = np.zeros(n, dtype=np.float64)
        self.filtered_value = 0.0
        self.initialized = -1

    def reset(self):
        """RTFilter_reset semantics: state untouched (src/RTFilter.c:55-58)."""
        self.initialized = -1
        self.filtered_value = 0.0

    def updaten(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        ch = _Chan(_p(self.state), self.filtered_value, self.initialized)
        rc = lib().orc_updaten(self.kind, self.init_mode, _p(self.b), len(self.b),
                               _p(self.a), len(self.a), C.byref(ch), _p(y), _p(x), len(x))
        assert rc == 0
        self.filtered_value = ch.filtered_value
        self.initialized = ch.initialized
        return y

    def snapshot(self):
        return dict(state=self.state.copy(), filtered_value=float(self.filtered_value),
                    initialized=int(self.initialized))


def iir(b, a):
    return Filter(KIND_IIR, b, a)


def fir(b, partial=False):
    return Filter(KIND_FIR, b, None, INIT_FIR_PARTIAL if partial else INIT_FIR_STABLE)


class Cascade:
    """Second-order sections in series: every section is an oracle IIR filter (nb = na = 3, the reference's
    RTIIRFilter_update / RTIIRFilter_stable_init, each pinned to the reference) fed with the output of the
    section before it.  The reference has no SOS execution (its FILTER_SOS flag is never read): this chain of
    reference filters is the definition the product's RTFB_KIND_SOS path is held to."""

    def __init__(self, sections):
        self.sections = [iir(np.asarray(b, dtype=np.float64), np.asarray(a, dtype=np.float64)) for b, a in sections]

    def reset(self):
        for s in self.sections:
            s.reset()

    def updaten(self, x):
        y = np.ascontiguousarray(x, dtype=np.float64)
        for s in self.sections:
            y = s.updaten(y)
        return y

    @property
    def state(self):
        """three slots per section, section after section (the layout of RTFilterBank_get_state)"""
        return np.concatenate([s.state for s in self.sections])

    @property
    def filtered_value(self):
        return self.sections[-1].filtered_value

    @property
    def initialized(self):
        return self.sections[0].initialized


class Bank:
    """Flat multichannel state: nch independent channels, each a reference filter.

    designs: list of dicts {kind, init_mode, b, a}; design_idx[c] picks one.
    """

    def __init__(self, designs, design_idx):
        self.nd = len(designs)
        self.kind = np.array([d["kind"] for d in designs], dtype=np.int32)
        self.init_mode = np.array([d["init_mode"] for d in designs], dtype=np.int32)
        self.nb = np.array([len(d["b"]) for d in designs], dtype=np.uintp)
        self.na = np.array([len(d.get("a", [])) if d["kind"] == KIND_IIR else 0 for d in designs],
                           dtype=np.uintp)
        offs, chunks, o = [], [], 0
        for d in designs:
            offs.append(o)
            bb = np.asarray(d["b"], dtype=np.float64)
            aa = np.asarray(d.get("a", []), dtype=np.float64) if d["kind"] == KIND_IIR else np.zeros(0)
            chunks += [bb, aa]
            o += len(bb) + len(aa)
        self.coef_off = np.array(offs, dtype=np.uintp)
        self.coefs = np.concatenate(chunks) if chunks else np.zeros(0)
        self.design = np.ascontiguousarray(design_idx, dtype=np.int32)
        self.nch = len(self.design)
        self.state_ld = int(max(max(int(self.nb[i]), int(self.na[i])) for i in range(self.nd)))
        self.state = np.zeros((self.nch, self.state_ld), dtype=np.float64)
        self.fval = np.zeros(self.nch, dtype=np.float64)
        self.init = np.full(self.nch, -1, dtype=np.int32)

    def reset(self):
        self.fval[:] = 0.0
        self.init[:] = -1

    def _run(self, out, x, n, lo, hi):
        rc = lib().orc_bank_updaten(
            self.nch, self.design.ctypes.data_as(_ip), self.kind.ctypes.data_as(_ip),
            self.init_mode.ctypes.data_as(_ip), self.nb.ctypes.data_as(_zp),
            self.na.ctypes.data_as(_zp), self.coef_off.ctypes.data_as(_zp), _p(self.coefs),
            _p(self.state), self.state_ld, _p(self.fval), self.init.ctypes.data_as(_ip),
            _p(out), out.strides[0] // 8, _p(x), x.strides[0] // 8, n, lo, hi)
        assert rc == 0

    def updaten(self, x, threads=1):
        """x: (nch, n) float64, rows contiguous (row stride free).  Returns (nch, n)."""
        assert x.dtype == np.float64 and x.shape[0] == self.nch and x.strides[1] == 8
        n = x.shape[1]
        out = np.empty((self.nch, n), dtype=np.float64)
        if threads <= 1 or self.nch < 2:
            self._run(out, x, n, 0, self.nch)
        else:
            threads = min(threads, self.nch)
            cuts = [self.nch * t // threads for t in range(threads + 1)]
            with ThreadPoolExecutor(threads) as ex:  # ctypes drops the GIL during the call
                list(ex.map(lambda t: self._run(out, x, n, cuts[t], cuts[t + 1]), range(threads)))
        return out


def iir_truth_ld(b, a, x):
    """Same recurrence in 80-bit long double (reports the float64 rounding floor)."""
    b = np.ascontiguousarray(b, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    st = (C.c_longdouble * (max(len(a), len(b)) + 1))()
    ini = C.c_int(-1)
    lib().orc_iir_updaten_ld(_p(b), len(b), _p(a), len(a), C.cast(st, C.c_void_p), C.byref(ini),
                             _p(y), _p(x), len(x))
    return y


def synth(seed, chan, n0, n):
    out = np.empty(n, dtype=np.float64)
    lib().orc_synth(_p(out), seed, chan, n0, n)
    return out
