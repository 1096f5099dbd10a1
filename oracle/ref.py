"""ctypes driver for the REAL reference library (oracle/_ref/libdsp_ref.so).

TEST INFRASTRUCTURE ONLY.  The shared object is compiled by `make -C oracle ref`
from the sources under /root/reference (never copied into this repo).  It is
used to (1) capture the golden fixtures under tests/golden/ and (2) pin
oracle/rtfilter_oracle.c bit-for-bit.  It exists only where /root/reference was
present at build time; everything that imports this module must cope with
`available() == False`.

Structure layouts follow include/RTFilter.h:30-59 of the reference (sizes in
SURVEY.md 8(a) row a9); they are re-declared here, not imported from the
reference's python/cdsp.py (which cannot be imported on Linux).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libdsp_ref.so")

_lib = None


class RTFilter(C.Structure):
    _fields_ = [("update", C.c_void_p), ("initialize", C.c_void_p), ("del_", C.c_void_p),
                ("filtered_value", C.c_double), ("flags", C.c_uint), ("initialized", C.c_int)]


class FilterBank(C.Structure):
    _fields_ = [("b", C.POINTER(C.c_double)), ("nb", C.c_size_t)]


class IIRFilterBank(C.Structure):
    _fields_ = [("fb", FilterBank), ("na", C.c_size_t)]


class RTIIRFilter(C.Structure):
    _fields_ = [("rtf", RTFilter), ("ifb", IIRFilterBank), ("state", C.POINTER(C.c_double))]


class RTFIRFilter(C.Structure):
    _fields_ = [("rtf", RTFilter), ("fb", FilterBank), ("state", C.POINTER(C.c_double))]


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        dp = C.POINTER(C.c_double)
        L.RTFilter_update.argtypes = [C.c_void_p, C.c_double]
        L.RTFilter_update.restype = C.c_double
        L.RTFilter_updaten.argtypes = [dp, C.c_void_p, dp, C.c_size_t]
        L.RTFilter_updaten.restype = C.c_int
        L.RTFilter_reset.argtypes = [C.c_void_p]
        L.RTFilter_reset.restype = None
        L.RTIIRFilter_init.argtypes = [C.c_void_p, dp, dp, C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p]
        L.RTIIRFilter_init.restype = None
        L.RTFIRFilter_init.argtypes = [C.c_void_p, dp, dp, C.c_size_t, C.c_uint, C.c_void_p]
        L.RTFIRFilter_init.restype = None
        for name in ("butterworth",):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_uint, C.c_void_p]
            getattr(L, name).restype = C.c_int
        for name in ("chebyshev1", "chebyshev2"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                         C.c_uint, C.c_void_p]
            getattr(L, name).restype = C.c_int
        L.thiran.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]
        L.thiran.restype = C.c_int
        L.moving_average.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.moving_average.restype = C.c_int
        _lib = L
    return _lib


def _fn_addr(name):
    return C.cast(getattr(lib(), name), C.c_void_p).value


class RefIIR:
    """A caller-owned RTIIRFilter (like python/dsp.py:16-24 builds one)."""

    def __init__(self, b, a, initialize="stable"):
        b = np.asarray(b, dtype=np.float64)
        a = np.asarray(a, dtype=np.float64)
        self.nb, self.na = len(b), len(a)
        self.bank = np.concatenate([b, a]).copy()
        self.state = np.zeros(max(self.na, self.nb) + 1, dtype=np.float64)
        self.s = RTIIRFilter()
        init = _fn_addr("RTIIRFilter_stable_init") if initialize == "stable" else None
        lib().RTIIRFilter_init(C.byref(self.s), self.bank.ctypes.data_as(C.POINTER(C.c_double)),
                               self.state.ctypes.data_as(C.POINTER(C.c_double)),
                               self.na, self.nb, 0, init)

    @property
    def nstate(self):
        return max(self.s.ifb.na, self.s.ifb.fb.nb)

    def b(self):
        return self.bank[: self.s.ifb.fb.nb].copy()

    def a(self):
        nb = self.s.ifb.fb.nb
        return self.bank[nb: nb + self.s.ifb.na].copy()

    def updaten(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        rc = lib().RTFilter_updaten(y.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.s),
                                    x.ctypes.data_as(C.POINTER(C.c_double)), len(x))
        assert rc == 0
        return y

    def update(self, v):
        return lib().RTFilter_update(C.byref(self.s), float(v))

    def reset(self):
        lib().RTFilter_reset(C.byref(self.s))

    def snapshot(self):
        return dict(state=self.state[: self.nstate].copy(),
                    filtered_value=float(self.s.rtf.filtered_value),
                    initialized=int(self.s.rtf.initialized))


class RefFIR:
    def __init__(self, b, initialize="stable"):
        b = np.asarray(b, dtype=np.float64)
        self.nb = len(b)
        self.bank = b.copy()
        self.state = np.zeros(self.nb + 1, dtype=np.float64)
        self.s = RTFIRFilter()
        init = {"stable": "RTFIRFilter_stable_init", "partial": "RTFIRFilter_partial_init"}[initialize]
        lib().RTFIRFilter_init(C.byref(self.s), self.bank.ctypes.data_as(C.POINTER(C.c_double)),
                               self.state.ctypes.data_as(C.POINTER(C.c_double)),
                               self.nb, 0, _fn_addr(init))

    @property
    def nstate(self):
        return self.s.fb.nb

    def b(self):
        return self.bank[: self.s.fb.nb].copy()

    updaten = RefIIR.updaten
    update = RefIIR.update
    reset = RefIIR.reset
    snapshot = RefIIR.snapshot


def design_iir(kind, order, wl=0.0, wu=0.0, ripple=None, tau=None):
    """Run a reference design routine on a zeroed bank sized as python/dsp.py:79-96 does."""
    if kind == "thiran":
        f = RefIIR([0.0], [0.0] * (order + 1))
        rc = lib().thiran(C.byref(f.s), order, tau, None)
    else:
        mult = 2 if (wl > 0.0 and wu > 0.0) else 1
        z = [0.0] * (order * mult + 1)
        f = RefIIR(z, z)
        if kind == "butterworth":
            rc = lib().butterworth(C.byref(f.s), order, wl, wu, 0, None)
        else:
            rc = getattr(lib(), kind)(C.byref(f.s), order, ripple, wl, wu, 0, None)
    return rc, f
