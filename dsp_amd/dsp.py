"""Python face of libdsp: the callable filter objects of the reference's python/dsp.py.

Behaviour mirrored from python/dsp.py (reference file:line in each docstring):
  * RTIIRFilter(b, a) / RTFIRFilter(b) own a ctypes coefficient bank and state
    and a caller-instantiated C struct (dsp.py:15-24, 48-56);
  * calling one with a number -> RTFilter_update; with a Sequence -> one
    RTFilter_updaten, list out (None on a non-zero status); with any other
    iterable -> a lazy per-sample iterator (dsp.py:25-38);
  * design helpers size the bank like the reference does (dsp.py:79-124).

Known reference bugs are NOT reproduced: chebyshev1/2 pass `ripple` to C,
moving_average has no undefined `flags`.
"""
from collections.abc import Iterable, Sequence
from ctypes import c_double, c_size_t

from . import cdsp
from .cdsp import byref_, dspdll


class FilterIterator:
    """Lazy per-sample filtering of an iterator (python/dsp.py:6-13)."""

    def __init__(self, filter, samples):
        self.filter = filter
        self.samples = iter(samples)

    def __iter__(self):
        return self

    def __next__(self):
        return self.filter(next(self.samples))


class _RTFilterObject:
    """Shared call protocol; subclasses provide self._cstruct (first member: RTFilter)."""

    def _rtf(self):
        return byref_(self._cstruct.rtf)

    def __call__(self, value):
        if isinstance(value, Sequence):
            n = len(value)
            buf_t = c_double * n
            cin, cout = buf_t(*value), buf_t()
            status = dspdll.RTFilter_updaten(cout, self._rtf(), cin, n)
            return None if status else list(cout)
        if isinstance(value, Iterable):
            return FilterIterator(self, value)
        return dspdll.RTFilter_update(self._rtf(), value)

    def reset(self):
        dspdll.RTFilter_reset(self._rtf())

    def cobj(self):
        return self._cstruct

    @property
    def filtered_value(self):
        return self._cstruct.rtf.filtered_value

    @property
    def initialized(self):
        return self._cstruct.rtf.initialized


class RTIIRFilter(_RTFilterObject):
    """python/dsp.py:15-46.  Bank = b then a in one array; state over-sized to na+nb."""

    def __init__(self, b, a, flags=0, initialize=None):
        self.na, self.nb = len(a), len(b)
        total = self.na + self.nb
        self._b = (c_double * total)(*b, *a)
        self._state = (c_double * total)()
        self._rtif = self._cstruct = cdsp.RTIIRFilter_()
        self.initialize = initialize
        dspdll.RTIIRFilter_init(byref_(self._rtif), self._b, self._state, self.na, self.nb, flags,
                                byref_(self.initialize))
        if initialize is None:
            # RTIIRFilter_init stores the initialize argument verbatim; a NULL would be called
            # through on the first sample.  Install the library default,
            # which is what every reference design routine ends up doing (RTFilter.c:1224-1226).
            self.use_stable_init()

    def use_stable_init(self):
        self._rtif.rtf.initialize = cdsp.i_RTFilter__d(cdsp.fn_address("RTIIRFilter_stable_init"))

    def print(self):
        dspdll.RTIIRFilter_print(byref_(self._rtif))

    def b(self):
        if self.is_sos():
            raise ValueError("a FILTER_SOS filter holds packed sections, not b | a: use sections()")
        return tuple(self._b[: self._rtif.ifb.fb.nb])

    def a(self):
        if self.is_sos():
            raise ValueError("a FILTER_SOS filter holds packed sections, not b | a: use sections()")
        nb = self._rtif.ifb.fb.nb
        return tuple(self._b[nb: nb + self._rtif.ifb.na])

    def state(self):
        if self.is_sos():
            return tuple(self._state[: 2 * self._rtif.ifb.fb.nb])
        return tuple(self._state[: max(self._rtif.ifb.na, self._rtif.ifb.fb.nb)])

    def is_sos(self):
        """True for a filter designed with FILTER_SOS: the bank holds packed sections, not b | a."""
        return bool(self._rtif.rtf.flags & cdsp.FILTER_SOS) and self._rtif.ifb.na <= 1

    def sections(self):
        """[(b, a), ...] of a FILTER_SOS filter (three taps each)."""
        arr = (cdsp.RTFilterSection_ * cdsp.DSP_MAX_SECTIONS)()
        n = c_size_t()
        if dspdll.RTIIRFilter_sos_sections(byref_(self._rtif), arr, cdsp.DSP_MAX_SECTIONS, byref_(n)):
            raise ValueError("not a FILTER_SOS filter")
        return [(tuple(arr[k].b), tuple(arr[k].a)) for k in range(n.value)]


class RTFIRFilter(_RTFilterObject):
    """python/dsp.py:48-76."""

    def __init__(self, b, flags=0, initialize=None):
        self.nb = len(b)
        self._b = (c_double * self.nb)(*b)
        self._state = (c_double * self.nb)()
        self._rtff = self._cstruct = cdsp.RTFIRFilter_()
        self.initialize = initialize
        dspdll.RTFIRFilter_init(byref_(self._rtff), self._b, self._state, self.nb, flags,
                                byref_(self.initialize))
        if initialize is None:
            self.use_stable_init()

    def use_stable_init(self):
        self._rtff.rtf.initialize = cdsp.i_RTFilter__d(cdsp.fn_address("RTFIRFilter_stable_init"))

    def use_partial_init(self):
        self._rtff.rtf.initialize = cdsp.i_RTFilter__d(cdsp.fn_address("RTFIRFilter_partial_init"))

    def print(self):
        dspdll.RTFIRFilter_print(byref_(self._rtff))

    def b(self):
        return tuple(self._b[: self._rtff.ifb.nb])

    def state(self):
        return tuple(self._state[: self._rtff.ifb.nb])


def _two_edges(wlow, whigh):
    return 2 if wlow > 0.0 and whigh > 0.0 else 1


def _design(name, order, nb, na, call, flags, initialize):
    if initialize is not None:
        initialize = cdsp.i_RTFilter__d(initialize)
    rtif = RTIIRFilter([0.0] * nb, [0.0] * na, flags, initialize)
    status = call(rtif, initialize)
    if status:
        raise ValueError("%s: status %d" % (name, status))
    rtif.nb, rtif.na = rtif.cobj().ifb.fb.nb, rtif.cobj().ifb.na
    return rtif


def butterworth(order, wlow, whigh, flags=0, initialize=None):
    """python/dsp.py:79-87: bank of order*(2 if band else 1)+1 taps each for b and a."""
    n = order * _two_edges(wlow, whigh) + 1
    return _design("butterworth", order, n, n,
                   lambda f, cb: dspdll.butterworth(byref_(f.cobj()), order, wlow, whigh, flags, byref_(cb)),
                   flags, initialize)


def chebyshev1(order, ripple, wlow, whigh, flags=0, initialize=None):
    """python/dsp.py:98-106, with `ripple` actually passed to C (the reference omits it)."""
    n = order * _two_edges(wlow, whigh) + 1
    return _design("chebyshev1", order, n, n,
                   lambda f, cb: dspdll.chebyshev1(byref_(f.cobj()), order, ripple, wlow, whigh, flags, byref_(cb)),
                   flags, initialize)


def chebyshev2(order, ripple, wlow, whigh, flags=0, initialize=None):
    """python/dsp.py:108-116, with `ripple` passed."""
    n = order * _two_edges(wlow, whigh) + 1
    return _design("chebyshev2", order, n, n,
                   lambda f, cb: dspdll.chebyshev2(byref_(f.cobj()), order, ripple, wlow, whigh, flags, byref_(cb)),
                   flags, initialize)


def thiran(order, tau, flags=0, initialize=None):
    """python/dsp.py:89-96: b has one tap, a has order+1."""
    return _design("thiran", order, 1, order + 1,
                   lambda f, cb: dspdll.thiran(byref_(f.cobj()), order, tau, byref_(cb)), flags, initialize)


def moving_average(window, initialize=None):
    """python/dsp.py:118-124 (with the undefined `flags` fixed): window equal taps, ramp-in init."""
    if initialize is not None:
        initialize = cdsp.i_RTFilter__d(initialize)
    rtff = RTFIRFilter([0.0] * window, 0, initialize)
    status = dspdll.moving_average(byref_(rtff.cobj()), window, byref_(initialize))
    if status:
        raise ValueError("moving_average: status %d" % status)
    return rtff


def filter_response_pzg(poles, zeros, kgain=1, freqs=None, n=200):
    """(gain, phase, freqs) of kgain * prod (z - zero_j) / (z - pole_j) at z = exp(i f); poles and zeros as
    (re, im) pairs.  Mirror of the reference's python/dsp.py:126-144 (same default grid i/(n-1))."""
    from ctypes import c_double, c_size_t
    if freqs is None:
        df = 1.0 / (n - 1)
        freqs = list(i * df for i in range(n))
    else:
        n = len(freqs)
    c_freqs = (c_double * n)(*freqs)
    npoles, nz = len(poles), len(zeros)
    c_poles_r = (c_double * npoles)(*[p[0] for p in poles])
    c_poles_i = (c_double * npoles)(*[p[1] for p in poles])
    c_zeros_r = (c_double * nz)(*[z[0] for z in zeros])
    c_zeros_i = (c_double * nz)(*[z[1] for z in zeros])
    c_gain = (c_double * n)()
    c_phase = (c_double * n)()
    dspdll.filter_response_pzg_noc(c_gain, c_phase, c_size_t(n), c_zeros_r, c_zeros_i, c_size_t(nz), c_poles_r,
                                   c_poles_i, c_size_t(npoles), c_double(kgain), c_freqs)
    return list(c_gain), list(c_phase), freqs


# ---- additive design helpers (include/RTFilterDesign.h; no counterpart in python/dsp.py) ----------------------

FILTER_SOS = cdsp.FILTER_SOS
_FAMILIES = {"butterworth": cdsp.DSP_FAMILY_BUTTERWORTH, "chebyshev1": cdsp.DSP_FAMILY_CHEBYSHEV1,
             "chebyshev2": cdsp.DSP_FAMILY_CHEBYSHEV2}
_WINDOWS = {"rectangular": cdsp.DSP_WINDOW_RECTANGULAR, "hamming": cdsp.DSP_WINDOW_HAMMING,
            "hann": cdsp.DSP_WINDOW_HANN, "blackman": cdsp.DSP_WINDOW_BLACKMAN}


def design_sos(family, order, wlow, whigh, ripple=0.0):
    """Second-order sections [(b, a), ...] of a classical design; the gain rides on section 0."""
    arr = (cdsp.RTFilterSection_ * cdsp.DSP_MAX_SECTIONS)()
    n = c_size_t()
    status = dspdll.dsp_design_sos(_FAMILIES[family], order, ripple, wlow, whigh, arr, cdsp.DSP_MAX_SECTIONS,
                                   byref_(n))
    if status:
        raise ValueError("dsp_design_sos(%s): status %d" % (family, status))
    return [(tuple(arr[k].b), tuple(arr[k].a)) for k in range(n.value)]


def design_batch(family, order, wlow, whigh, ripple=None, threads=0):
    """Many designs of one family and order at once (host threads): arrays wlow[i], whigh[i] (and ripple[i]).
    Returns (b, a, ncoef): float64 arrays of shape (n, 2*order+1) and the taps used per design; rows hold the
    same bits butterworth / chebyshev1 / chebyshev2 produce one filter at a time."""
    import numpy as np
    wlow = np.ascontiguousarray(np.atleast_1d(wlow), dtype=np.float64)
    whigh = np.ascontiguousarray(np.atleast_1d(whigh), dtype=np.float64)
    n = len(wlow)
    assert len(whigh) == n
    ld = 2 * order + 1
    b = np.zeros((n, ld))
    a = np.zeros((n, ld))
    nc = np.zeros(n, dtype=np.uintp)
    rp = None
    if ripple is not None:
        rp = np.ascontiguousarray(np.broadcast_to(np.asarray(ripple, dtype=np.float64), (n,)))
    dp = cdsp.POINTER(c_double)
    status = dspdll.dsp_design_batch(_FAMILIES[family], n, order,
                                     None if rp is None else rp.ctypes.data_as(dp),
                                     wlow.ctypes.data_as(dp), whigh.ctypes.data_as(dp),
                                     b.ctypes.data_as(dp), a.ctypes.data_as(dp), ld,
                                     nc.ctypes.data_as(cdsp.POINTER(c_size_t)), threads)
    if status:
        raise ValueError("dsp_design_batch(%s): status %d" % (family, status))
    return b, a, nc


def windowed_sinc(taps, wlow, whigh, window="hamming", initialize=None):
    """Windowed-sinc FIR (include/RTFilterDesign.h: fir_windowed_sinc); band conventions of butterworth."""
    if initialize is not None:
        initialize = cdsp.i_RTFilter__d(initialize)
    rtff = RTFIRFilter([0.0] * taps, 0, initialize)
    status = dspdll.fir_windowed_sinc(byref_(rtff.cobj()), taps, wlow, whigh, _WINDOWS[window], byref_(initialize))
    if status:
        raise ValueError("fir_windowed_sinc: status %d" % status)
    return rtff
