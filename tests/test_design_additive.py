"""Host tests of include/RTFilterDesign.h: second-order sections, batched design, windowed-sinc FIR, and the
host execution of FILTER_SOS filters against a chain of oracle (reference-pinned) biquads.  No GPU."""
import numpy as np
import pytest

from dsp_amd import cdsp, dsp
from oracle import oracle as orc
from tests import util

SOS_CASES = [
    ("butterworth", 4, 0.3, 0.6, 0.0),       # config 2's design: 4 biquads
    ("butterworth", 2, 0.038, 0.042, 0.0),   # the demo's narrow band-pass
    ("butterworth", 5, 0.0, 0.25, 0.0),      # odd low-pass: one first-order section
    ("butterworth", 3, 0.4, 0.0, 0.0),       # high-pass
    ("butterworth", 8, 0.25, 0.75, 0.0),     # config 5's design: 8 biquads
    ("butterworth", 16, 0.19, 0.21, 0.0),    # unstable as a transfer function
    ("chebyshev1", 4, 0.2, 0.5, 0.5),
    ("chebyshev2", 5, 0.4, 0.2, 0.1),        # band-stop, zeros on the unit circle
]


def _multiply_out(secs):
    b, a = np.array([1.0]), np.array([1.0])
    for sb, sa in secs:
        b, a = np.convolve(b, sb), np.convolve(a, sa)
    return b, a


@pytest.mark.parametrize("family,order,wl,wu,ripple", SOS_CASES)
def test_sections_multiply_back_to_the_transfer_function(family, order, wl, wu, ripple):
    secs = dsp.design_sos(family, order, wl, wu, ripple)
    f = dsp.butterworth(order, wl, wu) if family == "butterworth" else getattr(dsp, family)(order, ripple, wl, wu)
    fb, fa = np.array(f.b()), np.array(f.a())
    two = wl > 0 and wu > 0
    assert len(secs) == (order if two else (order + 1) // 2)
    for sb, sa in secs:
        assert sa[0] == 1.0 and len(sb) == 3 and len(sa) == 3
    b, a = _multiply_out(secs)
    n = len(fb)
    assert not np.any(b[n:]) and not np.any(a[n:])
    # high orders: the multiplied-out coefficients themselves are only defined to ~1e-16 x their largest entry
    assert np.max(np.abs(b[:n] - fb)) <= 4e-15 * np.max(np.abs(fb))
    assert np.max(np.abs(a[:n] - fa)) <= 4e-15 * np.max(np.abs(fa))
    # every section is stable on its own
    for _, sa in secs:
        assert np.max(np.abs(np.roots(np.trim_zeros(np.array(sa), "b")))) < 1.0


def test_sos_keeps_a_design_stable_that_the_transfer_function_form_loses():
    """butterworth(8, 0.19, 0.21): as one 16th-order transfer function the rounded coefficients have a pole at
    radius 1.0028; the same design as sections has every pole inside the unit circle."""
    f = dsp.butterworth(8, 0.19, 0.21)
    assert np.max(np.abs(np.roots(np.array(f.a())))) > 1.0
    for _, sa in dsp.design_sos("butterworth", 8, 0.19, 0.21):
        assert np.max(np.abs(np.roots(sa))) < 1.0


def test_design_batch_has_the_bits_of_the_single_filter_routines():
    rng = np.random.default_rng(7)
    n = 3000
    wl = 0.05 + 0.3 * rng.random(n)
    wu = wl + 0.05 + 0.3 * rng.random(n)
    b, a, nc = dsp.design_batch("butterworth", 4, wl, wu, threads=4)
    assert np.all(nc == 9)
    for i in rng.integers(0, n, 40):
        f = dsp.butterworth(4, float(wl[i]), float(wu[i]))
        assert util.bits_equal(np.array(f.b()), b[i]) and util.bits_equal(np.array(f.a()), a[i])
    # low-pass (wl == 0), chebyshev with per-design ripple
    wu2 = 0.1 + 0.5 * rng.random(50)
    rp = 0.1 + rng.random(50)
    b, a, nc = dsp.design_batch("chebyshev1", 5, np.zeros(50), wu2, ripple=rp)
    assert np.all(nc == 6)
    for i in range(50):
        f = dsp.chebyshev1(5, float(rp[i]), 0.0, float(wu2[i]))
        assert util.bits_equal(np.array(f.b()), b[i, :6]) and util.bits_equal(np.array(f.a()), a[i, :6])
    # a bad design (wl == wu) reports status 1 like butterworth does
    with pytest.raises(ValueError):
        dsp.design_batch("butterworth", 4, [0.3, 0.2], [0.6, 0.2])


def test_design_batch_65536_designs_take_well_under_a_second():
    import time
    rng = np.random.default_rng(1)
    n = 1 << 16
    wl = 0.25 + 0.1 * rng.random(n)
    wu = wl + 0.25 + 0.05 * rng.random(n)
    t0 = time.perf_counter()
    b, a, nc = dsp.design_batch("butterworth", 4, wl, wu)
    dt = time.perf_counter() - t0
    assert np.all(nc == 9) and dt < 1.0, dt
    f = dsp.butterworth(4, float(wl[-1]), float(wu[-1]))
    assert util.bits_equal(np.array(f.b()), b[-1]) and util.bits_equal(np.array(f.a()), a[-1])


@pytest.mark.parametrize("key,taps,fc", [("fir512_lp_0.25", 512, 0.25), ("fir32_lp_0.25", 32, 0.25),
                                         ("fir7_lp_0.4", 7, 0.4)])
def test_windowed_sinc_reproduces_the_benchmark_taps(key, taps, fc):
    """The taps configs 3 and 4 use (tests/golden/designs.json, made with numpy) come out of the library's own
    routine to the last bits (libm vs numpy, normalisation order: within 1e-15 of the largest tap)."""
    _, bref, _ = util.design_coefs(key)
    g = dsp.windowed_sinc(taps, 0.0, fc, "hamming")
    b = np.array(g.b())
    assert len(b) == taps and np.max(np.abs(b - bref)) <= 1e-15 * np.max(np.abs(bref))
    assert abs(np.sum(b) - 1.0) < 1e-14


def test_windowed_sinc_forms_and_errors():
    w = np.pi * np.array([0.0, 0.1, 0.5, 0.9, 1.0])

    def mag(b):
        k = np.arange(len(b))
        return np.abs(np.exp(-1j * np.outer(w, k)) @ np.asarray(b))
    lp = mag(dsp.windowed_sinc(101, 0.0, 0.3, "blackman").b())
    assert abs(lp[0] - 1) < 1e-12 and lp[1] > 0.99 and lp[2] < 1e-3
    hp = mag(dsp.windowed_sinc(101, 0.7, 0.0, "hann").b())
    assert abs(hp[4] - 1) < 1e-12 and hp[3] > 0.99 and hp[2] < 1e-3
    bp = mag(dsp.windowed_sinc(101, 0.4, 0.6).b())
    assert abs(bp[2] - 1) < 1e-12 and bp[1] < 1e-2 and bp[3] < 1e-2
    bs = mag(dsp.windowed_sinc(101, 0.6, 0.4).b())
    assert abs(bs[0] - 1) < 1e-12 and bs[2] < 1e-2 and bs[4] > 0.98
    for args in [(0, 0.0, 0.3), (16, 0.2, 0.2), (16, 0.7, 0.0)]:      # no taps; wl == wu; even high-pass
        with pytest.raises(ValueError):
            dsp.windowed_sinc(*args)
    f = dsp.RTFIRFilter([0.0] * 8)
    assert cdsp.dspdll.fir_windowed_sinc(cdsp.byref_(f.cobj()), 16, 0.0, 0.3, 1, None) == 2     # bank too small


@pytest.mark.parametrize("family,order,wl,wu,ripple", SOS_CASES)
def test_host_sos_filter_is_a_chain_of_reference_biquads(family, order, wl, wu, ripple):
    """butterworth(..., flags=FILTER_SOS): RTFilter_updaten / RTFilter_update on the host, bit for bit the chain of
    oracle IIR filters (each pinned to the reference), blocks and single samples mixed, reset in between."""
    if family == "butterworth":
        f = dsp.butterworth(order, wl, wu, flags=dsp.FILTER_SOS)
    else:
        f = getattr(dsp, family)(order, ripple, wl, wu, flags=dsp.FILTER_SOS)
    assert f.is_sos()
    secs = f.sections()
    assert secs == dsp.design_sos(family, order, wl, wu, ripple)
    chain = orc.Cascade(secs)
    x = util.synth(99, 3, 0, 3000) * 1e3 + 5.0
    y = f(list(x[:1000])) + [f(float(v)) for v in x[1000:1100]] + f(list(x[1100:]))
    yo = chain.updaten(x)
    assert util.bits_equal(np.array(y), yo)
    st = np.array(f.state()).reshape(-1, 2)
    assert util.bits_equal(st, chain.state.reshape(-1, 3)[:, :2])
    assert f.filtered_value == chain.filtered_value and f.initialized == 0
    f.reset()
    chain.reset()
    assert util.bits_equal(np.array(f(list(x[:50]))), chain.updaten(x[:50]))


def test_sos_meets_the_bar_where_the_transfer_function_form_cannot():
    """The demo's filter, butterworth(2, 0.038, 0.042): against an 80-bit run of the same recurrence the float64
    transfer-function form is off by 1e-9..1e-7 of the output peak (conditioning); the cascade of
    its two biquads stays within 1e-12 of ITS 80-bit run."""
    x = util.synth(5, 0, 0, 1 << 15)
    f = dsp.butterworth(2, 0.038, 0.042)
    b, a = np.array(f.b()), np.array(f.a())
    y_tf = orc.iir(b, a).updaten(x)
    t_tf = orc.iir_truth_ld(b, a, x)
    secs = dsp.design_sos("butterworth", 2, 0.038, 0.042)
    y_sos = orc.Cascade(secs).updaten(x)
    t = x
    for sb, sa in secs:
        t = orc.iir_truth_ld(np.array(sb), np.array(sa), t)
    e_tf = np.max(np.abs(y_tf - t_tf)) / np.max(np.abs(t_tf))
    e_sos = np.max(np.abs(y_sos - t)) / np.max(np.abs(t))
    assert e_sos <= 1e-12 < e_tf, (e_sos, e_tf)
