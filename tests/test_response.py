"""Frequency-response helpers (include/filterutils.h): the symbols the reference's own ctypes binding needs at
import time.  filter_response_pzg(_noc) is pinned bit for bit by vectors captured from the reference
(tests/golden/response.json, explicit grids: the reference's default-grid path overruns its heap block);
filter_response_ab(_noc) is a documented correction (the reference returns 1+0i / dereferences NULL), checked
against the textbook formula."""
import ctypes as C

import numpy as np

from dsp_amd import cdsp, dsp
from dsp_amd.cdsp import dspdll
from tests import util

DP = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(DP)


def test_pzg_noc_bit_exact_vs_reference_vectors():
    cases = util.load_json("response.json")
    for name, c in cases.items():
        if name == "errors":
            continue
        zr, zi = util.unhex(c["zeros_r"]), util.unhex(c["zeros_i"])
        pr, pi = util.unhex(c["poles_r"]), util.unhex(c["poles_i"])
        freq = util.unhex(c["freq"])
        ng = c["ng"]
        gain, phase = np.zeros(ng), np.zeros(ng)
        rc = dspdll.filter_response_pzg_noc(_p(gain), _p(phase), ng, _p(zr), _p(zi), len(zr), _p(pr), _p(pi), len(pr),
                                            float.fromhex(c["kgain"]), _p(freq))
        assert rc == c["status"] == 0
        assert util.bits_equal(gain, util.unhex(c["gain"])), name
        assert util.bits_equal(phase, util.unhex(c["phase"])), name
    e = cases["errors"]
    g1 = np.zeros(1)
    assert dspdll.filter_response_pzg_noc(None, _p(g1), 1, None, None, 0, None, None, 0, 1.0, None) == e["null_gain"] == 1
    assert dspdll.filter_response_pzg_noc(_p(g1), _p(g1), 0, None, None, 0, None, None, 0, 1.0, None) == e["ng0"] == 0


def test_pzg_default_grid_is_the_accumulated_one_and_python_mirror():
    zr, zi = np.array([0.3, -0.2]), np.array([0.1, 0.4])
    pr, pi = np.array([0.5, 0.5, -0.6]), np.array([0.2, -0.2, 0.0])
    ng = 25
    g0, p0, g1, p1 = (np.zeros(ng) for _ in range(4))
    assert dspdll.filter_response_pzg_noc(_p(g0), _p(p0), ng, _p(zr), _p(zi), 2, _p(pr), _p(pi), 3, 2.0, None) == 0
    grid = np.zeros(ng)
    df = 1.0 / (ng - 1)
    for i in range(1, ng):
        grid[i] = grid[i - 1] + df                       # ref: src/filterutils.c:57-61
    assert dspdll.filter_response_pzg_noc(_p(g1), _p(p1), ng, _p(zr), _p(zi), 2, _p(pr), _p(pi), 3, 2.0, _p(grid)) == 0
    assert util.bits_equal(g0, g1) and util.bits_equal(p0, p1)
    gain, phase, freqs = dsp.filter_response_pzg(list(zip(pr, pi)), list(zip(zr, zi)), kgain=2.0, n=ng)
    z = np.exp(1j * np.array(freqs))
    h = 2.0 * np.prod([z - complex(a, b) for a, b in zip(zr, zi)], axis=0) / np.prod(
        [z - complex(a, b) for a, b in zip(pr, pi)], axis=0)
    assert np.allclose(gain, np.abs(h), rtol=1e-12) and np.allclose(phase, np.angle(h), atol=1e-12)


def test_ab_noc_evaluates_the_transfer_function():
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    freq = np.linspace(0.01, np.pi - 0.01, 64)
    gain, phase = np.zeros(64), np.zeros(64)
    assert dspdll.filter_response_ab_noc(_p(gain), _p(phase), 64, _p(a), len(a), _p(b), len(b), _p(freq)) == 0
    zinv = np.exp(-1j * freq)
    h = np.polyval(b[::-1], zinv) / np.polyval(a[::-1], zinv)
    assert np.allclose(gain, np.abs(h), rtol=1e-10, atol=1e-13)
    assert np.allclose(np.exp(1j * phase), np.exp(1j * np.angle(h)), atol=1e-9)
    mid = np.argmin(np.abs(freq - 0.45 * np.pi))        # butterworth(4, 0.3, 0.6): unit gain inside the band
    assert abs(gain[mid] - 1.0) < 0.05
    assert dspdll.filter_response_ab_noc(None, _p(phase), 4, _p(a), len(a), _p(b), len(b), None) == 1
    assert dspdll.filter_response_ab_noc(_p(gain), _p(phase), 0, _p(a), len(a), _p(b), len(b), None) == 0
    assert dspdll.filter_response_ab_noc(_p(gain), _p(phase), 8, _p(a), len(a), _p(b), len(b), None) == 0   # own grid


def test_pzg_to_rtiirfilter_reports_failure_like_the_reference():
    fn = dspdll.pzg_to_RTIIRFilter
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double]
    assert fn(None, None, 0, None, 0, 1.0, 0.1, 0.2) == 1
    f = dsp.butterworth(2, 0.1, 0.2)
    before = (list(f.b()), list(f.a()))
    poles = (C.c_double * 4)(0.1, 0.2, 0.1, -0.2)
    assert fn(C.addressof(f.cobj()), poles, 2, None, 0, 1.0, 0.1, 0.2) == 1
    assert (list(f.b()), list(f.a())) == before          # untouched
