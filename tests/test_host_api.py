"""CPU tests of the drop-in boundary: libdsp.so's host RTFilter API (include/RTFilter.h) must give the
reference's bits, keep the reference's struct layouts and error codes, and export every declared symbol.
No GPU is touched; the device API must refuse loudly when no GPU exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dsp_amd import cdsp, dsp
from dsp_amd.cdsp import dspdll
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_layouts_match_reference_abi():
    # SURVEY.md 8(a) row a9 (measured on the reference, LP64)
    assert C.sizeof(cdsp.RTFilter_) == 40
    assert [getattr(cdsp.RTFilter_, f).offset for f in
            ("update", "initialize", "del", "filtered_value", "flags", "initialized")] == [0, 8, 16, 24, 32, 36]
    assert C.sizeof(cdsp.FilterBank_) == 16 and cdsp.FilterBank_.nb.offset == 8
    assert C.sizeof(cdsp.IIRFilterBank_) == 24 and cdsp.IIRFilterBank_.na.offset == 16
    assert C.sizeof(cdsp.RTFIRFilter_) == 64
    assert (cdsp.RTFIRFilter_.ifb.offset, cdsp.RTFIRFilter_.state.offset) == (40, 56)
    assert C.sizeof(cdsp.RTIIRFilter_) == 72
    assert (cdsp.RTIIRFilter_.ifb.offset, cdsp.RTIIRFilter_.state.offset) == (40, 64)


def test_every_declared_symbol_is_exported():
    names = set()
    for hdr in ("RTFilter.h", "RTFilterBank.h", "RTFilterStream.h", "filterutils.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"EXPORT\s+[\w\s\*]+?\b(\w+)\s*\(", text))
    assert len(names) > 50
    missing = [n for n in sorted(names) if not hasattr(dspdll, n)]
    assert not missing, missing


def _mk(kind, b, a):
    if kind == "iir":
        return dsp.RTIIRFilter(list(b), list(a))
    f = dsp.RTFIRFilter(list(b))
    if kind == "fir_partial":
        f.use_partial_init()
    return f


RUNS = util.load_npz("runs.npz")
RUN_KEYS = sorted({k.split("/")[0] for k in RUNS.files})


@pytest.mark.parametrize("key", RUN_KEYS)
def test_host_updaten_bit_exact_vs_reference_golden(key):
    kind, b, a = util.design_coefs(key)
    yref = RUNS[key + "/y"]
    x = util.synth(int(RUNS[key + "/seed"][0]), 0, 0, len(yref))
    f = _mk(kind, b, a)
    pos = 0
    for cut in RUNS[key + "/cuts"]:
        cut = int(cut)
        y = f(x[pos:cut].tolist())
        assert util.bits_equal(y, yref[pos:cut]), (key, cut)
        ref_state = RUNS["%s/state@%d" % (key, cut)]
        assert util.bits_equal(f.state()[:len(ref_state)], ref_state)
        assert util.bits_equal([f.filtered_value], RUNS["%s/fv@%d" % (key, cut)])
        assert f.initialized == int(RUNS["%s/init@%d" % (key, cut)][0])
        pos = cut
    # per-sample entry == batch entry (python/dsp.py:178-179 of the reference demo)
    g = _mk(kind, b, a)
    assert util.bits_equal([g(float(v)) for v in x[:600]], yref[:600])
    # lazy iterator form (python/dsp.py:35-36)
    h = _mk(kind, b, a)
    assert util.bits_equal(list(h(iter(x[:50].tolist()))), yref[:50])
    # reset-and-reuse
    f.reset()
    x2 = util.synth(int(RUNS[key + "/seed_after_reset"][0]), 1, 0, len(RUNS[key + "/y_after_reset"]))
    assert util.bits_equal(f(x2.tolist()), RUNS[key + "/y_after_reset"])


def test_demo_config1_bitwise():
    d = util.load_npz("demo.npz")
    f = dsp.RTIIRFilter(list(d["b"]), list(d["a"]))
    assert util.bits_equal(f(d["u"].tolist()), d["y"])
    g = dsp.RTIIRFilter(list(d["b"]), list(d["a"]))
    assert util.bits_equal([g(float(v)) for v in d["u"]], d["y"])


def test_kats_through_heap_constructors():
    kats = util.load_json("kats.json")
    dp = C.POINTER(C.c_double)
    dspdll.RTIIRFilter_new.restype = C.c_void_p
    dspdll.RTIIRFilter_new.argtypes = [dp, C.c_size_t, dp, C.c_size_t, C.c_uint, C.c_void_p]
    dspdll.RTFIRFilter_new.restype = C.c_void_p
    dspdll.RTFIRFilter_new.argtypes = [dp, C.c_size_t, C.c_uint, C.c_void_p]
    dspdll.RTFilter_del.argtypes = [C.c_void_p]
    dspdll.RTFilter_del.restype = None
    upd = dspdll.RTFilter_updaten
    x = (C.c_double * 6)(*range(1, 7))
    for name, b, a in (("kat1_iir", [0.2, 0.3, 0.1], [1, -0.5, 0.25]), ("kat2_iir_a0", [0.2, 0.3, 0.1], [2, -1, 0.5]),
                       ("kat3_iir_nb_lt_na", [0.5], [1, -0.5, 0.25]),
                       ("kat4_iir_nb_gt_na", [0.1, 0.2, 0.3, 0.4], [1, -0.3])):
        ca, cb = (C.c_double * len(a))(*a), (C.c_double * len(b))(*b)
        h = dspdll.RTIIRFilter_new(ca, len(a), cb, len(b), 0, None)     # NULL -> stable_init default
        assert h
        y = (C.c_double * 6)()
        assert upd(y, C.cast(h, C.POINTER(cdsp.RTFilter_)), x, 6) == 0
        assert util.bits_equal(list(y), util.unhex(kats[name]["y"])), name
        st = C.cast(h, C.POINTER(cdsp.RTIIRFilter_)).contents
        ref_state = util.unhex(kats[name]["state"])
        assert util.bits_equal([st.state[i] for i in range(len(ref_state))], ref_state)
        dspdll.RTFilter_del(h)
    for name, b in (("kat5_fir", [0.5, 0.3, 0.2]), ("kat6_fir_sum12", [0.5, 0.3, 0.4])):
        cb = (C.c_double * len(b))(*b)
        h = dspdll.RTFIRFilter_new(cb, len(b), 0, None)
        y = (C.c_double * 6)()
        assert upd(y, C.cast(h, C.POINTER(cdsp.RTFilter_)), x, 6) == 0
        assert util.bits_equal(list(y), util.unhex(kats[name]["y"])), name
        dspdll.RTFilter_del(h)
    assert dspdll.RTIIRFilter_new(None, 0, None, 0, 0, None) is None        # na == 0 -> NULL
    assert dspdll.RTFIRFilter_new(None, 0, 0, None) is None                 # nb == 0 -> NULL


def test_moving_average_kat7_kat8_and_aliasing():
    kats = util.load_json("kats.json")
    f = dsp.RTFIRFilter([0.0] * 4)
    assert dspdll.moving_average(C.byref(f.cobj()), 3, None) == 0
    x = util.unhex(kats["kat7_mavg3_partial"]["x"])
    assert util.bits_equal([f(float(v)) for v in x], util.unhex(kats["kat7_mavg3_partial"]["y"]))
    f.reset()
    # in-place batch call (out aliases samples), allowed by the reference loop
    buf = (C.c_double * 6)(*x)
    assert dspdll.RTFilter_updaten(buf, C.byref(f.cobj().rtf), buf, 6) == 0
    assert util.bits_equal(list(buf), util.unhex(kats["kat8_mavg3_reset_stale"]["y"]))


def test_error_codes_match_reference():
    err = util.load_json("errors.json")
    buf = (C.c_double * 4)()
    f = dsp.RTIIRFilter([0.0] * 5, [0.0] * 5)
    rtf = C.byref(f.cobj().rtf)
    assert dspdll.RTFilter_updaten(None, rtf, buf, 4) == err["updaten_null_out"] == 1
    assert dspdll.RTFilter_updaten(buf, None, buf, 4) == err["updaten_null_rtf"] == 1
    assert dspdll.RTFilter_updaten(buf, rtf, None, 4) == err["updaten_null_in"] == 1
    h = dsp.RTFIRFilter([0.0] * 4)
    assert dspdll.moving_average(None, 3, None) == err["mavg_null"]
    assert dspdll.moving_average(C.byref(h.cobj()), 0, None) == err["mavg_window0"]
    assert dspdll.moving_average(C.byref(h.cobj()), 5, None) == err["mavg_too_wide"]
    assert dspdll.moving_average(C.byref(h.cobj()), 4, None) == err["mavg_ok"] == 0


def test_user_callbacks_are_honoured_by_updaten():
    """RTFilter_new with foreign callbacks: updaten must go through them (acceleration rule)."""
    calls = []

    @cdsp.i_RTFilter__d
    def my_init(rtf, x):
        rtf.contents.filtered_value = -x
        rtf.contents.initialized = 0
        calls.append("i")
        return 0

    @cdsp.i_RTFilter__d
    def my_update(rtf, x):
        rtf.contents.filtered_value = 2 * x
        calls.append("u")
        return 0

    dspdll.RTFilter_new.restype = C.POINTER(cdsp.RTFilter_)
    dspdll.RTFilter_new.argtypes = [C.c_uint, cdsp.i_RTFilter__d, cdsp.i_RTFilter__d, C.c_void_p]
    p = dspdll.RTFilter_new(7, my_init, my_update, None)
    assert p.contents.flags == 7 and p.contents.initialized == -1
    x = (C.c_double * 3)(1.0, 2.0, 3.0)
    y = (C.c_double * 3)()
    assert dspdll.RTFilter_updaten(y, p, x, 3) == 0
    assert list(y) == [-1.0, 4.0, 6.0] and calls == ["i", "u", "u"]


def test_scan_probe_separates_pinned_from_ill_conditioned_designs():
    def probe(key):
        _, b, a = util.design_coefs(key)
        d = cdsp.RTFilterDesign_()
        d.kind, d.nb, d.na = cdsp.RTFB_KIND_IIR, len(b), len(a)
        d.b = b.ctypes.data_as(C.POINTER(C.c_double))
        d.a = a.ctypes.data_as(C.POINTER(C.c_double))
        e = C.c_double()
        assert dspdll.dsp_scan_probe(C.byref(d), C.byref(e)) == 0
        return e.value
    for key in ("bw4_bp_0.3_0.6", "bw8_bp_0.25_0.75", "bw5_lp_0.25", "bw16_lp_0.5"):
        assert probe(key) < 1e-13, key
    assert probe("bw2_bp_demo") > 1e-12        # narrow band-pass: SURVEY.md 0.5
    assert probe("bw4_bp_0.2_0.3") > 1e-12


@pytest.mark.skipif(dspdll.dsp_device_count() > 0, reason="a GPU is present")
def test_device_api_fails_loudly_without_a_gpu():
    from dsp_amd import bank as gbank
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    with pytest.raises(gbank.DSPError) as ei:
        gbank.FilterBank([gbank.iir_design(b, a)], [0, 0])
    assert ei.value.status == cdsp.RTFB_EDEVICE
    from dsp_amd import stream as gstream
    with pytest.raises(gbank.DSPError) as ei:
        gstream.FilterStream(gbank.iir_design(b, a))
    assert ei.value.status == cdsp.RTFB_EDEVICE


def test_bank_table_size_check_is_host_only():
    """dsp_bank_table_doubles: the 32-bit table-offset limit of the kernels is checked on the host before a bank is
    built (no device needed): a bank of 500 000 distinct order-16 designs is refused with RTFB_ESIZE."""
    import ctypes as C
    import numpy as np
    from dsp_amd import cdsp
    b = np.ones(17)
    a = np.ones(17)

    def designs(n, taps):
        arr = (cdsp.RTFilterDesign_ * n)()
        for i in range(n):
            arr[i].kind = cdsp.RTFB_KIND_IIR
            arr[i].nb = taps
            arr[i].na = taps
            arr[i].b = b.ctypes.data_as(C.POINTER(C.c_double))
            arr[i].a = a.ctypes.data_as(C.POINTER(C.c_double))
        return arr

    out = C.c_size_t()
    # one order-8 design: A^(64 2^k) 6x64 + G64 65x8 + A^(32 2^k) 7x64 + G32 33x8 + 64 row propagators of 64
    assert cdsp.dspdll.dsp_bank_table_doubles(designs(1, 9), 1, C.byref(out)) == cdsp.RTFB_OK
    assert out.value == 6 * 64 + 65 * 8 + 7 * 64 + 33 * 8 + 64 * 64
    # more than 1024 designs: no row propagators
    assert cdsp.dspdll.dsp_bank_table_doubles(designs(2000, 9), 2000, C.byref(out)) == cdsp.RTFB_OK
    assert out.value == 2000 * (6 * 64 + 65 * 8 + 7 * 64 + 33 * 8)
    n = 500_000
    assert cdsp.dspdll.dsp_bank_table_doubles(designs(n, 17), n, C.byref(out)) == cdsp.RTFB_ESIZE
    assert out.value == n * (6 * 256 + 65 * 16 + 7 * 256 + 33 * 16) > 2 ** 31
    assert b"2^31" in cdsp.dspdll.RTFilterBank_last_error()
