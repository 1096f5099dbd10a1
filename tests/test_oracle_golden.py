"""Pins oracle/rtfilter_oracle.c bit-for-bit against vectors captured from the reference.

Golden files were written by tests/golden/make_golden.py from oracle/_ref/libdsp_ref.so
(the reference compiled from /root/reference).  CPU-only.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import ref
from tests import util


def _mk(kind, b, a):
    if kind == "iir":
        return orc.iir(b, a)
    return orc.fir(b, partial=(kind == "fir_partial"))


KATS = util.load_json("kats.json")


@pytest.mark.parametrize("name,kind,b,a", [
    ("kat1_iir", "iir", [0.2, 0.3, 0.1], [1, -0.5, 0.25]),
    ("kat2_iir_a0", "iir", [0.2, 0.3, 0.1], [2, -1, 0.5]),
    ("kat3_iir_nb_lt_na", "iir", [0.5], [1, -0.5, 0.25]),
    ("kat4_iir_nb_gt_na", "iir", [0.1, 0.2, 0.3, 0.4], [1, -0.3]),
    ("kat5_fir", "fir", [0.5, 0.3, 0.2], None),
    ("kat6_fir_sum12", "fir", [0.5, 0.3, 0.4], None),
])
def test_kat(name, kind, b, a):
    k = KATS[name]
    f = _mk(kind, b, a)
    y = f.updaten(util.unhex(k["x"]))
    assert util.bits_equal(y, util.unhex(k["y"]))
    assert util.bits_equal(f.state, util.unhex(k["state"]))
    assert f.initialized == k["initialized"]
    assert util.bits_equal([f.filtered_value], util.unhex([k["filtered_value"]]))


def test_kat_survey_table_values():
    # literal values quoted in SURVEY.md 8(a) (independent of the json capture)
    y = orc.iir([0.2, 0.3, 0.1], [1, -0.5, 0.25]).updaten(np.arange(1.0, 7.0))
    assert y.tolist() == [1.0, 1.05, 1.5750000000000002, 2.425, 3.3187499999999996, 4.153125]
    y = orc.iir([0.1, 0.2, 0.3, 0.4], [1, -0.3]).updaten(np.arange(1.0, 7.0))
    assert y.tolist() == [0.42857142857142866, 1.2285714285714284, 1.7685714285714285,
                          2.5305714285714287, 3.7591714285714284, 5.127751428571429]
    y = orc.fir([0.5, 0.3, 0.4]).updaten(np.arange(1.0, 7.0))
    assert y.tolist() == [1.0, 1.7, 2.5, 3.7, 4.9, 6.1]


def test_kat7_kat8_moving_average_partial_and_stale_reset():
    b = util.design_coefs("mavg3")[1]
    f = orc.fir(b, partial=True)
    x = util.unhex(KATS["kat7_mavg3_partial"]["x"])
    y7 = np.concatenate([f.updaten(x[i:i + 1]) for i in range(len(x))])  # per-sample like the KAT
    assert util.bits_equal(y7, util.unhex(KATS["kat7_mavg3_partial"]["y"]))
    assert y7.tolist() == [1.0, 1.5, 2.0, 3.0, 3.9999999999999996, 5.0]
    f.reset()
    y8 = f.updaten(x)
    assert util.bits_equal(y8, util.unhex(KATS["kat8_mavg3_reset_stale"]["y"]))
    assert y8.tolist() == [1.0, 4.5, 2.0, 3.0, 3.9999999999999996, 5.0]


RUNS = util.load_npz("runs.npz")
RUN_KEYS = sorted({k.split("/")[0] for k in RUNS.files})


@pytest.mark.parametrize("key", RUN_KEYS)
def test_runs_full_snapshots_split_and_reset(key):
    kind, b, a = util.design_coefs(key)
    seed = int(RUNS[key + "/seed"][0])
    yref = RUNS[key + "/y"]
    n = len(yref)
    x = util.synth(seed, 0, 0, n)
    assert util.bits_equal(x, orc.synth(seed, 0, 0, n))
    f = _mk(kind, b, a)
    pos = 0
    for cut in RUNS[key + "/cuts"]:
        cut = int(cut)
        y = f.updaten(x[pos:cut])
        assert util.bits_equal(y, yref[pos:cut]), (key, cut)
        assert util.bits_equal(f.state, RUNS["%s/state@%d" % (key, cut)]), (key, cut)
        assert util.bits_equal([f.filtered_value], RUNS["%s/fv@%d" % (key, cut)])
        assert f.initialized == int(RUNS["%s/init@%d" % (key, cut)][0])
        pos = cut
    # one call == split calls
    g = _mk(kind, b, a)
    assert util.bits_equal(g.updaten(x), yref)
    # reset-and-reuse (state left stale on purpose)
    f.reset()
    x2 = util.synth(int(RUNS[key + "/seed_after_reset"][0]), 1, 0, len(RUNS[key + "/y_after_reset"]))
    assert util.bits_equal(f.updaten(x2), RUNS[key + "/y_after_reset"])
    assert util.bits_equal(f.state, RUNS[key + "/state_after_reset"])


def test_demo_config1():
    d = util.load_npz("demo.npz")
    f = orc.iir(d["b"], d["a"])
    y = f.updaten(d["u"])
    assert util.bits_equal(y, d["y"])
    # values quoted in SURVEY.md 8(a) "demo" row
    np.testing.assert_allclose(y[:4], [-1.82457222, -0.37724194, 6.73985245, 24.77571678], rtol=1e-8)
    np.testing.assert_allclose(y[-2:], [-31230.17139547, -16460.75833781], rtol=1e-9)


def test_bank_driver_matches_single_channel_and_threads():
    kind, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    _, bf, _ = util.design_coefs("fir32_lp_0.25")
    _, bm, _ = util.design_coefs("mavg32")
    designs = [dict(kind=orc.KIND_IIR, init_mode=orc.INIT_IIR_STABLE, b=b, a=a),
               dict(kind=orc.KIND_FIR, init_mode=orc.INIT_FIR_STABLE, b=bf),
               dict(kind=orc.KIND_FIR, init_mode=orc.INIT_FIR_PARTIAL, b=bm)]
    idx = [0, 1, 2, 0, 2, 1, 0]
    x = np.stack([util.synth(7, c, 0, 500) for c in range(len(idx))])
    bank = orc.Bank(designs, idx)
    y1 = bank.updaten(x[:, :200], threads=3)
    y2 = bank.updaten(x[:, 200:], threads=1)
    y = np.concatenate([y1, y2], axis=1)
    for c, di in enumerate(idx):
        d = designs[di]
        f = orc.Filter(d["kind"], d["b"], d.get("a"), d["init_mode"])
        assert util.bits_equal(f.updaten(x[c]), y[c]), c
        assert util.bits_equal(f.state, bank.state[c, :len(f.state)])


def test_long_double_floor_is_small_for_pinned_designs():
    # the float64 recurrence sits ~1e-15 from an 80-bit run for the pinned designs
    for key, bound in (("bw4_bp_0.3_0.6", 1e-13), ("bw8_bp_0.25_0.75", 1e-13)):
        _, b, a = util.design_coefs(key)
        x = util.synth(11, 0, 0, 20000)
        y = orc.iir(b, a).updaten(x)
        floor = float(util.rel_to_peak(y, orc.iir_truth_ld(b, a, x)))
        assert floor < bound, (key, floor)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_vs_live_reference_random_filters():
    rng = np.random.default_rng(5)
    for trial in range(40):
        nb = int(rng.integers(1, 10))
        na = int(rng.integers(1, 10))
        b = rng.standard_normal(nb)
        a = rng.standard_normal(na) * 0.3
        a[0] = rng.choice([1.0, 2.0, 0.75, -1.5])
        x = rng.standard_normal(257)
        r = ref.RefIIR(b, a)
        o = orc.iir(b, a)
        assert util.bits_equal(r.updaten(x[:100]), o.updaten(x[:100]))
        assert util.bits_equal(r.updaten(x[100:]), o.updaten(x[100:]))
        assert util.bits_equal(r.snapshot()["state"], o.state)
    for trial in range(20):
        nb = int(rng.integers(1, 40))
        b = rng.standard_normal(nb)
        x = rng.standard_normal(150)
        for mode in ("stable", "partial"):
            r = ref.RefFIR(b, mode)
            o = orc.fir(b, partial=(mode == "partial"))
            assert util.bits_equal(r.updaten(x), o.updaten(x)), (nb, mode)
            assert util.bits_equal(r.snapshot()["state"], o.state)
            assert r.snapshot()["initialized"] == o.initialized
