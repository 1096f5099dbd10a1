#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference (oracle/_ref/libdsp_ref.so).

Run once in the build container (where /root/reference exists):

    make -C oracle ref && python tests/golden/make_golden.py

Writes tests/golden/{kats.json, designs.json, runs.npz, demo.npz, errors.json, response.json}.
Everything written is DATA (inputs, coefficient bit patterns, outputs, state
snapshots); no reference source text is stored.  The GPU box has no reference,
so tests there read only these files.
"""
import ctypes as C
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402


def hexes(arr):
    return [struct.pack(">d", float(v)).hex() for v in np.asarray(arr, dtype=np.float64).ravel()]


def splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth(seed, chan, n0, n):
    """SURVEY.md 8(d) generator: uniform [-1,1)."""
    with np.errstate(over="ignore"):
        idx = (np.uint64(chan) << np.uint64(32)) + np.arange(n0, n0 + n, dtype=np.uint64)
        h = splitmix64(np.uint64(seed) ^ idx)
    return (h >> np.uint64(11)).astype(np.int64).astype(np.float64) * 2.0 ** -52 - 1.0


def windowed_sinc(ntaps, fc):
    """Hamming windowed-sinc low-pass, normalised to sum 1 (SURVEY.md 8(d) cfg 3).

    The reference has no FIR design routine except moving_average; this formula is
    ours and only its resulting bit patterns matter (they are stored)."""
    k = np.arange(ntaps, dtype=np.float64)
    m = (ntaps - 1) / 2.0
    h = fc * np.sinc(fc * (k - m)) * (0.54 - 0.46 * np.cos(2 * np.pi * k / (ntaps - 1)))
    return h / h.sum()


def make_kats():
    x = np.arange(1.0, 7.0)
    out = {}

    def rec(name, f, y):
        snap = f.snapshot()
        out[name] = dict(x=hexes(x), y=hexes(y), y_float=[float(v) for v in y],
                         state=hexes(snap["state"]), filtered_value=hexes([snap["filtered_value"]])[0],
                         initialized=snap["initialized"])

    f = ref.RefIIR([0.2, 0.3, 0.1], [1, -0.5, 0.25]); rec("kat1_iir", f, f.updaten(x))
    f = ref.RefIIR([0.2, 0.3, 0.1], [2, -1, 0.5]); rec("kat2_iir_a0", f, f.updaten(x))
    f = ref.RefIIR([0.5], [1, -0.5, 0.25]); rec("kat3_iir_nb_lt_na", f, f.updaten(x))
    f = ref.RefIIR([0.1, 0.2, 0.3, 0.4], [1, -0.3]); rec("kat4_iir_nb_gt_na", f, f.updaten(x))
    f = ref.RefFIR([0.5, 0.3, 0.2]); rec("kat5_fir", f, f.updaten(x))
    f = ref.RefFIR([0.5, 0.3, 0.4]); rec("kat6_fir_sum12", f, f.updaten(x))
    # KAT7/8: moving_average(window=3) on a 4-slot FIR, per-sample updates, then reset + replay.
    f = ref.RefFIR([0.0, 0.0, 0.0, 0.0], initialize="partial")
    rc = ref.lib().moving_average(C.byref(f.s), 3, None)
    assert rc == 0
    y7 = np.array([f.update(v) for v in x]); rec("kat7_mavg3_partial", f, y7)
    f.reset()
    y8 = np.array([f.update(v) for v in x]); rec("kat8_mavg3_reset_stale", f, y8)
    return out


DESIGNS = [
    # name, kind, order, wl, wu, ripple, tau
    ("bw4_bp_0.3_0.6", "butterworth", 4, 0.3, 0.6, None, None),       # cfg 2 / cfg 4 IIR
    ("bw8_bp_0.25_0.75", "butterworth", 8, 0.25, 0.75, None, None),   # cfg 5
    ("bw2_bp_demo", "butterworth", 2, None, None, None, None),         # cfg 1 (edges computed below)
    ("bw4_bp_0.2_0.3", "butterworth", 4, 0.2, 0.3, None, None),
    ("bw5_lp_0.25", "butterworth", 5, 0.0, 0.25, None, None),
    ("bw4_lp_0.5", "butterworth", 4, 0.0, 0.5, None, None),
    ("bw3_hp_0.4", "butterworth", 3, 0.4, 0.0, None, None),
    ("bw6_hp_0.3", "butterworth", 6, 0.3, 0.0, None, None),
    ("bw3_bs_0.6_0.3", "butterworth", 3, 0.6, 0.3, None, None),
    ("bw2_bs_0.5_0.2", "butterworth", 2, 0.5, 0.2, None, None),
    ("bw1_lp_0.1", "butterworth", 1, 0.0, 0.1, None, None),
    ("bw16_lp_0.5", "butterworth", 16, 0.0, 0.5, None, None),
    ("ch1_4_lp_0.3", "chebyshev1", 4, 0.0, 0.3, 0.5, None),
    ("ch1_3_bp_0.2_0.5", "chebyshev1", 3, 0.2, 0.5, 0.3, None),
    ("ch1_5_hp_0.4", "chebyshev1", 5, 0.4, 0.0, 0.7, None),
    ("ch2_4_lp_0.3", "chebyshev2", 4, 0.0, 0.3, 0.1, None),
    ("ch2_3_bp_0.2_0.5", "chebyshev2", 3, 0.2, 0.5, 0.05, None),
    ("thiran_5_4", "thiran", 5, None, None, None, 4.0),
    ("thiran_3_2.5", "thiran", 3, None, None, None, 2.5),
]


def demo_edges():
    # python/dsp.py:150-168 arithmetic, reproduced with numpy as the demo does.
    dt = 0.1
    w0 = 2 * np.pi / 5
    Ny = np.pi / dt
    return float(w0 / Ny * 0.95), float(w0 / Ny * 1.05)


def make_designs():
    out = {}
    for name, kind, order, wl, wu, ripple, tau in DESIGNS:
        if name == "bw2_bp_demo":
            wl, wu = demo_edges()
        rc, f = ref.design_iir(kind, order, wl or 0.0, wu or 0.0, ripple, tau)
        assert rc == 0, (name, rc)
        out[name] = dict(kind=kind, order=order, wl=wl, wu=wu, ripple=ripple, tau=tau,
                         wl_hex=None if wl is None else hexes([wl])[0],
                         wu_hex=None if wu is None else hexes([wu])[0],
                         b=hexes(f.b()), a=hexes(f.a()),
                         initialize_is_stable=bool(f.s.rtf.initialize ==
                                                   ref._fn_addr("RTIIRFilter_stable_init")))
    # FIR designs (ours; taps stored as bit patterns) + the reference's moving_average taps.
    out["fir512_lp_0.25"] = dict(kind="fir", b=hexes(windowed_sinc(512, 0.25)))
    out["fir32_lp_0.25"] = dict(kind="fir", b=hexes(windowed_sinc(32, 0.25)))
    out["fir7_lp_0.4"] = dict(kind="fir", b=hexes(windowed_sinc(7, 0.4)))
    for w in (3, 32, 512):
        f = ref.RefFIR([0.0] * w, initialize="partial")
        assert ref.lib().moving_average(C.byref(f.s), w, None) == 0
        out["mavg%d" % w] = dict(kind="fir_partial", b=hexes(f.b()),
                                 initialize_is_partial=bool(f.s.rtf.initialize ==
                                                            ref._fn_addr("RTFIRFilter_partial_init")))
    return out


def unhex(hs):
    return np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in hs], dtype=np.float64)


def make_runs(designs):
    """4096 seeded samples through each key design: full run, snapshots, split run, reset-reuse."""
    arrs = {}
    n = 4096
    keys = ["bw4_bp_0.3_0.6", "bw8_bp_0.25_0.75", "bw2_bp_demo", "bw5_lp_0.25", "bw3_hp_0.4",
            "bw3_bs_0.6_0.3", "ch1_4_lp_0.3", "thiran_5_4",
            "fir512_lp_0.25", "fir32_lp_0.25", "fir7_lp_0.4", "mavg3", "mavg32", "mavg512"]
    for ki, key in enumerate(keys):
        d = designs[key]
        x = synth(0x5EED0000 + 100 + ki, 0, 0, n)

        def mk():
            if d["kind"] == "fir":
                return ref.RefFIR(unhex(d["b"]), "stable")
            if d["kind"] == "fir_partial":
                return ref.RefFIR(unhex(d["b"]), "partial")
            return ref.RefIIR(unhex(d["b"]), unhex(d["a"]), "stable")

        f = mk()
        K = f.nstate
        cuts = sorted(set([1, min(K, n), 1024, n]))
        y = np.empty(n)
        pos = 0
        for cpos in cuts:
            y[pos:cpos] = f.updaten(x[pos:cpos])
            s = f.snapshot()
            arrs["%s/state@%d" % (key, cpos)] = s["state"]
            arrs["%s/fv@%d" % (key, cpos)] = np.array([s["filtered_value"]])
            arrs["%s/init@%d" % (key, cpos)] = np.array([s["initialized"]], dtype=np.int32)
            pos = cpos
        # x is NOT stored: it is synth(seed, chan=0, n0=0, n) of SURVEY.md 8(d); tests regenerate it.
        arrs[key + "/seed"] = np.array([0x5EED0000 + 100 + ki], dtype=np.int64)
        arrs[key + "/y"] = y
        arrs[key + "/cuts"] = np.array(cuts, dtype=np.int64)
        # split-block 1000 + 3096 must be identical (block carry)
        g = mk()
        y2 = np.concatenate([g.updaten(x[:1000]), g.updaten(x[1000:])])
        assert y2.tobytes() == y.tobytes(), key
        # per-sample path identical
        h = mk()
        y3 = np.array([h.update(v) for v in x[:300]])
        assert y3.tobytes() == y[:300].tobytes(), key
        # reset-and-reuse: RTFilter_reset after the run, then 512 more samples of a new stream
        x2 = synth(0x5EED0000 + 200 + ki, 1, 0, 700)
        f.reset()
        arrs[key + "/seed_after_reset"] = np.array([0x5EED0000 + 200 + ki], dtype=np.int64)  # chan=1
        arrs[key + "/y_after_reset"] = f.updaten(x2)
        s = f.snapshot()
        arrs[key + "/state_after_reset"] = s["state"]
    return arrs


def make_demo(designs):
    # inputs of python/dsp.py:150-166
    T, dt = 150, 0.1
    w0 = 2 * np.pi / 5
    t = np.arange(0.1, T, .1)
    y = np.sin(w0 * t)
    z = np.sin(w0 * 0.85 * t)
    w = np.sin(w0 * 1.15 * t)
    u = (y + z + w) * 100000 + 100000
    d = designs["bw2_bp_demo"]
    f = ref.RefIIR(unhex(d["b"]), unhex(d["a"]))
    out_n = f.updaten(u)
    g = ref.RefIIR(unhex(d["b"]), unhex(d["a"]))
    out_1 = np.array([g.update(v) for v in u])
    assert out_1.tobytes() == out_n.tobytes()
    return dict(u=u, y=out_n, b=unhex(d["b"]), a=unhex(d["a"]))


def make_errors():
    L = ref.lib()
    out = {}
    dp = C.POINTER(C.c_double)
    buf = (C.c_double * 4)()
    f = ref.RefIIR([0.0] * 5, [0.0] * 5)
    out["updaten_null_out"] = L.RTFilter_updaten(None, C.byref(f.s), buf, 4)
    out["updaten_null_rtf"] = L.RTFilter_updaten(buf, None, buf, 4)
    out["updaten_null_in"] = L.RTFilter_updaten(buf, C.byref(f.s), None, 4)
    out["butterworth_null"] = L.butterworth(None, 2, 0.1, 0.2, 0, None)
    out["butterworth_equal_edges"] = L.butterworth(C.byref(f.s), 2, 0.2, 0.2, 0, None)
    out["butterworth_order0"] = L.butterworth(C.byref(f.s), 0, 0.1, 0.2, 0, None)
    out["butterworth_bank_small"] = L.butterworth(C.byref(f.s), 3, 0.1, 0.2, 0, None)
    out["butterworth_ok"] = L.butterworth(C.byref(f.s), 2, 0.1, 0.2, 0, None)
    g = ref.RefIIR([0.0], [0.0] * 3)
    out["thiran_tau0"] = L.thiran(C.byref(g.s), 2, 0.0, None)
    out["thiran_order0"] = L.thiran(C.byref(g.s), 0, 1.0, None)
    out["thiran_bank_small"] = L.thiran(C.byref(g.s), 3, 1.0, None)
    out["thiran_ok"] = L.thiran(C.byref(g.s), 2, 1.0, None)
    h = ref.RefFIR([0.0] * 4, "partial")
    out["mavg_null"] = L.moving_average(None, 3, None)
    out["mavg_window0"] = L.moving_average(C.byref(h.s), 0, None)
    out["mavg_too_wide"] = L.moving_average(C.byref(h.s), 5, None)
    out["mavg_ok"] = L.moving_average(C.byref(h.s), 4, None)
    del dp
    return out


def make_response():
    """filter_response_pzg_noc of the reference (src/filterutils.c:137-191), always with a caller's grid -- as the
    reference's own python/dsp.py:126-143 calls it.  With freq == NULL the reference allocates room for ONE
    double behind its complex scratch and then writes ng of them (src/filterutils.c:151-161): heap corruption,
    so that path has no golden (the rebuilt library builds the same default grid in memory of its own)."""
    L = ref.lib()
    dp = C.POINTER(C.c_double)
    L.filter_response_pzg_noc.argtypes = [dp, dp, C.c_size_t, dp, dp, C.c_size_t, dp, dp, C.c_size_t, C.c_double, dp]
    L.filter_response_pzg_noc.restype = C.c_int
    rng = np.random.default_rng(20240)
    cases = {}
    for name, nz, npoles, ng, own_grid in (("z3p4", 3, 4, 33, True), ("z4p2", 4, 2, 17, True),
                                           ("z0p5", 0, 5, 9, True), ("z2p0", 2, 0, 8, True)):
        zr, zi = rng.uniform(-1, 1, nz), rng.uniform(-1, 1, nz)
        pr, pi = rng.uniform(-0.9, 0.9, npoles), rng.uniform(-0.9, 0.9, npoles)
        freq = np.sort(rng.uniform(0, np.pi, ng))
        if name == "z3p4":                         # the grid python/dsp.py builds: i * 1/(n-1)
            freq = np.array([i * (1.0 / (ng - 1)) for i in range(ng)])
        gain, phase = np.zeros(ng), np.zeros(ng)
        k = 0.75
        rc = L.filter_response_pzg_noc(gain.ctypes.data_as(dp), phase.ctypes.data_as(dp), ng,
                                       zr.ctypes.data_as(dp), zi.ctypes.data_as(dp), nz,
                                       pr.ctypes.data_as(dp), pi.ctypes.data_as(dp), npoles, k,
                                       freq.ctypes.data_as(dp) if own_grid else None)
        cases[name] = dict(status=rc, ng=ng, kgain=float(k).hex(), zeros_r=hexes(zr), zeros_i=hexes(zi),
                           poles_r=hexes(pr), poles_i=hexes(pi), freq=hexes(freq) if own_grid else None,
                           gain=hexes(gain), phase=hexes(phase))
    g1 = np.zeros(1)
    cases["errors"] = dict(null_gain=L.filter_response_pzg_noc(None, g1.ctypes.data_as(dp), 1, None, None, 0, None,
                                                               None, 0, 1.0, None),
                           ng0=L.filter_response_pzg_noc(g1.ctypes.data_as(dp), g1.ctypes.data_as(dp), 0, None, None,
                                                         0, None, None, 0, 1.0, None))
    return cases


def main():
    assert ref.available(), "build oracle/_ref first: make -C oracle ref"
    with open(os.path.join(HERE, "response.json"), "w") as fh:
        json.dump(make_response(), fh, indent=1, sort_keys=True)
    kats = make_kats()
    designs = make_designs()
    with open(os.path.join(HERE, "kats.json"), "w") as fh:
        json.dump(kats, fh, indent=1, sort_keys=True)
    with open(os.path.join(HERE, "designs.json"), "w") as fh:
        json.dump(designs, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "runs.npz"), **make_runs(designs))
    np.savez_compressed(os.path.join(HERE, "demo.npz"), **make_demo(designs))
    with open(os.path.join(HERE, "errors.json"), "w") as fh:
        json.dump(make_errors(), fh, indent=1, sort_keys=True)
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
