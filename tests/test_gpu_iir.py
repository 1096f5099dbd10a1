"""GPU parity tests for the IIR kernels, through the C ABI (RTFilterBank_*).

Bars: sequential kernel BIT-EXACT vs the oracle (and vs the reference's own golden
outputs); scan kernel max|dy|/max|y_ref| <= 1e-12 per channel on the pinned well-conditioned designs.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

TOL_SCAN = 1e-12   # north_star: "within 1e-12 relative error in float64"


@pytest.fixture(scope="module")
def G():
    from tests import gpu_util
    gpu_util.torch_mod()
    return gpu_util


def test_synth_fill_matches_checker(G):
    torch = G.torch_mod()
    from dsp_amd import bank as gbank
    t = torch.empty((5, 1000), dtype=torch.float64, device="cuda")
    gbank.synth_fill(t, 0x5EED0002, chan0=3, n0=77)
    ref = np.stack([util.synth(0x5EED0002, 3 + c, 77, 1000) for c in range(5)])
    assert util.bits_equal(t.cpu().numpy(), ref)


IIR_KEYS = ["bw4_bp_0.3_0.6", "bw8_bp_0.25_0.75", "bw2_bp_demo", "bw5_lp_0.25", "bw3_hp_0.4",
            "bw3_bs_0.6_0.3", "ch1_4_lp_0.3", "thiran_5_4"]        # every key has a captured reference run


@pytest.mark.parametrize("key", IIR_KEYS)
def test_sequential_bit_exact_vs_reference_golden(G, key):
    """One channel through the GPU sequential kernel == the reference's own output, bit for bit."""
    from dsp_amd import bank as gbank
    runs = util.load_npz("runs.npz")
    if key + "/y" not in runs.files:
        pytest.skip("no golden run for " + key)
    yref = runs[key + "/y"]
    x = util.synth(int(runs[key + "/seed"][0]), 0, 0, len(yref))
    gb = gbank.FilterBank([G.golden_design(key)], [0], mode=gbank.MODE_SEQUENTIAL)
    pos, outs = 0, []
    for cut in runs[key + "/cuts"]:
        cut = int(cut)
        outs.append(gb(G.to_dev(x[None, pos:cut])).cpu().numpy()[0])
        st, fv, ini = gb.get_state()
        ref_state = runs["%s/state@%d" % (key, cut)]
        assert util.bits_equal(st[0, :len(ref_state)], ref_state), (key, cut)
        assert util.bits_equal(fv, runs["%s/fv@%d" % (key, cut)])
        assert ini[0] == runs["%s/init@%d" % (key, cut)][0]
        pos = cut
    assert util.bits_equal(np.concatenate(outs), yref)
    # RTFilter_reset leaves state stale, then a new stream
    gb.reset()
    x2 = util.synth(int(runs[key + "/seed_after_reset"][0]), 1, 0, len(runs[key + "/y_after_reset"]))
    assert util.bits_equal(gb(G.to_dev(x2[None])).cpu().numpy()[0], runs[key + "/y_after_reset"])


@pytest.mark.parametrize("blocks", [[1], [2, 1], [31, 33], [64, 1000, 7], [4096 + 17, 515]])
def test_sequential_bit_exact_multichannel(G, blocks):
    from dsp_amd import bank as gbank
    designs = [G.golden_design(k) for k in ("bw4_bp_0.3_0.6", "bw8_bp_0.25_0.75", "bw3_hp_0.4",
                                            "thiran_5_4", "bw1_lp_0.1")]
    nch = 203                                    # not a multiple of 64; interleaved designs => index lists
    idx = [(3 * c + c // 7) % len(designs) for c in range(nch)]
    ys_g, ys_o, gb, ob = G.run_pair(designs, idx, blocks, gbank.MODE_SEQUENTIAL, seed=99)
    for yg, yo in zip(ys_g, ys_o):
        assert util.bits_equal(yg, yo)
    G.assert_state_equal(gb, ob, exact=True)


def test_sequential_kats_a0_not_one_and_ragged_taps(G):
    """a0 != 1 (true divide) and nb != na (generic kernel)."""
    from dsp_amd import bank as gbank
    kats = util.load_json("kats.json")
    cases = [("kat1_iir", [0.2, 0.3, 0.1], [1, -0.5, 0.25]), ("kat2_iir_a0", [0.2, 0.3, 0.1], [2, -1, 0.5]),
             ("kat3_iir_nb_lt_na", [0.5], [1, -0.5, 0.25]), ("kat4_iir_nb_gt_na", [0.1, 0.2, 0.3, 0.4], [1, -0.3])]
    designs = [gbank.iir_design(b, a) for _, b, a in cases]
    gb = gbank.FilterBank(designs, list(range(len(cases))), mode=gbank.MODE_SEQUENTIAL)
    x = np.tile(np.arange(1.0, 7.0), (len(cases), 1))
    y = gb(G.to_dev(x)).cpu().numpy()
    st, fv, ini = gb.get_state()
    for c, (name, b, a) in enumerate(cases):
        assert util.bits_equal(y[c], util.unhex(kats[name]["y"])), name
        ref_state = util.unhex(kats[name]["state"])
        assert util.bits_equal(st[c, :len(ref_state)], ref_state), name
        assert ini[c] == 0


def test_sequential_random_coefficients_and_a0(G):
    from dsp_amd import bank as gbank
    rng = np.random.default_rng(3)
    designs = []
    for n in (2, 3, 5, 9, 12, 17):
        b = rng.standard_normal(n)
        a = rng.standard_normal(n) * 0.1
        a[0] = rng.choice([1.0, 2.0, -0.5, 3.7])
        designs.append(gbank.iir_design(b, a))
    idx = [c % len(designs) for c in range(70)]
    ys_g, ys_o, gb, ob = G.run_pair(designs, idx, [130, 61], gbank.MODE_SEQUENTIAL, seed=5)
    for yg, yo in zip(ys_g, ys_o):
        assert util.bits_equal(yg, yo)
    G.assert_state_equal(gb, ob, exact=True)


def test_sequential_leading_zero_samples_keep_zero_signs(G):
    """Streams that start with exact zeros exercise the reference's `0.0 + b*x` sign behaviour."""
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    d = G.golden_design("bw4_bp_0.3_0.6")
    x = util.synth(17, 0, 0, 300)
    x[:40] = 0.0
    x[5] = -0.0
    x[100:120] = -0.0
    gb = gbank.FilterBank([d], [0, 0], mode=gbank.MODE_SEQUENTIAL)
    xx = np.stack([x, -x])
    y = gb(G.to_dev(xx)).cpu().numpy()
    for c in range(2):
        assert util.bits_equal(y[c], orc.iir(d["b"], d["a"]).updaten(xx[c]))


@pytest.mark.parametrize("key,blocks", [
    ("bw4_bp_0.3_0.6", [64 * 64 * 3]),                  # full segments only
    ("bw4_bp_0.3_0.6", [1, 5, 64 * 64 + 1, 100, 9000]),  # seed alone, ragged tails, carry across blocks
    ("bw8_bp_0.25_0.75", [5000, 64 * 64 * 2, 333]),
    ("bw5_lp_0.25", [4097, 4095]),
])
def test_scan_within_1e12_of_oracle(G, key, blocks):
    from dsp_amd import bank as gbank
    designs = [G.golden_design(key)]
    nch = 9
    ys_g, ys_o, gb, ob = G.run_pair(designs, [0] * nch, blocks, gbank.MODE_SCAN, seed=31)
    yg, yo = np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)
    err = util.rel_to_peak(yg, yo)
    assert np.all(err <= TOL_SCAN), (key, err.max())
    G.assert_state_equal(gb, ob, exact=False, tol=1e-12)


def test_scan_mixed_orders_noncontiguous_groups(G):
    from dsp_amd import bank as gbank
    keys = ("bw4_bp_0.3_0.6", "bw8_bp_0.25_0.75", "bw3_hp_0.4", "bw16_lp_0.5")
    designs = [G.golden_design(k) for k in keys]
    idx = [c % 4 for c in range(22)]
    ys_g, ys_o, gb, ob = G.run_pair(designs, idx, [64 * 64 + 77, 2000], gbank.MODE_SCAN, seed=8)
    yg, yo = np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)
    assert np.all(util.rel_to_peak(yg, yo) <= TOL_SCAN)


@pytest.mark.parametrize("order,band", [(9, None), (10, (0.2, 0.5)), (11, None), (12, (0.3, 0.7)), (13, None),
                                        (14, (0.25, 0.6)), (15, None), (16, (0.2, 0.6))])
def test_matrix_core_scan_orders_9_to_16(G, order, band):
    """Orders 9..16 run on the MFMA scan kernel with the states above N as zero padding of its 16-wide tiles:
    every order, full segments (the MFMA path), ragged and seeded ones (the general path) and the carry between them."""
    from dsp_amd import bank as gbank, dsp
    if band is None:
        f = dsp.butterworth(order, 0.0, 0.3 + 0.01 * order)               # low-pass of that order
    else:
        f = dsp.butterworth(order // 2, band[0], band[1])                 # band-pass: twice the prototype order
    b, a = np.array(f.b()), np.array(f.a())
    assert len(a) - 1 == order
    designs = [gbank.iir_design(b, a)]
    blocks = [64 * 64 * 2 + 5, 64 * 64, 100, 64 * 64 * 3]
    ys_g, ys_o, gb, ob = G.run_pair(designs, [0] * 5, blocks, gbank.MODE_SCAN, seed=40 + order)
    yg, yo = np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)
    err = util.rel_to_peak(yg, yo)
    assert np.all(err <= TOL_SCAN), (order, err.max())
    G.assert_state_equal(gb, ob, exact=False, tol=1e-12)


def test_matrix_core_scan_a0_not_one_unaligned_inplace(G):
    """Order-12 design with a0 = 2 (every coefficient doubled: the same filter), odd row pitch and an 8-byte aligned
    base (scalar access path), in place."""
    torch = G.torch_mod()
    from dsp_amd import bank as gbank, dsp
    from oracle import oracle as orc
    f = dsp.butterworth(6, 0.3, 0.7)
    b, a = 2.0 * np.array(f.b()), 2.0 * np.array(f.a())
    d = gbank.iir_design(b, a)
    nch, n, ld = 3, 64 * 64 * 2 + 77, 64 * 64 * 2 + 79
    x = G.make_inputs(23, nch, n)
    buf = torch.zeros(nch * ld + 1, dtype=torch.float64, device="cuda")
    view = buf[1:].view(nch, ld)[:, :n]
    view.copy_(G.to_dev(x))
    gb = gbank.FilterBank([d], [0] * nch, mode=gbank.MODE_SCAN)
    gb(view, out=view)
    yo = orc.Bank([G.to_orc_design(d)], [0] * nch).updaten(x)
    assert np.all(util.rel_to_peak(view.cpu().numpy(), yo) <= TOL_SCAN)


def test_unaligned_pointers_and_inplace(G):
    """Odd leading dimension / 8-byte-aligned base take the scalar access path; in == out allowed."""
    torch = G.torch_mod()
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    d = G.golden_design("bw4_bp_0.3_0.6")
    nch, n, ld = 5, 5001, 5003
    x = G.make_inputs(21, nch, n)
    for mode, exact in ((gbank.MODE_SEQUENTIAL, True), (gbank.MODE_SCAN, False)):
        buf = torch.zeros(nch * ld + 1, dtype=torch.float64, device="cuda")
        view = buf[1:].view(nch, ld)[:, :n]          # base 8-byte aligned only, odd ld
        view.copy_(G.to_dev(x))
        gb = gbank.FilterBank([d], [0] * nch, mode=mode)
        gb(view, out=view)                            # in place
        y = view.cpu().numpy()
        yo = orc.Bank([G.to_orc_design(d)], [0] * nch).updaten(x)
        if exact:
            assert util.bits_equal(y, yo)
        else:
            assert np.all(util.rel_to_peak(y, yo) <= TOL_SCAN)


def test_from_filters_and_sync_to_host(G):
    """Bank built from host structs; after sync the structs hold what the reference would have left."""
    from dsp_amd import bank as gbank, dsp
    from oracle import oracle as orc
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    filters = [dsp.RTIIRFilter(list(b), list(a)) for _ in range(3)]
    x = G.make_inputs(77, 3, 700)
    # channel 1 already consumed 100 samples on the host before the bank takes over
    pre = filters[1](x[1, :100].tolist())
    gb = gbank.FilterBank.from_filters(filters, mode=gbank.MODE_SEQUENTIAL)
    xin = x.copy()
    xin[1, :600] = x[1, 100:]
    y = gb(G.to_dev(xin[:, :600])).cpu().numpy()
    gb.sync_to_host()
    for c in range(3):
        o = orc.iir(b, a)
        if c == 1:
            yo = o.updaten(x[1])
            assert util.bits_equal(pre, yo[:100]) and util.bits_equal(y[1], yo[100:])
        else:
            yo = o.updaten(x[c, :600])
            assert util.bits_equal(y[c], yo)
        assert util.bits_equal(np.array(filters[c].state()), o.state)
        assert filters[c].filtered_value == o.filtered_value and filters[c].initialized == 0
    # and the host path continues seamlessly from the synced state
    nxt = filters[0](x[0, 600:].tolist())
    o = orc.iir(b, a); o.updaten(x[0, :600])
    assert util.bits_equal(nxt, o.updaten(x[0, 600:]))


def test_pool_reuses_blocks(G):
    import ctypes as C
    from dsp_amd.cdsp import dspdll
    p1, p2 = C.c_void_p(), C.c_void_p()
    assert dspdll.dsp_pool_alloc(C.byref(p1), 3 << 20, 0) == 0
    used = dspdll.dsp_pool_bytes_in_use(0)
    assert dspdll.dsp_pool_free(p1) == 0
    assert dspdll.dsp_pool_bytes_cached(0) >= (3 << 20)
    assert dspdll.dsp_pool_alloc(C.byref(p2), 3 << 20, 0) == 0
    assert p2.value == p1.value and dspdll.dsp_pool_bytes_in_use(0) == used
    assert dspdll.dsp_pool_free(p2) == 0
    assert dspdll.dsp_pool_free(C.c_void_p(12345)) != 0      # foreign pointer rejected
    assert dspdll.dsp_pool_trim(0) == 0


def test_pool_free_async_is_stream_ordered(G):
    """dsp_pool_free_async: the block is cached at once but not handed out again while work enqueued on the
    freeing stream before the free is still running."""
    import ctypes as C
    import torch
    from dsp_amd.cdsp import dspdll
    size = 5 << 20
    p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert dspdll.dsp_pool_trim(0) == 0
    assert dspdll.dsp_pool_alloc(C.byref(p1), size, 0) == 0
    big = torch.empty((64, 1 << 22), dtype=torch.float64, device="cuda")        # 2 GiB
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    for _ in range(40):                                   # tens of milliseconds of queued work on s
        assert dspdll.dsp_synth_fill(C.c_void_p(big.data_ptr()), 1 << 22, 64, 1 << 22, 1, 0, 0, sp) == 0
    assert dspdll.dsp_pool_free_async(p1, sp) == 0
    assert dspdll.dsp_pool_bytes_cached(0) >= size
    assert dspdll.dsp_pool_alloc(C.byref(p2), size, 0) == 0
    assert p2.value != p1.value, "block handed out while its free was still pending on the stream"
    s.synchronize()
    assert dspdll.dsp_pool_alloc(C.byref(p3), size, 0) == 0
    assert p3.value == p1.value                            # reusable once the stream has passed the free
    assert dspdll.dsp_pool_free(p2) == 0 and dspdll.dsp_pool_free(p3) == 0
    assert dspdll.dsp_pool_trim(0) == 0


def test_many_distinct_designs_take_the_scan2_kernel(G):
    """More than 1024 distinct designs (no per-design row propagators): orders <= 8 run on the two-wave scan kernel;
    the bank is built from packed arrays (dsp.design_batch -> FilterBank.from_arrays) and checked against the oracle
    on sampled channels, <= 1e-12, then bit-exact in SEQUENTIAL mode."""
    from dsp_amd import bank as gbank
    from dsp_amd import dsp
    from oracle import oracle as orc
    rng = np.random.default_rng(11)
    nd, n = 1500, 6000
    wl = 0.25 + 0.1 * rng.random(nd)
    wu = wl + 0.25 + 0.05 * rng.random(nd)
    b, a, nc = dsp.design_batch("butterworth", 4, wl, wu)
    x = G.make_inputs(21, nd, n)
    xd = G.to_dev(x)
    for mode, exact in ((gbank.MODE_AUTO, False), (gbank.MODE_SEQUENTIAL, True)):
        gb = gbank.FilterBank.from_arrays(gbank.KIND_IIR, b, a, mode=mode)
        y = gb(xd).cpu().numpy()
        for c in (0, 1, 777, 1023, 1024, 1499):
            yo = orc.iir(b[c], a[c]).updaten(x[c])
            if exact:
                assert util.bits_equal(y[c], yo), c
            else:
                assert np.max(np.abs(y[c] - yo)) / np.max(np.abs(yo)) <= 1e-12, c
        gb.close()


def test_groups_a_little_wider_than_a_resident_round_are_launched_round_by_round():
    """1024 < channels <= 4096 on the pipelined kernel: the launcher walks the group in resident rounds (offsets into
    the per-channel arrays, contiguous and listed channels).  Two designs interleaved unevenly so that one group is a
    channel LIST of 1300 entries and the other a list of 1200; against the bit-exact kernel on every channel and the
    oracle on a few; state carried over a second, ragged block."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    from tests import gpu_util as G
    d8 = G.golden_design("bw4_bp_0.3_0.6")
    d5 = G.golden_design("bw5_lp_0.25")
    nch = 2500
    idx = [0 if (c % 25) < 13 else 1 for c in range(nch)]
    x = G.make_inputs(41, nch, 8192 + 4096 + 77)
    outs = {}
    for name, mode in (("auto", gbank.MODE_AUTO), ("seq", gbank.MODE_SEQUENTIAL)):
        gb = gbank.FilterBank([d8, d5], idx, mode=mode)
        ys = [gb(G.to_dev(x[:, :8192])).cpu().numpy(), gb(G.to_dev(x[:, 8192:])).cpu().numpy()]
        outs[name] = (np.concatenate(ys, axis=1), gb.get_state())
        gb.close()
    err = util.rel_to_peak(outs["auto"][0], outs["seq"][0])
    assert float(np.max(err)) <= 1e-12
    st_a, st_s = outs["auto"][1][0], outs["seq"][1][0]
    assert np.max(np.abs(st_a - st_s)) <= 1e-12 * max(1.0, float(np.max(np.abs(st_s))))
    for c in (0, 13, 1299, 1300, 2499):
        ob = orc.Bank([G.to_orc_design(d8), G.to_orc_design(d5)], [idx[c]])
        want = ob.updaten(x[c:c + 1], threads=1)
        assert util.bits_equal(outs["seq"][0][c:c + 1], want)
