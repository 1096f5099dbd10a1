"""The tiled block layout and the chained block updates (include/RTFilterBank.h: RTFilterBank_updaten_tiled,
RTFilterBank_updaten_chained) against RTFilterBank_updaten and the oracle.  Both entry points run the pipelined
IIR kernel with the same arithmetic in the same order, so their outputs are held to the BITS of
RTFilterBank_updaten on the same blocks; that entry in turn is held to 1e-12 of the oracle (the re-associated
kernel's bar), and so are these."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def G():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests import gpu_util
    return gpu_util


def _bank(G, key, nch, mode=None):
    from dsp_amd import bank as gbank
    return gbank.FilterBank([G.golden_design(key)], [0] * nch, mode=gbank.MODE_AUTO if mode is None else mode)


def _oracle_rows(G, key, x):
    from oracle import oracle as orc
    d = G.to_orc_design(G.golden_design(key))
    ob = orc.Bank([d], [0] * x.shape[0])
    return ob.updaten(x, threads=8), ob


def test_tile_round_trip(G):
    import torch
    from dsp_amd import bank as gbank
    x = torch.rand((37, 3 * gbank.TILE), dtype=torch.float64, device="cuda")
    t = gbank.tile_rows(x)
    assert t.shape == (3, 37, gbank.TILE)
    assert torch.equal(t[1, 5], x[5, gbank.TILE:2 * gbank.TILE])
    assert torch.equal(gbank.untile_rows(t), x)


@pytest.mark.parametrize("key,nch", [("bw4_bp_0.3_0.6", 96), ("bw4_lp_0.5", 40), ("bw5_lp_0.25", 17)])
def test_tiled_layout_matches_rows_to_the_bit(G, key, nch):
    """Blocks of 3, 1 and 5 tiles, state carried; in place for the last block."""
    import torch
    from dsp_amd import bank as gbank
    blocks = [3 * gbank.TILE, gbank.TILE, 5 * gbank.TILE]
    a_bank, b_bank = _bank(G, key, nch), _bank(G, key, nch)
    pos = 0
    ys = []
    for i, n in enumerate(blocks):
        x = G.to_dev(G.make_inputs(77, nch, n, pos))
        want = a_bank(x)
        xt = gbank.tile_rows(x)
        yt = xt if i == 2 else torch.empty_like(xt)
        b_bank.updaten_tiled_ptr(yt.data_ptr(), xt.data_ptr(), n)
        got = gbank.untile_rows(yt)
        assert torch.equal(got, want), (key, i)
        ys.append(got.cpu().numpy())
        pos += n
    want_o, ob = _oracle_rows(G, key, G.make_inputs(77, nch, sum(blocks), 0))
    assert util.rel_to_peak(np.concatenate(ys, axis=1), want_o).max() <= TOL
    sa, sb = a_bank.get_state(), b_bank.get_state()
    for p, q in zip(sa, sb):
        assert util.bits_equal(p, q)


@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("key,nch", [("bw4_bp_0.3_0.6", 1024), ("bw4_bp_0.3_0.6", 300), ("bw5_lp_0.25", 64)])
def test_chained_blocks_match_stream_order_to_the_bit(G, key, nch, tiled):
    """Twelve blocks of 4 tiles through the chain (fresh stream: the head launch runs stable_init), a join in the
    middle (get_state), then more blocks: outputs, end state, filtered_value and initialized equal those of
    RTFilterBank_updaten on the same blocks bit for bit; the whole stream within 1e-12 of the oracle."""
    import torch
    from dsp_amd import bank as gbank
    n, nblocks = 4 * gbank.TILE, 12
    x = G.to_dev(G.make_inputs(91, nch, n * nblocks, 0))
    ref_bank, ch_bank = _bank(G, key, nch), _bank(G, key, nch)
    want = torch.empty_like(x)
    for j in range(nblocks):
        ref_bank.updaten_ptr(want.data_ptr() + j * n * 8, x.stride(0), x.data_ptr() + j * n * 8, x.stride(0), n)
    got = torch.empty_like(x)
    events = []

    def run(j):
        if tiled:
            xt = gbank.tile_rows(x[:, j * n:(j + 1) * n])
            yt = torch.empty_like(xt)
            events.append((ch_bank.updaten_chained_ptr(yt.data_ptr(), 0, xt.data_ptr(), 0, n, tiled=True,
                                                       want_event=True), j, yt, xt))
        else:
            events.append((ch_bank.updaten_chained_ptr(got.data_ptr() + j * n * 8, got.stride(0),
                                                       x.data_ptr() + j * n * 8, x.stride(0), n, want_event=True),
                           j, None, None))

    for j in range(7):
        run(j)
    st_mid = ch_bank.get_state()                     # joins the chain, synchronises
    assert st_mid[2].max() == 0
    for j in range(7, nblocks):
        run(j)
    ch_bank.chain_join()
    torch.cuda.synchronize()
    if tiled:
        for ev, j, yt, _ in events:
            assert ev is not None
            got[:, j * n:(j + 1) * n] = gbank.untile_rows(yt)
    torch.cuda.synchronize()
    assert torch.equal(got, want), (key, nch, tiled)
    for p, q in zip(ref_bank.get_state(), ch_bank.get_state()):
        assert util.bits_equal(p, q)
    want_o, ob = _oracle_rows(G, key, x.cpu().numpy())
    assert util.rel_to_peak(got.cpu().numpy(), want_o).max() <= TOL


def test_chained_call_that_does_not_qualify_runs_in_stream_order(G):
    """A ragged block length, a sequential-mode bank and a FIR bank: same results as RTFilterBank_updaten,
    done_event valid."""
    import torch
    from dsp_amd import bank as gbank
    nch = 48
    for key, mode, n in (("bw4_bp_0.3_0.6", gbank.MODE_AUTO, 5000), ("bw4_bp_0.3_0.6", gbank.MODE_SEQUENTIAL, 4096),
                         ("fir32_lp_0.25", gbank.MODE_AUTO, 4096)):
        x = G.to_dev(G.make_inputs(5, nch, 2 * n, 0))
        a_bank, b_bank = _bank(G, key, nch, mode), _bank(G, key, nch, mode)
        want, got = torch.empty_like(x), torch.empty_like(x)
        for j in range(2):
            a_bank.updaten_ptr(want.data_ptr() + j * n * 8, 2 * n, x.data_ptr() + j * n * 8, 2 * n, n)
            ev = b_bank.updaten_chained_ptr(got.data_ptr() + j * n * 8, 2 * n, x.data_ptr() + j * n * 8, 2 * n, n,
                                            want_event=True)
            assert ev
        b_bank.chain_join()
        torch.cuda.synchronize()
        assert torch.equal(got, want), key


def test_chain_then_reset_then_chain(G):
    """RTFilter_reset between two chains (the benchmark's wrap): the second chain's head runs stable_init again."""
    import torch
    from dsp_amd import bank as gbank
    key, nch, n = "bw4_bp_0.3_0.6", 128, 2 * gbank.TILE
    x = G.to_dev(G.make_inputs(17, nch, 3 * n, 0))
    a_bank, b_bank = _bank(G, key, nch), _bank(G, key, nch)
    want, got = torch.empty((2,) + tuple(x.shape), dtype=torch.float64, device="cuda"), None
    got = torch.empty_like(want)
    for rep in range(2):
        for j in range(3):
            a_bank.updaten_ptr(want[rep].data_ptr() + j * n * 8, 3 * n, x.data_ptr() + j * n * 8, 3 * n, n)
            b_bank.updaten_chained_ptr(got[rep].data_ptr() + j * n * 8, 3 * n, x.data_ptr() + j * n * 8, 3 * n, n)
        a_bank.reset()
        b_bank.reset()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(got[0], got[1])


@pytest.mark.parametrize("nch,ntiles", [(2500, 12), (4500, 3)])
def test_tiled_wide_bank_round_by_round_matches_rows_to_the_bit(G, nch, ntiles):
    """Banks wider than one resident round of workgroups: the tiled entry launches them round by round (full
    rounds of >= 12 tiles on the schedule), two blocks with the state carried."""
    import torch
    from dsp_amd import bank as gbank
    key, n = "bw4_bp_0.3_0.6", ntiles * gbank.TILE
    a_bank, b_bank = _bank(G, key, nch), _bank(G, key, nch)
    for i in range(2):
        x = torch.empty((nch, n), dtype=torch.float64, device="cuda")
        gbank.synth_fill(x, 0x5EED0002, chan0=0, n0=i * n)
        want = a_bank(x)
        xt = gbank.tile_rows(x)
        yt = torch.empty_like(xt)
        b_bank.updaten_tiled_ptr(yt.data_ptr(), xt.data_ptr(), n)
        assert torch.equal(gbank.untile_rows(yt), want), i
    for p, q in zip(a_bank.get_state(), b_bank.get_state()):
        assert util.bits_equal(p, q)
    # against the oracle: a few channels of the second block's end state would need the whole run; the rows entry
    # is held to the oracle at this shape in test_gpu_iir / test_gpu_fullsize, and the outputs above are its bits


@pytest.mark.parametrize("mode", ["auto", "exact"])
@pytest.mark.parametrize("key,nch", [("fir32_lp_0.25", 40), ("fir7_lp_0.4", 17), ("fir7_lp_0.4", 1), ("fir512_lp_0.25", 6),
                                     ("mavg32", 9), ("mavg3", 5)])
def test_tiled_fir_matches_rows(G, key, nch, mode):
    """FIR banks on the tiled layout: blocks of 2, 4 and 2 tiles, the first on a fresh stream (stable_init), the
    last in place.  Filters of <= 128 taps run on the streaming kernel (fir_tile_kernel): in RTFB_MODE_SEQUENTIAL
    with the reference's arithmetic -- the BITS of the rows entry in that mode and of the oracle --, in AUTO with
    fused multiply-adds (within 1e-12).  Longer filters run on the matrix-core kernel in AUTO (the bits of
    RTFilterBank_updaten in AUTO) and are refused in SEQUENTIAL.  The carried delay line is bit-exact in every case."""
    import torch
    from dsp_amd import bank as gbank
    exact = mode == "exact"
    m = gbank.MODE_SEQUENTIAL if exact else gbank.MODE_AUTO
    K = len(util.design_coefs(key)[1])
    blocks = [2 * gbank.TILE, 4 * gbank.TILE, 2 * gbank.TILE]
    a_bank, b_bank = _bank(G, key, nch, mode=m), _bank(G, key, nch, mode=m)
    if exact and K > 128:
        x = torch.zeros((nch, blocks[0]), dtype=torch.float64, device="cuda")
        with pytest.raises(Exception):
            b_bank.updaten_tiled_ptr(x.data_ptr(), x.data_ptr(), blocks[0])
        return
    pos, ys = 0, []
    for i, n in enumerate(blocks):
        x = G.to_dev(G.make_inputs(78, nch, n, pos))
        want = a_bank(x)
        xt = gbank.tile_rows(x)
        yt = xt if i == 2 else torch.empty_like(xt)
        b_bank.updaten_tiled_ptr(yt.data_ptr(), xt.data_ptr(), n)
        got = gbank.untile_rows(yt)
        if exact or K > 128:
            assert torch.equal(got, want), (key, i)
        else:
            assert util.rel_to_peak(got.cpu().numpy(), want.cpu().numpy()).max() <= TOL, (key, i)
        ys.append(got.cpu().numpy())
        pos += n
    sa, sb = a_bank.get_state(), b_bank.get_state()
    assert util.bits_equal(sa[0], sb[0]) and np.array_equal(sa[2], sb[2])
    want_o, ob = _oracle_rows(G, key, G.make_inputs(78, nch, sum(blocks), 0))
    got_all = np.concatenate(ys, axis=1)
    if exact:
        assert util.bits_equal(got_all, want_o)
    assert util.rel_to_peak(got_all, want_o).max() <= TOL
    assert util.bits_equal(b_bank.get_state()[0][:, :ob.state.shape[1]], ob.state)     # bit-exact end-state kernels


def _rel_dev(y, yref):
    peak = yref.abs().amax(dim=1).clamp_min(1e-300)
    return ((y - yref).abs().amax(dim=1) / peak).max().item()


def test_tiled_fir_wide_bank(G):
    """More channels than three resident rounds: the streaming kernel goes round by round on the schedule.
    Against the rows entry and the oracle on sampled channels."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    key, nch, n = "fir32_lp_0.25", 3400, 2 * gbank.TILE
    a_bank, b_bank = _bank(G, key, nch), _bank(G, key, nch)
    outs = []
    for i in range(2):
        x = torch.empty((nch, n), dtype=torch.float64, device="cuda")
        gbank.synth_fill(x, 0x5EED0004, chan0=0, n0=i * n)
        want = a_bank(x)
        xt = gbank.tile_rows(x)
        yt = torch.empty_like(xt)
        b_bank.updaten_tiled_ptr(yt.data_ptr(), xt.data_ptr(), n)
        got = gbank.untile_rows(yt)
        assert _rel_dev(got, want) <= TOL, i           # streaming kernel (fused) vs matrix-core kernel
        outs.append(got)
    assert util.bits_equal(a_bank.get_state()[0], b_bank.get_state()[0])
    d = G.to_orc_design(G.golden_design(key))
    for c in (0, 767, 768, 3399):
        xin = util.synth(0x5EED0004, c, 0, 2 * n)
        want_o = orc.Bank([d], [0]).updaten(xin[None, :], threads=1)[0]
        got = torch.cat([o[c] for o in outs]).cpu().numpy()
        assert util.rel_to_peak(got, want_o) <= TOL, c


@pytest.mark.parametrize("mode", ["auto", "exact"])
@pytest.mark.parametrize("K,partial,nch", [(33, False, 9), (64, False, 40), (64, True, 5), (100, False, 17), (128, False, 12),
                                           (128, True, 3)])
def test_tiled_fir_streaming_kernel_33_to_128_taps(G, K, partial, nch, mode):
    """The 64- and 128-tap instantiations of the streaming kernel (halo of 64 / 128 samples in front of a tile, taps
    read where they are used), stable_init and partial_init ramps, two designs of different lengths in one group:
    the oracle's bits in RTFB_MODE_SEQUENTIAL, within 1e-12 with fused multiply-adds in AUTO; a fresh stream, a
    carried block, an in-place block; the delay line bit-exact."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    rng = np.random.default_rng(1000 + K)
    ds = [gbank.fir_design(rng.normal(size=K) / K ** 0.5, partial=partial),
          gbank.fir_design(rng.normal(size=K - 20) / K ** 0.5, partial=False)]
    cmap = [c % 2 for c in range(nch)]
    gb = gbank.FilterBank(ds, cmap, mode=gbank.MODE_SEQUENTIAL if mode == "exact" else gbank.MODE_AUTO)
    ob = orc.Bank([G.to_orc_design(d) for d in ds], cmap)
    pos = 0
    for i, n in enumerate([2 * gbank.TILE, 4 * gbank.TILE, 2 * gbank.TILE]):
        x = G.make_inputs(79, nch, n, pos)
        want = ob.updaten(x, threads=4)
        xt = gbank.tile_rows(G.to_dev(x))
        yt = xt if i == 2 else torch.empty_like(xt)
        gb.updaten_tiled_ptr(yt.data_ptr(), xt.data_ptr(), n)
        got = gbank.untile_rows(yt).cpu().numpy()
        if mode == "exact":
            assert util.bits_equal(got, want), (K, i)
        else:
            assert util.rel_to_peak(got, want).max() <= TOL, (K, i)
        st = gb.get_state()
        assert util.bits_equal(st[0][:, :ob.state.shape[1]], ob.state) and np.array_equal(st[2], ob.init)
        pos += n


def test_tiled_entry_refuses_what_it_does_not_serve(G):
    import torch
    from dsp_amd import bank as gbank
    x = torch.zeros((4, 2 * gbank.TILE), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    fir = _bank(G, "fir32_lp_0.25", 4)
    with pytest.raises(Exception):                          # one tile: not a multiple of the 4096-output item
        fir.updaten_tiled_ptr(y.data_ptr(), x.data_ptr(), gbank.TILE)
    seq = _bank(G, "fir512_lp_0.25", 4, mode=gbank.MODE_SEQUENTIAL)        # > 128 taps: only the matrix cores serve it
    with pytest.raises(Exception):
        seq.updaten_tiled_ptr(y.data_ptr(), x.data_ptr(), 2 * gbank.TILE)
    _, b, _ = util.design_coefs("mavg512")                  # partial_init with more than 128 taps: no tiled kernel
    ramp = gbank.FilterBank([gbank.fir_design(b, partial=True)], [0] * 4)
    with pytest.raises(Exception):
        ramp.updaten_tiled_ptr(y.data_ptr(), x.data_ptr(), 2 * gbank.TILE)
    assert np.all(fir.get_state()[2] != 0)                  # nothing was enqueued: every channel is still fresh
    assert torch.all(fir(x) == 0)                           # and the bank still works


@pytest.mark.parametrize("key,mode", [("fir32_lp_0.25", "auto"), ("fir32_lp_0.25", "exact"), ("fir7_lp_0.4", "auto")])
def test_tiled_fir_streaming_kernel_with_inf_and_nan(G, key, mode):
    """The streaming FIR kernel multiplies a sample by the K taps it meets and nothing else: Inf / NaN at tile
    boundaries, in the 32 samples in front of a tile, in the first and last samples of a block and packed densely
    show in exactly the outputs the oracle has them in, with the same class; every other output within the bar
    (the reference's bits in the exact mode); the carried delay line as the oracle's."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    nch, blocks = 6, [2 * gbank.TILE, 4 * gbank.TILE]
    K = len(util.design_coefs(key)[1])
    x = G.make_inputs(4243, nch, sum(blocks))
    x[0, 0] = np.inf
    x[0, 2047] = -np.inf
    x[0, 2048] = np.nan
    x[1, 4095] = np.inf                                     # last sample of the first block
    x[1, 4096 + 5] = np.nan
    x[2, 3000:3000 + 3 * K:3] = np.inf
    x[2, 3001] = -np.inf
    x[3, 1] = np.nan
    x[4, sum(blocks) - 1] = -np.inf
    d = G.golden_design(key)
    gb = gbank.FilterBank([d], [0] * nch, mode=gbank.MODE_SEQUENTIAL if mode == "exact" else gbank.MODE_AUTO)
    ob = orc.Bank([G.to_orc_design(d)], [0] * nch)
    pos = 0
    for n in blocks:
        xb = np.ascontiguousarray(x[:, pos:pos + n])
        with np.errstate(invalid="ignore", over="ignore"):
            want = ob.updaten(xb, threads=4)
        xt = gbank.tile_rows(G.to_dev(xb))
        yt = torch.empty_like(xt)
        gb.updaten_tiled_ptr(yt.data_ptr(), xt.data_ptr(), n)
        got = gbank.untile_rows(yt).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want)), (key, pos)
        assert np.array_equal(np.isposinf(got), np.isposinf(want)) and np.array_equal(np.isneginf(got), np.isneginf(want))
        fin = np.isfinite(want)
        if mode == "exact" or K <= 16:
            assert np.array_equal(got[fin].tobytes(), want[fin].tobytes()), (key, pos)
        else:
            assert np.max(np.abs(got[fin] - want[fin])) <= TOL * max(1.0, np.max(np.abs(want[fin]))), (key, pos)
        st = gb.get_state()[0][:, :ob.state.shape[1]]
        assert np.array_equal(np.isnan(st), np.isnan(ob.state))
        assert st[np.isfinite(ob.state)].tobytes() == ob.state[np.isfinite(ob.state)].tobytes()
        pos += n
