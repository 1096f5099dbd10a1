"""Parity at BASELINE.json's full sizes, where the oracle cannot follow sample by sample: size-independent
properties of the path (block-split invariance, agreement of the re-associated kernels with the bit-exact
kernels -- which the small tests pin to the reference bit for bit --, linearity) plus spot checks of whole
channels against the oracle.  Tolerance 1e-12 relative to each channel's output peak (north star)."""
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


def _rel_to_peak_dev(y, yref):
    """per-row max|y - yref| / max|yref| on the device"""
    peak = yref.abs().amax(dim=1).clamp_min(1e-300)
    return ((y - yref).abs().amax(dim=1) / peak).max().item()


def test_config2_full_size_scan_vs_bit_exact_kernel_and_oracle_spot_checks(G):
    """1024-channel order-8 bank, 2^20 samples per channel streamed in 64K blocks (config 2)."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    C_, L, B = 1024, 1 << 20, 1 << 16
    pitch = L + 128
    xb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
    x = xb[:, :L]
    gbank.synth_fill(x, 0x5EED0002)
    ys = {}
    for name, mode in (("scan", gbank.MODE_AUTO), ("seq", gbank.MODE_SEQUENTIAL)):
        bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_, mode=mode)
        y = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
        for j in range(L // B):
            bank.updaten_ptr(y.data_ptr() + j * B * 8, pitch, x.data_ptr() + j * B * 8, pitch, B)
        torch.cuda.synchronize()
        ys[name] = (y[:, :L], bank.get_state())
        bank.close()
    # the re-associated kernel agrees with the bit-exact one on every channel over the whole stream
    assert _rel_to_peak_dev(ys["scan"][0], ys["seq"][0]) <= TOL
    st_s, st_q = ys["scan"][1][0], ys["seq"][1][0]
    assert np.max(np.abs(st_s - st_q)) <= TOL * max(1.0, np.max(np.abs(st_q)))
    # whole channels against the oracle: the sequential kernel to the bit, the scan kernel to 1e-12
    for c in (0, 511, 1023):
        want = orc.iir(b, a).updaten(util.synth(0x5EED0002, c, 0, L))
        assert util.bits_equal(ys["seq"][0][c].cpu().numpy(), want)
        got = ys["scan"][0][c].cpu().numpy()
        assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))
    # block-split invariance: one 2^20 call == sixteen 64K calls
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_)
    y1 = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
    bank.updaten_ptr(y1.data_ptr(), pitch, x.data_ptr(), pitch, L)
    torch.cuda.synchronize()
    assert _rel_to_peak_dev(y1[:, :L], ys["scan"][0]) <= TOL
    bank.close()


def test_config2_linearity_at_full_block(G):
    """filter(4 x1 + x2) == 4 filter(x1) + filter(x2) for streams already running (no stable_init in the way)."""
    import torch
    from dsp_amd import bank as gbank
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    C_, B = 1024, 1 << 16
    x1 = torch.empty((C_, B), dtype=torch.float64, device="cuda")
    x2 = torch.empty((C_, B), dtype=torch.float64, device="cuda")
    gbank.synth_fill(x1, 1)
    gbank.synth_fill(x2, 2)
    outs = []
    for x in (x1, x2, 4.0 * x1 + x2):
        bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_)
        st, fv, ini = bank.get_state()
        bank.set_state(st, fv, np.zeros_like(ini))          # initialised, zero state
        outs.append(bank(x))
        bank.close()
    torch.cuda.synchronize()
    assert _rel_to_peak_dev(outs[2], 4.0 * outs[0] + outs[1]) <= 4 * TOL


def test_config3_full_size_mfma_vs_bit_exact_kernel_and_oracle(G):
    """Config 3 as benchmarked: 512-tap FIR, 65 536 channels x 65 536 samples (32 GiB in).  The matrix-core kernel
    against the bit-exact kernel on every output; against the oracle: 64 sampled channels in
    full and ALL channels for their first 1024 samples (stable_init + the head that sits on the carried state)."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    _, b, _ = util.design_coefs("fir512_lp_0.25")
    C_, B = 1 << 16, 1 << 16
    x = torch.empty((C_, B), dtype=torch.float64, device="cuda")
    gbank.synth_fill(x, 0x5EED0003)
    res = {}
    for name, mode in (("mfma", gbank.MODE_AUTO), ("exact", gbank.MODE_SEQUENTIAL)):
        bank = gbank.FilterBank([gbank.fir_design(b)], [0] * C_, mode=mode)
        y = torch.empty_like(x)
        bank.updaten_ptr(y.data_ptr(), B, x.data_ptr(), B, B)
        torch.cuda.synchronize()
        res[name] = (y, bank.get_state()[0])
        bank.close()
    worst = 0.0
    for c0 in range(0, C_, 8192):                              # in slabs: the difference needs a temporary
        worst = max(worst, _rel_to_peak_dev(res["mfma"][0][c0:c0 + 8192], res["exact"][0][c0:c0 + 8192]))
    assert worst <= TOL, worst
    assert util.bits_equal(res["mfma"][1], res["exact"][1])    # the carried delay line: exact in both modes
    sampled = sorted(set([0, 1, C_ - 1] + list(np.random.default_rng(3).integers(0, C_, 61))))
    for c in sampled:
        want = orc.fir(b).updaten(util.synth(0x5EED0003, c, 0, B))
        assert util.bits_equal(res["exact"][0][c].cpu().numpy(), want), c
        got = res["mfma"][0][c].cpu().numpy()
        assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want)), c
    head = 1024
    ob = orc.Bank([dict(kind=orc.KIND_FIR, init_mode=orc.INIT_FIR_STABLE, b=b)], [0] * 8192)
    for c0 in range(0, C_, 8192):
        xh = x[c0:c0 + 8192, :head].cpu().numpy()
        ob.state[:] = 0.0
        ob.reset()
        want = ob.updaten(np.ascontiguousarray(xh), threads=16)
        assert util.bits_equal(res["exact"][0][c0:c0 + 8192, :head].cpu().numpy(), want), c0
        err = util.rel_to_peak(res["mfma"][0][c0:c0 + 8192, :head].cpu().numpy(), want)
        assert err.max() <= TOL, (c0, err.max())


def test_config4_wide_mixed_bank_auto_vs_bit_exact(G):
    """262 144 channels, four interleaved designs (IIR 8 / IIR 16 / FIR 32 / IIR 5), 4096-sample blocks:
    every group gathers its channels through a list; the IIR groups have 65 536 channels (scan kernels for
    orders 8 and 5, lane-per-channel kernel for order 16 in AUTO), the FIR group runs on the matrix cores in AUTO."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    keys = ["bw4_bp_0.3_0.6", "bw8_bp_0.25_0.75", "fir32_lp_0.25", "bw5_lp_0.25"]
    designs = [G.golden_design(k) for k in keys]
    C_, B = 1 << 18, 4096
    idx = [c % 4 for c in range(C_)]
    x = torch.empty((C_, 2 * B), dtype=torch.float64, device="cuda")
    gbank.synth_fill(x, 44)
    res = {}
    for name, mode in (("auto", gbank.MODE_AUTO), ("exact", gbank.MODE_SEQUENTIAL)):
        bank = gbank.FilterBank(designs, idx, mode=mode)
        y = torch.empty_like(x)
        for j in range(2):
            bank.updaten_ptr(y.data_ptr() + j * B * 8, 2 * B, x.data_ptr() + j * B * 8, 2 * B, B)
        torch.cuda.synchronize()
        res[name] = y
        bank.close()
    assert _rel_to_peak_dev(res["auto"], res["exact"]) <= TOL
    for c in (0, 1, 2, 3, C_ - 4, C_ - 3, C_ - 2, C_ - 1):
        d = G.to_orc_design(designs[c % 4])
        f = orc.Filter(d["kind"], d["b"], d.get("a"), d["init_mode"])
        want = f.updaten(util.synth(44, c, 0, 2 * B))
        assert util.bits_equal(res["exact"][c].cpu().numpy(), want), c


def test_config5_full_stream_vs_oracle(G):
    """Config 5 as benchmarked: ONE 2^30-sample stream through the 16th-order filter, split in time on the GPU,
    against the oracle over the whole stream (one host thread, ~10 s), compared in 2^26-sample pieces."""
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    _, b, a = util.design_coefs("bw8_bp_0.25_0.75")
    n, piece = 1 << 30, 1 << 26
    x = torch.empty((1, n), dtype=torch.float64, device="cuda")
    gbank.synth_fill(x, 0x5EED0005)
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    y = fs(x.view(-1))
    torch.cuda.synchronize()
    assert fs.last_split >= 1024
    ref = orc.iir(b, a)
    peak, worst = 0.0, 0.0
    for p0 in range(0, n, piece):
        want = ref.updaten(util.synth(0x5EED0005, 0, p0, piece))
        got = y[p0:p0 + piece].cpu().numpy()
        peak = max(peak, float(np.max(np.abs(want))))
        worst = max(worst, float(np.max(np.abs(got - want))))
    assert worst <= TOL * peak, (worst, peak)
    st, fv, ini = fs.get_state()
    assert np.max(np.abs(st[:len(ref.state)] - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state)))
    assert abs(fv - ref.filtered_value) <= TOL * peak and ini == 0
    fs.close()


def test_config2_every_channel_every_sample_vs_oracle(G):
    """SURVEY.md 8(d), config 2's parity bar: ALL 1024 channels x 2^20 samples against the CPU restatement of the
    reference (2^30 samples, 16 host threads): the bit-exact kernel to the bit, the scan kernel <= 1e-12 relative
    to each channel's peak.  Streamed in 64K blocks with the state carried on the device, as benchmarked."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    C_, L, B = 1024, 1 << 20, 1 << 16
    pitch = L + 128
    xb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
    x = xb[:, :L]
    gbank.synth_fill(x, 0x5EED0002)
    ys = {}
    for name, mode in (("scan", gbank.MODE_AUTO), ("seq", gbank.MODE_SEQUENTIAL)):
        bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_, mode=mode)
        y = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
        for j in range(L // B):
            bank.updaten_ptr(y.data_ptr() + j * B * 8, pitch, x.data_ptr() + j * B * 8, pitch, B)
        torch.cuda.synchronize()
        ys[name] = y[:, :L]
        bank.close()
    # the same 16 blocks through the tiled layout (as benchmarked) and through the chained entry in both layouts:
    # the bits of RTFilterBank_updaten's scan kernel on every channel and sample, hence the oracle's to 1e-12 below
    for name, tiled, chained in (("tiled", True, False), ("tiled+chained", True, True), ("rows+chained", False, True)):
        bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_)
        xt = torch.empty((B // gbank.TILE, C_, gbank.TILE), dtype=torch.float64, device="cuda")
        yt = [torch.empty_like(xt) for _ in range(2)]
        yr = torch.empty((C_, B), dtype=torch.float64, device="cuda")
        for j in range(L // B):
            xj = x[:, j * B:(j + 1) * B]
            if tiled:
                gbank.tile_rows(xj, xt)
                if chained:
                    bank.updaten_chained_ptr(yt[j & 1].data_ptr(), 0, xt.data_ptr(), 0, B, tiled=True)
                    bank.chain_join()              # xt is refilled for the next block
                else:
                    bank.updaten_tiled_ptr(yt[j & 1].data_ptr(), xt.data_ptr(), B)
                gbank.untile_rows(yt[j & 1], yr)
            else:
                bank.updaten_chained_ptr(yr.data_ptr(), B, xj.data_ptr(), pitch, B)
                bank.chain_join()
            assert torch.equal(yr, ys["scan"][:, j * B:(j + 1) * B]), (name, j)
        bank.close()
    # sixteen chained blocks in flight back to back (no join in between), rows layout, straight into a 2^20 buffer
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_)
    yc = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
    for j in range(L // B):
        bank.updaten_chained_ptr(yc.data_ptr() + j * B * 8, pitch, x.data_ptr() + j * B * 8, pitch, B)
    bank.chain_join()
    torch.cuda.synchronize()
    assert torch.equal(yc[:, :L], ys["scan"])
    bank.close()
    del yc
    step = 128
    worst = 0.0
    for c0 in range(0, C_, step):
        xh = np.stack([util.synth(0x5EED0002, c, 0, L) for c in range(c0, c0 + step)])
        assert util.bits_equal(x[c0:c0 + step].cpu().numpy(), xh)            # device generator == host generator
        ob = orc.Bank([dict(kind=orc.KIND_IIR, init_mode=orc.INIT_IIR_STABLE, b=b, a=a)], [0] * step)
        want = ob.updaten(xh, threads=16)
        assert util.bits_equal(ys["seq"][c0:c0 + step].cpu().numpy(), want), c0
        err = util.rel_to_peak(ys["scan"][c0:c0 + step].cpu().numpy(), want)
        worst = max(worst, float(err.max()))
    assert worst <= TOL, worst


def test_config4_benchmarked_wave_shape_both_kinds(G):
    """Config 4 as bench.py times it: sorted waves of 65 536 channels x 65 536 samples at row pitch 64K + 128
    (32 GiB each way, row offsets beyond 2^32 bytes) -- one wave of the order-8 IIR (the pipelined kernel 64 resident
    rounds deep) and one of the 32-tap FIR (matrix-core kernel).  AUTO against the bit-exact mode on every channel
    and sample, and both against the oracle on channels 0, 32 767 and 65 535."""
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    C_, B = 1 << 16, 1 << 16
    pitch = B + 128
    xb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
    x = xb[:, :B]
    gbank.synth_fill(x, 0x5EED0004)
    for key in ("bw4_bp_0.3_0.6", "fir32_lp_0.25"):
        d = G.golden_design(key)
        ys = {}
        for name, mode in (("auto", gbank.MODE_AUTO), ("exact", gbank.MODE_SEQUENTIAL)):
            bank = gbank.FilterBank([d], [0] * C_, mode=mode)
            yb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
            bank.updaten_ptr(yb.data_ptr(), pitch, x.data_ptr(), pitch, B)
            torch.cuda.synchronize()
            ys[name] = (yb[:, :B], bank.get_state())
            bank.close()
        worst = 0.0
        for c0 in range(0, C_, 8192):
            worst = max(worst, _rel_to_peak_dev(ys["auto"][0][c0:c0 + 8192], ys["exact"][0][c0:c0 + 8192]))
        assert worst <= TOL, (key, worst)
        st_a, st_e = ys["auto"][1][0], ys["exact"][1][0]
        assert np.max(np.abs(st_a - st_e)) <= TOL * max(1.0, np.max(np.abs(st_e))), key
        od = G.to_orc_design(d)
        for c in (0, C_ // 2 - 1, C_ - 1):
            want = orc.Filter(od["kind"], od["b"], od.get("a"), od["init_mode"]).updaten(util.synth(0x5EED0004, c, 0, B))
            assert util.bits_equal(ys["exact"][0][c].cpu().numpy(), want), (key, c)
            got = ys["auto"][0][c].cpu().numpy()
            assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want)), (key, c)
        # The layout the bench times since round 3: the wave's block as [32 tiles][65 536 channels][2048]
        # (RTFilterBank_updaten_tiled: 64 launches of one resident round each on the schedule -- the pipelined kernel
        # for the IIR wave, the streaming FIR kernel for the 32-tap wave).  Against AUTO on rows on every channel and
        # sample (IIR: the same arithmetic, equal bits; FIR: within the bar), and an equal carried state.
        want_rows, want_state = ys["auto"]
        del ys
        xt = torch.empty((B // gbank.TILE, C_, gbank.TILE), dtype=torch.float64, device="cuda")
        gbank.tile_rows(x, tiles=xt)
        yt = torch.empty_like(xt)
        bank = gbank.FilterBank([d], [0] * C_, mode=gbank.MODE_AUTO)
        bank.updaten_tiled_ptr(yt.data_ptr(), xt.data_ptr(), B)
        torch.cuda.synchronize()
        del xt
        iir = key.startswith("bw")
        for t in range(B // gbank.TILE):
            ref = want_rows[:, t * gbank.TILE:(t + 1) * gbank.TILE]
            if iir:                                        # the same kernel arithmetic: equal bits
                assert torch.equal(yt[t], ref), (key, t)
            else:                                          # 32 taps: the streaming kernel (fused) against the matrix cores
                assert (yt[t] - ref).abs().max().item() <= TOL * ref.abs().max().item(), (key, t)
        got_state = bank.get_state()
        assert util.bits_equal(got_state[0], want_state[0]), key        # carried state / delay line: bit-exact either way
        assert np.array_equal(got_state[2], want_state[2]), key
        if iir:
            assert util.bits_equal(got_state[1], want_state[1]), key
        bank.close()
        del yt, want_rows
