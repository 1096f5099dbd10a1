"""RTFilterStream (include/RTFilterStream.h) against the oracle: one filter, one long stream, split in
time on the GPU (BASELINE.json config 5's single-GPU leg), plus the state hand-off entry points the
multi-GPU split uses.  Tolerance: 1e-12 relative to the output peak (north star)."""
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


def _bw8():
    _, b, a = util.design_coefs("bw8_bp_0.25_0.75")   # order 16: config 5's filter
    return b, a


def _designs():
    _, b4, a4 = util.design_coefs("bw4_bp_0.3_0.6")
    b8, a8 = _bw8()
    return {"bw4": (b4, a4), "bw8": (b8, a8)}


@pytest.mark.parametrize("name", ["bw4", "bw8"])
@pytest.mark.parametrize("blocks", [[1 << 20], [300000, 5, 700001, 40000]])
def test_stream_updaten_matches_oracle(G, name, blocks):
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    b, a = _designs()[name]
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    assert fs.decay_length > 0
    ref = orc.iir(b, a)
    pos = 0
    for n in blocks:
        x = util.synth(11, 3, pos, n)
        xd = torch.from_numpy(x).cuda()
        yd = fs(xd)
        torch.cuda.synchronize()
        want = ref.updaten(x)
        err = np.max(np.abs(yd.cpu().numpy() - want)) / np.max(np.abs(want))
        assert err <= TOL, (name, n, err)
        if n >= 300000:
            assert fs.last_split >= 8        # really went through the time split
        pos += n
    st, fv, ini = fs.get_state()
    assert np.max(np.abs(st[:len(ref.state)] - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state)))
    assert abs(fv - ref.filtered_value) <= TOL * max(1.0, abs(ref.filtered_value))
    assert ini == ref.initialized == 0
    fs.close()


def test_stream_in_place_and_reset(G):
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    b, a = _designs()["bw4"]
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    ref = orc.iir(b, a)
    for rep in range(2):
        x = util.synth(5, 0, 0, 400000)
        xd = torch.from_numpy(x).cuda()
        fs(xd, out=xd)
        torch.cuda.synchronize()
        want = ref.updaten(x)
        assert np.max(np.abs(xd.cpu().numpy() - want)) <= TOL * np.max(np.abs(want))
        fs.reset()                            # RTFilter_reset: the stale state stays (ref :55-58)
        ref.reset()
    fs.close()


def test_stream_short_and_fir_paths(G):
    """Calls too short to split and FIR designs run as a one-channel bank."""
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    b, a = _designs()["bw4"]
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    ref = orc.iir(b, a)
    x = util.synth(1, 0, 0, 5000)
    y = fs(torch.from_numpy(x).cuda()).cpu().numpy()
    want = ref.updaten(x)
    assert fs.last_split == 1
    assert np.max(np.abs(y - want)) <= TOL * np.max(np.abs(want))
    fs.close()
    _, bf, _ = util.design_coefs("fir512_lp_0.25")
    ff = gstream.FilterStream(gbank.fir_design(bf))
    assert ff.decay_length == 0
    rf = orc.fir(bf)
    x = util.synth(2, 0, 0, 300000)
    y = ff(torch.from_numpy(x).cuda()).cpu().numpy()
    want = rf.updaten(x)
    assert np.max(np.abs(y - want)) <= TOL * np.max(np.abs(want))
    ff.close()


def test_tail_state_handoff_equals_one_long_run(G):
    """The multi-GPU protocol on one GPU: two FilterStream objects play rank 0 and rank 1."""
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    b, a = _designs()["bw8"]
    n = 1 << 19
    x = util.synth(9, 0, 0, 2 * n)
    xd = torch.from_numpy(x).cuda()
    r0 = gstream.FilterStream(gbank.iir_design(b, a))
    r1 = gstream.FilterStream(gbank.iir_design(b, a))
    pack = r0.new_pack()
    assert r0.tail_state(xd[:n], pack)        # rank 0: hand-off vector of its own chunk
    ref0 = orc.iir(b, a)                      # ... = state, filtered_value, initialized of a filter after that chunk
    y_ref0 = ref0.updaten(x[:n])
    pk = pack.cpu().numpy()
    ns = len(ref0.state)
    assert np.max(np.abs(pk[:ns] - ref0.state)) <= TOL * max(1.0, np.max(np.abs(ref0.state)))
    assert abs(pk[-2] - y_ref0[-1]) <= TOL * np.max(np.abs(y_ref0)) and pk[-1] == 0.0
    r1.import_state(pack)                     # rank 1 starts from it ...
    y1 = r1(xd[n:])                           # ... and both filter concurrently
    y0 = r0(xd[:n])
    torch.cuda.synchronize()
    want = orc.iir(b, a).updaten(x)
    got = np.concatenate([y0.cpu().numpy(), y1.cpu().numpy()])
    assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))
    # chunk shorter than the decay length: the shortcut is refused, the caller chains
    assert not r0.tail_state(xd[:r0.decay_length - 1], pack)
    # export/import of the true state
    r0.export_state(pack)
    r1.import_state(pack)
    s0, s1 = r0.get_state(), r1.get_state()
    assert util.bits_equal(s0[0], s1[0]) and s0[1:] == s1[1:]
    r0.close()
    r1.close()


def test_stream_from_filter_and_sync_to_host(G):
    import torch
    from dsp_amd import dsp
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    f = dsp.butterworth(4, 0.3, 0.6)
    b, a = np.array(f.b(), dtype=np.float64), np.array(f.a(), dtype=np.float64)
    ref = orc.iir(b, a)
    head = util.synth(3, 0, 0, 1000)
    f(list(head))                             # host filter advances first; the stream takes its state over
    ref.updaten(head)
    fs = gstream.FilterStream.from_filter(f)
    x = util.synth(3, 0, 1000, 500000)
    y = fs(torch.from_numpy(x).cuda()).cpu().numpy()
    want = ref.updaten(x)
    assert np.max(np.abs(y - want)) <= TOL * np.max(np.abs(want))
    fs.sync_to_host()
    assert np.max(np.abs(np.array(f.state()) - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state)))
    assert f.initialized == 0
    fs.close()


@pytest.mark.parametrize("design", ["bw8", "lp11", "comb10"])
def test_long_stream_in_short_sub_chunks_with_in_kernel_run_ups(G, design):
    """Streams of orders 9..16 longer than 1024 x 2^17 samples are cut into sub-chunks of 2^17 samples whose start
    states iir_scan16_kernel computes itself (the zero-state response of the 4096 samples in front of each).  Config
    5's band-pass (numerator taps b[odd] == 0: the even-b instantiation; decay length 1024: the run-up reads the
    last 1024 samples), an order-11 low-pass (no zero taps) and an order-10 comb whose decay length is 4096 (the
    run-up reads the whole segment): a fresh stream, a ragged continuation, and an in-place call (which must fall back to the run-up launch: the
    samples in front of a sub-chunk would already be outputs) -- all within 1e-12 of the oracle, state included."""
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import dsp
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    if design == "bw8":
        b, a = _bw8()
        assert all(v == 0.0 for v in b[1::2])
    elif design == "lp11":
        f = dsp.butterworth(11, 0.0, 0.35)
        b, a = np.array(f.b()), np.array(f.a())
    else:                                              # poles on a circle of radius 0.985: 0.86^(M/10) < 2^-80 from M = 3680
        a = np.zeros(11); a[0], a[10] = 1.0, -0.86
        b = np.zeros(11); b[0], b[1], b[10] = 0.2, 0.1, 0.05
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    assert 0 < fs.decay_length <= 4096
    assert (fs.decay_length > 1024) == (design == "comb10"), fs.decay_length
    ref = orc.iir(b, a)
    pos = 0
    for i, n in enumerate(((1 << 27) + 3 * (1 << 17) + 777, (1 << 27) + (1 << 18), (1 << 27) + (1 << 18))):
        x = util.synth(91, 0, pos, n)
        want = ref.updaten(x)
        xd = torch.from_numpy(x).cuda()
        if i == 2:
            fs.updaten_ptr(xd.data_ptr(), xd.data_ptr(), n)
            yd = xd
            # in place: short sub-chunks too where the run-up is the last 1024 samples (set aside before the launch);
            # the comb's whole-segment run-ups keep the long sub-chunks and the run-up launch
            assert (fs.last_split == 1024) == (design == "comb10"), fs.last_split
        else:
            yd = fs(xd)
            assert fs.last_split > 1024, fs.last_split
        got = yd.cpu().numpy()
        del xd, yd
        assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want)), (design, i)
        st, fv, ini = fs.get_state()
        assert np.max(np.abs(st[:len(ref.state)] - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state)))
        assert abs(fv - ref.filtered_value) <= TOL * max(1.0, abs(ref.filtered_value)) and ini == 0
        pos += n
    fs.close()


def _resonator(radius=0.9999, theta=0.3 * np.pi):
    """Second-order resonator with its pole pair at `radius`: well conditioned (passes the scan probe) and slow to
    forget -- radius^M < 2^-80 needs M = 5.5e5 samples, so the decay shortcut cannot split a 2^24-sample call."""
    b = np.array([(1.0 - radius) * 0.5, 0.0, -(1.0 - radius) * 0.5])
    a = np.array([1.0, -2.0 * radius * np.cos(theta), radius * radius])
    return b, a


def test_slow_decay_design_is_split_by_composing_chunk_maps(G):
    """A design that does not forget within a sub-chunk: the chunk maps s_end = A^L s_start + v are composed
    exactly (no truncation) instead of running the stream as one workgroup.  2^24 samples in one call and a second,
    ragged call that continues it, <= 1e-12 of the oracle relative to the output peak, state included."""
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    b, a = _resonator()
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    assert fs.exact_split and (fs.decay_length == 0 or fs.decay_length >= (1 << 19))
    ref = orc.iir(b, a)
    pos = 0
    for n, want_split in (((1 << 24), 8), ((1 << 22) + 12345, 8)):
        x = util.synth(77, 0, pos, n)
        y = fs(torch.from_numpy(x).cuda()).cpu().numpy()
        want = ref.updaten(x)
        assert fs.last_split >= want_split, fs.last_split
        assert np.max(np.abs(y - want)) <= TOL * np.max(np.abs(want))
        st, fv, ini = fs.get_state()
        assert np.max(np.abs(st[:len(ref.state)] - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state)))
        assert ini == 0
        pos += n
    # reset: the next call is fresh again (sample 0 goes to stable_init) and is still split
    fs.reset()
    ref.reset()
    x = util.synth(78, 0, 0, 1 << 21)
    y = fs(torch.from_numpy(x).cuda()).cpu().numpy()
    want = ref.updaten(x)
    assert fs.last_split >= 8 and np.max(np.abs(y - want)) <= TOL * np.max(np.abs(want))
    fs.close()


def _rank_worker(rank, world, port, n, q, design="bw8"):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, a = _bw8() if design == "bw8" else _resonator()
        fs = gstream.FilterStream(gbank.iir_design(b, a), device=0)
        outs = []
        for blk in range(2):
            lo, hi = gstream.time_range(n, world, rank)
            x = torch.from_numpy(util.synth(21, 0, blk * n + lo, hi - lo)).cuda()
            # with and without the static plan; the second block IN PLACE (in the composed-maps mode the middle
            # ranks run two passes over x: the first must not overwrite it)
            y = gstream.split_update(fs, x, x if blk else None, total=n if blk else None)
            torch.cuda.synchronize()
            parts = [None] * world
            dist.all_gather_object(parts, y.cpu().numpy())
            outs.append(np.concatenate(parts))
        gstream.sync_end_state(fs)
        st = fs.get_state()[0]
        if rank == 0:
            q.put((np.concatenate(outs), st))
        fs.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("design,world", [("bw8", 2), ("resonator", 3)], ids=["tail-handoff", "composed-maps"])
def test_ranks_on_one_gpu_split_the_stream(G, design, world):
    """Rehearsal of the multi-GPU leg: several processes (gloo for the hand-off, all on cuda:0) each own one
    contiguous chunk of every block of the stream; the result equals one long run of the reference.  The
    16th-order band-pass hands over the decay tail (one all_gather); the slow resonator composes chunk maps."""
    import socket
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    n = (1 << 20) + 4096
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, n, q, design)) for r in range(world)]
    for p in procs:
        p.start()
    got, st = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b, a = _bw8() if design == "bw8" else _resonator()
    ref = orc.iir(b, a)
    want = ref.updaten(util.synth(21, 0, 0, 2 * n))
    assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))
    assert np.max(np.abs(st[:len(ref.state)] - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state)))


def _c_rank_worker(rank, world, port, n, q, design):
    """The same rehearsal through the C entry RTFilterStream_split_updaten: the collective is a callback on device
    pointers (here: hipMemcpy to the host + a gloo all_gather), everything else is the library's."""
    import ctypes as C
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]

        def allgather(send, recv, count, stream_ptr):
            assert hip.hipStreamSynchronize(stream_ptr) == 0
            mine = torch.empty(count, dtype=torch.float64)
            assert hip.hipMemcpy(mine.data_ptr(), send, count * 8, 2) == 0          # device -> host
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            every = torch.cat(parts)
            assert hip.hipMemcpy(recv, every.data_ptr(), world * count * 8, 1) == 0  # host -> device
            return 0

        b, a = _bw8() if design == "bw8" else _resonator()
        fs = gstream.FilterStream(gbank.iir_design(b, a), device=0)
        lens = [hi - lo for lo, hi in (gstream.time_range(n, world, r) for r in range(world))]
        outs = []
        for blk in range(2):
            lo, hi = gstream.time_range(n, world, rank)
            x = torch.from_numpy(util.synth(21, 0, blk * n + lo, hi - lo)).cuda()
            y = fs.split_updaten(x, x if blk else None, lens, rank, allgather)      # the second block in place
            torch.cuda.synchronize()
            parts = [None] * world
            dist.all_gather_object(parts, y.cpu().numpy())
            outs.append(np.concatenate(parts))
        fs.split_end_state(rank, world, allgather)
        st = fs.get_state()[0]
        if rank == 0:
            q.put((np.concatenate(outs), st))
        fs.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("design,world", [("bw8", 2), ("resonator", 3)], ids=["tail-handoff", "composed-maps"])
def test_ranks_on_one_gpu_split_the_stream_through_the_c_entry(G, design, world):
    """RTFilterStream_split_updaten / _split_end_state (the multi-GPU leg as a C caller drives it): two blocks of a
    stream, each cut over the ranks, the second in place; equals one long run of the reference, end state too."""
    import socket
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    n = (1 << 20) + 4096
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c_rank_worker, args=(r, world, port, n, q, design)) for r in range(world)]
    for p in procs:
        p.start()
    got, st = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b, a = _bw8() if design == "bw8" else _resonator()
    ref = orc.iir(b, a)
    want = ref.updaten(util.synth(21, 0, 0, 2 * n))
    assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))
    assert np.max(np.abs(st[:len(ref.state)] - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state)))


def test_host_buffers_through_the_gpu(G):
    """RTFilterStream_updaten_host: pageable host arrays, several chunks, state carried across calls, in place."""
    import time
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    b, a = _designs()["bw8"]
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    ref = orc.iir(b, a)
    pos = 0
    for n in ((1 << 24) + 12345, 1000, (1 << 23) * 3):
        x = util.synth(31, 0, pos, n)
        t0 = time.perf_counter()
        y = fs.filter_host(x)
        dt = time.perf_counter() - t0
        want = ref.updaten(x)
        assert np.max(np.abs(y - want)) <= TOL * np.max(np.abs(want))
        if n > (1 << 24):
            print("host round trip: %.0f Msamples/s (%.1f GB/s each way)" % (n / dt / 1e6, n * 8 / dt / 1e9))
        pos += n
    x = util.synth(31, 0, pos, 1 << 22)
    want = ref.updaten(x)
    fs.filter_host(x, out=x)
    assert np.max(np.abs(x - want)) <= TOL * np.max(np.abs(want))
    fs.close()
