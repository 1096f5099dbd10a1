#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json config 2.

  1024-channel bank of 8th-order IIR band-pass (butterworth(4, 0.3, 0.6): na = nb = 9),
  float64, 2^20 samples per channel resident in HBM, streamed in 64K-sample blocks with the
  filter state carried on the device between blocks.

One STEP = one RTFilterBank_updaten call = one 65 536-sample block for all 1024 channels of this
rank's bank (512 MiB read + 512 MiB written).  Steps walk the 16 blocks of the stream in order;
when the stream wraps the bank is RTFilter_reset (inside the timed region).

N GPUs: one process per GPU (torch.distributed / RCCL only for the barrier and the max-reduce of the
time; the path has no data-path collective).  The bank is channel-sharded: rank r owns channels
[r*1024, (r+1)*1024) => weak scaling; value = samples all ranks filtered / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CHANNELS = 1024
STREAM = 1 << 20
BLOCK = 1 << 16
NBLOCKS = STREAM // BLOCK
PITCH = STREAM + 128             # row pitch in HBM (doubles): 1 KiB past a power of two, so that the 1024
                                 # rows do not all start on the same HBM channel
SEED = 0x5EED0002
DESIGN = "bw4_bp_0.3_0.6"
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_SAMPLE = 16            # 8 B read x + 8 B write y


def _init_dist(dist, local):
    """RCCL (backend "nccl") with one rank per GPU.  BENCH_DIST_BACKEND=gloo together with BENCH_ONE_GPU=1 is a
    rehearsal switch: several ranks share cuda:0 (RCCL refuses that) so that the N > 1 code path of this script
    can be exercised on a one-GPU box; it is never what the driver runs."""
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)


def _reduce_device(dist, dev):
    return dev if dist.get_backend() == "nccl" else torch.device("cpu")


def cpu_baseline(b, a):
    """Reference (oracle/_ref, if it travelled) or the oracle port, on the host cores: every thread
    filters CHANNELS/threads channels of 2^20 samples with its own filter object."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from oracle import ref
    from tests import util
    threads = max(1, min(16, os.cpu_count() or 1))
    per_thread = max(1, CHANNELS // threads)          # bounded sample: one pass of the 1024 streams
    kind = "reference" if ref.available() else "port"
    xs = [util.synth(SEED, t, 0, STREAM) for t in range(threads)]

    def work(t):
        for _ in range(per_thread):
            if kind == "reference":
                ref.RefIIR(b, a).updaten(xs[t])
            else:
                orc.iir(b, a).updaten(xs[t])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    nsamp = threads * per_thread * STREAM
    # SURVEY.md 8(d) column (a): one thread, one channel, 2^22 samples, median of 5 after a warm-up
    x1 = util.synth(SEED, 0, 0, 1 << 22)
    one = []
    for rep in range(6):
        f = ref.RefIIR(b, a) if kind == "reference" else orc.iir(b, a)
        t1 = time.perf_counter()
        f.updaten(x1)
        one.append(time.perf_counter() - t1)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=nsamp / dt / 1e6, unit="Msamples/s", cores=threads, kind=kind,
                single_thread_value=(1 << 22) / float(np.median(one[1:])) / 1e6,
                cpu_model=model, host_cpus_visible=os.cpu_count(),
                sample="%d channels x 2^20 samples of config 2 (%s), RTFilter_updaten per channel, "
                       "%d host threads (one channel range each, SMT siblings not distinguished), %.1f s; "
                       "single_thread_value: one channel, 2^22 samples, median of 5"
                       % (threads * per_thread, DESIGN, threads, dt))


def device_state_under_load(run_step, sync):
    """Clocks and power as rocm-smi reports them WHILE the hot path runs (untimed; the pool's boxes differ by a
    few per cent from process to process: the copy ceiling measured in the same run is the yardstick, these
    readings are the context for it).  None where rocm-smi is missing or refuses."""
    import re
    import subprocess
    import threading
    out = {}

    def probe():
        try:
            r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True,
                               timeout=20)
            out["text"] = r.stdout
        except Exception as exc:                         # noqa: BLE001  (diagnostic only)
            out["error"] = repr(exc)

    th = threading.Thread(target=probe)
    th.start()
    t0 = time.perf_counter()
    while th.is_alive() and time.perf_counter() - t0 < 20.0:
        for _ in range(64):
            run_step()
        sync()
    th.join()
    txt = out.get("text")
    if not txt:
        return None
    state = {}
    for key, pat in (("mclk_mhz", r"mclk clock level: \d+: \((\d+)Mhz\)"), ("sclk_mhz", r"sclk clock level: \d+: \((\d+)Mhz\)"),
                     ("fclk_mhz", r"fclk clock level: \d+: \((\d+)Mhz\)"),
                     ("power_w", r"Power \(W\): ([0-9.]+)")):
        m = re.search(pat, txt)
        if m:
            state[key] = float(m.group(1))
    return state or None


def measured_traffic(name="r3_traffic.json"):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes (profiles/:
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, KiB -> bytes).  A stored constant, not
    something this run measured (bench.py does not run the profiler): the bench line names the file under
    `traffic_source`.  None if the profile file is absent."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as fh:
            return json.load(fh)["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--mode", choices=["auto", "scan", "seq"], default="auto")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--jitter", action="store_true",
                    help="config 2 variant of SURVEY.md 8(d): every channel its own butterworth(4, wl, wu), band edges "
                         "jittered inside [0.25, 0.65], so that no coefficient table is shared")
    ap.add_argument("--workload", choices=["iir8", "fir512", "stream16", "mixed"], default="iir8",
                    help="iir8 = BASELINE.json config 2 (the headline line); fir512 = config 3; "
                         "stream16 = config 5 (one 2^30-sample stream, 16th-order IIR, split in time)")
    ap.add_argument("--stream-log2", type=int, default=30)
    ap.add_argument("--mixed-channels", type=int, default=65536,
                    help="config 4: channels per wave (SURVEY.md 8(d): 2^16)")
    ap.add_argument("--mixed-total", type=int, default=1 << 20, help="config 4: channels of the whole bank (2^20)")
    ap.add_argument("--mixed-layout", choices=["sorted", "interleaved"], default="sorted")
    ap.add_argument("--gather", action="store_true",
                    help="config 4: all_gather one WAVE-SIZED output block per step over RCCL/xGMI (every rank "
                         "contributes wave/N channels x 64K samples of the wave it just filtered; 32 GiB assembled "
                         "on every rank), on its own stream, overlapped with the next wave's filtering; reports "
                         "filter-only vs filter+gather and GB/s per xGMI link")
    ap.add_argument("--gather-fval", action="store_true",
                    help="config 4: all_gather one double per channel (the last output) after each block instead")
    ap.add_argument("--fir-channels", type=int, default=65536)
    ap.add_argument("--layout", choices=["tiles", "blocks", "rows"], default="tiles",
                    help="config 2: how the resident 2^20-sample streams lie in HBM: `rows` = one row of 2^20 samples per "
                         "channel, `blocks` = 16 block buffers of 1024 x 64K (dense rows), `tiles` = 16 block buffers in the "
                         "tiled block layout [32][1024][2048] (RTFilterBank_updaten_tiled)")
    ap.add_argument("--chained", action="store_true",
                    help="config 2: hand the blocks over with RTFilterBank_updaten_chained (block k+1 overlaps the drain "
                         "of block k on the bank's internal streams) instead of in stream order")
    ap.add_argument("--block-pad", type=int, default=0, help="layout blocks: extra doubles per row")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed: keep the GPU busy with the hot path for this long BEFORE the warm-up steps, so "
                         "that the shader/memory clocks have left their idle state (a step takes ~4 %% longer "
                         "during the first ~25 ms after an idle period: tools/rampup.py); 0 disables")
    ap.add_argument("--no-others", action="store_true",
                    help="config 2: skip the untimed legs that time the other hand-overs (rows / chained) and the cold "
                         "pass -- profiling runs, so that every launch of the dominant kernel has the timed shape")
    ap.add_argument("--no-extra", action="store_true",
                    help="default run only: skip the compact config-3 and config-5 legs attached under `other_configs`")
    args = ap.parse_args()
    if args.workload == "fir512":
        return main_fir512(args)
    if args.workload == "stream16":
        return main_stream16(args)
    if args.workload == "mixed":
        return main_mixed(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":     # the latter: RCCL code path with ONE rank (tests)
        import torch.distributed as dist
        _init_dist(dist, local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from dsp_amd import bank as gbank
    from tests import util
    _, b, a = util.design_coefs(DESIGN)
    mode = dict(auto=gbank.MODE_AUTO, scan=gbank.MODE_SCAN, seq=gbank.MODE_SEQUENTIAL)[args.mode]
    if args.jitter:
        from dsp_amd import dsp
        rng = np.random.default_rng(SEED + rank)
        wl = 0.25 + 0.1 * rng.random(CHANNELS)
        wu = wl + 0.25 + 0.05 * rng.random(CHANNELS)
        jb, ja, _ = dsp.design_batch("butterworth", 4, wl, wu)       # host threads, bits of dsp.butterworth
        bank = gbank.FilterBank.from_arrays(gbank.KIND_IIR, jb[:, :9], ja[:, :9], device=local, mode=mode)
    else:
        bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CHANNELS, device=local, mode=mode)

    tiled = args.layout == "tiles"
    if tiled:
        # the stream as it is streamed, every 64K block in the tiled block layout [32 segments][1024 channels][2048]
        pitch = gbank.TILE
        xbuf = torch.empty((NBLOCKS, BLOCK // gbank.TILE, CHANNELS, gbank.TILE), dtype=torch.float64, device=dev)
        ybuf = torch.empty_like(xbuf)
        rows_tmp = torch.empty((CHANNELS, BLOCK), dtype=torch.float64, device=dev)
        for j in range(NBLOCKS):
            gbank.synth_fill(rows_tmp, SEED, chan0=rank * CHANNELS, n0=j * BLOCK)
            gbank.tile_rows(rows_tmp, xbuf[j])
        del rows_tmp
        xptr = [xbuf[j].data_ptr() for j in range(NBLOCKS)]
        yptr = [ybuf[j].data_ptr() for j in range(NBLOCKS)]
    elif args.layout == "rows":
        # every channel's 2^20 samples contiguous (row pitch 2^20 + 128 doubles); block j = columns [j*64K, (j+1)*64K)
        pitch = PITCH
        xbuf = torch.empty((CHANNELS, PITCH), dtype=torch.float64, device=dev)
        ybuf = torch.empty((CHANNELS, PITCH), dtype=torch.float64, device=dev)
        gbank.synth_fill(xbuf[:, :STREAM], SEED, chan0=rank * CHANNELS, n0=0)
        xptr = [xbuf.data_ptr() + j * BLOCK * 8 for j in range(NBLOCKS)]
        yptr = [ybuf.data_ptr() + j * BLOCK * 8 for j in range(NBLOCKS)]
    else:
        # the stream as it is streamed: 16 block buffers of 1024 x 64K (each 512 MiB, rows dense), all resident
        pitch = BLOCK + args.block_pad
        xbuf = torch.empty((NBLOCKS, CHANNELS, pitch), dtype=torch.float64, device=dev)
        ybuf = torch.empty((NBLOCKS, CHANNELS, pitch), dtype=torch.float64, device=dev)
        for j in range(NBLOCKS):
            gbank.synth_fill(xbuf[j][:, :BLOCK], SEED, chan0=rank * CHANNELS, n0=j * BLOCK)
        xptr = [xbuf[j].data_ptr() for j in range(NBLOCKS)]
        yptr = [ybuf[j].data_ptr() for j in range(NBLOCKS)]
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream()
    step_no = [0]

    def step(ev=None):
        j = step_no[0] % NBLOCKS
        if j == 0 and step_no[0] > 0:
            bank.reset(stream)
        if ev is not None:
            ev[0].record(stream)
        if args.chained:
            bank.updaten_chained_ptr(yptr[j], pitch, xptr[j], pitch, BLOCK, tiled=tiled, stream=stream)
            if ev is not None:
                bank.chain_join(stream)
        elif tiled:
            bank.updaten_tiled_ptr(yptr[j], xptr[j], BLOCK, stream)
        else:
            bank.updaten_ptr(yptr[j], pitch, xptr[j], pitch, BLOCK, stream)
        if ev is not None:
            ev[1].record(stream)
        step_no[0] += 1

    # untimed extra, first thing after an idle pause: the same K steps from cold clocks (no settle pass, no warm-up)
    cold_value = None
    if args.settle_ms > 0 and not args.no_others:
        time.sleep(0.25)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(args.steps):
            step()
        if args.chained:
            bank.chain_join(stream)
        c1.record(stream)
        torch.cuda.synchronize()
        cold_value = CHANNELS * BLOCK * args.steps / (c0.elapsed_time(c1) / 1e3) / 1e6
        step_no[0] = 0
        bank.reset(stream)
    # untimed, before the warm-up: bring the device out of its idle clocks (profiles/r2_rampup.txt)
    settle_steps = 0
    t_settle = time.perf_counter()
    while args.settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(32):
            step()
        settle_steps += 32
        torch.cuda.synchronize()
    dev_state = None
    if rank == 0 and args.settle_ms > 0 and not args.no_others:
        dev_state = device_state_under_load(step, torch.cuda.synchronize)
    if settle_steps:
        step_no[0] = 0
        bank.reset(stream)
    for _ in range(args.warmup):
        step()
    # ONE pair of HIP events brackets the timed steps on the launch stream: average launch duration =
    # (e_end - e_begin) / steps.  (An event pair around every launch puts two marker packets between consecutive
    # kernels and costs ~6 us of wall time per step; the per-launch pairs are taken in an untimed pass below.)
    e_begin, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e_begin.record(stream)
    for k in range(args.steps):
        step()
    if args.chained:
        bank.chain_join(stream)              # `stream` waits for the last link: inside the timed region
    e_end.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=_reduce_device(dist, dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_avg_s = e_begin.elapsed_time(e_end) / 1e3 / args.steps

    # untimed extra: per-launch event pairs (spread of single launches)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(16)]
    for k in range(16):
        step(events[k])
    torch.cuda.synchronize()
    kern_ms = [e0.elapsed_time(e1) for e0, e1 in events]

    # Streaming-copy ceiling of this device for the same shape (1024 rows x one 64K block, read + write,
    # no arithmetic): what a 16 B/sample kernel could reach at best here.  Untimed extra, rank 0 only.
    copy_gbs = None
    if rank == 0:
        import ctypes as C
        from dsp_amd.cdsp import dspdll
        cts = []
        for k in range(12):
            j = k % NBLOCKS
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if tiled:
                rc = dspdll.dsp_copy_tiles(C.c_void_p(yptr[j]), C.c_void_p(xptr[j]), CHANNELS, BLOCK,
                                           C.c_void_p(stream.cuda_stream))
            else:
                rc = dspdll.dsp_copy_rows(C.c_void_p(yptr[j]), pitch, C.c_void_p(xptr[j]), pitch, CHANNELS, BLOCK,
                                          256, 8192, C.c_void_p(stream.cuda_stream))
            e1.record(stream)
            torch.cuda.synchronize()
            if rc == 0 and k >= 4:
                cts.append(e0.elapsed_time(e1))
        if cts:
            copy_gbs = CHANNELS * BLOCK * BYTES_PER_SAMPLE / (float(np.median(cts)) / 1e3) / 1e9

    # untimed extra, rank 0: the other ways the same blocks can be handed over (same bank design, fresh banks)
    others = {}
    if rank == 0 and not args.jitter and not args.no_others:
        def jitter_bank():
            # SURVEY.md 8(d)'s variant: every channel its own butterworth(4, wl, wu), band edges jittered inside
            # [0.25, 0.65] -- no coefficient table is shared (the --jitter run of this script, compact)
            from dsp_amd import dsp
            rng = np.random.default_rng(SEED + rank)
            wl = 0.25 + 0.1 * rng.random(CHANNELS)
            wu = wl + 0.25 + 0.05 * rng.random(CHANNELS)
            jb, ja, _ = dsp.design_batch("butterworth", 4, wl, wu)
            return gbank.FilterBank.from_arrays(gbank.KIND_IIR, jb[:, :9], ja[:, :9], device=local, mode=mode)

        def leg(layout_tiled, chained, jitter=False):
            bk = jitter_bank() if jitter else gbank.FilterBank([gbank.iir_design(b, a)], [0] * CHANNELS, device=local, mode=mode)
            if layout_tiled == tiled:
                xs, ys, p = xptr, yptr, pitch
            else:                             # four block buffers in the other layout (re-used by the 16 blocks)
                shape = (4, BLOCK // gbank.TILE, CHANNELS, gbank.TILE) if layout_tiled else (4, CHANNELS, BLOCK)
                xo = torch.empty(shape, dtype=torch.float64, device=dev).normal_()
                yo = torch.empty_like(xo)
                xs = [xo[j % 4].data_ptr() for j in range(NBLOCKS)]
                ys = [yo[j % 4].data_ptr() for j in range(NBLOCKS)]
                p = gbank.TILE if layout_tiled else BLOCK
            t0e, t1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for timed in (False, True):
                if timed:
                    t0e.record(stream)
                for k in range(32):
                    j = k % NBLOCKS
                    if chained:
                        bk.updaten_chained_ptr(ys[j], p, xs[j], p, BLOCK, tiled=layout_tiled, stream=stream)
                    elif layout_tiled:
                        bk.updaten_tiled_ptr(ys[j], xs[j], BLOCK, stream)
                    else:
                        bk.updaten_ptr(ys[j], p, xs[j], p, BLOCK, stream)
                if chained:
                    bk.chain_join(stream)
            t1e.record(stream)
            torch.cuda.synchronize()
            bk.close()
            return CHANNELS * BLOCK * 32 / (t0e.elapsed_time(t1e) / 1e3) / 1e6
        for name, lt, ch in (("dense rows (RTFilterBank_updaten), stream order", False, False),
                             ("dense rows, RTFilterBank_updaten_chained", False, True),
                             ("tiles (RTFilterBank_updaten_tiled), stream order", True, False),
                             ("tiles, RTFilterBank_updaten_chained", True, True)):
            if (lt, ch) != (tiled, args.chained):
                others[name] = leg(lt, ch)
        others["1024 distinct designs (--jitter), %s, stream order" % ("tiles" if tiled else "dense rows")] = \
            leg(tiled, False, jitter=True)

    if rank == 0:
        samples_per_step = CHANNELS * BLOCK
        value = world * samples_per_step * args.steps / elapsed / 1e6
        achieved = samples_per_step * BYTES_PER_SAMPLE / kern_avg_s / 1e9
        line = {
            "metric": "filtered Msamples/s (whole node)", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 2: 1024-channel 8th-order IIR band-pass bank "
                                   "(butterworth(4,0.3,0.6)), 2^20 samples/channel resident, "
                                   "64K-sample blocks, state carried in HBM",
                       "channels_per_gpu": CHANNELS, "block": BLOCK, "stream": STREAM,
                       "layout": args.layout + (": every 64K block stored [32 segments][1024 channels][2048 samples], "
                                                "RTFilterBank_updaten_tiled" if tiled else ""),
                       "hand_over": "RTFilterBank_updaten_chained" if args.chained else "stream order",
                       "row_pitch_doubles": pitch, "settle_steps_untimed": settle_steps,
                       "distinct_designs": CHANNELS if args.jitter else 1,
                       "kernel": args.mode, "parallelism": "channel-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(),
                         "traffic_source": "profiles/r3_traffic.json (rocprofv3 --pmc passes of this command, committed; "
                                           "not re-measured by this run)",
                         "algorithmic_bytes_per_launch": samples_per_step * BYTES_PER_SAMPLE,
                         "copy_ceiling_measured": copy_gbs,
                         "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                         "kernel_ms_avg": kern_avg_s * 1e3,
                         "single_launch_ms_avg": float(np.mean(kern_ms)), "single_launch_ms_min": float(np.min(kern_ms))},
            "cold_start_msamples_s": cold_value,
            "device_state_under_load": dev_state,
            "other_hand_overs_msamples_s": others,
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(b, a)
        if world == 1 and not args.no_extra and not args.jitter:
            # compact same-run results for configs 3 and 5 (a few steps each, no CPU leg), so that the driver's
            # default run also times the Toeplitz-MFMA FIR kernel and the single-stream split
            del bank, xbuf, ybuf
            torch.cuda.empty_cache()
            import copy
            other = {}
            a3 = copy.copy(args)
            a3.steps, a3.warmup, a3.no_cpu = 4, 1, True
            other["config3_fir512"] = _compact(main_fir512(a3, emit=False))
            torch.cuda.empty_cache()
            a5 = copy.copy(args)
            a5.steps, a5.warmup, a5.no_cpu = 8, 2, True
            other["config5_stream16"] = _compact(main_stream16(a5, emit=False))
            torch.cuda.empty_cache()
            # config 4, compact: ONE IIR-8 wave and ONE FIR-32 wave of 65 536 channels x 64K (the two kinds of wave
            # the 2^20-channel bank consists of), alternating
            a4 = copy.copy(args)
            a4.steps, a4.warmup, a4.no_cpu = 6, 2, True
            a4.mixed_total, a4.mixed_channels, a4.mixed_layout = 2 * 65536, 65536, "sorted"
            a4.gather = a4.gather_fval = False
            other["config4_mixed"] = _compact(main_mixed(a4, emit=False))
            line["other_configs"] = other
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _compact(line):
    keep = ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline")
    return {k: line[k] for k in keep if k in line}


def main_fir512(args, emit=True):
    """BASELINE.json config 3: 512-tap FIR low-pass, 65 536 channels x 64K samples per GPU.  One STEP =
    one RTFilterBank_updaten over the whole 65 536 x 65 536 block (32 GiB read + 32 GiB written), the
    state carried on (the stream simply continues with the same input block).  Dominant kernel:
    fir_mfma_kernel (Toeplitz tiles on v_mfma_f64_16x16x4); roofline bound = FP64 matrix peak with
    ALGORITHMIC flops = 2 x 512 per output sample."""
    FP64_MFMA_PEAK_TF = 78.6     # AMD MI355X data sheet, FP64 matrix (dense); the guide lists no f64 MFMA peak
    TAPS = 512
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":     # the latter: RCCL code path with ONE rank (tests)
        import torch.distributed as dist
        _init_dist(dist, local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from dsp_amd import bank as gbank
    from tests import util
    _, b, _ = util.design_coefs("fir512_lp_0.25")
    nch = args.fir_channels
    mode = dict(auto=gbank.MODE_AUTO, scan=gbank.MODE_AUTO, seq=gbank.MODE_SEQUENTIAL)[args.mode]
    bank = gbank.FilterBank([gbank.fir_design(b)], [0] * nch, device=local, mode=mode)
    x = torch.empty((nch, BLOCK), dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    gbank.synth_fill(x, SEED, chan0=rank * nch, n0=0)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        bank.updaten_ptr(y.data_ptr(), BLOCK, x.data_ptr(), BLOCK, BLOCK, stream)
        if ev is not None:
            ev[1].record(stream)

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=_reduce_device(dist, dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [e0.elapsed_time(e1) for e0, e1 in events]
    kern_avg_s = float(np.mean(kern_ms)) / 1e3
    if rank == 0:
        samples = nch * BLOCK
        flops = 2.0 * TAPS * samples
        achieved = flops / kern_avg_s / 1e12
        line = {
            "metric": "filtered Msamples/s (whole node)", "value": world * samples * args.steps / elapsed / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 3: 512-tap FIR low-pass (windowed sinc 0.25), %d channels x 64K "
                                   "samples, one block per step, state carried in HBM" % nch,
                       "channels_per_gpu": nch, "block": BLOCK, "taps": TAPS, "kernel": args.mode,
                       "parallelism": "channel-sharded x%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TF,
                         "traffic": measured_traffic("r3_fir512_traffic.json") if nch == 65536 else None,
                         "algorithmic_flops_per_launch": flops,
                         "hbm_gbs_at_16B_per_sample": samples * BYTES_PER_SAMPLE / kern_avg_s / 1e9,
                         "kernel_ms_avg": kern_avg_s * 1e3, "kernel_ms_min": float(np.min(kern_ms))},
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline_fir(b)
        if not emit:
            bank.close()
            return line
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def fv_needs_rows(args):
    """--gather-fval reads the last column of the output rows."""
    return bool(getattr(args, "gather_fval", False))


def main_mixed(args, emit=True):
    """BASELINE.json config 4 as SURVEY.md 8(d) pins it: a 2^20-channel bank, channels sorted by type -- the
    first 2^19 are the 8th-order IIR of config 2, the next 2^19 a 32-tap windowed-sinc FIR -- x 65 536 samples,
    contiguous channel ranges of 2^20/N per GPU, processed in WAVES of 2^16 channels (32 GiB in + 32 GiB out:
    the whole bank does not fit below 8 GPUs, and the same wave size at every GPU count keeps numbers comparable).
    One STEP = one RTFilterBank_updaten over one wave; steps walk the rank's waves, alternating types where the
    rank owns both (N = 1, 2), so that `value` averages over the bank's mix.  Every wave has its own bank (its own
    carried state); the sample buffer is shared between waves (its content does not affect the timing).
    `--mixed-layout interleaved` is the harder variant: four designs (IIR 8 / IIR 16 / FIR 32 / IIR 5) interleaved
    channel by channel, so that every kernel group gathers its channels through a list.
    With --gather the ranks all_gather the filtered_value of the wave's channels after each step (the real-time
    consumer's view); the sample blocks themselves stay sharded."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":     # the latter: RCCL code path with ONE rank (tests)
        import torch.distributed as dist
        _init_dist(dist, local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from dsp_amd import bank as gbank
    from tests import util
    mode = dict(auto=gbank.MODE_AUTO, scan=gbank.MODE_SCAN, seq=gbank.MODE_SEQUENTIAL)[args.mode]

    def design(key):
        kind, b, a = util.design_coefs(key)
        return gbank.fir_design(b) if kind != "iir" else gbank.iir_design(b, a)

    wave = args.mixed_channels                       # channels per wave (2^16)
    total = args.mixed_total
    per_rank = total // world
    nwaves = max(1, per_rank // wave)
    banks, kinds = [], []
    if args.mixed_layout == "sorted":
        for w in range(nwaves):
            g0 = rank * per_rank + w * wave
            key = "bw4_bp_0.3_0.6" if g0 < total // 2 else "fir32_lp_0.25"
            banks.append(gbank.FilterBank([design(key)], [0] * wave, device=local, mode=mode))
            kinds.append("iir8" if g0 < total // 2 else "fir32")
        order = sorted(range(nwaves), key=lambda w: (w % (nwaves // 2 or 1), w))   # alternate the two halves
    else:
        keys = ["bw4_bp_0.3_0.6", "bw8_bp_0.25_0.75", "fir32_lp_0.25", "bw5_lp_0.25"]
        for w in range(min(nwaves, 2)):
            banks.append(gbank.FilterBank([design(k) for k in keys], [c % 4 for c in range(wave)], device=local,
                                          mode=mode))
            kinds.append("interleaved")
        order = list(range(len(banks)))
    pitch = BLOCK + 128
    gather_rows = dist is not None and args.gather
    # --layout tiles (default): the wave's block in the tiled block layout [32 tiles][wave channels][2048]
    # (RTFilterBank_updaten_tiled; the IIR waves go round by round on the schedule).  The row gather and
    # the interleaved variant keep dense rows at pitch 64K + 128.
    tiled = args.layout == "tiles" and args.mixed_layout == "sorted" and not gather_rows and fv_needs_rows(args) is False
    nybuf = 2 if gather_rows else 1                  # the gather of wave w reads y while wave w+1 is written
    xbuf = torch.empty((wave, pitch), dtype=torch.float64, device=dev)
    ybufs = [torch.empty((wave, pitch), dtype=torch.float64, device=dev) for _ in range(nybuf)]
    x = xbuf[:, :BLOCK]
    gbank.synth_fill(x, 0x5EED0004, chan0=rank * per_rank, n0=0)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    fv_all = (torch.empty(world * wave, dtype=torch.float64, device=dev)
              if (dist is not None and args.gather_fval) else None)
    slice_rows = wave // world                       # this rank's contribution to the wave-sized gather
    gathered = gstream_g = None
    g_events, y_free = [], [None, None]
    if gather_rows:
        from dsp_amd import shard
        gathered = torch.empty((world * slice_rows, BLOCK), dtype=torch.float64, device=dev)
        send = torch.empty((slice_rows, BLOCK), dtype=torch.float64, device=dev)
        gstream_g = torch.cuda.Stream(device=dev)
    step_no = [0]

    def step(ev=None):
        k = step_no[0]
        bank = banks[order[k % len(order)]]
        ybuf = ybufs[k % nybuf]
        step_no[0] += 1
        if gather_rows and y_free[k % nybuf] is not None:
            stream.wait_event(y_free[k % nybuf])     # the gather that read this buffer two steps ago is done
        if ev is not None:
            ev[0].record(stream)
        if tiled:
            bank.updaten_tiled_ptr(ybuf.data_ptr(), xbuf.data_ptr(), BLOCK, stream)
        else:
            bank.updaten_ptr(ybuf.data_ptr(), pitch, x.data_ptr(), pitch, BLOCK, stream)
        if ev is not None:
            ev[1].record(stream)
        if fv_all is not None:
            dist.all_gather_into_tensor(fv_all, ybuf[:, BLOCK - 1].contiguous())
        if gather_rows:
            done = torch.cuda.Event()
            done.record(stream)
            with torch.cuda.stream(gstream_g):
                gstream_g.wait_event(done)
                g0 = torch.cuda.Event(enable_timing=True)
                g1 = torch.cuda.Event(enable_timing=True)
                g0.record(gstream_g)
                send.copy_(ybuf[:slice_rows, :BLOCK])              # dense rows for the collective
                shard.gather_rows_into(gathered, send)
                g1.record(gstream_g)
                if ev is not None:
                    g_events.append((g0, g1))
                y_free[k % nybuf] = g1

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    first_timed = step_no[0]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=_reduce_device(dist, dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = np.array([e0.elapsed_time(e1) for e0, e1 in events])
    avg_s = float(np.mean(step_ms)) / 1e3
    # Streaming-copy ceiling for the wave's shape and footprint (65 536 rows x 64K, 32 GiB each way): the
    # compute-free twin of the kernels' access pattern on the very buffers.  Untimed extra, rank 0 only.
    copy_gbs = None
    if rank == 0:
        import ctypes as C
        from dsp_amd.cdsp import dspdll
        cts = []
        for k in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if tiled:
                rc = dspdll.dsp_copy_tiles(C.c_void_p(ybufs[0].data_ptr()), C.c_void_p(xbuf.data_ptr()), wave, BLOCK,
                                           C.c_void_p(stream.cuda_stream))
            else:
                rc = dspdll.dsp_copy_rows(C.c_void_p(ybufs[0].data_ptr()), pitch, C.c_void_p(x.data_ptr()), pitch, wave,
                                          BLOCK, 256, 8192, C.c_void_p(stream.cuda_stream))
            e1.record(stream)
            torch.cuda.synchronize()
            if rc == 0 and k >= 1:
                cts.append(e0.elapsed_time(e1))
        if cts:
            copy_gbs = wave * BLOCK * BYTES_PER_SAMPLE / (float(np.median(cts)) / 1e3) / 1e9
    if rank == 0:
        samples = wave * BLOCK
        achieved = samples * BYTES_PER_SAMPLE / avg_s / 1e9
        per_kind = {}
        for k, ms in enumerate(step_ms):
            per_kind.setdefault(kinds[order[(first_timed + k) % len(order)]], []).append(ms)
        line = {
            "metric": "filtered Msamples/s (whole node)", "value": world * samples * args.steps / elapsed / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 4: %d-channel mixed bank (%s) x 64K samples, channel-sharded, waves of %d "
                                   "channels" % (total, "%d IIR order 8 + %d FIR 32 taps, sorted by type" % (total // 2, total - total // 2)
                                                 if args.mixed_layout == "sorted" else
                                                 "IIR 8 / IIR 16 / FIR 32 / IIR 5 interleaved", wave),
                       "channels_per_gpu": per_rank, "wave_channels": wave, "waves_per_gpu": nwaves, "block": BLOCK,
                       "layout": ("tiles: every wave's block stored [32 tiles][%d channels][2048 samples], "
                                  "RTFilterBank_updaten_tiled" % wave) if tiled else "dense rows",
                       "row_pitch_doubles": gbank.TILE if tiled else pitch, "kernel": args.mode,
                       "gather": "rows" if gather_rows else ("filtered_value" if fv_all is not None else None),
                       "rank0_wave_ms": {k: float(np.mean(v)) for k, v in per_kind.items()},
                       "parallelism": "channel-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic("r3_config4_traffic.json")
                                    if (tiled and wave == 65536 and args.mixed_layout == "sorted") else None,
                         "traffic_source": "profiles/r3_config4_traffic.json (committed PMC passes of --workload mixed; "
                                           "per wave, mean of the two kinds; not re-measured by this run)",
                         "algorithmic_bytes_per_launch": samples * BYTES_PER_SAMPLE,
                         "copy_ceiling_measured": copy_gbs,
                         "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                         "note": "all kernels of one RTFilterBank_updaten%s over one wave, rank 0's device time per "
                                 "call; copy ceiling = the row-per-workgroup copy twin in the same layout (unpaced)"
                                 % ("_tiled" if tiled else ""),
                         "step_ms_avg": avg_s * 1e3, "step_ms_min": float(np.min(step_ms))},
        }
        if gather_rows and g_events:
            g_ms = np.array([g0.elapsed_time(g1) for g0, g1 in g_events])
            recv = (world - 1) * slice_rows * BLOCK * 8                      # bytes this rank receives per gather
            line["gather"] = {
                "what": "all_gather (RCCL) of one wave-sized block per step: every rank contributes %d channels x 64K "
                        "samples (%.2f GiB) of the wave it just filtered, %.2f GiB assembled on every rank; own stream, "
                        "overlapped with the next wave's filtering" % (slice_rows, slice_rows * BLOCK * 8 / 2 ** 30,
                                                                        world * slice_rows * BLOCK * 8 / 2 ** 30),
                "gather_ms_avg": float(np.mean(g_ms)), "filter_ms_avg": avg_s * 1e3,
                "filter_plus_gather_ms_per_step": elapsed / args.steps * 1e3,
                "filter_only_msamples_s": world * samples / avg_s / 1e6,
                "filter_plus_gather_msamples_s": world * samples * args.steps / elapsed / 1e6,
                "bytes_received_per_rank": recv,
                "xgmi_gbs_per_rank": recv / (float(np.mean(g_ms)) / 1e3) / 1e9 if world > 1 else None,
                "xgmi_gbs_per_link": recv / (float(np.mean(g_ms)) / 1e3) / 1e9 / 7 if world > 1 else None,
            }
        if not emit:
            for bk in banks:
                bk.close()
            return line
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_stream16(args, emit=True):
    """BASELINE.json config 5: ONE 2^30-sample stream through a 16th-order IIR (butterworth(8, 0.25, 0.75),
    na = nb = 17).  One STEP = RTFilter_updaten over the whole stream: on N GPUs rank r holds samples
    [r L, (r+1) L) resident, the ranks exchange the state hand-off vectors (dsp_amd.stream.split_update:
    one all_gather of 19 doubles per rank + the end-state broadcast, RCCL over xGMI) and filter their
    chunks concurrently => STRONG scaling (total work fixed).  Consecutive steps continue the stream."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":     # the latter: RCCL code path with ONE rank (tests)
        import torch.distributed as dist
        _init_dist(dist, local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from tests import util
    _, b, a = util.design_coefs("bw8_bp_0.25_0.75")
    total = 1 << args.stream_log2
    lo, hi = gstream.time_range(total, world, rank)
    n = hi - lo
    fs = gstream.FilterStream(gbank.iir_design(b, a), device=local)
    x = torch.empty((1, n), dtype=torch.float64, device=dev)
    gbank.synth_fill(x, SEED, chan0=0, n0=lo)
    x = x.view(-1)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        if world > 1:
            gstream.split_update(fs, x, y, stream=stream, total=total)
        else:
            fs(x, y, stream)
        if ev is not None:
            ev[1].record(stream)

    # untimed, before the warm-up: bring the device out of its idle clocks (same count on every rank: the
    # multi-GPU step holds a collective)
    settle_steps = min(128, int(round(args.settle_ms / 3.7 * world))) if args.settle_ms > 0 else 0
    for _ in range(settle_steps):
        step()
    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=_reduce_device(dist, dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = [e0.elapsed_time(e1) for e0, e1 in events]
    avg_s = float(np.mean(step_ms)) / 1e3
    # untimed extra: what a plain copy of the same bytes reaches with the same access pattern -- rows = the stream's
    # sub-chunks, one workgroup per row, taken in dispatch order (tools/stream16_copy_ceiling.py)
    copy_gbs = None
    if world == 1 and not args.no_extra and fs.last_split >= 8 and n % fs.last_split == 0 \
            and (n // fs.last_split) % 2048 == 0:
        import ctypes as C
        from dsp_amd.cdsp import dspdll
        rows, ln = fs.last_split, n // fs.last_split
        sp = C.c_void_p(stream.cuda_stream)

        def copy():
            return dspdll.dsp_copy_rows(C.c_void_p(y.data_ptr()), ln, C.c_void_p(x.data_ptr()), ln, rows, ln, 256, 2048, sp)
        if copy() == 0:
            copy()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(stream)
            for _ in range(4):
                copy()
            c1.record(stream)
            torch.cuda.synchronize()
            copy_gbs = n * BYTES_PER_SAMPLE / (c0.elapsed_time(c1) / 4 / 1e3) / 1e9
    if rank == 0:
        achieved = n * BYTES_PER_SAMPLE / avg_s / 1e9
        line = {
            "metric": "filtered Msamples/s (whole node)", "value": total * args.steps / elapsed / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 5: single 2^%d-sample stream, 16th-order IIR (butterworth(8,0.25,0.75)), "
                                   "chunked state-space scan split in time, state hand-off between chunks"
                                   % args.stream_log2,
                       "samples_per_gpu": n, "sub_chunks": fs.last_split, "decay_length": fs.decay_length,
                       "settle_steps_untimed": settle_steps,
                       "parallelism": "time-split x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (measured_traffic("r3_config5_traffic.json")
                                     if world == 1 and args.stream_log2 == 30 else None),
                         "algorithmic_bytes_per_launch": n * BYTES_PER_SAMPLE,
                         "note": "per-GPU bytes of one step (ONE launch of the scan kernel over all sub-chunks -- it "
                                 "computes their run-ups itself --, two small state kernels) over the step's device "
                                 "time on rank 0; traffic = PMC bytes of that launch "
                                 "(profiles/r3_config5_traffic.json, committed; not re-measured by this run)",
                         "copy_ceiling_measured": copy_gbs,
                         "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                         "step_ms_avg": avg_s * 1e3, "step_ms_min": float(np.min(step_ms))},
        }
        if world == 1 and not args.no_cpu:
            from oracle import oracle as orc
            from oracle import ref
            m = 1 << 24                                   # one core: the path is serial on the CPU
            xs = util.synth(SEED, 0, 0, m)
            kind = "reference" if ref.available() else "port"
            f = ref.RefIIR(b, a) if kind == "reference" else orc.iir(b, a)
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 10.0:
                f.updaten(xs)
                reps += 1
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = dict(value=reps * m / dt / 1e6, unit="Msamples/s", cores=1, kind=kind,
                                        sample="%d x 2^24 consecutive samples of the config-5 stream on one host core "
                                               "(one filter is serial in time on the CPU), %.1f s" % (reps, dt))
        if not emit:
            return line
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_fir(b):
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from oracle import ref
    from tests import util
    threads = max(1, min(16, os.cpu_count() or 1))
    per_thread = 960          # ~15 s of host work at ~67 Msamples/s
    kind = "reference" if ref.available() else "port"
    xs = [util.synth(SEED, t, 0, BLOCK) for t in range(threads)]

    def work(t):
        for _ in range(per_thread):
            (ref.RefFIR(b) if kind == "reference" else orc.fir(b)).updaten(xs[t])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    return dict(value=threads * per_thread * BLOCK / dt / 1e6, unit="Msamples/s", cores=threads, kind=kind,
                sample="%d channels x 64K samples of config 3 (512 taps), RTFilter_updaten per channel, "
                       "%d host threads, %.1f s" % (threads * per_thread, threads, dt))


if __name__ == "__main__":
    main()
