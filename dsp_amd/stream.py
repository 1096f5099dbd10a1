"""One filter over a very long device-resident stream (include/RTFilterStream.h), and its split in
time across ranks (BASELINE.json config 5).

`FilterStream.updaten` is `RTFilter_updaten(out, filter, in, n)` (ref: src/RTFilter.c:111-119) with
device pointers, time-parallel on one GPU.  `split_update` is the same call with the stream cut
into one contiguous chunk per rank: the only exchange is the state hand-off vector (N + 3 doubles)
from each rank to the next, over `torch.distributed` (RCCL/xGMI on GPUs, gloo in the CPU tests).
"""
import ctypes as C

import numpy as np

from .bank import DSPError, _check, _stream_ptr
from .cdsp import RTFB_EUNSUPPORTED, RTFilterDesign_, RTFilterStream_allgather_fn, dspdll


class FilterStream:
    """Wraps an RTFilterStream handle.  `design` as returned by bank.iir_design / bank.fir_design."""

    def __init__(self, design, device=0):
        self._h = C.c_void_p()
        self._keep = design
        kind, b, a = design["kind"], design["b"], design.get("a")
        d = RTFilterDesign_(kind, len(b), 0 if a is None else len(a),
                            b.ctypes.data_as(C.POINTER(C.c_double)),
                            None if a is None else a.ctypes.data_as(C.POINTER(C.c_double)))
        _check(dspdll.RTFilterStream_from_design(C.byref(self._h), C.byref(d), device), "RTFilterStream_from_design")
        self.device = device

    @classmethod
    def from_filter(cls, filt, device=0):
        """From a dsp.RTIIRFilter / dsp.RTFIRFilter host object (its current state is taken over)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self._keep = filt
        _check(dspdll.RTFilterStream_from_filter(C.byref(self._h), C.c_void_p(C.addressof(filt.cobj())), device),
               "RTFilterStream_from_filter")
        self.device = device
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            dspdll.RTFilterStream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the time split across ranks, run by the library (RTFilterStream_split_updaten) ----
    def split_updaten(self, x_local, out_local, lens, rank, allgather, stream=None):
        """One call of the C entry: `lens` = chunk lengths of all ranks, `allgather(send_ptr, recv_ptr, count,
        stream_ptr) -> int` the caller's collective on device pointers (see include/RTFilterStream.h)."""
        import torch
        if out_local is None:
            out_local = torch.empty_like(x_local)
        arr = (C.c_size_t * len(lens))(*lens)
        cb = RTFilterStream_allgather_fn(lambda ctx, send, recv, count, sp: int(allgather(send, recv, count, sp)))
        _check(dspdll.RTFilterStream_split_updaten(self._h, C.c_void_p(out_local.data_ptr()), C.c_void_p(x_local.data_ptr()),
                                                   arr, rank, len(lens), cb, None, _stream_ptr(stream, self.device)),
               "RTFilterStream_split_updaten")
        return out_local

    def split_end_state(self, rank, world, allgather, stream=None):
        cb = RTFilterStream_allgather_fn(lambda ctx, send, recv, count, sp: int(allgather(send, recv, count, sp)))
        _check(dspdll.RTFilterStream_split_end_state(self._h, rank, world, cb, None, _stream_ptr(stream, self.device)),
               "RTFilterStream_split_end_state")

    # ---- the path ----
    def updaten_ptr(self, out_ptr, in_ptr, n, stream=None):
        _check(dspdll.RTFilterStream_updaten(self._h, C.c_void_p(out_ptr), C.c_void_p(in_ptr), n, _stream_ptr(stream, self.device)),
               "RTFilterStream_updaten")

    def __call__(self, x, out=None, stream=None):
        """x: 1-D float64 device tensor; returns the filtered tensor (a new one unless `out`)."""
        import torch
        if out is None:
            out = torch.empty_like(x)
        self.updaten_ptr(out.data_ptr(), x.data_ptr(), x.numel(), stream)
        return out

    def filter_host(self, x, out=None):
        """x: 1-D float64 numpy array in host memory; the reference call on host buffers, moved through the
        GPU in double-buffered chunks (bounded by the host link).  Returns `out` (a new array by default)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        if out is None:
            out = np.empty_like(x)
        _check(dspdll.RTFilterStream_updaten_host(self._h, out.ctypes.data_as(C.POINTER(C.c_double)),
                                                  x.ctypes.data_as(C.POINTER(C.c_double)), x.size),
               "RTFilterStream_updaten_host")
        return out

    def reset(self, stream=None):
        _check(dspdll.RTFilterStream_reset(self._h, _stream_ptr(stream, self.device)), "RTFilterStream_reset")

    def sync_to_host(self, stream=None):
        _check(dspdll.RTFilterStream_sync_to_host(self._h, _stream_ptr(stream, self.device)), "RTFilterStream_sync_to_host")

    @property
    def decay_length(self):
        return dspdll.RTFilterStream_decay_length(self._h)

    @property
    def last_split(self):
        return dspdll.RTFilterStream_last_split(self._h)

    # ---- state hand-off ----
    @property
    def pack_len(self):
        return dspdll.RTFilterStream_pack_len(self._h)

    def new_pack(self):
        import torch
        return torch.zeros(self.pack_len, dtype=torch.float64, device=torch.device("cuda", self.device))

    def export_state(self, pack, stream=None):
        _check(dspdll.RTFilterStream_export_state(self._h, C.c_void_p(pack.data_ptr()), _stream_ptr(stream, self.device)),
               "RTFilterStream_export_state")
        return pack

    def import_state(self, pack, stream=None):
        _check(dspdll.RTFilterStream_import_state(self._h, C.c_void_p(pack.data_ptr()), _stream_ptr(stream, self.device)),
               "RTFilterStream_import_state")

    def tail_state(self, x, pack, stream=None):
        """Hand-off vector after x (zero-state response of its last decay_length samples).
        Returns False when the design or the chunk does not allow it (RTFB_EUNSUPPORTED)."""
        rc = dspdll.RTFilterStream_tail_state(self._h, C.c_void_p(x.data_ptr()), x.numel(),
                                              C.c_void_p(pack.data_ptr()), _stream_ptr(stream, self.device))
        if rc == RTFB_EUNSUPPORTED:
            return False
        _check(rc, "RTFilterStream_tail_state")
        return True

    # ---- exact composition of chunk maps (include/RTFilterStream.h) ----
    @property
    def exact_split(self):
        return bool(dspdll.RTFilterStream_exact_split(self._h))

    # ---- interleaved segments (include/RTFilterStream.h) ----
    @property
    def interleaved(self):
        return bool(dspdll.RTFilterStream_interleaved(self._h))

    def set_interleave(self, workgroups):
        """> 0: deal the stream's segments to that many workgroups (clamped to one resident round); <= 0: off (default)."""
        _check(dspdll.RTFilterStream_set_interleave(self._h, int(workgroups)), "RTFilterStream_set_interleave")

    def zero_state(self, stream=None):
        _check(dspdll.RTFilterStream_zero_state(self._h, _stream_ptr(stream, self.device)), "RTFilterStream_zero_state")

    def compose_state(self, packs, count, chunk_len, stream=None):
        """packs: (>= count, stride) device tensor of hand-off vectors p_0 .. p_(count-1); the stream's state
        becomes A^L(...(A^L p_0 + p_1)...) + p_(count-1) for chunks of L = chunk_len samples."""
        _check(dspdll.RTFilterStream_compose_state(self._h, C.c_void_p(packs.data_ptr()), packs.stride(0), count,
                                                   chunk_len, _stream_ptr(stream, self.device)),
               "RTFilterStream_compose_state")

    def get_state(self, stream=None):
        ld = self.pack_len - 2
        st = np.zeros(ld, dtype=np.float64)
        fv = np.zeros(1, dtype=np.float64)
        ini = np.zeros(1, dtype=np.int32)
        _check(dspdll.RTFilterStream_get_state(self._h, st.ctypes.data_as(C.POINTER(C.c_double)),
                                               fv.ctypes.data_as(C.POINTER(C.c_double)),
                                               ini.ctypes.data_as(C.POINTER(C.c_int32)), _stream_ptr(stream, self.device)),
               "RTFilterStream_get_state")
        return st, float(fv[0]), int(ini[0])


def time_range(n, world_size, rank, grain=2048):
    """Contiguous sample range [lo, hi) of `rank`: equal chunks, multiples of `grain`, the last rank
    takes the remainder."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    per = n // world_size // grain * grain
    if per == 0:
        return (0, n) if rank == 0 else (n, n)
    lo = rank * per
    return lo, (n if rank == world_size - 1 else lo + per)


def _plan(engine, numel, world, rank, group, total):
    """Chunk lengths of all ranks and the hand-off mode, decided once per (engine, chunk length, world) and kept:
    'tail' (one all_gather, nothing else), 'exact' (chunk maps composed), 'chain' (send/recv of true end states)."""
    import torch
    import torch.distributed as dist
    key = (numel, world, rank, total)
    plans = engine.__dict__.setdefault("_split_plans", {})
    if key in plans:
        return plans[key]
    if total is not None:
        lens = [hi - lo for lo, hi in (time_range(total, world, r) for r in range(world))]
        assert lens[rank] == numel, "x_local does not hold time_range(total, world, rank)"
    else:                                                   # one small host collective, first call only
        mine = torch.tensor([numel], dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            mine = mine.cuda()
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine, group=group)
        lens = [int(t.item()) for t in got]
    decay = int(getattr(engine, "decay_length", 0) or 0)
    feeders = lens[:-1]                                     # the last rank hands nothing on
    if decay and all(n >= decay for n in feeders):
        mode = "tail"
    elif getattr(engine, "exact_split", False) and all(n == lens[1] for n in lens[1:-1]) and all(n > 0 for n in feeders):
        mode = "exact"
    else:
        mode = "chain"
    plans[key] = (lens, mode)
    return plans[key]


def split_update(engine, x_local, out_local=None, group=None, stream=None, total=None):
    """`RTFilter_updaten` over a stream whose consecutive chunks live on consecutive ranks (time_range).

    `engine`: a FilterStream (or anything with its hand-off methods).  The state carried INTO the stream is rank
    0's engine state on the first call; afterwards the stream's end state lives on the LAST rank and rides back to
    rank 0 inside the next call's all_gather, so consecutive calls continue one stream (`sync_end_state` puts it on
    every rank when a caller needs that, e.g. before repartitioning).  `total`: the stream length of this call, if
    the caller knows it -- the hand-off mode is then decided without any communication; otherwise the chunk lengths
    are exchanged once per (chunk length, world size) and remembered.

    Modes (fixed per engine and partition, no per-call negotiation, no host synchronisation on RCCL):
      tail   the design forgets within its decay length M <= every feeding chunk: each rank computes the hand-off
             vector of its own chunk from its last M samples; ONE all_gather (2 x pack_len doubles per rank: that
             vector and the rank's state at entry) moves everything; all ranks filter concurrently.
      exact  no usable decay: chunk maps are composed (RTFilterStream.h): ranks >= 1 run their chunk from the zero
             state first, one all_gather, compose, run again (2 passes on ranks >= 1; exact, no truncation).  An
             in-place call (out_local overlapping x_local) is allowed: the first pass then writes to a scratch
             chunk kept with the plan.
      chain  engines that can do neither (FIR, designs off the scan path): rank r waits for rank r-1's true end
             state (send/recv); exact and serial.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return engine(x_local, out_local, stream)
    lens, mode = _plan(engine, x_local.numel(), world, rank, group, total)

    # gloo (the CPU tests, or a rehearsal with several ranks on one GPU) moves host tensors only: the few doubles
    # of the hand-off are staged through the host there (which synchronises); RCCL takes the device tensors, and
    # the collectives are enqueued on the caller's stream like the kernels around them.
    host_hop = dist.get_backend(group) == "gloo"
    on_gpu = x_local.is_cuda
    ctx = torch.cuda.stream(stream) if (on_gpu and stream is not None and hasattr(stream, "cuda_stream")) else _Null()

    def wire(t):
        return t.cpu() if (host_hop and t.is_cuda) else t

    def peer(r):
        return dist.get_global_rank(group, r) if group is not None else r

    # Hand-off buffers are allocated ONCE per (engine, plan) and reused: at 8 GPUs a config-5 step is ~0.45 ms of
    # kernel time, so per-call tensor allocation and torch.stack traffic would sit on the critical path.
    bufs = engine.__dict__.setdefault("_split_bufs", {})
    bkey = (x_local.numel(), world, rank, mode)
    if bkey not in bufs:
        proto = engine.new_pack()
        plen = proto.numel()
        rows = 2 if mode == "tail" else 1
        bufs[bkey] = dict(mine=torch.zeros((rows, plen), dtype=torch.float64, device=proto.device),
                          everyone=torch.zeros((world, rows, plen), dtype=torch.float64, device=proto.device),
                          scratch=None)
    buf = bufs[bkey]

    def gather(mine):
        """all ranks' `mine` in buf["everyone"] (world, rows, plen), on mine's device; no allocation under RCCL"""
        every = buf["everyone"]
        if host_hop:
            m = wire(mine)
            parts = [torch.empty_like(m) for _ in range(world)]
            dist.all_gather(parts, m, group=group)
            for r, part in enumerate(parts):
                every[r].copy_(part)
        else:
            dist.all_gather_into_tensor(every, mine, group=group)
        return every

    carried_remote = bool(engine.__dict__.get("_carry_on_last_rank", False))
    with ctx:
        if mode == "tail":
            pack2 = buf["mine"]                                     # [0]: hand-off of my chunk (stays zero on the last rank)
            engine.export_state(pack2[1], stream)                   # my state at entry (the last rank's matters)
            if rank < world - 1:
                ok = engine.tail_state(x_local, pack2[0], stream)
                assert ok, "tail hand-off planned but refused by the engine"
            everyone = gather(pack2)
            if rank == 0:
                if carried_remote:
                    engine.import_state(everyone[world - 1, 1], stream)
            else:
                engine.import_state(everyone[rank - 1, 0], stream)
            out = engine(x_local, out_local, stream)
        elif mode == "exact":
            pack = buf["mine"][0]
            if carried_remote:                                      # rank 0 needs its start state before its pass
                engine.export_state(pack, stream)
                last = wire(pack)
                dist.broadcast(last, src=peer(world - 1), group=group)
                if rank == 0:
                    engine.import_state(last.to(pack.device), stream)
            if rank == 0:
                out = engine(x_local, out_local, stream)            # true pass; the vector is the true end state
                engine.export_state(pack, stream)
            elif rank < world - 1:
                # Zero-state pass: only its end state v_r is wanted.  Its outputs must NOT land on the inputs (an
                # in-place call: out_local is x_local, or the two overlap), which the true pass below reads again:
                # they go to a scratch chunk kept with the plan in that case.
                first_out = out_local
                if out_local is not None and _overlap(out_local, x_local):
                    if buf["scratch"] is None or buf["scratch"].numel() < x_local.numel():
                        buf["scratch"] = torch.empty(x_local.numel(), dtype=x_local.dtype, device=x_local.device)
                    first_out = buf["scratch"][: x_local.numel()].view(x_local.shape)
                engine.zero_state(stream)
                out = engine(x_local, first_out, stream)
                engine.export_state(pack, stream)
                if first_out is not out_local:
                    out = out_local
            else:
                out = out_local
                pack.zero_()
            everyone = gather(buf["mine"])[:, 0]
            if rank > 0:
                engine.compose_state(everyone, rank, lens[1], stream)
                out = engine(x_local, out, stream)
        else:
            pack = buf["mine"][0]
            if carried_remote:
                engine.export_state(pack, stream)
                last = wire(pack)
                dist.broadcast(last, src=peer(world - 1), group=group)
                if rank == 0:
                    engine.import_state(last.to(pack.device), stream)
            if rank > 0:
                got = wire(pack)
                dist.recv(got, src=peer(rank - 1), group=group)
                engine.import_state(got.to(pack.device), stream)
            out = engine(x_local, out_local, stream)
            if rank < world - 1:
                engine.export_state(pack, stream)
                dist.send(wire(pack), dst=peer(rank + 1), group=group)
    engine.__dict__["_carry_on_last_rank"] = True
    return out


def _overlap(p, q):
    """True if two tensors share memory (same storage range)."""
    if p.device != q.device or p.numel() == 0 or q.numel() == 0:
        return False
    p0, q0 = p.data_ptr(), q.data_ptr()
    return p0 < q0 + q.numel() * q.element_size() and q0 < p0 + p.numel() * p.element_size()


def sync_end_state(engine, group=None, stream=None):
    """After split_update: give every rank's engine the stream's end state (the last rank's), e.g. before the
    stream is partitioned differently or handed to single-GPU code.  One broadcast of pack_len doubles."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return
    host_hop = dist.get_backend(group) == "gloo"
    pack = engine.new_pack()
    engine.export_state(pack, stream)
    last = pack.cpu() if (host_hop and pack.is_cuda) else pack
    src = dist.get_global_rank(group, world - 1) if group is not None else world - 1
    dist.broadcast(last, src=src, group=group)
    engine.import_state(last.to(pack.device), stream)
    engine.__dict__["_carry_on_last_rank"] = False


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
