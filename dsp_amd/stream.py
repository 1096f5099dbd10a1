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
from .cdsp import RTFB_EUNSUPPORTED, RTFilterDesign_, dspdll


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


def split_update(engine, x_local, out_local=None, group=None, stream=None):
    """`RTFilter_updaten` over a stream whose consecutive chunks live on consecutive ranks.

    `engine`: a FilterStream (or anything with its hand-off methods: new_pack, tail_state,
    export_state, import_state, __call__).  Rank 0's engine holds the state carried into the stream;
    on return EVERY rank's engine holds the state at the end of the whole stream (so that the
    next call may repartition), and each rank's chunk of the output is returned.

    Hand-off: ranks whose chunk allows it compute the zero-state tail vector of their own chunk and
    all ranks all_gather these vectors (one small collective: pack_len doubles per rank); rank r
    then imports rank r-1's vector and all ranks filter concurrently.  If any rank reports that the
    tail shortcut does not apply (filter does not decay, chunk too short) the ranks chain instead:
    rank r waits for rank r-1's true end state (send/recv), which serialises them but is exact.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return engine(x_local, out_local, stream)

    # gloo (the CPU tests, or a rehearsal with several ranks on one GPU) moves host tensors only:
    # stage the few doubles of the hand-off through the host there; RCCL takes the device tensors.
    host_hop = dist.get_backend(group) == "gloo"

    def wire(t):
        return t.cpu() if (host_hop and t.is_cuda) else t

    def peer(r):
        return dist.get_global_rank(group, r) if group is not None else r

    pack = engine.new_pack()
    ok = engine.tail_state(x_local, pack, stream) if x_local.numel() else False
    flag = wire(torch.tensor([1 if ok else 0], dtype=torch.int32, device=pack.device))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 1:
        mine = wire(pack)
        packs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(packs, mine, group=group)
        if rank > 0:
            engine.import_state(packs[rank - 1].to(pack.device), stream)
        out = engine(x_local, out_local, stream)
    else:
        if rank > 0:
            got = wire(pack)
            dist.recv(got, src=peer(rank - 1), group=group)
            engine.import_state(got.to(pack.device), stream)
        out = engine(x_local, out_local, stream)
        if rank < world - 1:
            engine.export_state(pack, stream)
            dist.send(wire(pack), dst=peer(rank + 1), group=group)
    # everyone ends with the stream's end state (the last rank's true state)
    engine.export_state(pack, stream)
    last = wire(pack)
    dist.broadcast(last, src=peer(world - 1), group=group)
    engine.import_state(last.to(pack.device), stream)
    return out
