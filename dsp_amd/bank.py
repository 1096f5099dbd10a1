"""FilterBank: many RTFilter channels advanced together on an MI355X.

Thin host mirror of include/RTFilterBank.h.  Semantics are the reference's
per-filter loop (python/dsp.py:25-34 -> RTFilter_updaten, src/RTFilter.c:111-119)
applied to every channel of the bank; samples stay in device memory.

torch is used only as the owner of device buffers and streams: tensors are
passed to the C ABI as raw pointers.
"""
import ctypes as C

import numpy as np

from . import cdsp
from .cdsp import dspdll

KIND_IIR, KIND_FIR, KIND_FIR_PARTIAL, KIND_SOS = (cdsp.RTFB_KIND_IIR, cdsp.RTFB_KIND_FIR, cdsp.RTFB_KIND_FIR_PARTIAL,
                                                   cdsp.RTFB_KIND_SOS)
MODE_AUTO, MODE_SEQUENTIAL, MODE_SCAN = cdsp.RTFB_MODE_AUTO, cdsp.RTFB_MODE_SEQUENTIAL, cdsp.RTFB_MODE_SCAN


class DSPError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = dspdll.RTFilterBank_last_error() or b""
        super().__init__("%s failed with status %d: %s" % (where, status, msg.decode()))


def _check(status, where):
    if status != cdsp.RTFB_OK:
        raise DSPError(status, where)


def device_count():
    return dspdll.dsp_device_count()


def noalloc_max_doubles():
    return dspdll.RTFilterBank_noalloc_max_doubles()


def _stream_ptr(stream, device=None):
    """hipStream_t for the C ABI.  None = torch's current stream OF `device` (the bank's device, which need not be
    torch's current device), or the default stream without torch."""
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        except ImportError:
            pass
        return None
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


def iir_design(b, a):
    return dict(kind=KIND_IIR, b=np.ascontiguousarray(b, dtype=np.float64),
                a=np.ascontiguousarray(a, dtype=np.float64))


def sos_design(sections):
    """sections: [(b, a), ...] with three taps each (dsp.design_sos, RTIIRFilter.sections())."""
    b = np.ascontiguousarray([t for sb, _ in sections for t in sb], dtype=np.float64)
    a = np.ascontiguousarray([t for _, sa in sections for t in sa], dtype=np.float64)
    assert len(b) == len(a) == 3 * len(sections)
    return dict(kind=KIND_SOS, b=b, a=a)


def fir_design(b, partial=False):
    return dict(kind=KIND_FIR_PARTIAL if partial else KIND_FIR,
                b=np.ascontiguousarray(b, dtype=np.float64), a=None)


class FilterBank:
    """nch channels; channel c runs designs[design_of_channel[c]]."""

    def __init__(self, designs, design_of_channel, device=0, mode=MODE_AUTO):
        self._handle = C.c_void_p()
        self.device = device
        self._keep = designs                      # coefficient arrays stay alive during the call
        arr = (cdsp.RTFilterDesign_ * len(designs))()
        for i, d in enumerate(designs):
            arr[i].kind = d["kind"]
            arr[i].nb = len(d["b"])
            arr[i].b = d["b"].ctypes.data_as(C.POINTER(C.c_double))
            if d["kind"] in (KIND_IIR, KIND_SOS):
                arr[i].na = len(d["a"])
                arr[i].a = d["a"].ctypes.data_as(C.POINTER(C.c_double))
        idx = np.ascontiguousarray(design_of_channel, dtype=np.int32)
        _check(dspdll.RTFilterBank_from_designs(C.byref(self._handle), arr, len(designs),
                                                idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx),
                                                device), "RTFilterBank_from_designs")
        self.nch = len(idx)
        self._filters = None
        if mode != MODE_AUTO:
            self.set_mode(mode)

    @classmethod
    def from_arrays(cls, kind, b, a=None, design_of_channel=None, device=0, mode=MODE_AUTO):
        """Many designs of one kind and size from packed arrays (e.g. dsp.design_batch): b is (ndesign, nb), a is
        (ndesign, na) for IIR.  The RTFilterDesign table is laid out with numpy instead of a Python loop over
        ctypes structs (1.3 s per 2^18 designs); channel c runs design design_of_channel[c] (default: c)."""
        self = cls.__new__(cls)
        self._handle = C.c_void_p()
        self.device = device
        b = np.ascontiguousarray(b, dtype=np.float64)
        nd, nb = b.shape
        rec = np.zeros(nd, dtype=np.dtype(dict(names=["kind", "nb", "na", "b", "a"],
                                               formats=[np.int32, np.uintp, np.uintp, np.uintp, np.uintp],
                                               offsets=[0, 8, 16, 24, 32], itemsize=C.sizeof(cdsp.RTFilterDesign_))))
        rec["kind"] = kind
        rec["nb"] = nb
        rec["b"] = b.ctypes.data + np.arange(nd, dtype=np.uintp) * np.uintp(b.strides[0])
        if a is not None:
            a = np.ascontiguousarray(a, dtype=np.float64)
            assert a.shape[0] == nd
            rec["na"] = a.shape[1]
            rec["a"] = a.ctypes.data + np.arange(nd, dtype=np.uintp) * np.uintp(a.strides[0])
        self._keep = (b, a, rec)
        idx = np.ascontiguousarray(np.arange(nd) if design_of_channel is None else design_of_channel, dtype=np.int32)
        _check(dspdll.RTFilterBank_from_designs(C.byref(self._handle),
                                                C.cast(rec.ctypes.data, C.POINTER(cdsp.RTFilterDesign_)), nd,
                                                idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx), device),
               "RTFilterBank_from_designs")
        self.nch = len(idx)
        self._filters = None
        if mode != MODE_AUTO:
            self.set_mode(mode)
        return self

    @classmethod
    def from_filters(cls, filters, device=0, mode=MODE_AUTO):
        """filters: dsp.RTIIRFilter / dsp.RTFIRFilter objects (their C structs are referenced)."""
        self = cls.__new__(cls)
        self._handle = C.c_void_p()
        self.device = device
        self._filters = list(filters)
        ptrs = (C.c_void_p * len(self._filters))(*[C.addressof(f.cobj()) for f in self._filters])
        _check(dspdll.RTFilterBank_from_filters(C.byref(self._handle), ptrs, len(self._filters), device),
               "RTFilterBank_from_filters")
        self.nch = len(self._filters)
        self._keep = None
        if mode != MODE_AUTO:
            self.set_mode(mode)
        return self

    def close(self):
        if self._handle:
            for feed in list(getattr(self, "_feeds", ())):      # a feed refers to the C handle: close it first
                feed.close()
            dspdll.RTFilterBank_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, mode):
        _check(dspdll.RTFilterBank_set_mode(self._handle, mode), "RTFilterBank_set_mode")

    def updaten_ptr(self, out_ptr, ld_out, in_ptr, ld_in, n, stream=None):
        _check(dspdll.RTFilterBank_updaten(self._handle, C.c_void_p(out_ptr), ld_out, C.c_void_p(in_ptr),
                                           ld_in, n, _stream_ptr(stream, self.device)), "RTFilterBank_updaten")

    def updaten_ptr_noalloc(self, out_ptr, ld_out, in_ptr, ld_in, n, stream=None):
        """RTFilterBank_updaten_noalloc: the same update without any allocation (in-place FIR calls stage
        through the library's static device array; RTFB_ESIZE beyond noalloc_max_doubles())."""
        _check(dspdll.RTFilterBank_updaten_noalloc(self._handle, C.c_void_p(out_ptr), ld_out, C.c_void_p(in_ptr),
                                                   ld_in, n, _stream_ptr(stream, self.device)), "RTFilterBank_updaten_noalloc")

    def updaten_tiled_ptr(self, out_ptr, in_ptr, n, stream=None):
        """RTFilterBank_updaten_tiled: in / out are [n / TILE][nch][TILE] device blocks."""
        _check(dspdll.RTFilterBank_updaten_tiled(self._handle, C.c_void_p(out_ptr), C.c_void_p(in_ptr), n,
                                                 _stream_ptr(stream, self.device)), "RTFilterBank_updaten_tiled")

    def updaten_chained_ptr(self, out_ptr, ld_out, in_ptr, ld_in, n, tiled=False, stream=None, want_event=False):
        """RTFilterBank_updaten_chained: block k+1 overlaps the drain of block k on the bank's internal streams.
        Returns the block's done event (a raw hipEvent_t as int) when want_event, else None."""
        ev = C.c_void_p()
        _check(dspdll.RTFilterBank_updaten_chained(self._handle, C.c_void_p(out_ptr), ld_out, C.c_void_p(in_ptr), ld_in,
                                                   n, cdsp.RTFB_CHAIN_TILED if tiled else 0,
                                                   _stream_ptr(stream, self.device),
                                                   C.byref(ev) if want_event else None),
               "RTFilterBank_updaten_chained")
        return ev.value if want_event else None

    def set_pace(self, ns_per_segment):
        """Schedule of tiled launches: > 0 period per 2048-sample segment in ns, 0 off, < 0 the library's choice."""
        _check(dspdll.RTFilterBank_set_pace(self._handle, float(ns_per_segment)), "RTFilterBank_set_pace")

    def chain_join(self, stream=None):
        """`stream` waits for every chained call so far; the carried state is back in place."""
        _check(dspdll.RTFilterBank_chain_join(self._handle, _stream_ptr(stream, self.device)), "RTFilterBank_chain_join")

    def filter_host(self, x, out=None):
        """x: (nch, n) float64 numpy array in host memory (rows contiguous): RTFilterBank_updaten_host, the
        reference's per-filter call on host arrays, staged through pinned rings.  Returns `out`."""
        x = np.asarray(x)
        assert x.dtype == np.float64 and x.ndim == 2 and x.shape[0] == self.nch and x.strides[1] == 8
        if out is None:
            out = np.empty_like(x)
        assert out.shape == x.shape and out.dtype == np.float64 and out.strides[1] == 8
        dp = C.POINTER(C.c_double)
        _check(dspdll.RTFilterBank_updaten_host(self._handle, out.ctypes.data_as(dp), out.strides[0] // 8,
                                                x.ctypes.data_as(dp), x.strides[0] // 8, x.shape[1]),
               "RTFilterBank_updaten_host")
        return out

    def __call__(self, x, out=None, stream=None):
        """x: (nch, n) float64 device tensor with contiguous rows.  Returns the output tensor."""
        import torch
        assert x.dtype == torch.float64 and x.dim() == 2 and x.shape[0] == self.nch
        assert x.stride(1) == 1 or x.shape[1] <= 1
        if out is None:
            out = torch.empty_like(x, memory_format=torch.contiguous_format)
        assert out.shape == x.shape and out.dtype == torch.float64
        self.updaten_ptr(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), x.shape[1], stream)
        return out

    def reset(self, stream=None):
        _check(dspdll.RTFilterBank_reset(self._handle, _stream_ptr(stream, self.device)), "RTFilterBank_reset")

    def sync_to_host(self, stream=None):
        _check(dspdll.RTFilterBank_sync_to_host(self._handle, _stream_ptr(stream, self.device)), "RTFilterBank_sync_to_host")

    @property
    def state_ld(self):
        return dspdll.RTFilterBank_state_ld(self._handle)

    def get_state(self, stream=None):
        ld = self.state_ld
        st = np.empty((self.nch, ld), dtype=np.float64)
        fv = np.empty(self.nch, dtype=np.float64)
        ini = np.empty(self.nch, dtype=np.int32)
        _check(dspdll.RTFilterBank_get_state(self._handle, st.ctypes.data_as(C.POINTER(C.c_double)),
                                             fv.ctypes.data_as(C.POINTER(C.c_double)),
                                             ini.ctypes.data_as(C.POINTER(C.c_int32)), _stream_ptr(stream, self.device)),
               "RTFilterBank_get_state")
        return st, fv, ini

    def set_state(self, state=None, fval=None, init=None, stream=None):
        def ptr(a, t, ct):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=t)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(ct))
        keep = []
        if state is not None:
            assert np.shape(state) == (self.nch, self.state_ld)
        _check(dspdll.RTFilterBank_set_state(self._handle, ptr(state, np.float64, C.c_double),
                                             ptr(fval, np.float64, C.c_double),
                                             ptr(init, np.int32, C.c_int32), _stream_ptr(stream, self.device)),
               "RTFilterBank_set_state")


TILE = cdsp.RTFB_TILE


def tile_rows(rows, tiles=None, stream=None):
    """(nch, n) device tensor (row stride free) -> the tiled block layout [n / TILE][nch][TILE]."""
    import torch
    nch, n = rows.shape
    assert n % TILE == 0 and rows.dtype == torch.float64 and (rows.stride(1) == 1 or n <= 1)
    if tiles is None:
        tiles = torch.empty((n // TILE, nch, TILE), dtype=torch.float64, device=rows.device)
    assert tiles.is_contiguous() and tiles.numel() == nch * n
    _check(dspdll.dsp_tile_rows(C.c_void_p(tiles.data_ptr()), C.c_void_p(rows.data_ptr()), rows.stride(0), nch, n,
                                _stream_ptr(stream, rows.device)), "dsp_tile_rows")
    return tiles


def untile_rows(tiles, rows=None, stream=None):
    """[n / TILE][nch][TILE] -> (nch, n)."""
    import torch
    nseg, nch, t = tiles.shape
    assert t == TILE and tiles.is_contiguous() and tiles.dtype == torch.float64
    if rows is None:
        rows = torch.empty((nch, nseg * TILE), dtype=torch.float64, device=tiles.device)
    assert rows.shape == (nch, nseg * TILE) and rows.stride(1) == 1
    _check(dspdll.dsp_untile_rows(C.c_void_p(rows.data_ptr()), rows.stride(0), C.c_void_p(tiles.data_ptr()), nch,
                                  nseg * TILE, _stream_ptr(stream, tiles.device)), "dsp_untile_rows")
    return rows


def synth_fill(tensor, seed, chan0=0, n0=0, stream=None):
    """Fill a (nch, n) float64 device tensor with the SURVEY.md 8(d) synthetic input."""
    nch, n = tensor.shape
    _check(dspdll.dsp_synth_fill(C.c_void_p(tensor.data_ptr()), tensor.stride(0), nch, n, seed, chan0, n0,
                                 _stream_ptr(stream, tensor.device)), "dsp_synth_fill")
    return tensor


class BankFeed:
    """RTFilterBankFeed: frames (one sample per channel) in, filtered frames out, coalesced into device blocks --
    the bank-wide form of the reference's per-sample RTFilter_update / FilterIterator (python/dsp.py:6-13,35-38)."""

    def __init__(self, bank, block_frames):
        self._h = C.c_void_p()
        self.bank = bank
        self.nch = bank.nch
        _check(dspdll.RTFilterBankFeed_create(C.byref(self._h), bank._handle, block_frames), "RTFilterBankFeed_create")
        bank.__dict__.setdefault("_feeds", []).append(self)       # FilterBank.close() closes its live feeds first

    def close(self):
        if self._h:
            dspdll.RTFilterBankFeed_destroy(self._h)
            self._h = C.c_void_p()
            feeds = self.bank.__dict__.get("_feeds", [])
            if self in feeds:
                feeds.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, frames):
        """frames: (nframes, nch) float64, C-contiguous (or one frame of nch values)."""
        frames = np.ascontiguousarray(np.atleast_2d(frames), dtype=np.float64)
        assert frames.shape[1] == self.nch
        took = C.c_size_t()
        _check(dspdll.RTFilterBankFeed_push(self._h, frames.ctypes.data_as(C.POINTER(C.c_double)), frames.shape[0],
                                            C.byref(took)), "RTFilterBankFeed_push")
        return took.value                    # < len(frames): two blocks wait to be popped; pop, then push the rest

    def pop(self, max_frames, wait=False):
        """Filtered frames that are back (at most max_frames), oldest first; wait=True blocks until the oldest
        submitted block has come back."""
        if wait:
            _check(dspdll.RTFilterBankFeed_wait(self._h), "RTFilterBankFeed_wait")
        out = np.empty((max_frames, self.nch), dtype=np.float64)
        got = C.c_size_t()
        _check(dspdll.RTFilterBankFeed_pop(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), max_frames,
                                           C.byref(got)), "RTFilterBankFeed_pop")
        return out[: got.value]

    def available(self):
        return dspdll.RTFilterBankFeed_available(self._h)

    def pending(self):
        return dspdll.RTFilterBankFeed_pending(self._h)

    def flush(self):
        _check(dspdll.RTFilterBankFeed_flush(self._h), "RTFilterBankFeed_flush")
