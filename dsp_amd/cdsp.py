"""ctypes binding of libdsp.so.

Mirrors the reference's python/cdsp.py: the same module-level names (`dspdll`,
`byref_`, `RTFilter_`, `FilterBank_`, `IIRFilterBank_`, `RTFIRFilter_`,
`RTIIRFilter_`, `i_RTFilter__d`, `v_RTFilter_`), the same Structure layouts
(cdsp.py:23-52) and the same prototypes (cdsp.py:56-74), so code written
against the reference binding runs unchanged.  Two things differ on purpose:

* the loader resolves an absolute path (the reference's `cdll.LoadLibrary
  ('dsp.so')` after editing PATH cannot work on Linux);
* the additive RTFilterBank_* / dsp_pool_* prototypes of
  include/RTFilterBank.h are declared as well.

The library is REQUIRED: there is no Python or CPU re-implementation behind
this module; if lib/libdsp.so is missing the import fails.
"""
import ctypes as _C
import os as _os
from ctypes import (CFUNCTYPE, POINTER, Structure, byref, c_double, c_int, c_int32, c_size_t,
                    c_uint, c_uint64, c_void_p, c_char_p)

_LIBDIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lib")
LIBRARY_PATH = _os.environ.get("DSP_AMD_LIBRARY", _os.path.join(_LIBDIR, "libdsp.so"))

if not _os.path.exists(LIBRARY_PATH):
    raise ImportError(
        "dsp_amd: %s not found. Build it with `make -C dsp_amd/csrc` "
        "(or __graft_entry__.build()); there is no fallback implementation." % LIBRARY_PATH)



def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) next to
    libtorch_hip.so, while libdsp.so is linked against the system ROCm's libamdhip64.so.7.  If both
    copies get loaded the second one finds "no ROCm-capable device".  Loading torch's copy first --
    by path, without importing torch -- makes the dynamic linker satisfy libdsp.so's NEEDED entry
    with it (SONAME match), and a later `import torch` finds the very same object.  Without torch
    installed libdsp.so simply uses the system runtime.
    """
    import importlib.util
    import sys
    override = _os.environ.get("DSP_AMD_HIP_RUNTIME")
    if override:
        _C.CDLL(override, mode=_C.RTLD_GLOBAL)
        return
    if "torch" in sys.modules:
        return                      # already loaded: SONAME match happens by itself
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = _os.path.join(_os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if _os.path.exists(cand):
        _C.CDLL(cand, mode=_C.RTLD_GLOBAL)


_share_hip_runtime_with_torch()
dspdll = _C.CDLL(LIBRARY_PATH)


def byref_(obj, offset=0):
    """byref that lets None through (python/cdsp.py:17-21)."""
    return byref(obj, offset) if obj is not None else None


class RTFilter_(Structure):
    pass


i_RTFilter__d = CFUNCTYPE(c_int, POINTER(RTFilter_), c_double)
v_RTFilter_ = CFUNCTYPE(None, POINTER(RTFilter_))

RTFilter_._fields_ = [("update", i_RTFilter__d),
                      ("initialize", i_RTFilter__d),
                      ("del", v_RTFilter_),
                      ("filtered_value", c_double),
                      ("flags", c_uint),
                      ("initialized", c_int)]


class FilterBank_(Structure):
    _fields_ = [("b", POINTER(c_double)), ("nb", c_size_t)]


class IIRFilterBank_(Structure):
    _fields_ = [("fb", FilterBank_), ("na", c_size_t)]


class RTFIRFilter_(Structure):
    # the reference names this member `ifb` too (cdsp.py:44-47); same offset as C's `fb`
    _fields_ = [("rtf", RTFilter_), ("ifb", FilterBank_), ("state", POINTER(c_double))]


class RTIIRFilter_(Structure):
    _fields_ = [("rtf", RTFilter_), ("ifb", IIRFilterBank_), ("state", POINTER(c_double))]


class RTFilterDesign_(Structure):
    """include/RTFilterBank.h: RTFilterDesign."""
    _fields_ = [("kind", c_int), ("nb", c_size_t), ("na", c_size_t),
                ("b", POINTER(c_double)), ("a", POINTER(c_double))]


_dp = POINTER(c_double)
_cb = POINTER(i_RTFilter__d)   # the reference declares callbacks as pointer-to-function-pointer


def _proto(name, restype, argtypes):
    fn = getattr(dspdll, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


# ---- prototypes of python/cdsp.py:56-74 ------------------------------------------------------
_proto("RTFilter_update", c_double, [POINTER(RTFilter_), c_double])
_proto("RTFilter_updaten", c_int, [_dp, POINTER(RTFilter_), _dp, c_size_t])
_proto("RTIIRFilter_init", None, [POINTER(RTIIRFilter_), _dp, _dp, c_size_t, c_size_t, c_uint, _cb])
_proto("RTFIRFilter_init", None, [POINTER(RTFIRFilter_), _dp, _dp, c_size_t, c_uint, _cb])
_proto("moving_average", c_int, [POINTER(RTFIRFilter_), c_size_t, _cb])
_proto("butterworth", c_int, [POINTER(RTIIRFilter_), c_size_t, c_double, c_double, c_uint, _cb])
_proto("chebyshev1", c_int, [POINTER(RTIIRFilter_), c_size_t, c_double, c_double, c_double, c_uint, _cb])
_proto("chebyshev2", c_int, [POINTER(RTIIRFilter_), c_size_t, c_double, c_double, c_double, c_uint, _cb])
_proto("thiran", c_int, [POINTER(RTIIRFilter_), c_size_t, c_double, _cb])

# ---- the rest of include/RTFilter.h that Python callers use ---------------------------------
_proto("RTFilter_reset", None, [POINTER(RTFilter_)])
_proto("RTIIRFilter_print", None, [POINTER(RTIIRFilter_)])
_proto("RTFIRFilter_print", None, [POINTER(RTFIRFilter_)])

# ---- include/filterutils.h (ref: python/cdsp.py:73-74 binds filter_response_pzg_noc at import) --------
_proto("filter_response_pzg_noc", c_int, [_dp, _dp, c_size_t, _dp, _dp, c_size_t, _dp, _dp, c_size_t, c_double, _dp])
_proto("filter_response_ab_noc", c_int, [_dp, _dp, c_size_t, _dp, c_size_t, _dp, c_size_t, _dp])

# ---- include/RTFilterDesign.h (additive host design entry points) ---------------------------------
FILTER_FINITE, FILTER_SOS = 0x1, 0x2
DSP_FAMILY_BUTTERWORTH, DSP_FAMILY_CHEBYSHEV1, DSP_FAMILY_CHEBYSHEV2 = 0, 1, 2
DSP_WINDOW_RECTANGULAR, DSP_WINDOW_HAMMING, DSP_WINDOW_HANN, DSP_WINDOW_BLACKMAN = 0, 1, 2, 3
DSP_MAX_SECTIONS = 64


class RTFilterSection_(Structure):
    _fields_ = [("b", c_double * 3), ("a", c_double * 3)]


_proto("dsp_design_sos", c_int, [c_int, c_size_t, c_double, c_double, c_double, POINTER(RTFilterSection_), c_size_t,
                                 POINTER(c_size_t)])
_proto("RTIIRFilter_sos_sections", c_int, [POINTER(RTIIRFilter_), POINTER(RTFilterSection_), c_size_t,
                                           POINTER(c_size_t)])
_proto("dsp_design_batch", c_int, [c_int, c_size_t, c_size_t, _dp, _dp, _dp, _dp, _dp, c_size_t, POINTER(c_size_t),
                                   c_int])
_proto("fir_windowed_sinc", c_int, [POINTER(RTFIRFilter_), c_size_t, c_double, c_double, c_int, _cb])

# ---- include/RTFilterBank.h -------------------------------------------------------------------
RTFB_OK, RTFB_EINVAL, RTFB_ESIZE, RTFB_EUNSUPPORTED, RTFB_EDEVICE, RTFB_ENOMEM = range(6)
RTFB_KIND_IIR, RTFB_KIND_FIR, RTFB_KIND_FIR_PARTIAL, RTFB_KIND_SOS = 0, 1, 2, 3
RTFB_MODE_AUTO, RTFB_MODE_SEQUENTIAL, RTFB_MODE_SCAN = 0, 1, 2

_proto("RTFilterBank_from_filters", c_int, [POINTER(c_void_p), POINTER(c_void_p), c_size_t, c_int])
_proto("RTFilterBank_from_designs", c_int,
       [POINTER(c_void_p), POINTER(RTFilterDesign_), c_size_t, POINTER(c_int32), c_size_t, c_int])
_proto("dsp_scan_probe", c_int, [POINTER(RTFilterDesign_), POINTER(c_double)])
_proto("dsp_scan_probe_kernel", c_int, [POINTER(RTFilterDesign_), c_int, POINTER(c_double)])
_proto("dsp_bank_table_doubles", c_int, [POINTER(RTFilterDesign_), c_size_t, POINTER(c_size_t)])
_proto("RTFilterBank_destroy", None, [c_void_p])
_proto("RTFilterBank_channels", c_size_t, [c_void_p])
_proto("RTFilterBank_set_mode", c_int, [c_void_p, c_int])
_proto("RTFilterBank_updaten", c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_void_p])
_proto("RTFilterBank_updaten_noalloc", c_int,
       [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_void_p])
_proto("RTFilterBank_noalloc_max_doubles", c_size_t, [])
RTFB_TILE, RTFB_CHAIN_TILED = 2048, 1
_proto("RTFilterBank_updaten_tiled", c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p])
_proto("RTFilterBank_updaten_chained", c_int,
       [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_uint, c_void_p, POINTER(c_void_p)])
_proto("RTFilterBank_chain_join", c_int, [c_void_p, c_void_p])
_proto("RTFilterBank_set_pace", c_int, [c_void_p, c_double])
_proto("dsp_tile_rows", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p])
_proto("dsp_untile_rows", c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_void_p])
_proto("RTFilterBank_updaten_host", c_int, [c_void_p, _dp, c_size_t, _dp, c_size_t, c_size_t])
_proto("RTFilterBankFeed_create", c_int, [POINTER(c_void_p), c_void_p, c_size_t])
_proto("RTFilterBankFeed_destroy", None, [c_void_p])
_proto("RTFilterBankFeed_push", c_int, [c_void_p, _dp, c_size_t, POINTER(c_size_t)])
_proto("RTFilterBankFeed_pop", c_int, [c_void_p, _dp, c_size_t, POINTER(c_size_t)])
_proto("RTFilterBankFeed_available", c_size_t, [c_void_p])
_proto("RTFilterBankFeed_flush", c_int, [c_void_p])
_proto("RTFilterBankFeed_wait", c_int, [c_void_p])
_proto("RTFilterBankFeed_pending", c_size_t, [c_void_p])
_proto("RTFilterBank_reset", c_int, [c_void_p, c_void_p])
_proto("RTFilterBank_sync_to_host", c_int, [c_void_p, c_void_p])
_proto("RTFilterBank_state_ld", c_size_t, [c_void_p])
_proto("RTFilterBank_get_state", c_int, [c_void_p, _dp, _dp, POINTER(c_int32), c_void_p])
_proto("RTFilterBank_set_state", c_int, [c_void_p, _dp, _dp, POINTER(c_int32), c_void_p])
_proto("RTFilterBank_last_error", c_char_p, [])
_proto("dsp_pool_alloc", c_int, [POINTER(c_void_p), c_size_t, c_int])
_proto("dsp_pool_free", c_int, [c_void_p])
_proto("dsp_pool_free_async", c_int, [c_void_p, c_void_p])
_proto("dsp_pool_trim", c_int, [c_int])
_proto("dsp_pool_bytes_in_use", c_size_t, [c_int])
_proto("dsp_pool_bytes_cached", c_size_t, [c_int])
_proto("dsp_synth_fill", c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_uint64, c_uint64, c_uint64, c_void_p])
_proto("dsp_copy_rows", c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_int, c_int, c_void_p])
_proto("dsp_copy_tiles", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p])
_proto("dsp_device_count", c_int, [])
_proto("dsp_memcpy_h2d", c_int, [c_void_p, c_void_p, c_size_t, c_int])
_proto("dsp_memcpy_d2h", c_int, [c_void_p, c_void_p, c_size_t, c_int])
_proto("dsp_stream_synchronize", c_int, [c_void_p, c_int])

# ---- include/RTFilterStream.h -------------------------------------------------------------------
_proto("RTFilterStream_from_design", c_int, [POINTER(c_void_p), POINTER(RTFilterDesign_), c_int])
_proto("RTFilterStream_from_filter", c_int, [POINTER(c_void_p), c_void_p, c_int])
_proto("RTFilterStream_destroy", None, [c_void_p])
_proto("RTFilterStream_updaten", c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p])
_proto("RTFilterStream_updaten_host", c_int, [c_void_p, _dp, _dp, c_size_t])
_proto("RTFilterStream_reset", c_int, [c_void_p, c_void_p])
_proto("RTFilterStream_sync_to_host", c_int, [c_void_p, c_void_p])
_proto("RTFilterStream_decay_length", c_size_t, [c_void_p])
_proto("RTFilterStream_last_split", c_size_t, [c_void_p])
_proto("RTFilterStream_pack_len", c_size_t, [c_void_p])
_proto("RTFilterStream_export_state", c_int, [c_void_p, c_void_p, c_void_p])
_proto("RTFilterStream_import_state", c_int, [c_void_p, c_void_p, c_void_p])
_proto("RTFilterStream_tail_state", c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p])
_proto("RTFilterStream_exact_split", c_int, [c_void_p])
_proto("RTFilterStream_set_interleave", c_int, [c_void_p, c_int])
_proto("RTFilterStream_interleaved", c_int, [c_void_p])
_proto("RTFilterStream_zero_state", c_int, [c_void_p, c_void_p])
_proto("RTFilterStream_compose_state", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p])
RTFilterStream_allgather_fn = CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p)
_proto("RTFilterStream_split_updaten", c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_size_t), c_int, c_int,
                                               RTFilterStream_allgather_fn, c_void_p, c_void_p])
_proto("RTFilterStream_split_end_state", c_int, [c_void_p, c_int, c_int, RTFilterStream_allgather_fn, c_void_p, c_void_p])
_proto("RTFilterStream_split_updaten_rccl", c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_size_t), c_int, c_int,
                                                    c_void_p, c_void_p])
_proto("RTFilterStream_split_end_state_rccl", c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p])
_proto("dsp_debug_rccl_allgather", c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p])
_proto("RTFilterStream_get_state", c_int, [c_void_p, _dp, _dp, POINTER(c_int32), c_void_p])
_proto("RTFilterStream_set_state", c_int, [c_void_p, _dp, _dp, POINTER(c_int32), c_void_p])


def fn_address(name):
    """Address of an exported function (for pointer-identity checks and struct fields)."""
    return _C.cast(getattr(dspdll, name), c_void_p).value
