"""dsp_amd -- MI355X-native build of the bondeje/DSP RTFilter update path.

Layout:
  csrc/      hand-written HIP kernels (gfx950) + the C-ABI library sources
  lib/       libdsp.so (built in-tree by csrc/Makefile; also as dsp.so)
  cdsp.py    ctypes binding, layout-compatible with the reference's python/cdsp.py
  dsp.py     RTIIRFilter / RTFIRFilter / design helpers, mirroring python/dsp.py
  bank.py    FilterBank: many channels on device-resident blocks (RTFilterBank_* C ABI)
"""
from . import cdsp  # noqa: F401
