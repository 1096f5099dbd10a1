#!/usr/bin/env python3
"""Diagnostic (stamped library build, DSP_AMD_SCAN_STAMPS=1): the idle time between two back-to-back launches of
the pipelined kernel on one stream -- last workgroup of launch k done -> first workgroup of launch k+1 entered --
from the 100 MHz realtime stamps of two banks launched alternately."""
import ctypes as C, os, sys
os.environ["DSP_AMD_SCAN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
CH, n = 1024, 65536
dspdll.dsp_debug_scan_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
banks = [gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH) for _ in range(2)]
x = torch.empty((4, CH, n), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
for j in range(4):
    gbank.synth_fill(x[j], 1, n0=j * n)
torch.cuda.synchronize()
def stamps(bk):
    buf = np.zeros(CH * 32, dtype=np.uint64)
    assert dspdll.dsp_debug_scan_stamps(bk._handle, buf.ctypes.data, buf.size) == 0
    return buf.reshape(CH, 32).astype(np.float64) / 100.0
for rep in range(4):
    for k in range(64):          # steady state, alternating banks; the last two launches are the ones read back
        banks[k % 2].updaten_ptr(y[k % 4].data_ptr(), n, x[k % 4].data_ptr(), n, n)
    torch.cuda.synchronize()
    s0, s1 = stamps(banks[0]), stamps(banks[1])      # launch 62 (bank 0), launch 63 (bank 1)
    print("launch k: first entry %.1f, last done %.1f (span %.1f us) | launch k+1: first entry +%.1f us after that, "
          "p50 entry +%.1f, last entry +%.1f; span %.1f us; start-to-start %.1f us"
          % (0.0, s0[:, 5].max() - s0[:, 0].min(), s0[:, 5].max() - s0[:, 0].min(),
             s1[:, 0].min() - s0[:, 5].max(), np.median(s1[:, 0]) - s0[:, 5].max(), s1[:, 0].max() - s0[:, 5].max(),
             s1[:, 5].max() - s1[:, 0].min(), s1[:, 0].min() - s0[:, 0].min()), flush=True)
