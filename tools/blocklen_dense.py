#!/usr/bin/env python3
"""Diagnostic: config-2 bank, kernel and row-copy time by block length with DENSE rows (pitch = n), back-to-back
launches over a ring of buffers (steady clocks): how much of a 64K step is the launch's ramps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
CH = 1024
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
s = torch.cuda.current_stream()
TOTAL = 1 << 20
x = torch.empty(CH * TOTAL, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
gbank.synth_fill(x.view(CH, TOTAL), 1)
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(reps):
        fn(k)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (16384, 32768, 65536, 131072, 262144, 524288):
    nb = TOTAL // n                                   # block buffers of CH x n, dense
    kern = lambda k: bank.updaten_ptr(y.data_ptr() + (k % nb) * CH * n * 8, n, x.data_ptr() + (k % nb) * CH * n * 8, n, n, s)
    copy = lambda k: dspdll.dsp_copy_rows(C.c_void_p(y.data_ptr() + (k % nb) * CH * n * 8), n, C.c_void_p(x.data_ptr() + (k % nb) * CH * n * 8), n, CH, n, 256, 8192, C.c_void_p(s.cuda_stream))
    reps = max(16, (1 << 23) // n)
    timed(kern, reps)
    tk = timed(kern, reps)
    timed(copy, 8)
    tc = timed(copy, reps)
    print("n = %7d dense: kernel %8.1f us = %.0f GB/s (%.1f us per 64K)   copy %8.1f us = %.0f GB/s   kernel/copy %.3f"
          % (n, tk, CH * n * 16 / tk / 1e3, tk * 65536 / n, tc, CH * n * 16 / tc / 1e3, tk / tc), flush=True)
