#!/usr/bin/env python3
"""Diagnostic: config-2 bank on one 64K block per launch, ring of 8 block buffers, by ROW PITCH of the block buffer
(65536 doubles = 512 KiB, a power of two, plus a pad): kernel and row-copy time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
s = torch.cuda.current_stream()
CH, N, RING = 1024, 65536, 8
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(reps):
        fn(k)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for pad in (0, 16, 32, 64, 128, 256, 528, 1040, 2064, 4112, 0):
    P = N + pad
    x = torch.empty(RING * CH * P, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    gbank.synth_fill(x.view(RING * CH, P), 1)
    kern = lambda k: bank.updaten_ptr(y.data_ptr() + (k % RING) * CH * P * 8, P, x.data_ptr() + (k % RING) * CH * P * 8, P, N, s)
    copy = lambda k: dspdll.dsp_copy_rows(C.c_void_p(y.data_ptr() + (k % RING) * CH * P * 8), P, C.c_void_p(x.data_ptr() + (k % RING) * CH * P * 8), P, CH, N, 256, 8192, C.c_void_p(s.cuda_stream))
    timed(kern, 300)
    tk = timed(kern, 200)
    timed(copy, 16)
    tc = timed(copy, 100)
    print("pitch 65536 + %5d: kernel %7.1f us = %.0f GB/s   copy %7.1f us = %.0f GB/s   kernel/copy %.3f"
          % (pad, tk, CH * N * 16 / tk / 1e3, tc, CH * N * 16 / tc / 1e3, tk / tc), flush=True)
    del x, y
