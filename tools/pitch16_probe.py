#!/usr/bin/env python3
"""Diagnostic for config 5: the order-16 scan kernel on C rows of n samples by ROW PITCH (the pitch is the
sub-chunk length of the time-split stream: 2^20 samples = a power of two in bytes today): kernel and row copy."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
s = torch.cuda.current_stream()


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(reps):
        fn(k)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


SEG = 4096
for CH, nseg, pad in ((1024, 256, 0), (1024, 256, 128), (1024, 256, 2048), (1024, 256, 4096), (1020, 257, 0),
                      (1024, 255, 0), (1008, 260, 0), (1024, 256, 0)):
    N = nseg * SEG
    P = N + pad
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
    x = torch.empty(CH * P + 4096, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    gbank.synth_fill(x[:CH * P].view(CH, P), 1)
    kern = lambda k: bank.updaten_ptr(y.data_ptr(), P, x.data_ptr(), P, N, s)
    copy = lambda k: dspdll.dsp_copy_rows(C.c_void_p(y.data_ptr()), P, C.c_void_p(x.data_ptr()), P, CH, N, 256, 8192, C.c_void_p(s.cuda_stream))
    timed(kern, 6)
    tk = timed(kern, 10)
    timed(copy, 4)
    tc = timed(copy, 10)
    print("%4d rows x %3d segments, pitch n + %4d: kernel %7.1f us = %.0f GB/s (%.0f Msamples/s)   copy %7.1f us = %.0f GB/s   kernel/copy %.3f"
          % (CH, nseg, pad, tk, CH * N * 16 / tk / 1e3, CH * N / tk, tc, CH * N * 16 / tc / 1e3, tk / tc), flush=True)
    bank.close()
    del x, y
