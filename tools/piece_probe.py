#!/usr/bin/env python3
"""Diagnostic for config 5: one 2^30-sample array cut into independent rows of n samples (dense: row c = samples
[c n, (c+1) n)), by row length: row copy and the order-16 scan kernel on independent channels.  Shorter rows =
smaller window of concurrently touched memory (the launch's workgroups are dispatched in row order)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
s = torch.cuda.current_stream()
TOTAL = 1 << 30
x = torch.empty(TOTAL, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
gbank.synth_fill(x.view(1024, -1), 1)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(reps):
        fn(k)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for lg in (20, 18, 16, 15, 14, 13, 12, 20):
    N = 1 << lg
    CH = TOTAL // N
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
    kern = lambda k: bank.updaten_ptr(y.data_ptr(), N, x.data_ptr(), N, N, s)
    res = []
    for thr, seg in ((256, 8192), (64, 2048)):
        if N < seg:
            res.append(float("nan"))
            continue
        copy = lambda k: dspdll.dsp_copy_rows(C.c_void_p(y.data_ptr()), N, C.c_void_p(x.data_ptr()), N, CH, N, thr, seg, C.c_void_p(s.cuda_stream))
        timed(copy, 3)
        res.append(timed(copy, 8))
    timed(kern, 4)
    tk = timed(kern, 8)
    print("%7d rows x 2^%d: kernel %7.1f us = %.0f GB/s (%.0f Msamples/s)   copy(256 thr) %.0f GB/s  copy(64 thr) %.0f GB/s"
          % (CH, lg, tk, TOTAL * 16 / tk / 1e3, TOTAL / tk, TOTAL * 16 / res[0] / 1e3, TOTAL * 16 / res[1] / 1e3), flush=True)
    bank.close()
