#!/usr/bin/env python3
"""Diagnostic: does the distance between the input and the output block buffers matter?  One arena; x at its
base, y at base + 8 GiB + delta; config-2 kernel and row copy timed per delta (and per process: run it twice)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util
CH, BLOCK, NB = 1024, 65536, 16
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
stream = torch.cuda.current_stream()
BUF = NB * CH * BLOCK * 8
arena = torch.empty(2 * BUF + (1 << 30), dtype=torch.uint8, device="cuda")
base = (arena.data_ptr() + (1 << 21) - 1) >> 21 << 21
x = base
for j in range(NB):
    t = torch.empty(0)
    dspdll.dsp_synth_fill(C.c_void_p(x + j * CH * BLOCK * 8), BLOCK, CH, BLOCK, 1, 0, j * BLOCK, C.c_void_p(stream.cuda_stream))
torch.cuda.synchronize()
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(reps):
        fn(k)
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("arena @ %#x" % arena.data_ptr())
for delta in [0, 256, 4096, 1 << 16, 1 << 18, 1 << 20, 1 << 21, 3 << 20, 1 << 24, (1 << 24) + (1 << 12), 1 << 27, (1 << 28) + (1 << 21) + 4096, 1 << 29]:
    y = base + BUF + delta
    kern = lambda k: bank.updaten_ptr(y + (k % NB) * CH * BLOCK * 8, BLOCK, x + (k % NB) * CH * BLOCK * 8, BLOCK, BLOCK, stream)
    copy = lambda k: dspdll.dsp_copy_rows(C.c_void_p(y + (k % NB) * CH * BLOCK * 8), BLOCK, C.c_void_p(x + (k % NB) * CH * BLOCK * 8), BLOCK, CH, BLOCK, 256, 8192, C.c_void_p(stream.cuda_stream))
    timed(kern, 200)
    tk = timed(kern, 128)
    timed(copy, 8)
    tc = timed(copy, 64)
    print("delta %12d : kernel %.1f us (%.0f Gsamples/s)  copy %.1f us (%.0f GB/s)" % (delta, tk, CH * BLOCK / tk / 1e3, tc, CH * BLOCK * 16 / tc / 1e3), flush=True)
