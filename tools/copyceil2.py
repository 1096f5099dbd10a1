#!/usr/bin/env python3
"""Diagnostic: device copy ceilings (torch copy_, row-streaming copy) vs size and leading dimension."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd.cdsp import dspdll
fn = dspdll.dsp_copy_rows
fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]

def timeit(f, reps=8):
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:]))

for mib in (256, 512, 2048, 8192):
    nel = mib * (1 << 20) // 8
    x = torch.rand(nel, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print("torch copy_ %5d MiB each way: %.3f ms  %.0f GB/s" % (mib, t, 2 * nel * 8 / t / 1e6))
    del x, y
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 1 << 16
for pad in (0, 96, 1056):
    ld = (1 << 20) + pad
    xb = torch.rand(1024 * ld, dtype=torch.float64, device="cuda"); yb = torch.empty_like(xb)
    for n, label in ((B, "one 64K block"), (1 << 20, "whole 1M stream")):
        for threads, seg in ((128, 4096), (256, 8192)):
            t = timeit(lambda: fn(yb.data_ptr(), ld, xb.data_ptr(), ld, 1024, n, threads, seg, st))
            print("row copy ld=2^20+%-4d %-16s threads %3d seg %4d: %.3f ms  %.0f GB/s" % (pad, label, threads, seg, t, 1024 * n * 16 / t / 1e6))
    del xb, yb
