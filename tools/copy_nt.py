#!/usr/bin/env python3
"""Diagnostic: does a nontemporal hint on the loads / stores move the copy ceiling? (DSP_COPY_NT is read per call)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd.cdsp import dspdll
fn = dspdll.dsp_copy_rows
fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
L, B, nch = (1 << 20) + 128, 1 << 16, 1024
x = torch.rand((nch, L), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
for rnd in range(2):
    for nt in (0, 1, 2, 3):
        os.environ["DSP_COPY_NT"] = str(nt)
        ts = []
        for rep in range(20):
            j = rep % 16
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            assert fn(y.data_ptr() + j * B * 8, L, x.data_ptr() + j * B * 8, L, nch, B, 256, 4096,
                      C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = np.median(ts[4:])
        print("nt=%d: %.4f ms  %.0f GB/s" % (nt, t, nch * B * 16 / t / 1e6))
