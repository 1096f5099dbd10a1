#!/usr/bin/env python3
"""Diagnostic: is the run-to-run spread of config 2 (and of the row copy) a property of WHERE the block buffers
lie, or of WHEN the measurement is taken?  One process, several allocations of the same 2 x 16 block buffers
(the cache emptied and a dummy of random size allocated in between, so that the virtual and physical placement
change), each measured twice a second apart: kernel and row copy, per block buffer."""
import ctypes as C
import os
import sys
import time
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
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(reps):
        fn(k)
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def measure(x, y, pitch):
    xp = [x[j].data_ptr() for j in range(NB)]
    yp = [y[j].data_ptr() for j in range(NB)]
    kern = lambda k: bank.updaten_ptr(yp[k % NB], pitch, xp[k % NB], pitch, BLOCK, stream)
    copy = lambda k: dspdll.dsp_copy_rows(C.c_void_p(yp[k % NB]), pitch, C.c_void_p(xp[k % NB]), pitch, CH, BLOCK,
                                          256, 8192, C.c_void_p(stream.cuda_stream))
    timed(kern, 16)
    tk = timed(kern, 64)
    timed(copy, 4)
    tc = timed(copy, 32)
    per_block = [timed(lambda k, j=j: kern(j), 4) for j in range(NB)]
    return tk, tc, per_block


for trial in range(8):
    pitch = BLOCK
    dummy = torch.empty(int(rng.integers(1, 4096)) << 18, dtype=torch.uint8, device="cuda") if trial else None
    x = torch.empty((NB, CH, pitch), dtype=torch.float64, device="cuda")
    y = torch.empty((NB, CH, pitch), dtype=torch.float64, device="cuda")
    for j in range(NB):
        gbank.synth_fill(x[j], 1, n0=j * BLOCK)
    torch.cuda.synchronize()
    for rep in range(2):
        tk, tc, pb = measure(x, y, pitch)
        print("trial %d.%d  x @ %#x  y @ %#x : kernel %.1f us (%.0f Gsamples/s)  copy %.1f us (%.0f GB/s)  per block min %.1f max %.1f"
              % (trial, rep, x.data_ptr(), y.data_ptr(), tk, CH * BLOCK / tk / 1e3, tc, CH * BLOCK * 16 / tc / 1e3,
                 min(pb), max(pb)), flush=True)
        if rep == 0:
            time.sleep(1.0)
    del x, y, dummy
    torch.cuda.empty_cache()
