#!/usr/bin/env python3
"""Diagnostic: the same 1 GiB of config-2 work per launch as (channels x samples) = 1024 x 64K (one resident round
of workgroups), 2048 x 32K, 4096 x 16K, 8192 x 8K (2, 4, 8 rounds) -- what a launch that is several rounds deep would
gain; and more channels at 64K per channel.  Dense block buffers, back-to-back launches over a ring of 8 buffers."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
s = torch.cuda.current_stream()
RING = 8
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(reps):
        fn(k)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for ch, n in ((1024, 65536), (2048, 32768), (4096, 16384), (8192, 8192), (1024, 65536), (2048, 65536), (4096, 65536), (8192, 65536)):
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * ch)
    x = torch.empty(RING * ch * n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    gbank.synth_fill(x.view(RING * ch, n), 1)
    kern = lambda k: bank.updaten_ptr(y.data_ptr() + (k % RING) * ch * n * 8, n, x.data_ptr() + (k % RING) * ch * n * 8, n, n, s)
    copy = lambda k: dspdll.dsp_copy_rows(C.c_void_p(y.data_ptr() + (k % RING) * ch * n * 8), n, C.c_void_p(x.data_ptr() + (k % RING) * ch * n * 8), n, ch, n, 256, 8192, C.c_void_p(s.cuda_stream))
    timed(kern, 300)
    tk = timed(kern, 200)
    timed(copy, 16)
    tc = timed(copy, 100)
    print("%5d channels x %6d samples: kernel %8.1f us = %.0f GB/s   copy %8.1f us = %.0f GB/s   kernel/copy %.3f"
          % (ch, n, tk, ch * n * 16 / tk / 1e3, tc, ch * n * 16 / tc / 1e3, tk / tc), flush=True)
    bank.close(); del x, y
