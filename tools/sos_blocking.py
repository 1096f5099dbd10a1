#!/usr/bin/env python3
"""Diagnostic: a 1024-channel bank of butterworth(4, 0.3, 0.6) as FOUR second-order sections (one launch per
section, in place on the output rows) on a 64K block -- as one call, and as calls on sub-blocks of P samples (all
sections on one sub-block before the next), so that a sub-block's rows can stay in the 256 MiB Infinity Cache between
the sections.  Reported per 64K block; the transfer-function form of the same filter beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank, dsp
from tests import util
s = torch.cuda.current_stream()
CH, N, RING = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 65536, 8
secs = dsp.design_sos("butterworth", 4, 0.3, 0.6)
sos = gbank.FilterBank([gbank.sos_design(secs)], [0] * CH)
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
tf = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
x = torch.empty(RING * CH * N, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
gbank.synth_fill(x.view(RING * CH, N), 1)
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(reps):
        fn(k)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def run(bank, P):
    def fn(k):
        base = (k % RING) * CH * N * 8
        for off in range(0, N, P):
            bank.updaten_ptr(y.data_ptr() + base + off * 8, N, x.data_ptr() + base + off * 8, N, P, s)
    return fn
t = timed(run(tf, N), 50); t = timed(run(tf, N), 100)
print("%d channels, transfer-function form, one call per 64K block: %8.1f us = %.0f Msamples/s" % (CH, t, CH * N / t), flush=True)
for P in (65536, 32768, 16384, 8192, 4096):
    timed(run(sos, P), 20)
    t = timed(run(sos, P), 50)
    print("%d channels, 4 sections, sub-blocks of %5d samples: %8.1f us per 64K block = %.0f Msamples/s" % (CH, P, t, CH * N / t), flush=True)
