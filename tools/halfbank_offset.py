#!/usr/bin/env python3
"""Config 2 as two half-banks of 512 channels on two streams, tiled layout, each paced by itself: does one half's
steady state cover the other's fill and drain when the two are half a launch out of phase?
usage: halfbank_offset.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from tests import util
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, NB = 65536, 8
d = gbank.iir_design(b, a)

def buffers(ch):
    x = [torch.empty(ch * n, dtype=torch.float64, device="cuda").uniform_(-1, 1) for _ in range(NB)]
    y = [torch.empty_like(t) for t in x]
    return x, y

def run(label, banks, streams, offsets_us):
    bufs = [buffers(bk.nch) for bk in banks]
    def launch_all(k):
        for bk, st, (x, y) in zip(banks, streams, bufs):
            bk.updaten_tiled_ptr(y[k % NB].data_ptr(), x[k % NB].data_ptr(), n, st)
    for k in range(64): launch_all(k)
    torch.cuda.synchronize()
    for st, off in zip(streams, offsets_us):
        if off:
            with torch.cuda.stream(st): torch.cuda._sleep(int(off * 2000))          # ~2 GHz cycles
    e0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    t0 = torch.cuda.Event(enable_timing=True); t0.record(torch.cuda.current_stream())
    for st, e in zip(streams, e0): e.record(st)
    for k in range(steps): launch_all(k)
    for st, e in zip(streams, e1): e.record(st)
    torch.cuda.synchronize()
    ms = max(a_.elapsed_time(b_) for a_, b_ in zip(e0, e1))
    tot = sum(bk.nch for bk in banks) * n * steps
    print("%-52s %.1f us/step  %.0f Msamples/s  %.0f GB/s" % (label, ms / steps * 1e3, tot / ms / 1e3, tot * 16 / ms / 1e6), flush=True)

s0 = torch.cuda.current_stream()
one = gbank.FilterBank([d], [0] * 1024)
run("one bank of 1024, one stream", [one], [s0], [0])
halves = [gbank.FilterBank([d], [0] * 512), gbank.FilterBank([d], [0] * 512)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for off in (0, 50, 100, 150):
    run("two banks of 512, two streams, offset %3d us" % off, halves, [sa, sb], [0, off])
