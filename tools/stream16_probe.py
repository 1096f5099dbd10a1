#!/usr/bin/env python3
"""Config 5's filter (order 16) over one stream: the interleaved-segment kernel (pipe16_kernel.hip, stream mode)
against the oracle at small sizes, then its speed on 2^30 samples next to the contiguous time split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from dsp_amd import stream as gstream
from oracle import oracle as orc
from tests import util

_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
s = torch.cuda.current_stream()

if "--no-check" not in sys.argv:
    for grid in (1, 3, 8, 64, 1024):
        fs = gstream.FilterStream(gbank.iir_design(b, a))
        assert fs.interleaved
        fs.set_interleave(grid)
        ref = orc.iir(b, a)
        pos = 0
        for n in (200000, 2048 * 64, 5, 1 << 20, 2048 * 700 + 13):
            x = util.synth(11, 3, pos, n)
            xd = torch.from_numpy(x).cuda()
            yd = fs(xd)
            torch.cuda.synchronize()
            want = ref.updaten(x)
            err = np.max(np.abs(yd.cpu().numpy() - want)) / np.max(np.abs(want))
            print("grid %4d  n %8d  split %4d  err %.3e" % (grid, n, fs.last_split, err), flush=True)
            assert err <= 1e-12
            pos += n
        st, fv, ini = fs.get_state()
        assert np.max(np.abs(st[:len(ref.state)] - ref.state)) <= 1e-12 * max(1.0, np.max(np.abs(ref.state)))
        fs.close()

lg = 30
n = 1 << lg
x = torch.empty((1, n), dtype=torch.float64, device="cuda")
gbank.synth_fill(x, 7)
x = x.view(-1)
y = torch.empty_like(x)
for mode in (1024, -1, 1024, -1, 512, 768):
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    fs.set_interleave(mode)
    for _ in range(3):
        fs(x, y, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(6):
        fs(x, y, s)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 6
    print("interleave %4d: %.3f ms per 2^%d samples = %.0f Msamples/s = %.0f GB/s (split %d)"
          % (mode, ms, lg, n / ms / 1e3, n * 16 / ms / 1e6, fs.last_split), flush=True)
    fs.close()
