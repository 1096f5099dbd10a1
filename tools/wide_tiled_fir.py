#!/usr/bin/env python3
"""Config 4's FIR wave (65536 channels x 64K, 32 taps) on dense rows and in the tiled block layout.
usage: wide_tiled_fir.py [channels] [n] [key]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
key = sys.argv[3] if len(sys.argv) > 3 else "fir32_lp_0.25"
pitch = n + 128
x = torch.empty(Cn * pitch, dtype=torch.float64, device=dev).uniform_(-1, 1)
y = torch.empty_like(x)
if key.startswith("rand"):
    b = np.random.default_rng(5).normal(size=int(key[4:])) / int(key[4:])
else:
    _, b, _ = util.design_coefs(key)

def timed(call, nsteps=4):
    call(); call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(nsteps): call()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nsteps

def show(what, ms):
    print("%-34s %.3f ms  %.0f Msamples/s  %.2f TB/s" % (what, ms, Cn * n / ms / 1e3, Cn * n * 16 / ms / 1e9), flush=True)

for mode, name in ((gbank.MODE_AUTO, "AUTO"), (gbank.MODE_SEQUENTIAL, "SEQUENTIAL (bit-exact)")):
    bank = gbank.FilterBank([gbank.fir_design(b)], [0] * Cn, device=0, mode=mode)
    show("dense rows, pitch n+128, %s" % name, timed(lambda: bank.updaten_ptr(y.data_ptr(), pitch, x.data_ptr(), pitch, n, stream)))
    bank.close()
for mode, name in ((gbank.MODE_AUTO, "AUTO"), (gbank.MODE_SEQUENTIAL, "SEQUENTIAL (bit-exact)")):
    for pace in (-1, 0):
        bank = gbank.FilterBank([gbank.fir_design(b)], [0] * Cn, device=0, mode=mode)
        bank.set_pace(pace)
        try:
            show("tiled, %s, %s" % (name, "paced" if pace else "no schedule"),
                 timed(lambda: bank.updaten_tiled_ptr(y.data_ptr(), x.data_ptr(), n, stream)))
        except Exception as e:
            print("tiled, %s: refused (%s)" % (name, str(e)[:60]))
        bank.close()
