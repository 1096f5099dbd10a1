#!/usr/bin/env python3
"""Config 2 (1024 x order-8 IIR, 64K blocks, state carried) through the four ways a block can be handed over:
rows / tiled layout x stream order / chained launches.  Same box, alternating, Msamples/s per variant."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dsp_amd import bank as gbank  # noqa: E402
from tests import util  # noqa: E402

C, B, NB = 1024, 1 << 16, 16
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
dev = torch.device("cuda", 0)
xrows = torch.empty((NB, C, B), dtype=torch.float64, device=dev)
for j in range(NB):
    gbank.synth_fill(xrows[j], 0x5EED0002, chan0=0, n0=j * B)
xtiles = torch.empty((NB, B // gbank.TILE, C, gbank.TILE), dtype=torch.float64, device=dev)
for j in range(NB):
    gbank.tile_rows(xrows[j], xtiles[j])
yrows = torch.empty_like(xrows)
ytiles = torch.empty_like(xtiles)
torch.cuda.synchronize()
stream = torch.cuda.current_stream()


def run(variant, nsteps):
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C, device=0)
    tiled = "tiled" in variant
    chained = "chained" in variant
    join_every = int(variant.split("/")[1]) if "/" in variant else 0
    nblk = int(variant.split("x")[1]) if "x" in variant else 1      # blocks per call (slope of one launch)
    x, y = (xtiles, ytiles) if tiled else (xrows, yrows)

    def step(k):
        j = k % NB
        if j == 0 and k > 0:
            bank.reset(stream)
        if nblk > 1:
            # nblk consecutive blocks as ONE tiled call: [nblk * 32][C][2048] is contiguous over block buffers
            if j % nblk == 0:
                bank.updaten_tiled_ptr(y[j].data_ptr(), x[j].data_ptr(), B * nblk, stream)
        elif chained:
            bank.updaten_chained_ptr(y[j].data_ptr(), B, x[j].data_ptr(), B, B, tiled=tiled, stream=stream)
            if join_every and (k + 1) % join_every == 0:
                bank.chain_join(stream)
        elif tiled:
            bank.updaten_tiled_ptr(y[j].data_ptr(), x[j].data_ptr(), B, stream)
        else:
            bank.updaten_ptr(y[j].data_ptr(), B, x[j].data_ptr(), B, B, stream)

    for k in range(2 * NB):
        step(k)
    bank.chain_join(stream)
    bank.reset(stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for k in range(nsteps):
        step(k)
    bank.chain_join(stream)
    e1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = e0.elapsed_time(e1)
    bank.close()
    return C * B * nsteps / (ms / 1e3) / 1e6, C * B * nsteps / wall / 1e6


variants = ["rows", "rows+chained", "tiled", "tiled+chained", "tiled+chained/2", "tiled+chained/4", "tiledx2", "tiledx4", "tiledx16"]
res = {v: [] for v in variants}
for r in range(reps):
    for v in variants:
        res[v].append(run(v, steps))
for v in variants:
    ev = [p[0] for p in res[v]]
    wl = [p[1] for p in res[v]]
    print("%-14s events: %s  median %.0f Msamples/s = %.3f of 8 TB/s | wall-clock median %.0f"
          % (v, " ".join("%.0f" % t for t in ev), np.median(ev), np.median(ev) * 16 / 8e6, np.median(wl)), flush=True)

# parity of the four hand-over forms on the last pass: tiled outputs == row outputs, bit for bit
for j in range(NB):
    assert torch.equal(gbank.untile_rows(ytiles[j].reshape(B // gbank.TILE, C, gbank.TILE)), yrows[j]), j
print("tiled+chained outputs == rows+chained outputs, all 16 blocks, bit for bit")
