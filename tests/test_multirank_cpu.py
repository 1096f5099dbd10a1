"""N > 1 path on CPU: two gloo ranks shard a bank by channel, each filters its shard (oracle as the
stand-in compute, since there is no GPU here), the gather reassembles the full output, and the result
equals the single-process run bit for bit.  Covers dsp_amd/shard.py, which bench.py --gpus N uses."""
import os
import socket
import sys

import numpy as np
import pytest

from dsp_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_channel_ranges_partition_in_order():
    for nch in (1, 2, 7, 64, 1000, 1024, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            ranges = shard.all_ranges(nch, world)
            assert ranges[0][0] == 0 and ranges[-1][1] == nch
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.channel_range(10, 2, 2)


def _worker(rank, world, port, nch, n, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from tests import util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
        _, bf, _ = util.design_coefs("fir32_lp_0.25")
        designs = [dict(kind=orc.KIND_IIR, init_mode=orc.INIT_IIR_STABLE, b=b, a=a),
                   dict(kind=orc.KIND_FIR, init_mode=orc.INIT_FIR_STABLE, b=bf)]
        lo, hi = shard.channel_range(nch, world, rank)
        idx = [c % 2 for c in range(lo, hi)]
        bank = orc.Bank(designs, idx)
        outs = []
        for blk in range(2):                       # two blocks: state carried inside the shard
            x = np.stack([util.synth(42, c, blk * n, n) for c in range(lo, hi)])
            y = torch.from_numpy(bank.updaten(x))
            outs.append(shard.gather_rows(y, nch).numpy())
        if rank == 0:
            q.put(np.concatenate(outs, axis=1))
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_shard_and_gather_equals_single_process():
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    from tests import util
    nch, n, world = 7, 300, 2                       # odd channel count: uneven shards
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nch, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    _, bf, _ = util.design_coefs("fir32_lp_0.25")
    designs = [dict(kind=orc.KIND_IIR, init_mode=orc.INIT_IIR_STABLE, b=b, a=a),
               dict(kind=orc.KIND_FIR, init_mode=orc.INIT_FIR_STABLE, b=bf)]
    full = orc.Bank(designs, [c % 2 for c in range(nch)])
    ref = np.concatenate([full.updaten(np.stack([util.synth(42, c, blk * n, n) for c in range(nch)]))
                          for blk in range(2)], axis=1)
    assert util.bits_equal(got, ref)
