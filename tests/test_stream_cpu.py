"""Config 5's N > 1 logic on CPU: two gloo ranks split ONE stream in time and hand the filter state
over (dsp_amd/stream.py: time_range, split_update).  There is no GPU here, so the engine behind the
hand-off protocol is the oracle (test infrastructure); on the GPU the same protocol drives
RTFilterStream (tests/test_gpu_stream.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from dsp_amd import stream as gstream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-12      # north star: float64 outputs within 1e-12 (relative to the output peak)


def test_time_ranges_partition_in_order():
    for n in (0, 1, 2047, 2048, 4096, 100000, (1 << 20) + 17, 1 << 30):
        for world in (1, 2, 3, 4, 8):
            rs = [gstream.time_range(n, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert all(lo % 2048 == 0 for lo, hi in rs if lo < n)
    with pytest.raises(ValueError):
        gstream.time_range(10, 2, 2)


class OracleEngine:
    """The hand-off protocol of dsp_amd.stream.FilterStream on top of the CPU oracle."""

    def __init__(self, b, a, decay):
        from oracle import oracle as orc
        self._orc = orc
        self.b, self.a = b, a
        self.f = orc.iir(b, a)
        self.decay = abs(decay)               # 0: no tail shortcut
        self.decay_length = self.decay
        self.exact_split = decay == 0         # decay 0 -> chunk maps composed; decay < 0 -> neither: ranks chain
        if decay < 0:
            self.decay = self.decay_length = 0

    def zero_state(self, stream=None):
        self.f.state[:] = 0.0
        self.f.filtered_value = 0.0
        self.f.initialized = 0

    def compose_state(self, packs, count, chunk_len, stream=None):
        """s = A^L( ... (A^L p_0 + p_1) ... ) + p_(count-1): A^L applied by running L zero samples (x = 0 gives
        s' = A s exactly as the kernels' matrix does up to rounding)."""
        s = packs[0, :-2].numpy().copy()
        for j in range(1, count):
            h = self._orc.iir(self.b, self.a)
            h.initialized = 0
            h.state[:] = s
            h.updaten(np.zeros(chunk_len))
            s = h.state + packs[j, :-2].numpy()
        self.f.state[:] = s
        self.f.filtered_value = float(packs[count - 1, -2])
        self.f.initialized = 0

    def new_pack(self):
        import torch
        return torch.zeros(len(self.f.state) + 2, dtype=torch.float64)

    def tail_state(self, x, pack, stream=None):
        if not self.decay or x.numel() < self.decay:
            return False
        t = self._orc.iir(self.b, self.a)
        t.initialized = 0                     # already initialised, zero state
        t.updaten(x.numpy()[-self.decay:])
        pack[:-2] = __import__("torch").from_numpy(t.state)
        pack[-2] = t.filtered_value
        pack[-1] = 0.0
        return True

    def export_state(self, pack, stream=None):
        import torch
        pack[:-2] = torch.from_numpy(self.f.state)
        pack[-2] = self.f.filtered_value
        pack[-1] = float(self.f.initialized)
        return pack

    def import_state(self, pack, stream=None):
        self.f.state[:] = pack[:-2].numpy()
        self.f.filtered_value = float(pack[-2])
        self.f.initialized = int(pack[-1])

    def __call__(self, x, out=None, stream=None):
        import torch
        return torch.from_numpy(self.f.updaten(x.numpy()))


def _worker(rank, world, port, n, decay, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from tests import util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
        eng = OracleEngine(b, a, decay)
        outs = []
        for blk in range(2):                  # two calls: the end state must reach rank 0 again
            lo, hi = gstream.time_range(n, world, rank)
            x = torch.from_numpy(util.synth(7, 0, blk * n + lo, hi - lo))
            y = gstream.split_update(eng, x)
            parts = [None] * world
            dist.all_gather_object(parts, y.numpy())
            outs.append(np.concatenate(parts))
        gstream.sync_end_state(eng)
        snap = eng.f.snapshot()
        states = [None] * world
        dist.all_gather_object(states, snap["state"])
        if rank == 0:
            q.put((np.concatenate(outs), states))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("decay", [1024, 0, -1], ids=["tail-handoff", "composed", "chained"])
def test_two_gloo_ranks_split_one_stream_in_time(decay):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    from tests import util
    n, world = 3 * 4096 + 777, 3 if decay == 0 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, decay, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, states = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    ref = orc.iir(b, a)
    want = ref.updaten(util.synth(7, 0, 0, 2 * n))
    if decay < 0:
        assert util.bits_equal(got, want)                       # chained hand-off of the true state: exact
        assert all(util.bits_equal(s, ref.state) for s in states)
    else:
        assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))
        assert all(np.max(np.abs(s - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state))) for s in states)
