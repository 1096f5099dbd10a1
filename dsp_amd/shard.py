"""Channel sharding of a filter bank across ranks (one process per GPU).

The path has no cross-channel term (src/RTFilter.c:235-247, 396-439 touch one filter's own state),
so a bank of `nch` channels splits into contiguous, independent channel ranges; no collective is
needed to filter.  `gather_rows` is the optional output gather (RCCL all_gather on GPUs, gloo on
CPU) for a consumer that wants every channel's block in one place.
"""


def channel_range(nch, world_size, rank):
    """Contiguous range [lo, hi) of rank's channels; sizes differ by at most one, order preserved."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(nch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_ranges(nch, world_size):
    return [channel_range(nch, world_size, r) for r in range(world_size)]


def gather_rows(local_rows, nch, group=None):
    """all_gather of per-rank (rows_r, n) blocks into the full (nch, n) tensor, in channel order.

    Shards may differ by one row, so each rank pads to the largest shard before the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    ranges = all_ranges(nch, world)
    biggest = max(hi - lo for lo, hi in ranges)
    n = local_rows.shape[1]
    if nch % world == 0 and local_rows.is_contiguous():
        # equal shards: one collective straight into the result, no padding or reassembly copies
        full = torch.empty((nch, n), dtype=local_rows.dtype, device=local_rows.device)
        dist.all_gather_into_tensor(full, local_rows, group=group)
        return full
    padded = torch.zeros((biggest, n), dtype=local_rows.dtype, device=local_rows.device)
    padded[: local_rows.shape[0]] = local_rows
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(ranges)], dim=0)


def gather_rows_into(full, local_rows, group=None):
    """Equal shards only: all_gather of (rows, n) blocks into the preallocated (world*rows, n) tensor `full`
    (no allocation: what a streaming consumer calls once per wave).  Enqueued on torch's current stream."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(full, local_rows, group=group)
    return full
