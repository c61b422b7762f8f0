"""Multi-GPU plumbing of the path: one process per GPU, streams partitioned across ranks, no data-path collective
(all synthesis state is per stream, reference hpp:1117-1123).  Collectives used: ONE broadcast of the stream setup
(the job split) and scalar all-reduces of clock / counters; PCM stays where it was produced.  Backend-agnostic:
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  torch.distributed is plumbing here, not the product.
"""
import numpy as np

from .binding import SetupSpec


def shard_range(num_items, rank, world):
    """Contiguous, balanced partition of `num_items` streams (or files): -> (first, count) for `rank`."""
    base, extra = divmod(num_items, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def encode_spec(spec):
    """SetupSpec -> flat int32 vector (what rank 0 broadcasts)."""
    v = [spec.channels, spec.blocksize0, spec.blocksize1, len(spec.floors), len(spec.mappings), len(spec.modes)]
    for mult, xs in spec.floors:
        v += [mult, len(xs)] + [int(x) for x in xs]
    for coups, chfloor in spec.mappings:
        v += [len(coups)] + [int(t) for c in coups for t in c] + [int(f) for f in chfloor]
    for bf, m in spec.modes:
        v += [int(bf), int(m)]
    return np.asarray(v, np.int32)


def decode_spec(v):
    v = [int(x) for x in v]
    C, bs0, bs1, nf, nm, nmodes = v[:6]
    p = 6
    floors, maps, modes = [], [], []
    for _ in range(nf):
        mult, n = v[p], v[p + 1]
        floors.append((mult, v[p + 2:p + 2 + n]))
        p += 2 + n
    for _ in range(nm):
        nc = v[p]
        p += 1
        coups = [(v[p + 2 * i], v[p + 2 * i + 1]) for i in range(nc)]
        p += 2 * nc
        maps.append((coups, v[p:p + C]))
        p += C
    for _ in range(nmodes):
        modes.append((v[p], v[p + 1]))
        p += 2
    assert p == len(v)
    return SetupSpec(C, bs0, bs1, floors, maps, modes)


def broadcast_spec(spec, device, src=0):
    """Rank `src` owns the setup; everyone returns an identical SetupSpec. No-op without an initialised group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return spec
    rank = dist.get_rank()
    n = torch.tensor([len(encode_spec(spec)) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    buf = torch.from_numpy(encode_spec(spec)).to(device) if rank == src else torch.zeros(int(n.item()), dtype=torch.int32, device=device)
    dist.broadcast(buf, src)
    return decode_spec(buf.cpu().numpy())


def aggregate(dt_seconds, units, device, extra_sum=()):
    """-> (max dt over ranks, total units, tuple of summed extras). Scalars only."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dt_seconds, units, tuple(extra_sum)
    t = torch.tensor([dt_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([float(units)] + [float(x) for x in extra_sum], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(s[0].item())), tuple(float(x) for x in s[1:])
