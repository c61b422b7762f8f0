"""Multi-GPU plumbing of the path: one process per GPU, streams partitioned across ranks, no data-path collective
(all synthesis state is per stream, reference hpp:1117-1123).  Collectives used: ONE broadcast of the stream setup
(the job split) and scalar all-reduces of clock / counters; PCM stays where it was produced in the throughput configurations and
is gathered (gather_pcm: two all_gathers) only in the small parity configuration.  Backend-agnostic:
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  torch.distributed is plumbing here, not the product.
"""
import numpy as np

from .binding import SetupSpec


def shard_range(num_items, rank, world):
    """Contiguous, balanced partition of `num_items` streams (or files): -> (first, count) for `rank`."""
    base, extra = divmod(num_items, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def stream_positions(spec, packets, abs0=0):
    """Host restatement of the stream bookkeeping (reference hpp:1019-1067, what the layout kernel scans on the device) for ONE
    stream starting at absolute position abs0 with no block before it: -> (abs_before [P], emit [P]) as int64.
    L_q = n_{q-1}/4 + n_q/4 (0 for the first packet); abs_after_q = granule_q if granule_q >= 0 else abs_before_q + L_q."""
    P = len(packets)
    n = np.where(np.asarray([spec.modes[int(m)][0] for m in packets["mode"]], bool), spec.blocksize1, spec.blocksize0).astype(np.int64)
    L = np.zeros(P, np.int64)
    L[1:] = n[:-1] // 4 + n[1:] // 4
    gran = packets["granule"].astype(np.int64)
    has = gran >= 0
    # abs_after[q] = granule of the last packet j <= q that carries one, plus the natural frames after it
    idx = np.where(has, np.arange(P), -1)
    last = np.maximum.accumulate(idx)
    cl = np.cumsum(L)
    base = np.where(last >= 0, gran[np.maximum(last, 0)] - cl[np.maximum(last, 0)], abs0)
    abs_after = base + cl
    abs_before = np.concatenate([[abs0], abs_after[:-1]])
    return abs_before, abs_after - abs_before


def shard_packets(spec, batch, rank, world):
    """Second partitioning of SURVEY 8e: ONE long stream cut into contiguous packet ranges, one per rank, no communication.
    Packet p depends on p-1 only through the overlap-add (hpp:1008-1017, 1061-1109), so a rank re-decodes the last packet of
    its left neighbour's range (a one-packet halo) and drops that packet's output: its segment starts at the halo with
    VSYN_SEG_RESET — the first block of a stream emits nothing (hpp:1021-1027) — and the page granules are rebased to the
    segment's own origin. The concatenation of the ranks' PCM is the unsharded PCM, bit for bit (same blocks, same sums).

    batch: a ONE-segment batch (dict packets/segments/ys/residue/plane_stride as tests.workloads.synth_batch makes it) whose
    segment starts its stream. -> dict with the rank's batch (packets, segments, ys, residue, plane_stride) plus
    `first`, `count` (its packets of the original stream), `halo` (0 or 1 leading packets whose output is dropped) and
    `frame_offset` (absolute position of its first emitted frame)."""
    from .binding import VSYN_SEG_RESET
    assert len(batch["segments"]) == 1
    pk_all = batch["packets"]
    P = len(pk_all)
    C = spec.channels
    first, count = shard_range(P, rank, world)
    halo = 1 if (first > 0 and count > 0) else 0
    lo, hi = first - halo, first + count
    n = np.where(np.asarray([spec.modes[int(m)][0] for m in pk_all["mode"]], bool), spec.blocksize1, spec.blocksize0).astype(np.int64)
    res_off = int(batch["segments"][0]["residue_off"]) + np.concatenate([[0], np.cumsum(C * (n // 2))])
    abs_before, _ = stream_positions(spec, pk_all)
    pk = pk_all[lo:hi].copy()
    origin = int(abs_before[first]) if count else 0  # the halo emits nothing: the segment's frame 0 is the first frame of packet `first`
    if halo:
        pk["granule"][0] = -1  # its page end belongs to the left neighbour
    g = pk["granule"].astype(np.int64)
    pk["granule"] = np.where(g >= 0, g - origin, -1)
    seg = batch["segments"][:1].copy()
    seg[0] = (0, 0, hi - lo, VSYN_SEG_RESET, 0)
    return dict(packets=pk, segments=seg, ys=batch["ys"][lo:hi], residue=batch["residue"][int(res_off[lo]):int(res_off[hi])],
                plane_stride=max(int(count) * (spec.blocksize1 // 2), 1) + 64, first=first, count=count, halo=halo, frame_offset=origin)


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


def gather_pcm(pcm, frames, device):
    """The parity configuration's PCM gather ("ncclAllGather/gatherv of PCM only in the small parity config").
    pcm: this rank's planar PCM [S_local][C][plane] (numpy or torch, float32); frames: emitted frames per local stream [S_local].
    Two collectives: an all_gather of the per-rank shapes (stream count, plane length) and stream frame counts, then an
    all_gather of the PCM padded to the largest (streams x plane) of any rank — the gatherv. Every rank returns
    (pcm_all [S_total][C][max plane], frames_all [S_total]) with the ranks' streams in rank order, i.e. in shard_range order.
    Without an initialised group (or world size 1): the inputs, unchanged in content."""
    import torch
    import torch.distributed as dist
    pcm_t = torch.as_tensor(np.ascontiguousarray(pcm) if isinstance(pcm, np.ndarray) else pcm, dtype=torch.float32)
    fr_t = torch.as_tensor(np.asarray(frames, np.int64))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pcm_t.cpu().numpy(), fr_t.cpu().numpy()
    world = dist.get_world_size()
    S, C, plane = (int(x) for x in pcm_t.shape) if pcm_t.numel() or pcm_t.dim() == 3 else (0, 0, 0)
    shape = torch.tensor([S, C, plane], dtype=torch.int64, device=device)
    shapes = [torch.zeros(3, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(shapes, shape)
    shapes = [[int(v) for v in t.cpu()] for t in shapes]
    max_s = max(sh[0] for sh in shapes)
    max_plane = max(sh[2] for sh in shapes)
    chans = max(sh[1] for sh in shapes)
    assert all(sh[1] in (0, chans) for sh in shapes), "ranks disagree on the channel count"
    # frame counts and PCM, padded to the largest shard (all_gather needs equal shapes on every rank)
    fpad = torch.zeros(max(max_s, 1), dtype=torch.int64, device=device)
    fpad[:S] = fr_t.to(device)
    fall = [torch.zeros_like(fpad) for _ in range(world)]
    dist.all_gather(fall, fpad)
    ppad = torch.zeros((max(max_s, 1), max(chans, 1), max(max_plane, 1)), dtype=torch.float32, device=device)
    if S:
        ppad[:S, :, :plane] = pcm_t.to(device)
    pall = [torch.zeros_like(ppad) for _ in range(world)]
    dist.all_gather(pall, ppad)
    pcm_all = torch.cat([pall[r][:shapes[r][0]] for r in range(world)], dim=0)
    frames_all = torch.cat([fall[r][:shapes[r][0]] for r in range(world)], dim=0)
    return pcm_all.cpu().numpy(), frames_all.cpu().numpy()
