"""N > 1 path on CPU: two gloo ranks exercise the sharding plumbing bench.py uses on GPUs (setup broadcast from rank 0,
stream partition, scalar aggregation, the parity configuration's PCM gather) and check that sharded synthesis == unsharded synthesis, stream for stream
(the oracle stands in for the GPU here: the point is the partitioning, not the arithmetic)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parseoggvorbis_amd import sharding
from tests.workloads import fixture_like_spec, synth_batch

STREAMS, PPK = 5, 9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle_binding import OracleSynth
        dev = torch.device("cpu")
        spec = fixture_like_spec(2) if rank == 0 else fixture_like_spec(1, 64, 64)  # only rank 0 knows the real setup
        spec = sharding.broadcast_spec(spec, dev, src=0)
        assert (spec.channels, spec.blocksize0, spec.blocksize1) == (2, 256, 2048)
        full = synth_batch(spec, STREAMS, PPK, "mixed", seed=77)  # same seeded corpus on every rank
        first, count = sharding.shard_range(STREAMS, rank, world)
        seg = full["segments"][first:first + count].copy()
        orc = OracleSynth(spec, STREAMS)
        r = orc.submit_host(full["packets"], seg, full["ys"], full["residue"], full["plane_stride"])
        assert r["rc"] == 0
        emitted = int(r["emit_len"].sum())
        dt, units, (samples,) = sharding.aggregate(0.5 + rank, count * PPK, dev, extra_sum=(emitted,))
        assert dt == pytest.approx(0.5 + world - 1) and units == STREAMS * PPK
        # the parity configuration's PCM gather: every rank ends up with every stream's PCM, in stream order
        frames = [int(r["emit_len"][int(sg["first_packet"]):int(sg["first_packet"]) + int(sg["num_packets"])].sum()) for sg in seg]
        pcm_all, frames_all = sharding.gather_pcm(r["pcm"], frames, dev)
        assert pcm_all.shape[0] == STREAMS and int(frames_all.sum()) == int(samples)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), first=first, count=count, samples=samples, pcm_all=pcm_all, frames_all=frames_all)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle.oracle_binding import OracleSynth
    spec = fixture_like_spec(2)
    full = synth_batch(spec, STREAMS, PPK, "mixed", seed=77)
    want = OracleSynth(spec, STREAMS).submit_host(full["packets"], full["segments"], full["ys"], full["residue"], full["plane_stride"])
    covered = []
    want_frames = [int(want["emit_len"][int(sg["first_packet"]):int(sg["first_packet"]) + int(sg["num_packets"])].sum()) for sg in full["segments"]]
    for rank in range(world):
        z = np.load(tmp_path / ("rank%d.npz" % rank))
        first, count = int(z["first"]), int(z["count"])
        covered += list(range(first, first + count))
        # what EVERY rank holds after the gather: all streams, bit for bit what one process computes (a rank's streams do not depend
        # on the others), with the frame counts that say how much of each plane is PCM
        assert list(z["frames_all"]) == want_frames
        plane = min(z["pcm_all"].shape[2], want["pcm"].shape[2])
        assert np.array_equal(z["pcm_all"][:, :, :plane].view(np.uint32), want["pcm"][:, :, :plane].view(np.uint32))
        assert not z["pcm_all"][:, :, plane:].any()
        assert int(z["samples"]) == int(want["emit_len"].sum())            # all-reduced total
    assert covered == list(range(STREAMS))


def test_shard_range_properties():
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            parts = [sharding.shard_range(n, r, w) for r in range(w)]
            assert sum(c for _, c in parts) == n
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_spec_codec_roundtrip():
    spec = fixture_like_spec(2)
    back = sharding.decode_spec(sharding.encode_spec(spec))
    assert (back.channels, back.blocksize0, back.blocksize1) == (spec.channels, spec.blocksize0, spec.blocksize1)
    assert [(m, list(x)) for m, x in back.floors] == [(m, list(x)) for m, x in spec.floors]
    assert back.mappings[0][0] == spec.mappings[0][0] and list(back.modes) == [tuple(m) for m in spec.modes]


# ---- second partitioning: packet ranges of ONE long stream, one-packet halo, no communication -----------------------
LONG_PPK = 61


def _long_stream(spec):
    b = synth_batch(spec, 1, LONG_PPK, "mixed", seed=123, granule_last=True)
    # page granules inside the stream too (every 7th packet ends a page), consistent with the block sizes
    abs_before, emit = sharding.stream_positions(spec, b["packets"])
    for q in range(6, LONG_PPK - 1, 7):
        b["packets"]["granule"][q] = int(abs_before[q] + emit[q])
    return b


def _range_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle_binding import OracleSynth
        dev = torch.device("cpu")
        spec = sharding.broadcast_spec(fixture_like_spec(2) if rank == 0 else fixture_like_spec(1, 64, 64), dev, src=0)
        sh = sharding.shard_packets(spec, _long_stream(spec), rank, world)
        r = OracleSynth(spec, 1).submit_host(sh["packets"], sh["segments"], sh["ys"], sh["residue"], sh["plane_stride"])
        assert r["rc"] == 0, r["flags"]
        assert int(r["emit_len"][:sh["halo"]].sum()) == 0  # the halo's output is dropped by construction
        frames = int(r["emit_len"].sum())
        _, units, (total,) = sharding.aggregate(0.0, sh["count"], dev, extra_sum=(frames,))
        assert units == LONG_PPK
        np.savez(os.path.join(out_dir, "range%d.npz" % rank), pcm=r["pcm"][0][:, :frames], offset=sh["frame_offset"], total=total,
                 emit=r["emit_len"][sh["halo"]:])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_packet_range_sharding_matches_unsharded_bit_for_bit(tmp_path, world):
    mp.spawn(_range_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle.oracle_binding import OracleSynth
    spec = fixture_like_spec(2)
    full = _long_stream(spec)
    want = OracleSynth(spec, 1).submit_host(full["packets"], full["segments"], full["ys"], full["residue"], full["plane_stride"])
    assert want["rc"] == 0
    total = int(want["emit_len"].sum())
    parts, emits, at = [], [], 0
    for rank in range(world):
        z = np.load(tmp_path / ("range%d.npz" % rank))
        assert int(z["offset"]) == at and int(z["total"]) == total
        parts.append(z["pcm"])
        emits.append(z["emit"])
        at += z["pcm"].shape[1]
    assert np.array_equal(np.concatenate(emits), want["emit_len"])
    got = np.concatenate(parts, axis=1)
    assert got.shape[1] == total
    assert np.array_equal(got.view(np.uint32), want["pcm"][0][:, :total].view(np.uint32))


def test_stream_positions_match_the_oracle_layout():
    from oracle.oracle_binding import OracleSynth
    spec = fixture_like_spec(2)
    b = _long_stream(spec)
    want = OracleSynth(spec, 1).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    _, emit = sharding.stream_positions(spec, b["packets"])
    assert np.array_equal(emit.astype(np.uint32), want["emit_len"])
