"""N > 1 path on CPU: two gloo ranks exercise the sharding plumbing bench.py uses on GPUs (setup broadcast from rank 0,
stream partition, scalar aggregation) and check that sharded synthesis == unsharded synthesis, stream for stream
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
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), first=first, count=count, pcm=r["pcm"], samples=samples)
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
    for rank in range(world):
        z = np.load(tmp_path / ("rank%d.npz" % rank))
        first, count = int(z["first"]), int(z["count"])
        covered += list(range(first, first + count))
        assert np.array_equal(z["pcm"], want["pcm"][first:first + count])  # a rank's streams do not depend on the others
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
