"""tools/stress_u.py [first_seed] [count] — one-off stress of the size-generic fused kernel on the GPU box: random channel counts,
block-size pairs, coupling lists, floor shapes and block patterns, one submit == oracle and cut into several submits == oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle_binding as ob  # noqa: E402
from parseoggvorbis_amd import binding  # noqa: E402
from parseoggvorbis_amd.binding import SetupSpec  # noqa: E402
from tests.workloads import synth_batch  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
SIZES = [64, 128, 256, 512, 1024, 2048, 4096, 8192]
worst = 0.0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    C = int(rng.choice([1, 2, 2, 3, 4, 5, 6, 2, 13, 16]))
    i1 = int(rng.integers(0, len(SIZES)))
    i0 = int(rng.integers(0, min(i1, 5) + 1))
    if rng.random() < 0.15:
        i0 = i1  # equal sizes, 4096 / 8192 included: every block on the register-set path
    bs0, bs1 = SIZES[i0], SIZES[i1]

    def xs(n2, posts):
        posts = min(posts, n2 + 1)
        inner = rng.choice(np.arange(1, n2), posts - 2, replace=False) if posts > 2 else np.zeros(0, np.int64)
        return [0, n2] + [int(v) for v in inner]
    floors = [(int(rng.integers(1, 5)), xs(bs0 // 2, int(rng.choice([65, 64, 33]) if rng.random() < 0.1 else rng.integers(2, 30)))),
              (int(rng.integers(1, 5)), xs(bs1 // 2, 65 if rng.random() < 0.15 else int(rng.integers(2, 65))))]
    def coup():
        out = []
        for _ in range(int(rng.integers(0, 4)) if 1 < C <= (16 if bs1 <= 2048 else 8) else 0):  # (more channels than a workgroup has waves: uncoupled, the layout that stays fused)
            m, a = rng.choice(C, 2, replace=False)
            out.append((int(m), int(a)))
        return out
    spec = SetupSpec(C, bs0, bs1, floors, [(coup(), [0] * C), (coup(), [1] * C)], [(0, 0), (1, 1)])
    npk = int(rng.integers(3, 60 if bs1 <= 2048 else 24))
    streams = int(rng.integers(1, 4))
    flags = (rng.random(npk) < 0.5).astype(np.uint8)
    b = synth_batch(spec, streams, npk, flags, seed=seed, unused_frac=0.2, granule_last=bool(rng.integers(0, 2)), ylo=20, yhi=70)
    if rng.random() < 0.5:
        os.environ["VSYN_RUN_LEN"] = str(int(rng.integers(2, 12)))
    else:
        os.environ.pop("VSYN_RUN_LEN", None)
    want = ob.OracleSynth(spec, streams).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    gpu = binding.Synth(spec, max_streams=streams)
    got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert got["rc"] == want["rc"] == 0, (seed, got["rc"], got["flags"], want["rc"])
    assert np.array_equal(got["emit_len"], want["emit_len"]), seed
    scale = max(1.0, float(np.abs(want["pcm"]).max()))
    err = float(np.abs(got["pcm"] - want["pcm"]).max()) / scale
    assert err < 1e-5, (seed, err, C, bs0, bs1)
    worst = max(worst, err)
    # the first stream again, cut into submits
    n_of = np.where(b["packets"]["mode"][:npk] == 1, bs1, bs0)
    off = np.concatenate([[0], np.cumsum(C * n_of // 2)])
    cuts = sorted(set([0, npk] + [int(v) for v in rng.integers(1, npk, 4)]))
    g2 = binding.Synth(spec, max_streams=2)
    parts = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        seg = b["segments"][:1].copy()
        seg["stream"], seg["first_packet"], seg["num_packets"], seg["flags"], seg["residue_off"] = 1, 0, e - a, 1 if a == 0 else 0, 0
        r = g2.submit_host(b["packets"][a:e], seg, b["ys"][a:e], b["residue"][off[a]:off[e]], b["plane_stride"])
        assert r["rc"] == 0, (seed, a, e, r["flags"])
        assert np.array_equal(r["emit_len"], want["emit_len"][a:e]), (seed, a, e)
        parts.append(r["pcm"][0][:, :int(r["emit_len"].sum())])
    cat = np.concatenate(parts, axis=1)
    tot = cat.shape[1]
    e2 = float(np.abs(cat - want["pcm"][0][:, :tot]).max()) / scale
    assert e2 < 1e-5, (seed, "cut", e2, C, bs0, bs1, cuts)
    print("seed %d: C=%d %d/%d paths=%d posts=%d/%d npk=%d err %.2g cut %.2g" % (seed, C, bs0, bs1, gpu.fused_paths, len(floors[0][1]), len(floors[1][1]), npk, err, e2), flush=True)
print("all ok, worst relative error %.3g" % worst)
