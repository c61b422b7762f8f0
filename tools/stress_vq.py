"""tools/stress_vq.py [first_seed] [count] — one-off stress of the residue VQ kernel on the GPU box: random residue setups (format
0 / 1 / 2, partition sizes that are and are not multiples of 8, vector lengths 1..16 incl. non powers of two, 1-2 submaps, begin /
end anywhere, 2..14 classes, random cascades) with random well-formed entry streams over long and short blocks: device == oracle
bit for bit (same helper as tests/test_gpu_vq.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from parseoggvorbis_amd.binding import Synth, VqSpec, VsynError  # noqa: E402
from tests.test_gpu_vq import _random_vq_batch  # noqa: E402
from tests.workloads import fixture_like_spec  # noqa: E402


def random_spec(rng, channels=2, bs1=2048):
    psize = int(rng.choice([8, 12, 16, 24, 32, 48]))
    dims_ok = [d for d in (1, 2, 3, 4, 6, 8, 12, 16) if psize % d == 0]
    nb = int(rng.integers(3, 8))
    books = []
    for _ in range(nb):
        d = int(rng.choice(dims_ok))
        n = int(rng.integers(2, 300))
        amp = int(rng.integers(1, 6))
        books.append((d, n, rng.integers(-amp, amp + 1, (n, d)).astype(np.float32).ravel()))

    def cascade(nclass, density):
        c = np.full((nclass, 8), -1, np.int16)
        for k in range(nclass):
            for ps in range(8):
                if rng.random() < density:
                    c[k, ps] = int(rng.integers(0, nb))
        return c.ravel()

    def residue():
        t = int(rng.integers(0, 3))
        n2 = bs1 // 2
        ln = n2 * (channels if t == 2 else 1)
        begin = int(rng.integers(0, ln // 3))
        end = int(rng.integers(begin, ln + 200))
        nclass = int(rng.integers(2, 15))
        ps = psize if rng.random() < 0.7 else int(rng.choice([8, 16, 32]))
        if any(ps % b[0] for b in books):
            ps = psize
        return dict(type=t, begin=begin, end=end, partition_size=ps, num_classifications=nclass, classwords=int(rng.integers(1, 4)),
                    books=cascade(nclass, float(rng.uniform(0.1, 0.5))))

    if rng.random() < 0.3:
        res = [residue(), residue()]
        mux = [i % 2 for i in range(channels)]
        maps = [(mux, [0, 1]), (mux, [1, 0])]
    else:
        res = [residue()]
        maps = [([0] * channels, [0]), ([0] * channels, [0])]
    return VqSpec(books, res, maps)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
spec = fixture_like_spec(2)
done = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    vqs = random_spec(rng)
    syn = Synth(spec, max_streams=4)
    try:
        syn.attach_vq(vqs)
    except VsynError as e:  # a setup outside the stage's limits (refused with a reason): fine, next
        print("seed", seed, "refused:", str(e)[:80], flush=True)
        continue
    pk, seg, vqp, cls, ent, want = _random_vq_batch(spec, vqs, 4, 10, [1, 0, 1, 1, 0], seed=seed)
    ys = np.zeros((len(pk), 2, syn.ys_stride), np.uint16)
    out = syn.submit_host_vq(pk, seg, ys, vqp, cls, ent, want.size, 10 * spec.blocksize1 // 2)
    assert out["rc"] == 0, (seed, out)
    assert np.array_equal(out["residue"].view(np.uint32), want.view(np.uint32)), seed
    done += 1
    print("seed", seed, "ok: type", [r["type"] for r in vqs.residues], "psize", [r["partition_size"] for r in vqs.residues],
          "entries/packet up to", int(vqp["num_entries"].max()), flush=True)
print("all ok:", done, "setups")
