"""tools/u_cutdiff.py seed run_len — where does cut != uncut (bitwise) in test_random_block_patterns_runs_and_cuts?"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
seed, run_len = int(sys.argv[1]), int(sys.argv[2])
if run_len:
    os.environ["VSYN_RUN_LEN"] = str(run_len)
from parseoggvorbis_amd import binding
from tests.workloads import fixture_like_spec, synth_batch
rng = np.random.default_rng(100 + seed)
npk = 90
flags = np.ones(npk, np.uint8)
q = 0
while q < npk:
    q += int(rng.integers(1, 7)); k = int(rng.integers(1, 10)); flags[q:q + k] = 0; q += k
spec = fixture_like_spec(2)
b = synth_batch(spec, 3, npk, flags, seed=seed, unused_frac=0.15, granule_last=True)
one = binding.Synth(spec, max_streams=3).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
n_of = np.where(b["packets"]["mode"][:npk] == 1, spec.blocksize1, spec.blocksize0)
off = np.concatenate([[0], np.cumsum(2 * n_of // 2)])
first_of = {}
for c in range(npk - 1, 0, -1):
    first_of[(int(flags[c - 1]), int(flags[c]))] = c
cuts = sorted(set([0, npk] + [int(c) for c in rng.integers(1, npk, 12)] + list(first_of.values())))
print("flags", "".join(str(int(f)) for f in flags)); print("cuts", cuts)
gpu = binding.Synth(spec, max_streams=2)
parts = []
for a, e in zip(cuts[:-1], cuts[1:]):
    seg = b["segments"][:1].copy()
    seg["stream"], seg["first_packet"], seg["num_packets"], seg["flags"], seg["residue_off"] = 1, 0, e - a, 1 if a == 0 else 0, 0
    r = gpu.submit_host(b["packets"][a:e], seg, b["ys"][a:e], b["residue"][off[a]:off[e]], b["plane_stride"])
    parts.append(r["pcm"][0][:, :int(r["emit_len"].sum())])
got = np.concatenate(parts, axis=1)
total = int(one["emit_len"][:npk].sum())
ref = one["pcm"][0][:, :total]
d = got.view(np.uint32) != ref.view(np.uint32)
starts = np.concatenate([[0], np.cumsum(one["emit_len"][:npk])])
for qq in range(npk):
    a, e = int(starts[qq]), int(starts[qq + 1])
    if e > a and d[:, a:e].any():
        idx = np.nonzero(d[:, a:e].any(axis=0))[0]
        print("packet %d (%s, prev %s) cut-before:%s  differing frames %d..%d (%d of %d) max abs diff %.3g" % (
            qq, "L" if flags[qq] else "S", "L" if qq and flags[qq - 1] else "S", qq in cuts, idx.min(), idx.max(), len(idx), e - a,
            np.abs(got[:, a:e] - ref[:, a:e]).max()))
