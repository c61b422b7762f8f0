"""tools/u_debug.py C bs0 bs1 pattern streams npk [seed] — per-packet error map of the fused path against the oracle (GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle_binding as ob
from parseoggvorbis_amd import binding
from tests.workloads import fixture_like_spec, synth_batch

C, bs0, bs1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pattern = sys.argv[4]
if set(pattern) <= set("01"):
    pattern = [int(ch) for ch in pattern]
streams, npk = int(sys.argv[5]), int(sys.argv[6])
seed = int(sys.argv[7]) if len(sys.argv) > 7 else 5
spec = fixture_like_spec(C, bs0, bs1)
import os
b = synth_batch(spec, streams, npk, pattern, seed=seed, unused_frac=float(os.environ.get("UNUSED", "0")), granule_last=bool(int(os.environ.get("GRANULE", "0"))))
want = ob.OracleSynth(spec, streams).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
gpu = binding.Synth(spec, max_streams=streams)
print("fused_paths", gpu.fused_paths)
got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=int(os.environ.get("FLAGS", "0")))
print("rc", got["rc"], got["flags"], "emit equal", np.array_equal(got["emit_len"], want["emit_len"]))
long_mode = [i for i, (bf, _) in enumerate(spec.modes) if bf][0]
for s in range(streams):
    at = 0
    for q in range(npk):
        p = s * npk + q
        e = int(want["emit_len"][p])
        lng = int(b["packets"]["mode"][p] == long_mode)
        if e:
            d = np.abs(got["pcm"][s][:, at:at + e] - want["pcm"][s][:, at:at + e])
            worst = np.unravel_index(np.argmax(d), d.shape)
            flag = "" if d.max() < 1e-5 * max(1, np.abs(want["pcm"]).max()) else "  <<<<"
            if flag or q < 3:
                bad = np.nonzero(d.max(axis=0) > 1e-5)[0]
                rng = (int(bad.min()), int(bad.max()), len(bad)) if len(bad) else None
                print("s%d q%2d %s emit %4d err %.3g at ch%d pos %d bad-range %s%s" % (s, q, "L" if lng else "S", e, d.max(), worst[0], worst[1], rng, flag))
        at += e
