"""tools/stress_synth_gpu.py <dir> — one-off on the GPU box: every synth_NN.ogg of <dir> (written by oracle/make_synth_ogg.py with
its reference vectors beside it) through the corpus decoder, PCM against the reference's, in both residue modes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_gpu_host_decoder import _run_corpus  # noqa: E402

d = sys.argv[1]
names = sorted(f[:-4] for f in os.listdir(d) if f.startswith("synth_") and f.endswith(".ogg"))
gold = [np.load(os.path.join(d, n + ".npz")) for n in names]
blobs = [open(os.path.join(d, n + ".ogg"), "rb").read() for n in names]
chans = [int(z["channels"]) for z in gold]
worst = 0.0
for vq in ("1", "0"):
    os.environ["PARSEOGGVORBIS_VQ"] = vq
    frames, sums, ok, pcm, stats = _run_corpus(blobs, chans, threads=4, feeders=2, files_per_submit=7, cap=262144)
    for i, z in enumerate(gold):
        want = z["pcm"]
        assert ok[i] and frames[i] == want.shape[1], (names[i], ok[i], frames[i], want.shape)
        rel = float(np.abs(pcm[i][:, :frames[i]] - want).max()) / max(1.0, float(np.abs(want).max()))
        worst = max(worst, rel)
        assert rel <= 4e-6, (names[i], vq, rel)
print("all ok:", len(names), "streams x 2 residue modes; worst error / max(peak, 1) =", worst)
