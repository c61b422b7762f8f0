"""GPU parity tests proper: the HIP path, called through the C-ABI, against (a) the committed dumps of the
reference decoder and (b) the CPU oracle on seeded synthetic batches.
Bars: integer/index work (emit_len, unwrapped floor posts) exact; inverse coupling + floor product
("after_envelope") bit-exact; IMDCT / PCM within 1e-5 absolute at |pcm| <= 1 (compare-debug-out.py:90,
BASELINE.json north_star)."""
import numpy as np
import pytest

from oracle import oracle_binding as ob
from parseoggvorbis_amd import binding
from parseoggvorbis_amd.binding import SetupSpec
from tests.workloads import fixture_like_spec, load_golden, synth_batch

pytestmark = pytest.mark.gpu
TOL = 1e-5
PATHS = [0, binding.VSYN_SUBMIT_PRE_KERNELS, binding.VSYN_SUBMIT_STAGED]  # preparation kernel (default), chained pre-kernels, staged kernels
HIDDEN_PRE = binding.VSYN_SUBMIT_INPUTS_READY | binding.VSYN_SUBMIT_PRE_KERNELS  # chained pre-kernels on the internal stream


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def check(got, want, scale_tol=True):
    assert got["rc"] == want["rc"] == 0, (got["rc"], got["flags"], want["rc"])
    assert np.array_equal(got["emit_len"], want["emit_len"])
    scale = max(1.0, float(np.abs(want["pcm"]).max())) if scale_tol else 1.0
    err = float(np.abs(got["pcm"] - want["pcm"]).max())
    assert err < TOL * scale, (err, scale)
    return err


@pytest.mark.parametrize("flags", PATHS)
@pytest.mark.parametrize("name", ["test.stereo44khz", "test.mono44khz"])
def test_reference_fixture_dumps(name, flags):
    """Inputs = the reference's own hooks on its .ogg fixtures; outputs vs its 'pcm' hook (max |pcm| 0.58)."""
    spec, b, z = load_golden(name)
    gpu = binding.Synth(spec, max_streams=1)
    total = b["pcm"].shape[1]
    r = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], total + 8, flags=flags)
    assert r["rc"] == 0 and r["flags"] == 0
    assert np.array_equal(r["emit_len"], b["emit_len"])
    assert int(r["emit_len"].sum()) == total
    assert np.abs(r["pcm"][0][:, :total] - b["pcm"]).max() < TOL
    assert not r["pcm"][0][:, total:].any()


@pytest.mark.parametrize("name", ["test.stereo44khz", "test.mono44khz"])
def test_reference_fixture_taps(name):
    spec, b, z = load_golden(name)
    gpu = binding.Synth(spec, max_streams=1)
    total = b["pcm"].shape[1]
    r = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], total, want_taps=True)
    assert r["rc"] == 0
    C = spec.channels
    n_of = np.where(z["mode"] == 1, spec.blocksize1, spec.blocksize0)
    off = np.concatenate([[0], np.cumsum(C * n_of // 2)])
    for k in z["tap_packets"]:
        n = int(n_of[k])
        for c in range(C):
            key = "p%d_c%d_" % (k, c)
            env = r["taps"]["after_envelope"][off[k] + c * n // 2: off[k] + (c + 1) * n // 2]
            assert np.array_equal(bits(env), bits(z[key + "env"])), key  # exact: adds/subs and one multiply
            md = r["taps"]["pcm_after_mdct"][2 * off[k] + c * n: 2 * off[k] + (c + 1) * n]
            assert np.abs(md - z[key + "mdct"]).max() < TOL, key
            if key + "final_ys" in z.files:
                mult, xs = spec.floors[int(z["mode"][k])]
                row = r["taps"]["floor_final"].reshape(-1, C, gpu.ys_stride)[k, c, :len(xs)]
                assert np.array_equal(row & 0x7FFF, z[key + "final_ys"] * mult)
                assert np.array_equal(row >> 15, z[key + "flag"])
                cv = r["taps"]["floor_curve"][off[k] + c * n // 2: off[k] + (c + 1) * n // 2]
                assert np.array_equal(cv, z[key + "floor"][:n // 2]), key  # "floor1 floor" (SURVEY 8 f-4 feature tap)
                tail = int(row[1] & 0x7FFF) if row[1] >> 15 else int(cv[-1])  # flat from the last flagged post on (hpp:583-584)
                assert (z[key + "floor"][n // 2:] == tail).all()


SHAPES = [  # (channels, bs0, bs1, pattern, streams, packets)
    (2, 256, 2048, "mixed", 5, 40),
    (2, 256, 2048, "long", 3, 37),
    (1, 256, 2048, "short", 2, 50),
    (1, 64, 64, "long", 2, 9),
    (2, 64, 8192, "mixed", 2, 14),
    (3, 128, 1024, "mixed", 3, 21),
    (2, 512, 512, "short", 2, 8),
    (2, 2048, 4096, "mixed", 2, 12),
    (2, 128, 1024, "mixed", 3, 40),    # the size-generic fused kernel: 2 long blocks / 8 short blocks per pass
    (1, 128, 1024, "mixed", 2, 33),
    (2, 512, 1024, "mixed", 2, 30),
    (2, 1024, 1024, "long", 2, 20),
    (2, 256, 256, "short", 3, 50),
    (2, 64, 256, "mixed", 2, 61),
    (1, 64, 128, "mixed", 2, 47),
    (2, 2048, 2048, "long", 2, 12),
    (2, 128, 2048, "mixed", 2, 45),
    (2, 1024, 2048, "mixed", 2, 25),
    (2, 512, 4096, "mixed", 2, 35),    # blocks above 2048: register sets (UBig)
    (1, 256, 4096, "mixed", 2, 30),
    (2, 4096, 4096, "long", 2, 10),
    (2, 4096, 4096, "mixed", 2, 14),   # blocksize0 == blocksize1 above 2048: every block on the register-set path, whatever its mode
    (1, 8192, 8192, "mixed", 2, 9),
    (2, 1024, 8192, "mixed", 2, 28),
    (1, 64, 8192, "mixed", 1, 40),
    (2, 256, 2048, "mixed", 2, 2300),  # segments of several thousand packets: the layout kernel's bursts
    (1, 256, 2048, "mixed", 1, 16500),  # one very long segment: the layout kernel's 1024-thread block
]


@pytest.mark.parametrize("flags", PATHS)
@pytest.mark.parametrize("C,bs0,bs1,pattern,streams,npk", SHAPES)
def test_synthetic_vs_oracle(C, bs0, bs1, pattern, streams, npk, flags):
    spec = fixture_like_spec(C, bs0, bs1)
    b = synth_batch(spec, streams, npk, pattern, seed=bs0 + bs1 + C, unused_frac=0.1, granule_last=True)
    want = ob.OracleSynth(spec, streams).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    got = binding.Synth(spec, max_streams=streams).submit_host(b["packets"], b["segments"], b["ys"], b["residue"],
                                                             b["plane_stride"], flags=flags)
    check(got, want)


def test_many_channels_chained_couplings():
    """5.1-style mapping: three coupling steps sharing channels (reverse-order application matters)."""
    C = 6
    xs_s = [0, 64, 8, 32, 16, 48]
    xs_l = [0, 512] + [int(v) for v in np.random.default_rng(1).permutation(np.arange(1, 512))[:40]]
    coup = [(0, 1), (0, 2), (3, 4)]
    spec = SetupSpec(C, 128, 1024, [(1, xs_s), (3, xs_l)], [(coup, [0] * C), (coup, [1] * C)], [(0, 0), (1, 1)])
    b = synth_batch(spec, 2, 16, "mixed", seed=11, unused_frac=0.3)
    want = ob.OracleSynth(spec, 2).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps=True)
    got = binding.Synth(spec, max_streams=2).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps=True)
    check(got, want)
    assert np.array_equal(bits(got["taps"]["after_envelope"]), bits(want["taps"]["after_envelope"]))
    assert np.array_equal(got["taps"]["floor_final"], want["taps"]["floor_final"])


@pytest.mark.parametrize("flags", PATHS)
def test_streaming_across_submits(flags):
    """Overlap carry between batches: cutting a stream into several submits == one submit."""
    spec = fixture_like_spec(2)
    b = synth_batch(spec, 1, 60, "mixed", seed=3)
    one = binding.Synth(spec, max_streams=2).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=flags)
    total = int(one["emit_len"].sum())
    n_of = np.where(b["packets"]["mode"] == 1, spec.blocksize1, spec.blocksize0)
    off = np.concatenate([[0], np.cumsum(2 * n_of // 2)])
    gpu = binding.Synth(spec, max_streams=2)
    cuts = [0, 1, 2, 13, 14, 40, 60]
    parts = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        seg = b["segments"].copy()
        seg["stream"], seg["first_packet"], seg["num_packets"], seg["flags"] = 1, 0, e - a, 1 if a == 0 else 0
        r = gpu.submit_host(b["packets"][a:e], seg, b["ys"][a:e], b["residue"][off[a]:off[e]], b["plane_stride"], flags=flags)
        assert r["rc"] == 0
        assert np.array_equal(r["emit_len"], one["emit_len"][a:e])
        parts.append(r["pcm"][0][:, :int(r["emit_len"].sum())])
    got = np.concatenate(parts, axis=1)
    if flags:  # staged kernels everywhere: same arithmetic, same bits
        assert np.array_equal(got, one["pcm"][0][:, :total])
    else:      # fused kernel: the carry of a cut is the same rounded product the uncut run keeps in registers
        assert np.abs(got - one["pcm"][0][:, :total]).max() < TOL


@pytest.mark.parametrize("seed,run_len", [(1, 4), (2, 5), (3, 7), (4, 0)])
def test_random_block_patterns_runs_and_cuts(seed, run_len, monkeypatch):
    """The fused kernel's mixed-block path: random bursts of short blocks, short runs (so that run boundaries and their
    one-packet halos fall on every kind of transition), then the same streams cut into several submits at random places
    (carry-in long->long, long->short, short->short, short->long). One submit == oracle; cut == uncut, bit for bit."""
    rng = np.random.default_rng(100 + seed)
    npk = 90
    flags = np.ones(npk, np.uint8)
    q = 0
    while q < npk:  # alternating stretches: long 1..6, short 1..9
        q += int(rng.integers(1, 7))
        k = int(rng.integers(1, 10))
        flags[q:q + k] = 0
        q += k
    if run_len:
        monkeypatch.setenv("VSYN_RUN_LEN", str(run_len))
    spec = fixture_like_spec(2)
    b = synth_batch(spec, 3, npk, flags, seed=seed, unused_frac=0.15, granule_last=True)
    want = ob.OracleSynth(spec, 3).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    one = binding.Synth(spec, max_streams=3).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    check(one, want)
    # stream 0 again, cut at random places (every kind of size pair occurs at some cut)
    n_of = np.where(b["packets"]["mode"][:npk] == 1, spec.blocksize1, spec.blocksize0)
    off = np.concatenate([[0], np.cumsum(2 * n_of // 2)])
    first_of = {}
    for c in range(npk - 1, 0, -1):
        first_of[(int(flags[c - 1]), int(flags[c]))] = c
    assert len(first_of) == 4
    cuts = sorted(set([0, npk] + [int(c) for c in rng.integers(1, npk, 12)] + list(first_of.values())))
    gpu = binding.Synth(spec, max_streams=2)
    parts = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        seg = b["segments"][:1].copy()
        seg["stream"], seg["first_packet"], seg["num_packets"], seg["flags"], seg["residue_off"] = 1, 0, e - a, 1 if a == 0 else 0, 0
        r = gpu.submit_host(b["packets"][a:e], seg, b["ys"][a:e], b["residue"][off[a]:off[e]], b["plane_stride"])
        assert r["rc"] == 0
        assert np.array_equal(r["emit_len"], one["emit_len"][a:e])
        parts.append(r["pcm"][0][:, :int(r["emit_len"].sum())])
    got = np.concatenate(parts, axis=1)
    total = int(one["emit_len"][:npk].sum())
    assert np.array_equal(bits(got), bits(one["pcm"][0][:, :total]))


def test_empty_and_ragged_batches():
    spec = fixture_like_spec(2)
    gpu = binding.Synth(spec, max_streams=4)
    orc = ob.OracleSynth(spec, 4)
    # ragged: segments of very different lengths, one of a single packet, one empty
    bs = [synth_batch(spec, 1, k, "mixed", seed=k) for k in (1, 2, 31)]
    pk = np.concatenate([x["packets"] for x in bs])
    ys = np.concatenate([x["ys"] for x in bs])
    res = np.concatenate([x["residue"] for x in bs])
    seg = np.zeros(4, binding.SEGMENT_DTYPE)
    first, roff = 0, 0
    for i, x in enumerate(bs):
        seg[i] = (i, first, len(x["packets"]), 1, roff)
        first += len(x["packets"])
        roff += x["residue"].size
    seg[3] = (3, 0, 0, 1, 0)  # empty segment
    plane = bs[2]["plane_stride"]
    for flags in PATHS:
        gpu.reset()
        check(gpu.submit_host(pk, seg, ys, res, plane, flags=flags), orc.submit_host(pk, seg, ys, res, plane))
    # empty batch is a no-op
    r = gpu.submit_host(pk[:0], seg[:0], ys[:0], res[:0], 16)
    assert r["rc"] == 0


def test_error_flags_match_oracle():
    spec = fixture_like_spec(2)
    b = synth_batch(spec, 2, 6, "mixed", seed=5)
    gpu = binding.Synth(spec, max_streams=2)
    orc = ob.OracleSynth(spec, 2)
    bad = b["packets"].copy()
    bad["granule"][4] = 10 ** 9
    for flags in PATHS:
        r = gpu.submit_host(bad, b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=flags)
        w = orc.submit_host(bad, b["segments"], b["ys"], b["residue"], b["plane_stride"])
        assert r["rc"] == w["rc"] == 4 and r["flags"] & 4 and r["first_bad"] == w["first_bad"] == 4
    ys = b["ys"].copy()
    ys[7, 0, 2:] = 255
    pk = b["packets"].copy()
    pk["floor_used"][7] |= 1
    for flags in PATHS:
        r = gpu.submit_host(pk, b["segments"], ys, b["residue"], b["plane_stride"], flags=flags)
        w = orc.submit_host(pk, b["segments"], ys, b["residue"], b["plane_stride"])
        assert r["rc"] == w["rc"] == 4 and (r["flags"] & 3) and r["first_bad"] == w["first_bad"] == 7
    r = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], 100)
    assert r["flags"] & 8
    badmode = b["packets"].copy()
    badmode["mode"][3] = 9
    r = gpu.submit_host(badmode, b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert r["flags"] & 16 and r["first_bad"] == 3
    # after errors the handle is still usable
    gpu.reset()
    check(gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"]),
          orc.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"]))


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_imdct_only(n):
    """BASELINE config 2 shape (n=256: 4096 mono packets, sigma=0.05) and every other blocksize."""
    import torch
    count = 4096 if n == 256 else 257
    rng = np.random.default_rng(1234)
    x = (rng.standard_normal((count, n // 2)) * 0.05 * np.sqrt(128.0 / max(128, n // 2))).astype(np.float32)  # |out| <~ 1
    spec = fixture_like_spec(1, min(n, 256) if n != 64 else 64, max(n, 256) if n != 64 else 64)
    if n not in (spec.blocksize0, spec.blocksize1):
        spec = fixture_like_spec(1, n, n)
    gpu = binding.Synth(spec, max_streams=1)
    din = torch.from_numpy(x).cuda()
    dout = torch.zeros((count, n), dtype=torch.float32, device="cuda")
    gpu.imdct_device(n, count, din.data_ptr(), dout.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dout.cpu().numpy()
    want = ob.imdct(n, x)
    peak = float(np.abs(want).max())
    assert peak < 3.0
    assert np.abs(got - want).max() < TOL * max(1.0, peak)
    # analytic oracle on a few rows (double-precision closed form)
    cf = np.empty(n, np.float64)
    for r in (0, count - 1):
        ob.oracle().orc_imdct_closed_form(n, ob.p(x[r]), ob.p(cf))
        assert np.abs(got[r] - cf).max() < TOL * max(1.0, peak)


@pytest.mark.parametrize("flags", [binding.VSYN_SUBMIT_INPUTS_READY, binding.VSYN_SUBMIT_INPUTS_READY | binding.VSYN_SUBMIT_PRE_KERNELS])
def test_device_resident_back_to_back_submits_overlap_safely(flags):
    """vsyn_submit_device with VSYN_SUBMIT_INPUTS_READY, back to back without a sync in between, each == oracle: with the preparation
    kernel on the caller's stream (the default), and with the chained pre-kernels hidden beside the previous submit's synthesis kernel on
    the internal stream, on double-buffered workspaces (+ VSYN_SUBMIT_PRE_KERNELS: the default of rounds 1-3)."""
    import torch
    spec = fixture_like_spec(2)
    S, ppk = 6, 40
    gpu = binding.Synth(spec, max_streams=S)
    orc = ob.OracleSynth(spec, S)
    stream = torch.cuda.current_stream().cuda_stream
    keep, outs = [], []
    for i in range(5):
        b = synth_batch(spec, S, ppk, "long" if i % 2 == 0 else "mixed", seed=100 + i)
        d = dict(pk=torch.from_numpy(b["packets"].view(np.uint8)).cuda(), seg=torch.from_numpy(b["segments"].view(np.uint8)).cuda(),
                 ys=torch.from_numpy(b["ys"].astype(np.int16)).cuda(), res=torch.from_numpy(b["residue"]).cuda(),
                 pcm=torch.zeros((S, 2, b["plane_stride"]), device="cuda"), emit=torch.zeros(S * ppk, dtype=torch.int32, device="cuda"))
        keep.append(d)
        outs.append(b)
    torch.cuda.synchronize()
    for d, b in zip(keep, outs):
        gpu.submit_device(S * ppk, d["pk"].data_ptr(), S, d["seg"].data_ptr(), ppk, d["ys"].data_ptr(), d["res"].data_ptr(),
                          d["pcm"].data_ptr(), b["plane_stride"], d["emit"].data_ptr(), None, flags, stream)
    fl, bad = gpu.sync_status(stream)
    assert fl == 0, (fl, bad)
    for d, b in zip(keep, outs):
        want = orc.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
        assert np.array_equal(d["emit"].cpu().numpy().astype(np.uint32), want["emit_len"])
        assert np.abs(d["pcm"].cpu().numpy() - want["pcm"]).max() < TOL


@pytest.mark.parametrize("flags", PATHS)
def test_garbage_descriptors_and_posts(flags):
    """Untrusted input: random mode numbers / window flags / granules / floor-used masks and random coded posts. Whatever the
    verdict (the oracle gives the same one), nothing is read or written outside the batch and the handle is exact again on the
    next clean batch."""
    spec = fixture_like_spec(2)
    b = synth_batch(spec, 3, 40, "mixed", seed=8)
    gpu = binding.Synth(spec, max_streams=3)
    orc = ob.OracleSynth(spec, 3)
    rng = np.random.default_rng(17)
    for trial in range(8):
        pk = b["packets"].copy()
        ys = b["ys"].copy()
        hit = rng.random(len(pk)) < (0.08 if trial % 2 else 0.5)
        raw = pk.view(np.uint8).reshape(len(pk), -1)
        raw[hit] = rng.integers(0, 256, (int(hit.sum()), raw.shape[1]), dtype=np.uint8)
        if trial >= 4:
            yh = rng.random(ys.shape) < 0.1
            ys[yh] = rng.integers(0, 65536, int(yh.sum()))
        # the residue layout follows the (possibly changed) modes: give every packet room for a long block
        seg = b["segments"].copy()
        res = np.zeros(len(pk) * 2 * (spec.blocksize1 // 2) + 64, np.float32)
        res[:b["residue"].size] = b["residue"]
        per = len(pk) // len(seg) * 2 * (spec.blocksize1 // 2)
        seg["residue_off"] = np.arange(len(seg), dtype=np.uint64) * per
        r = gpu.submit_host(pk, seg, ys, res, b["plane_stride"] + 40 * 1024, flags=flags)
        w = orc.submit_host(pk, seg, ys, res, b["plane_stride"] + 40 * 1024)
        assert (r["rc"] == 0) == (w["rc"] == 0), (trial, r["rc"], w["rc"], r["flags"], w["flags"])
        gpu.reset()
        orc = ob.OracleSynth(spec, 3)
        check(gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=flags),
              orc.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"]))
        gpu.reset()
        orc = ob.OracleSynth(spec, 3)


@pytest.mark.parametrize("bs0,bs1", [(256, 2048), (128, 1024)])
def test_absurd_coded_posts_unwrap_alike_on_every_preparation(bs0, bs1):
    """Coded floor values far outside a valid stream's range (up to 65535: |dy| * dx beyond 2^21, where the preparation kernel's float
    form of hpp:533's division is no longer exact and the row is redone with the integer division): `floor_final` of the
    dependency-free preparation kernel == the chained pre-kernels == the staged path, bit for bit, flagged rows included; and equal
    to the oracle's rows wherever the oracle raises nothing."""
    spec = fixture_like_spec(2, bs0, bs1)
    b = synth_batch(spec, 3, 40, "mixed", seed=31)
    rng = np.random.default_rng(5)
    for trial in range(4):
        ys = b["ys"].copy()
        yh = rng.random(ys.shape) < (0.02 if trial < 2 else 0.3)
        ys[yh] = rng.integers(0, 65536 if trial % 2 else 4096, int(yh.sum()))
        rows = []
        for flags in PATHS:
            gpu = binding.Synth(spec, max_streams=3)
            r = gpu.submit_host(b["packets"], b["segments"], ys, b["residue"], b["plane_stride"], flags=flags, want_taps="features")
            rows.append(r["taps"]["floor_final"])
        assert np.array_equal(rows[0], rows[1]) and np.array_equal(rows[0], rows[2]), trial
        w = ob.OracleSynth(spec, 3).submit_host(b["packets"], b["segments"], ys, b["residue"], b["plane_stride"], want_taps=True)
        if w["rc"] == 0:
            assert np.array_equal(rows[0], w["taps"]["floor_final"]), trial


@pytest.mark.parametrize("pattern,C,bs0,bs1", [("long", 2, 256, 2048), ("mixed", 2, 256, 2048), ("mixed", 1, 256, 2048),
                                               ("mixed", 2, 128, 1024), ("long", 2, 128, 1024), ("mixed", 3, 128, 1024),   # size-generic kernel
                                               ("mixed", 2, 512, 4096), ("mixed", 1, 64, 8192), ("mixed", 2, 1024, 2048)])  # ... its register sets; both kernels
def test_feature_taps_from_the_fused_kernel(pattern, C, bs0, bs1):
    """the feature taps alone ("floor1 floor" curve + unwrapped posts) do not leave the fast path, whatever the block
    sizes (returnn_import.py:74-115 extracts them from any file) — the tap variants of the fused kernels write the curve on the way.
    Curve and posts equal the oracle's (integers: exact), PCM as without taps, and the staged kernels produce the very same tap."""
    spec = fixture_like_spec(C, bs0, bs1)
    if C > 2:  # (a coupling list for three channels)
        spec = SetupSpec(C, bs0, bs1, spec.floors, [([(0, 1), (0, 2)], [0] * C), ([(0, 1), (0, 2)], [1] * C)], [(0, 0), (1, 1)])
    b = synth_batch(spec, 4, 45, pattern, seed=31, unused_frac=0.15, granule_last=True)
    want = ob.OracleSynth(spec, 4).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps=True)
    gpu = binding.Synth(spec, max_streams=4)
    assert gpu.fused_paths & 2, gpu.fused_paths  # every run of this setup is taken by a fused kernel
    got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps="features")
    check(got, want)
    assert np.array_equal(got["taps"]["floor_curve"], want["taps"]["floor_curve"])
    assert np.array_equal(got["taps"]["floor_final"], want["taps"]["floor_final"])
    gpu.reset()
    staged = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps=True)
    assert np.array_equal(staged["taps"]["floor_curve"], got["taps"]["floor_curve"])
    gpu.reset()
    plain = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert np.array_equal(bits(plain["pcm"]), bits(got["pcm"]))  # the tap changes nothing else


# ---- the fused path at its limits (hpp:484-492 multipliers/ranges, hpp:521-589 floor decode) -----------------------------------
def _limit_spec(C, posts_long, mult_long, mult_short, coupled, posts_short=9, seed=5):
    """256/2048 setup whose long-block floor has `posts_long` posts at random distinct x (header order shuffled, x[0] = 0 and
    x[1] = n/2 as the format requires) and the given multipliers."""
    rng = np.random.default_rng(seed)

    def xs(n2, posts):
        inner = rng.choice(np.arange(1, n2), posts - 2, replace=False) if posts > 2 else np.zeros(0, np.int64)
        return [0, n2] + [int(v) for v in inner]
    coup = [(0, 1)] if (coupled and C >= 2) else []
    return SetupSpec(C, 256, 2048, [(mult_short, xs(128, posts_short)), (mult_long, xs(1024, posts_long))],
                     [(coup, [0] * C), (coup, [1] * C)], [(0, 0), (1, 1)])


LIMITS = [
    # C, posts_long, mult_long, mult_short, coupled, pattern, fused paths expected (vsyn_fused_paths)
    (2, 64, 2, 4, True, "long", 3),    # the most posts one ballot covers
    (2, 64, 1, 3, True, "mixed", 3),   # ... with multipliers 1 and 3 (ranges 256 and 86)
    (2, 2, 1, 1, True, "mixed", 3),    # the fewest: one segment per block, both floors
    (1, 33, 3, 2, False, "mixed", 3),  # mono, multiplier 3 on the long floor
    (2, 40, 4, 1, False, "mixed", 3),  # stereo WITHOUT a coupling step
    (2, 17, 2, 2, False, "long", 3),
    (2, 65, 2, 4, True, "mixed", 2),   # 65 posts (the spec's maximum): one more than the tuned kernel's one-ballot set-up covers ->
    (2, 65, 1, 1, False, "long", 2),   # every run on the size-generic kernel, which keeps the last sorted post wave-uniform
    (1, 65, 3, 2, False, "mixed", 2),
]


@pytest.mark.parametrize("C,posts,ml,ms,coupled,pattern,paths", LIMITS)
def test_fused_path_limits(C, posts, ml, ms, coupled, pattern, paths):
    """Floors with 2 / 64 / 65 posts, every multiplier, with and without coupling: the run class is asserted (so a silent fall
    back to the staged kernels cannot pass for the fused path), posts and PCM against the oracle, and the fused kernel against
    the staged kernels bit for bit where both exist."""
    posts_short = 2 if posts == 2 else 9
    spec = _limit_spec(C, posts, ml, ms, coupled, posts_short)
    b = synth_batch(spec, 3, 37, pattern, seed=posts * 7 + ml, unused_frac=0.1, granule_last=True, ylo=20, yhi=70)
    gpu = binding.Synth(spec, max_streams=3)
    assert gpu.fused_paths == paths, (gpu.fused_paths, paths)
    want = ob.OracleSynth(spec, 3).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps=True)
    got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps="features")
    check(got, want)
    assert np.array_equal(got["taps"]["floor_final"], want["taps"]["floor_final"])
    assert np.array_equal(got["taps"]["floor_curve"], want["taps"]["floor_curve"])
    gpu.reset()
    plain = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert np.array_equal(bits(plain["pcm"]), bits(got["pcm"]))
    gpu.reset()
    staged = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=binding.VSYN_SUBMIT_STAGED)
    check(staged, want)


@pytest.mark.parametrize("bs0,bs1,posts_short,posts_long", [(256, 2048, 65, 65), (128, 1024, 65, 40), (512, 4096, 30, 65), (1024, 8192, 65, 65)])
def test_65_post_floors_stay_fused(bs0, bs1, posts_short, posts_long):
    """The spec's largest floor (65 posts, hpp:416-471) on short blocks, long blocks and blocks above 2048 samples (the register-set
    path of the size-generic kernel): fused, posts / curve / PCM of the tap variant, the plain kernel and the staged kernels against the
    oracle."""
    rng = np.random.default_rng(bs1 + posts_short)

    def xs(n2, posts):
        return [0, n2] + [int(v) for v in rng.choice(np.arange(1, n2), posts - 2, replace=False)]
    spec = SetupSpec(2, bs0, bs1, [(2, xs(bs0 // 2, posts_short)), (1, xs(bs1 // 2, posts_long))],
                     [([(0, 1)], [0, 0]), ([(0, 1)], [1, 1])], [(0, 0), (1, 1)])
    b = synth_batch(spec, 3, 29, "mixed", seed=posts_long + bs0, unused_frac=0.1, granule_last=True, ylo=20, yhi=70)
    gpu = binding.Synth(spec, max_streams=3)
    assert gpu.fused_paths & 2, gpu.fused_paths
    want = ob.OracleSynth(spec, 3).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps=True)
    got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps="features")
    check(got, want)
    assert np.array_equal(got["taps"]["floor_final"], want["taps"]["floor_final"])
    assert np.array_equal(got["taps"]["floor_curve"], want["taps"]["floor_curve"])
    gpu.reset()
    plain = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert np.array_equal(bits(plain["pcm"]), bits(got["pcm"]))
    gpu.reset()
    staged = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=binding.VSYN_SUBMIT_STAGED)
    check(staged, want)


@pytest.mark.parametrize("pattern", ["long", "mixed"])
def test_absolute_gate_at_unit_peak(pattern):
    """The north-star gate as stated: |pcm - reference| < 1e-5 ABSOLUTE on a batch with |pcm| <= 1 (compare-debug-out.py:90)."""
    spec = fixture_like_spec(2)
    b = synth_batch(spec, 4, 48, pattern, seed=77, ylo=20, yhi=74)
    want = ob.OracleSynth(spec, 4).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    peak = float(np.abs(want["pcm"]).max())
    assert 0.05 < peak <= 1.0, peak
    for flags in PATHS:
        got = binding.Synth(spec, max_streams=4).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=flags)
        err = check(got, want, scale_tol=False)
        assert err < TOL, "max |err| %.3g at peak %.3f" % (err, peak)


def test_submit_flags_may_alternate_between_submits():
    """vsyn_submit_device with a different preparation in every submit — the dependency-free preparation kernel on the caller's stream,
    the chained pre-kernels hidden on the internal stream (VSYN_SUBMIT_INPUTS_READY | VSYN_SUBMIT_PRE_KERNELS), the chained pre-kernels
    on the caller's stream (VSYN_SUBMIT_PRE_KERNELS), the staged kernels — on continuing streams, with work queued on the caller's stream
    in front: consecutive preparations chain through the tagged stream-state records whichever kernel writes them and whichever stream it
    runs on."""
    FLAG_CYCLE = [0, HIDDEN_PRE, binding.VSYN_SUBMIT_PRE_KERNELS, HIDDEN_PRE, binding.VSYN_SUBMIT_INPUTS_READY,
                  binding.VSYN_SUBMIT_STAGED, HIDDEN_PRE, 0]
    import torch
    spec = fixture_like_spec(2)
    S, ppk, parts = 4, 24, 6
    b = synth_batch(spec, S, ppk * parts, "mixed", seed=5)
    orc = ob.OracleSynth(spec, S)
    want = orc.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    gpu = binding.Synth(spec, max_streams=S)
    stream = torch.cuda.current_stream().cuda_stream
    n_of = np.where(b["packets"]["mode"] == 1, spec.blocksize1, spec.blocksize0).reshape(S, ppk * parts)
    keep = []
    busy = torch.zeros(64 << 20, device="cuda")
    for k in range(parts):
        pk = np.concatenate([b["packets"][s * ppk * parts + k * ppk: s * ppk * parts + (k + 1) * ppk] for s in range(S)])
        ys = np.concatenate([b["ys"][s * ppk * parts + k * ppk: s * ppk * parts + (k + 1) * ppk] for s in range(S)])
        seg = b["segments"].copy()
        res_parts, off = [], 0
        for s in range(S):
            base = int(b["segments"][s]["residue_off"]) + int(2 * (n_of[s, :k * ppk] // 2).sum())
            ln = int(2 * (n_of[s, k * ppk:(k + 1) * ppk] // 2).sum())
            res_parts.append(b["residue"][base:base + ln])
            seg[s] = (s, s * ppk, ppk, binding.VSYN_SEG_RESET if k == 0 else 0, off)
            off += ln
        d = dict(pk=torch.from_numpy(pk.view(np.uint8)).cuda(), seg=torch.from_numpy(seg.view(np.uint8)).cuda(),
                 ys=torch.from_numpy(ys.astype(np.int16)).cuda(), res=torch.from_numpy(np.concatenate(res_parts)).cuda(),
                 pcm=torch.zeros((S, 2, ppk * 1024 + 64), device="cuda"), emit=torch.zeros(S * ppk, dtype=torch.int32, device="cuda"))
        keep.append(d)
    torch.cuda.synchronize()
    for k, d in enumerate(keep):
        for _ in range(4):
            busy.add_(1.0)  # delayed work in front of the submit on the caller's stream
        gpu.submit_device(S * ppk, d["pk"].data_ptr(), S, d["seg"].data_ptr(), ppk, d["ys"].data_ptr(), d["res"].data_ptr(),
                          d["pcm"].data_ptr(), ppk * 1024 + 64, d["emit"].data_ptr(), None,
                          FLAG_CYCLE[k % len(FLAG_CYCLE)], stream)
    fl, bad = gpu.sync_status(stream)
    assert fl == 0, (fl, bad)
    emit_want = want["emit_len"].reshape(S, parts, ppk)
    for s in range(S):
        at = 0
        for k, d in enumerate(keep):
            e = d["emit"].cpu().numpy().astype(np.uint32).reshape(S, ppk)[s]
            assert np.array_equal(e, emit_want[s, k])
            n = int(e.sum())
            got = d["pcm"][s, :, :n].cpu().numpy()
            assert np.abs(got - want["pcm"][s][:, at:at + n]).max() < TOL * max(1.0, float(np.abs(want["pcm"]).max())), (s, k)
            at += n


MULTI = [
    # C, bs0, bs1, couplings
    (6, 128, 1024, [(0, 1), (0, 2), (3, 4)]),   # 5.1-style: three steps, channel 0 in two of them (order matters)
    (6, 256, 2048, [(0, 2), (3, 4), (1, 0)]),
    (3, 256, 2048, [(1, 2)]),
    (5, 512, 512, []),
    (4, 64, 256, [(0, 1), (2, 3), (0, 2), (1, 3)]),
    (2, 256, 1024, [(1, 0), (0, 1)]),            # stereo with a two-step chain: not the pairwise swap
    (16, 128, 1024, []),                         # more channels than one workgroup has waves: fused while no mapping couples any
    (13, 256, 2048, []),
    (3, 512, 4096, []),                          # uncoupled channels on the register-set path
]


@pytest.mark.parametrize("C,bs0,bs1,coup", MULTI)
def test_multichannel_and_chained_couplings_stay_fused(C, bs0, bs1, coup):
    """More than two channels / chained coupling steps (hpp:1213-1241, 765-814) on the size-generic fused kernel: the channel waves
    of a run replay the steps in place. Run class asserted; PCM and posts against the oracle, fused == staged within the gate."""
    base = fixture_like_spec(1, bs0, bs1)
    spec = SetupSpec(C, bs0, bs1, base.floors, [(coup, [0] * C), (coup, [1] * C)], [(0, 0), (1, 1)])
    b = synth_batch(spec, 3, 29, "mixed", seed=C * 13 + bs0, unused_frac=0.25, granule_last=True)
    gpu = binding.Synth(spec, max_streams=3)
    assert gpu.fused_paths & 2, gpu.fused_paths
    want = ob.OracleSynth(spec, 3).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    check(got, want)
    gpu.reset()
    staged = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=binding.VSYN_SUBMIT_STAGED)
    check(staged, want)


@pytest.mark.parametrize("C,bs,pattern", [(2, 4096, "mixed"), (2, 8192, "mixed"), (1, 8192, "short"), (3, 4096, "long")])
def test_equal_block_sizes_above_2048_stay_fused(C, bs, pattern):
    """blocksize0 == blocksize1 of 4096 / 8192 (hpp:1294-1298: one code path for every size): blocks of both modes take the
    size-generic kernel's register-set path. Run class asserted; posts, curve and PCM against the oracle; staged kernels beside it."""
    spec = fixture_like_spec(C, bs, bs)
    b = synth_batch(spec, 2, 11, pattern, seed=bs + C, unused_frac=0.15, granule_last=True)
    gpu = binding.Synth(spec, max_streams=2)
    assert gpu.fused_paths & 2, gpu.fused_paths
    want = ob.OracleSynth(spec, 2).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps=True)
    got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], want_taps="features")
    check(got, want)
    assert np.array_equal(got["taps"]["floor_final"], want["taps"]["floor_final"])
    assert np.array_equal(got["taps"]["floor_curve"], want["taps"]["floor_curve"])
    gpu.reset()
    plain = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert np.array_equal(bits(plain["pcm"]), bits(got["pcm"]))
    gpu.reset()
    check(gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=binding.VSYN_SUBMIT_STAGED), want)


@pytest.mark.parametrize("C,bs0,bs1,paths", [(14, 128, 1024, 2), (16, 256, 2048, 2), (17, 128, 1024, 0), (14, 512, 4096, 0)])
def test_many_coupled_channels(C, bs0, bs1, paths):
    """The channel waves of a COUPLED run share one workgroup — 16 waves of the size-generic kernel, 8 where blocks above 2048 need
    register sets: up to that many coupled channels stay fused, one more is the channel layout left to the staged kernels. Asserted
    either way, and equal to the oracle."""
    base = fixture_like_spec(1, bs0, bs1)
    coup = [(0, C - 1), (1, 2)]
    spec = SetupSpec(C, bs0, bs1, base.floors, [(coup, [0] * C), (coup, [1] * C)], [(0, 0), (1, 1)])
    b = synth_batch(spec, 2, 21, "mixed", seed=C, unused_frac=0.2, granule_last=True)
    gpu = binding.Synth(spec, max_streams=2)
    assert gpu.fused_paths == paths, gpu.fused_paths
    want = ob.OracleSynth(spec, 2).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    check(gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"]), want)


MIXED_MAPPINGS = [
    # C, bs0, bs1, couplings of mapping 0 (short-block mode), couplings of mapping 1 (long-block mode)
    (2, 128, 1024, [], [(0, 1)]),                      # the shape of round 2's hang: an uncoupled mapping beside a coupled one
    (2, 256, 2048, [(0, 1)], []),                      # 256/2048 with mappings that disagree: not the tuned kernel's pairwise swap
    (3, 256, 1024, [], [(0, 1), (0, 2)]),
    (4, 128, 512, [(0, 1), (2, 3)], [(2, 3)]),
]


@pytest.mark.parametrize("C,bs0,bs1,coup0,coup1", MIXED_MAPPINGS)
def test_mappings_with_different_coupling_lists_stay_fused(C, bs0, bs1, coup0, coup1):
    """The mappings of one stream carry DIFFERENT coupling lists, one of them empty (hpp:765-814): passes of the uncoupled mapping
    take no part in the channel-group hand-off of the size-generic kernel, whose wait target therefore counts coupled passes only
    (round 2: a counter that counted every pass hung the kernel). Run class asserted; PCM against the oracle, fused == staged."""
    base = fixture_like_spec(1, bs0, bs1)
    spec = SetupSpec(C, bs0, bs1, base.floors, [(coup0, [0] * C), (coup1, [1] * C)], [(0, 0), (1, 1)])
    b = synth_batch(spec, 3, 37, "mixed", seed=C * 7 + bs1, unused_frac=0.2, granule_last=True)
    gpu = binding.Synth(spec, max_streams=3)
    assert gpu.fused_paths & 2, gpu.fused_paths
    want = ob.OracleSynth(spec, 3).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    for flags in (0, binding.VSYN_SUBMIT_PRE_KERNELS, binding.VSYN_SUBMIT_STAGED):
        gpu.reset()
        got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=flags)
        check(got, want)


@pytest.mark.parametrize("streams,npk", [(1, 10007), (2, 9000)])
def test_chunked_layout_scan_twice_on_one_handle(streams, npk):
    """Segments beyond LAYOUT_CHUNK_PACKETS (4096): the layout kernel scans them in chunks chained by a look-back whose records are
    tagged with the submit number and never cleared — so the SAME handle runs the batch twice (the second run meets the first run's
    records). Mixed blocks, page granules inside the stream; emit_len and PCM against the oracle both times."""
    from parseoggvorbis_amd import sharding
    spec = fixture_like_spec(2)
    b = synth_batch(spec, streams, npk, "mixed", seed=npk, granule_last=True)
    for s in range(streams):  # page granules mid-stream (every 9th packet ends a page), consistent with the block sizes
        pk = b["packets"][s * npk:(s + 1) * npk]
        abs_before, emit = sharding.stream_positions(spec, pk)
        for q in range(8, npk - 1, 9):
            pk["granule"][q] = int(abs_before[q] + emit[q])
    want = ob.OracleSynth(spec, streams).submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert want["rc"] == 0
    gpu = binding.Synth(spec, max_streams=streams)
    for rep in range(2):
        gpu.reset()
        got = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
        check(got, want)


@pytest.mark.parametrize("run_len", [3, 130, 300])
@pytest.mark.parametrize("C", [1, 2, 3])
def test_preparation_kernel_geometries(run_len, C, monkeypatch):
    """vsyn_prep_kernel (the dependency-free preparation of submits without VSYN_SUBMIT_INPUTS_READY) across its geometries: runs far
    shorter than a workgroup's 256 rows (many runs per workgroup), runs longer than one pass (a run's packets spread over several
    passes of a workgroup: class, scan carry and block size in front cross the pass boundary), 1-3 channels (rows per packet),
    ragged segments, unused channels, page granules inside the streams. Against the oracle, and bit-identical to the two chained
    pre-kernels (VSYN_SUBMIT_PRE_KERNELS) in emit_len, unwrapped posts and PCM."""
    from parseoggvorbis_amd import sharding
    monkeypatch.setenv("VSYN_RUN_LEN", str(run_len))
    spec = fixture_like_spec(C)
    if C > 2:
        spec = SetupSpec(C, 256, 2048, spec.floors, [([(0, 1)], [0] * C), ([(0, 1)], [1] * C)], [(0, 0), (1, 1)])
    lens = (701, 1, 333, 64)
    bs = [synth_batch(spec, 1, k, "mixed", seed=100 + k, unused_frac=0.2, granule_last=True) for k in lens]
    for x in bs:
        pk = x["packets"]
        abs_before, emit = sharding.stream_positions(spec, pk)
        for q in range(10, len(pk) - 1, 11):
            pk["granule"][q] = int(abs_before[q] + emit[q])
    pk = np.concatenate([x["packets"] for x in bs])
    ys = np.concatenate([x["ys"] for x in bs])
    res = np.concatenate([x["residue"] for x in bs])
    seg = np.zeros(len(bs), binding.SEGMENT_DTYPE)
    first, roff = 0, 0
    for i, x in enumerate(bs):
        seg[i] = (i, first, len(x["packets"]), 1, roff)
        first += len(x["packets"])
        roff += x["residue"].size
    plane = max(x["plane_stride"] for x in bs)
    want = ob.OracleSynth(spec, len(bs)).submit_host(pk, seg, ys, res, plane)
    gpu = binding.Synth(spec, max_streams=len(bs))
    got = gpu.submit_host(pk, seg, ys, res, plane, want_taps="features")
    check(got, want)
    gpu.reset()
    pre = gpu.submit_host(pk, seg, ys, res, plane, want_taps="features", flags=binding.VSYN_SUBMIT_PRE_KERNELS)
    assert np.array_equal(pre["emit_len"], got["emit_len"])
    assert np.array_equal(pre["taps"]["floor_final"], got["taps"]["floor_final"])
    assert np.array_equal(bits(pre["pcm"]), bits(got["pcm"]))
