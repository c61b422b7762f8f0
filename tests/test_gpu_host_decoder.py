"""GPU: the drop-in decoder end to end (BASELINE config 1 on the GPU box, where compare-debug-out.py and the
reference cannot travel): ours_hip.bin decodes the reference's .ogg fixtures with --debug_out, and its dump is
compared entry by entry with the committed dumps of the reference decoder — same grammar and tolerances as
tests/compare-debug-out.py (ints exact, floats 1e-5; reference lines 90-151, 524-542)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.dump_reader import read_dump, split_packets
from tests.workloads import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "parseoggvorbis_amd", "host")
CLI = os.path.join(HOST, "ours_hip.bin")
CLI_TESTING = os.path.join(HOST, "ours_hip_testing.bin")  # the same CLI over the TESTING build of the library (fault injection)
TOL = 1e-5
INVDB = np.ctypeslib.as_array(__import__("oracle.oracle_binding", fromlist=["x"]).oracle().orc_inverse_db_table(), (256,)).copy()


@pytest.mark.parametrize("batch", ["2048", "7"])
@pytest.mark.parametrize("name", ["test.stereo44khz", "test.mono44khz"])
def test_cli_dump_matches_reference(name, batch, tmp_path):
    spec, b, z = load_golden(name)
    dump = str(tmp_path / "d.bin")
    env = dict(os.environ, PARSEOGGVORBIS_BATCH=batch)  # 7: many flushes, overlap carried between GPU batches
    r = subprocess.run([CLI, "--in", os.path.join(GOLDEN, name + ".ogg"), "--debug_out", dump], capture_output=True,
                       text=True, env=env)
    assert r.returncode == 0, r.stderr
    total = b["pcm"].shape[1]
    assert "got eof. sample count: %d" % total in r.stdout and "\nok\n" in r.stdout
    assert "Ogg total packets count: %d" % (len(b["packets"]) + 3) in r.stdout
    header, entries = read_dump(dump)
    Cn = spec.channels
    assert int(header["decoder-num-channels"][0]) == Cn and int(header["decoder-sample-rate"][0]) == int(z["sample_rate"])
    setup, packets, pcm = split_packets(entries, Cn)
    # setup: (floor1_unpack multiplier, floor1_unpack xs)* finish_setup
    assert [e[0] for e in setup] == ["floor1_unpack multiplier", "floor1_unpack xs"] * len(spec.floors)
    for f, (mult, xs) in enumerate(spec.floors):
        assert int(setup[2 * f][2][0]) == mult and list(setup[2 * f + 1][2]) == xs
    assert len(packets) == len(b["packets"])
    n_of = np.where(z["mode"] == 1, spec.blocksize1, spec.blocksize0)
    off = np.concatenate([[0], np.cumsum(Cn * n_of // 2)])
    pos = 0
    for k, p in enumerate(packets):
        assert p["abs_total_pos"] == pos and p["expected_ending_total_pos"] == int(b["packets"]["granule"][k])
        pos += int(b["emit_len"][k])
        n = int(n_of[k])
        for c in range(Cn):
            assert p["floor_number"][c] == int(z["floor_number"][k, c])
            used = (int(b["packets"]["floor_used"][k]) >> c) & 1
            assert (c in p["ys"]) == bool(used)
            if used:
                posts = len(spec.floors[p["floor_number"][c]][1])
                assert np.array_equal(p["ys"][c], b["ys"][k, c, :posts])
            res = b["residue"][off[k] + c * n // 2: off[k] + (c + 1) * n // 2]
            assert np.array_equal(p["after_residue"][c].view(np.uint32), res.view(np.uint32))
            key = "p%d_c%d_" % (k, c)
            if key + "env" in z.files:
                assert np.array_equal(p["after_envelope"][c].view(np.uint32), z[key + "env"].view(np.uint32))
                assert np.abs(p["pcm_after_mdct"][c] - z[key + "mdct"]).max() < TOL
                if used:
                    assert np.array_equal(p["final_ys"][c], z[key + "final_ys"])
                    assert np.array_equal(p["flag"][c], z[key + "flag"])
                    assert np.array_equal(p["floor"][c], z[key + "floor"])  # "floor1 floor": all n rendered values
                    want_fo = INVDB[z[key + "floor"].astype(np.int64)]             # "floor_outputs" = its inverse-dB image
                    assert np.array_equal(p["floor_outputs"][c].view(np.uint32), want_fo.view(np.uint32))
            assert len(p["pcm_after_mdct"][c]) == n
    for c in range(Cn):
        assert len(pcm[c]) == total
        assert np.abs(pcm[c] - b["pcm"][c]).max() < TOL


def test_c_api_full_read(tmp_path):
    lib = C.CDLL(os.path.join(HOST, "libparseoggvorbis_amd.so"))
    lib.ogg_vorbis_full_read.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
    lib.ogg_vorbis_full_read_from_memory.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p)]
    err = C.c_char_p()
    assert lib.ogg_vorbis_full_read(os.path.join(GOLDEN, "test.mono44khz.ogg").encode(), C.byref(err)) == 0
    data = open(os.path.join(GOLDEN, "test.stereo44khz.ogg"), "rb").read()
    assert lib.ogg_vorbis_full_read_from_memory(data, len(data), C.byref(err)) == 0
    assert lib.ogg_vorbis_full_read_from_memory(data[:5000], 5000, C.byref(err)) == 1  # torn page
    assert b"check failed" in err.value
    assert lib.ogg_vorbis_full_read(b"/nonexistent.ogg", None) == 1  # error_out may be NULL


def _corpus_lib():
    lib = C.CDLL(os.path.join(HOST, "libparseoggvorbis_amd.so"))
    lib.ogg_vorbis_decode_corpus.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int,
                                             C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint8),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                             C.POINTER(C.c_char_p)]
    lib.ogg_vorbis_decode_corpus.restype = C.c_int
    return lib


def _run_corpus(blobs, channels, threads, feeders, files_per_submit, cap=131072):
    lib = _corpus_lib()
    n = len(blobs)
    datas = (C.c_char_p * n)(*blobs)
    lens = (C.c_size_t * n)(*[len(b) for b in blobs])
    frames = (C.c_uint64 * n)()
    sums = (C.c_double * n)()
    ok = (C.c_uint8 * n)()
    pcm = [np.zeros((channels[i], cap), np.float32) for i in range(n)]
    ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in pcm])
    caps = (C.c_uint64 * n)(*([cap] * n))
    stats = (C.c_double * 8)()
    err = C.c_char_p()
    rc = lib.ogg_vorbis_decode_corpus(datas, lens, n, threads, feeders, files_per_submit, 0, frames, sums, ok, ptrs, caps, stats,
                                      C.byref(err))
    assert rc == 0, err.value
    return list(frames), list(sums), list(ok), pcm, list(stats)


@pytest.mark.parametrize("files_per_submit,feeders", [(4, 1), (3, 3), (64, 2)])
def test_corpus_decoder_matches_reference_pcm(files_per_submit, feeders):
    """Many files, several entropy threads, merged GPU submits (one stream slot per file, two setups -> two handles):
    every file's PCM equals the reference decoder's (tests/golden, 1e-5), a corrupt file fails alone."""
    names = ["test.stereo44khz", "test.mono44khz"]
    gold = {n: load_golden(n) for n in names}
    raw = {n: open(os.path.join(GOLDEN, n + ".ogg"), "rb").read() for n in names}
    order = [names[i % 2] for i in range(13)] + [names[0]] * 4
    blobs = [raw[n] for n in order]
    bad = bytearray(raw[names[1]])
    bad[4000] ^= 0x55  # page CRC
    blobs.insert(5, bytes(bad))
    order.insert(5, names[1])
    chans = [gold[n][0].channels for n in order]
    frames, sums, ok, pcm, stats = _run_corpus(blobs, chans, threads=4, feeders=feeders, files_per_submit=files_per_submit)
    for i, n in enumerate(order):
        want = gold[n][1]["pcm"]
        if i == 5:
            assert ok[i] == 0
            # what the reference's gotPcmData would have delivered before its CRC check fires is still a prefix of the truth
            assert frames[i] < want.shape[1]
            if frames[i]:
                assert np.abs(pcm[i][:, :frames[i]] - want[:, :frames[i]]).max() < TOL
            continue
        assert ok[i] == 1 and frames[i] == want.shape[1]
        assert np.abs(pcm[i][:, :frames[i]] - want).max() < TOL
        assert abs(sums[i] - np.abs(pcm[i][:, :frames[i]].astype(np.float64)).sum()) < 1e-6 * max(1.0, sums[i])
    assert stats[5] >= 2  # at least one submit per setup
    # replicas of one file are bit-identical whatever slot / batch they landed in
    first = {n: order.index(n) for n in names}
    for i, n in enumerate(order):
        if i != 5:
            assert np.array_equal(pcm[i], pcm[first[n]])


def test_corpus_survives_damaged_files(tmp_path):
    """Damaged but partly parsable files (bit flips / overwritten bytes with the page CRCs re-computed, truncation) through the
    whole pipeline, GPU included: good files still decode to the reference PCM, bad ones fail alone, nothing hangs or faults."""
    import json
    import subprocess
    from tests.test_host_decoder import CORPUS_CLI  # noqa: F401
    rng = np.random.default_rng(77)
    tab = []
    for i in range(256):
        r = i << 24
        for _ in range(8):
            r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
        tab.append(r)

    def fix_crcs(b):
        o = 0
        while o + 27 <= len(b) and b[o:o + 4] == b"OggS":
            ns = b[o + 26]
            if o + 27 + ns > len(b):
                break
            ln = 27 + ns + sum(b[o + 27:o + 27 + ns])
            if o + ln > len(b):
                break
            b[o + 22:o + 26] = b"\0\0\0\0"
            c = 0
            for x in b[o:o + ln]:
                c = ((c << 8) & 0xFFFFFFFF) ^ tab[((c >> 24) & 0xFF) ^ x]
            b[o + 22:o + 26] = c.to_bytes(4, "little")
            o += ln

    names = ["test.stereo44khz", "test.mono44khz"]
    base = [open(os.path.join(GOLDEN, n + ".ogg"), "rb").read() for n in names]
    paths = [os.path.join(GOLDEN, names[0] + ".ogg")]  # file 0 is intact
    for k in range(40):
        b = bytearray(base[k % 2])
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(4500, len(b)))  # past the headers: audio packets
            if rng.random() < 0.7:
                b[pos] ^= 1 << int(rng.integers(0, 8))
            else:
                b[pos] = int(rng.integers(0, 256))
        if k % 5 == 0 and len(b) > 9000:
            del b[int(rng.integers(6000, len(b))):]
        fix_crcs(b)
        p = tmp_path / ("d%02d.ogg" % k)
        p.write_bytes(bytes(b))
        paths.append(str(p))
    r = subprocess.run(["timeout", "-k", "10", "120", CORPUS_CLI, "--threads", "4", "--feeders", "2", "--files_per_submit", "8"] + paths,
                       capture_output=True, text=True)
    assert r.returncode in (0, 1), (r.returncode, r.stderr[-800:])
    out = json.loads(r.stdout)
    assert out["files"] == 41
    assert out["first_file"]["frames"] == load_golden(names[0])[1]["pcm"].shape[1]  # the intact file is unaffected by its neighbours


def test_bench_config5_line():
    """bench.py --workload config5 (real files end to end) runs and prints one JSON line; it asserts itself
    that every replica has the granule-derived frame count and the same checksum."""
    import json
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "config5", "--files-per-gpu", "96", "--steps", "1",
                          "--warmup", "0", "--host-threads", "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["config"]["packets_per_gpu"] == 96 * 94 and j["replicas_bit_identical"] and j["value"] > 0


def test_bench_config5_full_size():
    """BASELINE config 5 at its FULL per-GPU size — 10 640 replicas of the stereo fixture = 1 000 160 audio packets — through
    bench.py exactly as the driver would run it on one GPU (one untimed pass, one timed): every replica must carry the reference
    decoder's granule-derived frame count and the same checksum (bench.py asserts both), and the whole pass has to be a real-time
    factor in the tens of thousands (it takes a quarter of a second on the box's 16 cores; the bound is generous)."""
    import json
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "config5", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["config"]["packets_per_gpu"] == 10640 * 94 == 1000160 and j["replicas_bit_identical"]
    assert j["frames_per_file"] == 91136          # the granule-derived total of the fixture
    assert j["value"] > 2.5e5 and j["realtime_factor"] > 5000


@pytest.mark.parametrize("vq", ["1", "0"])
def test_synthetic_streams_pcm_matches_reference(vq, monkeypatch):
    """The 16 synthetic streams of tests/golden (oracle/make_synth_ogg.py: setups the real fixtures do not have, golden PCM from
    the reference decoder) end to end through the corpus decoder — host entropy half, device VQ stage or float residue,
    synthesis kernels (most of these shapes take the staged kernels: 3 channels, other block sizes, chained couplings).
    Frame counts equal the reference's; PCM within 4e-6 of the stream's peak (the reference's own harness allows 1e-5 at
    |pcm| <= 1, compare-debug-out.py:90; these streams peak between 0.2 and 160)."""
    monkeypatch.setenv("PARSEOGGVORBIS_VQ", vq)
    names = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("synth_") and f.endswith(".ogg"))
    assert len(names) >= 16
    gold = [np.load(os.path.join(GOLDEN, n + ".npz")) for n in names]
    blobs = [open(os.path.join(GOLDEN, n + ".ogg"), "rb").read() for n in names]
    chans = [int(z["channels"]) for z in gold]
    frames, sums, ok, pcm, stats = _run_corpus(blobs, chans, threads=3, feeders=2, files_per_submit=5, cap=16384)
    for i, z in enumerate(gold):
        want = z["pcm"]
        assert ok[i], names[i]
        assert frames[i] == want.shape[1], (names[i], frames[i], want.shape)
        peak = float(np.abs(want).max())
        err = float(np.abs(pcm[i][:, :frames[i]] - want).max())
        assert err <= 4e-6 * max(peak, 1.0), (names[i], err, peak)


def _synth_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("synth_") and f.endswith(".ogg"))


@pytest.mark.parametrize("name", _synth_names())
def test_cli_dump_matches_reference_on_synthetic_streams(name, tmp_path):
    """ours_hip.bin --debug_out on the synthetic streams: the SAME hook stream as the reference decoder's — every entry name,
    channel and length in the same order (what tests/compare-debug-out.py of the reference walks, 154-198 / 380-401), integer
    hooks equal (CRC over the values), float hooks equal to 1e-5 of their magnitude sums, PCM as in the corpus test."""
    import zlib
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    dump = str(tmp_path / "d.bin")
    r = subprocess.run([CLI, "--in", os.path.join(GOLDEN, name + ".ogg"), "--debug_out", dump], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    header, entries = read_dump(dump)
    Cn = int(z["channels"])
    assert int(header["decoder-num-channels"][0]) == Cn
    hk = [(nm, ch, v) for nm, ch, v, _ in entries if nm != "pcm"]
    assert [nm.encode() for nm, _, _ in hk] == list(z["hook_names"])
    assert [ch for _, ch, _ in hk] == list(z["hook_ch"])
    assert [len(v) for _, _, v in hk] == list(z["hook_len"])
    for i, (nm, ch, v) in enumerate(hk):
        if z["hook_float"][i]:
            assert v.dtype.kind == "f", (i, nm)
            a = v.astype(np.float64)
            tol = 1e-5 * (float(z["hook_abs"][i]) + 1e-30) + 1e-12
            assert abs(float(a.sum()) - float(z["hook_sum"][i])) <= tol and abs(float(np.abs(a).sum()) - float(z["hook_abs"][i])) <= tol, (i, nm, ch)
        else:
            assert zlib.crc32(v.astype(np.int64).tobytes()) == int(z["hook_crc"][i]), (i, nm, ch)
    pcm = [np.concatenate([v for nm, ch, v, _ in entries if nm == "pcm" and ch == c] or [np.zeros(0, np.float32)]) for c in range(Cn)]
    want = z["pcm"]
    peak = max(1.0, float(np.abs(want).max()))
    for c in range(Cn):
        assert len(pcm[c]) == want.shape[1] and np.abs(pcm[c] - want[c]).max() <= 4e-6 * peak


def test_corpus_decoder_int16_output():
    """CorpusOptions::pcm_s16 (SURVEY 8 f-3 on the host path): the PCM leaves the device as interleaved int16 — ov_read's
    conversion applied on the GPU, half the bytes over the bus. Equal to the oracle's conversion of the float PCM the same
    decoder delivers otherwise, for real and synthetic files (1-6 channels)."""
    from oracle import oracle_binding as ob
    names = ["test.stereo44khz", "test.mono44khz"] + _synth_names()[:6]
    blobs = [open(os.path.join(GOLDEN, n + ".ogg"), "rb").read() for n in names]
    chans = [int(np.load(os.path.join(GOLDEN, n + ".npz"))["channels"]) for n in names]
    frames, sums, ok, pcm, stats = _run_corpus(blobs, chans, threads=3, feeders=2, files_per_submit=3, cap=131072)
    assert all(ok)
    lib = _corpus_lib()
    lib.ogg_vorbis_decode_corpus_s16.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                                 C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
    lib.ogg_vorbis_decode_corpus_s16.restype = C.c_int
    n = len(blobs)
    datas = (C.c_char_p * n)(*blobs)
    lens = (C.c_size_t * n)(*[len(b) for b in blobs])
    fr = (C.c_uint64 * n)()
    ok16 = (C.c_uint8 * n)()
    out = [np.zeros((131072, chans[i]), np.int16) for i in range(n)]
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in out])
    caps = (C.c_uint64 * n)(*([131072] * n))
    err = C.c_char_p()
    assert lib.ogg_vorbis_decode_corpus_s16(datas, lens, n, 3, 2, 3, 0, fr, ok16, ptrs, caps, None, C.byref(err)) == 0, err.value
    for i in range(n):
        assert ok16[i] and fr[i] == frames[i], names[i]
        want = ob.pcm_interleave(1, np.ascontiguousarray(pcm[i][:, :frames[i]]), frames[i])
        assert np.array_equal(out[i][:frames[i]], want), names[i]


@pytest.mark.parametrize("vq", ["1", "0"])
def test_error_in_mid_stream_still_delivers_the_packets_before_it(vq, tmp_path):
    """The reference hands PCM out packet by packet, so when a CHECK fires in packet k the hooks and gotPcmData have seen everything
    before it (hpp:1045-1054). Here those packets sit in a batch: the reader synthesises and delivers them, then reports the error
    (fault injection: the 20th audio packet fails half way, after its floor rows and residue were appended)."""
    name, k = "test.stereo44khz", 20
    spec, b, z = load_golden(name)
    dump = str(tmp_path / "d.bin")
    env = dict(os.environ, PARSEOGGVORBIS_TEST_FAIL_AT=str(k), PARSEOGGVORBIS_VQ=vq)
    r = subprocess.run([CLI_TESTING, "--in", os.path.join(GOLDEN, name + ".ogg"), "--debug_out", dump], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "injected failure" in (r.stdout + r.stderr)
    # the product build holds no fault injection: the same variable changes nothing there
    r2 = subprocess.run([CLI, "--in", os.path.join(GOLDEN, name + ".ogg")], capture_output=True, text=True, env=env)
    assert r2.returncode == 0, r2.stderr[-500:]
    header, entries = read_dump(dump)
    setup, packets, pcm = split_packets(entries, spec.channels)
    assert len(packets) == k
    want = int(b["emit_len"][:k].sum())
    for c in range(spec.channels):
        got = np.concatenate([np.atleast_1d(v) for v in pcm[c]]) if len(pcm[c]) else np.zeros(0, np.float32)
        assert got.shape[0] == want
        assert np.abs(got - b["pcm"][c, :want]).max() < TOL
