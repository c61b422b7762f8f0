"""CPU-side tests of the host decoder (parseoggvorbis_amd/host): its entropy half against the reference decoder's
hooks (tests/golden), the debug-hook dump format, the CLI contract and the "no CPU path" rule."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as entry
from parseoggvorbis_amd.binding import PACKET_DTYPE
from tests.dump_reader import read_dump
from tests.workloads import GOLDEN, load_golden, read_entropy_dump

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "parseoggvorbis_amd", "host")
CLI = os.path.join(HOST, "ours_hip.bin")


@pytest.fixture(scope="module")
def built():
    entry.build_hip()
    entry.build_host()
    return True


@pytest.fixture(scope="module")
def probe(built, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("probe") / "host_entropy_dump")
    csrc = os.path.join(ROOT, "parseoggvorbis_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", out, os.path.join(ROOT, "tests", "host_entropy_dump.cpp"),
                    "-L" + HOST, "-lparseoggvorbis_amd", "-L" + csrc, "-lvorbis_synth_hip", "-Wl,-rpath," + HOST,
                    "-Wl,-rpath," + csrc, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return out


def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.parametrize("name", ["test.stereo44khz", "test.mono44khz"])
def test_entropy_half_matches_reference_hooks(probe, name, tmp_path):
    """mode / window flags / page granules / 'floor1 ys' / 'after_residue' of every packet == reference (exact).
    Float-residue mode of the host decoder (PARSEOGGVORBIS_VQ=0)."""
    spec, b, _ = load_golden(name)
    out = str(tmp_path / "e.bin")
    r = subprocess.run([probe, os.path.join(GOLDEN, name + ".ogg"), out], capture_output=True, text=True,
                       env=dict(os.environ, PARSEOGGVORBIS_VQ="0"))
    assert r.returncode == 0, r.stderr
    d = read_entropy_dump(out)
    assert (d["P"], d["channels"], d["blocksize0"], d["blocksize1"]) == (len(b["packets"]), spec.channels, spec.blocksize0,
                                                                         spec.blocksize1)
    for k in ("mode", "prev_long", "next_long", "floor_used", "granule"):
        assert np.array_equal(d["packets"][k], b["packets"][k]), k
    assert np.array_equal(d["ys"], b["ys"])
    assert np.array_equal(d["residue"].view(np.uint32), b["residue"].view(np.uint32))
    assert "vq_packets" not in d


@pytest.mark.parametrize("name", ["test.stereo44khz", "test.mono44khz"])
def test_vq_entries_rebuild_reference_residue(probe, name, tmp_path):
    """VQ mode (default): the host ships classification + entry numbers; the ORACLE's accumulate stage
    (hpp:725-757 restated) turns them back into the reference decoder's 'after_residue', bit for bit, for every packet
    of both fixtures (format 2 stereo, format 1 mono). Pins oracle + host entry stream against the reference."""
    from oracle import oracle_binding as ob
    spec, b, _ = load_golden(name)
    out = str(tmp_path / "e.bin")
    r = subprocess.run([probe, os.path.join(GOLDEN, name + ".ogg"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = read_entropy_dump(out)
    assert d["residue"].size == 0 and d["residue_floats"] == b["residue"].size and len(d["vq_packets"]) == d["P"]
    assert np.array_equal(d["ys"], b["ys"])
    Cn = spec.channels
    off = 0
    for p in range(d["P"]):
        mode = int(d["packets"]["mode"][p])
        n2 = spec.blocksize_of_mode(mode) // 2
        vp = d["vq_packets"][p]
        c0 = int(vp["cls_off"])
        c1 = int(d["vq_packets"][p + 1]["cls_off"]) if p + 1 < d["P"] else d["cls"].size
        e0, ne = int(vp["entry_off"]), int(vp["num_entries"])
        # floor_output_used after the nonzero propagate (hpp:1174-1180)
        used = int(d["packets"]["floor_used"][p])
        for mag, ang in spec.mappings[spec.modes[mode][1]][0]:
            if (used >> mag) & 1 or (used >> ang) & 1:
                used |= (1 << mag) | (1 << ang)
        rc, res = ob.residue_vq(d["vq_spec"], spec.modes[mode][1], Cn, n2, used, d["cls"][c0:c1], d["entries"][e0:e0 + ne])
        assert rc == 0, (p, rc)
        want = b["residue"][off:off + Cn * n2]
        assert np.array_equal(res.view(np.uint32), want.view(np.uint32)), p
        off += Cn * n2
    assert off == b["residue"].size
    # a damaged entry stream is reported, not mis-decoded
    vp = d["vq_packets"][5]
    e0, ne = int(vp["entry_off"]), int(vp["num_entries"])
    mode = int(d["packets"]["mode"][5])
    rc, _ = ob.residue_vq(d["vq_spec"], spec.modes[mode][1], Cn, spec.blocksize_of_mode(mode) // 2, 3, d["cls"][int(vp["cls_off"]):],
                          d["entries"][e0:e0 + ne - 1])
    assert rc == 64  # VSYN_ST_BAD_VQ


def test_cli_contract(built, tmp_path):
    """--help / bad args -> usage + exit 1 (reference: src/Callbacks.cpp:392-440); missing file -> exit 1."""
    r = subprocess.run([CLI, "--help"], capture_output=True, text=True)
    assert r.returncode == 1 and "--in ogg_filename" in r.stdout
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 1 and "need to provide --in" in r.stderr
    r = subprocess.run([CLI, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "unexpected arg" in r.stderr
    r = subprocess.run([CLI, "--in", str(tmp_path / "missing.ogg")], capture_output=True, text=True)
    assert r.returncode == 1 and "check failed" in r.stderr  # "file:line: check failed: expr" convention


def test_cli_fails_loudly_without_gpu(built):
    if has_gpu():
        pytest.skip("GPU present")
    r = subprocess.run([CLI, "--in", os.path.join(GOLDEN, "test.stereo44khz.ogg")], capture_output=True, text=True)
    assert r.returncode == 1
    assert "Setup: num codebooks: 38, num floors: 2, num mappings: 2, num modes: 2, num residues: 2" in r.stdout
    assert "no HIP device" in r.stderr  # never a silent CPU fallback


def test_corrupt_page_is_rejected(built, tmp_path):
    data = bytearray(open(os.path.join(GOLDEN, "test.mono44khz.ogg"), "rb").read())
    data[4000] ^= 0x55  # breaks a page CRC
    p = tmp_path / "bad.ogg"
    p.write_bytes(bytes(data))
    r = subprocess.run([CLI, "--in", str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "check failed: want_crc == crc" in r.stderr


def test_hook_dump_format(built, tmp_path):
    """The hook module writes the TLV layout the reference's harness parses (types, channel field, NULL payloads)."""
    lib = C.CDLL(os.path.join(HOST, "libparseoggvorbis_amd.so"))
    lib.push_data_float.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.push_data_u32.argtypes = lib.push_data_u8.argtypes = lib.push_data_i64.argtypes = lib.push_data_float.argtypes
    lib.register_decoder_ref.argtypes = [C.c_void_p, C.c_char_p, C.c_long, C.c_int]
    lib.unregister_decoder_ref.argtypes = [C.c_void_p]
    lib.set_data_output_file.argtypes = [C.c_char_p]
    lib.generic_itoa.restype = C.c_char_p
    path = str(tmp_path / "d.bin")
    ref = C.c_void_p(0x1234)
    lib.set_data_output_file(path.encode())
    lib.register_decoder_ref(ref, b"unit", 44100, 2)
    f = np.arange(5, dtype=np.float32)
    u = np.array([7, 8, 9], np.uint32)
    g = np.array([-1], np.int64)
    lib.push_data_u8(ref, b"finish_setup", -1, None, 0)
    lib.push_data_float(ref, b"after_residue", 1, f.ctypes.data, 5)
    lib.push_data_u32(ref, b"floor1 ys", -1, u.ctypes.data, 3)
    lib.push_data_i64(ref, b"expected_ending_total_pos", -1, g.ctypes.data, 1)
    lib.unregister_decoder_ref(ref)  # closes the file
    header, entries = read_dump(path)
    assert header["decoder-name"].tobytes() == b"unit" and int(header["decoder-sample-rate"][0]) == 44100
    assert int(header["decoder-num-channels"][0]) == 2
    assert [(n, c) for n, c, _, _ in entries] == [("finish_setup", -1), ("after_residue", 1), ("floor1 ys", -1),
                                                   ("expected_ending_total_pos", -1)]
    assert len(entries[0][2]) == 0 and np.array_equal(entries[1][2], f) and np.array_equal(entries[2][2], u)
    assert entries[3][3] == 6 and int(entries[3][2][0]) == -1
    assert lib.generic_itoa(5, 2, 8) == b"00000101"


CORPUS_CLI = os.path.join(HOST, "corpus_hip.bin")


def test_corpus_entropy_workers_count_packets(built):
    """Corpus front-end, workers only (no GPU involved): every replica of both fixtures is parsed, on several threads,
    and the audio packet totals are those of the reference's decode (tests/golden)."""
    import json
    names = ["test.stereo44khz", "test.mono44khz"]
    per = sum(len(load_golden(n)[1]["packets"]) for n in names)
    r = subprocess.run([CORPUS_CLI, "--threads", "3", "--replicas", "5", "--entropy_only"] +
                       [os.path.join(GOLDEN, n + ".ogg") for n in names], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["files"] == 10 and out["failed"] == 0 and out["submits"] == 0
    assert out["audio_packets"] == 5 * per


def test_corpus_fails_loudly_without_gpu(built):
    if has_gpu():
        pytest.skip("GPU present")
    r = subprocess.run([CORPUS_CLI, "--threads", "2", "--replicas", "3", os.path.join(GOLDEN, "test.mono44khz.ogg")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and r.stdout == ""


def test_mutated_files_never_crash_the_front_end(built, tmp_path):
    """Robustness (no GPU involved): damaged files — flipped bits, overwritten bytes, truncation, with page CRCs
    re-computed so that the damage reaches the codec layer — end in 'ok' or in a per-file error, never in a crash.
    (tools/fuzz_host.cpp is the sanitizer-instrumented long-running version of this.)"""
    import json
    import zlib  # noqa: F401  (only to make sure the interpreter's C extensions load before the subprocess storm)

    def crc_tab():
        t = []
        for i in range(256):
            r = i << 24
            for _ in range(8):
                r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
            t.append(r)
        return t

    tab = crc_tab()

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

    rng = np.random.default_rng(2024)
    base = [open(os.path.join(GOLDEN, n + ".ogg"), "rb").read() for n in ("test.stereo44khz", "test.mono44khz")]
    paths = []
    for k in range(48):
        b = bytearray(base[k % 2])
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(0, len(b)))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                b[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                b[pos] = int(rng.integers(0, 256))
            elif len(b) > 200:
                del b[len(b) - int(rng.integers(1, len(b) // 2)):]
        if k % 4:
            fix_crcs(b)
        p = tmp_path / ("m%02d.ogg" % k)
        p.write_bytes(bytes(b))
        paths.append(str(p))
    r = subprocess.run([CORPUS_CLI, "--threads", "4", "--entropy_only"] + paths, capture_output=True, text=True)
    assert r.returncode in (0, 1), (r.returncode, r.stderr[-500:])  # 1: some files failed (expected); negative: a crash
    out = json.loads(r.stdout)
    assert out["files"] == 48 and 0 < out["failed"] <= 48


@pytest.mark.parametrize("vq", ["1", "0"])
def test_packet_that_fails_half_way_leaves_the_batch_consistent(probe, tmp_path, vq):
    """Damage INSIDE audio packets (page CRCs re-computed): the entropy half may give up in the middle of a packet, after it has
    appended floor rows / residue / entry numbers. The batch it then hands over (the reader delivers the good packets in front of
    the error, as the reference does packet by packet, hpp:1045-1054) must account for every row: ys, floor numbers, residue or
    entries exactly as long as the packet list says. (CorpusDecoder's feeders size their staging buffers from the packet list.)"""
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
            ln = 27 + ns + sum(b[o + 27:o + 27 + ns])
            if o + ln > len(b):
                break
            b[o + 22:o + 26] = b"\0\0\0\0"
            c = 0
            for x in b[o:o + ln]:
                c = ((c << 8) & 0xFFFFFFFF) ^ tab[((c >> 24) & 0xFF) ^ x]
            b[o + 22:o + 26] = c.to_bytes(4, "little")
            o += ln

    # deterministic first: the 20th audio packet fails after its floor rows and residue were appended (fault injection in
    # VorbisStream::parse_audio) — the reader must deliver exactly the 20 packets in front of it, consistently
    from tests.workloads import build_probe
    probe_t = build_probe(tmp_path, testing=True)  # (the fault injection exists in the TESTING build of the library only)
    for name, want in (("test.stereo44khz", 20), ("test.mono44khz", 7)):
        r = subprocess.run([probe_t, "--check", os.path.join(GOLDEN, name + ".ogg")], capture_output=True, text=True,
                           env=dict(os.environ, PARSEOGGVORBIS_VQ=vq, PARSEOGGVORBIS_TEST_FAIL_AT=str(want)))
        assert r.returncode == 0 and r.stdout.startswith("consistent batches=1 packets=%d error=1" % want), (r.returncode, r.stdout, r.stderr[-300:])
    rng = np.random.default_rng(77)
    base = open(os.path.join(GOLDEN, "test.stereo44khz.ogg"), "rb").read()
    env = dict(os.environ, PARSEOGGVORBIS_VQ=vq)
    partial = 0
    for k in range(40):
        b = bytearray(base)
        # the audio pages start behind the three header packets (~ the first 4.3 KB of this file)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(6000, len(b) - 100))
            b[pos] = int(rng.integers(0, 256))
        fix_crcs(b)
        p = tmp_path / ("h%02d.ogg" % k)
        p.write_bytes(bytes(b))
        r = subprocess.run([probe, "--check", str(p)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, (k, r.returncode, r.stdout, r.stderr[-300:])
        assert r.stdout.startswith("consistent")
        if "error=1" in r.stdout and "packets=0" not in r.stdout:
            partial += 1
    assert partial > 0  # some damage did end a stream half way with good packets in front


SYNTH = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("synth_") and f.endswith(".ogg"))


@pytest.mark.parametrize("name", SYNTH)
def test_synthetic_streams_entropy_half_matches_reference(probe, name, tmp_path):
    """Streams written by oracle/make_synth_ogg.py from the Vorbis I specification — setups the two real fixtures do not have:
    1-3 channels, other block sizes, floor multipliers / post counts / subclass books, residue formats 0 / 1 / 2, vector lengths
    that are not powers of two, lookup types 1 and 2, sequence_p, sparse and ordered codebooks, two submaps, several coupling
    steps — with the REFERENCE decoder's hooks on them as golden vectors. The host's entropy half must reproduce the
    reference's 'floor1 ys' and 'after_residue' exactly, both as floats (PARSEOGGVORBIS_VQ=0) and as entry numbers pushed
    through the oracle's accumulate stage (VQ mode)."""
    from oracle import oracle_binding as ob
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = str(tmp_path / "e.bin")
    for vq in ("0", "1"):
        r = subprocess.run([probe, os.path.join(GOLDEN, name + ".ogg"), out], capture_output=True, text=True,
                           env=dict(os.environ, PARSEOGGVORBIS_VQ=vq))
        assert r.returncode == 0, r.stderr
        d = read_entropy_dump(out)
        Cn = d["channels"]
        assert d["P"] == int(z["packets"]) and Cn == int(z["channels"])
        assert (d["blocksize0"], d["blocksize1"]) == (int(z["blocksize0"]), int(z["blocksize1"]))
        off = 0
        for ln, where in zip(z["ys_len"], z["ys_where"]):
            p, c = divmod(int(where), 1000)
            assert np.array_equal(d["ys"][p, c, :ln].astype(np.uint32), z["ys"][off:off + ln]), (p, c)
            off += ln
        if vq == "0":
            assert "vq_spec" not in d
            assert np.array_equal(d["residue"].view(np.uint32), z["residue"].view(np.uint32))
            continue
        assert d["residue"].size == 0 and d["residue_floats"] == z["residue"].size
        roff = 0
        for p in range(d["P"]):
            mode = int(d["packets"]["mode"][p])
            mapping = int(z["mode_mapping"][mode])
            n2 = (d["blocksize1"] if int(z["mode_blockflag"][mode]) else d["blocksize0"]) // 2
            used = int(d["packets"]["floor_used"][p])
            for mag, ang in z["coupling_m%d" % mapping]:  # nonzero propagate, hpp:1174-1180
                if (used >> int(mag)) & 1 or (used >> int(ang)) & 1:
                    used |= (1 << int(mag)) | (1 << int(ang))
            vp = d["vq_packets"][p]
            c0 = int(vp["cls_off"])
            c1 = int(d["vq_packets"][p + 1]["cls_off"]) if p + 1 < d["P"] else d["cls"].size
            e0, ne = int(vp["entry_off"]), int(vp["num_entries"])
            rc, res = ob.residue_vq(d["vq_spec"], mapping, Cn, n2, used, d["cls"][c0:c1], d["entries"][e0:e0 + ne])
            assert rc == 0, (p, rc)
            assert np.array_equal(res.view(np.uint32), z["residue"][roff:roff + Cn * n2].view(np.uint32)), p
            roff += Cn * n2
        assert roff == z["residue"].size
