"""CPU oracle vs the committed dumps of the reference decoder on the reference's own .ogg fixtures
(tests/golden/*.npz, made by oracle/make_golden.py).  Integer hooks exact; float hooks BIT-exact (the oracle
performs the reference's operations in the reference's order)."""
import numpy as np
import pytest

from oracle import oracle_binding as ob
from tests.workloads import load_golden, synth_batch, fixture_like_spec


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["test.stereo44khz", "test.mono44khz"])
def test_full_path_matches_reference_dump(name):
    spec, b, z = load_golden(name)
    orc = ob.OracleSynth(spec, max_streams=1)
    total = b["pcm"].shape[1]
    r = orc.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], total + 8, want_taps=True)
    assert r["rc"] == 0 and r["flags"] == 0
    assert np.array_equal(r["emit_len"], b["emit_len"])
    assert int(r["emit_len"].sum()) == total  # 91136 / 63488 samples, SURVEY 4
    assert np.array_equal(bits(r["pcm"][0][:, :total]), bits(b["pcm"]))
    # intermediate hooks on the tapped subset
    C, bs0, bs1 = spec.channels, spec.blocksize0, spec.blocksize1
    n_of = np.where(z["mode"] == 1, bs1, bs0)
    off = np.concatenate([[0], np.cumsum(C * n_of // 2)])
    for k in z["tap_packets"]:
        n = int(n_of[k])
        for c in range(C):
            key = "p%d_c%d_" % (k, c)
            env = r["taps"]["after_envelope"][off[k] + c * n // 2: off[k] + (c + 1) * n // 2]
            assert np.array_equal(bits(env), bits(z[key + "env"])), key
            md = r["taps"]["pcm_after_mdct"][2 * off[k] + c * n: 2 * off[k] + (c + 1) * n]
            assert np.array_equal(bits(md), bits(z[key + "mdct"])), key
            if key + "final_ys" in z.files:
                mult, xs = spec.floors[int(z["mode"][k])]
                row = r["taps"]["floor_final"].reshape(-1, C, orc.ys_stride)[k, c, :len(xs)]
                assert np.array_equal(row & 0x7FFF, z[key + "final_ys"] * mult)
                assert np.array_equal(row >> 15, z[key + "flag"])
                cv = r["taps"]["floor_curve"][off[k] + c * n // 2: off[k] + (c + 1) * n // 2]
                assert np.array_equal(cv, z[key + "floor"][:n // 2]), key  # "floor1 floor" (SURVEY 8 f-4 feature tap)
                tail = int(row[1] & 0x7FFF) if row[1] >> 15 else int(cv[-1])  # flat from the last flagged post on (hpp:583-584)
                assert (z[key + "floor"][n // 2:] == tail).all()


def test_block_sizes_seen_in_fixture():
    _, b, z = load_golden("test.stereo44khz")
    assert sorted(set(b["emit_len"].tolist())) == [0, 64, 128, 576, 1024]
    assert list(z["mode"][:5]) == [0, 0, 0, 0, 1]


def test_streaming_across_submits_equals_one_submit():
    """Overlap state carried between batches (SURVEY 5 'checkpoint/resume'): splitting a stream over several
    submits gives the same PCM as one submit."""
    spec, b, _ = load_golden("test.stereo44khz")
    orc = ob.OracleSynth(spec, max_streams=1)
    total = b["pcm"].shape[1]
    P = len(b["packets"])
    n_of = np.where(b["packets"]["mode"] == 1, spec.blocksize1, spec.blocksize0)
    off = np.concatenate([[0], np.cumsum(spec.channels * n_of // 2)])
    cuts = [0, 3, 4, 5, 40, P]
    got = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        seg = b["segments"].copy()
        seg["first_packet"], seg["num_packets"], seg["flags"] = 0, e - a, 1 if a == 0 else 0
        r = orc.submit_host(b["packets"][a:e], seg, b["ys"][a:e], b["residue"][off[a]:off[e]], total)
        assert r["rc"] == 0
        got.append(r["pcm"][0][:, :int(r["emit_len"].sum())])
    assert np.array_equal(np.concatenate(got, axis=1), b["pcm"])


def test_error_flags():
    spec = fixture_like_spec(2)
    orc = ob.OracleSynth(spec, max_streams=2)
    b = synth_batch(spec, 2, 6, "mixed", seed=5)
    ok = orc.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert ok["rc"] == 0
    bad = b["packets"].copy()
    bad["granule"][4] = 10 ** 9  # page claims more samples than the packets provide -> hpp:1041
    r = orc.submit_host(bad, b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert r["rc"] == 4 and r["flags"] & 4 and r["first_bad"] == 4
    ys = b["ys"].copy()
    ys[7, 0, 2:] = 255  # nonsense amplitudes -> floor CHECKs (hpp:536 / 587)
    pk = b["packets"].copy()
    pk["floor_used"][7] |= 1
    r = orc.submit_host(pk, b["segments"], ys, b["residue"], b["plane_stride"])
    assert r["rc"] == 4 and r["flags"] & 3 and r["first_bad"] == 7
    r = orc.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], 100)
    assert r["flags"] & 8  # plane overflow


def _synth_names():
    import os
    from tests.workloads import GOLDEN
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("synth_") and f.endswith(".ogg"))


@pytest.mark.parametrize("name", _synth_names())
def test_full_path_matches_reference_on_synthetic_streams(name, tmp_path_factory):
    """The oracle's whole synthesis path (floor unwrap + curve, propagate, coupling chains, product, IMDCT of every block size,
    windows, overlap, granule clipping) on the synthetic streams of oracle/make_synth_ogg.py — 2-6 channels, floor multipliers
    1-4, up to three coupling steps, three modes, block sizes 64 ... 4096: bit-identical to the REFERENCE decoder's PCM.
    Inputs are the host decoder's entropy output (tests/host_entropy_dump.cpp), itself checked against the reference's hooks
    in tests/test_host_decoder.py."""
    import os
    import subprocess
    from parseoggvorbis_amd.binding import SEGMENT_DTYPE, SetupSpec
    from tests.workloads import GOLDEN, build_probe, read_entropy_dump
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    td = tmp_path_factory.mktemp("probe_" + name)
    probe = build_probe(td)
    out = os.path.join(td, "e.bin")
    r = subprocess.run([probe, os.path.join(GOLDEN, name + ".ogg"), out], capture_output=True, text=True,
                       env=dict(os.environ, PARSEOGGVORBIS_VQ="0"))
    assert r.returncode == 0, r.stderr
    d = read_entropy_dump(out)
    C = int(z["channels"])
    floors = [(int(z["floor%d_mult" % k]), [int(x) for x in z["floor%d_xs" % k]]) for k in range(int(z["num_floors"]))]
    nmap = int(z["mode_mapping"].max()) + 1
    mappings = [([(int(a), int(b)) for a, b in z["coupling_m%d" % k]], [int(f) for f in z["chfloor_m%d" % k]]) for k in range(nmap)]
    modes = [(int(bf), int(mp)) for bf, mp in zip(z["mode_blockflag"], z["mode_mapping"])]
    spec = SetupSpec(C, int(z["blocksize0"]), int(z["blocksize1"]), floors, mappings, modes)
    seg = np.zeros(1, SEGMENT_DTYPE)
    seg["num_packets"], seg["flags"] = d["P"], 1
    total = z["pcm"].shape[1]
    orc = ob.OracleSynth(spec, max_streams=1)
    assert orc.ys_stride == d["ys_stride"]
    res = orc.submit_host(d["packets"], seg, d["ys"], d["residue"], total + 8)
    assert res["rc"] == 0 and res["flags"] == 0, res
    assert int(res["emit_len"].sum()) == total
    assert np.array_equal(bits(res["pcm"][0][:, :total]), bits(z["pcm"]))
