"""GPU: the residue VQ stage (include/vorbis_synth_hip.h "residue VQ stage").  The kernel rebuilds
'after_residue' from classification + entry numbers; checked bit-exactly against (a) the reference decoder's own
'after_residue' dumps for both fixtures (tests/golden), with the PCM of the whole path behind it, and (b) the oracle's
restatement of hpp:725-757 on random well-formed entry streams, plus the error reporting."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_binding as ob
from parseoggvorbis_amd.binding import (PACKET_DTYPE, SEGMENT_DTYPE, VQ_PACKET_DTYPE, VSYN_ERR_STREAM, VSYN_SEG_RESET,
                                        VSYN_ST_BAD_VQ, Synth, VsynError, VqSpec)
from tests.workloads import GOLDEN, build_probe, load_golden, read_entropy_dump, synth_vq_packet

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    return build_probe(tmp_path_factory.mktemp("probe"))


def _dump(probe, name, tmp_path):
    out = str(tmp_path / (name + ".bin"))
    r = subprocess.run([probe, os.path.join(GOLDEN, name + ".ogg"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return read_entropy_dump(out)


def _propagated(spec, mode, own):
    used = int(own)
    for mag, ang in spec.mappings[spec.modes[mode][1]][0]:
        if (used >> mag) & 1 or (used >> ang) & 1:
            used |= (1 << mag) | (1 << ang)
    return used


@pytest.mark.parametrize("name", ["test.stereo44khz", "test.mono44khz"])
def test_vq_stage_reproduces_reference_residue_and_pcm(probe, name, tmp_path):
    spec, b, _ = load_golden(name)
    d = _dump(probe, name, tmp_path)
    P = d["P"]
    syn = Synth(spec, max_streams=1)
    syn.attach_vq(d["vq_spec"])
    seg = np.zeros(1, SEGMENT_DTYPE)
    seg[0] = (0, 0, P, VSYN_SEG_RESET, 0)
    out = syn.submit_host_vq(d["packets"], seg, d["ys"], d["vq_packets"], d["cls"], d["entries"], d["residue_floats"],
                             P * spec.blocksize1 // 2)
    assert out["rc"] == 0 and out["flags"] == 0
    assert np.array_equal(out["residue"].view(np.uint32), b["residue"].view(np.uint32))  # == reference 'after_residue'
    total = b["pcm"].shape[1]
    assert int(out["emit_len"].sum()) == total
    assert np.abs(out["pcm"][0][:, :total] - b["pcm"]).max() < TOL
    # same batch cut into two submits (overlap carried on the device) gives the same PCM
    syn.reset()
    cut = 41
    got = []
    for lo, hi, flag in ((0, cut, VSYN_SEG_RESET), (cut, P, 0)):
        sg = np.zeros(1, SEGMENT_DTYPE)
        sg[0] = (0, 0, hi - lo, flag, 0)
        vqp = d["vq_packets"][lo:hi].copy()
        c0 = int(vqp["cls_off"][0])
        e0 = int(vqp["entry_off"][0])
        c1 = int(d["vq_packets"]["cls_off"][hi]) if hi < P else d["cls"].size
        e1 = int(d["vq_packets"]["entry_off"][hi]) if hi < P else d["entries"].size
        vqp["cls_off"] -= c0
        vqp["entry_off"] -= e0
        n_of = [spec.blocksize_of_mode(int(m)) // 2 * spec.channels for m in d["packets"]["mode"][lo:hi]]
        o = syn.submit_host_vq(d["packets"][lo:hi], sg, d["ys"][lo:hi], vqp, d["cls"][c0:c1], d["entries"][e0:e1], int(sum(n_of)),
                               (hi - lo) * spec.blocksize1 // 2, want_residue=False)
        assert o["rc"] == 0
        got.append(o["pcm"][0][:, :int(o["emit_len"].sum())])
    assert np.abs(np.concatenate(got, axis=1) - b["pcm"]).max() < TOL


def _random_vq_batch(spec, vq_spec, streams, per_stream, pattern, seed):
    rng = np.random.default_rng(seed)
    Cn = spec.channels
    long_mode = [i for i, (bf, _) in enumerate(spec.modes) if bf][0]
    short_mode = [i for i, (bf, _) in enumerate(spec.modes) if not bf][0]
    P = streams * per_stream
    pk = np.zeros(P, PACKET_DTYPE)
    vqp = np.zeros(P, VQ_PACKET_DTYPE)
    seg = np.zeros(streams, SEGMENT_DTYPE)
    cls_l, ent_l, want = [], [], []
    c_off = e_off = r_off = 0
    for s in range(streams):
        seg[s] = (s, s * per_stream, per_stream, VSYN_SEG_RESET, r_off)
        for q in range(per_stream):
            p = s * per_stream + q
            lng = pattern[q % len(pattern)]
            mode = long_mode if lng else short_mode
            pk[p]["mode"], pk[p]["prev_long"], pk[p]["next_long"], pk[p]["granule"] = mode, 1, 1, -1
            if lng:
                pk[p]["prev_long"] = pattern[(q - 1) % len(pattern)] if q else 1
                pk[p]["next_long"] = pattern[(q + 1) % len(pattern)] if q + 1 < per_stream else 1
            own = int(rng.integers(0, 1 << Cn)) if rng.random() < 0.3 else (1 << Cn) - 1
            pk[p]["floor_used"] = own  # ys stay zero: a flat floor; this test is about the residue
            n2 = spec.blocksize_of_mode(mode) // 2
            used = _propagated(spec, mode, own)
            cls, ent = synth_vq_packet(vq_spec, spec.modes[mode][1], Cn, n2, used, rng)
            rc, res = ob.residue_vq(vq_spec, spec.modes[mode][1], Cn, n2, used, cls, ent)
            assert rc == 0
            vqp[p] = (e_off, ent.size, c_off)
            cls_l.append(cls)
            ent_l.append(ent)
            want.append(res)
            c_off += cls.size
            e_off += ent.size
            r_off += Cn * n2
    return pk, seg, vqp, np.concatenate(cls_l), np.concatenate(ent_l), np.concatenate(want)


@pytest.mark.parametrize("name,pattern", [("test.stereo44khz", [1]), ("test.stereo44khz", [1, 1, 0, 0, 0, 1]), ("test.mono44khz", [1, 0, 0])])
def test_vq_stage_matches_oracle_on_random_entries(probe, name, pattern, tmp_path):
    """Random classifications / entry numbers over the fixtures' real codebooks and residue setups (format 2 stereo with
    partially unused channels, format 1 mono), long and short blocks, 8 streams: device == oracle, bit for bit."""
    spec, _, _ = load_golden(name)
    d = _dump(probe, name, tmp_path)
    pk, seg, vqp, cls, ent, want = _random_vq_batch(spec, d["vq_spec"], 8, 24, pattern, seed=7)
    syn = Synth(spec, max_streams=8)
    syn.attach_vq(d["vq_spec"])
    ys = np.zeros((len(pk), spec.channels, syn.ys_stride), np.uint16)
    out = syn.submit_host_vq(pk, seg, ys, vqp, cls, ent, want.size, 24 * spec.blocksize1 // 2)
    assert out["rc"] == 0, out
    assert np.array_equal(out["residue"].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("tables_in_lds", [True, False])
def test_vq_stage_tables_in_lds_and_in_global_memory(tables_in_lds, monkeypatch):
    """The two ways the VQ kernel reads codebook value tables — one copy per workgroup in LDS when the setup's tables fit (the
    synthetic setup: 15 KB; workgroups of several waves), or gathered from global memory (every setup with a large lattice book, e.g.
    the stereo fixture's 205 KB table; here forced with VSYN_VQ_NO_LDS_TABLES=1) — rebuild the same residue, bit for bit the
    oracle's, on long and short blocks with partially unused channels. Which way ran is asserted (vsyn_fused_paths bit 8)."""
    from tests.workloads import fixture_like_spec, synthetic_vq_spec
    if not tables_in_lds:
        monkeypatch.setenv("VSYN_VQ_NO_LDS_TABLES", "1")
    spec = fixture_like_spec(2)
    vqs = synthetic_vq_spec(2, spec.blocksize1)
    pk, seg, vqp, cls, ent, want = _random_vq_batch(spec, vqs, 40, 12, [1, 1, 0, 1, 0, 0], seed=33)
    syn = Synth(spec, max_streams=40)
    syn.attach_vq(vqs)
    assert bool(syn.fused_paths & 0x100) == tables_in_lds, hex(syn.fused_paths)
    ys = np.zeros((len(pk), 2, syn.ys_stride), np.uint16)
    out = syn.submit_host_vq(pk, seg, ys, vqp, cls, ent, want.size, 12 * spec.blocksize1 // 2)
    assert out["rc"] == 0, out
    assert np.array_equal(out["residue"].view(np.uint32), want.view(np.uint32))
    assert np.abs(want).max() > 0


@pytest.mark.parametrize("variant", ["format0", "format1x2", "submaps"])
def test_vq_stage_other_residue_formats(variant):
    """Residue formats / shapes the fixtures do not have (format 0, two format-1 vectors in one submap with unused channels,
    vector lengths that are not powers of two, partitions that are not multiples of 8, begin > 0, two submaps): device ==
    oracle bit for bit on random well-formed entry streams, long and short blocks."""
    from tests.workloads import exotic_vq_spec, fixture_like_spec
    spec = fixture_like_spec(2)
    vqs = exotic_vq_spec(variant)
    pk, seg, vqp, cls, ent, want = _random_vq_batch(spec, vqs, 6, 20, [1, 0, 1, 1, 0], seed=21)
    syn = Synth(spec, max_streams=6)
    syn.attach_vq(vqs)
    ys = np.zeros((len(pk), 2, syn.ys_stride), np.uint16)
    out = syn.submit_host_vq(pk, seg, ys, vqp, cls, ent, want.size, 20 * spec.blocksize1 // 2)
    assert out["rc"] == 0, out
    assert np.array_equal(out["residue"].view(np.uint32), want.view(np.uint32))
    assert np.abs(want).max() > 0  # the streams do carry values
    if variant != "format0":
        # both ways the kernel reads entry numbers are covered: staged in LDS (<= 2048 per packet) and straight from memory
        assert vqp["num_entries"].min() <= 2048 < vqp["num_entries"].max()


def test_vq_stage_reports_bad_streams(probe, tmp_path):
    spec, _, _ = load_golden("test.stereo44khz")
    d = _dump(probe, "test.stereo44khz", tmp_path)
    pk, seg, vqp, cls, ent, want = _random_vq_batch(spec, d["vq_spec"], 2, 6, [1], seed=3)
    syn = Synth(spec, max_streams=2)
    with pytest.raises(VsynError):  # not attached yet
        syn.submit_host_vq(pk, seg, np.zeros((len(pk), 2, syn.ys_stride), np.uint16), vqp, cls, ent, want.size, 6 * 1024)
    syn.attach_vq(d["vq_spec"])
    ys = np.zeros((len(pk), 2, syn.ys_stride), np.uint16)
    # an entry number beyond its codebook
    bad = ent.copy()
    k = int(vqp["entry_off"][7])
    bad[k] = 65535
    out = syn.submit_host_vq(pk, seg, ys, vqp, cls, bad, want.size, 6 * 1024)
    assert out["rc"] == VSYN_ERR_STREAM and out["flags"] & VSYN_ST_BAD_VQ and out["first_bad"] == 7
    # an entry count that does not match the classifications
    short = vqp.copy()
    short["num_entries"][3] -= 1
    out = syn.submit_host_vq(pk, seg, ys, short, cls, ent, want.size, 6 * 1024)
    assert out["rc"] == VSYN_ERR_STREAM and out["flags"] & VSYN_ST_BAD_VQ and out["first_bad"] == 3
    # a clean batch afterwards is clean
    out = syn.submit_host_vq(pk, seg, ys, vqp, cls, ent, want.size, 6 * 1024)
    assert out["rc"] == 0 and np.array_equal(out["residue"].view(np.uint32), want.view(np.uint32))
    # setups outside the stage's limits are refused with a reason
    books = list(d["vq_spec"].codebooks)
    res = [dict(r) for r in d["vq_spec"].residues]
    res[0]["partition_size"] = 7
    with pytest.raises(VsynError, match="divide"):
        syn.attach_vq(VqSpec(books, res, d["vq_spec"].mappings))


def test_vq_stage_survives_garbage(probe, tmp_path):
    """Untrusted input: random classification bytes, random entry numbers, descriptors whose counts do not match. The stage
    must flag the packets (or, by luck, accept them), never read outside what it was given, and be exact again on the next
    clean batch."""
    spec, _, _ = load_golden("test.stereo44khz")
    d = _dump(probe, "test.stereo44khz", tmp_path)
    pk, seg, vqp, cls, ent, want = _random_vq_batch(spec, d["vq_spec"], 4, 12, [1, 0, 1, 1], seed=11)
    syn = Synth(spec, max_streams=4)
    syn.attach_vq(d["vq_spec"])
    ys = np.zeros((len(pk), 2, syn.ys_stride), np.uint16)
    rng = np.random.default_rng(5)
    flagged = 0
    for trial in range(12):
        c2, e2, v2 = cls.copy(), ent.copy(), vqp.copy()
        kind = trial % 4
        if kind == 0:
            c2[:] = rng.integers(0, 256, c2.size)
        elif kind == 1:
            e2[:] = rng.integers(0, 65536, e2.size)
        elif kind == 2:
            hit = rng.random(c2.size) < 0.05
            c2[hit] = rng.integers(0, 256, int(hit.sum()))
            hit = rng.random(e2.size) < 0.05
            e2[hit] = rng.integers(0, 65536, int(hit.sum()))
        else:  # counts / offsets that do not describe the stream (still inside the arrays, which the host checks)
            v2["num_entries"] = rng.integers(0, 3000, len(v2))
            v2["entry_off"] = rng.integers(0, max(1, e2.size - 3000), len(v2))
            v2["cls_off"] = rng.integers(0, c2.size, len(v2))
        out = syn.submit_host_vq(pk, seg, ys, v2, c2, e2, want.size, 12 * 1024)
        assert out["rc"] in (0, VSYN_ERR_STREAM)
        flagged += out["rc"] == VSYN_ERR_STREAM and bool(out["flags"] & VSYN_ST_BAD_VQ)
        clean = syn.submit_host_vq(pk, seg, ys, vqp, cls, ent, want.size, 12 * 1024)
        assert clean["rc"] == 0 and np.array_equal(clean["residue"].view(np.uint32), want.view(np.uint32))
    assert flagged >= 9
