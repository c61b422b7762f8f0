#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE decoder (oracle/_ref/ours.bin, built by oracle/Makefile
from /root/reference/src) run on the reference's two .ogg fixtures.  TEST INFRASTRUCTURE, authoring
container only; the .npz files are committed, the reference itself never travels.

Per fixture the .npz holds (all packets): mode / window flags / page granules parsed from the Ogg framing,
"floor1 ys", "after_residue" (the hot path's inputs) and the streamed "pcm" (its output); for a subset of
packets also the intermediate hooks "floor1 final_ys", "floor1 step2_flag", "floor1 floor",
"after_envelope", "pcm_after_mdct" (src/ParseOggVorbis.hpp:518,560-561,585,1211,1254,1265,1051).
The .ogg files themselves (data files of the reference's tests) are copied next to the vectors.
"""
import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("REF", "/root/reference")
OUT = os.path.join(HERE, "..", "tests", "golden")
TYPES = {1: np.float32, 2: np.int32, 3: np.uint32, 4: np.uint8, 5: np.uint8, 6: np.int64, 7: np.uint64}


def read_dump(path):
    """TLV dump (src/Callbacks.cpp:136-201,317-324): returns (header dict, list of (name, channel, ndarray))."""
    buf = open(path, "rb").read()
    pos = 0

    def rec():
        nonlocal pos
        (ln,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        b = buf[pos:pos + ln]
        pos += ln
        return b

    def kv():
        key = rec().decode()
        tid = rec()[0]
        rec()  # element size
        data = np.frombuffer(rec(), dtype=TYPES[tid]).copy()
        return key, data

    assert rec() == b"ParseOggVorbis-header-v1"
    header = {}
    for _ in range(3):
        k, v = kv()
        header[k] = v
    entries = []
    while pos < len(buf):
        k, v = kv()
        assert k == "entry-name", k
        name = v.tobytes().decode()
        k, v = kv()
        ch = -1
        if k == "entry-channel":
            ch = int(v[0])
            k, v = kv()
        assert k == "entry-data", k
        entries.append((name, ch, v))
    return header, entries


def ogg_packets(path):
    """[(packet bytes, granule of its page if last-on-page else -1)] for a single-stream file."""
    buf = open(path, "rb").read()
    pos, out = 0, []
    while pos < len(buf):
        assert buf[pos:pos + 4] == b"OggS"
        granule = struct.unpack_from("<q", buf, pos + 6)[0]
        nseg = buf[pos + 26]
        table = buf[pos + 27:pos + 27 + nseg]
        data = pos + 27 + nseg
        ln, page = 0, []
        for s in table:
            ln += s
            if s < 255:
                page.append(buf[data:data + ln])
                data += ln
                ln = 0
        assert ln == 0, "page-spanning packet (the reference rejects these too, hpp:89)"
        for i, pk in enumerate(page):
            out.append((pk, granule if i == len(page) - 1 else -1))
        pos = data
    return out


def make(name):
    ogg = os.path.join(REF, "tests", "audio", name)
    ours = os.path.join(HERE, "_ref", "ours.bin")
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "d.bin")
        subprocess.run([ours, "--in", ogg, "--debug_out", dump], check=True, stdout=subprocess.DEVNULL)
        header, entries = read_dump(dump)
    C = int(header["decoder-num-channels"][0])

    floors, i = [], 0
    while entries[i][0] != "finish_setup":
        assert entries[i][0] == "floor1_unpack multiplier" and entries[i + 1][0] == "floor1_unpack xs"
        floors.append((int(entries[i][2][0]), entries[i + 1][2].astype(np.uint32)))
        i += 2
    i += 1

    packets, cur, pcm = [], None, [[] for _ in range(C)]
    for nm, ch, v in entries[i:]:
        if nm == "start_audio_packet":
            cur = dict(floor_number=[0] * C, ys=[None] * C, final_ys=[None] * C, flag=[None] * C, floor=[None] * C,
                       res=[None] * C, env=[None] * C, mdct=[None] * C, pcm_before=sum(len(a) for a in pcm[0]), last_floor_ch=-1)
            packets.append(cur)
        elif nm == "floor_number":
            cur["floor_number"][ch] = int(v[0])
            cur["last_floor_ch"] = ch
        elif nm == "floor1 ys":
            cur["ys"][cur["last_floor_ch"]] = v
        elif nm == "floor1 final_ys":
            cur["final_ys"][cur["last_floor_ch"]] = v
        elif nm == "floor1 step2_flag":
            cur["flag"][cur["last_floor_ch"]] = v
        elif nm == "floor1 floor":
            cur["floor"][cur["last_floor_ch"]] = v
        elif nm == "after_residue":
            cur["res"][ch] = v
        elif nm == "after_envelope":
            cur["env"][ch] = v
        elif nm == "pcm_after_mdct":
            cur["mdct"][ch] = v
        elif nm == "pcm":
            pcm[ch].append(v)
    P = len(packets)
    pcm = np.stack([np.concatenate(c) for c in pcm]).astype(np.float32)

    # framing side info: these fixtures have 2 modes (0 short, 1 long): bit0 type, bit1 mode, bits2/3 prev/next
    audio = ogg_packets(ogg)[3:]
    assert len(audio) == P
    bs = sorted({2 * len(p["res"][0]) for p in packets})
    bs0, bs1 = bs[0], bs[-1]
    mode = np.zeros(P, np.uint8)
    prevf = np.zeros(P, np.uint8)
    nextf = np.zeros(P, np.uint8)
    gran = np.zeros(P, np.int64)
    for k, (pk, g) in enumerate(audio):
        b = pk[0]
        assert (b & 1) == 0
        mode[k] = (b >> 1) & 1
        n = 2 * len(packets[k]["res"][0])
        assert n == (bs1 if mode[k] else bs0)
        if mode[k]:
            prevf[k], nextf[k] = (b >> 2) & 1, (b >> 3) & 1
        gran[k] = g

    stride = (max(len(xs) for _, xs in floors) + 3) & ~3
    ys = np.zeros((P, C, stride), np.uint16)
    used = np.zeros(P, np.uint32)
    res = []
    for k, p in enumerate(packets):
        for c in range(C):
            if p["ys"][c] is not None:
                ys[k, c, :len(p["ys"][c])] = p["ys"][c]
                used[k] |= 1 << c
            res.append(p["res"][c].astype(np.float32))
    # "pcm" entries follow their packet's finish_audio_packet (hpp:1270-1271) and precede the next start
    emit = np.diff([p["pcm_before"] for p in packets] + [pcm.shape[1]]).astype(np.uint32)

    sub = sorted(set(list(range(0, min(8, P))) + list(range(8, P, 9)) + [P - 3, P - 2, P - 1]))
    tap = {}
    for k in sub:
        p = packets[k]
        for c in range(C):
            key = "p%d_c%d_" % (k, c)
            tap[key + "env"] = p["env"][c].astype(np.float32)
            tap[key + "mdct"] = p["mdct"][c].astype(np.float32)
            if p["ys"][c] is not None:
                tap[key + "final_ys"] = p["final_ys"][c].astype(np.uint16)
                tap[key + "flag"] = p["flag"][c].astype(np.uint8)
                tap[key + "floor"] = p["floor"][c].astype(np.uint8)
                assert p["floor"][c].max() < 256

    floor_of = np.array([[p["floor_number"][c] for c in range(C)] for p in packets], np.uint8)
    base = os.path.splitext(name)[0]
    np.savez_compressed(
        os.path.join(OUT, base + ".npz"),
        channels=C, sample_rate=int(header["decoder-sample-rate"][0]), blocksize0=bs0, blocksize1=bs1,
        num_floors=len(floors), **{"floor%d_mult" % f: floors[f][0] for f in range(len(floors))},
        **{"floor%d_xs" % f: floors[f][1] for f in range(len(floors))},
        mode=mode, prev_long=prevf, next_long=nextf, granule=gran, floor_used=used, floor_number=floor_of,
        ys=ys, residue=np.concatenate(res), pcm=pcm, emit_len=emit, tap_packets=np.array(sub, np.int32), **tap)
    shutil.copyfile(ogg, os.path.join(OUT, name))
    print("%s: %d audio packets, %d ch, pcm %s, taps on %d packets" % (name, P, C, pcm.shape, len(sub)))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for nm in ("test.stereo44khz.ogg", "test.mono44khz.ogg"):
        make(nm)
