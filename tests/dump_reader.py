"""Reader for the debug-hook dump format ("ParseOggVorbis-header-v1"; writer: parseoggvorbis_amd/host/hooks.cpp,
reference writer src/Callbacks.cpp:136-201,317-324) and a grouping of its entries into audio packets following the
grammar tests/compare-debug-out.py of the reference expects (154-198, 380-401)."""
import struct

import numpy as np

TYPES = {1: np.float32, 2: np.int32, 3: np.uint32, 4: np.uint8, 5: np.uint8, 6: np.int64, 7: np.uint64}


def read_dump(path):
    buf = open(path, "rb").read()
    pos = 0

    def rec():
        nonlocal pos
        (ln,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        b = buf[pos:pos + ln]
        assert len(b) == ln, "truncated record"
        pos += ln
        return b

    def kv():
        key = rec().decode()
        tid = rec()
        assert len(tid) == 1 and tid[0] in TYPES, tid
        size = rec()
        assert len(size) == 1 and size[0] == np.dtype(TYPES[tid[0]]).itemsize
        data = np.frombuffer(rec(), dtype=TYPES[tid[0]]).copy()
        return key, data, tid[0]

    assert rec() == b"ParseOggVorbis-header-v1"
    header = {}
    for want in ("decoder-name", "decoder-sample-rate", "decoder-num-channels"):
        k, v, _ = kv()
        assert k == want
        header[k] = v
    entries = []
    while pos < len(buf):
        k, v, _ = kv()
        assert k == "entry-name", k
        name = v.tobytes().decode()
        k, v, tid = kv()
        ch = -1
        if k == "entry-channel":
            ch = int(v[0])
            k, v, tid = kv()
        assert k == "entry-data", k
        entries.append((name, ch, v, tid))
    return header, entries


def split_packets(entries, channels):
    """-> (setup entries, [packet dict], pcm [C] arrays). Asserts the entry grammar."""
    i = 0
    setup = []
    while entries[i][0] != "finish_setup":
        setup.append(entries[i])
        i += 1
    i += 1
    packets, pcm = [], [[] for _ in range(channels)]
    cur = None
    for name, ch, v, tid in entries[i:]:
        if name == "pcm":
            assert cur is None or cur.get("finished"), "pcm entries only between packets"
            pcm[ch].append(v)
            if cur is not None:
                cur.setdefault("pcm_len", [0] * channels)[ch] += len(v)
            continue
        if name == "start_audio_packet":
            assert cur is None or cur.get("finished")
            cur = {"floor_number": {}, "ys": {}, "final_ys": {}, "flag": {}, "floor": {}, "after_residue": {}, "after_envelope": {},
                   "pcm_after_mdct": {}, "last_ch": -1}
            packets.append(cur)
            continue
        assert cur is not None and not cur.get("finished"), name
        if name == "finish_audio_packet":
            cur["finished"] = True
        elif name == "floor_number":
            cur["floor_number"][ch] = int(v[0])
            cur["last_ch"] = ch
        elif name == "floor1 ys":
            cur["ys"][cur["last_ch"]] = v
        elif name == "floor1 final_ys":
            cur["final_ys"][cur["last_ch"]] = v
        elif name == "floor1 step2_flag":
            assert tid == 5
            cur["flag"][cur["last_ch"]] = v
        elif name in ("after_residue", "after_envelope", "pcm_after_mdct"):
            assert tid == 1
            cur[name][ch] = v
        elif name in ("abs_total_pos", "expected_ending_total_pos"):
            cur[name] = int(v[0])
        elif name == "floor1 floor":
            cur["floor"][cur["last_ch"]] = v
        elif name == "floor_outputs":
            assert tid == 1
            cur.setdefault("floor_outputs", {})[ch] = v
        else:
            assert name in ("floor1 fit_value unwrapped",), name
    pcm = [np.concatenate(c) if c else np.zeros(0, np.float32) for c in pcm]
    return setup, packets, pcm
