"""Shared inputs for tests and bench: the committed golden fixtures and the seeded synthetic batches of
BASELINE.json's configs.  numpy only."""
import os

import numpy as np

from parseoggvorbis_amd.binding import PACKET_DTYPE, SEGMENT_DTYPE, VSYN_SEG_RESET, SetupSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# floor-1 X lists of the reference's fixtures (hook "floor1_unpack xs", hpp:1367)
FIX_XS_SHORT = [0, 128, 14, 4, 58, 2, 8, 28, 90]
FIX_XS_LONG_SORTED = [0, 3, 6, 10, 14, 18, 23, 28, 33, 39, 46, 55, 65, 79, 93, 111, 130, 158, 186, 220, 260, 312, 372,
                      464, 556, 650, 750, 850, 1024]


def load_golden(name):
    """-> (SetupSpec, dict of arrays, npz) for 'test.stereo44khz' / 'test.mono44khz'."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    C = int(z["channels"])
    floors = [(int(z["floor%d_mult" % f]), [int(v) for v in z["floor%d_xs" % f]]) for f in range(int(z["num_floors"]))]
    coup = [(0, 1)] if C == 2 else []
    # fixtures: mode 0 short -> mapping 0 -> floor 0, mode 1 long -> mapping 1 -> floor 1
    spec = SetupSpec(C, int(z["blocksize0"]), int(z["blocksize1"]), floors,
                     [(coup, [0] * C), (coup, [1] * C)], [(0, 0), (1, 1)])
    P = len(z["mode"])
    assert np.array_equal(z["floor_number"], np.repeat(z["mode"][:, None], C, 1))
    pk = np.zeros(P, PACKET_DTYPE)
    pk["mode"], pk["prev_long"], pk["next_long"] = z["mode"], z["prev_long"], z["next_long"]
    pk["floor_used"], pk["granule"] = z["floor_used"], z["granule"]
    seg = np.zeros(1, SEGMENT_DTYPE)
    seg["num_packets"], seg["flags"] = P, VSYN_SEG_RESET
    return spec, dict(packets=pk, segments=seg, ys=z["ys"], residue=z["residue"], pcm=z["pcm"],
                      emit_len=z["emit_len"]), z


def fixture_like_spec(channels=2, bs0=256, bs1=2048):
    """The fixtures' stream setup, generalised to other blocksizes by scaling the X lists."""
    def scale(xs, n2, base):
        out = sorted({min(n2, max(0, (x * n2) // base)) for x in xs})
        if len(out) < 2:
            out = [0, n2]
        out = [0, n2] + [x for x in out if x not in (0, n2)]
        return out
    short = scale(FIX_XS_SHORT, bs0 // 2, 128)
    long_sorted = scale(FIX_XS_LONG_SORTED, bs1 // 2, 1024)
    rng = np.random.default_rng(99)
    inner = long_sorted[2:]
    rng.shuffle(inner)  # header order is not sorted in real streams
    long_ = [0, bs1 // 2] + [int(v) for v in inner]
    coup = [(0, 1)] if channels >= 2 else []
    return SetupSpec(channels, bs0, bs1, [(4, short), (2, long_)],
                     [(coup, [0] * channels), (coup, [1] * channels)], [(0, 0), (1, 1)])


def _encode_ys(xs, mult, target, rng, zero_frac):
    """Coded floor-1 ys whose unwrap gives (approximately) `target` amplitudes; always in range (valid stream)."""
    rng_of = {1: 256, 2: 128, 3: 86, 4: 64}[mult]
    posts = len(xs)
    ys = np.zeros(posts, np.int64)
    fy = np.zeros(posts, np.int64)
    lo_n = [max([j for j in range(i) if xs[j] < xs[i]], key=lambda j: xs[j], default=-1) for i in range(posts)]
    hi_n = [min([j for j in range(i) if xs[j] > xs[i]], key=lambda j: xs[j], default=-1) for i in range(posts)]
    fy[0] = ys[0] = min(rng_of - 1, max(0, int(target[0])))
    fy[1] = ys[1] = min(rng_of - 1, max(0, int(target[1])))
    for i in range(2, posts):
        lo, hi = lo_n[i], hi_n[i]
        adx = xs[hi] - xs[lo]
        dy = fy[hi] - fy[lo]
        off = (abs(dy) * (xs[i] - xs[lo])) // adx
        pred = fy[lo] + off if dy >= 0 else fy[lo] - off
        d = min(rng_of - 1, max(0, int(target[i]))) - pred
        hr, lr = rng_of - pred, pred
        room = min(hr, lr) * 2
        val = 0
        if d != 0 and rng.random() >= zero_frac:
            if d > 0:
                val = 2 * d if 2 * d < room else (d + lr if hr > lr else 0)
            else:
                val = -2 * d - 1 if -2 * d - 1 < room else (hr - d - 1 if hr <= lr else 0)
        if val == 0:
            f = pred
        elif val >= room:
            f = val - lr + pred if hr > lr else pred - val + hr - 1
        else:
            f = pred - (val + 1) // 2 if val % 2 else pred + val // 2
        if not (0 <= f < rng_of):
            val, f = 0, pred
        ys[i], fy[i] = val, f
    return ys


def synth_batch(spec, streams, packets_per_stream, pattern="long", seed=1234, ylo=28, yhi=88, unused_frac=0.0,
                granule_last=False, roll=True):
    """Synthetic batch in the shape of BASELINE configs 3/4:
    residue = round(Laplace(b=1.5)) with 60 % zeros; floor amplitudes a random walk in [ylo,yhi] (step +-6)
    over the setup's X list, wrapped into coded ys.  pattern: 'long', 'short', 'mixed' (L L L S*8 repeating, rotated per
    stream) or an explicit sequence of block flags (1 = long), the same for every stream.
    Returns dict(packets, segments, ys, residue, plane_stride)."""
    rng = np.random.default_rng(seed)
    C = spec.channels
    if not isinstance(pattern, str):
        flags1 = np.asarray(pattern, np.uint8)
        assert len(flags1) == packets_per_stream
        roll = False
    elif pattern == "long":
        flags1 = np.ones(packets_per_stream, np.uint8)
    elif pattern == "short":
        flags1 = np.zeros(packets_per_stream, np.uint8)
    else:
        unit = [1, 1, 1] + [0] * 8
        flags1 = np.array((unit * (packets_per_stream // len(unit) + 1))[:packets_per_stream], np.uint8)
    long_mode = [i for i, (bf, _) in enumerate(spec.modes) if bf][0]
    short_mode = [i for i, (bf, _) in enumerate(spec.modes) if not bf][0]
    P = streams * packets_per_stream
    pk = np.zeros(P, PACKET_DTYPE)
    seg = np.zeros(streams, SEGMENT_DTYPE)
    stride = spec.ys_stride
    ys = np.zeros((P, C, stride), np.uint16)
    res_parts = []
    off = 0
    for s in range(streams):
        flags = flags1 if (not roll or pattern != "mixed") else np.roll(flags1, s % 11)
        seg[s] = (s, s * packets_per_stream, packets_per_stream, VSYN_SEG_RESET, off)
        for q in range(packets_per_stream):
            p = s * packets_per_stream + q
            lng = int(flags[q])
            mode = long_mode if lng else short_mode
            n = spec.blocksize1 if lng else spec.blocksize0
            pk[p]["mode"] = mode
            if lng:
                pk[p]["prev_long"] = flags[q - 1] if q > 0 else 1
                pk[p]["next_long"] = flags[q + 1] if q + 1 < packets_per_stream else 1
            pk[p]["granule"] = -1
            used = 0
            for c in range(C):
                if rng.random() < unused_frac:
                    continue
                used |= 1 << c
                mult, xs = spec.floors[spec.mappings[spec.modes[mode][1]][1][c]]
                order = np.argsort(xs)
                walk = np.clip(np.cumsum(rng.integers(-6, 7, len(xs))) + rng.integers(ylo, yhi), ylo, yhi)
                target = np.zeros(len(xs), np.int64)
                target[order] = walk * 2 // mult  # same dB range whatever the multiplier
                ys[p, c, :len(xs)] = _encode_ys(xs, mult, target, rng, 0.25)
            pk[p]["floor_used"] = used
            r = np.round(rng.laplace(0.0, 1.5, (C, n // 2)))
            r[rng.random((C, n // 2)) < 0.6] = 0
            res_parts.append(r.astype(np.float32).ravel())
            off += C * (n // 2)
    if granule_last:
        for s in range(streams):
            fl = flags1 if (not roll or pattern != "mixed") else np.roll(flags1, s % 11)
            sizes = np.where(fl, spec.blocksize1, spec.blocksize0)
            total = int(sum(sizes[i - 1] // 4 + sizes[i] // 4 for i in range(1, packets_per_stream)))
            last_l = int(sizes[-2] // 4 + sizes[-1] // 4) if packets_per_stream > 1 else 0
            pk[s * packets_per_stream + packets_per_stream - 1]["granule"] = total - min(37, last_l // 2)  # clipped last packet
    plane = packets_per_stream * (spec.blocksize1 // 2) + 64
    return dict(packets=pk, segments=seg, ys=ys, residue=np.concatenate(res_parts), plane_stride=plane)


def read_entropy_dump(path):
    """Parse the file written by tests/host_entropy_dump.cpp -> dict (packets, ys, residue, and in VQ mode vq_packets,
    cls, entries, residue_floats, vq_spec)."""
    from parseoggvorbis_amd.binding import PACKET_DTYPE, VQ_PACKET_DTYPE, VqSpec
    raw = open(path, "rb").read()
    P, Cn, stride, nres, bs0, bs1 = (int(x) for x in np.frombuffer(raw[:24], np.uint32))
    off = 24
    d = dict(P=P, channels=Cn, ys_stride=stride, blocksize0=bs0, blocksize1=bs1)
    d["packets"] = np.frombuffer(raw[off:off + 16 * P], PACKET_DTYPE).copy()
    off += 16 * P
    d["ys"] = np.frombuffer(raw[off:off + 2 * P * Cn * stride], np.uint16).reshape(P, Cn, stride).copy()
    off += 2 * P * Cn * stride
    d["residue"] = np.frombuffer(raw[off:off + 4 * nres], np.float32).copy()
    off += 4 * nres
    if off == len(raw):
        return d

    def u32():
        nonlocal off
        v = int(np.frombuffer(raw[off:off + 4], np.uint32)[0])
        off += 4
        return v

    assert u32() == 0x31305156
    nvq, ncls, nent, rfl = u32(), u32(), u32(), u32()
    d["vq_packets"] = np.frombuffer(raw[off:off + 16 * nvq], VQ_PACKET_DTYPE).copy()
    off += 16 * nvq
    d["cls"] = np.frombuffer(raw[off:off + ncls], np.uint8).copy()
    off += ncls
    d["entries"] = np.frombuffer(raw[off:off + 2 * nent], np.uint16).copy()
    off += 2 * nent
    d["residue_floats"] = rfl
    books = []
    for _ in range(u32()):
        dims, n, has = u32(), u32(), u32()
        tab = None
        if has:
            tab = np.frombuffer(raw[off:off + 4 * dims * n], np.float32).copy()
            off += 4 * dims * n
        books.append((dims, n, tab))
    residues = []
    for _ in range(u32()):
        r = dict(type=u32(), begin=u32(), end=u32(), partition_size=u32(), num_classifications=u32(), classwords=u32())
        nb = r["num_classifications"] * 8
        r["books"] = np.frombuffer(raw[off:off + 2 * nb], np.int16).copy()
        off += 2 * nb
        residues.append(r)
    maps = []
    for _ in range(u32()):
        ns = u32()
        mux = list(raw[off:off + Cn])
        off += Cn
        sres = list(raw[off:off + ns])
        off += ns
        maps.append((mux, sres))
    assert off == len(raw)
    d["vq_spec"] = VqSpec(books, residues, maps)
    return d


def build_probe(tmpdir, testing=False):
    """Compile tests/host_entropy_dump.cpp against the built host library (testing: the build with the fault injection of
    PARSEOGGVORBIS_TEST_FAIL_AT compiled in); returns the executable's path."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "parseoggvorbis_amd", "host")
    csrc = os.path.join(root, "parseoggvorbis_amd", "csrc")
    out = os.path.join(str(tmpdir), "host_entropy_dump" + ("_testing" if testing else ""))
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", out, os.path.join(root, "tests", "host_entropy_dump.cpp"),
                    "-L" + host, "-lparseoggvorbis_amd" + ("_testing" if testing else ""), "-L" + csrc, "-lvorbis_synth_hip", "-Wl,-rpath," + host,
                    "-Wl,-rpath," + csrc, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return out


def synth_vq_packet(vq_spec, mapping, channels, n2, used_mask, rng):
    """Random but well-formed (cls, entries) of one packet for `mapping`, in the decode order of hpp:708-760."""
    mux, sres = vq_spec.mappings[mapping]
    cls_all, ent_all = [], []
    for s in range(len(sres)):
        chans = [c for c in range(channels) if mux[c] == s]
        if not chans:
            continue
        r = vq_spec.residues[sres[s]]
        fmt2 = r["type"] == 2
        vch = 1 if fmt2 else len(chans)
        ln = len(chans) * n2 if fmt2 else n2
        parts = (min(r["end"], ln) - min(r["begin"], ln)) // r["partition_size"]
        cls = rng.integers(0, r["num_classifications"], (vch, parts)).astype(np.uint8)
        used = [True] if fmt2 else [bool((used_mask >> c) & 1) for c in chans]
        for ps in range(8):
            for pc in range(parts):
                for j in range(vch):
                    if not used[j]:
                        continue
                    book = int(r["books"][int(cls[j, pc]) * 8 + ps])
                    if book < 0:
                        continue
                    dims, n, _ = vq_spec.codebooks[book]
                    ent_all.append(rng.integers(0, n, r["partition_size"] // dims).astype(np.uint16))
        cls_all.append(cls.ravel())
    cls = np.concatenate(cls_all) if cls_all else np.zeros(0, np.uint8)
    ent = np.concatenate(ent_all) if ent_all else np.zeros(0, np.uint16)
    return cls, ent


def synthetic_vq_spec(channels=2, bs1=2048, seed=5):
    """A residue VQ setup in the shape of the stereo fixture's long-block residue (format 2, begin 0, end 1600 of 2048
    interleaved bins, partition 32, 10 classes, three cascade passes, vector lengths 8/4/2/1), with small-integer value
    tables. Mapping 0 (short blocks) and 1 (long blocks) share it. For benchmarks and tests that need codebooks without an
    .ogg file."""
    from parseoggvorbis_amd.binding import VqSpec
    rng = np.random.default_rng(seed)
    books = []
    for dims, n, amp in ((8, 256, 1), (4, 256, 2), (4, 81, 1), (2, 128, 3), (2, 64, 1), (1, 32, 8), (1, 16, 2)):
        tab = rng.integers(-amp, amp + 1, (n, dims)).astype(np.float32)
        tab[rng.random((n, dims)) < 0.5] = 0
        books.append((dims, n, tab.ravel()))
    nclass = 10
    cas = np.full((nclass, 8), -1, np.int16)
    cas[1, 0] = 0
    cas[2, 0], cas[2, 1] = 1, 4
    cas[3, 0] = 2
    cas[4, 0], cas[4, 1] = 3, 6
    cas[5, 1] = 4
    cas[6, 0], cas[6, 1], cas[6, 2] = 1, 3, 5
    cas[7, 2] = 6
    cas[8, 0], cas[8, 2] = 0, 5
    cas[9, 0], cas[9, 1], cas[9, 2] = 2, 4, 6
    n2 = bs1 // 2
    end = (channels * n2 * 25 // 32) // 32 * 32  # 1600 of 2048 for stereo 2048
    res = dict(type=2, begin=0, end=end, partition_size=32, num_classifications=nclass, classwords=2, books=cas.ravel())
    mux = [0] * channels
    return VqSpec(books, [res], [(mux, [0]), (mux, [0])])


def exotic_vq_spec(variant, channels=2, bs0=256, bs1=2048, seed=9):
    """Residue setups the fixtures do not have, for oracle-vs-device tests of the VQ stage:
    'format0'   one submap, residue format 0 (interleaved components, hpp:738-746), partition 16, vector lengths 2/4/8/16
    'format1x2' one submap holding both channels as two format-1 vectors (partition counter shared, hpp:726-757), partition 12,
                vector lengths 1/2/3/4/6 (not powers of two, partition not a multiple of 8: the general path and its tails),
                begin > 0 and end below the block
    'submaps'   channel 0 -> submap 0 (format 1), channel 1 -> submap 1 (format 0): two residues per packet"""
    from parseoggvorbis_amd.binding import VqSpec
    rng = np.random.default_rng(seed)

    def book(dims, n, amp):
        tab = rng.integers(-amp, amp + 1, (n, dims)).astype(np.float32)
        return (dims, n, tab.ravel())

    def cascade(nclass, nbooks, density):
        c = np.full((nclass, 8), -1, np.int16)
        for k in range(1, nclass):
            for ps in range(8):
                if rng.random() < density:
                    c[k, ps] = int(rng.integers(0, nbooks))
        return c.ravel()

    if variant == "format0":
        books = [book(2, 50, 3), book(4, 81, 2), book(8, 100, 1), book(16, 37, 1)]
        res = [dict(type=0, begin=0, end=bs1 // 2, partition_size=16, num_classifications=6, classwords=2, books=cascade(6, 4, 0.35))]
        maps = [([0] * channels, [0]), ([0] * channels, [0])]
    elif variant == "format1x2":
        books = [book(1, 9, 5), book(2, 25, 3), book(3, 27, 2), book(4, 40, 2), book(6, 64, 1)]
        res = [dict(type=1, begin=24, end=bs1 // 2 - 100, partition_size=12, num_classifications=7, classwords=3,
                    books=cascade(7, 5, 0.3))]
        maps = [([0] * channels, [0]), ([0] * channels, [0])]
    else:
        books = [book(1, 16, 4), book(2, 64, 2), book(4, 128, 2), book(8, 60, 1)]
        res = [dict(type=1, begin=0, end=bs1 // 2, partition_size=32, num_classifications=5, classwords=2, books=cascade(5, 4, 0.4)),
               dict(type=0, begin=8, end=bs1 // 4, partition_size=8, num_classifications=4, classwords=1, books=cascade(4, 4, 0.4))]
        mux = [i % 2 for i in range(channels)]
        maps = [(mux, [0, 1]), (mux, [0, 1])]
    # the class codebook's vector length only matters to the host's bit reader; books here always have a value table
    return VqSpec(books, res, maps)
