#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: audio packets/s of the batched spectral-synthesis hot path.

A step = one pass of the hot path (layout + floor unwrap + fused floor-render/coupling/IMDCT/window/overlap-add)
over one device-resident synthetic batch: BASELINE config 3, 65 536 stereo packets, blocksize 2048, as 64
streams x 1024 packets, per GPU (weak scaling: every rank owns its own streams; no data-path
collective — streams are independent, hpp:1117-1123).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def alg_bytes_per_packet(spec, n, posts):
    """per stereo packet, residue in + coded posts in + PCM out once."""
    C = spec.channels
    return C * ((n // 2) * 4 + posts * 2 + (n // 2) * 4)


def build_batch(spec, streams, ppk, pattern, seed, device):
    """Device-resident synthetic batch; value distributions as tests/workloads.synth_batch."""
    import torch
    from parseoggvorbis_amd.binding import PACKET_DTYPE, SEGMENT_DTYPE, VSYN_SEG_RESET
    from tests.workloads import synth_batch
    C = spec.channels
    # coded floor rows: a pool of valid rows made on the host (the unwrap chain is serial per row), tiled on device
    pool_pk = 64 if pattern != "mixed" else 66
    pool = synth_batch(spec, 4, pool_pk, pattern, seed=seed, roll=False)  # every pool stream shares one block pattern
    P = streams * ppk
    pk = np.zeros(P, PACKET_DTYPE)
    seg = np.zeros(streams, SEGMENT_DTYPE)
    pool_n = len(pool["packets"]) // 4  # packets per pool stream (same flag pattern per stream when not rolled)
    reps = (ppk + pool_n - 1) // pool_n
    one = np.concatenate([pool["packets"][:pool_n]] * reps)[:ppk].copy()
    one["granule"] = -1
    if pattern == "mixed":  # re-derive consistent window flags after tiling
        lng = one["mode"] == [i for i, (bf, _) in enumerate(spec.modes) if bf][0]
        one["prev_long"] = np.where(lng, np.concatenate([[1], lng[:-1]]), 0)
        one["next_long"] = np.where(lng, np.concatenate([lng[1:], [1]]), 0)
    n_of = np.where(one["mode"] == [i for i, (bf, _) in enumerate(spec.modes) if bf][0], spec.blocksize1, spec.blocksize0)
    per_stream_floats = int((C * n_of // 2).sum())
    for s in range(streams):
        pk[s * ppk:(s + 1) * ppk] = one
        seg[s] = (s, s * ppk, ppk, VSYN_SEG_RESET, s * per_stream_floats)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    # ys rows: tile the pool's rows packet-wise (row q of every stream = pool row q mod pool_n, stream-varied)
    ys_pool = torch.from_numpy(pool["ys"].astype(np.int16)).to(device)  # [4*pool_n][C][stride]
    idx = (torch.arange(ppk, device=device) % pool_n)[None, :] + pool_n * torch.randint(0, 4, (streams, 1), device=device, generator=g)
    ys = ys_pool[idx.reshape(-1)].contiguous()
    total = streams * per_stream_floats
    u = torch.rand(total, device=device, generator=g)
    v = torch.rand(total, device=device, generator=g)
    lap = -1.5 * torch.sign(u - 0.5) * torch.log1p(-2 * (u - 0.5).abs().clamp(max=0.4999999))
    res = torch.where(v < 0.6, torch.zeros_like(lap), torch.round(lap)).contiguous()
    del u, v, lap
    plane = int(sum(int(n_of[i - 1]) // 4 + int(n_of[i]) // 4 for i in range(1, ppk))) + 64
    plane = (plane + 63) // 64 * 64
    return dict(P=P, S=streams, ppk=ppk, packets=torch.from_numpy(pk.view(np.uint8)).to(device),
                segments=torch.from_numpy(seg.view(np.uint8)).to(device), ys=ys, residue=res, plane=plane,
                host_packets=pk, host_segments=seg, per_stream_floats=per_stream_floats, n_of=n_of)


def reference_end_to_end_baseline(blob, audio_packets_per_file, seconds=10.0):
    """The REFERENCE decoder itself (its ogg_vorbis_full_read_from_memory, /root/reference/src/ParseOggVorbis.cpp:28, built from
    its own sources into oracle/_ref/libref_shim.so) on the same file: one thread, then every core of this process's share.
    Returns the cpu_baseline object, or None when oracle/_ref did not travel to this box."""
    import ctypes as C
    import threading
    from oracle import oracle_binding as ob
    if not ob.have_ref():
        return None
    lib = C.CDLL(ob.REF_SHIM)
    lib.ogg_vorbis_full_read_from_memory.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p)]
    lib.ogg_vorbis_full_read_from_memory.restype = C.c_int
    err = C.c_char_p()
    assert lib.ogg_vorbis_full_read_from_memory(blob, len(blob), C.byref(err)) == 0, err.value
    reps, t = 0, 0.0
    while t < seconds:
        c0 = time.perf_counter()
        lib.ogg_vorbis_full_read_from_memory(blob, len(blob), C.byref(err))
        t += time.perf_counter() - c0
        reps += 1
    cpu = {"value": round(audio_packets_per_file * reps / t, 1), "unit": "packets/s", "cores": 1, "kind": "reference",
           "sample": "ogg_vorbis_full_read_from_memory (the reference decoder, g++ -O2, no hooks) on tests/golden/test.stereo44khz.ogg, "
                     "%d decodes of the file (%d audio packets each) in %.1f s" % (reps, audio_packets_per_file, t)}
    ncores = max(1, min(16, os.cpu_count() or 1))
    counts = [0] * ncores
    t_end = time.perf_counter() + 4.0

    def work(k):
        e = C.c_char_p()
        while time.perf_counter() < t_end:
            lib.ogg_vorbis_full_read_from_memory(blob, len(blob), C.byref(e))
            counts[k] += 1

    c0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(ncores)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    t_all = time.perf_counter() - c0
    cpu["all_cores"] = {"value": round(audio_packets_per_file * sum(counts) / t_all, 1), "unit": "packets/s", "cores": ncores, "kind": "reference",
                        "sample": "%d threads, %d decodes of the file in %.1f s" % (ncores, sum(counts), t_all)}
    return cpu


def csrc_fingerprint():
    """sha256 over the kernel sources: a PMC summary is only quoted for the sources it was collected on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "parseoggvorbis_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "parseoggvorbis_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "parseoggvorbis_amd", "csrc", "*.inc")) + [os.path.join(ROOT, "include", "vorbis_synth_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def run_config5(args, rank, world, local, device):
    """BASELINE config 5: real files end to end. Every rank decodes its own shard of files — here `files_per_gpu`
    replicas of the stereo fixture, the only real stereo file there is offline — with the product's corpus decoder
    (parseoggvorbis_amd/host/CorpusDecoder: entropy worker threads -> merged multi-file submits on this rank's GPU). No data-path
    collective: files are independent. Checked: every replica yields the granule-derived frame count and the same checksum."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from parseoggvorbis_amd import sharding
    here = os.path.dirname(os.path.abspath(__file__))
    lib = C.CDLL(os.path.join(here, "parseoggvorbis_amd", "host", "libparseoggvorbis_amd.so"))
    lib.ogg_vorbis_decode_corpus.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
    lib.ogg_vorbis_decode_corpus.restype = C.c_int
    lib.ogg_vorbis_decode_corpus_s16.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                                 C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
    lib.ogg_vorbis_decode_corpus_s16.restype = C.c_int
    blob = open(os.path.join(here, "tests", "golden", "test.stereo44khz.ogg"), "rb").read()
    gold = np.load(os.path.join(here, "tests", "golden", "test.stereo44khz.npz"))
    want_frames = int(gold["pcm"].shape[-1])  # the reference decoder's total for this file (granule-derived)
    n = args.files_per_gpu
    cores = len(RANK_CPUS) if world > 1 else (os.cpu_count() or 16)
    share = max(2, min(16, cores if world > 1 else cores // max(1, world)))   # this rank's cores: one entropy worker each; the feeders mostly wait on the GPU
    feeders = args.feeders or (3 if share >= 12 else (2 if share >= 6 else 1))
    threads = args.host_threads or share
    datas = (C.c_char_p * n)(*([blob] * n))
    lens = (C.c_size_t * n)(*([len(blob)] * n))
    frames = (C.c_uint64 * n)()
    sums = (C.c_double * n)()
    ok = (C.c_uint8 * n)()
    stats = (C.c_double * 8)()
    err = C.c_char_p()

    def step():
        if args.pcm_s16:  # the PCM leaves the device as interleaved int16 (half the bytes over the bus); no per-file digest in this form
            rc = lib.ogg_vorbis_decode_corpus_s16(datas, lens, n, threads, feeders, 64, local, frames, ok, None, None, stats, C.byref(err))
        else:
            rc = lib.ogg_vorbis_decode_corpus(datas, lens, n, threads, feeders, 64, local, frames, sums, ok, None, None, stats, C.byref(err))
        assert rc == 0, err.value

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    fr = np.frombuffer(frames, np.uint64)
    sm = np.frombuffer(sums, np.float64)
    assert np.frombuffer(ok, np.uint8).all(), "a replica failed"
    assert (fr == want_frames).all(), "frame count differs from the granule-derived total of the fixture"
    assert args.pcm_s16 or (sm == sm[0]).all(), "replicas are not bit-identical"
    packets = int(round(stats[6]))
    dt, total, extra = sharding.aggregate(dt, packets, device, extra_sum=(stats[1], stats[2], float(fr.sum())))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = reference_end_to_end_baseline(blob, packets // n)
    if rank == 0:
        line = {"metric": "audio packets/sec", "value": round(total * args.steps / dt, 1), "unit": "packets/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "tests/golden/test.stereo44khz.ogg replicated",
                "config": {"workload": "config5: real files end to end, %d files (%d audio packets) per GPU, %d entropy threads + %d feeders "
                                       "per rank%s" % (n, packets, threads, feeders, ", int16 PCM" if args.pcm_s16 else ""), "packets_per_gpu": packets,
                           "parallelism": "files sharded over %d GPU(s), no data-path collective" % world},
                # end to end this workload is bound by the host's entropy decode (the GPU is busy a few % of the time): no kernel
                # roofline applies; the kernel workloads carry those
                # end to end this workload is bound outside the kernels (the GPU is busy a few % of the time): the host's entropy decode
                # (entropy_cpu_s_per_step / threads of the step) and the PCIe copy of the decoded PCM back to the host; no kernel
                # roofline applies, the kernel workloads carry those
                "roofline": {"bound": "host", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                             "note": "host entropy decode (Ogg paging + Huffman/VQ bit parse: %.0f packets/s per entropy thread while it runs) and the "
                                     "PCM copy to the host (%.1f GB/s device-to-host%s) bound the pipeline" % (
                                         total / max(extra[0], 1e-9), extra[2] * 2 * (2 if args.pcm_s16 else 4) * args.steps / dt / 1e9,  # (the fixture is stereo)
                                         ", int16" if args.pcm_s16 else ", f32")},
                "cpu_baseline": cpu,
                "packets_per_s_per_host_thread": round(total * args.steps / dt / max(1, threads * world), 1),
                "host_placement": {"rank0_cpus": _fmt_cpus(RANK_CPUS), "rank0_numa_node": RANK_NUMA, "pinned": world > 1,
                                   "note": "every rank pins itself to its GPU's NUMA-local cores before the first GPU call (sysfs topology); "
                                           "the whole-job rate is packets_per_s_per_host_thread x entropy threads per rank x ranks while the "
                                           "host cores last"},
                "realtime_factor": round(extra[2] * args.steps / dt / 44100.0, 1),
                "entropy_cpu_s_per_step": round(extra[0] / world, 3), "gpu_call_s_per_step": round(extra[1] / world, 3),
                "replicas_bit_identical": None if args.pcm_s16 else True, "frames_per_file": want_frames}
        print(json.dumps(line))


RANK_CPUS, RANK_NUMA = [], None


def _fmt_cpus(cpus):
    """[0,1,2,3,8] -> '0-3,8'"""
    out, i = [], 0
    cpus = sorted(cpus)
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def _parse_cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def rank_cpu_set(local, local_world):
    """The cores this rank's host threads (entropy workers, feeders, pinned-buffer first touch) should run on: the cores of the NUMA
    node its GPU hangs off, shared evenly by the ranks whose GPUs sit on the same node; without topology information, an even
    contiguous share of the cores the process may use. Read from sysfs only (KFD topology -> PCI address -> local_cpulist): no GPU
    call is made, so this can run before anything initialises the device. -> (sorted core list, numa node or None)."""
    allowed = sorted(os.sched_getaffinity(0))
    if local_world <= 1:
        return allowed, None
    gpus = []  # (kfd node order) -> pci address
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        for nd in sorted(os.listdir(base), key=int):
            props = dict(l.split(None, 1) for l in open(os.path.join(base, nd, "properties")).read().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0:
                loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
                gpus.append("%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
        if vis and all(v.strip().isdigit() for v in vis.split(",")):
            gpus = [gpus[int(v)] for v in vis.split(",") if int(v) < len(gpus)]
    except Exception:
        gpus = []
    node_of, cpus_of = {}, {}
    for i, bdf in enumerate(gpus):
        try:
            node_of[i] = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
            cpus_of[i] = [c for c in _parse_cpulist(open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read()) if c in allowed]
        except Exception:
            pass
    if local in cpus_of and cpus_of[local]:
        peers = sorted(i for i in range(local_world) if node_of.get(i, -2) == node_of[local])  # ranks whose GPUs share the node
        k, n = peers.index(local), len(peers)
        cores = cpus_of[local]
        per = max(1, len(cores) // n)
        mine = cores[k * per:(k + 1) * per] if k < n - 1 else cores[k * per:]
        if mine:
            return mine, node_of[local]
    per = max(1, len(allowed) // local_world)
    return allowed[local * per:(local + 1) * per] or allowed, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--feature-taps", action="store_true",
                    help="config3/config4: also write the feature taps of SURVEY 8 f-4 (rendered floor curve u16 + unwrapped posts) — the tap "
                         "variant of the fused kernel; +4 KB per stereo long packet")
    ap.add_argument("--pcm-s16", action="store_true",
                    help="config3/config4: append the PCM post-stage (planar f32 -> interleaved int16, SURVEY 8 f-3) to every step")
    ap.add_argument("--workload", default="config3", choices=["config3", "config4", "config2", "config3_vq", "config5"],
                    help="config3_vq: config 3 with the residue given as VQ entry numbers (device VQ stage, SURVEY 8 f-1); "
                         "config5: real .ogg files end to end (host entropy threads + GPU), files sharded over the ranks; use few steps, "
                         "e.g. --steps 3 --warmup 1 (one step = one pass over the rank's files)")
    ap.add_argument("--files-per-gpu", type=int, default=10640, help="config5: replicas of the stereo fixture per rank (94 audio packets each)")
    ap.add_argument("--feeders", type=int, default=0, help="config5: feeder threads per rank (pack, GPU call, deliver; 0: 3 on a 16-core share)")
    ap.add_argument("--host-threads", type=int, default=0, help="config5: entropy worker threads per rank (0: this rank's share of the cores, at most 16)")
    ap.add_argument("--blocksizes", default="256,2048", help="config3/config4 with another blocksize pair, e.g. 128,1024 (diagnostic: the "
                                                             "headline configurations are 256/2048)")
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--packets-per-stream", type=int, default=0)
    ap.add_argument("--vq-books", default="synthetic", choices=["synthetic", "fixture"],
                    help="config3_vq: codebooks and entry streams — a synthetic setup in the fixture's shape, or the stereo fixture's own "
                         "codebooks with its long packets' classifications / entry numbers replicated (tests/golden/test.stereo44khz.ogg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="diagnostic builds only (VSYN_KNOCKOUT): do not gate on the oracle spot check")
    ap.add_argument("--staged", action="store_true", help="time the staged kernels instead of the fused one")
    ap.add_argument("--no-steady", action="store_true", help="skip the longer run reported as steady_state beside a short timed region")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="diagnostic: no HIP events around the dominant kernel (roofline.achieved is then null)")
    ap.add_argument("--parity-gather", action="store_true",
                    help="after the timed region: the small parity configuration of SURVEY 8e — every rank synthesises its share of a tiny "
                         "seeded batch, the PCM is gathered over the process group (sharding.gather_pcm) and rank 0 checks every stream "
                         "against the oracle")
    ap.add_argument("--no-overlap", action="store_true",
                    help="submit without VSYN_SUBMIT_INPUTS_READY (diagnostic: nothing of a submit may overlap the previous one)")
    ap.add_argument("--hidden-pre-kernels", action="store_true",
                    help="A/B: prepare every batch with the chained layout + unwrap kernels on the library's internal stream, hidden beside "
                         "the previous submit's synthesis kernel (the default of rounds 1-3) instead of the preparation kernel on the "
                         "caller's stream")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rank-local host placement, before anything touches the GPU: this rank's threads (and the page-locked buffers they first touch)
    # stay on the cores of its GPU's NUMA node. Matters for config 5, which is host-bound; harmless elsewhere.
    global RANK_CPUS, RANK_NUMA
    RANK_CPUS, RANK_NUMA = rank_cpu_set(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if world > 1:
        try:
            os.sched_setaffinity(0, RANK_CPUS)
        except OSError:
            pass
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    # Rehearsal knobs for a ONE-GPU box (never set by the driver): BENCH_FORCE_DEVICE=0 puts every rank on that device and
    # BENCH_DIST_BACKEND=gloo carries the three tiny collectives (RCCL refuses two ranks on one GPU), so that the N > 1 code path
    # — setup broadcast, barriers, max-clock / summed-count reduction — can be run for real before a multi-GPU node does it.
    if os.environ.get("BENCH_FORCE_DEVICE"):
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, "launch with WORLD_SIZE == --gpus (one process per GPU)"

    if args.workload == "config5":
        run_config5(args, rank, world, local, device)
        if world > 1:
            dist.destroy_process_group()
        return

    from parseoggvorbis_amd.binding import Synth, VSYN_SUBMIT_STAGED, VSYN_SUBMIT_INPUTS_READY
    from tests.workloads import fixture_like_spec

    # rank 0 owns the stream setup; ONE broadcast of the (tiny) setup block splits the job, then ranks are independent
    from parseoggvorbis_amd import sharding
    bs0, bs1 = (int(v) for v in args.blocksizes.split(","))
    spec = fixture_like_spec(2 if args.workload != "config2" else 1, bs0, bs1) if rank == 0 else fixture_like_spec(1, 64, 64)
    spec = sharding.broadcast_spec(spec, device, src=0)

    stream = torch.cuda.current_stream().cuda_stream
    # the synthetic descriptors are resident and final before the timed region: consecutive submits may overlap their
    # pre-kernels with the previous synthesis kernel (every kernel of every step still runs inside the timed region)
    flags = VSYN_SUBMIT_STAGED if args.staged else (0 if args.no_overlap else VSYN_SUBMIT_INPUTS_READY)
    if args.hidden_pre_kernels and not args.staged:
        from parseoggvorbis_amd.binding import VSYN_SUBMIT_PRE_KERNELS
        flags = VSYN_SUBMIT_INPUTS_READY | VSYN_SUBMIT_PRE_KERNELS

    if args.workload == "config2":
        n, count = 256, 4096
        gpu = Synth(spec, device=local, max_streams=1)
        g = torch.Generator(device=device)
        g.manual_seed(1234 + rank)
        x = (torch.randn((count, n // 2), device=device, generator=g) * 0.05).contiguous()
        y = torch.zeros((count, n), device=device)
        step = lambda: gpu.imdct_device(n, count, x.data_ptr(), y.data_ptr(), stream)
        units, bytes_per_unit, wl = count, 1536, "config2: 4096 mono packets, blocksize 256, IMDCT only"
        b = None
    else:
        pattern = "long" if args.workload in ("config3", "config3_vq") else "mixed"
        ppk = args.packets_per_stream or (1024 if pattern == "long" else 512)
        b = build_batch(spec, args.streams, ppk, pattern, 1234 + rank, device)
        gpu = Synth(spec, device=local, max_streams=args.streams)
        pcm = torch.zeros((b["S"], spec.channels, b["plane"]), device=device)
        emit = torch.zeros(b["P"], dtype=torch.int32, device=device)
        tap_tuple = None
        if args.feature_taps:
            d_curve = torch.zeros(b["residue"].numel(), dtype=torch.int16, device=device)
            d_final = torch.zeros(b["ys"].numel(), dtype=torch.int16, device=device)
            tap_tuple = (None, None, d_final.data_ptr(), d_curve.data_ptr())
        step = lambda: gpu.submit_device(b["P"], b["packets"].data_ptr(), b["S"], b["segments"].data_ptr(), b["ppk"],
                                         b["ys"].data_ptr(), b["residue"].data_ptr(), pcm.data_ptr(), b["plane"],
                                         emit.data_ptr(), tap_tuple, flags, stream)
        vq_entries_per_packet = None
        if args.workload == "config3_vq":
            # same batch, but "after_residue" is rebuilt on the device from classification + entry numbers
            from tests.workloads import synthetic_vq_spec, synth_vq_packet
            from parseoggvorbis_amd.binding import VQ_PACKET_DTYPE
            rng = np.random.default_rng(77 + rank)
            if args.vq_books == "fixture":
                # the host decoder's entropy half over the real file (tests/host_entropy_dump.cpp): real codebooks (the 8 x 6561
                # lattice book among them), real class / entry statistics
                import subprocess
                import tempfile
                from tests.workloads import GOLDEN, build_probe, read_entropy_dump
                td = tempfile.mkdtemp()
                probe = build_probe(td)
                subprocess.run([probe, os.path.join(GOLDEN, "test.stereo44khz.ogg"), os.path.join(td, "d.bin")], check=True, capture_output=True)
                d = read_entropy_dump(os.path.join(td, "d.bin"))
                vqs = d["vq_spec"]
                fp = d["vq_packets"]
                pool = []
                for k in range(d["P"]):
                    if int(d["packets"]["mode"][k]) != 1:
                        continue
                    c0, e0, ne = int(fp["cls_off"][k]), int(fp["entry_off"][k]), int(fp["num_entries"][k])
                    c1 = int(fp["cls_off"][k + 1]) if k + 1 < d["P"] else d["cls"].size
                    pool.append((d["cls"][c0:c1].copy(), d["entries"][e0:e0 + ne].copy()))
            else:
                vqs = synthetic_vq_spec(spec.channels, spec.blocksize1)
                pool = [synth_vq_packet(vqs, 1, spec.channels, spec.blocksize1 // 2, 3, rng) for _ in range(256)]
            gpu.attach_vq(vqs)
            pick = rng.integers(0, len(pool), b["P"])
            cls = np.concatenate([pool[i][0] for i in pick])
            ent = np.concatenate([pool[i][1] for i in pick])
            vqp = np.zeros(b["P"], VQ_PACKET_DTYPE)
            vqp["num_entries"] = [pool[i][1].size for i in pick]
            vqp["entry_off"] = np.concatenate([[0], np.cumsum(vqp["num_entries"][:-1], dtype=np.uint64)])
            vqp["cls_off"] = np.concatenate([[0], np.cumsum([pool[i][0].size for i in pick][:-1])])
            d_vqp = torch.from_numpy(vqp.view(np.uint8)).to(device)
            d_cls = torch.from_numpy(cls).to(device)
            d_ent = torch.from_numpy(ent.view(np.int16)).to(device)
            d_res = [torch.zeros_like(b["residue"]) for _ in range(2)]  # double buffered: submits overlap
            vq_entries_per_packet = float(ent.size) / b["P"]
            count = [0]

            def step():
                count[0] += 1
                gpu.submit_device_vq(b["P"], b["packets"].data_ptr(), b["S"], b["segments"].data_ptr(), b["ppk"], b["ys"].data_ptr(),
                                     d_vqp.data_ptr(), d_cls.data_ptr(), cls.size, d_ent.data_ptr(), ent.size,
                                     d_res[count[0] & 1].data_ptr(), pcm.data_ptr(), b["plane"], emit.data_ptr(), flags, stream)
        pcm_stage_ms = None
        if args.pcm_s16:
            from parseoggvorbis_amd.binding import VSYN_PCM_S16
            d_s16 = torch.zeros((b["S"], b["plane"], spec.channels), dtype=torch.int16, device=device)
            inner = step

            def step():
                inner()
                gpu.pcm_interleave_device(VSYN_PCM_S16, pcm.data_ptr(), b["plane"], d_s16.data_ptr(), b["plane"], None, stream)
        units = b["P"]
        posts = {0: len(spec.floors[0][1]), 1: len(spec.floors[1][1])}
        lng_frac = float((b["n_of"] == spec.blocksize1).mean())
        # algorithmic bytes / packet: residue in + coded posts in + emitted PCM out
        in_b = float(np.mean([spec.channels * ((int(n) // 2) * 4 + posts[int(n == spec.blocksize1)] * 2) for n in b["n_of"]]))
        out_b = spec.channels * 4.0 * (b["plane"] - 64) / ppk
        bytes_per_unit = in_b + out_b
        if args.feature_taps:  # + the u16 curve of every bin
            bytes_per_unit += float(np.mean([spec.channels * (int(n) // 2) * 2 for n in b["n_of"]]))
        wl = ("config3: %d stereo packets, blocksize %d, %d streams x %d, floor+coupling+IMDCT+window+overlap-add"
              % (b["P"], spec.blocksize1, b["S"], ppk)) if pattern == "long" else \
             ("config4: %d stereo packets, mixed %d/%d (%.0f%% long), %d streams x %d" % (b["P"], spec.blocksize1, spec.blocksize0, 100 * lng_frac, b["S"], ppk))
        if vq_entries_per_packet is not None:
            # the roofline object of this workload describes the residue VQ kernel: entry + classification numbers and
            # the two 16/32-byte descriptors in, the rebuilt residue out
            bytes_per_unit = vq_entries_per_packet * 2 + float(cls.size) / b["P"] + 16 + 32 + spec.channels * (spec.blocksize1 // 2) * 4
            wl = ("config3_vq: config 3 with the residue as VQ entry numbers (%.0f entries/packet, %s); "
                  "residue VQ kernel + synthesis" % (vq_entries_per_packet, "synthetic format-2 setup" if args.vq_books == "synthetic"
                                                     else "the stereo fixture's codebooks and long-packet entry streams"))

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    fl, bad = gpu.sync_status(stream)
    assert fl == 0, "device flagged the synthetic batch: 0x%x at packet %d" % (fl, bad)

    gpu.profile(0 if args.no_kernel_timing else {"config4": 2, "config3_vq": 3}.get(args.workload, 1))
    gpu.profile_read()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms, launches, kern_name = gpu.profile_read()
    gpu.profile(0)

    dt, total_units, _ = sharding.aggregate(dt, units, device)  # max clock over ranks, summed packet count

    # Beside the contract's number: the same step over a longer stretch, timed right after it. A short run that starts on an idle
    # GPU sits in a clock-management transient (2-3 ms after the first launch the kernels run 5-10 % slower for ~5 ms:
    # profiles/r02_experiments/step_anatomy_kernel_trace.txt), so K = 20 reads worse than the rate a long job sees. Never `value`.
    steady = None
    if args.steps < 100 and not args.no_steady:
        ks = 200
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ks):
            step()
        torch.cuda.synchronize()
        barrier()
        dts, tot_s, _ = sharding.aggregate(time.perf_counter() - t1, units, device)
        steady = {"steps": ks, "value": round(tot_s * ks / dts, 1), "ms_per_step": round(dts / ks * 1e3, 4),
                  "note": "the same step, %d more of them timed after the contract's %d (not part of `value`)" % (ks, args.steps)}

    # parity spot check against the CPU oracle on the first streams of this rank's batch (not timed)
    max_err, cpu = None, None
    if b is not None and rank == 0:
        from oracle.oracle_binding import OracleSynth
        ns = min(2, b["S"])
        hp = b["host_packets"][:ns * b["ppk"]]
        hs = b["host_segments"][:ns].copy()
        hy = b["ys"][:ns * b["ppk"]].cpu().numpy().view(np.uint16)
        hr = b["residue"][:ns * b["per_stream_floats"]].cpu().numpy()
        if vq_entries_per_packet is not None:
            # the synthesis consumed the residue the VQ kernel rebuilt: check that against the oracle's VQ stage on a few
            # packets, then feed the synthesis oracle with it
            from oracle import oracle_binding as ob
            hr = d_res[count[0] & 1][:ns * b["per_stream_floats"]].cpu().numpy()
            per_pk = spec.channels * (spec.blocksize1 // 2)
            for q in (0, 1, ns * b["ppk"] - 1):
                e0, ne, c0 = int(vqp["entry_off"][q]), int(vqp["num_entries"][q]), int(vqp["cls_off"][q])
                c1 = int(vqp["cls_off"][q + 1]) if q + 1 < len(vqp) else cls.size
                rc_o, want_r = ob.residue_vq(vqs, 1, spec.channels, spec.blocksize1 // 2, 3, cls[c0:c1], ent[e0:e0 + ne])
                assert args.no_check or (rc_o == 0 and np.array_equal(want_r.view(np.uint32), hr[q * per_pk:(q + 1) * per_pk].view(np.uint32))), q
        orc = OracleSynth(spec, ns)
        want = orc.submit_host(hp, hs, hy, hr, b["plane"])
        got = pcm[:ns].cpu().numpy()
        max_err = float(np.abs(got - want["pcm"]).max())
        pcm_peak = float(np.abs(want["pcm"]).max())
        assert np.array_equal(emit[:ns * b["ppk"]].cpu().numpy().astype(np.uint32), want["emit_len"])
        # the north-star gate, absolute (compare-debug-out.py:90): a numerically broken kernel must not print a headline line
        assert args.no_check or max_err < 1e-5, "PCM max |err| vs oracle %.3g at peak %.3f" % (max_err, pcm_peak)
        if world == 1 and not args.no_cpu_baseline:
            # CPU baseline: the oracle (a port of the reference's arithmetic), single thread, bounded sample
            reps, t_cpu = 0, 0.0
            while t_cpu < 12.0:  # a bounded sample: ~12 s of single-thread CPU work
                c0 = time.perf_counter()
                orc.submit_host(hp, hs, hy, hr, b["plane"])
                t_cpu += time.perf_counter() - c0
                reps += 1
            cpu = {"value": round(len(hp) * reps / t_cpu, 1), "unit": "packets/s", "cores": 1, "kind": "port",
                   "sample": "%d streams x %d packets of the same batch, oracle/liboracle.so (gcc -O2), %d repetitions, %.1f s"
                             % (ns, b["ppk"], reps, t_cpu)}
            # the same port on every core of this process's share (packets of different streams are independent: one oracle
            # instance per thread, the C calls release the GIL) — SURVEY 8d asks for the all-cores number beside the single-thread one
            try:
                import threading
                ncores = max(1, min(16, os.cpu_count() or 1))
                orcs = [OracleSynth(spec, ns) for _ in range(ncores)]
                counts = [0] * ncores
                t_end = time.perf_counter() + 4.0

                def work(k):
                    while time.perf_counter() < t_end:
                        orcs[k].submit_host(hp, hs, hy, hr, b["plane"])
                        counts[k] += 1

                c0 = time.perf_counter()
                th = [threading.Thread(target=work, args=(k,)) for k in range(ncores)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                t_all = time.perf_counter() - c0
                cpu["all_cores"] = {"value": round(len(hp) * sum(counts) / t_all, 1), "unit": "packets/s", "cores": ncores, "kind": "port",
                                    "sample": "%d threads, %d passes over the same sample in %.1f s" % (ncores, sum(counts), t_all)}
            except Exception as exc:
                cpu["all_cores"] = {"error": str(exc)[:200]}
            # the reference's own src/mdct.cpp, built from its sources into oracle/_ref (when present on this box): IMDCT only
            try:
                from oracle import oracle_binding as ob
                if ob.have_ref():
                    n = spec.blocksize1
                    xin = np.ascontiguousarray(hr[:4096 * (n // 2)].reshape(-1, n // 2))
                    yout = np.zeros((xin.shape[0], n), np.float32)
                    r_reps, r_t = 0, 0.0
                    while r_t < 3.0:
                        c0 = time.perf_counter()
                        ob.ref().ref_mdct_backward_batch(n, xin.shape[0], xin.ctypes.data, yout.ctypes.data)
                        r_t += time.perf_counter() - c0
                        r_reps += 1
                    cpu["reference_mdct_backward"] = {"us_per_block": round(r_t / (r_reps * xin.shape[0]) * 1e6, 3), "n": n, "cores": 1,
                                                      "kind": "reference", "sample": "%d blocks x %d repetitions, oracle/_ref/libref_shim.so "
                                                      "(the reference's src/mdct.cpp, g++ -O2)" % (xin.shape[0], r_reps)}
            except Exception as exc:  # the baseline legs never take the bench down
                cpu["reference_mdct_backward"] = {"error": str(exc)[:200]}
    elif b is None and rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_binding as ob
        hx = x.cpu().numpy()
        want = ob.imdct(256, hx)
        max_err = float(np.abs(y.cpu().numpy() - want).max())
        pcm_peak = float(np.abs(want).max())
        assert args.no_check or max_err < 1e-5, "IMDCT max |err| vs oracle %.3g at peak %.3f" % (max_err, pcm_peak)
        reps, t_cpu = 0, 0.0
        while t_cpu < 5.0:
            c0 = time.perf_counter()
            ob.imdct(256, hx)
            t_cpu += time.perf_counter() - c0
            reps += 1
        cpu = {"value": round(4096 * reps / t_cpu, 1), "unit": "packets/s", "cores": 1, "kind": "port",
               "sample": "the same 4096 blocks, %d repetitions, %.1f s" % (reps, t_cpu)}

    parity_gather = None
    if args.parity_gather and b is not None:
        # SURVEY 8e's small parity configuration: a tiny seeded corpus, its streams partitioned over the ranks exactly as the big
        # batch is, synthesised on each rank's GPU, then ONE gather of the PCM over the process group (RCCL all_gather on GPUs; the
        # one-GPU rehearsal carries it over gloo on host tensors) and rank 0 checks every stream of every rank against the oracle.
        from tests.workloads import synth_batch
        from oracle.oracle_binding import OracleSynth
        ps_total, ps_ppk = 2 * world + 1, 14
        small = synth_batch(spec, ps_total, ps_ppk, "mixed", seed=4242)  # the same corpus on every rank
        first, count = sharding.shard_range(ps_total, rank, world)
        sseg = small["segments"][first:first + count].copy()
        sg = Synth(spec, device=local, max_streams=ps_total)
        r = sg.submit_host(small["packets"], sseg, small["ys"], small["residue"], small["plane_stride"])
        assert r["rc"] == 0, r["flags"]
        frames = [int(r["emit_len"][int(x["first_packet"]):int(x["first_packet"]) + int(x["num_packets"])].sum()) for x in sseg]
        gdev = device if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")
        pcm_all, frames_all = sharding.gather_pcm(r["pcm"], frames, gdev)
        if rank == 0:
            want = OracleSynth(spec, ps_total).submit_host(small["packets"], small["segments"], small["ys"], small["residue"], small["plane_stride"])
            wf = [int(want["emit_len"][int(x["first_packet"]):int(x["first_packet"]) + int(x["num_packets"])].sum()) for x in small["segments"]]
            assert list(frames_all) == wf, "gathered frame counts differ from the oracle's"
            plane = min(pcm_all.shape[2], want["pcm"].shape[2])
            gerr = float(np.abs(pcm_all[:, :, :plane] - want["pcm"][:, :, :plane]).max())
            assert args.no_check or gerr < 1e-5, "gathered PCM max |err| vs oracle %.3g" % gerr
            parity_gather = {"streams": int(pcm_all.shape[0]), "ranks": world, "max_abs_err_vs_oracle": gerr,
                             "collective": "all_gather (%s)" % (dist.get_backend() if world > 1 else "single process")}

    pcm_stage = None
    if b is not None and args.pcm_s16 and rank == 0:
        # the post-stage on its own: same buffers, torch events on the launch stream
        from parseoggvorbis_amd.binding import VSYN_PCM_S16
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gpu.pcm_interleave_device(VSYN_PCM_S16, pcm.data_ptr(), b["plane"], d_s16.data_ptr(), b["plane"], None, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        frames = int(emit.sum().item())
        nbytes = frames * spec.channels * (4 + 2)
        pcm_stage = {"kernel": "vsyn_pcm_interleave_kernel", "kernel_ms": round(ms, 5), "algorithmic_bytes": nbytes,
                     "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process; the committed summary of
    # the separate rocprofv3 --pmc passes over this same command (tools/profile_round.sh -> profiles/) is reported when it
    # describes this kernel and workload
    traffic, traffic_detail = None, None
    if rank == 0 and args.workload == "config3" and not args.staged:
        try:
            import glob
            latest = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_summary.json")))[-1]
            pm = json.load(open(latest))
            hb = pm.get("long_kernel_hbm_bytes_per_launch")
            # only a summary collected on THESE kernel sources is quoted (tools/pmc_summarize.py records their fingerprint)
            if hb and kern_name in pm.get("kernels", {}) and pm.get("csrc_fingerprint") == csrc_fingerprint():
                traffic = round(hb["read_corrected"] + hb["write"])  # HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE)
                traffic_detail = {"read_corrected": round(hb["read_corrected"]), "write": round(hb["write"]),
                                  "algorithmic_bytes_per_launch": None, "source": "profiles/" + os.path.basename(latest)}
        except Exception:
            traffic, traffic_detail = None, None
    if rank == 0:
        value = total_units * args.steps / dt
        achieved = bytes_per_unit * units / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        if traffic_detail is not None:
            traffic_detail["algorithmic_bytes_per_launch"] = round(bytes_per_unit * units)
        line = {
            "metric": "audio packets/sec (blocksize 2048, stereo)" if (args.workload == "config3" and spec.blocksize1 == 2048) else "audio packets/sec",
            "value": round(value, 1), "unit": "packets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "packets_per_gpu": units, "parallelism": "streams sharded over %d GPU(s), no data-path collective" % world},
            # what bounds the dominant kernel, from its counters: config 2 is a 6 MB batch (launch + one memory round trip);
            # the residue VQ kernel spends 58 % of its wave cycles in s_waitcnt and is paced by ~2 900 instructions per packet on the CU's
            # scalar unit and issue slots (profiles/r02_b_vq_kernel_pmc.json) — its fraction of the HBM peak is reported all the same
            "roofline": {"bound": "latency" if args.workload == "config2" else ("instruction issue" if args.workload == "config3_vq" else "hbm"), "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_detail": traffic_detail,
                         "kernel": kern_name, "kernel_ms": round(kern_ms, 5), "launches": launches,
                         "algorithmic_bytes_per_packet": round(bytes_per_unit, 1)},
            "cpu_baseline": cpu, "pcm_stage_s16": pcm_stage,
            "pcm_max_abs_err_vs_oracle": max_err, "pcm_peak": None if max_err is None else pcm_peak,
            "steady_state": steady, "parity_gather": parity_gather,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
