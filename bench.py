#!/usr/bin/env python3
"""bench.py — frames/s of encode+decode on the ScanNet-scale 1M-point frame.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under torch.distributed.run — RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment — or
     plain, in which case this process starts the N ranks itself: launch_ranks())

One step = the reference's operator contract: CompressionPipeline.compress(gop of host numpy frames as the
capturer leaves them — int16 points, float64 colours — at the three quality settings of shared/config.yaml:12-15)
+ DecompressionPipeline.decompress(container of the last quality) -> host numpy frames
(sender/encoder/codec_pipeline.py:239-267, receiver/decoder/codec_parallel.py:474-502).  `value` is that rate, PCIe
legs included; `value_hbm_resident` is the same K steps with the frame already in HBM and the reconstruction left
there (the figure the kernels' roofline refers to).  D1-PSNR / Y-PSNR of the reconstruction (metrics.py, outside
the timed region) are reported for the HIP path and for the CPU oracle; the two reconstructions must be equal.

N = 1: the C2 frame (BASELINE.json configs[1]).  N > 1: BASELINE.json configs[4]'s tiling — a fused scan of N C2
rooms side by side (N x 1M voxels), cut into 2N octree blocks of 256 x 512 x 256 voxels (tiled.cut_tiles), two per
rank: the same 1M voxels per GPU and step as at N = 1, and N = 4 is configs[4] itself (4M points in 8 blocks).
Every rank codes its two blocks as one GOP, the sub-bitstreams of all ranks are exchanged by the variable-length
all-gather of tiled.py (RCCL) INSIDE the timed region, and every rank decodes the blocks its neighbour coded, out of
the gathered bundle.  value = 1M-voxel frames of all ranks / max time over ranks.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import gc
import json
import os
import sys
import time


def _cap_thread_pools():
    """numpy's BLAS and every OpenMP runtime size their pools by the VISIBLE cores (256 on the GPU boxes) while the
    cgroup grants 16: one parallel region then spends the whole quota of a scheduling period in a few milliseconds and
    the kernel freezes every thread of the process until the period ends (cpu.stat: throttled 3.6 s of thread time
    while the workload was generated).  Pools are capped at the cgroup's share before numpy / torch are imported."""
    budget = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            budget = max(1, min(budget, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(var, str(min(budget, 16)))


_cap_thread_pools()

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"

SETTINGS = [[1.0, 0.0], [0.0, 1.0], [1, 1]]      # /root/reference/shared/config.yaml:12-15
HBM_PEAK_GBS = 8000.0                             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3                      # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def conv_algorithmic(n_out, n_in, cin, cout, k_vol, pairs):
    """bytes = 4(N_in Cin + N_out Cout) + 4 K Cin Cout + 4 P + 4 (N_out+1); flops = 2 P Cin Cout"""
    nbytes = 4 * (n_in * cin + n_out * cout) + 4 * k_vol * cin * cout + 4 * pairs + 4 * (n_out + 1)
    return nbytes, 2 * pairs * cin * cout


def cpu_baseline(wl, frame, n_sample, threads, runs=5, warmups=2):
    """SURVEY.md §8(d) protocol for the CPU leg: the oracle (oracle/codec_ref.py, the C restatement under numpy
    orchestration, OpenMP over rows) on the same frame — the WHOLE frame unless --cpu-sample asks for a spatial crop —
    `warmups` untimed runs, then the median of `runs` timed runs of encode (Q = 3) + decode (last quality), with the
    reference's stage split (E1-E7: codec_pipeline.py:218-225, D1-D6: codec_parallel.py:157-163) as medians per
    stage.  Raises on any failure: a bench line without this leg is an unmeasured line."""
    from oracle.codec_ref import Oracle
    pts = frame["points"].astype(np.int64)
    whole = n_sample <= 0 or n_sample >= pts.shape[0]
    if whole:
        sample = frame
    else:   # the points nearest (Chebyshev distance) to the median keep the surface statistics
        c = np.median(pts, axis=0)
        keep = np.argsort(np.abs(pts - c).max(axis=1), kind="stable")[:n_sample]
        sample = {"points": frame["points"][keep], "colors": frame["colors"][keep]}
    n = sample["points"].shape[0]
    o = Oracle(threads=threads)
    enc_s, dec_s, enc_st, dec_st, full = [], [], [], [], None
    for it in range(warmups + runs):
        t0 = time.perf_counter()
        out, _dbg = o.compress([dict(sample)], SETTINGS)
        t1 = time.perf_counter()
        rec = o.decompress(out[len(SETTINGS)])
        t2 = time.perf_counter()
        if it >= warmups:
            enc_s.append(t1 - t0)
            dec_s.append(t2 - t1)
            enc_st.append(dict(o.enc_times))
            dec_st.append(dict(o.dec_times))
        if whole:
            full = (out, rec)
    med = lambda v: float(np.median(np.asarray(v)))    # noqa: E731
    enc, dec = med(enc_s), med(dec_s)
    total = med([a + b for a, b in zip(enc_s, dec_s)])
    stages = {"encode": {k: med([d[k] for d in enc_st]) for k in enc_st[0]},
              "decode": {k: med([d[k] for d in dec_st]) for k in dec_st[0]}}
    return {"value": (n / 1.0e6) / total, "unit": "frames/s (1M-point equivalent, linear in points)",
            "cores": threads, "kind": "port", "runs": runs, "warmups": warmups,
            "encode_s": enc, "decode_s": dec, "total_s": total, "stages": stages,
            "sample": (f"{'the whole' if whole else 'a spatial crop of the'} bench frame ({n} points), Q=3 encode + "
                       f"decode of the last quality; median of {runs} runs after {warmups} warm-ups; oracle/ C "
                       f"restatement with OpenMP on {threads} threads")}, full


def bench_configs(pkg, wl, device, threads, steps=5, warmup=2, with_oracle=True):
    """The other configurations of BASELINE.json and the reference's own operating point, each timed on this GPU
    after the headline (outside its timed region): `warmup` untimed + `steps` timed passes of the operator contract
    (host numpy in / host numpy out, Q = 3 unless geometry-only), medians.  `equals_oracle`: containers (blob) and
    reconstruction compared with the CPU oracle's on the same input — the checker, never the thing measured; None
    where the oracle run is not affordable inside a bench run (C5: 4M points)."""
    import torch
    tiled = importlib.import_module(PKG + ".tiled")
    utils = importlib.import_module(PKG + ".utils")
    runtime = importlib.import_module(PKG + ".runtime")
    oracle = None
    if with_oracle:
        from oracle.codec_ref import Oracle
        oracle = Oracle(threads=threads)
    q_dec = len(SETTINGS)
    med = lambda v: float(np.median(np.asarray(v)))    # noqa: E731
    res = {}

    def codec_case(name, frames, what, check):
        enc = pkg.CompressionPipeline(SETTINGS, device=device, slots=1)
        dec = pkg.DecompressionPipeline(device=device, slots=1, output="numpy")
        e_ms, d_ms = [], []
        for it in range(warmup + steps):
            gop = wl.gop([dict(f) for f in frames])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out, side = enc.compress(gop)
            t1 = time.perf_counter()
            rec, _ = dec.decompress(out[q_dec])
            t2 = time.perf_counter()
            if it >= warmup:
                e_ms.append(1e3 * (t1 - t0))
                d_ms.append(1e3 * (t2 - t1))
        n_pts = int(sum(f["points"].shape[0] for f in frames))
        same = None
        if check and oracle is not None:
            o_out, _ = oracle.compress([dict(f) for f in frames], SETTINGS)
            o_rec = oracle.decompress(o_out[q_dec])
            same = bool(all(out[q] == o_out[q] for q in range(1, q_dec + 1)) and len(rec) == len(o_rec) and
                        all(np.array_equal(a["points"], b["points"]) and np.array_equal(a["colors"], b["colors"])
                            for a, b in zip(rec, o_rec)))
            assert same, f"{name}: HIP path and CPU oracle disagree"
        res[name] = {"workload": what, "frames": len(frames), "points": n_pts, "encode_ms": med(e_ms), "decode_ms": med(d_ms),
                     "gops_per_s": 1e3 / (med(e_ms) + med(d_ms)), "frames_per_s": 1e3 * len(frames) / (med(e_ms) + med(d_ms)),
                     "bpp": [float(b) for b in side["gop_info"]["bpp"]], "steps": steps, "equals_oracle": same}
        del enc, dec

    codec_case("C1", [wl.sphere_shell(64, 25.2, seed=1)],
               "BASELINE.json configs[0]: 64^3 sphere shell (~8k voxels), one frame", True)
    # C3: geometry only — utils.gpcc_encode / gpcc_decode (shared/utils.py:169-240) from host int16 points to the blob on
    # the host and back to host coordinates: upload, Morton keys, sort, octree levels + occupancy coder on the GPU
    # (blob version 2, csrc/octree2.hip), decode on the GPU, download
    sweep = wl.lidar_sweep()
    pts = sweep["points"]
    rt = runtime.Runtime(device)
    with rt:
        e_ms, d_ms, k_ms = [], [], []
        for it in range(warmup + steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            coords = np.concatenate([np.zeros((pts.shape[0], 1), np.int32), pts.astype(np.int32)], 1)
            keys = rt.morton_keys(rt.to_device(coords))
            rt.sort_pairs(keys)
            torch.cuda.synchronize()
            tk = time.perf_counter()
            blob = utils.gpcc_encode(keys, None, 0, keys.shape[0], 0)
            t1 = time.perf_counter()
            dec_pts = utils.gpcc_decode(blob, 1)
            t2 = time.perf_counter()
            if it >= warmup:
                e_ms.append(1e3 * (t1 - t0))
                d_ms.append(1e3 * (t2 - t1))
                k_ms.append(1e3 * (tk - t0))
        same = None
        if oracle is not None:
            same = bool(blob == oracle.octree_encode(pts.astype(np.int32), 32768) and
                        np.array_equal(dec_pts, oracle.octree_decode(blob)))
            assert same, "C3: HIP blob / decoded points differ from the oracle's"
        res["C3"] = {"workload": "BASELINE.json configs[2]: KITTI-like LiDAR sweep, geometry-only octree occupancy coding "
                                 "(lossless), host int16 points -> blob -> host int32 points",
                     "frames": 1, "points": int(pts.shape[0]), "encode_ms": med(e_ms), "decode_ms": med(d_ms),
                     "encode_ms_keys_and_sort": med(k_ms), "frames_per_s": 1e3 / (med(e_ms) + med(d_ms)),
                     "bpp": [8.0 * len(blob) / pts.shape[0]], "blob_version": int(blob[1]), "steps": steps, "equals_oracle": same}
    rt.close()
    codec_case("C4", [wl.body(800_000)], "BASELINE.json configs[3]: 8iVFB-like dense body with RGB, 800k voxels, one frame", True)
    tiles, _ = tiled.cut_tiles(wl.fused_scan(4_000_000), (512, 512, 256))
    codec_case("C5_one_gpu", tiles, "BASELINE.json configs[4] on ONE GPU: 4M-point fused scan cut into 8 octree blocks of 500k "
                                    "voxels, the 8 blocks coded as the 8 frames of one GOP (the N-GPU form deals them to ranks)", False)
    del tiles
    with np.load(os.path.join(ROOT, "tests", "golden", "zed_seq25.npz")) as f:
        zed = [{"points": f[f"points_{i}"], "colors": f[f"colors_u8_{i}"].astype(np.float64) / 255.0} for i in range(5)]
    codec_case("zed_gop5", zed, "the reference's operating point: GOP of 5 recorded ZED frames (evaluation/data/test_sequence, the "
                                "first five its encoder service samples), Q = 3; the reference logs 841 ms encode / 715 ms decode "
                                "per 66k-point GOP on a Jetson AGX Orin (BASELINE.md: other hardware, other weights)", True)
    return res


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it (the driver's command): start the N ranks here, one
    fresh child process per GPU, with the environment torch.distributed.run would give them.  This parent never
    touches the GPU (torch.cuda.device_count() does not initialise it) and never execs: it relays the children's
    output (rank 0 prints the JSON line), waits for all of them and returns the worst exit code.  A rank that fails
    is not restarted; the others are then ended by their own PIDs, since they would wait for it in a barrier."""
    import socket
    import subprocess
    backend = os.environ.get("PCC_BENCH_BACKEND", "nccl")
    import torch
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs for one RCCL rank each, this node shows {have}; no line printed",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst, failed_at = 0, None
    live = dict(enumerate(procs))
    t_start, limit = time.time(), float(os.environ.get("PCC_BENCH_RANK_TIMEOUT", "1500"))
    while live:
        if failed_at is None and time.time() - t_start > limit:   # a rank that never returns (a collective that hangs)
            log(f"bench.py: ranks still running after {limit:.0f} s: ending them, no line printed")
            worst, failed_at = 3, time.time()
            for q in live.values():
                q.terminate()
        for r, p in list(live.items()):
            rc = p.poll()
            if rc is None:
                continue
            del live[r]
            if rc != 0:
                log(f"bench.py: rank {r} ended with code {rc}")
                worst = worst or rc
                if failed_at is None:
                    failed_at = time.time()
                    for q in live.values():
                        q.terminate()
        if failed_at is not None and time.time() - failed_at > 20.0:
            for q in live.values():
                q.kill()
        time.sleep(0.05)
    return worst if worst > 0 else (1 if worst else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample", type=int, default=0, help="points of the CPU baseline's sample (0 = the whole frame)")
    ap.add_argument("--cpu-runs", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--inflight", type=int, default=3, help="also report throughput with this many GOPs in flight (0/1 = skip)")
    ap.add_argument("--no-configs", action="store_true", help="skip the C1 / C3 / C4 / C5-on-one-GPU / ZED-GOP entries")
    ap.add_argument("--config-steps", type=int, default=5)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))      # plain `python bench.py --gpus N`: this process only starts the ranks

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus}: refusing to print a line for another world size")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # one rank per GPU; PCC_BENCH_BACKEND=gloo lets several ranks share a GPU (used only to rehearse
    # the multi-rank control flow on a one-GPU box — RCCL refuses two ranks on one device)
    backend = os.environ.get("PCC_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    torch.cuda.set_device(local)
    # PCC_BENCH_TILED=1: the tiled (N > 1) workload and its RCCL exchange with a world of one rank — rehearses that
    # leg on a one-GPU box; never the headline
    tiled_mode = world > 1 or os.environ.get("PCC_BENCH_TILED") == "1"
    if tiled_mode and world == 1:
        os.environ["PCC_TILED_FORCE_COLLECTIVE"] = "1"    # the world of one still goes through the two collectives
    dist = None
    if tiled_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    tiled = importlib.import_module(PKG + ".tiled")
    dev = torch.device("cuda", local)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    q_dec = len(SETTINGS)

    # ---- workload.  N = 1: the C2 frame.  N > 1: two 500k-voxel octree blocks of the tiled scan per rank.
    t0 = time.time()
    if not tiled_mode:
        frames = [wl.room(args.points, seed=0)]
    else:
        # room `rank` of a fused scan of `world` C2 rooms side by side along x, cut into its two octree blocks
        room = wl.room(args.points, seed=rank, offset=(512 * rank - 256 * world, -512, -256))
        frames, origins = tiled.cut_tiles(room, (256, 512, 256))
        assert len(frames) == 2, origins
    n_pts = int(sum(f["points"].shape[0] for f in frames))
    log(f"[rank {rank}] workload: {len(frames)} frame(s), {n_pts} voxels generated in {time.time() - t0:.1f}s")
    assert all(f["points"].dtype == np.int16 and f["colors"].dtype == np.float64 for f in frames)
    d_frames = [{"points": torch.from_numpy(f["points"].astype(np.int32)).to(dev),
                 "colors": torch.from_numpy(f["colors"].astype(np.float32)).to(dev)} for f in frames]

    enc = pkg.CompressionPipeline(SETTINGS, device=local, slots=1)
    dec = pkg.DecompressionPipeline(device=local, slots=1, output="numpy")
    dec_dev = pkg.DecompressionPipeline(device=local, slots=1, output="device")

    enc_v1 = pkg.CompressionPipeline(SETTINGS, device=local, slots=1, container_version=1) if not tiled_mode else None

    def step(host=True, enc=enc):
        """one pass of the operator contract.  host=True: numpy frames in, numpy frames out (the contract of the
        reference's operators); host=False: inputs resident in HBM, reconstruction left in HBM.  enc: the pipeline
        that writes the reference's container (default) or the flagged version-1 container (GPU-coded strings)."""
        src = frames if host else d_frames
        d = dec if host else dec_dev
        if not tiled_mode:
            out, side = enc.compress({"frames": [dict(f) for f in src], "timestamps": {}})
            rec, dside = d.decompress(out[q_dec])
            return out, side, rec, dside, out[q_dec]
        # tiled scan: code my blocks, exchange the sub-bitstreams of every quality, decode what my neighbour coded
        bundles, side = tiled.compress_tiled(enc.compress, [dict(f) for f in src], list(range(1, q_dec + 1)), coll_dev)
        theirs = bundles[(rank + 1) % world][q_dec - 1]
        rec, dside = d.decompress(theirs)
        return {q: bundles[rank][q - 1] for q in range(1, q_dec + 1)}, side, rec, dside, theirs

    def prof_table(rts, n_steps):
        recs = []
        for r in rts:
            recs += r.prof_records()
            r.prof_enable(False)
        groups = {}
        for op, ms, dims in recs:
            g = groups.setdefault((op, dims), [0, 0.0])
            g[0] += 1
            g[1] += ms
        return sorted(groups.items(), key=lambda kv: -kv[1][1]), sum(v[1] for v in groups.values()) / n_steps

    # warm-up; its last step (untimed) runs with an event pair around EVERY C-ABI call: the per-op table below
    # and the choice of the dominant kernel come from it
    rts = enc.runtimes + dec.runtimes
    for i in range(max(args.warmup, 1)):
        if i == max(args.warmup, 1) - 1:
            for r in rts:
                r.prof_enable(True, reserve=400)
        out, side, rec, dside, decoded = step()
    assert sum(int(r["points"].shape[0]) for r in rec) == n_pts or tiled_mode
    torch.cuda.synchronize()
    table_all, sum_all = prof_table(rts, 1)
    layer_all = [(k, v) for k, v in table_all if k[0] in ("sparse_conv", "convT_gen")]
    step(host=False)                       # warm the HBM-resident variant's decoder slot too

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    per_rank = {}   # N > 1: ms per step of every rank, per timed region

    def timed(host, enc=enc):
        # the interpreter's cyclic garbage collector stays out of the timed steps, as in timeit: with torch imported a
        # full pass walks ~170k objects (8-13 ms on these hosts, 40-50 in the build container) and used to land in one
        # step of every timed region — 20 steps of 8.9 ms read as 9.4-9.9 (per-step spread on stderr: max 13 ms)
        gc.collect()
        gc.disable()
        try:
            fence()
            t_start = time.perf_counter()
            e_ms, d_ms, last = [], [], None
            for _ in range(args.steps):
                last = step(host, enc)
                e_ms.append(1e3 * (last[1]["timestamps"]["codec_end"] - last[1]["timestamps"]["codec_start"]))
                d_ms.append(1e3 * (last[3]["timestamps"]["codec_end"] - last[3]["timestamps"]["codec_start"]))
            fence()
            dt = time.perf_counter() - t_start
        finally:
            gc.enable()
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
            dist.all_gather(every, t)
            per_rank[("host" if host else "hbm", id(enc))] = [1e3 * float(v.item()) / args.steps for v in every]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # per-step spread on stderr: one GOP that takes twice as long (a one-time set-up inside the runtime, a throttled
        # host) shows here, not in the mean
        log(f"[rank {rank}] per-step ms ({'host' if host else 'hbm'} frames): encode min / median / max "
            f"{min(e_ms):.2f} / {float(np.median(e_ms)):.2f} / {max(e_ms):.2f}, decode "
            f"{min(d_ms):.2f} / {float(np.median(d_ms)):.2f} / {max(d_ms):.2f}; every step: "
            + " ".join(f"{a:.1f}+{b:.1f}" for a, b in zip(e_ms, d_ms)))
        return dt, float(np.mean(e_ms)), float(np.mean(d_ms)), last

    # timed region (`value`): K steps of the operator contract, host numpy in / host numpy out.  Only the dominant
    # kernel is bracketed by HIP events (one pair per step); bracketing all ~150 calls of a step costs the step
    # ~0.5 ms of stream bubbles
    if layer_all:
        (dom_op, dom_dims), _ = layer_all[0]
        for r in rts:
            r.prof_enable(True, reserve=8 * (args.steps + 1), only=dom_op, rows=dom_dims[0])
    elapsed, enc_ms, dec_ms, (out, side, rec, dside, decoded) = timed(host=True)
    # ---- launches of the dominant kernel inside the timed region (HIP events on the ctx streams)
    table, _ = prof_table(rts, args.steps)
    # secondary: the same K steps with the frame resident in HBM and the reconstruction left there (no events)
    elapsed_hbm, enc_ms_hbm, dec_ms_hbm, _ = timed(host=False)
    # the flagged container (version 1): y / z strings coded by the GPU's interleaved rANS in both directions, no
    # serial host coder on the path.  One profiled step for the coder kernels' own figures, then K timed steps.
    gpu_rans = None
    if not tiled_mode:
        step(True, enc_v1)
        rts1 = enc_v1.runtimes + dec.runtimes
        for r in rts1:
            r.prof_enable(True, reserve=400)
        out1, side1, rec1, dside1, _ = step(True, enc_v1)
        torch.cuda.synchronize()
        table1, _ = prof_table(rts1, 1)
        elapsed_v1, enc_ms_v1, dec_ms_v1, (out1, side1, rec1, dside1, _) = timed(host=True, enc=enc_v1)
        same = all(np.array_equal(a["points"], b["points"]) and np.array_equal(a["colors"], b["colors"])
                   for a, b in zip(rec1, rec))
        assert same, "version-1 container decodes to a different reconstruction"
        gpu_rans = {"value": args.steps / elapsed_v1, "ms_per_step": 1e3 * elapsed_v1 / args.steps,
                    "encode_ms": enc_ms_v1, "decode_ms": dec_ms_v1, "bpp": [float(b) for b in side1["gop_info"]["bpp"]],
                    "reconstruction_equals_version0": bool(same), "kernels": {}}
        for (op, dims), (cnt, tot) in table1:
            if op in ("rans_encode_dev", "rans_decode_dev"):
                n_sym, n_str, t_steps, n_chunks = dims
                if n_sym < 100000:
                    continue                      # the z stream: a single small launch
                stream = (len(out1[q_dec]) if op == "rans_decode_dev" else sum(len(out1[q]) for q in range(1, q_dec + 1)))
                nbytes = 5 * n_sym * n_str + stream     # int32 symbol + uint8 index per symbol, + the stream(s)
                ms = tot / cnt
                gpu_rans["kernels"][op] = {
                    "symbols": int(n_sym * n_str), "chunks": int(n_chunks * n_str), "steps_per_chunk": int(t_steps),
                    "avg_ms": ms, "algorithmic_bytes": int(nbytes),
                    "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None},
                    "note": "one wave per chunk of 64 x steps symbols, steps coded one after the other: the launch "
                            "lasts as long as ONE chunk (latency of a dependent 32-bit state update and a binary search "
                            "in LDS per step), far from the HBM roof by construction; what it buys is the removal of the "
                            "serial host coder and of the symbols' PCIe round trip"}
    # the reference's container + seek-point trailer (pcc_codec_set_seek_points): everything the reference's reader reads
    # is unchanged — it stops at the last frame record, in front of the trailer — and this library's decoder decodes the
    # y string in pieces on host threads.  Informational like value_gpu_rans; `value` stays the plain container.
    seek = None
    if not tiled_mode:
        pieces = int(os.environ.get("PCC_BENCH_SEEK_POINTS", "16"))
        enc_sk = pkg.CompressionPipeline(SETTINGS, device=local, slots=1, seek_points=pieces)
        step(True, enc_sk)
        elapsed_sk, enc_ms_sk, dec_ms_sk, (out_sk, side_sk, rec_sk, dside_sk, _) = timed(host=True, enc=enc_sk)
        same = all(np.array_equal(a["points"], b["points"]) and np.array_equal(a["colors"], b["colors"]) for a, b in zip(rec_sk, rec))
        prefix = all(out_sk[q][:len(out[q])] == out[q] for q in range(1, q_dec + 1))
        assert same and prefix, "seek-point containers: not the plain container + trailer, or another reconstruction"
        seek = {"value": args.steps / elapsed_sk, "ms_per_step": 1e3 * elapsed_sk / args.steps, "encode_ms": enc_ms_sk,
                "decode_ms": dec_ms_sk, "pieces": pieces, "trailer_bytes": [len(out_sk[q]) - len(out[q]) for q in range(1, q_dec + 1)],
                "bpp": [float(b) for b in side_sk["gop_info"]["bpp"]], "container_is_version0_plus_trailer": bool(prefix),
                "reconstruction_equals_version0": bool(same),
                "decode_stages_ms": {k: round(1e3 * v, 3) for k, v in dside_sk["time_measurements"].items()}}
        del enc_sk
    table = [kv for kv in table if layer_all and kv[0] == (dom_op, dom_dims)]
    if rank == 0:
        log("per-op device time of the last warm-up step (HIP events around every C-ABI call):")
        for (op, dims), (cnt, tot) in table_all[:25]:
            log(f"  {op:14s} {str(dims):34s} x{cnt:4d}  total {tot:9.3f} ms  avg {tot / cnt:8.4f} ms")
        log(f"  sum of recorded ops of that step: {sum_all:.3f} ms; "
            f"timed region: wall per step {1e3 * elapsed / args.steps:.3f} ms (enc {np.mean(enc_ms):.2f} + dec {np.mean(dec_ms):.2f})")
        st = side["enc_time_measurements"]
        log("  encode stages ms:", {k: (round(1e3 * v, 3) if not isinstance(v, list) else [round(1e3 * x, 3) for x in v])
                                    for k, v in st.items()})
        log("  decode stages ms:", {k: round(1e3 * v, 3) for k, v in dside["time_measurements"].items()})

    # ---- roofline of the dominant kernel (largest total device time among the layer kernels)
    roofline = None
    layer_ops = table
    if layer_ops and rank == 0:
        (op, dims), (cnt, tot) = layer_ops[0]
        avg_s = tot / cnt / 1e3
        # one extra untimed step through the op-by-op engine (same kernels, same rule books) to count the
        # active pairs of every rule book
        enc_o = pkg.CompressionPipeline(SETTINGS, device=local, slots=1, engine="ops")
        dec_o = pkg.DecompressionPipeline(device=local, slots=1, output="device", engine="ops")
        for r in enc_o.runtimes + dec_o.runtimes:
            r.pairs_log = {}
        out_o, _ = enc_o.compress({"frames": [dict(f) for f in d_frames], "timestamps": {}})
        assert out_o[q_dec] == out[q_dec], "the two engines wrote different containers"
        dec_o.decompress(decoded)          # the container the timed steps decoded (N > 1: the neighbour's blocks)
        pairs = {}
        for r in enc_o.runtimes + dec_o.runtimes:
            pairs.update(r.pairs_log)
            r.pairs_log = None
        del enc_o, dec_o
        n_out, cin, cout, k_vol = dims
        if op == "sparse_conv" and k_vol == 27:
            p = pairs.get(n_out, 0)
            nbytes, flops = conv_algorithmic(n_out, n_out, cin, cout, 27, p)
        elif op == "sparse_conv":
            # stride-2 kernel-2: every input row feeds exactly one output row
            p = max(pairs.values()) if pairs else n_out
            nbytes, flops = conv_algorithmic(n_out, p, cin, cout, k_vol, p)
        else:
            p = n_out * 8
            nbytes = 4 * (n_out * cin + 8 * n_out * cout) + 4 * 8 * cin * cout
            flops = 2 * p * cin * cout
        t_hbm, t_mfma = nbytes / (HBM_PEAK_GBS * 1e9), flops / (MFMA_F32_PEAK_TFLOPS * 1e12)
        if t_mfma >= t_hbm:
            ach = flops / avg_s / 1e12
            roofline = {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None}
        else:
            ach = nbytes / avg_s / 1e9
            roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None}
        # HBM traffic per launch from the committed PMC passes (separate rocprofv3 --pmc runs of this
        # same command, tools/pmc_summary.py); matched by kernel and grid size, else null
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            kname = {"sparse_conv": "k_gconv", "convT_gen": "k_convT_mfma"}[op]
            # k_gconv16: one 64-thread workgroup per 64-row window, k_gconv_up (the g_s layers): one per 128-row window,
            # grids rounded up to a multiple of 8 workgroups (per-XCD window order); k_convT_mfma: 4 waves x 32 rows
            grids = (((n_out + 63) // 64 + 7) // 8 * 8 * 64, ((n_out + 127) // 128 + 7) // 8 * 8 * 64,
                     ((n_out + 127) // 128) * 256)
            recs = [r for r in pmc["kernels"] if kname in r["kernel"] and r["grid_threads"] in grids]
            # the timed (native) engine runs the g_s convs in the form that makes the child rule book in-kernel on
            # channel-permuted rows (<true, true, true, 32>); the PMC passes also hold the explicit-rule-book form
            # from the op-by-op pairs count
            for tag in ("k_gconv_up<true>", "k_gconv_up", "<true, true, true", "<true, true"):
                hit = [r for r in recs if tag in r["kernel"]]
                if hit:
                    recs = hit
                    break
            if recs:
                roofline["traffic"] = recs[0]["hbm_bytes_corrected"]
                roofline["traffic_kernel"] = recs[0]["kernel"].split("(")[0]
                roofline["traffic_source"] = "profiles/pmc_latest.json: (2*FETCH_SIZE+WRITE_SIZE)*1024"
        except (OSError, KeyError, ValueError):
            pass
        # the other roof, for the record: the same launch priced against the one that does NOT bind it
        roofline["other_roof"] = ({"bound": "hbm", "achieved": nbytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBS, "min_ms": 1e3 * t_hbm}
                                  if roofline["bound"] == "mfma" else
                                  {"bound": "mfma", "achieved": flops / avg_s / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": flops / avg_s / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                   "min_ms": 1e3 * t_mfma})
        roofline["min_ms"] = 1e3 * max(t_hbm, t_mfma)
        roofline.update({"kernel": f"{op}{list(dims)}", "avg_ms": avg_s * 1e3, "launches": cnt,
                         "algorithmic_bytes": nbytes, "algorithmic_flops": flops, "active_pairs": p})

    # ---- informational: the same K steps with 3 GOPs in flight, the way the reference's services call
    # the codec (3 pool threads, sender/encoder/encoder.py:50, receiver/decoder/decoder.py:47): host
    # entropy coding of one frame overlaps GPU work of another.  Not the reported `value`.
    inflight = None
    if args.inflight > 1 and not tiled_mode:
        import concurrent.futures as cf
        enc_n = pkg.CompressionPipeline(SETTINGS, device=local, slots=args.inflight)
        dec_n = pkg.DecompressionPipeline(device=local, slots=args.inflight, output="device")

        def step_n(_):
            gop = {"frames": [dict(f) for f in d_frames], "timestamps": {}}
            out_n, _s = enc_n.compress(gop)
            rec_n, _d = dec_n.decompress(out_n[q_dec])
            return rec_n[0]["points"].shape[0]

        with cf.ThreadPoolExecutor(max_workers=args.inflight) as ex:
            list(ex.map(step_n, range(2 * args.inflight)))          # warm the slots
            fence()
            t0 = time.perf_counter()
            sizes = list(ex.map(step_n, range(args.steps)))
            fence()
            dt = time.perf_counter() - t0
        assert all(s == n_pts for s in sizes)
        inflight = {"in_flight": args.inflight, "value": args.steps * world / dt, "unit": "frames/s",
                    "ms_per_step": 1e3 * dt / args.steps,
                    "note": "same K steps (frame resident in HBM, reconstruction left there) issued from a pool of "
                            "worker threads, one codec slot (HIP stream + arena) each; informational, `value` above "
                            "is the one-frame-at-a-time figure"}
        del enc_n, dec_n

    cpu, oracle_full = None, None
    if rank == 0 and not tiled_mode and not args.no_cpu_baseline:
        abi = importlib.import_module(PKG + "._abi")
        t0 = time.time()
        cpu, oracle_full = cpu_baseline(wl, frames[0], args.cpu_sample, abi.host_cpu_budget(), runs=args.cpu_runs)
        log(f"cpu_baseline in {time.time() - t0:.1f}s: encode {cpu['encode_s']:.2f}s decode {cpu['decode_s']:.2f}s", cpu["stages"])

    # ---- distortion (BASELINE.json's metric names D1-PSNR; the reference logs none — SURVEY.md §8d): MPEG pc_error
    # point-to-point D1 and luma PSNR of the decoded frame against the source, outside the timed region; for the HIP
    # reconstruction and for the CPU oracle's, which must be the same arrays
    quality = None
    if rank == 0 and not tiled_mode and not args.no_psnr:
        metrics = importlib.import_module(PKG + ".metrics")
        t0 = time.time()
        quality = {"hip": metrics.frame_quality(frames[0], rec[0])}
        if oracle_full is not None:
            o_out, o_rec = oracle_full
            quality["oracle"] = metrics.frame_quality(frames[0], o_rec[0])
            quality["containers_equal_oracle"] = all(out[q] == o_out[q] for q in range(1, q_dec + 1))
            quality["reconstruction_equals_oracle"] = bool(np.array_equal(rec[0]["points"], o_rec[0]["points"]) and
                                                           np.array_equal(rec[0]["colors"], o_rec[0]["colors"]))
            assert quality["containers_equal_oracle"] and quality["reconstruction_equals_oracle"], \
                "HIP path and CPU oracle disagree on the bench frame"
            assert quality["oracle"] == quality["hip"]
        log(f"distortion figures in {time.time() - t0:.1f}s:", quality)

    configs = None
    if rank == 0 and not tiled_mode and not args.no_configs:
        abi = importlib.import_module(PKG + "._abi")
        t0 = time.time()
        configs = bench_configs(pkg, wl, local, abi.host_cpu_budget(), steps=args.config_steps,
                                with_oracle=not args.no_cpu_baseline)
        log(f"configs in {time.time() - t0:.1f}s:", json.dumps(configs))

    if rank == 0:
        frames_total = args.steps * world
        if not tiled_mode:
            workload = ("C2 ScanNet-scale 1M-point frame (BASELINE.json configs[1]): seeded indoor scene, 512x512x256 "
                        "grid, F=1 frame per GOP, Q=3 settings, hyperprior model demo_small; host numpy in (int16 points, "
                        "float64 colours) / host numpy out")
            sharding = "one frame per GPU, no data-path collective"
        else:
            workload = (f"BASELINE.json configs[4] tiling: fused scan of {world} C2 rooms ({world * args.points} voxels) "
                        f"cut into {2 * world} octree blocks of 256x512x256, two blocks ({args.points} voxels) per GPU "
                        "(configs[4] itself is N=4: 4M points, 8 blocks): each rank codes its blocks as one GOP (Q=3), "
                        "all-gather of the sub-bitstreams of every quality, each rank decodes the blocks its neighbour "
                        "coded; host numpy in / host numpy out; hyperprior model demo_small")
            sharding = ("tiles dealt to ranks, one variable-length all-gather of sub-bitstreams per step "
                        f"({backend}) inside the timed region")
        line = {
            "metric": "point-cloud frames/sec encode+decode @1M pts",
            "value": frames_total / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "timing": "K steps one after the other between two fences; the interpreter's cyclic garbage collector is "
                      "collected before and disabled inside every timed region (timeit's practice)",
            "config": {"workload": workload,
                       "weights": "UNTRAINED: seeded synthetic checkpoint (tools/make_checkpoint.py) — the reference's "
                                  "weights are not in its tree; bpp and PSNR below are those of this checkpoint, not of "
                                  "a trained codec, and the decoder's neighbourhood statistics come from its top-k",
                       "points_per_step_per_gpu": n_pts, "frames_per_step_per_gpu": len(frames),
                       "qualities": len(SETTINGS), "decoded_quality": q_dec, "sharding": sharding,
                       "value_basis": "operator contract: host numpy frames in, host numpy frames out (PCIe legs inside "
                                      "the timed region), the reference's container (version 0: y / z strings single "
                                      "rANS streams coded on the host); value_hbm_resident = frame already in HBM, "
                                      "reconstruction left in HBM; value_gpu_rans = operator contract with the flagged "
                                      "version-1 container (strings coded by the GPU's interleaved rANS); "
                                      "value_seek_points = operator contract with the reference's container + a trailer "
                                      "behind its last frame record (invisible to the reference's reader) from which "
                                      "this decoder decodes the y string on 16 host threads"},
            "encode_ms": enc_ms, "decode_ms": dec_ms,
            "value_hbm_resident": frames_total / elapsed_hbm,
            "ms_per_step_hbm_resident": 1e3 * elapsed_hbm / args.steps,
            "encode_ms_hbm_resident": enc_ms_hbm, "decode_ms_hbm_resident": dec_ms_hbm,
            "bpp": [float(b) for b in side["gop_info"]["bpp"]],
            "value_gpu_rans": gpu_rans["value"] if gpu_rans else None,
            "gpu_rans": gpu_rans,
            "value_seek_points": seek["value"] if seek else None,
            "seek_points": seek,
            "d1_psnr": quality["hip"]["d1_psnr"] if quality else None,
            "y_psnr": quality["hip"]["y_psnr"] if quality else None,
            "d1_psnr_oracle": quality["oracle"]["d1_psnr"] if quality and "oracle" in quality else None,
            "y_psnr_oracle": quality["oracle"]["y_psnr"] if quality and "oracle" in quality else None,
            "psnr": ({"definition": "MPEG pc_error D1 point-to-point, 10 log10(3 p^2 / max(mse A->B, mse B->A)); "
                                    "Y = BT.709 luma over nearest-neighbour pairs; decoded quality 3 vs source frame",
                      "peak": quality["hip"]["peak"],
                      "reconstruction_equals_oracle": quality.get("reconstruction_equals_oracle"),
                      "containers_equal_oracle": quality.get("containers_equal_oracle")} if quality else None),
            "ranks_seen": dist.get_world_size() if dist is not None else 1,
            "backend": (backend + (" (RCCL over xGMI)" if backend == "nccl" else "")) if dist is not None else None,
            "ms_per_step_per_rank": per_rank.get(("host", id(enc))),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "throughput_in_flight": inflight,
            "configs": configs,
        }
        if line["ranks_seen"] != line["n_gpus"] or line["n_gpus"] != args.gpus:
            sys.exit(f"bench.py: {line['ranks_seen']} ranks seen for --gpus {args.gpus}: no line printed")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
