#!/usr/bin/env python3
"""bench.py -- Mphotons/s traced + binned + gathered (BASELINE.json metric).

One "step" = one frame of the hot path over resident inputs: trace -> bin -> gather
(+ one RCCL all-reduce of the irradiance grid when N > 1).  Workload at N = 1: BASELINE
config 2 (256^3 heterogeneous volume, 1 048 576 photons, 128^3 light volume).  Multi-GPU is
weak scaling: every rank traces its own 1 048 576-photon shard of a 1024 x (1024 N) emission
lattice into a full-size grid; the grids are summed with one all-reduce per frame.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
# multi-process GPU work on this pool needs dmabuf IPC (the image exports this already; harmless when set)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (volume dim, photons per rank lattice (nx, ny), grid dim)
    "config2": (256, (1024, 1024), 128),
    "config1": (64, (256, 256), 32),
    "config4": (512, (2048, 2048), 256),
}


def algorithmic_bytes(n, I, vol_bytes, tf_width, cells, channels, tiles, passes):
    """Algorithmic bytes per launch of each kernel of the path (DESIGN.md 'Kernels')."""
    m = n * I
    rec = 16 if channels == 1 else 32
    return {
        "trace_kernel": n * (32 + 8 + 8) + m * 32 + vol_bytes + tf_width * 4,
        "bin_keys_kernel": m * 16 + m * 8,
        "radix_hist_kernel": m * 4 + 256 * tiles * 4,
        "radix_rowscan_kernel": 2 * 256 * tiles * 4,
        "radix_scatter_kernel": m * 16 + 256 * tiles * 4,
        "bin_finalize_kernel": m * 4 + m * rec + m * 4 + m * rec,
        "cell_start_kernel": m * 4 + (cells + 1) * 4,
        "gather_kernel": m * rec + (cells + 1) * 4 + cells * channels * 4,
    }


def cpu_baseline(workload, vol_np, tf, n_lattice, grid_dim, light_dir):
    """The oracle (a plain-C port of the reference path) on this host's cores: one full frame of
    the same workload (trace + bin + gather).  Reported beside the GPU number, never measured
    as the product."""
    sys.path.insert(0, str(REPO / "tests"))
    import numpy as np
    from oracle_binding import Oracle, OTraceParams
    import cpm_amd
    P, S = cpm_amd.pipeline, cpm_amd.synthetic
    o = Oracle()
    host_threads = os.cpu_count() or 1
    nx, ny = n_lattice
    n = nx * ny
    d = P._normalize(light_dir)
    origin = np.array([0.5, 0.5, 0.5], np.float32) - np.float32(2.0) * d
    po_, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
    area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
    s = o.uniform_samples_2d(nx, ny)
    ls = o.directional_light_samples(s, (1, 1, 1), d, po_, u, v, area)
    isect = o.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
    st = np.zeros((n, 2), np.uint32)
    st[:, 0] = o.glibc_rand_sequence(0, n)
    o.seed_streams(st, 1 << 40)
    ovol = o.volume(vol_np)
    og = o.grid((grid_dim,) * 3, 1)
    p = OTraceParams()
    p.step_size = 1.0 / vol_np.shape[0]
    p.n_light_samples = n
    p.max_interactions = 1
    p.total_photons = n
    photons = np.zeros((n, 8), np.float32)
    radius = S.photon_radius_texture(vol_np.shape[::-1], 1.0)
    scale = o.relative_irradiance_scale(radius, n)
    out = np.zeros(grid_dim ** 3, np.float32)
    def one_frame():
        t0 = time.perf_counter()
        steps = o.trace(ovol, tf, S.UNIT_CUBE_AABB, p, ls, isect, st, photons)
        t1 = time.perf_counter()
        _, cs, srt = o.bin(photons, n, og)
        t2 = time.perf_counter()
        o.gather(srt, cs, n, og, radius, scale, out)
        t3 = time.perf_counter()
        return (t3 - t0, t1 - t0, t2 - t1, t3 - t2), steps

    # The OpenMP team size that serves this host best (all hardware threads is NOT it on a 2 x 64-core SMT box: 0.04 s
    # per frame at 32 threads, 0.4 s at 256): try a few sizes, keep the fastest.  The RNG state is not written back, so
    # every frame traces the same photons.
    candidates = sorted({t for t in (16, 32, 64, 128, host_threads) if t <= host_threads} | {host_threads})
    trial = {}
    for t in candidates:
        o.set_threads(t)
        one_frame()
        trial[t] = min(one_frame()[0][0], one_frame()[0][0])
    cores = min(trial, key=trial.get)
    o.set_threads(cores)
    # one warm-up frame, then the median of 5
    frames = []
    steps = 0
    for rep in range(6):
        f, steps = one_frame()
        if rep > 0:
            frames.append(f)
    frames.sort()
    total, tt, tb, tg = frames[len(frames) // 2]
    # the reference formulation on the CPU (sequential splat, the order its CAS loop would have on one thread)
    sp = np.zeros(grid_dim ** 3, np.float32)
    ts = time.perf_counter()
    o.splat(photons, n, og, radius, scale, sp)
    splat_s = time.perf_counter() - ts
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(n / total / 1e6, 4), "unit": "Mphotons/s", "cores": cores, "kind": "port", "cpu_model": model,
        "host_threads": host_threads, "frame_s_by_threads": {str(k): round(v, 4) for k, v in trial.items()},
        "sample": f"median of 5 full frames (after 1 warm-up) of {workload} ({n} photons): trace {tt:.3f} s (OpenMP x{cores}) + "
                  f"bin {tb:.3f} s (OpenMP x{min(cores, 64)}, counting sort on <= 16 of them) + gather {tg:.3f} s (OpenMP x{cores}), at the fastest of the OpenMP team sizes tried ({cores} of {host_threads} hardware threads); the reference's OpenCL cannot be timed "
                  f"here (no CPU OpenCL device, Inviwo absent): this is the oracle, a plain-C port of the same path",
        "ms_per_frame": round(total * 1e3, 1), "woodcock_steps": int(steps),
        "splat_formulation_ms": round(splat_s * 1e3, 1),
    }


def pmc_traffic(kernel, workload):
    """L2<->fabric bytes per launch of `kernel` from the newest committed PMC summary (profiles/*pmc_traffic.json,
    written by tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied).
    PMC counters cannot be collected from inside this process, so the figure is the committed measurement of the
    same command; None when no summary exists for the workload's dominant kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*pmc_traffic.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
            k = d["kernels"].get(kernel) if d.get("workload") == workload else None
        except (OSError, ValueError, KeyError):
            continue
        if k:
            return int(k["traffic_bytes"]), "profiles/" + os.path.basename(f)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--splat", action="store_true", help="also time the reference formulation (atomic splat)")
    ap.add_argument("--streams", type=int, default=4,
                    help="frames in flight for the extra 'pipelined' figure (0 = skip it); 'value' is always one stream")
    ap.add_argument("--test-backend", default="nccl", choices=["nccl", "gloo"],
                    help="(testing) process-group backend; gloo lets the N > 1 code path run with every rank on one GPU")
    ap.add_argument("--test-one-device", action="store_true", help="(testing) every rank uses cuda:0")
    ap.add_argument("--graph", action="store_true",
                    help="replay a captured HIP graph of the frame instead of eager launches (measured slower here: 0.299 vs 0.251 ms)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import cpm_amd
    import importlib
    S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
    sharding = importlib.import_module(cpm_amd.__name__ + ".sharding")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.test_one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcpm_hip has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # launched by torch.distributed.run (also at N = 1)
        import torch.distributed as dist
        if args.test_backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # backend nccl = RCCL on ROCm

    vdim, (nx, ny), gdim = WORKLOADS[args.workload]
    light_dir = (0.3, 0.5, -1.0)
    vol_np = S.heterogeneous_volume(vdim) if args.workload != "config1" else S.homogeneous_volume(vdim)
    tf = S.workspace_tf() if args.workload != "config1" else S.homogeneous_tf(0.25)
    n_rank = nx * ny
    ctx = B.Context(local_rank)
    fr = P.PhotonFrame(ctx, vol_np, tf, (nx, ny * world), (gdim,) * 3, light_travel_direction=light_dir,
                       photon_range=sharding.shard_range(n_rank * world, rank, world))

    use_graph = args.graph
    if use_graph:
        fr.capture()

    # the one exchange step -- the sum of the per-rank grids over xGMI -- overlaps the next frame's trace and bin
    # (double-buffered grids, sharding.OverlappedGridReducer); everything outstanding is waited for inside the timed region
    reducer = sharding.OverlappedGridReducer(fr.light_volume)
    frame_no = [0]

    def step():
        k = frame_no[0]
        frame_no[0] += 1
        if use_graph:
            k = 0  # the captured gather writes frame.light_volume (= buffer 0): no double buffering under replay
            reducer.acquire(k)
            fr.replay()
        else:
            fr.trace()
            fr.bin()
            fr.gather(out=reducer.acquire(k))
        reducer.reduce(k)

    def barrier():
        reducer.flush()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- second pass with per-kernel HIP events (library hook) on the same stream
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(args.steps):
        fr.trace()
        fr.bin()
        fr.gather()
    kern = ctx.profile_collect()
    ctx.profile_enable(False)
    # ---- one more trace with the Woodcock iteration counter on (statistics; its atomics are not timed)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.set_step_counter(counter)
    fr.trace()
    torch.cuda.synchronize()
    ctx.set_step_counter(None)
    woodcock_steps = int(counter.item())

    # ---- extra figure (never `value`): S independent frames in flight on S streams.  The kernels of one
    # 1 M-photon frame are latency-bound, so frames of different time steps / progressive
    # batches overlap; each frame owns its context, buffers and stream.
    pipelined = None
    if world == 1 and args.streams > 1:
        ctxs = [B.Context(local_rank) for _ in range(args.streams)]
        frames = [P.PhotonFrame(c, vol_np, tf, (nx, ny), (gdim,) * 3, light_travel_direction=light_dir) for c in ctxs]
        streams = [torch.cuda.Stream() for _ in range(args.streams)]
        rounds = max(1, args.steps // args.streams)
        for it in range(rounds + 2):
            if it == 2:
                torch.cuda.synchronize()
                tp = time.perf_counter()
            for f, st in zip(frames, streams):
                with torch.cuda.stream(st):
                    f.trace(); f.bin(); f.gather()
        torch.cuda.synchronize()
        dtp = time.perf_counter() - tp
        same = all(bool(torch.equal(f.light_volume, fr.light_volume)) for f in frames)
        pipelined = {"streams": args.streams, "frames": rounds * args.streams,
                     "value": round(rounds * args.streams * n_rank / dtp / 1e6, 2), "unit": "Mphotons/s",
                     "ms_per_frame": round(dtp / (rounds * args.streams) * 1e3, 4),
                     "light_volumes_identical_to_single_stream": same}
        del frames, ctxs

    splat_ms = None
    if args.splat:
        tmp = torch.zeros_like(fr.light_volume)
        for _ in range(3):
            fr.splat(tmp)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            fr.splat(tmp)
        torch.cuda.synchronize()
        splat_ms = (time.perf_counter() - ts) / args.steps * 1e3

    if rank == 0:
        n_photons = n_rank * world
        ms_per_step = elapsed / args.steps * 1e3
        value = n_photons * args.steps / elapsed / 1e6
        per_frame = {k: (tot / args.steps, calls / args.steps, tot / calls) for k, (tot, calls) in kern.items()}

        def stage(names):
            return round(sum(per_frame[k][0] for k in per_frame if any(k.startswith(n) for n in names)), 4)

        stages = {"trace": stage(["trace_kernel"]),
                  "bin": stage(["bin_", "radix_", "cell_start"]),
                  "gather": stage(["gather"])}
        tile = 256 * (4 if n_rank <= (1 << 15) else 8 if n_rank <= (1 << 23) else 16)
        tiles = -(-n_rank // tile)
        key_bits = int(gdim ** 3).bit_length()
        passes = -(-key_bits // 8)
        ab = algorithmic_bytes(n_rank, 1, vol_np.size, tf.shape[0], gdim ** 3, 1, tiles, passes)
        dom = max(per_frame, key=lambda k: per_frame[k][0])
        dom_base = dom.split("<")[0]
        if dom_base.startswith("gather"):
            dom_base = "gather_kernel"
        dom_avg_ms = per_frame[dom][2]
        achieved = ab[dom_base] / (dom_avg_ms * 1e-3) / 1e9
        frame_bytes = (ab["trace_kernel"] + ab["bin_keys_kernel"] + passes * (ab["radix_hist_kernel"] + ab["radix_rowscan_kernel"] + ab["radix_scatter_kernel"])
                       + ab["bin_finalize_kernel"] + ab["cell_start_kernel"] + ab["gather_kernel"])
        traffic, traffic_src = pmc_traffic(dom.split("<")[0], args.workload)
        out = {
            "metric": "Mphotons/s traced+binned+gathered",
            "value": round(value, 2), "unit": "Mphotons/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE {args.workload}: {vdim}^3 u8 heterogeneous volume, {n_rank} photons per GPU "
                                   f"({nx}x{ny * world} lattice, one directional light), {gdim}^3 x1 f32 light volume, "
                                   f"I=1, r=1 voxel, MWC64X streams from glibc srand(0)",
                       "photons_per_gpu": n_rank, "volume": [vdim] * 3, "light_volume": [gdim] * 3,
                       "parallelism": (f"photon-sharded x{world}, one RCCL all-reduce of the grid per frame, overlapped with the next "
                                       f"frame's trace + bin (double-buffered grid)") if world > 1 else "single GPU",
                       "launch": "captured HIP graph replay" if use_graph else "eager launches"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ab[dom_base], "avg_launch_ms": round(dom_avg_ms, 5),
                         "launches_per_frame": round(per_frame[dom][1], 2)},
            "frame": {"kernel_ms_per_frame": {k: round(v[0], 5) for k, v in sorted(per_frame.items(), key=lambda kv: -kv[1][0])},
                      "stage_ms": stages, "sum_kernel_ms": round(sum(v[0] for v in per_frame.values()), 4),
                      "algorithmic_bytes_per_frame": frame_bytes,
                      "frame_hbm_frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                      "woodcock_steps_per_frame": woodcock_steps,
                      "gsamples_per_s": round(woodcock_steps / max(stages["trace"], 1e-9) / 1e6, 3)},
        }
        if pipelined is not None:
            out["pipelined"] = pipelined
        if splat_ms is not None:
            out["frame"]["reference_formulation_splat_ms"] = round(splat_ms, 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, vol_np, tf, (nx, ny), gdim, light_dir)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
