#!/usr/bin/env python3
"""bench.py -- Mphotons/s traced + binned + gathered (BASELINE.json metric).

One "step" = one frame of the hot path over resident inputs: trace -> bin -> gather (+ one RCCL all-reduce of the
irradiance grid, through the C-ABI, when N > 1).  Workload at N = 1: BASELINE config 2 (256^3 heterogeneous volume,
1 048 576 photons, 128^3 light volume).

  --formulation fast   (default) the tolerance-mode MI355X formulation: brick bin + fixed-point LDS-tile gather
                       (cpm_bin_fast + cpm_gather_fast; light volume within rtol 2e-5 / atol 1e-5 max of the reference
                       semantics, bitwise reproducible)
                exact  cell sort + sequential per-voxel gather (cpm_bin + cpm_gather; bit-exact contract)
  --scaling weak       (default at config 2) every rank traces its own 1024 x 1024 lattice rows (1024 x 1024 N in all)
            strong     (default at config 4 / 5: BASELINE fixes their photon counts) ONE lattice, sharded over the ranks
  --workload config5   a step is one time step of the 256^3 sequence: volume step (difference, min/max, importance),
                       correlated re-trace of the rank's shard, delta light-volume update, touched-brick reduce

  python bench.py --gpus N --steps K --warmup W        (N > 1 without WORLD_SIZE in the environment: this process touches
                                                        no GPU and starts the N ranks itself as child processes)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
                                                       (either way a rank runs its end of a communicator set-up in a probe
                                                        process first -- `--rccl-probe`, killed when it hangs -- and sets up
                                                        in-process only what passed on every rank: C-ABI RCCL -> torch's NCCL
                                                        group -> gloo; config.transport_probes says which)

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
# the CPU baseline's OpenMP teams: one thread per core, neighbours first (measured on the GPU box: 59 vs 62 ms per frame)
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_PROC_BIND", "close")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # ... and the copy rate measured there (6.29 TB/s): the ceiling a streaming kernel can reach

WORKLOADS = {
    # name: (volume dim, lattice (nx, ny), grid dim, default scaling)
    "config2": (256, (1024, 1024), 128, "weak"),
    "config1": (64, (256, 256), 32, "weak"),
    "config4": (512, (2048, 2048), 256, "strong"),
    "config5": (256, (1024, 1024), 128, "strong"),
}
LIGHT_DIR = (0.3, 0.5, -1.0)


def algorithmic_bytes(n, I, vol_bytes, tf_width, cells, channels, tiles):
    """Algorithmic bytes per launch of each kernel of the path."""
    m = n * I
    rec = 16 if channels == 1 else 32
    return {
        "trace_kernel": n * (32 + 8 + 8) + m * 32 + vol_bytes + tf_width * 4,
        # exact formulation
        "bin_keys_kernel": m * 16 + m * 8,
        "radix_hist_kernel": m * 4 + 256 * tiles * 4,
        "radix_rowscan_kernel": 2 * 256 * tiles * 4,
        "radix_scatter_kernel": m * 16 + 256 * tiles * 4,
        "bin_finalize_kernel": m * 4 + m * rec + m * 4 + m * rec,
        "cell_start_kernel": m * 4 + (cells + 1) * 4,
        "gather_kernel": m * rec + (cells + 1) * 4 + cells * channels * 4,
        # tolerance-mode formulation: what must move (records in; records out; light volume out)
        "fast_count_kernel": m * rec,
        "fast_scatter_kernel": m * rec + m * rec,
        "fast_brick_kernel": m * rec + cells * channels * 4,
    }


def cpu_quota_cores():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota); None when unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(workload, vol_np, tf, n_lattice, grid_dim):
    """The oracle (a plain-C port of the reference path) on this host's cores: full frames of the same workload
    (trace + bin + gather).  Reported beside the GPU number, never measured as the product."""
    sys.path.insert(0, str(REPO / "tests"))
    import numpy as np
    from oracle_binding import Oracle, OTraceParams
    import cpm_amd
    P, S = cpm_amd.pipeline, cpm_amd.synthetic
    o = Oracle()
    host_threads = os.cpu_count() or 1
    quota = cpu_quota_cores()
    nx, ny = n_lattice
    n = nx * ny
    d = P._normalize(LIGHT_DIR)
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

    # OpenMP team size.  The GPU box's container carries a CPU quota (cgroup cpu.max: 16 CPUs of a 2 x 64-core host when this
    # was written): a bigger team finishes a stage faster and is then throttled for the rest of the 100 ms period -- "more
    # threads" measured 0.06 s at 16, 0.10 s at 64, 0.40 s at 256.  The team is the quota when there is one, else one
    # thread per physical core; the sizes around it are tried too and the fastest kept.  The RNG state is not written back,
    # so every frame traces the same photons.
    physical = max(1, host_threads // 2)
    base = int(quota) if quota and quota >= 1 else physical
    candidates = sorted({t for t in (base // 2, base, base * 2, physical) if 1 <= t <= host_threads})
    trial = {}
    for t in candidates:
        o.set_threads(t)
        one_frame()
        trial[t] = sorted(one_frame()[0][0] for _ in range(3))[1]   # median of 3: a throttled team has fast and slow frames
    cores = min(trial, key=trial.get)
    o.set_threads(cores)
    frames = []
    steps = 0
    for rep in range(6):  # one warm-up frame, then the median of 5
        f, steps = one_frame()
        if rep > 0:
            frames.append(f)
    frames.sort()
    total, tt, tb, tg = frames[len(frames) // 2]
    sp = np.zeros(grid_dim ** 3, np.float32)
    ts = time.perf_counter()
    o.splat(photons, n, og, radius, scale, sp)  # the reference formulation, sequential (the order its CAS loop has on one thread)
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
    why = (f"the container's CPU quota is {quota:g} CPUs (cgroup cpu.max) of {host_threads} hardware threads: bigger teams are throttled"
           if quota else f"one thread per physical core of {host_threads} hardware threads")
    return {
        "value": round(n / total / 1e6, 4), "unit": "Mphotons/s", "cores": cores, "kind": "port", "cpu_model": model,
        "host_threads": host_threads, "cpu_quota_cores": quota, "frame_s_by_threads": {str(k): round(v, 4) for k, v in trial.items()},
        "sample": f"median of 5 full frames (after 1 warm-up) of {workload} ({n} photons): trace {tt:.3f} s + bin {tb:.3f} s + gather "
                  f"{tg:.3f} s, OpenMP x{cores} ({why}; fastest of the team sizes tried); the reference's OpenCL cannot be timed here "
                  f"(no CPU OpenCL device, Inviwo absent): this is the oracle, a plain-C port of the same path",
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


def timed(torch, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


LAUNCH_BUDGET_S = 200.0   # wall clock one set of ranks gets from the moment they are UP (torch imported, process group formed): set-up + frames + extras of a
                          # default run take < 60 s; what comes before -- the first `import torch` on a fresh box -- can take minutes and is not RCCL's fault
LAUNCH_TOTAL_S = (400.0, 170.0)   # ... and from their start at most this, first and second set (together inside the driver's 600 s)
RANKS_UP_MARKER = "bench.py: ranks up"


def _run_children(cmd, env, budget_s, total_s=None):
    """Start `cmd` in its own process group, relay its stderr, collect its stdout; kill the group budget_s seconds after the ranks reported
    that they are up (RANKS_UP_MARKER on stderr) or total_s seconds after the start, whichever comes first.
    -> (return code or None when killed, rank 0's JSON line or None, the last lines the children wrote, what the clock that ran out was)."""
    import signal
    import subprocess
    import threading
    from collections import deque
    t0 = time.monotonic()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    tail, found, up_at = deque(maxlen=60), [], []

    def pump(stream, is_stdout):
        for line in stream:
            if is_stdout and line.startswith("{") and '"metric"' in line:
                found.append(line)
            else:
                if RANKS_UP_MARKER in line and not up_at:
                    up_at.append(time.monotonic())
                tail.append(line)
                sys.stderr.write(line)
    threads = [threading.Thread(target=pump, args=(proc.stdout, True), daemon=True), threading.Thread(target=pump, args=(proc.stderr, False), daemon=True)]
    for t in threads:
        t.start()
    rc, why = None, None
    while True:
        try:
            rc = proc.wait(timeout=0.25)
            break
        except subprocess.TimeoutExpired:
            now = time.monotonic()
            if up_at and now - up_at[0] > budget_s:
                why = f"{budget_s:.0f} s after they were up"
            elif total_s is not None and now - t0 > total_s:
                why = f"{total_s:.0f} s after their start" + ("" if up_at else " (they never reported being up)")
            elif not up_at and total_s is None and now - t0 > budget_s:
                why = f"{budget_s:.0f} s after their start"
            if why:
                break
    if why:
        # (children of a parent that never touched a GPU; the whole group: the launcher and every rank it started)
        for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
    for t in threads:
        t.join(timeout=5)
    return rc, (found[-1] if found else None), list(tail), why


def launch_ranks(n_ranks, argv=None, budget_s=None, make_cmd=None):
    """`python bench.py --gpus N` run plainly (no WORLD_SIZE): start the N ranks -- one process per GPU, torch.distributed.run
    on 127.0.0.1 -- as CHILD processes of this one, which has not touched a GPU and never does; relay rank 0's JSON line and exit
    with the children's return code.  The run always ends with a line or an error: the ranks get `budget_s` seconds of wall clock from the
    moment they report being up (CPM_BENCH_LAUNCH_BUDGET_S; a set-up that hangs inside RCCL never returns by itself) and LAUNCH_TOTAL_S from
    their start (importing torch on a fresh box takes minutes: not charged to the first clock); when they overrun either or die without a
    line they are killed and a FRESH set is started ONCE with `--transport torch --exchange union` (torch.distributed's own NCCL
    group, the exchange whose collectives have run before) -- that line says so in config.launcher_note; a second failure exits
    non-zero with the ranks' last lines.  make_cmd (testing): argv -> the command to run instead of torch.distributed.run."""
    import socket
    argv = list(sys.argv[1:] if argv is None else argv)
    budget = float(os.environ.get("CPM_BENCH_LAUNCH_BUDGET_S", LAUNCH_BUDGET_S) if budget_s is None else budget_s)

    def command(args):
        if make_cmd is not None:
            return make_cmd(args), dict(os.environ)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
                "--master-port", str(port), str(Path(__file__).resolve())] + args, env
    notes = []
    for attempt, args in enumerate((argv, argv + ["--transport", "torch", "--exchange", "union"])):
        cmd, env = command(args)
        t0 = time.perf_counter()
        total = None if budget_s is not None else (float(os.environ["CPM_BENCH_LAUNCH_TOTAL_S"]) if "CPM_BENCH_LAUNCH_TOTAL_S" in os.environ else LAUNCH_TOTAL_S[attempt])
        rc, line, tail, expired = _run_children(cmd, env, budget, total)
        took = time.perf_counter() - t0
        if rc == 0 and line:
            if notes:   # the line of the second set of ranks says why it is not the first's
                try:
                    d = json.loads(line)
                    d.setdefault("config", {})["launcher_note"] = "; ".join(notes) + "; this line: a fresh set of ranks with --transport torch --exchange union"
                    line = json.dumps(d) + "\n"
                except ValueError:
                    pass
            sys.stdout.write(line)
            sys.stdout.flush()
            return 0
        why = (f"the ranks did not finish ({expired}) and were killed" if expired else
               f"the ranks exited with code {rc} after {took:.0f} s" if rc else "the ranks exited cleanly but rank 0 printed no JSON line")
        notes.append(f"attempt {attempt + 1} ({' '.join(args[-4:])}): {why}")
        print(f"bench.py: {notes[-1]}" + ("; starting a fresh set of ranks with --transport torch --exchange union" if attempt == 0 else ""), file=sys.stderr)
        if attempt == 1:
            print("bench.py: no JSON line from either set of ranks; their last lines:\n" + "".join(tail[-40:]), file=sys.stderr)
    return 1


PROBE_BUDGET_S = 60.0   # wall clock a probe process gets (CPM_BENCH_PROBE_BUDGET_S): torch is in the page cache by then, a communicator of 8 takes seconds


def rccl_probe_child(argv):
    """`bench.py --rccl-probe KIND RANK WORLD DEVICE WHERE`: one rank's end of a communicator set-up + the collectives the frames use, in a
    process of its OWN -- started by a rank (probe_in_child) before that rank sets anything up itself.  A set-up that never returns
    (ncclCommInitRank waiting for a peer, a bootstrap socket nobody answers) cannot be left from inside the process it hangs in; here it
    hangs in a child the rank kills by PID, and the rank goes on with the next transport.  KIND cabi: the C-ABI's path (cpm_comm_create
    from the id WHERE in hex, cpm_allreduce_grid, one cpm_comm_send / cpm_comm_recv pair between ranks 1 and 0); KIND torch:
    torch.distributed's own NCCL group (rendezvous in the file WHERE) and one all_reduce.  Exit code 0 = every result was right."""
    import ctypes
    import signal
    kind, rank, world, device, where = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), argv[4]
    try:
        ctypes.CDLL("libc.so.6").prctl(1, int(signal.SIGKILL))   # PR_SET_PDEATHSIG: never outlives the rank that started it
    except OSError:
        pass
    signal.alarm(int(float(os.environ.get("CPM_BENCH_PROBE_BUDGET_S", PROBE_BUDGET_S))) + 30)   # ... nor its own budget by much
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py --rccl-probe needs a GPU")
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    if kind == "cabi":
        import importlib
        import cpm_amd
        sharding = importlib.import_module(cpm_amd.__name__ + ".sharding")
        ctx = cpm_amd.binding.Context(device)
        tr = sharding.RcclTransport(ctx, rank, world, root=None, unique_id=bytes.fromhex(where))
        probe = torch.ones(8, dtype=torch.float32, device=dev)
        tr.wait(tr.start(probe))
        torch.cuda.synchronize()
        ok = bool((probe == float(world)).all().item())
        if world > 1 and rank < 2:   # the brick lists' calls: rank 1's send meets the display GPU's receive
            buf = torch.full((256,), 3.0 if rank == 1 else 0.0, dtype=torch.float32, device=dev)
            with torch.cuda.stream(tr.stream):
                (ctx.comm_send if rank == 1 else ctx.comm_recv)(tr.comm, buf, buf.numel() * 4, 1 - rank)
            tr.stream.synchronize()
            ok = ok and bool((buf == 3.0).all().item())
        tr.close()
    elif kind == "torch":
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="file://" + where, rank=rank, world_size=world, device_id=dev)
        t = torch.ones(8, dtype=torch.float32, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        ok = bool((t == float(world)).all().item())
        dist.destroy_process_group()
    else:
        raise SystemExit(f"bench.py --rccl-probe: unknown kind {kind!r}")
    return 0 if ok else 4


def probe_in_child(kind, where, rank, world, device, budget_s):
    """Run this rank's end of the probe (rccl_probe_child) as a child process; -> (passed, what went wrong).  The child is killed by its PID
    when the budget runs out (it has no children of its own; it stays in the rank's process group, so whatever ends the rank's group ends it)."""
    import subprocess
    cmd = [sys.executable, str(Path(__file__).resolve()), "--rccl-probe", kind, str(rank), str(world), str(device), where]
    p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    try:
        _, err = p.communicate(timeout=budget_s)
    except subprocess.TimeoutExpired:
        p.kill()
        p.communicate()
        return False, f"rank {rank}'s probe process did not finish in {budget_s:.0f} s and was killed"
    if p.returncode == 0:
        return True, ""
    last = [l for l in (err or "").strip().splitlines() if l.strip()]
    said = [l for l in last if "rror" in l and "Warning" not in l] or last   # (the exception's line rather than an exit-time warning behind it)
    return False, f"rank {rank}'s probe process exited with code {p.returncode}" + (f" ({said[-1].strip()[:200]})" if said else "")


TIMED_BATCHES = 7   # the timed region is repeated; the median batch is reported
MIN_WARM_STEPS = 200 # untimed frames before the first timed batch, whatever --warmup says (see main)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--rccl-probe":
        raise SystemExit(rccl_probe_child(sys.argv[2:]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--formulation", default="fast", choices=["fast", "exact"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--shards", default="auto", choices=["auto", "tiles", "range"],
                    help="N > 1: a rank takes the lattice's 4096-sample tiles t = rank (mod N) (every rank sees every lit brick at 1/N of "
                         "the density: the gather stays balanced, every rank's brick set is the union) or a contiguous range (a slab of the "
                         "light plane: brick sets nearly disjoint).  auto: range with --exchange lists, else tiles")
    ap.add_argument("--records", default="planar", choices=["planar", "float8"],
                    help="fast formulation: photon records in the two-plane layout the tracer can write and the brick bin reads half of "
                         "(default), or the reference's float8 records")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the labelled extra figures (other formulations, I = 4, sparse TF, configs 3 / 5, pipelined)")
    ap.add_argument("--streams", type=int, default=4,
                    help="frames in flight for the extra 'pipelined' figure (0 = skip it); 'value' is always one stream")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch"],
                    help="the grid reduce: cpm_allreduce_grid through the C-ABI (default) or torch.distributed")
    ap.add_argument("--collective", default="reduce", choices=["allreduce", "reduce"],
                    help="N > 1: only rank 0, the display GPU, receives the summed light volume (the north-star's single RCCL reduce; "
                         "default) or every rank does (all-reduce: twice the wire traffic of a ring reduce)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "union", "lists", "dense"],
                    help="N > 1, full frames -- what crosses xGMI: the union of the ranks' non-zero 4x4x4 bricks (cpm_allreduce_grid_sparse: mask "
                         "max-reduce + packed payload sized on the host from the union two frames before), every rank's OWN bricks as a list "
                         "to the root (cpm_reduce_grid_bricklists: one send per rank; for shards with disjoint brick sets) or the whole grid.  "
                         "No form synchronises the stream.  auto: lists where the photon count is fixed (strong scaling: the per-rank compute "
                         "shrinks, the union does not), union where it grows with the ranks (weak)")
    ap.add_argument("--reduce", default=None, choices=["sparse", "dense"], help="(older spelling of --exchange union / dense)")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2, 3, 4],
                    help="fast formulation, full frames: 2 = every other frame on a second stream with its own context and buffers (frame k + 1's "
                         "trace and bin run beside frame k's gather: a frame of a small shard is four launch-latency chains that leave most of the "
                         "GPU idle -- measured on one GPU at config 4 / N = 8: 66 -> 46 - 53 us per frame).  Default 1: `value` is one stream at every N")
    ap.add_argument("--sequence", default="resident", choices=["resident", "streamed"],
                    help="--workload config5: the 32 time steps live on the device as volumes (default: what Inviwo's representations are after the "
                         "first loop) or in pinned host memory, every rank uploading step t + 1 on the library's copy stream while it updates step t "
                         "(cpm_volume_stream: SURVEY 8d's step with the upload in it)")
    ap.add_argument("--test-backend", default="nccl", choices=["nccl", "gloo"],
                    help="(testing) process-group backend; gloo lets the N > 1 code path run with every rank on one GPU")
    ap.add_argument("--test-one-device", action="store_true", help="(testing) every rank uses cuda:0")
    ap.add_argument("--graph", action="store_true",
                    help="replay a captured HIP graph of the frame instead of eager launches (measured slower here)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

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
    if os.environ.get("CPM_BENCH_SPIN", "1") != "0":
        # the timed region ends in a synchronize: a host thread that sleeps in it wakes tens of microseconds after the device is done --
        # 2 - 3 % of a 20-frame batch.  Spin instead (set before the device's context exists)
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            if hip.hipSetDevice(ctypes.c_int(local_rank)) == 0:       # (the flags are the current device's)
                hip.hipSetDeviceFlags(ctypes.c_uint(0x1))             # hipDeviceScheduleSpin
        except OSError:
            pass
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcpm_hip has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # Control plane: a gloo group (communicator id, agreement flags, max-over-ranks time).  The DATA path's exchange is RCCL through the
    # C-ABI; its communicator is set up BEFORE anything else touches RCCL on the devices -- torch's own NCCL group is only created
    # when that set-up failed on some rank and every rank falls back (so a fall-back never shares a device with a half-built
    # communicator; a set-up that never returns is the launcher's timeout to end).
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # launched by torch.distributed.run (also at N = 1)
        import torch.distributed as dist
        dist.init_process_group("gloo")
        if rank == 0:
            print(f"{RANKS_UP_MARKER} ({world})", file=sys.stderr, flush=True)   # (launch_ranks starts its set-up clock here)

    vdim, (nx, ny), gdim, default_scaling = WORKLOADS[args.workload]
    scaling = args.scaling or default_scaling
    fast = args.formulation == "fast"
    correlated = args.workload == "config5"
    vol_np = S.heterogeneous_volume(vdim) if args.workload != "config1" else S.homogeneous_volume(vdim)
    tf = S.workspace_tf() if args.workload != "config1" else S.homogeneous_tf(0.25)
    if scaling == "weak":
        lattice, n_total = (nx, ny * world), nx * ny * world
    else:
        lattice, n_total = (nx, ny), nx * ny
    # what crosses xGMI per frame, and which photons a rank owns (static rules: every rank takes the same decision from the same flags)
    exchange = args.exchange
    if args.reduce is not None:
        exchange = {"sparse": "union", "dense": "dense"}[args.reduce]
    root = 0 if args.collective == "reduce" else None
    lists_possible = fast and not correlated and root is not None and not args.graph
    if exchange == "lists" and not lists_possible:
        raise SystemExit("bench.py: --exchange lists is the full frames' reduce to the display GPU (fast formulation, --collective reduce, not config5 / --graph)")
    # shards: a fixed photon count (strong) shrinks a rank's compute with N while tile shards keep every rank's brick set the union --
    # contiguous ranges there (brick sets nearly disjoint: profiles/r05_shard_exchange_bytes_config4.json); a photon count that grows
    # with N (weak) keeps the balanced tile shards.  exchange = auto: decided below, once a probe frame has counted the bricks.
    shards_kind = args.shards if args.shards != "auto" else ("range" if (scaling == "strong" and lists_possible and exchange in ("auto", "lists")) else "tiles")
    if shards_kind == "tiles":
        shard = sharding.shard_tiles(n_total, rank, world)
    else:
        lo, hi = sharding.shard_range(n_total, rank, world)
        shard = np.arange(lo, hi, dtype=np.int64)
    n_rank = int(shard.size)
    ctx = B.Context(local_rank)
    transport_note = None
    # (testing: with CPM_RCCL_LIBRARY naming the tests' RCCL double -- shared memory between ranks on ONE GPU -- the C-ABI's multi-rank
    # path runs over ranks that share a device)
    fake_rccl = bool(os.environ.get("CPM_RCCL_LIBRARY"))

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def torch_transport():
        """torch.distributed carries the sums: over its own NCCL (= RCCL) group on real multi-GPU runs, over gloo in the tests."""
        if world > 1 and real_nccl:
            return sharding.TorchTransport(group=dist.new_group(backend="nccl"), root=root)
        return sharding.TorchTransport(root=root)

    # Probe processes (rccl_probe_child): a set-up that HANGS inside RCCL cannot be left from the process it hangs in -- whoever started
    # the ranks (torch.distributed.run, the driver's form for N > 1: nothing of ours above the ranks) would wait until its own limit.  So
    # every rank first runs its end of the set-up + the frames' collectives in a child it can kill, the ranks agree, and only a path whose
    # probe passed on every rank is set up in the rank itself: C-ABI RCCL -> torch.distributed's own NCCL group -> gloo (sums staged
    # through the host: slow, labelled, but a line).  Off with CPM_BENCH_RCCL_PROBE=0; ranks that share one GPU (the tests) probe only
    # when it is 1 (two processes per rank on one card).
    probe_env = os.environ.get("CPM_BENCH_RCCL_PROBE", "")
    probing = world > 1 and probe_env != "0" and (not args.test_one_device or probe_env == "1")
    probe_budget = float(os.environ.get("CPM_BENCH_PROBE_BUDGET_S", PROBE_BUDGET_S))
    # (testing, CPM_BENCH_TEST_SHARED_NCCL=1: the REAL RCCL with every rank on one GPU -- it refuses such a communicator, which walks the
    # ladder below with real errors from both RCCL paths)
    real_nccl = args.test_backend == "nccl" and (not args.test_one_device or os.environ.get("CPM_BENCH_TEST_SHARED_NCCL") == "1")
    probes = {}

    def probe_all(kind):
        """Every rank's child at once (the set-up is collective); -> (passed everywhere, the first rank's complaint)."""
        if kind == "cabi":
            where = [ctx.comm_unique_id().hex() if rank == 0 else None]
        else:
            import tempfile
            where = [os.path.join(tempfile.gettempdir(), f"cpm_bench_probe_{os.getpid()}_{int(time.time())}") if rank == 0 else None]
        dist.broadcast_object_list(where, src=0)
        t0 = time.perf_counter()
        good, why = probe_in_child(kind, where[0], rank, world, local_rank, probe_budget)
        whys = [None] * world
        dist.all_gather_object(whys, why)
        if kind == "torch" and rank == 0:
            try:
                os.unlink(where[0])
            except OSError:
                pass
        good = agree(good)
        why = next((w for w in whys if w), "")
        probes[kind] = "passed" if good else why
        probes[kind + "_s"] = round(time.perf_counter() - t0, 1)
        return good, why

    def fallback(reason):
        """The C-ABI's RCCL is out (reason): torch.distributed carries the sums -- over its own NCCL group when THAT passes its probe."""
        if probing and real_nccl:
            good, why = probe_all("torch")
            if not good:
                return (sharding.TorchTransport(root=root),
                        f"gloo: the sums are staged through the host ({reason}; torch.distributed's own NCCL group did not pass its probe either: {why})")
        return torch_transport(), f"torch.distributed ({reason})"

    transport = None
    if world > 1 and args.transport == "rccl" and (real_nccl or fake_rccl):
        # every rank first checks locally that RCCL can be bound (no communication); the ranks agree; the probe processes; then the
        # collective communicator set-up and a probe of the collectives the frames will use.  Should a rank fail, ALL ranks fall back to
        # torch.distributed (the same wire) and the JSON line says so.
        err = ""
        try:
            ctx.comm_unique_id()
            ok = True
        except Exception as e:  # noqa: BLE001
            ok, err = False, str(e)
        ok = agree(ok)
        if ok and probing:
            ok, err = probe_all("cabi")
        if ok:
            try:
                transport = sharding.RcclTransport(ctx, rank, world, root=root)
                probe = torch.ones(8, dtype=torch.float32, device=ctx.device)
                transport.wait(transport.start(probe))
                torch.cuda.synchronize()
                ok = bool((probe == float(world)).all().item()) if (root is None or rank == root) else True
                if not ok:
                    err = "probe reduce returned a wrong sum"
            except Exception as e:  # noqa: BLE001
                ok, err = False, str(e)
            if not agree(ok):
                if transport is not None:
                    try:
                        transport.close()
                    except Exception:  # noqa: BLE001
                        pass
                transport = None
        if transport is None:
            transport, transport_note = fallback("RCCL through the C-ABI was not usable on some rank" + (f": {err}" if err else ""))
    if transport is None:
        if dist is not None and world > 1 and args.transport == "torch" and probing and real_nccl:
            transport, note = fallback("--transport torch")
            transport_note = note if "gloo" in note else None
        else:
            transport = torch_transport() if dist is not None else sharding.TorchTransport(root=root)
    rccl = isinstance(transport, sharding.RcclTransport)
    # what one message costs over THIS node's links (SURVEY 8e / DESIGN 6: the exchange model's constants were assumptions): a 1 KB and a
    # 2 MB ping-pong between the display GPU and rank 1 through the C-ABI; every rank gets the figures (gloo) and prices the exchanges with them
    p2p = None
    if rccl and world > 1:
        ok = True
        try:
            p2p = sharding.measure_p2p(transport)
        except Exception as e:  # noqa: BLE001
            ok, p2p = False, None
            transport_note = f"send / receive pairs not usable over this transport ({e})"
        have = [p2p if rank == (0 if root is None else root) else None]
        dist.broadcast_object_list(have, src=0 if root is None else root)
        p2p = have[0] if agree(ok) else None
    p2p_kw = {"link_gbs": p2p["link_gbs"], "latency_us": p2p["latency_us"]} if p2p else {}

    def ranks_barrier():
        """Every rank's device work is done and every rank is here: the timed region's brackets."""
        torch.cuda.synchronize()
        if dist is not None and world > 1:
            if rccl:
                transport.barrier()
            else:
                dist.barrier(group=transport.group)
        torch.cuda.synchronize()

    correlated = args.workload == "config5"
    if correlated:
        n_steps = 32
        vols = [S.heterogeneous_volume(vdim, S.sequence_blob_center(t, n_steps)) for t in range(n_steps)]
        streamed = args.sequence == "streamed"
        if streamed:   # the sequence stays in (pinned) host memory: three device slots, uploads on the library's copy stream
            pinned_seq = B.PinnedSequence(ctx, vols)
            vstream = B.VolumeStream(ctx, vols[0], n_slots=3)
            vstream.prefetch(1, pinned_seq.steps[1])
        else:
            dvols = [ctx.volume_create(v) for v in vols]  # resident as volumes (as a sequence's VolumeCL representations are): no upload in the step
        fr = P.CorrelatedPhotonMapper(ctx, vols[0], tf, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR,
                                      tf_points=list(S.WORKSPACE_TF_POINTS), photon_indices=shard)
        fr.full_frame()
        total_grid = fr.light_volume.clone()
        if world > 1:
            sharding.allreduce_light_volume(total_grid, transport)
        fr.touched_mask = torch.zeros(((gdim + 3) // 4) ** 3, dtype=torch.uint8, device=ctx.device)
        fractions = []
        step_no = [0]
        delta_sr = ctx.sparse_reduce_create(transport.comm, fr.grid) if world > 1 and rccl else None
        delta_pending = []
        reduce_info = []

        def step():
            t = 1 + step_no[0] % (n_steps - 1)
            step_no[0] += 1
            if streamed:
                t_next = 1 + step_no[0] % (n_steps - 1)
                vstream.prefetch(t_next, pinned_seq.steps[t_next])   # crosses PCIe while this step is computed
                fr.set_volume(vstream.acquire(t, pinned_seq.steps[t]))
            else:
                fr.set_volume(dvols[t])
            fr.touched_mask.zero_()
            n = fr.correlated_update()
            fractions.append(n / max(fr.n, 1))
            if delta_sr is not None:
                # the delta path: only bricks touched by a re-traced photon (old or new position) changed on any rank; the
                # previous step's ticket is completed here (its count has long arrived), this step's at the next one.
                # A rank whose update REBUILT its volume (a rank-local decision: its own re-trace count against its own threshold)
                # hands in every brick as touched -- ONE reduce object, ONE kind of mask on every rank whatever path each took: the union
                # then covers everything and the frame's sum is the dense one, on all ranks alike (two objects with
                # separate histories let ranks issue collectives of different sizes).
                while delta_pending:
                    sr_, t_ = delta_pending.pop()
                    reduce_info.append(sr_.complete(t_))
                if fr.last_path == "full":
                    fr.touched_mask.fill_(1)
                delta_pending.append((delta_sr, delta_sr.start(fr.light_volume, total_grid, brick_mask=fr.touched_mask)))
            elif world > 1:
                total_grid.copy_(fr.light_volume)
                sharding.allreduce_light_volume(total_grid, transport)

        def barrier():
            while delta_pending:
                sr_, t_ = delta_pending.pop()
                reduce_info.append(sr_.complete(t_))
            ranks_barrier()
        use_graph = False
    else:
        fr = P.PhotonFrame(ctx, vol_np, tf, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR, photon_indices=shard)
        # the frame's records in the two-plane layout (CPM_TRACE_PHOTONS_PLANAR: the same records at other addresses; the brick bin
        # then reads 16 instead of 32 bytes per photon).  --records float8 times the reference's interleaved layout instead.
        planar = fast and args.records == "planar"
        fr.set_planar_records(planar)
        use_graph = args.graph and not fast
        if use_graph:
            fr.capture()
        # the sum of the per-rank grids overlaps the next frame's trace and bin (double-buffered grids,
        # sharding.OverlappedGridReducer); everything outstanding is waited for inside the timed region
        desc = fr.grid if rccl else ((gdim,) * 3, 1)
        exchange_choice = None
        if use_graph or world == 1:
            exchange = "dense"
        elif exchange == "auto":
            # the byte model at set-up (sharding.exchange_model): one probe frame, its lit 4x4x4 bricks counted on every rank and OR-ed over
            # the ranks (gloo: every rank ends with the same numbers and takes the same decision)
            exchange = "union"
            if lists_possible:
                probe_out = torch.empty_like(fr.light_volume)
                fr.trace(); fr.bin_fast(); fr.gather_fast(out=probe_out)
                lit = (sharding.brick_view(probe_out, (gdim,) * 3, 1).reshape(-1, 64) != 0).any(dim=1)
                own = torch.tensor([int(lit.sum().item())], dtype=torch.int64)
                dist.all_reduce(own, op=dist.ReduceOp.MAX)
                um = lit.to(torch.uint8).cpu()
                dist.all_reduce(um, op=dist.ReduceOp.MAX)
                model = sharding.exchange_model(int(lit.numel()), 1, world, int(um.sum().item()), int(own.item()), gdim ** 3, **p2p_kw)
                exchange = "lists" if model["brick_lists"]["model_us"] < model["union_reduce"]["model_us"] else "union"
                exchange_choice = {"chosen": exchange, "from": model, "lit_bricks_max_per_rank": int(own.item()), "union_bricks": int(um.sum().item())}
                del probe_out
        if exchange == "lists" and rccl:
            # (send / receive pairs have not run over this communicator yet: a small exchange now, so that a transport that cannot do them
            # is found here -- every rank then takes the union form -- and not in the middle of the frames)
            ok = True
            try:
                pd = B.default_grid_desc((16, 16, 16), 1)
                pbr = ctx.bricklist_reduce_create(transport.comm, pd, 0 if root is None else root)
                pg = torch.full((16 ** 3,), 1.0, dtype=torch.float32, device=ctx.device)
                with torch.cuda.stream(transport.stream):
                    pbr.complete(pbr.start(pg))
                transport.stream.synchronize()
                ok = bool((pg == float(world)).all().item()) if rank == (0 if root is None else root) else True
                pbr.close()
            except Exception as e:  # noqa: BLE001
                ok = False
                transport_note = f"brick lists not usable over this transport ({e}): union of bricks instead"
            if not agree(ok):
                exchange = "union"
                if exchange_choice:
                    exchange_choice["chosen"] = "union (the list exchange's probe failed on some rank)"
        reducer = sharding.OverlappedGridReducer(fr.light_volume, transport, sparse=desc if exchange == "union" else None,
                                                 lists=desc if exchange == "lists" else None, root=0 if root is None else root)
        # --frames-in-flight 2: a second frame object (own context: the library's scratch is per context; own buffers, own stream) takes the odd
        # frames; the reducer's two buffers already alternate, so each buffer -- and each ticket's exchange -- stays with one of the two streams
        in_flight = args.frames_in_flight if (fast and not use_graph) else 1
        if world > 1:
            in_flight = min(in_flight, 2)   # (the exchange's buffers are two)
        lanes = [(fr, torch.cuda.current_stream())]
        if in_flight > 1:
            lanes = [(fr, torch.cuda.Stream(device=ctx.device))]
            for _ in range(in_flight - 1):
                cj = B.Context(local_rank)
                fj = P.PhotonFrame(cj, vol_np, tf, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR, photon_indices=shard)
                fj.set_planar_records(planar)
                lanes.append((fj, torch.cuda.Stream(device=ctx.device)))
        segments = exchange == "lists" and rccl   # the step-by-step form: a sender may gather straight into its segment
        reducer_root = 0 if root is None else root
        sender_gather = None
        if segments:
            # which form this rank's frame takes is measured here, on this rank, with its shard (both fill the same segment: a rank-local choice)
            mine = sharding.choose_sender_gather(fr, reducer.transport.lists.n_bricks) if rank != reducer_root else {"chosen": "root: dense volume"}
            reducer.use_segments = mine["chosen"] != "pack"
            rows = [None] * world
            dist.all_gather_object(rows, mine)
            sender_gather = {"per_rank": rows, "note": "frame time (trace + bin + gather [+ pack launch]) in us, measured on each rank at set-up: cpm_gather_fast_segment "
                                                       "against cpm_gather_fast_marked + the pack launch; the faster form is what the rank's frames run"}
        frame_no = [0]

        def step():
            k = frame_no[0]
            frame_no[0] += 1
            if use_graph:
                k = 0  # the captured gather writes frame.light_volume (= buffer 0): no double buffering under replay
                reducer.acquire(k)
                fr.replay()
            elif fast:
                f, st = lanes[k % len(lanes)]
                with torch.cuda.stream(st):
                    f.trace()
                    f.bin_fast()
                    if world == 1:                     # (no exchange: every frame object gathers into its own light volume)
                        f.gather_fast()
                        return
                    out = reducer.acquire(k)
                    seg = reducer.segment_for(k)       # brick lists over the C-ABI: a rank that is not the display GPU has no dense volume at all --
                    if seg is not None:                # its gather writes the non-zero 4x4x4 bricks into the ticket's segment (cpm_gather_fast_segment)
                        f.gather_fast_segment(seg)
                        reducer.reduce(k)
                        return
                    # (the union reduce takes the gather's marks; so does a brick-list exchange that starts from a dense volume: the torch twin,
                    # or a sender whose set-up measurement chose the pack launch)
                    marks = reducer.marks_for(k) if (reducer.sparse or (reducer.lists and not (segments and rank == reducer_root))) else None
                    f.gather_fast(out=out, nonzero_bricks=marks)
                    reducer.reduce(k, marked=marks is not None)
                    return
            else:
                fr.trace()
                fr.bin()
                fr.gather(out=reducer.acquire(k))
            reducer.reduce(k)

        def barrier():
            reducer.flush()
            ranks_barrier()

    # W untimed steps, then TIMED_BATCHES batches of exactly K steps, each between two barrier + synchronize pairs and taken as the
    # MAX over the ranks; ms_per_step / value are the MEDIAN batch's (K = 20 frames of config 2 last 1.4 ms: one batch is one sample of
    # the box's clocks; every batch's figure is in the line)
    for _ in range(args.warmup):
        step()
    # (a fresh box's clocks are still rising after W = 5 frames -- 0.3 ms of work: the first batches of a short run read 2 - 3 % slow.
    # Untimed frames up to MIN_WARM_STEPS in all; the K timed steps of every batch are untouched)
    for _ in range(max(0, MIN_WARM_STEPS - args.warmup)):
        step()
    batch_elapsed = []
    for _ in range(TIMED_BATCHES):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        batch_elapsed.append(dt)
    elapsed = sorted(batch_elapsed)[len(batch_elapsed) // 2]
    # (N > 1: the last timed frame's sum, as the exchange left it at the display GPU -- held against an independent dense sum below)
    final_sum = None
    if world > 1 and not correlated and rank == (0 if root is None else root):
        final_sum = reducer.result(frame_no[0] - 1).detach().clone()

    # ---- second pass with per-kernel HIP events (library hook) on the same stream
    prof_steps = min(args.steps, 100)
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(prof_steps):
        if correlated:
            step()
        else:
            fr.trace()
            if fast:
                fr.bin_fast(); fr.gather_fast()
            else:
                fr.bin(); fr.gather()
    kern = ctx.profile_collect()
    ctx.profile_enable(False)
    woodcock_steps = None
    extras = {}
    if not correlated:
        # one more trace with the Woodcock iteration counter on (statistics; its atomics are not timed)
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        ctx.set_step_counter(counter)
        fr.trace()
        torch.cuda.synchronize()
        ctx.set_step_counter(None)
        woodcock_steps = int(counter.item())

    if not correlated:
        fr.set_planar_records(False)   # (the extras below read fr.photons as float8 records: other formulations, the identity checks)
    # ---- labelled extras (never `value`), rank 0 of a single-GPU run only
    if world == 1 and not args.no_extras and not correlated:
        reps = max(10, min(args.steps, 50))
        # the other formulations of the same frame on the same photons
        other = (lambda: (fr.trace(), fr.bin(), fr.gather())) if fast else (lambda: (fr.trace(), fr.bin_fast(), fr.gather_fast()))
        extras["other_formulation"] = {"name": "exact" if fast else "fast", "ms_per_frame": round(timed(torch, other, reps), 4)}
        tmp = torch.zeros_like(fr.light_volume)
        extras["reference_formulation_splat"] = {"ms_per_frame": round(timed(torch, lambda: (fr.trace(), fr.splat(tmp)), reps), 4),
                                                 "splat_only_ms": round(timed(torch, lambda: fr.splat(tmp), reps), 4),
                                                 "note": "trace + clear + atomic splat (the reference's own formulation, cpm_splat); order-dependent sums"}
        # the same frame with the light samples evaluated inside the tracer (cpm_trace_emitted: a caller that owns its light and
        # intersects the volume's box needs no sample / intersection buffers -- 40 of the 48 input bytes per photon not read);
        # photons bit-identical.  Not `value`: the reference's tracer takes sample buffers, and so does the headline.
        if fast:
            fe = P.PhotonFrame(ctx, fr.vol, fr.tf, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR, emit_in_tracer=True)
            ms_e = timed(torch, fe.frame_fast, reps)
            extras["samples_emitted_in_tracer"] = {"ms_per_frame": round(ms_e, 4), "mphotons_per_s": round(n_rank / ms_e / 1e3, 2),
                                                   "trace_ms": round(timed(torch, fe.trace, reps), 4),
                                                   "photons_identical": bool(torch.equal(fe.photons.view(torch.int32), fr.photons.view(torch.int32)))}
            del fe
        # the launch order's gain when the order is STALE: costs measured under the base TF, the trace timed under the edited
        # TF (config 3's edit) without re-measuring -- what the first frames after an edit see -- then with a fresh order, and in lattice order
        if fast and args.workload == "config2":
            fo = P.PhotonFrame(ctx, fr.vol, tf, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR)
            fo.trace(); fo.trace()
            fo.tf.update(S.workspace_tf(moved_point4=0.26))
            fo._traces_since_order = 1                       # (no re-measure: the order stays the base TF's)
            t_stale = timed(torch, fo.trace, reps)
            fo._traces_since_order = 0                       # measure + re-sort under the edited TF
            fo.trace()
            t_fresh = timed(torch, fo.trace, reps)
            fo.adaptive_order = False
            t_lattice = timed(torch, fo.trace, reps)
            extras["trace_order_after_tf_edit"] = {"stale_order_ms": round(t_stale, 4), "fresh_order_ms": round(t_fresh, 4), "lattice_order_ms": round(t_lattice, 4),
                                                   "note": "trace alone under config 3's edited TF: order measured under the base TF / re-measured / none"}
            del fo
        # config 2 at I = 4 (multiple scattering, Henyey-Greenstein g = 0.3): N photons, up to 4 records each
        f4 = P.PhotonFrame(ctx, fr.vol, fr.tf, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR, max_interactions=4,
                           material=(0.3, 0.0, 0.0, 0.0))
        run4 = (lambda: (f4.trace(), f4.bin_fast(), f4.gather_fast())) if fast else (lambda: (f4.trace(), f4.bin(), f4.gather()))
        ms4 = timed(torch, run4, reps)
        stored4 = int((f4.photons[:, 0] != 3.402823466e+38).sum().item())
        extras["i4"] = {"max_interactions": 4, "ms_per_frame": round(ms4, 4), "mphotons_per_s": round(n_rank / ms4 / 1e3, 2),
                        "records_stored": stored4, "mrecords_per_s": round(stored4 / ms4 / 1e3, 2)}
        del f4
        # the tracer where it IS the frame: a sparse transfer function (~100 Woodcock steps per photon)
        sparse = S.homogeneous_tf(0.01)
        fs = P.PhotonFrame(ctx, fr.vol, sparse, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR)
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        ctx.set_step_counter(counter)
        fs.trace()
        torch.cuda.synchronize()
        ctx.set_step_counter(None)
        ms_s = timed(torch, fs.trace, 10)
        extras["sparse_tf_trace"] = {"tf": "constant alpha 0.01", "steps_per_photon": round(int(counter.item()) / n_rank, 2),
                                     "trace_ms": round(ms_s, 4), "gsamples_per_s": round(int(counter.item()) / ms_s / 1e6, 3)}
        del fs
        if args.workload == "config2":
            # BASELINE configs 3 and 5 on this GPU.  Timed through the C++ Processor/Port layer (libcpm_host.so: what an Inviwo
            # evaluation of the workspace's network costs, wall clock from the edit to an idle device), medians of >= 50
            # updates; the Python driver's figure for the same update beside it.
            hostlayer = importlib.import_module(cpm_amd.__name__ + ".hostlayer")
            hl = hostlayer.load()
            base_pts = list(S.WORKSPACE_TF_POINTS)
            edit = list(base_pts)
            edit[3] = (0.26,) + base_pts[3][1:]
            dnorm = P._normalize(LIGHT_DIR)
            lpos = np.array([0.5, 0.5, 0.5], np.float32) - np.float32(2.0) * dnorm
            reps_c = 60
            net = hostlayer.HostNetwork(hl, vol_np, nx, lpos, dnorm, base_pts, size_option=vdim // gdim, correlated=True)
            net.evaluate(first=True)
            full_ms = net.bench_full_frames(reps_c)
            # the headline's frame through the product's boundary -- the C++ Processor/Port layer: frames enqueued back to back
            # with one synchronisation (throughput: what `value` measures through the Python driver) and every frame from an idle
            # device until it is idle again (latency: what one Inviwo evaluation costs)
            net.bench_frames_back_to_back(20)
            thr_ms, enq_ms = net.bench_frames_back_to_back(max(100, args.steps))
            extras["host_network"] = {
                "throughput_ms_per_frame": round(thr_ms, 4), "mphotons_per_s": round(n_rank / thr_ms / 1e3, 2),
                "latency_ms_from_idle": round(float(np.median(full_ms[10:])), 4), "host_enqueue_ms_per_frame": round(enq_ms, 4),
                "measured": "libcpm_host.so: ProgressivePhotonTracerCL + PhotonToLightVolumeProcessorCL of the workspace's network (config 2, importance grid "
                            "connected), full frames: invalidate -> tracer.process() -> lightVolume.process()"}
            # as shipped: importanceBranchPolicy = adaptive -- the tracer serves an edit with whichever of importance branch +
            # add-remove and full frame it has measured cheaper (the first edit probes the branch)
            ms_auto, n_auto = net.bench_tf_edits(edit, base_pts, reps_c)
            auto_costs = {k: round(float(v), 4) for k, v in net.path_costs().items()}
            auto_decision = net.last_decision
            net.set_string("tracer", "importanceBranchPolicy", "never")     # ... the same edits served by full frames only
            ms_never, _ = net.bench_tf_edits(edit, base_pts, reps_c)
            net.set_string("tracer", "importanceBranchPolicy", "always")    # ... and by the importance branch only (the reference's behaviour)
            ms, nre = net.bench_tf_edits(edit, base_pts, reps_c)
            net.set_float("tracer", "fusedImportanceBranch", 0.0)      # the launch-by-launch branch with its host wait, for comparison
            net.evaluate()
            ms_legacy, _ = net.bench_tf_edits(edit, base_pts, reps_c)
            net.set_float("tracer", "fusedImportanceBranch", 1.0)
            n_host = net.n_photons
            # break-even: an equal-importance selection of every (100 / p)-th photon (exactly that fraction re-traced) through the
            # same branch, the light volume updated by add-remove and by a rebuild; against the full frame of the same network
            sweep = {}
            net.set_float("tracer", "equalImportance", 1.0)
            for pct in (1, 2, 5, 10, 20, 25, 50, 100):
                net.set_float("tracer", "equalImportancePercentage", float(pct))
                row = {}
                for thr, name in ((100.1, "add_remove_ms"), (0.0, "rebuild_ms")):
                    net.set_float("lightvolume", "incrementalRecomputationThreshold", thr)
                    m, nn = net.bench_tf_edits(edit, base_pts, 14)
                    row[name] = round(float(np.median(m[4:])), 4)
                    row["fraction"] = round(float(nn[-1]) / max(n_host, 1), 4)
                sweep[str(pct)] = row
            net.set_float("tracer", "equalImportance", 0.0)
            net.set_float("tracer", "equalImportancePercentage", 0.0)
            net.set_float("lightvolume", "incrementalRecomputationThreshold", 50.0)
            full_med = float(np.median(full_ms[10:]))
            cheaper = [r["fraction"] for r in sweep.values() if min(r["add_remove_ms"], r["rebuild_ms"]) < full_med]
            break_even = max(cheaper) if cheaper else 0.0
            extras["config3_tf_edit"] = {
                "update_ms": round(float(np.median(ms_auto[10:])), 4),
                "update_ms_p10_p90": [round(float(np.percentile(ms_auto[10:], q)), 4) for q in (10, 90)],
                "served_by": auto_decision, "full_frames_among_updates": round(float(np.mean(n_auto[10:] < 0)), 3),
                "measured_path_costs_ms": auto_costs,
                "importance_branch_update_ms": round(float(np.median(ms[10:])), 4),
                "full_frame_on_edit_ms": round(float(np.median(ms_never[10:])), 4),
                "fraction_retraced": round(float(np.mean(nre[10:])) / max(n_host, 1), 5), "updates": int(reps_c - 10),
                "full_frame_ms_same_network": round(float(np.median(full_ms[10:])), 4),
                "launch_by_launch_update_ms": round(float(np.median(ms_legacy[10:])), 4),
                "by_fraction_retraced": sweep, "largest_fraction_cheaper_than_full_frame": break_even,
                "measured": "libcpm_host.so (C++ processors): TF property edit -> importance grid -> tracer importance branch -> light-volume add-remove, until the device is idle",
                "includes": "TF LUT upload, importance grid, per-photon importance + selection, re-trace, - old / + new splat"}
            net.close()
            # the Python driver on the same update (ctypes launches, one mailbox read at the end)
            cm = P.CorrelatedPhotonMapper(ctx, vol_np, tf, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR, tf_points=base_pts)  # its own volume: time steps replace it
            cm.full_frame()
            res = []
            for rep in range(40):  # alternate edit / revert so that every update re-traces
                pts = edit if rep % 2 == 0 else base_pts
                torch.cuda.synchronize(); ta = time.perf_counter()
                cm.set_transfer_function(pts)
                n = cm.correlated_update()
                torch.cuda.synchronize()
                res.append(((time.perf_counter() - ta) * 1e3, n / cm.n))
            extras["config3_tf_edit"]["python_driver_update_ms"] = round(float(np.median([r[0] for r in res[8:]])), 4)
            # config 5: time steps of the 256^3 sequence.  (a) the Python driver: volume step (difference, min/max, importance)
            # + fused update, volumes resident; (b) the C++ network in its time-varying form (players -> importance -> tracer ->
            # light volume), displayed times a quarter step apart
            n_steps = 8
            seq_np = [S.heterogeneous_volume(vdim, S.sequence_blob_center(t, 32)) for t in range(n_steps)]
            vols = [ctx.volume_create(v) for v in seq_np]
            cm.set_transfer_function(base_pts)
            cm.full_frame()
            res = []
            for k in range(4 * (n_steps - 1)):
                t = 1 + k % (n_steps - 1) if (k // (n_steps - 1)) % 2 == 0 else n_steps - 2 - k % (n_steps - 1)   # forth and back
                torch.cuda.synchronize(); ta = time.perf_counter()
                cm.set_volume(vols[t])
                torch.cuda.synchronize(); tb = time.perf_counter()
                n = cm.correlated_update()
                torch.cuda.synchronize()
                res.append(((tb - ta) * 1e3, (time.perf_counter() - tb) * 1e3, n / cm.n))
            res = res[4:]
            extras["config5_time_step"] = {"steps": len(res), "volume_step_ms": round(float(np.median([r[0] for r in res])), 4),
                                           "update_ms": round(float(np.median([r[1] for r in res])), 4),
                                           "fraction_retraced": round(float(np.mean([r[2] for r in res])), 5),
                                           "note": "Python driver; RESIDENT: time steps on the device as volumes (no upload, copy or re-layout in the step); "
                                                   "volume_step_ms and update_ms each from an idle device"}
            # SURVEY 8(d)'s step: "upload volume, min/max, mean-abs-diff, ...".  The same walk through the sequence (a) RESIDENT, steps enqueued
            # back to back with one synchronisation at the end (the figure the streamed one compares with), (b) STREAMED from pinned host memory:
            # cpm_volume_stream, three device slots, step t + 1 put on the library's copy stream before step t's update is enqueued
            # (round and round through the steps, so that EVERY step needs an upload -- forth and back would find two of them resident at each end)
            walk = [k % n_steps for k in range(1, 6 * n_steps + 1)]

            def run_walk(volume_of, before=None):
                if before is not None:
                    before(-1)          # (the first step's upload is under way when the clock starts, as every later step's is)
                torch.cuda.synchronize(); ta = time.perf_counter()
                for j, t in enumerate(walk):
                    if before is not None:
                        before(j)
                    cm.set_volume(volume_of(j, t))
                    cm.correlated_update()
                torch.cuda.synchronize()
                return (time.perf_counter() - ta) * 1e3 / len(walk)
            run_walk(lambda j, t: vols[t])
            resident_ms = min(run_walk(lambda j, t: vols[t]) for _ in range(3))
            pinned = B.PinnedSequence(ctx, seq_np)
            vstream = B.VolumeStream(ctx, seq_np[0], n_slots=3)

            def ahead(j):   # the step after this one starts crossing PCIe now
                if j + 1 < len(walk):
                    vstream.prefetch(walk[j + 1], pinned.steps[walk[j + 1]])
            run_walk(lambda j, t: vstream.acquire(t, pinned.steps[t]), ahead)
            streamed_ms = min(run_walk(lambda j, t: vstream.acquire(t, pinned.steps[t]), ahead) for _ in range(3))
            torch.cuda.synchronize()
            si = vstream.stats()
            h2d_ms = si.upload_ms_total / max(si.uploads_timed, 1)
            # ... and with the upload IN the step's own stream, not ahead of it (what cpm_volume_update from host memory does): the sum
            own = [ctx.volume_create(seq_np[0]), ctx.volume_create(seq_np[0])]   # (two, used in turn: the mapper still reads the step before)
            inline_ms = run_walk(lambda j, t: (own[j & 1].update(pinned.steps[t]), own[j & 1])[1])
            extras["config5_time_step"].update({
                "resident_step_ms_back_to_back": round(resident_ms, 4),
                "streamed_step_ms": round(streamed_ms, 4),
                "upload_ms": round(h2d_ms, 4), "h2d_gbps": round(si.bytes_per_step / max(h2d_ms, 1e-9) / 1e6, 2),
                "bytes_per_step": int(si.bytes_per_step),
                "upload_hidden_fraction": round(min(1.0, max(0.0, 1.0 - (streamed_ms - resident_ms) / max(h2d_ms, 1e-9))), 3),
                "bound_ms": round(max(h2d_ms, resident_ms), 4),   # max(upload, update): what the streamed step is held to (+ 10 %)
                "upload_in_the_step_unhidden_ms": round(inline_ms, 4),
                "uploads_started_at_the_acquire": int(si.uploads_at_acquire),
                "walk": f"round and round through {n_steps} steps ({len(walk)} steps timed, one synchronisation at the end): every step uploads; the wrap from the last "
                        "step to the first is one large change per round, in the resident and the streamed walk alike",
                "streamed": "cpm_volume_stream: sequence in pinned host memory, 3 device slots (linear block + footprint copy each), H2D + re-layout on the "
                            "library's copy stream while the step before runs; upload_ms = the H2D copy alone (HIP events on the copy stream, mean); "
                            "upload_hidden_fraction = 1 - (streamed - resident) / upload"})
            vstream.close(); pinned.close()
            del cm, vols, own
            seq = hostlayer.HostSequence(hl, np.stack(seq_np))
            net = hostlayer.HostNetwork(hl, seq_np[0], nx, lpos, dnorm, base_pts, size_option=vdim // gdim, correlated=True)
            seq.attach(net)
            net.evaluate(first=True)
            rows = []
            for k in range(1, 4 * (n_steps - 1)):
                rows.append(seq.step(net, 0.25 * k))
            rows = rows[3:]
            totals = [seq.step_total(net, 0.25 * (4 * (n_steps - 1) - k))[1] for k in range(1, 4 * (n_steps - 1))][5:]   # (back down the sequence)
            extras["config5_time_step"]["host_network_step_ms"] = round(float(np.median(totals)), 4)
            # the same network with the sequence KEPT IN HOST MEMORY (VolumeSequencePlayer.keepSequenceOnDevice = false: its ring of three device
            # volumes, the next element prefetched on the library's copy stream): whole elements apart, so that every step needs a new upload
            seq.keep_on_device(False)
            for k in range(1, n_steps):
                seq.step_total(net, float(k))
            tot_s = [seq.step_total(net, float(k % n_steps))[1] for k in range(n_steps, 3 * n_steps)]
            st = seq.stream_stats() or {}
            seq.keep_on_device(True)
            for k in range(0, n_steps):
                seq.step_total(net, float(k))
            tot_r = [seq.step_total(net, float(k % n_steps))[1] for k in range(n_steps, 3 * n_steps)]
            extras["config5_time_step"]["host_network_streamed"] = {
                "step_ms_streamed": round(float(np.median(tot_s)), 4), "step_ms_resident_same_walk": round(float(np.median(tot_r)), 4),
                "upload_ms": round(st.get("upload_ms", 0.0) / max(st.get("uploads", 0), 1), 4), "uploads_at_acquire": st.get("uploads_at_acquire"),
                "note": "each step from an idle device to an idle device (cpmh_sequence_step_total synchronises on both sides): the prefetch of the NEXT "
                        "element is what overlaps here -- it was started during the step before"}
            extras["config5_time_step"]["host_network"] = {
                "steps": len(rows), "displayed_times": "a quarter of a sequence step apart",
                "players_ms": round(float(np.median([r[1] for r in rows])), 4), "update_ms": round(float(np.median([r[2] for r in rows])), 4),
                "fraction_retraced": round(float(np.mean([max(r[0], 0) for r in rows])) / max(net.n_photons, 1), 5),
                "full_frames_among_steps": round(float(np.mean([r[0] < 0 for r in rows])), 3), "served_by": net.last_decision,
                "measured": "libcpm_host.so: VolumeSequencePlayer + 2 UniformGrid3D players (players_ms); importance (time-varying) -> tracer -> light volume (update_ms)"}
            net.close(); seq.close()
            del seq_np
        # the sparse full-frame reduce's own work on ONE GPU (a real RCCL communicator of size 1: mask, list, pack, unpack run;
        # the two collectives are what N > 1 adds): what it costs beside a frame, and what it would put on the wire
        if fast:
            try:
                tr1 = sharding.RcclTransport(ctx, 0, 1)
                red1 = sharding.OverlappedGridReducer(fr.light_volume, tr1, sparse=fr.grid, force=True)
                kf = [0]

                def frame_with_reduce():
                    fr.trace(); fr.bin_fast()
                    fr.gather_fast(out=red1.acquire(kf[0]), nonzero_bricks=red1.marks_for(kf[0]))
                    red1.reduce(kf[0], marked=True); kf[0] += 1
                ms_red = timed(torch, frame_with_reduce, reps)
                ctx.profile_reset(); ctx.profile_enable(True)
                for _ in range(20):
                    frame_with_reduce()
                red1.flush(); torch.cuda.synchronize()
                kk = ctx.profile_collect(); ctx.profile_enable(False)
                last = red1.info[-1]
                extras["sparse_reduce_one_gpu"] = {
                    "ms_per_frame_with_reduce_chain": round(ms_red, 4),
                    "kernel_us": {k: round(tot / calls * 1e3, 2) for k, (tot, calls) in kk.items() if k.startswith("brick_") or "brick_" in k.split("<")[0]},
                    "n_bricks_4x4x4": last["n_bricks"], "n_union": last["n_union"], "capacity": last["capacity"], "mode": last["mode"],
                    "reduce_bytes_per_frame": last["reduce_bytes"], "dense_bytes": last["dense_bytes"],
                    "fraction_of_dense": round(last["reduce_bytes"] / last["dense_bytes"], 4),
                    "host_waits": "none on the stream: the union count reaches the host through a pinned mailbox two frames later",
                    "note": "communicator of size 1 on this GPU: pack / unpack / mask / list launches measured, the two collectives are no-ops"}
                tr1.close()
            except Exception as e:  # noqa: BLE001
                extras["sparse_reduce_one_gpu"] = {"error": str(e)[:200]}
        # the workspace's own operating point through the C++ processors (workspaces/CorrelatedPhotonMappingSingleVolume.inv: two
        # directional lights on the tracer's multi-inport, 2 x 1024^2 samples, a 512 x 512 x 96 volume clipped as its proxy
        # geometry is, a light volume of half the size): full frames and BASELINE config 3's TF edit on it
        if args.workload == "config2" and fast:
            try:
                hostlayer = importlib.import_module(cpm_amd.__name__ + ".hostlayer")
                hl = hostlayer.load()
                wl = []
                for w in ((-90.045471, 104.828, 312.07489), (94.269867, 148.44716, 302.45557)):   # the lights' world positions (:214,1089)
                    dd = P._normalize(tuple(-x for x in w))
                    wl.append((np.array([0.5, 0.5, 0.5], np.float32) - np.float32(2.0) * dd, dd))
                vol_w = S.heterogeneous_volume((512, 512, 96))
                base_pts = list(S.WORKSPACE_TF_POINTS)
                edit = list(base_pts)
                edit[3] = (0.26,) + base_pts[3][1:]
                net = hostlayer.HostNetwork(hl, vol_w, 1024, wl[0][0], wl[0][1], base_pts, size_option=2, correlated=True)
                net.add_light(*wl[1])
                net.set_clip(73, 512, 7, 512, 0, 96)
                net.evaluate(first=True)
                net.bench_frames_back_to_back(10)
                thr_w, _ = net.bench_frames_back_to_back(60)
                lat_w = float(np.median(net.bench_full_frames(40)[10:]))
                kern_w = net.profile_full_frames(30)
                ms_e, n_e = net.bench_tf_edits(edit, base_pts, 40)
                extras["workspace_point"] = {
                    "ms_per_frame": round(thr_w, 4), "mphotons_per_s": round(2 * 1024 * 1024 / thr_w / 1e3, 2),
                    "latency_ms_from_idle": round(lat_w, 4), "photons_per_frame": 2 * 1024 * 1024, "lights": 2,
                    "volume": [512, 512, 96], "light_volume": [256, 256, 48], "clip": [73, 512, 7, 512, 0, 96],
                    "candidate_box": "6 x 6 x 2 voxels (r = 2.8 / 2.8 / 0.5 light-volume voxels: |indexToTexture (1, 1, 1)| of a 4 : 4 : 0.75 volume)",
                    "kernel_ms_per_frame": {k: round(v, 5) for k, v in sorted(kern_w.items(), key=lambda kv: -kv[1])},
                    "tf_edit_update_ms": round(float(np.median(ms_e[10:])), 4),
                    "tf_edit_fraction_retraced": round(float(np.mean(np.maximum(n_e[10:], 0))) / (2 * 1024 * 1024), 5),
                    "tf_edit_served_by": net.last_decision,
                    "launches": "both lights' samples traced by one launch (cpm_trace_lights; the TF edit's importance pass + re-trace likewise: "
                                "cpm_photon_importance_retrace_lights); the wide box's gather: a photon filed under one brick of 16 x 16 x 8 voxels, 64-bit LDS tiles with "
                                "a halo (two copies), staged per brick and merged by a second launch",
                    "measured": "libcpm_host.so: the workspace's network (two light samplers -> tracer multi-inport -> light volume), frames back to "
                                "back with one synchronisation; tests/test_workspace_point_gpu.py holds the same network to the oracle"}
                net.close()
                del vol_w
            except Exception as e:  # noqa: BLE001
                extras["workspace_point"] = {"error": str(e)[:300]}
        # Frames in flight: S independent frames on S streams (each frame owns its context, buffers and stream; --frames-in-flight runs the
        # timed region itself that way).  A frame is four dependent launches of 10 - 30 us, each with its ramp and tail: a second frame's
        # launches fill what the first leaves idle.  Never `value` (one stream, as in every round): the labelled throughput of the same frames.
        if args.streams > 1:
            most = max(args.streams, 3)
            ctxs = [B.Context(local_rank) for _ in range(most)]
            frames = [P.PhotonFrame(c, vol_np, tf, lattice, (gdim,) * 3, light_travel_direction=LIGHT_DIR) for c in ctxs]
            streams = [torch.cuda.Stream() for _ in range(most)]
            for f in frames:
                f.set_planar_records(fast and args.records == "planar")

            def run(S, n_frames):
                for it in range(n_frames):
                    f, st = frames[it % S], streams[it % S]
                    with torch.cuda.stream(st):
                        f.trace()
                        if fast:
                            f.bin_fast(); f.gather_fast()
                        else:
                            f.bin(); f.gather()
            by = {}
            n_timed = max(args.steps, 60)
            for S in sorted({2, 3, args.streams}):
                run(S, 120)
                samples = []
                for _ in range(3):
                    torch.cuda.synchronize(); tp = time.perf_counter()
                    run(S, n_timed)
                    torch.cuda.synchronize()
                    samples.append((time.perf_counter() - tp) / n_timed)
                dtp = sorted(samples)[1]
                by[str(S)] = {"value": round(n_rank / dtp / 1e6, 2), "ms_per_frame": round(dtp * 1e3, 4)}
            if fast:
                fr.frame_fast()
            else:
                fr.frame()
            same = all(bool(torch.equal(f.light_volume, fr.light_volume)) for f in frames)
            best = max(by, key=lambda k: by[k]["value"])
            extras["pipelined"] = {"streams": int(best), "frames": n_timed, "value": by[best]["value"], "unit": "Mphotons/s",
                                   "ms_per_frame": by[best]["ms_per_frame"], "by_streams": by,
                                   "light_volumes_identical_to_single_stream": same,
                                   "note": "throughput of the same frames with several in flight (own stream, context and buffers each; median of 3 batches); "
                                           "`value` above is one stream -- a frame's latency and the per-kernel roofline are that run's"}
            del frames, ctxs

    # what every rank's light volume lights, for the exchange's byte model (all ranks: the counts are gathered)
    exchange_rows = None
    if not correlated and world > 1 and fast:
        tmp = torch.empty_like(fr.light_volume)
        fr.trace(); fr.bin_fast(); fr.gather_fast(out=tmp)
        lit = (sharding.brick_view(tmp, (gdim,) * 3, 1).reshape(-1, 64) != 0).any(dim=1)
        tail = reducer.info[-args.steps:] if reducer.info else []
        mine = torch.tensor([float(lit.sum().item()),
                             float(np.median([i.get("sent_bytes", 0) for i in tail])) if tail else 0.0,
                             float(np.median([i.get("received_bytes", 0) for i in tail])) if tail else 0.0,
                             float(sum(i.get("resent", 0) for i in tail))], dtype=torch.float64)
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        union = lit.to(torch.uint8).cpu()
        dist.all_reduce(union, op=dist.ReduceOp.MAX)
        exchange_rows = {"per_rank": [r.tolist() for r in rows], "union": int(union.sum().item()), "n_bricks": int(lit.numel())}
        # self-check of the exchange that was timed: the display GPU's last summed volume against the sum of every rank's own dense volume carried by
        # gloo over host memory (nothing of RCCL or of the brick bookkeeping in it); the photons do not change from frame to frame
        ref = tmp.detach().cpu()
        dist.reduce(ref, dst=0 if root is None else root)
        if final_sum is not None:
            got = final_sum.cpu()
            peak = float(ref.abs().max())
            worst = float((got - ref).abs().max())
            exchange_rows["self_check"] = {"max_abs_difference": worst, "volume_max": peak, "relative": worst / max(peak, 1e-30),
                                           "ok": bool(worst <= 2e-5 * peak), "nonzero_voxels": int((got != 0).sum().item()),
                                           "against": "the ranks' own dense volumes summed by gloo over host memory (float sums in another order: tolerance 2e-5 of the maximum)"}
        del tmp
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed / 1e6
        per_frame = {k: (tot / prof_steps, calls / prof_steps, tot / calls) for k, (tot, calls) in kern.items()}

        def stage(names):
            return round(sum(per_frame[k][0] for k in per_frame if any(k.startswith(n) for n in names)), 4)

        stages = {"trace": stage(["trace_kernel"]),
                  "bin": stage(["bin_", "radix_", "cell_start", "fast_count", "fast_scan", "fast_scatter"]),
                  "gather": stage(["gather", "fast_brick", "fast_halo"]),
                  "reduce": stage(["rccl_"])}
        tile = 256 * (4 if n_rank <= (1 << 15) else 8 if n_rank <= (1 << 23) else 16)
        tiles = -(-n_rank // tile)
        ab = algorithmic_bytes(n_rank, 1, vol_np.size, tf.shape[0], gdim ** 3, 1, tiles)
        path_kernels = {k: v for k, v in per_frame.items() if k.split("<")[0].replace("cpm::", "") in ab or k.startswith("gather")}
        dom = max(path_kernels or per_frame, key=lambda k: per_frame[k][0])
        dom_base = dom.split("<")[0].replace("cpm::", "")
        if dom_base.startswith("gather"):
            dom_base = "gather_kernel"
        dom_avg_ms = per_frame[dom][2]
        achieved = ab.get(dom_base, 0) / (dom_avg_ms * 1e-3) / 1e9
        if fast:
            frame_bytes = sum(ab[k] for k in ("trace_kernel", "fast_count_kernel", "fast_scatter_kernel", "fast_brick_kernel"))
        else:
            passes = -(-int(gdim ** 3).bit_length() // 8)
            frame_bytes = (ab["trace_kernel"] + ab["bin_keys_kernel"] + passes * (ab["radix_hist_kernel"] + ab["radix_rowscan_kernel"] + ab["radix_scatter_kernel"])
                           + ab["bin_finalize_kernel"] + ab["cell_start_kernel"] + ab["gather_kernel"])
        traffic, traffic_src = pmc_traffic(dom.split("<")[0], args.workload)
        what = {"fast": "brick bin + one-launch fixed-point LDS-tile gather (cpm_bin_fast + cpm_gather_fast: tolerance mode, rtol 2e-5 / atol 1e-5 max "
                        "vs the reference semantics, bitwise reproducible)",
                "exact": "cell sort + sequential per-voxel gather (cpm_bin + cpm_gather: bit-exact contract)"}[args.formulation]
        out = {
            "metric": "Mphotons/s traced+binned+gathered",
            "value": round(value, 2), "unit": "Mphotons/s", "n_gpus": world, "steps": args.steps,
            # the untimed frames that RAN before the first timed batch (never fewer than MIN_WARM_STEPS), and what --warmup asked for
            "warmup": max(args.warmup, MIN_WARM_STEPS), "warmup_requested": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "timing": {"what": f"median of {TIMED_BATCHES} timed batches of {args.steps} steps each (every batch between barrier + synchronize pairs, "
                               "MAX over ranks); value and ms_per_step are that batch's; untimed frames before the first batch: "
                               f"{max(args.warmup, MIN_WARM_STEPS)} (--warmup {args.warmup}, at least {MIN_WARM_STEPS}: a fresh box's clocks are still rising after 5)",
                       "batch_ms_per_step": [round(e / args.steps * 1e3, 4) for e in batch_elapsed]},
            "config": {"workload": (f"BASELINE {args.workload}: {vdim}^3 u8 heterogeneous volume, {n_total} photons per frame in all "
                                    f"({lattice[0]}x{lattice[1]} lattice, one directional light; {n_rank} on rank 0), {gdim}^3 x1 f32 light volume, "
                                    f"I=1, r=1 voxel, MWC64X streams from glibc srand(0)"
                                    + ("; a step = one time step of the 32-step sequence: " + ("UPLOAD of the step from pinned host memory (cpm_volume_stream, "
                                       "ahead on the copy stream), " if (correlated and args.sequence == "streamed") else "(sequence resident on the device) ")
                                       + "volume difference / min-max / importance, correlated re-trace, delta light-volume update" if correlated else "")),
                       "formulation": what, "photons_per_frame": n_total, "photons_rank0": n_rank, "volume": [vdim] * 3,
                       "light_volume": [gdim] * 3,
                       "parallelism": (f"photon-sharded x{world} ({scaling} scaling), one "
                                       + ("touched-brick reduce (cpm_allreduce_grid_sparse with the touched-brick mask)" if correlated else
                                          ({"union": "sparse ", "lists": "brick-list ", "dense": ""}[exchange] + ("all-reduce" if root is None else "reduce to rank 0")
                                           + " of the grid per frame (" + {"union": "cpm_allreduce_grid_sparse", "lists": "cpm_reduce_grid_bricklists: every rank's own non-zero bricks, one send per rank",
                                                                           "dense": "cpm_allreduce_grid" if root is None else "cpm_reduce_grid"}[exchange])
                                          + ": RCCL on a side stream), overlapped with the next "
                                          "frame's trace + bin (double-buffered grid)")
                                       + f", transport {type(transport).__name__}"
                                       + (f" [{transport_note}]" if transport_note else "")) if world > 1 else "single GPU",
                       "launch": "captured HIP graph replay" if use_graph else "eager launches",
                       "frames_in_flight": (in_flight if not correlated else 1),
                       "photon_records": ("two-plane layout (CPM_TRACE_PHOTONS_PLANAR: the same records, position + first power channel in one plane; "
                                          "the brick bin reads 16 of a record's 32 bytes)" if (not correlated and planar) else
                                          "float8 records (the reference's layout, cl/photon.cl:49-63)"),
                       "trace_workgroup_order": "costliest chunks first, from the costs a measured launch recorded (cpm_trace_order: the first "
                                                "frame and every 256th are measured)" if getattr(fr, "adaptive_order", False) else "lattice order",
                       "shards": (f"4096-sample lattice tiles dealt round-robin (rank r: tiles t = r mod {world})" if shards_kind == "tiles"
                                  else "contiguous photon ranges (slabs of the light plane)") if world > 1 else "one shard",
                       "exchange": exchange if world > 1 else "none",
                       "sender_gather": sender_gather if (world > 1 and not correlated) else None,
                       "exchange_chosen_by": (exchange_choice if (world > 1 and not correlated and exchange_choice) else "flag / not applicable"),
                       # the transport the reduce really used, and the size RCCL itself reports for the communicator
                       # (cpm_comm_size; 0 = the reduce did not go through the C-ABI's RCCL communicator)
                       "transport": type(transport).__name__,
                       "rccl_ranks": transport.comm.size if rccl else 0, **({"transport_probes": probes} if probes else {})},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         # also against what a copy kernel reaches on this part (6.29 TB/s, MI355X_MICROARCH.md)
                         "copy_ceiling": HBM_COPY_GBS, "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ab.get(dom_base, 0), "avg_launch_ms": round(dom_avg_ms, 5),
                         "launches_per_frame": round(per_frame[dom][1], 2)},
            "frame": {"kernel_ms_per_frame": {k: round(v[0], 5) for k, v in sorted(per_frame.items(), key=lambda kv: -kv[1][0])},
                      "stage_ms": stages, "sum_kernel_ms": round(sum(v[0] for v in per_frame.values()), 4),
                      "kernel_times_from": "a second pass of the same frames with a HIP-event pair around every launch (about 1 us more per launch "
                                           "than in the timed region: sum_kernel_ms exceeds ms_per_step by that)",
                      "algorithmic_bytes_per_frame": frame_bytes,
                      "frame_hbm_frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if woodcock_steps is not None:
            out["frame"]["woodcock_steps_per_frame"] = woodcock_steps
            out["frame"]["gsamples_per_s"] = round(woodcock_steps / max(stages["trace"], 1e-9) / 1e6, 3)
        if correlated:
            out["frame"]["fraction_retraced"] = round(float(np.mean(fractions[-args.steps:])), 5)
        infos = None
        if correlated and reduce_info:
            infos = [{"n_union": i.n_union, "capacity": i.capacity, "mode": i.mode, "reduce_bytes": i.reduce_bytes, "dense_bytes": i.dense_bytes,
                      "n_bricks": i.n_bricks} for i in reduce_info]
        elif not correlated and reducer.sparse and reducer.info:
            infos = reducer.info
        if infos:
            tail = infos[-args.steps:]
            out["reduce"] = {"kind": "sparse: union of the ranks' " + ("touched" if correlated else "non-zero") + " 4x4x4 bricks (cpm_allreduce_grid_sparse)",
                             "reduce_bytes_per_frame": int(np.median([i["reduce_bytes"] for i in tail])),
                             "dense_bytes": int(tail[-1]["dense_bytes"]), "n_bricks": int(tail[-1]["n_bricks"]),
                             "n_union_median": int(np.median([i["n_union"] for i in tail])),
                             "capacity_median": int(np.median([i["capacity"] for i in tail])),
                             "frames_sparse": int(sum(i["mode"] == 0 for i in tail)), "frames_dense_by_policy": int(sum(i["mode"] == 1 for i in tail)),
                             "frames_dense_after_overflow": int(sum(i["mode"] == 2 for i in tail)),
                             "stream_synchronisations_per_frame": 0}
        elif world > 1 and not correlated and reducer.lists and exchange_rows:
            pr = exchange_rows["per_rank"]
            out["reduce"] = {"kind": "brick lists: every rank's own non-zero 4x4x4 bricks to rank 0 (cpm_reduce_grid_bricklists)",
                             "sent_bytes_per_rank_per_frame": [int(r[1]) for r in pr], "received_bytes_at_root_per_frame": int(pr[0][2]),
                             "reduce_bytes_per_frame": int(max(r[1] for r in pr)),   # the busiest link: a sender's one segment
                             "exchanges_repeated_at_exact_size": int(sum(r[3] for r in pr[1:])),
                             "dense_bytes": gdim ** 3 * 4, "n_bricks": exchange_rows["n_bricks"], "stream_synchronisations_per_frame": 0}
        elif world > 1:
            out["reduce"] = {"kind": "dense", "reduce_bytes_per_frame": gdim ** 3 * 4, "dense_bytes": gdim ** 3 * 4}
        if exchange_rows and "self_check" in exchange_rows:
            out.setdefault("reduce", {})["self_check"] = exchange_rows["self_check"]
            if not exchange_rows["self_check"]["ok"]:
                print("bench.py: the summed light volume differs from the dense reference sum: " + json.dumps(exchange_rows["self_check"]), file=sys.stderr)
        if exchange_rows:
            counts = [int(r[0]) for r in exchange_rows["per_rank"]]
            out.setdefault("reduce", {})["bricks"] = {"lit_per_rank": counts, "union": exchange_rows["union"], "of": exchange_rows["n_bricks"]}
            out["reduce"]["model"] = dict(sharding.exchange_model(exchange_rows["n_bricks"], 1, world, exchange_rows["union"], max(counts), gdim ** 3, **p2p_kw),
                                          measured=p2p,
                                          note="bytes on a rank's busiest xGMI link per frame and a modelled time for the three forms from this run's brick "
                                               "counts: messages x latency + bytes / link rate, with the constants under `constants` -- measured at set-up "
                                               "(`measured`: a 1 KB and a 2 MB ping-pong over the communicator) when the C-ABI transport ran, else assumed")
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline and not correlated:
            out["cpu_baseline"] = cpu_baseline(args.workload, vol_np, tf, lattice, gdim)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
