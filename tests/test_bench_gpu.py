"""bench.py end to end on the GPU: the single-GPU line, and the N > 1 code path (photon shards, overlapped grid
reduction, max-over-ranks timing, rank-0 JSON) with two ranks sharing the one GPU of the test box over gloo."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "10", "--warmup", "2", "--workload", "config1", "--streams", "2"],
                       capture_output=True, text=True, timeout=900, cwd=str(REPO))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and {"cpu_baseline", "pipelined", "frame", "other_formulation", "reference_formulation_splat", "i4",
                               "sparse_tf_trace"} <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 200 and d["warmup_requested"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "cpm_bin_fast" in d["config"]["formulation"] and d["other_formulation"]["name"] == "exact"
    assert any(k.startswith("fast_brick_kernel") for k in d["frame"]["kernel_ms_per_frame"])
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert d["pipelined"]["light_volumes_identical_to_single_stream"] is True
    sp = d["sparse_reduce_one_gpu"]
    assert "error" not in sp, sp
    assert sp["n_union"] <= sp["n_bricks_4x4x4"] and sp["reduce_bytes_per_frame"] > 0 and sp["dense_bytes"] == 32 ** 3 * 4


def test_bench_two_ranks_code_path():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "config1",
           "--test-backend", "gloo", "--test-one-device", "--exchange", "union"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
    assert "reduce to rank 0" in d["config"]["parallelism"] and d["scaling"] == "weak"      # the north-star's reduce to the display GPU
    assert d["config"]["exchange"] == "union" and "tiles" in d["config"]["shards"]             # weak scaling: tile shards, union of bricks
    assert len(d["timing"]["batch_ms_per_step"]) == 7 and d["reduce"]["bricks"]["union"] >= max(d["reduce"]["bricks"]["lit_per_rank"])
    assert set(d["reduce"]["model"]) >= {"dense_reduce", "union_reduce", "brick_lists"}
    assert d["config"]["photons_rank0"] == 65536 and d["config"]["photons_per_frame"] == 131072   # weak: the per-rank work is fixed
    # the default reduce is the sparse one (here carried out by torch ops over gloo): no stream synchronisation on the frame's path
    assert "sparse" in d["config"]["parallelism"] and d["reduce"]["stream_synchronisations_per_frame"] == 0
    assert d["reduce"]["frames_sparse"] + d["reduce"]["frames_dense_by_policy"] + d["reduce"]["frames_dense_after_overflow"] == 6


def test_bench_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment -- the form the driver uses: the parent starts the two
    ranks itself (child processes), relays rank 0's line and reports n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "config1",
           "--test-backend", "gloo", "--test-one-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]            # ONE JSON line on stdout
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["photons_per_frame"] == 131072 and d["config"]["photons_rank0"] == 65536
    assert "tiles" in d["config"]["shards"] and d["config"]["transport"] == "TorchTransport" and d["config"]["rccl_ranks"] == 0
    # --exchange auto: the byte model decided from a probe frame's brick counts, the same decision on both ranks
    chosen = d["config"]["exchange_chosen_by"]
    assert d["config"]["exchange"] in ("union", "lists") and chosen["chosen"] == d["config"]["exchange"]
    assert set(chosen["from"]) >= {"dense_reduce", "union_reduce", "brick_lists"} and chosen["union_bricks"] >= chosen["lit_bricks_max_per_rank"] > 0


def test_bench_strong_scaling_and_exact_formulation():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", str(REPO / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "config1",
           "--scaling", "strong", "--formulation", "exact", "--test-backend", "gloo", "--test-one-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = _last_json(r.stdout)
    assert d["scaling"] == "strong" and d["config"]["photons_per_frame"] == 65536 and d["config"]["photons_rank0"] == 32768
    assert "cpm_bin + cpm_gather" in d["config"]["formulation"]
    assert any(k.startswith("gather") for k in d["frame"]["kernel_ms_per_frame"])


def test_bench_single_process_rccl_world1():
    """torch.distributed.run with one rank: the RCCL transport (cpm_comm through the C-ABI) is constructed and the line is
    the single-GPU line."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29535", str(REPO / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--workload", "config1",
           "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0


def test_bench_two_ranks_through_the_c_abi_reduce():
    """bench.py --gpus 2 with the C-ABI's own transport (RcclTransport -> cpm_comm_create(rank, 2), cpm_allreduce_grid_sparse with the
    gather's marks) -- over the RCCL test double (tests/fake_rccl: librccl refuses two ranks on the box's one GPU), ranks started by
    gloo.  The communicator reports two ranks, every frame's reduce went through the sparse path or its dense fall-back."""
    sys.path.insert(0, str(REPO / "tests" / "fake_rccl"))
    import build as fake_build
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CPM_RCCL_LIBRARY=str(fake_build.build()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29536", str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--workload", "config1",
           "--test-backend", "gloo", "--test-one-device", "--transport", "rccl", "--exchange", "union"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["transport"] == "RcclTransport" and d["config"]["rccl_ranks"] == 2
    red = d["reduce"]
    assert red["stream_synchronisations_per_frame"] == 0 and "cpm_allreduce_grid_sparse" in red["kind"]
    assert red["frames_sparse"] + red["frames_dense_by_policy"] + red["frames_dense_after_overflow"] == 6
    assert red["n_union_median"] > 0 and red["reduce_bytes_per_frame"] > 0


def test_bench_two_ranks_brick_lists_through_the_c_abi():
    """A fixed photon count sharded over two ranks (--scaling strong): contiguous ranges + cpm_reduce_grid_bricklists by default -- every
    rank's own bricks as one segment to rank 0, over the RCCL test double on the box's one GPU."""
    sys.path.insert(0, str(REPO / "tests" / "fake_rccl"))
    import build as fake_build
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CPM_RCCL_LIBRARY=str(fake_build.build()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--workload", "config1",
           "--scaling", "strong", "--test-backend", "gloo", "--test-one-device", "--transport", "rccl"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["transport"] == "RcclTransport" and d["config"]["rccl_ranks"] == 2
    assert d["config"]["exchange"] == "lists" and "contiguous" in d["config"]["shards"]
    red = d["reduce"]
    assert "cpm_reduce_grid_bricklists" in red["kind"] and red["stream_synchronisations_per_frame"] == 0
    assert red["sent_bytes_per_rank_per_frame"][0] == 0 and red["sent_bytes_per_rank_per_frame"][1] > 0    # the root sends nothing
    assert red["received_bytes_at_root_per_frame"] == sum(red["sent_bytes_per_rank_per_frame"])
    assert red["bricks"]["union"] <= sum(red["bricks"]["lit_per_rank"])
    # round 6: the exchange that was timed is checked against an independent dense sum; the model's constants were measured over the communicator;
    # every sender's gather form was chosen by measurement on that rank
    assert red["self_check"]["ok"] and red["self_check"]["nonzero_voxels"] > 0 and red["self_check"]["relative"] < 2e-5
    assert red["model"]["measured"]["latency_us"] > 0 and red["model"]["constants"]["source"].startswith("measured")
    sg = d["config"]["sender_gather"]["per_rank"]
    assert sg[0]["chosen"].startswith("root") and sg[1]["chosen"] in ("segment", "pack") and sg[1]["segment_us"] > 0 and sg[1]["pack_us"] > 0


def test_bench_two_ranks_brick_lists_over_gloo():
    """The same exchange carried out by torch ops (TorchTransport's twin of the protocol) when the C-ABI's transport is not in use."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29538", str(REPO / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3", "--workload", "config1",
           "--scaling", "strong", "--test-backend", "gloo", "--test-one-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = _last_json(r.stdout)
    assert d["config"]["exchange"] == "lists" and d["config"]["transport"] == "TorchTransport"
    assert "cpm_reduce_grid_bricklists" in d["reduce"]["kind"] and d["reduce"]["received_bytes_at_root_per_frame"] > 0
    assert d["reduce"]["self_check"]["ok"] and d["reduce"]["model"]["constants"]["source"] == "assumed"


def test_bench_four_ranks_weak_scaling_chooses_its_exchange():
    """`bench.py --gpus 4` as the driver runs it for the scaling curve (config 2's weak scaling rule on the small workload): tile shards, the
    exchange picked by the byte model from a probe frame's brick counts (the same decision on every rank), carried out by the C-ABI over
    the RCCL test double with FOUR ranks on the box's one GPU (4 ranks + this process: within the box's limit of 6)."""
    sys.path.insert(0, str(REPO / "tests" / "fake_rccl"))
    import build as fake_build
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CPM_RCCL_LIBRARY=str(fake_build.build()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29539", str(REPO / "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "3", "--workload", "config1",
           "--test-backend", "gloo", "--test-one-device", "--transport", "rccl"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 4 and d["value"] > 0 and d["scaling"] == "weak" and "tiles" in d["config"]["shards"]
    assert d["config"]["transport"] == "RcclTransport" and d["config"]["rccl_ranks"] == 4
    assert d["config"]["photons_per_frame"] == 4 * 65536 and d["config"]["photons_rank0"] == 65536
    chosen = d["config"]["exchange_chosen_by"]
    assert chosen["chosen"] == d["config"]["exchange"] and d["config"]["exchange"] in ("lists", "union")
    red = d["reduce"]
    assert red["stream_synchronisations_per_frame"] == 0 and len(red["bricks"]["lit_per_rank"]) == 4
    if d["config"]["exchange"] == "lists":
        assert len(red["sent_bytes_per_rank_per_frame"]) == 4 and red["sent_bytes_per_rank_per_frame"][0] == 0
        assert all(b > 0 for b in red["sent_bytes_per_rank_per_frame"][1:])
        assert red["received_bytes_at_root_per_frame"] == sum(red["sent_bytes_per_rank_per_frame"])


def test_bench_two_frames_in_flight():
    """--frames-in-flight 2: every other frame on a second stream with its own context and buffers -- the lever for shards whose frames are
    launch-latency chains.  One GPU: a line, labelled; two ranks over the RCCL double with the brick-list exchange: the summed volume still
    passes its self-check (each of the reducer's two buffers, and each ticket, stays with one of the two streams)."""
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "10", "--warmup", "2", "--workload", "config1", "--no-extras", "--no-cpu-baseline",
                        "--frames-in-flight", "2"], capture_output=True, text=True, timeout=900, cwd=str(REPO))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["config"]["frames_in_flight"] == 2 and d["value"] > 0
    sys.path.insert(0, str(REPO / "tests" / "fake_rccl"))
    import build as fake_build
    # (CPM_BENCH_RCCL_PROBE=1: every rank first runs its end of the communicator set-up in a probe process -- what real multi-GPU runs do)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CPM_RCCL_LIBRARY=str(fake_build.build()), CPM_BENCH_RCCL_PROBE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--workload", "config1",
           "--scaling", "strong", "--test-backend", "gloo", "--test-one-device", "--transport", "rccl", "--frames-in-flight", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = _last_json(r.stdout)
    assert d["config"]["frames_in_flight"] == 2 and d["config"]["exchange"] == "lists" and d["reduce"]["self_check"]["ok"]
    assert d["config"]["transport"] == "RcclTransport" and d["config"]["transport_probes"]["cabi"] == "passed"


def test_bench_survives_a_communicator_set_up_that_never_returns():
    """the first contact with N GPUs must end with a line.  `python bench.py --gpus 2` -- the driver's form -- over an RCCL whose
    ncclCommInitRank blocks for ever (the test double, FAKE_RCCL_BLOCK_INIT): the ranks hang inside the C-ABI's communicator set-up, the parent
    (which never touched a GPU) kills their process group when the set-up budget runs out and starts a FRESH set with --transport torch
    --exchange union; ONE JSON line comes out and says what happened."""
    sys.path.insert(0, str(REPO / "tests" / "fake_rccl"))
    import build as fake_build
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(CPM_RCCL_LIBRARY=str(fake_build.build()), FAKE_RCCL_BLOCK_INIT="1", CPM_BENCH_LAUNCH_BUDGET_S="20", CPM_BENCH_LAUNCH_TOTAL_S="240")
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "config1",
           "--test-backend", "gloo", "--test-one-device", "--transport", "rccl"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["transport"] == "TorchTransport" and d["config"]["rccl_ranks"] == 0
    note = d["config"]["launcher_note"]
    assert "20 s after they were up" in note and "were killed" in note and "--transport torch --exchange union" in note
    assert "starting a fresh set of ranks" in r.stderr


def test_ranks_started_by_torchrun_leave_a_set_up_that_never_returns():
    """The driver starts N > 1 ranks with torch.distributed.run itself: nothing of bench.py's stands above them, so a communicator set-up
    that hangs would hang until the driver's limit.  Every rank therefore runs its end of the set-up in a PROBE PROCESS first (bench.py
    --rccl-probe) and kills it when its budget runs out; the ranks agree and carry on over torch.distributed -- same processes, one JSON
    line, which says what happened.  Here: the RCCL double whose ncclCommInitRank blocks for ever, two ranks on one GPU."""
    sys.path.insert(0, str(REPO / "tests" / "fake_rccl"))
    import build as fake_build
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CPM_RCCL_LIBRARY=str(fake_build.build()), FAKE_RCCL_BLOCK_INIT="1",
               CPM_BENCH_RCCL_PROBE="1", CPM_BENCH_PROBE_BUDGET_S="15")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "config1",
           "--test-backend", "gloo", "--test-one-device", "--transport", "rccl", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["transport"] == "TorchTransport" and d["config"]["rccl_ranks"] == 0
    probes = d["config"]["transport_probes"]
    assert "probe process did not finish in 15 s and was killed" in probes["cabi"] and 15 <= probes["cabi_s"] < 60
    assert "probe process did not finish in 15 s" in json.dumps(d["config"])      # ... and in the line's own description of its reduce


def test_probe_processes_over_the_real_rccl_with_one_rank(ctx, tmp_path):
    """Both kinds of probe process against the REAL RCCL (one rank is all one GPU allows): the C-ABI's set-up from an id made in this
    process + its all-reduce, and torch.distributed's own NCCL group from a file rendezvous."""
    sys.path.insert(0, str(REPO))
    import importlib
    bench = importlib.import_module("bench")
    env_was = os.environ.pop("CPM_RCCL_LIBRARY", None)
    try:
        ok, why = bench.probe_in_child("cabi", ctx.comm_unique_id().hex(), 0, 1, 0, 300.0)
        assert ok, why
        ok, why = bench.probe_in_child("torch", str(tmp_path / "rendezvous"), 0, 1, 0, 300.0)
        assert ok, why
        ok, why = bench.probe_in_child("nonsense", "-", 0, 1, 0, 300.0)
        assert not ok and "unknown kind" in why
    finally:
        if env_was is not None:
            os.environ["CPM_RCCL_LIBRARY"] = env_was


def test_the_ladder_ends_on_gloo_when_the_real_rccl_refuses_both_paths():
    """The ladder's lower rungs with REAL errors: two ranks on one GPU and the real RCCL (CPM_BENCH_TEST_SHARED_NCCL=1), which refuses a
    communicator whose ranks share a device.  The C-ABI's probe process fails, torch.distributed's own NCCL group fails its probe too, the
    ranks -- the same processes, nothing set up in them yet -- carry the sums over gloo and the line says all of that."""
    env = {k: v for k, v in os.environ.items() if k != "CPM_RCCL_LIBRARY"}
    env.update(MASTER_ADDR="127.0.0.1", CPM_BENCH_RCCL_PROBE="1", CPM_BENCH_TEST_SHARED_NCCL="1", CPM_BENCH_PROBE_BUDGET_S="40")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29549", str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "config1",
           "--test-one-device", "--transport", "rccl", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    probes = d["config"]["transport_probes"]
    assert probes["cabi"] != "passed" and probes["torch"] != "passed", probes
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["transport"] == "TorchTransport" and d["config"]["rccl_ranks"] == 0
    assert "gloo: the sums are staged through the host" in json.dumps(d["config"])
    print(probes)
