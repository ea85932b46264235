"""bench.py's launcher on a host without a GPU: `python bench.py --gpus N` (no WORLD_SIZE -- the driver's form) must start
N ranks as child processes.  Without a GPU every rank stops at "needs a GPU" (there is no CPU fallback), which is what is
counted here; the GPU form of the test is tests/test_bench_gpu.py::test_bench_plain_command_starts_its_own_ranks."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def test_plain_command_starts_n_ranks():
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by tests/test_bench_gpu.py on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "config1"],
                       capture_output=True, text=True, timeout=600, cwd=str(REPO), env=env)
    assert r.returncode != 0                                  # the children's failure is the parent's
    # one per rank and attempt: ranks that die are replaced ONCE by a fresh set (--transport torch --exchange union), then the parent gives up
    assert (r.stdout + r.stderr).count("bench.py needs a GPU") == 4
    assert "starting a fresh set of ranks" in r.stderr and "no JSON line from either set of ranks" in r.stderr
    assert r.stdout.strip() == ""                             # nothing but rank 0's JSON line ever goes to stdout


FAKE_CHILD = r"""
import json, sys, time
mode, args = sys.argv[1], sys.argv[2:]
second = "--transport" in args and args[args.index("--transport") + 1] == "torch" and "--exchange" in args
print("fake child", mode, "second" if second else "first", file=sys.stderr, flush=True)
if mode == "hang_then_ok" and not second:
    time.sleep(600)
if mode == "die_then_ok" and not second:
    sys.exit(3)
if mode == "hang_always":
    time.sleep(600)
if mode == "silent" :
    sys.exit(0)
print("some chatter on stdout")
print(json.dumps({"metric": "Mphotons/s traced+binned+gathered", "value": 1.0, "config": {"transport": "TorchTransport" if second else "RcclTransport"}}), flush=True)
"""


def _launch(tmp_path, mode, budget):
    sys.path.insert(0, str(REPO))
    import importlib
    bench = importlib.import_module("bench")
    child = tmp_path / "fake_child.py"
    child.write_text(FAKE_CHILD)
    import io
    import contextlib
    out, err = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        rc = bench.launch_ranks(2, argv=["--gpus", "2", "--steps", "2"], budget_s=budget,
                                make_cmd=lambda args: [sys.executable, str(child), mode] + args)
    return rc, out.getvalue(), err.getvalue()


def test_a_set_up_that_hangs_is_killed_and_replaced_once(tmp_path):
    """`python bench.py --gpus N` always ends with a line.  Ranks that overrun their wall-clock budget (a set-up that never
    returns from RCCL) are killed -- their whole process group -- and a FRESH set runs once with --transport torch --exchange union; its
    line says so."""
    import json
    import time
    t0 = time.time()
    rc, out, err = _launch(tmp_path, "hang_then_ok", 3.0)
    assert rc == 0 and time.time() - t0 < 60
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1                                     # the JSON line alone on stdout (the chatter went to stderr)
    d = json.loads(lines[0])
    assert d["config"]["transport"] == "TorchTransport"
    assert "did not finish (3 s after their start) and were killed" in d["config"]["launcher_note"] and "--transport torch --exchange union" in d["config"]["launcher_note"]
    assert "fake child hang_then_ok first" in err and "fake child hang_then_ok second" in err


def test_the_set_up_clock_starts_when_the_ranks_are_up(tmp_path):
    """Importing torch on a fresh box takes minutes and is nobody's hang: the budget runs from the ranks' "up" line, a separate (longer) clock from
    their start.  A child that is slow to come up and then quick is left alone; one that comes up and then hangs is killed `budget` later."""
    import importlib
    import io
    import contextlib
    import time
    sys.path.insert(0, str(REPO))
    bench = importlib.import_module("bench")
    child = tmp_path / "slow_child.py"
    child.write_text(r"""
import json, sys, time
mode = sys.argv[1]
second = "--transport" in sys.argv
time.sleep(3.0)                                         # "importing torch"
print("bench.py: ranks up (2)", file=sys.stderr, flush=True)
if mode == "up_then_hang" and not second:
    time.sleep(600)
print(json.dumps({"metric": "Mphotons/s traced+binned+gathered", "value": 1.0, "config": {"second": second}}), flush=True)
""")
    for mode, second in (("up_then_quick", False), ("up_then_hang", True)):
        out, err = io.StringIO(), io.StringIO()
        os.environ["CPM_BENCH_LAUNCH_BUDGET_S"], os.environ["CPM_BENCH_LAUNCH_TOTAL_S"] = "2", "60"
        try:
            t0 = time.time()
            with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
                rc = bench.launch_ranks(2, argv=["--gpus", "2"], make_cmd=lambda args: [sys.executable, str(child), mode] + args)
        finally:
            del os.environ["CPM_BENCH_LAUNCH_BUDGET_S"], os.environ["CPM_BENCH_LAUNCH_TOTAL_S"]
        import json
        d = json.loads(out.getvalue().strip())
        assert rc == 0 and d["config"]["second"] is second, (mode, err.getvalue())
        if second:
            assert "2 s after they were up" in d["config"]["launcher_note"] and time.time() - t0 < 40


def test_ranks_that_die_are_replaced_once(tmp_path):
    import json
    rc, out, err = _launch(tmp_path, "die_then_ok", 30.0)
    assert rc == 0
    d = json.loads(out.strip())
    assert "exited with code 3" in d["config"]["launcher_note"]


def test_a_second_failure_ends_the_run_with_the_ranks_last_lines(tmp_path):
    rc, out, err = _launch(tmp_path, "hang_always", 2.0)
    assert rc == 1 and out.strip() == ""
    assert err.count("did not finish (2 s after their start) and were killed") >= 2 and "no JSON line from either set of ranks" in err
    rc, out, err = _launch(tmp_path, "silent", 30.0)
    assert rc == 1 and "printed no JSON line" in err


def test_a_first_set_that_succeeds_is_relayed_untouched(tmp_path):
    import json
    rc, out, err = _launch(tmp_path, "ok", 30.0)
    assert rc == 0
    d = json.loads(out.strip())
    assert d["config"]["transport"] == "RcclTransport" and "launcher_note" not in d["config"]


def test_a_probe_process_reports_how_it_ended(monkeypatch):
    """The ranks' probe processes (bench.py --rccl-probe: a communicator set-up that may hang runs in a child the rank can kill): one that
    fails says so with its last line, one that overruns its budget is killed by PID and said to have been."""
    import time
    import torch
    sys.path.insert(0, str(REPO))
    import importlib
    bench = importlib.import_module("bench")
    if not torch.cuda.is_available():
        ok, why = bench.probe_in_child("cabi", "00" * 128, 0, 2, 0, 120.0)
        assert not ok and "rank 0's probe process exited with code 1" in why and "needs a GPU" in why
    t0 = time.time()
    ok, why = bench.probe_in_child("cabi", "00" * 128, 1, 2, 0, 0.05)     # (still importing when the budget ends)
    assert not ok and why == "rank 1's probe process did not finish in 0 s and was killed" and time.time() - t0 < 30
