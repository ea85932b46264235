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
    assert (r.stdout + r.stderr).count("bench.py needs a GPU") == 2   # one per rank
    assert r.stdout.strip() == ""                             # nothing but rank 0's JSON line ever goes to stdout
