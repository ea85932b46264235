"""The checker under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not
available on the pool): oracle/selftest.c calls every oracle function once on small, exact-size heap buffers."""
import os
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def test_oracle_selftest_under_asan_ubsan():
    r = subprocess.run(["make", "-C", str(REPO / "oracle"), "san"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available here: " + r.stderr[-300:])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([str(REPO / "oracle" / "_san" / "selftest")], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "selftest ok" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
