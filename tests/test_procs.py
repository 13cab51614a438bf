"""CPU: the rank runner the two-ranks-on-one-GPU tests use (tests/_procs.py) fails fast instead of waiting out a dead peer."""
import os
import subprocess
import sys
import time

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _procs import run_ranks  # noqa: E402


def test_all_ranks_succeed_and_outputs_come_back():
    outs = run_ranks(lambda r, port: [sys.executable, "-c", f"print('rank', {r}, 'port', {port})"], 2, timeout=60)
    assert [o.split()[:2] for o in outs] == [["rank", "0"], ["rank", "1"]]


def test_a_dead_rank_ends_the_run_at_once():
    t0 = time.monotonic()
    with pytest.raises(AssertionError, match="rank 0 exited with code 3"):
        run_ranks(lambda r, port: [sys.executable, "-c", "import sys, time; print('boom'); sys.exit(3)" if r == 0 else "import time; time.sleep(120)"],
                  2, timeout=100)
    assert time.monotonic() - t0 < 30


def test_port_race_is_retried_once(tmp_path):
    marker = tmp_path / "first_attempt"
    code = ("import os, sys\n"
            f"m = {str(marker)!r}\n"
            "if not os.path.exists(m):\n"
            "    open(m, 'w').close(); print('RuntimeError: Address already in use'); sys.exit(1)\n"
            "print('ok')\n")
    outs = run_ranks(lambda r, port: [sys.executable, "-c", code], 1, timeout=60)
    assert outs[0].strip() == "ok"


def test_timeout_is_reported():
    with pytest.raises(AssertionError, match="no result after"):
        run_ranks(lambda r, port: [sys.executable, "-c", "import time; time.sleep(60)"], 1, timeout=1)


def test_bench_deadline_kills_a_run_that_hangs():
    """`--deadline`: a rank that is still running after the limit exits with code 124 by itself (no GPU needed: the timer is
    armed before any device work)."""
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench; "
            "bench.arm_deadline(0.5, 0); time.sleep(30)" % root)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 124 and time.monotonic() - t0 < 60
    assert b"giving up" in r.stderr
