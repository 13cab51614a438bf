"""Helpers for the tests that start one Python process per rank themselves (two ranks on the one test GPU).

`run_ranks` waits for ALL ranks at once: the moment one exits with an error the others are killed and the test fails with that
rank's output, instead of the survivors sitting in a rendezvous or a collective until a timeout (a rank that cannot bind the
rendezvous port — the port is probed free, then released, then bound by rank 0: another process can take it in between — used
to cost the whole timeout).  A run that failed on exactly that race is repeated once on a fresh port.
"""
import os
import socket
import subprocess
import tempfile
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_once(make_cmd, world, port, env, timeout):
    logs = [tempfile.NamedTemporaryFile(prefix=f"rank{r}_", suffix=".log", delete=False) for r in range(world)]
    procs = [subprocess.Popen(make_cmd(r, port), env=env, stdout=logs[r], stderr=subprocess.STDOUT) for r in range(world)]
    deadline = time.monotonic() + timeout
    failed = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank {bad[0]} exited with code {codes[bad[0]]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                failed = f"no result after {timeout} s (exit codes so far: {codes})"
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    outs = []
    for f in logs:
        f.close()
        with open(f.name, "rb") as fh:
            outs.append(fh.read().decode(errors="replace"))
        os.unlink(f.name)
    return failed, outs


def run_ranks(make_cmd, world, *, env=None, timeout=500):
    """make_cmd(rank, port) -> argv.  Returns the ranks' outputs; raises AssertionError (with the output tails) on failure."""
    for attempt in range(2):
        failed, outs = _run_once(make_cmd, world, free_port(), env, timeout)
        if failed is None:
            return outs
        port_race = any("ddress already in use" in o or "EADDRINUSE" in o for o in outs)
        if not (port_race and attempt == 0):
            break
    tails = "\n".join(f"--- rank {r} ---\n{o[-3000:]}" for r, o in enumerate(outs))
    raise AssertionError(f"{failed}\n{tails}")
