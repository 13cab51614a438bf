"""Sweep the scan's segment schedule (EVI_SCAN_FIRST x EVI_SCAN_GROWTH) at several shard sizes.

Each point is a fresh `bench.py` child process (the knobs are read once per process).
usage: python tools/segment_sweep.py [rows ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [1 << 20, 1 << 23]
    for rows in rows_list:
        for first in (8192, 16384, 32768, 65536):
            for growth in (8, 16, 32, 64):
                env = dict(os.environ, EVI_SCAN_FIRST=str(first), EVI_SCAN_GROWTH=str(growth))
                out = subprocess.run(
                    [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", str(rows), "--no-graph-eval",
                     "--no-cpu-baseline", "--steps", "40", "--warmup", "5"],
                    env=env, capture_output=True, text=True, timeout=300)
                line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
                try:
                    d = json.loads(line)
                except Exception:
                    print(rows, first, growth, "FAILED", out.stderr[-300:], flush=True)
                    continue
                r = d["roofline"]
                print(f"rows={rows} first={first} growth={growth} step={d['ms_per_step']:.4f} ms "
                      f"scan={r['kernel_ms_per_step']:.4f} select={r['select_ms_per_step']:.4f} "
                      f"launches={r['launches_per_step']}", flush=True)


if __name__ == "__main__":
    main()
