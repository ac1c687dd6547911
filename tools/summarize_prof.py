#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (kernel stats and --pmc passes) into a small text/JSON
report for profiles/.  Usage: summarize_prof.py <pmc_dir> [<kernel_stats.csv>] > report.md"""
import csv
import glob
import json
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def main():
    pmc_dir = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(f"{pmc_dir}/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            c = r["Counter_Name"]
            agg[k][c] += float(r["Counter_Value"])
            calls[k][c] += 1
    out = {}
    for k in agg:
        out[k] = {c: agg[k][c] / max(1, calls[k][c]) for c in agg[k]}   # per-launch averages
    n = None
    for ref in ("k_predict", "k_finalize"):                   # one thread per particle, wave64
        if ref in out and "SQ_WAVES" in out[ref]:
            n = int(round(out[ref]["SQ_WAVES"] * 64))
            break
    from bench import kernel_source_sha     # the fingerprint bench.py checks before it quotes these counters
    try:
        head = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True,
                              text=True).stdout.strip() or None
    except OSError:
        head = None
    print(json.dumps({"particles_per_launch": n, "units": "per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB",
                      "kernel_source_sha": kernel_source_sha(), "git_head_when_summarised": head,
                      "kernels": out}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
