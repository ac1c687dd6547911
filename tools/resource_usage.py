#!/usr/bin/env python3
"""ISA-level resource usage of every kernel in libpbf_hip.so, from the compiler's own remarks
(`hipcc -Rpass-analysis=kernel-resource-usage`, the flags of __graft_entry__.build_hip): VGPRs, SGPRs,
scratch (spills), LDS and the occupancy the registers / LDS allow, one line per kernel.  CPU-only
(hipcc cross-compiles gfx950); writes profiles/<tag>_resource_usage.txt.
    python tools/resource_usage.py r03
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    rows = {}
    for src in ("pbf_hip.hip", "pbf_comm.hip"):
        with tempfile.TemporaryDirectory() as td:
            flags = [f for f in g.HIP_FLAGS if f not in ("-shared", "-ldl")]
            cmd = [g.HIPCC, *flags, "-c", "-Rpass-analysis=kernel-resource-usage", str(g.CSRC / src), "-o",
                   str(Path(td) / "x.o")]
            err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
        cur = None
        for ln in err.splitlines():
            m = re.search(r"remark: (?:Function Name|Kernel Name): (\S+)", ln)
            if m:
                cur = rows.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\]| \[waves/SIMD\])?: (\d+)", ln)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
    names = demangle(list(rows))
    keys = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize", "LDS Size", "Occupancy"]
    lines = [f"# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, flags of __graft_entry__.build_hip "
             f"(kernel sources {__import__('bench').kernel_source_sha()})",
             f"# {'kernel':<78} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch B/lane':>15} {'LDS B/block':>12} {'occ waves/SIMD':>15}"]
    for mangled, r in sorted(rows.items(), key=lambda kv: names[kv[0]]):
        nm = re.sub(r"^void ", "", names[mangled].replace("(anonymous namespace)::", ""))
        nm = nm[:nm.index("(")] if "(" in nm else nm
        if not r:
            continue
        lines.append(f"{nm:<80} " + " ".join(f"{r.get(k, 0):>{w}}" for k, w in zip(keys, (5, 5, 5, 15, 12, 15))))
    out = ROOT / "profiles" / f"{tag}_resource_usage.txt"
    out.write_text("\n".join(lines) + "\n")
    print(out)


if __name__ == "__main__":
    main()
