#!/usr/bin/env python3
"""Builds ab/libfp32.so for tools/tolerance_probe.py: the library with the pair evaluation in fp32
(pair geometry, W, grad W, s_corr in fp32, hardware square root) -- a MEASUREMENT variant, not a
product path.  The product header csrc/pbf_device.h holds one arithmetic only; this script patches a
private copy of it (three function bodies) into a scratch include directory that precedes csrc/ on
the include path, and compiles the unchanged translation units against it.
    python tools/build_fp32_probe.py && python tools/tolerance_probe.py ab/libfp32.so
"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

PATCHES = [
    # poly6_pair: W and the gradient scalar from fp32 q = max(h*h - d*d, 0)
    ("    const double d = (double)distance;\n    const double q = fmax(fma(-d, d, k.h2), 0.0);",
     "    { const float qf = fmaxf(fmaf(-distance, distance, k.hh), 0.f); const float q2f = qf * qf;\n"
     "      w = (float)k.w_coef * (q2f * qf); gs = k.g_coef * q2f; return; }\n"
     "    const double d = (double)distance;\n    const double q = fmax(fma(-d, d, k.h2), 0.0);"),
    # sqrt_rn: the 1-ulp hardware square root
    ("    if (x >= 1.2621774e-29f && x <= 3.4028235e38f) {",
     "    return __builtin_amdgcn_sqrtf(x);\n    if (x >= 1.2621774e-29f && x <= 3.4028235e38f) {"),
    # scorr_strict: x^4 and the products in fp32
    ("    float xf = (float)((double)w * k.inv_w_dq);",
     "    { const float xs = w * (float)k.inv_w_dq, x2 = xs * xs;\n"
     "      return ((float)k.scorr_k * (x2 * x2)) * (float)k.dt_pow; }\n"
     "    float xf = (float)((double)w * k.inv_w_dq);"),
]


def main():
    out_dir = ROOT / "ab"
    inc = out_dir / "fp32_probe_include"
    inc.mkdir(parents=True, exist_ok=True)
    text = (g.CSRC / "pbf_device.h").read_text()
    for old, new in PATCHES:
        if text.count(old) != 1:
            raise SystemExit(f"pbf_device.h no longer holds exactly one {old.splitlines()[0]!r}")
        text = text.replace(old, new)
    (inc / "pbf_device.h").write_text(text)
    flags = [f for f in g.HIP_FLAGS if f != f"-I{g.CSRC}"]
    out = out_dir / "libfp32.so"
    # the patched header must win over the one next to pbf_hip.hip: quote-includes search the
    # including file's directory first, so compile copies of the two .hip files from the scratch dir
    for name in ("pbf_hip.hip", "pbf_comm.hip"):
        (inc / name).write_text((g.CSRC / name).read_text())
    cmd = [g.HIPCC, *flags, f"-I{inc}", f"-I{g.CSRC}", str(inc / "pbf_hip.hip"), str(inc / "pbf_comm.hip"),
           "-o", str(out)]
    subprocess.run(cmd, check=True)
    print(out)


if __name__ == "__main__":
    main()
