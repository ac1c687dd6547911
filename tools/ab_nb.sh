L=fluidsimulation_amd/csrc/libpbf_hip.so
for v in "$@" "$@"; do cp ab/lib$v.so $L; echo "== $v"; timeout -k 10 120 python -u tools/nb_probe.py 2>/dev/null || exit 1; done
