#!/bin/bash
# A/B of builds on the same box (copy candidate libraries to ab/lib<tag>.so first): alternates them twice
L=fluidsimulation_amd/csrc/libpbf_hip.so
for v in "$@" "$@"; do
  cp ab/lib$v.so $L
  timeout -k 10 200 python bench.py --cpu-steps 0 > gpurun_out/ab_$v.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', round(d['ms_per_step'],4), {k:round(v['ms_per_launch']*1e3,1) for k,v in d['phases'].items()})"
done
