#!/bin/bash
# A/B of builds on the same box (copy candidate libraries to ab/lib<tag>.so first): alternates them twice
L=fluidsimulation_amd/csrc/libpbf_hip.so
for v in "$@" "$@"; do
  cp ab/lib$v.so $L
  timeout -k 10 200 python bench.py --cpu-steps 0 ${AB_ARGS:-} > gpurun_out/ab_$v.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); e=d.get('evolved') or {}
print('$v', round(d['ms_per_step'],4), {k:round(v['ms_per_launch']*1e3,1) for k,v in d['phases'].items()}, 'evolved', round(e.get('ms_per_step',0),4), {k:round(v['ms_per_launch']*1e3,1) for k,v in (e.get('phases') or {}).items()}, 'health', d['health']['mean_neighbors'], d['health']['nonfinite'])"
done
