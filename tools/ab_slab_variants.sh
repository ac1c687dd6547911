#!/bin/bash
# A/B of library builds (ab/lib<tag>.so) on the native slab step, one GPU: tools/ab_slab_variants.sh tagA tagB ...
L=fluidsimulation_amd/csrc/libpbf_hip.so
cp $L ab/libkeep.so
for v in "$@" "$@"; do
  cp ab/lib$v.so $L
  echo "== $v"
  ( cd fluidsimulation_amd/host && timeout -k 10 100 ./pbf_slab_run --dims 100 100 100 --slabs 1 --steps 200 --ghost-layers 2 && timeout -k 10 100 ./pbf_slab_run --dims 200 100 100 --slabs 2 --steps 100 --ghost-layers 2 && timeout -k 10 100 ./pbf_slab_run --dims 400 100 100 --slabs 4 --steps 100 --ghost-layers 2 --check && timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 8 --steps 20 --ghost-layers 2 ) || exit 1
done
cp ab/libkeep.so $L
