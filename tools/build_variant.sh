#!/bin/bash
# builds ab/lib<tag>.so from the tree's sources with extra compiler flags:  tools/build_variant.sh tag -DNB_PASSES=1
tag=$1; shift
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -Wall -Wno-unused-function -ldl \
  -Iinclude -Ifluidsimulation_amd/csrc "$@" fluidsimulation_amd/csrc/pbf_hip.hip fluidsimulation_amd/csrc/pbf_comm.hip -o ab/lib$tag.so && echo "ab/lib$tag.so"
