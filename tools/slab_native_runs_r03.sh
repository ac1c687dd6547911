# One-GPU rehearsal of the eight-slab step on the round-3 kernels: every autotune candidate of
# bench.py (ghost depth 4 / 2 / 2 with the overlapped halo / 1), then the four-slab parity runs.
set -e
cd fluidsimulation_amd/host
R=../../gpurun_out/r03_slab_native.txt
: > $R
timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 1 --steps 20 --ghost-layers 4 >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 8 --steps 20 --ghost-layers 4 >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 8 --steps 20 --ghost-layers 2 >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 8 --steps 20 --ghost-layers 2 --overlap >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 8 --steps 20 --ghost-layers 1 >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 400 100 100 --slabs 4 --steps 60 --ghost-layers 4 --check >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 400 100 100 --slabs 4 --steps 60 --ghost-layers 2 --overlap --check >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 400 100 100 --slabs 4 --steps 60 --ghost-layers 1 --check >> $R 2>&1
cat $R
