set -e
cd fluidsimulation_amd/host
R=../../gpurun_out/r02_slab_native_win16.txt
: > $R
timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 8 --steps 20 --ghost-layers 2 >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 8 --steps 20 --ghost-layers 4 >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 256 250 250 --slabs 1 --steps 20 --ghost-layers 4 >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 400 100 100 --slabs 4 --steps 60 --ghost-layers 4 --check >> $R 2>&1
timeout -k 10 200 ./pbf_slab_run --dims 400 100 100 --slabs 4 --steps 60 --ghost-layers 2 >> $R 2>&1
timeout -k 10 400 ./pbf_slab_run --dims 400 100 100 --slabs 4 --steps 6000 --ghost-layers 4 --rebalance 50 --check >> $R 2>&1
cat $R
