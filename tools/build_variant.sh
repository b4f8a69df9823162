#!/bin/bash
# experiment builds of the RX kernel objects: tools/build_variant.sh NAME [-DFLAG ...] -> t41_sdr_amd/abl/libt41rx_NAME.so
# (select with T41RX_LIB=...; the host objects are the product's).  Every variant is built with -DT41RX_EXPERIMENT=1
# (rx_experiments.hpp): a variant whose flags change results or add diagnostics reports itself and t41rx_create() refuses
# it unless T41RX_ALLOW_EXPERIMENT=1 is in the environment -- the tools that time variants set it.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../t41_sdr_amd/csrc"
make -s -j8 rx_host.o design.o nr_kernels.o nr_tables.o tx_kernels.o tx_host.o tx_tables.o
mkdir -p ../abl
B=/tmp/rxk_$NAME; mkdir -p $B
for TU in rx512_ssb rx512_am rx512_nfm rx512_sam rx_long fastconv display_kernel rx_dispatch; do
  hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -fno-slp-vectorize -I../../include -DT41RX_EXPERIMENT=1 "$@" -c $TU.hip -o $B/$TU.o &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o rx_host.o design.o nr_kernels.o nr_tables.o tx_kernels.o tx_host.o tx_tables.o -o ../abl/libt41rx_$NAME.so
