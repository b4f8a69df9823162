#!/bin/bash
# experiment builds of the kernel object: tools/build_variant.sh NAME [-DFLAG ...] -> t41_sdr_amd/abl/libt41rx_NAME.so
# (select with T41RX_LIB=...; the host objects are the product's)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../t41_sdr_amd/csrc"
make -s rx_host.o design.o nr_kernels.o nr_tables.o tx_kernels.o tx_host.o tx_tables.o
mkdir -p ../abl
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize "$@" -c rx_kernels.hip -o /tmp/rxk_$NAME.o
hipcc -shared -fPIC --offload-arch=gfx950 /tmp/rxk_$NAME.o rx_host.o design.o nr_kernels.o nr_tables.o tx_kernels.o tx_host.o tx_tables.o -o ../abl/libt41rx_$NAME.so
