#!/bin/bash
# builds gpurun_out-independent experiment variants: t41_sdr_amd/abl/libt41rx_ablN.so
set -e
cd "$(dirname "$0")/../t41_sdr_amd/csrc"
mkdir -p ../abl
for N in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DT41RX_ABLATE=$N -c rx_kernels.hip -o /tmp/rxk_abl$N.o
  hipcc -shared -fPIC --offload-arch=gfx950 /tmp/rxk_abl$N.o rx_host.o design.o -o ../abl/libt41rx_abl$N.so
done
