#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (the oracle and the host coefficient designer).
# GPU sanitizers are not available on this pool; the kernels are covered by the parity tests.
set -e
cd "$(dirname "$0")/../.."
OUT=${TMPDIR:-/tmp}/t41_sanitize
mkdir -p $OUT
gcc -O1 -g -std=c11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -Ioracle \
    tools/sanitize/oracle_harness.c oracle/t41_oracle.c -o $OUT/oracle_harness -lm -lpthread
$OUT/oracle_harness
g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude \
    tools/sanitize/designer_harness.cpp t41_sdr_amd/csrc/design.cpp -o $OUT/designer_harness
$OUT/designer_harness
