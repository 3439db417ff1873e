#!/bin/bash
# builds and runs the standalone reproducer on the GPU box:  gpurun -- bash tools/mfma_neighbour/run_repro.sh [trials]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -ffp-contract=off -c -o /tmp/repro_pk.o repro_pk.hip
$HIPCC --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -c -o /tmp/repro_scalar.o repro_scalar.hip 2>/dev/null
$HIPCC --offload-arch=gfx950 -O3 -c -o /tmp/repro_main.o repro.hip
$HIPCC --offload-arch=gfx950 -o /tmp/repro /tmp/repro_main.o /tmp/repro_pk.o /tmp/repro_scalar.o
timeout -k 10 120 /tmp/repro ${1:-40}
