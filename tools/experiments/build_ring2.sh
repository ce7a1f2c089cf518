#!/bin/bash
# Build the two-role ring experiment (tools/experiments/libring2_exp.so); extra flags, e.g. -DSMRF_RING2_NP=3, are passed on.
set -e
cd "$(dirname "$0")/../.."
hipcc -std=c++20 -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -Iinclude -Ineilpy_amd/csrc -Itools/experiments "$@" \
      tools/experiments/ring2_exp.hip -o tools/experiments/libring2_exp${RING2_TAG}.so
