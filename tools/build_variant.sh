#!/bin/bash
# Build a variant of the library with extra -D flags into tools/_diag/libvariant_NAME.so (for A/B timing with tools/ab_time.py)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -shared "$@" \
    sup-nerf_amd/csrc/snr_aux.hip sup-nerf_amd/csrc/snr_wgrad.hip sup-nerf_amd/csrc/snr_mlp.hip sup-nerf_amd/csrc/snr_mlp_bwd.hip sup-nerf_amd/csrc/snr_bf16.hip -o tools/_diag/libvariant_${name}.so
echo tools/_diag/libvariant_${name}.so
