#!/bin/bash
# Diagnostic builds of the library (timing experiments; outputs of -DSNR_EXP_* builds are garbage by design).
# usage: tools/build_diag.sh NAME [-DFLAG ...]   ->  tools/_diag/libsupnerf_stamps_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/_diag/libsupnerf_stamps_${name}.so
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -shared -DSNR_STAMPS "$@" \
    -Isup-nerf_amd/csrc sup-nerf_amd/csrc/snr_aux.hip sup-nerf_amd/csrc/snr_mlp.hip sup-nerf_amd/csrc/snr_mlp16.hip sup-nerf_amd/csrc/snr_mlp16_bwd.hip sup-nerf_amd/csrc/snr_mlp_bwd.hip sup-nerf_amd/csrc/snr_bf16.hip -o $out
echo $out
