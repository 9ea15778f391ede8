#!/bin/bash
# Build a variant of the library with extra -D flags into tools/_diag/libvariant_NAME.so (for A/B timing with tools/ab_time.py)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
# SNR_EXTRA_SRC: more sources.  -DSNR_FWD32: the fp32 forward on the one-wave-per-SIMD 32x32x2 kernel of rounds 1-3; -DSNR16_FORCE_WAVES=8: the 16x16x4 kernel as one 512-thread workgroup per CU; -DSNR_BWD32: the fp32 backward on the one-wave 32x32x2 kernel (the split backward has no 32x32x16 form any more)
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -shared -Isup-nerf_amd/csrc "$@" $SNR_EXTRA_SRC \
    sup-nerf_amd/csrc/snr_aux.hip sup-nerf_amd/csrc/snr_wgrad.hip sup-nerf_amd/csrc/snr_mlp.hip sup-nerf_amd/csrc/snr_mlp16.hip sup-nerf_amd/csrc/snr_mlp16_bwd.hip sup-nerf_amd/csrc/snr_mlp_bwd.hip sup-nerf_amd/csrc/snr_bf16.hip -o tools/_diag/libvariant_${name}.so
echo tools/_diag/libvariant_${name}.so
