// Backward of the fused decoder / render kernels (placeholder until the kernel lands).
#include "snr_device.hpp"
#include "snr_host.hpp"
extern "C" {
size_t snr_decoder_bwd_ws_bytes(int64_t, int, int) { return 16; }
int snr_decoder_bwd(const float*, const float*, const float*, const float*, const void*, const float*, const float*, const float*,
                    int64_t, int64_t, int, int, float*, float*, float*, void*, size_t, void*) { return SNR_E_UNSUPPORTED; }
size_t snr_render_bwd_ws_bytes(const snr_render_args*) { return 16; }
int snr_render_bwd(const snr_render_args*, const float*, const float*, const void*, const float*, const float*, const float*,
                   float*, float*, float*, float*, void*, size_t, void*) { return SNR_E_UNSUPPORTED; }
}
