// freqencoder.hip — sinusoidal positional encoding (freqencoder/src/freqencoder.cu:30-94).
// One thread per output element; a wave writes 256 contiguous bytes. The reference uses the
// __sinf fast intrinsic (freqencoder/setup.py:10 builds with -use_fast_math); here sinf()
// is the ocml implementation — at least as accurate, so parity holds within 1e-6 of libm.
#include "common.h"

__global__ void __launch_bounds__(256) k_freq_fwd(const float *__restrict__ inputs, uint32_t B, uint32_t D, uint32_t C,
                                                  float *__restrict__ outputs) {
    const float HALF_PI = 3.141592653589793f / 2;
    const uint64_t total = (uint64_t)B * C;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
        const float *in = inputs + (uint64_t)b * D;
        float o;
        if (c < D) o = in[c];
        else {
            const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
            const float phase_shift = (float)(col % 2) * HALF_PI;
            o = sinf(scalbnf(in[d], (int)freq) + phase_shift);
        }
        outputs[t] = o;
    }
}

__global__ void __launch_bounds__(256) k_freq_bwd(const float *__restrict__ grad, const float *__restrict__ outputs, uint32_t B, uint32_t D,
                                                  uint32_t deg, uint32_t C, float *__restrict__ grad_inputs) {
    const uint64_t total = (uint64_t)B * D;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (uint64_t)b * D);
        const float *g = grad + (uint64_t)b * C, *o = outputs + (uint64_t)b * C;
        float result = g[d];
        g += D; o += D;
        for (uint32_t f = 0; f < deg; f++) {
            result = fmaf(scalbnf(1.0f, (int)f), fmaf(g[d], o[D + d], -(g[D + d] * o[d])), result);
            g += 2 * D; o += 2 * D;
        }
        grad_inputs[t] = result;
    }
}

extern "C" {

int foc_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *outputs, void *stream) {
    FocDeviceGuard foc_guard_(stream);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(inputs && outputs, FOC_E_INVALID, "freq_encode_forward: null pointer");
    FOC_REQUIRE(D >= 1 && C == D + 2 * D * deg, FOC_E_INVALID, "freq_encode_forward: C must equal D + 2*D*deg (D=%u deg=%u C=%u)", D, deg, C);
    hipLaunchKernelGGL(k_freq_fwd, dim3(foc_grid_1d((uint64_t)B * C, 256)), dim3(256), 0, (hipStream_t)stream, inputs, B, D, C, outputs);
    FOC_CHECK_LAUNCH("freq_encode_forward");
    return FOC_OK;
}

int foc_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                             float *grad_inputs, void *stream) {
    FocDeviceGuard foc_guard_(stream);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(grad && outputs && grad_inputs, FOC_E_INVALID, "freq_encode_backward: null pointer");
    FOC_REQUIRE(D >= 1 && C == D + 2 * D * deg, FOC_E_INVALID, "freq_encode_backward: C must equal D + 2*D*deg (D=%u deg=%u C=%u)", D, deg, C);
    hipLaunchKernelGGL(k_freq_bwd, dim3(foc_grid_1d((uint64_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream, grad, outputs, B, D, deg, C, grad_inputs);
    FOC_CHECK_LAUNCH("freq_encode_backward");
    return FOC_OK;
}

} // extern "C"
