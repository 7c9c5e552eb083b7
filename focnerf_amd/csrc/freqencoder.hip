// freqencoder.hip — sinusoidal positional encoding (semantics: freqencoder/src/freqencoder.cu:30-94).
//   out[b] = [x_0..x_{D-1}, sin(2^0 x)..(D), sin(2^0 x + pi/2)..(D), sin(2^1 x).., ...]   (the cosine IS sin(v + pi/2), :48-57)
//   dx_d   = g_d + sum_f 2^f (g_sin out_cos - g_cos out_sin)                               (from the saved outputs, :63-94)
// The reference uses the __sinf fast intrinsic (freqencoder/setup.py:10 builds with -use_fast_math); here sinf() is the ocml
// implementation — at least as accurate, so parity holds within 1e-6 of libm.
//
// Layout of the work (not the reference's one-thread-per-output-element with a 64-bit division each): a workgroup owns a TILE of 256
// consecutive rows. Forward: the tile's inputs (256 x D floats) are staged in LDS with coalesced loads; the tile's 256 x C outputs are
// one contiguous range of memory which the threads walk with stride 256 — every wave stores 256 contiguous bytes — and (row, column)
// advance incrementally (no division in the loop, one 32-bit division per thread and tile). Backward: one thread per (row, dimension)
// of the tile, 32-bit index arithmetic; the 2 deg pairs it reads from grad / outputs lie 2 D floats apart in rows that neighbouring
// lanes share. Tiles are dealt grid-stride, so any B fits any grid.
#include "common.h"

#define FQ_ROWS 256u

// R = rows per tile: 256, fewer for wide inputs so that the staged tile stays within 32 KiB of LDS (fq_tile_rows)
__global__ void __launch_bounds__(256) k_freq_fwd(const float *__restrict__ inputs, uint32_t B, uint32_t D, uint32_t C,
                                                  float *__restrict__ outputs, uint32_t R) {
    extern __shared__ float s_in[];                        // [R][D]
    const float HALF_PI = 3.141592653589793f / 2;
    const uint32_t n_tiles = (B + R - 1) / R;
    const uint32_t step_r = 256u / C, step_c = 256u % C;   // what a stride of 256 elements does to (row, column)
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * R, rows = min(R, B - row0);
        const float *in = inputs + (uint64_t)row0 * D;
        __syncthreads();                                   // the previous tile's reads of s_in are done
        for (uint32_t i = threadIdx.x; i < rows * D; i += 256) s_in[i] = in[i];
        __syncthreads();
        float *out = outputs + (uint64_t)row0 * C;
        const uint32_t n_out = rows * C;
        uint32_t r = threadIdx.x / C, c = threadIdx.x - r * C;
        for (uint32_t e = threadIdx.x; e < n_out; e += 256) {
            float o;
            if (c < D) o = s_in[r * D + c];
            else {
                const uint32_t col = c / D - 1, d = c - (col + 1) * D, freq = col >> 1;
                o = sinf(scalbnf(s_in[r * D + d], (int)freq) + (float)(col & 1u) * HALF_PI);
            }
            out[e] = o;
            r += step_r; c += step_c;
            if (c >= C) { c -= C; r++; }
        }
    }
}

__global__ void __launch_bounds__(256) k_freq_bwd(const float *__restrict__ grad, const float *__restrict__ outputs, uint32_t B, uint32_t D,
                                                  uint32_t deg, uint32_t C, float *__restrict__ grad_inputs) {
    const uint32_t n_tiles = (B + FQ_ROWS - 1) / FQ_ROWS;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * FQ_ROWS, rows = min(FQ_ROWS, B - row0);
        float *gi = grad_inputs + (uint64_t)row0 * D;
        for (uint32_t e = threadIdx.x; e < rows * D; e += 256) {
            const uint32_t r = e / D, d = e - r * D;
            const float *g = grad + ((uint64_t)row0 + r) * C + d, *o = outputs + ((uint64_t)row0 + r) * C + d;
            float result = g[0];
            g += D; o += D;
            for (uint32_t f = 0; f < deg; f++) {
                result = fmaf(scalbnf(1.0f, (int)f), fmaf(g[0], o[D], -(g[D] * o[0])), result);
                g += 2 * D; o += 2 * D;
            }
            gi[e] = result;
        }
    }
}

// rows of a forward tile: 256 while 256 x D floats fit 32 KiB (D <= 32: every NeRF encoder), else as many as do (the reference takes
// any input dimension, freqencoder.cu:30-94; one row of up to 8192 floats still fits)
static uint32_t fq_tile_rows(uint32_t D) { return D <= 32u ? FQ_ROWS : (8192u / D > 0u ? 8192u / D : 1u); }

extern "C" {

int foc_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *outputs, void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(inputs && outputs, FOC_E_INVALID, "freq_encode_forward: null pointer");
    FOC_REQUIRE(D >= 1 && C == D + 2 * D * deg, FOC_E_INVALID, "freq_encode_forward: C must equal D + 2*D*deg (D=%u deg=%u C=%u)", D, deg, C);
    FOC_REQUIRE(D <= 8192, FOC_E_INVALID, "freq_encode_forward: input dimension %u is beyond what one tile row holds (<= 8192)", D);
    const uint32_t R = fq_tile_rows(D);
    hipLaunchKernelGGL(k_freq_fwd, dim3(foc_grid_1d((uint64_t)foc_div_up(B, R) * 256, 256)), dim3(256), (size_t)R * D * sizeof(float), (hipStream_t)stream, inputs,
                       B, D, C, outputs, R);
    FOC_CHECK_LAUNCH("freq_encode_forward");
    return FOC_OK;
}

int foc_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                             float *grad_inputs, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(grad && outputs && grad_inputs, FOC_E_INVALID, "freq_encode_backward: null pointer");
    FOC_REQUIRE(D >= 1 && C == D + 2 * D * deg, FOC_E_INVALID, "freq_encode_backward: C must equal D + 2*D*deg (D=%u deg=%u C=%u)", D, deg, C);
    hipLaunchKernelGGL(k_freq_bwd, dim3(foc_grid_1d((uint64_t)foc_div_up(B, FQ_ROWS) * 256, 256)), dim3(256), 0, (hipStream_t)stream, grad, outputs, B, D, deg, C,
                       grad_inputs);
    FOC_CHECK_LAUNCH("freq_encode_backward");
    return FOC_OK;
}

} // extern "C"
