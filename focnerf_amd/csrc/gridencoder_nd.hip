// gridencoder_nd.hip — the grid encoder for input dimensions 4 and 5 (gridencoder.cu:393-398, 437-442, 633-638 dispatch D in {2,3,4,5}).
//
// The NeRF networks of the path are D = 3 (and D = 2 for the background), which gridencoder.hip serves with kernels specialised on D at
// compile time (shared index terms, row-pair loads, the binned backward). D = 4 / 5 — space-time grids, dnerf-style callers — are kept
// to ONE kernel per operation with D as a RUNTIME value: 16 / 32 corners per (point, level) fully unrolled for every (D, C, dtype)
// multiplied the build time of the library several times over for shapes no FOC network uses. Same arithmetic as the compile-time forms
// (explicit fmaf where nvcc contracts, fp32 accumulation, one rounding to the table dtype), same layouts:
//   forward   thread = (point, level), level-major launch, outputs [L,B,C] or [B,L*C], optional dy_dx [B,L,D,C]   (kernel_grid :87-245)
//   backward  thread = (point, level): w * grad to the 2^D corners with packed fp16 / fp32 atomics                 (kernel_grid_backward :248-340)
//             + grad_inputs[b,d] = sum grad * dy_dx                                                               (kernel_input_backward :343-369)
//   grad_tv                                                                                                        (kernel_grad_tv :506-610)
#include "ge_common.h"

#define GE_ND_MAXD 5

__device__ __forceinline__ uint32_t nd_index(uint32_t D, uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution,
                                              const uint32_t *pos_grid) {                                  // gridencoder.cu:50-84
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t result = 0;
        for (uint32_t i = 0; i < D; i++) result ^= pos_grid[i] * primes[i];
        index = result;
    }
    return index % hashmap_size;
}

template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_fwd_nd(const float *__restrict__ inputs, const T *__restrict__ grid, const int32_t *__restrict__ offsets,
                                                     T *__restrict__ outputs, uint32_t B, uint32_t D, uint32_t L, GeLevels lv, T *__restrict__ dy_dx,
                                                     uint32_t gridtype, bool align_corners, uint32_t interp, bool out_bl) {
    const uint32_t level = blockIdx.y;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float x[GE_ND_MAXD];
    bool oob = false;
    for (uint32_t d = 0; d < D; d++) { x[d] = inputs[(uint64_t)b * D + d]; oob |= (x[d] < 0 || x[d] > 1); }
    T *out = out_bl ? outputs + ((uint64_t)b * L + level) * C : outputs + ((uint64_t)level * B + b) * C;
    T *dy = dy_dx ? dy_dx + ((uint64_t)b * L + level) * D * C : nullptr;
    float results[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) results[c] = 0.0f;
    if (oob) {                                                   // :119-135
        GeVec<T, C>::st(out, results);
        if (dy) for (uint32_t i = 0; i < D * C; i++) GeT<T>::st(dy + i, 0.0f);
        return;
    }
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    const T *table = grid + (uint64_t)off0 * C;
    float pos[GE_ND_MAXD], pos_deriv[GE_ND_MAXD];
    uint32_t pos_grid[GE_ND_MAXD];
    for (uint32_t d = 0; d < D; d++) {                           // :147-159
        pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) {
            const float v = pos[d];
            pos_deriv[d] = 6 * v * (1.0f - v);
            pos[d] = v * v * fmaf(-2.0f, v, 3.0f);
        } else pos_deriv[d] = 1.0f;
    }
    for (uint32_t idx = 0; idx < (1u << D); idx++) {             // :167-191
        float w = 1;
        uint32_t pgl[GE_ND_MAXD];
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
        }
        float v[C];
        GeVec<T, C>::ld(table + (uint64_t)nd_index(D, gridtype, align_corners, hashmap_size, resolution, pgl) * C, v);
#pragma unroll
        for (uint32_t c = 0; c < C; c++) results[c] = fmaf(w, v[c], results[c]);
    }
    GeVec<T, C>::st(out, results);
    if (dy) {                                                    // :201-244
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) rg[c] = 0.0f;
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float w = scale;
                uint32_t pgl[GE_ND_MAXD];
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                pgl[gd] = pos_grid[gd];
                const uint32_t rl = nd_index(D, gridtype, align_corners, hashmap_size, resolution, pgl);
                pgl[gd] = pos_grid[gd] + 1;
                const uint32_t rr = nd_index(D, gridtype, align_corners, hashmap_size, resolution, pgl);
                float vl[C], vr[C];
                GeVec<T, C>::ld(table + (uint64_t)rl * C, vl);
                GeVec<T, C>::ld(table + (uint64_t)rr * C, vr);
#pragma unroll
                for (uint32_t c = 0; c < C; c++) rg[c] = fmaf(w * (vr[c] - vl[c]), pos_deriv[gd], rg[c]);
            }
#pragma unroll
            for (uint32_t c = 0; c < C; c++) GeT<T>::st(dy + gd * C + c, rg[c]);
        }
    }
}

template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_bwd_nd(const T *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets,
                                                     T *__restrict__ grad_grid, uint32_t B, uint32_t D, uint32_t L, GeLevels lv, uint32_t gridtype,
                                                     bool align_corners, uint32_t interp, bool grad_bl) {
    const uint32_t level = blockIdx.y;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float x[GE_ND_MAXD];
    for (uint32_t d = 0; d < D; d++) { x[d] = inputs[(uint64_t)b * D + d]; if (x[d] < 0 || x[d] > 1) return; }      // :276-281
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    float g[C];
    GeVec<T, C>::ld(grad_bl ? grad + ((uint64_t)b * L + level) * C : grad + ((uint64_t)level * B + b) * C, g);
    bool any = false;
#pragma unroll
    for (uint32_t c = 0; c < C; c++) any |= (g[c] != 0.0f);
    if (!any) return;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    T *grad_table = grad_grid + (uint64_t)off0 * C;
    float pos[GE_ND_MAXD];
    uint32_t pos_grid[GE_ND_MAXD];
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) { const float v = pos[d]; pos[d] = v * v * fmaf(-2.0f, v, 3.0f); }
    }
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pgl[GE_ND_MAXD];
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
        }
        float v[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) v[c] = w * g[c];
        GeAtomic<T>::template add<C>(grad_table + (uint64_t)nd_index(D, gridtype, align_corners, hashmap_size, resolution, pgl) * C, v);
    }
}

template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_input_bwd_nd(const T *__restrict__ grad, const T *__restrict__ dy_dx, T *__restrict__ grad_inputs, uint32_t B,
                                                           uint32_t D, uint32_t L, bool grad_bl) {
    const uint64_t total = (uint64_t)B * D;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (uint64_t)b * D);
        float r = 0;
        for (uint32_t l = 0; l < L; l++) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                const float gv = GeT<T>::ld(grad_bl ? grad + ((uint64_t)b * L + l) * C + c : grad + ((uint64_t)l * B + b) * C + c);
                r = fmaf(gv, GeT<T>::ld(dy_dx + (((uint64_t)b * L + l) * D + d) * C + c), r);
            }
        }
        GeT<T>::st(grad_inputs + t, r);
    }
}

template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_grad_tv_nd(const T *__restrict__ inputs, const T *__restrict__ grid, T *__restrict__ grad, const int32_t *__restrict__ offsets,
                                                    float weight, uint32_t B, uint32_t D, uint32_t L, GeLevels lv, uint32_t gridtype, bool align_corners) {
    const uint32_t level = blockIdx.y;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float x[GE_ND_MAXD];
    for (uint32_t d = 0; d < D; d++) { x[d] = GeT<T>::ld(inputs + (uint64_t)b * D + d); if (x[d] < 0 || x[d] > 1) return; }
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    const T *tab = grid + (uint64_t)off0 * C;
    uint32_t pos_grid[GE_ND_MAXD];
    for (uint32_t d = 0; d < D; d++) pos_grid[d] = (uint32_t)floorf(fmaf(x[d], scale, align_corners ? 0.0f : 0.5f));
    float results[C], idelta[C], center[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) { results[c] = 0; idelta[c] = 0; }
    const uint32_t row = nd_index(D, gridtype, align_corners, hashmap_size, resolution, pos_grid);
    GeVec<T, C>::ld(tab + (uint64_t)row * C, center);
    const float w = weight / (float)(2 * D);
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t cur = pos_grid[d];
        if (cur < resolution) {
            pos_grid[d] = cur + 1;
            float o[C];
            GeVec<T, C>::ld(tab + (uint64_t)nd_index(D, gridtype, align_corners, hashmap_size, resolution, pos_grid) * C, o);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) { const float gv = center[c] - o[c]; results[c] += gv; idelta[c] = fmaf(gv, gv, idelta[c]); }
        }
        if (cur > 0) {
            pos_grid[d] = cur - 1;
            float o[C];
            GeVec<T, C>::ld(tab + (uint64_t)nd_index(D, gridtype, align_corners, hashmap_size, resolution, pos_grid) * C, o);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) { const float gv = center[c] - o[c]; results[c] += gv; idelta[c] = fmaf(gv, gv, idelta[c]); }
        }
        pos_grid[d] = cur;
    }
    float v[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) v[c] = w * results[c] * (1.0f / sqrtf(idelta[c] + 1e-9f));
    GeAtomic<T>::template add<C>(grad + ((uint64_t)off0 + row) * C, v);
}

// ================================================================= host side
#define ND_DISPATCH_C(CALL)                                                                              \
    switch (C) {                                                                                         \
        case 1: CALL(1); break;                                                                          \
        case 2: CALL(2); break;                                                                          \
        case 4: CALL(4); break;                                                                          \
        case 8: CALL(8); break;                                                                          \
        default: foc_set_error("GridEncoding: C must be 1, 2, 4, or 8."); return FOC_E_INVALID;          \
    }

int ge_nd_forward(int dtype, uint32_t D, uint32_t C, const float *inputs, const void *emb, const int32_t *offsets, void *outputs, uint32_t B, uint32_t L,
                  const GeLevels &lv, void *dy_dx, uint32_t gridtype, bool ac, uint32_t interp, bool bl, hipStream_t st) {
    if (D < 1 || D > GE_ND_MAXD) { foc_set_error("GridEncoding: D must be 2, 3, 4 or 5 (got %u)", D); return FOC_E_INVALID; }
    const dim3 grid(foc_div_up(B, 256), L);
#define FWD(CC)                                                                                                                                       \
    if (dtype == FOC_F16) hipLaunchKernelGGL((k_grid_fwd_nd<__half, CC>), grid, dim3(256), 0, st, inputs, (const __half *)emb, offsets, (__half *)outputs, B, D, L, \
                                             lv, (__half *)dy_dx, gridtype, ac, interp, bl);                                                         \
    else hipLaunchKernelGGL((k_grid_fwd_nd<float, CC>), grid, dim3(256), 0, st, inputs, (const float *)emb, offsets, (float *)outputs, B, D, L, lv,    \
                            (float *)dy_dx, gridtype, ac, interp, bl)
    ND_DISPATCH_C(FWD)
#undef FWD
    FOC_CHECK_LAUNCH("grid_encode_forward");
    return FOC_OK;
}

int ge_nd_backward(int dtype, uint32_t D, uint32_t C, const void *grad, const float *inputs, const int32_t *offsets, void *grad_emb, uint32_t B, uint32_t L,
                   const GeLevels &lv, const void *dy_dx, void *grad_inputs, uint32_t gridtype, bool ac, uint32_t interp, bool bl, hipStream_t st) {
    if (D < 1 || D > GE_ND_MAXD) { foc_set_error("GridEncoding: D must be 2, 3, 4 or 5 (got %u)", D); return FOC_E_INVALID; }
    const dim3 grid(foc_div_up(B, 256), L);
    const uint32_t g1 = foc_grid_1d((uint64_t)B * D, 256);
#define BWD(CC)                                                                                                                                       \
    if (dtype == FOC_F16) {                                                                                                                           \
        hipLaunchKernelGGL((k_grid_bwd_nd<__half, CC>), grid, dim3(256), 0, st, (const __half *)grad, inputs, offsets, (__half *)grad_emb, B, D, L, lv, gridtype, ac, \
                           interp, bl);                                                                                                               \
        if (dy_dx && grad_inputs) hipLaunchKernelGGL((k_grid_input_bwd_nd<__half, CC>), dim3(g1), dim3(256), 0, st, (const __half *)grad, (const __half *)dy_dx,    \
                                                     (__half *)grad_inputs, B, D, L, bl);                                                            \
    } else {                                                                                                                                          \
        hipLaunchKernelGGL((k_grid_bwd_nd<float, CC>), grid, dim3(256), 0, st, (const float *)grad, inputs, offsets, (float *)grad_emb, B, D, L, lv, gridtype, ac,   \
                           interp, bl);                                                                                                               \
        if (dy_dx && grad_inputs) hipLaunchKernelGGL((k_grid_input_bwd_nd<float, CC>), dim3(g1), dim3(256), 0, st, (const float *)grad, (const float *)dy_dx,        \
                                                     (float *)grad_inputs, B, D, L, bl);                                                             \
    }
    ND_DISPATCH_C(BWD)
#undef BWD
    FOC_CHECK_LAUNCH("grid_encode_backward");
    return FOC_OK;
}

int ge_nd_tv(int dtype, uint32_t D, uint32_t C, const void *inputs, const void *emb, void *grad, const int32_t *offsets, float weight, uint32_t B, uint32_t L,
             const GeLevels &lv, uint32_t gridtype, bool ac, hipStream_t st) {
    if (D < 1 || D > GE_ND_MAXD) { foc_set_error("GridEncoding: D must be 2, 3, 4 or 5 (got %u)", D); return FOC_E_INVALID; }
    const dim3 grid(foc_div_up(B, 256), L);
#define TV(CC)                                                                                                                                        \
    if (dtype == FOC_F16) hipLaunchKernelGGL((k_grad_tv_nd<__half, CC>), grid, dim3(256), 0, st, (const __half *)inputs, (const __half *)emb, (__half *)grad, offsets, \
                                             weight, B, D, L, lv, gridtype, ac);                                                                     \
    else hipLaunchKernelGGL((k_grad_tv_nd<float, CC>), grid, dim3(256), 0, st, (const float *)inputs, (const float *)emb, (float *)grad, offsets, weight, B, D, L,  \
                            lv, gridtype, ac)
    ND_DISPATCH_C(TV)
#undef TV
    FOC_CHECK_LAUNCH("grad_total_variation");
    return FOC_OK;
}
