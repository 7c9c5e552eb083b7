// head.hip — the per-sample glue of the NeRF network between the encoder and the two MLPs, for callers that evaluate
// arbitrary sample lists (the occupancy-grid paths: march_rays_train / march_rays feed [M,3] points and [M,3] directions).
//
// Reference: the torch expressions of nerf/network_ff.py:51-75 (forward) —
//     sigma = trunc_exp(h[..., 0]); geo_feat = h[..., 1:]; d = SH4(d); p = zeros(1)
//     h2 = cat([d, geo_feat, p]); rgb = sigmoid(color_net(h2))
// with activation.py:8-18 (trunc_exp: exp in fp32, backward g * exp(clamp(x, -15, 15))). In the reference these are ~25
// elementwise torch kernels per call (the SH encoder alone is 16 column kernels + a stack); here two kernels forward and
// two backward, one thread per sample, 16-byte accesses. fixedstep.hip holds the one-wave-per-ray forms of the same glue
// for the fixed-step renderer. These entry points have no reference binding.
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// degree-4 real spherical harmonics (focnerf_amd/shencoder.py), same expressions in fp32
__device__ __forceinline__ void hd_sh16(float x, float y, float z, float (&o)[16]) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// h [M,16] fp16 (sigma-net output), dirs [M,3] fp32 -> sigma [M] fp32 = exp(h[:,0]), cin [M,32] fp16 = [SH16(dir) | h[:,1:16] | 0]
__global__ void __launch_bounds__(256) k_head_fwd(const _Float16 *__restrict__ h, const float *__restrict__ dirs, uint64_t M,
                                                  float *__restrict__ sigma, _Float16 *__restrict__ cin, const _Float16 *__restrict__ obj, uint32_t cin_ld) {
    // 48-wide form (FOC network with a per-image object feature, nerf/network_tcnn.py:641): [SH16 | h[:,1:16] | obj16 | 0]
    h8 ob1 = {0, 0, 0, 0, 0, 0, 0, 0}, ob2 = {0, 0, 0, 0, 0, 0, 0, 0};
    _Float16 ob0 = (_Float16)0;
    if (cin && obj) {
        ob0 = obj[0];
#pragma unroll
        for (int k = 0; k < 8; k++) ob1[k] = obj[1 + k];
#pragma unroll
        for (int k = 0; k < 7; k++) ob2[k] = obj[9 + k];
    }
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < M; s += (uint64_t)gridDim.x * 256) {
        const h8 r0 = *reinterpret_cast<const h8 *>(h + s * 16), r1 = *reinterpret_cast<const h8 *>(h + s * 16 + 8);
        if (sigma) sigma[s] = expf((float)r0[0]);
        if (cin) {
            float sh[16];
            hd_sh16(dirs[s * 3], dirs[s * 3 + 1], dirs[s * 3 + 2], sh);
            h8 c0, c1, c2, c3;
#pragma unroll
            for (int k = 0; k < 8; k++) { c0[k] = foc_f2h(sh[k]); c1[k] = foc_f2h(sh[8 + k]); }
#pragma unroll
            for (int k = 0; k < 7; k++) { c2[k] = r0[k + 1]; c3[k] = r1[k + 1]; }
            c2[7] = r1[0]; c3[7] = ob0;
            h8 *dst = reinterpret_cast<h8 *>(cin + s * cin_ld);
            dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = c3;
            if (cin_ld == 48) { dst[4] = ob1; dst[5] = ob2; }
        }
    }
}

// grad_sigma [M] fp32, grad_cin [M,32] fp16 (either may be null) -> grad_h [M,16] fp16:
// column 0 = grad_sigma * exp(clamp(h0, -15, 15)) (activation.py:16-18), columns 1..15 = grad_cin[:, 16:31]
__global__ void __launch_bounds__(256) k_head_bwd(const _Float16 *__restrict__ h, const float *__restrict__ grad_sigma,
                                                  const _Float16 *__restrict__ grad_cin, uint64_t M, _Float16 *__restrict__ grad_h, uint32_t cin_ld) {
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < M; s += (uint64_t)gridDim.x * 256) {
        h8 g2 = {0, 0, 0, 0, 0, 0, 0, 0}, g3 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (grad_cin) { g2 = *reinterpret_cast<const h8 *>(grad_cin + s * cin_ld + 16); g3 = *reinterpret_cast<const h8 *>(grad_cin + s * cin_ld + 24); }
        float g0 = 0.0f;
        if (grad_sigma) {
            float x = (float)h[s * 16];
            x = x < -15.0f ? -15.0f : (x > 15.0f ? 15.0f : x);
            g0 = grad_sigma[s] * expf(x);
        }
        h8 o0, o1;
        o0[0] = foc_f2h(g0);
#pragma unroll
        for (int k = 0; k < 7; k++) { o0[k + 1] = g2[k]; o1[k + 1] = g3[k]; }
        o1[0] = g2[7];
        *reinterpret_cast<h8 *>(grad_h + s * 16) = o0;
        *reinterpret_cast<h8 *>(grad_h + s * 16 + 8) = o1;
    }
}

// torch.sigmoid on a half tensor: evaluated in fp32, rounded to half
__device__ __forceinline__ float hd_sigmoid_h(float x) { return (float)(_Float16)(1.0f / (1.0f + expf(-x))); }

// c [M,16] fp16 (colour-net output) -> rgb [M,3] fp32 holding the half-rounded sigmoid of columns 0..2
__global__ void __launch_bounds__(256) k_rgb_fwd(const _Float16 *__restrict__ c, uint64_t M, float *__restrict__ rgb) {
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < M; s += (uint64_t)gridDim.x * 256) {
        const _Float16 *p = c + s * 16;
        rgb[s * 3] = hd_sigmoid_h((float)p[0]); rgb[s * 3 + 1] = hd_sigmoid_h((float)p[1]); rgb[s * 3 + 2] = hd_sigmoid_h((float)p[2]);
    }
}

// grad_rgb [M,3] fp32 -> grad_c [M,16] fp16: columns 0..2 = half(g) * (1 - y) * y (torch's half sigmoid backward), the rest 0
__global__ void __launch_bounds__(256) k_rgb_bwd(const _Float16 *__restrict__ c, const float *__restrict__ grad_rgb, uint64_t M,
                                                 _Float16 *__restrict__ grad_c) {
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < M; s += (uint64_t)gridDim.x * 256) {
        h8 o0 = {0, 0, 0, 0, 0, 0, 0, 0};
        const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float y = hd_sigmoid_h((float)c[s * 16 + k]);
            const float g = (float)(_Float16)grad_rgb[s * 3 + k];
            o0[k] = foc_f2h(g * (1.0f - y) * y);
        }
        *reinterpret_cast<h8 *>(grad_c + s * 16) = o0;
        *reinterpret_cast<h8 *>(grad_c + s * 16 + 8) = z;
    }
}

// dirs [M,3] fp32 -> SH degree 4 [M,16] fp32 (the module encoding.get_encoder('sphere_harmonics') returns; 17 torch kernels as an expression)
__global__ void __launch_bounds__(256) k_sh_encode(const float *__restrict__ dirs, uint64_t M, float *__restrict__ out) {
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < M; s += (uint64_t)gridDim.x * 256) {
        float sh[16];
        hd_sh16(dirs[s * 3], dirs[s * 3 + 1], dirs[s * 3 + 2], sh);
        float4 *dst = reinterpret_cast<float4 *>(out + s * 16);
#pragma unroll
        for (int k = 0; k < 4; k++) dst[k] = make_float4(sh[4 * k], sh[4 * k + 1], sh[4 * k + 2], sh[4 * k + 3]);
    }
}

extern "C" {

int foc_sh_encode(const float *dirs, uint64_t M, float *out, void *stream) {
    FocDeviceGuard foc_guard_(stream, dirs);
    if (M == 0) return FOC_OK;
    FOC_REQUIRE(dirs && out, FOC_E_INVALID, "sh_encode: null pointer");
    hipLaunchKernelGGL(k_sh_encode, dim3(foc_grid_1d(M, 256)), dim3(256), 0, (hipStream_t)stream, dirs, M, out);
    FOC_CHECK_LAUNCH("sh_encode");
    return FOC_OK;
}

int foc_sample_head_forward(const void *h, const float *dirs, uint64_t M, float *sigma, void *cin, const void *obj_feat, uint32_t cin_width, void *stream) {
    FocDeviceGuard foc_guard_(stream, h);
    if (M == 0) return FOC_OK;
    FOC_REQUIRE(h && (sigma || cin) && (!cin || dirs), FOC_E_INVALID, "sample_head_forward: null pointer");
    FOC_REQUIRE(cin_width == 32 || cin_width == 48, FOC_E_INVALID, "sample_head_forward: cin_width must be 32 or 48 (got %u)", cin_width);
    FOC_REQUIRE(!obj_feat || cin_width == 48, FOC_E_INVALID, "sample_head_forward: an object feature needs the 48-wide colour input");
    hipLaunchKernelGGL(k_head_fwd, dim3(foc_grid_1d(M, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)h, dirs, M, sigma, (_Float16 *)cin,
                       (const _Float16 *)obj_feat, cin_width);
    FOC_CHECK_LAUNCH("sample_head_forward");
    return FOC_OK;
}

int foc_sample_head_backward(const void *h, const float *grad_sigma, const void *grad_cin, uint64_t M, void *grad_h, uint32_t cin_width, void *stream) {
    FocDeviceGuard foc_guard_(stream, h);
    if (M == 0) return FOC_OK;
    FOC_REQUIRE(h && grad_h, FOC_E_INVALID, "sample_head_backward: null pointer");
    FOC_REQUIRE(cin_width == 32 || cin_width == 48, FOC_E_INVALID, "sample_head_backward: cin_width must be 32 or 48 (got %u)", cin_width);
    hipLaunchKernelGGL(k_head_bwd, dim3(foc_grid_1d(M, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)h, grad_sigma,
                       (const _Float16 *)grad_cin, M, (_Float16 *)grad_h, cin_width);
    FOC_CHECK_LAUNCH("sample_head_backward");
    return FOC_OK;
}

int foc_rgb_head_forward(const void *c, uint64_t M, float *rgb, void *stream) {
    FocDeviceGuard foc_guard_(stream, c);
    if (M == 0) return FOC_OK;
    FOC_REQUIRE(c && rgb, FOC_E_INVALID, "rgb_head_forward: null pointer");
    hipLaunchKernelGGL(k_rgb_fwd, dim3(foc_grid_1d(M, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)c, M, rgb);
    FOC_CHECK_LAUNCH("rgb_head_forward");
    return FOC_OK;
}

int foc_rgb_head_backward(const void *c, const float *grad_rgb, uint64_t M, void *grad_c, void *stream) {
    FocDeviceGuard foc_guard_(stream, c);
    if (M == 0) return FOC_OK;
    FOC_REQUIRE(c && grad_rgb && grad_c, FOC_E_INVALID, "rgb_head_backward: null pointer");
    hipLaunchKernelGGL(k_rgb_bwd, dim3(foc_grid_1d(M, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)c, grad_rgb, M, (_Float16 *)grad_c);
    FOC_CHECK_LAUNCH("rgb_head_backward");
    return FOC_OK;
}

} // extern "C"
