// fixedstep.hip — FOC's default render path (fixed num_steps per ray, no occupancy grid) as fused ops.
//
// Reference: the torch code of nerf/renderer.py:145-221 (identical in COMBINED.py:451-534), plus the
// glue of nerf/network_ff.py:51-134 that sits between the encoder and the two MLPs:
//     z = near + (far-near)*linspace(0,1,T) [+ (rand-0.5)*(far-near)/T]; xyz = clip(o + d z, aabb)
//     sigma = trunc_exp(h[:,0]); alpha = 1 - exp(-delta*density_scale*sigma)
//     weights = alpha * cumprod([1, 1-alpha+1e-15])[:-1]
//     colour input = [SH16(dir) | h[:,1:16] | 0]; rgb = sigmoid(colour_net(.)) where weights > thresh, else 0
//     image = sum w rgb + (1 - sum w) bg ; depth = sum w clamp((z-near)/(far-near),0,1)
// In the reference these are ~60 small torch kernels per step (≈35 % of the measured step once the
// encoder and MLP kernels are native). Here: one sample-generation kernel, and ONE WAVE PER RAY kernels for
// the density head (sigma, weights by a DPP product-scan, colour-net input rows) and for the composite,
// forward and backward. The colour network is evaluated densely and masked (same values as the reference's
// gather -> MLP -> scatter, without the nonzero/index round trips).
// SURVEY.md §8(f)-3; these entry points have no reference binding.
#include "common.h"
#include <float.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct FsGeom { float near, far, span, sample_dist, step; };

// Block-interleaved sample order of the inference path (k_fs_sample): rows of 64 consecutive rays interleaved by depth.
#define FS_RAY_BLOCK 64u
__host__ __device__ __forceinline__ uint64_t fs_block_row(uint32_t n, uint32_t i, uint32_t T) {
    return (uint64_t)(n / FS_RAY_BLOCK) * FS_RAY_BLOCK * T + (uint64_t)i * FS_RAY_BLOCK + n % FS_RAY_BLOCK;
}

__device__ __forceinline__ FsGeom fs_geom(const float *__restrict__ nears, const float *__restrict__ fars, uint32_t n, uint32_t T) {
    FsGeom g;
    g.near = nears[n]; g.far = fars[n];
    g.span = g.far - g.near;
    g.sample_dist = g.span / (float)T;
    g.step = 1.0f / (float)(T - 1);
    return g;
}
// torch.linspace(0, 1, T) as torch's DEVICE kernel fills it: symmetric halves, and the upper half `end - step*k` is one
// fused multiply-add (the device compilers — nvcc for the reference, hipcc for torch-ROCm — contract it; torch's CPU kernel
// and therefore the CPU oracle round twice). Then z = near + span * lin [+ (u - 0.5) * sample_dist], separate torch ops.
__device__ __forceinline__ float fs_z(const FsGeom &g, uint32_t i, uint32_t T, const float *__restrict__ noise, uint64_t s) {
    const float lin = (i < T / 2) ? (g.step * (float)i) : fmaf(-g.step, (float)(T - 1 - i), 1.0f);
    float z = g.near + g.span * lin;
    if (noise) z = z + (noise[s] - 0.5f) * g.sample_dist;
    return z;
}

// the same value from a noise draw that is already in a register (NOISE = false: no jitter term at all, like fs_z with a null pointer).
// The tail kernels load both draws a sample needs up front, unconditionally, so that all loads of an iteration are in flight together:
// behind `if (noise)` / `if (i + 1 < T)` each load was its own round trip (s_waitcnt vmcnt(0) after every one of them).
template <bool NOISE>
__device__ __forceinline__ float fs_zu(const FsGeom &g, uint32_t i, uint32_t T, float u) {
    const float lin = (i < T / 2) ? (g.step * (float)i) : fmaf(-g.step, (float)(T - 1 - i), 1.0f);
    float z = g.near + g.span * lin;
    if (NOISE) z = z + (u - 0.5f) * g.sample_dist;
    return z;
}

// degree-4 real spherical harmonics (focnerf_amd/shencoder.py), same expressions in fp32
__device__ __forceinline__ void fs_sh16(float x, float y, float z, float (&o)[16]) {
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

// ---------------------------------------------------------------- sample generation
__global__ void __launch_bounds__(256) k_fs_sample(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ nears,
                                                   const float *__restrict__ fars, const float *__restrict__ aabb, const float *__restrict__ noise,
                                                   uint32_t N, uint32_t T, float bound, float *__restrict__ xyzs, float *__restrict__ enc_in,
                                                   _Float16 *__restrict__ ray_sh, uint32_t ray_block) {
    // ray_block == 0: sample (n, i) stands at row n*T + i. ray_block == FS_RAY_BLOCK: the rows of FS_RAY_BLOCK consecutive rays are
    // interleaved, row = (n / 64)*64*T + i*64 + n % 64 (fs_block_row), the last block padded with copies of ray N-1 — the lanes of a wave
    // of the kernels downstream are then 64 NEIGHBOURING RAYS AT ONE DEPTH instead of 64 depths of one ray, and the level-major encoder
    // forward finds most of a wave's corner rows in a few cache lines (0.34 -> 0.17 ms per 2 M samples of an 800 x 800 view).
    const uint64_t total = ray_block ? (uint64_t)((N + FS_RAY_BLOCK - 1) / FS_RAY_BLOCK) * FS_RAY_BLOCK * T : (uint64_t)N * T;
    const float a0 = aabb[0], a1 = aabb[1], a2 = aabb[2], a3 = aabb[3], a4 = aabb[4], a5 = aabb[5];
    const float two_b = 2 * bound;
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < total; s += (uint64_t)gridDim.x * 256) {
        uint32_t n, i;
        bool own = true;                                   // false for the padding rows of the last block
        if (ray_block) {
            const uint32_t blk = (uint32_t)(s / ((uint64_t)FS_RAY_BLOCK * T)), in = (uint32_t)(s - (uint64_t)blk * FS_RAY_BLOCK * T);
            i = in / FS_RAY_BLOCK;
            n = blk * FS_RAY_BLOCK + in % FS_RAY_BLOCK;
            own = n < N;
            n = own ? n : N - 1;
        } else {
            n = (uint32_t)(s / T); i = (uint32_t)(s - (uint64_t)n * T);
        }
        const FsGeom g = fs_geom(nears, fars, n, T);
        const float z = fs_z(g, i, T, noise, (uint64_t)n * T + i);     // the noise array is ray-major in both orders
        // torch: rays_o + rays_d * z (two kernels, two roundings), then min(max(., aabb_lo), aabb_hi)
        float x = rays_o[n * 3] + rays_d[n * 3] * z, y = rays_o[n * 3 + 1] + rays_d[n * 3 + 1] * z, w = rays_o[n * 3 + 2] + rays_d[n * 3 + 2] * z;
        x = fminf(fmaxf(x, a0), a3); y = fminf(fmaxf(y, a1), a4); w = fminf(fmaxf(w, a2), a5);
        if (xyzs) { xyzs[s * 3] = x; xyzs[s * 3 + 1] = y; xyzs[s * 3 + 2] = w; }
        if (enc_in) { enc_in[s * 3] = (x + bound) / two_b; enc_in[s * 3 + 1] = (y + bound) / two_b; enc_in[s * 3 + 2] = (w + bound) / two_b; }
        if (ray_sh && i == 0 && own) {                     // the ray's SH row as it stands in the colour-net input (fp16), once per ray
            float sh[16];
            fs_sh16(rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2], sh);
            h8 lo, hi;
#pragma unroll
            for (int k = 0; k < 8; k++) { lo[k] = foc_f2h(sh[k]); hi[k] = foc_f2h(sh[8 + k]); }
            h8 *dst = reinterpret_cast<h8 *>(ray_sh + (uint64_t)n * 16);
            dst[0] = lo; dst[1] = hi;
        }
    }
}

// ---------------------------------------------------------------- density head, forward (one wave per ray)
// h [M,16] fp16 (sigma-net output); outputs sigma [M], trans [M] (transmittance BEFORE each sample), weights [M],
// weights_sum [N], depth [N], colour-net input rows cin [M,32] fp16 (may be null).
__global__ void __launch_bounds__(256) k_fs_head_fwd(const _Float16 *__restrict__ h, const float *__restrict__ rays_d, const float *__restrict__ nears,
                                                     const float *__restrict__ fars, const float *__restrict__ noise, uint32_t N, uint32_t T,
                                                     float density_scale, float *__restrict__ sigma_out, float *__restrict__ trans_out,
                                                     float *__restrict__ weights_out, float *__restrict__ weights_sum, float *__restrict__ depth,
                                                     _Float16 *__restrict__ cin, const _Float16 *__restrict__ obj, uint32_t cin_ld) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const FsGeom g = fs_geom(nears, fars, n, T);
    h8 shlo, shhi;
    h8 ob1 = {0, 0, 0, 0, 0, 0, 0, 0}, ob2 = {0, 0, 0, 0, 0, 0, 0, 0};   // cin columns 32..39 / 40..47 of the 48-wide form: obj[1..8], obj[9..15] | 0
    _Float16 ob0 = (_Float16)0;                                          // column 31: obj[0] (the 32-wide form has its zero pad there)
    if (cin) {
        float sh[16];
        fs_sh16(rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2], sh);
#pragma unroll
        for (int k = 0; k < 8; k++) { shlo[k] = foc_f2h(sh[k]); shhi[k] = foc_f2h(sh[8 + k]); }
        if (obj) {
            ob0 = obj[0];
#pragma unroll
            for (int k = 0; k < 8; k++) ob1[k] = obj[1 + k];
#pragma unroll
            for (int k = 0; k < 7; k++) ob2[k] = obj[9 + k];
        }
    }
    float Tc = 1.0f, ws = 0, dp = 0;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < T;
        const uint64_t s = (uint64_t)n * T + (valid ? i : T - 1);
        h8 r0, r1;
        if (cin) { r0 = *reinterpret_cast<const h8 *>(h + s * 16); r1 = *reinterpret_cast<const h8 *>(h + s * 16 + 8); }
        else r0[0] = h[s * 16];
        const float sigma = expf((float)r0[0]);                                  // trunc_exp forward (activation.py:9)
        const float z = fs_z(g, valid ? i : T - 1, T, noise, s);
        float delta = g.sample_dist;
        if (i + 1 < T) delta = fs_z(g, i + 1, T, noise, s + 1) - z;
        const float alpha = valid ? 1 - expf((-delta * density_scale) * sigma) : 0.0f;
        const float om = valid ? (1 - alpha + 1e-15f) : 1.0f;
        const float P = wave_incl_prod(om, (int)lane);
        float Pex = __shfl_up(P, 1, 64);
        if (lane == 0) Pex = 1.0f;
        const float Tb = Tc * Pex;
        const float w = alpha * Tb;
        if (valid) {
            sigma_out[s] = sigma; trans_out[s] = Tb; weights_out[s] = w;
            float oz = (z - g.near) / g.span;
            oz = oz < 0.0f ? 0.0f : (oz > 1.0f ? 1.0f : oz);                       // keeps NaN (0/0 on rays that miss the box), like torch.clamp
            ws += w; dp += w * oz;
            if (cin) {
                h8 c2, c3;
#pragma unroll
                for (int k = 0; k < 7; k++) { c2[k] = r0[k + 1]; c3[k] = r1[k + 1]; }
                c2[7] = r1[0]; c3[7] = ob0;
                h8 *dst = reinterpret_cast<h8 *>(cin + s * cin_ld);
                dst[0] = shlo; dst[1] = shhi; dst[2] = c2; dst[3] = c3;
                if (cin_ld == 48) { dst[4] = ob1; dst[5] = ob2; }
            }
        }
        Tc *= __shfl(P, 63, 64);
    }
    ws = wave_sum(ws); dp = wave_sum(dp);
    if (lane == 0) { weights_sum[n] = ws; depth[n] = dp; }
}

// reverse (suffix) inclusive sum across the wave
__device__ __forceinline__ float wave_suffix_incl_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_down(v, o, 64);
        if (lane + o < 64) v += u;
    }
    return v;
}

// ---------------------------------------------------------------- density head, backward
// grad_w [M] (from the composite), grad_ws [N], grad_depth [N] (either may be null), grad_cin [M,32] fp16 (may be null)
// -> grad_h [M,16] fp16: column 0 through weights -> alpha -> sigma -> trunc_exp, columns 1..15 = grad_cin[:,16:31].
__global__ void __launch_bounds__(256) k_fs_head_bwd(const _Float16 *__restrict__ h, const float *__restrict__ sigma_in, const float *__restrict__ trans_in,
                                                     const float *__restrict__ nears, const float *__restrict__ fars, const float *__restrict__ noise,
                                                     const float *__restrict__ grad_w, const float *__restrict__ grad_ws, const float *__restrict__ grad_depth,
                                                     const _Float16 *__restrict__ grad_cin, uint32_t N, uint32_t T, float density_scale,
                                                     _Float16 *__restrict__ grad_h, uint32_t cin_ld) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const FsGeom g = fs_geom(nears, fars, n, T);
    const float gws = grad_ws ? grad_ws[n] : 0.0f, gdp = grad_depth ? grad_depth[n] : 0.0f;
    float S_carry = 0.0f;                      // sum over samples AFTER the current chunk of g_j * w_j
    const uint32_t n_chunks = (T + 63) / 64;
    for (uint32_t cidx = n_chunks; cidx-- > 0;) {
        const uint32_t i = cidx * 64 + lane;
        const bool valid = i < T;
        const uint64_t s = (uint64_t)n * T + (valid ? i : T - 1);
        const float sigma = sigma_in[s], Tb = trans_in[s];
        const float z = fs_z(g, valid ? i : T - 1, T, noise, s);
        float delta = g.sample_dist;
        if (i + 1 < T) delta = fs_z(g, i + 1, T, noise, s + 1) - z;
        const float ex = expf((-delta * density_scale) * sigma);            // 1 - alpha
        const float alpha = 1 - ex;
        const float om = 1 - alpha + 1e-15f;
        float oz = (z - g.near) / g.span;
        oz = oz < 0.0f ? 0.0f : (oz > 1.0f ? 1.0f : oz);
        float gi = (grad_w ? grad_w[s] : 0.0f) + gws;
        if (gdp != 0.0f) gi += gdp * oz;
        const float gw_i = valid ? gi * (alpha * Tb) : 0.0f;                // g_i * w_i
        const float incl = wave_suffix_incl_sum(gw_i, (int)lane);
        const float S_i = S_carry + (incl - gw_i);                          // strictly after i
        const float dalpha = gi * Tb - S_i / om;
        const float dsigma = dalpha * (delta * density_scale) * ex;
        const _Float16 h0 = h[s * 16];
        const float dh0 = dsigma * expf(fminf(fmaxf((float)h0, -15.0f), 15.0f));   // trunc_exp backward (activation.py:15)
        if (valid) {
            h8 o0, o1;
            if (grad_cin) {
                const h8 c2 = *reinterpret_cast<const h8 *>(grad_cin + s * cin_ld + 16), c3 = *reinterpret_cast<const h8 *>(grad_cin + s * cin_ld + 24);
#pragma unroll
                for (int k = 0; k < 7; k++) { o0[k + 1] = c2[k]; o1[k + 1] = c3[k]; }
                o1[0] = c2[7];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) { o0[k] = (_Float16)0; o1[k] = (_Float16)0; }
            }
            o0[0] = foc_f2h(dh0);
            h8 *dst = reinterpret_cast<h8 *>(grad_h + s * 16);
            dst[0] = o0; dst[1] = o1;
        }
        S_carry += __shfl(incl, 0, 64);
    }
}

// ---------------------------------------------------------------- composite, forward / backward
// c [M,16] fp16 (colour-net output, rgb logits in columns 0..2), weights [M]; bg: per-ray [N,3] or scalar.
__device__ __forceinline__ float fs_sigmoid_h(float x) {
    // the reference applies torch.sigmoid to the HALF tensor (network_ff.py:117): fp32 math, one rounding to fp16
    return (float)(_Float16)(1.0f / (1.0f + expf(-x)));
}

__global__ void __launch_bounds__(256) k_fs_composite_fwd(const _Float16 *__restrict__ c, const float *__restrict__ weights, const float *__restrict__ bg_ray,
                                                          float bg_scalar, uint32_t N, uint32_t T, float thresh, float *__restrict__ image) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float r = 0, g = 0, b = 0, ws = 0;
    for (uint32_t i = lane; i < T; i += 64) {
        const uint64_t s = (uint64_t)n * T + i;
        const float w = weights[s];
        ws += w;
        if (w > thresh) {
            const uint2 raw = *reinterpret_cast<const uint2 *>(c + s * 16);
            const _Float16 *cc = reinterpret_cast<const _Float16 *>(&raw);
            r += w * fs_sigmoid_h((float)cc[0]); g += w * fs_sigmoid_h((float)cc[1]); b += w * fs_sigmoid_h((float)cc[2]);
        }
    }
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws);
    if (lane == 0) {
        const float b0 = bg_ray ? bg_ray[n * 3] : bg_scalar, b1 = bg_ray ? bg_ray[n * 3 + 1] : bg_scalar, b2 = bg_ray ? bg_ray[n * 3 + 2] : bg_scalar;
        image[n * 3] = r + (1 - ws) * b0; image[n * 3 + 1] = g + (1 - ws) * b1; image[n * 3 + 2] = b + (1 - ws) * b2;
    }
}

__global__ void __launch_bounds__(256) k_fs_composite_bwd(const float *__restrict__ grad_image, const _Float16 *__restrict__ c, const float *__restrict__ weights,
                                                          const float *__restrict__ bg_ray, float bg_scalar, uint32_t N, uint32_t T, float thresh,
                                                          _Float16 *__restrict__ grad_c, float *__restrict__ grad_w) {
    const uint64_t total = (uint64_t)N * T;
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < total; s += (uint64_t)gridDim.x * 256) {
        const uint32_t n = (uint32_t)(s / T);
        const float g0 = grad_image[n * 3], g1 = grad_image[n * 3 + 1], g2 = grad_image[n * 3 + 2];
        const float b0 = bg_ray ? bg_ray[n * 3] : bg_scalar, b1 = bg_ray ? bg_ray[n * 3 + 1] : bg_scalar, b2 = bg_ray ? bg_ray[n * 3 + 2] : bg_scalar;
        const float w = weights[s];
        float gw = -(g0 * b0 + g1 * b1 + g2 * b2);
        h8 o0, o1;
#pragma unroll
        for (int k = 0; k < 8; k++) { o0[k] = (_Float16)0; o1[k] = (_Float16)0; }
        if (w > thresh) {
            const uint2 raw = *reinterpret_cast<const uint2 *>(c + s * 16);
            const _Float16 *cc = reinterpret_cast<const _Float16 *>(&raw);
            const float y0 = fs_sigmoid_h((float)cc[0]), y1 = fs_sigmoid_h((float)cc[1]), y2 = fs_sigmoid_h((float)cc[2]);
            gw += g0 * y0 + g1 * y1 + g2 * y2;
            o0[0] = foc_f2h(g0 * w * y0 * (1 - y0)); o0[1] = foc_f2h(g1 * w * y1 * (1 - y1)); o0[2] = foc_f2h(g2 * w * y2 * (1 - y2));
        }
        grad_w[s] = gw;
        h8 *dst = reinterpret_cast<h8 *>(grad_c + s * 16);
        dst[0] = o0; dst[1] = o1;
    }
}

// ---------------------------------------------------------------- training tail: density head + composite in one pass per direction
// What k_fs_head_fwd (without the colour-net input) and k_fs_composite_fwd compute, one wave per ray, with the same per-lane
// accumulation order, so the results are the same bits; the weights are not re-read. c [M,16] fp16 = colour-net output.
template <bool NOISE>
__global__ void __launch_bounds__(256) k_fs_tail_fwd(const _Float16 *__restrict__ h, const _Float16 *__restrict__ c, const float *__restrict__ nears,
                                                     const float *__restrict__ fars, const float *__restrict__ noise, const float *__restrict__ bg_ray,
                                                     float bg_scalar, uint32_t N, uint32_t T, float density_scale, float thresh,
                                                     float *__restrict__ sigma_out, float *__restrict__ trans_out, float *__restrict__ weights_out,
                                                     float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image, uint32_t c_ld,
                                                     float *__restrict__ ray_sumsq) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const FsGeom g = fs_geom(nears, fars, n, T);
    float Tc = 1.0f, ws = 0, dp = 0, r = 0, gg = 0, b = 0, sq = 0;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < T;
        const uint32_t ic = valid ? i : T - 1;
        const uint64_t s = (uint64_t)n * T + ic;
        const _Float16 h0 = h[s * 16];
        const uint2 raw = *reinterpret_cast<const uint2 *>(c + s * c_ld);
        float u0 = 0.0f, u1 = 0.0f;
        if (NOISE) { u0 = noise[s]; u1 = noise[(uint64_t)n * T + min(ic + 1u, T - 1u)]; }
        const float sigma = expf((float)h0);                                    // trunc_exp forward (activation.py:9)
        const float z = fs_zu<NOISE>(g, ic, T, u0);
        float delta = g.sample_dist;
        if (i + 1 < T) delta = fs_zu<NOISE>(g, i + 1, T, u1) - z;
        const float alpha = valid ? 1 - expf((-delta * density_scale) * sigma) : 0.0f;
        const float om = valid ? (1 - alpha + 1e-15f) : 1.0f;
        const float P = wave_incl_prod(om, (int)lane);
        float Pex = __shfl_up(P, 1, 64);
        if (lane == 0) Pex = 1.0f;
        const float Tb = Tc * Pex;
        const float w = alpha * Tb;
        if (valid) {
            sigma_out[s] = sigma; trans_out[s] = Tb; weights_out[s] = w;
            float oz = (z - g.near) / g.span;
            oz = oz < 0.0f ? 0.0f : (oz > 1.0f ? 1.0f : oz);
            ws += w; dp += w * oz; sq = fmaf(sigma, sigma, sq);
            if (w > thresh) {
                const _Float16 *cc = reinterpret_cast<const _Float16 *>(&raw);
                r += w * fs_sigmoid_h((float)cc[0]); gg += w * fs_sigmoid_h((float)cc[1]); b += w * fs_sigmoid_h((float)cc[2]);
            }
        }
        Tc *= __shfl(P, 63, 64);
    }
    ws = wave_sum(ws); dp = wave_sum(dp); r = wave_sum(r); gg = wave_sum(gg); b = wave_sum(b);
    if (ray_sumsq) {                                       // wave-uniform
        sq = wave_sum(sq);
        if (lane == 0) ray_sumsq[n] = sq;
    }
    if (lane == 0) {
        const float b0 = bg_ray ? bg_ray[n * 3] : bg_scalar, b1 = bg_ray ? bg_ray[n * 3 + 1] : bg_scalar, b2 = bg_ray ? bg_ray[n * 3 + 2] : bg_scalar;
        image[n * 3] = r + (1 - ws) * b0; image[n * 3 + 1] = gg + (1 - ws) * b1; image[n * 3 + 2] = b + (1 - ws) * b2;
        weights_sum[n] = ws; depth[n] = dp;
    }
}

// a value the compiler cannot fold: exp(+-15) below must come out of the device's expf like every other exp of this file
__device__ __forceinline__ float fs_opaque(float v) { asm volatile("" : "+v"(v)); return v; }

// k_fs_composite_bwd and k_fs_head_bwd (column 0 only) in one pass: grad_image [N,3], grad_ws / grad_depth [N] (may be null)
// -> grad_c [M,16] fp16 and grad_h0 [M] fp16. The gradient of the weights never leaves the lane. trunc_exp's backward factor
// exp(clamp(h0, -15, 15)) is taken as clamp(sigma, exp(-15), exp(15)) — the same bits, expf being monotonic — so h is not read.
template <bool NOISE>
__global__ void __launch_bounds__(256) k_fs_tail_bwd(const float *__restrict__ grad_image, const float *__restrict__ grad_ws, const float *__restrict__ grad_depth,
                                                     const _Float16 *__restrict__ c, const float *__restrict__ sigma_in, const float *__restrict__ trans_in,
                                                     const float *__restrict__ weights, const float *__restrict__ nears, const float *__restrict__ fars,
                                                     const float *__restrict__ noise, const float *__restrict__ bg_ray, float bg_scalar, uint32_t N, uint32_t T,
                                                     float density_scale, float thresh, _Float16 *__restrict__ grad_c, _Float16 *__restrict__ grad_h0, uint32_t c_ld,
                                                     const float *__restrict__ grad_sumsq) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const FsGeom g = fs_geom(nears, fars, n, T);
    const float gws = grad_ws ? grad_ws[n] : 0.0f, gdp = grad_depth ? grad_depth[n] : 0.0f;
    const float gsq2 = grad_sumsq ? 2.0f * grad_sumsq[n] : 0.0f;      // d(sum sigma^2)/d sigma = 2 sigma
    const float g0 = grad_image[n * 3], g1 = grad_image[n * 3 + 1], g2 = grad_image[n * 3 + 2];
    const float b0 = bg_ray ? bg_ray[n * 3] : bg_scalar, b1 = bg_ray ? bg_ray[n * 3 + 1] : bg_scalar, b2 = bg_ray ? bg_ray[n * 3 + 2] : bg_scalar;
    const float e_lo = expf(fs_opaque(-15.0f)), e_hi = expf(fs_opaque(15.0f));
    float S_carry = 0.0f;
    const uint32_t n_chunks = (T + 63) / 64;
    for (uint32_t cidx = n_chunks; cidx-- > 0;) {
        const uint32_t i = cidx * 64 + lane;
        const bool valid = i < T;
        const uint32_t ic = valid ? i : T - 1;
        const uint64_t s = (uint64_t)n * T + ic;
        const float sigma = sigma_in[s], Tb = trans_in[s], w = weights[s];
        const uint2 raw = *reinterpret_cast<const uint2 *>(c + s * c_ld);       // loaded for every sample (used above the weight threshold only)
        float u0 = 0.0f, u1 = 0.0f;
        if (NOISE) { u0 = noise[s]; u1 = noise[(uint64_t)n * T + min(ic + 1u, T - 1u)]; }
        // ---- composite backward (k_fs_composite_bwd)
        float gw = -(g0 * b0 + g1 * b1 + g2 * b2);
        h8 o0, o1;
#pragma unroll
        for (int k = 0; k < 8; k++) { o0[k] = (_Float16)0; o1[k] = (_Float16)0; }
        if (w > thresh) {
            const _Float16 *cc = reinterpret_cast<const _Float16 *>(&raw);
            const float y0 = fs_sigmoid_h((float)cc[0]), y1 = fs_sigmoid_h((float)cc[1]), y2 = fs_sigmoid_h((float)cc[2]);
            gw += g0 * y0 + g1 * y1 + g2 * y2;
            o0[0] = foc_f2h(g0 * w * y0 * (1 - y0)); o0[1] = foc_f2h(g1 * w * y1 * (1 - y1)); o0[2] = foc_f2h(g2 * w * y2 * (1 - y2));
        }
        if (valid) {
            if (c_ld == 4u) *reinterpret_cast<uint2 *>(grad_c + s * 4) = *reinterpret_cast<const uint2 *>(&o0);     // rgb + 1 zero: the columns that exist
            else { h8 *dst = reinterpret_cast<h8 *>(grad_c + s * 16); dst[0] = o0; dst[1] = o1; }
        }
        // ---- density head backward (k_fs_head_bwd)
        const float z = fs_zu<NOISE>(g, ic, T, u0);
        float delta = g.sample_dist;
        if (i + 1 < T) delta = fs_zu<NOISE>(g, i + 1, T, u1) - z;
        const float ex = expf((-delta * density_scale) * sigma);
        const float alpha = 1 - ex;
        const float om = 1 - alpha + 1e-15f;
        float oz = (z - g.near) / g.span;
        oz = oz < 0.0f ? 0.0f : (oz > 1.0f ? 1.0f : oz);
        float gi = gw + gws;
        if (gdp != 0.0f) gi += gdp * oz;
        const float gw_i = valid ? gi * (alpha * Tb) : 0.0f;
        const float incl = wave_suffix_incl_sum(gw_i, (int)lane);
        const float S_i = S_carry + (incl - gw_i);
        const float dalpha = gi * Tb - S_i / om;
        float dsigma = dalpha * (delta * density_scale) * ex;
        if (gsq2 != 0.0f) dsigma = fmaf(gsq2, sigma, dsigma);          // the outside-mask density criterion's share (nerf/renderer.py:163-165)
        const float dh0 = dsigma * fminf(fmaxf(sigma, e_lo), e_hi);
        if (valid) grad_h0[s] = foc_f2h(dh0);
        S_carry += __shfl(incl, 0, 64);
    }
}

// ---------------------------------------------------------------- inference tail: weights + mask + composite in one pass
// sigma [M] fp32 and rgb [M,3] fp32 (foc_nerf_field_inference) -> image [N,3], depth [N], weights_sum [N]; one wave per ray, the
// transmittance scan of k_fs_head_fwd and the masked sum of k_fs_composite_fwd without the weights / trans arrays in between.
// rgb_masked (may be NULL): rgb * [w > thresh], the per-sample colour field the reference's run() returns (renderer.py:187).
// PACK: additionally (or only: image / depth / weights_sum may then be null) writes the object's per-sample field as ONE float4 per sample,
// field4[s] = (sigma, rgb where w > thresh else 0) — the (`densities`, `rgbs`) pair COMBINED.py merges across objects (:598-618), in the
// layout the per-ray exchange and the fused select + composite read with one 16-byte access per lane (csrc/combine.hip).
struct FsRayAcc { float Tc, ws, dp, r, g, b; };

// 64 samples of ray n, sample i on the lane (sigma / c0..c2 are that sample's values; unread where i >= T): weights by wave scan, the
// masked sums on the lane, the per-sample outputs written ray-major.
template <bool PACK>
__device__ __forceinline__ void fs_infer_tile(FsRayAcc &a, const FsGeom &g, uint32_t n, uint32_t i, uint32_t lane, uint32_t T, float sigma, float c0, float c1,
                                              float c2, const float *__restrict__ noise, float density_scale, float thresh, float *__restrict__ rgb_masked,
                                              float4 *__restrict__ field4, float *__restrict__ sigma_rm) {
    const bool valid = i < T;
    const uint64_t s = (uint64_t)n * T + (valid ? i : T - 1);
    const float z = fs_z(g, valid ? i : T - 1, T, noise, s);
    float delta = g.sample_dist;
    if (i + 1 < T) delta = fs_z(g, i + 1, T, noise, s + 1) - z;
    const float alpha = valid ? 1 - expf((-delta * density_scale) * sigma) : 0.0f;
    const float om = valid ? (1 - alpha + 1e-15f) : 1.0f;
    const float P = wave_incl_prod(om, (int)lane);
    float Pex = __shfl_up(P, 1, 64);
    if (lane == 0) Pex = 1.0f;
    const float w = alpha * (a.Tc * Pex);
    if (valid) {
        float oz = (z - g.near) / g.span;
        oz = oz < 0.0f ? 0.0f : (oz > 1.0f ? 1.0f : oz);
        a.ws += w; a.dp += w * oz;
        const bool on = w > thresh;
        if (on) { a.r += w * c0; a.g += w * c1; a.b += w * c2; }
        if (rgb_masked) { rgb_masked[s * 3] = on ? c0 : 0.0f; rgb_masked[s * 3 + 1] = on ? c1 : 0.0f; rgb_masked[s * 3 + 2] = on ? c2 : 0.0f; }
        if (PACK) field4[s] = make_float4(sigma, on ? c0 : 0.0f, on ? c1 : 0.0f, on ? c2 : 0.0f);
        if (sigma_rm) sigma_rm[s] = sigma;
    }
    a.Tc *= __shfl(P, 63, 64);
}

template <bool PACK>
__device__ __forceinline__ void fs_infer_finish(FsRayAcc &a, uint32_t n, uint32_t lane, const float *__restrict__ bg_ray, float bg_scalar,
                                                float *__restrict__ image, float *__restrict__ depth, float *__restrict__ weights_sum) {
    const float ws = wave_sum(a.ws), dp = wave_sum(a.dp), r = wave_sum(a.r), gg = wave_sum(a.g), b = wave_sum(a.b);
    if (lane == 0 && (!PACK || image)) {
        const float b0 = bg_ray ? bg_ray[n * 3] : bg_scalar, b1 = bg_ray ? bg_ray[n * 3 + 1] : bg_scalar, b2 = bg_ray ? bg_ray[n * 3 + 2] : bg_scalar;
        image[n * 3] = r + (1 - ws) * b0; image[n * 3 + 1] = gg + (1 - ws) * b1; image[n * 3 + 2] = b + (1 - ws) * b2;
        depth[n] = dp;
        weights_sum[n] = ws;
    }
}

template <bool PACK>
__global__ void __launch_bounds__(256) k_fs_render_infer(const float *__restrict__ sigma_in, const float *__restrict__ rgb_in, const float *__restrict__ nears,
                                                         const float *__restrict__ fars, const float *__restrict__ noise, const float *__restrict__ bg_ray,
                                                         float bg_scalar, uint32_t N, uint32_t T, float density_scale, float thresh,
                                                         float *__restrict__ image, float *__restrict__ depth, float *__restrict__ weights_sum,
                                                         float *__restrict__ rgb_masked, float4 *__restrict__ field4) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const FsGeom g = fs_geom(nears, fars, n, T);
    FsRayAcc a = {1.0f, 0, 0, 0, 0, 0};
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < T;
        const uint64_t s = (uint64_t)n * T + (valid ? i : T - 1);
        const float sigma = sigma_in[s];
        float c0 = 0, c1 = 0, c2 = 0;
        if (valid) { c0 = rgb_in[s * 3]; c1 = rgb_in[s * 3 + 1]; c2 = rgb_in[s * 3 + 2]; }
        fs_infer_tile<PACK>(a, g, n, i, lane, T, sigma, c0, c1, c2, noise, density_scale, thresh, rgb_masked, field4, nullptr);
    }
    fs_infer_finish<PACK>(a, n, lane, bg_ray, bg_scalar, image, depth, weights_sum);
}

// The same pass over sigma / rgb in the block-interleaved order (fs_block_row): a workgroup of 16 waves takes 16 rays (a quarter of a
// 64-ray block: 64 B of every sigma row and 192 B of every rgb row), stages 64 depths of them through LDS with 16-byte loads (the next
// tile's loads are in flight while this one is consumed) and each wave runs the per-ray pass above on one ray, reading its samples
// transposed out of LDS (odd row strides: no bank conflicts). Per-sample outputs leave ray-major; sigma_rm (may be NULL) = the
// densities in ray-major order for callers that return them.
#define FS_BLK_RAYS 16u
template <bool PACK>
__global__ void __launch_bounds__(1024) k_fs_render_infer_blk(const float *__restrict__ sigma_in, const float *__restrict__ rgb_in, const float *__restrict__ nears,
                                                              const float *__restrict__ fars, const float *__restrict__ noise, const float *__restrict__ bg_ray,
                                                              float bg_scalar, uint32_t N, uint32_t T, float density_scale, float thresh,
                                                              float *__restrict__ image, float *__restrict__ depth, float *__restrict__ weights_sum,
                                                              float *__restrict__ rgb_masked, float4 *__restrict__ field4, float *__restrict__ sigma_rm) {
    __shared__ float tile_s[64][FS_BLK_RAYS + 1];
    __shared__ float tile_c[64][FS_BLK_RAYS * 3 + 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n0 = blockIdx.x * FS_BLK_RAYS;                         // first ray of the workgroup; its rows start at fs_block_row(n0, i, T)
    const uint64_t row0 = fs_block_row(n0, 0, T);
    // loads of one tile: threads 0..255 -> (depth, 16-byte part) of the sigma rows, threads 256..1023 -> of the rgb rows
    const bool is_s = tid < 256;
    const uint32_t li = is_s ? tid : tid - 256;
    const uint32_t l_step = is_s ? li >> 2 : li / 12, l_part = is_s ? (li & 3) : li % 12;
    const float *src = is_s ? sigma_in + row0 + l_part * 4 : rgb_in + row0 * 3 + l_part * 4;
    const uint32_t src_ld = is_s ? FS_RAY_BLOCK : FS_RAY_BLOCK * 3;
    float *dst = is_s ? &tile_s[l_step][l_part * 4] : &tile_c[l_step][l_part * 4];
    float4 nx = make_float4(0, 0, 0, 0);
    if (l_step < T) nx = *reinterpret_cast<const float4 *>(src + (uint64_t)l_step * src_ld);
    const uint32_t n = n0 + wv;
    const bool own = n < N;                                               // wave-uniform
    const FsGeom g = fs_geom(nears, fars, own ? n : N - 1, T);
    FsRayAcc a = {1.0f, 0, 0, 0, 0, 0};
    for (uint32_t base = 0; base < T; base += 64) {
        dst[0] = nx.x; dst[1] = nx.y; dst[2] = nx.z; dst[3] = nx.w;
        __syncthreads();
        if (base + 64 + l_step < T) nx = *reinterpret_cast<const float4 *>(src + (uint64_t)(base + 64 + l_step) * src_ld);
        if (own)
            fs_infer_tile<PACK>(a, g, n, base + lane, lane, T, tile_s[lane][wv], tile_c[lane][wv * 3], tile_c[lane][wv * 3 + 1], tile_c[lane][wv * 3 + 2],
                                noise, density_scale, thresh, rgb_masked, field4, sigma_rm);
        __syncthreads();
    }
    if (own) fs_infer_finish<PACK>(a, n, lane, bg_ray, bg_scalar, image, depth, weights_sum);
}

// ================================================================= host entry points
// ---------------------------------------------------------------- tile order of a view's rays, found and built on the device
// focnerf_amd/rayorder.py: a caller hands a view over row by row; the staged render walks it in th x tw pixel tiles (the 64 rays of a block
// are then a compact patch at every level of the hash grid). Whether the ray list IS a row-major H x W pixel grid is read off the
// directions: inside a row consecutive steps point the same way, the step from a row's last pixel to the next row's first points back
// across the image — it is anti-parallel to the step before it and to the step after it. All of it here, no host round trip (the
// torch form — nonzero, int(), bool() — waited for the GPU three times per view, i.e. until the PREVIOUS view had drained, before the
// first chunk of the next one could be enqueued: 1-3 ms per 800 x 800 view).
//   pass 1: turns t_i = [ (d[i+2] - d[i+1]) . (d[i+1] - d[i]) < 0 ], i < N - 2: their number and the first one (W = first + 2)
//   pass 2: every t_i must be where an H x W grid puts it: (i + 2) % W == 0 or (i + 1) % W == 0
//   pass 3: perm[p] = the ray at position p of the tile order (closed form, ragged last tiles included), or p when it is no such grid
struct VtState { uint32_t count, first_enc, bad, pad; };      // first_enc = N - (index of the first turn), 0 = none (zero-initialised)

__device__ __forceinline__ bool vt_turn(const float *__restrict__ d, uint32_t i) {
    const float ax = d[(i + 1) * 3] - d[i * 3], ay = d[(i + 1) * 3 + 1] - d[i * 3 + 1], az = d[(i + 1) * 3 + 2] - d[i * 3 + 2];
    const float bx = d[(i + 2) * 3] - d[(i + 1) * 3], by = d[(i + 2) * 3 + 1] - d[(i + 1) * 3 + 1], bz = d[(i + 2) * 3 + 2] - d[(i + 1) * 3 + 2];
    return (bx * ax + by * ay) + bz * az < 0.0f;
}

__global__ void __launch_bounds__(256) k_vt_find(const float *__restrict__ d, uint32_t N, VtState *__restrict__ st) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i + 2 < N; i += gridDim.x * 256)
        if (vt_turn(d, i)) { atomicAdd(&st->count, 1u); atomicMax(&st->first_enc, N - i); }
}

__global__ void __launch_bounds__(256) k_vt_check(const float *__restrict__ d, uint32_t N, VtState *__restrict__ st) {
    const uint32_t fe = st->first_enc;
    if (fe == 0u) return;
    const uint32_t W = N - fe + 2u;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i + 2 < N; i += gridDim.x * 256) {
        const bool expect = (i + 2u) % W == 0u || (i + 1u) % W == 0u;
        if (vt_turn(d, i) != expect) st->bad = 1u;
    }
}

__global__ void __launch_bounds__(256) k_vt_perm(uint32_t N, uint32_t th, uint32_t tw, const VtState *__restrict__ st, int64_t *__restrict__ perm) {
    const uint32_t fe = st->first_enc, W = fe ? N - fe + 2u : 0u;
    const uint32_t H = W ? N / W : 0u;
    const bool grid = W >= 16u && N % W == 0u && H >= 8u && st->count == 2u * (H - 1u) && st->bad == 0u;
    for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < N; p += gridDim.x * 256) {
        uint32_t src = p;
        if (grid) {
            // tile rows hold W * th rays each (the last one fewer rows), a tile rows_here * tw rays (the last one of a tile row fewer columns)
            const uint64_t wt = (uint64_t)W * th;            // 64 bits: a very wide view times a tall tile passes 2^32
            const uint32_t ty = (uint32_t)(p / wt), rem = (uint32_t)(p - ty * wt);
            const uint32_t rows_here = min(th, H - ty * th);
            const uint32_t tx = min(rem / (rows_here * tw), (W - 1u) / tw), r2 = rem - tx * rows_here * tw;
            const uint32_t cols_here = min(tw, W - tx * tw);
            src = (ty * th + r2 / cols_here) * W + tx * tw + r2 % cols_here;
        }
        perm[p] = (int64_t)src;
    }
}

extern "C" {

int foc_fixed_sample(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *aabb, const float *noise,
                     uint32_t N, uint32_t T, float bound, float *xyzs, float *enc_in, void *ray_sh, uint32_t ray_block, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_o);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(rays_o && rays_d && nears && fars && aabb && (xyzs || enc_in), FOC_E_INVALID, "fixed_sample: null pointer");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "fixed_sample: T must be >= 2");
    FOC_REQUIRE(ray_block == 0 || ray_block == FS_RAY_BLOCK, FOC_E_INVALID, "fixed_sample: ray_block must be 0 (ray-major) or %u (got %u)", FS_RAY_BLOCK, ray_block);
    const uint64_t rows = ray_block ? (uint64_t)foc_div_up(N, FS_RAY_BLOCK) * FS_RAY_BLOCK * T : (uint64_t)N * T;
    hipLaunchKernelGGL(k_fs_sample, dim3(foc_grid_1d(rows, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, nears, fars, aabb,
                       noise, N, T, bound, xyzs, enc_in, (_Float16 *)ray_sh, ray_block);
    FOC_CHECK_LAUNCH("fixed_sample");
    return FOC_OK;
}

int foc_fixed_head_forward(const void *h, const float *rays_d, const float *nears, const float *fars, const float *noise, uint32_t N, uint32_t T,
                           float density_scale, float *sigma, float *trans, float *weights, float *weights_sum, float *depth, void *cin,
                           const void *obj_feat, uint32_t cin_width, void *stream) {
    FocDeviceGuard foc_guard_(stream, h);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(h && rays_d && nears && fars && sigma && trans && weights && weights_sum && depth, FOC_E_INVALID, "fixed_head_forward: null pointer");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "fixed_head_forward: T must be >= 2");
    FOC_REQUIRE(cin_width == 32 || cin_width == 48, FOC_E_INVALID, "fixed_head_forward: cin_width must be 32 or 48 (got %u)", cin_width);
    FOC_REQUIRE(!obj_feat || cin_width == 48, FOC_E_INVALID, "fixed_head_forward: an object feature needs the 48-wide colour input");
    hipLaunchKernelGGL(k_fs_head_fwd, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)h, rays_d, nears, fars, noise, N, T,
                       density_scale, sigma, trans, weights, weights_sum, depth, (_Float16 *)cin, (const _Float16 *)obj_feat, cin_width);
    FOC_CHECK_LAUNCH("fixed_head_forward");
    return FOC_OK;
}

int foc_fixed_head_backward(const void *h, const float *sigma, const float *trans, const float *nears, const float *fars, const float *noise,
                            const float *grad_w, const float *grad_ws, const float *grad_depth, const void *grad_cin, uint32_t N, uint32_t T,
                            float density_scale, void *grad_h, uint32_t cin_width, void *stream) {
    FocDeviceGuard foc_guard_(stream, h);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(h && sigma && trans && nears && fars && grad_h, FOC_E_INVALID, "fixed_head_backward: null pointer");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "fixed_head_backward: T must be >= 2");
    FOC_REQUIRE(cin_width == 32 || cin_width == 48, FOC_E_INVALID, "fixed_head_backward: cin_width must be 32 or 48 (got %u)", cin_width);
    hipLaunchKernelGGL(k_fs_head_bwd, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)h, sigma, trans, nears, fars, noise,
                       grad_w, grad_ws, grad_depth, (const _Float16 *)grad_cin, N, T, density_scale, (_Float16 *)grad_h, cin_width);
    FOC_CHECK_LAUNCH("fixed_head_backward");
    return FOC_OK;
}

int foc_fixed_composite_forward(const void *c, const float *weights, const float *bg_ray, float bg_scalar, uint32_t N, uint32_t T, float thresh,
                                float *image, void *stream) {
    FocDeviceGuard foc_guard_(stream, c);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(c && weights && image, FOC_E_INVALID, "fixed_composite_forward: null pointer");
    hipLaunchKernelGGL(k_fs_composite_fwd, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)c, weights, bg_ray, bg_scalar, N, T,
                       thresh, image);
    FOC_CHECK_LAUNCH("fixed_composite_forward");
    return FOC_OK;
}

int foc_fixed_composite_backward(const float *grad_image, const void *c, const float *weights, const float *bg_ray, float bg_scalar, uint32_t N,
                                 uint32_t T, float thresh, void *grad_c, float *grad_w, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad_image);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(grad_image && c && weights && grad_c && grad_w, FOC_E_INVALID, "fixed_composite_backward: null pointer");
    hipLaunchKernelGGL(k_fs_composite_bwd, dim3(foc_grid_1d((uint64_t)N * T, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, grad_image,
                       (const _Float16 *)c, weights, bg_ray, bg_scalar, N, T, thresh, (_Float16 *)grad_c, grad_w);
    FOC_CHECK_LAUNCH("fixed_composite_backward");
    return FOC_OK;
}

int foc_fixed_tail_forward(const void *h, const void *c, const float *nears, const float *fars, const float *noise, const float *bg_ray, float bg_scalar,
                           uint32_t N, uint32_t T, float density_scale, float thresh, float *sigma, float *trans, float *weights, float *weights_sum,
                           float *depth, float *image, uint32_t c_width, float *ray_sumsq, void *stream) {
    FocDeviceGuard foc_guard_(stream, h);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(c_width == 16 || c_width == 4, FOC_E_INVALID, "fixed_tail_forward: c_width must be 16 or 4 (got %u)", c_width);
    FOC_REQUIRE(h && c && nears && fars && sigma && trans && weights && weights_sum && depth && image, FOC_E_INVALID, "fixed_tail_forward: null pointer");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "fixed_tail_forward: T must be >= 2");
    auto kern = noise ? k_fs_tail_fwd<true> : k_fs_tail_fwd<false>;
    hipLaunchKernelGGL(kern, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)h, (const _Float16 *)c, nears, fars, noise,
                       bg_ray, bg_scalar, N, T, density_scale, thresh, sigma, trans, weights, weights_sum, depth, image, c_width, ray_sumsq);
    FOC_CHECK_LAUNCH("fixed_tail_forward");
    return FOC_OK;
}

int foc_fixed_tail_backward(const float *grad_image, const float *grad_ws, const float *grad_depth, const void *c, const float *sigma, const float *trans,
                            const float *weights, const float *nears, const float *fars, const float *noise, const float *bg_ray, float bg_scalar,
                            uint32_t N, uint32_t T, float density_scale, float thresh, void *grad_c, void *grad_h0, uint32_t c_width, const float *grad_sumsq,
                            void *stream) {
    FocDeviceGuard foc_guard_(stream, grad_image);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(c_width == 16 || c_width == 4, FOC_E_INVALID, "fixed_tail_backward: c_width must be 16 or 4 (got %u)", c_width);
    FOC_REQUIRE(grad_image && c && sigma && trans && weights && nears && fars && grad_c && grad_h0, FOC_E_INVALID, "fixed_tail_backward: null pointer");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "fixed_tail_backward: T must be >= 2");
    auto kern = noise ? k_fs_tail_bwd<true> : k_fs_tail_bwd<false>;
    hipLaunchKernelGGL(kern, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, grad_image, grad_ws, grad_depth, (const _Float16 *)c, sigma,
                       trans, weights, nears, fars, noise, bg_ray, bg_scalar, N, T, density_scale, thresh, (_Float16 *)grad_c, (_Float16 *)grad_h0, c_width,
                       grad_sumsq);
    FOC_CHECK_LAUNCH("fixed_tail_backward");
    return FOC_OK;
}

int foc_fixed_render_inference(const float *sigma, const float *rgb, const float *nears, const float *fars, const float *noise, const float *bg_ray,
                               float bg_scalar, uint32_t N, uint32_t T, float density_scale, float thresh, float *image, float *depth, float *weights_sum,
                               float *rgb_masked, uint32_t ray_block, float *sigma_raymajor, void *stream) {
    FocDeviceGuard foc_guard_(stream, sigma);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(sigma && rgb && nears && fars && image && depth && weights_sum, FOC_E_INVALID, "fixed_render_inference: null pointer");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "fixed_render_inference: T must be >= 2");
    FOC_REQUIRE(ray_block == 0 || ray_block == FS_RAY_BLOCK, FOC_E_INVALID, "fixed_render_inference: ray_block must be 0 or %u (got %u)", FS_RAY_BLOCK, ray_block);
    FOC_REQUIRE(ray_block || !sigma_raymajor, FOC_E_INVALID, "fixed_render_inference: sigma_raymajor is the un-blocked copy of a blocked sigma (ray_block == 0: sigma already is ray-major)");
    if (ray_block) {
        FOC_REQUIRE((((uintptr_t)sigma | (uintptr_t)rgb) & 15) == 0, FOC_E_INVALID, "fixed_render_inference: blocked sigma / rgb must be 16-byte aligned");
        hipLaunchKernelGGL(k_fs_render_infer_blk<false>, dim3(foc_div_up(N, FS_BLK_RAYS)), dim3(1024), 0, (hipStream_t)stream, sigma, rgb, nears, fars, noise, bg_ray, bg_scalar,
                           N, T, density_scale, thresh, image, depth, weights_sum, rgb_masked, (float4 *)nullptr, sigma_raymajor);
        FOC_CHECK_LAUNCH("fixed_render_inference");
        return FOC_OK;
    }
    hipLaunchKernelGGL(k_fs_render_infer<false>, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, sigma, rgb, nears, fars, noise, bg_ray, bg_scalar, N, T,
                       density_scale, thresh, image, depth, weights_sum, rgb_masked, (float4 *)nullptr);
    FOC_CHECK_LAUNCH("fixed_render_inference");
    return FOC_OK;
}

int foc_fixed_field_pack(const float *sigma, const float *rgb, const float *nears, const float *fars, const float *noise, const float *bg_ray,
                         float bg_scalar, uint32_t N, uint32_t T, float density_scale, float thresh, float *image, float *depth, float *weights_sum,
                         float *field4, uint32_t ray_block, void *stream) {
    FocDeviceGuard foc_guard_(stream, sigma);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(sigma && rgb && nears && fars && field4, FOC_E_INVALID, "fixed_field_pack: null pointer");
    FOC_REQUIRE((image && depth && weights_sum) || (!image && !depth && !weights_sum), FOC_E_INVALID,
                "fixed_field_pack: image, depth and weights_sum go together (all or none)");
    FOC_REQUIRE(((uintptr_t)field4 & 15) == 0, FOC_E_INVALID, "fixed_field_pack: field4 must be 16-byte aligned");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "fixed_field_pack: T must be >= 2");
    FOC_REQUIRE(ray_block == 0 || ray_block == FS_RAY_BLOCK, FOC_E_INVALID, "fixed_field_pack: ray_block must be 0 or %u (got %u)", FS_RAY_BLOCK, ray_block);
    if (ray_block) {
        FOC_REQUIRE((((uintptr_t)sigma | (uintptr_t)rgb) & 15) == 0, FOC_E_INVALID, "fixed_field_pack: blocked sigma / rgb must be 16-byte aligned");
        hipLaunchKernelGGL(k_fs_render_infer_blk<true>, dim3(foc_div_up(N, FS_BLK_RAYS)), dim3(1024), 0, (hipStream_t)stream, sigma, rgb, nears, fars, noise, bg_ray, bg_scalar,
                           N, T, density_scale, thresh, image, depth, weights_sum, (float *)nullptr, (float4 *)field4, (float *)nullptr);
        FOC_CHECK_LAUNCH("fixed_field_pack");
        return FOC_OK;
    }
    hipLaunchKernelGGL(k_fs_render_infer<true>, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, sigma, rgb, nears, fars, noise, bg_ray, bg_scalar, N, T,
                       density_scale, thresh, image, depth, weights_sum, (float *)nullptr, (float4 *)field4);
    FOC_CHECK_LAUNCH("fixed_field_pack");
    return FOC_OK;
}

int foc_view_tile_order(const float *rays_d, uint32_t N, uint32_t tile_h, uint32_t tile_w, int64_t *perm, void *state16, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_d);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(rays_d && perm && state16, FOC_E_INVALID, "view_tile_order: null pointer");
    FOC_REQUIRE(tile_h >= 1 && tile_w >= 1 && tile_h * tile_w <= 4096, FOC_E_INVALID, "view_tile_order: tile %u x %u", tile_h, tile_w);
    FOC_REQUIRE(N < (1u << 31), FOC_E_INVALID, "view_tile_order: N too large");
    hipStream_t st = (hipStream_t)stream;
    VtState *state = reinterpret_cast<VtState *>(state16);
    if (foc_zero_async(state, sizeof(VtState), st) != hipSuccess) { foc_set_error("view_tile_order: zero fill failed"); return FOC_E_LAUNCH; }
    if (N >= 3) {
        hipLaunchKernelGGL(k_vt_find, dim3(foc_grid_1d(N, 256)), dim3(256), 0, st, rays_d, N, state);
        hipLaunchKernelGGL(k_vt_check, dim3(foc_grid_1d(N, 256)), dim3(256), 0, st, rays_d, N, state);
    }
    hipLaunchKernelGGL(k_vt_perm, dim3(foc_grid_1d(N, 256)), dim3(256), 0, st, N, tile_h, tile_w, state, perm);
    FOC_CHECK_LAUNCH("view_tile_order");
    return FOC_OK;
}

} // extern "C"
