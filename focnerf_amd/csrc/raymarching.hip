// raymarching.hip — gfx950 kernels behind the _raymarching entry points of include/focnerf.h.
//
// Semantics follow raymarching/src/raymarching.cu of the reference (line cites per kernel);
// the kernels themselves are organised for CDNA4:
//   * marching is split into count -> ordered scan -> write, so slot reservation is a
//     deterministic prefix sum in ray order instead of the reference's atomicAdd race
//     (raymarching.cu:405-406);
//   * train-time compositing runs ONE WAVE PER RAY: 64 consecutive samples are loaded
//     coalesced, transmittance is a wave product-scan (DPP shuffles), early termination
//     is a ballot, the per-ray sums are wave reductions — instead of the reference's one
//     thread walking its own ray segment (uncoalesced, 128 rays per block).
// Floating-point policy: compiled with -ffp-contract=off; fused multiply-adds are written
// as explicit fmaf() exactly where oracle/oracle.c has them, so the marching control flow
// (and therefore every sample index and position) is bit-identical to the oracle.
#include "common.h"
#include <float.h>

#define RM_SQRT3 1.7320508075688772f
#define RM_RPI   0.3183098861837907f

__device__ __forceinline__ float rm_sign(float x) { return copysignf(1.0f, x); }
__device__ __forceinline__ float rm_clamp(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

__device__ __forceinline__ uint32_t rm_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t rm_morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return rm_expand_bits(x) | (rm_expand_bits(y) << 1) | (rm_expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t rm_morton3D_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// frexpf exponent of a non-negative finite float without the libcall: for x = 0 frexpf
// returns exponent 0; subnormals never reach a positive exponent, and only max(0, e) is used.
__device__ __forceinline__ int rm_frexp_exp(float x) {
    int e;
    (void)frexpf(x, &e);
    return e;
}

// ---------------------------------------------------------------- R1 (raymarching.cu:92-145)
// slab test of one ray against the box a[0..5]; a miss gives FLT_MAX twice
__device__ __forceinline__ void rm_near_far(float ox, float oy, float oz, float dx, float dy, float dz, float a0, float a1, float a2, float a3, float a4, float a5,
                                            float min_near, float &near_out, float &far_out) {
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near = (a0 - ox) * rdx, far = (a3 - ox) * rdx, t;
    if (near > far) { t = near; near = far; far = t; }
    float near_y = (a1 - oy) * rdy, far_y = (a4 - oy) * rdy;
    if (near_y > far_y) { t = near_y; near_y = far_y; far_y = t; }
    bool miss = (near > far_y || near_y > far);
    if (!miss) {
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (a2 - oz) * rdz, far_z = (a5 - oz) * rdz;
        if (near_z > far_z) { t = near_z; near_z = far_z; far_z = t; }
        miss = (near > far_z || near_z > far);
        if (!miss) {
            if (near_z > near) near = near_z;
            if (far_z < far) far = far_z;
            if (near < min_near) near = min_near;
        }
    }
    near_out = miss ? FLT_MAX : near;
    far_out = miss ? FLT_MAX : far;
}

__global__ void __launch_bounds__(256) k_near_far_from_aabb(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                     const float *__restrict__ aabb, uint32_t N, float min_near,
                                     float *__restrict__ nears, float *__restrict__ fars) {
    const float a0 = aabb[0], a1 = aabb[1], a2 = aabb[2], a3 = aabb[3], a4 = aabb[4], a5 = aabb[5];
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        float near, far;
        rm_near_far(rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2], rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2], a0, a1, a2, a3, a4, a5, min_near, near, far);
        nears[n] = near;
        fars[n] = far;
    }
}

// ---------------------------------------------------------------- R2 (raymarching.cu:163-198)
__global__ void __launch_bounds__(256) k_sph_from_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float radius,
                               uint32_t N, float *__restrict__ coords) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
        const float C = fmaf(-radius, radius, fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
        const float t = (-B + sqrtf(fmaf(B, B, -(A * C)))) / A;
        const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
        const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[n * 2] = fmaf(2 * theta, RM_RPI, -1.0f);
        coords[n * 2 + 1] = phi * RM_RPI;
    }
}

// ---------------------------------------------------------------- R3/R4 (raymarching.cu:214-254)
__global__ void __launch_bounds__(256) k_morton3D(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x)
        indices[n] = (int32_t)rm_morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void __launch_bounds__(256) k_morton3D_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const int32_t ind = indices[n];
        coords[n * 3] = (int32_t)rm_morton3D_invert((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)rm_morton3D_invert((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)rm_morton3D_invert((uint32_t)(ind >> 2));
    }
}

// ---------------------------------------------------------------- R5 (raymarching.cu:267-289)
// One thread packs 32 cells (8 x float4 loads, 128 B contiguous per lane) into one dword, so a
// wave streams 8 KiB per iteration and stores 256 contiguous bytes.
__global__ void __launch_bounds__(256) k_packbits_x4(const float4 *__restrict__ grid4, uint32_t N4, float thresh,
                              uint32_t *__restrict__ bitfield4) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N4; n += gridDim.x * blockDim.x) {
        uint32_t bits = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float4 v = grid4[(uint64_t)n * 8 + q];
            bits |= (v.x > thresh ? 1u : 0u) << (q * 4 + 0);
            bits |= (v.y > thresh ? 1u : 0u) << (q * 4 + 1);
            bits |= (v.z > thresh ? 1u : 0u) << (q * 4 + 2);
            bits |= (v.w > thresh ? 1u : 0u) << (q * 4 + 3);
        }
        bitfield4[n] = bits;
    }
}
__global__ void __launch_bounds__(256) k_packbits_tail(const float *__restrict__ grid, uint32_t n0, uint32_t N, float thresh,
                                uint8_t *__restrict__ bitfield) {
    for (uint32_t n = n0 + blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) bits |= (grid[(uint64_t)n * 8 + i] > thresh ? 1u : 0u) << i;
        bitfield[n] = (uint8_t)bits;
    }
}

// ---------------------------------------------------------------- marching core
// One cell lookup (raymarching.cu:361-379 == :429-448 == :752-770).
struct RmCell { float x, y, z, dt, mip_bound; int nx, ny, nz; };

struct RmParams {
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Cf, Hm1;
    uint32_t H;
    float norm_inv;                    // 0: sample positions leave as they are; else they leave as (x + bound) * norm_inv, the encoder's [0,1] coordinates
    int rederive;                      // != 0: after every emitted sample t continues from last_t + (t - last_t), the value composite_rays hands to the NEXT call
                                       // (raymarching.cu:871, 899) — a burst of k samples then marches what k calls of one sample would (see foc_march_rays_two_phase)
    int gamma_pow2;                    // dt_gamma is a power of two (1/128, the reference's default for its scenes): t * dt_gamma is exact, so t + t * dt_gamma
                                       // rounds once — it IS fmaf(t, dt_gamma, t)
};
// position of an emitted sample as it is stored (the native render step asks for the normalised form: csrc/occrender.hip)
__device__ __forceinline__ float rm_out(const RmParams &p, float x) { return p.norm_inv != 0.0f ? (x + p.bound) * p.norm_inv : x; }

__device__ __forceinline__ bool rm_cell(const uint8_t *__restrict__ grid, const RmParams &p, float ox, float oy, float oz,
                                        float dx, float dy, float dz, float t, RmCell &c) {
    c.x = rm_clamp(fmaf(t, dx, ox), -p.bound, p.bound);
    c.y = rm_clamp(fmaf(t, dy, oy), -p.bound, p.bound);
    c.z = rm_clamp(fmaf(t, dz, oz), -p.bound, p.bound);
    c.dt = rm_clamp(t * p.dt_gamma, p.dt_min, p.dt_max);
    // mip_from_pos / mip_from_dt (:42-54): float min/max, then truncation
    const float mx = fmaxf(fabsf(c.x), fmaxf(fabsf(c.y), fabsf(c.z)));
    const int l1 = (int)fminf(p.Cf - 1, fmaxf(0.0f, (float)rm_frexp_exp(mx)));
    const float mdt = (float)((double)(c.dt * p.Hf) * 0.5);
    const int l2 = (int)fminf(p.Cf - 1, fmaxf(0.0f, (float)rm_frexp_exp(mdt)));
    const int level = l1 > l2 ? l1 : l2;
    c.mip_bound = fminf(scalbnf(1.0f, level), p.bound);
    const float mip_rbound = 1 / c.mip_bound;
    // 0.5 * (x * mip_rbound + 1) * H in double, narrowed to float by clamp()'s parameter (:374-376)
    c.nx = (int)rm_clamp((float)(0.5 * (double)fmaf(c.x, mip_rbound, 1.0f) * (double)p.H), 0.0f, p.Hm1);
    c.ny = (int)rm_clamp((float)(0.5 * (double)fmaf(c.y, mip_rbound, 1.0f) * (double)p.H), 0.0f, p.Hm1);
    c.nz = (int)rm_clamp((float)(0.5 * (double)fmaf(c.z, mip_rbound, 1.0f) * (double)p.H), 0.0f, p.Hm1);
    // level * H3 + morton in float (:339,:378)
    const uint32_t index = (uint32_t)fmaf((float)level, p.H3, (float)rm_morton3D((uint32_t)c.nx, (uint32_t)c.ny, (uint32_t)c.nz));
    return (grid[index >> 3] & (1u << (index & 7u))) != 0;
}

// Empty cell: jump to the voxel exit (:389-398).
__device__ __forceinline__ float rm_skip_target(const RmParams &p, const RmCell &c, float t, float dx, float dy, float dz,
                                                float rdx, float rdy, float rdz) {
    const float tx = fmaf(fmaf(fmaf(0.5f, rm_sign(dx), (float)c.nx + 0.5f) * p.rH, 2.0f, -1.0f), c.mip_bound, -c.x) * rdx;
    const float ty = fmaf(fmaf(fmaf(0.5f, rm_sign(dy), (float)c.ny + 0.5f) * p.rH, 2.0f, -1.0f), c.mip_bound, -c.y) * rdy;
    const float tz = fmaf(fmaf(fmaf(0.5f, rm_sign(dz), (float)c.nz + 0.5f) * p.rH, 2.0f, -1.0f), c.mip_bound, -c.z) * rdz;
    return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
}

__device__ __forceinline__ float rm_skip(const RmParams &p, const RmCell &c, float t, float dx, float dy, float dz,
                                         float rdx, float rdy, float rdz) {
    const float tt = rm_skip_target(p, c, t, dx, dy, dz, rdx, rdy, rdz);
    do { t += rm_clamp(t * p.dt_gamma, p.dt_min, p.dt_max); } while (t < tt);
    return t;
}

static RmParams rm_make_params(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    RmParams p;
    p.bound = bound; p.dt_gamma = dt_gamma;
    // same float expressions as raymarching.cu:345-346 (IEEE division on host == device)
    p.dt_min = 2 * RM_SQRT3 / (float)max_steps;
    p.dt_max = 2 * RM_SQRT3 * (float)(1 << (C - 1)) / (float)H;
    p.rH = 1 / (float)H;
    p.H3 = (float)(H * H * H);
    p.Hf = (float)H; p.Cf = (float)C; p.Hm1 = (float)(H - 1);
    p.H = H;
    p.norm_inv = 0.0f;
    p.rederive = 0;
    int e = 0;
    p.gamma_pow2 = (dt_gamma > 0.0f && frexpf(dt_gamma, &e) == 0.5f) ? 1 : 0;
    return p;
}

// ---------------------------------------------------------------- R6 pass 1: count (raymarching.cu:348-400)
// aabb != NULL (both count kernels): nears / fars are OUTPUTS — the ray's slab test (k_near_far_from_aabb's expressions) is done here
__global__ void __launch_bounds__(64) k_march_count(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                              const uint8_t *__restrict__ grid, RmParams p, uint32_t max_steps, uint32_t N,
                              float *__restrict__ nears, float *__restrict__ fars,
                              const float *__restrict__ noises, int32_t *__restrict__ counts, float *__restrict__ tstrip,
                              const float *__restrict__ aabb, float min_near) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near_n, far;
    if (aabb) {
        rm_near_far(ox, oy, oz, dx, dy, dz, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5], min_near, near_n, far);
        nears[n] = near_n; fars[n] = far;
    } else { near_n = nears[n]; far = fars[n]; }
    float t = near_n;
    t = fmaf(rm_clamp(t * p.dt_gamma, p.dt_min, p.dt_max), noises[n], t);
    uint32_t num_steps = 0;
    RmCell c;
    // The marching loop is a chain of dependent L2 lookups: running it a second time to write the samples (as the reference does,
    // raymarching.cu:415-479) doubles that latency-bound cost. The `t` of every emitted sample is kept in a per-ray strip instead
    // (max_steps floats per ray), from which k_march_emit rebuilds position, dt and the two deltas with the loop's own expressions.
    float *strip = tstrip + (uint64_t)n * max_steps;
    while (t < far && num_steps < max_steps) {
        if (rm_cell(grid, p, ox, oy, oz, dx, dy, dz, t, c)) { strip[num_steps] = t; num_steps++; t += c.dt; }
        else t = rm_skip(p, c, t, dx, dy, dz, rdx, rdy, rdz);
    }
    counts[n] = (int32_t)num_steps;
}

// ---------------------------------------------------------------- R6 pass 1, one WAVE per ray
// The loop above is a chain of dependent cell lookups, one ray per lane: with 4096 rays that is 64 waves, each as slow as its longest
// ray (0.135 ms per training batch). But the values `t` can take do not depend on the occupancy grid at all: both branches advance it
// by t += clamp(t * dt_gamma, dt_min, dt_max) (:385 and :397), so every ray walks a fixed lattice t_0, t_1, ... and the grid only
// decides which lattice points are visited and which of those are emitted. Here a wave takes 64 consecutive lattice points of ONE ray
// (lane j = point j, generated with the loop's own expression), looks all 64 cells up at once, and then replays the loop's control
// flow on wave-uniform bit masks: an occupied visited point emits and moves to the next lane, an empty one moves to the first lane
// whose t is not below its voxel exit (carried into the next 64 points when there is none). Same visits, same emitted t, bit for bit.
#define RM_MAX_ROUNDS (1u << 22)                           // x 64 lattice points: far beyond any real ray; makes the loop finite whatever the inputs
template <bool MED3>                                       // dt_min <= dt_max (any real setting): the clamp of the recurrence is one v_med3_f32
__global__ void __launch_bounds__(256) k_march_count_wave(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                              const uint8_t *__restrict__ grid, RmParams p, uint32_t max_steps, uint32_t N,
                              float *__restrict__ nears, float *__restrict__ fars,
                              const float *__restrict__ noises, int32_t *__restrict__ counts, float *__restrict__ tstrip,
                              const float *__restrict__ aabb, float min_near, const int32_t *__restrict__ counter_in) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (n >= N) return;                                    // whole wave
    // the emit pass does the slot reservation itself (k_march_emit<.., true>): it takes the counter's entry values from a snapshot behind the
    // counts, because its last workgroup overwrites the counter while others may not have started
    if (counter_in && n == 0 && lane == 0) { counts[N] = counter_in[0]; counts[N + 1] = counter_in[1]; }
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near_n, far;
    if (aabb) {                                            // wave-uniform
        rm_near_far(ox, oy, oz, dx, dy, dz, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5], min_near, near_n, far);
        if (lane == 0) { nears[n] = near_n; fars[n] = far; }
    } else { near_n = nears[n]; far = fars[n]; }
    float t_cur = near_n;
    t_cur = fmaf(rm_clamp(t_cur * p.dt_gamma, p.dt_min, p.dt_max), noises[n], t_cur);
    float *strip = tstrip + (uint64_t)n * max_steps;
    uint32_t num_steps = 0;
    bool skipping = false;                                 // inside the do-while of an empty cell whose exit `skip_to` lies beyond the last 64 points
    float skip_to = 0.0f;
    for (uint32_t round = 0; round < RM_MAX_ROUNDS; round++) {
        // lane j: t_cur advanced j times. Every lane runs the same recurrence; each new value enters at lane 63 while the earlier
        // ones move down one lane (DPP wave_shl:1), so the 64th insertion leaves value j in lane j and costs one move per step.
        float T = t_cur, t_next = t_cur;
        // The step clamp(t dt_gamma, dt_min, dt_max) has three regimes and t only grows: a ray walks dt_min steps near the camera, then
        // t dt_gamma, then dt_max. A round that lies in ONE regime needs no clamp — its 64 steps are `t += dt_min`, `t += t dt_gamma`
        // (one fma when dt_gamma is a power of two) or `t += dt_max`, the same values bit for bit — which halves the instructions of the
        // recurrence (move + 1 instead of move + mul + med3 + add; the recurrence is half of this kernel's VALU work). The regime is
        // read off the round's first point and confirmed on its last one (lane 63 after the loop); a round that crosses a boundary —
        // at most two per ray — is generated again with the general step.
        bool fast = false;
        if constexpr (MED3) {
            const float raw0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t_cur * p.dt_gamma)));
            const int regime = raw0 <= p.dt_min ? 0 : (raw0 >= p.dt_max ? 2 : 1);
            if (regime == 0 || regime == 2) {
                const float step = regime == 0 ? p.dt_min : p.dt_max;
#pragma unroll 8
                for (uint32_t j = 0; j < 64; j++) {
                    T = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t_next), __builtin_bit_cast(int, T), 0x130, 0xf, 0xf, false));
                    t_next += step;
                }
            } else if (p.gamma_pow2) {
#pragma unroll 8
                for (uint32_t j = 0; j < 64; j++) {
                    T = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t_next), __builtin_bit_cast(int, T), 0x130, 0xf, 0xf, false));
                    t_next = fmaf(t_next, p.dt_gamma, t_next);
                }
            } else {
#pragma unroll 8
                for (uint32_t j = 0; j < 64; j++) {
                    T = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t_next), __builtin_bit_cast(int, T), 0x130, 0xf, 0xf, false));
                    t_next += t_next * p.dt_gamma;
                }
            }
            // the last point the general step would have clamped: lattice point 63 (in lane 63 now)
            const float raw63 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, T), 63)) * p.dt_gamma;
            fast = regime == 0 ? raw63 <= p.dt_min : (regime == 2 ? true : raw63 <= p.dt_max);
        }
        if (!fast) {
            T = t_cur; t_next = t_cur;
#pragma unroll 8
            for (uint32_t j = 0; j < 64; j++) {
                T = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t_next), __builtin_bit_cast(int, T), 0x130, 0xf, 0xf, false));
                const float raw = t_next * p.dt_gamma;             // for ordered operands the median IS fminf(hi, fmaxf(lo, x)), bit for bit
                t_next += MED3 ? __builtin_amdgcn_fmed3f(raw, p.dt_min, p.dt_max) : rm_clamp(raw, p.dt_min, p.dt_max);
            }
        }
        const uint64_t in_range = __ballot(T < far);
        uint32_t k = 0;                                    // lane of the lattice point the loop is at
        if (skipping) {
            const uint64_t landed = __ballot(!(T < skip_to));
            if (landed == 0ull) {
                if (!((in_range >> 63) & 1ull)) break;     // t only grows: wherever the skip lands, it is beyond `far`
                t_cur = t_next;
                continue;
            }
            k = (uint32_t)__builtin_ctzll(landed);
            skipping = false;
        }
        RmCell c;
        const bool occupied_here = rm_cell(grid, p, ox, oy, oz, dx, dy, dz, T, c);
        const float exit_here = rm_skip_target(p, c, T, dx, dy, dz, rdx, rdy, rdz);
        const uint64_t emit_ok = __ballot(occupied_here) & in_range;
        uint64_t emitted = 0ull;
        uint32_t room = (uint32_t)__builtin_amdgcn_readfirstlane((int)(max_steps - num_steps));
        // The replay is scalar code — one iteration per visited point, ~50 of them per round in empty space — and the CU has ONE scalar unit
        // for its sixteen waves: the kernel was bound by it (SQ counters: 3970 scalar against 3040 vector instructions per wave, the
        // compiler's lowering of the loop with its `done` / `skipping` flags ran to ~40 scalar instructions per iteration). Hence the
        // spare form: in_range is a PREFIX mask (t only grows), so "the point is out of range" is k >= k_far; one exit code instead of
        // flags; the mask of the lanes above k as one shift; and the walk over empty points as a hand-written block.
        const uint32_t k_far = ~in_range ? (uint32_t)__builtin_ctzll(~in_range) : 64u;       // first lattice point at or beyond `far`
        int state = room == 0u ? 1 : 0;                    // 0: ran off the 64 points; 1: the loop of :359 ended (far / step cap); 2: a skip carries over
        uint32_t ks = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        // (wave-uniform by construction; said again so that the block's scalar operands are scalar registers)
        const uint32_t k_far_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)k_far);
        const uint64_t emit_s = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(emit_ok >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)emit_ok);
        while (state == 0) {
            // Walk over EMPTY points until an occupied one, the end of the range or a skip that leaves the 64 points. Hand-written: nine
            // instructions per visited point (seven of them scalar) where the compiler's lowering of the same loop took about thirty.
            //   why 0: ks >= k_far      why 1: the point at ks is occupied      why 2: no point of this round lies at or beyond the voxel exit
            uint32_t why, tt_bits;
            uint64_t tmp;
            asm volatile("1:\n\t"
                         "s_cmp_ge_u32 %[k], %[kfar]\n\t"
                         "s_cbranch_scc1 2f\n\t"
                         "s_bitcmp1_b64 %[emit], %[k]\n\t"
                         "s_cbranch_scc1 3f\n\t"
                         "s_nop 3\n\t"                                      // SALU wrote %[k]: four wait states before it selects a lane
                         "v_readlane_b32 %[tt], %[exitv], %[k]\n\t"         // the voxel exit of the point at ks
                         "s_lshl_b64 %[tmp], -2, %[k]\n\t"                  // the lanes above ks: do { advance } while (t < tt) advances at least once
                         "v_cmp_ngt_f32_e32 vcc, %[tt], %[T]\n\t"           // !(tt > T) == !(T < tt)
                         "s_and_b64 %[tmp], vcc, %[tmp]\n\t"
                         "s_cbranch_scc0 4f\n\t"
                         "s_ff1_i32_b64 %[k], %[tmp]\n\t"
                         "s_branch 1b\n\t"
                         "2: s_mov_b32 %[why], 0\n\t"
                         "s_branch 5f\n\t"
                         "3: s_mov_b32 %[why], 1\n\t"
                         "s_branch 5f\n\t"
                         "4: s_mov_b32 %[why], 2\n\t"
                         "5:"
                         : [k] "+s"(ks), [tt] "=&s"(tt_bits), [tmp] "=&s"(tmp), [why] "=&s"(why)
                         : [kfar] "s"(k_far_s), [emit] "s"(emit_s), [exitv] "v"(exit_here), [T] "v"(T)
                         : "vcc", "scc");
            if (why == 0u) { state = ks >= 64u ? 0 : 1; break; }
            if (why == 2u) { state = 2; skip_to = __builtin_bit_cast(float, tt_bits); break; }
            // a run of occupied points: each emits and steps to its successor
            const uint64_t rest = ~(emit_s >> ks);         // bit 0 clear; for ks > 0 the bits shifted in end the run at lane 63
            uint32_t len = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
            len = (uint32_t)__builtin_amdgcn_readfirstlane((int)(len < room ? len : room));      // (uniform; a scalar register for the block above)
            emitted |= (len >= 64u ? ~0ull : ((1ull << len) - 1ull)) << ks;
            room -= len;
            ks += len;
            if (room == 0u) state = 1;                     // what the loop condition finds at its next test (:359)
        }
        skipping = state == 2;
        const bool done = state == 1;
        if ((emitted >> lane) & 1ull) strip[num_steps + (uint32_t)__builtin_popcountll(emitted & ((1ull << lane) - 1ull))] = T;
        num_steps += (uint32_t)__builtin_popcountll(emitted);
        if (done) break;
        t_cur = t_next;
    }
    if (lane == 0) counts[n] = (int32_t)num_steps;
}

// ---------------------------------------------------------------- R6 pass 2: ordered slot reservation
// One 1024-thread workgroup scans the N counts in ray order (each thread owns a contiguous
// chunk), writes rays[n] = (n, base + exclusive_prefix, count) and bumps the two counters —
// the deterministic stand-in for the atomicAdd pair of raymarching.cu:405-413.
__global__ void __launch_bounds__(1024) k_march_scan(const int32_t *__restrict__ counts, uint32_t N,
                                                     int32_t *__restrict__ rays, int32_t *__restrict__ counter) {
    __shared__ int s_wave[16];
    __shared__ int s_total;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (N + 1023) / 1024;
    const uint32_t lo = tid * per, hi = min(N, lo + per);
    int local = 0;
    for (uint32_t i = lo; i < hi; i++) local += counts[i];
    const int incl = wave_incl_sum_i(local, (int)lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int v = lane < 16 ? s_wave[lane] : 0;
        const int vi = wave_incl_sum_i(v, (int)lane);
        if (lane < 16) s_wave[lane] = vi - v;   // exclusive base per wave
        if (lane == 15) s_total = vi;
    }
    __syncthreads();
    const int base0 = counter[0];
    const int ray0 = counter[1];
    int run = base0 + s_wave[wave] + (incl - local);
    for (uint32_t i = lo; i < hi; i++) {
        const int c = counts[i];
        const uint32_t r = i;   // rows in ray order; a non-zero counter[1] would index past rays[N,3] in the reference
        rays[r * 3] = (int32_t)i; rays[r * 3 + 1] = run; rays[r * 3 + 2] = c;
        run += c;
    }
    __syncthreads();
    if (tid == 0) { counter[0] = base0 + s_total; counter[1] = ray0 + (int32_t)N; }
}

// degree-4 real spherical harmonics of a direction, the expressions of head.hip hd_sh16 (focnerf_amd/shencoder.py)
__device__ __forceinline__ void rm_sh16(float x, float y, float z, float (&o)[16]) {
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

// ---------------------------------------------------------------- R6 pass 3: emit (raymarching.cu:415-479)
// One wave per ray, lane k = sample k: t_k from the strip; xyz = clamp(o + t d), dt = clamp(t dt_gamma) exactly as rm_cell forms them;
// deltas = (dt_k, (t_k + dt_k) - last_t) with last_t = t_{k-1} + dt_{k-1} (the start t for k = 0), as the loop accumulates them.
// Stores are contiguous across the wave (768 B of xyz per 64 samples).
// FIELD: the sample list in the layout the fused training path consumes (foc_march_rays_train_field) — `xyzs` receives the encoder's
// [0,1] coordinates (x + bound) * 1 / (2 bound) (rm_out), `dirs` is not written, and `sh` [M,16] fp16 receives each sample's degree-4
// SH row (the first k-chunk of the colour network's input, head.hip hd_sh16 rounded like k_head_fwd rounds it): one row per sample of
// a ray, all equal. Every row of both arrays is written — rays that do not fit the list and the rows behind the last ray get zeros
// (spare workgroups; `counter` from k_march_scan) — so the caller needs no zero fill.
#define RM_PAD_BLOCKS 64u
// SCAN (round 5): the ordered slot reservation of k_march_scan done HERE, by every workgroup for itself — a workgroup's rays start at the sum
// of the counts of all rays before them, 256 threads add those up (at most 16 384 ints, L2 resident) and the wave of ray n adds the counts of
// the workgroup's earlier rays: the same integers as the one-workgroup scan, with no launch of its own (8.6 us + a launch gap of the configs[2]
// step), no atomics and no ticket. The `rays` table is written here (lane 0 of each ray's wave), the counter by the workgroup of the last ray;
// the entry values of the counter come from the snapshot the count pass left behind the counts (`counts[N], counts[N + 1]`).
template <bool FIELD, bool SCAN>
__global__ void __launch_bounds__(256) k_march_emit(const float *__restrict__ rays_o, const float *__restrict__ rays_d, RmParams p, uint32_t max_steps,
                                                    uint32_t N, uint32_t M, const float *__restrict__ nears, const float *__restrict__ noises,
                                                    int32_t *__restrict__ rays, const float *__restrict__ tstrip,
                                                    float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                    _Float16 *__restrict__ sh, int32_t *__restrict__ counter, uint32_t pad_align,
                                                    const int32_t *__restrict__ counts) {
    typedef _Float16 rm_h8 __attribute__((ext_vector_type(8)));
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t scan_point = 0, scan_steps = 0, scan_total = 0;
    if constexpr (SCAN) {
        __shared__ int s_part[4];
        const uint32_t first = min(blockIdx.x * 4u, N);            // spare workgroups (FIELD): all N rays lie before them
        int part = 0;
        for (uint32_t i = threadIdx.x; i < first; i += 256u) part += counts[i];
        part = wave_sum_i(part);
        if (lane == 0) s_part[threadIdx.x >> 6] = part;
        __syncthreads();
        const int before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        const int base0 = counts[N], ray0 = counts[N + 1];
        const uint32_t w = threadIdx.x >> 6;
        int mine = 0, earlier = 0, here = 0;
        for (uint32_t j = 0; j < 4u; j++) {
            const int cj = first + j < N ? counts[first + j] : 0;
            if (j < w) earlier += cj;
            if (j == w) mine = cj;
            here += cj;
        }
        scan_point = (uint32_t)(base0 + before + earlier);
        scan_steps = (uint32_t)mine;
        scan_total = (uint32_t)(base0 + before + (blockIdx.x * 4u < N ? here : 0));
        if (n < N && lane == 0) { rays[n * 3] = (int32_t)n; rays[n * 3 + 1] = (int32_t)scan_point; rays[n * 3 + 2] = mine; }
        if (blockIdx.x == (N + 3u) / 4u - 1u && threadIdx.x == 0) { counter[0] = (int32_t)scan_total; counter[1] = ray0 + (int32_t)N; }
    }
    if constexpr (FIELD) {
        if (blockIdx.x >= (N + 3u) / 4u) {
            // pad_align > 0: the caller cuts the list to the samples marched, rounded up like raymarching.py:226 — nothing behind that is read
            const uint32_t total = SCAN ? scan_total : (uint32_t)counter[0], pb = blockIdx.x - (N + 3u) / 4u;
            const uint64_t end = pad_align ? min((uint64_t)M, (uint64_t)total + (pad_align - total % pad_align)) : (uint64_t)M;
            for (uint64_t s = (uint64_t)total + pb * 256u + threadIdx.x; s < end; s += (uint64_t)RM_PAD_BLOCKS * 256u) {
                xyzs[s * 3] = 0.0f; xyzs[s * 3 + 1] = 0.0f; xyzs[s * 3 + 2] = 0.0f;
                deltas[s * 2] = 0.0f; deltas[s * 2 + 1] = 0.0f;
                *reinterpret_cast<uint4 *>(sh + s * 16) = make_uint4(0u, 0u, 0u, 0u); *reinterpret_cast<uint4 *>(sh + s * 16 + 8) = make_uint4(0u, 0u, 0u, 0u);
            }
            return;
        }
    }
    if (n >= N) return;
    const uint32_t point_index = SCAN ? scan_point : (uint32_t)rays[n * 3 + 1];
    const uint32_t num_steps = SCAN ? scan_steps : (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0) return;
    if (point_index + num_steps > M) {                             // raymarching.cu:413
        if constexpr (FIELD) {
            const uint64_t end = min((uint64_t)point_index + num_steps, (uint64_t)M);
            for (uint64_t s = (uint64_t)point_index + lane; s < end; s += 64) {
                xyzs[s * 3] = 0.0f; xyzs[s * 3 + 1] = 0.0f; xyzs[s * 3 + 2] = 0.0f;
                deltas[s * 2] = 0.0f; deltas[s * 2 + 1] = 0.0f;
                *reinterpret_cast<uint4 *>(sh + s * 16) = make_uint4(0u, 0u, 0u, 0u); *reinterpret_cast<uint4 *>(sh + s * 16 + 8) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        return;
    }
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    float t0 = nears[n];
    t0 = fmaf(rm_clamp(t0 * p.dt_gamma, p.dt_min, p.dt_max), noises[n], t0);
    const float *strip = tstrip + (uint64_t)n * max_steps;
    rm_h8 sh0, sh1;
    if constexpr (FIELD) {
        float o[16];
        rm_sh16(dx, dy, dz, o);
#pragma unroll
        for (int k = 0; k < 8; k++) { sh0[k] = foc_f2h(o[k]); sh1[k] = foc_f2h(o[8 + k]); }
    }
    for (uint32_t k = lane; k < num_steps; k += 64) {
        const float t = strip[k];
        const float dt = rm_clamp(t * p.dt_gamma, p.dt_min, p.dt_max);
        float last_t = t0;
        if (k > 0) { const float tp = strip[k - 1]; last_t = tp + rm_clamp(tp * p.dt_gamma, p.dt_min, p.dt_max); }
        const uint64_t s = (uint64_t)point_index + k;
        xyzs[s * 3] = rm_out(p, rm_clamp(fmaf(t, dx, ox), -p.bound, p.bound));
        xyzs[s * 3 + 1] = rm_out(p, rm_clamp(fmaf(t, dy, oy), -p.bound, p.bound));
        xyzs[s * 3 + 2] = rm_out(p, rm_clamp(fmaf(t, dz, oz), -p.bound, p.bound));
        if constexpr (FIELD) { *reinterpret_cast<rm_h8 *>(sh + s * 16) = sh0; *reinterpret_cast<rm_h8 *>(sh + s * 16 + 8) = sh1; }
        else { dirs[s * 3] = dx; dirs[s * 3 + 1] = dy; dirs[s * 3 + 2] = dz; }
        deltas[s * 2] = dt;
        deltas[s * 2 + 1] = (t + dt) - last_t;
    }
}

// ---------------------------------------------------------------- R7 (raymarching.cu:500-577), one wave per ray
__global__ void __launch_bounds__(256) k_composite_train_fwd(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                      const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                      uint32_t M, uint32_t N, float T_thresh,
                                      float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) {
        if (lane == 0) { weights_sum[index] = 0; depth[index] = 0; image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0; }
        return;
    }
    float T_carry = 1.0f, t_carry = 0.0f;
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    for (uint32_t base = 0; base < num_steps; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < num_steps;
        float sigma = 0, dt0 = 0, dt1 = 0, c0 = 0, c1 = 0, c2 = 0;
        if (valid) {
            const uint64_t s = (uint64_t)offset + i;
            sigma = sigmas[s];
            const float2 dl = *reinterpret_cast<const float2 *>(deltas + s * 2);
            dt0 = dl.x; dt1 = dl.y;
            c0 = rgbs[s * 3]; c1 = rgbs[s * 3 + 1]; c2 = rgbs[s * 3 + 2];
        }
        const float alpha = valid ? 1.0f - __expf(-sigma * dt0) : 0.0f;
        const float om = 1.0f - alpha;
        const float P = wave_incl_prod(om, (int)lane);
        float Pex = __shfl_up(P, 1, 64);
        if (lane == 0) Pex = 1.0f;
        const float T_before = T_carry * Pex;
        const float T_after = T_carry * P;
        const float tsum = t_carry + wave_incl_sum(dt1, (int)lane);
        // the reference breaks AFTER accumulating the sample whose T drops below the threshold
        const unsigned long long term = __ballot(valid && (T_after < T_thresh));
        const int first = term ? (int)__ffsll((long long)term) - 1 : 64;
        const float w = (valid && (int)lane <= first) ? alpha * T_before : 0.0f;
        r = fmaf(w, c0, r); g = fmaf(w, c1, g); b = fmaf(w, c2, b);
        d = fmaf(w, tsum, d);
        ws += w;
        if (term) break;
        T_carry = __shfl(T_after, 63, 64);
        t_carry = __shfl(tsum, 63, 64);
    }
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); d = wave_sum(d);
    if (lane == 0) {
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

// ---------------------------------------------------------------- R8 (raymarching.cu:601-682), one wave per ray
__global__ void __launch_bounds__(256) k_composite_train_bwd(const float *__restrict__ grad_weights_sum, const float *__restrict__ grad_image,
                                      const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                      const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                      const float *__restrict__ weights_sum, const float *__restrict__ image,
                                      uint32_t M, uint32_t N, float T_thresh,
                                      float *__restrict__ grad_sigmas, float *__restrict__ grad_rgbs) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float g0 = grad_image[index * 3], g1 = grad_image[index * 3 + 1], g2 = grad_image[index * 3 + 2];
    const float gws = grad_weights_sum ? grad_weights_sum[index] : 0.0f;
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
    const float ws_term = gws * (1 - weights_sum[index]);
    float T_carry = 1.0f;
    float r_carry = 0, g_carry = 0, b_carry = 0;
    for (uint32_t base = 0; base < num_steps; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < num_steps;
        const uint64_t s = (uint64_t)offset + (valid ? i : 0);
        float sigma = 0, dt0 = 0, c0 = 0, c1 = 0, c2 = 0;
        if (valid) {
            sigma = sigmas[s];
            dt0 = deltas[s * 2];
            c0 = rgbs[s * 3]; c1 = rgbs[s * 3 + 1]; c2 = rgbs[s * 3 + 2];
        }
        const float alpha = valid ? 1.0f - __expf(-sigma * dt0) : 0.0f;
        const float om = 1.0f - alpha;
        const float P = wave_incl_prod(om, (int)lane);
        float Pex = __shfl_up(P, 1, 64);
        if (lane == 0) Pex = 1.0f;
        const float T_before = T_carry * Pex;
        const float T_after = T_carry * P;
        const unsigned long long term = __ballot(valid && (T_after < T_thresh));
        const int first = term ? (int)__ffsll((long long)term) - 1 : 64;
        const bool act = valid && (int)lane <= first;
        const float w = act ? alpha * T_before : 0.0f;
        // running colour INCLUDING this sample (:648-650)
        const float r_acc = r_carry + wave_incl_sum(w * c0, (int)lane);
        const float g_acc = g_carry + wave_incl_sum(w * c1, (int)lane);
        const float b_acc = b_carry + wave_incl_sum(w * c2, (int)lane);
        if (act) {
            grad_rgbs[s * 3] = g0 * w; grad_rgbs[s * 3 + 1] = g1 * w; grad_rgbs[s * 3 + 2] = g2 * w;
            float acc = g0 * fmaf(T_after, c0, -(r_final - r_acc));
            acc = fmaf(g1, fmaf(T_after, c1, -(g_final - g_acc)), acc);
            acc = fmaf(g2, fmaf(T_after, c2, -(b_final - b_acc)), acc);
            acc += ws_term;
            grad_sigmas[s] = dt0 * acc;
        }
        if (term) break;
        T_carry = __shfl(T_after, 63, 64);
        r_carry = __shfl(r_acc, 63, 64); g_carry = __shfl(g_acc, 63, 64); b_carry = __shfl(b_acc, 63, 64);
    }
}

// ---------------------------------------------------------------- R9 (raymarching.cu:700-805)
// The serial loop of one list entry n (ray `index`). BAIL: stop at the first empty cell and report it (the two-phase form below: such a
// ray is a "walker" and is marched again, from its start, by the walkers' kernel).
template <bool BAIL>
__device__ __forceinline__ bool rm_lane_walk(uint32_t n, int index, uint32_t n_step, const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                             const float *__restrict__ rays_d, const uint8_t *__restrict__ grid, const RmParams &p, const float *__restrict__ fars,
                                             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas, const float *__restrict__ noises) {
    const float ox = rays_o[index * 3], oy = rays_o[index * 3 + 1], oz = rays_o[index * 3 + 2];
    const float dx = rays_d[index * 3], dy = rays_d[index * 3 + 1], dz = rays_d[index * 3 + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float *px = xyzs + (uint64_t)n * n_step * 3, *pd = dirs + (uint64_t)n * n_step * 3, *pl = deltas + (uint64_t)n * n_step * 2;
    float t = rays_t[index];
    const float far = fars[index];
    t = fmaf(rm_clamp(t * p.dt_gamma, p.dt_min, p.dt_max), noises[n], t);
    float last_t = t;
    uint32_t step = 0;
    RmCell c;
    while (t < far && step < n_step) {
        if (rm_cell(grid, p, ox, oy, oz, dx, dy, dz, t, c)) {
            px[0] = rm_out(p, c.x); px[1] = rm_out(p, c.y); px[2] = rm_out(p, c.z);
            pd[0] = dx; pd[1] = dy; pd[2] = dz;
            t += c.dt;
            const float span = t - last_t;
            pl[0] = c.dt; pl[1] = span;
            if (p.rederive) t = last_t + span;
            last_t = t;
            px += 3; pd += 3; pl += 2; step++;
        } else {
            if (BAIL) return true;
            t = rm_skip(p, c, t, dx, dy, dz, rdx, rdy, rdz);
        }
    }
    return false;
}

__global__ void __launch_bounds__(64) k_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                             const float *__restrict__ rays_t, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                             const uint8_t *__restrict__ grid, RmParams p, const float *__restrict__ fars,
                             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                             const float *__restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    if (index < 0) return;             // a list entry marked dead (-1, as composite_rays leaves them): its slots stay zero = "terminated"
    (void)rm_lane_walk<false>(n, index, n_step, rays_t, rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises);
}

// ---------------------------------------------------------------- R9, one ray per lane, the wave's samples staged in LDS
// k_march_rays writes a ray's samples as single dwords at a 32 n_step-byte lane stride (a burst of 8: 64 stores per lane, each touching 64
// cache lines) and relies on the caller for the zeros of the slots a ray does not fill. Here a lane collects its ray's burst in LDS
// ([3 n_step] positions | [2 n_step] deltas | samples filled, direction) and the wave then writes the three arrays of its 64 list entries as
// runs of consecutive floats, zeros included: every slot of every entry is written, whole cache lines at a time. Same loop, same bits.
__global__ void __launch_bounds__(64) k_march_rays_staged(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                             const float *__restrict__ rays_t, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                             const uint8_t *__restrict__ grid, RmParams p, const float *__restrict__ fars,
                             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                             const float *__restrict__ noises, int sample_major) {
    extern __shared__ float rm_stage[];                    // [64][5 n_step + 4]
    const uint32_t lane = threadIdx.x, n0 = blockIdx.x * 64u, n = n0 + lane;
    const uint32_t stride = 5u * n_step + 4u;
    float *mine = rm_stage + lane * stride;
    const int index = n < n_alive ? rays_alive[n] : -1;
    uint32_t step = 0;
    float dx = 0.0f, dy = 0.0f, dz = 0.0f;
    if (index >= 0) {
        const float ox = rays_o[index * 3], oy = rays_o[index * 3 + 1], oz = rays_o[index * 3 + 2];
        dx = rays_d[index * 3]; dy = rays_d[index * 3 + 1]; dz = rays_d[index * 3 + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        float t = rays_t[index];
        const float far = fars[index];
        t = fmaf(rm_clamp(t * p.dt_gamma, p.dt_min, p.dt_max), noises[n], t);
        float last_t = t;
        float *px = mine, *pl = mine + 3u * n_step;
        RmCell c;
        while (t < far && step < n_step) {
            if (rm_cell(grid, p, ox, oy, oz, dx, dy, dz, t, c)) {
                px[0] = rm_out(p, c.x); px[1] = rm_out(p, c.y); px[2] = rm_out(p, c.z);
                t += c.dt;
                const float span = t - last_t;
                pl[0] = c.dt; pl[1] = span;
                if (p.rederive) t = last_t + span;
                last_t = t;
                px += 3; pl += 2; step++;
            } else t = rm_skip(p, c, t, dx, dy, dz, rdx, rdy, rdz);
        }
    }
    float *tail = mine + 5u * n_step;
    tail[0] = __builtin_bit_cast(float, step); tail[1] = dx; tail[2] = dy; tail[3] = dz;
    __syncthreads();                                       // one wave per workgroup: orders the LDS traffic, costs nothing
    const uint32_t cnt = min(64u, n_alive - n0);           // list entries of this wave (n0 < n_alive by the launch)
    if (sample_major) {
        // [n_step][n_alive] arrays: slot s of the wave's entries is one run of 3 cnt (2 cnt) consecutive floats. The 64 rows of an encoder /
        // network wave are then 64 NEIGHBOURING rays at the same burst slot — closer to each other than 8 consecutive samples of 8 rays
        // (a pixel apart against a step apart) — and the composite kernel's loads are contiguous across lanes.
        for (uint32_t sl = 0; sl < n_step; sl++) {
            float *gx = xyzs + ((uint64_t)sl * n_alive + n0) * 3u, *gd = dirs + ((uint64_t)sl * n_alive + n0) * 3u, *gl = deltas + ((uint64_t)sl * n_alive + n0) * 2u;
            for (uint32_t k = lane; k < cnt * 3u; k += 64u) {
                const uint32_t r = k / 3u, cc = k - r * 3u;
                const float *rec = rm_stage + r * stride;
                const bool on = sl < __builtin_bit_cast(uint32_t, rec[5u * n_step]);
                gx[k] = on ? rec[sl * 3u + cc] : 0.0f;
                gd[k] = on ? rec[5u * n_step + 1u + cc] : 0.0f;
            }
            for (uint32_t k = lane; k < cnt * 2u; k += 64u) {
                const uint32_t r = k >> 1;
                const float *rec = rm_stage + r * stride;
                gl[k] = sl < __builtin_bit_cast(uint32_t, rec[5u * n_step]) ? rec[3u * n_step + sl * 2u + (k & 1u)] : 0.0f;
            }
        }
        return;
    }
    const uint32_t lx = 3u * n_step, ll = 2u * n_step;
    const float inv_lx = 1.0f / (float)lx, inv_ll = 1.0f / (float)ll;
    float *gx = xyzs + (uint64_t)n0 * lx, *gd = dirs + (uint64_t)n0 * lx, *gl = deltas + (uint64_t)n0 * ll;
    for (uint32_t i = lane; i < cnt * lx; i += 64u) {
        const uint32_t r = (uint32_t)(((float)i + 0.5f) * inv_lx), j = i - r * lx, slot = j / 3u, cc = j - slot * 3u;
        const float *rec = rm_stage + r * stride;
        const bool on = slot < __builtin_bit_cast(uint32_t, rec[5u * n_step]);
        gx[i] = on ? rec[j] : 0.0f;
        gd[i] = on ? rec[5u * n_step + 1u + cc] : 0.0f;
    }
    for (uint32_t i = lane; i < cnt * ll; i += 64u) {
        const uint32_t r = (uint32_t)(((float)i + 0.5f) * inv_ll), j = i - r * ll;
        const float *rec = rm_stage + r * stride;
        gl[i] = (j >> 1) < __builtin_bit_cast(uint32_t, rec[5u * n_step]) ? rec[3u * n_step + j] : 0.0f;
    }
}

// ---------------------------------------------------------------- R9, G lanes per ray
// k_march_rays above is a chain of dependent bitfield lookups per ray, and a render iteration has few rays left alive (the host loop keeps
// live x n_step <= N, so most iterations march ~N/8 rays by 8 samples: about one wave per SIMD, nothing to hide a lookup's latency
// behind, and the launch lasts as long as its longest ray — one that leaves the object and walks ~100 cells to the box's far side).
// As in k_march_count_wave the values t can take are a fixed lattice per ray (both branches of the loop advance t by
// clamp(t dt_gamma, dt_min, dt_max), raymarching.cu:758-795), so G consecutive lattice points of a ray are generated on G lanes (the
// recurrence itself, each value entering at the row's last lane through a DPP row_shl:1 move), looked up with ONE round of loads, and the
// loop's control flow is replayed on the G-bit masks of the ray's lane group: an occupied visited point emits (position, direction, dt,
// t_new - last_t) into the ray's next slot and moves on by one lane; an empty one moves to the first lane whose t is not below the voxel
// exit, carried into the next G points when there is none. Same visits, same samples, bit for bit — with chains G times shorter. A full
// wave per ray (G = 64, the training kernel's form) would spend 64 recurrence steps for the 8 samples an iteration asks for; G = 16
// (one DPP row, 4 rays per wave) generates what a burst of 8 typically consumes in one or two rounds.
#define RM_ROW_MAX_ROUNDS (1u << 14)                       // x 16 lattice points: far beyond any real ray; makes the loop finite whatever the inputs
// all 64 lanes of a wave call this together; `have`, `n`, `index` are uniform over each 16-lane group
// STAGE: the samples of a ray are collected in LDS (`stage`: 5 n_step floats per 16-lane group) and leave in rm_row_flush — whole runs of
// consecutive floats per group, zeros in the slots the ray did not fill — instead of as single dwords at a 32 n_step-byte lane stride.
template <bool MED3, bool STAGE = false>
__device__ __forceinline__ uint32_t rm_row_walk(bool have, uint32_t n, int index, uint32_t n_step, const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                            const float *__restrict__ rays_d, const uint8_t *__restrict__ grid, const RmParams &p, const float *__restrict__ fars,
                                            float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas, const float *__restrict__ noises,
                                            float *stage = nullptr) {
    constexpr uint32_t G = 16u;
    const uint32_t lane = threadIdx.x & 63u, sub = lane & (G - 1u), gbase = lane & ~(G - 1u), gshift = gbase;
    const float ox = rays_o[index * 3], oy = rays_o[index * 3 + 1], oz = rays_o[index * 3 + 2];
    const float dx = rays_d[index * 3], dy = rays_d[index * 3 + 1], dz = rays_d[index * 3 + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float far = fars[index];
    float t_cur = rays_t[index];
    t_cur = fmaf(rm_clamp(t_cur * p.dt_gamma, p.dt_min, p.dt_max), noises[have ? n : 0u], t_cur);
    float last_t = t_cur;
    float *px = xyzs + (uint64_t)n * n_step * 3, *pd = dirs + (uint64_t)n * n_step * 3, *pl = deltas + (uint64_t)n * n_step * 2;
    uint32_t step = 0;
    bool skipping = false, live = have;                    // live: this ray's loop has not ended (uniform over the ray's G lanes)
    float skip_to = 0.0f;
    const uint32_t below = (1u << sub) - 1u;
    for (uint32_t round = 0; round < RM_ROW_MAX_ROUNDS && __builtin_amdgcn_ballot_w64(live) != 0ull; round++) {
        float T = t_cur, t_next = t_cur;
#pragma unroll
        for (uint32_t j = 0; j < G; j++) {
            T = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t_next), __builtin_bit_cast(int, T), 0x101, 0xf, 0xf, false));
            const float raw = t_next * p.dt_gamma;
            t_next += MED3 ? __builtin_amdgcn_fmed3f(raw, p.dt_min, p.dt_max) : rm_clamp(raw, p.dt_min, p.dt_max);
        }
        // every lane of the wave evaluates its lattice point (rays that have ended idle along: their results are not looked at)
        RmCell c;
        const bool occupied_here = rm_cell(grid, p, ox, oy, oz, dx, dy, dz, T, c);
        const float exit_here = rm_skip_target(p, c, T, dx, dy, dz, rdx, rdy, rdz);
        const float t_new = T + c.dt;                      // the loop's `t += dt` of an emitting visit (== the next lattice point)
        const uint32_t in_range = (uint32_t)(__ballot(T < far) >> gshift) & 0xFFFFu;
        const uint32_t occ = (uint32_t)(__ballot(occupied_here) >> gshift) & 0xFFFFu;
        uint32_t emitted = 0u;
        if (live) {
            uint32_t k = 0;
            bool go = true;
            if (skipping) {
                const uint32_t landed = (uint32_t)(__ballot(!(T < skip_to)) >> gshift) & 0xFFFFu;
                if (landed == 0u) {
                    if (!((in_range >> (G - 1u)) & 1u)) live = false;      // t only grows: wherever the skip lands, it is beyond `far`
                    go = false;
                } else { k = (uint32_t)__builtin_ctz(landed); skipping = false; }
            }
            const uint32_t emit_ok = occ & in_range;
            uint32_t room = n_step - step;
            while (go && k < G) {
                if (!((in_range >> k) & 1u) || room == 0u) { live = false; break; }        // the loop condition of raymarching.cu:745
                if ((emit_ok >> k) & 1u) {                 // a run of occupied points: each emits and steps to its successor
                    const uint32_t rest = ~(emit_ok >> k) & (0xFFFFu >> k);
                    uint32_t len = rest ? (uint32_t)__builtin_ctz(rest) : G - k;
                    len = len < room ? len : room;
                    emitted |= ((1u << len) - 1u) << k;
                    room -= len;
                    k += len;
                } else {
                    const float tt = __shfl(exit_here, (int)(gbase + k), 64);
                    const uint32_t above = k >= G - 1u ? 0u : ((0xFFFFu << (k + 1u)) & 0xFFFFu);
                    const uint32_t landed = (uint32_t)(__ballot(!(T < tt)) >> gshift) & above;      // do { advance } while (t < tt): at least one advance
                    if (landed == 0u) { skipping = true; skip_to = tt; break; }
                    k = (uint32_t)__builtin_ctz(landed);
                }
            }
        }
        // the emitting lanes write their samples; delta[1] spans from the previous emitted sample's end (or the carried one)
        const uint32_t before = emitted & below;
        const int prev_lane = before ? (int)(gbase + 31u - (uint32_t)__builtin_clz(before)) : (int)lane;
        const float prev_end = __shfl(t_new, prev_lane, 64);
        if ((emitted >> sub) & 1u) {
            const uint32_t slot = step + (uint32_t)__builtin_popcount(before);
            if (STAGE) {
                stage[slot * 3] = rm_out(p, c.x); stage[slot * 3 + 1] = rm_out(p, c.y); stage[slot * 3 + 2] = rm_out(p, c.z);
                stage[3 * n_step + slot * 2] = c.dt; stage[3 * n_step + slot * 2 + 1] = t_new - (before ? prev_end : last_t);
            } else {
                px[slot * 3] = rm_out(p, c.x); px[slot * 3 + 1] = rm_out(p, c.y); px[slot * 3 + 2] = rm_out(p, c.z);
                pd[slot * 3] = dx; pd[slot * 3 + 1] = dy; pd[slot * 3 + 2] = dz;
                pl[slot * 2] = c.dt; pl[slot * 2 + 1] = t_new - (before ? prev_end : last_t);
            }
        }
        const int top_lane = emitted ? (int)(gbase + 31u - (uint32_t)__builtin_clz(emitted)) : (int)lane;
        const float top_end = __shfl(t_new, top_lane, 64);
        if (emitted) last_t = top_end;
        step += (uint32_t)__builtin_popcount(emitted);
        t_cur = t_next;
    }
    return step;
}

// the staged samples of list entry n (`filled` of its n_step slots) -> memory; every lane of the 16-lane group calls this after the block's barrier
__device__ __forceinline__ void rm_row_flush(uint32_t n, int index, uint32_t n_step, uint32_t filled, const float *__restrict__ rays_d, const float *stage,
                                             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas) {
    const uint32_t sub = threadIdx.x & 15u;
    const float dx = rays_d[index * 3], dy = rays_d[index * 3 + 1], dz = rays_d[index * 3 + 2];
    float *px = xyzs + (uint64_t)n * n_step * 3, *pd = dirs + (uint64_t)n * n_step * 3, *pl = deltas + (uint64_t)n * n_step * 2;
    for (uint32_t i = sub; i < n_step * 3u; i += 16u) {
        const uint32_t slot = i / 3u, c = i - slot * 3u;
        const bool on = slot < filled;
        px[i] = on ? stage[i] : 0.0f;
        pd[i] = on ? (c == 0u ? dx : c == 1u ? dy : dz) : 0.0f;
    }
    for (uint32_t i = sub; i < n_step * 2u; i += 16u) pl[i] = (i >> 1) < filled ? stage[3 * n_step + i] : 0.0f;
}

template <bool MED3>
__global__ void __launch_bounds__(256) k_march_rays_row(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                             const float *__restrict__ rays_t, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                             const uint8_t *__restrict__ grid, RmParams p, const float *__restrict__ fars,
                             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                             const float *__restrict__ noises) {
    __shared__ float stage[16][5 * 16];                     // per 16-lane group: positions [n_step, 3] | deltas [n_step, 2], n_step <= 16
    const uint32_t n = (blockIdx.x * 256u + threadIdx.x) / 16u;
    const int listed = n < n_alive ? rays_alive[n] : -1;
    const bool have = listed >= 0;     // beyond the list, or an entry marked dead (-1): nothing to march
    float *mine = stage[threadIdx.x >> 4];
    const uint32_t filled = rm_row_walk<MED3, true>(have, n, have ? listed : 0, n_step, rays_t, rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises, mine);
    __syncthreads();
    // every entry of the list leaves with all of its slots written (a dead one: zeros = "terminated"), so the caller need not have zeroed them
    if (n < n_alive) rm_row_flush(n, have ? listed : 0, n_step, have ? filled : 0u, rays_d, mine, xyzs, dirs, deltas);
}

// ---------------------------------------------------------------- R9 in two phases (the native render step, csrc/occrender.hip)
// A launch with hundreds of thousands of live rays asks each of them for one sample: a ray inside the object evaluates one cell and is
// done, while the few percent that have just left it walk ~100 empty cells to the far side of the box — and every wave runs as long as
// its longest lane (48 us per launch, of which the cell evaluations proper are ~3 us). Phase 1 (k_march_rays_first): the serial loop
// per lane, but a ray that meets an EMPTY cell stops there and puts its list entry on a worklist (one atomic per wave). Phase 2
// (k_march_walkers): the worklist's rays are marched again from their start — densely packed: 16 lanes per ray while there are few of them
// (chains 16 times shorter), one ray per lane when the list is long (the first iteration of a view, where every ray starts in empty
// space). Same samples as the single kernel, bit for bit: a walker's slots are simply written twice with the same values.
__global__ void __launch_bounds__(256) k_march_rays_first(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                             const float *__restrict__ rays_t, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                             const uint8_t *__restrict__ grid, RmParams p, const float *__restrict__ fars,
                             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                             const float *__restrict__ noises, int32_t *__restrict__ worklist, int32_t *__restrict__ wl_count) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    bool walker = false;
    if (n < n_alive) {
        const int index = rays_alive[n];
        if (index >= 0) walker = rm_lane_walk<true>(n, index, n_step, rays_t, rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises);
    }
    const uint64_t mask = __ballot(walker);
    if (mask != 0ull) {
        const uint32_t lane = threadIdx.x & 63u;
        int base = 0;
        if (lane == (uint32_t)__builtin_ctzll(mask)) base = atomicAdd(wl_count, (int)__builtin_popcountll(mask));
        base = __shfl(base, (int)__builtin_ctzll(mask), 64);
        if (walker) worklist[base + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (int32_t)n;
    }
}

template <bool MED3>
__global__ void __launch_bounds__(256) k_march_walkers(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                             const float *__restrict__ rays_t, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                             const uint8_t *__restrict__ grid, RmParams p, const float *__restrict__ fars,
                             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                             const float *__restrict__ noises, const int32_t *__restrict__ worklist, const int32_t *__restrict__ wl_count, uint32_t row_max) {
    const uint32_t count = min((uint32_t)max(*wl_count, 0), n_alive);         // grid-uniform
    if (count <= row_max) {
        const uint32_t groups_total = gridDim.x * 16u;                          // 16-lane groups of the grid
        const uint32_t wave_first = (blockIdx.x * 256u + (threadIdx.x & ~63u)) / 16u;
        for (uint32_t base = wave_first; base < count; base += groups_total) {  // wave-uniform trip count: the four groups walk together
            const uint32_t i = base + ((threadIdx.x & 63u) >> 4);
            const bool have = i < count;
            const uint32_t n = have ? (uint32_t)worklist[i] : 0u;
            rm_row_walk<MED3>(have, n, have ? rays_alive[n] : 0, n_step, rays_t, rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises);
        }
    } else {
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
            const uint32_t n = (uint32_t)worklist[i];
            (void)rm_lane_walk<false>(n, rays_alive[n], n_step, rays_t, rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises);
        }
    }
}

// Where rays die, for a caller that deals the samples over iterations differently from the reference and must know what the reference's
// own loop would have done (focnerf_amd/renderer.py): hist[min(base + j, len - 1)][slice] += 1 for a ray that ends at slot j of this call
// (no sample there, or the transmittance test after it). hist == nullptr: nothing recorded. A view's rays die in a few iterations, tens of
// thousands per launch into n_step bins: the wave adds its rays per bin with one atomic, and the bins come in RM_DEATH_SLICES copies
// (slice = wave index mod RM_DEATH_SLICES; the reader sums them) — one add per ray on 8 addresses cost 5 ms per view.
#define RM_DEATH_SLICES 64u
struct RmDeaths { int32_t *hist; uint32_t base, len; int sample_major; };      // (sample_major: the layout of the sample arrays rides along)
__device__ __forceinline__ void rm_record_deaths(const RmDeaths &dh, bool died, uint32_t at, uint32_t n_step, uint32_t n) {
    if (!dh.hist) return;
    if (__ballot(died) == 0ull) return;
    const uint32_t slice = (n >> 6) & (RM_DEATH_SLICES - 1u);
    for (uint32_t j = 0; j < n_step; j++) {
        const uint64_t m = __ballot(died && at == j);
        if (m != 0ull && (threadIdx.x & 63u) == 0u)
            atomicAdd(&dh.hist[(uint64_t)min(dh.base + j, dh.len - 1u) * RM_DEATH_SLICES + slice], (int)__builtin_popcountll(m));
    }
}

// ---------------------------------------------------------------- R10 (raymarching.cu:818-905)
// COUNT: the wave also adds its number of surviving entries to block_counts[n / 1024] (zeroed by the caller) — the first pass of the ordered
// compaction that follows in the native render step (k_compact_count otherwise)
template <bool COUNT>
__global__ void __launch_bounds__(64) k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                                 int32_t *__restrict__ rays_alive, float *__restrict__ rays_t,
                                 const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                 float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image, int32_t *__restrict__ block_counts,
                                 RmDeaths dh) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    const int index = n < n_alive ? rays_alive[n] : -1;
    bool survives = false, died = false;
    uint32_t died_at = 0;
    if (index >= 0) {                  // beyond the list or already marked dead (stays dead, nothing to accumulate)
    // ray-major [n_alive][n_step] (the reference's layout) or sample-major [n_step][n_alive] (dh.sample_major: the native render step)
    const uint64_t first = dh.sample_major ? (uint64_t)n : (uint64_t)n * n_step, hop = dh.sample_major ? (uint64_t)n_alive : 1ull;
    const float *s = sigmas + first, *c = rgbs + first * 3, *dl = deltas + first * 2;
    float t = rays_t[index];
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0) break;
        const float alpha = 1.0f - __expf(-s[0] * dl[0]);
        const float T = 1 - weight_sum;
        const float weight = alpha * T;
        weight_sum += weight;
        t += dl[1];
        d = fmaf(weight, t, d);
        r = fmaf(weight, c[0], r); g = fmaf(weight, c[1], g); b = fmaf(weight, c[2], b);
        if (T < T_thresh) break;
        s += hop; c += 3 * hop; dl += 2 * hop; step++;
    }
    if (step < n_step) { rays_alive[n] = -1; died = true; died_at = step; } else { rays_t[index] = t; survives = true; }
    weights_sum[index] = weight_sum; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
    if (COUNT) {
        const uint64_t alive = __ballot(survives);
        if ((threadIdx.x & 63u) == 0u && alive != 0ull) atomicAdd(&block_counts[n >> 10], (int)__builtin_popcountll(alive));
        rm_record_deaths(dh, died, died_at, n_step, n);
    }
}

// The same with a compile-time burst length that is a multiple of 4: the lane first loads ALL of its ray's sigmas / colours / deltas with
// 16-byte loads (NS, 3 NS and 2 NS consecutive floats: whole cache lines per lane, every load in flight at once) and then runs the serial
// accumulation on registers — with the pointer-chasing loop above a burst of 8 is 8 dependent rounds of 4-byte loads at a 32-byte lane stride
// (86 us per 5.1 M samples, 1.4 TB/s). Same operations in the same order: same bits.
template <int NS, bool COUNT>
__global__ void __launch_bounds__(64) k_composite_rays_pre(uint32_t n_alive, float T_thresh, int32_t *__restrict__ rays_alive, float *__restrict__ rays_t,
                                     const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                     float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image, int32_t *__restrict__ block_counts,
                                     RmDeaths dh) {
    static_assert(NS % 4 == 0, "whole float4 loads");
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    const int index = n < n_alive ? rays_alive[n] : -1;
    bool survives = false, died = false;
    uint32_t died_at = 0;
    if (index >= 0) {
        float sg[NS], cl[3 * NS], dl[2 * NS];
        if (dh.sample_major) {             // [NS][n_alive]: consecutive lanes read consecutive rays of one slot
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const uint64_t m = (uint64_t)i * n_alive + n;
                sg[i] = sigmas[m];
                cl[3 * i] = rgbs[m * 3]; cl[3 * i + 1] = rgbs[m * 3 + 1]; cl[3 * i + 2] = rgbs[m * 3 + 2];
                const float2 v = *reinterpret_cast<const float2 *>(deltas + m * 2);
                dl[2 * i] = v.x; dl[2 * i + 1] = v.y;
            }
        } else {
        const float4 *ps = reinterpret_cast<const float4 *>(sigmas + (uint64_t)n * NS);
        const float4 *pc = reinterpret_cast<const float4 *>(rgbs + (uint64_t)n * NS * 3);
        const float4 *pd = reinterpret_cast<const float4 *>(deltas + (uint64_t)n * NS * 2);
#pragma unroll
        for (int i = 0; i < NS / 4; i++) { const float4 v = ps[i]; sg[4 * i] = v.x; sg[4 * i + 1] = v.y; sg[4 * i + 2] = v.z; sg[4 * i + 3] = v.w; }
#pragma unroll
        for (int i = 0; i < 3 * NS / 4; i++) { const float4 v = pc[i]; cl[4 * i] = v.x; cl[4 * i + 1] = v.y; cl[4 * i + 2] = v.z; cl[4 * i + 3] = v.w; }
#pragma unroll
        for (int i = 0; i < 2 * NS / 4; i++) { const float4 v = pd[i]; dl[4 * i] = v.x; dl[4 * i + 1] = v.y; dl[4 * i + 2] = v.z; dl[4 * i + 3] = v.w; }
        }
        float t = rays_t[index];
        float weight_sum = weights_sum[index], d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        bool ended = false;
        uint32_t ended_at = 0;
#pragma unroll
        for (int step = 0; step < NS; step++) {
            if (!ended) {
                ended_at = (uint32_t)step;
                if (dl[2 * step] == 0) ended = true;
                else {
                    const float alpha = 1.0f - __expf(-sg[step] * dl[2 * step]);
                    const float T = 1 - weight_sum;
                    const float weight = alpha * T;
                    weight_sum += weight;
                    t += dl[2 * step + 1];
                    d = fmaf(weight, t, d);
                    r = fmaf(weight, cl[3 * step], r); g = fmaf(weight, cl[3 * step + 1], g); b = fmaf(weight, cl[3 * step + 2], b);
                    if (T < T_thresh) ended = true;
                }
            }
        }
        if (ended) { rays_alive[n] = -1; died = true; died_at = ended_at; } else { rays_t[index] = t; survives = true; }
        weights_sum[index] = weight_sum; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
    if (COUNT) {
        const uint64_t alive = __ballot(survives);
        if ((threadIdx.x & 63u) == 0u && alive != 0ull) atomicAdd(&block_counts[n >> 10], (int)__builtin_popcountll(alive));
        rm_record_deaths(dh, died, died_at, (uint32_t)NS, n);
    }
}

template <bool COUNT>
static void rm_launch_composite(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t, const float *sigmas, const float *rgbs,
                                const float *deltas, float *weights_sum, float *depth, float *image, int32_t *block_counts, hipStream_t st,
                                RmDeaths dh = RmDeaths{nullptr, 0u, 1u, 0}) {
    const dim3 grid(foc_div_up(n_alive, 64)), block(64);
    const bool aligned = ((reinterpret_cast<uintptr_t>(sigmas) | reinterpret_cast<uintptr_t>(rgbs) | reinterpret_cast<uintptr_t>(deltas)) & 15u) == 0;
    if (aligned && n_step == 4u)
        hipLaunchKernelGGL((k_composite_rays_pre<4, COUNT>), grid, block, 0, st, n_alive, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, block_counts, dh);
    else if (aligned && n_step == 8u)
        hipLaunchKernelGGL((k_composite_rays_pre<8, COUNT>), grid, block, 0, st, n_alive, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, block_counts, dh);
    else if (aligned && n_step == 16u)
        hipLaunchKernelGGL((k_composite_rays_pre<16, COUNT>), grid, block, 0, st, n_alive, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, block_counts, dh);
    else
        hipLaunchKernelGGL(k_composite_rays<COUNT>, grid, block, 0, st, n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                           block_counts, dh);
}

// ---------------------------------------------------------------- ordered compaction of rays_alive >= 0
__global__ void __launch_bounds__(1024) k_compact_count(const int32_t *__restrict__ in, uint32_t n, int32_t *__restrict__ block_counts) {
    __shared__ int s_w[16];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const int keep = (i < n && in[i] >= 0) ? 1 : 0;
    const int c = wave_sum_i(keep);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; k++) t += s_w[k]; block_counts[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(1024) k_compact_scan(int32_t *__restrict__ block_counts, uint32_t nb, int32_t *__restrict__ n_out) {
    // single workgroup: exclusive scan of nb block counts in place
    __shared__ int s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = tid * per, hi = min(nb, lo + per);
    int local = 0;
    for (uint32_t i = lo; i < hi; i++) local += block_counts[i];
    const int incl = wave_incl_sum_i(local, (int)lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int v = lane < 16 ? s_wave[lane] : 0;
        const int vi = wave_incl_sum_i(v, (int)lane);
        if (lane < 16) s_wave[lane] = vi - v;
        if (lane == 15) n_out[0] = vi;
    }
    __syncthreads();
    int run = s_wave[wave] + (incl - local);
    for (uint32_t i = lo; i < hi; i++) { const int c = block_counts[i]; block_counts[i] = run; run += c; }
}
__global__ void __launch_bounds__(1024) k_compact_scatter(const int32_t *__restrict__ in, uint32_t n, const int32_t *__restrict__ block_base,
                                                          int32_t *__restrict__ out) {
    __shared__ int s_w[16];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = i < n ? in[i] : -1;
    const int keep = v >= 0 ? 1 : 0;
    const int incl = wave_incl_sum_i(keep, (int)lane);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (uint32_t k = 0; k < wave; k++) wbase += s_w[k];
    if (keep) out[block_base[blockIdx.x] + wbase + incl - 1] = v;
}

// ================================================================= host entry points
extern "C" {

int foc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                           float *nears, float *fars, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_o);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(rays_o && rays_d && aabb && nears && fars, FOC_E_INVALID, "near_far_from_aabb: null pointer");
    hipLaunchKernelGGL(k_near_far_from_aabb, dim3(foc_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, aabb, N, min_near, nears, fars);
    FOC_CHECK_LAUNCH("near_far_from_aabb");
    return FOC_OK;
}

int foc_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_o);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(rays_o && rays_d && coords, FOC_E_INVALID, "sph_from_ray: null pointer");
    hipLaunchKernelGGL(k_sph_from_ray, dim3(foc_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, radius, N, coords);
    FOC_CHECK_LAUNCH("sph_from_ray");
    return FOC_OK;
}

int foc_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream) {
    FocDeviceGuard foc_guard_(stream, coords);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(coords && indices, FOC_E_INVALID, "morton3D: null pointer");
    hipLaunchKernelGGL(k_morton3D, dim3(foc_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream, coords, N, indices);
    FOC_CHECK_LAUNCH("morton3D");
    return FOC_OK;
}

int foc_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream) {
    FocDeviceGuard foc_guard_(stream, indices);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(coords && indices, FOC_E_INVALID, "morton3D_invert: null pointer");
    hipLaunchKernelGGL(k_morton3D_invert, dim3(foc_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream, indices, N, coords);
    FOC_CHECK_LAUNCH("morton3D_invert");
    return FOC_OK;
}

int foc_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, void *stream) {
    FocDeviceGuard foc_guard_(stream, grid);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(grid && bitfield, FOC_E_INVALID, "packbits: null pointer");
    uint32_t N4 = 0;
    if ((((uintptr_t)grid) & 15u) == 0 && (((uintptr_t)bitfield) & 3u) == 0) N4 = N / 4;
    if (N4) {
        hipLaunchKernelGGL(k_packbits_x4, dim3(foc_grid_1d(N4, 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4 *>(grid), N4, density_thresh, reinterpret_cast<uint32_t *>(bitfield));
        FOC_CHECK_LAUNCH("packbits");
    }
    if (N4 * 4 < N) {
        hipLaunchKernelGGL(k_packbits_tail, dim3(foc_grid_1d(N - N4 * 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           grid, N4 * 4, N, density_thresh, bitfield);
        FOC_CHECK_LAUNCH("packbits(tail)");
    }
    return FOC_OK;
}

// counts [N] (+ pad), then the per-ray strips of sample positions [N, max_steps]
static uint64_t rm_strip_offset(uint32_t N) { return (((uint64_t)N + 64) * sizeof(int32_t) + 255) & ~(uint64_t)255; }
uint64_t foc_march_rays_train_scratch_bytes(uint32_t N, uint32_t max_steps) { return rm_strip_offset(N) + (uint64_t)N * max_steps * sizeof(float); }

static int rm_march_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                          uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                          float *nears, float *fars, float *xyzs, float *dirs, float *deltas,
                          int32_t *rays, int32_t *counter, const float *noises, int32_t *scratch, void *sh_rows, bool field, uint32_t pad_align,
                          const float *aabb, float min_near, void *stream) {
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(rays_o && rays_d && grid && nears && fars && rays && counter && noises && scratch, FOC_E_INVALID,
                "march_rays_train: null pointer");
    FOC_REQUIRE(M == 0 || (xyzs && deltas && (field ? sh_rows != nullptr : dirs != nullptr)), FOC_E_INVALID, "march_rays_train: null output with M > 0");
    FOC_REQUIRE(C >= 1 && C <= 8 && H >= 2 && H <= 512 && max_steps >= 1, FOC_E_INVALID,
                "march_rays_train: unsupported C=%u H=%u max_steps=%u", C, H, max_steps);
    // the float index `level*H^3 + morton` of raymarching.cu:378 is exact only below 2^24
    FOC_REQUIRE((uint64_t)C * H * H * H <= (1ull << 24), FOC_E_INVALID, "march_rays_train: C*H^3 exceeds 2^24");
    RmParams p = rm_make_params(bound, dt_gamma, max_steps, C, H);
    hipStream_t st = (hipStream_t)stream;
    float *tstrip = reinterpret_cast<float *>(reinterpret_cast<char *>(scratch) + rm_strip_offset(N));
    // A wave per ray repeats the lattice recurrence on 64 lanes: about 4x the instructions of a ray per lane, in exchange for a chain
    // of dependent lookups 64x shorter. That pays while the lane-per-ray kernel is latency-bound. Measured per call (tools/
    // time_march_modes.py, wave / lane): 4096 rays 77 / 138 us, 8192 127 / 178, 16384 221 / 267, 32768 417 / 278, 65536 825 / 460.
    // FOC_MARCH_SERIAL=1 / 0 forces one.
    const int forced = foc_opt(FOC_OPT_MARCH_SERIAL);
    const bool serial = forced >= 0 ? forced != 0 : N > 16384u;
    if (serial)
        hipLaunchKernelGGL(k_march_count, dim3(foc_div_up(N, 64)), dim3(64), 0, st, rays_o, rays_d, grid, p, max_steps, N, nears, fars, noises, scratch, tstrip, aabb, min_near);
    else
        hipLaunchKernelGGL(p.dt_min <= p.dt_max ? k_march_count_wave<true> : k_march_count_wave<false>, dim3(foc_div_up(N, 4)), dim3(256), 0, st, rays_o, rays_d,
                           grid, p, max_steps, N, nears, fars, noises, scratch, tstrip, aabb, min_near, (const int32_t *)counter);
    FOC_CHECK_LAUNCH("march_rays_train(count)");
    // The reference's callers always pass a freshly zeroed counter (legacy/nerf/renderer.py:281-283):
    // rays rows are written at index i (ray order); counter[0] is honoured as the base offset.
    // Batches of the wave form (<= 16 384 rays: every emit workgroup can afford to add up the counts before its rays) reserve their slots in
    // the emit pass; larger ones keep the one-workgroup scan.
    const bool scan_in_emit = !serial && N <= 16384u;     // (FOC_MARCH_SERIAL=0 can force the wave form on larger batches)
    if (!scan_in_emit) {
        hipLaunchKernelGGL(k_march_scan, dim3(1), dim3(1024), 0, st, scratch, N, rays, counter);
        FOC_CHECK_LAUNCH("march_rays_train(scan)");
    }
    if (field) {
        p.norm_inv = 1.0f / (2.0f * bound);                // (x + bound) / (2 bound) as torch evaluates it: times the reciprocal (grid.py:149)
        auto emit = k_march_emit<true, false>;
        if (scan_in_emit) emit = k_march_emit<true, true>;
        hipLaunchKernelGGL(emit, dim3(foc_div_up(N, 4) + RM_PAD_BLOCKS), dim3(256), 0, st, rays_o, rays_d, p, max_steps, N, M, nears, noises, rays, tstrip, xyzs,
                           (float *)nullptr, deltas, (_Float16 *)sh_rows, counter, pad_align, (const int32_t *)scratch);
    } else {
        auto emit = k_march_emit<false, false>;
        if (scan_in_emit) emit = k_march_emit<false, true>;
        hipLaunchKernelGGL(emit, dim3(foc_div_up(N, 4)), dim3(256), 0, st, rays_o, rays_d, p, max_steps, N, M, nears, noises, rays, tstrip, xyzs, dirs, deltas,
                           (_Float16 *)nullptr, counter, 0u, (const int32_t *)scratch);
    }
    FOC_CHECK_LAUNCH("march_rays_train(emit)");
    return FOC_OK;
}

int foc_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                         int32_t *rays, int32_t *counter, const float *noises, int32_t *scratch, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_o);
    return rm_march_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, const_cast<float *>(nears), const_cast<float *>(fars), xyzs, dirs, deltas, rays,
                          counter, noises, scratch, nullptr, false, 0u, nullptr, 0.0f, stream);
}

int foc_march_rays_train_field(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                               uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                               float *nears, float *fars, float *enc_in, void *sh_rows, float *deltas,
                               int32_t *rays, int32_t *counter, const float *noises, int32_t *scratch, uint32_t pad_align,
                               const float *aabb, float min_near, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_o);
    return rm_march_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, enc_in, nullptr, deltas, rays, counter, noises, scratch, sh_rows, true,
                          pad_align, aabb, min_near, stream);
}

int foc_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                                     uint32_t M, uint32_t N, float T_thresh, float *weights_sum, float *depth, float *image,
                                     void *stream) {
    FocDeviceGuard foc_guard_(stream, sigmas);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(rays && weights_sum && depth && image, FOC_E_INVALID, "composite_rays_train_forward: null pointer");
    FOC_REQUIRE(M == 0 || (sigmas && rgbs && deltas), FOC_E_INVALID, "composite_rays_train_forward: null input with M > 0");
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream,
                       sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
    FOC_CHECK_LAUNCH("composite_rays_train_forward");
    return FOC_OK;
}

int foc_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas,
                                      const float *rgbs, const float *deltas, const int32_t *rays, const float *weights_sum,
                                      const float *image, uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                      float *grad_rgbs, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad_image);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(grad_image && rays && weights_sum && image, FOC_E_INVALID, "composite_rays_train_backward: null pointer");
    FOC_REQUIRE(M == 0 || (sigmas && rgbs && deltas && grad_sigmas && grad_rgbs), FOC_E_INVALID,
                "composite_rays_train_backward: null buffer with M > 0");
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream,
                       grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh,
                       grad_sigmas, grad_rgbs);
    FOC_CHECK_LAUNCH("composite_rays_train_backward");
    return FOC_OK;
}

int foc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                   const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                   const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                   const float *noises, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_alive);
    (void)nears;
    if (n_alive == 0) return FOC_OK;
    FOC_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas && noises, FOC_E_INVALID,
                "march_rays: null pointer");
    FOC_REQUIRE(C >= 1 && C <= 8 && H >= 2 && H <= 512 && max_steps >= 1 && n_step >= 1, FOC_E_INVALID,
                "march_rays: unsupported C=%u H=%u max_steps=%u n_step=%u", C, H, max_steps, n_step);
    FOC_REQUIRE((uint64_t)C * H * H * H <= (1ull << 24), FOC_E_INVALID, "march_rays: C*H^3 exceeds 2^24");
    const RmParams p = rm_make_params(bound, dt_gamma, max_steps, C, H);
    // 16 lanes per ray (k_march_rays_row) while the launch is latency-bound, one ray per lane beyond (FOC_MARCH_RAYS_ROW_MAX = most rays
    // the row form takes; 0 = never). Measured on the 800 x 800 occupancy render (profiles/, tools/quick_render_stats.sh; 549 launches,
    // most of them with > 262 144 rays alive and one sample per ray): one ray per lane 48.1 us average; the row form for every launch
    // 79.9 us (16x the cell evaluations where a ray needs one); row form up to 65 536 / 131 072 rays: 26-28 us on those launches (120 of
    // 549) against ~40 us. A lane form looking 8 lattice points ahead measured 53.8 us against 48.1 (round 3; removed): with
    // ~10 waves per SIMD the big launches are bound by the divergent walks' instruction count, not by their lookup chains.
    const long row_max = foc_opt(FOC_OPT_MARCH_RAYS_ROW_MAX);
    if ((long)n_alive <= row_max && n_step <= 16u)
        hipLaunchKernelGGL(p.dt_min <= p.dt_max ? k_march_rays_row<true> : k_march_rays_row<false>, dim3(foc_div_up((uint64_t)n_alive * 16u, 256)), dim3(256), 0,
                           (hipStream_t)stream, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises);
    else
        hipLaunchKernelGGL(k_march_rays, dim3(foc_div_up(n_alive, 64)), dim3(64), 0, (hipStream_t)stream, n_alive, n_step, rays_alive,
                           rays_t, rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises);
    FOC_CHECK_LAUNCH("march_rays");
    return FOC_OK;
}

// which kernels foc_march_rays_two_phase takes for a burst of n_step samples of n_alive rays: 0 = the two phases, 1 = 16 lanes per ray (samples
// staged in LDS), 2 = one ray per lane, 3 = one ray per lane with the wave's samples staged in LDS. Measured on the 800 x 800 occupancy view
// (tools/time_occ_burst.py; 640 000 rays alive for most of it, bursts of 8): form 3 15.8 ms per view, form 1 17.4, form 2 18.2, form 0 18.8
// — with ten waves per SIMD the lanes' lookup chains hide each other and what counts is instructions and how the samples reach memory.
static int rm_burst_form(uint32_t n_step, uint32_t n_alive, bool rederive = false) {
    const int forced = foc_opt(FOC_OPT_OCC_MARCH_FORM);
    // the forms that generate a ray's lattice 16 points ahead (row, the walkers of the two phases) cannot re-derive t sample by sample
    if (rederive && n_step > 1u) return forced == 2 ? 2 : 3;
    if (forced >= 0) return forced;
    (void)n_alive;
    return n_step <= 2u ? 0 : 3;       // (16 lanes per ray for short lists, as foc_march_rays does: 14.7 against 14.0 ms per view — its samples stay ray-major)
}
/* 1 when foc_march_rays_two_phase writes every slot of every list entry for this burst length (the caller need not zero them) */
int foc_march_rays_two_phase_fills(uint32_t n_step, int flags) { const int f = rm_burst_form(n_step, 1u << 30, (flags & 2) != 0); return (f == 1 || f == 3) ? 1 : 0; }
/* 1 when foc_march_rays_two_phase honours flag bit 2 (sample-major output) for this call: the staged one-ray-per-lane kernel does */
int foc_march_rays_two_phase_sample_major(uint32_t n_alive, uint32_t n_step, int flags) { return rm_burst_form(n_step, n_alive, (flags & 2) != 0) == 3 ? 1 : 0; }

/* R9 in two phases (k_march_rays_first + k_march_walkers): `scratch` = int32[n_alive + 4], its first word the worklist length, which the
 * caller has zeroed on this stream. Same arguments and results as foc_march_rays. */
int foc_march_rays_two_phase(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                             const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                             const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                             const float *noises, int32_t *scratch, int normalised, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_alive);
    (void)nears;
    if (n_alive == 0) return FOC_OK;
    FOC_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas && noises && scratch, FOC_E_INVALID,
                "march_rays_two_phase: null pointer");
    FOC_REQUIRE(C >= 1 && C <= 8 && H >= 2 && H <= 512 && max_steps >= 1 && n_step >= 1 && n_step <= 16, FOC_E_INVALID,
                "march_rays_two_phase: unsupported C=%u H=%u max_steps=%u n_step=%u", C, H, max_steps, n_step);
    FOC_REQUIRE((uint64_t)C * H * H * H <= (1ull << 24), FOC_E_INVALID, "march_rays_two_phase: C*H^3 exceeds 2^24");
    RmParams p = rm_make_params(bound, dt_gamma, max_steps, C, H);
    if (normalised & 1) p.norm_inv = 1.0f / (2.0f * bound);    // `xyzs` receives (x + bound) / (2 bound) as torch evaluates it: times the reciprocal (grid.py:149)
    p.rederive = (normalised & 2) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    // A burst of several samples per ray meets an empty cell on most rays (every one of them would be marched twice): the two phases are
    // for bursts of one or two samples. Longer bursts take one launch — one ray per lane with the wave's samples staged in LDS ("staged"),
    // 16 lanes per ray when few rays are left ("row"); FOC_OCC_MARCH_FORM = two | row | lane | staged overrides the choice (A/B runs, tests).
    const int form = rm_burst_form(n_step, n_alive, p.rederive != 0);
    if (form == 1) {
        hipLaunchKernelGGL(p.dt_min <= p.dt_max ? k_march_rays_row<true> : k_march_rays_row<false>, dim3(foc_div_up((uint64_t)n_alive * 16u, 256)), dim3(256), 0, st,
                           n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises);
        FOC_CHECK_LAUNCH("march_rays(row form)");
        return FOC_OK;
    }
    if (form == 3) {
        hipLaunchKernelGGL(k_march_rays_staged, dim3(foc_div_up(n_alive, 64)), dim3(64), 64u * (5u * n_step + 4u) * sizeof(float), st, n_alive, n_step, rays_alive, rays_t,
                           rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises, (normalised & 4) ? 1 : 0);
        FOC_CHECK_LAUNCH("march_rays(staged lane form)");
        return FOC_OK;
    }
    if (form == 2) {
        hipLaunchKernelGGL(k_march_rays, dim3(foc_div_up(n_alive, 64)), dim3(64), 0, st, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, grid, p, fars, xyzs,
                           dirs, deltas, noises);
        FOC_CHECK_LAUNCH("march_rays(lane form)");
        return FOC_OK;
    }
    int32_t *wl_count = scratch, *worklist = scratch + 4;
    hipLaunchKernelGGL(k_march_rays_first, dim3(foc_div_up(n_alive, 256)), dim3(256), 0, st, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, grid, p, fars,
                       xyzs, dirs, deltas, noises, worklist, wl_count);
    FOC_CHECK_LAUNCH("march_rays(first visits)");
    const uint32_t row_max = (uint32_t)max(0, foc_opt(FOC_OPT_MARCH_RAYS_ROW_MAX));
    uint32_t blocks = foc_div_up((uint64_t)n_alive * 16u, 256);
    if (blocks > 4096u) blocks = 4096u;                     // 16 workgroups per CU; longer worklists are walked grid-stride
    hipLaunchKernelGGL(p.dt_min <= p.dt_max ? k_march_walkers<true> : k_march_walkers<false>, dim3(blocks), dim3(256), 0, st, n_alive, n_step, rays_alive, rays_t,
                       rays_o, rays_d, grid, p, fars, xyzs, dirs, deltas, noises, worklist, wl_count, row_max);
    FOC_CHECK_LAUNCH("march_rays(walkers)");
    return FOC_OK;
}

int foc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                       const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth,
                       float *image, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_alive);
    if (n_alive == 0) return FOC_OK;
    FOC_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, FOC_E_INVALID,
                "composite_rays: null pointer");
    rm_launch_composite<false>(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, nullptr, (hipStream_t)stream);
    FOC_CHECK_LAUNCH("composite_rays");
    return FOC_OK;
}

/* foc_composite_rays + the ordered compaction of the survivors into `out` (count in n_out), the compaction's counting pass done by the
 * composite kernel itself: block_counts = int32[n_alive / 1024 + 2], ZEROED by the caller on this stream. (csrc/occrender.hip) */
int foc_composite_compact(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                          const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth,
                          float *image, int32_t *out, int32_t *n_out, int32_t *block_counts, int32_t *deaths, uint32_t deaths_base, uint32_t deaths_len,
                          int sample_major, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_alive);
    FOC_REQUIRE(n_alive > 0 && rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image && out && n_out && block_counts, FOC_E_INVALID,
                "composite_compact: null pointer");
    FOC_REQUIRE(!deaths || deaths_len >= 1, FOC_E_INVALID, "composite_compact: empty death histogram");
    hipStream_t st = (hipStream_t)stream;
    rm_launch_composite<true>(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, block_counts, st,
                              RmDeaths{deaths, deaths_base, deaths ? deaths_len : 1u, sample_major ? 1 : 0});
    FOC_CHECK_LAUNCH("composite_compact(composite)");
    const uint32_t nb = foc_div_up(n_alive, 1024);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, st, block_counts, nb, n_out);
    FOC_CHECK_LAUNCH("composite_compact(scan)");
    hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(1024), 0, st, rays_alive, n_alive, block_counts, out);
    FOC_CHECK_LAUNCH("composite_compact(scatter)");
    return FOC_OK;
}

int foc_compact_alive(const int32_t *rays_alive, uint32_t n_alive, int32_t *out, int32_t *n_out, int32_t *scratch, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_alive);
    FOC_REQUIRE(n_out && scratch, FOC_E_INVALID, "compact_alive: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nb = foc_div_up(n_alive, 1024);
    if (n_alive == 0) { (void)foc_zero_async(n_out, sizeof(int32_t), st); return FOC_OK; }
    FOC_REQUIRE(rays_alive && out, FOC_E_INVALID, "compact_alive: null pointer");
    hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(1024), 0, st, rays_alive, n_alive, scratch);
    FOC_CHECK_LAUNCH("compact_alive(count)");
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, st, scratch, nb, n_out);
    FOC_CHECK_LAUNCH("compact_alive(scan)");
    hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(1024), 0, st, rays_alive, n_alive, scratch, out);
    FOC_CHECK_LAUNCH("compact_alive(scatter)");
    return FOC_OK;
}

} // extern "C"
