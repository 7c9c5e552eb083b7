// mlp_common.h — shared device helpers of the fused MLP kernels (ffmlp.hip; a second translation unit, ffmlp_bwd_priv.hip, lived beside it in round 5: NOTEBOOK.md): MFMA fragment layouts, the
// chained-operand k permutation, weight staging into LDS, colour-head input modes. Semantics: ffmlp/src/ffmlp.cu of the reference
// (see ffmlp.hip for the design notes).
#pragma once
#include "common.h"
#include "activations.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s4 lds_s4;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define MLP_BLOCK 256
#define MLP_WAVES 4
// k_mlp_bwd_fused hands its weight-gradient tiles over in per-workgroup slots: [stage][wave][16 registers][64 lanes] fp32
#define MLP_DW_SLOT_STAGE 4096u        // floats per stage and slot (4 wave tiles of 32 x 32)
#define MLP_DW_MAX_SLOTS 1024u         // cap on the workgroups of a launch (2 per CU)
#ifndef FOC_MLP_SETPRIO
#define FOC_MLP_SETPRIO 1              // issue priority of the MFMA sections of k_mlp_bwd_fused (0 = none: A/B builds, tools/build_variant.sh)
#endif

// neuron (row) index inside a 32-row accumulator tile held in register `reg` by lane-half `h`
// (C/D map of v_mfma_f32_32x32x*: row = (reg&3) + 8*(reg>>2) + 4*h)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// k index carried by element j of lane-half h when an accumulator tile is reused as a B operand
__device__ __forceinline__ int chain_k(int kc, int h, int j) { return 16 * kc + 8 * (j >> 2) + 4 * h + (j & 3); }

__device__ __forceinline__ f16v mfma16(const h8 a, const h8 b, const f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ h8 ld_frag(const _Float16 *lds, uint32_t frag, uint32_t lane) {
    return *reinterpret_cast<const h8 *>(lds + (size_t)frag * 512 + lane * 8);
}

// Build the B fragment of k-chunk (2*mt_prev + s) from accumulator tile `acc` (optionally ReLU'd).
template <bool RELU>
__device__ __forceinline__ h8 acc_to_frag(const f16v &acc, int s) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h8 r;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        h2 p = {(_Float16)acc[8 * s + j], (_Float16)acc[8 * s + j + 1]};       // one v_cvt_pk_f16_f32 (round to nearest even)
        if (RELU) p = __builtin_elementwise_max(p, h2{(_Float16)0, (_Float16)0});  // one v_pk_max_f16 for the pair: max(x, 0), NaN -> 0 like `x > 0 ? x : 0`
        r[j] = p[0]; r[j + 1] = p[1];
    }
    return r;
}

// The same with any of the reference's hidden activations (activations.h): the sum rounded to half, then the function on it.
__device__ __forceinline__ h8 acc_to_frag_act(const f16v &acc, int s, int act) {
    h8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = foc_act_forward((_Float16)acc[8 * s + j], act);
    return r;
}

// ReLU gate on a chained fragment: d where a > 0, else +0. `a` is a post-ReLU activation (never negative, never -0), so "a > 0" is
// "its 16 bits are not zero": min(bits, 1) is a 0/1 factor per half and a packed 16-bit integer multiply applies it to d's bits —
// 2 packed instructions per 2 values instead of a compare and a select per value.
__device__ __forceinline__ h8 relu_gate(const h8 d, const h8 a) {
    const u32x4 db = __builtin_bit_cast(u32x4, d), ab = __builtin_bit_cast(u32x4, a);
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t m, o;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(ab[i]), "v"(0x00010001u));
        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o) : "v"(db[i]), "v"(m));
        r[i] = o;
    }
    return __builtin_bit_cast(h8, r);
}

// Planar network inputs: [in_dim/2][B] dwords (half2), i.e. the hash-grid encoder's native [L, B, C=2] output (gridencoder.cu:218)
// read without the permute to [B, L*C]. Element (row, 16kc + 8h + 2j + {0,1}) lives in plane 8kc + 4h + j; for one j the 32 lanes
// of a lane-half read 128 contiguous bytes.
__device__ __forceinline__ h8 ld_planar8(const _Float16 *__restrict__ base, uint64_t B, uint64_t row, uint32_t kc, int h) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(base) + (uint64_t)(8 * kc + 4 * h) * B + row;
    const u32x4 v = {p[0], p[B], p[2 * B], p[3 * B]};
    return __builtin_bit_cast(h8, v);
}

// Colour-network inputs taken from where they already are (input mode 2, in_dim = 32): the row [SH16(ray) | h[1:16] | 0] the
// reference concatenates per sample (network_ff.py:104-108) is never materialised. k-chunk 0 = the ray's 16 SH values (one 32-byte
// row per ray, shared by its samples); k-chunk 1 = columns 1..16 of the sigma network's output row h [B,16], i.e. the row shifted by
// one half with a zero shifted in at the end — every value sits at the k position it has in the concatenated row, so the
// products and their summation order are those of the materialised form.
struct MlpHead {
    const _Float16 *ray_sh;        // [B / samples_per_ray, 16]
    const _Float16 *grad_h0;       // backward: [B], gradient of h[:,0] (the density path), merged into grad_h column 0
    uint32_t samples_per_ray;
    uint32_t out_width;            // 16: [B,16] outputs / output gradients; 4: only columns 0..3 exist in memory ([B,4]: rgb logits + 1 pad)
    // FOC's object-conditioned colour network (nerf/network_tcnn.py:611-640): the input row is [SH16 | h[1:16] | obj 16 | 0] = 48 wide, and
    // the encoded object feature `obj` [16] is ONE vector for every sample of the launch. W0[:, 31:47] . obj is therefore a constant per
    // neuron: it enters as the initial value of the layer-0 accumulators (obj_bias in LDS) and the k-chunks stay the two of the 32-wide
    // form. Backward: column 31 of the layer-0 input tile is set to 1, so the weight-gradient MFMAs deliver the column sum of delta_0 in
    // dW0[:, 31]; dW0[:, 31:47] = colsum (x) obj and grad_obj = W0[:, 31:47]^T colsum follow in the finalize kernel. W0 rows are 48 wide.
    const _Float16 *obj;           // [16] or null
};
#define HEAD_OBJ_LD 48u
__device__ __forceinline__ uint32_t head_ld0(const MlpHead &hd) { return hd.obj ? HEAD_OBJ_LD : 32u; }

// obj_bias[mt][h][reg] (fp32, accumulator-register order of acc_row) = sum_j W0[n][31 + j] * obj[j], n = 32 mt + acc_row(reg, h):
// 64 threads, sequential fmaf in j order. `W0` has HEAD_OBJ_LD-wide rows.
__device__ __forceinline__ void stage_obj_bias(const _Float16 *__restrict__ W0, const _Float16 *__restrict__ obj, float *bias, uint32_t hidden) {
    if (threadIdx.x < hidden) {
        const uint32_t n = threadIdx.x, r = n & 31u, mt = n >> 5;
        const uint32_t h = (r >> 2) & 1u, reg = (r & 3u) + 4u * (r >> 3);
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) a = fmaf((float)W0[(size_t)n * HEAD_OBJ_LD + 31 + j], (float)obj[j], a);
        bias[(mt * 2 + h) * 16 + reg] = a;
    }
}
__device__ __forceinline__ f16v ld_obj_bias(const float *bias, int mt, int h) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 *p = reinterpret_cast<const f4 *>(bias + (mt * 2 + h) * 16);
    const f4 a = p[0], b = p[1], c = p[2], d = p[3];
    return f16v{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
}
// the loads and the shift are separate so that a prefetching caller can keep the raw dwords in flight
__device__ __forceinline__ void ld_head_raw(const _Float16 *__restrict__ hrows, uint64_t row, int h, u32x4 &v, uint32_t &nxt) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(hrows + row * 16) + 4 * h;
    v = *reinterpret_cast<const u32x4 *>(p);
    nxt = h == 0 ? p[4] : 0u;
}
// prefetching form: the dword after the lane's 16 bytes is loaded by EVERY lane (upper lane-half: a dword of its own 16 bytes, dropped by
// the caller when it shifts) so that the load sits in no exec-masked branch
__device__ __forceinline__ void ld_head_raw_all(const _Float16 *__restrict__ hrows, uint64_t row, int h, u32x4 &v, uint32_t &nxt) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(hrows + row * 16) + 4 * h;
    v = *reinterpret_cast<const u32x4 *>(p);
    nxt = p[h == 0 ? 4 : 3];
}
__device__ __forceinline__ h8 head_shift(const u32x4 v, uint32_t nxt) {
    const u32x4 r = {__builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16), __builtin_amdgcn_alignbit(v.w, v.z, 16),
                     __builtin_amdgcn_alignbit(nxt, v.w, 16)};
    return __builtin_bit_cast(h8, r);
}
__device__ __forceinline__ h8 ld_head8(const _Float16 *__restrict__ hrows, const MlpHead &hd, uint64_t row, uint32_t kc, int h) {
    if (kc == 0) return *reinterpret_cast<const h8 *>(hd.ray_sh + (uint64_t)((uint32_t)row / hd.samples_per_ray) * 16 + 8 * h);
    u32x4 v; uint32_t nxt;
    ld_head_raw_all(hrows, row, h, v, nxt);
    return head_shift(v, h == 0 ? nxt : 0u);
}

// ---------------------------------------------------------------- weight staging
// Forward image. Fragment f holds, for lane (r = lane&31, h = lane>>5), the 8 halfs
//   layer 0      : W0[32*mt + r][16*kc + 8*h + j]                       (natural k: B comes from global inputs)
//   hidden l>=1  : Wl[32*mt + r][chain_k(kc, h, j)]
//   output layer : Wout[r][chain_k(kc, h, j)]  for r < 16, else 0
// Fragment order: layer0 [mt][kc0] | hidden layers [l][mt][kc] | out [kc].
// ld0 = 0: W0 rows are in_dim wide and all in_dim / 16 k-chunks are staged; ld0 > in_dim (head with an object feature): rows are ld0 wide,
// the first in_dim / 16 chunks are staged
template <int HIDDEN>
__device__ void stage_weights_fwd(const _Float16 *__restrict__ W, _Float16 *lds, uint32_t in_dim, uint32_t num_layers, bool with_out = true, uint32_t ld0 = 0) {
    constexpr int MT = (HIDDEN + 31) / 32, KC = HIDDEN / 16;
    const uint32_t KS0 = in_dim / 16;
    if (!ld0) ld0 = in_dim;
    const uint32_t n0 = MT * KS0, nh = (num_layers - 1) * MT * KC, total = n0 + nh + (with_out ? KC : 0);
    const _Float16 *Wh = W + (size_t)HIDDEN * ld0;
    const _Float16 *Wo = Wh + (size_t)(num_layers - 1) * HIDDEN * HIDDEN;
    for (uint32_t idx = threadIdx.x; idx < total * 64; idx += MLP_BLOCK) {
        const uint32_t f = idx >> 6, lane = idx & 63, r = lane & 31, h = lane >> 5;
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (_Float16)0;
        if (f < n0) {
            const uint32_t mt = f / KS0, kc = f % KS0, row = 32 * mt + r;
            if (row < HIDDEN) v = *reinterpret_cast<const h8 *>(W + (size_t)row * ld0 + 16 * kc + 8 * h);
        } else if (f < n0 + nh) {
            const uint32_t g = f - n0, l = g / (MT * KC), mt = (g / KC) % MT, kc = g % KC, row = 32 * mt + r;
            if (row < HIDDEN) {
                const _Float16 *p = Wh + (size_t)l * HIDDEN * HIDDEN + (size_t)row * HIDDEN + 16 * kc + 4 * h;
                const h4 lo = *reinterpret_cast<const h4 *>(p), hi = *reinterpret_cast<const h4 *>(p + 8);
#pragma unroll
                for (int j = 0; j < 4; j++) { v[j] = lo[j]; v[4 + j] = hi[j]; }
            }
        } else {
            const uint32_t kc = f - n0 - nh;
            if (r < 16) {
                const _Float16 *p = Wo + (size_t)r * HIDDEN + 16 * kc + 4 * h;
                const h4 lo = *reinterpret_cast<const h4 *>(p), hi = *reinterpret_cast<const h4 *>(p + 8);
#pragma unroll
                for (int j = 0; j < 4; j++) { v[j] = lo[j]; v[4 + j] = hi[j]; }
            }
        }
        *reinterpret_cast<h8 *>(lds + (size_t)f * 512 + lane * 8) = v;
    }
}

// Backward image (transposed weights: rows = INPUT neuron i of the layer, k = OUTPUT neuron o).
//   out layer   [mt]        : Wout[8*h + j][32*mt + r]                (k = o natural, K = 16; B = grad from global)
//   hidden l    [l][mt][kc] : Wl[chain_k(kc,h,j)][32*mt + r]          (l = 0 .. num_layers-2, matrix l maps fwd[l] -> fwd[l+1])
//   dX          [mt0][kc]   : W0[chain_k(kc,h,j)][32*mt0 + r]  (i < in_dim, else 0)
// Fragment order: out [mt] | hidden [l][mt][kc] | dX [mt0][kc].
template <int HIDDEN>
__device__ void stage_weights_bwd(const _Float16 *__restrict__ W, _Float16 *lds, uint32_t in_dim, uint32_t num_layers, bool with_dx, bool head = false,
                                  uint32_t ld0 = 0) {
    constexpr int MT = (HIDDEN + 31) / 32, KC = HIDDEN / 16;
    const uint32_t MT0 = (in_dim + 31) / 32;
    if (!ld0) ld0 = in_dim;
    const uint32_t no = MT, nh = (num_layers - 1) * MT * KC, nx = with_dx ? MT0 * KC : 0, total = no + nh + nx;
    const _Float16 *Wh = W + (size_t)HIDDEN * ld0;
    const _Float16 *Wo = Wh + (size_t)(num_layers - 1) * HIDDEN * HIDDEN;
    for (uint32_t idx = threadIdx.x; idx < total * 64; idx += MLP_BLOCK) {
        const uint32_t f = idx >> 6, lane = idx & 63, r = lane & 31, h = lane >> 5;
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (_Float16)0;
        if (f < no) {
            const uint32_t i = 32 * f + r;
            if (i < HIDDEN) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = Wo[(size_t)(8 * h + j) * HIDDEN + i];
            }
        } else if (f < no + nh) {
            const uint32_t g = f - no, l = g / (MT * KC), mt = (g / KC) % MT, kc = g % KC, i = 32 * mt + r;
            if (i < HIDDEN) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = Wh[(size_t)l * HIDDEN * HIDDEN + (size_t)chain_k(kc, h, j) * HIDDEN + i];
            }
        } else {
            const uint32_t g = f - no - nh, mt0 = g / KC, kc = g % KC, i = 32 * mt0 + r;
            if (head) {
                // input mode 2: result row 16 + k is the gradient of h column k = input column 15 + k (k = 1..15); rows 0..16 are not used
                if (i >= 17 && i < 32) {
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = W[(size_t)chain_k(kc, h, j) * ld0 + (i - 1)];
                }
            } else if (i < in_dim) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = W[(size_t)chain_k(kc, h, j) * ld0 + i];
            }
        }
        *reinterpret_cast<h8 *>(lds + (size_t)f * 512 + lane * 8) = v;
    }
}

// Store one accumulator tile (32 neurons x 32 samples) as fp16 into a row-major [B, ld] buffer:
// lane (c, h) owns sample `row0 + c` and, per register quad q, the 4 consecutive neurons
// col0 + 8q + 4h .. +3  -> one 8-byte store per quad.
template <bool RELU>
__device__ __forceinline__ void store_tile(_Float16 *__restrict__ dst, uint32_t ld, uint64_t row, uint64_t nrows, uint32_t col0, uint32_t ncols,
                                           const f16v &acc, int h) {
    if (row >= nrows) return;          // ragged last tile: rows past B are computed on clamped inputs and dropped
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t col = col0 + 8 * q + 4 * h;
        if (col < ncols) {
            h4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                _Float16 x = (_Float16)acc[4 * q + e];
                if (RELU) x = x > (_Float16)0 ? x : (_Float16)0;
                v[e] = x;
            }
            *reinterpret_cast<h4 *>(dst + row * ld + col) = v;
        }
    }
}

// store_tile with a general activation on the half-rounded sums
__device__ __forceinline__ void store_tile_act(_Float16 *__restrict__ dst, uint32_t ld, uint64_t row, uint64_t nrows, uint32_t col0, uint32_t ncols,
                                               const f16v &acc, int h, int act) {
    if (row >= nrows) return;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t col = col0 + 8 * q + 4 * h;
        if (col < ncols) {
            h4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = foc_act_forward((_Float16)acc[4 * q + e], act);
            *reinterpret_cast<h4 *>(dst + row * ld + col) = v;
        }
    }
}
