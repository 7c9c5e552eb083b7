// gridencoder.hip — multiresolution hash / tiled grid encoding for gfx950.
//
// Semantics: gridencoder/src/gridencoder.cu of the reference (kernel_grid :87-245,
// kernel_grid_backward :248-340, kernel_input_backward :343-369, kernel_grad_tv :506-610).
//
// MI355X organisation:
//  * Level placement is XCD-aware. Workgroups are dealt round-robin over the 8 XCDs
//    (block b and b+8 share one), each XCD has a private 4 MiB L2, and one hashed level
//    of the default table is 2^19 rows x 2 ch = 2 MiB (fp16) / 4 MiB (fp32). The 1-D grid
//    is therefore decoded as  xcd = id % 8, level = xcd + 8 * (j / chunks), so a level's
//    gathers all come from ONE XCD's L2 and stay resident there, instead of the reference's
//    blockIdx.y = level (gridencoder.cu:103) which spreads every level over all eight L2s.
//    This is a speed choice only: any placement gives the same result.
//  * Per-level scale / resolution are computed on the host (same libm expression as the
//    oracle) and passed by value: no exp2f per thread, and the integer index math cannot
//    drift with a device transcendental.
//  * The [B, L*C] output variant assigns the L levels of one point to adjacent lanes, so a
//    wave stores 256 contiguous bytes — the reference writes [L,B,C] and pays a permute copy
//    in Python (grid.py:57).
//  * Backward, D = 3 / C = 2 tables (every FOC network): NO scattered atomics — samples are partitioned into 8192-row segments per
//    level (k_gbin_count or the counting workgroups of k_grid_fwd_counted -> k_gbin_scans -> k_gbin_scatter_pms writing two-corner /
//    factored records) and each segment is summed on chip and written once (k_gbin_reduce); DESIGN.md section 4, derivation and measured
//    variants in NOTEBOOK.md. Other shapes (k_grid_bwd): one packed global_atomic_pk_add_f16 per corner for fp16 C = 2 tables,
//    global_atomic_add_f32 for fp32; zero gradients are not issued (adding +-0 is a no-op).
// Compiled with -ffp-contract=off; explicit fmaf() mirrors oracle/oracle.c.
#include "common.h"
#include <math.h>
#include <stdlib.h>

#include "ge_common.h"

// ---- index math (gridencoder.cu:50-84) -------------------------------------------------
template <uint32_t D>
__device__ __forceinline__ uint32_t ge_index(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution,
                                             const uint32_t (&pos_grid)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pos_grid[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t result = 0;
#pragma unroll
        for (uint32_t i = 0; i < D; i++) result ^= pos_grid[i] * primes[i];
        index = result;
    }
    // `index % hashmap_size` (gridencoder.cu:83) without the integer division in the common cases: dense levels have
    // index < size already, hashed levels have a power-of-two size (2^log2_hashmap_size); the generic path remains for
    // tiled grids / odd sizes. A runtime u32 modulo is ~35 VALU instructions and this runs 8x per (point, level).
    if (index >= hashmap_size) index = ((hashmap_size & (hashmap_size - 1u)) == 0u) ? (index & (hashmap_size - 1u)) : (index % hashmap_size);
    return index;                  // row index; caller multiplies by C
}

// Decode a linear block id into (level, chunk) so that all blocks of a level share
// blockIdx % 8, i.e. one XCD under the observed round-robin dispatch.
__device__ __forceinline__ bool ge_decode_block(uint32_t id, uint32_t chunks, uint32_t L, uint32_t &level, uint32_t &chunk) {
    const uint32_t xcd = id & 7u, j = id >> 3;
    level = xcd + 8u * (j / chunks);
    chunk = j % chunks;
    return level < L;
}

template <uint32_t D>
__device__ __forceinline__ bool ge_load_point(const float *__restrict__ inputs, uint32_t b, float (&x)[D]) {
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) { x[d] = inputs[(uint64_t)b * D + d]; oob |= (x[d] < 0 || x[d] > 1); }
    return oob;
}

// The same 8 rows for D = 3 with the per-level decisions (which dimensions enter the dense index, dense or hashed, power-of-two
// size) taken once on wave-uniform values and the per-axis terms shared between the corners: (p+1)*k = p*k + k in uint32, so every
// row is two adds or two xors instead of ge_index's per-corner multiplies and branches. Bit-identical to ge_index per corner (corner idx: bit d set = +1 along axis d).
__device__ __forceinline__ void ge_rows3(const uint32_t (&pos_grid)[3], uint32_t hashmap_size, uint32_t resolution, uint32_t gridtype,
                                              bool align_corners, uint32_t (&rows)[8], bool *is_hashed = nullptr) {
    const uint32_t r1 = align_corners ? resolution : resolution + 1u;
    uint32_t stride = 1u, st[3];
    bool part[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { part[d] = stride <= hashmap_size; st[d] = part[d] ? stride : 0u; if (part[d]) stride *= r1; }
    const bool hashed = gridtype == 0u && stride > hashmap_size;
    if (is_hashed) *is_hashed = hashed;
    uint32_t t[3][2];
    if (hashed) {
        constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};
#pragma unroll
        for (int d = 0; d < 3; d++) { t[d][0] = pos_grid[d] * primes[d]; t[d][1] = t[d][0] + primes[d]; }
    } else {
#pragma unroll
        for (int d = 0; d < 3; d++) { t[d][0] = pos_grid[d] * st[d]; t[d][1] = t[d][0] + st[d]; }
    }
    // `index % hashmap_size` (gridencoder.cu:83), decided once per level: a dense level that takes all three axes has index < size
    // already; a power-of-two size (every hashed level: 2^log2_hashmap_size) is a mask; the division is left for tiled grids with odd sizes
    const bool need_mod = hashed || stride > hashmap_size || !part[2];
    const bool pow2 = (hashmap_size & (hashmap_size - 1u)) == 0u;
    if (!need_mod) {
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) rows[idx] = t[0][idx & 1u] + t[1][(idx >> 1) & 1u] + t[2][(idx >> 2) & 1u];
    } else if (pow2) {
        const uint32_t mask = hashmap_size - 1u;
        if (hashed) {
#pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) rows[idx] = (t[0][idx & 1u] ^ t[1][(idx >> 1) & 1u] ^ t[2][(idx >> 2) & 1u]) & mask;
        } else {
#pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) rows[idx] = (t[0][idx & 1u] + t[1][(idx >> 1) & 1u] + t[2][(idx >> 2) & 1u]) & mask;
        }
    } else {
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            const uint32_t a = t[0][idx & 1u], b = t[1][(idx >> 1) & 1u], c = t[2][(idx >> 2) & 1u];
            rows[idx] = (hashed ? (a ^ b ^ c) : (a + b + c)) % hashmap_size;
        }
    }
}

// ---- forward: one (point, level) ---------------------------------------------------------
// out points at the C outputs of this (point, level); dy points at its [D,C] block or null.
template <typename T, uint32_t D, uint32_t C, bool PAIRS = false>
__device__ __forceinline__ void ge_forward_one(const float (&x)[D], bool oob, const T *__restrict__ table, uint32_t hashmap_size,
                                               float scale, uint32_t resolution, T *__restrict__ out, T *__restrict__ dy,
                                               uint32_t gridtype, bool align_corners, uint32_t interp) {
    float results[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) results[c] = 0.0f;
    if (oob) {                                                   // gridencoder.cu:119-135
        GeVec<T, C>::st(out, results);
        if (dy) {
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) GeT<T>::st(dy + i, 0.0f);
        }
        return;
    }
    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {                           // :147-159
        pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pos_grid[d] = (uint32_t)fl;
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) {
            const float v = pos[d];
            pos_deriv[d] = 6 * v * (1.0f - v);
            pos[d] = v * v * fmaf(-2.0f, v, 3.0f);
        } else pos_deriv[d] = 1.0f;
    }
    // issue all 2^D gathers before consuming them (latency-bound: keep them in flight)
    float vals[1u << D][C];
    float ws[1u << D];
    uint32_t rows3[8];
    if constexpr (D == 3) ge_rows3(pos_grid, hashmap_size, resolution, gridtype, align_corners, rows3);
    constexpr bool paired = PAIRS && D == 3 && C == 2 && sizeof(T) == 2;
    if constexpr (paired) {
        // The two corners along x of a (y, z) pair are ROW NEIGHBOURS whenever row(x+1) == row(x) ^ 1: always on a dense level with an
        // even row, and on a hashed level for every even x (the hash xors x in with prime 1, so x -> x+1 flips bit 0 only). One
        // 8-byte load of the aligned row pair then serves both corners — the gathers are bound by the number of distinct cache-line
        // requests, not by bytes (tools/bench_gather.hip) — and only the other lanes issue the second, 4-byte load.
        {
            const uint32_t *tw = reinterpret_cast<const uint32_t *>(table);
            uint2 wide[4];
            uint32_t nar[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wide[j] = *reinterpret_cast<const uint2 *>(tw + (rows3[2 * j] & ~1u));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                nar[j] = 0u;
                if (rows3[2 * j + 1] != (rows3[2 * j] ^ 1u)) nar[j] = tw[rows3[2 * j + 1]];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool odd = (rows3[2 * j] & 1u) != 0u, pr = rows3[2 * j + 1] == (rows3[2 * j] ^ 1u);
                const uint32_t v0 = odd ? wide[j].y : wide[j].x;
                const uint32_t v1 = pr ? (odd ? wide[j].x : wide[j].y) : nar[j];
                const __half2 h0 = *reinterpret_cast<const __half2 *>(&v0), h1 = *reinterpret_cast<const __half2 *>(&v1);
                vals[2 * j][0] = __low2float(h0); vals[2 * j][1] = __high2float(h0);
                vals[2 * j + 1][0] = __low2float(h1); vals[2 * j + 1][1] = __high2float(h1);
            }
        }
    }
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {             // :167-191
        float w = 1;
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
        }
        ws[idx] = w;
        if constexpr (!paired) {
            uint32_t row;
            if constexpr (D == 3) row = rows3[idx];
            else row = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
            GeVec<T, C>::ld(table + (uint64_t)row * C, vals[idx]);
        }
    }
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
#pragma unroll
        for (uint32_t c = 0; c < C; c++) results[c] = fmaf(ws[idx], vals[idx][c], results[c]);
    }
    GeVec<T, C>::st(out, results);

    if (dy) {                                                    // :201-244
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) rg[c] = 0.0f;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float w = scale;
                uint32_t pgl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                pgl[gd] = pos_grid[gd];
                const uint32_t rl = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
                pgl[gd] = pos_grid[gd] + 1;
                const uint32_t rr = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
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

// ---- forward, the shape FOC runs: hash gridtype, align_corners off, linear interpolation, fp16 tables with C = 2, D = 3, no dy_dx, row pairs
// 8-byte aligned. ge_forward_one with every per-call decision already taken: what is left at run time is dense-or-hashed (wave-uniform,
// decided by the caller with ge_level_hashed) and, on a hashed level, ONE lane-divergent branch (x odd: the four x+1 corners are no row
// neighbours and take their own 4-byte loads) instead of four. Same arithmetic in the same order: bit-identical results.
__host__ __device__ __forceinline__ bool ge_level_hashed(uint32_t hashmap_size, uint32_t resolution, uint32_t &st1, uint32_t &st2) {
    const uint32_t r1 = resolution + 1u;                 // the reference's uint32 stride walk (gridencoder.cu:70-83), D = 3
    uint32_t stride = 1u;
    st1 = st2 = 0u;
    if (stride <= hashmap_size) stride *= r1;
    if (stride <= hashmap_size) { st1 = stride; stride *= r1; }
    if (stride <= hashmap_size) { st2 = stride; stride *= r1; }
    return stride > hashmap_size;
}

__device__ __forceinline__ void ge_forward_hash3(const float (&x)[3], bool oob, const uint32_t *__restrict__ tw, uint32_t hashmap_size, bool hashed,
                                                 uint32_t st1, uint32_t st2, float scale, __half *__restrict__ out) {
    if (oob) { *reinterpret_cast<uint32_t *>(out) = 0u; return; }
    float pos[3];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float p = fmaf(x[d], scale, 0.5f);         // >= 0.5 here (0 <= x <= 1): truncation is the floor, and p - floor(p) is exact —
        pg[d] = (uint32_t)p;                             // v_fract_f32 returns that same value in one instruction
        pos[d] = __builtin_amdgcn_fractf(p);
    }
    // rows as 32-bit BYTE offsets from the level's base: the loads take the base from scalar registers (global_load ... v_off, s[base])
    // instead of a 64-bit address per lane and load (the launch wrapper admits this path for levels below 2^30 rows only)
    const char *tb = reinterpret_cast<const char *>(tw);
    // b0[j]: BYTE offset of the row of corner (x, y_j, z_j), j = y bit | z bit << 1; the row pair it belongs to starts at b0 & ~7.
    uint32_t b0[4];
    uint2 wide[4];
    uint32_t nar[4] = {0u, 0u, 0u, 0u};
    bool pr[4];
    if (hashmap_size <= (1u << 22)) {
        // Tables of at most 2^22 rows (every NeRF grid: log2_hashmap_size 19): the index arithmetic runs on byte offsets below 2^24 with
        // FULL-RATE 24-bit multiplies. Only the low bits of the hash survive the `% hashmap_size` (a power of two here), and they depend on
        // the low bits of the factors alone: (x ^ y P1 ^ z P2) & m, times 4, == ((x << 2) ^ y (4 (P1 & m)) ^ z (4 (P2 & m))) & (m << 2). The
        // generic form below spends two quarter-rate 32-bit multiplies, a shift and a mask per corner; this one a shift for x, two
        // v_mul_u32_u24 and one xor + and per corner (~20 of the ~100 VALU slots per point and level). A dense level likewise: strides
        // below 2^22, rows below the table size.
        const uint32_t x4 = pg[0] << 2;
        uint32_t yz4[4];
        if (hashed) {
            const uint32_t m = hashmap_size - 1u, m4 = m << 2;
            const uint32_t p1 = (2654435761u & m) << 2, p2 = (805459861u & m) << 2;
            const uint32_t t1 = __umul24(pg[1], p1), t2 = __umul24(pg[2], p2);
            yz4[0] = t1 ^ t2; yz4[1] = (t1 + p1) ^ t2; yz4[2] = t1 ^ (t2 + p2); yz4[3] = (t1 + p1) ^ (t2 + p2);
#pragma unroll
            for (int j = 0; j < 4; j++) b0[j] = (x4 ^ yz4[j]) & m4;
#pragma unroll
            for (int j = 0; j < 4; j++) wide[j] = *reinterpret_cast<const uint2 *>(tb + (b0[j] & ~7u));
            const bool even = (x4 & 4u) == 0u;           // x even: x + 1 == x ^ 1, every (x, x+1) corner pair is an aligned row pair
            if (!even) {
#pragma unroll
                for (int j = 0; j < 4; j++) nar[j] = *reinterpret_cast<const uint32_t *>(tb + (((x4 + 4u) ^ yz4[j]) & m4));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) pr[j] = even;
        } else {
            const uint32_t s1 = st1 << 2, s2 = st2 << 2;
            const uint32_t t1 = __umul24(pg[1], s1), t2 = __umul24(pg[2], s2);
            yz4[0] = t1 + t2; yz4[1] = (t1 + s1) + t2; yz4[2] = t1 + (t2 + s2); yz4[3] = (t1 + s1) + (t2 + s2);
#pragma unroll
            for (int j = 0; j < 4; j++) b0[j] = x4 + yz4[j];
#pragma unroll
            for (int j = 0; j < 4; j++) wide[j] = *reinterpret_cast<const uint2 *>(tb + (b0[j] & ~7u));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                pr[j] = (b0[j] & 4u) == 0u;
                if (!pr[j]) nar[j] = *reinterpret_cast<const uint32_t *>(tb + (b0[j] + 4u));
            }
        }
    } else if (hashed) {
        const uint32_t mask = hashmap_size - 1u;
        const uint32_t t1 = pg[1] * 2654435761u, t2 = pg[2] * 805459861u;
        const uint32_t yz[4] = {t1 ^ t2, (t1 + 2654435761u) ^ t2, t1 ^ (t2 + 805459861u), (t1 + 2654435761u) ^ (t2 + 805459861u)};
#pragma unroll
        for (int j = 0; j < 4; j++) b0[j] = ((pg[0] ^ yz[j]) & mask) << 2;
#pragma unroll
        for (int j = 0; j < 4; j++) wide[j] = *reinterpret_cast<const uint2 *>(tb + (b0[j] & ~7u));
        const bool even = (pg[0] & 1u) == 0u;
        if (!even) {
#pragma unroll
            for (int j = 0; j < 4; j++) nar[j] = *reinterpret_cast<const uint32_t *>(tb + ((((pg[0] + 1u) ^ yz[j]) & mask) << 2));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) pr[j] = even;
    } else {
        const uint32_t t1 = pg[1] * st1, t2 = pg[2] * st2;
        const uint32_t yz[4] = {t1 + t2, (t1 + st1) + t2, t1 + (t2 + st2), (t1 + st1) + (t2 + st2)};
#pragma unroll
        for (int j = 0; j < 4; j++) b0[j] = (pg[0] + yz[j]) << 2;
#pragma unroll
        for (int j = 0; j < 4; j++) wide[j] = *reinterpret_cast<const uint2 *>(tb + (b0[j] & ~7u));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            pr[j] = (b0[j] & 4u) == 0u;
            if (!pr[j]) nar[j] = *reinterpret_cast<const uint32_t *>(tb + (b0[j] + 4u));
        }
    }
    const float wx[2] = {1 - pos[0], pos[0]}, wy[2] = {1 - pos[1], pos[1]}, wz[2] = {1 - pos[2], pos[2]};
    float res0 = 0.0f, res1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool odd = (b0[j] & 4u) != 0u;
        const uint32_t v0 = odd ? wide[j].y : wide[j].x;
        const uint32_t v1 = pr[j] ? (odd ? wide[j].x : wide[j].y) : nar[j];
        const __half2 h0 = *reinterpret_cast<const __half2 *>(&v0), h1 = *reinterpret_cast<const __half2 *>(&v1);
        const float w0 = (wx[0] * wy[j & 1]) * wz[j >> 1], w1 = (wx[1] * wy[j & 1]) * wz[j >> 1];
        res0 = fmaf(w0, __low2float(h0), res0); res1 = fmaf(w0, __high2float(h0), res1);
        res0 = fmaf(w1, __low2float(h1), res0); res1 = fmaf(w1, __high2float(h1), res1);
    }
    *reinterpret_cast<__half2 *>(out) = __halves2half2(__float2half_rn(ge_opaque(res0)), __float2half_rn(ge_opaque(res1)));
}

// One level of the fast shape or of the general one (`fast`: the launch-wide part of the decision, taken on the host).
// ge_forward_hash3 addresses rows by 32-bit BYTE offsets inside the level: levels of up to 2^30 rows of 4 bytes; a hashed level must
// have a power-of-two size (its `% hashmap_size` is a mask there). Within it, levels of at most 2^22 rows take the 24-bit multiplies.
__host__ __device__ __forceinline__ bool ge_hash3_admits(uint32_t hashmap_size, bool hashed) {
    return (!hashed || (hashmap_size & (hashmap_size - 1u)) == 0u) && hashmap_size <= (1u << 30);
}

template <typename T, uint32_t D, uint32_t C>
__device__ __forceinline__ void ge_forward_level(const float (&x)[D], bool oob, const T *__restrict__ grid, uint32_t off0, uint32_t hashmap_size, float scale,
                                                 uint32_t resolution, T *__restrict__ out, T *__restrict__ dy, uint32_t gridtype, bool align_corners,
                                                 uint32_t interp, uint32_t pairs) {
    const bool aligned = ((off0 | hashmap_size) & 1u) == 0u;                   // row pairs of this level are 8-byte aligned and inside it
    if constexpr (sizeof(T) == 2 && D == 3 && C == 2) {
        if (pairs == 2u && aligned) {
            uint32_t st1, st2;
            const bool hashed = ge_level_hashed(hashmap_size, resolution, st1, st2);
            if (ge_hash3_admits(hashmap_size, hashed)) {
                ge_forward_hash3(x, oob, reinterpret_cast<const uint32_t *>(grid) + off0, hashmap_size, hashed, st1, st2, scale,
                                 reinterpret_cast<__half *>(out));
                return;
            }
        }
    }
    if (pairs && aligned)
        ge_forward_one<T, D, C, true>(x, oob, grid + (uint64_t)off0 * C, hashmap_size, scale, resolution, out, dy, gridtype, align_corners, interp);
    else
        ge_forward_one<T, D, C, false>(x, oob, grid + (uint64_t)off0 * C, hashmap_size, scale, resolution, out, dy, gridtype, align_corners, interp);
}

// Encoding workgroup f of a plain level walk -> chunk of 256 points and the range of levels it encodes: the `lc` leading levels (the
// small dense tables, 1.4 MiB together for the default grid) are done by ONE workgroup per chunk, which loads its points once; the
// levels from `lc` up follow one at a time. At those small levels a launch moves 12 bytes of position per 4 bytes of result: what
// they cost is reading the positions (0.015 ms per level and 2 M points), not the gathers.
__device__ __forceinline__ void ge_walk(uint32_t f, uint32_t chunks, uint32_t lc, uint32_t &chunk, uint32_t &l0, uint32_t &l1) {
    const uint32_t grp = f / chunks;
    chunk = f - grp * chunks;
    if (lc < 2u) { l0 = grp; l1 = grp + 1u; }
    else if (grp == 0u) { l0 = 0u; l1 = lc; }
    else { l0 = lc + grp - 1u; l1 = l0 + 1u; }
}

// Level-major launch, outputs [L,B,C]: thread = point. `plain`: blocks are numbered level by level, so the whole chip walks ONE
// level at a time and every XCD's 4 MiB L2 holds that level's table (<= 2 MiB at 2^19 x half2); otherwise a level is pinned to one
// XCD (ge_decode_block), which balances badly because the cost of a level grows ~5x from level 0 to 15 (tools/time_levels.py).
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_fwd_lbc(const float *__restrict__ inputs, const T *__restrict__ grid,
                                                      const int32_t *__restrict__ offsets, T *__restrict__ outputs,
                                                      uint32_t B, uint32_t L, GeLevels lv, T *__restrict__ dy_dx,
                                                      uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t chunks, uint32_t plain,
                                                      uint32_t pairs, uint32_t lc) {
    uint32_t chunk, l0, l1;
    if (plain) ge_walk(blockIdx.x, chunks, lc, chunk, l0, l1);
    else { if (!ge_decode_block(blockIdx.x, chunks, L, l0, chunk)) return; l1 = l0 + 1u; }
    const uint32_t b = chunk * 256 + threadIdx.x;
    if (b >= B || l0 >= L) return;
    float x[D];
    const bool oob = ge_load_point<D>(inputs, b, x);
    for (uint32_t level = l0; level < l1; level++) {
        const uint32_t off0 = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
        T *dy = dy_dx ? dy_dx + ((uint64_t)b * L + level) * D * C : nullptr;
        ge_forward_level<T, D, C>(x, oob, grid, off0, hashmap_size, lv.scale[level], lv.resolution[level], outputs + ((uint64_t)level * B + b) * C, dy,
                                  gridtype, align_corners, interp, pairs);
    }
}

// Point-major launch, outputs [B, L*C]: consecutive lanes = consecutive levels of one point, so
// the wave's store is contiguous. Thread id g -> b = g / L, level = g % L.
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_fwd_bl(const float *__restrict__ inputs, const T *__restrict__ grid,
                                                     const int32_t *__restrict__ offsets, T *__restrict__ outputs,
                                                     uint32_t B, uint32_t L, GeLevels lv, T *__restrict__ dy_dx,
                                                     uint32_t gridtype, bool align_corners, uint32_t interp) {
    const uint64_t total = (uint64_t)B * L;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (uint64_t)gridDim.x * 256) {
        const uint32_t b = (uint32_t)(g / L), level = (uint32_t)(g - (uint64_t)b * L);
        const uint32_t off0 = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
        float x[D];
        const bool oob = ge_load_point<D>(inputs, b, x);
        T *dy = dy_dx ? dy_dx + ((uint64_t)b * L + level) * D * C : nullptr;
        ge_forward_one<T, D, C>(x, oob, grid + (uint64_t)off0 * C, hashmap_size, lv.scale[level], lv.resolution[level],
                                outputs + g * C, dy, gridtype, align_corners, interp);
    }
}

// ---- backward ---------------------------------------------------------------------------
template <typename T, uint32_t D, uint32_t C>
__device__ __forceinline__ void ge_backward_one(const float (&x)[D], const float (&g)[C], T *__restrict__ grad_table,
                                                uint32_t hashmap_size, float scale, uint32_t resolution,
                                                uint32_t gridtype, bool align_corners, uint32_t interp) {
    float pos[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) { const float v = pos[d]; pos[d] = v * v * fmaf(-2.0f, v, 3.0f); }
    }
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
        }
        const uint32_t row = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
        float v[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) v[c] = w * g[c];
        GeAtomic<T>::template add<C>(grad_table + (uint64_t)row * C, v);
    }
}

// grad layout: GRAD_BL ? [B, L*C] : [L, B, C].  Level-major XCD-aware launch, thread = point.
template <typename T, uint32_t D, uint32_t C, bool GRAD_BL>
__global__ void __launch_bounds__(256) k_grid_bwd(const T *__restrict__ grad, const float *__restrict__ inputs,
                                                  const int32_t *__restrict__ offsets, T *__restrict__ grad_grid,
                                                  uint32_t B, uint32_t L, GeLevels lv, uint32_t gridtype, bool align_corners,
                                                  uint32_t interp, uint32_t chunks) {
    uint32_t level, chunk;
    if (!ge_decode_block(blockIdx.x, chunks, L, level, chunk)) return;
    const uint32_t b = chunk * 256 + threadIdx.x;
    if (b >= B) return;
    float x[D];
    if (ge_load_point<D>(inputs, b, x)) return;                   // :276-281
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    float g[C];
    const T *gp = GRAD_BL ? grad + ((uint64_t)b * L + level) * C : grad + ((uint64_t)level * B + b) * C;
    GeVec<T, C>::ld(gp, g);
    bool any = false;
#pragma unroll
    for (uint32_t c = 0; c < C; c++) any |= (g[c] != 0.0f);
    if (!any) return;
    ge_backward_one<T, D, C>(x, g, grad_grid + (uint64_t)off0 * C, hashmap_size, lv.scale[level], lv.resolution[level],
                             gridtype, align_corners, interp);
}


// ---------------------------------------------------------------------------------------------
// Backward without scattered atomics: partition (bin) -> per-segment LDS accumulation.
//
// The atomic scatter above is bound by the chip's memory-side atomic unit: 64 lanes hitting 64
// different 64-B lines run at ~20 G atomics/s whatever the schedule (MI355X_MICROARCH.md "Global
// float atomics"; measured here: 2.1 M points x 128 atomics in 15.7 ms). This path turns the
// scatter into streaming traffic:
//   1 count    every (point, level) computes its 8 corner rows; row >> 13 is the row's SEGMENT
//              (8192 rows = 64 KiB of fp32 pairs, one LDS image); per-workgroup LDS histogram,
//              then one global add per non-empty (level, segment);
//   2 scan     exclusive prefix over the <= 64 x L segment counts, chunk table for step 4;
//   3 scatter  recompute the rows, rank each record inside its segment with an LDS returning
//              atomic, reserve the workgroup's range per segment with ONE global returning atomic,
//              write {local row, w*grad} records (8 B for fp16, 4+8 B for fp32) with plain stores;
//   4 reduce   one workgroup per (segment, chunk of <= 65536 records): accumulate in a 128 KiB f64
//              LDS image with ds_add_f64, then add the image to the gradient table with
//              CONTIGUOUS atomics (256 B per wave instruction: the full-rate shape).
// Sums are f64 in LDS and rounded to the table dtype once per chunk — more accurate than the
// reference's half2 atomicAdd per addend (gridencoder.cu:325-331). C = 2, D = 3 (the NeRF tables).
#define GB_SEG_SHIFT 13u
#define GB_SEG (1u << GB_SEG_SHIFT)            // rows per segment
#define GB_MAX_SEGS 64u                        // per level: covers 2^19-row levels
#define GB_CHUNK 32768u                        // records (two corners each) per reduce workgroup

struct GbHeader {                              // lives at the start of the workspace
    uint32_t counts[GE_MAX_LEVELS * GB_MAX_SEGS];
    uint32_t base[GE_MAX_LEVELS * GB_MAX_SEGS + 1];
    uint32_t cursor[GE_MAX_LEVELS * GB_MAX_SEGS];
    uint32_t chunk_prefix[GE_MAX_LEVELS * GB_MAX_SEGS + 1];
};

struct GbSizes { uint32_t size[GE_MAX_LEVELS]; };          // rows per level (offsets[l + 1] - offsets[l]), from the host copy of the offsets

template <typename T> struct GbRec;
template <> struct GbRec<__half> { static constexpr uint32_t bytes = 8; };    // {u32 local row, half2}
template <> struct GbRec<float> { static constexpr uint32_t bytes = 12; };    // u32 rows[] + float2 vals[]

// cell and in-cell position of one (point, level): same arithmetic as ge_backward_one
template <uint32_t D>
__device__ __forceinline__ void gb_cell(const float (&x)[D], float scale, bool align_corners, uint32_t interp, uint32_t (&pos_grid)[D], float (&pos)[D]) {
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pos_grid[d] = (uint32_t)fl;
        pos[d] -= fl;                               // == (float)pos_grid[d] for every in-range point (0 <= fl < 2^24): one conversion less per axis
        if (interp == 1) { const float v = pos[d]; pos[d] = v * v * fmaf(-2.0f, v, 3.0f); }
    }
}
template <uint32_t D>
__device__ __forceinline__ void gb_cell_rows(const uint32_t (&pos_grid)[D], uint32_t hashmap_size, uint32_t resolution, uint32_t gridtype,
                                             bool align_corners, uint32_t (&rows)[1u << D]) {
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) pgl[d] = pos_grid[d] + ((idx >> d) & 1u);
        rows[idx] = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
    }
}
template <uint32_t D>
__device__ __forceinline__ void gb_cell_weights(const float (&pos)[D], float (&ws)[1u << D]) {
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) w *= (idx & (1u << d)) ? pos[d] : 1 - pos[d];
        ws[idx] = w;
    }
}
// D = 3, two channels: the 8 trilinear weights and the 16 products pv[2 i + c] = ws[i] * g[c], corner i = bit d set -> +1 along axis d.
__device__ __forceinline__ void gb_weighted_grads(const float (&pf)[3], float g0, float g1, float (&pv)[16]) {
    // (packed fp32 multiplies, v_pk_mul_f32, were measured slower here: 14 packed + the moves that pair their operands vs 28 plain multiplies)
    float ws[8];
    gb_cell_weights<3>(pf, ws);
#pragma unroll
    for (int i = 0; i < 8; i++) { pv[2 * i] = ws[i] * g0; pv[2 * i + 1] = ws[i] * g1; }
}

// corner rows + weights of one (point, level)
template <uint32_t D>
__device__ __forceinline__ void gb_corners(const float (&x)[D], uint32_t hashmap_size, float scale, uint32_t resolution, uint32_t gridtype,
                                           bool align_corners, uint32_t interp, uint32_t (&rows)[1u << D], float (&ws)[1u << D]) {
    float pos[D];
    uint32_t pos_grid[D];
    gb_cell<D>(x, scale, align_corners, interp, pos_grid, pos);
    gb_cell_weights<D>(pos, ws);
    gb_cell_rows<D>(pos_grid, hashmap_size, resolution, gridtype, align_corners, rows);
}

// ---- run merging (one-point-per-thread kernels) -------------------------------------------------
// Consecutive points are consecutive samples of a ray, so at the coarser levels neighbouring lanes fall into the SAME cell and
// address the same 8 rows (512 samples/ray: ~18 samples per cell at resolution 16, ~1 at 300). Within each aligned group of 16
// lanes a run of lanes with equal cells is summed on the lanes (segmented DPP scan, fp32) and only the run's last lane emits
// records. The count and the scatter kernel derive the run structure from the same values with the same code, so their record
// counts agree by construction.
#define GB_MERGE_MAX_RES 480u                  // default; levels above this resolution are not merged (hashed ones travel as factored 8-byte records); cells < 2^10 per axis are needed for the key
template <int CTRL>
__device__ __forceinline__ uint32_t gb_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
struct GbRun { uint32_t f0, f1, f2, f3; float k0, k1, k2, k3; bool tail; };       // f_k: "do not add from lane - 2^k" at scan step k; k_k = f_k ? 0 : 1
__device__ __forceinline__ GbRun gb_run_flags(bool inside, const uint32_t (&pos_grid)[3]) {
    const uint32_t r = threadIdx.x & 15u;
    const uint32_t key = inside ? (pos_grid[0] | (pos_grid[1] << 10) | (pos_grid[2] << 20)) : 0xFFFFFFFFu;
    const uint32_t prev = gb_dpp<0x111>(key);                               // row_shr:1
    const uint32_t head = (r == 0u || key != prev || !inside) ? 1u : 0u;
    GbRun run;
    run.f0 = head;
    run.f1 = run.f0 | gb_dpp<0x111>(run.f0);
    run.f2 = run.f1 | gb_dpp<0x112>(run.f1);
    run.f3 = run.f2 | gb_dpp<0x114>(run.f2);
    const uint32_t next_head = gb_dpp<0x101>(head);                         // row_shl:1
    run.tail = inside && (r == 15u || next_head != 0u);
    run.k0 = run.f0 ? 0.0f : 1.0f; run.k1 = run.f1 ? 0.0f : 1.0f; run.k2 = run.f2 ? 0.0f : 1.0f; run.k3 = run.f3 ? 0.0f : 1.0f;
    return run;
}
// One scan step = ONE instruction, v_fmac_f32_dpp: v += k * (v of lane - 2^step), k in {0, 1}. fma(u, 1, v) is the correctly rounded
// u + v and fma(u, 0, v) is v for every finite u, so finite data give the bits of the select form (add, then v_cndmask: two
// instructions per step and value, 128 per merged level). A non-finite addend of a neighbouring run turns into NaN here (inf * 0)
// where the select form would not have looked at it — in a step whose gradients already hold inf / NaN, which the AMP scaler discards
// whole (FOC_GB_MERGE_SELECT=1 at build time keeps the select form).
__device__ __forceinline__ float gb_run_sum(float v, const GbRun &run) {
    float u;
#ifdef FOC_GB_MERGE_SELECT
    u = __uint_as_float(gb_dpp<0x111>(__float_as_uint(v))); v = run.f0 ? v : v + u;
    u = __uint_as_float(gb_dpp<0x112>(__float_as_uint(v))); v = run.f1 ? v : v + u;
    u = __uint_as_float(gb_dpp<0x114>(__float_as_uint(v))); v = run.f2 ? v : v + u;
    u = __uint_as_float(gb_dpp<0x118>(__float_as_uint(v))); v = run.f3 ? v : v + u;
#else
    u = __uint_as_float(gb_dpp<0x111>(__float_as_uint(v))); v = fmaf(u, run.k0, v);
    u = __uint_as_float(gb_dpp<0x112>(__float_as_uint(v))); v = fmaf(u, run.k1, v);
    u = __uint_as_float(gb_dpp<0x114>(__float_as_uint(v))); v = fmaf(u, run.k2, v);
    u = __uint_as_float(gb_dpp<0x118>(__float_as_uint(v))); v = fmaf(u, run.k3, v);
#endif
    return v;
}
// The same scan over the 16 values of a (point, level) — 8 corners x 2 channels — with every step ONE v_fmac_f32_dpp per value (the
// compiler does not fold the DPP move into the fmac: it emits v_mov_b32_dpp + v_fmac, two instructions, like the select form). The 16
// instructions of a step are independent, so the "VALU write -> DPP read of the same VGPR" hazard (2 wait states on gfx9) only exists
// at the head of a block: one s_nop there covers it whatever precedes the block.
#ifndef FOC_GB_MERGE_SELECT
#define GB_FMAC16(CTRL)                                                                                                                  \
    asm volatile("s_nop 1\n\t"                                                                                                           \
                 "v_fmac_f32_dpp %0, %0, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %1, %1, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %2, %2, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %3, %3, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %4, %4, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %5, %5, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %6, %6, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %7, %7, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %8, %8, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %9, %9, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                       \
                 "v_fmac_f32_dpp %10, %10, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                     \
                 "v_fmac_f32_dpp %11, %11, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                     \
                 "v_fmac_f32_dpp %12, %12, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                     \
                 "v_fmac_f32_dpp %13, %13, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                     \
                 "v_fmac_f32_dpp %14, %14, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                     \
                 "v_fmac_f32_dpp %15, %15, %16 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                          \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),   \
                   "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])                                           \
                 : "v"(k))
#endif
__device__ __forceinline__ void gb_run_sum16(float (&v)[16], const GbRun &run) {
#ifdef FOC_GB_MERGE_SELECT
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = gb_run_sum(v[i], run);
#else
    // a scan step in which no lane of the wave adds anything is skipped (f_k set everywhere: every run is shorter than 2^k + 1 lanes) —
    // from resolution ~150 up runs of more than four samples are rare, and a step is 16 instructions; f_{k+1} clear implies f_k clear
    float k;
    if (__builtin_amdgcn_ballot_w64(run.f0 == 0u) != 0ull) {
        k = run.k0; GB_FMAC16("row_shr:1");
        if (__builtin_amdgcn_ballot_w64(run.f1 == 0u) != 0ull) {
            k = run.k1; GB_FMAC16("row_shr:2");
            if (__builtin_amdgcn_ballot_w64(run.f2 == 0u) != 0ull) {
                k = run.k2; GB_FMAC16("row_shr:4");
                if (__builtin_amdgcn_ballot_w64(run.f3 == 0u) != 0ull) { k = run.k3; GB_FMAC16("row_shr:8"); }
            }
        }
    }
#endif
}

// ---- one-point-per-thread count / scatter -------------------------------------------------------
// A workgroup owns GB_PM_TILE points x all levels, keeps one LDS counter per (level, segment) and talks to the global counters
// once per non-empty (level, segment); its own counts per slot give it a deterministic record range (k_gbin_scans).
#define GB_PM_TILE 1024u
#define GB_PMS_WG 1024u                         // threads of the count / scatter workgroups: one point per thread

// Both scans in one launch of L * 64 + 1 workgroups (they are independent: the scatter adds a slot's base to its workgroup prefix itself):
//   workgroups 0 .. L*64-1: per slot, exclusive scan over the workgroups' counts (in place);
//   the last workgroup    : exclusive scans over the L * 64 slot counts -> record base and reduce-chunk prefix of every slot.
__global__ void __launch_bounds__(256) k_gbin_scans(GbHeader *__restrict__ hdr, uint32_t *__restrict__ wg_hist, uint32_t n_wg, uint32_t L) {
    __shared__ uint32_t s_wave[2][4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = L * GB_MAX_SEGS;
    if (blockIdx.x == n) {
        const uint32_t per = (n + 255) / 256;
        const uint32_t lo = threadIdx.x * per, hi = min(n, lo + per);
        uint32_t recs = 0, chks = 0;
        for (uint32_t i = lo; i < hi; i++) { const uint32_t c = hdr->counts[i]; recs += c; chks += (c + GB_CHUNK - 1) / GB_CHUNK; }
        const uint32_t ir = (uint32_t)wave_incl_sum_i((int)recs, (int)lane), ic = (uint32_t)wave_incl_sum_i((int)chks, (int)lane);
        if (lane == 63) { s_wave[0][wave] = ir; s_wave[1][wave] = ic; }
        __syncthreads();
        uint32_t r = ir - recs, k = ic - chks;
        for (uint32_t w = 0; w < wave; w++) { r += s_wave[0][w]; k += s_wave[1][w]; }
        for (uint32_t i = lo; i < hi; i++) {
            const uint32_t c = hdr->counts[i];
            hdr->base[i] = r; hdr->chunk_prefix[i] = k;
            r += c; k += (c + GB_CHUNK - 1) / GB_CHUNK;
        }
        if (threadIdx.x == 255) { hdr->base[n] = r; hdr->chunk_prefix[n] = k; }   // the last thread's range ends at n (empty ranges pass r, k through)
        return;
    }
    const uint32_t slot = blockIdx.x;
    uint32_t *col = wg_hist + (uint64_t)slot * n_wg;
    const uint32_t per = (n_wg + 255) / 256;
    const uint32_t lo = min(n_wg, threadIdx.x * per), hi = min(n_wg, lo + per);
    uint32_t local = 0;
    for (uint32_t i = lo; i < hi; i++) local += col[i];
    const uint32_t incl = (uint32_t)wave_incl_sum_i((int)local, (int)lane);
    if (lane == 63) s_wave[0][wave] = incl;
    __syncthreads();
    uint32_t run = incl - local;
    for (uint32_t k = 0; k < wave; k++) run += s_wave[0][k];
    for (uint32_t i = lo; i < hi; i++) { const uint32_t c = col[i]; col[i] = run; run += c; }
}

// count of one 1024-point tile by a workgroup of THREADS threads (1024 / THREADS points per thread, one after the other)
template <uint32_t THREADS>
__device__ __forceinline__ void gb_count_tile(uint32_t *hist, uint32_t tile, uint32_t n_wg, const float *__restrict__ inputs, const int32_t *__restrict__ offsets,
                                              GbHeader *__restrict__ hdr, uint32_t *__restrict__ wg_hist, uint32_t B, uint32_t L, const GeLevels &lv,
                                              uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t l_begin, uint32_t l_end) {
    // levels [l_begin, l_end) of the tile: the slots of different levels are disjoint, so a tile's levels may be counted by different workgroups
    static_assert(GB_PM_TILE % THREADS == 0 && THREADS % 64 == 0, "whole waves, whole tile");
    const uint32_t slot_lo = l_begin * GB_MAX_SEGS, slot_hi = l_end * GB_MAX_SEGS;
    for (uint32_t i = slot_lo + threadIdx.x; i < slot_hi; i += THREADS) hist[i] = 0;
    __syncthreads();
    for (uint32_t it = 0; it < GB_PM_TILE / THREADS; it++) {
        const uint32_t b = tile * GB_PM_TILE + it * THREADS + threadIdx.x;
        float x[3] = {0.f, 0.f, 0.f};
        const bool inside = b < B && !ge_load_point<3>(inputs, b, x);
        for (uint32_t level = l_begin; level < l_end; level++) {
            const uint32_t resolution = lv.resolution[level];
            uint32_t pg[3]; float pf[3];
            gb_cell<3>(x, lv.scale[level], align_corners, interp, pg, pf);
            bool emit = inside;
            if (resolution <= lv.merge_max_res) emit = gb_run_flags(inside, pg).tail;
            if (emit) {
                const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - (uint32_t)offsets[level];
                uint32_t rows[8];
                bool hashed;
                ge_rows3(pg, hashmap_size, resolution, gridtype, align_corners, rows, &hashed);
                // one record per pair of corners along x (they share their segment unless they straddle an 8192-row boundary: then one each).
                // On a hashed level (wave-uniform) the pair's rows differ in the bits x ^ (x + 1) < 2^13 only (resolution < 8191, gb_check): never.
                if (hashed) {
#pragma unroll
                    for (int j = 0; j < 4; j++) atomicAdd(&hist[level * GB_MAX_SEGS + (rows[2 * j] >> GB_SEG_SHIFT)], 1u);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t s0 = rows[2 * j] >> GB_SEG_SHIFT, s1 = rows[2 * j + 1] >> GB_SEG_SHIFT;
                        atomicAdd(&hist[level * GB_MAX_SEGS + s0], 1u);
                        if (s0 != s1) atomicAdd(&hist[level * GB_MAX_SEGS + s1], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = slot_lo + threadIdx.x; i < slot_hi; i += THREADS) {
        const uint32_t hcount = hist[i];
        if (hcount) (void)__hip_atomic_fetch_add(&hdr->counts[i], hcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wg_hist[(uint64_t)i * n_wg + tile] = hcount;
    }
}

__global__ void __launch_bounds__(GB_PMS_WG) k_gbin_count_pt(const float *__restrict__ inputs, const int32_t *__restrict__ offsets, GbHeader *__restrict__ hdr,
                                                             uint32_t *__restrict__ wg_hist, uint32_t B, uint32_t L, GeLevels lv, uint32_t gridtype,
                                                             bool align_corners, uint32_t interp) {
    __shared__ uint32_t hist[GE_MAX_LEVELS * GB_MAX_SEGS];
    gb_count_tile<GB_PMS_WG>(hist, blockIdx.x, gridDim.x, inputs, offsets, hdr, wg_hist, B, L, lv, gridtype, align_corners, interp, 0u, L);
}

// The level-major forward (k_grid_fwd_lbc, plain walk, fp16 C = 2 D = 3, no dy_dx) with the backward's count pass riding along: every
// `period`-th workgroup of the launch counts one 1024-point tile instead of encoding 256 points of one level. The gathers are bound by
// cache-line requests and leave the VALU idle; the count is VALU/LDS work on the same positions. (As a separate kernel on a side
// stream the count starved: its 1024-thread workgroups never found 16 free wave slots on a CU while the forward's 4-wave workgroups
// kept back-filling, and it then ran into the MLP kernels and slowed those.)
template <typename T>
__global__ void __launch_bounds__(256) k_grid_fwd_counted(const float *__restrict__ inputs, const T *__restrict__ grid, const int32_t *__restrict__ offsets,
                                                          T *__restrict__ outputs, uint32_t B, uint32_t L, GeLevels lv, uint32_t gridtype, bool align_corners,
                                                          uint32_t interp, uint32_t chunks, uint32_t pairs, GbHeader *__restrict__ hdr,
                                                          uint32_t *__restrict__ wg_hist, uint32_t n_tiles, uint32_t period, uint32_t w0, uint32_t lc, uint32_t csplit) {
    __shared__ uint32_t hist[GE_MAX_LEVELS * GB_MAX_SEGS];
    // the counting workgroups sit among the encoding workgroups of the FINE levels (from block w0 on, one in `period`): those are the
    // request-bound ones; the coarse levels are instruction-bound themselves, and the last level is left alone so that no long-running
    // counting workgroup starts at the very end of the launch. A tile is counted by `csplit` workgroups, L / csplit levels each: one
    // workgroup walking all 16 levels of its 4 x 256 points is ~20 us of dependent VALU work on one wave per SIMD — longer than the
    // encoding workgroups behind it in a launch of 0.5 M points, whose end it then sets (forward 100 us, forward + count 120 us).
    const uint32_t n_count = n_tiles * csplit;
    uint32_t f = blockIdx.x;                                // index among the encoding workgroups
    if (blockIdx.x >= w0) {
        const uint32_t g = blockIdx.x - w0, q = g / period, r = g - q * period;
        if (r == period - 1u && q < n_count) {
            const uint32_t part = q % csplit;
            gb_count_tile<256>(hist, q / csplit, n_tiles, inputs, offsets, hdr, wg_hist, B, L, lv, gridtype, align_corners, interp, (part * L) / csplit, ((part + 1u) * L) / csplit);
            return;
        }
        f = blockIdx.x - min(q, n_count);
    }
    uint32_t chunk, l0, l1;
    ge_walk(f, chunks, lc, chunk, l0, l1);
    const uint32_t b = chunk * 256 + threadIdx.x;
    if (l0 >= L || b >= B) return;
    float x[3];
    const bool oob = ge_load_point<3>(inputs, b, x);
    for (uint32_t level = l0; level < l1; level++) {
        const uint32_t off0 = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
        ge_forward_level<T, 3, 2>(x, oob, grid, off0, hashmap_size, lv.scale[level], lv.resolution[level], outputs + ((uint64_t)level * B + b) * 2,
                                  (T *)nullptr, gridtype, align_corners, interp, pairs);
    }
}

// Level-sequential, LDS-sorted scatter, one point per thread (1024-thread workgroups, two per CU for fp16 tables).
// Writing each 8-byte record straight to its slot is bound by the number of store REQUESTS: 268 M partial 64-byte-line writes per
// 2 M-point step (measured 1.2 of 1.5 ms; 0.26 ms with the stores compiled out). Here the point and its gradient (a 64-byte
// [B, L*C] row, or 16 coalesced dwords from the [L,B,C] planes) are loaded once and stay in registers; the workgroup walks the levels
// one at a time: its record count per segment is the difference of two neighbouring per-workgroup bases, so the segment prefix
// inside the LDS staging array is known up front; each record takes its place with one LDS cursor atomic, and the staging array is
// copied out flat, so a wave stores 512 contiguous bytes. Two barriers per level; the bases of the next level are fetched while the
// current one is ranked.
// The two barriers per level order LDS traffic only (foc_lds_barrier, common.h): with `__syncthreads()` every wave sat at each of the
// 2 x L barriers until its copy-out stores to the record arrays had been acknowledged by memory and the prefetched gradient of the next
// level had landed. The records are consumed by the next kernel, never by this one; only the LDS staging arrays cross the barrier.
template <typename T>
__global__ void __launch_bounds__(GB_PMS_WG, (sizeof(T) == 2 ? 8 : 4)) k_gbin_scatter_pms(
    const T *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets, const GbHeader *__restrict__ hdr,
    const uint32_t *__restrict__ wg_base, void *__restrict__ recs, uint64_t max_recs, uint32_t B, uint32_t L, GeLevels lv, uint32_t gridtype,
    bool align_corners, uint32_t interp, bool grad_bl, uint32_t fact_mask, GbSizes sz, uint32_t n_tiles, uint32_t n_whole, uint32_t split) {
    // Workgroups 0 .. n_whole-1 take one tile each through all levels. The tiles behind them — the launch's last, partial round of
    // workgroups (gb_run) — are dealt out `split` workgroups per tile, each walking L / split of the levels: a tile's (level, segment)
    // record ranges are fixed by the count pass, so the levels of a tile are independent of each other. 517 tiles on a chip that holds
    // 512 of these workgroups used to be two rounds, the second one five workgroups wide and a full tile long (105 us where 512 tiles take 60).
    uint32_t tile = blockIdx.x, l_begin = 0u, l_end = L;
    if (blockIdx.x >= n_whole) {
        const uint32_t r = blockIdx.x - n_whole, part = r % split;
        tile = n_whole + r / split;
        l_begin = (part * L) / split; l_end = ((part + 1u) * L) / split;
    }
    static_assert(GB_PM_TILE == GB_PMS_WG, "one point per thread");
    // a record carries the two corners along x of one (y, z) corner pair: 4 per point, 5 when one pair straddles a segment boundary
    // (two pairs of a point cannot: their rows differ by less than 8192 and not by a multiple of it on a dense level, and a hashed
    // level's x never reaches 8191)
    constexpr uint32_t NREC = GB_PM_TILE * 5u;
    constexpr uint32_t VW = sizeof(T) == 2 ? 1 : 2;         // dwords per corner value
    __shared__ uint32_t cur[2][GB_MAX_SEGS];               // next free staging position per segment (double-buffered by level parity)
    __shared__ uint32_t pre[2][GB_MAX_SEGS + 1];           // first staging position per segment; [64] = records of this level
    __shared__ uint32_t gb[2][GB_MAX_SEGS];                // this workgroup's first global record per segment MINUS its first staging position
    __shared__ uint32_t s_rows[NREC];                      // local row of corner 0 | local row of corner 1 << 13 | segment << 26
    __shared__ uint32_t s_val[2 * VW][NREC];               // corner 0 value words, corner 1 value words
    const uint32_t n_wg = n_tiles;
    const uint32_t b = tile * GB_PM_TILE + threadIdx.x;
    float x[3] = {0.f, 0.f, 0.f};
    const bool inside = b < B && !ge_load_point<3>(inputs, b, x);
    // Gradient of the tile, one level at a time (fp16 -> one dword per point, fp32 -> two), through LDS: WAVE 0 requests the 1024 points'
    // values of level l + 1 with LDS-direct loads (global_load_lds_dword: no register destination; 16 x 64 lanes, coalesced on the
    // [L,B,C] planes of gridencoder.cu:283, one sector per lane on [B, L*C] rows) behind the first barrier of level l; everybody reads its
    // value after the first barrier of level l + 1. Wave 0 issues every LOAD of the kernel (these and the segment bases) and takes no
    // part in the copy-out, the other fifteen waves issue nothing but the copy-out STORES — so no wave ever waits for a store:
    // vmcnt is one in-order counter for loads and stores, and with the per-thread prefetch register the kernel had before, each wave's
    // wait for its gradient at the top of a level was a wait for every record it had stored in the level before (timing builds: 292 us,
    // 206 us without the stores — they did not overlap the next level's work at all).
    constexpr uint32_t GW = sizeof(T) == 2 ? 1 : 2;
    __shared__ uint32_t s_grad[2][GW][GB_PM_TILE];
    auto fetch_grad = [&](uint32_t level) {                 // wave 0 only
        if (level < l_end) {
            const uint32_t *gbase = reinterpret_cast<const uint32_t *>(grad);
#pragma unroll 4
            for (uint32_t k = 0; k < GB_PM_TILE / 64u; k++) {
                const uint32_t pt = tile * GB_PM_TILE + k * 64u + threadIdx.x;
                const uint32_t pq = pt < B ? pt : 0u;
                const uint32_t *src = grad_bl ? gbase + ((uint64_t)pq * L + level) * GW : gbase + ((uint64_t)level * B + pq) * GW;
                // as inline asm: with the builtin, hipcc 7.2 puts `s_waitcnt vmcnt(0)` in front of EVERY later LDS access of every wave (it cannot
                // tell the LDS-DMA's destination from the other arrays) — the stores the other waves have in flight would be waited for again
#pragma unroll
                for (uint32_t w = 0; w < GW; w++) {
                    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(
                        (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&s_grad[level & 1u][w][k * 64u]);
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src + w), "s"(dst) : "memory");
                }
            }
        }
    };
    // wave 0 keeps the (base, end) pair of the level about to be processed in registers
    uint32_t nsb = 0, nw0 = 0, nw1 = 0;
    auto fetch_bases = [&](uint32_t level) {
        if (threadIdx.x < GB_MAX_SEGS && level < l_end) {
            const uint32_t slot = level * GB_MAX_SEGS + threadIdx.x;
            const uint32_t sb = hdr->base[slot];           // wg_base holds the prefix inside the slot
            const uint32_t wi = slot * n_wg + tile;      // < 2^26 (gb_check: B * 8 * L < 2^32); a 32-bit offset from the scalar base keeps one register live
            // loaded values only, no arithmetic: the sums are taken where the next level consumes them, so wave 0 does not wait for these
            // loads in front of the barrier the other fifteen waves are parked at
            nsb = sb;
            nw0 = wg_base[wi];
            nw1 = *(tile + 1 < n_wg ? &wg_base[wi + 1u] : &hdr->counts[slot]);
        }
    };
    // Wave 0 prepares the segment tables of level l + 1 (cur / pre / gb of the other parity) right AFTER the first barrier of level l,
    // from bases fetched a level earlier: in front of the barrier it made the other fifteen waves wait for its loads (and, vmcnt being
    // one in-order counter on gfx950, for the acknowledgement of its copy-out stores). The tables of that parity were last read by the
    // copy-out of level l - 1, which every thread has left when it passes the barrier.
    auto setup_tables = [&](uint32_t pbn) {
        const uint32_t nb0 = nsb + nw0;
        const uint32_t hc = nw1 - nw0;
        const uint32_t incl = (uint32_t)wave_incl_sum_i((int)hc, (int)threadIdx.x);
        pre[pbn][threadIdx.x] = incl - hc;
        cur[pbn][threadIdx.x] = incl - hc;
        gb[pbn][threadIdx.x] = nb0 - (incl - hc);
        if (threadIdx.x == GB_MAX_SEGS - 1) pre[pbn][GB_MAX_SEGS] = incl;
    };
    fetch_bases(l_begin);
    if (threadIdx.x < GB_MAX_SEGS) { if (l_begin < l_end) setup_tables(l_begin & 1u); fetch_bases(l_begin + 1u); fetch_grad(l_begin); }
    // The per-level parameters (scale, resolution, rows of the level) come out of the KERNEL ARGUMENTS with a scalar index: as vector
    // loads (`offsets[level]` from memory, `lv.scale[level]` with the level in a VGPR) each of them was followed by `s_waitcnt vmcnt(0)`,
    // which on gfx950 also waits for the gradient prefetch just issued and for every copy-out store of the level before.
    for (uint32_t level_v = l_begin; level_v < l_end; level_v++) {
        {
        const uint32_t level = (uint32_t)__builtin_amdgcn_readfirstlane((int)level_v);
        const uint32_t pb = level & 1u;
        if (threadIdx.x < GB_MAX_SEGS) __builtin_amdgcn_s_waitcnt(0x0F70);       // wave 0: this level's gradients (and bases) have landed: vmcnt(0), issued a level ago
        foc_lds_barrier();
        if (threadIdx.x < GB_MAX_SEGS && level + 1u < l_end) { setup_tables(pb ^ 1u); fetch_bases(level + 2u); fetch_grad(level + 1u); }
        float g[2];
        if constexpr (sizeof(T) == 2) {
            const uint32_t gw = s_grad[pb][0][threadIdx.x];
            g[0] = __half2float(__ushort_as_half((unsigned short)(gw & 0xFFFFu)));
            g[1] = __half2float(__ushort_as_half((unsigned short)(gw >> 16)));
        } else {
            g[0] = __uint_as_float(s_grad[pb][0][threadIdx.x]); g[1] = __uint_as_float(s_grad[pb][GW - 1][threadIdx.x]);
        }
        if (!inside) { g[0] = 0.0f; g[1] = 0.0f; }
        const bool fact = sizeof(T) == 2 && ((fact_mask >> level) & 1u) != 0u;      // kernel-uniform per level (gb_fact_mask)
        if (fact) {
            // FACTORED record (unmerged hashed levels, fp16 tables): the two corners along x of a (y, z) pair get (1 - fx) p and fx p of the
            // SAME product p = w_y w_z g, and on a hashed level their rows differ by x ^ (x + 1) = 2^(jb+1) - 1 (jb = trailing ones of x,
            // < 13 below resolution 8191) — so the pair travels as 8 bytes {local row 0 (13) | jb (4) | fx (15 bits), half2 p} instead of
            // 12 {row 0 | row 1, half2 v0, half2 v1}; the reduce rebuilds row 1 and both addends. One-third fewer record bytes written and
            // read back on the levels where nothing merges, four weights instead of eight, no run scan.
            uint32_t pg[3]; float pf[3];
            gb_cell<3>(x, lv.scale[level], align_corners, interp, pg, pf);
            if (inside) {
                const uint32_t mask = sz.size[level] - 1u;
                const uint32_t ty0 = pg[1] * 2654435761u, ty1 = ty0 + 2654435761u, tz0 = pg[2] * 805459861u, tz1 = tz0 + 805459861u;
                const uint32_t jb = (uint32_t)__builtin_ctz(~pg[0]);
                const uint32_t fxq = min((uint32_t)rintf(pf[0] * 32768.0f), 32767u);
                const uint32_t meta = (jb << 13) | (fxq << 17);
#pragma unroll 1
                for (int j = 0; j < 4; j++) {
                    const uint32_t row = (pg[0] ^ ((j & 1) ? ty1 : ty0) ^ ((j >> 1) ? tz1 : tz0)) & mask;
                    const float w = ((j & 1) ? pf[1] : 1 - pf[1]) * ((j >> 1) ? pf[2] : 1 - pf[2]);
                    typedef _Float16 gb_h2 __attribute__((ext_vector_type(2)));
                    const gb_h2 hv = {(_Float16)ge_opaque(w * g[0]), (_Float16)ge_opaque(w * g[1])};
                    const uint32_t seg = row >> GB_SEG_SHIFT;
                    const uint32_t pos = atomicAdd(&cur[pb][seg], 1u);
                    if (pos < NREC) {
                        s_rows[pos] = (row & (GB_SEG - 1u)) | meta;
                        s_val[0][pos] = *reinterpret_cast<const uint32_t *>(&hv);
                        s_val[VW][pos] = seg;
                    }
                }
            }
        } else {
            const uint32_t resolution = lv.resolution[level];
            uint32_t pg[3]; float pf[3];
            gb_cell<3>(x, lv.scale[level], align_corners, interp, pg, pf);
            uint32_t pv0[8], pv1[8];                       // record values: fp16 -> half2 bits in pv0; fp32 -> two floats
            bool emit = inside;
            float pv[16];
            gb_weighted_grads(pf, g[0], g[1], pv);
            if (resolution <= lv.merge_max_res) {
                const GbRun run = gb_run_flags(inside, pg);
                emit = run.tail;
                gb_run_sum16(pv, run);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float v0 = pv[2 * i], v1 = pv[2 * i + 1];
                if constexpr (sizeof(T) == 2) {
                    typedef _Float16 gb_h2 __attribute__((ext_vector_type(2)));
                    const gb_h2 hv = {(_Float16)ge_opaque(v0), (_Float16)ge_opaque(v1)};        // one v_cvt_pk_f16_f32 (round to nearest even) per pair
                    pv0[i] = *reinterpret_cast<const uint32_t *>(&hv);
                } else { pv0[i] = __float_as_uint(v0); pv1[i] = __float_as_uint(v1); }
            }
            if (emit) {
                const uint32_t hashmap_size = sz.size[level];
                uint32_t rows[8];
                bool hashed;
                ge_rows3(pg, hashmap_size, resolution, gridtype, align_corners, rows, &hashed);
                auto place = [&](uint32_t seg, uint32_t r0, uint32_t r1, int i0, int i1) {      // i1 < 0: corner 1 carries nothing
                    const uint32_t pos = atomicAdd(&cur[pb][seg], 1u);
                    if (pos >= NREC) return;               // cannot happen when count and scatter agree
                    s_rows[pos] = (r0 & (GB_SEG - 1u)) | ((r1 & (GB_SEG - 1u)) << 13) | (seg << 26);
                    s_val[0][pos] = pv0[i0];
                    s_val[VW][pos] = i1 >= 0 ? pv0[i1 >= 0 ? i1 : 0] : 0u;
                    if constexpr (sizeof(T) != 2) { s_val[1][pos] = pv1[i0]; s_val[VW + 1][pos] = i1 >= 0 ? pv1[i1 >= 0 ? i1 : 0] : 0u; }
                };
                if (hashed) {                              // wave-uniform: a hashed level's pairs never straddle a segment boundary (see the count pass)
#pragma unroll
                    for (int j = 0; j < 4; j++) place(rows[2 * j] >> GB_SEG_SHIFT, rows[2 * j], rows[2 * j + 1], 2 * j, 2 * j + 1);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t r0 = rows[2 * j], r1 = rows[2 * j + 1];
                        const uint32_t s0 = r0 >> GB_SEG_SHIFT, s1 = r1 >> GB_SEG_SHIFT;
                        if (s0 == s1) place(s0, r0, r1, 2 * j, 2 * j + 1);
                        else { place(s0, r0, r0, 2 * j, -1); place(s1, r1, r1, 2 * j + 1, -1); }
                    }
                }
            }
        }
        foc_lds_barrier();
        const uint32_t total = min(pre[pb][GB_MAX_SEGS], NREC);
        uint32_t *rec_rows = reinterpret_cast<uint32_t *>(recs);                       // [max_recs] rows, then [max_recs] values (16-byte aligned)
        uint32_t *rec_vals = rec_rows + ((max_recs + 3) & ~(uint64_t)3);
        if (fact) {
            if constexpr (sizeof(T) == 2) {
                for (uint32_t j = threadIdx.x - 64u; j < total; j += GB_PMS_WG - 64u) {      // wave 0 (j wraps past total) copies nothing
                    const uint32_t at = gb[pb][s_val[1][j] & (GB_MAX_SEGS - 1u)] + j;
                    if (at >= max_recs) continue;
                    reinterpret_cast<uint2 *>(rec_vals)[at] = make_uint2(s_rows[j], s_val[0][j]);      // the rows array is not touched on these levels
                }
            }
        } else
        for (uint32_t j = threadIdx.x - 64u; j < total; j += GB_PMS_WG - 64u) {
            const uint32_t rw = s_rows[j];
            const uint32_t at = gb[pb][rw >> 26] + j;
            if (at >= max_recs) continue;                  // cannot happen when count and scatter agree; keeps a logic slip from faulting
            rec_rows[at] = rw;                             // bits 26.. (the staging segment) ride along: the reduce reads two 13-bit rows and nothing else
            if constexpr (sizeof(T) == 2) reinterpret_cast<uint2 *>(rec_vals)[at] = make_uint2(s_val[0][j], s_val[1][j]);
            else reinterpret_cast<uint4 *>(rec_vals)[at] = make_uint4(s_val[0][j], s_val[1][j], s_val[2][j], s_val[3][j]);
        }
        // no barrier here: the next level's setup writes the other parity of cur/pre/gb, and its staging writes come after
        // the barrier that follows the setup, which every thread reaches only once its copy-out loop is done
        }
    }
}

// LDS accumulation is done in DOUBLE: on gfx950 ds_add_f32 on random addresses runs ~23x slower than
// ds_add_u32 (measured 101 vs 2349 G records/s, tools/bench_lds_atomic.hip) while ds_add_f64 runs at
// 1823 G/s — so the fp32-quality sum is kept in a 128 KiB f64 image (one workgroup per CU).
// A finite fp16 value times 2^24 as a two's-complement 64-bit integer, straight from its bits: normal numbers are
// (1024 | mantissa) << (exponent - 1), subnormals the mantissa itself (fp32 -> int64 conversion has no instruction on gfx950 and
// expands to a dozen; this is a shift and a conditional negate).
__device__ __forceinline__ unsigned long long gb_half_to_fixed(uint32_t h) {
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    const unsigned long long mag = e ? ((unsigned long long)(1024u | m) << (e - 1u)) : (unsigned long long)m;
    return (h & 0x8000u) ? (0ull - mag) : mag;
}

#define GB_RTHREADS 1024u
template <typename T>
__global__ void __launch_bounds__(GB_RTHREADS) k_gbin_reduce(const GbHeader *__restrict__ hdr, const void *__restrict__ recs, uint64_t max_recs,
                                                             const int32_t *__restrict__ offsets, T *__restrict__ grad_grid, uint32_t L, uint32_t fact_mask) {
    // 128 KiB (fp32 tables: f64 sums; fp16 tables: the same bytes as 2^24-scaled int64), one PLANE per channel: with the two channels
    // of a row side by side a wave instruction (one channel of 64 random rows) could only ever touch every other pair of banks —
    // half of the LDS's banks idle, twice the conflict cycles; planes spread a channel's 64 addends over all 64 banks
    __shared__ double acc[GB_SEG * 2];
    unsigned long long *acci = reinterpret_cast<unsigned long long *>(acc);
    __shared__ uint32_t s_bad[GB_SEG / 32];        // fp16 tables: rows that received an inf/NaN addend (an overflowed AMP step) -> NaN out
    __shared__ uint32_t s_slot, s_lo, s_hi;
    const uint32_t n = L * GB_MAX_SEGS;
    const uint32_t total_chunks = hdr->chunk_prefix[n];
    // (one workgroup per chunk; a persistent form — one workgroup per CU walking the chunks with the grid's stride — measured 263 vs
    // 256 us, and with larger chunks worse: the dynamic dispatch balances the uneven chunks better than a stride does)
    const uint32_t chunk_id = blockIdx.x;
    if (chunk_id >= total_chunks) return;
    {
    // Which (level, segment) slot and which chunk of it this workgroup owns: the slot with chunk_prefix[slot] <= blockIdx.x < chunk_prefix[slot + 1]
    // (empty slots have equal neighbours and match nothing). Every thread tests one or two slots: one round trip to the header instead of
    // thread 0's eleven dependent loads of a binary search (no measurable difference: the header sits in L2; kept for the shorter chain).
    if (threadIdx.x == 0) { s_slot = 0u; s_lo = 0u; s_hi = 0u; }      // a header that matches nothing (not this launch's) reduces nothing
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += GB_RTHREADS) {
        const uint32_t p0 = hdr->chunk_prefix[i], p1 = hdr->chunk_prefix[i + 1];
        const uint32_t cnt = hdr->counts[i];
        const uint64_t b0 = hdr->base[i];
        if (p0 <= chunk_id && chunk_id < p1) {
            const uint32_t c = chunk_id - p0;
            s_slot = i;
            // clamped to the record arrays: a header that does not belong to these records must not turn into an out-of-bounds read
            s_lo = (uint32_t)min(b0 + (uint64_t)c * GB_CHUNK, max_recs);
            s_hi = (uint32_t)min(b0 + min(cnt, (c + 1) * GB_CHUNK), max_recs);
        }
    }
    for (uint32_t i = threadIdx.x; i < GB_SEG * 2; i += GB_RTHREADS) acc[i] = 0.0;
    if (threadIdx.x < GB_SEG / 32) s_bad[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t slot = s_slot, lo = s_lo, hi = s_hi;
    // Software pipeline: the next UNR records per lane are in flight while the current ones go through the LDS adds
    // (with a plain load -> wait -> add loop the HBM latency was exposed once per iteration: 1.15 ms for 2.1 GB).
    constexpr uint32_t UNR = 4;
    const uint32_t *rec_rows = reinterpret_cast<const uint32_t *>(recs);           // record = rows word + the values of its two corners
    const uint32_t *rec_vals = rec_rows + ((max_recs + 3) & ~(uint64_t)3);
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    if constexpr (sizeof(T) == 2) {
        const uint2 *vv = reinterpret_cast<const uint2 *>(rec_vals);
        // fp16 addends are exact multiples of 2^-24 below 2^16: as 2^24-scaled 64-bit integers their sum is EXACT (and order
        // independent); ds_add_u64 is also the fastest LDS atomic that can hold it (tools/bench_lds_atomic.hip: 1335 G two-channel records/s on
        // planes, ds_add_f64 600, ds_add_f32 100, ds_pk_add_f16 200). The kernel ran at a third of that rate: its VALU was 59 % busy with the
        // conversion (a 64-bit shift, a 64-bit negate and four selects per value, behind a per-value inf/NaN branch). Now:
        //   * half -> 2^24-scaled int64 in 8 plain instructions, no select, no branch, sign and subnormals included: s = 16 f is exact,
        //     H = floor(s) < 2^21, L = (s - H) 2^20 < 2^20 are exact fp32 integers, value = H 2^20 + L: lo word (H << 20) | L, hi word H >> 12;
        //   * ONE inf/NaN test per record on its four halves ((x & 0x7C00) + 0x0400 carries into bit 15 only for an all-ones exponent);
        //     such a record (an overflowed AMP step) takes the slow path that marks its rows.
        // (v_cvt_flr_i32_f32 = floor and convert in one instruction, v_fract_f32 = s - floor(s), exact here: 7 instructions per value)
        auto fixed = [](float f, uint32_t &wlo, uint32_t &whi) {
            const float sc = f * 16.0f;
            int32_t Hi;
            asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(Hi) : "v"(sc));
            wlo = ((uint32_t)Hi << 20) | (uint32_t)(__builtin_amdgcn_fractf(sc) * 1048576.0f);
            whi = (uint32_t)(Hi >> 12);
        };
        auto add_fast = [&](uint32_t row, uint32_t hv) {
            const __half2 h2 = *reinterpret_cast<const __half2 *>(&hv);
            uint32_t l0, h0, l1, h1;
            fixed(__low2float(h2), l0, h0);
            fixed(__high2float(h2), l1, h1);
            atomicAdd(&acci[row], ((unsigned long long)h0 << 32) | l0);
            atomicAdd(&acci[GB_SEG + row], ((unsigned long long)h1 << 32) | l1);
        };
        auto add_slow = [&](uint32_t row, uint32_t hv) {
            if ((hv & 0x7C00u) == 0x7C00u || (hv & 0x7C000000u) == 0x7C000000u) {       // inf / NaN: the reference's half2 atomics would leave inf/NaN in the row
                atomicOr(&s_bad[row >> 5], 1u << (row & 31u));
                return;
            }
            add_fast(row, hv);
        };
        // Software pipeline over batches of UNR x 1024 records: the loads of batch k + 1 are issued, then batch k goes through the LDS adds.
        // The loads of a COMPLETE batch carry no per-lane guard: with `i < hi ? load : 0` every load sat in its own exec-masked branch, the
        // compiler could no longer count the loads in flight and put `s_waitcnt vmcnt(0)` right behind the prefetch — each iteration
        // waited out the full latency of the loads it had just issued. Only the last batch of a chunk (partial) takes guarded loads;
        // its missing records read as "row 0 += 0".
        constexpr uint32_t BATCH = GB_RTHREADS * UNR;
        auto pipeline = [&](auto load_full, auto load_tail, auto process) {
            uint32_t cr[UNR], nr[UNR]; uint2 cv[UNR], nv[UNR];
            uint32_t base = lo;                                    // workgroup-uniform
            if (base + BATCH <= hi) load_full(base, cr, cv); else load_tail(base, cr, cv);
            while (base + 3u * BATCH <= hi) {                      // two more complete batches: ping-pong, no register rotation (a rotation
                load_full(base + BATCH, nr, nv);                   // is a v_mov of loaded registers, i.e. a wait for the loads just issued)
                process(cr, cv);
                load_full(base + 2u * BATCH, cr, cv);
                process(nr, nv);
                base += 2u * BATCH;
            }
            while (base + 2u * BATCH <= hi) {                      // batch k + 1 is complete too
                load_full(base + BATCH, nr, nv);
                process(cr, cv);
#pragma unroll
                for (uint32_t u = 0; u < UNR; u++) { cr[u] = nr[u]; cv[u] = nv[u]; }
                base += BATCH;
            }
            if (base + BATCH < hi) { load_tail(base + BATCH, nr, nv); process(cr, cv); process(nr, nv); }
            else if (base < hi) process(cr, cv);
        };
        if ((fact_mask >> (slot / GB_MAX_SEGS)) & 1u) {
            // factored records (k_gbin_scatter_pms): {local row 0 | jb << 13 | fx << 17, half2 p} -> row 1 = row 0 ^ (2^(jb+1) - 1),
            // addends (p - fx p, fx p) per channel. They are fp32 values, not halves: `fixedf` takes any |f| < 2^17 and drops what lies below
            // 2^-24, the spacing of the smallest halves — the sum stays order independent.
            // v_fract_f32 instead of s - floor(s): for a tiny negative s the difference rounds to 1.0 and the fraction word would carry
            // into H (an addend off by 2^-4); the instruction returns at most 1 - 2^-24
            auto fixedf = [](float f, uint32_t &wlo, uint32_t &whi) {
                const float sc = f * 16.0f;
                int32_t Hi;
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(Hi) : "v"(sc));
                wlo = ((uint32_t)Hi << 20) | (uint32_t)(__builtin_amdgcn_fractf(sc) * 1048576.0f);
                whi = (uint32_t)(Hi >> 12);
            };
            auto add2 = [&](uint32_t row, float a, float b) {
                uint32_t l0, h0, l1, h1;
                fixedf(a, l0, h0);
                fixedf(b, l1, h1);
                atomicAdd(&acci[row], ((unsigned long long)h0 << 32) | l0);
                atomicAdd(&acci[GB_SEG + row], ((unsigned long long)h1 << 32) | l1);
            };
            pipeline(
                [&](uint32_t base, uint32_t (&)[UNR], uint2 (&v)[UNR]) {
#pragma unroll
                    for (uint32_t u = 0; u < UNR; u++) v[u] = vv[base + threadIdx.x + u * GB_RTHREADS];
                },
                [&](uint32_t base, uint32_t (&)[UNR], uint2 (&v)[UNR]) {
#pragma unroll
                    for (uint32_t u = 0; u < UNR; u++) { const uint32_t i = base + threadIdx.x + u * GB_RTHREADS; v[u] = i < hi ? vv[i] : make_uint2(0u, 0u); }
                },
                [&](const uint32_t (&)[UNR], const uint2 (&v)[UNR]) {
#pragma unroll
                    for (uint32_t u = 0; u < UNR; u++) {
                        const uint32_t w = v[u].x;
                        const uint32_t r0 = w & (GB_SEG - 1u), r1 = (r0 ^ ((2u << ((w >> 13) & 15u)) - 1u)) & (GB_SEG - 1u);
                        const float fx = (float)(w >> 17) * (1.0f / 32768.0f);
                        const uint32_t nonfinite = ((v[u].y & 0x7C007C00u) + 0x04000400u) & 0x80008000u;
                        if (__builtin_expect(nonfinite == 0u, 1)) {
                            const __half2 h2 = *reinterpret_cast<const __half2 *>(&v[u].y);
                            const float p0 = __low2float(h2), p1 = __high2float(h2);
                            const float a1 = fx * p0, b1 = fx * p1;
                            add2(r0, p0 - a1, p1 - b1);
                            add2(r1, a1, b1);
                        } else {
                            atomicOr(&s_bad[r0 >> 5], 1u << (r0 & 31u));
                            atomicOr(&s_bad[r1 >> 5], 1u << (r1 & 31u));
                        }
                    }
                });
        } else {
            pipeline(
                [&](uint32_t base, uint32_t (&r)[UNR], uint2 (&v)[UNR]) {
#pragma unroll
                    for (uint32_t u = 0; u < UNR; u++) { const uint32_t i = base + threadIdx.x + u * GB_RTHREADS; r[u] = rec_rows[i]; v[u] = vv[i]; }
                },
                [&](uint32_t base, uint32_t (&r)[UNR], uint2 (&v)[UNR]) {
#pragma unroll
                    for (uint32_t u = 0; u < UNR; u++) {
                        const uint32_t i = base + threadIdx.x + u * GB_RTHREADS;
                        const bool ok = i < hi;
                        r[u] = ok ? rec_rows[i] : 0u; v[u] = ok ? vv[i] : make_uint2(0u, 0u);
                    }
                },
                [&](const uint32_t (&r)[UNR], const uint2 (&v)[UNR]) {
#pragma unroll
                    for (uint32_t u = 0; u < UNR; u++) {
                        const uint32_t r0 = r[u] & (GB_SEG - 1u), r1 = (r[u] >> 13) & (GB_SEG - 1u);
                        const uint32_t nonfinite = (((v[u].x & 0x7C007C00u) + 0x04000400u) | ((v[u].y & 0x7C007C00u) + 0x04000400u)) & 0x80008000u;
                        if (__builtin_expect(nonfinite == 0u, 1)) { add_fast(r0, v[u].x); add_fast(r1, v[u].y); }
                        else { add_slow(r0, v[u].x); add_slow(r1, v[u].y); }
                    }
                });
        }
    } else {
        const float4 *vv = reinterpret_cast<const float4 *>(rec_vals);
        uint32_t cr[UNR], nr[UNR]; float4 cv[UNR], nv[UNR];
        uint32_t i0 = lo + threadIdx.x;
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++) { const uint32_t i = i0 + u * GB_RTHREADS; const bool ok = i < hi; cr[u] = ok ? rec_rows[i] : NONE; cv[u] = ok ? vv[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
        while (i0 < hi) {
            const uint32_t i1 = i0 + GB_RTHREADS * UNR;
#pragma unroll
            for (uint32_t u = 0; u < UNR; u++) { const uint32_t i = i1 + u * GB_RTHREADS; const bool ok = i < hi; nr[u] = ok ? rec_rows[i] : NONE; nv[u] = ok ? vv[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (uint32_t u = 0; u < UNR; u++) {
                if (cr[u] != NONE) {
                    const uint32_t r0 = cr[u] & (GB_SEG - 1u), r1 = (cr[u] >> 13) & (GB_SEG - 1u);
                    atomicAdd(&acc[r0], (double)cv[u].x);
                    atomicAdd(&acc[GB_SEG + r0], (double)cv[u].y);
                    atomicAdd(&acc[r1], (double)cv[u].z);
                    atomicAdd(&acc[GB_SEG + r1], (double)cv[u].w);
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < UNR; u++) { cr[u] = nr[u]; cv[u] = nv[u]; }
            i0 = i1;
        }
    }
    __syncthreads();
    const uint32_t level = slot / GB_MAX_SEGS, seg = slot % GB_MAX_SEGS;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const uint32_t row0 = seg * GB_SEG;
    T *dst = grad_grid + ((uint64_t)off0 + row0) * 2;
    const uint32_t nrows = hashmap_size > row0 ? min(GB_SEG, hashmap_size - row0) : 0u;
    if constexpr (sizeof(T) == 2) {
        typedef _Float16 __attribute__((ext_vector_type(2))) v2h;
        for (uint32_t r = threadIdx.x; r < nrows; r += GB_RTHREADS) {
            float a = (float)((double)(long long)acci[r] * (1.0 / 16777216.0)), b = (float)((double)(long long)acci[GB_SEG + r] * (1.0 / 16777216.0));
            if ((s_bad[r >> 5] >> (r & 31u)) & 1u) { a = __builtin_nanf(""); b = __builtin_nanf(""); }
            if (a == 0.0f && b == 0.0f) continue;
            v2h hv; hv[0] = (_Float16)ge_opaque(a); hv[1] = (_Float16)ge_opaque(b);
            (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) v2h *)(dst + 2 * r), hv);
        }
    } else {
        for (uint32_t e = threadIdx.x; e < nrows * 2; e += GB_RTHREADS) {
            const float a = (float)acc[(e & 1u) * GB_SEG + (e >> 1)];
            if (a != 0.0f) (void)__hip_atomic_fetch_add(dst + e, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    }
}

// gridencoder.cu:343-369: grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]   (fp32 accumulate)
template <typename T, uint32_t D, uint32_t C, bool GRAD_BL>
__global__ void __launch_bounds__(256) k_grid_input_bwd(const T *__restrict__ grad, const T *__restrict__ dy_dx, T *__restrict__ grad_inputs,
                                                        uint32_t B, uint32_t L) {
    const uint64_t total = (uint64_t)B * D;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (uint64_t)b * D);
        float r = 0;
        for (uint32_t l = 0; l < L; l++) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                const float gv = GeT<T>::ld(GRAD_BL ? grad + ((uint64_t)b * L + l) * C + c : grad + ((uint64_t)l * B + b) * C + c);
                r = fmaf(gv, GeT<T>::ld(dy_dx + (((uint64_t)b * L + l) * D + d) * C + c), r);
            }
        }
        GeT<T>::st(grad_inputs + t, r);
    }
}

// gridencoder.cu:506-610 kernel_grad_tv
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grad_tv(const T *__restrict__ inputs, const T *__restrict__ grid, T *__restrict__ grad,
                                                 const int32_t *__restrict__ offsets, float weight, uint32_t B, uint32_t L,
                                                 GeLevels lv, uint32_t gridtype, bool align_corners, uint32_t chunks) {
    uint32_t level, chunk;
    if (!ge_decode_block(blockIdx.x, chunks, L, level, chunk)) return;
    const uint32_t b = chunk * 256 + threadIdx.x;
    if (b >= B) return;
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) { x[d] = GeT<T>::ld(inputs + (uint64_t)b * D + d); oob |= (x[d] < 0 || x[d] > 1); }
    if (oob) return;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    const T *tab = grid + (uint64_t)off0 * C;
    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) pos_grid[d] = (uint32_t)floorf(fmaf(x[d], scale, align_corners ? 0.0f : 0.5f));
    float results[C], idelta[C], center[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) { results[c] = 0; idelta[c] = 0; }
    const uint32_t row = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid);
    GeVec<T, C>::ld(tab + (uint64_t)row * C, center);
    const float w = weight / (float)(2 * D);
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t cur = pos_grid[d];
        if (cur < resolution) {
            pos_grid[d] = cur + 1;
            float o[C];
            GeVec<T, C>::ld(tab + (uint64_t)ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C, o);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) { const float gv = center[c] - o[c]; results[c] += gv; idelta[c] = fmaf(gv, gv, idelta[c]); }
        }
        if (cur > 0) {
            pos_grid[d] = cur - 1;
            float o[C];
            GeVec<T, C>::ld(tab + (uint64_t)ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C, o);
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
// [L, B] units -> [B, L] units (a unit = the C features of one (point, level): 4 or 8 bytes). Thread = point: L coalesced plane reads,
// one contiguous row store. Lets the public [B, L*C] op use the level-major forward kernel (0.39 vs 0.74 ms per 2 M random points).
template <typename U>
__global__ void __launch_bounds__(256) k_planes_to_rows(const U *__restrict__ planes, U *__restrict__ rows, uint32_t B, uint32_t L) {
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        U v[GE_MAX_LEVELS];
#pragma unroll
        for (uint32_t l = 0; l < GE_MAX_LEVELS; l++) if (l < L) v[l] = planes[(uint64_t)l * B + b];
#pragma unroll
        for (uint32_t l = 0; l < GE_MAX_LEVELS; l++) if (l < L) rows[(uint64_t)b * L + l] = v[l];
    }
}

// the inverse: [B, L] units -> [L, B] units (one contiguous row load per thread, L coalesced plane stores) — the permute + copy of
// grid.py:75 as one kernel, so that the binned backward reads the incoming [B, L*C] gradient as planes (0.69 -> 0.58 ms per 2 M points:
// read as rows, every level of the scatter touches one 32-byte sector per point)
template <typename U>
__global__ void __launch_bounds__(256) k_rows_to_planes(const U *__restrict__ rows, U *__restrict__ planes, uint32_t B, uint32_t L) {
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        U v[GE_MAX_LEVELS];
#pragma unroll
        for (uint32_t l = 0; l < GE_MAX_LEVELS; l++) if (l < L) v[l] = rows[(uint64_t)b * L + l];
#pragma unroll
        for (uint32_t l = 0; l < GE_MAX_LEVELS; l++) if (l < L) planes[(uint64_t)l * B + b] = v[l];
    }
}

static int ge_make_levels(uint32_t L, float S, uint32_t H, GeLevels &lv) {
    if (L > GE_MAX_LEVELS) return 1;
    for (uint32_t l = 0; l < L; l++) {
        // gridencoder.cu:138-139, evaluated with the host libm (identical to oracle/oracle.c)
        const float sc = exp2f((float)l * S) * (float)H - 1.0f;
        lv.scale[l] = sc;
        lv.resolution[l] = (uint32_t)ceil((double)sc) + 1;
    }
    for (uint32_t l = L; l < GE_MAX_LEVELS; l++) { lv.scale[l] = 0; lv.resolution[l] = 1; }
    int merge_res = foc_opt(FOC_OPT_GB_MERGE_MAX_RES);                 // <= 1023: the run key packs 10 bits per axis
    if (merge_res > 1023) merge_res = 1023;
    if (merge_res < 0) merge_res = 0;
    lv.merge_max_res = (uint32_t)merge_res;
    return 0;
}

static inline uint32_t ge_xcd_grid(uint32_t chunks, uint32_t L) { return 8u * chunks * ((L + 7u) / 8u); }

// Leading levels encoded by one workgroup per chunk (ge_walk): resolution <= 160 and at most L / 2 of them; fewer than two -> plain
// walk. Measured on FOC's 16-level grid (forward + count, 2 M points): threshold 63 0.358 ms, 120 0.353, 160 0.350 (8 levels),
// 250 without the L / 2 cap 0.353, 600 0.422, every level 0.640. FOC_GRID_FUSE_SMALL=0 switches it off.
static uint32_t ge_small_levels(uint32_t L, const GeLevels &lv) {
    const int on = foc_opt(FOC_OPT_GRID_FUSE_SMALL);
    if (!on) return 0u;
    const uint32_t finest = on > 1 ? (uint32_t)on : 160u;            // a value above 1 is taken as the resolution threshold (A/B runs)
    const uint32_t most = L / 2;                                     // most levels in the shared group (10 / 12 / 14 / 16 of 16 measured: NOTEBOOK.md, rounds 1-4 section 9)
    uint32_t lc = 0;
    while (lc < most && lc < L && lv.resolution[lc] <= finest) lc++;
    return lc >= 2u ? lc : 0u;
}

// FOC_GRID_PAIRS=0: one 4-byte load per corner (A/B runs). 16-byte groups of 4 rows (which would also cover x = 1 mod 4 on a hashed
// level) were measured SLOWER: 0.064 vs 0.052 ms per 2 M random points and level — the 16-byte gather is not free like the 8-byte one.
static bool ge_pairs_enabled() {
    return foc_opt(FOC_OPT_GRID_PAIRS) != 0;
}
// The `pairs` argument of the level-major forward kernels: 0 = one load per corner, 1 = row pairs, 2 = row pairs and the call is of the
// shape ge_forward_hash3 serves (the per-level part of that decision is taken in the kernel). FOC_GRID_FAST=0: never 2 (A/B runs).
static uint32_t ge_pairs_mode(const void *emb, const void *dy_dx, size_t elem, uint32_t D, uint32_t C, uint32_t gridtype, bool ac, uint32_t interp) {
    const int fast = foc_opt(FOC_OPT_GRID_FAST);
    if (!ge_pairs_enabled() || ((uintptr_t)emb & 7u) != 0u || dy_dx) return 0u;
    return (fast && elem == 2 && D == 3 && C == 2 && gridtype == 0u && !ac && interp == 0u) ? 2u : 1u;
}

template <typename T, uint32_t D, uint32_t C>
static int ge_forward_launch(const float *inputs, const void *emb, const int32_t *offsets, void *outputs, uint32_t B, uint32_t L,
                             const GeLevels &lv, void *dy_dx, uint32_t gridtype, bool ac, uint32_t interp, bool bl, hipStream_t st) {
    const int lm_plain = 1;                      // blocks numbered level by level: the whole chip walks one level at a time (pinning a level to one XCD balanced badly)
    if (bl) {
        const uint64_t total = (uint64_t)B * L;
        const uint32_t grid = (uint32_t)((total + 255) / 256 > 0x7FFFFFFFull ? 0x7FFFFFFFull : (total + 255) / 256);
        hipLaunchKernelGGL((k_grid_fwd_bl<T, D, C>), dim3(grid), dim3(256), 0, st, inputs, (const T *)emb, offsets, (T *)outputs, B, L, lv,
                           (T *)dy_dx, gridtype, ac, interp);
    } else {
        const uint32_t chunks = foc_div_up(B, 256);
        const uint32_t lc = lm_plain ? ge_small_levels(L, lv) : 0u;
        const uint32_t groups = lc >= 2u ? L - lc + 1u : L;
        // FOC_GRID_FWD_LDS=<bytes>: unused dynamic LDS per workgroup — caps the resident workgroups per CU (occupancy experiments: what the
        // forward's gathers cost with fewer waves in flight, NOTEBOOK.md, rounds 1-4 section 9)
        const int pad_lds = 0;
        hipLaunchKernelGGL((k_grid_fwd_lbc<T, D, C>), dim3(lm_plain ? chunks * groups : ge_xcd_grid(chunks, L)), dim3(256), (size_t)pad_lds, st, inputs, (const T *)emb, offsets,
                           (T *)outputs, B, L, lv, (T *)dy_dx, gridtype, ac, interp, chunks, (uint32_t)lm_plain,
                           ge_pairs_mode(emb, dy_dx, sizeof(T), D, C, gridtype, ac, interp), lc);
    }
    FOC_CHECK_LAUNCH("grid_encode_forward");
    return FOC_OK;
}

template <typename T, uint32_t D>
static int ge_forward_c(uint32_t C, const float *inputs, const void *emb, const int32_t *offsets, void *outputs, uint32_t B, uint32_t L,
                        const GeLevels &lv, void *dy_dx, uint32_t gridtype, bool ac, uint32_t interp, bool bl, hipStream_t st) {
    switch (C) {
        case 1: return ge_forward_launch<T, D, 1>(inputs, emb, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, bl, st);
        case 2: return ge_forward_launch<T, D, 2>(inputs, emb, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, bl, st);
        case 4: return ge_forward_launch<T, D, 4>(inputs, emb, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, bl, st);
        case 8: return ge_forward_launch<T, D, 8>(inputs, emb, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, bl, st);
        default: foc_set_error("GridEncoding: C must be 1, 2, 4, or 8."); return FOC_E_INVALID;   // gridencoder.cu:381
    }
}

template <typename T>
static int ge_forward_d(uint32_t D, uint32_t C, const float *inputs, const void *emb, const int32_t *offsets, void *outputs, uint32_t B,
                        uint32_t L, const GeLevels &lv, void *dy_dx, uint32_t gridtype, bool ac, uint32_t interp, bool bl, hipStream_t st) {
    switch (D) {
        case 2: return ge_forward_c<T, 2>(C, inputs, emb, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, bl, st);
        case 3: return ge_forward_c<T, 3>(C, inputs, emb, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, bl, st);
        case 4: case 5:                                    // gridencoder.cu:393-398: runtime-D kernels (gridencoder_nd.hip)
            return ge_nd_forward(sizeof(T) == 2 ? FOC_F16 : FOC_F32, D, C, inputs, emb, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, bl, st);
        default: foc_set_error("GridEncoding: D must be 2, 3, 4 or 5 (got %u)", D); return FOC_E_INVALID;
    }
}

static int ge_forward(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B, uint32_t D,
                      uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp,
                      int dtype, bool bl, void *stream) {
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(inputs && embeddings && offsets && outputs, FOC_E_INVALID, "grid_encode_forward: null pointer");
    FOC_REQUIRE(dtype == FOC_F32 || dtype == FOC_F16, FOC_E_DTYPE, "grid_encode_forward: dtype must be FOC_F32 or FOC_F16");
    FOC_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS, FOC_E_INVALID, "grid_encode_forward: L must be in [1,%d]", GE_MAX_LEVELS);
    FOC_REQUIRE(gridtype <= 1 && interp <= 1, FOC_E_INVALID, "grid_encode_forward: bad gridtype/interp");
    if (B == 0) return FOC_OK;
    GeLevels lv;
    ge_make_levels(L, S, H, lv);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FOC_F32) return ge_forward_d<float>(D, C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, align_corners != 0, interp, bl, st);
    return ge_forward_d<__half>(D, C, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, align_corners != 0, interp, bl, st);
}

template <typename T, uint32_t D, uint32_t C>
static int ge_backward_launch(const void *grad, const float *inputs, const int32_t *offsets, void *grad_emb, uint32_t B, uint32_t L,
                              const GeLevels &lv, const void *dy_dx, void *grad_inputs, uint32_t gridtype, bool ac, uint32_t interp,
                              bool bl, hipStream_t st) {
    const uint32_t chunks = foc_div_up(B, 256);
    if (bl) hipLaunchKernelGGL((k_grid_bwd<T, D, C, true>), dim3(ge_xcd_grid(chunks, L)), dim3(256), 0, st, (const T *)grad, inputs, offsets,
                               (T *)grad_emb, B, L, lv, gridtype, ac, interp, chunks);
    else hipLaunchKernelGGL((k_grid_bwd<T, D, C, false>), dim3(ge_xcd_grid(chunks, L)), dim3(256), 0, st, (const T *)grad, inputs, offsets,
                            (T *)grad_emb, B, L, lv, gridtype, ac, interp, chunks);
    FOC_CHECK_LAUNCH("grid_encode_backward");
    if (dy_dx && grad_inputs) {
        const uint32_t g = foc_grid_1d((uint64_t)B * D, 256);
        if (bl) hipLaunchKernelGGL((k_grid_input_bwd<T, D, C, true>), dim3(g), dim3(256), 0, st, (const T *)grad, (const T *)dy_dx, (T *)grad_inputs, B, L);
        else hipLaunchKernelGGL((k_grid_input_bwd<T, D, C, false>), dim3(g), dim3(256), 0, st, (const T *)grad, (const T *)dy_dx, (T *)grad_inputs, B, L);
        FOC_CHECK_LAUNCH("grid_encode_backward(inputs)");
    }
    return FOC_OK;
}

template <typename T, uint32_t D>
static int ge_backward_c(uint32_t C, const void *grad, const float *inputs, const int32_t *offsets, void *grad_emb, uint32_t B, uint32_t L,
                         const GeLevels &lv, const void *dy_dx, void *grad_inputs, uint32_t gridtype, bool ac, uint32_t interp, bool bl,
                         hipStream_t st) {
    switch (C) {
        case 1: return ge_backward_launch<T, D, 1>(grad, inputs, offsets, grad_emb, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
        case 2: return ge_backward_launch<T, D, 2>(grad, inputs, offsets, grad_emb, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
        case 4: return ge_backward_launch<T, D, 4>(grad, inputs, offsets, grad_emb, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
        case 8: return ge_backward_launch<T, D, 8>(grad, inputs, offsets, grad_emb, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
        default: foc_set_error("GridEncoding: C must be 1, 2, 4, or 8."); return FOC_E_INVALID;
    }
}

template <typename T, uint32_t D, uint32_t C>
static int ge_tv_launch(const void *inputs, const void *emb, void *grad, const int32_t *offsets, float weight, uint32_t B, uint32_t L,
                        const GeLevels &lv, uint32_t gridtype, bool ac, hipStream_t st) {
    const uint32_t chunks = foc_div_up(B, 256);
    hipLaunchKernelGGL((k_grad_tv<T, D, C>), dim3(ge_xcd_grid(chunks, L)), dim3(256), 0, st, (const T *)inputs, (const T *)emb, (T *)grad, offsets,
                       weight, B, L, lv, gridtype, ac, chunks);
    FOC_CHECK_LAUNCH("grad_total_variation");
    return FOC_OK;
}

template <typename T, uint32_t D>
static int ge_tv_c(uint32_t C, const void *inputs, const void *emb, void *grad, const int32_t *offsets, float weight, uint32_t B, uint32_t L,
                   const GeLevels &lv, uint32_t gridtype, bool ac, hipStream_t st) {
    switch (C) {
        case 1: return ge_tv_launch<T, D, 1>(inputs, emb, grad, offsets, weight, B, L, lv, gridtype, ac, st);
        case 2: return ge_tv_launch<T, D, 2>(inputs, emb, grad, offsets, weight, B, L, lv, gridtype, ac, st);
        case 4: return ge_tv_launch<T, D, 4>(inputs, emb, grad, offsets, weight, B, L, lv, gridtype, ac, st);
        case 8: return ge_tv_launch<T, D, 8>(inputs, emb, grad, offsets, weight, B, L, lv, gridtype, ac, st);
        default: foc_set_error("GridEncoding: C must be 1, 2, 4, or 8."); return FOC_E_INVALID;
    }
}


// ---- binned backward: host side --------------------------------------------------------------
// records (pairs of corners along x) a pass can produce: 4 per (point, level), 5 when a pair straddles a segment boundary
static uint64_t gb_max_recs(uint32_t B, uint32_t L) { return (uint64_t)B * 5u * L; }
static uint64_t gb_rec_array_bytes(uint64_t m, int dtype) { return ((m + 3) & ~(uint64_t)3) * 4 + m * (dtype == FOC_F16 ? 8 : 16); }
static uint64_t gb_workspace_bytes(uint32_t B, uint32_t L, int dtype) {
    const uint64_t hdr = (sizeof(GbHeader) + 255) & ~(uint64_t)255;
    const uint64_t m = gb_max_recs(B, L);
    const uint64_t wg = (uint64_t)foc_div_up(B, GB_PM_TILE) * L * GB_MAX_SEGS * 4;      // per-workgroup counts / bases of the point-major passes
    return hdr + gb_rec_array_bytes(m, dtype) + 256 + wg + 256;
}
static uint64_t gb_recs_bytes(uint32_t B, uint32_t L, int dtype) {
    const uint64_t m = gb_max_recs(B, L);
    return (gb_rec_array_bytes(m, dtype) + 255) & ~(uint64_t)255;
}

// The gradient-independent half of the pass (counts -> record ranges): needs the sample positions only, so a caller may run it ahead of
// time, e.g. next to the forward pass (foc_grid_encode_backward_count).
static int gb_count(const float *inputs, const int32_t *offsets, uint32_t B, uint32_t L, const GeLevels &lv, uint32_t gridtype, bool ac, uint32_t interp,
                    int dtype, void *workspace, hipStream_t st) {
    GbHeader *hdr = reinterpret_cast<GbHeader *>(workspace);
    void *recs = reinterpret_cast<char *>(workspace) + ((sizeof(GbHeader) + 255) & ~(uint64_t)255);
    if (foc_zero_async(hdr->counts, sizeof(hdr->counts), st) != hipSuccess) { foc_set_error("grid_encode_backward: memset failed"); return FOC_E_LAUNCH; }
    const uint32_t n_wg = foc_div_up(B, GB_PM_TILE);
    uint32_t *wg_hist = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(recs) + gb_recs_bytes(B, L, dtype));
    hipLaunchKernelGGL(k_gbin_count_pt, dim3(n_wg), dim3(GB_PMS_WG), 0, st, inputs, offsets, hdr, wg_hist, B, L, lv, gridtype, ac, interp);
    FOC_CHECK_LAUNCH("grid_encode_backward(count)");
    hipLaunchKernelGGL(k_gbin_scans, dim3(L * GB_MAX_SEGS + 1), dim3(256), 0, st, hdr, wg_hist, n_wg, L);
    FOC_CHECK_LAUNCH("grid_encode_backward(scans)");
    return FOC_OK;
}

// forward [L,B,2] + the count pass in one launch, then the two scan kernels (foc_grid_encode_forward_counted)
template <typename T>
static int gb_forward_counted(const float *inputs, const void *emb, const int32_t *offsets, void *outputs, uint32_t B, uint32_t L, const GeLevels &lv,
                              uint32_t gridtype, bool ac, uint32_t interp, void *workspace, hipStream_t st) {
    GbHeader *hdr = reinterpret_cast<GbHeader *>(workspace);
    void *recs = reinterpret_cast<char *>(workspace) + ((sizeof(GbHeader) + 255) & ~(uint64_t)255);
    if (foc_zero_async(hdr->counts, sizeof(hdr->counts), st) != hipSuccess) { foc_set_error("grid_encode_forward_counted: memset failed"); return FOC_E_LAUNCH; }
    const uint32_t n_tiles = foc_div_up(B, GB_PM_TILE), chunks = foc_div_up(B, 256);
    const uint32_t lc = ge_small_levels(L, lv);
    const uint32_t groups = lc >= 2u ? L - lc + 1u : L;                          // the small levels are one group of workgroups
    const uint32_t fwd_blocks = chunks * groups;
    const uint32_t lo = groups >= 4 ? groups / 2 : 0, hi = groups >= 4 ? groups - 1 : groups;   // groups whose workgroups the counting ones are spread over
    // an ODD period: workgroups go to the 8 XCDs round-robin, so an even one puts every counting workgroup on the same 1, 2 or 4 XCDs
    // (period 16, which any B that is not a multiple of 1024 gave for 4 host groups, made this launch 3-4x slower than for B = k * 1024)
    const uint32_t csplit = L >= 4u ? 4u : 1u, n_count = n_tiles * csplit;
    const uint32_t w0 = lo * chunks, fit = ((hi - lo) * chunks) / n_count + 1u, period = (fit & 1u) ? fit : fit - 1u;
    uint32_t *wg_hist = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(recs) + gb_recs_bytes(B, L, sizeof(T) == 2 ? FOC_F16 : FOC_F32));
    hipLaunchKernelGGL((k_grid_fwd_counted<T>), dim3(fwd_blocks + n_count), dim3(256), 0, st, inputs, (const T *)emb, offsets, (T *)outputs, B, L, lv, gridtype, ac,
                       interp, chunks, sizeof(T) == 2 ? ge_pairs_mode(emb, nullptr, sizeof(T), 3, 2, gridtype, ac, interp) : 0u, hdr, wg_hist, n_tiles, period, w0, lc, csplit);
    FOC_CHECK_LAUNCH("grid_encode_forward_counted");
    hipLaunchKernelGGL(k_gbin_scans, dim3(L * GB_MAX_SEGS + 1), dim3(256), 0, st, hdr, wg_hist, n_tiles, L);
    FOC_CHECK_LAUNCH("grid_encode_forward_counted(scans)");
    return FOC_OK;
}

// Levels whose two-corner records travel in the factored 8-byte form (k_gbin_scatter_pms): fp16 tables, a hashed level (ge_rows3's
// decision, in its uint32 arithmetic) with a power-of-two size of at least one segment, above the run-merging threshold.
// FOC_GB_FACTORED=0 keeps the 12-byte records everywhere (A/B runs).
static uint32_t gb_fact_mask(uint32_t L, const GeLevels &lv, const int32_t *offsets_host, uint32_t gridtype, bool ac, int dtype) {
    const int on = foc_opt(FOC_OPT_GB_FACTORED);
    if (!on || dtype != FOC_F16 || gridtype != 0u) return 0u;
    uint32_t m = 0;
    for (uint32_t l = 0; l < L && l < 32u; l++) {
        const uint32_t size = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        const uint32_t r1 = ac ? lv.resolution[l] : lv.resolution[l] + 1u;
        uint32_t stride = 1u;
        for (int d = 0; d < 3; d++) if (stride <= size) stride *= r1;
        const bool hashed = stride > size;
        if (hashed && (size & (size - 1u)) == 0u && size >= GB_SEG && lv.resolution[l] > lv.merge_max_res) m |= 1u << l;
    }
    return m;
}

// workgroups of k_gbin_scatter_pms one CU holds (fp16 tables: two 1024-thread workgroups, fp32 tables: one)
static uint32_t gb_scatter_resident(size_t elem) {
    static uint32_t cus[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = (uint32_t)n;
    }
    return cus[dev] * (elem == 2 ? 2u : 1u);
}

template <typename T>
static int gb_run(const void *grad, const float *inputs, const int32_t *offsets, void *grad_emb, uint32_t B, uint32_t L, const GeLevels &lv,
                  uint32_t gridtype, bool ac, uint32_t interp, bool bl, void *workspace, bool counted, uint32_t fact_mask, const GbSizes &sz, hipStream_t st) {
    GbHeader *hdr = reinterpret_cast<GbHeader *>(workspace);
    void *recs = reinterpret_cast<char *>(workspace) + ((sizeof(GbHeader) + 255) & ~(uint64_t)255);
    const uint64_t max_recs = gb_max_recs(B, L);
    const uint32_t n_tiles = foc_div_up(B, GB_PM_TILE);
    uint32_t *wg_hist = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(recs) + gb_recs_bytes(B, L, sizeof(T) == 2 ? FOC_F16 : FOC_F32));
    if (!counted) {
        const int rc = gb_count(inputs, offsets, B, L, lv, gridtype, ac, interp, sizeof(T) == 2 ? FOC_F16 : FOC_F32, workspace, st);
        if (rc) return rc;
    }
    // The last, partial round of scatter workgroups is cut into shorter ones (see the kernel): `split` workgroups per tile, as many as
    // still fit one round, so that the launch ends a fraction of a tile's time after its last full round instead of a whole one.
    const uint32_t resident = gb_scatter_resident(sizeof(T));
    const uint32_t n_tail = n_tiles % resident, n_whole = n_tiles - n_tail;
    uint32_t split = 1u;
    const uint32_t split_max = (uint32_t)max(1, foc_opt(FOC_OPT_GB_TAIL_SPLIT));
    while (n_tail && split * 2u <= L && n_tail * split * 2u <= resident && split * 2u <= split_max) split *= 2u;
    hipLaunchKernelGGL((k_gbin_scatter_pms<T>), dim3(n_whole + n_tail * split), dim3(GB_PMS_WG), 0, st, (const T *)grad, inputs, offsets, hdr, wg_hist, recs, max_recs, B, L, lv,
                       gridtype, ac, interp, bl, fact_mask, sz, n_tiles, n_whole, split);
    FOC_CHECK_LAUNCH("grid_encode_backward(scatter)");
    const uint32_t ub = (uint32_t)(((uint64_t)B * 5u * L + GB_CHUNK - 1) / GB_CHUNK) + L * GB_MAX_SEGS;      // chunks in the worst case
    hipLaunchKernelGGL((k_gbin_reduce<T>), dim3(ub), dim3(GB_RTHREADS), 0, st, hdr, recs, max_recs, offsets, (T *)grad_emb, L, fact_mask);
    FOC_CHECK_LAUNCH("grid_encode_backward(reduce)");
    return FOC_OK;
}

extern "C" {

int foc_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B, uint32_t D,
                            uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners,
                            uint32_t interp, int dtype, const int32_t *offsets_host, void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs);
    (void)offsets_host;
    return ge_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, dtype, false, stream);
}

int foc_grid_encode_forward_bl(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B, uint32_t D,
                               uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners,
                               uint32_t interp, int dtype, const int32_t *offsets_host, void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs);
    (void)offsets_host;
    return ge_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, dtype, true, stream);
}

int foc_grid_forward_index_path(uint32_t level_rows, uint32_t resolution, uint32_t level_offset_rows) {
    // which index arithmetic the fp16 / D = 3 / C = 2 / hash-grid forward uses for a level (host mirror of ge_forward_level's decision)
    if (((level_offset_rows | level_rows) & 1u) != 0u) return 0;
    uint32_t st1, st2;
    const bool hashed = ge_level_hashed(level_rows, resolution, st1, st2);
    if (!ge_hash3_admits(level_rows, hashed)) return 0;
    return level_rows <= (1u << 22) ? 2 : 1;
}

int foc_grid_planes_to_rows(const void *planes, void *rows, uint32_t B, uint32_t L, uint32_t unit_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, planes);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(planes && rows, FOC_E_INVALID, "grid_planes_to_rows: null pointer");
    FOC_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS && (unit_bytes == 4 || unit_bytes == 8), FOC_E_INVALID, "grid_planes_to_rows: L in [1,%d], 4- or 8-byte units", GE_MAX_LEVELS);
    hipStream_t st = (hipStream_t)stream;
    if (unit_bytes == 4) hipLaunchKernelGGL((k_planes_to_rows<uint32_t>), dim3(foc_grid_1d(B, 256)), dim3(256), 0, st, (const uint32_t *)planes, (uint32_t *)rows, B, L);
    else hipLaunchKernelGGL((k_planes_to_rows<uint2>), dim3(foc_grid_1d(B, 256)), dim3(256), 0, st, (const uint2 *)planes, (uint2 *)rows, B, L);
    FOC_CHECK_LAUNCH("grid_planes_to_rows");
    return FOC_OK;
}

int foc_grid_rows_to_planes(const void *rows, void *planes, uint32_t B, uint32_t L, uint32_t unit_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, rows);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(planes && rows, FOC_E_INVALID, "grid_rows_to_planes: null pointer");
    FOC_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS && (unit_bytes == 4 || unit_bytes == 8), FOC_E_INVALID, "grid_rows_to_planes: L in [1,%d], 4- or 8-byte units", GE_MAX_LEVELS);
    hipStream_t st = (hipStream_t)stream;
    if (unit_bytes == 4) hipLaunchKernelGGL((k_rows_to_planes<uint32_t>), dim3(foc_grid_1d(B, 256)), dim3(256), 0, st, (const uint32_t *)rows, (uint32_t *)planes, B, L);
    else hipLaunchKernelGGL((k_rows_to_planes<uint2>), dim3(foc_grid_1d(B, 256)), dim3(256), 0, st, (const uint2 *)rows, (uint2 *)planes, B, L);
    FOC_CHECK_LAUNCH("grid_rows_to_planes");
    return FOC_OK;
}

int foc_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets, void *grad_embeddings,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                             uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int grad_is_bl,
                             const int32_t *offsets_host, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad);
    (void)offsets_host; (void)embeddings;
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(grad && inputs && offsets && grad_embeddings, FOC_E_INVALID, "grid_encode_backward: null pointer");
    FOC_REQUIRE(dtype == FOC_F32 || dtype == FOC_F16, FOC_E_DTYPE, "grid_encode_backward: dtype must be FOC_F32 or FOC_F16");
    FOC_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS, FOC_E_INVALID, "grid_encode_backward: L must be in [1,%d]", GE_MAX_LEVELS);
    FOC_REQUIRE(gridtype <= 1 && interp <= 1, FOC_E_INVALID, "grid_encode_backward: bad gridtype/interp");
    if (B == 0) return FOC_OK;
    GeLevels lv;
    ge_make_levels(L, S, H, lv);
    hipStream_t st = (hipStream_t)stream;
    const bool ac = align_corners != 0, bl = grad_is_bl != 0;
    if (dtype == FOC_F32) {
        switch (D) {
            case 2: return ge_backward_c<float, 2>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
            case 3: return ge_backward_c<float, 3>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
        }
    } else {
        switch (D) {
            case 2: return ge_backward_c<__half, 2>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
            case 3: return ge_backward_c<__half, 3>(C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
        }
    }
    if (D == 4 || D == 5)                                      // gridencoder.cu:437-442
        return ge_nd_backward(dtype, D, C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, bl, st);
    foc_set_error("GridEncoding: D must be 2, 3, 4 or 5 (got %u)", D);
    return FOC_E_INVALID;
}

uint64_t foc_grid_encode_backward_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t L, int dtype) {
    if (D != 3 || C != 2 || L > GE_MAX_LEVELS) return 0;       // 0: the binned path does not apply; use foc_grid_encode_backward
    return gb_workspace_bytes(B, L, dtype);
}

static int gb_check(uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, uint32_t interp, int dtype,
                    const int32_t *offsets_host, uint64_t workspace_bytes) {
    FOC_REQUIRE(dtype == FOC_F32 || dtype == FOC_F16, FOC_E_DTYPE, "grid_encode_backward_binned: dtype must be FOC_F32 or FOC_F16");
    FOC_REQUIRE(D == 3 && C == 2, FOC_E_INVALID, "grid_encode_backward_binned: only D=3, C=2 (got D=%u C=%u)", D, C);
    FOC_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS && gridtype <= 1 && interp <= 1, FOC_E_INVALID, "grid_encode_backward_binned: bad L/gridtype/interp");
    FOC_REQUIRE(workspace_bytes >= gb_workspace_bytes(B, L, dtype), FOC_E_INVALID, "grid_encode_backward_binned: workspace too small");
    FOC_REQUIRE((uint64_t)B * 8u * L < (1ull << 32), FOC_E_INVALID, "grid_encode_backward_binned: B*8*L must stay below 2^32 records");
    for (uint32_t l = 0; l < L; l++)
        FOC_REQUIRE((uint32_t)(offsets_host[l + 1] - offsets_host[l]) <= GB_SEG * GB_MAX_SEGS, FOC_E_INVALID,
                    "grid_encode_backward_binned: level %u has more than %u rows", l, GB_SEG * GB_MAX_SEGS);
    // The scatter stages at most 5 two-corner records per point and level: a pair of corners along x splits into two records only when
    // its rows straddle an 8192-row boundary, which on a hash grid at most one pair of a point can do — a dense level is at most
    // 2^19 rows, so the pairs' rows differ by less than 8192 and not by a multiple of it; a hashed level's pairs split only at
    // x = 8191 (mod 8192), which a resolution below that never reaches. Tiled grids and finer levels take the atomic kernel.
    FOC_REQUIRE(gridtype == 0, FOC_E_INVALID, "grid_encode_backward_binned: hash grids only (gridtype 0); use foc_grid_encode_backward");
    GeLevels lv;
    ge_make_levels(L, S, H, lv);
    for (uint32_t l = 0; l < L; l++)
        FOC_REQUIRE(lv.resolution[l] <= GB_SEG - 2u, FOC_E_INVALID, "grid_encode_backward_binned: level %u has resolution %u (> %u)", l,
                    lv.resolution[l], GB_SEG - 2u);
    return FOC_OK;
}

int foc_grid_encode_backward_count(const float *inputs, const int32_t *offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   uint32_t gridtype, int align_corners, uint32_t interp, int dtype, const int32_t *offsets_host, void *workspace,
                                   uint64_t workspace_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(inputs && offsets && workspace && offsets_host, FOC_E_INVALID, "grid_encode_backward_count: null pointer");
    const int rc = gb_check(B, D, C, L, S, H, gridtype, interp, dtype, offsets_host, workspace_bytes);
    if (rc) return rc;
    GeLevels lv;
    ge_make_levels(L, S, H, lv);
    return gb_count(inputs, offsets, B, L, lv, gridtype, align_corners != 0, interp, dtype, workspace, (hipStream_t)stream);
}

int foc_grid_encode_forward_counted(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B, uint32_t D, uint32_t C,
                                    uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                    const int32_t *offsets_host, void *workspace, uint64_t workspace_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(inputs && embeddings && offsets && outputs && workspace && offsets_host, FOC_E_INVALID, "grid_encode_forward_counted: null pointer");
    const int rc = gb_check(B, D, C, L, S, H, gridtype, interp, dtype, offsets_host, workspace_bytes);
    if (rc) return rc;
    GeLevels lv;
    ge_make_levels(L, S, H, lv);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FOC_F32) return gb_forward_counted<float>(inputs, embeddings, offsets, outputs, B, L, lv, gridtype, align_corners != 0, interp, workspace, st);
    return gb_forward_counted<__half>(inputs, embeddings, offsets, outputs, B, L, lv, gridtype, align_corners != 0, interp, workspace, st);
}

static int gb_entry(const void *grad, const float *inputs, const int32_t *offsets, void *grad_embeddings,
                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                    uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int grad_is_bl,
                    const int32_t *offsets_host, void *workspace, uint64_t workspace_bytes, bool counted, void *stream) {
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(grad && inputs && offsets && grad_embeddings && workspace && offsets_host, FOC_E_INVALID, "grid_encode_backward_binned: null pointer");
    {
        const int rc0 = gb_check(B, D, C, L, S, H, gridtype, interp, dtype, offsets_host, workspace_bytes);
        if (rc0) return rc0;
    }
    GeLevels lv;
    ge_make_levels(L, S, H, lv);
    hipStream_t st = (hipStream_t)stream;
    const bool ac = align_corners != 0, bl = grad_is_bl != 0;
    const uint32_t fact_mask = gb_fact_mask(L, lv, offsets_host, gridtype, ac, dtype);
    GbSizes sz;
    for (uint32_t l = 0; l < GE_MAX_LEVELS; l++) sz.size[l] = l < L ? (uint32_t)(offsets_host[l + 1] - offsets_host[l]) : 0u;
    int rc = dtype == FOC_F32 ? gb_run<float>(grad, inputs, offsets, grad_embeddings, B, L, lv, gridtype, ac, interp, bl, workspace, counted, 0u, sz, st)
                              : gb_run<__half>(grad, inputs, offsets, grad_embeddings, B, L, lv, gridtype, ac, interp, bl, workspace, counted, fact_mask, sz, st);
    if (rc) return rc;
    if (dy_dx && grad_inputs) {
        const uint32_t g = foc_grid_1d((uint64_t)B * 3, 256);
        if (dtype == FOC_F32) {
            if (bl) hipLaunchKernelGGL((k_grid_input_bwd<float, 3, 2, true>), dim3(g), dim3(256), 0, st, (const float *)grad, (const float *)dy_dx, (float *)grad_inputs, B, L);
            else hipLaunchKernelGGL((k_grid_input_bwd<float, 3, 2, false>), dim3(g), dim3(256), 0, st, (const float *)grad, (const float *)dy_dx, (float *)grad_inputs, B, L);
        } else {
            if (bl) hipLaunchKernelGGL((k_grid_input_bwd<__half, 3, 2, true>), dim3(g), dim3(256), 0, st, (const __half *)grad, (const __half *)dy_dx, (__half *)grad_inputs, B, L);
            else hipLaunchKernelGGL((k_grid_input_bwd<__half, 3, 2, false>), dim3(g), dim3(256), 0, st, (const __half *)grad, (const __half *)dy_dx, (__half *)grad_inputs, B, L);
        }
        FOC_CHECK_LAUNCH("grid_encode_backward(inputs)");
    }
    return FOC_OK;
}

int foc_grid_encode_backward_binned(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets, void *grad_embeddings,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                                    uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int grad_is_bl,
                                    const int32_t *offsets_host, void *workspace, uint64_t workspace_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad);
    (void)embeddings;
    return gb_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp, dtype, grad_is_bl,
                    offsets_host, workspace, workspace_bytes, false, stream);
}

int foc_grid_encode_backward_binned_counted(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets, void *grad_embeddings,
                                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                                            uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int grad_is_bl,
                                            const int32_t *offsets_host, void *workspace, uint64_t workspace_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad);
    (void)embeddings;
    return gb_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp, dtype, grad_is_bl,
                    offsets_host, workspace, workspace_bytes, true, stream);
}

int foc_grad_total_variation(const void *inputs, const void *embeddings, void *grad, const int32_t *offsets, float weight, uint32_t B,
                             uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype,
                             void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(inputs && embeddings && grad && offsets, FOC_E_INVALID, "grad_total_variation: null pointer");
    FOC_REQUIRE(dtype == FOC_F32 || dtype == FOC_F16, FOC_E_DTYPE, "grad_total_variation: dtype must be FOC_F32 or FOC_F16");
    FOC_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS, FOC_E_INVALID, "grad_total_variation: L must be in [1,%d]", GE_MAX_LEVELS);
    if (B == 0) return FOC_OK;
    GeLevels lv;
    ge_make_levels(L, S, H, lv);
    hipStream_t st = (hipStream_t)stream;
    const bool ac = align_corners != 0;
    if (dtype == FOC_F32) {
        switch (D) {
            case 2: return ge_tv_c<float, 2>(C, inputs, embeddings, grad, offsets, weight, B, L, lv, gridtype, ac, st);
            case 3: return ge_tv_c<float, 3>(C, inputs, embeddings, grad, offsets, weight, B, L, lv, gridtype, ac, st);
        }
    } else {
        switch (D) {
            case 2: return ge_tv_c<__half, 2>(C, inputs, embeddings, grad, offsets, weight, B, L, lv, gridtype, ac, st);
            case 3: return ge_tv_c<__half, 3>(C, inputs, embeddings, grad, offsets, weight, B, L, lv, gridtype, ac, st);
        }
    }
    if (D == 4 || D == 5) return ge_nd_tv(dtype, D, C, inputs, embeddings, grad, offsets, weight, B, L, lv, gridtype, ac, st);      // gridencoder.cu:633-638
    foc_set_error("GridEncoding: D must be 2, 3, 4 or 5 (got %u)", D);
    return FOC_E_INVALID;
}

} // extern "C"
