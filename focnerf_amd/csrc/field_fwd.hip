// field_fwd.hip — the TRAINING forward of the whole field in one kernel: sigma network on the encoder's planes -> h [B,16] -> colour network
// fed from h and a per-ray SH table -> colour logits. nerf/network_ff.py:51-75 between the encoder and the compositing, as the two FFMLP calls
// `foc_ffmlp_forward_planar` + `foc_color_head_forward` compute it (ffmlp/src/ffmlp.cu:331-407, kernel_mlp_fused, twice) — here back to back in
// registers: h is written once (the tail and the backward read it) and NOT read back for the colour network (-32 B per sample, one launch).
//
// Bit for bit the two kernels' results. Every MFMA sees the operands, in the k positions and in the order, that k_mlp_fwd<64, 1, false, 1>
// (planar) and k_mlp_fwd<64, 1, false, 2> (head) give it. The one thing the head kernel does through memory is the colour network's second
// k-chunk: columns 1..16 of the h row (the row shifted by one half, a zero shifted in). Here the sigma network's output tile holds the row in
// the accumulator layout — lane c has rows {0-3, 8-11} of sample c, lane c + 32 rows {4-7, 12-15} — so the two lane halves swap four packed
// registers (v_permlane32_swap) and shift: the same 8 halves per lane that ld_head8 loads.
#include "mlp_common.h"

typedef uint32_t ff_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t ff_pack(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 p = {(_Float16)a, (_Float16)b};               // v_cvt_pk_f16_f32, round to nearest even: the rounding of store_tile
    return __builtin_bit_cast(uint32_t, p);
}

#ifndef FF_MIN_WAVES
#define FF_MIN_WAVES 2                 // A/B builds: tools/build_variant.sh ... -DFF_MIN_WAVES=4
#endif
// RELU as a compile-time fact: with the runtime flag every fragment conversion became a branch of its own (forty basic blocks per tile, nothing
// scheduled across them: 97 us per 2 M rows against NN)
template <int NLS, int NLC, bool RELU>
__global__ void __launch_bounds__(MLP_BLOCK, FF_MIN_WAVES) k_field_fwd_train(const _Float16 *__restrict__ planes, const _Float16 *__restrict__ w_sigma,
                                                               const _Float16 *__restrict__ w_color, _Float16 *__restrict__ h_out,
                                                               _Float16 *__restrict__ c_out, uint32_t B, MlpHead hd) {
    constexpr int HIDDEN = 64, MT = 2, KC = 4, KS0 = 2;
    constexpr int FS = MT * KS0 + (NLS - 1) * MT * KC + KC;     // fragments of the sigma image: layer 0 | hidden | out
    constexpr int FC = MT * KS0 + (NLC - 1) * MT * KC + KC;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    _Float16 *ldsS = lds, *ldsC = lds + FS * 512;
    const float *obj_bias = reinterpret_cast<const float *>(ldsC + FC * 512);
    stage_weights_fwd<HIDDEN>(w_sigma, ldsS, 32, NLS);
    stage_weights_fwd<HIDDEN>(w_color, ldsC, 32, NLC, true, head_ld0(hd));
    if (hd.obj) stage_obj_bias(w_color, hd.obj, const_cast<float *>(obj_bias), HIDDEN);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = (B + 31) / 32;
    // The rows of a tile are asked for one tile ahead (both k-chunks of the planes: eight dwords per lane, and the ray's SH row): a wave otherwise
    // waits out one memory latency per k-chunk and tile with nothing of its own to run meanwhile (96 -> NN us per 2 M rows, NOTEBOOK). Every load
    // is unconditional — the tile index is clamped to the last tile, whose rows are clamped to B - 1.
    const uint32_t tstride = gridDim.x * MLP_WAVES;
    uint32_t tile = blockIdx.x * MLP_WAVES + wave;
    h8 x_nxt[KS0], sh_nxt;
    auto fetch = [&](uint32_t t) {
#ifdef FF_TIMING_NO_LOAD
        const uint64_t r = (uint64_t)c + 0 * t;                                        // timing build: every tile reads the first 32 rows
#else
        const uint64_t r = min((uint64_t)min(t, n_tiles - 1u) * 32 + c, (uint64_t)B - 1);
#endif
#pragma unroll
        for (int kc = 0; kc < KS0; kc++) x_nxt[kc] = ld_planar8(planes, B, r, kc, h);
        sh_nxt = ld_head8(planes, hd, r, 0, h);
    };
    fetch(tile);
    for (; tile < n_tiles; tile += tstride) {
        // The weight fragments are re-read from LDS in every tile: with both layer counts compile-time the compiler otherwise hoists all 40 fragment
        // loads out of this loop (160 registers: 229 VGPRs, two waves per SIMD — or, under a tighter budget, spills them to scratch: 220 / 287 us for
        // 2 M rows at 3 / 4 waves against 98). A compiler-level memory barrier per tile keeps them where k_mlp_fwd has them.
#ifndef FF_TIMING_HOIST
        asm volatile("" ::: "memory");
#endif
        const uint64_t row_raw = (uint64_t)tile * 32 + c;
        h8 x_cur[KS0];
#pragma unroll
        for (int kc = 0; kc < KS0; kc++) x_cur[kc] = x_nxt[kc];
        const h8 sh = sh_nxt;
        fetch(tile + tstride);
        f16v acc[MT];
        // ---- sigma network, layer 0 from the planes: k-chunk 0 starts from the zero accumulator, k_mlp_fwd's order of operations
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mt][e] = 0.0f;
#pragma unroll
        for (int kc = 0; kc < KS0; kc++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = mfma16(ld_frag(ldsS, mt * KS0 + kc, lane), x_cur[kc], acc[mt]);
        f16v o;
        for (uint32_t l = 1; l <= (uint32_t)NLS; l++) {
            h8 bf[KC];
#pragma unroll
            for (int kc = 0; kc < KC; kc++) bf[kc] = acc_to_frag<RELU>(acc[kc >> 1], kc & 1);
            if (l < (uint32_t)NLS) {
                const uint32_t fbase = MT * KS0 + (l - 1) * MT * KC;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[mt][e] = 0.0f;
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) acc[mt] = mfma16(ld_frag(ldsS, fbase + mt * KC + kc, lane), bf[kc], acc[mt]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; e++) o[e] = 0.0f;
#pragma unroll
                for (int kc = 0; kc < KC; kc++) o = mfma16(ld_frag(ldsS, FS - KC + kc, lane), bf[kc], o);
            }
        }
        // ---- h: rounded to half once; the row goes to memory (two 8-byte stores per lane, store_tile's layout) and stays on the lanes
        uint32_t P0 = ff_pack(o[0], o[1]), P1 = ff_pack(o[2], o[3]), P2 = ff_pack(o[4], o[5]), P3 = ff_pack(o[6], o[7]);
#ifdef FF_TIMING_NO_STORE
        if (row_raw == 0xFFFFFFFFFFull) {
#else
        if (row_raw < B) {
#endif
            *reinterpret_cast<ff_u2 *>(h_out + row_raw * 16 + 4 * h) = ff_u2{P0, P1};
            *reinterpret_cast<ff_u2 *>(h_out + row_raw * 16 + 8 + 4 * h) = ff_u2{P2, P3};
        }
        // lane half 0 holds (h0 h1)(h2 h3)(h8 h9)(h10 h11), half 1 (h4 h5)(h6 h7)(h12 h13)(h14 h15): after the two swaps half 0 has columns 0..7
        // and half 1 columns 8..15 in natural order; the dword behind half 0's eight is its own old P2 = (h8 h9)
        const uint32_t nxt = h == 0 ? P2 : 0u;
        const ff_u2 s02 = __builtin_amdgcn_permlane32_swap(P0, P2, false, false);      // P0[32:63] <-> P2[0:31]
        const ff_u2 s13 = __builtin_amdgcn_permlane32_swap(P1, P3, false, false);
        const h8 hk = head_shift(u32x4{s02.x, s13.x, s02.y, s13.y}, nxt);
        // ---- colour network: layer 0 = [SH16 | h[1:16] | 0] (+ the object feature's share as the accumulators' start), k_mlp_fwd's head form
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            if (hd.obj) acc[mt] = ld_obj_bias(obj_bias, mt, h);
            else {
#pragma unroll
                for (int e = 0; e < 16; e++) acc[mt][e] = 0.0f;
            }
        }
#pragma unroll
        for (int kc = 0; kc < KS0; kc++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = mfma16(ld_frag(ldsC, mt * KS0 + kc, lane), kc == 0 ? sh : hk, acc[mt]);
        for (uint32_t l = 1; l <= (uint32_t)NLC; l++) {
            h8 bf[KC];
#pragma unroll
            for (int kc = 0; kc < KC; kc++) bf[kc] = acc_to_frag<RELU>(acc[kc >> 1], kc & 1);
            if (l < (uint32_t)NLC) {
                const uint32_t fbase = MT * KS0 + (l - 1) * MT * KC;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[mt][e] = 0.0f;
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) acc[mt] = mfma16(ld_frag(ldsC, fbase + mt * KC + kc, lane), bf[kc], acc[mt]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; e++) o[e] = 0.0f;
#pragma unroll
                for (int kc = 0; kc < KC; kc++) o = mfma16(ld_frag(ldsC, FC - KC + kc, lane), bf[kc], o);
                if (hd.out_width == 4u) {                   // only the rgb logits (+ one pad column) exist in memory: lanes of half 0 hold neurons 0..3
#ifdef FF_TIMING_NO_STORE
                    if (h == 0 && row_raw == 0xFFFFFFFFFFull) *reinterpret_cast<ff_u2 *>(c_out + row_raw * 4) = ff_u2{ff_pack(o[0], o[1]), ff_pack(o[2], o[3])};
#else
                    if (h == 0 && row_raw < B) *reinterpret_cast<ff_u2 *>(c_out + row_raw * 4) = ff_u2{ff_pack(o[0], o[1]), ff_pack(o[2], o[3])};
#endif
                } else store_tile<false>(c_out, 16, row_raw, B, 0, 16, o, h);
            }
        }
    }
}

static uint32_t ff_num_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return (uint32_t)n;
}

template <int NLS, int NLC, bool RELU>
static int ff_launch(const void *planes, const void *w_sigma, const void *w_color, void *h, void *c, uint32_t B, const MlpHead &hd, hipStream_t st) {
    auto kern = k_field_fwd_train<NLS, NLC, RELU>;
    const size_t lds = (size_t)((4 + (NLS - 1) * 8 + 4) + (4 + (NLC - 1) * 8 + 4)) * 1024 + 256;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static int resident = 0;                                 // workgroups of this kernel one CU holds (registers + LDS): the grid is capped at CUs x that
    if (!resident) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, MLP_BLOCK, lds) != hipSuccess || n < 1) n = 2;
        resident = n > 8 ? 8 : n;
    }
    uint32_t grid = foc_div_up(foc_div_up(B, 32), MLP_WAVES);
    const uint32_t cap = ff_num_cus() * (uint32_t)resident;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_BLOCK), lds, st, (const _Float16 *)planes, (const _Float16 *)w_sigma, (const _Float16 *)w_color,
                       (_Float16 *)h, (_Float16 *)c, B, hd);
    FOC_CHECK_LAUNCH("field_forward_train");
    return FOC_OK;
}

template <int NLS, int NLC>
static int ff_launch_act(const void *planes, const void *w_sigma, const void *w_color, void *h, void *c, uint32_t B, int relu, const MlpHead &hd, hipStream_t st) {
    return relu ? ff_launch<NLS, NLC, true>(planes, w_sigma, w_color, h, c, B, hd, st) : ff_launch<NLS, NLC, false>(planes, w_sigma, w_color, h, c, B, hd, st);
}

extern "C" {

int foc_field_forward_train(const void *planes, const void *sigma_weights, uint32_t sigma_layers, const void *ray_sh, uint32_t samples_per_ray,
                            const void *color_weights, uint32_t color_layers, uint32_t hidden_dim, uint32_t activation, uint32_t B, void *h, void *c,
                            uint32_t out_width, const void *obj_feat, void *stream) {
    FocDeviceGuard foc_guard_(stream, planes);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(planes && sigma_weights && ray_sh && color_weights && h && c, FOC_E_INVALID, "field_forward_train: null pointer");
    FOC_REQUIRE(hidden_dim == 64 && samples_per_ray >= 1, FOC_E_INVALID, "field_forward_train: hidden_dim must be 64 (got %u), samples_per_ray >= 1", hidden_dim);
    FOC_REQUIRE(activation == FOC_ACT_RELU || activation == FOC_ACT_NONE, FOC_E_INVALID, "field_forward_train: hidden activation must be relu(0) or none(6)");
    FOC_REQUIRE(out_width == 16 || out_width == 4, FOC_E_INVALID, "field_forward_train: out_width must be 16 or 4 (got %u)", out_width);
    const MlpHead hd{(const _Float16 *)ray_sh, nullptr, samples_per_ray, out_width, (const _Float16 *)obj_feat};
    const int relu = activation == FOC_ACT_RELU;
    hipStream_t st = (hipStream_t)stream;
    switch (sigma_layers * 10 + color_layers) {
        case 22: return ff_launch_act<2, 2>(planes, sigma_weights, color_weights, h, c, B, relu, hd, st);
        case 23: return ff_launch_act<2, 3>(planes, sigma_weights, color_weights, h, c, B, relu, hd, st);
        case 33: return ff_launch_act<3, 3>(planes, sigma_weights, color_weights, h, c, B, relu, hd, st);
        default: foc_set_error("field_forward_train: layer counts (%u, %u) are not built (2/2, 2/3, 3/3)", sigma_layers, color_layers); return FOC_E_INVALID;
    }
}

} // extern "C"
