// ffmlp_bwd_priv.hip — fused MLP backward (activation gradients + weight gradients, activations re-evaluated) with WAVE-PRIVATE
// weight-gradient tiles: no workgroup barrier inside the batch loop.
//
// Semantics: ffmlp/src/ffmlp.cu of the reference — kernel_mlp_fused_backward (:410-518) and the weight-gradient GEMMs (:749-895);
// same results as k_mlp_bwd_fused (ffmlp.hip) up to the order of the fp32 batch sums of dW.
//
// Why a second form. k_mlp_bwd_fused gives every wave of a workgroup ONE 32 x 32 tile of each layer's dW over the rows of all four
// waves: 64 accumulator registers, two waves per SIMD — and two workgroup barriers per layer (tiles written by four waves, read by
// four waves), four transposed LDS reads per weight-gradient MFMA (no operand is shared between a wave's MFMAs: the LDS array's
// 256 B/clk are exactly what 4 x 512 B per 32-cycle MFMA ask for), and a wave stalled half of its time at barriers and on LDS
// (SQ counters, profiles/r04_bench_pmc_sq.csv). Here a wave owns its 32 NB rows END TO END:
//   * it accumulates EVERY tile of every layer's dW for its own rows (4 NL tiles = 128 / 192 accumulator registers for 2 / 3
//     hidden layers), so its delta / activation tiles in LDS are private: written and read back transposed by the same wave, ordered
//     by the LDS queue itself — no s_barrier, no cross-wave skew, and a transposed operand serves two MFMAs (2 instead of 4 LDS
//     reads per weight-gradient MFMA);
//   * one wave per SIMD (up to 512 registers), NB independent 32-row chains interleaved in one instruction stream to cover the
//     MFMA -> convert -> LDS -> MFMA dependencies the second wave used to cover;
//   * LDS tiles XOR-swizzled by (row >> 4) on the 8-byte granule so that the chained-layout writes (32 rows x 8 bytes at a
//     36-dword row stride: rows r and r + 16 met in one bank) and the transposed reads are both conflict-free;
//   * the waves' accumulators meet once, after the loop, in LDS; one slot per WORKGROUP goes to k_mlp_dw_reduce.
// Shapes: hidden 64, input width 32, 2 or 3 hidden layers, ReLU, activations re-evaluated (no forward / backward buffer) — the two
// networks of nerf/network_ff.py and FOC's colour head. Everything else stays on k_mlp_bwd_fused.
#include "mlp_common.h"

#define PRIV_WD 72          // halves per row of the delta / activation tiles: 64 + 8 (36 dwords)
#define PRIV_WD0 40         // halves per row of the output-gradient tile: 16 + 16 zero columns + 8 (20 dwords)
#define PRIV_TILE_HALFS (32 * PRIV_WD0 + 2 * 32 * PRIV_WD)

__device__ __forceinline__ void priv_lds_order() {
    // the wave's own LDS writes before its transposed reads of OTHER lanes' data: the LDS queue keeps a wave's operations in order, the
    // compiler must keep them in program order too
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Every MFMA of this kernel is the compiler's builtin. With a 512-register budget hipcc (ROCm 7.2) selects the AGPR form for all of them, keeps
// the chain accumulators in a[..] too and moves them to VGPRs for the conversions (v_accvgpr_read: 16 per 32 x 32 tile). Pinning the
// weight-gradient tiles to AGPRs through inline asm ("+a") was tried and is WRONG here: the register allocator still moves tiles between
// their VGPR and AGPR homes around the asm, and a v_accvgpr_read it places behind an MFMA it cannot see carries no wait states (MFMA -> VALU
// read hazard: 3 % of the weight gradients came back stale, tests/test_gpu_ffmlp.py::test_backward_recompute_is_bit_identical).
__device__ __forceinline__ void priv_mfma_acc(f16v &d, const h8 a, const h8 b) { d = mfma16(a, b, d); }
__device__ __forceinline__ void priv_acc_settle(f16v &d) { (void)d; }

__device__ __forceinline__ h8 priv_tr_pair(const _Float16 *p, int row2) {
    const s4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(p));
    const s4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(p + row2));
    const u32x2 A0 = __builtin_bit_cast(u32x2, a0), A1 = __builtin_bit_cast(u32x2, a1);
    return __builtin_bit_cast(h8, (u32x4){A0.x, A0.y, A1.x, A1.y});
}

// IMODE: 0 = [B,32] rows, 1 = planar [16][B] half2 (the encoder's [L,B,C]), 2 = colour head with [B,16] output gradients, 3 = colour head with
// [B,4] output gradients (MlpHead, mlp_common.h)
template <int NL, int IMODE, int NB>
__global__ void __launch_bounds__(MLP_BLOCK, 1) k_mlp_bwd_priv(const _Float16 *__restrict__ grad, const _Float16 *__restrict__ inputs,
                                                            const _Float16 *__restrict__ weights, _Float16 *__restrict__ grad_inputs,
                                                            float *__restrict__ ws, uint32_t B, MlpHead hd) {
    constexpr int HIDDEN = 64, MT = 2, KC = 4, IN = 32, KS0 = 2, WD = PRIV_WD, WD0 = PRIV_WD0;
    constexpr bool planar = IMODE == 1, HEAD = IMODE >= 2, NARROW = IMODE == 3;
    constexpr int NT = 4 * NL;                             // tiles of dW a wave accumulates: 2 (output stage) + 4 (NL - 1) + 2 (input stage)
    constexpr int F_BWD = MT + (NL - 1) * MT * KC + KC;    // backward image: out [mt] | hidden [l][mt][kc] | dX [kc]
    constexpr int F_FWD = MT * KS0 + (NL - 1) * MT * KC;   // forward image: layer 0 [mt][kc] | hidden [l][mt][kc]
    f16v FZ;
#pragma unroll
    for (int e = 0; e < 16; e++) FZ[e] = 0.0f;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const bool with_dx = grad_inputs != nullptr;
    uint32_t ld0 = IN;
    bool has_obj = false;
    if constexpr (HEAD) { ld0 = head_ld0(hd); has_obj = hd.obj != nullptr; }
    _Float16 *ldsB = lds, *ldsF = lds + F_BWD * 512;
    float *obj_bias = reinterpret_cast<float *>(ldsF + F_FWD * 512);
    _Float16 *tiles = ldsF + F_FWD * 512 + 128;
    stage_weights_bwd<HIDDEN>(weights, ldsB, IN, NL, with_dx, HEAD, ld0);
    stage_weights_fwd<HIDDEN>(weights, ldsF, IN, NL, false, ld0);
    if constexpr (HEAD) { if (has_obj) stage_obj_bias(weights, hd.obj, obj_bias, HIDDEN); }

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5, swc = (c >> 4) & 1;
    const int q = (lane & 15) >> 2, p = lane & 3, cg = 16 * ((lane >> 4) & 1);
    _Float16 *myT = tiles + (size_t)(wave * NB) * PRIV_TILE_HALFS;
    // columns 16..31 of the output-gradient tiles are zeros for good (the 16-wide output stage fills a 32-row MFMA operand)
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
        *reinterpret_cast<h8 *>(myT + nb * PRIV_TILE_HALFS + c * WD0 + 16 + 8 * h) = h8{0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();

    // lane constants of the tile accesses (halves). Granule = 8 bytes; a row's granule g lives at g ^ ((row >> 4) & 1).
    const uint32_t wD0 = c * WD0 + 8 * h;                                 // output gradient row, natural 8-half chunk
    const uint32_t wCh = c * WD + 4 * (h ^ swc);                          // chained fragment: + 16 kc (elements 0..3), + 16 kc + 8 (4..7)
    const uint32_t wXlo = c * WD + 8 * h + 4 * swc, wXhi = c * WD + 8 * h + 4 * (1 - swc);     // natural 8-half chunk: + 16 kc
    const uint32_t rT0 = (4 * q + h) * WD + cg + 4 * p, rT1 = (4 * q + h) * WD + cg + 4 * (p ^ 1);   // transposed reads, k step 0 / 1
    const uint32_t rD0 = (4 * q + h) * WD0 + cg + 4 * p;

    f16v dwacc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) dwacc[t] = FZ;

    const uint32_t rows_per_tile = 32 * NB;
    const uint32_t n_tiles = (B + rows_per_tile - 1) / rows_per_tile;
    const uint32_t tstride = gridDim.x * MLP_WAVES;
    uint32_t tile = blockIdx.x * MLP_WAVES + wave;

    // prefetch registers: every load unconditional (rows clamped to B - 1, a readable dummy for a null pointer), zeros applied when they become current
    h8 g_nxt[NB];
    h8 x_nxt[KS0][NB];
    _Float16 h0_nxt[NB];
    uint32_t hx_nxt[NB];
    auto fetch = [&](uint32_t t) {
        const uint64_t r0 = (uint64_t)t * rows_per_tile;
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const uint64_t row = min(r0 + nb * 32 + c, (uint64_t)B - 1);
            if constexpr (HEAD) {
                const _Float16 *ph = hd.grad_h0 ? hd.grad_h0 + row : grad;
                h0_nxt[nb] = *ph;
            }
            if constexpr (NARROW) {                          // [B,4] output gradients: columns 4..15 are zeros that were never written
                const uint2 v = *reinterpret_cast<const uint2 *>(grad + row * 4);
                g_nxt[nb] = __builtin_bit_cast(h8, (u32x4){v.x, v.y, 0u, 0u});
            } else g_nxt[nb] = *reinterpret_cast<const h8 *>(grad + row * 16 + 8 * h);
#pragma unroll
            for (int kc = 0; kc < KS0; kc++) {
                h8 v;
                if constexpr (HEAD) {
                    if (kc == 0) v = ld_head8(inputs, hd, row, 0, h);
                    else { u32x4 raw; ld_head_raw_all(inputs, row, h, raw, hx_nxt[nb]); v = __builtin_bit_cast(h8, raw); }
                } else v = planar ? ld_planar8(inputs, B, row, kc, h) : *reinterpret_cast<const h8 *>(inputs + row * IN + 16 * kc + 8 * h);
                x_nxt[kc][nb] = v;
            }
        }
    };
    if (tile < n_tiles) fetch(tile);
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the first tile's rows are waited for once, in front of the loop

    for (; tile < n_tiles; tile += tstride) {
        const uint64_t row0 = (uint64_t)tile * rows_per_tile;
        f16v acc[MT][NB];
        h8 bf[KC][NB];
        h8 fa[NL][KC][NB];
        h8 g_cur[NB];
        h8 x_cur[KS0][NB];
        _Float16 h0_cur[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const bool live = row0 + nb * 32 + c < B && !(NARROW && h == 1), live_hi = live && !NARROW;
            const u32x4 gv = __builtin_bit_cast(u32x4, g_nxt[nb]);
            g_cur[nb] = __builtin_bit_cast(h8, (u32x4){live ? gv.x : 0u, live ? gv.y : 0u, live_hi ? gv.z : 0u, live_hi ? gv.w : 0u});
            x_cur[0][nb] = x_nxt[0][nb];
            if constexpr (HEAD) {
                x_cur[1][nb] = head_shift(__builtin_bit_cast(u32x4, x_nxt[1][nb]), h == 0 ? hx_nxt[nb] : 0u);
                h0_cur[nb] = (hd.grad_h0 && row0 + nb * 32 + c < B) ? h0_nxt[nb] : (_Float16)0;
            } else x_cur[1][nb] = x_nxt[1][nb];
        }
        if (tile + tstride < n_tiles) fetch(tile + tstride);

        // ---- forward re-evaluation (k_mlp_fwd's order of operations: the activations are the bits the forward pass saw)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            f16v b0 = FZ;
            if constexpr (HEAD) { if (has_obj) b0 = ld_obj_bias(obj_bias, mt, h); }
#pragma unroll
            for (int nb = 0; nb < NB; nb++) acc[mt][nb] = b0;
        }
#pragma unroll
        for (int kc = 0; kc < KS0; kc++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const h8 a = ld_frag(ldsF, mt * KS0 + kc, lane);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, x_cur[kc][nb], (kc == 0 && !HEAD) ? FZ : acc[mt][nb]);
            }
#pragma unroll
        for (int l = 0; l < NL; l++) {
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
#pragma unroll
                for (int nb = 0; nb < NB; nb++) fa[l][kc][nb] = acc_to_frag<true>(acc[kc >> 1][nb], kc & 1);
            if (l + 1 < NL) {
                const uint32_t fbase = MT * KS0 + l * MT * KC;
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const h8 a = ld_frag(ldsF, fbase + mt * KC + kc, lane);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, fa[l][kc][nb], kc == 0 ? FZ : acc[mt][nb]);
                    }
            }
        }

        // ---- backward stages: s = 0 the 16-wide output layer, s = 1 .. NL the hidden layers from the last to the first
#pragma unroll
        for (int s = 0; s <= NL; s++) {
            const int NTi = s < NL ? 2 : 1, MTo = s == 0 ? 1 : 2;                      // 32-wide tiles along the layer's inputs / outputs
            const int tbase = s == 0 ? 0 : 2 + 4 * (s - 1);
            // delta of this stage -> bf (chained layout, ReLU-gated by the layer's own activation) and the wave's D tile
            if (s == 0) {
#pragma unroll
                for (int nb = 0; nb < NB; nb++) *reinterpret_cast<h8 *>(myT + nb * PRIV_TILE_HALFS + wD0) = g_cur[nb];
            } else {
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        bf[kc][nb] = relu_gate(acc_to_frag<false>(acc[kc >> 1][nb], kc & 1), fa[NL - s][kc][nb]);
                        const h8 v = bf[kc][nb];
                        _Float16 *dst = myT + nb * PRIV_TILE_HALFS + 32 * WD0 + wCh + 16 * kc;
                        *reinterpret_cast<h4 *>(dst) = h4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<h4 *>(dst + 8) = h4{v[4], v[5], v[6], v[7]};
                    }
            }
            // the layer's input -> the wave's A tile
            if (s < NL) {
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const h8 v = fa[NL - 1 - s][kc][nb];
                        _Float16 *dst = myT + nb * PRIV_TILE_HALFS + 32 * WD0 + 32 * WD + wCh + 16 * kc;
                        *reinterpret_cast<h4 *>(dst) = h4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<h4 *>(dst + 8) = h4{v[4], v[5], v[6], v[7]};
                    }
            } else {
#pragma unroll
                for (int kc = 0; kc < KS0; kc++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        h8 v = x_cur[kc][nb];
                        // object feature: column 31 of the input tile (a zero of the shifted h row) becomes 1, so that dW0[:, 31] = sum_b delta_0
                        if constexpr (HEAD) { if (kc == 1 && has_obj && h == 1) v[7] = (_Float16)1.0f; }
                        _Float16 *dst = myT + nb * PRIV_TILE_HALFS + 32 * WD0 + 32 * WD + 16 * kc;
                        *reinterpret_cast<h4 *>(dst + wXlo) = h4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<h4 *>(dst + wXhi) = h4{v[4], v[5], v[6], v[7]};
                    }
            }
            priv_lds_order();
            // ---- input gradients (last stage) before the weight-gradient MFMAs: the stores have the rest of the tile to be acknowledged
            if (s == NL && with_dx) {
                f16v x[NB];
#pragma unroll
                for (int kc = 0; kc < KC; kc++) {
                    const h8 a = ld_frag(ldsB, MT + (NL - 1) * MT * KC + kc, lane);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) x[nb] = mfma16(a, bf[kc][nb], kc == 0 ? FZ : x[nb]);
                }
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const uint64_t row = row0 + nb * 32 + c;
                    if constexpr (HEAD) {
                        // rows 16..31 of the tile = gradient of h columns 0..15 (staged shifted); column 0 comes from the density path
                        if (row < B) {
                            h4 lo = {(_Float16)x[nb][8], (_Float16)x[nb][9], (_Float16)x[nb][10], (_Float16)x[nb][11]};
                            const h4 hi = {(_Float16)x[nb][12], (_Float16)x[nb][13], (_Float16)x[nb][14], (_Float16)x[nb][15]};
                            if (h == 0) lo[0] = h0_cur[nb];
                            *reinterpret_cast<h4 *>(grad_inputs + row * 16 + 4 * h) = lo;
                            *reinterpret_cast<h4 *>(grad_inputs + row * 16 + 8 + 4 * h) = hi;
                        }
                    } else if constexpr (planar) {
                        // [16][B] half2 planes (the encoder's [L,B,C] gradient layout): register quad q4 = features 8 q4 + 4 h .. + 3
                        if (row < B) {
                            uint32_t *gp = reinterpret_cast<uint32_t *>(grad_inputs);
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) {
                                const uint32_t col = 8 * q4 + 4 * h;
                                const h4 v = {(_Float16)x[nb][4 * q4], (_Float16)x[nb][4 * q4 + 1], (_Float16)x[nb][4 * q4 + 2], (_Float16)x[nb][4 * q4 + 3]};
                                const u32x2 w = __builtin_bit_cast(u32x2, v);
                                gp[(uint64_t)(col / 2) * B + row] = w.x;
                                gp[(uint64_t)(col / 2 + 1) * B + row] = w.y;
                            }
                        }
                    } else store_tile<false>(grad_inputs, IN, row, B, 0, IN, x[nb], h);
                }
            }
            // ---- dW_s += D_s^T A_s over this wave's rows: every tile of the stage, operands read transposed once per k step
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const _Float16 *tD0 = myT + nb * PRIV_TILE_HALFS, *tD = tD0 + 32 * WD0, *tA = tD + 32 * WD;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const uint32_t rT = ks == 0 ? rT0 : rT1;
                    h8 av[2], bv[2];
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
                        if (mt < MTo) av[mt] = s == 0 ? priv_tr_pair(tD0 + rD0 + 16 * ks * WD0, 2 * WD0) : priv_tr_pair(tD + rT + 16 * ks * WD + 32 * mt, 2 * WD);
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        if (nt < NTi) bv[nt] = priv_tr_pair(tA + rT + 16 * ks * WD + 32 * nt, 2 * WD);
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
#pragma unroll
                        for (int nt = 0; nt < 2; nt++)
                            if (mt < MTo && nt < NTi) priv_mfma_acc(dwacc[tbase + mt * NTi + nt], av[mt], bv[nt]);
                }
            }
            // ---- next delta (pre-activation), chained in registers
            if (s == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const h8 a = ld_frag(ldsB, mt, lane);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, g_cur[nb], FZ);
                }
            } else if (s < NL) {
                const uint32_t fbase = MT + (NL - 1 - s) * MT * KC;           // hidden matrix fl - 1 with fl = NL - s
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const h8 a = ld_frag(ldsB, fbase + mt * KC + kc, lane);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, bf[kc][nb], kc == 0 ? FZ : acc[mt][nb]);
                    }
            }
        }
    }

    // ---- the four waves' tiles meet in LDS (the weight images and tiles are done with), one slot per workgroup leaves:
    // [stage][tile][register][lane] fp32, the layout k_mlp_dw_reduce reads (ffmlp.hip), every stage's tiles in (mt, nt) order
#pragma unroll
    for (int t = 0; t < NT; t++) priv_acc_settle(dwacc[t]);      // the last weight-gradient MFMAs have left the matrix pipe before anything reads a tile
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);              // [4 waves][4 tiles][16][64]
    float *slot = ws + (uint64_t)blockIdx.x * ((NL + 1) * MLP_DW_SLOT_STAGE);
#pragma unroll
    for (int s = 0; s <= NL; s++) {
        const int nt_s = (s == 0 || s == NL) ? 2 : 4, tbase = s == 0 ? 0 : 2 + 4 * (s - 1);
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (t < nt_s) {
#pragma unroll
                for (int reg = 0; reg < 16; reg++) red[((wave * 4 + t) * 16 + reg) * 64 + lane] = dwacc[tbase + t][reg];
            }
        __syncthreads();
        if ((int)wave < nt_s) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                float v = red[((0 * 4 + wave) * 16 + reg) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; w++) v += red[((w * 4 + wave) * 16 + reg) * 64 + lane];
                slot[(s * 4 + wave) * 1024 + reg * 64 + lane] = v;
            }
        }
        __syncthreads();
    }
}

template <int NL, int IMODE, int NB>
static size_t priv_lds_bytes() {
    const size_t weights = (size_t)((2 + (NL - 1) * 8 + 4) + (4 + (NL - 1) * 8)) * 1024 + 256;
    const size_t tiles = (size_t)MLP_WAVES * NB * PRIV_TILE_HALFS * sizeof(_Float16);
    const size_t red = (size_t)4 * 4 * 16 * 64 * sizeof(float);
    return weights + tiles > red ? weights + tiles : red;
}

template <int NL, int IMODE, int NB>
static int priv_launch(const void *grad, const void *inputs, const void *weights, uint32_t B, void *grad_inputs, float *slots, uint32_t max_grid,
                       const MlpHead &hd, hipStream_t st, uint32_t *grid_out) {
    auto kern = k_mlp_bwd_priv<NL, IMODE, NB>;
    const size_t lds = priv_lds_bytes<NL, IMODE, NB>();
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    uint32_t grid = foc_div_up(foc_div_up(B, 32 * NB), MLP_WAVES);
    if (grid > max_grid) grid = max_grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_BLOCK), lds, st, (const _Float16 *)grad, (const _Float16 *)inputs, (const _Float16 *)weights,
                       (_Float16 *)grad_inputs, slots, B, hd);
    FOC_CHECK_LAUNCH("ffmlp_backward(private tiles)");
    *grid_out = grid;
    return FOC_OK;
}

// Launches the wave-private backward for (hidden 64, input 32, NL hidden layers, ReLU, re-evaluated activations); `imode` as the template
// parameter; nb = rows per wave and step / 32 (1 or 2). `slots` receives one slot per workgroup; *grid_out = their number (<= max_grid).
int mlp_bwd_priv_launch(int num_layers, int imode, int nb, const void *grad, const void *inputs, const void *weights, uint32_t B, void *grad_inputs,
                        float *slots, uint32_t max_grid, const MlpHead *head, hipStream_t st, uint32_t *grid_out) {
    const MlpHead hd = head ? *head : MlpHead{nullptr, nullptr, 1u, 16u, nullptr};
#define PRIV_CASE(NLv, IMv)                                                                                                                  \
    if (num_layers == NLv && imode == IMv)                                                                                                   \
        return nb == 2 ? priv_launch<NLv, IMv, 2>(grad, inputs, weights, B, grad_inputs, slots, max_grid, hd, st, grid_out)                  \
                       : priv_launch<NLv, IMv, 1>(grad, inputs, weights, B, grad_inputs, slots, max_grid, hd, st, grid_out);
    PRIV_CASE(2, 0) PRIV_CASE(2, 1) PRIV_CASE(2, 2) PRIV_CASE(2, 3)
    PRIV_CASE(3, 0) PRIV_CASE(3, 1) PRIV_CASE(3, 2) PRIV_CASE(3, 3)
#undef PRIV_CASE
    foc_set_error("ffmlp_backward(private tiles): shape not built (layers %d, input mode %d)", num_layers, imode);
    return FOC_E_INVALID;
}
