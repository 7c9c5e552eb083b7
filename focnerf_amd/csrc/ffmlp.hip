// ffmlp.hip — fully fused fp16 MLP on gfx950 matrix cores (v_mfma_f32_32x32x16_f16).
//
// Semantics: ffmlp/src/ffmlp.cu of the reference — kernel_mlp_fused (:331-407),
// kernel_mlp_fused_backward (:410-518), weight-gradient GEMMs (:800-876); layouts :631-634,
// :742-748; ReLU forward/backward ffmlp/src/utils.h:427-432, :540-545.
//
// CDNA4 design (not a translation of the WMMA/shared-memory structure):
//  * The network is evaluated TRANSPOSED: Y^T[neurons x batch] = W[neurons x in] * X^T[in x batch].
//    With the 32x32x16 MFMA the result tile then has the batch sample on the LANE and the
//    neuron index in the 16 accumulator REGISTERS, which is exactly the B-operand shape of the
//    next layer's MFMA (sum over neurons). Activations therefore chain from layer to layer in
//    registers — ReLU + v_cvt to fp16 on the accumulators, no LDS round trip, no barrier.
//    The k order of such a chained operand is permuted (element j of lane-half h is neuron
//    16*kc + 8*(j>>2) + 4*h + (j&3)); the permutation is folded into the WEIGHT fragments
//    when they are staged into LDS once per workgroup, so it costs nothing per sample.
//  * Every wave owns its batch tiles end to end (64 samples per iteration = 2 MFMA column
//    tiles sharing each weight fragment); the 4 waves of a workgroup share only the staged
//    weight image in LDS (read with conflict-free ds_read_b128, 1 KiB per fragment).
//  * fp32 accumulation (the reference accumulates in fp16, ffmlp.cu:68,169,256,458); layer
//    outputs are rounded to fp16 exactly once, like the reference's half fragments.
//  * Backward: activation gradients chain the same way with transposed weight fragments;
//    weight gradients are one split-K MFMA kernel over the batch (all layers in one launch,
//    fp32 atomics of whole 128-B row segments into a 16-KiB-per-layer workspace), replacing
//    the reference's CUTLASS split-K GEMMs on side streams (ffmlp.cu:800-876).
#include "common.h"
#include "activations.h"
#include <stdlib.h>

#include <type_traits>
#include "mlp_common.h"

// ---------------------------------------------------------------- M1: fused forward / inference
// GEN: a hidden activation other than ReLU / None (exponential, sine, sigmoid, squareplus, softplus — ffmlp/src/utils.h:424-470; no FOC
// network uses one): the same MFMA chain, the activation evaluated in fp32 on the half-rounded sums. `act` is the reference's code; the
// ReLU / None instantiations read it as a flag (act == 0) and keep their packed-max path.
template <int HIDDEN, int NB, bool TRAIN, int IMODE, bool GEN = false>
__global__ void __launch_bounds__(MLP_BLOCK, 2) k_mlp_fwd(const _Float16 *__restrict__ inputs, const _Float16 *__restrict__ weights,
                                                       _Float16 *__restrict__ fwd_buf, _Float16 *__restrict__ outputs,
                                                       uint32_t B, uint32_t in_dim, uint32_t num_layers, int act, MlpHead hd) {
    const int relu = act == FOC_ACT_RELU;
    constexpr int MT = (HIDDEN + 31) / 32, KC = HIDDEN / 16;
    constexpr bool planar = IMODE == 1;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const uint32_t KS0 = IMODE == 2 ? 2u : in_dim / 16;
    const uint32_t f_hidden = MT * KS0, f_out = f_hidden + (num_layers - 1) * MT * KC;
    const float *obj_bias = reinterpret_cast<const float *>(lds + (size_t)(f_out + KC) * 512);      // head mode with an object feature only
    if constexpr (IMODE == 2) {
        stage_weights_fwd<HIDDEN>(weights, lds, in_dim, num_layers, true, head_ld0(hd));
        if (hd.obj) stage_obj_bias(weights, hd.obj, const_cast<float *>(obj_bias), HIDDEN);
    } else stage_weights_fwd<HIDDEN>(weights, lds, in_dim, num_layers);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t tile_rows = 32 * NB;
    const uint32_t n_tiles = (B + tile_rows - 1) / tile_rows;

    for (uint32_t tile = blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += gridDim.x * MLP_WAVES) {
        const uint64_t row0 = (uint64_t)tile * tile_rows;
        f16v acc[MT][NB];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[mt][nb][e] = 0.0f;

        // ---- layer 0: B operand straight from global inputs (natural k order)
        if constexpr (IMODE == 2) {
            if (hd.obj) {                                  // the object feature's share of layer 0: a constant per neuron (MlpHead)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f16v b0 = ld_obj_bias(obj_bias, mt, h);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) acc[mt][nb] = b0;
                }
            }
            // head mode (two k-chunks: the ray's SH row, the shifted h row): both requested before the first MFMA and pinned — left alone the
            // compiler sinks the second chunk's loads below the MFMAs of the first (54 -> 52 us per 2 M rows). The same hoist made the
            // planar form SLOWER (53 -> 62 us: sixteen dword loads waited for at once instead of overlapping the first chunk's MFMAs).
            h8 b[2][NB];
#pragma unroll
            for (int kc = 0; kc < 2; kc++)
#pragma unroll
                for (int nb = 0; nb < NB; nb++) b[kc][nb] = ld_head8(inputs, hd, min(row0 + nb * 32 + c, (uint64_t)B - 1), kc, h);
#pragma unroll
            for (int kc = 0; kc < 2; kc++)
#pragma unroll
                for (int nb = 0; nb < NB; nb++) asm volatile("" : "+v"(b[kc][nb]));
#pragma unroll
            for (int kc = 0; kc < 2; kc++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const h8 a = ld_frag(lds, mt * KS0 + kc, lane);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, b[kc][nb], acc[mt][nb]);
                }
        } else
        for (uint32_t kc = 0; kc < KS0; kc++) {
            h8 b[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const uint64_t row = min(row0 + nb * 32 + c, (uint64_t)B - 1);
                b[nb] = planar ? ld_planar8(inputs, B, row, kc, h) : *reinterpret_cast<const h8 *>(inputs + row * in_dim + 16 * kc + 8 * h);
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const h8 a = ld_frag(lds, mt * KS0 + kc, lane);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, b[nb], acc[mt][nb]);
            }
        }
        // ---- hidden layers, chained in registers
        for (uint32_t l = 1; l <= num_layers; l++) {
            // acc holds the pre-activation of layer l-1
            if (TRAIN) {
                _Float16 *fb = fwd_buf + (uint64_t)(l - 1) * B * HIDDEN;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        if constexpr (GEN) store_tile_act(fb, HIDDEN, row0 + nb * 32 + c, B, 32 * mt, HIDDEN, acc[mt][nb], h, act);
                        else if (relu) store_tile<true>(fb, HIDDEN, row0 + nb * 32 + c, B, 32 * mt, HIDDEN, acc[mt][nb], h);
                        else store_tile<false>(fb, HIDDEN, row0 + nb * 32 + c, B, 32 * mt, HIDDEN, acc[mt][nb], h);
                    }
            }
            h8 bf[KC][NB];
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    if constexpr (GEN) bf[kc][nb] = acc_to_frag_act(acc[kc >> 1][nb], kc & 1, act);
                    else bf[kc][nb] = relu ? acc_to_frag<true>(acc[kc >> 1][nb], kc & 1) : acc_to_frag<false>(acc[kc >> 1][nb], kc & 1);
                }
            if (l < num_layers) {
                const uint32_t fbase = f_hidden + (l - 1) * MT * KC;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[mt][nb][e] = 0.0f;
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const h8 a = ld_frag(lds, fbase + mt * KC + kc, lane);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, bf[kc][nb], acc[mt][nb]);
                    }
            } else {
                // ---- output layer (16 neurons in the low half of one M tile), no activation
                f16v o[NB];
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
#pragma unroll
                    for (int e = 0; e < 16; e++) o[nb][e] = 0.0f;
#pragma unroll
                for (int kc = 0; kc < KC; kc++) {
                    const h8 a = ld_frag(lds, f_out + kc, lane);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) o[nb] = mfma16(a, bf[kc][nb], o[nb]);
                }
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    if (IMODE == 2 && hd.out_width == 4u) {
                        // of the colour network's 16 padded outputs only the rgb logits are ever read: lanes of half 0 hold neurons 0..3
                        const uint64_t row = row0 + nb * 32 + c;
                        if (h == 0 && row < B) {
                            const h4 v = {(_Float16)o[nb][0], (_Float16)o[nb][1], (_Float16)o[nb][2], (_Float16)o[nb][3]};
                            *reinterpret_cast<h4 *>(outputs + row * 4) = v;
                        }
                    } else store_tile<false>(outputs, 16, row0 + nb * 32 + c, B, 0, 16, o[nb], h);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- M2: fused activation-gradient backward
// backward_buffer[k] = gradient w.r.t. the (post-ReLU) output of forward layer num_layers-1-k.
// GEN: any hidden activation (warp_activation_backward, utils.h:533-589): the half-rounded delta times a factor of the stored
// post-activation, in half arithmetic (activations.h); otherwise `act` is read as the ReLU flag.
template <int HIDDEN, int NB, bool GEN = false>
__global__ void __launch_bounds__(MLP_BLOCK) k_mlp_bwd(const _Float16 *__restrict__ grad, const _Float16 *__restrict__ weights,
                                                       const _Float16 *__restrict__ fwd_buf, _Float16 *__restrict__ bwd_buf,
                                                       _Float16 *__restrict__ grad_inputs, uint32_t B, uint32_t in_dim,
                                                       uint32_t num_layers, int act) {
    const int relu = act == FOC_ACT_RELU;
    constexpr int MT = (HIDDEN + 31) / 32, KC = HIDDEN / 16;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const bool with_dx = grad_inputs != nullptr;
    stage_weights_bwd<HIDDEN>(weights, lds, in_dim, num_layers, with_dx);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t MT0 = (in_dim + 31) / 32;
    const uint32_t f_hidden = MT, f_dx = MT + (num_layers - 1) * MT * KC;
    const uint32_t tile_rows = 32 * NB;
    const uint32_t n_tiles = (B + tile_rows - 1) / tile_rows;

    for (uint32_t tile = blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += gridDim.x * MLP_WAVES) {
        const uint64_t row0 = (uint64_t)tile * tile_rows;
        f16v acc[MT][NB];
        // ---- through the output layer: delta = W_out^T * grad^T   (K = 16, one k-step)
        {
            h8 bg[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) bg[nb] = *reinterpret_cast<const h8 *>(grad + min(row0 + nb * 32 + c, (uint64_t)B - 1) * 16 + 8 * h);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const h8 a = ld_frag(lds, mt, lane);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    f16v z;
#pragma unroll
                    for (int e = 0; e < 16; e++) z[e] = 0.0f;
                    acc[mt][nb] = mfma16(a, bg[nb], z);
                }
            }
        }
        for (uint32_t k = 0; k < num_layers; k++) {
            const uint32_t fl = num_layers - 1 - k;   // forward layer whose output gradient `acc` is
            // ---- ReLU transfer with the stored forward activations (utils.h:540-545), then store
            if constexpr (GEN) {
                const _Float16 *fb = fwd_buf + (uint64_t)fl * B * HIDDEN;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t col = 32 * mt + 8 * q + 4 * h;
                            if (col < HIDDEN) {
                                const h4 f = *reinterpret_cast<const h4 *>(fb + min(row0 + nb * 32 + c, (uint64_t)B - 1) * HIDDEN + col);
#pragma unroll
                                for (int e = 0; e < 4; e++) acc[mt][nb][4 * q + e] = (float)foc_act_backward((_Float16)acc[mt][nb][4 * q + e], f[e], act);
                            }
                        }
            } else if (relu) {
                const _Float16 *fb = fwd_buf + (uint64_t)fl * B * HIDDEN;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t col = 32 * mt + 8 * q + 4 * h;
                            if (col < HIDDEN) {
                                const h4 f = *reinterpret_cast<const h4 *>(fb + min(row0 + nb * 32 + c, (uint64_t)B - 1) * HIDDEN + col);
#pragma unroll
                                for (int e = 0; e < 4; e++) if (!(f[e] > (_Float16)0)) acc[mt][nb][4 * q + e] = 0.0f;
                            }
                        }
            }
            {
                _Float16 *bb = bwd_buf + (uint64_t)k * B * HIDDEN;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) store_tile<false>(bb, HIDDEN, row0 + nb * 32 + c, B, 32 * mt, HIDDEN, acc[mt][nb], h);
            }
            if (fl == 0 && !with_dx) break;
            h8 bf[KC][NB];
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
#pragma unroll
                for (int nb = 0; nb < NB; nb++) bf[kc][nb] = acc_to_frag<false>(acc[kc >> 1][nb], kc & 1);
            if (fl > 0) {
                // through hidden matrix fl-1 (maps fwd[fl-1] -> fwd[fl])
                const uint32_t fbase = f_hidden + (fl - 1) * MT * KC;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[mt][nb][e] = 0.0f;
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const h8 a = ld_frag(lds, fbase + mt * KC + kc, lane);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, bf[kc][nb], acc[mt][nb]);
                    }
            } else {
                // ---- dL/dinput = delta_0 * W_0  (no activation)
                for (uint32_t mt0 = 0; mt0 < MT0; mt0++) {
                    f16v x[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int e = 0; e < 16; e++) x[nb][e] = 0.0f;
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        const h8 a = ld_frag(lds, f_dx + mt0 * KC + kc, lane);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) x[nb] = mfma16(a, bf[kc][nb], x[nb]);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) store_tile<false>(grad_inputs, in_dim, row0 + nb * 32 + c, B, 32 * mt0, in_dim, x[nb], h);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- M3: weight gradients, split-K over the batch
// Layer id j: 0 = input layer (dW0 = delta_0^T x), 1..num_layers-1 = hidden matrix j-1, num_layers = output layer.
//   dW[o][i] = sum_b D[b][o] * A[b][i]   with D = delta (or grad for the output layer), A = layer input.
// A workgroup streams 64-sample chunks of D and A through LDS (coalesced 16-byte loads), each wave owns
// 32x32 output tiles and reads its MFMA operands transposed out of LDS; partial sums go to the fp32
// workspace with float atomics shaped as whole 128-byte row segments.
#define DW_CHUNK 64
// AW: widest layer input the LDS tile holds — 128 (hidden <= 128, inputs up to 128), 256 (hidden 256, or the reference's dynamic input layer with
// up to 256 inputs at a narrower hidden width, ffmlp.cu:151-239: any 16 m that fits shared memory)
template <int HIDDEN, int AW = (HIDDEN > 128 ? HIDDEN : 128)>
__global__ void __launch_bounds__(MLP_BLOCK) k_mlp_dw(const _Float16 *__restrict__ grad, const _Float16 *__restrict__ inputs,
                                                      const _Float16 *__restrict__ fwd_buf, const _Float16 *__restrict__ bwd_buf,
                                                      float *__restrict__ ws, uint32_t B, uint32_t in_dim, uint32_t num_layers) {
    constexpr int LDP = 8;   // row padding (halfs) to spread the strided 2-byte reads over banks
    __shared__ __attribute__((aligned(16))) _Float16 sD[DW_CHUNK][(HIDDEN < 32 ? 32 : HIDDEN) + LDP];   // >= 32 columns: transposed reads span a whole 32-row tile
    __shared__ __attribute__((aligned(16))) _Float16 sA[DW_CHUNK][AW + LDP];

    const uint32_t j = blockIdx.y;
    const _Float16 *Dp; const _Float16 *Ap; uint32_t OUT, IN; uint64_t ws_off;
    const uint64_t first = (uint64_t)HIDDEN * in_dim, lsz = (uint64_t)HIDDEN * HIDDEN;
    if (j == 0) { Dp = bwd_buf + (uint64_t)(num_layers - 1) * B * HIDDEN; Ap = inputs; OUT = HIDDEN; IN = in_dim; ws_off = 0; }
    else if (j < num_layers) { Dp = bwd_buf + (uint64_t)(num_layers - 1 - j) * B * HIDDEN; Ap = fwd_buf + (uint64_t)(j - 1) * B * HIDDEN; OUT = HIDDEN; IN = HIDDEN; ws_off = first + (j - 1) * lsz; }
    else { Dp = grad; Ap = fwd_buf + (uint64_t)(num_layers - 1) * B * HIDDEN; OUT = 16; IN = HIDDEN; ws_off = first + (uint64_t)(num_layers - 1) * lsz; }

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const uint32_t MTo = (OUT + 31) / 32, NTi = (IN + 31) / 32, n_out_tiles = MTo * NTi;
    // up to 4 tiles per wave and workgroup: 16 tiles cover OUT, IN <= 128; hidden 256 (64 tiles) and wide input layers (up to 32) deal them
    // over blockIdx.z — a z slice with no tile of this layer has nothing to stage
    if (16 * blockIdx.z >= n_out_tiles) return;
    f16v acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.0f;

    const uint32_t n_chunks = (B + DW_CHUNK - 1) / DW_CHUNK;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t row0 = (uint64_t)chunk * DW_CHUNK;
        __syncthreads();
        // stage D [64 x OUT] and A [64 x IN], 8 halfs per thread-iteration
        for (uint32_t idx = threadIdx.x; idx < DW_CHUNK * (OUT / 8); idx += MLP_BLOCK) {
            const uint32_t rr = idx / (OUT / 8), cc = (idx % (OUT / 8)) * 8;
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};                       // rows past B contribute nothing
            if (row0 + rr < B) v = *reinterpret_cast<const h8 *>(Dp + (row0 + rr) * OUT + cc);
            *reinterpret_cast<h8 *>(&sD[rr][cc]) = v;
        }
        for (uint32_t idx = threadIdx.x; idx < DW_CHUNK * (IN / 8); idx += MLP_BLOCK) {
            const uint32_t rr = idx / (IN / 8), cc = (idx % (IN / 8)) * 8;
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row0 + rr < B) v = *reinterpret_cast<const h8 *>(Ap + (row0 + rr) * IN + cc);
            *reinterpret_cast<h8 *>(&sA[rr][cc]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t tile = wave + t * MLP_WAVES + 16 * blockIdx.z;
            if (tile < n_out_tiles) {
                const uint32_t mt = tile / NTi, nt = tile % NTi;
                const uint32_t o = 32 * mt + r, i = 32 * nt + r;
                // Both operands need 8 consecutive BATCH rows of one neuron column: a transposed read of the
                // row-major LDS tiles. ds_read_b64_tr_b16 hands lane i of each 16-lane group column i of a
                // 4-row x 16-column block (lane 4q+p supplies the address of row q, columns 4p..4p+3), so one
                // fragment is two such reads. All 64 lanes take part (the tile test above is wave-uniform);
                // lanes whose neuron is past OUT / IN read in-bounds garbage and are zeroed afterwards.
                const int q = (lane & 15) >> 2, p = lane & 3, cg = 16 * ((lane >> 4) & 1);
#pragma unroll
                for (int ks = 0; ks < DW_CHUNK / 16; ks++) {
                    const int k0 = 16 * ks + 4 * q + h;      // batch rows {0,4,8,12} + h (+2 for the second read): see k_mlp_bwd_fused
                    const s4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)&sD[k0][32 * mt + cg + 4 * p]);
                    const s4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)&sD[k0 + 2][32 * mt + cg + 4 * p]);
                    const s4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)&sA[k0][32 * nt + cg + 4 * p]);
                    const s4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)&sA[k0 + 2][32 * nt + cg + 4 * p]);
                    // assemble the fragments at dword granularity (hipcc 7.2 mis-folds per-element extracts of the
                    // builtin's v4i16 result: it reuses element 0 for every lane element)
                    const u32x2 A0 = __builtin_bit_cast(u32x2, a0), A1 = __builtin_bit_cast(u32x2, a1);
                    const u32x2 B0 = __builtin_bit_cast(u32x2, b0), B1 = __builtin_bit_cast(u32x2, b1);
                    u32x4 av = {A0.x, A0.y, A1.x, A1.y}, bv = {B0.x, B0.y, B1.x, B1.y};
                    if (!(o < OUT)) av = u32x4{0u, 0u, 0u, 0u};
                    if (!(i < IN)) bv = u32x4{0u, 0u, 0u, 0u};
                    acc[t] = mfma16(__builtin_bit_cast(h8, av), __builtin_bit_cast(h8, bv), acc[t]);
                }
            }
        }
    }
    // accumulate partial tiles: lane holds column i = 32*nt + r, rows o = 32*mt + acc_row(reg, h)
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t tile = wave + t * MLP_WAVES + 16 * blockIdx.z;
        if (tile < n_out_tiles) {
            const uint32_t mt = tile / NTi, nt = tile % NTi;
            const uint32_t i = 32 * nt + r;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const uint32_t o = 32 * mt + acc_row(reg, h);
                if (o < OUT && i < IN) (void)__hip_atomic_fetch_add(ws + ws_off + (uint64_t)o * IN + i, acc[t][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// fp32 workspace -> fp16 grad_weights (one rounding, like a full-K fp32 accumulation)
__global__ void __launch_bounds__(256) k_mlp_dw_finalize(const float *__restrict__ ws, _Float16 *__restrict__ gw, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) gw[i] = (_Float16)ws[i];
}

// ---------------------------------------------------------------- M2+M3 fused: activation gradients AND weight gradients in one pass
// The two-kernel form writes every activation gradient to HBM (backward_buffer) and reads it, the forward activations and the
// inputs back for the split-K weight-gradient GEMM: ~1.2 KB/sample. Here a workgroup (4 waves x 32*NB rows) walks the layers once:
//   stage s: every wave puts its rows of D_s (grad for s = 0, else the masked delta it just computed) and of A_s (the layer's
//            input: forward activations, or the network inputs for the last stage) into LDS as row-major tiles;
//   barrier; wave w accumulates output tile w of dW_s = D_s^T A_s over ALL four waves' rows (operands read transposed with
//            ds_read_b64_tr_b16) in registers that live across the whole batch loop;
//   the same A_s tile is the ReLU mask of the next delta, which chains in registers exactly as in k_mlp_bwd; barrier.
// HBM traffic drops to grad + forward activations + inputs (+ grad_inputs): ~0.4 KB/sample, and backward_buffer is only written
// if the caller asks for it. Compile-time layer count (the stage loop must unroll for the accumulators to stay in registers).
//
// RECOMP (forward_buffer == NULL at the ABI): the forward pass stored no activations; each wave re-evaluates the hidden layers of
// its 32 rows from the inputs first — the same MFMA sequence as k_mlp_fwd, so the activations are the bits the forward pass saw —
// and keeps them in registers in the chained layout: they are the ReLU masks as they are, and are written into the LDS A tiles
// where the stored form loads them from HBM. Traffic: grad + inputs (+ grad_inputs), 0.1-0.16 KB/sample. The next group's grad
// and input rows are fetched while the current group is processed.
// LEAN (round 5): no backward_buffer and input gradients wanted, both as compile-time facts — what every training step asks for. With the runtime
// pointers each stage carried a cold stored-form block, guarded stores and a runtime dX loop: a dozen branches per group, and nothing is scheduled
// across a branch (k_field_fwd_train lost a factor 1.2 to exactly that, NOTEBOOK 5.4).
template <int HIDDEN, int NL, int NB, bool RECOMP, int IMODE, bool RELU_CT, bool IN32 = false, bool LEAN = false>
__global__ void __launch_bounds__(MLP_BLOCK, 2) k_mlp_bwd_fused(const _Float16 *__restrict__ grad, const _Float16 *__restrict__ inputs,
                                                             const _Float16 *__restrict__ weights, const _Float16 *__restrict__ fwd_buf,
                                                             _Float16 *__restrict__ bwd_buf_arg, _Float16 *__restrict__ grad_inputs, float *__restrict__ ws,
                                                             uint32_t B, uint32_t in_dim, int relu_rt, uint32_t lds_w_halfs, MlpHead hd) {
    // RELU_CT: the activation as a compile-time fact (every NeRF network is ReLU): with the runtime flag the compiler evaluated BOTH forms of
    // every activation / gate and selected per register (224 v_cndmask per 128 rows); `false` keeps the runtime flag (activation 'none')
    const int relu = RELU_CT ? 1 : relu_rt;
    f16v FZ;                                               // constant zero C operand: folds into the first MFMA of a chain as the inline constant 0
#pragma unroll
    for (int e = 0; e < 16; e++) FZ[e] = 0.0f;
    constexpr int MT = (HIDDEN + 31) / 32, KC = HIDDEN / 16, RW = 32 * NB;
    // IMODE: 0 = [B, in_dim] rows, 1 = planar, 2 = colour-network head (MlpHead) with [B,16] output gradients, 3 = the head with [B,4]
    // output gradients (hd.out_width == 4) as a compile-time fact: as a runtime branch around the prefetch loads it cost a wait per group
    constexpr bool planar = IMODE == 1, HEAD = IMODE >= 2, NARROW = IMODE == 3;
    static_assert(!HEAD || RECOMP, "the head input modes have no stored-activation form");
    constexpr int WD = (HIDDEN < 32 ? 32 : HIDDEN) + 8;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const bool with_dx = LEAN ? true : grad_inputs != nullptr;
    _Float16 *const bwd_buf = LEAN ? nullptr : bwd_buf_arg;
    // row width of W0 in the blob and in the weight-gradient workspace: in_dim, or 48 for the colour head with an object feature (MlpHead)
    uint32_t ld0 = in_dim;
    bool has_obj = false;
    if constexpr (HEAD) { ld0 = head_ld0(hd); has_obj = hd.obj != nullptr; }
    stage_weights_bwd<HIDDEN>(weights, lds, in_dim, NL, with_dx, HEAD, ld0);
    const uint32_t WA = (in_dim > (uint32_t)HIDDEN ? in_dim : (uint32_t)(HIDDEN < 32 ? 32 : HIDDEN)) + 8;
    _Float16 *sD = lds + lds_w_halfs;                    // [4][RW][WD]
    _Float16 *sA = sD + 4 * RW * WD;                     // [4][RW][WA]
    _Float16 *ldsF = sA + 4 * RW * WA;                   // RECOMP: forward weight image (layer 0 + hidden matrices)
    const float *obj_bias = reinterpret_cast<const float *>(ldsF + (size_t)(((HIDDEN + 31) / 32) * (in_dim / 16) + (NL - 1) * ((HIDDEN + 31) / 32) * (HIDDEN / 16)) * 512);
    if constexpr (RECOMP) {
        stage_weights_fwd<HIDDEN>(weights, ldsF, in_dim, NL, false, ld0);
        if constexpr (HEAD) { if (has_obj) stage_obj_bias(weights, hd.obj, const_cast<float *>(obj_bias), HIDDEN); }
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t MT0 = (HEAD || IN32) ? 1u : (in_dim + 31) / 32;
    const uint32_t f_hidden = MT, f_dx = MT + (NL - 1) * MT * KC;
    _Float16 *myD = sD + wave * RW * WD, *myA = sA + wave * RW * WA;

    f16v dwacc[NL + 1];
#pragma unroll
    for (int s = 0; s <= NL; s++)
#pragma unroll
        for (int e = 0; e < 16; e++) dwacc[s][e] = 0.0f;

    const uint32_t rows_per_group = 4 * RW;
    const uint32_t n_groups = (B + rows_per_group - 1) / rows_per_group;
    // in_dim <= 64 on this path; the head's input is 32 wide by construction. IN32: in_dim <= 32 as a compile-time fact (the sigma network of
    // every NeRF field): with the runtime width alone the prefetch held FOUR k-chunks per row and loaded the last existing chunk again for the
    // two that do not exist — 8 of the planar form's 16 dword loads per lane and group, plus their registers and rotation moves (round 5)
    constexpr int KS0M = (HEAD || IN32) ? 2 : 4;
    const uint32_t KS0 = HEAD ? 2u : in_dim / 16;
    // RECOMP: this wave's grad rows (D_0 tile order) and input rows (layer-0 B operand order) of the group about to be processed
    h8 g_nxt[(RW * 2 + 63) / 64];
    h8 x_nxt[KS0M][NB];
    _Float16 h0_nxt[NB];                                 // input mode 2: the density path's gradient of h[:,0] for this lane's rows
    uint32_t hx_nxt[NB];                                 // input mode 2: the dword after the lane's 16 bytes of its h row
    // Every prefetch load is UNCONDITIONAL (rows clamped to B - 1, k-chunks to the last existing one, a readable dummy address where a
    // pointer is null); what must read as zero is zeroed when the registers become "current". With `cond ? load : 0` each load sat in an
    // exec-masked or uniform branch of its own, the compiler could not count the loads in flight and placed s_waitcnt vmcnt(0) between the
    // prefetch loads themselves — a full memory latency per group of 128 rows, in a kernel with two waves per SIMD to cover it.
    static_assert((RW * 2) % 64 == 0, "a wave's grad rows are whole 64-lane passes");
    auto fetch_group = [&](uint32_t grp) {
        const uint64_t r0 = (uint64_t)grp * rows_per_group + wave * RW;
        if constexpr (HEAD) {
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const uint64_t row = min(r0 + nb * 32 + c, (uint64_t)B - 1);
                const _Float16 *p = hd.grad_h0 ? hd.grad_h0 + row : grad;
                h0_nxt[nb] = *p;
            }
        }
#pragma unroll
        for (uint32_t it = 0; it < (RW * 2) / 64; it++) {
            const uint32_t idx = lane + 64 * it, rr = idx >> 1, cc = (idx & 1) * 8;
            const uint64_t row = min(r0 + rr, (uint64_t)B - 1);
            if constexpr (NARROW) {                          // [B,4] output gradients: columns 4..15 are zeros that were never written (odd lanes: dropped below)
                const uint2 q = *reinterpret_cast<const uint2 *>(grad + row * 4);
                g_nxt[it] = __builtin_bit_cast(h8, (u32x4){q.x, q.y, 0u, 0u});
            } else g_nxt[it] = *reinterpret_cast<const h8 *>(grad + row * 16 + cc);
        }
#pragma unroll
        for (int kc = 0; kc < KS0M; kc++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const uint32_t kce = min((uint32_t)kc, KS0 - 1u);           // chunks past in_dim repeat the last one; they are never used
                const uint64_t row = min(r0 + nb * 32 + c, (uint64_t)B - 1);
                h8 v;
                if constexpr (HEAD) {
                    if (kc == 0) v = ld_head8(inputs, hd, row, 0, h);
                    else { u32x4 raw; ld_head_raw_all(inputs, row, h, raw, hx_nxt[nb]); v = __builtin_bit_cast(h8, raw); }   // shifted when it becomes x_cur
                } else v = planar ? ld_planar8(inputs, B, row, kce, h) : *reinterpret_cast<const h8 *>(inputs + row * in_dim + 16 * kce + 8 * h);
                x_nxt[kc][nb] = v;
            }
    };
    if constexpr (RECOMP) {
        if (blockIdx.x < n_groups) fetch_group(blockIdx.x);
        // The first group's rows are waited for HERE, once: the compiler lets these loads land directly in the registers the loop reads
        // as "current" and then, merging that state over the back edge, waited in EVERY iteration for the oldest of the prefetch loads it
        // had just issued (s_waitcnt vmcnt(3) behind four loads: one exposed memory latency per group of 128 rows).
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), expcnt / lgkmcnt untouched
    }
    for (uint32_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint64_t row0 = (uint64_t)grp * rows_per_group + wave * RW;     // this wave's first row
        f16v acc[MT][NB];
        h8 bf[KC][NB];
        h8 fa[RECOMP ? NL : 1][KC][NB];                  // RECOMP: post-activation of forward layer l, chained layout
        h8 g_cur[(RW * 2 + 63) / 64];
        h8 x_cur[KS0M][NB];
        _Float16 h0_cur[NB];
        if constexpr (HEAD) {
#pragma unroll
            for (int nb = 0; nb < NB; nb++) h0_cur[nb] = (hd.grad_h0 && row0 + nb * 32 + c < B) ? h0_nxt[nb] : (_Float16)0;
        }
        if constexpr (RECOMP) {
#pragma unroll
            for (uint32_t it = 0; it < (RW * 2 + 63) / 64; it++) {
                const uint32_t idx = lane + 64 * it;
                const bool live = row0 + (idx >> 1) < B && !(NARROW && (idx & 1u)), live_hi = live && !NARROW;
                const u32x4 gv = __builtin_bit_cast(u32x4, g_nxt[it]);
                g_cur[it] = __builtin_bit_cast(h8, (u32x4){live ? gv.x : 0u, live ? gv.y : 0u, live_hi ? gv.z : 0u, live_hi ? gv.w : 0u});
            }
#pragma unroll
            for (int kc = 0; kc < KS0M; kc++)
#pragma unroll
                for (int nb = 0; nb < NB; nb++) x_cur[kc][nb] = x_nxt[kc][nb];
            if constexpr (HEAD) {
#pragma unroll
                for (int nb = 0; nb < NB; nb++) x_cur[1][nb] = head_shift(__builtin_bit_cast(u32x4, x_nxt[1][nb]), h == 0 ? hx_nxt[nb] : 0u);
            }
            // (LEAN: the last group fetches itself again instead of branching around the loads)
            if (LEAN) fetch_group(min(grp + gridDim.x, n_groups - 1u));
            else if (grp + gridDim.x < n_groups) fetch_group(grp + gridDim.x);
            // ---- forward re-evaluation: layer 0 from the inputs, hidden layers chained (k_mlp_fwd's order of operations)
            // The colour head's layer 0 starts from the object feature's share (a constant per neuron) when there is one, else — like every other
            // input mode — from the inline-constant zero of the chain's first MFMA. Two wave-uniform branches: as ONE code path the start value was a
            // select per accumulator register (32 v_mov per 128 rows of the head without an object feature, round 5).
            auto layer0 = [&](auto from_zero) {
#pragma unroll
                for (int kc = 0; kc < KS0M; kc++) {
                    if (kc == 0 || (uint32_t)kc < KS0) {         // in_dim >= 16: k-chunk 0 always exists
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) {
                            const h8 a = ld_frag(ldsF, mt * KS0 + kc, lane);
#pragma unroll
                            for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, x_cur[kc][nb], (kc == 0 && decltype(from_zero)::value) ? FZ : acc[mt][nb]);
                        }
                    }
                }
            };
            bool from_bias = false;
            if constexpr (HEAD) from_bias = has_obj;
            if (from_bias) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f16v b0 = ld_obj_bias(obj_bias, mt, h);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) acc[mt][nb] = b0;
                }
                layer0(std::false_type{});
            } else layer0(std::true_type{});
#pragma unroll
            for (int l = 0; l < NL; l++) {
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
                        fa[l][kc][nb] = relu ? acc_to_frag<true>(acc[kc >> 1][nb], kc & 1) : acc_to_frag<false>(acc[kc >> 1][nb], kc & 1);
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
        }
#pragma unroll
        for (int s = 0; s <= NL; s++) {
            const uint32_t OUT = s == 0 ? 16u : (uint32_t)HIDDEN;
            const uint32_t IN = s < NL ? (uint32_t)HIDDEN : in_dim;
            const _Float16 *Ap = s < NL ? fwd_buf + (uint64_t)(NL - 1 - s) * B * HIDDEN : inputs;
            // ---- D_s tile of this wave -> LDS
            if (s == 0) {
                if constexpr (RECOMP) {
#pragma unroll
                    for (uint32_t it = 0; it < (RW * 2 + 63) / 64; it++) {
                        const uint32_t idx = lane + 64 * it, rr = idx >> 1, cc = (idx & 1) * 8;
                        if (idx < RW * 2) *reinterpret_cast<h8 *>(myD + rr * WD + cc) = g_cur[it];
                    }
                } else {
                    for (uint32_t idx = lane; idx < RW * 2; idx += 64) {
                        const uint32_t rr = idx >> 1, cc = (idx & 1) * 8;
                        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (row0 + rr < B) v = *reinterpret_cast<const h8 *>(grad + (row0 + rr) * 16 + cc);
                        *reinterpret_cast<h8 *>(myD + rr * WD + cc) = v;
                    }
                }
            } else if constexpr (RECOMP) {
                // delta of forward layer NL-s: acc rounded to fp16 in the chained layout (the next stage's B operand), gated by that
                // layer's activation there, and the same dwords go to the LDS tile (quad q of tile mt = half (q&1) of fragment 2mt+(q>>1)).
                // Rows past B need no masking: their D_0 rows are zeros, so every later delta of such a row is an exact zero.
                if (bwd_buf && relu) {                   // stored form asked for as well (cold): the buffer holds the gated values
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
#pragma unroll
                        for (int nb = 0; nb < NB; nb++)
#pragma unroll
                            for (int q = 0; q < 4; q++)
#pragma unroll
                                for (int e = 0; e < 4; e++)
                                    if ((2 * mt + (q >> 1)) < KC && !(fa[NL - s][(2 * mt + (q >> 1)) < KC ? (2 * mt + (q >> 1)) : 0][nb][4 * (q & 1) + e] > (_Float16)0))
                                        acc[mt][nb][4 * q + e] = 0.0f;
                }
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        bf[kc][nb] = acc_to_frag<false>(acc[kc >> 1][nb], kc & 1);
                        if (relu) bf[kc][nb] = relu_gate(bf[kc][nb], fa[NL - s][kc][nb]);
                    }
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t col = 32 * mt + 8 * q + 4 * h;
                            if (col < HIDDEN) {
                                const h8 f = bf[(2 * mt + (q >> 1)) < KC ? (2 * mt + (q >> 1)) : 0][nb];
                                const h4 v = (q & 1) ? h4{f[4], f[5], f[6], f[7]} : h4{f[0], f[1], f[2], f[3]};
#ifndef FOC_TIMING_NO_TILES
                                *reinterpret_cast<h4 *>(myD + (nb * 32 + c) * WD + col) = v;
#endif
                            }
                        }
                        if (bwd_buf) store_tile<false>(bwd_buf + (uint64_t)(s - 1) * B * HIDDEN, HIDDEN, row0 + nb * 32 + c, B, 32 * mt, HIDDEN, acc[mt][nb], h);
                    }
            } else {
                // delta of forward layer NL-s: acc, already masked; rounded to fp16 (what the reference stores in backward_buffer)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        // rows past B need no masking: their D_0 rows are zeros, so every later delta of such a row is an exact zero
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t col = 32 * mt + 8 * q + 4 * h;
                            if (col < HIDDEN) {
                                h4 v;
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] = (_Float16)acc[mt][nb][4 * q + e];
                                *reinterpret_cast<h4 *>(myD + (nb * 32 + c) * WD + col) = v;
                            }
                        }
                        if (bwd_buf) store_tile<false>(bwd_buf + (uint64_t)(s - 1) * B * HIDDEN, HIDDEN, row0 + nb * 32 + c, B, 32 * mt, HIDDEN, acc[mt][nb], h);
                    }
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) bf[kc][nb] = acc_to_frag<false>(acc[kc >> 1][nb], kc & 1);
            }
            // ---- A_s tile of this wave -> LDS
            if constexpr (RECOMP) {
                // from registers; rows past B hold values computed from a clamped (finite) input row and meet all-zero rows of D_s
                if (s < NL) {
#pragma unroll
                    for (int kc = 0; kc < KC; kc++)
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) {
                            const h8 v = fa[NL - 1 - s][kc][nb];
                            _Float16 *dst = myA + (nb * 32 + c) * WA + 16 * kc + 4 * h;
#ifndef FOC_TIMING_NO_TILES
                            *reinterpret_cast<h4 *>(dst) = h4{v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<h4 *>(dst + 8) = h4{v[4], v[5], v[6], v[7]};
#endif
                        }
                } else {
#pragma unroll
                    for (int kc = 0; kc < KS0M; kc++)
#pragma unroll
                        for (int nb = 0; nb < NB; nb++)
                            if ((uint32_t)kc < KS0) {
                                h8 v = x_cur[kc][nb];
                                // object feature: column 31 of the input tile (a zero of the shifted h row) becomes 1, so that dW0[:, 31] = sum_b delta_0
                                if constexpr (HEAD) { if (kc == 1 && has_obj && h == 1) v[7] = (_Float16)1.0f; }
                                *reinterpret_cast<h8 *>(myA + (nb * 32 + c) * WA + 16 * kc + 8 * h) = v;
                            }
                }
            } else {
                // rows past B as zeros: they must not reach dW
                for (uint32_t idx = lane; idx < RW * (IN / 8); idx += 64) {
                    const uint32_t rr = idx / (IN / 8), cc = (idx % (IN / 8)) * 8;
                    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (row0 + rr < B) v = *reinterpret_cast<const h8 *>(Ap + (row0 + rr) * IN + cc);
                    *reinterpret_cast<h8 *>(myA + rr * WA + cc) = v;
                }
            }
#ifndef FOC_TIMING_NO_BARRIER
            foc_lds_barrier();     // LDS tiles only: the next group's prefetched rows stay in flight, grad_inputs stores are not waited for
#endif
            // ---- input gradients (last stage) BEFORE this stage's weight-gradient MFMAs: they need the delta fragments in registers only, and the
            // stores then have the whole dW section to be acknowledged — the wait for the prefetched rows at the top of the next group is a
            // vmcnt(0) (stores behind per-lane guards cannot be counted), which used to sit right behind these stores
            if (s == NL && with_dx) {
                for (uint32_t mt0 = 0; mt0 < MT0; mt0++) {
                    f16v x[NB];
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        const h8 a = ld_frag(lds, f_dx + mt0 * KC + kc, lane);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) x[nb] = mfma16(a, bf[kc][nb], kc == 0 ? FZ : x[nb]);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        if constexpr (HEAD) {
                            // rows 16..31 of the tile = gradient of h columns 0..15 (staged shifted); column 0 comes from the density path
                            const uint64_t row = row0 + nb * 32 + c;
                            if (row < B) {
                                h4 lo = {(_Float16)x[nb][8], (_Float16)x[nb][9], (_Float16)x[nb][10], (_Float16)x[nb][11]};
                                const h4 hi = {(_Float16)x[nb][12], (_Float16)x[nb][13], (_Float16)x[nb][14], (_Float16)x[nb][15]};
                                if (h == 0) lo[0] = h0_cur[nb];
                                *reinterpret_cast<h4 *>(grad_inputs + row * 16 + 4 * h) = lo;
                                *reinterpret_cast<h4 *>(grad_inputs + row * 16 + 8 + 4 * h) = hi;
                            }
                        } else if constexpr (RECOMP && IMODE == 1) {
                            // [in_dim/2][B] half2 planes (the encoder's [L,B,C] gradient layout): register quad q = features 32mt0 + 8q + 4h .. +3
                            const uint64_t row = row0 + nb * 32 + c;
                            if (row < B) {
                                uint32_t *gp = reinterpret_cast<uint32_t *>(grad_inputs);
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    const uint32_t col = 32 * mt0 + 8 * q + 4 * h;
                                    if (col < in_dim) {
                                        const h4 v = {(_Float16)x[nb][4 * q], (_Float16)x[nb][4 * q + 1], (_Float16)x[nb][4 * q + 2], (_Float16)x[nb][4 * q + 3]};
                                        const u32x2 w = __builtin_bit_cast(u32x2, v);
                                        gp[(uint64_t)(col / 2) * B + row] = w.x;
                                        gp[(uint64_t)(col / 2 + 1) * B + row] = w.y;
                                    }
                                }
                            }
                        } else {
                            store_tile<false>(grad_inputs, in_dim, row0 + nb * 32 + c, B, 32 * mt0, in_dim, x[nb], h);
                        }
                    }
                }
            
            }
            // ---- dW_s: output tile `wave` (MTo x NTi tiles, at most 4 for HIDDEN, in_dim <= 64)
            {
                // a stage with 4 output tiles gives every wave one of them over all four waves' rows; one with 2 (the 16-wide output
                // stage, a 32-wide input) is split along the batch as well: waves 2, 3 take the same tiles over the rows of waves 2, 3 —
                // every wave owns an accumulator for every stage anyway, and the flush adds them up
                const uint32_t NTi = (IN + 31) / 32, MTo = (OUT + 31) / 32, ntile = MTo * NTi;
                const uint32_t ksplit = (ntile * 2 <= 4) ? 2u : 1u;
#ifdef FOC_TIMING_NO_DW
                if (B == 0xFFFFFFFFu)                      // timing build: the weight-gradient section never runs
#endif
                if (HIDDEN == 64 || wave < ntile * ksplit) {       // hidden 64: every stage has 4 (tile, batch half) pairs, one per wave
                    // The MFMA sections (weight gradients, then the delta chain) run at raised issue priority: the other wave on this SIMD belongs to
                    // the other workgroup and is somewhere else in its group — when both can issue, the one feeding the matrix pipe goes first and
                    // the other one's conversions / LDS writes fill the gaps. rocprofv3 averages, same box: 186.6 + 141.7 -> 181.2 + 139.7 us
                    // (priority in the weight-gradient section alone 183.2 + 138.8; in the tile-writing section before a barrier: slower).
                    __builtin_amdgcn_s_setprio(FOC_MLP_SETPRIO);
                    const uint32_t tile = wave % ntile, kpart = wave / ntile;
                    const uint32_t mt = tile / NTi, nt = tile % NTi;
                    const uint32_t o = 32 * mt + (lane & 31), i = 32 * nt + (lane & 31);
                    const int q = (lane & 15) >> 2, p = lane & 3, cg = 16 * ((lane >> 4) & 1);
#pragma unroll
                    for (int wi = 0; wi < 4; wi++) {
                        if (ksplit == 2u && wi >= 2) break;
                        const uint32_t wv = ksplit == 2u ? 2u * kpart + wi : (uint32_t)wi;
                        const _Float16 *tD = sD + wv * RW * WD, *tA = sA + wv * RW * WA;
#pragma unroll
                        for (int ks = 0; ks < RW / 16; ks++) {
                            // Which batch row plays k = 8 h + j of the MFMA is free as long as both operands agree (the sum over k has no order).
                            // Rows 16 ks + 4 q + h (j = q) and + 2 (j = 4 + q): the four rows one 32-lane LDS group reads are 4 apart, and with the
                            // tiles' 36-dword row stride 4 rows are 16 banks apart — each q gets its own quarter of the 64 banks. Consecutive rows
                            // (stride 36 dwords: bank offsets 0, 36, 8, 44) overlapped pairwise: a 2-way conflict on half of these reads, the
                            // bulk of the kernel's LDS cycles (SQ_LDS_BANK_CONFLICT 40 % of SQ_LDS_IDX_ACTIVE, LDS 48 % busy).
                            const int k0 = 16 * ks + 4 * q + h;
                            const s4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(tD + k0 * WD + 32 * mt + cg + 4 * p));
                            const s4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(tD + (k0 + 2) * WD + 32 * mt + cg + 4 * p));
                            const s4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(tA + k0 * WA + 32 * nt + cg + 4 * p));
                            const s4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(tA + (k0 + 2) * WA + 32 * nt + cg + 4 * p));
                            const u32x2 A0 = __builtin_bit_cast(u32x2, a0), A1 = __builtin_bit_cast(u32x2, a1);
                            const u32x2 B0 = __builtin_bit_cast(u32x2, b0), B1 = __builtin_bit_cast(u32x2, b1);
                            u32x4 av = {A0.x, A0.y, A1.x, A1.y}, bv = {B0.x, B0.y, B1.x, B1.y};
                            // columns past OUT / IN of the LDS tiles hold stale values: only the 16-wide output stage and input widths
                            // that are not a multiple of 32 have such columns inside a 32-wide tile
                            if (s == 0 || (HIDDEN & 31)) { if (!(o < OUT)) av = u32x4{0u, 0u, 0u, 0u}; }
                            if (s < NL ? (HIDDEN & 31) != 0 : (in_dim & 31u) != 0u) { if (!(i < IN)) bv = u32x4{0u, 0u, 0u, 0u}; }
                            dwacc[s] = mfma16(__builtin_bit_cast(h8, av), __builtin_bit_cast(h8, bv), dwacc[s]);
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            // ---- next delta (chained in registers), masked by A_s read back in the accumulator layout from this wave's own tile
            __builtin_amdgcn_s_setprio(FOC_MLP_SETPRIO);
            if (s == 0) {
                h8 bg[NB];
#pragma unroll
                for (int nb = 0; nb < NB; nb++) bg[nb] = *reinterpret_cast<const h8 *>(myD + (nb * 32 + c) * WD + 8 * h);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const h8 a = ld_frag(lds, mt, lane);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, bg[nb], FZ);
                }
            } else if (s < NL) {
                const uint32_t fbase = f_hidden + (NL - 1 - s) * MT * KC;     // hidden matrix fl-1 with fl = NL - s
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const h8 a = ld_frag(lds, fbase + mt * KC + kc, lane);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, bf[kc][nb], kc == 0 ? FZ : acc[mt][nb]);
                    }
            }
            if (!RECOMP && s < NL && relu) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t col = 32 * mt + 8 * q + 4 * h;
                            if (col < HIDDEN) {
                                if constexpr (RECOMP) {
                                    // neuron 32mt + 8q + 4h + e = chain_k(2mt + (q>>1), h, 4(q&1) + e)
#pragma unroll
                                    for (int e = 0; e < 4; e++)
                                        if (!(fa[NL - 1 - s][2 * mt + (q >> 1)][nb][4 * (q & 1) + e] > (_Float16)0)) acc[mt][nb][4 * q + e] = 0.0f;
                                } else {
                                    const h4 f = *reinterpret_cast<const h4 *>(myA + (nb * 32 + c) * WA + col);
#pragma unroll
                                    for (int e = 0; e < 4; e++) if (!(f[e] > (_Float16)0)) acc[mt][nb][4 * q + e] = 0.0f;
                                }
                            }
                        }
            }
            __builtin_amdgcn_s_setprio(0);
#ifndef FOC_TIMING_NO_BARRIER
            foc_lds_barrier();     // every wave is done reading sD / sA of this stage
#endif
        }
    }
    // ---- flush the weight-gradient tiles: every workgroup writes ITS partial tiles to a slot of its own with plain coalesced stores
    // (slot layout [stage][wave][reg][lane]: a wave instruction stores 256 contiguous bytes); k_mlp_dw_reduce adds the slots up in a
    // fixed order. As fp32 atomics into ONE blob-shaped workspace the flush was 512 workgroups x 7-11 K adds on the same 28-45 KB:
    // 35-50 us of serialised same-address atomics at the end of every launch whatever the batch (the whole kernel takes 67 / 89 us on
    // the 0.5 M-sample batches of the occupancy sampler), and sums that depended on arrival order.
    // A stage with at most two tiles was split along the batch (waves 2, 3 hold the same tiles over the other half of the rows): the two halves
    // meet in LDS first (the tiles' LDS is free by now), so a slot carries one copy of every tile — a quarter fewer slot bytes to write here and
    // to read in k_mlp_dw_reduce (round 5).
    float *slot = ws + (uint64_t)blockIdx.x * ((NL + 1) * MLP_DW_SLOT_STAGE);
    float *meet = reinterpret_cast<float *>(sD);           // [2 tiles][16 registers][64 lanes] fp32 = 8 KiB <= the four D tiles
#pragma unroll
    for (int s = 0; s <= NL; s++) {
        const uint32_t OUT = s == 0 ? 16u : (uint32_t)HIDDEN;
        const uint32_t IN = s < NL ? (uint32_t)HIDDEN : in_dim;
        const uint32_t ntile = ((OUT + 31) / 32) * ((IN + 31) / 32), ksplit = (ntile * 2 <= 4) ? 2u : 1u;
        if (ksplit == 2u) {
            __syncthreads();                                 // the previous stage's meeting buffer has been read
            if (wave >= ntile && wave < 2u * ntile) {
#pragma unroll
                for (int reg = 0; reg < 16; reg++) meet[((wave - ntile) * 16 + reg) * 64 + lane] = dwacc[s][reg];
            }
            __syncthreads();
            if (wave < ntile) {
#pragma unroll
                for (int reg = 0; reg < 16; reg++) slot[(s * 4 + wave) * 1024 + reg * 64 + lane] = dwacc[s][reg] + meet[(wave * 16 + reg) * 64 + lane];
            }
        } else if (wave < ntile) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) slot[(s * 4 + wave) * 1024 + reg * 64 + lane] = dwacc[s][reg];
        }
    }
}

// Sum of the per-workgroup weight-gradient slots of k_mlp_bwd_fused -> the gradient blob (fp16, `gw`) or, for the object-conditioned
// colour head, the fp32 blob image k_mlp_dw_finalize_obj expands (`wsb`). One workgroup per 64 consecutive slot positions (same stage,
// wave tile and register: 256 contiguous bytes in every slot); its 16 waves split the slots (wave k takes slots k, k + 16, ...), the
// partial sums meet in LDS and are added in wave order: the result does not depend on timing (the atomic flush's did).
template <int HIDDEN>
__global__ void __launch_bounds__(1024) k_mlp_dw_reduce(const float *__restrict__ slots, uint32_t n_slots, uint32_t NL, uint32_t in_dim, uint32_t ld0,
                                                        _Float16 *__restrict__ gw, float *__restrict__ wsb) {
    __shared__ float part[16][64];
    const uint32_t lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x / 64, w = (blockIdx.x / 16) % 4, reg = blockIdx.x % 16;       // stage, wave tile, accumulator register
    const uint32_t OUT = s == 0 ? 16u : (uint32_t)HIDDEN, IN = s < NL ? (uint32_t)HIDDEN : in_dim;
    const uint32_t NTi = (IN + 31) / 32, MTo = (OUT + 31) / 32, ntile = MTo * NTi, ksplit = 1u;      // a slot holds ONE copy of every tile (k_mlp_bwd_fused's flush)
    if (w >= ntile) return;
    const uint32_t mt = w / NTi, nt = w % NTi, h = lane >> 5;
    const uint32_t o = 32 * mt + (uint32_t)acc_row((int)reg, (int)h), i = 32 * nt + (lane & 31);
    if (!(o < OUT) && !(32 * mt + (uint32_t)acc_row((int)reg, 1 - (int)h) < OUT)) return;      // rows past OUT in both lane halves (16-wide output stage)
    const uint64_t stride = (uint64_t)(NL + 1) * MLP_DW_SLOT_STAGE;
    const float *p = slots + (uint64_t)(s * 4 + w) * 1024 + reg * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (uint32_t k = 0; k < ksplit; k++) {
        const float *q = p + (uint64_t)k * ntile * 1024;
        uint32_t g = slice;
        for (; g + 48 < n_slots; g += 64) {
            a0 += q[(uint64_t)g * stride]; a1 += q[(uint64_t)(g + 16) * stride]; a2 += q[(uint64_t)(g + 32) * stride]; a3 += q[(uint64_t)(g + 48) * stride];
        }
        for (; g < n_slots; g += 16) a0 += q[(uint64_t)g * stride];
    }
    part[slice][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (slice == 0 && o < OUT && i < IN) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) v += part[k][lane];
        const uint64_t first = (uint64_t)HIDDEN * ld0, lsz = (uint64_t)HIDDEN * HIDDEN;
        const uint32_t fl = NL - s;                       // forward layer whose matrix this is (s = 0: output matrix)
        const uint64_t off = (s == 0 ? first + (uint64_t)(NL - 1) * lsz : (fl == 0 ? 0 : first + (uint64_t)(fl - 1) * lsz)) + (uint64_t)o * (s == NL ? ld0 : IN) + i;
        if (wsb) wsb[off] = v; else gw[off] = (_Float16)v;
    }
}

// Finalize of the colour head with an object feature (MlpHead): the workspace's W0 block has 48-wide rows of which the MFMAs filled
// columns 0..30 and column 31 = cs[o] = sum_b delta_0[b][o]. dW0[o][31 + j] = cs[o] * obj[j] (the input columns 31..46 hold the same
// obj[j] for every sample), dW0[o][47] = 0 (zero pad input), and grad_obj[j] = sum_o W0[o][31 + j] * cs[o] (fp32, [16]).
__global__ void __launch_bounds__(256) k_mlp_dw_finalize_obj(const float *__restrict__ ws, _Float16 *__restrict__ gw, uint32_t n, uint32_t hidden,
                                                             const _Float16 *__restrict__ W, const _Float16 *__restrict__ obj, float *__restrict__ grad_obj) {
    const uint32_t n0 = hidden * HEAD_OBJ_LD;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float v = ws[i];
        if (i < n0) {
            const uint32_t o = i / HEAD_OBJ_LD, col = i % HEAD_OBJ_LD;
            if (col >= 31) v = col < 47 ? ws[o * HEAD_OBJ_LD + 31] * (float)obj[col - 31] : 0.0f;
        }
        gw[i] = (_Float16)v;
    }
    if (blockIdx.x == 0 && threadIdx.x < 16 && grad_obj) {
        float a = 0.0f;
        for (uint32_t o = 0; o < hidden; o++) a = fmaf((float)W[o * HEAD_OBJ_LD + 31 + threadIdx.x], ws[o * HEAD_OBJ_LD + 31], a);
        grad_obj[threadIdx.x] = a;
    }
}

// ---------------------------------------------------------------- inference: sigma network -> head -> colour network in one kernel
// What NeRFNetwork.forward evaluates per sample (nerf/network_ff.py:51-75) without its intermediates ever leaving the registers:
//   h = sigma_net(enc)            [16]   MFMA chain as in k_mlp_fwd; h is rounded to fp16 like the stored network output
//   sigma = exp(h[0])                    fp32 (trunc_exp forward)
//   cin = [SH4(dir) | h[1:16] | 0]       the 16 SH values are computed on the lane, the geometry features ARE the chained operand:
//                                        element (h, j) of the sigma-net output fragment is neuron chain_k(0, h, j), so the colour net's
//                                        layer-0 weights for k-chunk 1 are staged permuted (column 15 + neuron, none for neuron 0)
//   rgb = sigmoid(color_net(cin)[0:3])   rounded to fp16 like the half sigmoid, stored as fp32
// Per sample this reads 64 B of encoding + the direction and writes 16 B, instead of also writing and re-reading h (32 B), cin (64 B)
// and the colour logits (32 B). Values are bit-identical to the separate kernels (same MFMA order, same roundings).
__device__ __forceinline__ void nf_sh16_half(float x, float y, float z, int h, h8 &out) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float o[8];
    if (h == 0) {
        o[0] = 0.28209479177387814f; o[1] = -0.48860251190291987f * y; o[2] = 0.48860251190291987f * z; o[3] = -0.48860251190291987f * x;
        o[4] = 1.0925484305920792f * xy; o[5] = -1.0925484305920792f * yz; o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        o[7] = -1.0925484305920792f * xz;
    } else {
        o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2; o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        o[2] = 2.8906114426405538f * xy * z; o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        o[4] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f); o[5] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        o[6] = 1.4453057213202769f * z * (x2 - y2); o[7] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[j] = (_Float16)o[j];
}

// BLK: the rows stand in 64-ray blocks (dir_block == 64 == the tile's rows, csrc/fixedstep.hip fs_block_row) — a tile is 64 rays at one
// depth, tile / dir_div is its block, and a wave takes CONSECUTIVE tiles: the SH values of its lanes' rays change once per block, so they
// are computed when the block changes and kept in registers (per tile they were 60 of the kernel's ~550 VALU instructions, plus the loads
// of the directions and the division that finds them).
// OBJ: FOC's object-conditioned colour network (nerf/network_tcnn.py:611-640, MlpHead above): 48-wide W0 rows, the encoded object feature's
// share W0[:, 31:47] . obj as the initial value of the colour layer-0 accumulators.
template <int NLS, int NLC, bool PLANAR, bool RELU_CT, bool BLK, bool OBJ>
__global__ void __launch_bounds__(MLP_BLOCK, 2) k_nerf_infer(const _Float16 *__restrict__ enc, const float *__restrict__ dirs, uint32_t dir_div,
                                                          uint32_t dir_block, uint32_t n_dirs, const _Float16 *__restrict__ w_sigma, const _Float16 *__restrict__ w_color, uint32_t B,
                                                          int relu_rt, float *__restrict__ sigma_out, float *__restrict__ rgb_out, const _Float16 *__restrict__ obj) {
    const int relu = RELU_CT ? 1 : relu_rt;            // ReLU as a compile-time fact (see k_mlp_bwd_fused); `false` keeps the runtime flag
    f16v FZ;                                           // constant zero C operand: the first MFMA of every chain takes the inline constant 0
#pragma unroll
    for (int e = 0; e < 16; e++) FZ[e] = 0.0f;
    constexpr int HIDDEN = 64, MT = 2, KC = 4, NB = 2, IN = 32, KS0 = 2;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    // sigma net: the forward image of k_mlp_fwd; colour net behind it: layer 0 custom (see above), hidden and output as usual
    constexpr uint32_t n_sigma = MT * KS0 + (NLS - 1) * MT * KC + KC;
    _Float16 *ldsC = lds + (size_t)n_sigma * 512;
    stage_weights_fwd<HIDDEN>(w_sigma, lds, IN, NLS);
    {
        constexpr uint32_t n0 = MT * KS0, nh = (NLC - 1) * MT * KC, total = n0 + nh + KC;
        constexpr uint32_t LD0 = OBJ ? HEAD_OBJ_LD : (uint32_t)IN;      // row width of the colour net's W0
        const _Float16 *Wh = w_color + (size_t)HIDDEN * LD0;
        const _Float16 *Wo = Wh + (size_t)(NLC - 1) * HIDDEN * HIDDEN;
        for (uint32_t idx = threadIdx.x; idx < total * 64; idx += MLP_BLOCK) {
            const uint32_t f = idx >> 6, lane = idx & 63, r = lane & 31, h = lane >> 5;
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (f < n0) {
                const uint32_t mt = f / KS0, kc = f % KS0, row = 32 * mt + r;
                if (kc == 0) v = *reinterpret_cast<const h8 *>(w_color + (size_t)row * LD0 + 8 * h);          // SH chunk, natural order
                else {
#pragma unroll
                    for (int j = 0; j < 8; j++) { const int n = chain_k(0, h, j); if (n >= 1) v[j] = w_color[(size_t)row * LD0 + 15 + n]; }
                }
            } else if (f < n0 + nh) {
                const uint32_t g = f - n0, l = g / (MT * KC), mt = (g / KC) % MT, kc = g % KC, row = 32 * mt + r;
                const _Float16 *p = Wh + (size_t)l * HIDDEN * HIDDEN + (size_t)row * HIDDEN + 16 * kc + 4 * h;
                const h4 lo = *reinterpret_cast<const h4 *>(p), hi = *reinterpret_cast<const h4 *>(p + 8);
#pragma unroll
                for (int j = 0; j < 4; j++) { v[j] = lo[j]; v[4 + j] = hi[j]; }
            } else {
                const uint32_t kc = f - n0 - nh;
                if (r < 16) {
                    const _Float16 *p = Wo + (size_t)r * HIDDEN + 16 * kc + 4 * h;
                    const h4 lo = *reinterpret_cast<const h4 *>(p), hi = *reinterpret_cast<const h4 *>(p + 8);
#pragma unroll
                    for (int j = 0; j < 4; j++) { v[j] = lo[j]; v[4 + j] = hi[j]; }
                }
            }
            *reinterpret_cast<h8 *>(ldsC + (size_t)f * 512 + lane * 8) = v;
        }
    }
    const float *obj_bias = reinterpret_cast<const float *>(ldsC + (size_t)(MT * KS0 + (NLC - 1) * MT * KC + KC) * 512);
    if constexpr (OBJ) stage_obj_bias(w_color, obj, const_cast<float *>(obj_bias), HIDDEN);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = (B + 32 * NB - 1) / (32 * NB);
    // one hidden stack: acc (pre-activation of layer 0) -> pre-activation of the output layer input; returns the B fragments of the last hidden layer
    auto hidden_stack = [&](f16v (&acc)[MT][NB], h8 (&bf)[KC][NB], const _Float16 *img, uint32_t f_hidden, int nl) {
#pragma unroll 1
        for (int l = 1; l <= nl; l++) {
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
#pragma unroll
                for (int nb = 0; nb < NB; nb++) bf[kc][nb] = relu ? acc_to_frag<true>(acc[kc >> 1][nb], kc & 1) : acc_to_frag<false>(acc[kc >> 1][nb], kc & 1);
            if (l < nl) {
                const uint32_t fbase = f_hidden + (l - 1) * MT * KC;
#pragma unroll
                for (int kc = 0; kc < KC; kc++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const h8 a = ld_frag(img, fbase + mt * KC + kc, lane);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, bf[kc][nb], kc == 0 ? FZ : acc[mt][nb]);
                    }
            }
        }
    };
    // The encoding rows and the directions of tile t + 1 are in flight while tile t goes through its 64 MFMAs: two register sets used in
    // turn (a rotation `cur = nxt` would be a v_mov of loaded registers, i.e. a wait for the loads just issued, and the kernel has two
    // waves per SIMD to cover a memory latency). Every prefetch load is unconditional — rows clamped to B - 1, the tile after the last
    // one repeats the last — so the compiler can count the loads in flight; the loop used to stop twice per tile for a full latency
    // (planes, then directions behind the sigma store).
    struct TileIn { h8 b[KS0][NB]; float d[BLK ? 1 : NB][3]; };
    auto fetch_tile = [&](uint32_t tile, TileIn &t) {
        const uint64_t row0 = (uint64_t)min(tile, n_tiles - 1u) * 32 * NB;
#pragma unroll
        for (int kc = 0; kc < KS0; kc++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const uint64_t row = min(row0 + nb * 32 + c, (uint64_t)B - 1);
                t.b[kc][nb] = PLANAR ? ld_planar8(enc, B, row, kc, h) : *reinterpret_cast<const h8 *>(enc + row * IN + 16 * kc + 8 * h);
            }
        if constexpr (!BLK)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            // row -> direction: dir_div consecutive rows per direction, or (dir_block > 0) the block-interleaved sample order of
            // csrc/fixedstep.hip (fs_block_row): dir_block rays x dir_div depths per block, the ray index fastest
            // (32-bit arithmetic: B is a uint32_t, and a 64-bit division is ~150 instructions per lane)
            const uint32_t r32 = (uint32_t)min(row0 + nb * 32 + c, (uint64_t)B - 1);
            uint32_t di = r32 / dir_div;
            if (dir_block) di = min((r32 / (dir_block * dir_div)) * dir_block + r32 % dir_block, n_dirs - 1);
            const float *dp = dirs + (uint64_t)di * 3;
            t.d[nb][0] = dp[0]; t.d[nb][1] = dp[1]; t.d[nb][2] = dp[2];
        }
    };
    uint32_t sh_blk = 0xFFFFFFFFu;                     // BLK: the block whose SH values sh_keep holds
    h8 sh_keep[NB];
    auto eval_tile = [&](uint32_t tile, const TileIn &t) {
        const uint64_t row0 = (uint64_t)tile * 32 * NB;
        f16v acc[MT][NB];
        h8 bf[KC][NB];
        // ---- sigma net
#pragma unroll
        for (int kc = 0; kc < KS0; kc++) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const h8 a = ld_frag(lds, mt * KS0 + kc, lane);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a, t.b[kc][nb], kc == 0 ? FZ : acc[mt][nb]);
            }
        }
        hidden_stack(acc, bf, lds, MT * KS0, NLS);
        f16v o[NB];
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            const h8 a = ld_frag(lds, MT * KS0 + (NLS - 1) * MT * KC + kc, lane);
#pragma unroll
            for (int nb = 0; nb < NB; nb++) o[nb] = mfma16(a, bf[kc][nb], kc == 0 ? FZ : o[nb]);
        }
        // ---- head: sigma from neuron 0 (lane half 0, element 0), SH on the lane, geometry features = the output fragment itself
        h8 geo[NB], sh[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            geo[nb] = acc_to_frag<false>(o[nb], 0);
            const uint64_t row = row0 + nb * 32 + c;
            if (h == 0 && row < B && sigma_out) sigma_out[row] = expf((float)geo[nb][0]);
            if constexpr (BLK) {
                const uint32_t blk = tile / dir_div;                          // wave-uniform
                if (blk != sh_blk) {
                    const uint32_t di = min(blk * dir_block + (uint32_t)(nb * 32 + c), n_dirs - 1u);
                    const float *dp = dirs + (uint64_t)di * 3;
                    nf_sh16_half(dp[0], dp[1], dp[2], h, sh_keep[nb]);
                    if (nb == NB - 1) sh_blk = blk;
                }
                sh[nb] = sh_keep[nb];
            } else nf_sh16_half(t.d[nb][0], t.d[nb][1], t.d[nb][2], h, sh[nb]);
        }
        // ---- colour net
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const h8 a0 = ld_frag(ldsC, mt * KS0 + 0, lane);
            if constexpr (OBJ) {
                // read per tile (pointer laundered): hoisted out of the tile loop the 32 values stay live across it and the kernel spills
                const float *bp = obj_bias;
                asm volatile("" : "+v"(bp));
                const f16v b0 = ld_obj_bias(bp, mt, h);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a0, sh[nb], b0);
            } else {
                // an empty volatile asm between the sigma network and the colour network: the scheduler no longer interleaves the two halves
                // of a tile, the kernel needs 163-168 VGPRs instead of 223-234 (three waves per SIMD instead of two) and a view renders
                // 3 % faster (0.0458 -> 0.0444 s, A/B on one box, round 3)
                { int sched_fence = 0; asm volatile("" : "+v"(sched_fence)); }
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a0, sh[nb], FZ);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const h8 a1 = ld_frag(ldsC, mt * KS0 + 1, lane);
#pragma unroll
            for (int nb = 0; nb < NB; nb++) acc[mt][nb] = mfma16(a1, geo[nb], acc[mt][nb]);
        }
        hidden_stack(acc, bf, ldsC, MT * KS0, NLC);
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            const h8 a = ld_frag(ldsC, MT * KS0 + (NLC - 1) * MT * KC + kc, lane);
#pragma unroll
            for (int nb = 0; nb < NB; nb++) o[nb] = mfma16(a, bf[kc][nb], kc == 0 ? FZ : o[nb]);
        }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const uint64_t row = row0 + nb * 32 + c;
            if (h == 0 && row < B) {
                // torch.sigmoid on the half logits: fp32 math, one rounding to fp16 (network_ff.py:73)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float x = (float)(_Float16)o[nb][k];
                    rgb_out[row * 3 + k] = (float)(_Float16)(1.0f / (1.0f + expf(-x)));
                }
            }
        }
    };
    // tiles of a wave: every (waves of the grid)-th one, or in BLK mode a run of consecutive ones
    const uint32_t n_waves = gridDim.x * MLP_WAVES, my_wave = blockIdx.x * MLP_WAVES + wave;
    const uint32_t per = BLK ? (n_tiles + n_waves - 1u) / n_waves : 1u;
    const uint32_t stride = BLK ? 1u : n_waves;
    uint32_t tile = BLK ? my_wave * per : my_wave;
    const uint32_t t_end = BLK ? min(n_tiles, tile + per) : n_tiles;
    TileIn ta, tb;
    if (tile < t_end) fetch_tile(tile, ta);
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) once, here: see k_mlp_bwd_fused
    while (tile < t_end) {
        fetch_tile(tile + stride, tb);
        eval_tile(tile, ta);
        tile += stride;
        if (tile >= t_end) break;
        fetch_tile(tile + stride, ta);
        eval_tile(tile, tb);
        tile += stride;
    }
}

// ================================================================= host side
// Per-device caches (the entry points make the stream's device current, common.h FocDeviceGuard): CU count and, per kernel, the number
// of workgroups one CU holds.
#define MLP_MAX_DEVICES 16
// floats of the fp32 blob image at the head of the backward workspace (rounded up to 256 bytes: the slots behind it stay aligned)
static inline uint64_t mlp_dw_blob_floats(uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers) {
    const uint64_t n = (uint64_t)hidden_dim * (input_dim + (uint64_t)hidden_dim * (num_layers - 1) + 16);
    return (n + 63) & ~(uint64_t)63;
}
static int mlp_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MLP_MAX_DEVICES) dev = 0;
    return dev;
}
static uint32_t mlp_num_cus() {
    static uint32_t num_cus[MLP_MAX_DEVICES];
    const int dev = mlp_device();
    if (!num_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        num_cus[dev] = (uint32_t)n;
    }
    return num_cus[dev];
}

// Workgroups of a persistent kernel that one CU holds at once (registers + LDS): the grid is capped at CUs x this, so that every
// workgroup is resident from the start — a grid of CUs x 4 on a kernel that fits 3 per CU runs a quarter of its workgroups in a second,
// three-quarters-empty round (measured: 2.025 -> 2.003 ms per training step for the two forward kernels).
static uint32_t mlp_resident_blocks(const void *kern, size_t lds) {
    static const void *seen[MLP_MAX_DEVICES][64];
    static uint32_t blocks[MLP_MAX_DEVICES][64];
    static int n_seen[MLP_MAX_DEVICES];
    const int dev = mlp_device();
    for (int i = 0; i < n_seen[dev]; i++) if (seen[dev][i] == kern) return blocks[dev][i];
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, MLP_BLOCK, lds) != hipSuccess || n < 1) n = 2;
    if (n > 8) n = 8;
    if (n_seen[dev] < 64) { seen[dev][n_seen[dev]] = kern; blocks[dev][n_seen[dev]] = (uint32_t)n; n_seen[dev]++; }
    return (uint32_t)n;
}

// hidden_dim 256 (ffmlp_wide.hip): layer-by-layer kernels with one matrix resident in LDS
int mlp_wide_forward(bool train, const void *inputs, const void *weights, uint32_t B, uint32_t in_dim, uint32_t hidden, uint32_t num_layers, int act, void *buffer,
                     void *outputs, hipStream_t st);
int mlp_wide_backward_activations(const void *grad, const void *weights, const void *fwd_buf, uint32_t B, uint32_t in_dim, uint32_t hidden, uint32_t num_layers, int act,
                                  void *bwd_buf, void *grad_inputs, hipStream_t st);

static int mlp_check(const char *who, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                     uint32_t activation, uint32_t output_activation) {
    FOC_REQUIRE(hidden_dim == 16 || hidden_dim == 32 || hidden_dim == 64 || hidden_dim == 128 || hidden_dim == 256, FOC_E_INVALID,
                "%s: hidden_dim should in [16, 32, 64, 128, 256] (got %u)", who, hidden_dim);                 // ffmlp.cu:658
    // ffmlp.cu:151-239: the reference's input layer takes any 16 m that fits shared memory; here up to 256 (the weight images must fit the 160 KiB LDS:
    // checked per launch)
    FOC_REQUIRE(input_dim > 0 && input_dim % 16 == 0 && input_dim <= 256u, FOC_E_INVALID, "%s: input_dim must be a multiple of 16 up to 256 (got %u)", who, input_dim);
    FOC_REQUIRE(output_dim <= 16, FOC_E_INVALID, "%s: output_dim must be <= 16 (got %u)", who, output_dim);
    FOC_REQUIRE(num_layers >= 2 && num_layers <= 16, FOC_E_INVALID, "%s: num_layers must be in [2,16] (got %u)", who, num_layers);
    FOC_REQUIRE(activation <= 6, FOC_E_INVALID, "%s: hidden activation must be one of relu(0) exponential(1) sine(2) sigmoid(3) squareplus(4) softplus(5) none(6) (got %u)", who, activation);
    FOC_REQUIRE(output_activation == 6, FOC_E_INVALID, "%s: output activation must be none(6) (got %u)", who, output_activation);
    return FOC_OK;
}

template <int HIDDEN>
static size_t mlp_fwd_lds(uint32_t in_dim, uint32_t num_layers) {
    constexpr int MT = (HIDDEN + 31) / 32, KC = HIDDEN / 16;
    return (size_t)(MT * (in_dim / 16) + (num_layers - 1) * MT * KC + KC) * 1024;
}
template <int HIDDEN>
static size_t mlp_bwd_lds(uint32_t in_dim, uint32_t num_layers, bool dx) {
    constexpr int MT = (HIDDEN + 31) / 32, KC = HIDDEN / 16;
    return (size_t)(MT + (num_layers - 1) * MT * KC + (dx ? ((in_dim + 31) / 32) * KC : 0)) * 1024;
}

template <int HIDDEN, bool TRAIN>
static int mlp_fwd_launch(const void *inputs, const void *weights, uint32_t B, uint32_t in_dim, uint32_t num_layers, int act,
                          void *fwd_buf, void *outputs, int planar, hipStream_t st, const MlpHead *head = nullptr) {
    const bool gen = act != FOC_ACT_RELU && act != FOC_ACT_NONE;
    FOC_REQUIRE(!gen || (!planar && !head), FOC_E_INVALID, "ffmlp_forward: activation %d is served on row-major inputs only", act);
    constexpr int NB = 1;                          // one 32-row tile per wave: 92 registers, 5 waves per SIMD (two tiles: 157 registers, 3 waves; 57 -> 54 us per 2 M rows)
    const size_t lds = mlp_fwd_lds<HIDDEN>(in_dim, num_layers) + (head && head->obj ? 256 : 0);       // + obj_bias [2][2][16] fp32
    FOC_REQUIRE(lds <= 160 * 1024, FOC_E_INVALID, "ffmlp_forward: weights (%zu B) do not fit the 160 KiB LDS", lds);
    FOC_REQUIRE(!(planar && TRAIN), FOC_E_INVALID, "ffmlp_forward: planar inputs go with the activation-free forward");
    auto kern = planar ? k_mlp_fwd<HIDDEN, NB, false, 1> : k_mlp_fwd<HIDDEN, NB, TRAIN, 0>;
    if (gen) kern = k_mlp_fwd<HIDDEN, NB, TRAIN, 0, true>;
    if constexpr (HIDDEN == 64 && !TRAIN) { if (head) kern = k_mlp_fwd<HIDDEN, NB, false, 2>; }
    else FOC_REQUIRE(!head, FOC_E_INVALID, "color_head_forward: hidden_dim must be 64");
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const uint32_t n_tiles = foc_div_up(B, 32 * NB);
    uint32_t grid = foc_div_up(n_tiles, MLP_WAVES);
    const uint32_t cap = mlp_num_cus() * mlp_resident_blocks(reinterpret_cast<const void *>(kern), lds);
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_BLOCK), lds, st, (const _Float16 *)inputs, (const _Float16 *)weights, (_Float16 *)fwd_buf,
                       (_Float16 *)outputs, B, in_dim, num_layers, act, head ? *head : MlpHead{nullptr, nullptr, 1u, 16u, nullptr});
    FOC_CHECK_LAUNCH(TRAIN ? "ffmlp_forward" : "ffmlp_inference");
    return FOC_OK;
}

template <bool TRAIN>
static int mlp_fwd(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                   uint32_t num_layers, uint32_t activation, uint32_t output_activation, void *buffer, void *outputs, void *stream, int planar = 0) {
    const char *who = TRAIN ? "ffmlp_forward" : "ffmlp_inference";
    int rc = mlp_check(who, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
    if (rc) return rc;
    if (B == 0) return FOC_OK;                      // empty tensors carry null data pointers
    FOC_REQUIRE(inputs && weights && outputs && (!TRAIN || buffer), FOC_E_INVALID, "%s: null pointer", who);
    const int act = (int)activation;
    hipStream_t st = (hipStream_t)stream;
    switch (hidden_dim) {
        case 16: return mlp_fwd_launch<16, TRAIN>(inputs, weights, B, input_dim, num_layers, act, buffer, outputs, planar, st);
        case 32: return mlp_fwd_launch<32, TRAIN>(inputs, weights, B, input_dim, num_layers, act, buffer, outputs, planar, st);
        case 64: return mlp_fwd_launch<64, TRAIN>(inputs, weights, B, input_dim, num_layers, act, buffer, outputs, planar, st);
        case 128: return mlp_fwd_launch<128, TRAIN>(inputs, weights, B, input_dim, num_layers, act, buffer, outputs, planar, st);
        case 256:                                       // one matrix fills the LDS: layer by layer (ffmlp_wide.hip)
            FOC_REQUIRE(!planar, FOC_E_INVALID, "%s: planar inputs are served up to hidden_dim 128", who);
            return mlp_wide_forward(TRAIN, inputs, weights, B, input_dim, 256, num_layers, act, buffer, outputs, st);
    }
    return FOC_E_INVALID;
}

template <int HIDDEN, int NL>
static int mlp_bwd_fused_launch(const void *grad, const void *inputs, const void *weights, const void *fwd_buf, uint32_t B, uint32_t in_dim, int relu,
                                void *bwd_buf, void *grad_inputs, void *grad_weights, float *ws, int planar, hipStream_t st, const MlpHead *head = nullptr,
                                float *grad_obj = nullptr) {
    FOC_REQUIRE(!planar || fwd_buf == nullptr, FOC_E_INVALID, "ffmlp_backward: planar inputs need the re-evaluating form (forward_buffer NULL)");
    constexpr int NB = 1, RW = 32 * NB, WD = (HIDDEN < 32 ? 32 : HIDDEN) + 8;
    const bool dx = grad_inputs != nullptr;
    const size_t lds_w = mlp_bwd_lds<HIDDEN>(in_dim, NL, dx);
    const uint32_t WA = (in_dim > (uint32_t)HIDDEN ? in_dim : (uint32_t)(HIDDEN < 32 ? 32 : HIDDEN)) + 8;
    const bool recomp = fwd_buf == nullptr;
    constexpr int MT = (HIDDEN + 31) / 32, KC = HIDDEN / 16;
    const bool has_obj = head && head->obj;
    const size_t lds = lds_w + (size_t)4 * RW * (WD + WA) * sizeof(_Float16) + (recomp ? (size_t)(MT * (in_dim / 16) + (NL - 1) * MT * KC) * 1024 : 0) + (has_obj ? 256 : 0);
    FOC_REQUIRE(lds <= 160 * 1024, FOC_E_INVALID, "ffmlp_backward: fused kernel needs %zu B of LDS", lds);
#ifdef FOC_TIMING_ONE_WG
    const size_t lds_launch = lds < 84 * 1024 ? 84 * 1024 : lds;      // timing build: one workgroup (one wave per SIMD) per CU
#else
    const size_t lds_launch = lds;
#endif
    // the re-evaluating forms exist twice: ReLU as a compile-time fact (what every NeRF network uses) and with the runtime flag
    auto kern = recomp ? (planar ? (relu ? k_mlp_bwd_fused<HIDDEN, NL, NB, true, 1, true> : k_mlp_bwd_fused<HIDDEN, NL, NB, true, 1, false>)
                                 : (relu ? k_mlp_bwd_fused<HIDDEN, NL, NB, true, 0, true> : k_mlp_bwd_fused<HIDDEN, NL, NB, true, 0, false>))
                       : k_mlp_bwd_fused<HIDDEN, NL, NB, false, 0, false>;
    const bool lean = !bwd_buf && dx;
    if constexpr (HIDDEN == 64) {
        if (recomp && relu && in_dim <= 32 && !head) {
            if (lean) kern = planar ? k_mlp_bwd_fused<HIDDEN, NL, NB, true, 1, true, true, true> : k_mlp_bwd_fused<HIDDEN, NL, NB, true, 0, true, true, true>;
            else kern = planar ? k_mlp_bwd_fused<HIDDEN, NL, NB, true, 1, true, true> : k_mlp_bwd_fused<HIDDEN, NL, NB, true, 0, true, true>;
        }
    }
    if constexpr (HIDDEN == 64 && NL <= 3) {
        if (head) kern = head->out_width == 4u ? (relu ? (lean ? k_mlp_bwd_fused<HIDDEN, NL, NB, true, 3, true, false, true> : k_mlp_bwd_fused<HIDDEN, NL, NB, true, 3, true>)
                                                       : k_mlp_bwd_fused<HIDDEN, NL, NB, true, 3, false>)
                                               : (relu ? (lean ? k_mlp_bwd_fused<HIDDEN, NL, NB, true, 2, true, false, true> : k_mlp_bwd_fused<HIDDEN, NL, NB, true, 2, true>)
                                                       : k_mlp_bwd_fused<HIDDEN, NL, NB, true, 2, false>);
    }
    else FOC_REQUIRE(!head, FOC_E_INVALID, "color_head_backward: hidden_dim must be 64 and num_layers 2 or 3");
    if (lds_launch > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch);
    const uint32_t n_w = HIDDEN * ((has_obj ? HEAD_OBJ_LD : in_dim) + HIDDEN * (NL - 1) + 16);
    uint32_t grid = foc_div_up(B, 4 * RW);
    const uint32_t cap = min(mlp_num_cus() * 2, MLP_DW_MAX_SLOTS);
    if (grid > cap) grid = cap;
    // workspace: [fp32 blob image (object-conditioned head only)] [one slot of (NL + 1) x 4096 floats per workgroup] — no zero fill, no atomics
    float *slots = ws + mlp_dw_blob_floats(has_obj ? HEAD_OBJ_LD : in_dim, HIDDEN, NL);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_BLOCK), lds_launch, st, (const _Float16 *)grad, (const _Float16 *)inputs, (const _Float16 *)weights,
                       (const _Float16 *)fwd_buf, (_Float16 *)bwd_buf, (_Float16 *)grad_inputs, slots, B, in_dim, relu, (uint32_t)(lds_w / sizeof(_Float16)),
                       head ? *head : MlpHead{nullptr, nullptr, 1u, 16u, nullptr});
    FOC_CHECK_LAUNCH("ffmlp_backward(fused)");
    hipLaunchKernelGGL((k_mlp_dw_reduce<HIDDEN>), dim3((NL + 1) * 64), dim3(1024), 0, st, (const float *)slots, grid, (uint32_t)NL, in_dim,
                       has_obj ? (uint32_t)HEAD_OBJ_LD : in_dim, (_Float16 *)grad_weights, has_obj ? ws : (float *)nullptr);
    FOC_CHECK_LAUNCH("ffmlp_backward(reduce)");
    if (has_obj) {
        hipLaunchKernelGGL(k_mlp_dw_finalize_obj, dim3(foc_grid_1d(n_w, 256)), dim3(256), 0, st, ws, (_Float16 *)grad_weights, n_w, (uint32_t)HIDDEN,
                           (const _Float16 *)weights, head->obj, grad_obj);
        FOC_CHECK_LAUNCH("ffmlp_backward(finalize)");
    }
    return FOC_OK;
}

// weight gradients of every layer from the stored activations and activation gradients: split-K over the batch into the fp32 workspace, one rounding
template <int HIDDEN>
static int mlp_dw_launch(const void *grad, const void *inputs, const void *fwd_buf, const void *bwd_buf, uint32_t B, uint32_t in_dim, uint32_t num_layers,
                         void *grad_weights, float *ws, hipStream_t st) {
    const uint32_t n_w = HIDDEN * (in_dim + HIDDEN * (num_layers - 1) + 16);
    if (foc_zero_async(ws, (size_t)n_w * sizeof(float), st) != hipSuccess) { foc_set_error("ffmlp_backward: memset of workspace failed"); return FOC_E_LAUNCH; }
    uint32_t gx = foc_div_up(B, DW_CHUNK);
    const int wgs_per_cu = 8;                    // split-K workgroups per CU over all layers
    // 16 output tiles per workgroup: hidden x hidden, or hidden x in_dim for an input layer wider than the hidden layers
    const uint32_t tiles = ((HIDDEN + 31) / 32) * ((max((uint32_t)HIDDEN, in_dim) + 31) / 32), GZ = (tiles + 15) / 16;
    const uint32_t capx = foc_div_up(mlp_num_cus() * (uint32_t)wgs_per_cu, (num_layers + 1) * GZ);
    if (gx > capx) gx = capx;
    if (gx < 1) gx = 1;
    auto kern = k_mlp_dw<HIDDEN>;
    if constexpr (HIDDEN <= 128) { if (in_dim > 128) kern = k_mlp_dw<HIDDEN, 256>; }
    hipLaunchKernelGGL(kern, dim3(gx, num_layers + 1, GZ), dim3(MLP_BLOCK), 0, st, (const _Float16 *)grad, (const _Float16 *)inputs,
                       (const _Float16 *)fwd_buf, (const _Float16 *)bwd_buf, ws, B, in_dim, num_layers);
    FOC_CHECK_LAUNCH("ffmlp_backward(weights)");
    hipLaunchKernelGGL(k_mlp_dw_finalize, dim3(foc_grid_1d(n_w, 256)), dim3(256), 0, st, ws, (_Float16 *)grad_weights, n_w);
    FOC_CHECK_LAUNCH("ffmlp_backward(finalize)");
    return FOC_OK;
}

template <int HIDDEN>
static int mlp_bwd_launch(const void *grad, const void *inputs, const void *weights, const void *fwd_buf, uint32_t B, uint32_t in_dim,
                          uint32_t num_layers, int act, void *bwd_buf, void *grad_inputs, void *grad_weights, float *ws, int planar, hipStream_t st) {
    const bool gen = act != FOC_ACT_RELU && act != FOC_ACT_NONE;      // the single-pass kernel is built for ReLU / None: the others take the reference's data flow
    const int relu = act == FOC_ACT_RELU;
    // fused single-pass kernel for the shapes the NeRF networks use; FOC_MLP_BWD_FUSED=0 forces the two-kernel form (tuning / tests)
    const int use_fused = foc_opt(FOC_OPT_MLP_BWD_FUSED);
    if constexpr (HIDDEN <= 64) {
        if (use_fused && in_dim <= 64 && !gen) {
            switch (num_layers) {
                case 2: return mlp_bwd_fused_launch<HIDDEN, 2>(grad, inputs, weights, fwd_buf, B, in_dim, relu, bwd_buf, grad_inputs, grad_weights, ws, planar, st);
                case 3: return mlp_bwd_fused_launch<HIDDEN, 3>(grad, inputs, weights, fwd_buf, B, in_dim, relu, bwd_buf, grad_inputs, grad_weights, ws, planar, st);
                case 4: return mlp_bwd_fused_launch<HIDDEN, 4>(grad, inputs, weights, fwd_buf, B, in_dim, relu, bwd_buf, grad_inputs, grad_weights, ws, planar, st);
                default: break;
            }
        }
    }
    FOC_REQUIRE(bwd_buf && fwd_buf, FOC_E_INVALID, "ffmlp_backward: forward_buffer and backward_buffer are required for this shape / activation (two-kernel path)");
    FOC_REQUIRE(!planar, FOC_E_INVALID, "ffmlp_backward: planar inputs are served by the fused kernel only (hidden_dim <= 64, input_dim <= 64, 2..4 layers)");
    constexpr int NB = 2;
    const bool dx = grad_inputs != nullptr;
    const size_t lds = mlp_bwd_lds<HIDDEN>(in_dim, num_layers, dx);
    FOC_REQUIRE(lds <= 160 * 1024, FOC_E_INVALID, "ffmlp_backward: weights (%zu B) do not fit the 160 KiB LDS", lds);
    auto kern = gen ? k_mlp_bwd<HIDDEN, NB, true> : k_mlp_bwd<HIDDEN, NB, false>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const uint32_t n_tiles = foc_div_up(B, 32 * NB);
    uint32_t grid = foc_div_up(n_tiles, MLP_WAVES);
    const uint32_t cap = mlp_num_cus() * 4;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_BLOCK), lds, st, (const _Float16 *)grad, (const _Float16 *)weights, (const _Float16 *)fwd_buf,
                       (_Float16 *)bwd_buf, (_Float16 *)grad_inputs, B, in_dim, num_layers, act);
    FOC_CHECK_LAUNCH("ffmlp_backward(activations)");
    return mlp_dw_launch<HIDDEN>(grad, inputs, fwd_buf, bwd_buf, B, in_dim, num_layers, grad_weights, ws, st);
}

template <int NLS, int NLC>
static int nerf_infer_launch(const void *enc, const float *dirs, uint32_t dir_div, uint32_t dir_block, uint32_t n_dirs, const void *w_sigma, const void *w_color, uint32_t B, int relu, int planar,
                             float *sigma, float *rgb, const void *obj, hipStream_t st) {
    const size_t lds = (size_t)((2 * 2 + (NLS - 1) * 8 + 4) + (2 * 2 + (NLC - 1) * 8 + 4)) * 1024 + (obj ? 256 : 0);
    const bool blk = planar && dir_block == 64u;     // the staged render's sample order (fixedstep.hip FS_RAY_BLOCK)
    auto kern = planar ? (blk ? (relu ? k_nerf_infer<NLS, NLC, true, true, true, false> : k_nerf_infer<NLS, NLC, true, false, true, false>)
                              : (relu ? k_nerf_infer<NLS, NLC, true, true, false, false> : k_nerf_infer<NLS, NLC, true, false, false, false>))
                       : (relu ? k_nerf_infer<NLS, NLC, false, true, false, false> : k_nerf_infer<NLS, NLC, false, false, false, false>);
    if (obj) {
        // the object-conditioned form is built for what FOC runs: ReLU networks on the encoder's planes
        FOC_REQUIRE(planar && relu, FOC_E_INVALID, "nerf_field_inference: an object feature needs planar encodings and ReLU networks");
        kern = blk ? k_nerf_infer<NLS, NLC, true, true, true, true> : k_nerf_infer<NLS, NLC, true, true, false, true>;
    }
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    uint32_t grid = foc_div_up(foc_div_up(B, 64), MLP_WAVES);
    const uint32_t cap = mlp_num_cus() * mlp_resident_blocks(reinterpret_cast<const void *>(kern), lds);
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_BLOCK), lds, st, (const _Float16 *)enc, dirs, dir_div, dir_block, n_dirs, (const _Float16 *)w_sigma,
                       (const _Float16 *)w_color, B, relu, sigma, rgb, (const _Float16 *)obj);
    FOC_CHECK_LAUNCH("nerf_field_inference");
    return FOC_OK;
}

extern "C" {

int foc_ffmlp_forward(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                      uint32_t num_layers, uint32_t activation, uint32_t output_activation, void *forward_buffer, void *outputs, void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs);
    // forward_buffer == NULL: nothing is kept for the backward pass (foc_ffmlp_backward re-evaluates the activations)
    if (!forward_buffer) return mlp_fwd<false>(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, nullptr, outputs, stream);
    return mlp_fwd<true>(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer, outputs, stream);
}

int foc_ffmlp_inference(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                        uint32_t num_layers, uint32_t activation, uint32_t output_activation, void *inference_buffer, void *outputs, void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs);
    return mlp_fwd<false>(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference_buffer, outputs, stream);
}

uint64_t foc_ffmlp_backward_workspace_bytes(uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers) {
    // the fp32 image of the weight blob (split-K sums of k_mlp_dw; the object-conditioned head's finalize) and, for the shapes
    // k_mlp_bwd_fused serves, one slot of partial tiles per workgroup of its launch
    uint64_t floats = mlp_dw_blob_floats(input_dim, hidden_dim, num_layers);
    if (hidden_dim <= 64 && input_dim <= 64 && num_layers >= 2 && num_layers <= 4) floats += (uint64_t)MLP_DW_MAX_SLOTS * (num_layers + 1) * MLP_DW_SLOT_STAGE;
    return floats * sizeof(float);
}

static int mlp_bwd_entry(const void *grad, const void *inputs, const void *weights, const void *forward_buffer, uint32_t B, uint32_t input_dim,
                         uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                         int calc_grad_inputs, void *backward_buffer, void *grad_inputs, void *grad_weights, void *workspace, uint64_t workspace_bytes, int planar,
                         void *stream) {
    // backward_buffer may be NULL: the fused kernel keeps activation gradients on chip; forward_buffer may be NULL: the fused kernel
    // then re-evaluates the activations from the inputs (the two-kernel path checks both again)
    int rc = mlp_check("ffmlp_backward", B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
    if (rc) return rc;
    if (B == 0) {                                   // empty batch (null data pointers): the weight gradient is all zeros
        FOC_REQUIRE(grad_weights, FOC_E_INVALID, "ffmlp_backward: null pointer");
        const size_t n_w = (size_t)hidden_dim * (input_dim + (size_t)hidden_dim * (num_layers - 1) + 16);
        if (foc_zero_async(grad_weights, n_w * sizeof(_Float16), (hipStream_t)stream) != hipSuccess) { foc_set_error("ffmlp_backward: memset failed"); return FOC_E_LAUNCH; }
        return FOC_OK;
    }
    FOC_REQUIRE(grad && inputs && weights && grad_weights && workspace, FOC_E_INVALID, "ffmlp_backward: null pointer");
    FOC_REQUIRE(!calc_grad_inputs || grad_inputs, FOC_E_INVALID, "ffmlp_backward: calc_grad_inputs set but grad_inputs is null");
    FOC_REQUIRE(workspace_bytes >= foc_ffmlp_backward_workspace_bytes(input_dim, hidden_dim, num_layers), FOC_E_INVALID,
                "ffmlp_backward: workspace of %llu bytes, foc_ffmlp_backward_workspace_bytes(%u, %u, %u) asks for %llu (ABI 2: blob image + per-workgroup slots)",
                (unsigned long long)workspace_bytes, input_dim, hidden_dim, num_layers,
                (unsigned long long)foc_ffmlp_backward_workspace_bytes(input_dim, hidden_dim, num_layers));
    const int relu = (int)activation;                     // the reference's activation code (0 = ReLU ... 6 = None), handed on as it is
    hipStream_t st = (hipStream_t)stream;
    void *gi = calc_grad_inputs ? grad_inputs : nullptr;
    switch (hidden_dim) {
        case 16: return mlp_bwd_launch<16>(grad, inputs, weights, forward_buffer, B, input_dim, num_layers, relu, backward_buffer, gi, grad_weights, (float *)workspace, planar, st);
        case 32: return mlp_bwd_launch<32>(grad, inputs, weights, forward_buffer, B, input_dim, num_layers, relu, backward_buffer, gi, grad_weights, (float *)workspace, planar, st);
        case 64: return mlp_bwd_launch<64>(grad, inputs, weights, forward_buffer, B, input_dim, num_layers, relu, backward_buffer, gi, grad_weights, (float *)workspace, planar, st);
        case 128: return mlp_bwd_launch<128>(grad, inputs, weights, forward_buffer, B, input_dim, num_layers, relu, backward_buffer, gi, grad_weights, (float *)workspace, planar, st);
        case 256: {
            FOC_REQUIRE(backward_buffer && forward_buffer, FOC_E_INVALID, "ffmlp_backward: hidden_dim 256 runs layer by layer and needs forward_buffer and backward_buffer");
            FOC_REQUIRE(!planar, FOC_E_INVALID, "ffmlp_backward: planar inputs are served up to hidden_dim 64");
            rc = mlp_wide_backward_activations(grad, weights, forward_buffer, B, input_dim, 256, num_layers, relu, backward_buffer, gi, st);
            if (rc) return rc;
            return mlp_dw_launch<256>(grad, inputs, forward_buffer, backward_buffer, B, input_dim, num_layers, grad_weights, (float *)workspace, st);
        }
    }
    return FOC_E_INVALID;
}

int foc_ffmlp_backward(const void *grad, const void *inputs, const void *weights, const void *forward_buffer, uint32_t B, uint32_t input_dim,
                       uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                       int calc_grad_inputs, void *backward_buffer, void *grad_inputs, void *grad_weights, void *workspace, uint64_t workspace_bytes,
                       void *stream) {
    FocDeviceGuard foc_guard_(stream, grad);
    return mlp_bwd_entry(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                         calc_grad_inputs, backward_buffer, grad_inputs, grad_weights, workspace, workspace_bytes, 0, stream);
}

int foc_ffmlp_forward_planar(const void *inputs_planar, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                             uint32_t num_layers, uint32_t activation, uint32_t output_activation, void *outputs, void *stream) {
    FocDeviceGuard foc_guard_(stream, inputs_planar);
    return mlp_fwd<false>(inputs_planar, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, nullptr, outputs, stream, 1);
}

int foc_ffmlp_backward_planar(const void *grad, const void *inputs_planar, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                              uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation, int calc_grad_inputs,
                              void *grad_inputs_planar, void *grad_weights, void *workspace, uint64_t workspace_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad);
    return mlp_bwd_entry(grad, inputs_planar, weights, nullptr, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                         calc_grad_inputs, nullptr, grad_inputs_planar, grad_weights, workspace, workspace_bytes, 1, stream);
}

int foc_nerf_field_inference(const void *enc, int enc_planar, const float *dirs, uint32_t dir_div, uint32_t dir_block, uint32_t n_dirs,
                             const void *sigma_weights, uint32_t sigma_layers,
                             const void *color_weights, uint32_t color_layers, uint32_t hidden_dim, uint32_t activation, uint32_t B, float *sigma,
                             float *rgb, const void *obj_feat, void *stream) {
    FocDeviceGuard foc_guard_(stream, enc);
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(enc && dirs && sigma_weights && color_weights && rgb, FOC_E_INVALID, "nerf_field_inference: null pointer");
    FOC_REQUIRE(hidden_dim == 64 && dir_div >= 1, FOC_E_INVALID, "nerf_field_inference: hidden_dim must be 64 (got %u)", hidden_dim);
    FOC_REQUIRE(dir_block == 0 || n_dirs >= 1, FOC_E_INVALID, "nerf_field_inference: the block-interleaved row order needs the number of directions");
    FOC_REQUIRE((uint64_t)dir_block * dir_div < (1ull << 32), FOC_E_INVALID, "nerf_field_inference: dir_block * dir_div must fit 32 bits");
    FOC_REQUIRE(activation == 0 || activation == 6, FOC_E_INVALID, "nerf_field_inference: hidden activation must be relu(0) or none(6)");
    const int relu = activation == 0;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t key = sigma_layers * 10 + color_layers;
    switch (key) {
        case 22: return nerf_infer_launch<2, 2>(enc, dirs, dir_div, dir_block, n_dirs, sigma_weights, color_weights, B, relu, enc_planar, sigma, rgb, obj_feat, st);
        case 23: return nerf_infer_launch<2, 3>(enc, dirs, dir_div, dir_block, n_dirs, sigma_weights, color_weights, B, relu, enc_planar, sigma, rgb, obj_feat, st);
        case 33: return nerf_infer_launch<3, 3>(enc, dirs, dir_div, dir_block, n_dirs, sigma_weights, color_weights, B, relu, enc_planar, sigma, rgb, obj_feat, st);
        default: foc_set_error("nerf_field_inference: layer counts (%u, %u) are not built (2/2, 2/3, 3/3)", sigma_layers, color_layers); return FOC_E_INVALID;
    }
}

// The colour network of the fixed-step training path, fed from the sigma network's output rows and a per-ray SH table (input mode 2).
int foc_color_head_forward(const void *h, const void *ray_sh, uint32_t samples_per_ray, const void *weights, uint32_t B, uint32_t hidden_dim,
                           uint32_t num_layers, uint32_t activation, void *outputs, uint32_t out_width, const void *obj_feat, void *stream) {
    FocDeviceGuard foc_guard_(stream, h);
    int rc = mlp_check("color_head_forward", B, 32, 16, hidden_dim, num_layers, activation, 6);
    if (rc) return rc;
    if (B == 0) return FOC_OK;
    FOC_REQUIRE(h && ray_sh && weights && outputs, FOC_E_INVALID, "color_head_forward: null pointer");
    FOC_REQUIRE(hidden_dim == 64 && samples_per_ray >= 1, FOC_E_INVALID, "color_head_forward: hidden_dim must be 64 (got %u), samples_per_ray >= 1", hidden_dim);
    FOC_REQUIRE(out_width == 16 || out_width == 4, FOC_E_INVALID, "color_head_forward: out_width must be 16 or 4 (got %u)", out_width);
    const MlpHead hd{(const _Float16 *)ray_sh, nullptr, samples_per_ray, out_width, (const _Float16 *)obj_feat};
    FOC_REQUIRE(activation == FOC_ACT_RELU || activation == FOC_ACT_NONE, FOC_E_INVALID, "color_head_forward: hidden activation must be relu(0) or none(6)");
    return mlp_fwd_launch<64, false>(h, weights, B, 32, num_layers, (int)activation, nullptr, outputs, 0, (hipStream_t)stream, &hd);
}

int foc_color_head_backward(const void *grad, const void *h, const void *ray_sh, uint32_t samples_per_ray, const void *grad_h0, const void *weights,
                            uint32_t B, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, void *grad_h, void *grad_weights, void *workspace,
                            uint64_t workspace_bytes, uint32_t out_width, const void *obj_feat, float *grad_obj, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad);
    int rc = mlp_check("color_head_backward", B, 32, 16, hidden_dim, num_layers, activation, 6);
    if (rc) return rc;
    FOC_REQUIRE(hidden_dim == 64 && (num_layers == 2 || num_layers == 3) && samples_per_ray >= 1, FOC_E_INVALID,
                "color_head_backward: hidden_dim must be 64 and num_layers 2 or 3 (got %u, %u)", hidden_dim, num_layers);
    if (B == 0) {
        FOC_REQUIRE(grad_weights, FOC_E_INVALID, "color_head_backward: null pointer");
        const size_t n_w = (size_t)64 * ((obj_feat ? HEAD_OBJ_LD : 32u) + (size_t)64 * (num_layers - 1) + 16);
        if (foc_zero_async(grad_weights, n_w * sizeof(_Float16), (hipStream_t)stream) != hipSuccess) { foc_set_error("color_head_backward: memset failed"); return FOC_E_LAUNCH; }
        if (grad_obj && foc_zero_async(grad_obj, 16 * sizeof(float), (hipStream_t)stream) != hipSuccess) { foc_set_error("color_head_backward: memset failed"); return FOC_E_LAUNCH; }
        return FOC_OK;
    }
    FOC_REQUIRE(grad && h && ray_sh && weights && grad_h && grad_weights && workspace, FOC_E_INVALID, "color_head_backward: null pointer");
    // with an object feature the slots start behind a 48-wide blob image: the buffer must have been sized for input_dim 48
    FOC_REQUIRE(workspace_bytes >= foc_ffmlp_backward_workspace_bytes(obj_feat ? HEAD_OBJ_LD : 32u, 64, num_layers), FOC_E_INVALID,
                "color_head_backward: workspace of %llu bytes, foc_ffmlp_backward_workspace_bytes(%u, 64, %u) asks for %llu", (unsigned long long)workspace_bytes,
                obj_feat ? HEAD_OBJ_LD : 32u, num_layers, (unsigned long long)foc_ffmlp_backward_workspace_bytes(obj_feat ? HEAD_OBJ_LD : 32u, 64, num_layers));
    FOC_REQUIRE(out_width == 16 || out_width == 4, FOC_E_INVALID, "color_head_backward: out_width must be 16 or 4 (got %u)", out_width);
    FOC_REQUIRE(activation == FOC_ACT_RELU || activation == FOC_ACT_NONE, FOC_E_INVALID, "color_head_backward: hidden activation must be relu(0) or none(6)");
    const MlpHead hd{(const _Float16 *)ray_sh, (const _Float16 *)grad_h0, samples_per_ray, out_width, (const _Float16 *)obj_feat};
    const int relu = activation == 0;
    if (num_layers == 2) return mlp_bwd_fused_launch<64, 2>(grad, h, weights, nullptr, B, 32, relu, nullptr, grad_h, grad_weights, (float *)workspace, 0, (hipStream_t)stream, &hd, grad_obj);
    return mlp_bwd_fused_launch<64, 3>(grad, h, weights, nullptr, B, 32, relu, nullptr, grad_h, grad_weights, (float *)workspace, 0, (hipStream_t)stream, &hd, grad_obj);
}

int foc_allocate_splitk(uint64_t size) { (void)size; return FOC_OK; }
int foc_free_splitk(void) { return FOC_OK; }

} // extern "C"
