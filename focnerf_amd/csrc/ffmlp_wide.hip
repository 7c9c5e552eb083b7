// ffmlp_wide.hip — the fully fused MLP at hidden_dim 256 (ffmlp.cu:652-658 dispatches hidden in {16, 32, 64, 128, 256}).
//
// The kernels of ffmlp.hip keep EVERY layer's weights of a network in LDS and chain the layers in registers; a 256 x 256 matrix is
// 128 KiB, so one layer fills the CU's 160 KiB and that form stops at 128. No network of the path is wider than 64 (nerf/network_ff.py:31-49):
// 256 is served layer by layer, each launch a persistent kernel with its ONE matrix resident in LDS as MFMA A-fragments:
//     Y[b, n] = epi( sum_k X[b, k] * Wop[n, k] ),   Wop = W (forward) or W^T (activation gradients)
// evaluated transposed like the fused kernels (Y^T tile = Wop tile x X^T tile on v_mfma_f32_32x32x16_f16): a wave owns 32 rows of the batch, reads its
// B operands straight from the row-major activations (16 bytes per lane and k-chunk), accumulates the <= 8 neuron tiles in 128 registers (fp32) and
// stores fp16 — after its last read, so a layer may run IN PLACE (inference needs one [B, hidden] buffer, as ffmlp.cu:673-709).
// Activations travel through memory between layers — the reference's own data flow (forward_buffer / backward_buffer [layers, B, hidden],
// ffmlp.py:31, :73), which this shape keeps; the weight gradients reuse the split-K kernel of ffmlp.hip (k_mlp_dw<256>, output tiles dealt over
// blockIdx.z). Same arithmetic as the narrower widths: fp16 operands, fp32 accumulation, one rounding per layer, ReLU on the rounded value.
#include "common.h"
#include "activations.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define WIDE_BLOCK 256
#define WIDE_MT 8                                   // neuron tiles of 32 a wave accumulates: N <= 256

// epilogues
#define WIDE_NONE 0                                 // y
#define WIDE_RELU 1                                 // max(y, 0)                         (forward hidden layers)
#define WIDE_MASK 2                                 // y where mask[b, n] > 0 else 0     (ReLU transfer of the backward pass, utils.h:540-545)
#define WIDE_ACT 3                                  // act(y), any of the reference's hidden activations (activations.h)
#define WIDE_TRANSFER 4                             // y * factor(mask[b, n]): their backward transfer on the stored post-activations

// W: the layer's [n_w, k_w] row-major matrix (neuron = row). TRANS = false: output neuron n = row n, reduction over the row (N = n_w, K = k_w).
// TRANS = true: output n = COLUMN n of W, reduction over the rows (N = k_w, K = n_w; rows of W past `n_valid` do not exist: the padded output layer).
template <bool TRANS, int EPI>
__global__ void __launch_bounds__(WIDE_BLOCK, 1) k_wide_layer(const _Float16 *__restrict__ X, uint32_t ldx, const _Float16 *__restrict__ W, uint32_t n_w, uint32_t k_w,
                                                             _Float16 *__restrict__ Y, uint32_t ldy, const _Float16 *__restrict__ mask, uint32_t ldm, uint32_t B, int act) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const uint32_t N = TRANS ? k_w : n_w, K = TRANS ? n_w : k_w;
    const uint32_t MT = (N + 31) / 32, KC = (K + 15) / 16;
    // stage the A fragments: fragment (mt, kc), lane (r, h), element j = Wop[32 mt + r][16 kc + 8 h + j]
    for (uint32_t idx = threadIdx.x; idx < MT * KC * 64; idx += WIDE_BLOCK) {
        const uint32_t f = idx >> 6, lane = idx & 63, r = lane & 31, h = lane >> 5;
        const uint32_t mt = f / KC, kc = f % KC, n = 32 * mt + r, k0 = 16 * kc + 8 * h;
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < N) {
            if (!TRANS) {
                if (k0 + 8 <= K) v = *reinterpret_cast<const h8 *>(W + (size_t)n * k_w + k0);
                else {
#pragma unroll
                    for (int j = 0; j < 8; j++) if (k0 + j < K) v[j] = W[(size_t)n * k_w + k0 + j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) if (k0 + j < K) v[j] = W[(size_t)(k0 + j) * k_w + n];
            }
        }
        *reinterpret_cast<h8 *>(lds + (size_t)f * 512 + lane * 8) = v;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = (B + 31) / 32;
    f16v FZ;
#pragma unroll
    for (int e = 0; e < 16; e++) FZ[e] = 0.0f;
    for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const uint64_t row = (uint64_t)tile * 32 + c;
        const uint64_t rl = row < B ? row : (uint64_t)B - 1;                 // ragged last tile: computed on a clamped row, not stored
        f16v acc[WIDE_MT];
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; mt++) acc[mt] = FZ;
        h8 b = *reinterpret_cast<const h8 *>(X + rl * ldx + 8 * h);
        for (uint32_t kc = 0; kc < KC; kc++) {
            h8 bn = b;
            if (kc + 1 < KC) bn = *reinterpret_cast<const h8 *>(X + rl * ldx + 16 * (kc + 1) + 8 * h);      // next k-chunk in flight under this one's MFMAs
#pragma unroll
            for (int mt = 0; mt < WIDE_MT; mt++) {
                if ((uint32_t)mt < MT) {
                    const h8 a = *reinterpret_cast<const h8 *>(lds + (size_t)(mt * KC + kc) * 512 + lane * 8);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[mt], 0, 0, 0);
                }
            }
            b = bn;
        }
        if (row < B) {
            // lane (c, h) owns sample `row` and, per register quad q of tile mt, the 4 consecutive neurons 32 mt + 8 q + 4 h .. + 3
#pragma unroll
            for (int mt = 0; mt < WIDE_MT; mt++) {
                if ((uint32_t)mt < MT) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t col = 32 * mt + 8 * q + 4 * h;
                        if (col < N) {
                            h4 v;
#pragma unroll
                            for (int e = 0; e < 4; e++) v[e] = (_Float16)acc[mt][4 * q + e];
                            if (EPI == WIDE_RELU) {
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] = v[e] > (_Float16)0 ? v[e] : (_Float16)0;
                            } else if (EPI == WIDE_MASK) {
                                const h4 m = *reinterpret_cast<const h4 *>(mask + row * ldm + col);
#pragma unroll
                                for (int e = 0; e < 4; e++) if (!(m[e] > (_Float16)0)) v[e] = (_Float16)0;
                            } else if (EPI == WIDE_ACT) {
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] = foc_act_forward(v[e], act);
                            } else if (EPI == WIDE_TRANSFER) {
                                const h4 m = *reinterpret_cast<const h4 *>(mask + row * ldm + col);
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] = foc_act_backward(v[e], m[e], act);
                            }
                            *reinterpret_cast<h4 *>(Y + row * ldy + col) = v;
                        }
                    }
                }
            }
        }
    }
}

static uint32_t wide_num_cus() {
    static uint32_t n_cus[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!n_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n_cus[dev] = (uint32_t)n;
    }
    return n_cus[dev];
}

template <bool TRANS, int EPI>
static int wide_launch(const char *what, const void *X, uint32_t ldx, const void *W, uint32_t n_w, uint32_t k_w, void *Y, uint32_t ldy, const void *mask, uint32_t ldm,
                       uint32_t B, hipStream_t st, int act = FOC_ACT_NONE) {
    const uint32_t N = TRANS ? k_w : n_w, K = TRANS ? n_w : k_w;
    FOC_REQUIRE(N <= 32 * WIDE_MT && K <= 256 && N % 4 == 0, FOC_E_INVALID, "%s: layer %u x %u is beyond the wide path (<= 256 x 256)", what, N, K);
    FOC_REQUIRE(ldx % 8 == 0 && ldx >= ((K + 15) / 16) * 16, FOC_E_INVALID, "%s: activation rows must hold whole 16-wide k-chunks (ld %u, K %u)", what, ldx, K);
    const size_t lds = (size_t)((N + 31) / 32) * ((K + 15) / 16) * 1024;
    auto kern = k_wide_layer<TRANS, EPI>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    uint32_t grid = foc_div_up(foc_div_up(B, 32), 4);
    const uint32_t cap = wide_num_cus() * (lds > 80 * 1024 ? 1u : 2u);
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WIDE_BLOCK), lds, st, (const _Float16 *)X, ldx, (const _Float16 *)W, n_w, k_w, (_Float16 *)Y, ldy, (const _Float16 *)mask, ldm, B, act);
    FOC_CHECK_LAUNCH(what);
    return FOC_OK;
}

// ---- forward: buffer = forward_buffer [num_layers, B, hidden] (training: every post-activation is kept) or inference_buffer [B, hidden] (layers in place)
int mlp_wide_forward(bool train, const void *inputs, const void *weights, uint32_t B, uint32_t in_dim, uint32_t hidden, uint32_t num_layers, int act, void *buffer,
                     void *outputs, hipStream_t st) {
    const int relu = act == FOC_ACT_RELU;
    const bool gen = act != FOC_ACT_RELU && act != FOC_ACT_NONE;
    const char *who = train ? "ffmlp_forward" : "ffmlp_inference";
    FOC_REQUIRE(buffer, FOC_E_INVALID, "%s: hidden_dim %u runs layer by layer and needs the %s", who, hidden, train ? "forward_buffer" : "inference_buffer [B, hidden_dim]");
    const _Float16 *W = (const _Float16 *)weights;
    _Float16 *buf = (_Float16 *)buffer;
    const void *x = inputs;
    uint32_t K = in_dim;
    for (uint32_t l = 0; l < num_layers; l++) {
        _Float16 *y = train ? buf + (size_t)l * B * hidden : buf;
        const int rc = gen ? wide_launch<false, WIDE_ACT>(who, x, K, W, hidden, K, y, hidden, nullptr, 0, B, st, act)
                     : relu ? wide_launch<false, WIDE_RELU>(who, x, K, W, hidden, K, y, hidden, nullptr, 0, B, st)
                            : wide_launch<false, WIDE_NONE>(who, x, K, W, hidden, K, y, hidden, nullptr, 0, B, st);
        if (rc) return rc;
        W += (size_t)hidden * K;
        x = y;
        K = hidden;
    }
    return wide_launch<false, WIDE_NONE>(who, x, hidden, W, 16, hidden, outputs, 16, nullptr, 0, B, st);
}

// ---- activation gradients: backward_buffer[k] = gradient w.r.t. the post-ReLU output of forward layer num_layers-1-k (ffmlp.cu:410-518); grad_inputs or NULL
int mlp_wide_backward_activations(const void *grad, const void *weights, const void *fwd_buf, uint32_t B, uint32_t in_dim, uint32_t hidden, uint32_t num_layers, int act,
                                  void *bwd_buf, void *grad_inputs, hipStream_t st) {
    const int relu = act == FOC_ACT_RELU;
    const bool gen = act != FOC_ACT_RELU && act != FOC_ACT_NONE;
    const _Float16 *W0 = (const _Float16 *)weights;
    const _Float16 *Wh = W0 + (size_t)hidden * in_dim;
    const _Float16 *Wo = Wh + (size_t)(num_layers - 1) * hidden * hidden;
    const _Float16 *fb = (const _Float16 *)fwd_buf;
    _Float16 *bb = (_Float16 *)bwd_buf;
    const void *src = grad;
    uint32_t ld_src = 16;
    for (uint32_t k = 0; k < num_layers; k++) {
        const uint32_t fl = num_layers - 1 - k;
        const _Float16 *Wm = k == 0 ? Wo : Wh + (size_t)fl * hidden * hidden;          // hidden matrix fl maps fwd[fl] -> fwd[fl+1]
        const uint32_t n_w = k == 0 ? 16u : hidden;
        _Float16 *dst = bb + (size_t)k * B * hidden;
        const _Float16 *m = fb + (size_t)fl * B * hidden;
        const int rc = gen ? wide_launch<true, WIDE_TRANSFER>("ffmlp_backward", src, ld_src, Wm, n_w, hidden, dst, hidden, m, hidden, B, st, act)
                     : relu ? wide_launch<true, WIDE_MASK>("ffmlp_backward", src, ld_src, Wm, n_w, hidden, dst, hidden, m, hidden, B, st)
                            : wide_launch<true, WIDE_NONE>("ffmlp_backward", src, ld_src, Wm, n_w, hidden, dst, hidden, nullptr, 0, B, st);
        if (rc) return rc;
        src = dst;
        ld_src = hidden;
    }
    if (grad_inputs) return wide_launch<true, WIDE_NONE>("ffmlp_backward", src, hidden, W0, hidden, in_dim, grad_inputs, in_dim, nullptr, 0, B, st);
    return FOC_OK;
}
