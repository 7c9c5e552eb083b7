// combine.hip — multi-object combine (COMBINED.py:247-251, 141-200) and library bookkeeping.
//
// foc_combine_select      : per-sample strict-'>' max-density select (first object wins ties).
// foc_combine_pack_keys / foc_combine_unpack : the same select expressed as an order-preserving
//   64-bit key, so K objects living on K GPUs can be merged with ONE RCCL all-reduce(MAX) of the
//   keys plus one all-reduce(SUM) of rgb masked to the winning rank (exactly one non-zero
//   contributor per sample -> bit-exact).
// foc_composite_fixed_steps : the fixed-step alpha composite the reference writes with
//   torch.cumprod (COMBINED.py:141-200 / nerf/renderer.py:169-221), one wave per ray with a
//   wave product-scan over 64 samples at a time.
#include "common.h"
#include <string.h>

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";

void foc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- kernels
__global__ void __launch_bounds__(256) k_combine_select(const float *__restrict__ dens, const float *__restrict__ rgb,
                                                        float *__restrict__ max_dens, float *__restrict__ best_rgb, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const float d = dens[i], m = max_dens[i];
        if (d > m) { best_rgb[i * 3] = rgb[i * 3]; best_rgb[i * 3 + 1] = rgb[i * 3 + 1]; best_rgb[i * 3 + 2] = rgb[i * 3 + 2]; }
        // torch.maximum propagates NaN
        max_dens[i] = (d != d || m != m) ? __builtin_nanf("") : (d > m ? d : m);
    }
}

__global__ void __launch_bounds__(256) k_combine_pack(const float *__restrict__ dens, uint32_t rank, uint64_t *__restrict__ keys, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        float d = dens[i];
        d = d > 0.0f ? d : 0.0f;                       // sigma >= 0 (trunc_exp): bit pattern is order preserving; -0/NaN -> 0
        keys[i] = ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)(0xFFFFFFFFu - rank);
    }
}

__global__ void __launch_bounds__(256) k_combine_unpack(const uint64_t *__restrict__ keys, uint32_t rank, const float *__restrict__ rgb,
                                                        float *__restrict__ max_dens, float *__restrict__ masked_rgb, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t k = keys[i];
        const bool mine = (uint32_t)(k & 0xFFFFFFFFu) == 0xFFFFFFFFu - rank;
        max_dens[i] = __uint_as_float((uint32_t)(k >> 32));
        masked_rgb[i * 3] = mine ? rgb[i * 3] : 0.0f;
        masked_rgb[i * 3 + 1] = mine ? rgb[i * 3 + 1] : 0.0f;
        masked_rgb[i * 3 + 2] = mine ? rgb[i * 3 + 2] : 0.0f;
    }
}

// One wave per ray. z_i = near + (far-near)*lin_i with lin = torch.linspace(0,1,T) evaluated the way
// torch fills it (symmetric halves); deltas = diff(z), last = (far-near)/T;
// w_i = alpha_i * prod_{j<i}(1 - alpha_j + 1e-15).
__global__ void __launch_bounds__(256) k_composite_fixed(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                         const float *__restrict__ nears, const float *__restrict__ fars,
                                                         uint32_t N, uint32_t T, float bg, float *__restrict__ image4, float *__restrict__ depth) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float near = nears[n], far = fars[n];
    const float span = far - near;
    const float sample_dist = span / (float)T;
    const float step = 1.0f / (float)(T - 1);
    float Tc = 1.0f, r = 0, g = 0, b = 0, a = 0, d = 0, ws = 0;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < T;
        float alpha = 0, sigma = 0, c0 = 0, c1 = 0, c2 = 0, oz = 0;
        if (valid) {
            const float l0 = (i < T / 2) ? (step * (float)i) : fmaf(-step, (float)(T - 1 - i), 1.0f);   // device linspace: see fixedstep.hip
            const float z = near + span * l0;
            float delta = sample_dist;
            if (i + 1 < T) {
                const uint32_t i1 = i + 1;
                const float l1 = (i1 < T / 2) ? (step * (float)i1) : fmaf(-step, (float)(T - 1 - i1), 1.0f);
                delta = (near + span * l1) - z;
            }
            const uint64_t s = (uint64_t)n * T + i;
            sigma = sigmas[s];
            c0 = rgbs[s * 3]; c1 = rgbs[s * 3 + 1]; c2 = rgbs[s * 3 + 2];
            alpha = 1 - __expf(-delta * sigma);
            oz = (z - near) / span;
            oz = oz < 0 ? 0 : (oz > 1 ? 1 : oz);
        }
        const float om = valid ? (1 - alpha + 1e-15f) : 1.0f;
        const float P = wave_incl_prod(om, (int)lane);
        float Pex = __shfl_up(P, 1, 64);
        if (lane == 0) Pex = 1.0f;
        const float w = alpha * (Tc * Pex);
        r = fmaf(w, c0, r); g = fmaf(w, c1, g); b = fmaf(w, c2, b);
        a = fmaf(w, sigma, a); d = fmaf(w, oz, d); ws += w;
        Tc *= __shfl(P, 63, 64);
    }
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); a = wave_sum(a); d = wave_sum(d); ws = wave_sum(ws);
    if (lane == 0) {
        const float rest = (1 - ws) * bg;
        float o[4] = {r + rest, g + rest, b + rest, a + rest};
#pragma unroll
        for (int k = 0; k < 4; k++) image4[(uint64_t)n * 4 + k] = fminf(1.0f, fmaxf(0.0f, o[k]));
        depth[n] = d;
    }
}

// ---------------------------------------------------------------- fused select + composite over K objects' packed fields
// fields.p[k] -> object k's field4 [N,T] (sigma, r, g, b) on the SAME rays and sample positions, k in checkpoint order. One wave per
// ray, 64 samples at a time: every lane reads its sample of each object with one 16-byte load, keeps the entry with the strictly larger
// density (object 0 first, `d > max` for the others: COMBINED.py:247-251, torch.maximum's NaN propagation included) and the composite of
// image_depth_generation (:141-200) runs on the merged values without the merged field ever being stored: 16 B read per sample and
// object, 20 B written per ray and background. n_bg backgrounds (the reference composites every view twice, white and black,
// compute_metrics_both_backgrounds): image4 [n_bg,N,4]. merged4 (optional) [N,T] receives (max density, best rgb) for callers that
// keep the reference's max_densities / max_rgbs.
#define FOC_COMBINE_MAX_OBJECTS 16
struct CombineFields { const float4 *p[FOC_COMBINE_MAX_OBJECTS]; };

__global__ void __launch_bounds__(256) k_combine_select_composite(CombineFields fields, uint32_t K, const float *__restrict__ nears,
                                                                  const float *__restrict__ fars, uint32_t N, uint32_t T, float bg0, float bg1,
                                                                  uint32_t n_bg, float *__restrict__ image4, float *__restrict__ depth,
                                                                  float4 *__restrict__ merged4) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float near = nears[n], far = fars[n];
    const float span = far - near;
    const float sample_dist = span / (float)T;
    const float step = 1.0f / (float)(T - 1);
    float Tc = 1.0f, r = 0, g = 0, b = 0, a = 0, d = 0, ws = 0;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < T;
        float alpha = 0, sigma = 0, c0 = 0, c1 = 0, c2 = 0, oz = 0;
        if (valid) {
            const uint64_t s = (uint64_t)n * T + i;
            float4 best = fields.p[0][s];
            for (uint32_t k = 1; k < K; k++) {
                const float4 f = fields.p[k][s];
                const float m = best.x;
                if (f.x > m) { best.y = f.y; best.z = f.z; best.w = f.w; }
                best.x = (f.x != f.x || m != m) ? __builtin_nanf("") : (f.x > m ? f.x : m);      // torch.maximum propagates NaN
            }
            if (merged4) merged4[s] = best;
            sigma = best.x; c0 = best.y; c1 = best.z; c2 = best.w;
            const float l0 = (i < T / 2) ? (step * (float)i) : fmaf(-step, (float)(T - 1 - i), 1.0f);   // device linspace: see fixedstep.hip
            const float z = near + span * l0;
            float delta = sample_dist;
            if (i + 1 < T) {
                const uint32_t i1 = i + 1;
                const float l1 = (i1 < T / 2) ? (step * (float)i1) : fmaf(-step, (float)(T - 1 - i1), 1.0f);
                delta = (near + span * l1) - z;
            }
            alpha = 1 - __expf(-delta * sigma);
            oz = (z - near) / span;
            oz = oz < 0 ? 0 : (oz > 1 ? 1 : oz);
        }
        const float om = valid ? (1 - alpha + 1e-15f) : 1.0f;
        const float P = wave_incl_prod(om, (int)lane);
        float Pex = __shfl_up(P, 1, 64);
        if (lane == 0) Pex = 1.0f;
        const float w = alpha * (Tc * Pex);
        r = fmaf(w, c0, r); g = fmaf(w, c1, g); b = fmaf(w, c2, b);
        a = fmaf(w, sigma, a); d = fmaf(w, oz, d); ws += w;
        Tc *= __shfl(P, 63, 64);
    }
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); a = wave_sum(a); d = wave_sum(d); ws = wave_sum(ws);
    if (lane == 0) {
        for (uint32_t q = 0; q < n_bg; q++) {
            const float rest = (1 - ws) * (q == 0 ? bg0 : bg1);
            const float o[4] = {r + rest, g + rest, b + rest, a + rest};
#pragma unroll
            for (int k = 0; k < 4; k++) image4[((uint64_t)q * N + n) * 4 + k] = fminf(1.0f, fmaxf(0.0f, o[k]));
        }
        depth[n] = d;
    }
}

// Pre-merge of the objects that share a rank before the exchange (K objects on fewer GPUs): acc4 <- select(acc4, f4), same rule. The
// select is associative over the object order, so merging a rank's consecutive objects first and the ranks afterwards equals the
// reference's one sequential pass.
__global__ void __launch_bounds__(256) k_combine_select4(const float4 *__restrict__ f4, float4 *__restrict__ acc4, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const float4 f = f4[i];
        float4 best = acc4[i];
        const float m = best.x;
        if (f.x > m) { best.y = f.y; best.z = f.z; best.w = f.w; }
        best.x = (f.x != f.x || m != m) ? __builtin_nanf("") : (f.x > m ? f.x : m);
        acc4[i] = best;
    }
}

// MONeRFNetwork's running select (nerf/multiobjectnetwork.py:66-82): torch.max over stack([new, best]) and take_along_dim of the rows that
// travel with the density (geo_feat [n,15] or colour [n,3]). torch.max returns the FIRST maximal index and treats NaN as maximal, and the
// new object is stacked first: the new object wins ties (the opposite of COMBINED.py's strict '>') and a NaN density of either side stays.
// One wave per 64 samples: the lanes decide their sample, the ballot is the row mask, and the wave copies the 64 x width elements of the
// taken rows as one contiguous run (rows are 30 or 6 bytes in fp16: no lane-per-row accesses).
template <typename T>
__global__ void __launch_bounds__(256) k_mo_select(const T *__restrict__ sig_new, const T *__restrict__ feat_new, T *__restrict__ sig_best,
                                                   T *__restrict__ feat_best, uint64_t n, uint32_t width) {
    const uint64_t s0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u;          // wave-uniform
    if (s0 >= n) return;
    const uint32_t lane = threadIdx.x & 63u;
    bool take = false;
    if (s0 + lane < n) {
        const T a_raw = sig_new[s0 + lane];
        const float a = (float)a_raw, b = (float)sig_best[s0 + lane];
        take = (a != a) || (b == b && a >= b);
        if (take) sig_best[s0 + lane] = a_raw;
    }
    const uint64_t rows = __ballot(take);
    const uint32_t live = (uint32_t)(n - s0 < 64u ? n - s0 : 64u);
    const uint64_t base = s0 * width;
    for (uint32_t e = lane; e < live * width; e += 64u)
        if ((rows >> (e / width)) & 1ull) feat_best[base + e] = feat_new[base + e];
}

// ---------------------------------------------------------------- options (common.h FocOpt)
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
// values are atomics and the table is filled from the environment exactly once (std::call_once): a second thread entering foc_opt() while the
// first one initialises waits for it instead of reading defaults; foc_set_option from one thread is seen by launches issued afterwards
struct FocOptionRow { const char *name; std::atomic<int> value; };
static FocOptionRow foc_option_table[FOC_OPT_COUNT] = {
    {"FOC_MLP_BWD_FUSED", 1}, {"FOC_FIELD_FWD_FUSED", 1}, {"FOC_GB_MERGE_MAX_RES", 480}, {"FOC_GB_FACTORED", 1}, {"FOC_GB_TAIL_SPLIT", 16}, {"FOC_GRID_FUSE_SMALL", 1},
    {"FOC_GRID_PAIRS", 1}, {"FOC_GRID_FAST", 1}, {"FOC_MARCH_SERIAL", -1}, {"FOC_MARCH_RAYS_ROW_MAX", 131072}, {"FOC_OCC_MARCH_FORM", -1},
    {"FOC_OCC_SAMPLE_MAJOR", 1}, {"FOC_OCC_FIELD_PIECE", 1 << 23},
};
static int foc_option_parse(int which, const char *text) {
    if (which == FOC_OPT_OCC_MARCH_FORM) {                  // the forms have names: two | row | lane | staged (or 0..3, -1 = by burst length)
        if (text[0] == 't') return 0;
        if (text[0] == 'r') return 1;
        if (text[0] == 'l') return 2;
        if (text[0] == 's') return 3;
    }
    return atoi(text);
}
static void foc_options_init() {
    static std::once_flag once;
    std::call_once(once, [] {
        for (int i = 0; i < FOC_OPT_COUNT; i++) {           // the ONE place the library reads its environment
            const char *e = getenv(foc_option_table[i].name);
            if (e && e[0]) foc_option_table[i].value.store(foc_option_parse(i, e), std::memory_order_relaxed);
        }
    });
}
int foc_opt(FocOpt which) { foc_options_init(); return foc_option_table[which].value.load(std::memory_order_relaxed); }

extern "C" {

int foc_set_option(const char *name, int value) {
    foc_options_init();
    for (int i = 0; i < FOC_OPT_COUNT; i++)
        if (name && strcmp(name, foc_option_table[i].name) == 0) { foc_option_table[i].value.store(value, std::memory_order_relaxed); return FOC_OK; }
    foc_set_error("set_option: unknown option '%s'", name ? name : "(null)");
    return FOC_E_INVALID;
}

int foc_get_option(const char *name, int *value) {
    foc_options_init();
    for (int i = 0; i < FOC_OPT_COUNT; i++)
        if (name && value && strcmp(name, foc_option_table[i].name) == 0) { *value = foc_option_table[i].value.load(std::memory_order_relaxed); return FOC_OK; }
    foc_set_error("get_option: unknown option '%s'", name ? name : "(null)");
    return FOC_E_INVALID;
}

int foc_guard_pick_device(int stream_is_null, int stream_device, int pointer_device, int current_device) {
    return foc_guard_pick(stream_is_null != 0, stream_device, pointer_device, current_device);
}


int foc_combine_select_composite(const float *const *fields4, uint32_t K, const float *nears, const float *fars, uint32_t N, uint32_t T,
                                 const float *bgs, uint32_t n_bg, float *image4, float *depth, float *merged4, void *stream) {
    FocDeviceGuard foc_guard_(stream, nears);
    FOC_REQUIRE(K >= 1 && K <= FOC_COMBINE_MAX_OBJECTS, FOC_E_INVALID, "combine_select_composite: 1 <= K <= %d objects per call (got %u)",
                FOC_COMBINE_MAX_OBJECTS, K);
    FOC_REQUIRE(n_bg >= 1 && n_bg <= 2 && bgs, FOC_E_INVALID, "combine_select_composite: one or two backgrounds");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "combine_select_composite: T must be >= 2");
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(fields4 && nears && fars && image4 && depth, FOC_E_INVALID, "combine_select_composite: null pointer");
    CombineFields f;
    for (uint32_t k = 0; k < FOC_COMBINE_MAX_OBJECTS; k++) {
        f.p[k] = reinterpret_cast<const float4 *>(k < K ? fields4[k] : fields4[0]);
        FOC_REQUIRE(f.p[k] && ((uintptr_t)f.p[k] & 15) == 0, FOC_E_INVALID, "combine_select_composite: field %u is null or not 16-byte aligned", k);
    }
    FOC_REQUIRE(((uintptr_t)merged4 & 15) == 0, FOC_E_INVALID, "combine_select_composite: merged4 must be 16-byte aligned");
    hipLaunchKernelGGL(k_combine_select_composite, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, f, K, nears, fars, N, T, bgs[0],
                       n_bg > 1 ? bgs[1] : 0.0f, n_bg, image4, depth, reinterpret_cast<float4 *>(merged4));
    FOC_CHECK_LAUNCH("combine_select_composite");
    return FOC_OK;
}

int foc_combine_select4(const float *field4, float *acc4, uint64_t n, void *stream) {
    FocDeviceGuard foc_guard_(stream, field4);
    FOC_REQUIRE(n == 0 || (field4 && acc4), FOC_E_INVALID, "combine_select4: null pointer");
    FOC_REQUIRE((((uintptr_t)field4 | (uintptr_t)acc4) & 15) == 0, FOC_E_INVALID, "combine_select4: fields must be 16-byte aligned");
    if (n == 0) return FOC_OK;
    hipLaunchKernelGGL(k_combine_select4, dim3(foc_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4 *>(field4),
                       reinterpret_cast<float4 *>(acc4), n);
    FOC_CHECK_LAUNCH("combine_select4");
    return FOC_OK;
}

int foc_mo_select(const void *sigma_new, const void *feat_new, void *sigma_best, void *feat_best, uint64_t n, uint32_t feat_width,
                  uint32_t elem_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, sigma_new);
    FOC_REQUIRE(elem_bytes == 2 || elem_bytes == 4, FOC_E_INVALID, "mo_select: elem_bytes must be 2 (half) or 4 (float), got %u", elem_bytes);
    FOC_REQUIRE(feat_width >= 1 && feat_width <= 64, FOC_E_INVALID, "mo_select: feat_width %u outside 1..64", feat_width);
    FOC_REQUIRE(n == 0 || (sigma_new && feat_new && sigma_best && feat_best), FOC_E_INVALID, "mo_select: null pointer");
    FOC_REQUIRE(n < (1ull << 38), FOC_E_INVALID, "mo_select: n too large");
    if (n == 0) return FOC_OK;
    const dim3 grid((uint32_t)((n + 255u) / 256u));
    if (elem_bytes == 2)
        hipLaunchKernelGGL(k_mo_select<__half>, grid, dim3(256), 0, (hipStream_t)stream, (const __half *)sigma_new, (const __half *)feat_new,
                           (__half *)sigma_best, (__half *)feat_best, n, feat_width);
    else
        hipLaunchKernelGGL(k_mo_select<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)sigma_new, (const float *)feat_new,
                           (float *)sigma_best, (float *)feat_best, n, feat_width);
    FOC_CHECK_LAUNCH("mo_select");
    return FOC_OK;
}

int foc_abi_version(void) { return FOC_ABI_VERSION; }
const char *foc_last_error(void) { return g_err; }
const char *foc_arch(void) { return "gfx950"; }

int foc_combine_select(const float *dens, const float *rgb, float *max_dens, float *best_rgb, uint64_t n, void *stream) {
    FocDeviceGuard foc_guard_(stream, dens);
    FOC_REQUIRE(n == 0 || (dens && rgb && max_dens && best_rgb), FOC_E_INVALID, "combine_select: null pointer");
    if (n == 0) return FOC_OK;
    hipLaunchKernelGGL(k_combine_select, dim3(foc_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, dens, rgb, max_dens, best_rgb, n);
    FOC_CHECK_LAUNCH("combine_select");
    return FOC_OK;
}

int foc_combine_pack_keys(const float *dens, uint32_t rank, uint64_t *keys, uint64_t n, void *stream) {
    FocDeviceGuard foc_guard_(stream, dens);
    FOC_REQUIRE(n == 0 || (dens && keys), FOC_E_INVALID, "combine_pack_keys: null pointer");
    if (n == 0) return FOC_OK;
    hipLaunchKernelGGL(k_combine_pack, dim3(foc_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, dens, rank, keys, n);
    FOC_CHECK_LAUNCH("combine_pack_keys");
    return FOC_OK;
}

int foc_combine_unpack(const uint64_t *keys, uint32_t rank, const float *rgb, float *max_dens, float *masked_rgb, uint64_t n, void *stream) {
    FocDeviceGuard foc_guard_(stream, keys);
    FOC_REQUIRE(n == 0 || (keys && rgb && max_dens && masked_rgb), FOC_E_INVALID, "combine_unpack: null pointer");
    if (n == 0) return FOC_OK;
    hipLaunchKernelGGL(k_combine_unpack, dim3(foc_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, keys, rank, rgb, max_dens, masked_rgb, n);
    FOC_CHECK_LAUNCH("combine_unpack");
    return FOC_OK;
}

int foc_composite_fixed_steps(const float *sigmas, const float *rgbs, const float *nears, const float *fars, uint32_t N, uint32_t T,
                              float bg, float *image4, float *depth, void *stream) {
    FocDeviceGuard foc_guard_(stream, sigmas);
    FOC_REQUIRE(N == 0 || (sigmas && rgbs && nears && fars && image4 && depth), FOC_E_INVALID, "composite_fixed_steps: null pointer");
    FOC_REQUIRE(T >= 2, FOC_E_INVALID, "composite_fixed_steps: T must be >= 2");
    if (N == 0) return FOC_OK;
    hipLaunchKernelGGL(k_composite_fixed, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, sigmas, rgbs, nears, fars, N, T, bg, image4, depth);
    FOC_CHECK_LAUNCH("composite_fixed_steps");
    return FOC_OK;
}

} // extern "C"
