// densitygrid.hip — occupancy-grid maintenance on the device.
//
// Reference: NeRFRenderer.mark_untrained_grid (nerf/renderer.py:356-418, legacy/nerf/renderer.py:380-443) and
// NeRFRenderer.update_extra_state (nerf/renderer.py:420-508, legacy :445-536). The reference runs them from Python: a five-level
// loop over 64^3 blocks x cascades x pose batches (hundreds of torch kernels) for the former; torch.nonzero / randint / index_put /
// masked max / mean().item() / packbits with two host synchronisations per cascade for the latter, every 16 training steps.
// Here the same arithmetic is a handful of launches with no host synchronisation:
//   foc_mark_untrained_grid        one thread per (cascade, cell), loop over the cameras
//   foc_grid_cells_xyz             query points of the full sweep (first 16 updates), cells enumerated in Morton order
//   foc_grid_update_sample         query points of the steady-state update: N uniformly random cells + N cells drawn uniformly from
//                                  the occupied ones (rank-select over 64-cell occupancy words instead of torch.nonzero + gather)
//   foc_grid_update_apply          scatter (max on duplicates) -> EMA max -> mean -> packbits with threshold min(mean, density_thresh)
// The network evaluation between sample and apply stays with the caller. Randomness comes in as tensors (like `noises` of
// march_rays_train), so the CPU oracle can replay a call exactly. These entry points have no reference binding.
#include "common.h"

__device__ __forceinline__ uint32_t dg_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t dg_morton3D(uint32_t x, uint32_t y, uint32_t z) { return dg_expand_bits(x) | (dg_expand_bits(y) << 1) | (dg_expand_bits(z) << 2); }
__device__ __forceinline__ uint32_t dg_compact_bits(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

struct DgCascades { float scale[8]; float half[8]; };       // per cascade: float(bound_c - bound_c/H), float(bound_c/H), bound_c = min(2^c, bound)

static void dg_make_cascades(uint32_t C, uint32_t H, float bound, DgCascades &cs) {
    for (uint32_t c = 0; c < 8; c++) {
        // Python: bound = min(2 ** cas, self.bound); half_grid_size = bound / self.grid_size  (float64), then used as fp32 scalars
        const double b = (double)(1u << c) < (double)bound ? (double)(1u << c) : (double)bound;
        const double half = b / (double)H;
        cs.scale[c] = c < C ? (float)(b - half) : 0.0f;
        cs.half[c] = c < C ? (float)half : 0.0f;
    }
}

// world coordinate of a cell: 2 * coords.float() / (H - 1) - 1   (renderer.py:385 / :437), then * (bound - half_grid_size)
__device__ __forceinline__ float dg_world(uint32_t c, float Hm1) { return (2.0f * (float)c) / Hm1 - 1.0f; }

// ---------------------------------------------------------------- mark_untrained_grid
__global__ void __launch_bounds__(256) k_dg_mark_untrained(const float *__restrict__ poses, uint32_t B, float kx, float ky, DgCascades cs, uint32_t C,
                                                           uint32_t H, float *__restrict__ density_grid, int32_t *__restrict__ count_out) {
    const uint32_t H3 = H * H * H;
    const float Hm1 = (float)(H - 1);
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < C * H3; g += gridDim.x * 256) {
        const uint32_t cas = g / H3, m = g - cas * H3;
        const float wx = dg_world(dg_compact_bits(m), Hm1) * cs.scale[cas], wy = dg_world(dg_compact_bits(m >> 1), Hm1) * cs.scale[cas],
                    wz = dg_world(dg_compact_bits(m >> 2), Hm1) * cs.scale[cas];
        const float h2 = cs.half[cas] * 2.0f;                  // `half_grid_size * 2` is exact in either precision
        int32_t count = 0;
        for (uint32_t b = 0; b < B; b++) {
            const float *P = poses + (uint64_t)b * 16;           // c2w, row-major 4x4
            const float dx = wx - P[3], dy = wy - P[7], dz = wz - P[11];
            // cam = d @ R (renderer.py:403): cam_j = sum_i d_i R[i][j]
            const float cx_ = fmaf(dz, P[8], fmaf(dy, P[4], dx * P[0]));
            const float cy_ = fmaf(dz, P[9], fmaf(dy, P[5], dx * P[1]));
            const float cz_ = fmaf(dz, P[10], fmaf(dy, P[6], dx * P[2]));
            const bool mz = cz_ > 0.0f;
            const bool mx = fabsf(cx_) < kx * cz_ + h2;
            const bool my = fabsf(cy_) < ky * cz_ + h2;
            count += (mz && mx && my) ? 1 : 0;
        }
        if (count_out) count_out[g] = count;
        if (count == 0) density_grid[g] = -1.0f;
    }
}

// ---------------------------------------------------------------- query points
__device__ __forceinline__ void dg_store_xyz(float *__restrict__ xyzs, uint64_t s, uint32_t m, float Hm1, float scale, float half,
                                             const float *__restrict__ jitter) {
    float x = dg_world(dg_compact_bits(m), Hm1) * scale, y = dg_world(dg_compact_bits(m >> 1), Hm1) * scale, z = dg_world(dg_compact_bits(m >> 2), Hm1) * scale;
    if (jitter) {   // cas_xyzs += (torch.rand_like(cas_xyzs) * 2 - 1) * half_grid_size
        x += (jitter[s * 3] * 2.0f - 1.0f) * half; y += (jitter[s * 3 + 1] * 2.0f - 1.0f) * half; z += (jitter[s * 3 + 2] * 2.0f - 1.0f) * half;
    }
    xyzs[s * 3] = x; xyzs[s * 3 + 1] = y; xyzs[s * 3 + 2] = z;
}

__global__ void __launch_bounds__(256) k_dg_cells_xyz(DgCascades cs, uint32_t C, uint32_t H, const float *__restrict__ jitter, float *__restrict__ xyzs) {
    const uint32_t H3 = H * H * H;
    const float Hm1 = (float)(H - 1);
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < C * H3; g += gridDim.x * 256) {
        const uint32_t cas = g / H3, m = g - cas * H3;
        dg_store_xyz(xyzs, g, m, Hm1, cs.scale[cas], cs.half[cas], jitter);
    }
}

// occupancy words: bit i of word w = density_grid[cas][64 w + i] > 0   (renderer.py:482 `torch.nonzero(self.density_grid[cas] > 0)`)
__global__ void __launch_bounds__(256) k_dg_occ_words(const float *__restrict__ density_grid, uint32_t total_cells, unsigned long long *__restrict__ words,
                                                      uint32_t *__restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t base = (blockIdx.x * 256 + threadIdx.x) - lane; base < total_cells; base += gridDim.x * 256) {
        const uint32_t i = base + lane;
        const bool occ = i < total_cells && density_grid[i] > 0.0f;
        const unsigned long long w = __ballot(occ);
        if (lane == 0) { words[base >> 6] = w; counts[base >> 6] = (uint32_t)__popcll(w); }
    }
}

// one workgroup per cascade: exclusive scan of the nb word counts -> prefix[0..nb]
__global__ void __launch_bounds__(1024) k_dg_occ_scan(const uint32_t *__restrict__ counts, uint32_t nb, uint32_t *__restrict__ prefix) {
    __shared__ uint32_t s_wave[16];
    const uint32_t cas = blockIdx.x;
    const uint32_t *cnt = counts + (uint64_t)cas * nb;
    uint32_t *pre = prefix + (uint64_t)cas * (nb + 1);
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = min(nb, lo + per);
    uint32_t local = 0;
    for (uint32_t i = lo; i < hi; i++) local += cnt[i];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = (uint32_t)wave_incl_sum_i((int)local, (int)lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t k = 0; k < wave; k++) wbase += s_wave[k];
    uint32_t run = wbase + (incl - local);
    for (uint32_t i = lo; i < hi; i++) { pre[i] = run; run += cnt[i]; }
    if (threadIdx.x == 1023) pre[nb] = run;          // the last thread's range ends at nb (or is empty and carries the total)
}

// position of the r-th (0-based) set bit of w
__device__ __forceinline__ uint32_t dg_select_bit(unsigned long long w, uint32_t r) {
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t width = 32; width >= 1; width >>= 1) {
        const uint32_t c = (uint32_t)__popcll(w & ((1ull << width) - 1ull));
        if (r >= c) { r -= c; w >>= width; pos += width; }
    }
    return pos;
}

__global__ void __launch_bounds__(256) k_dg_sample(DgCascades cs, uint32_t C, uint32_t H, uint32_t N, const int32_t *__restrict__ rand_coords,
                                                   const float *__restrict__ rand_pick, const float *__restrict__ jitter,
                                                   const unsigned long long *__restrict__ words, const uint32_t *__restrict__ prefix, uint32_t nb,
                                                   int32_t *__restrict__ indices, float *__restrict__ xyzs) {
    const float Hm1 = (float)(H - 1);
    const uint32_t per_cas = 2 * N;
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < C * per_cas; g += gridDim.x * 256) {
        const uint32_t cas = g / per_cas, i = g - cas * per_cas;
        uint32_t m;
        const uint32_t j = i < N ? i : i - N;
        const int32_t *rc = rand_coords + ((uint64_t)cas * N + j) * 3;
        const uint32_t m_rand = dg_morton3D((uint32_t)rc[0], (uint32_t)rc[1], (uint32_t)rc[2]);
        if (i < N) {
            m = m_rand;                                         // coords = torch.randint(0, H, (N,3)); indices = morton3D(coords)  (:479-480)
        } else {
            const uint32_t *pre = prefix + (uint64_t)cas * (nb + 1);
            const uint32_t total = pre[nb];
            if (total == 0) {
                m = m_rand;                                     // no occupied cell (torch.randint(0, 0) raises in the reference): repeat the random cell
            } else {
                // rand_mask = torch.randint(0, n_occ, [N]) drawn here as floor(u * n_occ) from u in [0,1): no host round trip for n_occ
                uint32_t k = (uint32_t)(rand_pick[(uint64_t)cas * N + j] * (float)total);
                if (k >= total) k = total - 1;
                uint32_t lo = 0, hi = nb;                       // largest word b with pre[b] <= k
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= k) lo = mid; else hi = mid; }
                m = lo * 64 + dg_select_bit(words[(uint64_t)cas * nb + lo], k - pre[lo]);
            }
        }
        indices[g] = (int32_t)m;
        dg_store_xyz(xyzs, g, m, Hm1, cs.scale[cas], cs.half[cas], jitter);
    }
}

// ---------------------------------------------------------------- apply
__global__ void __launch_bounds__(256) k_dg_fill(float *__restrict__ tmp, uint32_t n, double *__restrict__ sum) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) tmp[i] = -1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *sum = 0.0;
}

// tmp_grid[cas, indices] = sigmas * density_scale; duplicates: the reference's index_put keeps an arbitrary one, here the largest
__global__ void __launch_bounds__(256) k_dg_scatter(const float *__restrict__ sigmas, const int32_t *__restrict__ indices, uint32_t C, uint32_t Mc,
                                                    uint32_t H3, float density_scale, float *__restrict__ tmp) {
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < C * Mc; g += gridDim.x * 256) {
        const uint32_t cas = g / Mc;
        const uint32_t cell = indices ? (uint32_t)indices[g] : g - cas * Mc;
        if (cell >= H3) continue;
        const float v = sigmas[g] * density_scale;
        // non-negative floats order like their bit patterns as signed ints, and the -1.0f fill is negative as an int
        atomicMax(reinterpret_cast<int *>(tmp) + (uint64_t)cas * H3 + cell, __float_as_int(v));
    }
}

// valid = (grid >= 0) & (tmp >= 0); grid[valid] = max(grid*decay, tmp); sum of clamp(grid, min=0)   (:494-497)
__global__ void __launch_bounds__(256) k_dg_ema(float *__restrict__ grid, const float *__restrict__ tmp, uint32_t n, float decay, double *__restrict__ sum) {
    __shared__ double s_part[4];
    double local = 0.0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float g = grid[i];
        const float t = tmp[i];
        if (g >= 0.0f && t >= 0.0f) { const float d = g * decay; g = d > t ? d : t; grid[i] = g; }
        local += (double)(g > 0.0f ? g : 0.0f);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// mean = sum / n (fp32 result like torch.mean); bitfield = packbits(grid, min(mean, density_thresh))   (:497-503)
__global__ void __launch_bounds__(256) k_dg_pack(const float *__restrict__ grid, uint32_t n_bytes, const double *__restrict__ sum, double inv_n,
                                                 float density_thresh, uint8_t *__restrict__ bitfield, float *__restrict__ mean_out) {
    const float mean = (float)(*sum * inv_n);
    const float thresh = mean < density_thresh ? mean : density_thresh;
    if (blockIdx.x == 0 && threadIdx.x == 0 && mean_out) *mean_out = mean;
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < n_bytes; b += gridDim.x * 256) {
        const float4 v0 = *reinterpret_cast<const float4 *>(grid + (uint64_t)b * 8), v1 = *reinterpret_cast<const float4 *>(grid + (uint64_t)b * 8 + 4);
        uint32_t bits = 0;
        bits |= (v0.x > thresh ? 1u : 0u) << 0; bits |= (v0.y > thresh ? 1u : 0u) << 1; bits |= (v0.z > thresh ? 1u : 0u) << 2; bits |= (v0.w > thresh ? 1u : 0u) << 3;
        bits |= (v1.x > thresh ? 1u : 0u) << 4; bits |= (v1.y > thresh ? 1u : 0u) << 5; bits |= (v1.z > thresh ? 1u : 0u) << 6; bits |= (v1.w > thresh ? 1u : 0u) << 7;
        bitfield[b] = (uint8_t)bits;
    }
}

// ================================================================= host entry points
static int dg_check(const char *who, uint32_t C, uint32_t H) {
    FOC_REQUIRE(C >= 1 && C <= 8, FOC_E_INVALID, "%s: cascade must be in [1,8] (got %u)", who, C);
    FOC_REQUIRE(H >= 8 && H <= 1024 && (H & (H - 1)) == 0 && (H * H * H) % 64 == 0, FOC_E_INVALID, "%s: grid size must be a power of two in [8,1024] (got %u)", who, H);
    return FOC_OK;
}

extern "C" {

int foc_mark_untrained_grid(const float *poses, uint32_t B, float fx, float fy, float cx, float cy, float bound, uint32_t C, uint32_t H,
                            float *density_grid, int32_t *count, void *stream) {
    FocDeviceGuard foc_guard_(stream, poses);
    int rc = dg_check("mark_untrained_grid", C, H);
    if (rc) return rc;
    FOC_REQUIRE(density_grid && (poses || B == 0), FOC_E_INVALID, "mark_untrained_grid: null pointer");
    DgCascades cs;
    dg_make_cascades(C, H, bound, cs);
    // `cx / fx` and `cy / fy` are Python floats (float64) that multiply an fp32 tensor: rounded to fp32 once
    const float kx = (float)((double)cx / (double)fx), ky = (float)((double)cy / (double)fy);
    hipLaunchKernelGGL(k_dg_mark_untrained, dim3(foc_grid_1d((uint64_t)C * H * H * H, 256)), dim3(256), 0, (hipStream_t)stream, poses, B, kx, ky, cs, C, H,
                       density_grid, count);
    FOC_CHECK_LAUNCH("mark_untrained_grid");
    return FOC_OK;
}

int foc_grid_cells_xyz(uint32_t C, uint32_t H, float bound, const float *jitter, float *xyzs, void *stream) {
    FocDeviceGuard foc_guard_(stream, xyzs);
    int rc = dg_check("grid_cells_xyz", C, H);
    if (rc) return rc;
    FOC_REQUIRE(xyzs, FOC_E_INVALID, "grid_cells_xyz: null pointer");
    DgCascades cs;
    dg_make_cascades(C, H, bound, cs);
    hipLaunchKernelGGL(k_dg_cells_xyz, dim3(foc_grid_1d((uint64_t)C * H * H * H, 256)), dim3(256), 0, (hipStream_t)stream, cs, C, H, jitter, xyzs);
    FOC_CHECK_LAUNCH("grid_cells_xyz");
    return FOC_OK;
}

uint64_t foc_grid_update_sample_workspace_bytes(uint32_t C, uint32_t H) {
    const uint64_t nb = (uint64_t)H * H * H / 64;
    return C * nb * 8 + C * nb * 4 + C * (nb + 1) * 4 + 256;
}

int foc_grid_update_sample(const float *density_grid, uint32_t C, uint32_t H, float bound, uint32_t N, const int32_t *rand_coords, const float *rand_pick,
                           const float *jitter, int32_t *indices, float *xyzs, void *workspace, uint64_t workspace_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, density_grid);
    int rc = dg_check("grid_update_sample", C, H);
    if (rc) return rc;
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(density_grid && rand_coords && rand_pick && indices && xyzs && workspace, FOC_E_INVALID, "grid_update_sample: null pointer");
    FOC_REQUIRE(workspace_bytes >= foc_grid_update_sample_workspace_bytes(C, H), FOC_E_INVALID, "grid_update_sample: workspace too small");
    FOC_REQUIRE((uint64_t)C * 2 * N < (1ull << 31), FOC_E_INVALID, "grid_update_sample: too many samples");
    const uint32_t H3 = H * H * H, nb = H3 / 64;
    unsigned long long *words = reinterpret_cast<unsigned long long *>(workspace);
    uint32_t *counts = reinterpret_cast<uint32_t *>(words + (uint64_t)C * nb);
    uint32_t *prefix = counts + (uint64_t)C * nb;
    DgCascades cs;
    dg_make_cascades(C, H, bound, cs);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dg_occ_words, dim3(foc_grid_1d((uint64_t)C * H3, 256)), dim3(256), 0, st, density_grid, C * H3, words, counts);
    FOC_CHECK_LAUNCH("grid_update_sample(words)");
    hipLaunchKernelGGL(k_dg_occ_scan, dim3(C), dim3(1024), 0, st, counts, nb, prefix);
    FOC_CHECK_LAUNCH("grid_update_sample(scan)");
    hipLaunchKernelGGL(k_dg_sample, dim3(foc_grid_1d((uint64_t)C * 2 * N, 256)), dim3(256), 0, st, cs, C, H, N, rand_coords, rand_pick, jitter, words, prefix, nb,
                       indices, xyzs);
    FOC_CHECK_LAUNCH("grid_update_sample");
    return FOC_OK;
}

uint64_t foc_grid_update_apply_workspace_bytes(uint32_t C, uint32_t H) { return (uint64_t)C * H * H * H * 4 + 256; }

int foc_grid_update_apply(float *density_grid, uint32_t C, uint32_t H, const float *sigmas, const int32_t *indices, uint32_t Mc, float density_scale,
                          float decay, float density_thresh, uint8_t *bitfield, float *mean_out, void *workspace, uint64_t workspace_bytes, void *stream) {
    FocDeviceGuard foc_guard_(stream, density_grid);
    int rc = dg_check("grid_update_apply", C, H);
    if (rc) return rc;
    FOC_REQUIRE(density_grid && sigmas && bitfield && workspace, FOC_E_INVALID, "grid_update_apply: null pointer");
    FOC_REQUIRE(workspace_bytes >= foc_grid_update_apply_workspace_bytes(C, H), FOC_E_INVALID, "grid_update_apply: workspace too small");
    const uint32_t H3 = H * H * H, n = C * H3;
    FOC_REQUIRE(indices || Mc == H3, FOC_E_INVALID, "grid_update_apply: without indices, sigmas must cover every cell (Mc == H^3)");
    FOC_REQUIRE((((uintptr_t)density_grid) & 15u) == 0, FOC_E_INVALID, "grid_update_apply: density_grid must be 16-byte aligned");
    double *sum = reinterpret_cast<double *>(workspace);
    float *tmp = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dg_fill, dim3(foc_grid_1d(n, 256)), dim3(256), 0, st, tmp, n, sum);
    FOC_CHECK_LAUNCH("grid_update_apply(fill)");
    if (Mc) {
        hipLaunchKernelGGL(k_dg_scatter, dim3(foc_grid_1d((uint64_t)C * Mc, 256)), dim3(256), 0, st, sigmas, indices, C, Mc, H3, density_scale, tmp);
        FOC_CHECK_LAUNCH("grid_update_apply(scatter)");
    }
    hipLaunchKernelGGL(k_dg_ema, dim3(foc_grid_1d(n, 256)), dim3(256), 0, st, density_grid, tmp, n, decay, sum);
    FOC_CHECK_LAUNCH("grid_update_apply(ema)");
    hipLaunchKernelGGL(k_dg_pack, dim3(foc_grid_1d(n / 8, 256)), dim3(256), 0, st, density_grid, n / 8, sum, 1.0 / (double)n, density_thresh, bitfield, mean_out);
    FOC_CHECK_LAUNCH("grid_update_apply(pack)");
    return FOC_OK;
}

} // extern "C"
