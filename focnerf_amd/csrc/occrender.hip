// occrender.hip — one iteration of the occupancy-grid inference loop as ONE native call.
//
// The caller of the inference ops (legacy/nerf/renderer.py:323-372) runs, per iteration and from Python: march_rays -> the network on the
// new samples -> composite_rays -> compaction of the list of live rays. An 800 x 800 view is ~150 such iterations of ~0.17 ms of kernels
// each, and ~20 Python-level operations per iteration cost more than that to enqueue. foc_occ_render_step enqueues the whole iteration —
// nine launches — from C, on buffers the caller allocated once per view:
//
//   prepare   fill the output list with -1, reset the walkers' worklist and the compaction's block counts; zero the iteration's sample slots
//             (the reference's torch.zeros, raymarching.py:334-336) where the march kernel of this burst length does not write them all itself
//   march     foc_march_rays_two_phase (raymarching.hip). One sample per ray: first visits per lane, then the walkers densely packed. A burst
//             of several samples (focnerf_amd/renderer.py marches 8 per ray and iteration): one ray per lane, the wave's samples staged in
//             LDS and written as whole runs, SAMPLE-MAJOR ([n_step][n_alive]: the 64 rows of an encoder / network wave are 64 neighbouring
//             rays at one burst slot), t re-derived after every sample so that the samples are those of one-sample iterations. The
//             positions leave already normalised, x -> (x + bound) * (1 / (2 bound))  (gridencoder/grid.py:149 as torch evaluates it:
//             division by a scalar = multiplication by its reciprocal) — nobody but the encoder reads them here
//   encode    foc_grid_encode_forward ([L, M, 2] planes)
//   field     foc_nerf_field_inference (sigma net -> head -> colour net, per-sample directions)
//   composite foc_composite_compact: composite_rays in place (weights_sum, depth, image, rays_t; finished rays marked -1; a burst's samples
//             loaded at once), its waves counting their survivors and recording where rays died, then scan + ordered scatter into the output
//             list, whose tail stays -1 (march / composite skip such entries)
//
// The live count stays on the device (`count`, one int); the caller reads it late (focnerf_amd/renderer.py) and passes an upper bound.
#include "common.h"
#include <cstdlib>

__global__ void __launch_bounds__(256) k_occ_prepare(uint32_t *__restrict__ samples, uint64_t n_sample_words, int32_t *__restrict__ list_out, uint32_t n_list,
                                                     int32_t *__restrict__ wl_count, int32_t *__restrict__ block_counts, uint32_t n_blocks) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_sample_words; i += stride) samples[i] = 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_list; i += stride) list_out[i] = -1;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_blocks; i += stride) block_counts[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *wl_count = 0;
}

extern "C" {


uint64_t foc_occ_render_step_scratch_bytes(uint32_t n_rays) {
    // int32[n + 4] worklist | compaction block counts int32[n / 1024 + 2], both 256-byte aligned
    const uint64_t wl = (((uint64_t)n_rays + 4) * 4 + 255) & ~(uint64_t)255;
    const uint64_t cc = (((uint64_t)n_rays / 1024 + 2) * 4 + 255) & ~(uint64_t)255;
    return wl + cc;
}

int foc_occ_render_step(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, int32_t *rays_alive_out, int32_t *count,
                        float *rays_t, const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars, const float *noises,
                        float *samples /* [n_alive*n_step*8]: normalised xyzs | dirs | deltas */,
                        void *planes /* fp16 [L, n_alive*n_step, 2] */, float *sigma, float *rgb,
                        const void *embeddings, const int32_t *offsets, const int32_t *offsets_host, uint32_t L, float S, uint32_t base_res,
                        const void *sigma_weights, uint32_t sigma_layers, const void *color_weights, uint32_t color_layers, uint32_t activation,
                        const void *obj_feat, float T_thresh, float *weights_sum, float *depth, float *image, void *scratch, uint32_t flags, int32_t *deaths, uint32_t deaths_base,
                        uint32_t deaths_len, void *stream) {
    FocDeviceGuard foc_guard_(stream, rays_alive);
    FOC_REQUIRE(count, FOC_E_INVALID, "occ_render_step: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n_alive == 0) return foc_zero_async(count, sizeof(int32_t), st) == hipSuccess ? FOC_OK : FOC_E_LAUNCH;
    FOC_REQUIRE(rays_alive && rays_alive_out && rays_t && rays_o && rays_d && grid && fars && noises && samples && planes && sigma && rgb && embeddings &&
                offsets && sigma_weights && color_weights && weights_sum && depth && image && scratch, FOC_E_INVALID, "occ_render_step: null pointer");
    FOC_REQUIRE(n_step >= 1 && n_step <= 16, FOC_E_INVALID, "occ_render_step: n_step must be in [1, 16] (got %u)", n_step);
    const uint64_t M = (uint64_t)n_alive * n_step;
    FOC_REQUIRE(M < (1ull << 31), FOC_E_INVALID, "occ_render_step: n_alive * n_step must fit 31 bits");
    float *xyzs = samples, *dirs = samples + 3 * M, *deltas = samples + 6 * M;
    int32_t *worklist = reinterpret_cast<int32_t *>(scratch);
    int32_t *compact_scratch = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(scratch) + ((((uint64_t)n_alive + 4) * 4 + 255) & ~(uint64_t)255));
    // the sample slots are zeroed here (the reference's torch.zeros) unless the march kernel of this burst length writes every one of them itself
    int march_flags = 1 | ((flags & 1u) ? 2 : 0);          // normalised positions; flags bit 0: t re-derived after every sample (focnerf.h)
    // sample-major sample arrays ([n_step][n_alive]) where the march kernel of this call can write them (FOC_OCC_SAMPLE_MAJOR=0: ray-major)
    const int sample_major = (n_step > 1 && foc_opt(FOC_OPT_OCC_SAMPLE_MAJOR) != 0 && foc_march_rays_two_phase_sample_major(n_alive, n_step, march_flags)) ? 1 : 0;
    if (sample_major) march_flags |= 4;
    const uint64_t to_zero = foc_march_rays_two_phase_fills(n_step, march_flags) ? 0 : M * 8;
    hipLaunchKernelGGL(k_occ_prepare, dim3(foc_grid_1d((to_zero > n_alive ? to_zero : n_alive) + 1, 256)), dim3(256), 0, st, reinterpret_cast<uint32_t *>(samples), to_zero,
                       rays_alive_out, n_alive, worklist, compact_scratch, n_alive / 1024 + 2);
    FOC_CHECK_LAUNCH("occ_render_step(prepare)");
    int rc = foc_march_rays_two_phase(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas,
                                      noises, worklist, march_flags, stream);
    if (rc) return rc;
    // the field in pieces of at most `piece` samples: the [L, piece, 2] planes between the encoder and the whole-field kernel (64 B per sample)
    // are bounded whatever the burst length (FOC_OCC_FIELD_PIECE, samples; default 2^23. Measured on the 800 x 800 view, 5.1 M samples per iteration:
    // one piece 18.5 ms per view, pieces of 2^21 — planes that fit the 256 MiB Infinity Cache — 19.3, of 2^20 20.0: fewer launches win)
    uint64_t piece = (uint64_t)(uint32_t)foc_opt(FOC_OPT_OCC_FIELD_PIECE);
    if (piece < 1024) piece = 1024;
    piece &= ~(uint64_t)63;
    for (uint64_t m0 = 0; m0 < M; m0 += piece) {
        const uint32_t mc = (uint32_t)(M - m0 < piece ? M - m0 : piece);
        rc = foc_grid_encode_forward(xyzs + 3 * m0, embeddings, offsets, planes, mc, 3, 2, L, S, base_res, nullptr, 0, 0, 0, FOC_F16, offsets_host, stream);
        if (rc) return rc;
        rc = foc_nerf_field_inference(planes, 1, dirs + 3 * m0, 1, 0, mc, sigma_weights, sigma_layers, color_weights, color_layers, 64, activation, mc, sigma + m0,
                                      rgb + 3 * m0, obj_feat, stream);
        if (rc) return rc;
    }
    // composite marks finished rays in the INPUT list; the compaction then writes the survivors to the output list
    return foc_composite_compact(n_alive, n_step, T_thresh, const_cast<int32_t *>(rays_alive), rays_t, sigma, rgb, deltas, weights_sum, depth, image, rays_alive_out,
                                 count, compact_scratch, deaths, deaths_base, deaths_len, sample_major, stream);
}

} // extern "C"
