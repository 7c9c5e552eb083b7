/*
 * focnerf.h — C ABI of libfocnerf_hip.so, the MI355X (gfx950) implementation of
 * FOCNeRF's volume-rendering hot path.
 *
 * Every entry point replaces one function of the reference's four pybind11
 * extension modules (_raymarching, _gridencoder, _freqencoder, _ffmlp); the
 * reference declaration each one stands in for is cited as file:line into the
 * reference tree. Argument ORDER and MEANING follow the reference binding; the
 * differences are mechanical:
 *   - at::Tensor  -> raw device pointer (caller owns every buffer, as in the
 *                    reference where the Python wrapper allocates all outputs);
 *   - an explicit dtype code where the reference dispatches on scalar_type();
 *   - a trailing hipStream_t passed as void* (the reference launches on the
 *     legacy default stream; here the caller picks the stream);
 *   - int return: 0 = ok, non-zero = FOC_E_* with foc_last_error() holding a
 *     thread-local message (the reference raises through TORCH_CHECK /
 *     std::runtime_error, or checks nothing at all in _raymarching).
 * No function allocates, frees or synchronises; all are safe to capture into a
 * hipGraph. All pointers are DEVICE pointers unless stated otherwise.
 */
#ifndef FOCNERF_H
#define FOCNERF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOC_OK            0
#define FOC_E_INVALID     1   /* bad argument (null pointer, unsupported C/D/hidden_dim, ...) */
#define FOC_E_DTYPE       2   /* unsupported dtype code for this entry point */
#define FOC_E_LAUNCH      3   /* hipGetLastError() != hipSuccess after a launch */

#define FOC_F32 0
#define FOC_F16 1

/* ABI version of this header; bump on any change of a signature OR of a buffer contract.
 *   2 (round 5): the workspace of foc_ffmlp_backward / _backward_planar / foc_color_head_backward is
 *     [fp32 image of the weight blob | one slot of partial weight-gradient tiles per workgroup of the launch], written without a zero fill and
 *     tens of MB large — a buffer sized by version 1's blob formula (input_dim, hidden_dim, num_layers -> ~50 KB) is too small: size it with
 *     foc_ffmlp_backward_workspace_bytes(); the three entry points take the buffer's size (`workspace_bytes`) and refuse one that is too small;
 *     FocOccTrainNode carries `mlp_workspace_bytes`; foc_field_forward_train is new; input_dim up to 256 at every hidden width. */
#define FOC_ABI_VERSION 2
int         foc_abi_version(void);
/* Thread-local message of the last non-zero return on this thread ("" if none). */
const char *foc_last_error(void);
/* Tuning / test switches (csrc/common.h FocOpt): ints with a default, initialised once from the environment variable of the same name
 * (FOC_MLP_BWD_FUSED, FOC_FIELD_FWD_FUSED, FOC_GB_MERGE_MAX_RES, FOC_GB_FACTORED, FOC_GB_TAIL_SPLIT, FOC_GRID_FUSE_SMALL, FOC_GRID_PAIRS, FOC_GRID_FAST,
 * FOC_MARCH_SERIAL, FOC_MARCH_RAYS_ROW_MAX, FOC_OCC_MARCH_FORM, FOC_OCC_SAMPLE_MAJOR, FOC_OCC_FIELD_PIECE — INTEGRATION.md has the table)
 * and changeable at run time; no entry point reads the environment per call. Unknown name: FOC_E_INVALID. */
int foc_set_option(const char *name, int value);
int foc_get_option(const char *name, int *value);
/* Which device an entry point makes current for its call (host-only query of the rule, for tests): a non-null stream's device; for the
 * NULL stream (it exists on every device) the device the first pointer argument lives on; else the current device. -1 = unknown. */
int foc_guard_pick_device(int stream_is_null, int stream_device, int pointer_device, int current_device);
/* Compiled-for architecture string, e.g. "gfx950". */
const char *foc_arch(void);

/* ------------------------------------------------------------------------- *
 * _raymarching  (reference: raymarching/src/raymarching.h:7-18,
 *                           raymarching/src/bindings.cpp:5-19)
 * All floating tensors are fp32 (the reference's wrappers force fp32 with
 * custom_fwd(cast_inputs=torch.float32), raymarching/raymarching.py:21).
 * ------------------------------------------------------------------------- */

/* raymarching.cu:92-156  near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
 * rays_o/rays_d [N,3], aabb [6] (device), nears/fars [N]. Miss -> both FLT_MAX. */
int foc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                           uint32_t N, float min_near, float *nears, float *fars, void *stream);

/* raymarching.cu:163-209  sph_from_ray(rays_o, rays_d, radius, N, coords)   coords [N,2] */
int foc_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                     float *coords, void *stream);

/* raymarching.cu:214-232  morton3D(coords int32 [N,3], N, indices int32 [N]) */
int foc_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream);

/* raymarching.cu:237-260  morton3D_invert(indices int32 [N], N, coords int32 [N,3]) */
int foc_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream);

/* raymarching.cu:267-300  packbits(grid fp32 [N*8], N, density_thresh, bitfield u8 [N]) */
int foc_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, void *stream);

/* raymarching.cu:311-490  march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps,
 *                                          N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
 * grid u8 [C*H^3/8]; xyzs/dirs [M,3], deltas [M,2] (caller pre-zeroed, raymarching.py:205-207);
 * rays int32 [N,3] = (ray id, point offset, point count); counter int32[2] is ADDED to
 * (counter[0] += total points, counter[1] += N) exactly like the reference's atomicAdd pair
 * (raymarching.cu:405-406).
 * Order: the reference's slot order depends on atomicAdd arrival; this implementation
 * reserves slots by an exclusive scan in RAY ORDER (one of the reference's legal outcomes),
 * so the result is deterministic: rays[i] = (i, counter0_before + sum_{j<i} n_j, n_i).
 * scratch: caller-owned device bytes (foc_march_rays_train_scratch_bytes): per-ray counts + scan carry, and one
 * strip of max_steps sample positions per ray, so that the occupancy walk runs once (the reference walks every ray
 * twice, raymarching.cu:357-399 and :415-479). */
int foc_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid,
                         float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float *nears, const float *fars,
                         float *xyzs, float *dirs, float *deltas,
                         int32_t *rays, int32_t *counter, const float *noises,
                         int32_t *scratch, void *stream);
/* Bytes of `scratch` needed by foc_march_rays_train for N rays of at most max_steps samples. */
uint64_t foc_march_rays_train_scratch_bytes(uint32_t N, uint32_t max_steps);

/* Extension (no reference binding): the same march — same rays table, counter and per-ray samples — with the sample list in the
 * layout the fused occupancy-grid training path consumes (focnerf_amd/occtrain.py; the caller-side expressions of
 * legacy/nerf/renderer.py:288-300 and nerf/network_ff.py:51-68 folded into the emit pass):
 *   enc_in [M,3] fp32  = (xyz + bound) * (1 / (2 bound)), the encoder's [0,1] coordinates as torch evaluates (x + bound) / (2 bound);
 *   sh_rows [M,16] fp16 = the degree-4 SH values of each sample's ray direction (k-chunk 0 of the colour network's input);
 *   deltas [M,2] as above. No `dirs`. EVERY row of the three arrays is written (rays that do not fit the list and the rows behind the
 *   last ray receive zeros): the caller does not pre-zero them. pad_align > 0: the rows behind the last ray are zeroed only up to the
 *   next multiple of pad_align above counter[0] (a caller that cuts the list there, raymarching.py:223-229, reads nothing beyond).
 *   aabb != NULL: nears / fars [N] are OUTPUTS — the slab test of foc_near_far_from_aabb(aabb [6], min_near) is done by the count pass
 *   (same expressions, same bits); aabb == NULL: they are inputs as in foc_march_rays_train. */
int foc_march_rays_train_field(const float *rays_o, const float *rays_d, const uint8_t *grid,
                               float bound, float dt_gamma, uint32_t max_steps,
                               uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                               float *nears, float *fars,
                               float *enc_in, void *sh_rows, float *deltas,
                               int32_t *rays, int32_t *counter, const float *noises,
                               int32_t *scratch, uint32_t pad_align, const float *aabb, float min_near, void *stream);

/* raymarching.cu:500-588  composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh,
 *                                                      weights_sum, depth, image) */
int foc_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                     const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                     float *weights_sum, float *depth, float *image, void *stream);

/* raymarching.cu:601-693  composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs,
 *       deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs)
 * grad_sigmas/grad_rgbs must be pre-zeroed by the caller (raymarching.py:283-284). grad_weights_sum may be NULL (= all zero: the loss
 * does not read weights_sum). */
int foc_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                      const float *sigmas, const float *rgbs, const float *deltas,
                                      const int32_t *rays, const float *weights_sum, const float *image,
                                      uint32_t M, uint32_t N, float T_thresh,
                                      float *grad_sigmas, float *grad_rgbs, void *stream);

/* raymarching.cu:700-815  march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
 *       dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises)
 * xyzs/dirs/deltas [n_alive*n_step (+pad), 3|3|2] pre-zeroed (raymarching.py:334-336). */
int foc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                   const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                   uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                   const float *nears, const float *fars,
                   float *xyzs, float *dirs, float *deltas, const float *noises, void *stream);

/* raymarching.cu:818-914  composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs,
 *       deltas, weights_sum, depth, image)   — in place; rays_alive[n] = -1 marks a finished ray. */
int foc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                       int32_t *rays_alive, float *rays_t,
                       const float *sigmas, const float *rgbs, const float *deltas,
                       float *weights_sum, float *depth, float *image, void *stream);

/* Device-side compaction of rays_alive (replaces the reference caller's
 * `rays_alive = rays_alive[rays_alive >= 0]`, legacy/nerf/renderer.py:363, which costs a
 * host sync per marching iteration). Writes the surviving ids, in order, to `out` and
 * their count to n_out[0] (device int32). Extension: no reference binding. */
/* One iteration of the inference loop of legacy/nerf/renderer.py:323-372 as one call (csrc/occrender.hip): zero the iteration's sample
 * slots -> march_rays (two phases: first visits, then the walkers densely packed) -> x to [0,1] -> grid_encode_forward ([L,M,2] planes) ->
 * foc_nerf_field_inference (per-sample directions) -> composite_rays (in place; finished rays marked -1 in rays_alive) -> ordered
 * compaction into rays_alive_out, whose entries behind the count are -1. M = n_alive * n_step.
 * n_alive may be an UPPER BOUND of the live count: entries of rays_alive that are -1 are skipped by every stage, so a caller may read
 * `count` (device int) late. samples: fp32 [M * 8] = positions [M,3] (in the encoder's [0,1] coordinates) | dirs [M,3] | deltas [M,2]; planes fp16
 * [L, min(M, piece), 2]: the field is evaluated in pieces of FOC_OCC_FIELD_PIECE samples (default 2^23), whose planes stay in the Infinity Cache;
 * sigma [M], rgb [M,3] fp32; scratch: foc_occ_render_step_scratch_bytes(n_alive of the FIRST iteration) bytes. Hash grid D = 3, C = 2,
 * fp16 table (embeddings), linear interpolation; networks as foc_nerf_field_inference (hidden 64; obj_feat may be NULL). density_scale 1.
 * flags bit 0: the march re-derives t after every sample (foc_march_rays_two_phase, flag bit 1): a caller that marches n_step samples where
 * the reference's loop would march one per iteration sets it and receives the reference's samples. deaths / deaths_base / deaths_len: as
 * foc_composite_compact. */
/* The two building blocks of the step that have no reference counterpart, usable on their own:
 * foc_march_rays_two_phase — foc_march_rays with the same arguments and results (bit for bit), as two launches: first visits per lane, then
 *   the rays that met an empty cell ("walkers") compacted on a worklist and marched 16 lanes per ray (one ray per lane when the list is
 *   long). scratch: int32[n_alive + 4] whose first word the caller has zeroed on this stream. `normalised` is a flag word. Bit 0: xyzs
 *   receives (x + bound) * (1 / (2 bound)), the encoder's [0,1] coordinates, instead of x. Bit 1 ("re-derive"): after every emitted sample
 *   the march continues from last_t + (t - last_t) instead of t — the value composite_rays reconstructs from deltas[:,1] and stores in
 *   rays_t for the NEXT call (raymarching.cu:871, 899). The two differ by an ulp when t - last_t is not exact in fp32 (a skip over empty
 *   space that more than doubles t), so the reference's samples depend on where its bursts end; with bit 1 a burst of k samples marches
 *   exactly what k consecutive calls with n_step = 1 would, whatever k. Bit 2 ("sample-major"): xyzs / dirs / deltas are written as
 *   [n_step][n_alive] arrays instead of [n_alive][n_step] — honoured by the staged one-ray-per-lane kernel only
 *   (foc_march_rays_two_phase_sample_major(n_alive, n_step, flags) != 0 says whether this call would); foc_composite_compact reads that
 *   layout with sample_major != 0. The 64 rows an encoder / network wave works on are then 64 neighbouring rays at one burst slot. Bursts of more than two samples (most rays meet an
 *   empty cell inside them and would be marched twice) take ONE launch instead — one ray per lane — of a kernel that collects the
 *   samples in LDS and writes ALL n_step slots of every list entry
 *   as runs of consecutive floats (zeros where the ray ended early or the entry is -1): foc_march_rays_two_phase_fills(n_step) != 0 says
 *   so, and the caller may then skip zeroing xyzs / dirs / deltas. FOC_OCC_MARCH_FORM = two | row | lane | staged overrides the choice
 *   (A/B runs, tests; "lane" = foc_march_rays' serial kernel, which needs the zeros).
 * foc_composite_compact — foc_composite_rays followed by the ordered compaction of the surviving list entries into `out` (count in n_out),
 *   the compaction's counting pass done by the composite kernel. block_counts: int32[n_alive / 1024 + 2], zeroed by the caller.
 *   deaths (may be NULL): int32[deaths_len][64] histogram in 64 slices (sum them): deaths[min(deaths_base + j, deaths_len - 1)][slice] += 1
 *   for every ray that ends at slot j of this call (no sample there, or the transmittance test after it). With deaths_base = the samples marched per ray so far, the histogram
 *   over a view tells how many rays were alive after any number of samples — what the reference's burst rule (renderer.py:337) looks at —
 *   to a caller that deals the samples over iterations differently. */
int foc_march_rays_two_phase(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                             const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                             const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                             const float *noises, int32_t *scratch, int normalised, void *stream);
int foc_march_rays_two_phase_fills(uint32_t n_step, int flags);
int foc_march_rays_two_phase_sample_major(uint32_t n_alive, uint32_t n_step, int flags);
int foc_composite_compact(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                          const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth,
                          float *image, int32_t *out, int32_t *n_out, int32_t *block_counts, int32_t *deaths, uint32_t deaths_base,
                          uint32_t deaths_len, int sample_major, void *stream);
uint64_t foc_occ_render_step_scratch_bytes(uint32_t n_rays);
int foc_occ_render_step(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, int32_t *rays_alive_out, int32_t *count,
                        float *rays_t, const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars, const float *noises,
                        float *samples, void *planes, float *sigma, float *rgb,
                        const void *embeddings, const int32_t *offsets, const int32_t *offsets_host, uint32_t L, float S, uint32_t base_res,
                        const void *sigma_weights, uint32_t sigma_layers, const void *color_weights, uint32_t color_layers, uint32_t activation,
                        const void *obj_feat, float T_thresh, float *weights_sum, float *depth, float *image, void *scratch, uint32_t flags,
                        int32_t *deaths, uint32_t deaths_base, uint32_t deaths_len, void *stream);

int foc_compact_alive(const int32_t *rays_alive, uint32_t n_alive, int32_t *out, int32_t *n_out,
                      int32_t *scratch, void *stream);

/* ------------------------------------------------------------------------- *
 * _gridencoder  (reference: gridencoder/src/gridencoder.h:12-15,
 *                           gridencoder/src/bindings.cpp)
 * inputs fp32 [B,D] in [0,1]; embeddings/outputs/dy_dx/grad* share `dtype`
 * (FOC_F32 or FOC_F16 — the reference dispatches on embeddings.scalar_type(),
 * gridencoder.cu:467-470); offsets int32 [L+1] (device). D in {2,3}, C in {1,2,4,8}.
 * S = log2(per_level_scale). Per-level scale = exp2f(l*S)*H-1 is evaluated on the
 * HOST (libm) and handed to the kernel, so the integer index math does not depend on
 * a device transcendental. offsets_host: the same L+1 ints in HOST memory (needed to
 * size launches without a D2H copy); may be NULL, in which case dense-level LDS
 * fast paths are disabled.
 * ------------------------------------------------------------------------- */

/* gridencoder.cu:448-471  grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H,
 *                                             dy_dx?, gridtype, align_corners, interp)
 * outputs [L,B,C] (level-major, as the reference kernel writes it); dy_dx [B,L,D,C] or NULL. */
int foc_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets,
                            void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            float S, uint32_t H, void *dy_dx, uint32_t gridtype,
                            int align_corners, uint32_t interp, int dtype,
                            const int32_t *offsets_host, void *stream);

/* Same computation, but writes outputs as [B, L*C] (what the reference's Python wrapper
 * produces with a permute+reshape copy, gridencoder/grid.py:57). Extension used by this
 * repo's wrapper to skip that copy; results are element-for-element identical. */
int foc_grid_encode_forward_bl(const float *inputs, const void *embeddings, const int32_t *offsets,
                               void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                               float S, uint32_t H, void *dy_dx, uint32_t gridtype,
                               int align_corners, uint32_t interp, int dtype,
                               const int32_t *offsets_host, void *stream);
/* [L,B,C] -> [B,L*C] for C * sizeof(element) = unit_bytes in {4, 8}: the permute + copy of grid.py:57 as one kernel, so that
 * the [B,L*C] result can come from the level-major forward kernel (faster than the point-major one on incoherent points). */
int foc_grid_planes_to_rows(const void *planes, void *rows, uint32_t B, uint32_t L, uint32_t unit_bytes, void *stream);
/* Host-only query (tests): the index arithmetic foc_grid_encode_forward uses for one level of an fp16, D = 3, C = 2 hash grid —
 * 2: 32-bit byte offsets with 24-bit multiplies (levels of at most 2^22 rows: every NeRF grid), 1: 32-bit byte offsets, generic
 * multiplies (up to 2^30 rows; a hashed level must have a power-of-two size), 0: the general 64-bit form (everything else). */
int foc_grid_forward_index_path(uint32_t level_rows, uint32_t resolution, uint32_t level_offset_rows);
/* [B,L*C] -> [L,B,C]: the permute + copy of grid.py:75 (the gradient on its way into the backward kernel) as one kernel. */
int foc_grid_rows_to_planes(const void *rows, void *planes, uint32_t B, uint32_t L, uint32_t unit_bytes, void *stream);

/* gridencoder.cu:473-503  grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings,
 *       B, D, C, L, S, H, dy_dx?, grad_inputs?, gridtype, align_corners, interp)
 * grad [L,B,C]; grad_embeddings pre-zeroed (grid.py:77); grad_inputs [B,D] (dtype) or NULL.
 * grad_is_bl != 0: grad is laid out [B, L*C] instead (skips the wrapper's permute copy,
 * gridencoder/grid.py:75). */
int foc_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                             const int32_t *offsets, void *grad_embeddings,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             const void *dy_dx, void *grad_inputs, uint32_t gridtype,
                             int align_corners, uint32_t interp, int dtype, int grad_is_bl,
                             const int32_t *offsets_host, void *stream);

/* Same result as foc_grid_encode_backward for D = 3, C = 2 tables, computed WITHOUT scattered atomics:
 * the (row, w*grad) contributions are partitioned by 8192-row table segment and summed per segment in
 * LDS (fp32), then added to grad_embeddings with contiguous atomics. Scattered memory-side atomics cap
 * the atomic entry point at ~2x10^10 corner updates/s on MI355X; this path streams instead.
 * workspace: device scratch of foc_grid_encode_backward_workspace_bytes() bytes (0 = shape not
 * supported: call foc_grid_encode_backward). offsets_host (L+1 ints in HOST memory) is required.
 * Extension: no reference binding (the reference has only the atomic kernel, gridencoder.cu:248-340). */
uint64_t foc_grid_encode_backward_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t L, int dtype);
int foc_grid_encode_backward_binned(const void *grad, const float *inputs, const void *embeddings,
                                    const int32_t *offsets, void *grad_embeddings,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    const void *dy_dx, void *grad_inputs, uint32_t gridtype,
                                    int align_corners, uint32_t interp, int dtype, int grad_is_bl,
                                    const int32_t *offsets_host, void *workspace, uint64_t workspace_bytes,
                                    void *stream);

/* The binned pass in two halves, so that the gradient-independent one (record counts -> record ranges, which needs the
 * sample positions only) can run ahead of time: foc_grid_encode_backward_count fills the workspace header on its own;
 * foc_grid_encode_forward_counted is foc_grid_encode_forward ([L,B,C] outputs, no dy_dx; D = 3, C = 2) with that pass riding
 * along in the same launch — the forward gathers are bound by cache requests and leave the VALU idle, so the count costs
 * almost nothing there (a training forward). foc_grid_encode_backward_binned_counted then does
 * the scatter + reduce for the SAME inputs / offsets / B / dtype on the SAME workspace (nothing else may have used the
 * workspace in between; the caller orders the two calls, across streams with an event). */
int foc_grid_encode_forward_counted(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                    const int32_t *offsets_host, void *workspace, uint64_t workspace_bytes, void *stream);
int foc_grid_encode_backward_count(const float *inputs, const int32_t *offsets, uint32_t B, uint32_t D, uint32_t C,
                                   uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                   uint32_t interp, int dtype, const int32_t *offsets_host, void *workspace,
                                   uint64_t workspace_bytes, void *stream);
int foc_grid_encode_backward_binned_counted(const void *grad, const float *inputs, const void *embeddings,
                                            const int32_t *offsets, void *grad_embeddings,
                                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                            const void *dy_dx, void *grad_inputs, uint32_t gridtype,
                                            int align_corners, uint32_t interp, int dtype, int grad_is_bl,
                                            const int32_t *offsets_host, void *workspace, uint64_t workspace_bytes,
                                            void *stream);

/* gridencoder.cu:639-645  grad_total_variation(inputs, embeddings, grad, offsets, weight,
 *       B, D, C, L, S, H, gridtype, align_corners)   — inputs share `dtype` with embeddings. */
int foc_grad_total_variation(const void *inputs, const void *embeddings, void *grad,
                             const int32_t *offsets, float weight,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             uint32_t gridtype, int align_corners, int dtype, void *stream);

/* ------------------------------------------------------------------------- *
 * _freqencoder  (reference: freqencoder/src/freqencoder.h:7,10) — fp32 only.
 * ------------------------------------------------------------------------- */

/* freqencoder.cu:97-110  freq_encode_forward(inputs [B,D], B, D, deg, C, outputs [B,C]) */
int foc_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                            float *outputs, void *stream);
/* freqencoder.cu:113-129 freq_encode_backward(grad [B,C], outputs [B,C], B, D, deg, C, grad_inputs [B,D]) */
int foc_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D,
                             uint32_t deg, uint32_t C, float *grad_inputs, void *stream);

/* ------------------------------------------------------------------------- *
 * _ffmlp  (reference: ffmlp/src/ffmlp.h:8-14) — fp16 storage, fp32 MFMA accumulation.
 * weights: one fp16 blob [hidden*input_dim | (num_layers-1)*hidden*hidden | 16*hidden],
 * each block row-major with the OUTPUT neuron as the row (ffmlp.cu:631-634).
 * inputs [B,input_dim], outputs [B,16] (output_dim is the padded 16), row-major.
 * Any B >= 1 is accepted (the reference kernels need a multiple of 128 and its wrapper pads with a copy,
 * ffmlp/ffmlp.py:157-159; here the ragged last tile is handled in the kernels, so no padded copy is needed);
 * hidden_dim in {16,32,64,128,256}; input_dim % 16 == 0; output_dim <= 16; num_layers >= 2.
 * activation codes follow ffmlp.py:86-93: 0 relu, 1 exponential, 2 sine, 3 sigmoid, 4 squareplus, 5 softplus, 6 none — the hidden
 * activations of ffmlp/src/utils.h:424-589, evaluated as there (fp32 function of the half-rounded sum; the backward factor from the stored
 * post-activation in half arithmetic; Sine's backward a pass-through, as the reference leaves it). Codes 1..5 take the reference's data
 * flow: foc_ffmlp_backward then needs forward_buffer and backward_buffer (the single-pass kernel and the planar / head / whole-field forms
 * serve relu|none, which is what every FOC network uses). Output activation none (FFMLP can construct no other, ffmlp.py:107-108).
 * ------------------------------------------------------------------------- */

/* ffmlp.cu:635-671  ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers,
 *       activation, output_activation, forward_buffer [num_layers,B,hidden], outputs)
 * forward_buffer may be NULL: no activations are kept (outputs are the same bits); pass NULL
 * to foc_ffmlp_backward as well and it re-evaluates them from `inputs` on chip (hidden_dim <= 64,
 * input_dim <= 64, num_layers 2..4 — the shapes of the fused backward kernel). */
int foc_ffmlp_forward(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim,
                      uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                      uint32_t activation, uint32_t output_activation,
                      void *forward_buffer, void *outputs, void *stream);

/* ffmlp.cu:673-709  ffmlp_inference(..., inference_buffer [B,hidden] (unused here), outputs) */
int foc_ffmlp_inference(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim,
                        uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                        uint32_t activation, uint32_t output_activation,
                        void *inference_buffer, void *outputs, void *stream);

/* ffmlp.cu:749-895  ffmlp_backward(grad [B,16], inputs, weights, forward_buffer, B, input_dim,
 *       output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs,
 *       backward_buffer [num_layers,B,hidden], grad_inputs [B,input_dim], grad_weights (blob))
 * forward_buffer NULL: see foc_ffmlp_forward. backward_buffer may be NULL for the same shapes (the
 * activation gradients then never leave the chip); other shapes need both (FOC_E_INVALID otherwise).
 * workspace: device memory, foc_ffmlp_backward_workspace_bytes(input_dim, hidden_dim, num_layers) bytes, caller-owned, needs NO zero fill:
 * [fp32 image of the weight blob: the split-K sums of the two-kernel form, the object-conditioned head's finalize] followed, for the shapes
 * the single-pass kernel serves (hidden_dim <= 64, input_dim <= 64, 2..4 layers), by up to 1024 per-workgroup slots of
 * (num_layers + 1) x 4096 fp32 partial weight-gradient tiles that a second kernel sums in a fixed order (33 - 80 MB; the reference's CUTLASS
 * split-K workspace, cutlass_matmul.h:335-363, is a process-global map instead). `workspace_bytes` = the size of the caller's buffer: a
 * buffer smaller than foc_ffmlp_backward_workspace_bytes() is refused (FOC_E_INVALID) instead of being written past its end; for the colour
 * head with an object feature ask for input_dim 48 (the slots start behind a 48-wide blob image). */
int foc_ffmlp_backward(const void *grad, const void *inputs, const void *weights,
                       const void *forward_buffer, uint32_t B, uint32_t input_dim,
                       uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                       uint32_t activation, uint32_t output_activation, int calc_grad_inputs,
                       void *backward_buffer, void *grad_inputs, void *grad_weights,
                       void *workspace, uint64_t workspace_bytes, void *stream);
uint64_t foc_ffmlp_backward_workspace_bytes(uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers);

/* Planar-input forms for the encoder -> MLP pair. `inputs_planar` is [input_dim/2][B] half2 planes: exactly the
 * [L, B, C=2] tensor foc_grid_encode_forward writes (gridencoder.cu:218), consumed here without the
 * permute + copy to [B, L*C] the reference wrapper makes (grid.py:57); `grad_inputs_planar` has the same layout,
 * i.e. the [L, B, C] gradient foc_grid_encode_backward(_binned) reads (grid.py:75 builds it with another permute).
 * Semantics otherwise as foc_ffmlp_forward / foc_ffmlp_backward with forward_buffer and backward_buffer NULL
 * (same shape limits: hidden_dim <= 64, input_dim <= 64, num_layers 2..4 for the backward). */
int foc_ffmlp_forward_planar(const void *inputs_planar, const void *weights, uint32_t B, uint32_t input_dim,
                             uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                             uint32_t activation, uint32_t output_activation, void *outputs, void *stream);
int foc_ffmlp_backward_planar(const void *grad, const void *inputs_planar, const void *weights, uint32_t B,
                              uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                              uint32_t activation, uint32_t output_activation, int calc_grad_inputs,
                              void *grad_inputs_planar, void *grad_weights, void *workspace, uint64_t workspace_bytes, void *stream);

/* Inference of the whole NeRF field per sample in one kernel (nerf/network_ff.py:51-75): sigma network on the hash-grid
 * encoding `enc` ([B,32] fp16 row-major, or the encoder's [16,B,2] planes when enc_planar), trunc_exp, degree-4 SH of the
 * direction, colour network, sigmoid. dirs [B / dir_div, 3] fp32: sample s uses direction s / dir_div (dir_div = 1 for
 * per-sample directions, = samples per ray when they are per ray). sigma [B] fp32 (may be NULL), rgb [B,3] fp32 (values
 * rounded to fp16 like the half sigmoid). Same bits as foc_ffmlp_inference -> foc_sample_head_forward -> foc_ffmlp_inference
 * -> foc_rgb_head_forward; hidden_dim 64, (sigma_layers, color_layers) in {(2,2), (2,3), (3,3)}.
 * dir_block > 0: the rows stand in the block-interleaved sample order of foc_fixed_sample (ray_block = dir_block, dir_div =
 * samples per ray): row r uses direction min((r / (dir_block*dir_div))*dir_block + r % dir_block, n_dirs - 1). dir_block = 0:
 * n_dirs is not read.
 * obj_feat (may be NULL): FOC's object-conditioned colour network, nerf/network_tcnn.py:611-640 — [16] fp16, the ENCODED object
 * feature (yolo_feat_encoder output), one vector for every sample. The colour network's input row is then
 * [SH16 | h[1:16] | obj_feat 16 | 0] (48 wide, W0 rows of 48 in color_weights); its share W0[:,31:47] . obj_feat enters as the
 * initial value of the layer-0 accumulators (fp32), nothing else changes. Needs enc_planar = 1 and ReLU. */
int foc_nerf_field_inference(const void *enc, int enc_planar, const float *dirs, uint32_t dir_div,
                             uint32_t dir_block, uint32_t n_dirs, const void *sigma_weights, uint32_t sigma_layers, const void *color_weights,
                             uint32_t color_layers, uint32_t hidden_dim, uint32_t activation, uint32_t B,
                             float *sigma, float *rgb, const void *obj_feat, void *stream);

/* The colour network of a ray-ordered sample list without its materialised input (network_ff.py:104-108 builds
 * cin = [SH16(dir) | h[:,1:16] | 0] per sample, 64 B written and read twice): the kernels take the sigma network's output
 * rows h [B,16] fp16 and one SH row per ray, ray_sh [B / samples_per_ray, 16] fp16 (foc_fixed_sample), and place
 * every value at the k position it has in cin — same bits as foc_ffmlp_forward / foc_ffmlp_backward (forward_buffer NULL,
 * input_dim 32, output_dim 16) on that cin. outputs [B,16] fp16.
 * Backward: grad [B,16] fp16 -> grad_weights (blob layout of foc_ffmlp_backward, same workspace) and grad_h [B,16] fp16 =
 * [grad_h0 | grad_cin[:,16:31]] with grad_h0 [B] fp16 (NULL = zeros) the density path's gradient of h[:,0]
 * (foc_fixed_tail_backward) — the row the sigma network's backward consumes, written once. hidden_dim 64, 2 or 3 layers.
 * out_width 16: outputs / grad are [B,16]; 4: only the columns that are ever read exist, outputs / grad are [B,4] (of the 16 padded
 * outputs of a 3-output network columns 0..2 are the rgb logits; the gradient of the others is zero by construction).
 * obj_feat (may be NULL): FOC's object-conditioned colour network (nerf/network_tcnn.py:611-640; see foc_nerf_field_inference) —
 * [16] fp16, cin = [SH16 | h[:,1:16] | obj_feat | 0] (48 wide; weights and grad_weights then have 48-wide W0 rows, the workspace is
 * foc_ffmlp_backward_workspace_bytes(48, ...)). Same values as foc_ffmlp_forward / _backward on that cin up to fp32 summation order
 * (the object columns' products are summed first). Backward: grad_weights[:, 31:47] = (sum_b delta_0[b,:]) (x) obj_feat, and
 * grad_obj [16] fp32 (may be NULL) = W0[:,31:47]^T sum_b delta_0[b,:], the gradient the object-feature encoder consumes. */
int foc_color_head_forward(const void *h, const void *ray_sh, uint32_t samples_per_ray, const void *weights,
                           uint32_t B, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                           void *outputs, uint32_t out_width, const void *obj_feat, void *stream);
int foc_color_head_backward(const void *grad, const void *h, const void *ray_sh, uint32_t samples_per_ray,
                            const void *grad_h0, const void *weights, uint32_t B, uint32_t hidden_dim,
                            uint32_t num_layers, uint32_t activation, void *grad_h, void *grad_weights,
                            void *workspace, uint64_t workspace_bytes, uint32_t out_width, const void *obj_feat, float *grad_obj, void *stream);

/* The TRAINING forward of the whole field in ONE kernel (csrc/field_fwd.hip): foc_ffmlp_forward_planar on the encoder's planes ->
 * h [B,16] fp16 (written: the compositing tail and the backward read it) -> foc_color_head_forward fed from h and ray_sh, back to back in
 * registers — nerf/network_ff.py:51-75 as ffmlp.cu:331-407 evaluates it twice. Results are BIT FOR BIT those of the two separate calls
 * (same operands in the same k positions and order for every MFMA); h is not read back for the colour network (-32 B per sample, one
 * launch). planes [16][B] half2 (input width 32), hidden_dim 64, (sigma_layers, color_layers) in {(2,2), (2,3), (3,3)}, activation relu(0) /
 * none(6); ray_sh / samples_per_ray / out_width / obj_feat as foc_color_head_forward. The backward stays the two calls
 * foc_color_head_backward -> foc_ffmlp_backward_planar (they re-evaluate the activations from h and the planes). */
int foc_field_forward_train(const void *planes, const void *sigma_weights, uint32_t sigma_layers, const void *ray_sh, uint32_t samples_per_ray,
                            const void *color_weights, uint32_t color_layers, uint32_t hidden_dim, uint32_t activation, uint32_t B,
                            void *h, void *c, uint32_t out_width, const void *obj_feat, void *stream);

/* ffmlp.cu:721-740  allocate_splitk(size) / free_splitk(): the reference creates side
 * streams for its CUTLASS split-K GEMMs. Weight gradients here are produced inside the
 * backward launch sequence on the caller's stream, so both are no-ops kept for API parity. */
int foc_allocate_splitk(uint64_t size);
int foc_free_splitk(void);

/* ------------------------------------------------------------------------- *
 * Multi-object combine (reference: COMBINED.py:247-251 best_densities_and_colors_v3).
 * Per-sample strict-'>' max-density select; first object wins ties.
 * ------------------------------------------------------------------------- */

/* In place: where dens[i] > max_dens[i]: best_rgb[i,:] = rgb[i,:]; max_dens[i] = max(dens, max_dens).
 * n = number of samples. */
int foc_combine_select(const float *dens, const float *rgb, float *max_dens, float *best_rgb,
                       uint64_t n, void *stream);

/* Pack (sigma, rank) into the order-preserving 64-bit key used for the RCCL MAX
 * all-reduce of the one-object-per-GPU combine: key = (float_bits(max(sigma,0)) << 32) |
 * (0xFFFFFFFF - rank); and the inverse select of the winning rank's rgb. */
int foc_combine_pack_keys(const float *dens, uint32_t rank, uint64_t *keys, uint64_t n, void *stream);
int foc_combine_unpack(const uint64_t *keys, uint32_t rank, const float *rgb,
                       float *max_dens, float *masked_rgb, uint64_t n, void *stream);

/* Fixed-step composite of a merged field (COMBINED.py:141-200 image_depth_generation):
 * sigmas [N,T], rgbs [N,T,3], nears/fars [N]; out image4 [N,4] (rgb + sum w*sigma), depth [N].
 * bg: background value broadcast over the 4 channels; result clamped to [0,1]. */
int foc_composite_fixed_steps(const float *sigmas, const float *rgbs, const float *nears,
                              const float *fars, uint32_t N, uint32_t T, float bg,
                              float *image4, float *depth, void *stream);

/* The same select and composite fused over K objects' PACKED fields, the form the one-object-per-GPU exchange moves
 * (focnerf_amd/combine.py): fields4[k] -> device pointer to object k's [N,T,4] fp32 rows (sigma, r, g, b), k in checkpoint
 * order (host array of K device pointers, K <= 16; more objects: pre-merge with foc_combine_select4, the rule is
 * associative over the order). Replaces the object loop of COMBINED.py:598-618 + image_depth_generation :141-200 for one
 * ray chunk without storing the merged field. bgs: host array of n_bg (1 or 2) background values (the reference composites
 * white and black, compute_metrics_both_backgrounds); image4 [n_bg,N,4], depth [N]; merged4 [N,T,4] or NULL. */
int foc_combine_select_composite(const float *const *fields4, uint32_t K, const float *nears, const float *fars,
                                 uint32_t N, uint32_t T, const float *bgs, uint32_t n_bg, float *image4, float *depth,
                                 float *merged4, void *stream);

/* acc4 <- select(acc4, field4) on n packed samples (same rule; acc4 holds the earlier objects). */
int foc_combine_select4(const float *field4, float *acc4, uint64_t n, void *stream);

/* MONeRFNetwork's running select (reference: nerf/multiobjectnetwork.py:66-82 — torch.max over stack([new, best]) with
 * take_along_dim of the rows that travel with the density: geo_feat [n,15] in density(), colour [n,3] in color()).
 * In place: where the new object takes sample i, sigma_best[i] = sigma_new[i] and feat_best[i,:] = feat_new[i,:].
 * The rule is torch.max's: the first maximal index wins and the new object is stacked first, so the NEW object takes
 * ties (COMBINED.py's select keeps the earlier one); NaN counts as maximal (a NaN of either side stays, the new one's
 * on both). sigma and feat have the same element type: elem_bytes 2 (half, under autocast) or 4 (float);
 * feat_width 1..64 elements per row, rows contiguous. */
int foc_mo_select(const void *sigma_new, const void *feat_new, void *sigma_best, void *feat_best, uint64_t n,
                  uint32_t feat_width, uint32_t elem_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Fixed-step render path as fused ops. No reference binding: these replace the torch
 * code of nerf/renderer.py:145-221 (== COMBINED.py:451-534) and the glue of nerf/network_ff.py:51-134
 * between the encoder and the two MLPs. M = N*T samples, ray-major. `noise` ([M] uniform(0,1), the
 * reference's torch.rand for perturb, always ray-major) may be NULL.
 *
 * Sample order of the inference entry points (`ray_block`): 0 = ray-major, sample i of ray n at row n*T + i.
 * 64 = block-interleaved: row (n/64)*64*T + i*64 + n%64, ceil(N/64)*64*T rows, the last block padded with copies
 * of ray N-1 — 64 neighbouring rays at one depth are 64 consecutive rows, which is what makes the level-major
 * encoder forward fast on a rendered view (its wave then finds its corner rows in a few cache lines). What
 * leaves the path for the caller (image, depth, field4, rgb_masked, sigma_raymajor) is ray-major either way.
 * ------------------------------------------------------------------------- */

/* z = near + (far-near)*linspace(0,1,T)[i] (+ (noise-0.5)*(far-near)/T); xyz = clip(o + d z, aabb) -> xyzs [M,3]
 * and/or enc_in [M,3] = (xyz + bound)/(2 bound), the GridEncoder's normalised input (grid.py:149). */
int foc_fixed_sample(const float *rays_o, const float *rays_d, const float *nears, const float *fars,
                     const float *aabb, const float *noise, uint32_t N, uint32_t T, float bound,
                     float *xyzs, float *enc_in, void *ray_sh, uint32_t ray_block, void *stream);
/* ray_sh [N,16] fp16 or NULL: the degree-4 SH values of each ray's direction, rounded to fp16 as they stand in columns 0..15 of
 * the colour-net input (for foc_color_head_forward). */

/* h [M,16] fp16 = sigma-net output. sigma = exp(h[:,0]); weights = alpha * cumprod(1-alpha+1e-15);
 * trans [M] = transmittance before each sample; weights_sum/depth [N]; cin [M,cin_width] fp16 or NULL:
 *   cin_width 32: [SH16(dir) | h[:,1:16] | 0]                 (the colour-net input, network_ff.py:62-68)
 *   cin_width 48: [SH16(dir) | h[:,1:16] | obj_feat[16] | 0]  (FOC network with the encoded YOLO object feature,
 *                 nerf/network_tcnn.py:641-643; obj_feat = 16 fp16 values shared by all samples, NULL = zeros). */
int foc_fixed_head_forward(const void *h, const float *rays_d, const float *nears, const float *fars,
                           const float *noise, uint32_t N, uint32_t T, float density_scale,
                           float *sigma, float *trans, float *weights, float *weights_sum, float *depth,
                           void *cin, const void *obj_feat, uint32_t cin_width, void *stream);
/* grad_w [M], grad_ws [N], grad_depth [N], grad_cin [M,cin_width] fp16 (each may be NULL) -> grad_h [M,16] fp16
 * (the object-feature gradient is the column sum of grad_cin[:,31:47], left to the caller). */
int foc_fixed_head_backward(const void *h, const float *sigma, const float *trans, const float *nears,
                            const float *fars, const float *noise, const float *grad_w, const float *grad_ws,
                            const float *grad_depth, const void *grad_cin, uint32_t N, uint32_t T,
                            float density_scale, void *grad_h, uint32_t cin_width, void *stream);

/* Training tail in one pass per direction (one wave per ray): foc_fixed_head_forward (cin NULL) + foc_fixed_composite_forward,
 * and foc_fixed_composite_backward + column 0 of foc_fixed_head_backward (written as grad_h0 [M] fp16) — the same bits as the pairs; the weights are not
 * re-read, their gradient never leaves the lane, and the backward does not read h (exp(clamp(h0,-15,15)) = clamp(sigma, ...)).
 * c_width 16: c and grad_c are [M,16] rows as above; 4: they are [M,4] (rgb logits + one pad column), the compact form
 * foc_color_head_forward / _backward exchange with out_width 4.
 * ray_sumsq [N] (may be NULL): sum over the ray's samples of sigma^2 — what FOC's outside-mask density criterion
 * ||sigma[rays outside the object mask]||_2 (nerf/renderer.py:163-165) needs from the samples; grad_sumsq [N] (may be NULL) is its
 * gradient, added to the density path as 2 sigma grad_sumsq[ray] before trunc_exp's backward factor. */
int foc_fixed_tail_forward(const void *h, const void *c, const float *nears, const float *fars, const float *noise,
                           const float *bg_ray, float bg_scalar, uint32_t N, uint32_t T, float density_scale, float thresh,
                           float *sigma, float *trans, float *weights, float *weights_sum, float *depth, float *image,
                           uint32_t c_width, float *ray_sumsq, void *stream);
int foc_fixed_tail_backward(const float *grad_image, const float *grad_ws, const float *grad_depth, const void *c,
                            const float *sigma, const float *trans, const float *weights, const float *nears,
                            const float *fars, const float *noise, const float *bg_ray, float bg_scalar, uint32_t N,
                            uint32_t T, float density_scale, float thresh, void *grad_c, void *grad_h0, uint32_t c_width,
                            const float *grad_sumsq, void *stream);

/* Tail of the occupancy-grid TRAINING path on a ragged sample list (csrc/occtrain.hip; legacy/nerf/renderer.py:300-314):
 *   sigma = density_scale * exp(h[:,0]); rgb = sigmoid(c[:,0:3]) (rounded to fp16); composite_rays_train (raymarching.cu:500-588);
 *   image = raw + (1 - weights_sum) * bg (bg_ray [N,3] or NULL -> bg_scalar); depth = clamp(depth - nears, 0) / (fars - nears).
 * h [M,16] fp16 (density network output), c [M,c_width] fp16 (colour network output, c_width 4 or 16), deltas [M,2], rays [N,3].
 * Outputs [N] / [N,3] fp32: weights_sum, image_raw (the composite's own image: the backward needs it), image, depth.
 * Backward: grad_image [N,3] (of `image`), grad_ws [N] or NULL -> grad_c [M,c_width] fp16 and grad_h0 [M] fp16 (the gradient of
 * h[:,0], trunc_exp's factor applied) — what foc_color_head_backward consumes. EVERY row of both is written (zeros where a ray
 * stopped early, did not fit the list, and behind the last ray: `counter` = the march's counter, counter[0] = samples marched). */
int foc_occ_tail_forward(const void *h, const void *c, uint32_t c_width, const float *deltas, const int32_t *rays,
                         uint32_t M, uint32_t N, float T_thresh, float density_scale, const float *bg_ray, float bg_scalar,
                         const float *nears, const float *fars, float *weights_sum, float *image_raw, float *image,
                         float *depth, void *stream);
int foc_occ_tail_backward(const float *grad_image, const float *grad_ws, const void *h, const void *c, uint32_t c_width,
                          const float *deltas, const int32_t *rays, const int32_t *counter, const float *weights_sum,
                          const float *image_raw, uint32_t M, uint32_t N, float T_thresh, float density_scale,
                          const float *bg_ray, float bg_scalar, void *grad_c, void *grad_h0, void *stream);

/* The whole occupancy-grid TRAINING node as ONE call each way (csrc/occtrain.hip): what legacy/nerf/renderer.py:256-322 (`run_cuda`, training
 * branch, a fixed sample budget) + nerf/network_ff.py:51-75 do between the rays and the image, in the order
 *   forward:  foc_march_rays_train_field -> foc_grid_encode_forward_counted -> foc_ffmlp_forward_planar -> foc_color_head_forward -> foc_occ_tail_forward
 *   backward: foc_occ_tail_backward -> foc_color_head_backward -> foc_ffmlp_backward_planar -> foc_grid_encode_backward_binned[_counted]
 * — the SAME entry points with the same arguments, launched from C instead of from the host language: nine library calls per step cost a
 * Python caller ~0.25 ms of the ~0.8 ms a 4096-ray step takes on the GPU, and the step then depends on the host's speed. Every buffer is the
 * caller's (device memory, sizes as the entry points above document them; `cap` = rows of the sample arrays = the sample budget M);
 * the encoder must be one the binned backward serves (D = 3, C = 2, fp16 table: foc_grid_encode_backward_workspace_bytes != 0).
 * `struct_bytes` = sizeof(FocOccTrainNode): a binding built against another layout is refused. Backward: `precounted` != 0 when the
 * workspace header still holds the count pass of THIS node's forward (nothing else used `grid_workspace` in between). */
typedef struct FocOccTrainNode {
    uint32_t struct_bytes;
    /* rays, occupancy grid, march (foc_march_rays_train_field) */
    uint32_t n_rays, max_steps, cascade, grid_size, cap, pad_align;
    float bound, dt_gamma, min_near;
    const float *rays_o, *rays_d, *aabb, *jitter;
    const uint8_t *bitfield;
    float *nears, *fars, *enc_in, *deltas;
    void *sh_rows;
    int32_t *rays, *counter, *march_scratch;
    /* hash-grid encoder, [L,M,2] planes */
    uint32_t levels, base_resolution, gridtype, interp;
    int32_t align_corners, table_dtype;
    float per_level_scale_log2;
    const void *embeddings;
    const int32_t *offsets, *offsets_host;
    void *planes, *grid_workspace;
    uint64_t grid_workspace_bytes;
    /* density network (FFMLP, planar input) and colour head */
    uint32_t sigma_input_dim, sigma_hidden, sigma_layers, sigma_activation, sigma_output_activation;
    uint32_t color_hidden, color_layers, color_activation, c_width;
    const void *w_sigma, *w_color;
    void *h, *c;
    /* tail */
    float T_thresh, density_scale, bg_scalar;
    const float *bg_ray;
    float *weights_sum, *image_raw, *image, *depth;
    /* backward only */
    int32_t precounted;
    const float *grad_image, *grad_ws;
    void *grad_c, *grad_h0, *grad_h, *grad_planes, *grad_w_color, *grad_w_sigma, *grad_embeddings, *mlp_workspace;
    uint64_t mlp_workspace_bytes;       /* size of mlp_workspace: >= foc_ffmlp_backward_workspace_bytes() of both networks */
} FocOccTrainNode;
int foc_occ_train_forward(const FocOccTrainNode *node, void *stream);
int foc_occ_train_backward(const FocOccTrainNode *node, void *stream);

/* Extension (no reference binding; focnerf_amd/rayorder.py): perm [N] int64 = the order in which a staged render walks a view's rays —
 * tile_h x tile_w pixel tiles when rays_d [N,3] fp32 is a row-major H x W pixel grid (recognised from the directions: W >= 16, H >= 8),
 * else the identity. Found and built on the device, no host round trip. state16: 16 bytes of device scratch. */
int foc_view_tile_order(const float *rays_d, uint32_t N, uint32_t tile_h, uint32_t tile_w, int64_t *perm, void *state16,
                        void *stream);

/* c [M,16] fp16 = colour-net output; rgb = sigmoid(c[:, :3]) (rounded to fp16 like the reference's half
 * sigmoid) where weights > thresh, else 0; image [N,3] = sum w rgb + (1 - sum w) bg. bg_ray [N,3] or NULL
 * (then bg_scalar). */
int foc_fixed_composite_forward(const void *c, const float *weights, const float *bg_ray, float bg_scalar,
                                uint32_t N, uint32_t T, float thresh, float *image, void *stream);
int foc_fixed_composite_backward(const float *grad_image, const void *c, const float *weights,
                                 const float *bg_ray, float bg_scalar, uint32_t N, uint32_t T, float thresh,
                                 void *grad_c, float *grad_w, void *stream);

/* Inference tail of the fixed-step renderer in one pass: sigma [M] fp32, rgb [M,3] fp32 (e.g. from
 * foc_nerf_field_inference) -> weights (alpha * cumprod), image [N,3] = sum w rgb [w > thresh] + (1 - sum w) bg,
 * depth [N], weights_sum [N]; rgb_masked [M,3] (may be NULL) = rgb where w > thresh else 0, the `rgbs` field of
 * nerf/renderer.py:187. ray_block = 64: sigma / rgb stand in the block-interleaved order (16-byte aligned,
 * ceil(N/64)*64*T rows); sigma_raymajor [M] (may be NULL) then receives the densities in ray-major order, the
 * `densities` field. Same bits either way. */
int foc_fixed_render_inference(const float *sigma, const float *rgb, const float *nears, const float *fars,
                               const float *noise, const float *bg_ray, float bg_scalar, uint32_t N, uint32_t T,
                               float density_scale, float thresh, float *image, float *depth, float *weights_sum,
                               float *rgb_masked, uint32_t ray_block, float *sigma_raymajor, void *stream);

/* foc_fixed_render_inference that also (or only: image, depth and weights_sum may then all be NULL) writes the object's
 * per-sample field packed as field4 [N,T,4] fp32 = (sigma, rgb where w > thresh else 0): the (`densities`, `rgbs`) pair
 * of nerf/renderer.py:187 / COMBINED.py:598-600 in the layout foc_combine_select_composite reads. */
int foc_fixed_field_pack(const float *sigma, const float *rgb, const float *nears, const float *fars,
                         const float *noise, const float *bg_ray, float bg_scalar, uint32_t N, uint32_t T,
                         float density_scale, float thresh, float *image, float *depth, float *weights_sum,
                         float *field4, uint32_t ray_block, void *stream);

/* ---------------------------------------------------------------------------
 * Per-sample network glue for callers with arbitrary sample lists (the occupancy-grid paths): the torch
 * expressions of nerf/network_ff.py:51-75 between the sigma network, the SH-encoded direction and the
 * colour network. No reference binding (the reference runs them as ~25 torch kernels per call).
 * ------------------------------------------------------------------------- */

/* h [M,16] fp16 = sigma-net output, dirs [M,3] fp32. sigma [M] fp32 = trunc_exp(h[:,0]) (activation.py:8-13);
 * cin [M,cin_width] fp16 = [SH degree 4 of dir | h[:,1:16] | 0] (cin_width 32, network_ff.py:62-68) or
 * [SH | h[:,1:16] | obj_feat[16] | 0] (cin_width 48, see foc_fixed_head_forward). sigma or cin may be NULL. */
int foc_sample_head_forward(const void *h, const float *dirs, uint64_t M, float *sigma, void *cin,
                            const void *obj_feat, uint32_t cin_width, void *stream);
/* grad_sigma [M] fp32, grad_cin [M,cin_width] fp16 (either may be NULL) -> grad_h [M,16] fp16
 * (column 0: grad_sigma * exp(clamp(h0,-15,15)), activation.py:16-18; columns 1..15: grad_cin[:,16:31]). */
int foc_sample_head_backward(const void *h, const float *grad_sigma, const void *grad_cin, uint64_t M,
                             void *grad_h, uint32_t cin_width, void *stream);
/* Degree-4 real spherical harmonics of dirs [M,3] fp32 -> out [M,16] fp32: the direction encoder the reference imports as
 * encoding.get_encoder('sphere_harmonics') (network_ff.py:43) but does not ship. */
int foc_sh_encode(const float *dirs, uint64_t M, float *out, void *stream);
/* c [M,16] fp16 = colour-net output -> rgb [M,3] fp32 = sigmoid(c[:, :3]) rounded to fp16 (network_ff.py:73). */
int foc_rgb_head_forward(const void *c, uint64_t M, float *rgb, void *stream);
/* grad_rgb [M,3] fp32 -> grad_c [M,16] fp16 (columns 3..15 zero). */
int foc_rgb_head_backward(const void *c, const float *grad_rgb, uint64_t M, void *grad_c, void *stream);

/* ---------------------------------------------------------------------------
 * Occupancy-grid maintenance on the device (csrc/densitygrid.hip). Reference: the torch code of
 * NeRFRenderer.mark_untrained_grid (nerf/renderer.py:356-418) and NeRFRenderer.update_extra_state
 * (nerf/renderer.py:420-508), which the reference runs from Python (no binding). density_grid is
 * [cascade, H^3] fp32 in Morton order, bitfield uint8 [cascade * H^3 / 8]; random numbers come in as
 * arrays so that a call can be replayed by the oracle.
 * ------------------------------------------------------------------------- */

/* poses [B,4,4] fp32 camera-to-world, row-major. count (int32 [cascade,H^3], may be NULL) = cameras that see
 * the cell; density_grid[count == 0] = -1 (renderer.py:416). */
int foc_mark_untrained_grid(const float *poses, uint32_t B, float fx, float fy, float cx, float cy, float bound,
                            uint32_t cascade, uint32_t H, float *density_grid, int32_t *count, void *stream);
/* Query points of the full sweep (the first 16 updates, renderer.py:430-453): xyzs [cascade*H^3,3], cell m of a
 * cascade at row m (Morton order); jitter [cascade*H^3,3] uniform in [0,1) or NULL. */
int foc_grid_cells_xyz(uint32_t cascade, uint32_t H, float bound, const float *jitter, float *xyzs, void *stream);
/* Query points of the steady-state update (renderer.py:476-493): per cascade N uniformly random cells
 * (rand_coords int32 [cascade,N,3] in [0,H)) followed by N cells drawn uniformly from the occupied ones
 * (density_grid > 0; pick k = floor(rand_pick * n_occupied), rand_pick fp32 [cascade,N] in [0,1)).
 * Outputs indices int32 [cascade,2N] (Morton), xyzs [cascade*2N,3] jittered by jitter [cascade*2N,3]. */
uint64_t foc_grid_update_sample_workspace_bytes(uint32_t cascade, uint32_t H);
int foc_grid_update_sample(const float *density_grid, uint32_t cascade, uint32_t H, float bound, uint32_t N,
                           const int32_t *rand_coords, const float *rand_pick, const float *jitter,
                           int32_t *indices, float *xyzs, void *workspace, uint64_t workspace_bytes, void *stream);
/* tmp = -1; tmp[cas, indices] = sigmas * density_scale (largest on duplicates); where density_grid >= 0 and tmp >= 0:
 * density_grid = max(density_grid * decay, tmp); mean_out (device fp32, may be NULL) = mean(clamp(density_grid, 0));
 * bitfield = packbits(density_grid, min(mean, density_thresh))  (renderer.py:428, :493-503). sigmas fp32 [cascade*Mc];
 * indices int32 [cascade*Mc], or NULL when sigmas cover every cell in Morton order (Mc == H^3). No host synchronisation. */
uint64_t foc_grid_update_apply_workspace_bytes(uint32_t cascade, uint32_t H);
int foc_grid_update_apply(float *density_grid, uint32_t cascade, uint32_t H, const float *sigmas, const int32_t *indices,
                          uint32_t Mc, float density_scale, float decay, float density_thresh, uint8_t *bitfield,
                          float *mean_out, void *workspace, uint64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FOCNERF_H */
