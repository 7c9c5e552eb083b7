// occtrain.hip — the tail of the occupancy-grid TRAINING path (legacy/nerf/renderer.py:256-322) on ragged sample lists.
//
// What the caller's torch glue evaluates between the colour network and the loss, as one kernel per direction, one wave per ray:
//     sigma = density_scale * trunc_exp(h[:, 0])                      (network_ff.py:60, activation.py:8-18; legacy renderer :300)
//     rgb   = sigmoid(c[:, 0:3])                                      (network_ff.py:73, a half tensor: rounded to fp16)
//     weights_sum, depth, image = composite_rays_train(sigma, rgb, deltas, rays, T_thresh)     (raymarching.cu:500-588)
//     image = image + (1 - weights_sum) * bg_color                    (:313)
//     depth = clamp(depth - nears, min=0) / (fars - nears)            (:314, no gradient)
// and their derivatives (raymarching.cu:601-693; trunc_exp's g * exp(clamp(x, -15, 15)); the half sigmoid's g (1 - y) y).
// h [M,16] fp16 is the density network's output, c [M,c_width] fp16 the colour network's (c_width 4: rgb logits + one pad column,
// the form foc_color_head_forward / _backward exchange; 16: the padded FFMLP output). Per sample and lane the arithmetic is that of
// k_head_fwd / k_rgb_fwd / k_composite_train_fwd (head.hip, raymarching.hip) in the same order, so the forward values are the
// bits of the three-kernel chain; sigma [M], rgbs [M,3] and their gradients are never stored (4 x 16 B per sample and direction).
//
// The backward writes EVERY row of grad_c and grad_h0: a ray's wave covers the ray's whole slot range (zeros behind the sample at which
// the ray became opaque, zeros for a ray that did not fit the list), and the rows behind the last ray's range are zeroed by spare
// waves — the colour network's backward reads all M rows, and the caller's zero fill of them was a launch of its own.
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float ot_sigmoid_h(float x) { return (float)(_Float16)(1.0f / (1.0f + expf(-x))); }       // head.hip hd_sigmoid_h

struct OtRay { uint32_t index, offset, count; bool fits; };
__device__ __forceinline__ OtRay ot_ray(const int32_t *__restrict__ rays, uint32_t n, uint32_t M) {
    OtRay r;
    r.index = (uint32_t)rays[n * 3]; r.offset = (uint32_t)rays[n * 3 + 1]; r.count = (uint32_t)rays[n * 3 + 2];
    r.fits = r.count != 0u && (uint64_t)r.offset + r.count <= M;           // raymarching.cu:515: empty rays and rays past the list are skipped
    return r;
}

#define OT_PAD_BLOCKS 64u

__global__ void __launch_bounds__(256) k_occ_tail_fwd(const _Float16 *__restrict__ h, const _Float16 *__restrict__ c, uint32_t c_ld,
                                                      const float *__restrict__ deltas, const int32_t *__restrict__ rays, uint32_t M, uint32_t N,
                                                      float T_thresh, float density_scale, const float *__restrict__ bg_ray, float bg_scalar,
                                                      const float *__restrict__ nears, const float *__restrict__ fars,
                                                      float *__restrict__ weights_sum, float *__restrict__ image_raw, float *__restrict__ image,
                                                      float *__restrict__ depth) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const OtRay ry = ot_ray(rays, n, M);
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (ry.fits) {
        float T_carry = 1.0f, t_carry = 0.0f;
        for (uint32_t base = 0; base < ry.count; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < ry.count;
            float sigma = 0, dt0 = 0, dt1 = 0, c0 = 0, c1 = 0, c2 = 0;
            if (valid) {
                const uint64_t s = (uint64_t)ry.offset + i;
                sigma = expf((float)h[s * 16]);                              // k_head_fwd
                if (density_scale != 1.0f) sigma = density_scale * sigma;
                const float2 dl = *reinterpret_cast<const float2 *>(deltas + s * 2);
                dt0 = dl.x; dt1 = dl.y;
                const uint2 raw = *reinterpret_cast<const uint2 *>(c + s * c_ld);
                const _Float16 *cc = reinterpret_cast<const _Float16 *>(&raw);
                c0 = ot_sigmoid_h((float)cc[0]); c1 = ot_sigmoid_h((float)cc[1]); c2 = ot_sigmoid_h((float)cc[2]);      // k_rgb_fwd
            }
            // from here on: k_composite_train_fwd
            const float alpha = valid ? 1.0f - __expf(-sigma * dt0) : 0.0f;
            const float om = 1.0f - alpha;
            const float P = wave_incl_prod(om, (int)lane);
            float Pex = __shfl_up(P, 1, 64);
            if (lane == 0) Pex = 1.0f;
            const float T_before = T_carry * Pex;
            const float T_after = T_carry * P;
            const float tsum = t_carry + wave_incl_sum(dt1, (int)lane);
            const unsigned long long term = __ballot(valid && (T_after < T_thresh));
            const int first = term ? (int)__ffsll((long long)term) - 1 : 64;
            const float w = (valid && (int)lane <= first) ? alpha * T_before : 0.0f;
            r = fmaf(w, c0, r); g = fmaf(w, c1, g); b = fmaf(w, c2, b);
            d = fmaf(w, tsum, d);
            ws += w;
            if (term) break;
            T_carry = __shfl(T_after, 63, 64);
            t_carry = __shfl(tsum, 63, 64);
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); d = wave_sum(d);
    }
    if (lane == 0) {
        const uint32_t k = ry.index;
        weights_sum[k] = ws;
        image_raw[k * 3] = r; image_raw[k * 3 + 1] = g; image_raw[k * 3 + 2] = b;
        const float rest = 1 - ws;
        // `image + rest` for the default white background ((1 - w) * 1 is (1 - w)), `image + rest * bg` otherwise: the caller's two torch forms
        const float b0 = bg_ray ? bg_ray[k * 3] : bg_scalar, b1 = bg_ray ? bg_ray[k * 3 + 1] : bg_scalar, b2 = bg_ray ? bg_ray[k * 3 + 2] : bg_scalar;
        image[k * 3] = r + rest * b0; image[k * 3 + 1] = g + rest * b1; image[k * 3 + 2] = b + rest * b2;
        const float nr = nears[k], dd = d - nr;
        depth[k] = (dd < 0.0f ? 0.0f : dd) / (fars[k] - nr);
    }
}

// grad_image [N,3] (of the FINAL image), grad_ws [N] or NULL -> grad_c [M,c_ld] fp16, grad_h0 [M] fp16 (every row written).
__global__ void __launch_bounds__(256) k_occ_tail_bwd(const float *__restrict__ grad_image, const float *__restrict__ grad_ws,
                                                      const _Float16 *__restrict__ h, const _Float16 *__restrict__ c, uint32_t c_ld,
                                                      const float *__restrict__ deltas, const int32_t *__restrict__ rays, const int32_t *__restrict__ counter,
                                                      const float *__restrict__ weights_sum, const float *__restrict__ image_raw, uint32_t M, uint32_t N,
                                                      float T_thresh, float density_scale, const float *__restrict__ bg_ray, float bg_scalar,
                                                      _Float16 *__restrict__ grad_c, _Float16 *__restrict__ grad_h0) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x >= (N + 3u) / 4u) {
        // spare workgroups: the rows behind the last ray's slot range (the list is padded to a multiple of 128 rows, or sized by mean_count)
        const uint32_t total = (uint32_t)counter[0];
        const uint32_t pb = blockIdx.x - (N + 3u) / 4u;
        for (uint64_t s = (uint64_t)total + pb * 256u + threadIdx.x; s < M; s += (uint64_t)OT_PAD_BLOCKS * 256u) {
            grad_h0[s] = (_Float16)0;
            if (c_ld == 4) *reinterpret_cast<uint2 *>(grad_c + s * 4) = make_uint2(0u, 0u);
            else { *reinterpret_cast<uint4 *>(grad_c + s * 16) = make_uint4(0u, 0u, 0u, 0u); *reinterpret_cast<uint4 *>(grad_c + s * 16 + 8) = make_uint4(0u, 0u, 0u, 0u); }
        }
        return;
    }
    if (n >= N) return;
    const OtRay ry = ot_ray(rays, n, M);
    auto store = [&](uint64_t s, float gs, float q0, float q1, float q2) {
        grad_h0[s] = foc_f2h(gs);
        h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        o[0] = foc_f2h(q0); o[1] = foc_f2h(q1); o[2] = foc_f2h(q2);
        if (c_ld == 4) *reinterpret_cast<uint2 *>(grad_c + s * 4) = *reinterpret_cast<const uint2 *>(&o);
        else { *reinterpret_cast<h8 *>(grad_c + s * 16) = o; *reinterpret_cast<uint4 *>(grad_c + s * 16 + 8) = make_uint4(0u, 0u, 0u, 0u); }
    };
    if (!ry.fits) {                                            // its slots (the part of them that lies inside the list) carry no gradient
        const uint64_t end = min((uint64_t)ry.offset + ry.count, (uint64_t)M);
        for (uint64_t s = (uint64_t)ry.offset + lane; s < end; s += 64) store(s, 0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const uint32_t index = ry.index;
    const float g0 = grad_image[index * 3], g1 = grad_image[index * 3 + 1], g2 = grad_image[index * 3 + 2];
    // image = raw + (1 - ws) bg: the background term hands -(g . bg) to the opacity's gradient (white: -(g0 + g1 + g2), torch's sum over the channel axis)
    const float b0 = bg_ray ? bg_ray[index * 3] : bg_scalar, b1 = bg_ray ? bg_ray[index * 3 + 1] : bg_scalar, b2 = bg_ray ? bg_ray[index * 3 + 2] : bg_scalar;
    const float gws = (grad_ws ? grad_ws[index] : 0.0f) - ((g0 * b0 + g1 * b1) + g2 * b2);
    const float r_final = image_raw[index * 3], g_final = image_raw[index * 3 + 1], b_final = image_raw[index * 3 + 2];
    const float ws_term = gws * (1 - weights_sum[index]);
    float T_carry = 1.0f;
    float r_carry = 0, g_carry = 0, b_carry = 0;
    bool dead = false;                                         // wave-uniform: the ray became opaque in an earlier block of 64
    for (uint32_t base = 0; base < ry.count; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < ry.count;
        const uint64_t s = (uint64_t)ry.offset + (valid ? i : 0);
        if (dead) { if (valid) store(s, 0.0f, 0.0f, 0.0f, 0.0f); continue; }
        float sigma = 0, e0 = 0, dt0 = 0, c0 = 0, c1 = 0, c2 = 0;
        if (valid) {
            const float x = (float)h[s * 16];
            e0 = expf(x);
            sigma = density_scale != 1.0f ? density_scale * e0 : e0;
            dt0 = deltas[s * 2];
            const uint2 raw = *reinterpret_cast<const uint2 *>(c + s * c_ld);
            const _Float16 *cc = reinterpret_cast<const _Float16 *>(&raw);
            c0 = ot_sigmoid_h((float)cc[0]); c1 = ot_sigmoid_h((float)cc[1]); c2 = ot_sigmoid_h((float)cc[2]);
            // trunc_exp's backward factor exp(clamp(x, -15, 15)) (activation.py:16-18)
            const float xc = x < -15.0f ? -15.0f : (x > 15.0f ? 15.0f : x);
            if (xc != x) e0 = expf(xc);
        }
        const float alpha = valid ? 1.0f - __expf(-sigma * dt0) : 0.0f;
        const float om = 1.0f - alpha;
        const float P = wave_incl_prod(om, (int)lane);
        float Pex = __shfl_up(P, 1, 64);
        if (lane == 0) Pex = 1.0f;
        const float T_before = T_carry * Pex;
        const float T_after = T_carry * P;
        const unsigned long long term = __ballot(valid && (T_after < T_thresh));
        const int first = term ? (int)__ffsll((long long)term) - 1 : 64;
        const bool act = valid && (int)lane <= first;
        const float w = act ? alpha * T_before : 0.0f;
        const float r_acc = r_carry + wave_incl_sum(w * c0, (int)lane);
        const float g_acc = g_carry + wave_incl_sum(w * c1, (int)lane);
        const float b_acc = b_carry + wave_incl_sum(w * c2, (int)lane);
        if (act) {
            // k_composite_train_bwd: grad_rgbs = g w; grad_sigmas = dt0 (...)
            float acc = g0 * fmaf(T_after, c0, -(r_final - r_acc));
            acc = fmaf(g1, fmaf(T_after, c1, -(g_final - g_acc)), acc);
            acc = fmaf(g2, fmaf(T_after, c2, -(b_final - b_acc)), acc);
            acc += ws_term;
            float gs = dt0 * acc;
            if (density_scale != 1.0f) gs = density_scale * gs;       // through `density_scale * sigmas`
            // k_rgb_bwd: half(g) (1 - y) y;  k_head_bwd: grad_sigma * exp(clamp(h0))
            const float q0 = (float)(_Float16)(g0 * w), q1 = (float)(_Float16)(g1 * w), q2 = (float)(_Float16)(g2 * w);
            store(s, gs * e0, q0 * (1.0f - c0) * c0, q1 * (1.0f - c1) * c1, q2 * (1.0f - c2) * c2);
        } else if (valid) store(s, 0.0f, 0.0f, 0.0f, 0.0f);
        if (term) { dead = true; continue; }
        T_carry = __shfl(T_after, 63, 64);
        r_carry = __shfl(r_acc, 63, 64); g_carry = __shfl(g_acc, 63, 64); b_carry = __shfl(b_acc, 63, 64);
    }
}

extern "C" {

int foc_occ_tail_forward(const void *h, const void *c, uint32_t c_width, const float *deltas, const int32_t *rays, uint32_t M, uint32_t N,
                         float T_thresh, float density_scale, const float *bg_ray, float bg_scalar, const float *nears, const float *fars,
                         float *weights_sum, float *image_raw, float *image, float *depth, void *stream) {
    FocDeviceGuard foc_guard_(stream, h);
    if (N == 0) return FOC_OK;
    FOC_REQUIRE(c_width == 16 || c_width == 4, FOC_E_INVALID, "occ_tail_forward: c_width must be 16 or 4 (got %u)", c_width);
    FOC_REQUIRE(rays && nears && fars && weights_sum && image_raw && image && depth && (M == 0 || (h && c && deltas)), FOC_E_INVALID, "occ_tail_forward: null pointer");
    hipLaunchKernelGGL(k_occ_tail_fwd, dim3(foc_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)h, (const _Float16 *)c, c_width, deltas, rays,
                       M, N, T_thresh, density_scale, bg_ray, bg_scalar, nears, fars, weights_sum, image_raw, image, depth);
    FOC_CHECK_LAUNCH("occ_tail_forward");
    return FOC_OK;
}

int foc_occ_tail_backward(const float *grad_image, const float *grad_ws, const void *h, const void *c, uint32_t c_width, const float *deltas,
                          const int32_t *rays, const int32_t *counter, const float *weights_sum, const float *image_raw, uint32_t M, uint32_t N,
                          float T_thresh, float density_scale, const float *bg_ray, float bg_scalar, void *grad_c, void *grad_h0, void *stream) {
    FocDeviceGuard foc_guard_(stream, grad_image);
    if (N == 0 || M == 0) return FOC_OK;
    FOC_REQUIRE(c_width == 16 || c_width == 4, FOC_E_INVALID, "occ_tail_backward: c_width must be 16 or 4 (got %u)", c_width);
    FOC_REQUIRE(grad_image && h && c && deltas && rays && counter && weights_sum && image_raw && grad_c && grad_h0, FOC_E_INVALID, "occ_tail_backward: null pointer");
    hipLaunchKernelGGL(k_occ_tail_bwd, dim3(foc_div_up(N, 4) + OT_PAD_BLOCKS), dim3(256), 0, (hipStream_t)stream, grad_image, grad_ws, (const _Float16 *)h,
                       (const _Float16 *)c, c_width, deltas, rays, counter, weights_sum, image_raw, M, N, T_thresh, density_scale, bg_ray, bg_scalar,
                       (_Float16 *)grad_c, (_Float16 *)grad_h0);
    FOC_CHECK_LAUNCH("occ_tail_backward");
    return FOC_OK;
}

// ---------------------------------------------------------------- the node as one call each way
// Sequencing only: each step is the public entry point a caller would have called itself, with its own argument checks and device guard.
static int ot_check_node(const FocOccTrainNode *n, const char *who) {
    FOC_REQUIRE(n != nullptr, FOC_E_INVALID, "%s: null node", who);
    FOC_REQUIRE(n->struct_bytes == (uint32_t)sizeof(FocOccTrainNode), FOC_E_INVALID, "%s: node of %u bytes, this library's FocOccTrainNode has %zu", who,
                n->struct_bytes, sizeof(FocOccTrainNode));
    FOC_REQUIRE(n->cap > 0 && n->n_rays > 0, FOC_E_INVALID, "%s: empty node (cap %u, rays %u): nothing to sequence, call nothing", who, n->cap, n->n_rays);
    FOC_REQUIRE(n->grid_workspace && n->grid_workspace_bytes && n->offsets_host, FOC_E_INVALID, "%s: the binned encoder backward's workspace is required", who);
    return FOC_OK;
}

int foc_occ_train_forward(const FocOccTrainNode *n, void *stream) {
    int rc = ot_check_node(n, "occ_train_forward");
    if (rc != FOC_OK) return rc;
    const uint32_t M = n->cap;
    rc = foc_march_rays_train_field(n->rays_o, n->rays_d, n->bitfield, n->bound, n->dt_gamma, n->max_steps, n->n_rays, n->cascade, n->grid_size, M, n->nears, n->fars,
                                    n->enc_in, n->sh_rows, n->deltas, n->rays, n->counter, n->jitter, n->march_scratch, n->pad_align, n->aabb, n->min_near, stream);
    if (rc != FOC_OK) return rc;
    rc = foc_grid_encode_forward_counted(n->enc_in, n->embeddings, n->offsets, n->planes, M, 3, 2, n->levels, n->per_level_scale_log2, n->base_resolution, n->gridtype,
                                         n->align_corners, n->interp, n->table_dtype, n->offsets_host, n->grid_workspace, n->grid_workspace_bytes, stream);
    if (rc != FOC_OK) return rc;
    // both networks in one kernel when the shapes are FOC's (csrc/field_fwd.hip: bit for bit the two calls below)
    const uint32_t lk = n->sigma_layers * 10 + n->color_layers;
    if (n->sigma_input_dim == 32 && n->sigma_hidden == 64 && n->color_hidden == 64 && (lk == 22 || lk == 23 || lk == 33) &&
        n->sigma_activation == n->color_activation && (n->sigma_activation == 0 || n->sigma_activation == 6) && n->sigma_output_activation == 6 &&
        foc_opt(FOC_OPT_FIELD_FWD_FUSED)) {
        rc = foc_field_forward_train(n->planes, n->w_sigma, n->sigma_layers, n->sh_rows, 1, n->w_color, n->color_layers, 64, n->sigma_activation, M, n->h, n->c,
                                     n->c_width, nullptr, stream);
        if (rc != FOC_OK) return rc;
    } else {
        rc = foc_ffmlp_forward_planar(n->planes, n->w_sigma, M, n->sigma_input_dim, 16, n->sigma_hidden, n->sigma_layers, n->sigma_activation, n->sigma_output_activation,
                                      n->h, stream);
        if (rc != FOC_OK) return rc;
        rc = foc_color_head_forward(n->h, n->sh_rows, 1, n->w_color, M, n->color_hidden, n->color_layers, n->color_activation, n->c, n->c_width, nullptr, stream);
        if (rc != FOC_OK) return rc;
    }
    return foc_occ_tail_forward(n->h, n->c, n->c_width, n->deltas, n->rays, M, n->n_rays, n->T_thresh, n->density_scale, n->bg_ray, n->bg_scalar, n->nears, n->fars,
                                n->weights_sum, n->image_raw, n->image, n->depth, stream);
}

int foc_occ_train_backward(const FocOccTrainNode *n, void *stream) {
    int rc = ot_check_node(n, "occ_train_backward");
    if (rc != FOC_OK) return rc;
    const uint32_t M = n->cap;
    rc = foc_occ_tail_backward(n->grad_image, n->grad_ws, n->h, n->c, n->c_width, n->deltas, n->rays, n->counter, n->weights_sum, n->image_raw, M, n->n_rays, n->T_thresh,
                               n->density_scale, n->bg_ray, n->bg_scalar, n->grad_c, n->grad_h0, stream);
    if (rc != FOC_OK) return rc;
    rc = foc_color_head_backward(n->grad_c, n->h, n->sh_rows, 1, n->grad_h0, n->w_color, M, n->color_hidden, n->color_layers, n->color_activation, n->grad_h,
                                 n->grad_w_color, n->mlp_workspace, n->mlp_workspace_bytes, n->c_width, nullptr, nullptr, stream);
    if (rc != FOC_OK) return rc;
    rc = foc_ffmlp_backward_planar(n->grad_h, n->planes, n->w_sigma, M, n->sigma_input_dim, 16, n->sigma_hidden, n->sigma_layers, n->sigma_activation,
                                   n->sigma_output_activation, 1, n->grad_planes, n->grad_w_sigma, n->mlp_workspace, n->mlp_workspace_bytes, stream);
    if (rc != FOC_OK) return rc;
    return (n->precounted ? foc_grid_encode_backward_binned_counted : foc_grid_encode_backward_binned)(
        n->grad_planes, n->enc_in, n->embeddings, n->offsets, n->grad_embeddings, M, 3, 2, n->levels, n->per_level_scale_log2, n->base_resolution, nullptr, nullptr,
        n->gridtype, n->align_corners, n->interp, n->table_dtype, 0, n->offsets_host, n->grid_workspace, n->grid_workspace_bytes, stream);
}

} // extern "C"
