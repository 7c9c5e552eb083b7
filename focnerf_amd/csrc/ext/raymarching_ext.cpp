// _raymarching — raymarching/src/bindings.cpp:5-19, raymarching/src/raymarching.h:7-18 (same names, same positional arguments).
// The reference's functions check nothing and dispatch on the scalar type (raymarching.cu:148-156 ...); its Python wrappers cast every
// floating input to fp32 (custom_fwd(cast_inputs=torch.float32), raymarching.py:21) so fp32 is what arrives. Here a tensor that is not a
// contiguous fp32 (int32 for ray tables / indices, uint8 for the bitfield) device tensor raises instead of being reinterpreted.
#include "ext_common.h"

#define RM_F(x) do { FOC_CHECK_CUDA(x); FOC_CHECK_CONTIGUOUS(x); FOC_CHECK_IS_FLOAT(x); } while (0)
#define RM_I(x) do { FOC_CHECK_CUDA(x); FOC_CHECK_CONTIGUOUS(x); FOC_CHECK_IS_INT(x); } while (0)
#define RM_U8(x) do { FOC_CHECK_CUDA(x); FOC_CHECK_CONTIGUOUS(x); TORCH_CHECK((x).scalar_type() == at::ScalarType::Byte, #x " must be a uint8 tensor"); } while (0)

void near_far_from_aabb(const at::Tensor rays_o, const at::Tensor rays_d, const at::Tensor aabb, const uint32_t N, const float min_near, at::Tensor nears, at::Tensor fars) {
    RM_F(rays_o); RM_F(rays_d); RM_F(aabb); RM_F(nears); RM_F(fars);
    foc_ok(foc_near_far_from_aabb(foc_ptr<float>(rays_o), foc_ptr<float>(rays_d), foc_ptr<float>(aabb), N, min_near, foc_ptr<float>(nears), foc_ptr<float>(fars), foc_stream(rays_o)), "near_far_from_aabb");
}
void sph_from_ray(const at::Tensor rays_o, const at::Tensor rays_d, const float radius, const uint32_t N, at::Tensor coords) {
    RM_F(rays_o); RM_F(rays_d); RM_F(coords);
    foc_ok(foc_sph_from_ray(foc_ptr<float>(rays_o), foc_ptr<float>(rays_d), radius, N, foc_ptr<float>(coords), foc_stream(rays_o)), "sph_from_ray");
}
void morton3D(const at::Tensor coords, const uint32_t N, at::Tensor indices) {
    RM_I(coords); RM_I(indices);
    foc_ok(foc_morton3D(foc_ptr<int32_t>(coords), N, foc_ptr<int32_t>(indices), foc_stream(coords)), "morton3D");
}
void morton3D_invert(const at::Tensor indices, const uint32_t N, at::Tensor coords) {
    RM_I(indices); RM_I(coords);
    foc_ok(foc_morton3D_invert(foc_ptr<int32_t>(indices), N, foc_ptr<int32_t>(coords), foc_stream(indices)), "morton3D_invert");
}
void packbits(const at::Tensor grid, const uint32_t N, const float density_thresh, at::Tensor bitfield) {
    RM_F(grid); RM_U8(bitfield);
    foc_ok(foc_packbits(foc_ptr<float>(grid), N, density_thresh, foc_ptr<uint8_t>(bitfield), foc_stream(grid)), "packbits");
}
void march_rays_train(const at::Tensor rays_o, const at::Tensor rays_d, const at::Tensor grid, const float bound, const float dt_gamma, const uint32_t max_steps, const uint32_t N, const uint32_t C, const uint32_t H, const uint32_t M, const at::Tensor nears, const at::Tensor fars, at::Tensor xyzs, at::Tensor dirs, at::Tensor deltas, at::Tensor rays, at::Tensor counter, at::Tensor noises) {
    RM_F(rays_o); RM_F(rays_d); RM_U8(grid); RM_F(nears); RM_F(fars); RM_F(xyzs); RM_F(dirs); RM_F(deltas); RM_I(rays); RM_I(counter); RM_F(noises);
    void *scratch = foc_scratch("march", foc_march_rays_train_scratch_bytes(N, max_steps), rays_o);
    foc_ok(foc_march_rays_train(foc_ptr<float>(rays_o), foc_ptr<float>(rays_d), foc_ptr<uint8_t>(grid), bound, dt_gamma, max_steps, N, C, H, M, foc_ptr<float>(nears), foc_ptr<float>(fars),
                                foc_ptr<float>(xyzs), foc_ptr<float>(dirs), foc_ptr<float>(deltas), foc_ptr<int32_t>(rays), foc_ptr<int32_t>(counter), foc_ptr<float>(noises),
                                reinterpret_cast<int32_t *>(scratch), foc_stream(rays_o)), "march_rays_train");
}
void composite_rays_train_forward(const at::Tensor sigmas, const at::Tensor rgbs, const at::Tensor deltas, const at::Tensor rays, const uint32_t M, const uint32_t N, const float T_thresh, at::Tensor weights_sum, at::Tensor depth, at::Tensor image) {
    RM_F(sigmas); RM_F(rgbs); RM_F(deltas); RM_I(rays); RM_F(weights_sum); RM_F(depth); RM_F(image);
    foc_ok(foc_composite_rays_train_forward(foc_ptr<float>(sigmas), foc_ptr<float>(rgbs), foc_ptr<float>(deltas), foc_ptr<int32_t>(rays), M, N, T_thresh, foc_ptr<float>(weights_sum),
                                            foc_ptr<float>(depth), foc_ptr<float>(image), foc_stream(sigmas)), "composite_rays_train_forward");
}
void composite_rays_train_backward(const at::Tensor grad_weights_sum, const at::Tensor grad_image, const at::Tensor sigmas, const at::Tensor rgbs, const at::Tensor deltas, const at::Tensor rays, const at::Tensor weights_sum, const at::Tensor image, const uint32_t M, const uint32_t N, const float T_thresh, at::Tensor grad_sigmas, at::Tensor grad_rgbs) {
    RM_F(grad_weights_sum); RM_F(grad_image); RM_F(sigmas); RM_F(rgbs); RM_F(deltas); RM_I(rays); RM_F(weights_sum); RM_F(image); RM_F(grad_sigmas); RM_F(grad_rgbs);
    foc_ok(foc_composite_rays_train_backward(foc_ptr<float>(grad_weights_sum), foc_ptr<float>(grad_image), foc_ptr<float>(sigmas), foc_ptr<float>(rgbs), foc_ptr<float>(deltas), foc_ptr<int32_t>(rays),
                                             foc_ptr<float>(weights_sum), foc_ptr<float>(image), M, N, T_thresh, foc_ptr<float>(grad_sigmas), foc_ptr<float>(grad_rgbs), foc_stream(sigmas)),
           "composite_rays_train_backward");
}
void march_rays(const uint32_t n_alive, const uint32_t n_step, const at::Tensor rays_alive, const at::Tensor rays_t, const at::Tensor rays_o, const at::Tensor rays_d, const float bound, const float dt_gamma, const uint32_t max_steps, const uint32_t C, const uint32_t H, const at::Tensor grid, const at::Tensor nears, const at::Tensor fars, at::Tensor xyzs, at::Tensor dirs, at::Tensor deltas, at::Tensor noises) {
    RM_I(rays_alive); RM_F(rays_t); RM_F(rays_o); RM_F(rays_d); RM_U8(grid); RM_F(nears); RM_F(fars); RM_F(xyzs); RM_F(dirs); RM_F(deltas); RM_F(noises);
    foc_ok(foc_march_rays(n_alive, n_step, foc_ptr<int32_t>(rays_alive), foc_ptr<float>(rays_t), foc_ptr<float>(rays_o), foc_ptr<float>(rays_d), bound, dt_gamma, max_steps, C, H,
                          foc_ptr<uint8_t>(grid), foc_ptr<float>(nears), foc_ptr<float>(fars), foc_ptr<float>(xyzs), foc_ptr<float>(dirs), foc_ptr<float>(deltas), foc_ptr<float>(noises),
                          foc_stream(rays_o)), "march_rays");
}
void composite_rays(const uint32_t n_alive, const uint32_t n_step, const float T_thresh, at::Tensor rays_alive, at::Tensor rays_t, at::Tensor sigmas, at::Tensor rgbs, at::Tensor deltas, at::Tensor weights_sum, at::Tensor depth, at::Tensor image) {
    RM_I(rays_alive); RM_F(rays_t); RM_F(sigmas); RM_F(rgbs); RM_F(deltas); RM_F(weights_sum); RM_F(depth); RM_F(image);
    foc_ok(foc_composite_rays(n_alive, n_step, T_thresh, foc_ptr<int32_t>(rays_alive), foc_ptr<float>(rays_t), foc_ptr<float>(sigmas), foc_ptr<float>(rgbs), foc_ptr<float>(deltas),
                              foc_ptr<float>(weights_sum), foc_ptr<float>(depth), foc_ptr<float>(image), foc_stream(sigmas)), "composite_rays");
}

PYBIND11_MODULE(_raymarching, m) {
    m.def("packbits", &packbits, "packbits (HIP, gfx950)");
    m.def("near_far_from_aabb", &near_far_from_aabb, "near_far_from_aabb (HIP, gfx950)");
    m.def("sph_from_ray", &sph_from_ray, "sph_from_ray (HIP, gfx950)");
    m.def("morton3D", &morton3D, "morton3D (HIP, gfx950)");
    m.def("morton3D_invert", &morton3D_invert, "morton3D_invert (HIP, gfx950)");
    m.def("march_rays_train", &march_rays_train, "march_rays_train (HIP, gfx950)");
    m.def("composite_rays_train_forward", &composite_rays_train_forward, "composite_rays_train_forward (HIP, gfx950)");
    m.def("composite_rays_train_backward", &composite_rays_train_backward, "composite_rays_train_backward (HIP, gfx950)");
    m.def("march_rays", &march_rays, "march rays (HIP, gfx950)");
    m.def("composite_rays", &composite_rays, "composite rays (HIP, gfx950)");
}
