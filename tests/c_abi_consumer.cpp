// A consumer of the C ABI that is NOT Python and knows nothing of torch: device memory from the HIP runtime API, raw pointers and sizes into
// include/focnerf.h, the NULL stream. Built and run by tests/test_abi.py (mode "cpu": no device is touched) and tests/test_gpu_edge_cases.py
// (mode "gpu": three entry points on seeded inputs against the C oracle linked next to it, bit for bit). Test infrastructure, like the oracle.
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I<repo>/include c_abi_consumer.cpp -o consumer \
//       -L<repo>/focnerf_amd -lfocnerf_hip -L<repo>/oracle/_build -loracle -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,...
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "focnerf.h"

extern "C" {
void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near, float *nears, float *fars);
void orc_morton3D(const int32_t *coords, uint32_t N, int32_t *indices);
void orc_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield);
}

#define REQUIRE(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)
#define HIP_OK(call) do { hipError_t e_ = (call); REQUIRE(e_ == hipSuccess, "%s -> %s", #call, hipGetErrorString(e_)); } while (0)

static uint32_t lcg_state = 12345u;
static float unit() { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)(lcg_state >> 8) * (1.0f / 16777216.0f); }

template <typename T> static int to_device(const std::vector<T> &h, T **d) {
    HIP_OK(hipMalloc((void **)d, h.size() * sizeof(T)));
    HIP_OK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int host_checks() {
    REQUIRE(foc_abi_version() > 0, "abi version %d", foc_abi_version());
    REQUIRE(foc_near_far_from_aabb(nullptr, nullptr, nullptr, 4, 0.2f, nullptr, nullptr, nullptr) != 0, "null pointers were accepted");
    REQUIRE(strstr(foc_last_error(), "null pointer") != nullptr, "message: %s", foc_last_error());
    REQUIRE(foc_occ_train_forward(nullptr, nullptr) != 0 && strstr(foc_last_error(), "null node") != nullptr, "message: %s", foc_last_error());
    FocOccTrainNode node;
    memset(&node, 0, sizeof node);
    node.struct_bytes = (uint32_t)sizeof node;                       // the right size, but nothing to do: refused on the host
    REQUIRE(foc_occ_train_backward(&node, nullptr) != 0 && strstr(foc_last_error(), "empty node") != nullptr, "message: %s", foc_last_error());
    int v = -12345;
    REQUIRE(foc_get_option("FOC_GB_TAIL_SPLIT", &v) == 0 && v != -12345, "option table");
    const int before = v;
    REQUIRE(foc_set_option("FOC_GB_TAIL_SPLIT", 2) == 0 && foc_get_option("FOC_GB_TAIL_SPLIT", &v) == 0 && v == 2, "set_option");
    REQUIRE(foc_set_option("FOC_GB_TAIL_SPLIT", before) == 0, "restore");
    REQUIRE(foc_set_option("FOC_NO_SUCH_OPTION", 1) != 0, "an unknown option was accepted");
    return 0;
}

static int device_checks() {
    int n_dev = 0;
    HIP_OK(hipGetDeviceCount(&n_dev));
    REQUIRE(n_dev >= 1, "no device");
    // ---- rays against a box: hits, misses, origins inside (near = min_near), axis-parallel directions (1 / 0 = inf on one axis)
    const uint32_t N = 5003;
    std::vector<float> o(N * 3), d(N * 3), aabb = {-1.f, -1.f, -1.f, 1.f, 1.f, 1.f};
    for (uint32_t n = 0; n < N; n++) {
        for (int k = 0; k < 3; k++) { o[n * 3 + k] = (unit() * 2.f - 1.f) * (n % 3 ? 3.f : 0.9f); d[n * 3 + k] = unit() * 2.f - 1.f; }
        if (n % 97 == 0) d[n * 3 + n % 3] = 0.0f;
    }
    float *d_o, *d_d, *d_aabb, *d_near, *d_far;
    if (to_device(o, &d_o) || to_device(d, &d_d) || to_device(aabb, &d_aabb)) return 1;
    HIP_OK(hipMalloc((void **)&d_near, N * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_far, N * sizeof(float)));
    REQUIRE(foc_near_far_from_aabb(d_o, d_d, d_aabb, N, 0.2f, d_near, d_far, nullptr) == 0, "near_far_from_aabb: %s", foc_last_error());
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> near(N), far(N), near_ref(N), far_ref(N);
    HIP_OK(hipMemcpy(near.data(), d_near, N * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(far.data(), d_far, N * sizeof(float), hipMemcpyDeviceToHost));
    orc_near_far_from_aabb(o.data(), d.data(), aabb.data(), N, 0.2f, near_ref.data(), far_ref.data());
    uint32_t hits = 0;
    for (uint32_t n = 0; n < N; n++) {
        REQUIRE(memcmp(&near[n], &near_ref[n], 4) == 0 && memcmp(&far[n], &far_ref[n], 4) == 0, "ray %u: near %.9g / %.9g far %.9g / %.9g", n, near[n], near_ref[n], far[n], far_ref[n]);
        hits += far[n] < 1e30f;
    }
    REQUIRE(hits > N / 4 && hits < N, "%u of %u rays hit the box", hits, N);
    // ---- Morton codes of 10-bit cell coordinates
    const uint32_t M = 4099;
    std::vector<int32_t> coords(M * 3), idx(M), idx_ref(M);
    for (auto &c : coords) c = (int32_t)(unit() * 1024.f) & 1023;
    int32_t *d_coords, *d_idx;
    if (to_device(coords, &d_coords)) return 1;
    HIP_OK(hipMalloc((void **)&d_idx, M * sizeof(int32_t)));
    REQUIRE(foc_morton3D(d_coords, M, d_idx, nullptr) == 0, "morton3D: %s", foc_last_error());
    HIP_OK(hipMemcpy(idx.data(), d_idx, M * sizeof(int32_t), hipMemcpyDeviceToHost));      // a blocking copy on the NULL stream: ordered behind the kernel
    orc_morton3D(coords.data(), M, idx_ref.data());
    REQUIRE(idx == idx_ref, "morton3D differs");
    // ---- density grid -> bitfield: N counts the BYTES written, eight cells each (raymarching.cu kernel_packbits); values on both sides of the threshold
    const uint32_t G = 8 * 1237;
    std::vector<float> grid(G);
    for (auto &g : grid) g = unit() < 0.3f ? -1.0f : unit() * 0.02f;
    std::vector<uint8_t> bits(G / 8), bits_ref(G / 8);
    float *d_grid; uint8_t *d_bits;
    if (to_device(grid, &d_grid)) return 1;
    HIP_OK(hipMalloc((void **)&d_bits, G / 8));
    REQUIRE(foc_packbits(d_grid, G / 8, 0.01f, d_bits, nullptr) == 0, "packbits: %s", foc_last_error());
    HIP_OK(hipMemcpy(bits.data(), d_bits, G / 8, hipMemcpyDeviceToHost));
    orc_packbits(grid.data(), G / 8, 0.01f, bits_ref.data());
    REQUIRE(bits == bits_ref, "packbits differs");
    // ---- an error on the device path leaves a message and a non-zero code, and the library goes on working
    REQUIRE(foc_morton3D(nullptr, M, d_idx, nullptr) != 0, "null coords accepted");
    REQUIRE(foc_morton3D(d_coords, M, d_idx, nullptr) == 0, "the call after an error: %s", foc_last_error());
    for (void *p : {(void *)d_o, (void *)d_d, (void *)d_aabb, (void *)d_near, (void *)d_far, (void *)d_coords, (void *)d_idx, (void *)d_grid, (void *)d_bits}) (void)hipFree(p);
    return 0;
}

int main(int argc, char **argv) {
    const bool gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    if (host_checks()) return 1;
    if (gpu && device_checks()) return 1;
    printf("C_ABI_CONSUMER_OK %s abi %d\n", gpu ? "gpu" : "cpu", foc_abi_version());
    return 0;
}
