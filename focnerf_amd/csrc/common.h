// common.h — shared helpers for the gfx950 kernels of libfocnerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/focnerf.h"

#define FOC_WAVE 64

// thread-local error message (foc_last_error)
void foc_set_error(const char *fmt, ...);

// Tuning / test switches of the library, ONE table (combine.hip foc_option_table): each is an int with a default, initialised from the
// environment variable of the same name the first time any option is read (library load, in effect) and changeable at run time through
// foc_set_option (tests, A/B runs) — no getenv on any call path.
enum FocOpt {
    FOC_OPT_MLP_BWD_FUSED,        // 1: single-pass fused MLP backward where it applies; 0: the two-kernel form (stored activations)
    FOC_OPT_FIELD_FWD_FUSED,      // 1: training forward of both networks in one kernel (field_fwd.hip); 0: foc_ffmlp_forward_planar + foc_color_head_forward (parity aid)
    FOC_OPT_GB_MERGE_MAX_RES,     // binned grid backward: levels up to this resolution merge runs of equal cells (default 480)
    FOC_OPT_GB_FACTORED,          // 1: 8-byte factored records on the unmerged hashed levels; 0: 12-byte two-corner records everywhere
    FOC_OPT_GB_TAIL_SPLIT,        // scatter: most workgroups a tile of the last partial round is dealt out to (16; 1 = whole tiles only)
    FOC_OPT_GRID_FUSE_SMALL,      // 1: the coarse levels of the forward share a workgroup
    FOC_OPT_GRID_PAIRS,           // 1: 8-byte row-pair loads in the fp16 forward
    FOC_OPT_GRID_FAST,            // 1: the FOC-shape code path of the forward (ge_forward_hash3)
    FOC_OPT_MARCH_SERIAL,         // march_rays_train's walk: -1 by ray count, 1 one ray per lane, 0 one wave per ray
    FOC_OPT_MARCH_RAYS_ROW_MAX,   // march_rays: 16 lanes per ray up to this many live rays (131072; 0 = one ray per lane)
    FOC_OPT_OCC_MARCH_FORM,       // native occupancy loop's march: -1 by burst length, 0 two phases, 1 row, 2 lane, 3 staged
    FOC_OPT_OCC_SAMPLE_MAJOR,     // 1: sample-major sample arrays inside the native occupancy step
    FOC_OPT_OCC_FIELD_PIECE,      // samples per field evaluation inside the native occupancy step (2^23)
    FOC_OPT_COUNT
};
int foc_opt(FocOpt which);

#define FOC_REQUIRE(cond, code, ...)                    \
    do {                                                \
        if (!(cond)) {                                  \
            foc_set_error(__VA_ARGS__);                 \
            return (code);                              \
        }                                               \
    } while (0)

// Call after every launch: hipGetLastError catches bad launch configurations without
// synchronising (so the entry points stay graph-capturable).
#define FOC_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            foc_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return FOC_E_LAUNCH;                                                   \
        }                                                                          \
    } while (0)

// Every entry point runs with the device of ITS call current: a caller holding `cuda:1` tensors while device 0 is current (K resident
// objects on several GPUs of one process, focnerf_amd/checkpoint.py load_objects) gets the occupancy queries, LDS opt-ins, per-device
// statics and — above all — the launches of the right device. Which device that is:
//   * a non-null stream belongs to one device: hipStreamGetDevice;
//   * the NULL stream exists on every device (torch's default stream is the null handle everywhere), so it says nothing: the device
//     is then the one the first pointer argument lives on (hipPointerGetAttributes) — what the Python binding and the extension shims
//     already do on their side by making the tensors' device current before they call in;
//   * neither (null stream, no device pointer): the current device.
// foc_guard_pick is that rule as a pure function (host-only unit test: foc_guard_pick_device).
static inline int foc_guard_pick(bool stream_is_null, int stream_device, int pointer_device, int current_device) {
    if (!stream_is_null && stream_device >= 0) return stream_device;
    if (pointer_device >= 0) return pointer_device;
    return current_device;
}
struct FocDeviceGuard {
    int prev = -1;
    explicit FocDeviceGuard(void *stream, const void *first_pointer = nullptr) {
        static int n_devices = -1;                         // one visible device: nothing to choose (and no pointer query per call)
        if (n_devices < 0 && hipGetDeviceCount(&n_devices) != hipSuccess) { (void)hipGetLastError(); n_devices = 0; }
        if (n_devices == 1) return;
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return;
        int sdev = -1, pdev = -1;
        if (stream != nullptr) {
            if (hipStreamGetDevice(static_cast<hipStream_t>(stream), &sdev) != hipSuccess) { (void)hipGetLastError(); sdev = -1; }
        } else if (first_pointer != nullptr) {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, first_pointer) == hipSuccess && attr.type == hipMemoryTypeDevice) pdev = attr.device;
            else (void)hipGetLastError();                  // host memory, or not a HIP allocation: no opinion
        }
        const int want = foc_guard_pick(stream == nullptr, sdev, pdev, cur);
        if (want != cur && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~FocDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    FocDeviceGuard(const FocDeviceGuard &) = delete;
    FocDeviceGuard &operator=(const FocDeviceGuard &) = delete;
};

static inline uint32_t foc_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Grid for a grid-stride elementwise kernel: enough workgroups to fill 256 CUs x 8,
// capped so tiny problems do not launch empty blocks.
static inline uint32_t foc_grid_1d(uint64_t n, uint32_t block, uint32_t max_blocks = 256 * 8) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (uint32_t)g;
}

// fp32 -> fp16 of a value that was ROUNDED TO fp32 first, as a torch `.half()` of an fp32 tensor is. Without the barrier the
// compiler folds a preceding multiply into v_fma_mixlo_f16 (one rounding of the exact product, even with -ffp-contract=off),
// which differs from the two-step rounding in about one value in 2^13.
__device__ __forceinline__ _Float16 foc_f2h(float v) { asm volatile("" : "+v"(v)); return (_Float16)v; }

// Zero fill as a KERNEL. hipMemsetAsync must not be used in this library: captured into a HIP graph, its node fills with garbage from
// the second replay on (ROCm 7.2 / gfx950: the first 8 KiB header of the binned grid backward came back as 0x4f20_2200_2b4c_3438-like
// words that advance by the memset size per replay, tools/_diag: counts of 5 * 10^11 records and an out-of-bounds read in the reduce).
__global__ static void __launch_bounds__(256) k_foc_zero(uint32_t *__restrict__ p, uint64_t n_words) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256) p[i] = 0u;
}
static inline hipError_t foc_zero_async(void *p, size_t bytes, hipStream_t st) {       // p 4-byte aligned, bytes a multiple of 4
    const uint64_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_foc_zero, dim3((uint32_t)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t *>(p), n);
    return hipGetLastError();
}

// Workgroup barrier that orders LDS traffic only. `__syncthreads()` is a workgroup-scope release/acquire on ALL memory: the compiler
// drains vmcnt before it, so every wave sits at the barrier until its outstanding global stores have been acknowledged by memory and
// its prefetched global loads have landed — a software prefetch issued before a barrier buys nothing. Use where the only data shared
// across the barrier lives in LDS (global stores consumed by a LATER kernel, loads into registers the compiler waits for on use).
__device__ __forceinline__ void foc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave64 reductions / scans via DPP-backed shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive product scan across the 64 lanes
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(v, o, 64);
        if (lane >= o) v *= u;
    }
    return v;
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ int wave_incl_sum_i(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
