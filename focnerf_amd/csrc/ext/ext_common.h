// Shared by the four torch-extension shims (_raymarching, _gridencoder, _freqencoder, _ffmlp): the reference's pybind11 modules,
// name for name and argument for argument (raymarching/src/bindings.cpp, gridencoder/src/bindings.cpp, freqencoder/src/bindings.cpp,
// ffmlp/src/bindings.cpp), as thin host-only wrappers over the C ABI of libfocnerf_hip.so (include/focnerf.h). Each converts
// at::Tensor -> raw device pointer, takes torch's CURRENT HIP stream of the tensor's device (the reference launches on the legacy
// default stream) and turns a non-zero return into TORCH_CHECK(false, foc_last_error()) — a Python RuntimeError, like the reference's
// own checks. No kernel code lives here: these files compile with the host compiler alone.
#pragma once
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>
#include <map>
#include <tuple>
#include "../../../include/focnerf.h"

#define FOC_CHECK_CUDA(x) TORCH_CHECK((x).device().is_cuda(), #x " must be a CUDA tensor")
#define FOC_CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x " must be a contiguous tensor")
#define FOC_CHECK_IS_INT(x) TORCH_CHECK((x).scalar_type() == at::ScalarType::Int, #x " must be an int tensor")
#define FOC_CHECK_IS_FLOATING(x) TORCH_CHECK((x).scalar_type() == at::ScalarType::Float || (x).scalar_type() == at::ScalarType::Half, #x " must be a floating tensor")
#define FOC_CHECK_IS_HALF(x) TORCH_CHECK((x).scalar_type() == at::ScalarType::Half, #x " must be a Half tensor")
#define FOC_CHECK_IS_FLOAT(x) TORCH_CHECK((x).scalar_type() == at::ScalarType::Float, #x " must be a float32 tensor")

static inline void *foc_stream_handle(const at::Tensor &t) { return (void *)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
// The stream argument of a C-ABI call: torch's current stream on the TENSOR's device, and — for as long as the temporary lives, i.e. until
// the call it is an argument of has returned — that device made current. torch's default stream is the null handle on every device, so
// the library cannot learn the device from the handle (csrc/common.h FocDeviceGuard): cuda:1 tensors on their default stream while
// cuda:0 is current would otherwise launch on device 0.
struct FocStream {
    c10::DeviceGuard guard;          // the generic guard: ROCm builds of torch register their HIP guard under the "cuda" device type
    void *st;
    explicit FocStream(const at::Tensor &t) : guard(t.device()), st(foc_stream_handle(t)) {}
    operator void *() const { return st; }
};
static inline FocStream foc_stream(const at::Tensor &t) { return FocStream(t); }
static inline void foc_ok(int rc, const char *what) { TORCH_CHECK(rc == 0, "focnerf_amd ", what, ": ", foc_last_error(), " (code ", rc, ")"); }
static inline int foc_dtype(const at::Tensor &t) {
    if (t.scalar_type() == at::ScalarType::Float) return FOC_F32;
    TORCH_CHECK(t.scalar_type() == at::ScalarType::Half, "focnerf_amd: float32 or float16 tensor expected");
    return FOC_F16;
}
template <typename T> static inline T *foc_ptr(const at::Tensor &t) { return reinterpret_cast<T *>(t.data_ptr()); }
template <typename T> static inline T *foc_optr(const at::optional<at::Tensor> &t) { return t.has_value() ? reinterpret_cast<T *>(t->data_ptr()) : nullptr; }

// Grow-only device scratch per (purpose, device, stream), owned by the extension module for the life of the process (the reference's
// _ffmlp keeps its split-K workspace the same way, cutlass_matmul.h:335-363). Superseded buffers are kept: a captured graph may still
// hold their address.
static inline void *foc_scratch(const char *key, uint64_t bytes, const at::Tensor &like) {
    static std::map<std::tuple<std::string, int, void *>, at::Tensor> bufs;
    static std::vector<at::Tensor> retired;
    auto k = std::make_tuple(std::string(key), (int)like.device().index(), foc_stream_handle(like));
    auto it = bufs.find(k);
    if (it == bufs.end() || (uint64_t)it->second.numel() < bytes) {
        if (it != bufs.end()) retired.push_back(it->second);
        bufs[k] = at::empty({(int64_t)(bytes < 256 ? 256 : bytes)}, like.options().dtype(at::kByte));
        it = bufs.find(k);
    }
    return it->second.data_ptr();
}
