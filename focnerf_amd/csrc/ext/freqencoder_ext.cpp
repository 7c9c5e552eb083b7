// _freqencoder — freqencoder/src/bindings.cpp, freqencoder/src/freqencoder.h:7,10 (checks: freqencoder.cu:98-105, 114-124).
#include "ext_common.h"

void freq_encode_forward(at::Tensor inputs, const uint32_t B, const uint32_t D, const uint32_t deg, const uint32_t C, at::Tensor outputs) {
    FOC_CHECK_CUDA(inputs); FOC_CHECK_CUDA(outputs); FOC_CHECK_CONTIGUOUS(inputs); FOC_CHECK_CONTIGUOUS(outputs); FOC_CHECK_IS_FLOAT(inputs); FOC_CHECK_IS_FLOAT(outputs);
    foc_ok(foc_freq_encode_forward(foc_ptr<float>(inputs), B, D, deg, C, foc_ptr<float>(outputs), foc_stream(inputs)), "freq_encode_forward");
}
void freq_encode_backward(at::Tensor grad, at::Tensor outputs, const uint32_t B, const uint32_t D, const uint32_t deg, const uint32_t C, at::Tensor grad_inputs) {
    FOC_CHECK_CUDA(grad); FOC_CHECK_CUDA(outputs); FOC_CHECK_CUDA(grad_inputs); FOC_CHECK_CONTIGUOUS(grad); FOC_CHECK_CONTIGUOUS(outputs); FOC_CHECK_CONTIGUOUS(grad_inputs);
    FOC_CHECK_IS_FLOAT(grad); FOC_CHECK_IS_FLOAT(outputs); FOC_CHECK_IS_FLOAT(grad_inputs);
    foc_ok(foc_freq_encode_backward(foc_ptr<float>(grad), foc_ptr<float>(outputs), B, D, deg, C, foc_ptr<float>(grad_inputs), foc_stream(grad)), "freq_encode_backward");
}

PYBIND11_MODULE(_freqencoder, m) {
    m.def("freq_encode_forward", &freq_encode_forward, "freq encode forward (HIP, gfx950)");
    m.def("freq_encode_backward", &freq_encode_backward, "freq encode backward (HIP, gfx950)");
}
