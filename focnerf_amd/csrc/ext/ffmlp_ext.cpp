// _ffmlp — ffmlp/src/bindings.cpp, ffmlp/src/ffmlp.h:8-14 (checks: ffmlp.cu:636-642, 750-776).
#include "ext_common.h"

static void check_half(std::initializer_list<const at::Tensor *> ts) {
    for (const at::Tensor *t : ts) { TORCH_CHECK(t->device().is_cuda(), "tensor must be a CUDA tensor"); TORCH_CHECK(t->is_contiguous(), "tensor must be a contiguous tensor");
                                     TORCH_CHECK(t->scalar_type() == at::ScalarType::Half, "tensor must be a Half tensor"); }
}
void ffmlp_forward(const at::Tensor inputs, const at::Tensor weights, const uint32_t B, const uint32_t input_dim, const uint32_t output_dim, const uint32_t hidden_dim, const uint32_t num_layers, const uint32_t activation_, const uint32_t output_activation_, at::Tensor forward_buffer, at::Tensor outputs) {
    check_half({&inputs, &weights, &forward_buffer, &outputs});
    foc_ok(foc_ffmlp_forward(inputs.data_ptr(), weights.data_ptr(), B, input_dim, output_dim, hidden_dim, num_layers, activation_, output_activation_, forward_buffer.data_ptr(), outputs.data_ptr(),
                             foc_stream(inputs)), "ffmlp_forward");
}
void ffmlp_inference(const at::Tensor inputs, const at::Tensor weights, const uint32_t B, const uint32_t input_dim, const uint32_t output_dim, const uint32_t hidden_dim, const uint32_t num_layers, const uint32_t activation_, const uint32_t output_activation_, at::Tensor inference_buffer, at::Tensor outputs) {
    check_half({&inputs, &weights, &outputs});
    foc_ok(foc_ffmlp_inference(inputs.data_ptr(), weights.data_ptr(), B, input_dim, output_dim, hidden_dim, num_layers, activation_, output_activation_, inference_buffer.data_ptr(), outputs.data_ptr(),
                               foc_stream(inputs)), "ffmlp_inference");
}
void ffmlp_backward(const at::Tensor grad, const at::Tensor inputs, const at::Tensor weights, const at::Tensor forward_buffer, const uint32_t B, const uint32_t input_dim, const uint32_t output_dim, const uint32_t hidden_dim, const uint32_t num_layers, const uint32_t activation, const uint32_t output_activation, const bool calc_grad_inputs, at::Tensor backward_buffer, at::Tensor grad_inputs, at::Tensor grad_weights) {
    check_half({&grad, &inputs, &weights, &forward_buffer, &backward_buffer, &grad_inputs, &grad_weights});
    const uint64_t ws_bytes = foc_ffmlp_backward_workspace_bytes(input_dim, hidden_dim, num_layers);
    void *ws = foc_scratch("ffmlp_ws", ws_bytes, grad);
    foc_ok(foc_ffmlp_backward(grad.data_ptr(), inputs.data_ptr(), weights.data_ptr(), forward_buffer.data_ptr(), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                              calc_grad_inputs ? 1 : 0, backward_buffer.data_ptr(), grad_inputs.data_ptr(), grad_weights.data_ptr(), ws, ws_bytes, foc_stream(grad)), "ffmlp_backward");
}
void allocate_splitk(size_t size) { foc_ok(foc_allocate_splitk((uint64_t)size), "allocate_splitk"); }
void free_splitk() { foc_ok(foc_free_splitk(), "free_splitk"); }

PYBIND11_MODULE(_ffmlp, m) {
    m.def("ffmlp_forward", &ffmlp_forward, "ffmlp_forward (HIP, gfx950)");
    m.def("ffmlp_inference", &ffmlp_inference, "ffmlp_inference (HIP, gfx950)");
    m.def("ffmlp_backward", &ffmlp_backward, "ffmlp_backward (HIP, gfx950)");
    m.def("allocate_splitk", &allocate_splitk, "allocate_splitk (HIP, gfx950)");
    m.def("free_splitk", &free_splitk, "free_splitk (HIP, gfx950)");
}
