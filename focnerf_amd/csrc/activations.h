// activations.h — the hidden activations of the fused MLP (ffmlp/src/utils.h:424-589; codes of ffmlp/ffmlp.py:87-95).
//   0 relu   1 exponential   2 sine   3 sigmoid   4 squareplus   5 softplus   6 none
// Forward (warp_activation): the layer's sum, ROUNDED TO HALF (the reference's fragment type), goes through the function in fp32 and is
// rounded to half once more; ReLU and None act on the half itself. Backward (warp_activation_backward): the gradient, a half, times a
// factor of the POST-activation forward value — the factor formed as the reference forms it (sigmoid: forward * (1 - forward) in half
// arithmetic; squareplus / softplus: in fp32, rounded to half) and the product a half multiply. Sine has no backward in the reference
// (it would need the pre-activations, which are not stored): the function returns without touching the fragment, so the gradient passes
// through unchanged — reproduced here, not corrected. K_ACT = 10 (utils.h:41).
#pragma once
#include "common.h"

#define FOC_ACT_RELU 0
#define FOC_ACT_EXP 1
#define FOC_ACT_SINE 2
#define FOC_ACT_SIGMOID 3
#define FOC_ACT_SQUAREPLUS 4
#define FOC_ACT_SOFTPLUS 5
#define FOC_ACT_NONE 6
#define FOC_K_ACT 10.0f

__device__ __forceinline__ _Float16 foc_act_forward(_Float16 h, int act) {
    const float x = (float)h;
    switch (act) {
        case FOC_ACT_RELU: return x > 0.0f ? h : (_Float16)0;
        case FOC_ACT_EXP: return foc_f2h(expf(x));
        case FOC_ACT_SINE: return foc_f2h(sinf(x));
        case FOC_ACT_SIGMOID: return foc_f2h(1.0f / (1.0f + expf(-x)));
        case FOC_ACT_SQUAREPLUS: { const float s = x * FOC_K_ACT; return foc_f2h(0.5f * (s + sqrtf(s * s + 4.0f)) / FOC_K_ACT); }
        case FOC_ACT_SOFTPLUS: return foc_f2h(logf(expf(x * FOC_K_ACT) + 1.0f) / FOC_K_ACT);
        default: return h;
    }
}

// a * b in half arithmetic (one rounding of the exact product)
__device__ __forceinline__ _Float16 foc_hmul(_Float16 a, _Float16 b) { return foc_f2h((float)a * (float)b); }

__device__ __forceinline__ _Float16 foc_act_backward(_Float16 g, _Float16 fwd, int act) {
    const float f = (float)fwd;
    switch (act) {
        case FOC_ACT_RELU: return f > 0.0f ? g : (_Float16)0;
        case FOC_ACT_EXP: return foc_hmul(g, fwd);
        case FOC_ACT_SINE: return g;
        case FOC_ACT_SIGMOID: return foc_hmul(g, foc_hmul(fwd, foc_f2h(1.0f - f)));
        case FOC_ACT_SQUAREPLUS: { const float y = f * FOC_K_ACT; return foc_hmul(g, foc_f2h(y * y / (y * y + 1.0f))); }
        case FOC_ACT_SOFTPLUS: return foc_hmul(g, foc_f2h(1.0f - expf(-f * FOC_K_ACT)));
        default: return g;
    }
}
