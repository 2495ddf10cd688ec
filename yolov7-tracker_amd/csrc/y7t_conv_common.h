// y7t_conv_common.h -- vector types and the activation shared by the convolution kernels
#pragma once
#include "y7t_det.h"
#include <type_traits>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(16))) float floatx16;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// compile-time activation: the epilogues branch ONCE on p.act and run a specialised body (a per-element run-time switch cost ~3
// scalar branches + hazard nops per value: 40 % of the epilogue's instructions)
template <int ACT>
__device__ __forceinline__ float act_t(float v) {
    if (ACT == Y7T_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
    if (ACT == Y7T_ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
    return v;
}

// run `body(std::integral_constant<int, ACT>)` for the run-time activation code
template <typename F>
__device__ __forceinline__ void act_dispatch(int act, F&& body) {
    if (act == Y7T_ACT_SILU) body(std::integral_constant<int, Y7T_ACT_SILU>{});
    else if (act == Y7T_ACT_LEAKY) body(std::integral_constant<int, Y7T_ACT_LEAKY>{});
    else body(std::integral_constant<int, Y7T_ACT_NONE>{});
}

__device__ __forceinline__ float act_fn(float v, int act) {
    // SiLU = v * sigmoid(v) with the hardware exp2 / rcp (1 ulp each; the result is rounded to fp16 anyway): the IEEE
    // division + expf of the naive form made the epilogue's VALU work 27 % of the whole forward (Y7T_CONV_ABLATE=8)
    if (act == Y7T_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
    if (act == Y7T_ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
    return v;
}
