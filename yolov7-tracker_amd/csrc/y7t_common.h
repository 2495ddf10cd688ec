// y7t_common.h -- host-side helpers shared by the C-ABI translation units of liby7t.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/y7t.h"

void y7t_set_error(const char* fmt, ...);
void y7t_note_kernel(const char* fmt, ...);   // see y7t_last_kernel()

#define Y7T_HIP_CHECK(expr)                                                                 \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            y7t_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return Y7T_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

#define Y7T_ARG_CHECK(cond)                                                  \
    do {                                                                     \
        if (!(cond)) {                                                       \
            y7t_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
            return Y7T_E_ARG;                                                \
        }                                                                    \
    } while (0)

#define Y7T_LAUNCH_CHECK() Y7T_HIP_CHECK(hipGetLastError())
