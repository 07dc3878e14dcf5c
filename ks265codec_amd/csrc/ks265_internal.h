// ks265_internal.h — host-side context shared by the .hip translation units (not part of the C ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/ks265_hip.h"
#include "ks265_dev.h"

struct ks265_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string last_error;
};

// record a HIP error (if any) from the launch just issued; kernels are asynchronous, so this only
// catches launch-configuration errors — execution errors surface at ks265_synchronize()
static inline int ks265_check_launch(ks265_ctx *ctx)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return KS265_OK;
    ctx->last_error = hipGetErrorString(e);
    return KS265_FAIL;
}
static inline int ks265_hip(ks265_ctx *ctx, hipError_t e)
{
    if (e == hipSuccess) return KS265_OK;
    if (ctx) ctx->last_error = hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? KS265_OUTOFMEMORY : KS265_FAIL;
}
