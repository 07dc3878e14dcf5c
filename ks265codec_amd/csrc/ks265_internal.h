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
    // device-side error word: pinned, device-mapped host memory that kernels OR error bits into (KS_DEVERR_*); ks265_synchronize
    // reads and clears it after the stream has drained and turns a set bit into KS265_FAIL
    unsigned *err_host = nullptr, *err_dev = nullptr;
    bool capturing = false;                      // between ks265_capture_begin and ks265_capture_end on this context
    int wavefront_spin_limit = 1 << 22;          // ks265_debug_set(KS265_DBG_WAVEFRONT_SPINS): test hook for the timeout path
};
#define KS_DEVERR_WAVEFRONT_TIMEOUT 1u           // intra wavefront: the CTU row above did not make progress in time

// make the context's device the calling thread's current device (the runtime keeps it per thread; setting the device that is already current costs a
// thread-local compare inside the runtime - no caching here: several translation units and direct hipSetDevice calls would defeat it)
static inline void ks_use_device(const ks265_ctx *ctx) { (void)hipSetDevice(ctx->device); }
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
