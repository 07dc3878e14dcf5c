// context.hip — ks265_ctx lifetime, stream adoption, event timing (include/ks265_hip.h §1)
#include "ks265_internal.h"

// one-thread kernel that only exists to be found in a kernel trace (tools/rocpd_stats.py restricts its table to the dispatches
// between the first and the last marker = the timed region of bench.py)
__global__ void ks265_marker_kernel(int id) { (void)id; }

extern "C" {

const char *ks265_version(void) { return "ks265hip 0.1 (gfx950) — pixel-kernel path of libqycodec V2.6.1.3"; }

int ks265_create(ks265_ctx **out, int device) { return ks265_create_prio(out, device, 0); }
/* high_priority != 0: the context's stream gets the device's highest priority - for a host's long, narrow kernels (a key picture's intra wavefront keeps 34 work-groups
 * busy for 19 ms) that must make progress underneath wide ones from other streams */
int ks265_create_prio(ks265_ctx **out, int device, int high_priority)
{
    if (!out) return KS265_POINTER;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return KS265_NO_DEVICE;
    ks265_ctx *c = new ks265_ctx();
    c->device = device;
    int plo = 0, phi = 0;                                            // (numerically lower = higher priority)
    if (hipSetDevice(device) != hipSuccess) { delete c; return KS265_FAIL; }
    if (high_priority && hipDeviceGetStreamPriorityRange(&plo, &phi) == hipSuccess && phi < plo) {
        if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, phi) != hipSuccess) { delete c; return KS265_FAIL; }
    } else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return KS265_FAIL; }
    c->own_stream = true;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { ks265_destroy(c); return KS265_FAIL; }
    if (hipHostMalloc((void **)&c->err_host, sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->err_dev, c->err_host, 0) != hipSuccess) { ks265_destroy(c); return KS265_FAIL; }
    *c->err_host = 0;
    *out = c;
    return KS265_OK;
}

void ks265_destroy(ks265_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->err_host) (void)hipHostFree(c->err_host);
    delete c;
}

int ks265_set_stream(ks265_ctx *c, void *s)
{
    if (!c) return KS265_POINTER;
    if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return KS265_OK;
}

int ks265_synchronize(ks265_ctx *c)
{
    if (!c) return KS265_POINTER;
    ks_use_device(c);
    int r = ks265_hip(c, hipStreamSynchronize(c->stream));
    if (r) return r;
    const unsigned e = __atomic_exchange_n(c->err_host, 0u, __ATOMIC_ACQ_REL);      // kernels OR bits in; report once, then clear
    if (e) {
        c->last_error = (e & KS_DEVERR_WAVEFRONT_TIMEOUT) ? "intra wavefront timeout: a CTU row waited too long for the row above; the key picture is invalid, re-encode it"
                                                          : "device-side error flag set";
        return KS265_FAIL;
    }
    return KS265_OK;
}

/* the device error word without waiting for anything: pipelined hosts call it when a picture's completion event has fired */
int ks265_take_device_error(ks265_ctx *c)
{
    if (!c) return KS265_POINTER;
    const unsigned e = __atomic_exchange_n(c->err_host, 0u, __ATOMIC_ACQ_REL);
    if (!e) return KS265_OK;
    c->last_error = (e & KS_DEVERR_WAVEFRONT_TIMEOUT) ? "intra wavefront timeout: a CTU row waited too long for the row above; the picture is invalid, re-encode it"
                                                      : "device-side error flag set";
    return KS265_FAIL;
}

int ks265_debug_set(ks265_ctx *c, int what, int value)
{
    if (!c) return KS265_POINTER;
    if (what == KS265_DBG_WAVEFRONT_SPINS) { c->wavefront_spin_limit = value < 0 ? (1 << 22) : value; return KS265_OK; }
    return KS265_NOTSUPPORTED;
}

int ks265_marker(ks265_ctx *c, int id)
{
    if (!c) return KS265_POINTER;
    hipLaunchKernelGGL(ks265_marker_kernel, dim3(1), dim3(1), 0, c->stream, id);
    return ks265_hip(c, hipGetLastError());
}

/* device / pinned-host memory and stream-ordered copies for hosts that are plain C (no HIP headers needed above the C ABI) */
int ks265_dev_malloc(ks265_ctx *c, void **dev, size_t bytes)
{
    if (!c || !dev) return KS265_POINTER;
    (void)hipSetDevice(c->device);
    return ks265_hip(c, hipMalloc(dev, bytes ? bytes : 1));
}
int ks265_dev_free(ks265_ctx *c, void *dev)
{
    if (!c) return KS265_POINTER;
    (void)hipSetDevice(c->device);
    return dev ? ks265_hip(c, hipFree(dev)) : KS265_OK;
}
int ks265_host_malloc(ks265_ctx *c, void **host, size_t bytes)
{
    if (!c || !host) return KS265_POINTER;
    (void)hipSetDevice(c->device);
    return ks265_hip(c, hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocDefault));
}
int ks265_host_free(ks265_ctx *c, void *host)
{
    if (!c) return KS265_POINTER;
    return host ? ks265_hip(c, hipHostFree(host)) : KS265_OK;
}
/* the caller's own memory made DMA-able in place (hipHostRegister: the pages are pinned and mapped, the data does not move) - the encoder host uploads straight from the
 * application's picture buffers instead of copying them into pinned memory of its own first */
int ks265_host_register(ks265_ctx *c, void *host, size_t bytes)
{
    if (!c || !host) return KS265_POINTER;
    if (!bytes) return KS265_NOTSUPPORTED;
    (void)hipSetDevice(c->device);
    const hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) (void)hipGetLastError();                       /* (not sticky: the caller falls back to its copying path) */
    return ks265_hip(c, e);
}
int ks265_host_unregister(ks265_ctx *c, void *host)
{
    if (!c || !host) return KS265_POINTER;
    const hipError_t e = hipHostUnregister(host);
    if (e != hipSuccess) (void)hipGetLastError();
    return ks265_hip(c, e);
}
/* host -> device NOW, on no stream of the library's (the runtime's null stream; the contexts' streams are non-blocking and do not order against it): returns when the data is
 * on the device.  For memory of ks265_host_register / ks265_host_malloc this is one DMA */
int ks265_memcpy_h2d_sync(ks265_ctx *c, void *dev, const void *host, size_t bytes)
{
    if (!c || !dev || !host) return KS265_POINTER;
    ks_use_device(c);
    return ks265_hip(c, hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
}
int ks265_memcpy_h2d_async(ks265_ctx *c, void *dev, const void *host, size_t bytes)
{
    if (!c || !dev || !host) return KS265_POINTER;
    ks_use_device(c);
    return ks265_hip(c, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
}
int ks265_memcpy_d2h_async(ks265_ctx *c, void *host, const void *dev, size_t bytes)
{
    if (!c || !dev || !host) return KS265_POINTER;
    ks_use_device(c);
    return ks265_hip(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
}
int ks265_memcpy_d2d_async(ks265_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return KS265_POINTER;
    ks_use_device(c);
    return ks265_hip(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
}
// Device-to-host copy as a kernel with a SMALL grid writing straight into device-mapped pinned host memory: the runtime's own device-to-host copies
// may run as a full-grid shader copy that fills every compute unit with waves waiting on PCIe and stalls whatever the other streams want to launch;
// 32 work-groups keep enough stores in flight to fill the link and leave the machine to the pixel path.
__global__ __launch_bounds__(256) void copy_out_kernel(uint4 *dst, const uint4 *src, size_t n16, unsigned char *dst_tail, const unsigned char *src_tail, int ntail)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
int ks265_copy_out_async(ks265_ctx *c, void *pinned_host, const void *dev, size_t bytes)
{
    if (!c || !dev || !pinned_host) return KS265_POINTER;
    ks_use_device(c);
    if (((uintptr_t)pinned_host | (uintptr_t)dev) & 15) return ks265_hip(c, hipMemcpyAsync(pinned_host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    const size_t n16 = bytes >> 4;
    hipLaunchKernelGGL(copy_out_kernel, dim3(32), dim3(256), 0, c->stream, (uint4 *)pinned_host, (const uint4 *)dev, n16, (unsigned char *)pinned_host + (n16 << 4),
                       (const unsigned char *)dev + (n16 << 4), (int)(bytes & 15));
    return ks265_check_launch(c);
}
int ks265_memset_async(ks265_ctx *c, void *dev, int value, size_t bytes)
{
    if (!c || !dev) return KS265_POINTER;
    ks_use_device(c);
    return ks265_hip(c, hipMemsetAsync(dev, value, bytes, c->stream));
}
/* an event on the context's stream: record now, wait later from any host thread (pipelined hosts: "picture n has left the GPU") */
int ks265_event_create(ks265_ctx *c, void **ev)
{
    if (!c || !ev) return KS265_POINTER;
    (void)hipSetDevice(c->device);
    hipEvent_t e;
    const int r = ks265_hip(c, hipEventCreateWithFlags(&e, getenv("KS265_EVENT_BLOCKING") ? (hipEventDisableTiming | hipEventBlockingSync) : hipEventDisableTiming)   /* a waiting host thread spins inside the runtime: hosts keep ONE thread waiting (blocking-sync events wake up about a millisecond late) */);
    *ev = r ? nullptr : (void *)e;
    return r;
}
int ks265_event_record(ks265_ctx *c, void *ev) { if (!c || !ev) return KS265_POINTER; ks_use_device(c); return ks265_hip(c, hipEventRecord((hipEvent_t)ev, c->stream)); }
int ks265_event_wait(ks265_ctx *c, void *ev) { return (!c || !ev) ? KS265_POINTER : ks265_hip(c, hipEventSynchronize((hipEvent_t)ev)); }
/* has everything in front of the event's last record run?  Never blocks: a host thread that has other work looks again later */
int ks265_event_query(ks265_ctx *c, void *ev, int *done)
{
    if (!c || !ev || !done) return KS265_POINTER;
    const hipError_t r = hipEventQuery((hipEvent_t)ev);
    *done = r == hipSuccess;
    if (r == hipErrorNotReady) { (void)hipGetLastError(); return KS265_OK; }
    return ks265_hip(c, r);
}
/* make everything enqueued on c's stream AFTER this call wait for the event (recorded on another context's stream): the hand-over between the
 * copy-in, compute and copy-out streams of a pipelined host; no host thread blocks */
int ks265_stream_wait_event(ks265_ctx *c, void *ev) { if (!c || !ev) return KS265_POINTER; ks_use_device(c); return ks265_hip(c, hipStreamWaitEvent(c->stream, (hipEvent_t)ev, 0)); }
int ks265_capture_begin(ks265_ctx *c) { if (!c) return KS265_POINTER; ks_use_device(c); const int r = ks265_hip(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed)); c->capturing = r == KS265_OK; return r; }
int ks265_capture_end(ks265_ctx *c, void **graph_exec)
{
    if (!c || !graph_exec) return KS265_POINTER;
    *graph_exec = nullptr;
    hipGraph_t g = nullptr;
    ks_use_device(c);
    c->capturing = false;
    int r = ks265_hip(c, hipStreamEndCapture(c->stream, &g));
    if (r || !g) return r ? r : KS265_FAIL;
    hipGraphExec_t ex = nullptr;
    r = ks265_hip(c, hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    if (!r) *graph_exec = (void *)ex;
    return r;
}
int ks265_graph_launch(ks265_ctx *c, void *graph_exec) { if (!c || !graph_exec) return KS265_POINTER; ks_use_device(c); return ks265_hip(c, hipGraphLaunch((hipGraphExec_t)graph_exec, c->stream)); }
int ks265_graph_destroy(ks265_ctx *c, void *graph_exec) { return (!c || !graph_exec) ? KS265_POINTER : ks265_hip(c, hipGraphExecDestroy((hipGraphExec_t)graph_exec)); }
int ks265_event_destroy(ks265_ctx *c, void *ev) { return (!c || !ev) ? KS265_POINTER : ks265_hip(c, hipEventDestroy((hipEvent_t)ev)); }

const char *ks265_last_error(ks265_ctx *c) { return c ? c->last_error.c_str() : "null context"; }

int ks265_timer_start(ks265_ctx *c)
{
    if (!c) return KS265_POINTER;
    return ks265_hip(c, hipEventRecord(c->ev0, c->stream));
}

int ks265_timer_stop_ms(ks265_ctx *c, float *ms)
{
    if (!c || !ms) return KS265_POINTER;
    int r = ks265_hip(c, hipEventRecord(c->ev1, c->stream));
    if (r) return r;
    r = ks265_hip(c, hipEventSynchronize(c->ev1));
    if (r) return r;
    return ks265_hip(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
}

}  // extern "C"
