// api_misc.hip -- small C-ABI utilities.
#include <string.h>
#include "plf_common.h"

extern "C" const char *plf_version(void) { return "plf 0.1 (gfx950)"; }

extern "C" const char *plf_status_string(int status)
{
    switch (status) {
        case PLF_OK: return "ok";
        case PLF_E_EMPTY: return "empty input";
        case PLF_E_BADARG: return "bad argument / unsupported size";
        case PLF_E_CAPACITY: return "output capacity too small";
        case PLF_E_HIP: return "HIP runtime error";
        case PLF_E_NOMEM: return "out of memory";
        case PLF_E_RECTS: return "more LSD rectangles than the line handle holds";
        case PLF_W_TRUNCATED: return "warning: a frame ran out of its time budget (max_ms)";
        case PLF_W_SLOW: return "warning: the call took far longer than the handle's recent calls (outputs complete)";
        default: return "unknown";
    }
}

extern "C" int plf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ORBmatcher::DescriptorDistance (include/ORBmatcher.h:44, so@0x79d20; Thirdparty/DBoW2/DBoW2/FORB.cpp:82-102):
// 256-bit Hamming distance of two descriptors in HOST memory (scalar utility used by host-side callers).
extern "C" int plf_hamming256(const uint8_t *a, const uint8_t *b)
{
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

// ---- device memory helpers for host code that does not link the HIP runtime itself (the C++ adapters of include/plf.hpp move the
// Frame / MapPoint members the matchers read into HBM with these).  Plain wrappers: no state, no fallback.
extern "C" int plf_device_alloc(int32_t device, size_t bytes, void **out)
{
    if (!out) return PLF_E_BADARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return PLF_E_HIP; }
    if (device < 0 || device >= n) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(device));
    if (hipMalloc(out, bytes ? bytes : 256) != hipSuccess) { (void)hipGetLastError(); return PLF_E_NOMEM; }
    return PLF_OK;
}

extern "C" void plf_device_free(void *p) { if (p) (void)hipFree(p); }

// stream == NULL means "synchronous": the handles run on their own NON-BLOCKING streams, which the legacy null stream does not order with, so a
// null-stream upload must be complete before the next enqueue and a null-stream download must wait for everything the device is still running.
extern "C" int plf_upload(void *dst_device, const void *src_host, size_t bytes, void *stream)
{
    if (bytes == 0) return PLF_OK;
    if (!dst_device || !src_host) return PLF_E_BADARG;
    if (!stream) { PLF_HIP_TRY(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice)); return PLF_OK; }
    PLF_HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PLF_OK;
}

extern "C" int plf_download(void *dst_host, const void *src_device, size_t bytes, void *stream)
{
    if (bytes == 0) return PLF_OK;
    if (!dst_host || !src_device) return PLF_E_BADARG;
    if (!stream) {
        PLF_HIP_TRY(hipDeviceSynchronize());
        PLF_HIP_TRY(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
        return PLF_OK;
    }
    PLF_HIP_TRY(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    PLF_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return PLF_OK;
}

extern "C" int plf_fill(void *dst_device, int32_t byte_value, size_t bytes, void *stream)
{
    if (bytes == 0) return PLF_OK;
    if (!dst_device) return PLF_E_BADARG;
    PLF_HIP_TRY(hipMemsetAsync(dst_device, byte_value, bytes, (hipStream_t)stream));
    if (!stream) PLF_HIP_TRY(hipStreamSynchronize(nullptr));
    return PLF_OK;
}
