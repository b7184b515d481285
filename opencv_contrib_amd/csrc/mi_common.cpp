// Error channel, device selection and device-memory helpers of the C-ABI.
#include "mi_common.h"

namespace mi {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mi

extern "C" {

const char *mi_last_error(void) { return mi::g_err; }
const char *mi_version(void) { return "miflow 0.1 (gfx950)"; }

int mi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int mi_set_device(int device) { MI_HIP_TRY(hipSetDevice(device)); return MI_OK; }
int mi_get_device(int *device) { MI_REQUIRE(device, MI_ERR_BAD_ARG, "null device"); MI_HIP_TRY(hipGetDevice(device)); return MI_OK; }

int mi_malloc(void **dptr, size_t bytes)
{
    MI_REQUIRE(dptr, MI_ERR_BAD_ARG, "null dptr");
    MI_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return MI_OK;
}
int mi_malloc_pitch(void **dptr, size_t *step, size_t width_bytes, int rows)
{
    MI_REQUIRE(dptr && step && rows >= 0, MI_ERR_BAD_ARG, "bad mi_malloc_pitch arguments");
    // GpuMat default allocator: pitched when rows > 1 && cols > 1; 256-B pitch here
    const size_t st = rows > 1 ? (width_bytes + 255) / 256 * 256 : width_bytes;
    MI_HIP_TRY(hipMalloc(dptr, st * (size_t)(rows ? rows : 1) + 16));
    *step = st;
    return MI_OK;
}
int mi_free(void *dptr) { MI_HIP_TRY(hipFree(dptr)); return MI_OK; }
int mi_memcpy_h2d(void *dst, size_t dstep, const void *src, size_t sstep, size_t width_bytes, int rows, void *stream)
{
    MI_HIP_TRY(hipMemcpy2DAsync(dst, dstep, src, sstep, width_bytes, (size_t)rows, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MI_OK;
}
int mi_memcpy_d2h(void *dst, size_t dstep, const void *src, size_t sstep, size_t width_bytes, int rows, void *stream)
{
    MI_HIP_TRY(hipMemcpy2DAsync(dst, dstep, src, sstep, width_bytes, (size_t)rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MI_OK;
}
int mi_memset(void *dst, size_t dstep, int value, size_t width_bytes, int rows, void *stream)
{
    MI_HIP_TRY(hipMemset2DAsync(dst, dstep, value, width_bytes, (size_t)rows, (hipStream_t)stream));
    return MI_OK;
}
int mi_stream_create(void **stream)
{
    MI_REQUIRE(stream, MI_ERR_BAD_ARG, "null stream");
    hipStream_t s;
    MI_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return MI_OK;
}
int mi_stream_destroy(void *stream) { MI_HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return MI_OK; }
int mi_stream_synchronize(void *stream) { MI_HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return MI_OK; }

}  // extern "C"
