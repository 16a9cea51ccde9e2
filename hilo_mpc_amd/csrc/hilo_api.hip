// Library-level entry points: error text, device memory helpers (so that a binding needs nothing but this .so).
#include <string.h>

#include "hilo_common.h"

namespace hilo {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace hilo

using namespace hilo;

extern "C" int hilo_abi_version(void) { return HILO_ABI_VERSION; }
extern "C" const char* hilo_last_error(void) { return err_buf(); }

extern "C" int hilo_device_count(int* count) {
  HILO_REQUIRE(count, "hilo_device_count: NULL argument");
  HILO_HIP_CHECK(hipGetDeviceCount(count));
  return HILO_OK;
}

extern "C" int hilo_malloc(void** dptr, uint64_t bytes, int device) {
  HILO_REQUIRE(dptr, "hilo_malloc: NULL argument");
  HILO_HIP_CHECK(hipSetDevice(device));
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
  if (e != hipSuccess) return fail(HILO_ENOMEM, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
  return HILO_OK;
}
extern "C" int hilo_free(void* dptr) {
  if (dptr) HILO_HIP_CHECK(hipFree(dptr));
  return HILO_OK;
}
extern "C" int hilo_memcpy_h2d(void* dst, const void* src_host, uint64_t bytes, void* stream) {
  HILO_HIP_CHECK(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return HILO_OK;
}
extern "C" int hilo_memcpy_d2h(void* dst_host, const void* src, uint64_t bytes, void* stream) {
  HILO_HIP_CHECK(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HILO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return HILO_OK;
}
extern "C" int hilo_stream_sync(void* stream) {
  HILO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return HILO_OK;
}
