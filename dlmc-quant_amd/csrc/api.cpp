// Version and error strings of the C ABI (include/dlmcq.h).
#include <hip/hip_runtime.h>

#include "dlmcq.h"

extern "C" int dlmcq_version(void) { return DLMCQ_VERSION; }

extern "C" const char* dlmcq_strerror(int code) {
  switch (code) {
    case DLMCQ_OK: return "success";
    case DLMCQ_EINVAL: return "dlmcq: invalid argument (null pointer, negative size, lo > hi or unknown enum)";
    case DLMCQ_ERANGE: return "dlmcq: size out of the kernels' index range";
    case DLMCQ_ESCRATCH: return "dlmcq: scratch buffer missing or too small";
    case DLMCQ_EALIGN: return "dlmcq: pointer alignment violated";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "dlmcq: unknown error code";
}
