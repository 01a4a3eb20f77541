// Shared internals of libdynamite_amd: error plumbing and small types.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/dynamite_amd.h"

namespace dnm {

void set_error(const char *fmt, ...);

#define DNM_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      dnm::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),        \
                     __FILE__, __LINE__);                                          \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

#define DNM_CHECK(cond, ...)                                                       \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      dnm::set_error(__VA_ARGS__);                                                 \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

#define DNM_TRY(call)                                                              \
  do {                                                                             \
    int rc_ = (call);                                                              \
    if (rc_) return rc_;                                                           \
  } while (0)

struct cplx {
  double re, im;
};

static inline int parity64(uint64_t v) { return __builtin_parityll(v); }
static inline int ilog2(uint64_t v) { return 63 - __builtin_clzll(v); }

}  // namespace dnm
