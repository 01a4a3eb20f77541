// Shared internals of libdynamite_amd: error plumbing and small types.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

// Experiment knobs (DNM_* environment variables that change plans, kernels or solver internals) are honoured
// only under DNM_EXPERIMENTAL=1; a production process ignores them.  Tests and the A/B tools set the gate.
static inline const char *knob(const char *name) {
  const char *g = getenv("DNM_EXPERIMENTAL");
  if (!g || g[0] != '1') return nullptr;
  const char *v = getenv(name);
  return (v && *v) ? v : nullptr;
}

struct cplx {
  double re, im;
};

static inline int parity64(uint64_t v) { return __builtin_parityll(v); }
static inline int ilog2(uint64_t v) { return 63 - __builtin_clzll(v); }

}  // namespace dnm
