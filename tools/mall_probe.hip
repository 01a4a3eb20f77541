// Infinity-Cache blocking probe: would the two passes of the tiled multiply run faster if the second pass found
// its operands (x again, the y the first pass wrote) in the 256 MB Infinity Cache instead of in HBM?
//   pass A of a chunk: read x, write y            (32 B/amp, the window pass's streams)
//   pass B of a chunk: read x, read y, write y    (48 B/amp, the accumulating contiguous pass's streams)
// both through a 64 KB LDS tile per 1024-thread workgroup (the multiply's shape).  ONE launch holds the workgroups of
// both passes; the block order decides how far B trails A:
//   lag = 0 (reference): all of A, then all of B -- what two launches do
//   lag = k: A(0) .. A(k-1), then A(c+k), B(c) alternating -- chunk c's x and y are at most (k+1) chunks old when B reads
// (No dependency tracking: a bandwidth probe, the values are not checked.  Workgroups dispatch in block order, so B(c)
// starts after A(c .. c+k) were dispatched.)
// hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe [log2 n]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

constexpr int NT = 1024, R = 4, TILE = NT * R;

// logc: log2 of the tiles per chunk; lag as above; xcd: spread consecutive tiles of a chunk over the XCDs as the
// hardware does (block b runs on XCD b % 8) -- nothing to do, consecutive blocks are consecutive tiles
__global__ void __launch_bounds__(NT, 8)
fused_kernel(const d2v *__restrict__ x, d2v *__restrict__ y, int logc, int lag, uint32_t nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  const uint32_t tpc = 1u << logc;
  const uint32_t seg = blockIdx.x >> logc, within = blockIdx.x & (tpc - 1u);
  // segment order: lag == 0: A(0..n-1), B(0..n-1).  Otherwise A(0..lag-1), then pairs A(c+lag), B(c); the last lag
  // pairs have no A left
  uint32_t chunk;
  bool isB;
  if (lag == 0) {
    isB = seg >= nchunks;
    chunk = isB ? seg - nchunks : seg;
  } else if (seg < (uint32_t)lag) {
    isB = false;
    chunk = seg;
  } else {
    const uint32_t s = seg - lag;            // 0 .. 2 n - lag - 1
    const uint32_t npairs = nchunks - lag;   // pairs A(c+lag), B(c) for c < npairs
    if (s < 2 * npairs) {
      isB = s & 1u;
      chunk = isB ? (s >> 1) : (s >> 1) + lag;
    } else {
      isB = true;
      chunk = npairs + (s - 2 * npairs);
    }
  }
  const size_t base = ((size_t)chunk << logc) * TILE + (size_t)within * TILE;
  const uint32_t tid = threadIdx.x;
  d2v v[R], w[R];
#pragma unroll
  for (int k = 0; k < R; ++k) v[k] = __builtin_nontemporal_load(x + base + tid + k * NT);
  if (isB) {
#pragma unroll
    for (int k = 0; k < R; ++k) w[k] = __builtin_nontemporal_load(y + base + tid + k * NT);
  }
#pragma unroll
  for (int k = 0; k < R; ++k) tile[tid + k * NT] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    d2v a = tile[(tid ^ 1u) + k * NT] + tile[(tid ^ 5u) + k * NT];
    if (isB) a += w[k];
    __builtin_nontemporal_store(a, y + base + tid + k * NT);
  }
}

// the same with plain (cached) loads and stores, to see whether the non-temporal hint keeps lines out of the MALL
__global__ void __launch_bounds__(NT, 8)
fused_plain_kernel(const d2v *__restrict__ x, d2v *__restrict__ y, int logc, int lag, uint32_t nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  const uint32_t tpc = 1u << logc;
  const uint32_t seg = blockIdx.x >> logc, within = blockIdx.x & (tpc - 1u);
  uint32_t chunk;
  bool isB;
  if (lag == 0) {
    isB = seg >= nchunks;
    chunk = isB ? seg - nchunks : seg;
  } else if (seg < (uint32_t)lag) {
    isB = false;
    chunk = seg;
  } else {
    const uint32_t s = seg - lag;
    const uint32_t npairs = nchunks - lag;
    if (s < 2 * npairs) {
      isB = s & 1u;
      chunk = isB ? (s >> 1) : (s >> 1) + lag;
    } else {
      isB = true;
      chunk = npairs + (s - 2 * npairs);
    }
  }
  const size_t base = ((size_t)chunk << logc) * TILE + (size_t)within * TILE;
  const uint32_t tid = threadIdx.x;
  d2v v[R], w[R];
#pragma unroll
  for (int k = 0; k < R; ++k) v[k] = x[base + tid + k * NT];
  if (isB) {
#pragma unroll
    for (int k = 0; k < R; ++k) w[k] = y[base + tid + k * NT];
  }
#pragma unroll
  for (int k = 0; k < R; ++k) tile[tid + k * NT] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    d2v a = tile[(tid ^ 1u) + k * NT] + tile[(tid ^ 5u) + k * NT];
    if (isB) a += w[k];
    y[base + tid + k * NT] = a;
  }
}

static hipEvent_t e0, e1;
template <class F>
static double time_ms(F f, int reps) {
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const size_t N = (size_t)1 << lg;
  d2v *X, *Y;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipMalloc(&X, N * 16));
  CK(hipMalloc(&Y, N * 16));
  CK(hipMemset(X, 0, N * 16));
  CK(hipMemset(Y, 0, N * 16));
  const size_t lds = (size_t)TILE * 16;
  CK(hipFuncSetAttribute((const void *)fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void *)fused_plain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const uint32_t ntiles = (uint32_t)(N / TILE);
  printf("n = 2^%d amplitudes, 80 B/amp over both passes (A: 32, B: 48); tile 64 KB\n", lg);
  for (int plain = 0; plain < 2; ++plain)
    for (int logc = 6; logc <= 12; ++logc) {       // chunk = 2^logc tiles = 2^(logc+12) amplitudes
      const uint32_t nchunks = ntiles >> logc;
      if (nchunks < 8) continue;
      for (int lag : {0, 1, 2, 4}) {
        if (lag == 0 && logc != 6) continue;
        const double t = time_ms([&] {
          if (plain) hipLaunchKernelGGL(fused_plain_kernel, dim3(2 * ntiles), dim3(NT), lds, 0, X, Y, logc, lag, nchunks);
          else hipLaunchKernelGGL(fused_kernel, dim3(2 * ntiles), dim3(NT), lds, 0, X, Y, logc, lag, nchunks);
        }, 5);
        const double mb = (double)((size_t)TILE << logc) * 32 / 1048576.0;
        printf("%-5s chunk 2^%2d amps (x+y %6.0f MB) lag %d: live set %6.0f MB  %7.3f ms  %7.1f GB/s (80 B/amp)\n", plain ? "plain" : "nt",
               logc + 12, mb, lag, mb * (lag + 1), t, 80.0 * N / 1e9 / t * 1e3);
        fflush(stdout);
      }
    }
  return 0;
}
