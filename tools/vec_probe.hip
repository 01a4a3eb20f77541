// Launch shapes for the Krylov vector sweeps (axpby: x, y -> y, 48 B/amp; dot: x, y -> partials, 32 B/amp read-only;
// maxpy with 4 vectors: 5 reads + 1 write, 96 B/amp): grid-stride over 2048 workgroups (csrc/vec_kernels.hip today)
// against one element per thread, and a few in between.
// hipcc --offload-arch=gfx950 -O3 tools/vec_probe.hip -o /tmp/vec_probe && /tmp/vec_probe [log2 n]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));
constexpr int NT = 256;

__device__ __forceinline__ d2v ldn(const d2v *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stn(d2v *p, d2v v) { __builtin_nontemporal_store(v, p); }

// U elements per thread per trip, loads of a trip issued together
template <int U>
__global__ void __launch_bounds__(NT) axpby_kernel(d2v *y, const d2v *__restrict__ x, int64_t n, double a, double b) {
  const int64_t stride = (int64_t)gridDim.x * NT * U;
  for (int64_t i0 = (int64_t)blockIdx.x * NT * U + threadIdx.x; i0 < n; i0 += stride) {
    d2v xv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * NT < n) { xv[u] = ldn(x + i0 + u * NT); yv[u] = ldn(y + i0 + u * NT); }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * NT < n) stn(y + i0 + u * NT, a * xv[u] + b * yv[u]);
  }
}

template <int U>
__global__ void __launch_bounds__(NT) dot_kernel(const d2v *__restrict__ y, const d2v *__restrict__ x, int64_t n, double *partials) {
  const int64_t stride = (int64_t)gridDim.x * NT * U;
  double sr = 0.0, si = 0.0;
  for (int64_t i0 = (int64_t)blockIdx.x * NT * U + threadIdx.x; i0 < n; i0 += stride) {
    d2v xv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * NT < n) { xv[u] = ldn(x + i0 + u * NT); yv[u] = ldn(y + i0 + u * NT); }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * NT < n) {
      sr += xv[u].x * yv[u].x + xv[u].y * yv[u].y;
      si += xv[u].x * yv[u].y - xv[u].y * yv[u].x;
    }
  }
  for (int off = 32; off > 0; off >>= 1) { sr += __shfl_xor(sr, off, 64); si += __shfl_xor(si, off, 64); }
  __shared__ double red[NT / 64][2];
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sr; red[threadIdx.x >> 6][1] = si; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double s = 0.0;
    for (int w = 0; w < NT / 64; ++w) s += red[w][threadIdx.x];
    partials[2 * (int64_t)blockIdx.x + threadIdx.x] = s;
  }
}

template <int U, int NV>
__global__ void __launch_bounds__(NT) maxpy_kernel(d2v *w, const d2v *__restrict__ V, int64_t ldv, int64_t n, double c) {
  const int64_t stride = (int64_t)gridDim.x * NT * U;
  for (int64_t i0 = (int64_t)blockIdx.x * NT * U + threadIdx.x; i0 < n; i0 += stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * NT;
      if (i < n) {
        d2v acc = ldn(w + i);
        d2v v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = ldn(V + j * ldv + i);
#pragma unroll
        for (int j = 0; j < NV; ++j) acc += c * v[j];
        stn(w + i, acc);
      }
    }
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
  const int64_t n = (int64_t)1 << lg;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  d2v *X, *Y, *V;
  double *P;
  CK(hipMalloc(&X, n * 16));
  CK(hipMalloc(&Y, n * 16));
  CK(hipMalloc(&V, n * 16 * 4));
  CK(hipMalloc(&P, (size_t)(n / NT + 1) * 16));
  CK(hipMemset(X, 0, n * 16));
  CK(hipMemset(Y, 0, n * 16));
  CK(hipMemset(V, 0, n * 16 * 4));
  printf("n = 2^%d amplitudes\n", lg);
  auto blocks = [&](int U, int64_t cap) { int64_t nb = (n + (int64_t)NT * U - 1) / ((int64_t)NT * U); return (unsigned)(nb < cap ? nb : cap); };
  const int64_t caps[] = {2048, 8192, 65536, (int64_t)1 << 30};
  for (int64_t cap : caps) {
#define RUN(U)                                                                                                         \
    {                                                                                                                  \
      const unsigned nb = blocks(U, cap);                                                                              \
      double t = time_ms([&] { hipLaunchKernelGGL(axpby_kernel<U>, dim3(nb), dim3(NT), 0, 0, Y, X, n, 0.5, 0.25); }, 5); \
      printf("axpby  %d per trip, %9u workgroups: %7.3f ms  %7.1f GB/s (48 B/amp)\n", U, nb, t, 48.0 * n / 1e6 / t);     \
      t = time_ms([&] { hipLaunchKernelGGL(dot_kernel<U>, dim3(nb), dim3(NT), 0, 0, Y, X, n, P); }, 5);                 \
      printf("dot    %d per trip, %9u workgroups: %7.3f ms  %7.1f GB/s (32 B/amp)\n", U, nb, t, 32.0 * n / 1e6 / t);     \
      t = time_ms([&] { hipLaunchKernelGGL((maxpy_kernel<U, 4>), dim3(nb), dim3(NT), 0, 0, Y, V, n, n, 0.5); }, 5);     \
      printf("maxpy4 %d per trip, %9u workgroups: %7.3f ms  %7.1f GB/s (96 B/amp)\n", U, nb, t, 96.0 * n / 1e6 / t);     \
      fflush(stdout);                                                                                                  \
    }
    RUN(1)
    RUN(2)
    RUN(4)
#undef RUN
  }
  return 0;
}
