// Copy-rate probe 2: which WORKGROUP SHAPE streams 32 B/amp fastest?  (tools/copy_probe.hip showed that one 16-byte
// element per thread in 256-thread workgroups copies 2^30 amplitudes at 6.5 TB/s while the tiled multiply's shape --
// 64 KB per 1024-thread workgroup, 4 amplitudes per thread -- reaches 5.2-5.5, whatever the allocation.)
//   direct<NT, R, CONTIG>  a workgroup copies NT*R amplitudes; a thread loads its R values, then stores them.
//                          CONTIG: a wavefront's R loads are adjacent 1 KB segments; else they are NT*16 B apart
//   lds<NT, R>             the same through LDS with a barrier (the tile skeleton)
// hipcc --offload-arch=gfx950 -O3 tools/copy_probe2.hip -o /tmp/copy_probe2 && /tmp/copy_probe2 [log2 n]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

template <int NT, int R, bool CONTIG, bool NTL, bool NTS>
__global__ void __launch_bounds__(NT, (2048 / NT >= 8 ? 8 : 2048 / NT) * NT / 256)
direct_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  const size_t base = (size_t)blockIdx.x * (NT * R);
  const uint32_t tid = threadIdx.x;
  d2v v[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const size_t i = base + (CONTIG ? ((tid >> 6) * (64 * R) + k * 64 + (tid & 63)) : (tid + k * NT));
    v[k] = NTL ? __builtin_nontemporal_load(x + i) : x[i];
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const size_t i = base + (CONTIG ? ((tid >> 6) * (64 * R) + k * 64 + (tid & 63)) : (tid + k * NT));
    if (NTS) __builtin_nontemporal_store(v[k], y + i);
    else y[i] = v[k];
  }
}

template <int NT, int R, bool NTS>
__global__ void __launch_bounds__(NT, (2048 / NT >= 8 ? 8 : 2048 / NT) * NT / 256)
lds_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  const size_t base = (size_t)blockIdx.x * (NT * R);
  const uint32_t tid = threadIdx.x;
  d2v v[R];
#pragma unroll
  for (int k = 0; k < R; ++k) v[k] = x[base + tid + k * NT];
#pragma unroll
  for (int k = 0; k < R; ++k) tile[tid + k * NT] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const d2v a = tile[(tid ^ 1u) + k * NT], b = tile[(tid ^ 5u) + k * NT];
    if (NTS) __builtin_nontemporal_store(a + b, y + base + tid + k * NT);
    else y[base + tid + k * NT] = a + b;
  }
}

// the tile skeleton with the loads of a thread issued G at a time (G = R: all at once, as lds_kernel): fewer bytes in
// flight per wavefront, more round trips per tile
template <int NT, int R, int G>
__global__ void __launch_bounds__(NT, (2048 / NT >= 8 ? 8 : 2048 / NT) * NT / 256)
lds_serial_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  const size_t base = (size_t)blockIdx.x * (NT * R);
  const uint32_t tid = threadIdx.x;
#pragma unroll
  for (int k0 = 0; k0 < R; k0 += G) {
    d2v v[G];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = x[base + tid + (k0 + g) * NT];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < G; ++g) tile[tid + (k0 + g) * NT] = v[g];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const d2v a = tile[(tid ^ 1u) + k * NT], b = tile[(tid ^ 5u) + k * NT];
    __builtin_nontemporal_store(a + b, y + base + tid + k * NT);
  }
}
// ... and with the stores also spaced: store row k, then wait, before the next
template <int NT, int R>
__global__ void __launch_bounds__(NT, (2048 / NT >= 8 ? 8 : 2048 / NT) * NT / 256)
lds_serial2_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  const size_t base = (size_t)blockIdx.x * (NT * R);
  const uint32_t tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const d2v v = x[base + tid + k * NT];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile[tid + k * NT] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const d2v a = tile[(tid ^ 1u) + k * NT], b = tile[(tid ^ 5u) + k * NT];
    __builtin_nontemporal_store(a + b, y + base + tid + k * NT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

static d2v *X, *Y;
static size_t N;

template <int NT, int R, bool CONTIG, bool NTL, bool NTS>
static void run_direct() {
  const double t = time_ms([&] { hipLaunchKernelGGL((direct_kernel<NT, R, CONTIG, NTL, NTS>), dim3((unsigned)(N / (NT * R))), dim3(NT), 0, 0, X, Y); }, 10);
  printf("direct NT=%4d R=%d %-7s loads=%-5s stores=%-5s  %7.3f ms  %7.1f GB/s\n", NT, R, CONTIG ? "contig" : "strided", NTL ? "nt" : "plain",
         NTS ? "nt" : "plain", t, 32.0 * N / 1e9 / t * 1e3);
  fflush(stdout);
}
template <int NT, int R, bool NTS>
static void run_lds() {
  auto k = lds_kernel<NT, R, NTS>;
  const size_t lds = (size_t)NT * R * 16;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double t = time_ms([&] { hipLaunchKernelGGL(k, dim3((unsigned)(N / (NT * R))), dim3(NT), lds, 0, X, Y); }, 10);
  printf("lds    NT=%4d R=%d (tile %3zu KB)         stores=%-5s  %7.3f ms  %7.1f GB/s\n", NT, R, lds >> 10, NTS ? "nt" : "plain", t,
         32.0 * N / 1e9 / t * 1e3);
  fflush(stdout);
}

template <int NT, int R, int G>
static void run_lds_serial() {
  auto k = lds_serial_kernel<NT, R, G>;
  const size_t lds = (size_t)NT * R * 16;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double t = time_ms([&] { hipLaunchKernelGGL(k, dim3((unsigned)(N / (NT * R))), dim3(NT), lds, 0, X, Y); }, 10);
  printf("lds    NT=%4d R=%d (tile %3zu KB) loads %d at a time     %7.3f ms  %7.1f GB/s\n", NT, R, lds >> 10, G, t,
         32.0 * N / 1e9 / t * 1e3);
  fflush(stdout);
}
template <int NT, int R>
static void run_lds_serial2() {
  auto k = lds_serial2_kernel<NT, R>;
  const size_t lds = (size_t)NT * R * 16;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double t = time_ms([&] { hipLaunchKernelGGL(k, dim3((unsigned)(N / (NT * R))), dim3(NT), lds, 0, X, Y); }, 10);
  printf("lds    NT=%4d R=%d (tile %3zu KB) loads and stores 1 at a time %7.3f ms  %7.1f GB/s\n", NT, R, lds >> 10, t,
         32.0 * N / 1e9 / t * 1e3);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  N = (size_t)1 << lg;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipMalloc(&X, N * 16));
  CK(hipMalloc(&Y, N * 16));
  CK(hipMemset(X, 0, N * 16));
  CK(hipMemset(Y, 0, N * 16));
  printf("n = 2^%d amplitudes per vector, 32 B/amp\n", lg);
  run_direct<256, 1, false, true, true>();
  run_direct<256, 1, false, false, false>();
  run_direct<256, 1, false, false, true>();
  run_direct<512, 1, false, true, true>();
  run_direct<1024, 1, false, true, true>();
  run_direct<256, 2, false, true, true>();
  run_direct<256, 2, true, true, true>();
  run_direct<256, 4, false, true, true>();
  run_direct<256, 4, true, true, true>();
  run_direct<512, 2, false, true, true>();
  run_direct<512, 4, false, true, true>();
  run_direct<512, 4, true, true, true>();
  run_direct<512, 8, false, true, true>();
  run_direct<1024, 2, false, true, true>();
  run_direct<1024, 2, true, true, true>();
  run_direct<1024, 4, false, true, true>();
  run_direct<1024, 4, true, true, true>();
  run_direct<1024, 4, false, false, true>();
  run_lds<256, 1, true>();
  run_lds<256, 2, true>();
  run_lds<256, 4, true>();
  run_lds<512, 2, true>();
  run_lds<512, 4, true>();
  run_lds<512, 8, true>();
  run_lds<1024, 1, true>();
  run_lds<1024, 2, true>();
  run_lds<1024, 4, true>();
  run_lds<1024, 4, false>();
  run_lds_serial<1024, 4, 1>();
  run_lds_serial<1024, 4, 2>();
  run_lds_serial<1024, 4, 4>();
  run_lds_serial<512, 8, 1>();
  run_lds_serial<512, 8, 2>();
  run_lds_serial<512, 8, 4>();
  run_lds_serial2<1024, 4>();
  run_lds_serial2<512, 8>();
  return 0;
}
