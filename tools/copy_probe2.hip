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

// H1: is it the interleave granularity?  Workgroup b moves R pieces of NT amplitudes that lie nWG*NT amplitudes apart
// (SPREAD = 1: piece k at (k*nWG + b)*NT), so the workgroups running at one time cover contiguous memory the way the
// R = 1 shape does, or groups of 2^SG consecutive workgroups interleave their pieces (SPREAD = 2: piece k of workgroup
// b at ((b >> SG) * R * 2^SG + k * 2^SG + (b & (2^SG - 1))) * NT: a window tile whose k bits sit right above SG bits)
template <int NT, int R, int SPREAD, int SG>
__global__ void __launch_bounds__(NT, (2048 / NT >= 8 ? 8 : 2048 / NT) * NT / 256)
spread_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  const uint32_t tid = threadIdx.x, b = blockIdx.x, nwg = gridDim.x;
  d2v v[R];
  size_t off[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if (SPREAD == 1) off[k] = ((size_t)k * nwg + b) * NT + tid;
    else off[k] = (((size_t)(b >> SG) * R << SG) + ((size_t)k << SG) + (b & ((1u << SG) - 1u))) * NT + tid;
  }
#pragma unroll
  for (int k = 0; k < R; ++k) v[k] = x[off[k]];
#pragma unroll
  for (int k = 0; k < R; ++k) tile[tid + k * NT] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const d2v a = tile[(tid ^ 1u) + k * NT], c = tile[(tid ^ 5u) + k * NT];
    __builtin_nontemporal_store(a + c, y + off[k]);
  }
}

// The multiply's own shape under a bit-permuted vector layout: 1024 threads x 4 rows, tile coordinate t (12 bits);
// the tile lies in memory as 2^(12-P) pieces of 2^P amplitudes, the pieces of 8 consecutive workgroups (one per XCD)
// interleaved: offset = (t & (2^P - 1)) | (b & 7) << P | (t >> P) << (P + 3) | (b >> 3) << 15.  ACCUM: read y too.
template <int P, bool ACCUM>
__global__ void __launch_bounds__(1024, 8)
permuted_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  constexpr int NT = 1024, R = 4;
  const uint32_t tid = threadIdx.x, b = blockIdx.x;
  const size_t base = ((size_t)(b >> 3) << 15) | ((size_t)(b & 7u) << P);
  d2v v[R], w[R];
  size_t off[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t t = tid + k * NT;
    off[k] = base | (t & ((1u << P) - 1u)) | ((size_t)(t >> P) << (P + 3));
  }
#pragma unroll
  for (int k = 0; k < R; ++k) v[k] = __builtin_nontemporal_load(x + off[k]);
  if (ACCUM) {
#pragma unroll
    for (int k = 0; k < R; ++k) w[k] = __builtin_nontemporal_load(y + off[k]);
  }
#pragma unroll
  for (int k = 0; k < R; ++k) tile[tid + k * NT] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    d2v a = tile[(tid ^ 1u) + k * NT] + tile[(tid ^ 5u) + k * NT];
    if (ACCUM) a += w[k];
    __builtin_nontemporal_store(a, y + off[k]);
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

template <int NT, int R, int SPREAD, int SG>
static void run_spread() {
  auto k = spread_kernel<NT, R, SPREAD, SG>;
  const size_t lds = (size_t)NT * R * 16;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double t = time_ms([&] { hipLaunchKernelGGL(k, dim3((unsigned)(N / (NT * R))), dim3(NT), lds, 0, X, Y); }, 10);
  if (SPREAD == 1) printf("lds    NT=%4d R=%d (tile %3zu KB) pieces a whole pass apart          %7.3f ms  %7.1f GB/s\n", NT, R, lds >> 10, t, 32.0 * N / 1e9 / t * 1e3);
  else printf("lds    NT=%4d R=%d (tile %3zu KB) pieces of %4d workgroups interleaved %7.3f ms  %7.1f GB/s\n", NT, R, lds >> 10, 1 << SG, t, 32.0 * N / 1e9 / t * 1e3);
  fflush(stdout);
}

template <int P, bool ACCUM>
static void run_permuted() {
  auto k = permuted_kernel<P, ACCUM>;
  const size_t lds = (size_t)4096 * 16;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double t = time_ms([&] { hipLaunchKernelGGL(k, dim3((unsigned)(N / 4096)), dim3(1024), lds, 0, X, Y); }, 10);
  const double bpa = ACCUM ? 48.0 : 32.0;
  printf("tile   NT=1024 R=4 (64 KB) in %4d pieces of %5d B, 8 workgroups interleaved, %s  %7.3f ms  %7.1f GB/s\n", 4096 >> P, 16 << P,
         ACCUM ? "x,y -> y (48 B/amp)" : "x -> y (32 B/amp)  ", t, bpa * N / 1e9 / t * 1e3);
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
  run_permuted<12, false>();
  run_permuted<10, false>();
  run_permuted<9, false>();
  run_permuted<8, false>();
  run_permuted<7, false>();
  run_permuted<6, false>();
  run_permuted<12, true>();
  run_permuted<10, true>();
  run_permuted<9, true>();
  run_permuted<8, true>();
  run_permuted<7, true>();
  run_permuted<6, true>();
  run_spread<1024, 4, 1, 0>();
  run_spread<1024, 4, 2, 3>();
  run_spread<1024, 4, 2, 6>();
  run_spread<1024, 4, 2, 9>();
  run_spread<1024, 4, 2, 12>();
  run_spread<512, 4, 1, 0>();
  run_spread<512, 4, 2, 3>();
  run_spread<512, 4, 2, 6>();
  run_spread<256, 4, 1, 0>();
  run_spread<256, 4, 2, 3>();
  run_spread<256, 4, 2, 6>();
  run_spread<256, 16, 2, 3>();
  run_spread<256, 16, 2, 6>();
  return 0;
}
