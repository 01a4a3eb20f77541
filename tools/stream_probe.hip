// Streaming-rate probe for the tiled multiply's skeletons (GPU box; hipcc --offload-arch=gfx950 -O3).
// Every variant reads N complex128 amplitudes and writes N (32 B/amp), so GB/s = 32 N / t is comparable to
// the "32 B/amp" rate of a tile pass without gathered records.
//   copy        grid-stride 16 B/lane copy (plain | nt loads+stores)
//   tile        one 64 KB tile per workgroup: global -> registers -> LDS -> barrier -> LDS -> registers -> global
//   tile_dma    the same with global_load_lds staging
//   persist     256*k workgroups loop over tiles; tile t+1 is DMA-ed into the other LDS buffer while tile t is
//               read back and stored (one barrier per tile)
//   persist_reg the same with a register prefetch of tile t+1 and one LDS buffer
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
typedef double d2v __attribute__((ext_vector_type(2)));

template <bool NT_>
__global__ void __launch_bounds__(256) copy_kernel(const d2v *__restrict__ x, d2v *__restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT_) __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
    else y[i] = x[i];
  }
}

// 4 loads in flight per lane, workgroup-contiguous chunks
template <bool NT_>
__global__ void __launch_bounds__(256) copy4_kernel(const d2v *__restrict__ x, d2v *__restrict__ y, size_t n) {
  for (size_t b = (size_t)blockIdx.x * 1024; b < n; b += (size_t)gridDim.x * 1024) {
    d2v v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = NT_ ? __builtin_nontemporal_load(x + b + k * 256 + threadIdx.x) : x[b + k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (NT_) __builtin_nontemporal_store(v[k], y + b + k * 256 + threadIdx.x);
      else y[b + k * 256 + threadIdx.x] = v[k];
    }
  }
}

constexpr int B = 12;
template <int LOGR, bool DMA, bool NTL>
__global__ void __launch_bounds__(1 << (B - LOGR), (LOGR == 3 ? 4 : 2))
tile_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  constexpr int R = 1 << LOGR, NT = 1 << (B - LOGR);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  const uint32_t tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x << B;
  if (DMA) {
#pragma unroll
    for (int k = 0; k < R; ++k)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(x + base + tid + k * NT),
                                       (LDS_AS void *)(tile + (k * NT + (tid & ~63u))), 16, 0, NTL ? 2 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    d2v v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = NTL ? __builtin_nontemporal_load(x + base + tid + k * NT) : x[base + tid + k * NT];
#pragma unroll
    for (int k = 0; k < R; ++k) tile[tid + k * NT] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    d2v a = tile[(tid ^ 1u) + k * NT], b = tile[(tid ^ 5u) + k * NT];
    __builtin_nontemporal_store(a + b, y + base + tid + k * NT);
  }
}

// persistent, LDS double buffer filled by DMA
template <int LOGR, bool NTL>
__global__ void __launch_bounds__(1 << (B - LOGR), 1 << (B - LOGR) >> 8)
persist_kernel(const d2v *__restrict__ x, d2v *__restrict__ y, uint32_t ntiles) {
  constexpr int R = 1 << LOGR, NT = 1 << (B - LOGR);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *buf = reinterpret_cast<d2v *>(smem);
  const uint32_t tid = threadIdx.x;
  uint32_t t = blockIdx.x;
  if (t >= ntiles) return;
  {
    const size_t base = (size_t)t << B;
#pragma unroll
    for (int k = 0; k < R; ++k)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(x + base + tid + k * NT),
                                       (LDS_AS void *)(buf + (k * NT + (tid & ~63u))), 16, 0, NTL ? 2 : 0);
  }
  uint32_t cur = 0;
  for (; t < ntiles; t += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t tn = t + gridDim.x;
    if (tn < ntiles) {
      const size_t base = (size_t)tn << B;
      d2v *nb = buf + ((cur ^ 1u) << B);
#pragma unroll
      for (int k = 0; k < R; ++k)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(x + base + tid + k * NT),
                                         (LDS_AS void *)(nb + (k * NT + (tid & ~63u))), 16, 0, NTL ? 2 : 0);
    }
    const d2v *tile = buf + (cur << B);
    const size_t base = (size_t)t << B;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      d2v a = tile[(tid ^ 1u) + k * NT], b = tile[(tid ^ 5u) + k * NT];
      __builtin_nontemporal_store(a + b, y + base + tid + k * NT);
    }
    cur ^= 1u;
  }
}

// persistent, one LDS buffer, next tile prefetched into registers
template <int LOGR>
__global__ void __launch_bounds__(1 << (B - LOGR), (LOGR == 3 ? 4 : 2))
persist_reg_kernel(const d2v *__restrict__ x, d2v *__restrict__ y, uint32_t ntiles) {
  constexpr int R = 1 << LOGR, NT = 1 << (B - LOGR);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2v *tile = reinterpret_cast<d2v *>(smem);
  const uint32_t tid = threadIdx.x;
  uint32_t t = blockIdx.x;
  if (t >= ntiles) return;
  d2v v[R];
  {
    const size_t base = (size_t)t << B;
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = __builtin_nontemporal_load(x + base + tid + k * NT);
  }
  for (; t < ntiles; t += gridDim.x) {
#pragma unroll
    for (int k = 0; k < R; ++k) tile[tid + k * NT] = v[k];
    __syncthreads();
    const uint32_t tn = t + gridDim.x;
    if (tn < ntiles) {
      const size_t base = (size_t)tn << B;
#pragma unroll
      for (int k = 0; k < R; ++k) v[k] = __builtin_nontemporal_load(x + base + tid + k * NT);
    }
    const size_t base = (size_t)t << B;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      d2v a = tile[(tid ^ 1u) + k * NT], b = tile[(tid ^ 5u) + k * NT];
      __builtin_nontemporal_store(a + b, y + base + tid + k * NT);
    }
    __syncthreads();
  }
}

template <typename F>
static void timeit(const char *name, size_t n, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-44s %8.3f ms  %7.1f GB/s (32 B/amp)\n", name, ms, 32.0 * n / ms / 1e6);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 30;
  const size_t n = (size_t)1 << L;
  d2v *x, *y;
  CK(hipMalloc(&x, n * 16));
  CK(hipMalloc(&y, n * 16));
  CK(hipMemset(x, 1, n * 16));
  CK(hipMemset(y, 0, n * 16));
  const uint32_t ntiles = (uint32_t)(n >> B);
  for (int g : {2048, 8192, 65536}) {
    char nm[64];
    snprintf(nm, sizeof nm, "copy plain grid=%d", g);
    timeit(nm, n, [&] { copy_kernel<false><<<g, 256>>>(x, y, n); });
    snprintf(nm, sizeof nm, "copy nt grid=%d", g);
    timeit(nm, n, [&] { copy_kernel<true><<<g, 256>>>(x, y, n); });
    snprintf(nm, sizeof nm, "copy4 plain grid=%d", g);
    timeit(nm, n, [&] { copy4_kernel<false><<<g, 256>>>(x, y, n); });
    snprintf(nm, sizeof nm, "copy4 nt grid=%d", g);
    timeit(nm, n, [&] { copy4_kernel<true><<<g, 256>>>(x, y, n); });
  }
  const size_t lds1 = (size_t)16 << B;
#define SETLDS(k, b) CK(hipFuncSetAttribute((const void *)(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(b)))
  SETLDS((tile_kernel<3, false, false>), lds1); SETLDS((tile_kernel<3, false, true>), lds1);
  SETLDS((tile_kernel<3, true, false>), lds1);  SETLDS((tile_kernel<3, true, true>), lds1);
  SETLDS((tile_kernel<4, false, true>), lds1);  SETLDS((tile_kernel<4, true, true>), lds1);
  timeit("tile R=8 regs plain-load", n, [&] { tile_kernel<3, false, false><<<ntiles, 512, lds1>>>(x, y); });
  timeit("tile R=8 regs nt-load", n, [&] { tile_kernel<3, false, true><<<ntiles, 512, lds1>>>(x, y); });
  timeit("tile R=8 dma", n, [&] { tile_kernel<3, true, false><<<ntiles, 512, lds1>>>(x, y); });
  timeit("tile R=8 dma nt", n, [&] { tile_kernel<3, true, true><<<ntiles, 512, lds1>>>(x, y); });
  timeit("tile R=16 regs nt-load", n, [&] { tile_kernel<4, false, true><<<ntiles, 256, lds1>>>(x, y); });
  timeit("tile R=16 dma nt", n, [&] { tile_kernel<4, true, true><<<ntiles, 256, lds1>>>(x, y); });
  const size_t lds2 = 2 * lds1;
  SETLDS((persist_kernel<3, false>), lds2); SETLDS((persist_kernel<3, true>), lds2);
  SETLDS((persist_kernel<2, true>), lds2);  SETLDS((persist_kernel<4, true>), lds2);
  SETLDS((persist_reg_kernel<3>), lds1);    SETLDS((persist_reg_kernel<4>), lds1);
  timeit("persist dma 512thr grid=256", n, [&] { persist_kernel<3, false><<<256, 512, lds2>>>(x, y, ntiles); });
  timeit("persist dma nt 512thr grid=256", n, [&] { persist_kernel<3, true><<<256, 512, lds2>>>(x, y, ntiles); });
  timeit("persist dma nt 1024thr grid=256", n, [&] { persist_kernel<2, true><<<256, 1024, lds2>>>(x, y, ntiles); });
  timeit("persist dma nt 256thr grid=256", n, [&] { persist_kernel<4, true><<<256, 256, lds2>>>(x, y, ntiles); });
  timeit("persist reg 512thr grid=512", n, [&] { persist_reg_kernel<3><<<512, 512, lds1>>>(x, y, ntiles); });
  timeit("persist reg 256thr grid=512", n, [&] { persist_reg_kernel<4><<<512, 256, lds1>>>(x, y, ntiles); });
  timeit("persist reg 512thr grid=256", n, [&] { persist_reg_kernel<3><<<256, 512, lds1>>>(x, y, ntiles); });
  return 0;
}
