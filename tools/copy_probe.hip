// Copy-rate probe: a plain 32 B/amp copy (read 16 B, write 16 B per complex128 amplitude) against
//   * the footprint (256 MiB, 1, 4, 16 GiB per vector),
//   * the allocation (one hipMalloc per vector | both vectors inside one 64 GiB arena | hipMallocAsync pool),
//   * the kernel shape (one 16 B element per thread, no loop = the guide's "float4 copy" | grid-stride with 4 loads in
//     flight, non-temporal | 64 KB tiles per workgroup as the tiled multiply moves them).
// Question it answers: is the 5.4-5.9 TB/s of tools/stream_probe.hip against the guide's
// 6.29 TB/s a property of the 16 GiB footprint / the allocator, or of the kernel shape?
//   hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o gpurun_out/copy_probe && gpurun_out/copy_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) copy1_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  y[i] = x[i];
}
__global__ void __launch_bounds__(256) copy1nt_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
}
// 4 loads in flight per lane, workgroup-contiguous 16 KB chunks, grid-stride
__global__ void __launch_bounds__(256) copy4nt_kernel(const d2v *__restrict__ x, d2v *__restrict__ y, size_t n) {
  for (size_t b = (size_t)blockIdx.x * 1024; b < n; b += (size_t)gridDim.x * 1024) {
    d2v v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(x + b + k * 256 + threadIdx.x);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], y + b + k * 256 + threadIdx.x);
  }
}
// one 64 KB tile per 1024-thread workgroup, 4 amplitudes per thread (the tiled multiply's global access shape)
__global__ void __launch_bounds__(1024, 8) tile4_kernel(const d2v *__restrict__ x, d2v *__restrict__ y) {
  const size_t base = (size_t)blockIdx.x << 12;
  d2v v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = x[base + threadIdx.x + k * 1024];
#pragma unroll
  for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], y + base + threadIdx.x + k * 1024);
}
// read-only and write-only halves (which direction is short of the guide's rate?)
__global__ void __launch_bounds__(256) read4_kernel(const d2v *__restrict__ x, d2v *__restrict__ y, size_t n) {
  d2v acc = {0.0, 0.0};
  for (size_t b = (size_t)blockIdx.x * 1024; b < n; b += (size_t)gridDim.x * 1024) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += __builtin_nontemporal_load(x + b + k * 256 + threadIdx.x);
  }
  if (acc.x == 1.2345e300) y[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) write4_kernel(d2v *__restrict__ y, size_t n) {
  const d2v v = {1.0, 2.0};
  for (size_t b = (size_t)blockIdx.x * 1024; b < n; b += (size_t)gridDim.x * 1024) {
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v, y + b + k * 256 + threadIdx.x);
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

static void run_set(const char *tag, d2v *x, d2v *y, size_t n) {
  const double gb = 32.0 * n / 1e9;
  const int reps = n >= ((size_t)1 << 28) ? 10 : 40;
  double t;
  t = time_ms([&] { hipLaunchKernelGGL(copy1_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, x, y); }, reps);
  printf("%-34s n=2^%2d  copy1 plain        %8.3f ms  %7.1f GB/s\n", tag, (int)__builtin_ctzll(n), t, gb / t * 1e3);
  t = time_ms([&] { hipLaunchKernelGGL(copy1nt_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, x, y); }, reps);
  printf("%-34s n=2^%2d  copy1 nt           %8.3f ms  %7.1f GB/s\n", tag, (int)__builtin_ctzll(n), t, gb / t * 1e3);
  for (unsigned g : {8192u, 65536u}) {
    t = time_ms([&] { hipLaunchKernelGGL(copy4nt_kernel, dim3(g), dim3(256), 0, 0, x, y, n); }, reps);
    printf("%-34s n=2^%2d  copy4 nt grid=%-5u %8.3f ms  %7.1f GB/s\n", tag, (int)__builtin_ctzll(n), g, t, gb / t * 1e3);
  }
  t = time_ms([&] { hipLaunchKernelGGL(tile4_kernel, dim3((unsigned)(n >> 12)), dim3(1024), 0, 0, x, y); }, reps);
  printf("%-34s n=2^%2d  tile4 (64KB/wg)    %8.3f ms  %7.1f GB/s\n", tag, (int)__builtin_ctzll(n), t, gb / t * 1e3);
  t = time_ms([&] { hipLaunchKernelGGL(read4_kernel, dim3(65536), dim3(256), 0, 0, x, y, n); }, reps);
  printf("%-34s n=2^%2d  read only          %8.3f ms  %7.1f GB/s (16 B/amp)\n", tag, (int)__builtin_ctzll(n), t, gb / 2 / t * 1e3);
  t = time_ms([&] { hipLaunchKernelGGL(write4_kernel, dim3(65536), dim3(256), 0, 0, y, n); }, reps);
  printf("%-34s n=2^%2d  write only         %8.3f ms  %7.1f GB/s (16 B/amp)\n", tag, (int)__builtin_ctzll(n), t, gb / 2 / t * 1e3);
  fflush(stdout);
}

int main() {
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  size_t freeb, totalb;
  CK(hipMemGetInfo(&freeb, &totalb));
  printf("device memory: %.1f GiB free of %.1f\n", freeb / 1073741824.0, totalb / 1073741824.0);

  // (1) one hipMalloc per vector
  for (int lg : {24, 26, 28, 30}) {
    const size_t n = (size_t)1 << lg;
    d2v *x, *y;
    CK(hipMalloc(&x, n * 16));
    CK(hipMalloc(&y, n * 16));
    CK(hipMemset(x, 0, n * 16));
    CK(hipMemset(y, 0, n * 16));
    char tag[64];
    snprintf(tag, sizeof tag, "hipMalloc per vector");
    run_set(tag, x, y, n);
    CK(hipFree(x));
    CK(hipFree(y));
  }
  // (2) one 64 GiB arena, vectors at several offsets
  {
    unsigned char *arena;
    const size_t A = (size_t)64 << 30;
    CK(hipMalloc(&arena, A));
    CK(hipMemset(arena, 0, A));
    for (int lg : {26, 28, 30}) {
      const size_t n = (size_t)1 << lg;
      run_set("64 GiB arena, y right behind x", (d2v *)arena, (d2v *)(arena + n * 16), n);
    }
    const size_t n = (size_t)1 << 30;
    run_set("64 GiB arena, y at +32 GiB", (d2v *)arena, (d2v *)(arena + ((size_t)32 << 30)), n);
    run_set("64 GiB arena, y at +16 GiB+4 KiB", (d2v *)arena, (d2v *)(arena + ((size_t)16 << 30) + 4096), n);
    // small footprint, the same number of bytes: 16 x the first GiB (does the rate depend on the footprint or on the
    // bytes?) -- launched as 16 back-to-back copies of 2^26 amplitudes
    {
      const size_t m = (size_t)1 << 26;
      d2v *x = (d2v *)arena, *y = (d2v *)(arena + m * 16);
      const double t = time_ms([&] {
        for (int r = 0; r < 16; ++r) hipLaunchKernelGGL(copy4nt_kernel, dim3(65536), dim3(256), 0, 0, x, y, m);
      }, 5);
      printf("%-34s 16 x 2^26 copy4 nt grid=65536 %8.3f ms  %7.1f GB/s\n", "arena, 2 GiB footprint", t, 32.0 * m * 16 / 1e9 / t * 1e3);
    }
    CK(hipFree(arena));
  }
  // (3) stream-ordered pool
  {
    const size_t n = (size_t)1 << 30;
    d2v *x, *y;
    CK(hipMallocAsync((void **)&x, n * 16, 0));
    CK(hipMallocAsync((void **)&y, n * 16, 0));
    CK(hipMemsetAsync(x, 0, n * 16, 0));
    CK(hipMemsetAsync(y, 0, n * 16, 0));
    run_set("hipMallocAsync pool", x, y, n);
    CK(hipFreeAsync(x, 0));
    CK(hipFreeAsync(y, 0));
  }
  return 0;
}
