// L2 set-mapping probe: every workgroup repeatedly reads the same N cache lines, whose
// byte addresses are formed by depositing the line index into a chosen set of address
// bits.  If the N lines spread over the L2 sets the re-reads hit (>> fabric bandwidth);
// if they alias, the rate drops to the fabric/HBM rate.
//   hipcc --offload-arch=gfx950 -O3 tools/l2map_probe.hip -o gpurun_out/l2map_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) probe(const double2 *base, uint64_t bitmask, int logn, int iters, double *sink) {
  extern __shared__ uint32_t tab[];            // line offsets in 128 B units
  const int N = 1 << logn;
  for (int i = threadIdx.x; i < N; i += 256) {
    uint64_t m = bitmask, off = 0;
    int v = i;
    while (m && v) {
      const uint64_t low = m & (~m + 1);
      if (v & 1) off |= low;
      v >>= 1;
      m ^= low;
    }
    tab[i] = (uint32_t)(off >> 7);
  }
  __syncthreads();
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  unsigned idx = (unsigned)(wave * 2654435761u) >> 7;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    const unsigned l = (idx + (lane >> 3)) & (N - 1);
    const double2 v = base[(size_t)tab[l] * 8 + (lane & 7)];
    acc += v.x + v.y;
    idx += 8;
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

int main(int argc, char **argv) {
  const size_t bytes = (size_t)16 << 30;
  double2 *buf;
  double *sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 8));
  CK(hipMemset(buf, 0, bytes));
  struct Cfg { const char *name; std::vector<int> bits; };
  auto rng = [](int a, int b) { std::vector<int> v; for (int i = a; i < b; ++i) v.push_back(i); return v; };
  auto cat = [](std::vector<int> a, std::vector<int> b) { a.insert(a.end(), b.begin(), b.end()); return a; };
  std::vector<Cfg> cfgs = {
      {"contiguous 7..", rng(7, 34)},
      {"{7} + 21..", cat({7}, rng(21, 34))},
      {"{7,8} + 21..", cat({7, 8}, rng(21, 34))},
      {"{7,8,9} + 21..", cat({7, 8, 9}, rng(21, 34))},
      {"{7} + 16..", cat({7}, rng(16, 34))},
      {"{7} + 12..", cat({7}, rng(12, 34))},
      {"{7} + 25..", cat({7}, rng(25, 34))},
      {"{7} + 21..28 + 29..", cat({7}, rng(21, 34))},
      {"{7..11} + 21..", cat(rng(7, 12), rng(21, 34))},
      {"{7..14} + 21..", cat(rng(7, 15), rng(21, 34))},
  };
  const int grid = 2048, iters = 2048;
  for (auto &c : cfgs) {
    for (int logn = 8; logn <= 14; ++logn) {
      if (logn > (int)c.bits.size()) break;
      uint64_t mask = 0;
      for (int i = 0; i < logn; ++i) mask |= 1ull << c.bits[i];
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(probe, dim3(grid), dim3(256), (size_t)4 << logn, 0, buf, mask, logn, iters, sink);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe, dim3(grid), dim3(256), (size_t)4 << logn, 0, buf, mask, logn, iters, sink);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double gb = (double)grid * 4 * iters * 1024 / 1e9;
      printf("%-22s N=2^%-2d (%6.0f KB)  %8.3f ms  %9.1f GB/s\n", c.name, logn, (double)(128 << logn) / 1024, ms, gb / ms * 1e3);
      fflush(stdout);
    }
  }
  return 0;
}
