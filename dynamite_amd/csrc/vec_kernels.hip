// Vector kernels the Krylov loops need (what PETSc Vec / SLEPc BV provide to
// SLEPc's MFN/EPS in the reference): fills, axpby, fused multi-dot and
// multi-axpy over a block of basis vectors, in-place basis rotation, and a
// counter-based normal generator.  All are HBM-streaming; 16 B per lane,
// grid-stride, deterministic two-stage reductions (no atomics).
#include <algorithm>

#include "kernels.h"
#include "philox.h"

namespace dnm {

typedef double2 c128;
typedef double d2v __attribute__((ext_vector_type(2)));

// streaming (non-temporal) accesses for data touched once per sweep
__device__ __forceinline__ c128 ld_stream(const c128 *p) {
  d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ void st_stream(c128 *p, c128 a) {
  d2v v = {a.x, a.y};
  __builtin_nontemporal_store(v, reinterpret_cast<d2v *>(p));
}

constexpr int VNT = 256;
// Launch shapes, measured at 2^30 amplitudes (tools/vec_probe.hip, profiles/r03_vec_probe.txt).  A sweep that WRITES
// streams fastest with one element per thread and no loop at all: x, y -> y at 6.48 TB/s against 4.9 with a
// grid-stride loop over 2048 workgroups (5.3 over 65536); a read-only reduction reaches its 7.0-7.1 TB/s from 8192
// workgroups on.  So: the streaming kernels get one workgroup per 256 elements (their loops run once), the multi-dot
// 16384 workgroups, and the fused Lanczos sweeps one element per thread with a two-level sum of their partials.
// (HIP rejects launches of 2^32 or more threads per grid dimension: the cap keeps gridDim.x * VNT below that, the
// grid-stride loops take over from 2^32 - VNT elements -- a 64 GiB local vector -- on)
constexpr int64_t VMAX_BLOCKS = (((int64_t)1 << 32) - 1) / VNT + 1;
constexpr int64_t VRED_BLOCKS = 16384;                  // multi-dot: partials are 2 * nv doubles per workgroup
constexpr int VRED2 = 1024;                             // second-level partial sums of the one-element-per-thread sweeps

static inline unsigned vgrid(int64_t n, int per_thread = 1, int64_t cap = VMAX_BLOCKS - 1) {
  int64_t nb = (n + (int64_t)VNT * per_thread - 1) / ((int64_t)VNT * per_thread);
  if (nb < 1) nb = 1;
  return (unsigned)(nb < cap ? nb : cap);
}

__global__ void __launch_bounds__(VNT) set_kernel(c128 *x, int64_t n, double re, double im) {
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT)
    x[i] = make_double2(re, im);
}

__global__ void __launch_bounds__(VNT) scale_kernel(c128 *x, int64_t n, double re, double im) {
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT) {
    c128 v = ld_stream(x + i);
    st_stream(x + i, make_double2(re * v.x - im * v.y, re * v.y + im * v.x));
  }
}

__global__ void __launch_bounds__(VNT)
axpby_kernel(c128 *y, const c128 *__restrict__ x, int64_t n, double are, double aim, double bre,
             double bim, int beta_zero) {
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT) {
    c128 xv = ld_stream(x + i);
    double rr = are * xv.x - aim * xv.y, ri = are * xv.y + aim * xv.x;
    if (!beta_zero) {
      c128 yv = ld_stream(y + i);
      rr += bre * yv.x - bim * yv.y;
      ri += bre * yv.y + bim * yv.x;
    }
    st_stream(y + i, make_double2(rr, ri));
  }
}

// y = x: one element per thread streams at 6.5 TB/s, hipMemcpyAsync device-to-device at 4.7 (profiles/r03_vec_abi.txt)
__global__ void __launch_bounds__(VNT) copy_kernel(c128 *__restrict__ y, const c128 *__restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT) st_stream(y + i, ld_stream(x + i));
}
int vk_copy(void *y, const void *x, int64_t n, hipStream_t st) {
  if (n <= 0 || x == y) return 0;
  hipLaunchKernelGGL(copy_kernel, dim3(vgrid(n)), dim3(VNT), 0, st, (c128 *)y, (const c128 *)x, n);
  DNM_HIP(hipGetLastError());
  return 0;
}

int vk_set(void *x, int64_t n, double re, double im, hipStream_t st) {
  hipLaunchKernelGGL(set_kernel, dim3(vgrid(n)), dim3(VNT), 0, st, (c128 *)x, n, re, im);
  DNM_HIP(hipGetLastError());
  return 0;
}
int vk_scale(void *x, int64_t n, double re, double im, hipStream_t st) {
  hipLaunchKernelGGL(scale_kernel, dim3(vgrid(n)), dim3(VNT), 0, st, (c128 *)x, n, re, im);
  DNM_HIP(hipGetLastError());
  return 0;
}
int vk_axpby(void *y, const void *x, int64_t n, double are, double aim, double bre, double bim,
             hipStream_t st) {
  int bz = (bre == 0.0 && bim == 0.0);
  hipLaunchKernelGGL(axpby_kernel, dim3(vgrid(n)), dim3(VNT), 0, st, (c128 *)y, (const c128 *)x, n,
                     are, aim, bre, bim, bz);
  DNM_HIP(hipGetLastError());
  return 0;
}

// ---- Philox-4x32-10 counter-based generator (philox.h) -----------------------
__global__ void __launch_bounds__(VNT)
random_kernel(c128 *x, int64_t n, uint64_t seed, int64_t offset, int swz) {
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT) {
    uint64_t ctr = (uint64_t)(offset + vec_pos(i, swz));     // the element stored at position i (involution)
    x[i] = philox_normal(ctr, seed);
  }
}

int vk_random(void *x, int64_t n, uint64_t seed, int64_t offset, hipStream_t st, int swz) {
  hipLaunchKernelGGL(random_kernel, dim3(vgrid(n)), dim3(VNT), 0, st, (c128 *)x, n, seed, offset, swz);
  DNM_HIP(hipGetLastError());
  return 0;
}

// dst[i] = src[i ^ sw(i)]: 256-byte runs move as units, both sides coalesced
__global__ void __launch_bounds__(VNT) swizzle_copy_kernel(c128 *dst, const c128 *__restrict__ src, int64_t n, int swz) {
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT)
    st_stream(dst + i, ld_stream(src + vec_pos(i, swz)));
}
// real-packed vectors (DNM_MAT_REAL_PACKED): element j of src holds the real amplitudes of indices 2j (.x) and 2j + 1
// (.y); dst is the complex128 vector of 2 n elements they stand for (imaginary parts zero), each in its own layout
__global__ void __launch_bounds__(VNT) unpack_real_kernel(c128 *dst, const c128 *__restrict__ src, int64_t n2, int swz_src,
                                                          int swz_dst) {
  for (int64_t p = (int64_t)blockIdx.x * VNT + threadIdx.x; p < n2; p += (int64_t)gridDim.x * VNT) {
    const int64_t i = vec_pos(p, swz_dst);               // the index stored at position p (the map is an involution)
    const c128 v = src[vec_pos(i >> 1, swz_src)];
    st_stream(dst + p, make_double2((i & 1) ? v.y : v.x, 0.0));
  }
}
int vk_unpack_real(void *dst, const void *src, int64_t n_packed, int swz_src, int swz_dst, hipStream_t st) {
  hipLaunchKernelGGL(unpack_real_kernel, dim3(vgrid(2 * n_packed)), dim3(VNT), 0, st, (c128 *)dst, (const c128 *)src,
                     2 * n_packed, swz_src, swz_dst);
  DNM_HIP(hipGetLastError());
  return 0;
}
int vk_swizzle_copy(void *dst, const void *src, int64_t n, int swz, hipStream_t st) {
  hipLaunchKernelGGL(swizzle_copy_kernel, dim3(vgrid(n)), dim3(VNT), 0, st, (c128 *)dst, (const c128 *)src, n, swz);
  DNM_HIP(hipGetLastError());
  return 0;
}

// ---- fused multi-dot: h[j] = sum_i conj(V_j[i]) w[i] -------------------------
int vk_mdot_blocks(int64_t n) {
  // 4096 elements per workgroup, at least 2048 workgroups (256 CUs x 8) and at most VRED_BLOCKS: the second stage reads
  // every partial, which shows below 2^26 elements
  const int64_t want = std::max<int64_t>(2048, n / 4096);
  return (int)vgrid(n, 4, std::min<int64_t>(want, VRED_BLOCKS));
}
int vk_sweep_blocks(int64_t n) { return (int)vgrid(n); }
size_t vk_sweep_scratch(int64_t n, int ncols) { return ((size_t)vgrid(n) + 1 + VRED2) * (size_t)ncols; }

template <int NV>
__global__ void __launch_bounds__(VNT)
mdot_kernel(const c128 *__restrict__ V, int64_t ldv, const c128 *__restrict__ w, int64_t n,
            double *__restrict__ partials, int nv_total, int j0) {
  double sr[NV], si[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) sr[j] = si[j] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT) {
    const c128 wv = ld_stream(w + i);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const c128 v = ld_stream(V + (int64_t)(j0 + j) * ldv + i);
      sr[j] = fma(v.x, wv.x, sr[j]);
      sr[j] = fma(v.y, wv.y, sr[j]);
      si[j] = fma(v.x, wv.y, si[j]);
      si[j] = fma(-v.y, wv.x, si[j]);
    }
  }
  __shared__ double red[VNT / 64][2 * NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    for (int off = 32; off > 0; off >>= 1) {
      sr[j] += __shfl_xor(sr[j], off, 64);
      si[j] += __shfl_xor(si[j], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6][2 * j] = sr[j];
      red[threadIdx.x >> 6][2 * j + 1] = si[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * NV) {
    double s = 0.0;
    for (int wv = 0; wv < VNT / 64; ++wv) s += red[wv][threadIdx.x];
    partials[(int64_t)blockIdx.x * 2 * nv_total + 2 * j0 + threadIdx.x] = s;
  }
}

// second stage: out[c] = sum_b partials[b][c]
__global__ void __launch_bounds__(VNT)
reduce_partials_kernel(const double *__restrict__ partials, int nblocks, int ncols,
                       double *__restrict__ out) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += VNT) s += partials[(int64_t)b * ncols + c];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  __shared__ double red[VNT / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int wv = 0; wv < VNT / 64; ++wv) t += red[wv];
    out[c] = t;
  }
}

// first level for many partials: workgroup (c, g) sums the g-th of G slices of column c into tmp[g * ncols + c]
__global__ void __launch_bounds__(VNT)
reduce_slices_kernel(const double *__restrict__ partials, int64_t nblocks, int ncols, int G, double *__restrict__ tmp) {
  const int c = blockIdx.x % ncols, g = blockIdx.x / ncols;
  const int64_t chunk = (nblocks + G - 1) / G, b0 = g * chunk, b1 = b0 + chunk < nblocks ? b0 + chunk : nblocks;
  double s = 0.0;
  for (int64_t b = b0 + threadIdx.x; b < b1; b += VNT) s += partials[b * ncols + c];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  __shared__ double red[VNT / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int wv = 0; wv < VNT / 64; ++wv) t += red[wv];
    tmp[(int64_t)g * ncols + c] = t;
  }
}

// sums of the columns of partials[nblocks][ncols] -> out[ncols]; tmp: VRED2 * ncols doubles (used beyond 8192 rows)
static void reduce_columns(const double *partials, int64_t nblocks, int ncols, double *out, double *tmp, hipStream_t st) {
  if (nblocks <= 8192) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ncols), dim3(VNT), 0, st, partials, (int)nblocks, ncols, out);
    return;
  }
  hipLaunchKernelGGL(reduce_slices_kernel, dim3(ncols * VRED2), dim3(VNT), 0, st, partials, nblocks, ncols, VRED2, tmp);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(ncols), dim3(VNT), 0, st, (const double *)tmp, VRED2, ncols, out);
}

// partials_dev: [nblocks * 2*nv] scratch followed by [2*nv] results
int vk_mdot(const void *V, int64_t ldv, int nv, const void *w, int64_t n, double *partials_dev,
            hipStream_t st) {
  const unsigned nb = (unsigned)vk_mdot_blocks(n);
  const c128 *Vp = (const c128 *)V;
  const c128 *wp = (const c128 *)w;
  int j0 = 0;
  while (j0 < nv) {
    int rem = nv - j0;
    if (rem >= 8) {
      hipLaunchKernelGGL((mdot_kernel<8>), dim3(nb), dim3(VNT), 0, st, Vp, ldv, wp, n, partials_dev, nv, j0);
      j0 += 8;
    } else if (rem >= 4) {
      hipLaunchKernelGGL((mdot_kernel<4>), dim3(nb), dim3(VNT), 0, st, Vp, ldv, wp, n, partials_dev, nv, j0);
      j0 += 4;
    } else if (rem >= 2) {
      hipLaunchKernelGGL((mdot_kernel<2>), dim3(nb), dim3(VNT), 0, st, Vp, ldv, wp, n, partials_dev, nv, j0);
      j0 += 2;
    } else {
      hipLaunchKernelGGL((mdot_kernel<1>), dim3(nb), dim3(VNT), 0, st, Vp, ldv, wp, n, partials_dev, nv, j0);
      j0 += 1;
    }
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(2 * nv), dim3(VNT), 0, st, partials_dev, (int)nb,
                     2 * nv, partials_dev + (int64_t)nb * 2 * nv);
  DNM_HIP(hipGetLastError());
  return 0;
}

// ---- fused multi-axpy: w += sum_j c[j] V_j ----------------------------------
__global__ void __launch_bounds__(VNT)
maxpy_kernel(c128 *w, const c128 *__restrict__ V, int64_t ldv, int nv, int64_t n,
             const double *__restrict__ c) {
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT) {
    c128 acc = ld_stream(w + i);
    for (int j = 0; j < nv; ++j) {
      const double cr = c[2 * j], ci = c[2 * j + 1];
      const c128 v = ld_stream(V + (int64_t)j * ldv + i);
      acc.x = fma(cr, v.x, acc.x);
      acc.x = fma(-ci, v.y, acc.x);
      acc.y = fma(cr, v.y, acc.y);
      acc.y = fma(ci, v.x, acc.y);
    }
    st_stream(w + i, acc);
  }
}

int vk_maxpy(void *w, const void *V, int64_t ldv, int nv, int64_t n, const double *c_dev,
             hipStream_t st) {
  hipLaunchKernelGGL(maxpy_kernel, dim3(vgrid(n)), dim3(VNT), 0, st, (c128 *)w, (const c128 *)V, ldv,
                     nv, n, c_dev);
  DNM_HIP(hipGetLastError());
  return 0;
}

// ---- in-place basis rotation V[:, 0:nout) = V[:, 0:nin) S --------------------
// A workgroup stages 64 rows x nin columns in LDS, then writes the nout
// combinations back over the same rows.
constexpr int BU_ROWS = 64;

__global__ void __launch_bounds__(VNT)
basis_update_kernel(c128 *V, int64_t ldv, int nin, int nout, int64_t n,
                    const double *__restrict__ S, int rows) {
  // `rows` (a power of two <= BU_ROWS) rows of all nin vectors are staged per step: fewer rows when the basis is
  // too wide for 64 of them (nin * rows * 16 B <= 160 KB)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c128 *rowsbuf = reinterpret_cast<c128 *>(smem);   // [nin][rows]
  const int lane = threadIdx.x & (rows - 1), grp = threadIdx.x / rows, ngrp = VNT / rows;
  for (int64_t r0 = (int64_t)blockIdx.x * rows; r0 < n; r0 += (int64_t)gridDim.x * rows) {
    const int64_t row = r0 + lane;
    for (int j = grp; j < nin; j += ngrp)
      if (row < n) rowsbuf[j * rows + lane] = ld_stream(V + (int64_t)j * ldv + row);
    __syncthreads();
    for (int o = grp; o < nout; o += ngrp) {
      double ar = 0.0, ai = 0.0;
      for (int j = 0; j < nin; ++j) {
        const double sr = S[2 * ((int64_t)o * nin + j)], si = S[2 * ((int64_t)o * nin + j) + 1];
        const c128 v = rowsbuf[j * rows + lane];
        ar = fma(sr, v.x, ar);
        ar = fma(-si, v.y, ar);
        ai = fma(sr, v.y, ai);
        ai = fma(si, v.x, ai);
      }
      if (row < n) st_stream(V + (int64_t)o * ldv + row, make_double2(ar, ai));
    }
    __syncthreads();
  }
}

// Register variant for nout <= 16 (every thick restart with the default ncv): one thread per row keeps the
// nout outputs in registers while it streams the nin inputs of its row in chunks of 8 columns -- all loads of a
// chunk are in flight together and nothing goes through LDS.  In place: a row's outputs are written only after
// all of its inputs have been read.
constexpr int BU_MAXOUT = 16;
__global__ void __launch_bounds__(VNT)
basis_update_reg_kernel(c128 *V, int64_t ldv, int nin, int nout, int64_t n, const double *__restrict__ S) {
  for (int64_t row = (int64_t)blockIdx.x * VNT + threadIdx.x; row < n; row += (int64_t)gridDim.x * VNT) {
    double ar[BU_MAXOUT], ai[BU_MAXOUT];
#pragma unroll
    for (int o = 0; o < BU_MAXOUT; ++o) ar[o] = ai[o] = 0.0;
    for (int j0 = 0; j0 < nin; j0 += 8) {
      c128 v[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        v[jj] = (j0 + jj < nin) ? ld_stream(V + (int64_t)(j0 + jj) * ldv + row) : make_double2(0.0, 0.0);
#pragma unroll
      for (int o = 0; o < BU_MAXOUT; ++o) {
        if (o < nout) {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            if (j0 + jj < nin) {
              const double sr = S[2 * ((int64_t)o * nin + j0 + jj)], si = S[2 * ((int64_t)o * nin + j0 + jj) + 1];
              ar[o] = fma(sr, v[jj].x, ar[o]);
              ar[o] = fma(-si, v[jj].y, ar[o]);
              ai[o] = fma(sr, v[jj].y, ai[o]);
              ai[o] = fma(si, v[jj].x, ai[o]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < BU_MAXOUT; ++o)
      if (o < nout) st_stream(V + (int64_t)o * ldv + row, make_double2(ar[o], ai[o]));
  }
}

int vk_basis_update(void *V, int64_t ldv, int nin, int nout, int64_t n, const double *S_dev,
                    hipStream_t st) {
  DNM_CHECK(nin >= 1 && nout >= 0 && nout <= nin, "basis_update: bad shapes");
  if (nout <= BU_MAXOUT) {
    hipLaunchKernelGGL(basis_update_reg_kernel, dim3(vgrid(n)), dim3(VNT), 0, st, (c128 *)V, ldv, nin, nout, n,
                       S_dev);
    DNM_HIP(hipGetLastError());
    return 0;
  }
  int rows = BU_ROWS;
  while (rows > 1 && (size_t)nin * rows * sizeof(c128) > 160 * 1024) rows >>= 1;
  const size_t lds = (size_t)nin * rows * sizeof(c128);
  DNM_CHECK(lds <= 160 * 1024, "basis_update: too many vectors for one LDS stage");
  static size_t attr = 0;
  if (lds > attr) {
    DNM_HIP(hipFuncSetAttribute((const void *)basis_update_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  int64_t nb = (n + rows - 1) / rows;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(basis_update_kernel, dim3((unsigned)nb), dim3(VNT), lds, st, (c128 *)V, ldv, nin,
                     nout, n, S_dev, rows);
  DNM_HIP(hipGetLastError());
  return 0;
}

// ---- fused Lanczos update: p -= a v + b u, partial sums of |p|^2 ----------------
// (one sweep instead of multi-axpy + norm; u may be null)
__global__ void __launch_bounds__(VNT)
lanczos_update_kernel(c128 *p, const c128 *__restrict__ v, const c128 *__restrict__ u, int64_t n, double are,
                      double aim, double b, double scale, double *__restrict__ partials) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT) {
    c128 acc = ld_stream(p + i);
    const c128 vv = ld_stream(v + i);
    acc.x = fma(-are, vv.x, acc.x);
    acc.x = fma(aim, vv.y, acc.x);
    acc.y = fma(-are, vv.y, acc.y);
    acc.y = fma(-aim, vv.x, acc.y);
    if (u) {
      const c128 uv = ld_stream(u + i);
      acc.x = fma(-b, uv.x, acc.x);
      acc.y = fma(-b, uv.y, acc.y);
    }
    acc.x *= scale;
    acc.y *= scale;
    st_stream(p + i, acc);
    s = fma(acc.x, acc.x, s);
    s = fma(acc.y, acc.y, s);
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  __shared__ double red[VNT / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int wv = 0; wv < VNT / 64; ++wv) t += red[wv];
    partials[blockIdx.x] = t;
  }
}

// partials_dev: vk_sweep_scratch(n, 1) doubles -- [nblocks] partials, [1] result, second-level sums;
// nblocks = vk_sweep_blocks(n): one element per thread
int vk_lanczos_update(void *p, const void *v, const void *u, int64_t n, double are, double aim, double b,
                      double scale, double *partials_dev, hipStream_t st) {
  const unsigned nb = (unsigned)vk_sweep_blocks(n);
  hipLaunchKernelGGL(lanczos_update_kernel, dim3(nb), dim3(VNT), 0, st, (c128 *)p, (const c128 *)v,
                     (const c128 *)u, n, are, aim, b, scale, partials_dev);
  reduce_columns(partials_dev, nb, 1, partials_dev + nb, partials_dev + nb + 1, st);
  DNM_HIP(hipGetLastError());
  return 0;
}

// y = ys y - b z (z may be null; ys == 1 and no z: y untouched) with the partial sums of conj(x) y (re, im) and |y|^2
// of the result
__global__ void __launch_bounds__(VNT)
lanczos_dot_kernel(c128 *y, const c128 *__restrict__ z, const c128 *__restrict__ x, int64_t n, double b, double ys,
                   double *__restrict__ partials) {
  double dr = 0.0, di = 0.0, dn = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * VNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * VNT) {
    c128 acc = ld_stream(y + i);
    if (z || ys != 1.0) {
      acc.x *= ys;
      acc.y *= ys;
      if (z) {
        const c128 zv = ld_stream(z + i);
        acc.x = fma(-b, zv.x, acc.x);
        acc.y = fma(-b, zv.y, acc.y);
      }
      st_stream(y + i, acc);
    }
    const c128 xv = ld_stream(x + i);
    dr = fma(xv.x, acc.x, dr);
    dr = fma(xv.y, acc.y, dr);
    di = fma(xv.x, acc.y, di);
    di = fma(-xv.y, acc.x, di);
    dn = fma(acc.x, acc.x, dn);
    dn = fma(acc.y, acc.y, dn);
  }
  for (int off = 32; off > 0; off >>= 1) {
    dr += __shfl_xor(dr, off, 64);
    di += __shfl_xor(di, off, 64);
    dn += __shfl_xor(dn, off, 64);
  }
  __shared__ double red[3 * (VNT / 64)];
  if ((threadIdx.x & 63) == 0) {
    red[3 * (threadIdx.x >> 6)] = dr;
    red[3 * (threadIdx.x >> 6) + 1] = di;
    red[3 * (threadIdx.x >> 6) + 2] = dn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tr = 0.0, ti = 0.0, tn = 0.0;
    for (int wv = 0; wv < VNT / 64; ++wv) { tr += red[3 * wv]; ti += red[3 * wv + 1]; tn += red[3 * wv + 2]; }
    partials[3 * (int64_t)blockIdx.x] = tr;
    partials[3 * (int64_t)blockIdx.x + 1] = ti;
    partials[3 * (int64_t)blockIdx.x + 2] = tn;
  }
}

// partials_dev: vk_sweep_scratch(n, 3) doubles -- [3 * nblocks] partials, [3] results, second-level sums;
// nblocks = vk_sweep_blocks(n)
int vk_lanczos_dot(void *y, const void *z, const void *x, int64_t n, double b, double *partials_dev,
                   hipStream_t st, double yscale) {
  const unsigned nb = (unsigned)vk_sweep_blocks(n);
  hipLaunchKernelGGL(lanczos_dot_kernel, dim3(nb), dim3(VNT), 0, st, (c128 *)y, (const c128 *)z, (const c128 *)x, n,
                     b, yscale, partials_dev);
  reduce_columns(partials_dev, nb, 3, partials_dev + 3 * (int64_t)nb, partials_dev + 3 * (int64_t)nb + 3, st);
  DNM_HIP(hipGetLastError());
  return 0;
}

// out[c] = sum of column c of partials[nblocks][ncols]; tmp (vk_reduce_scratch(ncols) doubles, may be null) lets many
// partials be summed in two levels (one workgroup per column reads 2^18 partials of a tiled pass in 0.2 ms)
size_t vk_reduce_scratch(int ncols) { return (size_t)VRED2 * (size_t)ncols; }
int vk_reduce_partials(const double *partials, int nblocks, int ncols, double *out, hipStream_t st, double *tmp) {
  if (tmp && nblocks > 8192) {
    hipLaunchKernelGGL(reduce_slices_kernel, dim3(ncols * VRED2), dim3(VNT), 0, st, partials, (int64_t)nblocks, ncols, VRED2, tmp);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ncols), dim3(VNT), 0, st, (const double *)tmp, VRED2, ncols, out);
  } else {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ncols), dim3(VNT), 0, st, partials, nblocks, ncols, out);
  }
  DNM_HIP(hipGetLastError());
  return 0;
}

int vk_norm2_partials(const void *x, int64_t n, double *partials_dev, hipStream_t st) {
  return vk_mdot(x, n, 1, x, n, partials_dev, st);
}

}  // namespace dnm
