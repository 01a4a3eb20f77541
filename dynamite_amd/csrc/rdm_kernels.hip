// Reduced density matrix  rho[a,b] = sum_tr psi(a,tr) conj(psi(b,tr))  of a state on any
// subspace (reference: rdm_<SUBSPACE>, bpetsc_template_1.c:87-165, serial on rank 0).
//
// psi(a,tr) = x[S2I(deposit(a, kept bits) | deposit(tr, traced bits))], zero outside the
// subspace.  The sum is a Hermitian rank-T update (ZHERK) of a K x K matrix, K = 2^k, with
// the operand gathered on the fly -- the K x T matrix is never materialised.  One workgroup
// owns a TM x TM tile of the lower triangle and a slice of the traced configurations:
// chunks of 1024 amplitudes per operand are staged in LDS, every thread keeps a 4 x 4
// block of complex accumulators in registers; for tiles smaller than 64 x 64 the threads
// split the traced index among themselves and combine at the end (wave shuffles, then LDS).
// Partial tiles go to a scratch buffer and a second kernel sums the slices and mirrors the
// upper triangle, so the result is deterministic (no atomics).
#include "kernels.h"

namespace dnm {

typedef double2 c128;

constexpr int RDM_NT = 256;
constexpr int RDM_STAGE = 1024;   // amplitudes per operand per chunk

template <int ST>
__device__ __forceinline__ c128 rdm_fetch(const c128 *__restrict__ x, uint64_t state, const SubView &sub) {
  const int64_t idx = Sub<ST>::s2i((int64_t)state, sub);
  return idx >= 0 ? x[vec_pos(idx, sub.swz)] : make_double2(0.0, 0.0);
}

__device__ __forceinline__ uint64_t rdm_deposit(uint64_t v, const int8_t *len, const int8_t *pos, int nseg) {
  uint64_t out = 0;
  for (int i = 0; i < nseg; ++i) {
    out |= (v & (((uint64_t)1 << len[i]) - 1)) << pos[i];
    v >>= len[i];
  }
  return out;
}

template <int ST, int LOGTM>
__global__ void __launch_bounds__(RDM_NT)
rdm_tile_kernel(const c128 *__restrict__ x, const SubView sub, const RdmGeom geo, int64_t chunks_per_split,
                int ntiles, c128 *__restrict__ partial) {
  constexpr int TM = 1 << LOGTM;
  constexpr int SUBT = TM / 4;            // 4x4 register blocks per tile side
  constexpr int NSUB = SUBT * SUBT;       // threads that cover one tile
  constexpr int G = RDM_NT / NSUB;        // groups splitting the traced index
  constexpr int TK = RDM_STAGE / TM;      // traced configurations per chunk
  __shared__ c128 As[RDM_STAGE];
  __shared__ c128 Bs[RDM_STAGE];
  __shared__ uint64_t pa[TM], pb[TM];

  const int tid = threadIdx.x;
  // tile (ti, tj), tj <= ti, from the linear lower-triangle index
  const int tile = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > tile) --ti;
  const int tj = tile - (int)((int64_t)ti * (ti + 1) / 2);
  const bool diag_tile = ti == tj;
  const int64_t K = (int64_t)1 << geo.k, T = (int64_t)1 << (geo.L - geo.k);
  const int64_t a0 = (int64_t)ti * TM, b0 = (int64_t)tj * TM;

  if (tid < TM) {
    pa[tid] = rdm_deposit((uint64_t)(a0 + tid), geo.klen, geo.kpos, geo.nseg_keep);
    pb[tid] = rdm_deposit((uint64_t)(b0 + tid), geo.klen, geo.kpos, geo.nseg_keep);
  }
  __syncthreads();

  const int sub_id = tid % NSUB, g = tid / NSUB;
  const int ty = sub_id / SUBT, tx = sub_id % SUBT;
  double accr[4][4], acci[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) accr[i][j] = acci[i][j] = 0.0;

  const int64_t nchunks = (T + TK - 1) / TK;
  const int64_t c_begin = (int64_t)blockIdx.y * chunks_per_split;
  int64_t c_end = c_begin + chunks_per_split;
  if (c_end > nchunks) c_end = nchunks;
  const c128 *Bp = diag_tile ? As : Bs;

  for (int64_t c = c_begin; c < c_end; ++c) {
    // stage: element e -> (row r, traced slot t); consecutive threads take consecutive rows
#pragma unroll
    for (int e = tid; e < RDM_STAGE; e += RDM_NT) {
      const int r = e % TM, t = e / TM;
      const int64_t tr = c * TK + t;
      c128 va = make_double2(0.0, 0.0), vb = va;
      if (tr < T) {
        const uint64_t pt = rdm_deposit((uint64_t)tr, geo.tlen, geo.tpos, geo.nseg_tr);
        if (a0 + r < K) va = rdm_fetch<ST>(x, pa[r] | pt, sub);
        if (!diag_tile && b0 + r < K) vb = rdm_fetch<ST>(x, pb[r] | pt, sub);
      }
      As[e] = va;
      if (!diag_tile) Bs[e] = vb;
    }
    __syncthreads();
    for (int t = g; t < TK; t += G) {
      c128 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[t * TM + ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bp[t * TM + tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // a * conj(b)
          accr[i][j] = fma(a[i].x, b[j].x, accr[i][j]);
          accr[i][j] = fma(a[i].y, b[j].y, accr[i][j]);
          acci[i][j] = fma(a[i].y, b[j].x, acci[i][j]);
          acci[i][j] = fma(-a[i].x, b[j].y, acci[i][j]);
        }
    }
    __syncthreads();
  }

  // combine the groups: lanes of a wave that hold the same register block, then the waves
  if (G > 1) {
    if (NSUB < 64) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          for (int off = NSUB; off < 64; off <<= 1) {
            accr[i][j] += __shfl_xor(accr[i][j], off, 64);
            acci[i][j] += __shfl_xor(acci[i][j], off, 64);
          }
    }
    constexpr int LIVE = NSUB < 64 ? NSUB : 64;       // lanes per wave holding distinct blocks
    double *red = reinterpret_cast<double *>(As);     // LIVE * 32 doubles <= 16 KB
    const int wave = tid >> 6, lane = tid & 63;
    for (int w = 1; w < RDM_NT / 64; ++w) {
      __syncthreads();
      if (wave == w && lane < LIVE) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            red[(lane * 16 + i * 4 + j) * 2] = accr[i][j];
            red[(lane * 16 + i * 4 + j) * 2 + 1] = acci[i][j];
          }
      }
      __syncthreads();
      if (wave == 0 && lane < LIVE) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            accr[i][j] += red[(lane * 16 + i * 4 + j) * 2];
            acci[i][j] += red[(lane * 16 + i * 4 + j) * 2 + 1];
          }
      }
    }
  }
  if (tid < NSUB) {
    c128 *out = partial + ((int64_t)blockIdx.y * ntiles + tile) * (TM * TM);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) out[(ty * 4 + i) * TM + tx * 4 + j] = make_double2(accr[i][j], acci[i][j]);
  }
}

// ---- one to three kept spins: a streaming kernel -----------------------------------------------------------------
// K = 2^k <= 8 rows: the whole matrix fits a thread's registers.  A thread walks traced configurations (consecutive
// threads take consecutive configurations: their loads are neighbours wherever the lowest traced bits are), loads the
// K amplitudes psi(a, tr) and adds psi(a) conj(psi(b)) for a >= b to K (K + 1) / 2 complex accumulators; wavefronts
// reduce by shuffles, the workgroup through LDS, and every workgroup writes one slice of the ONE tile (TM = max(4, K))
// that rdm_reduce_kernel / rdm_finalize_kernel then sum -- the scratch format, the slice tree and the mirroring are
// those of the tiled form.  The tiled kernel moves such a state in 8 KB chunks between barriers (1.3-2.5 TB/s of x at
// L = 26); this one is bound by the read.
constexpr int RDM_SMALL_MAXK = 2;      // (three spins: 1.33x on the Full space, 0.76x on SpinConserve -- left to the tiles)
constexpr int rdm_small_cfg(int k) { return k <= 2 ? 4 : 1; }      // configurations a thread has in flight (registers)

template <int ST, int LOGK>
__global__ void __launch_bounds__(RDM_NT)
rdm_small_kernel(const c128 *__restrict__ x, const SubView sub, const RdmGeom geo, c128 *__restrict__ partial) {
  constexpr int K = 1 << LOGK;
  constexpr int TM = K < 4 ? 4 : K;
  constexpr int NP = K * (K + 1) / 2;
  constexpr int RDM_SMALL_CFG = rdm_small_cfg(LOGK);
  __shared__ double red[RDM_NT / 64][2 * NP];
  uint64_t pa[K];
#pragma unroll
  for (int a = 0; a < K; ++a) pa[a] = rdm_deposit((uint64_t)a, geo.klen, geo.kpos, geo.nseg_keep);
  const int64_t T = (int64_t)1 << (geo.L - geo.k);
  double ar[NP], ai[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) ar[p] = ai[p] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * RDM_NT;
  for (int64_t tr0 = (int64_t)blockIdx.x * RDM_NT + threadIdx.x; tr0 < T; tr0 += stride * RDM_SMALL_CFG) {
    c128 psi[RDM_SMALL_CFG][K];
#pragma unroll
    for (int c = 0; c < RDM_SMALL_CFG; ++c) {
      const int64_t tr = tr0 + c * stride;
      const uint64_t pt = rdm_deposit((uint64_t)(tr < T ? tr : 0), geo.tlen, geo.tpos, geo.nseg_tr);
#pragma unroll
      for (int a = 0; a < K; ++a) psi[c][a] = tr < T ? rdm_fetch<ST>(x, pa[a] | pt, sub) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int c = 0; c < RDM_SMALL_CFG; ++c) {
      int p = 0;
#pragma unroll
      for (int a = 0; a < K; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b, ++p) {      // psi(a) conj(psi(b))
          ar[p] = fma(psi[c][a].x, psi[c][b].x, ar[p]);
          ar[p] = fma(psi[c][a].y, psi[c][b].y, ar[p]);
          ai[p] = fma(psi[c][a].y, psi[c][b].x, ai[p]);
          ai[p] = fma(-psi[c][a].x, psi[c][b].y, ai[p]);
        }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ar[p] += __shfl_xor(ar[p], off, 64);
      ai[p] += __shfl_xor(ai[p], off, 64);
    }
    if (lane == 0) {
      red[wave][2 * p] = ar[p];
      red[wave][2 * p + 1] = ai[p];
    }
  }
  __syncthreads();
  // one slice of the tile, both triangles (a diagonal tile is not mirrored by rdm_finalize_kernel): element (a, b)
  // with b > a is the conjugate of (b, a); rows and columns beyond K stay zero
  c128 *out = partial + (int64_t)blockIdx.x * (TM * TM);
  for (int e = threadIdx.x; e < TM * TM; e += RDM_NT) {
    const int a = e / TM, b = e % TM;
    c128 v = make_double2(0.0, 0.0);
    if (a < K && b < K) {
      const int hi = a > b ? a : b, lo = a > b ? b : a;
      const int p = hi * (hi + 1) / 2 + lo;
      for (int w = 0; w < RDM_NT / 64; ++w) {
        v.x += red[w][2 * p];
        v.y += red[w][2 * p + 1];
      }
      if (b > a) v.y = -v.y;
    }
    out[e] = v;
  }
}

// ---- 64 x 64 tiles on the matrix cores (k >= 6) ------------------------------------------------------------
// The same tile, slices and scratch as rdm_tile_kernel<ST, 6>, with the rank-T update done by
// v_mfma_f64_16x16x4_f64: a wavefront owns a 32 x 32 block of the tile (2 x 2 MFMA blocks, real and imaginary
// accumulators: 64 VGPRs) and per four traced configurations reads two A and two B fragments from LDS -- each lane one
// complex amplitude (ds_read_b128: row / column = lane & 15, traced slot = lane >> 4) -- where the VALU form reads
// eight amplitudes per 16 complex products: an eighth of the LDS traffic per flop, and the FMAs leave the vector unit.
//   rho = A B^H:  Re += Ar Br^T + Ai Bi^T,   Im += Ai Br^T + (-Ar) Bi^T      (four real MFMAs per complex block)
// C/D layout of the f64 MFMA (not the f32 one): col = lane & 15, row = (lane >> 4) + 4 * reg.
// The next chunk's amplitudes are gathered into registers while the matrix cores work on the staged one.
typedef double mfma_acc __attribute__((ext_vector_type(4)));

// waves per SIMD the kernel is compiled for / chunk size: 4 waves (128 registers, 8 B/lane of scratch) run 2.4 % faster
// than 3; chunks of 32 traced configurations (64 KB of LDS, two workgroups per CU) 2.7 % slower (GPU session 39)
#ifndef DNM_RDM_WAVES
#define DNM_RDM_WAVES 4
#endif
#ifndef DNM_RDM_MSTAGE
#define DNM_RDM_MSTAGE 1024
#endif
template <int ST>
__global__ void __launch_bounds__(RDM_NT, DNM_RDM_WAVES)
rdm_mfma_kernel(const c128 *__restrict__ x, const SubView sub, const RdmGeom geo, int64_t chunks_per_split,
                int ntiles, c128 *__restrict__ partial) {
  constexpr int TM = 64;
  constexpr int MST = DNM_RDM_MSTAGE;     // amplitudes per operand in a staged chunk
  constexpr int TK = MST / TM;            // traced configurations per chunk (16: four MFMA steps)
  constexpr int EPT = MST / RDM_NT;       // amplitudes per thread and operand in a chunk
  __shared__ c128 As[MST];
  __shared__ c128 Bs[MST];
  __shared__ uint64_t pa[TM], pb[TM];

  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > tile) --ti;
  const int tj = tile - (int)((int64_t)ti * (ti + 1) / 2);
  const bool diag_tile = ti == tj;
  const int64_t K = (int64_t)1 << geo.k, T = (int64_t)1 << (geo.L - geo.k);
  const int64_t a0 = (int64_t)ti * TM, b0 = (int64_t)tj * TM;
  if (tid < TM) {
    pa[tid] = rdm_deposit((uint64_t)(a0 + tid), geo.klen, geo.kpos, geo.nseg_keep);
    pb[tid] = rdm_deposit((uint64_t)(b0 + tid), geo.klen, geo.kpos, geo.nseg_keep);
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;
  mfma_acc re[2][2], im[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) re[i][j] = im[i][j] = mfma_acc{0.0, 0.0, 0.0, 0.0};

  // the slice of traced configurations of this workgroup (rdm_plan counts it in chunks of RDM_STAGE / TM), walked in
  // chunks of TK from its first configuration
  constexpr int64_t PLAN_TK = RDM_STAGE / TM;
  const int64_t tr_begin = (int64_t)blockIdx.y * chunks_per_split * PLAN_TK;
  int64_t tr_end = tr_begin + chunks_per_split * PLAN_TK;
  if (tr_end > T) tr_end = T;
  const int64_t c_begin = 0, c_end = tr_end > tr_begin ? (tr_end - tr_begin + TK - 1) / TK : 0;
  const c128 *Bp = diag_tile ? As : Bs;

  // a thread's amplitudes of a chunk share the row (RDM_NT is a multiple of TM): its part of the state and the
  // bounds tests are loop-invariant; the traced parts of a chunk's TK configurations are deposited once per
  // workgroup (one lane each) instead of once per amplitude
  __shared__ uint64_t pts[2][TK];
  const int rr = tid % TM, t0 = tid / TM;
  const uint64_t par = pa[rr], pbr = pb[rr];
  const bool arow = a0 + rr < K, brow = !diag_tile && b0 + rr < K;
  auto deposit_chunk = [&](int64_t c) {
    if (tid < TK) {
      const int64_t tr = tr_begin + c * TK + tid;
      pts[c & 1][tid] = tr < tr_end ? rdm_deposit((uint64_t)tr, geo.tlen, geo.tpos, geo.nseg_tr) : ~(uint64_t)0;
    }
  };
  c128 va[EPT], vb[EPT];
  auto gather = [&](int64_t c) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const uint64_t pt = pts[c & 1][t0 + i * (RDM_NT / TM)];
      va[i] = vb[i] = make_double2(0.0, 0.0);
      if (pt != ~(uint64_t)0) {
        if (arow) va[i] = rdm_fetch<ST>(x, par | pt, sub);
        if (brow) vb[i] = rdm_fetch<ST>(x, pbr | pt, sub);
      }
    }
  };
  deposit_chunk(c_begin);
  deposit_chunk(c_begin + 1);
  __syncthreads();
  if (c_begin < c_end) gather(c_begin);
  __syncthreads();                              // every wave has read its slots before deposit_chunk(c_begin + 2)
  for (int64_t c = c_begin; c < c_end; ++c) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      As[tid + i * RDM_NT] = va[i];
      if (!diag_tile) Bs[tid + i * RDM_NT] = vb[i];
    }
    deposit_chunk(c + 2);                       // (its slot was last read by gather(c), before this barrier)
    __syncthreads();
    if (c + 1 < c_end) gather(c + 1);           // in flight under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < TK; kk += 4) {
      const int slot = (kk + (lane >> 4)) * TM + (lane & 15);
      c128 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[slot + wy * 32 + i * 16];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bp[slot + wx * 32 + j * 16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, re[i][j], 0, 0, 0);
          re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, re[i][j], 0, 0, 0);
          im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].x, im[i][j], 0, 0, 0);
          im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[i].x, b[j].y, im[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  c128 *out = partial + ((int64_t)blockIdx.y * ntiles + tile) * (TM * TM);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wy * 32 + i * 16 + (lane >> 4) + 4 * r, col = wx * 32 + j * 16 + (lane & 15);
        out[row * TM + col] = make_double2(re[i][j][r], im[i][j][r]);
      }
}

// rho = sum over slices of the partial tiles; the upper triangle is the conjugate transpose
template <int LOGTM>
__global__ void __launch_bounds__(RDM_NT)
rdm_finalize_kernel(const c128 *__restrict__ partial, int ntiles, int nsplit, int64_t K, c128 *__restrict__ rho) {
  constexpr int TM = 1 << LOGTM;
  const int tile = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > tile) --ti;
  const int tj = tile - (int)((int64_t)ti * (ti + 1) / 2);
  for (int e = threadIdx.x; e < TM * TM; e += RDM_NT) {
    const int r = e / TM, cidx = e % TM;
    const int64_t a = (int64_t)ti * TM + r, b = (int64_t)tj * TM + cidx;
    if (a >= K || b >= K) continue;
    // a diagonal tile holds both triangles, accumulated in different orders (MFMA: the imaginary part of (a, b) and
    // of (b, a) add the same products in different sequence): the lower one is taken and mirrored like every other
    // tile, so that rho is Hermitian to the last bit, as the reference's element-by-element sum is
    if (ti == tj && cidx > r) continue;
    double sr = 0.0, si = 0.0;
    for (int s = 0; s < nsplit; ++s) {
      const c128 v = partial[((int64_t)s * ntiles + tile) * (TM * TM) + e];
      sr += v.x;
      si += v.y;
    }
    if (a == b) si = 0.0;     // |psi|^2 sums: the reference's a * conj(a) has no imaginary part either
    rho[a * K + b] = make_double2(sr, si);
    if (a != b) rho[b * K + a] = make_double2(sr, -si);
  }
}

// tree stage of the slice sum: out[s'] = sum of up to `fan` consecutive slices of `in` (whole slices are
// nelem = ntiles * TM * TM amplitudes, consecutive threads read consecutive amplitudes)
constexpr int RDM_FAN = 32;
__global__ void __launch_bounds__(RDM_NT)
rdm_reduce_kernel(const c128 *__restrict__ in, c128 *__restrict__ out, int64_t nelem, int nsplit_in) {
  const int64_t e = (int64_t)blockIdx.x * RDM_NT + threadIdx.x;
  if (e >= nelem) return;
  const int s0 = blockIdx.y * RDM_FAN;
  int s1 = s0 + RDM_FAN;
  if (s1 > nsplit_in) s1 = nsplit_in;
  double sr = 0.0, si = 0.0;
  for (int s = s0; s < s1; ++s) {
    const c128 v = in[(int64_t)s * nelem + e];
    sr += v.x;
    si += v.y;
  }
  out[(int64_t)blockIdx.y * nelem + e] = make_double2(sr, si);
}

template <int ST, int LOGTM>
static int rdm_launch(const c128 *x, const SubView &sub, const RdmGeom &geo, int ntiles, int nsplit,
                      int64_t chunks_per_split, c128 *partial, c128 *rho, hipStream_t st) {
  constexpr int TM = 1 << LOGTM;
  // 64 x 64 tiles run on the matrix cores (DNM_RDM_MFMA=0: the vector-unit form, for comparison)
  static const bool use_mfma = [] { const char *e = knob("DNM_RDM_MFMA"); return !(e && e[0] == '0'); }();
  static const bool use_small = [] { const char *e = knob("DNM_RDM_SMALL"); return !(e && e[0] == '0'); }();
  if (geo.k >= 1 && geo.k <= RDM_SMALL_MAXK && LOGTM <= 3 && use_small) {
    // (nsplit is the number of workgroups of the streaming kernel: rdm_plan)
    switch (geo.k) {
      case 1: hipLaunchKernelGGL((rdm_small_kernel<ST, 1>), dim3((unsigned)nsplit), dim3(RDM_NT), 0, st, x, sub, geo, partial); break;
      case 2: hipLaunchKernelGGL((rdm_small_kernel<ST, 2>), dim3((unsigned)nsplit), dim3(RDM_NT), 0, st, x, sub, geo, partial); break;
      default: hipLaunchKernelGGL((rdm_small_kernel<ST, 3>), dim3((unsigned)nsplit), dim3(RDM_NT), 0, st, x, sub, geo, partial); break;
    }
  } else if (LOGTM == 6 && use_mfma)
    hipLaunchKernelGGL((rdm_mfma_kernel<ST>), dim3((unsigned)ntiles, (unsigned)nsplit), dim3(RDM_NT), 0, st, x, sub,
                       geo, chunks_per_split, ntiles, partial);
  else
    hipLaunchKernelGGL((rdm_tile_kernel<ST, LOGTM>), dim3((unsigned)ntiles, (unsigned)nsplit), dim3(RDM_NT), 0, st, x,
                       sub, geo, chunks_per_split, ntiles, partial);
  // sum the slices by a fan-in-32 tree (ping-pong inside the scratch), then mirror
  const int64_t nelem = (int64_t)ntiles * TM * TM;
  c128 *cur = partial, *nxt = partial + (int64_t)nsplit * nelem;
  while (nsplit > RDM_FAN) {
    const int nout = (nsplit + RDM_FAN - 1) / RDM_FAN;
    hipLaunchKernelGGL(rdm_reduce_kernel, dim3((unsigned)((nelem + RDM_NT - 1) / RDM_NT), (unsigned)nout),
                       dim3(RDM_NT), 0, st, cur, nxt, nelem, nsplit);
    cur = nxt;
    nxt = cur + (int64_t)nout * nelem;
    nsplit = nout;
  }
  hipLaunchKernelGGL((rdm_finalize_kernel<LOGTM>), dim3((unsigned)ntiles), dim3(RDM_NT), 0, st, cur, ntiles,
                     nsplit, (int64_t)1 << geo.k, rho);
  DNM_HIP(hipGetLastError());
  return 0;
}

template <int ST>
static int rdm_dispatch_tm(int logtm, const c128 *x, const SubView &sub, const RdmGeom &geo, int ntiles, int nsplit,
                           int64_t cps, c128 *partial, c128 *rho, hipStream_t st) {
  switch (logtm) {
    case 2: return rdm_launch<ST, 2>(x, sub, geo, ntiles, nsplit, cps, partial, rho, st);
    case 3: return rdm_launch<ST, 3>(x, sub, geo, ntiles, nsplit, cps, partial, rho, st);
    case 4: return rdm_launch<ST, 4>(x, sub, geo, ntiles, nsplit, cps, partial, rho, st);
    case 5: return rdm_launch<ST, 5>(x, sub, geo, ntiles, nsplit, cps, partial, rho, st);
    case 6: return rdm_launch<ST, 6>(x, sub, geo, ntiles, nsplit, cps, partial, rho, st);
  }
  set_error("internal: bad RDM tile size");
  return 1;
}

void rdm_plan(const RdmGeom &geo, int *logtm, int *ntiles, int *nsplit, int64_t *chunks_per_split,
              size_t *partial_bytes) {
  int ltm = geo.k < 2 ? 2 : (geo.k > 6 ? 6 : geo.k);
  const int64_t K = (int64_t)1 << geo.k, T = (int64_t)1 << (geo.L - geo.k);
  {
    static const bool use_small = [] { const char *e = knob("DNM_RDM_SMALL"); return !(e && e[0] == '0'); }();
    if (geo.k >= 1 && geo.k <= RDM_SMALL_MAXK && use_small) {
      // streaming kernel: one tile, one slice per workgroup; a thread takes RDM_SMALL_CFG configurations per trip
      const int64_t per_wg = (int64_t)RDM_NT * rdm_small_cfg(geo.k);
      int64_t ns = (T + per_wg - 1) / per_wg;
      if (ns > 4096) ns = 4096;
      if (ns < 1) ns = 1;
      const int64_t TMs = (int64_t)1 << ltm;
      *logtm = ltm;
      *ntiles = 1;
      *nsplit = (int)ns;
      *chunks_per_split = 0;
      *partial_bytes = ((size_t)ns + (size_t)ns / 31 + 2) * (size_t)(TMs * TMs) * sizeof(c128);
      return;
    }
  }
  const int64_t TM = (int64_t)1 << ltm, TK = RDM_STAGE / TM;
  const int64_t side = (K + TM - 1) / TM;
  const int64_t nt = side * (side + 1) / 2;
  const int64_t nchunks = (T + TK - 1) / TK;
  int64_t ns = (4096 + nt - 1) / nt;          // enough workgroups to fill 256 CUs several times over
  if (ns > nchunks) ns = nchunks;
  if (ns < 1) ns = 1;
  int64_t cps = (nchunks + ns - 1) / ns;
  ns = (nchunks + cps - 1) / cps;
  *logtm = ltm;
  *ntiles = (int)nt;
  *nsplit = (int)ns;
  *chunks_per_split = cps;
  // slices + the intermediate levels of the fan-in-32 sum (ns/32 + ns/1024 + ... < ns/31 + 2)
  *partial_bytes = ((size_t)ns + (size_t)ns / 31 + 2) * (size_t)nt * (size_t)(TM * TM) * sizeof(c128);
}

int launch_rdm(const void *x, const SubView &sub, const RdmGeom &geo, void *partial, void *rho, hipStream_t st) {
  int logtm, ntiles, nsplit;
  int64_t cps;
  size_t pbytes;
  rdm_plan(geo, &logtm, &ntiles, &nsplit, &cps, &pbytes);
  const c128 *xp = (const c128 *)x;
  c128 *pp = (c128 *)partial, *rp = (c128 *)rho;
  switch (sub.type) {
    case DNM_FULL: return rdm_dispatch_tm<DNM_FULL>(logtm, xp, sub, geo, ntiles, nsplit, cps, pp, rp, st);
    case DNM_PARITY: return rdm_dispatch_tm<DNM_PARITY>(logtm, xp, sub, geo, ntiles, nsplit, cps, pp, rp, st);
    case DNM_SPIN_CONSERVE:
      return rdm_dispatch_tm<DNM_SPIN_CONSERVE>(logtm, xp, sub, geo, ntiles, nsplit, cps, pp, rp, st);
    case DNM_EXPLICIT: return rdm_dispatch_tm<DNM_EXPLICIT>(logtm, xp, sub, geo, ntiles, nsplit, cps, pp, rp, st);
  }
  set_error("bad subspace type");
  return 1;
}

}  // namespace dnm
