// hsk_eval.hip -- full-catalogue evaluation for gfx950: exact-fp32 MFMA score GEMM
// (users x item-shard^T + biases), CSR exclusion mask, radix-select top-k, shard merge, rank metrics.
//
// Reference semantics restated: eval/eval.py:237-253 (scores, -inf mask), eval/eval.py:54-99
// (topk(100), k in {100,50,10,5}), eval/metrics.py:4-105 (precision / recall / ndcg).
#include "hsk_common.h"
#include "hsk_gemm_wide_h2.h"
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

typedef float hsk_f32x16 __attribute__((ext_vector_type(16)));

// =============================================================================================
// Score GEMM.  C[r, j] = sum_d U[u[r], d] * I[item_begin + j, d]  (+ biases), fp32 in, fp32 acc.
// Block tile 128 (users) x 128 (items) x 32 (k); 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain per output).  Both operands are
// k-contiguous in HBM; tiles are staged [row][32+4] in LDS (144-B rows: ds_read_b128 conflict-free)
// and each lane half h reads k = 16h .. 16h+15 of the tile, four floats per ds_read_b128, so MFMA
// step t multiplies k = t (half 0) and k = 16+t (half 1): a fixed permutation of the k order.
// =============================================================================================
#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 32
#define GEMM_LDS_STRIDE (GEMM_BK + 4)

template <bool VEC4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_score_gemm(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                    const float* __restrict__ Ib, const float* __restrict__ Ub,
                                                    const float* __restrict__ gb, int n_users, int D,
                                                    const int64_t* __restrict__ u_idx, int n_rows, long long item_begin,
                                                    int item_count, float* __restrict__ C, int32_t* status) {
  __shared__ __attribute__((aligned(16))) float As[GEMM_BM * GEMM_LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[GEMM_BN * GEMM_LDS_STRIDE];
  __shared__ int urow[GEMM_BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GEMM_BM;
  const int n0 = blockIdx.x * GEMM_BN;

  if (tid < GEMM_BM) {
    int r = m0 + tid;
    int u = 0;
    if (r < n_rows) {
      long long uu = u_idx[r];
      if (uu < 0 || uu >= n_users) {
        if (status) atomicOr(status, HSK_STATUS_BAD_INDEX);
        uu = 0;
      }
      u = (int)uu;
    }
    urow[tid] = u;
  }
  __syncthreads();

  hsk_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // staging assignment: 8 threads cover the 32 floats (128 B) of one row; 32 rows per pass, 4 passes
  const int srow = tid >> 3;
  const int scol = (tid & 7) * 4;
  const int half = lane >> 5;
  const int l32 = lane & 31;

  // Register-staged double buffering: the global loads of k-tile t+1 are issued before the MFMAs of tile t and
  // land in registers while the matrix cores work; they go to LDS after the tile's last read.  (Loading straight
  // into LDS inside the loop left the memory latency of every tile exposed: 78 TFLOP/s; staged: see DESIGN.md.)
  float4 ra[4], rb[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = srow + pass * 32;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + r < n_rows) {   // A: user rows (gathered through urow)
        const float* src = Uw + (long long)urow[r] * D + k0 + scol;
        if (VEC4) {
          if (k0 + scol < D) va = *reinterpret_cast<const float4*>(src);
        } else {
          if (k0 + scol + 0 < D) va.x = src[0];
          if (k0 + scol + 1 < D) va.y = src[1];
          if (k0 + scol + 2 < D) va.z = src[2];
          if (k0 + scol + 3 < D) va.w = src[3];
        }
      }
      if (n0 + r < item_count) {   // B: item rows of the shard
        const float* src = Iw + (item_begin + n0 + r) * (long long)D + k0 + scol;
        if (VEC4) {
          if (k0 + scol < D) vb = *reinterpret_cast<const float4*>(src);
        } else {
          if (k0 + scol + 0 < D) vb.x = src[0];
          if (k0 + scol + 1 < D) vb.y = src[1];
          if (k0 + scol + 2 < D) vb.z = src[2];
          if (k0 + scol + 3 < D) vb.w = src[3];
        }
      }
      ra[pass] = va;
      rb[pass] = vb;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = srow + pass * 32;
      *reinterpret_cast<float4*>(&As[r * GEMM_LDS_STRIDE + scol]) = ra[pass];
      *reinterpret_cast<float4*>(&Bs[r * GEMM_LDS_STRIDE + scol]) = rb[pass];
    }
  };
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int k0 = 0; k0 < D; k0 += GEMM_BK) {
    const bool has_next = k0 + GEMM_BK < D;
    if (has_next) load_tile(k0 + GEMM_BK);

#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + l32) * GEMM_LDS_STRIDE + half * 16 + q * 4]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + j * 32 + l32) * GEMM_LDS_STRIDE + half * 16 + q * 4]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (has_next) {
      store_tile();
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const float gbv = gb ? gb[0] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l32;
      if (col >= item_count) continue;
      const float ib = Ib ? Ib[item_begin + col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rloc = wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        const int row = m0 + rloc;
        if (row < n_rows) {
          float o = acc[i][j][q];
          if (Ub) o += Ub[urow[rloc]];  // reference order: += u_bias, += i_bias, += global_bias
          if (Ib) o += ib;
          if (gb) o += gbv;
          C[(long long)row * item_count + col] = o;
        }
      }
    }
}

// Which arithmetic the score GEMMs use (k_score_gemm* here, k_score_topk* in hsk_eval_fused.hip): 2 (default) two fp16
// pieces per operand / three products on the 256 x 256 kernels wherever the call brings the pieces' scratch and 16-byte
// aligned rows (everything else runs form 1); 1 three bf16 pieces / six products; 0 the exact-fp32 MFMA form.
// HSK_EVAL_X3 in the environment sets the initial value, hsk_eval_set_arith changes it (parity tests compare the forms).
static int g_eval_x3 = -1;
int hsk_eval_x3() {   // also called from hsk_eval_fused.hip
  if (g_eval_x3 < 0) {
    const int v = getenv("HSK_EVAL_X3") ? atoi(getenv("HSK_EVAL_X3")) : 2;
    g_eval_x3 = v < 0 ? 0 : (v > 2 ? 2 : v);
  }
  return g_eval_x3;
}
extern "C" void hsk_eval_set_arith(int form) { g_eval_x3 = form < 0 ? 0 : (form > 2 ? 2 : form); }

// ---------------------------------------------------------------------------------------------
// The same scores from bf16 matrix cores.  Every fp32 operand is cut into three bf16 pieces while its tile is staged into
// LDS (x = x1 + x2 + x3: 8 + 8 + 8 significant bits, each cut exact in fp32); of the nine products per (a, b) pair the six
// of weight >= 2^-16 are kept, each a v_mfma_f32_32x32x16_bf16 into the same fp32 accumulator.  A bf16 x bf16 product is
// exact in fp32 and the dropped terms are <= 2^-24 |a||b|: fp32-GEMM accuracy (measured: 5e-7 of the largest score
// against float64, profiles/probes/gemm_bf16x3.hip) -- at six 32-cycle MFMAs where the fp32 form needs sixteen 64-cycle
// ones per 32 x 32 x 32 block.  Non-finite operands keep their value in the first piece (the other two are zero).
// ---------------------------------------------------------------------------------------------
typedef __bf16 hsk_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hsk_bf16x4 __attribute__((ext_vector_type(4)));
#define GEMM_X3_LDK (GEMM_BK + 8)   // LDS row stride in bf16 elements: 80 B, ds_read_b128 conflict-free

__device__ __forceinline__ void hsk_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float hf = (float)h;
  const bool fin = __builtin_isfinite(hf);   // inf / nan (also a finite x that rounds to bf16 inf): one piece
  const float r1 = fin ? x - hf : 0.f;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// Pre-pass of the PLANES variants: rows [0, n_valid) of `src` (gathered through idx when given, else rows row0 + r) cut
// into their three bf16 pieces ONCE per call, laid out [Dp / 32 k-tiles][n_pad rows][3 pieces][32] -- a 128-row tile's
// pieces of one k-tile are 24 KB of consecutive bytes -- zero beyond the valid rows and beyond D.  One thread per 4
// consecutive elements (D % 4 == 0, 16-byte aligned rows).  The GEMM loops then only move 16-byte chunks global ->
// registers -> LDS: the split (two roundings and two subtractions per element, per tile, in every workgroup) was a sixth
// of the loop (profiles/probes/gemm_bf16x3.hip: 640 -> 545 us at the ml10m eval shape, 575 with this pre-pass).
template <int TK>   // depth of a k-tile: 32 ([k-tile][row][piece][32], the 128 x 128 kernels) or 16 (k_score_gemm_x3_wide)
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                      long long row0, long long n_src_rows, int n_valid, int n_pad, int D,
                                                      int Dp, __bf16* __restrict__ planes) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = Dp / 4;
  const long long r = t / per_row;
  const int k = (int)(t - r * per_row) * 4;
  if (r >= n_pad) return;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (r < n_valid && k < D) {
    long long sr = idx ? idx[r] : row0 + r;
    if (sr < 0 || sr >= n_src_rows) sr = 0;   // (a bad index is reported by the scoring kernel)
    const float4 x = *reinterpret_cast<const float4*>(src + sr * (long long)D + k);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
  hsk_bf16x4 p1, p2, p3;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    __bf16 x1, x2, x3;
    hsk_split3(v[e], x1, x2, x3);
    p1[e] = x1; p2[e] = x2; p3[e] = x3;
  }
  __bf16* dst = planes + ((long long)(k / TK) * n_pad + r) * (3 * TK) + (k % TK);
  *reinterpret_cast<hsk_bf16x4*>(dst) = p1;
  *reinterpret_cast<hsk_bf16x4*>(dst + TK) = p2;
  *reinterpret_cast<hsk_bf16x4*>(dst + 2 * TK) = p3;
}

// (also used by hsk_eval_fused.hip)  planes: 6 * n_pad * Dp bytes, Dp = dim rounded up to 32
static void hsk_eval_split_planes_k(const float* src, const int64_t* idx, long long row0, long long n_src_rows, int n_valid,
                                    int n_pad, int D, void* planes, hipStream_t stream, int tile_k) {
  const int Dp = (int)hsk_align_up(D, GEMM_BK);
  const unsigned nblk = (unsigned)hsk_ceil_div((long long)n_pad * (Dp / 4), 256);
  if (tile_k == 16)
    k_split_planes<16><<<nblk, 256, 0, stream>>>(src, idx, row0, n_src_rows, n_valid, n_pad, D, Dp, (__bf16*)planes);
  else
    k_split_planes<GEMM_BK><<<nblk, 256, 0, stream>>>(src, idx, row0, n_src_rows, n_valid, n_pad, D, Dp, (__bf16*)planes);
}
void hsk_eval_split_planes(const float* src, const int64_t* idx, long long row0, long long n_src_rows, int n_valid,
                           int n_pad, int D, void* planes, hipStream_t stream) {
  hsk_eval_split_planes_k(src, idx, row0, n_src_rows, n_valid, n_pad, D, planes, stream, GEMM_BK);
}
void hsk_eval_split_planes16(const float* src, const int64_t* idx, long long row0, long long n_src_rows, int n_valid,
                             int n_pad, int D, void* planes, hipStream_t stream) {
  hsk_eval_split_planes_k(src, idx, row0, n_src_rows, n_valid, n_pad, D, planes, stream, 16);
}

// ---------------------------------------------------------------------------------------------
// Form 2: two fp16 pieces per operand (hsk_gemm_wide_h2.h).  The pre-pass first takes the largest finite |x| of the rows
// it is about to cut -- those and no others: an item shard's table may be nothing but the shard (include/hassaku_hip.h).
// A power-of-two scale changes no piece that stays a normal fp16, so shards of one catalogue (different maxima) agree
// except in elements below 2^-17 of a maximum -- then writes hi / lo at the scale that puts that maximum in [2^14, 2^15),
// laid out [Dp / 16 k-tiles][2 pieces][n_pad rows][16]: one (k-tile, piece) image of a 256-row block is 8 KB of
// consecutive bytes.  The GEMM undoes both scales with one exact multiplication.
// ---------------------------------------------------------------------------------------------
typedef _Float16 hsk_f16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_absmax_rows(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                     long long row0, long long n_src_rows, long long n_rows, int D,
                                                     uint32_t* __restrict__ out) {
  const int per_row = D / 4;
  const long long total = n_rows * per_row;
  float m = 0.f;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long r = t / per_row;
    const int k = (int)(t - r * per_row) * 4;
    long long sr = idx ? idx[r] : row0 + r;
    if (sr < 0 || sr >= n_src_rows) sr = idx ? 0 : row0;   // (a bad index is reported by the scoring kernel)
    const float4 x = *reinterpret_cast<const float4*>(src + sr * (long long)D + k);
    const float v[4] = {fabsf(x.x), fabsf(x.y), fabsf(x.z), fabsf(x.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (v[e] > m && v[e] < INFINITY) m = v[e];   // (NaN and inf do not take part)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {   // one atomic per workgroup (8192 of them on one word took 70 us of a 10 us kernel)
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > 0.f) atomicMax(out, __float_as_uint(m));   // non-negative floats order like their bits
  }
}

__global__ __launch_bounds__(256) void k_split_planes_h2(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                         long long row0, long long n_src_rows, int n_valid, int n_pad, int D,
                                                         int Dp, const uint32_t* __restrict__ amax,
                                                         _Float16* __restrict__ planes) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = Dp / 4;
  const long long r = t / per_row;
  const int k = (int)(t - r * per_row) * 4;
  if (r >= n_pad) return;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (r < n_valid && k < D) {
    long long sr = idx ? idx[r] : row0 + r;
    if (sr < 0 || sr >= n_src_rows) sr = 0;   // (a bad index is reported by the scoring kernel)
    const float4 x = *reinterpret_cast<const float4*>(src + sr * (long long)D + k);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
  const int e = hsk_h2_scale_exp(__uint_as_float(amax[0]));
  hsk_f16x4 hi, lo;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    _Float16 a, b;
    hsk_split_h2(v[q], e, a, b);
    hi[q] = a; lo[q] = b;
  }
  _Float16* dst = planes + ((long long)(k / 16) * 2 * n_pad + r) * 16 + (k % 16);
  *reinterpret_cast<hsk_f16x4*>(dst) = hi;
  *reinterpret_cast<hsk_f16x4*>(dst + (long long)n_pad * 16) = lo;
}

// (also used by hsk_eval_fused.hip)  planes: 4 * n_pad * Dp bytes; amax: one zeroed word the call keeps for the GEMM to read
void hsk_eval_split_planes_h2(const float* src, const int64_t* idx, long long row0, long long n_src_rows, int n_valid,
                              int n_pad, int D, uint32_t* amax, void* planes, hipStream_t stream) {
  const int Dp = (int)hsk_align_up(D, GEMM_BK);
  const long long work = (long long)n_valid * (D / 4);
  const unsigned nb = (unsigned)std::min<long long>(1024, std::max<long long>(1, hsk_ceil_div(work, 256 * 8)));
  k_absmax_rows<<<nb, 256, 0, stream>>>(src, idx, row0, n_src_rows, n_valid, D, amax);
  const unsigned nblk = (unsigned)hsk_ceil_div((long long)n_pad * (Dp / 4), 256);
  k_split_planes_h2<<<nblk, 256, 0, stream>>>(src, idx, row0, n_src_rows, n_valid, n_pad, D, Dp, amax, (_Float16*)planes);
}

typedef unsigned hsk_vu32x4 __attribute__((ext_vector_type(4)));   // (an array of HIP's uint4 structs ends up in scratch)
typedef float hsk_f32x4 __attribute__((ext_vector_type(4)));

// PLANES: Apl / Bpl hold the operands' pieces (k_split_planes; a_rows / b_rows = their padded row counts)
template <bool VEC4, bool PLANES = false>
__global__ __launch_bounds__(256, 2) void k_score_gemm_x3(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                    const float* __restrict__ Ib, const float* __restrict__ Ub,
                                                    const float* __restrict__ gb, int n_users, int D,
                                                    const int64_t* __restrict__ u_idx, int n_rows, long long item_begin,
                                                    int item_count, float* __restrict__ C, int32_t* status,
                                                    const __bf16* __restrict__ Apl = nullptr,
                                                    const __bf16* __restrict__ Bpl = nullptr, int a_rows = 0,
                                                    int b_rows = 0) {
  __shared__ __attribute__((aligned(16))) __bf16 As[3][GEMM_BM * GEMM_X3_LDK];   // the three bf16 pieces of the tile
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3][GEMM_BN * GEMM_X3_LDK];
  __shared__ int urow[GEMM_BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GEMM_BM;
  const int n0 = blockIdx.x * GEMM_BN;

  if (tid < GEMM_BM) {
    int r = m0 + tid;
    int u = 0;
    if (r < n_rows) {
      long long uu = u_idx[r];
      if (uu < 0 || uu >= n_users) {
        if (status) atomicOr(status, HSK_STATUS_BAD_INDEX);
        uu = 0;
      }
      u = (int)uu;
    }
    urow[tid] = u;
  }
  __syncthreads();

  hsk_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // staging assignment: 8 threads cover the 32 floats (128 B) of one row; 32 rows per pass, 4 passes
  const int srow = tid >> 3;
  const int scol = (tid & 7) * 4;
  const int half = lane >> 5;
  const int l32 = lane & 31;

  // Register-staged double buffering: the global loads of k-tile t+1 are issued before the MFMAs of tile t and
  // land in registers while the matrix cores work; they go to LDS after the tile's last read.  (Loading straight
  // into LDS inside the loop left the memory latency of every tile exposed: 78 TFLOP/s; staged: see DESIGN.md.)
  float4 ra[4], rb[4];
  hsk_vu32x4 rpa[6], rpb[6];   // PLANES: 6 x 16 bytes of the tile's 24 KB per operand
  auto load_tile = [&](int k0) {
    if (PLANES) {   // chunk c of the tile's 1536 is the 16 bytes at 16 c
      const __bf16* a = Apl + ((long long)(k0 / GEMM_BK) * a_rows + m0) * 96;
      const __bf16* b = Bpl + ((long long)(k0 / GEMM_BK) * b_rows + n0) * 96;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        rpa[i] = *reinterpret_cast<const hsk_vu32x4*>(a + (tid + 256 * i) * 8);
        rpb[i] = *reinterpret_cast<const hsk_vu32x4*>(b + (tid + 256 * i) * 8);
      }
      return;
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = srow + pass * 32;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + r < n_rows) {   // A: user rows (gathered through urow)
        const float* src = Uw + (long long)urow[r] * D + k0 + scol;
        if (VEC4) {
          if (k0 + scol < D) va = *reinterpret_cast<const float4*>(src);
        } else {
          if (k0 + scol + 0 < D) va.x = src[0];
          if (k0 + scol + 1 < D) va.y = src[1];
          if (k0 + scol + 2 < D) va.z = src[2];
          if (k0 + scol + 3 < D) va.w = src[3];
        }
      }
      if (n0 + r < item_count) {   // B: item rows of the shard
        const float* src = Iw + (item_begin + n0 + r) * (long long)D + k0 + scol;
        if (VEC4) {
          if (k0 + scol < D) vb = *reinterpret_cast<const float4*>(src);
        } else {
          if (k0 + scol + 0 < D) vb.x = src[0];
          if (k0 + scol + 1 < D) vb.y = src[1];
          if (k0 + scol + 2 < D) vb.z = src[2];
          if (k0 + scol + 3 < D) vb.w = src[3];
        }
      }
      ra[pass] = va;
      rb[pass] = vb;
    }
  };
  auto store_tile = [&]() {
    if (PLANES) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int c = tid + 256 * i, r = c / 12, j = c - r * 12, pl = j >> 2, kc = (j & 3) * 8;
        *reinterpret_cast<hsk_vu32x4*>(&As[pl][r * GEMM_X3_LDK + kc]) = rpa[i];
        *reinterpret_cast<hsk_vu32x4*>(&Bs[pl][r * GEMM_X3_LDK + kc]) = rpb[i];
      }
      return;
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = srow + pass * 32;
      hsk_bf16x4 a1, a2, a3, b1, b2, b3;
      const float av[4] = {ra[pass].x, ra[pass].y, ra[pass].z, ra[pass].w};
      const float bv[4] = {rb[pass].x, rb[pass].y, rb[pass].z, rb[pass].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 x1, x2, x3;
        hsk_split3(av[e], x1, x2, x3);
        a1[e] = x1; a2[e] = x2; a3[e] = x3;
        hsk_split3(bv[e], x1, x2, x3);
        b1[e] = x1; b2[e] = x2; b3[e] = x3;
      }
      *reinterpret_cast<hsk_bf16x4*>(&As[0][r * GEMM_X3_LDK + scol]) = a1;
      *reinterpret_cast<hsk_bf16x4*>(&As[1][r * GEMM_X3_LDK + scol]) = a2;
      *reinterpret_cast<hsk_bf16x4*>(&As[2][r * GEMM_X3_LDK + scol]) = a3;
      *reinterpret_cast<hsk_bf16x4*>(&Bs[0][r * GEMM_X3_LDK + scol]) = b1;
      *reinterpret_cast<hsk_bf16x4*>(&Bs[1][r * GEMM_X3_LDK + scol]) = b2;
      *reinterpret_cast<hsk_bf16x4*>(&Bs[2][r * GEMM_X3_LDK + scol]) = b3;
    }
  };
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int k0 = 0; k0 < D; k0 += GEMM_BK) {
    const bool has_next = k0 + GEMM_BK < D;
    if (has_next) load_tile(k0 + GEMM_BK);

#pragma unroll
    for (int s16 = 0; s16 < 2; ++s16) {   // two k = 16 steps per tile; lane (r = l32, h = half) holds k = 8h .. 8h+7 of a step
      hsk_bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[pl][i] = *reinterpret_cast<const hsk_bf16x8*>(&As[pl][(wm * 64 + i * 32 + l32) * GEMM_X3_LDK + 16 * s16 + 8 * half]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[pl][j] = *reinterpret_cast<const hsk_bf16x8*>(&Bs[pl][(wn * 64 + j * 32 + l32) * GEMM_X3_LDK + 16 * s16 + 8 * half]);
      }
      // the six products of weight >= 2^-16, smallest first: (3,1) (1,3) (2,2) (2,1) (1,2) (1,1)
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][i], b[TB[t]][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (has_next) {
      store_tile();
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const float gbv = gb ? gb[0] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l32;
      if (col >= item_count) continue;
      const float ib = Ib ? Ib[item_begin + col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rloc = wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        const int row = m0 + rloc;
        if (row < n_rows) {
          float o = acc[i][j][q];
          if (Ub) o += Ub[urow[rloc]];  // reference order: += u_bias, += i_bias, += global_bias
          if (Ib) o += ib;
          if (gb) o += gbv;
          C[(long long)row * item_count + col] = o;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// The same GEMM on a 256 x 256 block tile, ONE wave per SIMD (profiles/probes/gemm_bf16x3_v4.hip).
//
// The 128 x 128 tiling above moves 6 bytes per operand element through the L2s for 128 flop per byte -- at the bf16
// MFMA peak 19.5 TB/s, more than the L2s deliver -- and its two workgroups per CU share every SIMD's matrix pipe.
// Here a workgroup owns 256 x 256 (256 flop/B), its four waves 128 x 128 each (4 x 4 accumulator tiles = 256 AGPRs,
// 0.25 fragment reads per MFMA), alone on their SIMDs.  The pieces ([k16-tile][row][piece][16] bf16 from
// k_split_planes<16>, rows padded to 256 with zeros) go global -> registers -> LDS one k-step ahead, LDS double-buffered
// at BK = 16, one barrier per k-step.  What made the difference in the probe was the ORDER of a step's instructions: its
// 12 LDS stores + 12 global loads as a burst at the head of the step held the wave -- and its SIMD's matrix pipe -- for
// ~1500 of the step's ~5000 cycles (the CU's L1 takes 64 B per clock, the LDS store path 13 cycles per 16-byte store);
// the compiler left to itself clusters them (the scheduling-group hints did not move it).  The step is therefore written
// as ONE hand-interleaved stream pinned by scheduling fences: a (store, load) pair after every 8 MFMAs, the fragment
// reads of term n+1 under the MFMAs of term n.  Probe, 8192 x 10752 x 512 on one box: 585 us (128 x 128, two workgroups
// per CU) -> 451 us, 39 cycles per MFMA against the pipe's 32 -- at an in-kernel clock the chip lowers to 1.7 GHz under
// this load (2.0-2.1 GHz under the looser loop): what bounds it now is power, not issue.
// Every output element sees the same sequence of MFMAs as in k_score_gemm_x3: the scores are bit-identical.
// ---------------------------------------------------------------------------------------------

// Epilogue of the 256 x 256 score GEMMs, through LDS.  One workgroup per CU: nothing covers this phase, so its length
// counts in full -- and as dword-per-lane stores (256 per wave, every 128-byte piece split over two cache lines when the
// row stride is not a multiple of 32 floats: item_count = 10 677) it took about a quarter of the kernel at the ml10m
// shape.  Per band of 32 rows the four waves put their accumulators into LDS -- scaled by cs (form 2; exact), biases added
// in the reference's order (+= u_bias, += i_bias, += global_bias), each row shifted by its own misalignment
// (row * item_count + n0) mod 4 so that 16-byte pieces of LDS are 16-byte pieces of global memory -- and every wave then
// writes 16 whole rows of the band's 256 columns with one ds_read_b128 + one global_store_dwordx4 per lane and row; the
// ragged ends go out as dwords.  st: 2 x 32 rows of 264 floats = 67.6 KB of the (dead) operand stages.
template <bool SCALE>
__device__ __forceinline__ void hsk_wide_store_scores(const hsk_w_f32x16 (&acc)[4][4], float* st, float cs,
                                                      const float* __restrict__ Ib, const float* __restrict__ Ub,
                                                      const float* __restrict__ gb, int n_users,
                                                      const int64_t* __restrict__ u_idx, int n_rows, long long item_begin,
                                                      int item_count, float* __restrict__ C, int32_t* status, int m0, int n0,
                                                      int lane, int wm, int wn, int r32, int h) {
  constexpr int TM = 4, TN = 4, WM = 128, WN = 128;
  constexpr int LDR = 256 + 8;                       // floats per staged row (shift <= 3, 16-byte chunks conflict-free)
  const float gbv = gb ? gb[0] : 0.f;
  const int ncols = min(256, item_count - n0);       // valid columns of this block
  const int ic3 = item_count & 3;
  float ibv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + j * 32 + r32;
    ibv[j] = (Ib && col < item_count) ? Ib[item_begin + col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();   // the previous band's rows have been read (first band: the last k-step's fragments)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int rloc = (q & 3) + 8 * (q >> 2) + 4 * h;
      const int row = m0 + wm * WM + i * 32 + rloc;
      float ub = 0.f;
      if (row < n_rows) {
        long long uu = u_idx[row];
        if (uu < 0 || uu >= n_users) {
          if (status) atomicOr(status, HSK_STATUS_BAD_INDEX);
          uu = 0;
        }
        if (Ub) ub = Ub[uu];
      }
      const int shift = ((row & 3) * ic3) & 3;       // (row * item_count + n0) mod 4, n0 a multiple of 256
      float* dst = st + (wm * 32 + rloc) * LDR + wn * WN + r32 + shift;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float o = SCALE ? acc[i][j][q] * cs : acc[i][j][q];
        if (Ub) o += ub;
        if (Ib) o += ibv[j];
        if (gb) o += gbv;
        dst[j * 32] = o;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      const int rloc = wn * 16 + rr;
      const int row = m0 + wm * WM + i * 32 + rloc;
      if (row >= n_rows) break;                      // wave-uniform
      const int shift = ((row & 3) * ic3) & 3;
      const float* src = st + (wm * 32 + rloc) * LDR;
      float* gdst = C + (long long)row * item_count + n0 - shift;   // 16-byte aligned
      const int f0 = 4 * lane;                       // this lane's chunk: staged floats [f0, f0 + 4) -> gdst[f0 ..]
      const hsk_f32x4 v = *reinterpret_cast<const hsk_f32x4*>(src + f0);
      const int lo = shift, hi = shift + ncols;      // valid staged floats
      if (f0 >= lo && f0 + 4 <= hi) {
        *reinterpret_cast<hsk_f32x4*>(gdst + f0) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (f0 + e >= lo && f0 + e < hi) gdst[f0 + e] = v[e];
      }
      if (lane < shift && 256 + lane < hi) gdst[256 + lane] = src[256 + lane];   // the chunk past lane 63
    }
  }
}

__global__ __launch_bounds__(256, 1) void k_score_gemm_x3_wide(const float* __restrict__ Ib, const float* __restrict__ Ub,
                                                               const float* __restrict__ gb, int n_users, int Dp,
                                                               const int64_t* __restrict__ u_idx, int n_rows,
                                                               long long item_begin, int item_count,
                                                               float* __restrict__ C, int32_t* status,
                                                               const __bf16* __restrict__ Apl,
                                                               const __bf16* __restrict__ Bpl, int a_rows, int b_rows) {
  extern __shared__ __attribute__((aligned(16))) __bf16 wlds[];
  constexpr int TM = 4, TN = 4;
  __bf16* As = wlds;
  __bf16* Bs = wlds + 2 * GEMM_W_A_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GEMM_W_BM, n0 = blockIdx.x * GEMM_W_BN;
  const int r32 = lane & 31, h = lane >> 5;
  hsk_w_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  hsk_wide_stage stg;
  hsk_wide_init(stg, tid);
  const int NT = Dp / GEMM_W_BK;
  const __bf16* a0 = Apl + (long long)m0 * 48;
  const __bf16* b0 = Bpl + (long long)n0 * 48;
  const long long a_step = (long long)a_rows * 48, b_step = (long long)b_rows * 48;
  hsk_wide_load(stg, a0, b0, tid);                     // k-tile 0 -> LDS stage 0
  hsk_wide_store(stg, As, Bs);
  hsk_wide_load(stg, a0 + (NT > 1 ? a_step : 0), b0 + (NT > 1 ? b_step : 0), tid);   // k-tile 1 -> registers
  __syncthreads();
  hsk_wide_kloop(acc, stg, As, Bs, a0, b0, a_step, b_step, NT, tid, wm, wn, r32, h);
  hsk_wide_store_scores<false>(acc, reinterpret_cast<float*>(wlds), 1.f, Ib, Ub, gb, n_users, u_idx, n_rows, item_begin, item_count,
                               C, status, m0, n0, lane, wm, wn, r32, h);
}

// The same on two fp16 pieces per operand, three products (hsk_gemm_wide_h2.h; planes from hsk_eval_split_planes_h2).
// amax[0] / amax[1]: the largest |x| the user / item pieces were scaled by.
__global__ __launch_bounds__(256, 1) void k_score_gemm_h2_wide(const float* __restrict__ Ib, const float* __restrict__ Ub,
                                                               const float* __restrict__ gb, int n_users, int Dp,
                                                               const int64_t* __restrict__ u_idx, int n_rows,
                                                               long long item_begin, int item_count,
                                                               float* __restrict__ C, int32_t* status,
                                                               const _Float16* __restrict__ Apl,
                                                               const _Float16* __restrict__ Bpl, int a_rows, int b_rows,
                                                               const uint32_t* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hlds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GEMM_W_BM, n0 = blockIdx.x * GEMM_W_BN;
  const int r32 = lane & 31, h = lane >> 5;
  hsk_w_f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const float cs = ldexpf(1.f, -(hsk_h2_scale_exp(__uint_as_float(amax[0])) + hsk_h2_scale_exp(__uint_as_float(amax[1]))));
  hsk_h2_stage stg;
  const int NT = Dp / GEMM_H_BK;
  hsk_h2_load(stg, Apl, Bpl, a_rows, b_rows, m0, n0, 0, tid);                  // k-step 0 -> LDS stage 0
  hsk_h2_store(stg, hlds, tid);
  hsk_h2_load(stg, Apl, Bpl, a_rows, b_rows, m0, n0, NT > 1 ? 1 : 0, tid);     // k-step 1 -> registers
  __syncthreads();
  hsk_h2_kloop(acc, stg, hlds, Apl, Bpl, a_rows, b_rows, m0, n0, NT, tid, wm, wn, r32, h);
  hsk_wide_store_scores<true>(acc, reinterpret_cast<float*>(hlds), cs, Ib, Ub, gb, n_users, u_idx, n_rows, item_begin, item_count,
                              C, status, m0, n0, lane, wm, wn, r32, h);
}

// excluded (user, item) pairs -> -inf.  One wave per eval row, lanes stride over the user's CSR row.
__global__ __launch_bounds__(256) void k_mask_excluded(const int64_t* __restrict__ u_idx, int n_rows, int n_users,
                                                       const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices, long long item_begin,
                                                       int item_count, float* __restrict__ C) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  long long u = u_idx[r];
  if (u < 0 || u >= n_users) u = 0;
  const long long lo = indptr[u], hi = indptr[u + 1];
  for (long long e = lo + lane; e < hi; e += 64) {
    const long long it = (long long)indices[e] - item_begin;
    if (it >= 0 && it < item_count) C[(long long)r * item_count + it] = -INFINITY;
  }
}

// =============================================================================================
// Top-k per row: 4-pass 8-bit radix select on order-preserving keys, collect, bitonic sort.
// Order: value descending, then index ascending.  One 256-thread block per row.
// =============================================================================================
__device__ __forceinline__ uint32_t hsk_f2key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending float order == ascending key order
}
__device__ __forceinline__ float hsk_key2f(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

#define TOPK_MAX 1024
#define TOPK_CAND_MAX 2048   // LDS candidates of the two-pass fast path (>= TOPK_MAX)

// sorts n (power of two, <= TOPK_MAX... up to 4096) composite keys in LDS descending
__device__ void hsk_bitonic_desc(unsigned long long* s, int n) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = s[lo], b = s[hi];
        if ((a < b) == desc) {
          s[lo] = b;
          s[hi] = a;
        }
      }
    }
  }
  __syncthreads();
}

// Of 256 bin counts h[0..255] (LDS): the highest bin b with (keys in bins above b) < need <= (those + h[b]), found by
// wave 0 -- lane l owns bins 4l .. 4l+3, a suffix sum over the lanes -- instead of one thread walking 256 dependent LDS
// reads (7 us per walk, and a selection makes several).  No such bin (fewer than `need` keys in bins 1..255): bin 0.
// Every lane of wave 0 returns the same (bin, keys above it, keys in it); other waves must not call.
__device__ __forceinline__ int hsk_topk_wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

struct hsk_bin_pick { unsigned int bin, above, count; };
__device__ __forceinline__ hsk_bin_pick hsk_pick_bin_256(const unsigned int* h, unsigned int need, int lane) {
  const unsigned int c0 = h[lane * 4], c1 = h[lane * 4 + 1], c2 = h[lane * 4 + 2], c3 = h[lane * 4 + 3];
  const unsigned int mine = c0 + c1 + c2 + c3;
  unsigned int suf = mine;   // inclusive suffix sum over lanes >= this one
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int t = __shfl_down(suf, off, 64);
    if (lane + off < 64) suf += t;
  }
  const unsigned int a3 = suf - mine, a2 = a3 + c3, a1 = a2 + c2, a0 = a1 + c1;
  int fb = -1;
  unsigned int fa = 0, fc = 0;
  if (a3 < need && need <= a3 + c3) { fb = lane * 4 + 3; fa = a3; fc = c3; }
  else if (a2 < need && need <= a2 + c2) { fb = lane * 4 + 2; fa = a2; fc = c2; }
  else if (a1 < need && need <= a1 + c1) { fb = lane * 4 + 1; fa = a1; fc = c1; }
  else if (a0 < need && need <= a0 + c0 && lane * 4 > 0) { fb = lane * 4; fa = a0; fc = c0; }
  const unsigned long long who = __ballot(fb > 0);
  hsk_bin_pick r;
  if (who) {
    const int src = 63 - __builtin_clzll(who);   // the highest such bin
    r.bin = (unsigned)__shfl(fb, src, 64);
    r.above = __shfl(fa, src, 64);
    r.count = __shfl(fc, src, 64);
  } else {                                       // bin 0 takes what is left
    const unsigned int total = __shfl(suf, 0, 64), h0 = __shfl(c0, 0, 64);
    r.bin = 0;
    r.above = total - h0;
    r.count = h0;
  }
  return r;
}

// TOPK_RPT keys per thread stay in registers for rows of up to 256 x TOPK_RPT columns: 48 (rows of <= 12 288 columns; the
// kernel then fits 4 waves per SIMD) or 64 (<= 16 384 columns -- the width of one rank's shard of the 131 072-item
// catalogue on 8 GPUs -- at 3 waves per SIMD)
template <typename IdxOut, int TOPK_RPT>
__device__ __forceinline__ void hsk_topk_rows_body(const float* __restrict__ X, long long ld, int cols, int k, int kpad,
                                                   long long idx_offset, float* __restrict__ out_vals,
                                                   IdxOut* __restrict__ out_idx) {
  __shared__ unsigned int hist[256];
  // the 12-bit histogram is dead (summed into part12, its b1 bin's count copied out) before the first candidate is
  // written: one 16 KB array serves both, and twice as many rows are in flight per CU
  __shared__ unsigned long long cand[TOPK_CAND_MAX];
  static_assert(sizeof(unsigned long long) * TOPK_CAND_MAX >= sizeof(unsigned int) * 4096, "hist12 lives in cand");
  unsigned int* const hist12 = reinterpret_cast<unsigned int*>(cand);
  __shared__ unsigned int sh_prefix, sh_need, sh_ngt, sh_neq, sh_taken;
  __shared__ unsigned int wave_cnt[4], wave_tot[4];
  __shared__ unsigned int part12[256];

  const int tid = threadIdx.x;
  const float* __restrict__ row = X + (long long)blockIdx.x * ld;

  // Fast path, two reads of the row instead of five: a 12-bit histogram (sign, exponent, 3 mantissa bits) finds the
  // bin b1 that holds the k-th largest key; everything in a higher bin is certainly in the top k, everything in b1
  // may be.  If those candidates fit into LDS they are gathered in one more pass and sorted there (key descending,
  // index ascending -- the same tie rule as below).  Rows with more candidates than that (long runs of equal scores,
  // e.g. -inf) take the general 4 x 8-bit radix select below.
  // Rows of up to 256 x TOPK_RPT columns are read ONCE: every thread keeps its TOPK_RPT keys in registers (all loads in
  // flight together) and the k-th largest key's prefix is narrowed 8 bits at a time by histograms over the registers:
  // first the top 8 bits -- where scores of one magnitude all meet in a few bins, so every bin has 16 replicas (by
  // lane) or the LDS atomics serialise: that contention, not the row's bytes, was what this kernel spent its time on --
  // then 8 more bits among the keys of that bin, until only a few keys beyond k are left to sort.
  // The row is read in 16-byte pieces from the 16-byte boundary below its first element (`shift` floats in front of it):
  // key i of a thread is column 4 * (tid + 256 * (i / 4)) + i % 4 - shift.  (One dword per lane per load moved the
  // 350 MB of an 8192 x 10 677 chunk at 2.5 TB/s and was two thirds of this kernel's time.)
  const int shift = (int)((reinterpret_cast<uintptr_t>(row) >> 2) & 3u);
  const bool in_regs = cols + shift <= 256 * TOPK_RPT;
  auto col_of = [&](int i) { return 4 * (tid + 256 * (i >> 2)) + (i & 3) - shift; };
  uint32_t kreg[TOPK_RPT];
  // k <= 256 on a row held in registers: NO histogram at all.  Every thread takes the maximum of its keys; every wave
  // the ceil(k/4)-th largest of its 64 maxima (ranked by counting over readlane broadcasts, no LDS); L = the smallest
  // of the four.  At least 4 * ceil(k/4) >= k distinct elements are >= L, so L is a lower bound of the row's k-th
  // largest key, and on scores without structure along the columns only k .. ~1.5 k keys are >= L: they are ranked
  // below by counting.  (The 12-bit histogram this replaces cost 48 LDS atomics per thread plus its 8-bit refinements.)
  // Keys of columns outside the row are 0 here (below every key a float maps to, NaNs with a set sign bit aside).
#ifndef HSK_TOPK_MAXIMA
#define HSK_TOPK_MAXIMA 1     // 0: the histogram selection for every k (comparison builds)
#endif
  const bool by_maxima = HSK_TOPK_MAXIMA && in_regs && k <= 256;
  unsigned int my_off = 0, mx_b1 = 0, mx_cand = 0;   // by_maxima: first candidate slot of this thread, L, candidates
  if (cols >= 4096) {
    if (!by_maxima)
      for (int c = tid; c < 4096; c += 256) hist12[c] = 0;
    if (in_regs) {
#pragma unroll
      for (int v = 0; v < TOPK_RPT / 4; ++v) {
        // No branch around a load: TOPK_RPT / 4 independent 16-byte loads, all in flight together (with a scalar path
        // for the pieces that straddle the row's ends every piece waited for the one before it: 12-16 memory round
        // trips per row, ~20 us, where the row's bytes need 2).  A piece that straddles an end is still ONE aligned
        // 16-byte load: it holds at least one element of the row, so it lies inside that element's page; what it
        // brings along from outside the row is never looked at (every use below tests the column).  Pieces wholly
        // outside re-read the row's first piece.
        const int c0 = col_of(4 * v);
        const bool some = c0 + 3 >= 0 && c0 < cols;
        const float4 x = *reinterpret_cast<const float4*>(row + (some ? c0 : -shift));   // 16-byte aligned by construction
        kreg[4 * v + 0] = (unsigned)(c0 + 0) < (unsigned)cols ? hsk_f2key(x.x) : 0u;
        kreg[4 * v + 1] = (unsigned)(c0 + 1) < (unsigned)cols ? hsk_f2key(x.y) : 0u;
        kreg[4 * v + 2] = (unsigned)(c0 + 2) < (unsigned)cols ? hsk_f2key(x.z) : 0u;
        kreg[4 * v + 3] = (unsigned)(c0 + 3) < (unsigned)cols ? hsk_f2key(x.w) : 0u;
      }
    }
    if (by_maxima) {
      uint32_t mx = 0;
#pragma unroll
      for (int i = 0; i < TOPK_RPT; ++i) mx = max(mx, kreg[i]);
#if defined(HSK_TOPK_STOP) && HSK_TOPK_STOP == 1
      if (mx == 12345u) out_vals[0] = 1.f;
      return;
#endif
      const int lane = tid & 63;
      const int cth = (k + 3) >> 2;        // <= 64
      // the cth-th largest of the wave's 64 maxima, bit by bit from the top: the largest T with >= cth maxima >= T (one
      // compare + a scalar population count per bit; ranking every lane against every other took 64 x 5 vector operations)
      uint32_t Lw = 0;
#pragma unroll
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t t = Lw | (1u << bit);
        if (__popcll(__ballot(mx >= t)) >= cth) Lw = t;
      }
      if (lane == 0) wave_cnt[tid >> 6] = Lw;
      __syncthreads();
      const uint32_t L = min(min(wave_cnt[0], wave_cnt[1]), min(wave_cnt[2], wave_cnt[3]));
      int mine_ge = 0;
#pragma unroll
      for (int i = 0; i < TOPK_RPT; ++i) mine_ge += (kreg[i] >= L) ? 1 : 0;
      // where this thread's candidates go: an exclusive scan of the counts over the workgroup (no atomics: ~k returning
      // LDS atomics on one word were a serial chain of their own)
      const int incl = hsk_topk_wave_incl_scan(mine_ge, lane);
      if (lane == 63) wave_tot[tid >> 6] = (unsigned)incl;
#if defined(HSK_TOPK_STOP) && HSK_TOPK_STOP == 2
      if (incl == 12345) out_vals[0] = 1.f;
      return;
#endif
      __syncthreads();
      unsigned int before = 0, total = 0;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) {
        const unsigned int t = wave_tot[ww];
        before += (ww < (tid >> 6)) ? t : 0u;
        total += t;
      }
      my_off = before + (unsigned)(incl - mine_ge);
      mx_b1 = L;
      // L == 0 (a wave with fewer than cth columns): everything is a candidate -> the general selection below
      mx_cand = L ? total : (unsigned)TOPK_CAND_MAX + 1u;
    } else {
    __syncthreads();
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < TOPK_RPT; ++i)
        if ((unsigned)col_of(i) < (unsigned)cols) atomicAdd(&hist12[(kreg[i] >> 24) * 16 + (tid & 15)], 1u);
    } else {
      for (int c = tid; c < cols; c += 256) atomicAdd(&hist12[hsk_f2key(row[c]) >> 20], 1u);
    }
    __syncthreads();
    {
      unsigned int sum = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += hist12[tid * 16 + j];
      part12[tid] = sum;
    }
    __syncthreads();
    if (tid < 64) {
      const hsk_bin_pick pk = hsk_pick_bin_256(part12, (unsigned)k, tid);   // group of 16 bins (or 8-bit bin)
      if (tid == 0) {
        unsigned int cum = pk.above;   // keys in bins above the current one
        const int g = (int)pk.bin;
        if (in_regs) {                       // 8-bit bins (16 replicas each)
          sh_prefix = (unsigned)g;
          sh_ngt = cum;
          sh_neq = pk.count;
        } else {
          int bin = g * 16 + 15;
          for (; bin > g * 16; --bin) {
            if (cum + hist12[bin] >= (unsigned)k) break;
            cum += hist12[bin];
          }
          sh_prefix = (unsigned)bin;         // b1
          sh_ngt = cum;                      // keys in higher bins (< k)
          sh_neq = hist12[bin];              // keys in b1
        }
        sh_taken = 0;
      }
    }
    __syncthreads();
    }
    unsigned int b1 = by_maxima ? mx_b1 : sh_prefix, n_cand = by_maxima ? mx_cand : sh_ngt + sh_neq;
    int cshift = by_maxima ? 0 : in_regs ? 24 : 20;   // candidates: keys with (key >> cshift) >= b1
    while (in_regs && cshift >= 8 && n_cand > 2u * (unsigned)kpad) {
      // 8 more bits among the keys that share the current prefix
      __syncthreads();
      hist[tid] = 0;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < TOPK_RPT; ++i)
        if ((unsigned)col_of(i) < (unsigned)cols && (kreg[i] >> cshift) == b1) atomicAdd(&hist[(kreg[i] >> (cshift - 8)) & 0xffu], 1u);
      __syncthreads();
      if (tid < 64) {
        const hsk_bin_pick pk = hsk_pick_bin_256(hist, (unsigned)k - sh_ngt, tid);   // need >= 1: the k-th largest carries the prefix
        if (tid == 0) {
          sh_prefix = (b1 << 8) | pk.bin;
          sh_ngt += pk.above;
          sh_neq = pk.count;
        }
      }
      __syncthreads();
      b1 = sh_prefix;
      n_cand = sh_ngt + sh_neq;
      cshift -= 8;
    }
    if (n_cand <= TOPK_CAND_MAX) {
      int npad = 1;
      while (npad < (int)n_cand) npad <<= 1;
      if (npad < kpad) npad = kpad;
      if (!by_maxima || n_cand > 256) {   // (ranking by counting reads cand[0 .. n_cand) only, and by_maxima has not
        for (int c = tid; c < npad; c += 256) cand[c] = 0ull;   // touched LDS: no pads, no barrier)   pads sort last
        __syncthreads();
      }
      if (by_maxima) {
        unsigned int slot = my_off;
#pragma unroll
        for (int i = 0; i < TOPK_RPT; ++i)
          if (kreg[i] >= b1) cand[slot++] = ((unsigned long long)kreg[i] << 32) | (uint32_t)(~(uint32_t)col_of(i));
      } else if (in_regs) {
#pragma unroll
        for (int i = 0; i < TOPK_RPT; ++i) {
          const int c = col_of(i);
          if ((unsigned)c < (unsigned)cols && (kreg[i] >> cshift) >= b1) {
            const unsigned int slot = atomicAdd(&sh_taken, 1u);
            cand[slot] = ((unsigned long long)kreg[i] << 32) | (uint32_t)(~(uint32_t)c);
          }
        }
      } else {
        for (int c = tid; c < cols; c += 256) {
          const uint32_t key = hsk_f2key(row[c]);
          if ((key >> 20) >= b1) {
            const unsigned int slot = atomicAdd(&sh_taken, 1u);
            cand[slot] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)c);
          }
        }
      }
      if (n_cand <= 256 && tid < 8) cand[n_cand + tid] = 0ull;   // (nobody's slot: see the ranking loop below)
      __syncthreads();
#if defined(HSK_TOPK_STOP) && HSK_TOPK_STOP == 3
      if (cand[0] == 12345ull) out_vals[0] = 1.f;
      return;
#endif
      if (n_cand <= 256) {
        // a few keys beyond k: every thread ranks its own candidate by counting the larger ones (composite keys are
        // unique; broadcast LDS reads) -- one pass instead of the 36 barrier-separated stages of a bitonic sort
        // (eight LDS reads in flight per step: one read per iteration was a chain of n_cand LDS latencies; the slots up
        // to the next multiple of 8 were zeroed in front of the barrier above and rank nobody down)
        const unsigned long long mine = (tid < (int)n_cand) ? cand[tid] : 0ull;
        int rank = 0;
        for (int j = 0; j < (int)n_cand; j += 8) {
          unsigned long long o[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = cand[j + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) rank += (o[q] > mine) ? 1 : 0;
        }
        if (tid < (int)n_cand && rank < k) {
          out_vals[(long long)blockIdx.x * k + rank] = hsk_key2f((uint32_t)(mine >> 32));
          out_idx[(long long)blockIdx.x * k + rank] = (IdxOut)((long long)(~(uint32_t)mine) + idx_offset);
        }
        return;
      }
      hsk_bitonic_desc(cand, npad);
      for (int c = tid; c < k; c += 256) {
        const unsigned long long v = cand[c];
        out_vals[(long long)blockIdx.x * k + c] = hsk_key2f((uint32_t)(v >> 32));
        out_idx[(long long)blockIdx.x * k + c] = (IdxOut)((long long)(~(uint32_t)v) + idx_offset);
      }
      return;
    }
    __syncthreads();
  }

  uint32_t prefix = 0, prefix_mask = 0;
  unsigned int need = (unsigned)k;  // how many still to take among keys matching the prefix
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    for (int c = tid; c < cols; c += 256) {
      const uint32_t key = hsk_f2key(row[c]);
      if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      const hsk_bin_pick pk = hsk_pick_bin_256(hist, need, tid);
      if (tid == 0) {
        sh_prefix = prefix | ((uint32_t)pk.bin << shift);
        sh_need = need - pk.above;
        sh_neq = pk.count;
      }
    }
    __syncthreads();
    prefix = sh_prefix;
    need = sh_need;
    prefix_mask |= (0xffu << shift);
    __syncthreads();
  }
  // prefix == key of the k-th largest element (T); `need` of the sh_neq elements equal to T are taken
  const uint32_t T = prefix;
  const unsigned int n_eq = sh_neq;
  if (tid == 0) {
    sh_ngt = 0;
    sh_taken = 0;
  }
  for (int c = tid; c < kpad; c += 256) cand[c] = 0ull;  // pads sort last
  __syncthreads();
  const unsigned int n_gt = (unsigned)k - need;

  if (n_eq == need) {
    // no tie straddles the cut: take every key >= T, any order
    for (int c = tid; c < cols; c += 256) {
      const uint32_t key = hsk_f2key(row[c]);
      if (key >= T) {
        const unsigned int slot = atomicAdd(&sh_ngt, 1u);
        cand[slot] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)c);
      }
    }
  } else {
    // ties at the cut: keys > T in any order, keys == T lowest index first
    for (int c = tid; c < cols; c += 256) {
      const uint32_t key = hsk_f2key(row[c]);
      if (key > T) {
        const unsigned int slot = atomicAdd(&sh_ngt, 1u);
        cand[slot] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)c);
      }
    }
    __syncthreads();
    for (int base = 0; base < cols; base += 256) {
      if (sh_taken >= need) break;  // uniform: sh_taken only changes between barriers
      const int c = base + tid;
      const bool eq = (c < cols) && (hsk_f2key(row[c]) == T);
      const unsigned long long m = __ballot(eq);
      const int lane = tid & 63, w = tid >> 6;
      if (lane == 0) wave_cnt[w] = (unsigned)__popcll(m);
      __syncthreads();
      unsigned int before = sh_taken;
      for (int ww = 0; ww < w; ++ww) before += wave_cnt[ww];
      const unsigned int rank = before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
      if (eq && rank < need) cand[n_gt + rank] = ((unsigned long long)T << 32) | (uint32_t)(~(uint32_t)c);
      __syncthreads();
      if (tid == 0) sh_taken += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
    }
  }
  __syncthreads();
  hsk_bitonic_desc(cand, kpad);
  for (int c = tid; c < k; c += 256) {
    const unsigned long long v = cand[c];
    out_vals[(long long)blockIdx.x * k + c] = hsk_key2f((uint32_t)(v >> 32));
    out_idx[(long long)blockIdx.x * k + c] = (IdxOut)((long long)(~(uint32_t)v) + idx_offset);
  }
}

template <typename IdxOut>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_topk_rows(const float* __restrict__ X, long long ld, int cols, int k, int kpad, long long idx_offset,
                 float* __restrict__ out_vals, IdxOut* __restrict__ out_idx) {
  hsk_topk_rows_body<IdxOut, 48>(X, ld, cols, k, kpad, idx_offset, out_vals, out_idx);
}
template <typename IdxOut>
__global__ __launch_bounds__(256) void k_topk_rows_wide(const float* __restrict__ X, long long ld, int cols, int k, int kpad,
                                                        long long idx_offset, float* __restrict__ out_vals,
                                                        IdxOut* __restrict__ out_idx) {
  hsk_topk_rows_body<IdxOut, 64>(X, ld, cols, k, kpad, idx_offset, out_vals, out_idx);
}

static int hsk_next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// merge of candidate lists: [n_parts, rows, k] -> [rows, k]
__global__ __launch_bounds__(256) void k_topk_merge(const float* __restrict__ vals, const int32_t* __restrict__ idx,
                                                    int n_parts, int rows, int k, int npad,
                                                    float* __restrict__ out_vals, int32_t* __restrict__ out_idx) {
  __shared__ unsigned long long cand[4096];
  const int r = blockIdx.x;
  const int total = n_parts * k;
  for (int c = threadIdx.x; c < npad; c += 256) {
    unsigned long long v = 0ull;
    if (c < total) {
      const int p = c / k, j = c - p * k;
      const long long src = ((long long)p * rows + r) * k + j;
      v = ((unsigned long long)hsk_f2key(vals[src]) << 32) | (uint32_t)(~(uint32_t)idx[src]);
    }
    cand[c] = v;
  }
  hsk_bitonic_desc(cand, npad);
  for (int c = threadIdx.x; c < k; c += 256) {
    const unsigned long long v = cand[c];
    out_vals[(long long)r * k + c] = hsk_key2f((uint32_t)(v >> 32));
    out_idx[(long long)r * k + c] = (int32_t)(~(uint32_t)v);
  }
}

// =============================================================================================
// rank metrics: one wave per evaluated user
// =============================================================================================
#define HSK_MAX_KS 8
struct hsk_ks {
  int k[HSK_MAX_KS];
  int n;
};

__global__ __launch_bounds__(256) void k_rank_metrics(const int32_t* __restrict__ topk, int n_rows, int k_max,
                                                      const int64_t* __restrict__ u_idx, int n_users,
                                                      const int64_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ indices, hsk_ks ks,
                                                      float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  long long u = u_idx[r];
  if (u < 0 || u >= n_users) u = 0;
  const long long lo = indptr[u], hi = indptr[u + 1];
  const int n_rel = (int)(hi - lo);
  float hits[HSK_MAX_KS], dcg[HSK_MAX_KS], idcg[HSK_MAX_KS];
#pragma unroll
  for (int t = 0; t < HSK_MAX_KS; ++t) hits[t] = dcg[t] = idcg[t] = 0.f;
  for (int rank = lane; rank < k_max; rank += 64) {
    const int item = topk[(long long)r * k_max + rank];
    long long l = lo, h = hi;
    while (l < h) {
      const long long mid = (l + h) >> 1;
      if (indices[mid] < item)
        l = mid + 1;
      else
        h = mid;
    }
    const bool hit = (l < hi) && (indices[l] == item);
    const float disc = 1.f / log2f((float)(rank + 2));
#pragma unroll
    for (int t = 0; t < HSK_MAX_KS; ++t) {
      if (t < ks.n && rank < ks.k[t]) {
        if (hit) {
          hits[t] += 1.f;
          dcg[t] += disc;
        }
        if (rank < n_rel) idcg[t] += disc;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < HSK_MAX_KS; ++t) {
    if (t < ks.n) {
      const float h_ = hsk_wave_sum(hits[t]);
      const float d_ = hsk_wave_sum(dcg[t]);
      const float i_ = hsk_wave_sum(idcg[t]);
      if (lane == 0) {
        float* o = out + ((long long)r * ks.n + t) * 3;
        o[0] = h_ / (float)ks.k[t];
        o[1] = n_rel > 0 ? h_ / (float)n_rel : 0.f;
        o[2] = n_rel > 0 ? fminf(d_ / i_, 1.f) : 0.f;
      }
    }
  }
}

// =============================================================================================
// C ABI
// =============================================================================================
static int hsk_launch_topk_i32(const float* X, int64_t rows, int64_t cols, int64_t ld, int64_t k, long long off,
                               float* out_vals, int32_t* out_idx, hipStream_t stream) {
  const int kpad = hsk_next_pow2((int)k);
  if (cols + 3 > 256 * 48 && cols <= 256 * 64)
    k_topk_rows_wide<int32_t><<<(unsigned)rows, 256, 0, stream>>>(X, ld, (int)cols, (int)k, kpad, off, out_vals, out_idx);
  else
    k_topk_rows<int32_t><<<(unsigned)rows, 256, 0, stream>>>(X, ld, (int)cols, (int)k, kpad, off, out_vals, out_idx);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

// the opt-in for > 64 KB of dynamic LDS is a per-device attribute of each kernel
static int hsk_eval_set_wide_lds() {
  static bool lds_set[64] = {};
  int dev = 0;
  HSK_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !lds_set[dev]) {
    HSK_HIP(hipFuncSetAttribute((const void*)k_score_gemm_x3_wide, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_W_LDS_BYTES));
    HSK_HIP(hipFuncSetAttribute((const void*)k_score_gemm_h2_wide, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_H_LDS_BYTES));
    lds_set[dev] = true;
  }
  return HSK_OK;
}

extern "C" int64_t hsk_mf_eval_planes_bytes(int64_t n_rows, int64_t item_count, int64_t dim) {
  if (n_rows <= 0 || item_count <= 0 || dim <= 0) return 0;
  const int64_t Dp = hsk_align_up(dim, GEMM_BK);
  // (rows padded to 256: what the widest block tile, k_score_gemm_x3_wide, reads)
  return hsk_align_up(6 * hsk_align_up(n_rows, GEMM_W_BM) * Dp, 256) + hsk_align_up(6 * hsk_align_up(item_count, GEMM_W_BN) * Dp, 256);
}

extern "C" int hsk_mf_eval_topk(const float* user_emb, const float* item_emb, const float* item_bias,
                                const float* user_bias, const float* global_bias, int64_t n_users, int64_t n_items,
                                int64_t dim, const int64_t* u_idx, int64_t n_rows, int64_t item_begin,
                                int64_t item_count, const int64_t* excl_indptr, const int32_t* excl_indices, int64_t k,
                                float* scores_ws, float* out_vals, int32_t* out_idx, int32_t* status,
                                hsk_stream_t stream_) {
  return hsk_mf_eval_topk_planes(user_emb, item_emb, item_bias, user_bias, global_bias, n_users, n_items, dim, u_idx,
                                 n_rows, item_begin, item_count, excl_indptr, excl_indices, k, scores_ws, nullptr, 0,
                                 out_vals, out_idx, status, stream_);
}

extern "C" int hsk_mf_eval_topk_planes(const float* user_emb, const float* item_emb, const float* item_bias,
                                       const float* user_bias, const float* global_bias, int64_t n_users,
                                       int64_t n_items, int64_t dim, const int64_t* u_idx, int64_t n_rows,
                                       int64_t item_begin, int64_t item_count, const int64_t* excl_indptr,
                                       const int32_t* excl_indices, int64_t k, float* scores_ws, void* planes_ws,
                                       int64_t planes_bytes, float* out_vals, int32_t* out_idx, int32_t* status,
                                       hsk_stream_t stream_) {
  HSK_REQUIRE(user_emb && item_emb && u_idx && scores_ws, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_users > 0 && n_items > 0 && dim > 0, HSK_ERR_INVALID, "bad table shape");
  HSK_REQUIRE(item_begin >= 0 && item_count > 0 && item_begin + item_count <= n_items, HSK_ERR_INVALID,
              "item shard [%lld, +%lld) outside [0, %lld)", (long long)item_begin, (long long)item_count,
              (long long)n_items);
  HSK_REQUIRE(n_rows >= 0 && item_count < 0x7fffffff && n_rows < 0x7fffffff, HSK_ERR_INVALID, "bad n_rows");
  HSK_REQUIRE((excl_indptr == nullptr) == (excl_indices == nullptr), HSK_ERR_INVALID,
              "exclude CSR needs both indptr and indices");
  HSK_REQUIRE(k >= 0 && k <= TOPK_MAX, HSK_ERR_UNSUPPORTED, "k %lld outside [0, %d]", (long long)k, TOPK_MAX);
  HSK_REQUIRE(k <= item_count, HSK_ERR_INVALID, "k %lld > item_count %lld", (long long)k, (long long)item_count);
  HSK_REQUIRE(k == 0 || (out_vals && out_idx), HSK_ERR_INVALID, "top-k outputs must not be NULL");
  if (n_rows == 0) return HSK_OK;
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid((unsigned)hsk_ceil_div(item_count, GEMM_BN), (unsigned)hsk_ceil_div(n_rows, GEMM_BM));
  const bool vec4 = (dim % 4 == 0) && ((((uintptr_t)user_emb | (uintptr_t)item_emb) & 15) == 0);
  const int x3 = hsk_eval_x3();
#define HSK_SCORE_GEMM(KERNEL)                                                                                        \
  KERNEL<<<grid, 256, 0, stream>>>(user_emb, item_emb, item_bias, user_bias, global_bias, (int)n_users, (int)dim, u_idx, \
                                   (int)n_rows, (long long)item_begin, (int)item_count, scores_ws, status)
  static const int planes_on = getenv("HSK_EVAL_PLANES") ? atoi(getenv("HSK_EVAL_PLANES")) : 1;
  if (x3 && vec4 && planes_on && planes_ws && ((uintptr_t)planes_ws & 255) == 0 &&
      planes_bytes >= hsk_mf_eval_planes_bytes(n_rows, item_count, dim)) {
    // the operands' pieces, made once for the call
    const int64_t Dp = hsk_align_up(dim, GEMM_BK);
    const int a_rows = (int)hsk_align_up(n_rows, GEMM_W_BM), b_rows = (int)hsk_align_up(item_count, GEMM_W_BN);
    if (x3 == 2) {
      // form 2: fp16 pairs (4 of the region's 6 bytes per element; the two scale words sit in its last 256 bytes),
      // the 256 x 256 kernel whatever the shape
      _Float16* Ah = (_Float16*)planes_ws;
      _Float16* Bh = (_Float16*)((char*)planes_ws + hsk_align_up(6 * (int64_t)a_rows * Dp, 256));
      uint32_t* amax = (uint32_t*)((char*)planes_ws + hsk_mf_eval_planes_bytes(n_rows, item_count, dim) - 256);
      HSK_HIP(hipMemsetAsync(amax, 0, 8, stream));
      hsk_eval_split_planes_h2(user_emb, u_idx, 0, n_users, (int)n_rows, a_rows, (int)dim, amax, Ah, stream);
      hsk_eval_split_planes_h2(item_emb, nullptr, item_begin, n_items, (int)item_count, b_rows, (int)dim, amax + 1, Bh, stream);
      HSK_LAUNCH_CHECK();
      int rc = hsk_eval_set_wide_lds();
      if (rc) return rc;
      dim3 wgrid((unsigned)hsk_ceil_div(item_count, GEMM_W_BN), (unsigned)hsk_ceil_div(n_rows, GEMM_W_BM));
      k_score_gemm_h2_wide<<<wgrid, 256, GEMM_H_LDS_BYTES, stream>>>(item_bias, user_bias, global_bias, (int)n_users, (int)Dp,
                                                                     u_idx, (int)n_rows, (long long)item_begin,
                                                                     (int)item_count, scores_ws, status, Ah, Bh, a_rows, b_rows,
                                                                     amax);
    } else {
    __bf16* Apl = (__bf16*)planes_ws;
    __bf16* Bpl = (__bf16*)((char*)planes_ws + hsk_align_up(6 * (int64_t)a_rows * Dp, 256));
    // 256 x 256 block tiles (one wave per SIMD) once there are enough of them to fill the chip a few times over;
    // below that the 128 x 128 kernel, whose four times as many workgroups fill it sooner.  Same bits either way.
    static const int wide_on = getenv("HSK_EVAL_WIDE") ? atoi(getenv("HSK_EVAL_WIDE")) : 1;
    const int64_t wide_blocks = hsk_ceil_div(n_rows, GEMM_W_BM) * hsk_ceil_div(item_count, GEMM_W_BN);
    const bool wide = wide_on == 2 || (wide_on && wide_blocks >= 3 * 256);
    hsk_eval_split_planes_k(user_emb, u_idx, 0, n_users, (int)n_rows, a_rows, (int)dim, Apl, stream, wide ? 16 : GEMM_BK);
    hsk_eval_split_planes_k(item_emb, nullptr, item_begin, n_items, (int)item_count, b_rows, (int)dim, Bpl, stream,
                            wide ? 16 : GEMM_BK);
    HSK_LAUNCH_CHECK();
    if (wide) {
      int rc = hsk_eval_set_wide_lds();
      if (rc) return rc;
      dim3 wgrid((unsigned)hsk_ceil_div(item_count, GEMM_W_BN), (unsigned)hsk_ceil_div(n_rows, GEMM_W_BM));
      k_score_gemm_x3_wide<<<wgrid, 256, GEMM_W_LDS_BYTES, stream>>>(item_bias, user_bias, global_bias, (int)n_users,
                                                                     (int)Dp, u_idx, (int)n_rows, (long long)item_begin,
                                                                     (int)item_count, scores_ws, status, Apl, Bpl, a_rows,
                                                                     b_rows);
    } else {
      k_score_gemm_x3<true, true><<<grid, 256, 0, stream>>>(user_emb, item_emb, item_bias, user_bias, global_bias,
                                                           (int)n_users, (int)dim, u_idx, (int)n_rows,
                                                           (long long)item_begin, (int)item_count, scores_ws, status, Apl,
                                                           Bpl, a_rows, b_rows);
    }
    }
  } else if (x3) {
    if (vec4) HSK_SCORE_GEMM(k_score_gemm_x3<true>); else HSK_SCORE_GEMM(k_score_gemm_x3<false>);
  } else {
    if (vec4) HSK_SCORE_GEMM(k_score_gemm<true>); else HSK_SCORE_GEMM(k_score_gemm<false>);
  }
#undef HSK_SCORE_GEMM
  HSK_LAUNCH_CHECK();
  if (excl_indptr) {
    k_mask_excluded<<<(unsigned)hsk_ceil_div(n_rows, 4), 256, 0, stream>>>(
        u_idx, (int)n_rows, (int)n_users, excl_indptr, excl_indices, (long long)item_begin, (int)item_count, scores_ws);
    HSK_LAUNCH_CHECK();
  }
  if (k > 0)
    return hsk_launch_topk_i32(scores_ws, n_rows, item_count, item_count, k, (long long)item_begin, out_vals, out_idx,
                               stream);
  return HSK_OK;
}

// Threshold seeding for the in-GEMM selection (hsk_eval_fused.hip: k_score_topk_wide).  The k-th best score of a row
// over ANY subset of the admissible items is a lower bound of its k-th best over all of them.  This scores the first
// `sample_count` items of the range for every row (the 256 x 256 GEMM on the pieces the fused call has made anyway, the
// exclusion mask, the row top-k) and leaves the k-th best as an ordered key in gthr[row]: the selection then starts with
// a threshold only ~k / sample_count of the scores reach, instead of appending whole tiles until its lists fill.
__global__ __launch_bounds__(256) void k_seed_keys(const float* __restrict__ vals, int n_rows, int k, uint32_t* __restrict__ gthr) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const uint32_t b = __float_as_uint(vals[(long long)r * k + k - 1]);
  gthr[r] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // = fg_f2key of hsk_eval_fused.hip
}

int hsk_eval_seed_thresholds(const float* item_bias, const float* user_bias, const float* global_bias, int n_users, int Dp,
                             const int64_t* u_idx, int n_rows, long long item_begin, int sample_count,
                             const int64_t* excl_indptr, const int32_t* excl_indices, int k, const void* Apl,
                             const void* Bpl, int a_rows, int b_rows, const uint32_t* amax, float* scores_ws, float* vals_ws,
                             int32_t* idx_ws, uint32_t* gthr, int32_t* status, hipStream_t stream) {
  int rc0 = hsk_eval_set_wide_lds();
  if (rc0) return rc0;
  dim3 wgrid((unsigned)hsk_ceil_div(sample_count, GEMM_W_BN), (unsigned)hsk_ceil_div(n_rows, GEMM_W_BM));
  if (amax)   // form 2: the pieces are fp16 pairs
    k_score_gemm_h2_wide<<<wgrid, 256, GEMM_H_LDS_BYTES, stream>>>(item_bias, user_bias, global_bias, n_users, Dp, u_idx, n_rows,
                                                                   item_begin, sample_count, scores_ws, status,
                                                                   (const _Float16*)Apl, (const _Float16*)Bpl, a_rows, b_rows, amax);
  else
    k_score_gemm_x3_wide<<<wgrid, 256, GEMM_W_LDS_BYTES, stream>>>(item_bias, user_bias, global_bias, n_users, Dp, u_idx, n_rows,
                                                                   item_begin, sample_count, scores_ws, status,
                                                                   (const __bf16*)Apl, (const __bf16*)Bpl, a_rows, b_rows);
  HSK_LAUNCH_CHECK();
  if (excl_indptr) {
    k_mask_excluded<<<(unsigned)hsk_ceil_div(n_rows, 4), 256, 0, stream>>>(u_idx, n_rows, n_users, excl_indptr, excl_indices,
                                                                          item_begin, sample_count, scores_ws);
    HSK_LAUNCH_CHECK();
  }
  int rc = hsk_launch_topk_i32(scores_ws, n_rows, sample_count, sample_count, k, item_begin, vals_ws, idx_ws, stream);
  if (rc) return rc;
  k_seed_keys<<<(unsigned)hsk_ceil_div(n_rows, 256), 256, 0, stream>>>(vals_ws, n_rows, k, gthr);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

extern "C" int hsk_topk_dense(const float* logits, int64_t rows, int64_t cols, int64_t ld, int64_t k, float* out_vals,
                              int64_t* out_idx, hsk_stream_t stream_) {
  HSK_REQUIRE(logits && out_vals && out_idx, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && cols < 0x7fffffff, HSK_ERR_INVALID, "bad matrix shape");
  HSK_REQUIRE(k >= 1 && k <= TOPK_MAX, HSK_ERR_UNSUPPORTED, "k %lld outside [1, %d]", (long long)k, TOPK_MAX);
  HSK_REQUIRE(k <= cols, HSK_ERR_INVALID, "k %lld > cols %lld", (long long)k, (long long)cols);
  if (rows == 0) return HSK_OK;
  const int kpad = hsk_next_pow2((int)k);
  if (cols + 3 > 256 * 48 && cols <= 256 * 64)
    k_topk_rows_wide<int64_t><<<(unsigned)rows, 256, 0, (hipStream_t)stream_>>>(logits, ld, (int)cols, (int)k, kpad, 0ll,
                                                                                 out_vals, out_idx);
  else
    k_topk_rows<int64_t><<<(unsigned)rows, 256, 0, (hipStream_t)stream_>>>(logits, ld, (int)cols, (int)k, kpad, 0ll,
                                                                            out_vals, out_idx);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

extern "C" int hsk_topk_merge(const float* vals, const int32_t* idx, int64_t n_parts, int64_t rows, int64_t k,
                              float* out_vals, int32_t* out_idx, hsk_stream_t stream_) {
  HSK_REQUIRE(vals && idx && out_vals && out_idx, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_parts >= 1 && rows >= 0 && k >= 1, HSK_ERR_INVALID, "bad sizes");
  HSK_REQUIRE(n_parts * k <= 4096, HSK_ERR_UNSUPPORTED, "n_parts*k = %lld > 4096", (long long)(n_parts * k));
  if (rows == 0) return HSK_OK;
  const int npad = hsk_next_pow2((int)(n_parts * k));
  k_topk_merge<<<(unsigned)rows, 256, 0, (hipStream_t)stream_>>>(vals, idx, (int)n_parts, (int)rows, (int)k, npad,
                                                                 out_vals, out_idx);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

extern "C" int hsk_rank_metrics(const int32_t* topk_idx, int64_t n_rows, int64_t k_max, const int64_t* u_idx,
                                int64_t n_users, const int64_t* label_indptr, const int32_t* label_indices, const int32_t* ks,
                                int32_t n_ks, float* out, hsk_stream_t stream_) {
  HSK_REQUIRE(topk_idx && u_idx && label_indptr && label_indices && ks && out, HSK_ERR_INVALID,
              "NULL pointer argument");
  HSK_REQUIRE(n_ks >= 1 && n_ks <= HSK_MAX_KS, HSK_ERR_UNSUPPORTED, "n_ks %d outside [1, %d]", n_ks, HSK_MAX_KS);
  HSK_REQUIRE(n_rows >= 0 && k_max >= 1 && n_users > 0 && n_users < 0x7fffffff, HSK_ERR_INVALID, "bad sizes");
  hsk_ks kk;
  kk.n = n_ks;
  for (int t = 0; t < HSK_MAX_KS; ++t) kk.k[t] = 0;
  for (int t = 0; t < n_ks; ++t) {
    HSK_REQUIRE(ks[t] >= 1 && ks[t] <= k_max, HSK_ERR_INVALID, "ks[%d]=%d outside [1, k_max=%lld]", t, ks[t],
                (long long)k_max);
    kk.k[t] = ks[t];
  }
  if (n_rows == 0) return HSK_OK;
  k_rank_metrics<<<(unsigned)hsk_ceil_div(n_rows, 4), 256, 0, (hipStream_t)stream_>>>(
      topk_idx, (int)n_rows, (int)k_max, u_idx, (int)n_users, label_indptr, label_indices, kk, out);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}
