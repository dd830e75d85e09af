// hsk_eval_fused.hip -- full-catalogue scoring with the top-k selection INSIDE the score GEMM: the [rows, items] score
// matrix (1 GB per 2048 users at 131 072 items) is never written to memory.
//
// Reference semantics: eval/eval.py:237-253 (scores = U_batch x I^T + biases, -inf on the excluded positives) followed
// by eval/eval.py:63 (logits.topk(k)).  Same arithmetic as k_score_gemm (hsk_eval.hip): exact-fp32 MFMA, biases added in
// the reference's order, order of the result = (score descending, item id ascending).
//
// Work split: grid = (S item splits, row blocks of 128 users).  A workgroup walks the 128-column tiles of its split;
// the 128 x 128 scores of a tile exist only in the MFMA accumulators.  Per row it keeps
//     thr   its current k-th best score (-inf while it holds fewer than k)
//     cand  up to HSK_SEL_CAP candidates (key << 32 | ~item id) in a global scratch slab, cnt of them
// and the epilogue appends the tile's scores that reach thr.  After the first tiles a random score reaches thr with
// probability ~k / (columns seen), so appends are rare: ~k ln(n / k) per row in total.  Whenever a row may overflow
// during the next tile (cnt > CAP - 128), one wave sorts its candidates (bitonic, in LDS), keeps the best k and raises
// thr.  At the end every row is compacted once more and its k best go out, sorted; hsk_topk_merge's kernel joins the S
// lists of a row.  Excluded (user, item) pairs become -inf exactly as in the reference (they can still appear in the
// top k of a user with fewer than k admissible items, lowest item id first).
#include "hsk_common.h"
#include "hsk_gemm_wide_h2.h"
#include <stdlib.h>

#include <algorithm>

typedef float hsk_f32x16 __attribute__((ext_vector_type(16)));
typedef float hsk_f32x4 __attribute__((ext_vector_type(4)));

#define FG_BM 128
#define FG_X3_LDK 40   // = GEMM_X3_LDK of hsk_eval.hip: LDS row stride of a bf16 plane, in elements
typedef __bf16 fg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fg_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fg_split3(float x, __bf16& h, __bf16& m, __bf16& l) {   // = hsk_split3 of hsk_eval.hip
  h = (__bf16)x;
  const float hf = (float)h;
  const bool fin = __builtin_isfinite(hf);
  const float r1 = fin ? x - hf : 0.f;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}
int hsk_eval_x3();   // hsk_eval.hip: which arithmetic the score GEMMs use
#define FG_BN 128
#define FG_BK 32
#define FG_LDS_STRIDE (FG_BK + 4)
#ifndef HSK_SEL_CAP
#define HSK_SEL_CAP 512            // candidates a row can hold between two compactions
#endif
#define HSK_SEL_KMAX 128           // k supported by the fused path
#ifndef HSK_SEL_TRIG
// a row is compacted when it holds more than this (<= CAP - 128, > KMAX).  Between two compactions the row's threshold is
// stale and lets too much through; a compaction is a wave-wide radix select.  Measured at the lfm2b shape with the
// bf16x3 core: 384: 895-902 k users/s, 320: 911-913, 256: 911-912, 192: 879-882
#define HSK_SEL_TRIG 320
#endif

__device__ __forceinline__ uint32_t fg_f2key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // ascending float order == ascending key order
}
__device__ __forceinline__ float fg_key2f(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

// one wave sorts n <= HSK_SEL_CAP composite keys of `s` (LDS, padded with 0 up to npad, a power of two) descending
__device__ __forceinline__ void fg_wave_bitonic_desc(unsigned long long* s, int npad, int lane) {
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __builtin_amdgcn_wave_barrier();   // LDS operations of one wave execute in order
      for (int t = lane; t < (npad >> 1); t += 64) {
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
  __builtin_amdgcn_wave_barrier();
}

// X3: the GEMM core of k_score_gemm_x3 (hsk_eval.hip): three bf16 pieces per fp32 operand, six bf16 MFMAs per block, the
// same order of operations -- so the two paths stay bit-equal in either arithmetic
// MODE 2: the same, but the three bf16 pieces of both operands come ready-made from global memory (k_split_planes, once
// per call): the split's arithmetic leaves the k loop.  MODE 3: only the items' pieces do.
typedef unsigned fg_u32x4 __attribute__((ext_vector_type(4)));   // (an array of HIP's uint4 structs ended up in scratch)
template <bool VEC4, int MODE>
__global__ __launch_bounds__(256, 2) void k_score_topk(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                    const float* __restrict__ Ib, const float* __restrict__ Ub,
                                                    const float* __restrict__ gb, int n_users, int D,
                                                    const int64_t* __restrict__ u_idx, int n_rows,
                                                    long long item_begin, int item_count, int tiles_per_split,
                                                    const int64_t* __restrict__ excl_indptr,
                                                    const int32_t* __restrict__ excl_indices, int k, int n_splits,
                                                    unsigned long long* __restrict__ cand_ws,
                                                    float* __restrict__ part_vals, int32_t* __restrict__ part_idx,
                                                    int32_t* status, const __bf16* __restrict__ Apl = nullptr,
                                                    const __bf16* __restrict__ Bpl = nullptr,
                                                    long long a_stride = 0, long long b_stride = 0) {
  constexpr bool X3 = MODE != 0, APL = MODE == 2, BPL = MODE >= 2;   // which operands come as ready-made planes
  // fp32 tiles [row][32+4], or (X3) three bf16 planes [row][32+8] each
  constexpr int TILE_BYTES = X3 ? 3 * FG_BM * FG_X3_LDK * 2 : FG_BM * FG_LDS_STRIDE * 4;
  __shared__ __attribute__((aligned(16))) unsigned char As_raw[TILE_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Bs_raw[TILE_BYTES];
  float* As = reinterpret_cast<float*>(As_raw);
  float* Bs = reinterpret_cast<float*>(Bs_raw);
  __bf16* Ap = reinterpret_cast<__bf16*>(As_raw);   // plane pl at Ap + pl * FG_BM * FG_X3_LDK
  __bf16* Bp = reinterpret_cast<__bf16*>(Bs_raw);
  __shared__ unsigned long long srt[4][HSK_SEL_KMAX];   // per-wave scratch of the final sort (k <= 128 keys)
  __shared__ unsigned int hist[4][256];                 // per-wave digit histogram of the selection
  __shared__ int urow[FG_BM];
  __shared__ float ubias[FG_BM];
  __shared__ float thr[FG_BM];      // current k-th best score of the row (-inf while it holds fewer than k)
  __shared__ int cnt[FG_BM];
  __shared__ uint32_t emask[FG_BM][4];      // exclusion bits of the current tile, one row of 128 bits per user
  __shared__ long long eptr[FG_BM], eend[FG_BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroup -> (row block, split): consecutive workgroup ids go round the 8 XCDs, so id % 8 picks the row block
  // inside a group of 8 and the next 8 ids the next split: one XCD then runs all the splits of only a few row blocks at
  // a time -- their user rows (256 KB per row block, re-read for every tile) stay in its 4 MB L2, and an item tile is
  // shared by the row blocks that walk the same split side by side.
  const int n_row_blocks = (n_rows + FG_BM - 1) / FG_BM;
  const int per_group = 8 * n_splits;
  const int grp = blockIdx.x / per_group, rem = blockIdx.x - grp * per_group;
  const int split = rem >> 3;
  const int rb = grp * 8 + (rem & 7);
  if (rb >= n_row_blocks) return;
  const int m0 = rb * FG_BM;
  const int n_tiles = (item_count + FG_BN - 1) / FG_BN;
  const int t_lo = split * tiles_per_split, t_hi = min(n_tiles, t_lo + tiles_per_split);
  unsigned long long* __restrict__ cand =
      cand_ws + ((long long)rb * n_splits + split) * ((long long)FG_BM * HSK_SEL_CAP);

  if (tid < FG_BM) {
    const int r = m0 + tid;
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
    ubias[tid] = Ub ? Ub[u] : 0.f;
    thr[tid] = -INFINITY;
    cnt[tid] = 0;
    long long lo = 0, hi = 0;
    if (excl_indptr && r < n_rows) {
      lo = excl_indptr[u];
      hi = excl_indptr[u + 1];
      // first excluded item at or beyond this split's first column (the row is sorted)
      const long long first_col = item_begin + (long long)t_lo * FG_BN;
      long long l = lo, h = hi;
      while (l < h) {
        const long long mid = (l + h) >> 1;
        if (excl_indices[mid] < first_col)
          l = mid + 1;
        else
          h = mid;
      }
      lo = l;
    }
    eptr[tid] = lo;
    eend[tid] = hi;
  }
  __syncthreads();
  // threads 0..127 walk "their" user's exclusion row; the next excluded item sits in a register, so a tile without
  // excluded items (the usual case) costs no memory access
  long long e_p = 0, e_end = 0, e_next = 0x7fffffffffffffffll;
  if (tid < FG_BM) {
    e_p = eptr[tid];
    e_end = eend[tid];
    if (e_p < e_end) e_next = excl_indices[e_p];
  }

  const int srow = tid >> 3;
  const int scol = (tid & 7) * 4;
  const int half = lane >> 5;
  const int l32 = lane & 31;
  const float gbv = gb ? gb[0] : 0.f;

  // One wave, one row with n > k candidates: keep the best k (unsorted) and raise thr to the k-th best's key.
  // The k-th largest composite key is found by an MSB-first radix SELECT (8 passes of 8 bits over <= CAP keys held in
  // registers, digit histogram in LDS) -- ~10x cheaper than sorting the list, and nothing here needs an order: only
  // the final lists are sorted.  Composite keys are unique (they end in the item id), so exactly k keys are >= it.
  constexpr int KPL = HSK_SEL_CAP / 64;   // keys per lane
  auto compact_row = [&](int rloc) {
    const int n = cnt[rloc];
    if (n <= k) return;
    unsigned long long kv[KPL];
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
      const int j = lane + 64 * m;
      kv[m] = (j < n) ? cand[(long long)rloc * HSK_SEL_CAP + j] : 0ull;
    }
    unsigned int* h = hist[wave];
    unsigned long long prefix = 0ull, pmask = 0ull;
    int need = k;
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
#pragma unroll
      for (int q = 0; q < 4; ++q) h[lane * 4 + q] = 0u;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int m = 0; m < KPL; ++m)
        if (kv[m] != 0ull && (kv[m] & pmask) == prefix) atomicAdd(&h[(unsigned)(kv[m] >> shift) & 255u], 1u);
      __builtin_amdgcn_wave_barrier();
      // lane l owns digits 4l .. 4l+3; keys with a higher digit than d: sum over digits > d
      const unsigned c0 = h[lane * 4], c1 = h[lane * 4 + 1], c2 = h[lane * 4 + 2], c3 = h[lane * 4 + 3];
      const int mine = (int)(c0 + c1 + c2 + c3);
      int above = mine;   // inclusive suffix sum over lanes >= this one
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_down(above, off, 64);
        if (lane + off < 64) above += t;
      }
      above -= mine;      // keys in digits owned by higher lanes
      // the digit that holds the need-th largest key: the unique (lane, q) with above_d < need <= above_d + count_d
      int found_digit = -1, found_need = 0;
      {
        int a3 = above, a2 = above + (int)c3, a1 = a2 + (int)c2, a0 = a1 + (int)c1;
        if (a3 < need && need <= a3 + (int)c3) { found_digit = lane * 4 + 3; found_need = need - a3; }
        else if (a2 < need && need <= a2 + (int)c2) { found_digit = lane * 4 + 2; found_need = need - a2; }
        else if (a1 < need && need <= a1 + (int)c1) { found_digit = lane * 4 + 1; found_need = need - a1; }
        else if (a0 < need && need <= a0 + (int)c0) { found_digit = lane * 4; found_need = need - a0; }
      }
      const unsigned long long who = __ballot(found_digit >= 0);
      const int src = __builtin_ctzll(who);
      const int digit = __shfl(found_digit, src, 64);
      need = __shfl(found_need, src, 64);
      prefix |= (unsigned long long)digit << shift;
      pmask |= 0xffull << shift;
    }
    // prefix == the k-th largest composite key
    int base = 0;
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
      const bool keep = kv[m] >= prefix && kv[m] != 0ull;
      const unsigned long long bm = __ballot(keep);
      if (keep) cand[(long long)rloc * HSK_SEL_CAP + base + __popcll(bm & ((1ull << lane) - 1ull))] = kv[m];
      base += __popcll(bm);
    }
    if (lane == 0) {
      cnt[rloc] = k;
      thr[rloc] = fg_key2f((uint32_t)(prefix >> 32));
    }
    __builtin_amdgcn_wave_barrier();
  };
  // final: one wave, one row: its <= k survivors sorted (key descending, item id ascending) in LDS
  auto sort_row = [&](int rloc, int keep) {
    unsigned long long* sr = srt[wave];
    for (int j = lane; j < HSK_SEL_KMAX; j += 64) sr[j] = (j < keep) ? cand[(long long)rloc * HSK_SEL_CAP + j] : 0ull;
    fg_wave_bitonic_desc(sr, HSK_SEL_KMAX, lane);
  };

  float4 ra[4], rbv[4];   // register staging of the next k-tile (A: user rows, B: item rows)
  fg_u32x4 rpa[6], rpb[6];   // planes: 6 x 16 bytes of the tile's 24 KB per operand
  for (int tile = t_lo; tile < t_hi; ++tile) {
    const int n0 = tile * FG_BN;
    // exclusion bits of this tile: the user's sorted CSR row is consumed as the tiles advance
    if (tid < FG_BM) {
      uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
      const long long col_lo = item_begin + n0, col_hi = col_lo + FG_BN;
      while (e_next < col_hi) {
        const int c = (int)(e_next - col_lo);
        if (c >= 0) {
          const uint32_t bit = 1u << (c & 31);
          if (c < 32) b0 |= bit; else if (c < 64) b1 |= bit; else if (c < 96) b2 |= bit; else b3 |= bit;
        }
        ++e_p;
        e_next = (e_p < e_end) ? (long long)excl_indices[e_p] : 0x7fffffffffffffffll;
      }
      emask[tid][0] = b0;
      emask[tid][1] = b1;
      emask[tid][2] = b2;
      emask[tid][3] = b3;
    }

    hsk_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    auto load_tile = [&](int n0, int k0) {
      // planes are padded to whole tiles in rows and to 32 in k: no bounds to check; [k-tile][row][piece][32] makes a
      // 128-row tile's three pieces of one k-tile 24 KB of consecutive bytes: chunk c of 1536 is 16 bytes at 16 c
      if (APL) {
        const __bf16* base = Apl + (long long)(k0 / FG_BK) * a_stride + (long long)m0 * 96;
#pragma unroll
        for (int i = 0; i < 6; ++i) rpa[i] = *reinterpret_cast<const fg_u32x4*>(base + (tid + 256 * i) * 8);
      }
      if (BPL) {
        const __bf16* base = Bpl + (long long)(k0 / FG_BK) * b_stride + (long long)n0 * 96;
#pragma unroll
        for (int i = 0; i < 6; ++i) rpb[i] = *reinterpret_cast<const fg_u32x4*>(base + (tid + 256 * i) * 8);
      }
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = srow + pass * 32;
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!APL && m0 + r < n_rows) {
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
        if (!BPL && n0 + r < item_count) {
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
        if (!APL) ra[pass] = va;
        if (!BPL) rbv[pass] = vb;
      }
    };
    auto store_tile = [&]() {
      if (APL || BPL) {
        constexpr int PL = FG_BM * FG_X3_LDK;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int c = tid + 256 * i, r = c / 12, j = c - r * 12, pl = j >> 2, kc = (j & 3) * 8;
          if (APL) *reinterpret_cast<fg_u32x4*>(&Ap[pl * PL + r * FG_X3_LDK + kc]) = rpa[i];
          if (BPL) *reinterpret_cast<fg_u32x4*>(&Bp[pl * PL + r * FG_X3_LDK + kc]) = rpb[i];
        }
      }
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = srow + pass * 32;
        if (X3) {
          constexpr int PL = FG_BM * FG_X3_LDK;
          if (!APL) {
            fg_bf16x4 a1, a2, a3;
            const float av[4] = {ra[pass].x, ra[pass].y, ra[pass].z, ra[pass].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              __bf16 x1, x2, x3;
              fg_split3(av[e], x1, x2, x3);
              a1[e] = x1; a2[e] = x2; a3[e] = x3;
            }
            *reinterpret_cast<fg_bf16x4*>(&Ap[0 * PL + r * FG_X3_LDK + scol]) = a1;
            *reinterpret_cast<fg_bf16x4*>(&Ap[1 * PL + r * FG_X3_LDK + scol]) = a2;
            *reinterpret_cast<fg_bf16x4*>(&Ap[2 * PL + r * FG_X3_LDK + scol]) = a3;
          }
          if (!BPL) {
            fg_bf16x4 b1, b2, b3;
            const float bv[4] = {rbv[pass].x, rbv[pass].y, rbv[pass].z, rbv[pass].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              __bf16 x1, x2, x3;
              fg_split3(bv[e], x1, x2, x3);
              b1[e] = x1; b2[e] = x2; b3[e] = x3;
            }
            *reinterpret_cast<fg_bf16x4*>(&Bp[0 * PL + r * FG_X3_LDK + scol]) = b1;
            *reinterpret_cast<fg_bf16x4*>(&Bp[1 * PL + r * FG_X3_LDK + scol]) = b2;
            *reinterpret_cast<fg_bf16x4*>(&Bp[2 * PL + r * FG_X3_LDK + scol]) = b3;
          }
        } else {
          *reinterpret_cast<float4*>(&As[r * FG_LDS_STRIDE + scol]) = ra[pass];
          *reinterpret_cast<float4*>(&Bs[r * FG_LDS_STRIDE + scol]) = rbv[pass];
        }
      }
    };
    if (tile == t_lo) load_tile(n0, 0);   // later tiles: loaded under the previous tile's epilogue
    store_tile();
    __syncthreads();
    for (int k0 = 0; k0 < D; k0 += FG_BK) {
      const bool has_next = k0 + FG_BK < D;
      if (has_next) load_tile(n0, k0 + FG_BK);
      if (X3) {
        constexpr int PL = FG_BM * FG_X3_LDK;
#pragma unroll
        for (int s16 = 0; s16 < 2; ++s16) {   // two k = 16 steps per tile (see k_score_gemm_x3)
          fg_bf16x8 a[3][2], b[3][2];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
              a[pl][i] = *reinterpret_cast<const fg_bf16x8*>(&Ap[pl * PL + (wm * 64 + i * 32 + l32) * FG_X3_LDK + 16 * s16 + 8 * half]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
              b[pl][j] = *reinterpret_cast<const fg_bf16x8*>(&Bp[pl * PL + (wn * 64 + j * 32 + l32) * FG_X3_LDK + 16 * s16 + 8 * half]);
          }
          constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][i], b[TB[t]][j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + l32) * FG_LDS_STRIDE + half * 16 + q * 4]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[j] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + j * 32 + l32) * FG_LDS_STRIDE + half * 16 + q * 4]);
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
      }
      __syncthreads();
      if (has_next) {
        store_tile();
        __syncthreads();
      }
    }

    if (tile + 1 < t_hi) load_tile(n0 + FG_BN, 0);   // in flight during the epilogue and the compactions
    // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // the 16 rows of this lane's accumulator registers: their thresholds in one burst of LDS reads (thr only changes in
      // the compactions, behind the barrier below) -- read one by one in front of each compare they were 64 exposed LDS
      // latencies per tile and wave.  Rows past n_rows: +inf, nothing passes.
      float t[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rloc = wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        t[q] = (m0 + rloc < n_rows) ? thr[rloc] : INFINITY;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cloc = wn * 64 + j * 32 + l32;
        const int col = n0 + cloc;
        if (col >= item_count) continue;
        const float ib = Ib ? Ib[item_begin + col] : 0.f;
        const uint32_t nid = ~(uint32_t)(item_begin + col);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rloc = wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
          float o = acc[i][j][q];
          if (Ub) o += ubias[rloc];   // reference order: += u_bias, += i_bias, += global_bias
          if (Ib) o += ib;
          if (gb) o += gbv;
          // the common case ends here: one compare per score (NaN passes, as in torch.topk)
          if (!(o < t[q]) && m0 + rloc < n_rows) {
            if ((emask[rloc][cloc >> 5] >> (cloc & 31)) & 1u) o = -INFINITY;
            if (!(o < t[q])) {   // (t[q] == thr[rloc] for a valid row)
              const int slot = atomicAdd(&cnt[rloc], 1);
              cand[(long long)rloc * HSK_SEL_CAP + slot] = ((unsigned long long)fg_f2key(o) << 32) | nid;
            }
          }
        }
      }
    }
    __syncthreads();
    // rows that the next tile could overflow: compact now (wave w owns rows 32w .. 32w+31)
    {   // (one LDS read per lane and a ballot, not 32 reads in a row)
      const int fill = (lane < 32) ? cnt[wave * 32 + lane] : 0;
      unsigned long long due = __ballot(fill > HSK_SEL_TRIG);
      while (due) {
        const int rr = __builtin_ctzll(due);
        due &= due - 1;
        compact_row(wave * 32 + rr);
      }
    }
    __syncthreads();
  }

  // final: every row down to its k best, sorted, out to this split's slice of the partial lists
  for (int rr = 0; rr < 32; ++rr) {
    const int rloc = wave * 32 + rr;
    const int row = m0 + rloc;
    if (row >= n_rows) continue;
    compact_row(rloc);
    const int keep = min(cnt[rloc], k);
    sort_row(rloc, keep);
    const long long dst = ((long long)split * n_rows + row) * k;
    for (int j = lane; j < k; j += 64) {
      float v = -INFINITY;
      int32_t id = 0x7fffffff;   // pad of a split that holds fewer than k items: sorts behind everything real
      if (j < keep) {
        const unsigned long long c = srt[wave][j];
        v = fg_key2f((uint32_t)(c >> 32));
        id = (int32_t)(~(uint32_t)c);
      }
      part_vals[dst + j] = v;
      part_idx[dst + j] = id;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// =============================================================================================
// The in-GEMM selection on the 256 x 256 / one-wave-per-SIMD core (hsk_gemm_wide.h)
// =============================================================================================
// Round 3 built this once with the 128 x 128 kernel's epilogue (a compare, an exclusion test and a possible append per
// score, ~12 000 instructions per tile and wave) and measured it SLOWER than the 128 x 128 kernel: with one wave per SIMD
// nothing covers the epilogue.  This one is lean where it has to be:
//   filter   per accumulator register (= one row, this lane's FOUR columns of it): the four scores against the row's
//            threshold, one ballot.  64 such sites per tile and wave; the common case ends there.
//   enqueue  lanes that passed put (4 scores, row, column) into a per-wave LDS queue at ballot-ranked positions
//            (no atomics, ~10 instructions for a site where any lane passed) -- the queue lives in the operand stages,
//            which are idle during the epilogue.
//   drain    after each band of 16 sites the wave walks its queue DENSELY, 64 entries at a time: exact re-test of each of
//            the four scores, exclusion bit, slot from the row's LDS counter, the 8-byte candidate out to the row's slab.
// The divergent part (which lanes pass where) thus costs a few instructions per site instead of an append path per site,
// and the append path runs on full waves.  ~1 700 instructions per tile and wave in steady state against ~120 000 cycles
// of MFMAs.  Candidates, thresholds, compactions (MSB-first radix select by one wave per row) and the final lists are the
// 128 x 128 kernel's; every score sees the same MFMA sequence: values, ids and order are bit-identical
// (test_fused_topk_equals_materialised_topk).
#define FGW_CAP 640            // candidates a row can hold between two compactions: TRIG + one whole tile of 256 columns
#define FGW_TRIG 384
#define FGW_Q 1536             // queue entries per wave: drained when it holds > 512 (a band adds at most 16 sites x 64 lanes)
#define FGW_STATE_BYTES (3 * 256 * 4 + 256 * 8 * 4)   // thr, cnt, ubias [256] + emask [256][8]
#define FGW_LDS_BYTES (GEMM_W_LDS_BYTES + FGW_STATE_BYTES)
#define FGW_H2_LDS_BYTES (GEMM_H_LDS_BYTES + FGW_STATE_BYTES)   // on the fp16-pair core (hsk_gemm_wide_h2.h)
static_assert(FGW_LDS_BYTES <= 160 * 1024, "LDS of k_score_topk_wide");
static_assert(4 * FGW_Q * 20 + 4 * 256 * 4 + 4 * HSK_SEL_KMAX * 8 <= GEMM_H_LDS_BYTES && GEMM_H_LDS_BYTES <= GEMM_W_LDS_BYTES,
              "queues + select scratch fit the idle stages");

// HAS_IB: item bias present; EXTRA: a user and / or global bias as well (the additions follow the reference's order and
// are left out, not replaced by + 0, where a bias is absent: -0 + 0 would change a sign bit the materialised path keeps)
// H2: the fp16-pair / three-product core (form 2 of hsk_eval_set_arith): Apl / Bpl are its planes, amax the two maxima the
// planes were scaled by; every score is the accumulator times 2^-(e_a + e_b), exact
template <bool HAS_IB, bool EXTRA, bool H2 = false>
__global__ __launch_bounds__(256, 1) void k_score_topk_wide(
    const float* __restrict__ Ib, const float* __restrict__ Ub, const float* __restrict__ gb, int n_users, int Dp,
    const int64_t* __restrict__ u_idx, int n_rows, long long item_begin, int item_count, int tiles_per_split,
    const int64_t* __restrict__ excl_indptr, const int32_t* __restrict__ excl_indices, int k, int n_splits,
    unsigned long long* __restrict__ cand_ws, float* __restrict__ part_vals, int32_t* __restrict__ part_idx,
    int32_t* status, const void* __restrict__ Apl_, const void* __restrict__ Bpl_, int a_rows, int b_rows,
    uint32_t* __restrict__ gthr, int dbg, int pw, const uint32_t* __restrict__ amax) {
  // pw: entries per row of this split's partial list: k (one split: the final, sorted list) or HSK_SEL_KMAX (several:
  // k_fused_merge sorts the union anyway, so a row that ends with <= pw survivors is handed over as it is -- no select)
  // gthr [n_rows] (zero-initialised keys): the best threshold any split has reached for the row.  A split's k-th best score
  // so far is a lower bound of the row's k-th best over the whole catalogue, so every split may filter with the LARGEST
  // of them: the splits of a row block run side by side (same XCD, one per CU) and would each warm up a list of their own
  // -- 4 x k (1 + ln(n/4k)) appends per row instead of k (1 + ln(n/k)).  Published at the compactions (atomicMax), read at
  // the head of every tile past this CU's L1; a stale value only filters less.  Lists may then end with fewer than k
  // entries (padded with -inf, which the merge expects anyway); the merged top-k is exact whatever the thresholds were.
  extern __shared__ __attribute__((aligned(16))) __bf16 wlds[];
  constexpr int BM = GEMM_W_BM, BN = GEMM_W_BN, TM = 4, TN = 4;
  __bf16* As = wlds;
  __bf16* Bs = wlds + 2 * GEMM_W_A_STAGE;
  const __bf16* __restrict__ Apl = reinterpret_cast<const __bf16*>(Apl_);
  const __bf16* __restrict__ Bpl = reinterpret_cast<const __bf16*>(Bpl_);
  const _Float16* __restrict__ Ah = reinterpret_cast<const _Float16*>(Apl_);
  const _Float16* __restrict__ Bh = reinterpret_cast<const _Float16*>(Bpl_);
  unsigned char* hlds = reinterpret_cast<unsigned char*>(wlds);
  unsigned char* state = reinterpret_cast<unsigned char*>(wlds) + (H2 ? GEMM_H_LDS_BYTES : GEMM_W_LDS_BYTES);
  float cs = 1.f;
  if constexpr (H2) cs = ldexpf(1.f, -(hsk_h2_scale_exp(__uint_as_float(amax[0])) + hsk_h2_scale_exp(__uint_as_float(amax[1]))));
  float* thr = reinterpret_cast<float*>(state);               // [256] current k-th best score of the row
  int* cnt = reinterpret_cast<int*>(state + 1024);            // [256] candidates the row holds
  float* ubias = reinterpret_cast<float*>(state + 2048);      // [256]
  uint32_t* emask = reinterpret_cast<uint32_t*>(state + 3072);   // [256][8] exclusion bits of the current tile
  // during the epilogue / the compactions the operand stages are idle: queues and select scratch live there
  unsigned char* idle = reinterpret_cast<unsigned char*>(wlds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  hsk_f32x4* qv = reinterpret_cast<hsk_f32x4*>(idle) + wave * FGW_Q;                    // 4 x 16 KB
  int* qm = reinterpret_cast<int*>(idle + 4 * FGW_Q * 16) + wave * FGW_Q;               // 4 x 4 KB
  unsigned int* hist = reinterpret_cast<unsigned int*>(idle + 4 * FGW_Q * 20) + wave * 256;   // 4 x 1 KB
  unsigned long long* srt = reinterpret_cast<unsigned long long*>(idle + 4 * FGW_Q * 20 + 4 * 1024) + wave * HSK_SEL_KMAX;

  const int n_row_blocks = (n_rows + BM - 1) / BM;
  const int per_group = 8 * n_splits;
  const int grp = blockIdx.x / per_group, rem = blockIdx.x - grp * per_group;
  const int split = rem >> 3;
  const int rb = grp * 8 + (rem & 7);
  if (rb >= n_row_blocks) return;
  const int m0 = rb * BM;
  const int n_tiles = (item_count + BN - 1) / BN;
  const int t_lo = split * tiles_per_split, t_hi = min(n_tiles, t_lo + tiles_per_split);
  unsigned long long* __restrict__ cand = cand_ws + ((long long)rb * n_splits + split) * ((long long)BM * FGW_CAP);

  // thread = row: its user, bias, threshold, and where its exclusion row starts for this split
  long long e_p = 0, e_end = 0, e_next = 0x7fffffffffffffffll;
  {
    const int r = m0 + tid;
    int u = 0;
    if (r < n_rows) {
      long long uu = u_idx[r];
      if (uu < 0 || uu >= n_users) {
        if (status) atomicOr(status, HSK_STATUS_BAD_INDEX);
        uu = 0;
      }
      u = (int)uu;
    }
    ubias[tid] = Ub ? Ub[u] : 0.f;
    // rows past n_rows: nothing ever passes (and nothing ever lowers it); the others start from the seeded lower bound
    const uint32_t k0 = (gthr && r < n_rows) ? gthr[r] : 0u;   // (key 0: no bound yet)
    thr[tid] = r < n_rows ? (k0 ? fg_key2f(k0) : -INFINITY) : INFINITY;
    cnt[tid] = 0;
    if (excl_indptr && r < n_rows) {
      long long lo = excl_indptr[u];
      const long long hi = excl_indptr[u + 1];
      const long long first_col = item_begin + (long long)t_lo * BN;
      long long l = lo, hh = hi;
      while (l < hh) {
        const long long mid = (l + hh) >> 1;
        if (excl_indices[mid] < first_col)
          l = mid + 1;
        else
          hh = mid;
      }
      e_p = l;
      e_end = hi;
      if (e_p < e_end) e_next = excl_indices[e_p];
    }
  }
  const float gbv = gb ? gb[0] : 0.f;

  constexpr int KPL = FGW_CAP / 64;   // keys per lane of a compaction
  // MSB-first radix select of the k-th largest key, 8 bits per pass.  Two things keep it to ~2 passes instead of 8: the
  // digits every live key shares (the keys of one row are scores of about one magnitude: sign, exponent and a few mantissa
  // bits agree, and a pass over such a digit is 640 LDS atomics on ONE counter) are taken from the keys' maximum and
  // minimum without a pass, and the passes stop as soon as the selected bin is taken whole (its members differ only below
  // the digits already fixed: nothing left to decide).  The threshold is the smallest key kept -- exact either way.
  auto compact_row = [&](int rloc) {   // one wave, one row with n > k candidates: keep the best k, raise thr
    const int n = cnt[rloc];
    if (n <= k) return;
    unsigned long long kv[KPL];
    unsigned long long kmax = 0ull, kmin = ~0ull;
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
      const int j = lane + 64 * m;
      kv[m] = (j < n) ? cand[(long long)rloc * FGW_CAP + j] : 0ull;
      if (kv[m] != 0ull) {
        kmax = kv[m] > kmax ? kv[m] : kmax;
        kmin = kv[m] < kmin ? kv[m] : kmin;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long a = __shfl_xor(kmax, off, 64), b = __shfl_xor(kmin, off, 64);
      kmax = a > kmax ? a : kmax;
      kmin = b < kmin ? b : kmin;
    }
    const unsigned long long diff = kmax ^ kmin;   // (n > k >= 1 distinct keys: diff != 0)
    const int first = __builtin_clzll(diff | 1ull) >> 3;   // passes 0 .. first - 1: every live key has the maximum's digit
    unsigned long long pmask = first ? ~0ull << (64 - 8 * first) : 0ull;
    unsigned long long prefix = kmax & pmask;
    int need = k;
    for (int pass = first; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
#pragma unroll
      for (int q = 0; q < 4; ++q) hist[lane * 4 + q] = 0u;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int m = 0; m < KPL; ++m)
        if (kv[m] != 0ull && (kv[m] & pmask) == prefix) atomicAdd(&hist[(unsigned)(kv[m] >> shift) & 255u], 1u);
      __builtin_amdgcn_wave_barrier();
      const unsigned c0 = hist[lane * 4], c1 = hist[lane * 4 + 1], c2 = hist[lane * 4 + 2], c3 = hist[lane * 4 + 3];
      const int mine = (int)(c0 + c1 + c2 + c3);
      int above = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_down(above, off, 64);
        if (lane + off < 64) above += t;
      }
      above -= mine;
      int found_digit = -1, found_need = 0, found_cnt = 0;
      {
        int a3 = above, a2 = above + (int)c3, a1 = a2 + (int)c2, a0 = a1 + (int)c1;
        if (a3 < need && need <= a3 + (int)c3) { found_digit = lane * 4 + 3; found_need = need - a3; found_cnt = (int)c3; }
        else if (a2 < need && need <= a2 + (int)c2) { found_digit = lane * 4 + 2; found_need = need - a2; found_cnt = (int)c2; }
        else if (a1 < need && need <= a1 + (int)c1) { found_digit = lane * 4 + 1; found_need = need - a1; found_cnt = (int)c1; }
        else if (a0 < need && need <= a0 + (int)c0) { found_digit = lane * 4; found_need = need - a0; found_cnt = (int)c0; }
      }
      const unsigned long long who = __ballot(found_digit >= 0);
      const int src = __builtin_ctzll(who);
      const int digit = __shfl(found_digit, src, 64);
      need = __shfl(found_need, src, 64);
      const int whole = __shfl(found_cnt, src, 64);
      prefix |= (unsigned long long)digit << shift;
      pmask |= 0xffull << shift;
      if (need == whole) break;   // the bin is kept whole: keys >= prefix (zeros below) are exactly the k best
    }
    int base = 0;
    unsigned long long kth = ~0ull;
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
      const bool keep = kv[m] >= prefix && kv[m] != 0ull;
      const unsigned long long bm = __ballot(keep);
      if (keep) {
        cand[(long long)rloc * FGW_CAP + base + __popcll(bm & ((1ull << lane) - 1ull))] = kv[m];
        kth = kv[m] < kth ? kv[m] : kth;
      }
      base += __popcll(bm);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long b = __shfl_xor(kth, off, 64);
      kth = b < kth ? b : kth;
    }
    if (lane == 0) {
      cnt[rloc] = k;
      const uint32_t kth32 = (uint32_t)(kth >> 32);
      const float told = thr[rloc], tnew = fg_key2f(kth32);
      thr[rloc] = (tnew < told) ? told : tnew;           // (the shared threshold may already be above this list's k-th)
      if (gthr) atomicMax(&gthr[m0 + rloc], kth32);
    }
    __builtin_amdgcn_wave_barrier();
  };

  hsk_w_f32x16 acc[TM][TN];
  hsk_wide_stage stg;
  hsk_h2_stage hstg;
  if constexpr (!H2) hsk_wide_init(stg, tid);
  const int NT = H2 ? Dp / GEMM_H_BK : Dp / GEMM_W_BK;
  const long long a_step = (long long)a_rows * 48, b_step = (long long)b_rows * 48;
  const __bf16* a0 = Apl + (long long)m0 * 48;
  // first tile of the split: k-step 0 -> LDS stage 0, k-step 1 -> registers
  if constexpr (H2) {
    hsk_h2_load(hstg, Ah, Bh, a_rows, b_rows, m0, t_lo * BN, 0, tid);
    hsk_h2_store(hstg, hlds, tid);
    hsk_h2_load(hstg, Ah, Bh, a_rows, b_rows, m0, t_lo * BN, NT > 1 ? 1 : 0, tid);
  } else {
    const __bf16* b0 = Bpl + (long long)t_lo * BN * 48;
    hsk_wide_load(stg, a0, b0, tid);
    hsk_wide_store(stg, As, Bs);
    hsk_wide_load(stg, a0 + (NT > 1 ? a_step : 0), b0 + (NT > 1 ? b_step : 0), tid);
  }
  for (int tile = t_lo; tile < t_hi; ++tile) {
    const int n0 = tile * BN;
    const __bf16* b0 = Bpl + (long long)n0 * 48;
    // the row's shared threshold, if another split has got further (read past the L1: agent-scope load)
    if (gthr && m0 + tid < n_rows) {
      const uint32_t ks = __hip_atomic_load(&gthr[m0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float ts = ks ? fg_key2f(ks) : -INFINITY;
      if (thr[tid] < ts) thr[tid] = ts;
    }
    // exclusion bits of this tile (the row's sorted CSR is consumed as the tiles advance); read in the drains, behind
    // the k loop's barriers
    {
      uint32_t* em = emask + tid * 8;
#pragma unroll
      for (int w = 0; w < 8; ++w) em[w] = 0u;
      const long long col_lo = item_begin + n0, col_hi = col_lo + BN;
      while (e_next < col_hi) {
        const int c = (int)(e_next - col_lo);
        if (c >= 0) em[c >> 5] |= 1u << (c & 31);
        ++e_p;
        e_next = (e_p < e_end) ? (long long)excl_indices[e_p] : 0x7fffffffffffffffll;
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    __syncthreads();   // stage 0 of this tile is in place (stored behind the previous tile's compactions)
    const bool more = tile + 1 < t_hi;
    if constexpr (H2) {
      hsk_h2_kloop(acc, hstg, hlds, Ah, Bh, a_rows, b_rows, m0, n0, NT, tid, wm, wn, r32, h);
      if (more) hsk_h2_load(hstg, Ah, Bh, a_rows, b_rows, m0, n0 + BN, 0, tid);   // next tile's k-step 0: in flight during the epilogue
    } else {
      hsk_wide_kloop(acc, stg, As, Bs, a0, b0, a_step, b_step, NT, tid, wm, wn, r32, h);
      if (more) hsk_wide_load(stg, a0, b0 + (long long)BN * 48, tid);
    }

    // ---- epilogue: filter -> per-wave queue -> dense drain, band by band ---------------------------------------
    float ibv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 128 + j * 32 + r32;
      ibv[j] = (HAS_IB && col < item_count) ? Ib[item_begin + col] : 0.f;
    }
    // this lane's rows of band i: row_base + i * 32 + (q & 3) + 8 * (q >> 2) -- one base register, immediate offsets
    const int row_base = wm * 128 + 4 * h;
    const float* thr_b = thr + row_base;
    const float* ub_b = ubias + row_base;
    const int meta_base = row_base | ((wn * 128 + r32) << 16);
    int qn = 0;   // wave-uniform queue fill (carried across the bands: drained when it runs high, and at the tile's end)
    if (dbg & 1) {   // (timing experiments only: HSK_FUSED_DEBUG) keep the accumulators alive, select nothing
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) sacc += acc[i][j][0] + acc[i][j][15];
      if (sacc == 12345.678f) thr[tid] = sacc;
    } else
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float t[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] = thr_b[i * 32 + (q & 3) + 8 * (q >> 2)];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        hsk_f32x4 o;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float x = H2 ? acc[i][j][q] * cs : acc[i][j][q];
          if (EXTRA) {   // reference order: += u_bias, += i_bias, += global_bias
            if (Ub) x += ub_b[i * 32 + (q & 3) + 8 * (q >> 2)];
            if (HAS_IB) x += ibv[j];
            if (gb) x += gbv;
          } else if (HAS_IB) {
            x += ibv[j];
          }
          o[j] = x;
        }
        const bool pass = !(o[0] < t[q]) || !(o[1] < t[q]) || !(o[2] < t[q]) || !(o[3] < t[q]);   // NaN passes (torch.topk)
        const unsigned long long mask = (dbg & 4) ? (__ballot(pass) & 0ull) : __ballot(pass);   // (4: filter only)
        if (mask) {   // wave-uniform
          if (pass) {
            const int pos = qn + __popcll(mask & ((1ull << lane) - 1ull));
            qv[pos] = o;
            qm[pos] = meta_base + (i * 32 + (q & 3) + 8 * (q >> 2));
          }
          qn += __popcll(mask);
        }
      }
      __builtin_amdgcn_wave_barrier();   // LDS operations of one wave execute in order
      if (dbg & 8) qn = 0;   // (8: filter + enqueue, no drain)
      // one drain per tile in steady state (a few dozen entries); more often only while the thresholds are still low
      if (qn <= 512 && i != TM - 1) continue;
      for (int e0 = 0; e0 < qn; e0 += 64) {
        const int e = e0 + lane;
        if (e < qn) {
          const hsk_f32x4 o = qv[e];
          const int meta = qm[e];
          const int rloc = meta & 0xffff, cbase = meta >> 16;
          const float tr = thr[rloc];
          const uint32_t* em = emask + rloc * 8;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int cloc = cbase + j * 32;
            float x = o[j];
            if (!(x < tr) && n0 + cloc < item_count) {
              if ((em[cloc >> 5] >> (cloc & 31)) & 1u) x = -INFINITY;
              if (!(x < tr)) {
                const int slot = atomicAdd(&cnt[rloc], 1);
                cand[(long long)rloc * FGW_CAP + slot] =
                    ((unsigned long long)fg_f2key(x) << 32) | (uint32_t)(~(uint32_t)(item_begin + n0 + cloc));
              }
            }
          }
        }
      }
      qn = 0;
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();   // every append of this tile has its slot
    {   // rows that the next tile could overflow: compact now (wave w owns rows 64 w .. 64 w + 63, one per lane)
      const int fill = cnt[wave * 64 + lane];
      unsigned long long due = __ballot(fill > FGW_TRIG);
      while (due) {
        const int rr = __builtin_ctzll(due);
        due &= due - 1;
        compact_row(wave * 64 + rr);
      }
    }
    __syncthreads();   // queues and select scratch are dead: the stages may be refilled
    if (more) {
      if constexpr (H2) {
        hsk_h2_store(hstg, hlds, tid);
        hsk_h2_load(hstg, Ah, Bh, a_rows, b_rows, m0, n0 + BN, NT > 1 ? 1 : 0, tid);
      } else {
        const __bf16* b1 = b0 + (long long)BN * 48;
        hsk_wide_store(stg, As, Bs);
        hsk_wide_load(stg, a0 + (NT > 1 ? a_step : 0), b1 + (NT > 1 ? b_step : 0), tid);
      }
    }
  }

  // final: every row down to its k best, sorted, out to this split's slice of the partial lists
  __syncthreads();
  for (int rr = 0; rr < 64; ++rr) {
    const int rloc = wave * 64 + rr;
    const int row = m0 + rloc;
    if (row >= n_rows) break;
    if (cnt[rloc] > pw) compact_row(rloc);
    const int keep = min(cnt[rloc], pw);
    for (int j = lane; j < HSK_SEL_KMAX; j += 64) srt[j] = (j < keep) ? cand[(long long)rloc * FGW_CAP + j] : 0ull;
    if (n_splits == 1) fg_wave_bitonic_desc(srt, HSK_SEL_KMAX, lane);   // (several splits: k_fused_merge sorts the union)
    else __builtin_amdgcn_wave_barrier();
    const long long dst = ((long long)split * n_rows + row) * pw;
    for (int j = lane; j < pw; j += 64) {
      float v = -INFINITY;
      int32_t id = 0x7fffffff;   // pad of a split that holds fewer than k items: sorts behind everything real
      if (j < keep) {
        const unsigned long long c = srt[j];
        v = fg_key2f((uint32_t)(c >> 32));
        id = (int32_t)(~(uint32_t)c);
      }
      part_vals[dst + j] = v;
      part_idx[dst + j] = id;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// merge of the per-split lists: [n_parts, rows, k] -> [rows, k]  (same kernel contract as hsk_topk_merge)
// pw: entries per row of a partial list (>= k: the 256 x 256 kernel hands over up to HSK_SEL_KMAX unsorted survivors)
__global__ __launch_bounds__(256) void k_fused_merge(const float* __restrict__ vals, const int32_t* __restrict__ idx,
                                                     int n_parts, int rows, int k, int npad,
                                                     float* __restrict__ out_vals, int32_t* __restrict__ out_idx,
                                                     int pw = 0) {
  extern __shared__ unsigned long long merge_keys[];   // npad keys (a fixed 4096 held a CU to five rows at a time)
  const int r = blockIdx.x;
  if (pw <= 0) pw = k;
  const int total = n_parts * pw;
  for (int c = threadIdx.x; c < npad; c += 256) {
    unsigned long long v = 0ull;
    if (c < total) {
      const int p = c / pw, j = c - p * pw;
      const long long src = ((long long)p * rows + r) * pw + j;
      v = ((unsigned long long)fg_f2key(vals[src]) << 32) | (uint32_t)(~(uint32_t)idx[src]);
    }
    merge_keys[c] = v;
  }
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (npad >> 1); t += 256) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = merge_keys[lo], b = merge_keys[hi];
        if ((a < b) == desc) {
          merge_keys[lo] = b;
          merge_keys[hi] = a;
        }
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < k; c += 256) {
    const unsigned long long v = merge_keys[c];
    out_vals[(long long)r * k + c] = fg_key2f((uint32_t)(v >> 32));
    out_idx[(long long)r * k + c] = (int32_t)(~(uint32_t)v);
  }
}

// the pre-pass of MODE 2 / 3 (hsk_eval.hip, k_split_planes): pieces laid out [k-tile][row][piece][32]
void hsk_eval_split_planes(const float* src, const int64_t* idx, long long row0, long long n_src_rows, int n_valid,
                           int n_pad, int D, void* planes, hipStream_t stream);
int hsk_eval_seed_thresholds(const float* item_bias, const float* user_bias, const float* global_bias, int n_users, int Dp,
                             const int64_t* u_idx, int n_rows, long long item_begin, int sample_count,
                             const int64_t* excl_indptr, const int32_t* excl_indices, int k, const void* Apl,
                             const void* Bpl, int a_rows, int b_rows, const uint32_t* amax, float* scores_ws, float* vals_ws,
                             int32_t* idx_ws, uint32_t* gthr, int32_t* status, hipStream_t stream);   // hsk_eval.hip
// form 2: fp16 pairs [k-tile of 16][piece][row][16] at the scale of the rows' largest |x| (hsk_gemm_wide_h2.h)
void hsk_eval_split_planes_h2(const float* src, const int64_t* idx, long long row0, long long n_src_rows, int n_valid,
                              int n_pad, int D, uint32_t* amax, void* planes, hipStream_t stream);
// ... laid out [k-tile of 16][row][piece][16]: what the 256 x 256 core reads
void hsk_eval_split_planes16(const float* src, const int64_t* idx, long long row0, long long n_src_rows, int n_valid,
                             int n_pad, int D, void* planes, hipStream_t stream);

// bytes of the pre-split pieces of a call (MODE 2 / 3)
static int64_t hsk_fused_plane_bytes(int64_t n_rows, int64_t item_count, int64_t dim) {
  if (dim <= 0) return 0;
  const int64_t Dp = hsk_align_up(dim, FG_BK);
  // (rows padded to 256: what the 256 x 256 core reads; the 128 x 128 kernels need 128)
  const int64_t a_rows = hsk_align_up(n_rows, GEMM_W_BM), b_rows = hsk_align_up(item_count, GEMM_W_BN);
  return hsk_align_up(3 * a_rows * Dp * 2, 256) + hsk_align_up(3 * b_rows * Dp * 2, 256);
}

// number of item splits: enough workgroups to fill the chip (2 per CU), lists short enough for the merge
static int hsk_fused_splits(int64_t n_rows, int64_t item_count, int64_t k) {
  const int64_t row_blocks = hsk_ceil_div(n_rows, FG_BM), n_tiles = hsk_ceil_div(item_count, FG_BN);
  // two workgroups of this kernel fit a CU (256 VGPRs): 512 of them are one full round; 1024 (two rounds, twice as many
  // lists to warm up) measured 748 vs 804 k users/s at the lfm2b shape, 256: 573 k
  static const int target_wgs = getenv("HSK_FUSED_WGS") ? atoi(getenv("HSK_FUSED_WGS")) : 512;
  int64_t s = hsk_ceil_div(target_wgs, row_blocks);
  s = std::min<int64_t>(s, 4096 / std::max<int64_t>(k, 1));   // the merge sorts <= 4096 keys per row
  s = std::min<int64_t>(s, std::max<int64_t>(1, n_tiles / 8));   // a split of fewer than 8 tiles is all warm-up
  s = std::max<int64_t>(s, 1);
  if (s >= 8) s = s / 8 * 8;   // splits follow the XCDs: blockIdx.x % 8
  return (int)s;
}

// The 256 x 256 kernel: ONE workgroup per CU, so 256 of them are a full round.  Splits of the catalogue for it, or 0 when
// the call is not for it (too few tiles per split: a split's first tiles are all warm-up -- every score passes until the
// rows hold k candidates).  HSK_FUSED_WIDE: 0 never, 2 whenever the shape allows, 1 (default) by this rule.
// force: form 2 of the arithmetic -- the fp16-pair core only exists at 256 x 256, so every shape takes this kernel
static int hsk_fused_wide_splits(int64_t n_rows, int64_t item_count, int64_t k, bool force = false) {
  static const int wide_env = getenv("HSK_FUSED_WIDE") ? atoi(getenv("HSK_FUSED_WIDE")) : 1;
  const int wide_on = force ? 2 : wide_env;
  if (!wide_on) return 0;
  const int64_t row_blocks = hsk_ceil_div(n_rows, GEMM_W_BM), n_tiles = hsk_ceil_div(item_count, GEMM_W_BN);
  static const int target_wgs = getenv("HSK_FUSED_WIDE_WGS") ? atoi(getenv("HSK_FUSED_WIDE_WGS")) : 256;
  int64_t s = std::max<int64_t>(1, target_wgs / row_blocks);
  s = std::min<int64_t>(s, 4096 / HSK_SEL_KMAX);   // the merge sorts <= 4096 keys per row, HSK_SEL_KMAX per split
  const int64_t min_tiles = wide_on == 2 ? 2 : 16;
  s = std::min<int64_t>(s, std::max<int64_t>(1, n_tiles / min_tiles));
  if (s >= 8) s = s / 8 * 8;
  if (wide_on != 2 && (n_tiles / s < min_tiles || row_blocks * s < 128)) return 0;   // (less than half a round)
  return (int)s;
}

// columns of the threshold-seeding sample (hsk_eval_seed_thresholds): 0 = no seeding
static int64_t hsk_fused_seed_cols(int64_t item_count, int64_t k) {
  // (lfm2b shape, one box: no seeding 1.09, 2048 columns 1.11, 4096 columns 1.155 M users/s)
  static const int seed_on = getenv("HSK_FUSED_SEED") ? atoi(getenv("HSK_FUSED_SEED")) : 4096;
  if (seed_on <= 0 || item_count < 4 * (int64_t)seed_on || k > seed_on / 4) return 0;
  return hsk_align_up(seed_on, GEMM_W_BN);
}

// candidate slabs + partial lists of one layout
static int64_t hsk_fused_sel_bytes(int64_t n_rows, int64_t k, int64_t s, int64_t bm, int64_t cap, int64_t seed_cols = 0) {
  const int64_t row_blocks = hsk_ceil_div(n_rows, bm);
  if (bm == GEMM_W_BM && s > 1) k = std::max<int64_t>(k, HSK_SEL_KMAX);   // (partial lists of the 256 x 256 kernel: pw entries)
  return hsk_align_up(row_blocks * s * bm * cap * 8, 256) + hsk_align_up(s * n_rows * k * 8, 256) +
         hsk_align_up(row_blocks * bm * 4, 256) +          // slabs, partial lists, shared thresholds
         (seed_cols ? hsk_align_up(n_rows * seed_cols * 4, 256) + hsk_align_up(n_rows * k * 8, 256) : 0) + 256;
}

extern "C" int64_t hsk_mf_eval_fused_ws_bytes(int64_t n_rows, int64_t item_count, int64_t k) {
  if (n_rows <= 0 || item_count <= 0 || k <= 0 || k > HSK_SEL_KMAX) return -1;
  // (the larger of the two kernels' layouts: which one runs also depends on whether the pieces' scratch is there)
  const int64_t narrow = hsk_fused_sel_bytes(n_rows, k, hsk_fused_splits(n_rows, item_count, k), FG_BM, HSK_SEL_CAP);
  // (... and, since the arithmetic can be switched between sizing and running a call, on both rules for the 256 x 256 one)
  int64_t wide = 0;
  for (int force = 0; force < 2; ++force) {
    const int sw = hsk_fused_wide_splits(n_rows, item_count, k, force != 0);
    if (sw) wide = std::max(wide, hsk_fused_sel_bytes(n_rows, k, sw, GEMM_W_BM, FGW_CAP, hsk_fused_seed_cols(item_count, k)));
  }
  return std::max(narrow, wide);
}

extern "C" int64_t hsk_mf_eval_fused_ws_bytes_dim(int64_t n_rows, int64_t item_count, int64_t k, int64_t dim) {
  const int64_t base = hsk_mf_eval_fused_ws_bytes(n_rows, item_count, k);
  return base < 0 ? base : base + hsk_fused_plane_bytes(n_rows, item_count, dim);
}

extern "C" int hsk_mf_eval_topk_fused(const float* user_emb, const float* item_emb, const float* item_bias,
                                      const float* user_bias, const float* global_bias, int64_t n_users,
                                      int64_t n_items, int64_t dim, const int64_t* u_idx, int64_t n_rows,
                                      int64_t item_begin, int64_t item_count, const int64_t* excl_indptr,
                                      const int32_t* excl_indices, int64_t k, void* ws, int64_t ws_bytes,
                                      float* out_vals, int32_t* out_idx, int32_t* status, hsk_stream_t stream_) {
  HSK_REQUIRE(user_emb && item_emb && u_idx && ws && out_vals && out_idx, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_users > 0 && n_items > 0 && dim > 0, HSK_ERR_INVALID, "bad table shape");
  HSK_REQUIRE(item_begin >= 0 && item_count > 0 && item_begin + item_count <= n_items, HSK_ERR_INVALID,
              "item shard [%lld, +%lld) outside [0, %lld)", (long long)item_begin, (long long)item_count,
              (long long)n_items);
  HSK_REQUIRE(n_rows >= 0 && n_items < 0x7fffffff && n_rows < 0x7fffffff, HSK_ERR_INVALID, "bad n_rows");
  HSK_REQUIRE((excl_indptr == nullptr) == (excl_indices == nullptr), HSK_ERR_INVALID,
              "exclude CSR needs both indptr and indices");
  HSK_REQUIRE(k >= 1 && k <= HSK_SEL_KMAX, HSK_ERR_UNSUPPORTED, "k %lld outside [1, %d]", (long long)k, HSK_SEL_KMAX);
  HSK_REQUIRE(k <= item_count, HSK_ERR_INVALID, "k %lld > item_count %lld", (long long)k, (long long)item_count);
  if (n_rows == 0) return HSK_OK;
  const int64_t need = hsk_mf_eval_fused_ws_bytes(n_rows, item_count, k);
  HSK_REQUIRE(ws_bytes >= need && ((uintptr_t)ws & 255) == 0, HSK_ERR_INVALID,
              "workspace: %lld bytes (256-byte aligned) needed, %lld given", (long long)need, (long long)ws_bytes);
  hipStream_t stream = (hipStream_t)stream_;
  {
    // the 256 x 256 / one-wave-per-SIMD kernel: needs the operands' pieces (k-tiles of 16) and enough tiles per split
    static const int planes_on_w = getenv("HSK_EVAL_PLANES") ? atoi(getenv("HSK_EVAL_PLANES")) : 1;
    const bool h2 = hsk_eval_x3() == 2;
    const int SW = hsk_fused_wide_splits(n_rows, item_count, k, h2);
    const bool vec4w = (dim % 4 == 0) && ((((uintptr_t)user_emb | (uintptr_t)item_emb) & 15) == 0);
    if (SW && hsk_eval_x3() && planes_on_w && planes_on_w != 3 && vec4w &&
        ws_bytes >= need + hsk_fused_plane_bytes(n_rows, item_count, dim)) {
      const int64_t row_blocks = hsk_ceil_div(n_rows, GEMM_W_BM), n_tiles = hsk_ceil_div(item_count, GEMM_W_BN);
      const int tiles_per_split = (int)hsk_ceil_div(n_tiles, SW);
      char* p = (char*)ws;
      unsigned long long* slab = (unsigned long long*)p;
      p += hsk_align_up(row_blocks * SW * GEMM_W_BM * FGW_CAP * 8, 256);
      const int pw = SW > 1 ? HSK_SEL_KMAX : (int)k;   // entries per row of a partial list
      float* part_vals = (float*)p;
      int32_t* part_idx = (int32_t*)(p + (int64_t)SW * n_rows * pw * 4);
      p += hsk_align_up((int64_t)SW * n_rows * pw * 8, 256);
      uint32_t* gthr = (uint32_t*)p;   // shared / seeded thresholds (ordered keys; 0 = below everything)
      p += hsk_align_up(row_blocks * GEMM_W_BM * 4, 256);
      const int64_t seed_cols = hsk_fused_seed_cols(item_count, k);
      float* seed_scores = (float*)p;
      float* seed_vals = (float*)(p + hsk_align_up(n_rows * seed_cols * 4, 256));
      int32_t* seed_idx = (int32_t*)(seed_vals + n_rows * k);
      HSK_HIP(hipMemsetAsync(gthr, 0, (size_t)row_blocks * GEMM_W_BM * 4, stream));
      const int Dp = (int)hsk_align_up(dim, FG_BK);
      const int a_rows = (int)(row_blocks * GEMM_W_BM), b_rows = (int)(n_tiles * GEMM_W_BN);
      void* Apl = (char*)ws + need;
      void* Bpl = (char*)Apl + hsk_align_up(3 * (int64_t)a_rows * Dp * 2, 256);
      // (form 2: fp16 pairs fill 4 of the regions' 6 bytes per element; the two scale words sit in the last 256 bytes)
      uint32_t* amax = h2 ? (uint32_t*)((char*)Apl + hsk_fused_plane_bytes(n_rows, item_count, dim) - 256) : nullptr;
      if (h2) {
        HSK_HIP(hipMemsetAsync(amax, 0, 8, stream));
        hsk_eval_split_planes_h2(user_emb, u_idx, 0, n_users, (int)n_rows, a_rows, (int)dim, amax, Apl, stream);
        hsk_eval_split_planes_h2(item_emb, nullptr, item_begin, n_items, (int)item_count, b_rows, (int)dim, amax + 1, Bpl,
                                 stream);
      } else {
        hsk_eval_split_planes16(user_emb, u_idx, 0, n_users, (int)n_rows, a_rows, (int)dim, Apl, stream);
        hsk_eval_split_planes16(item_emb, nullptr, item_begin, n_items, (int)item_count, b_rows, (int)dim, Bpl, stream);
      }
      HSK_LAUNCH_CHECK();
      if (seed_cols) {
        int src = hsk_eval_seed_thresholds(item_bias, user_bias, global_bias, (int)n_users, Dp, u_idx, (int)n_rows,
                                           (long long)item_begin, (int)seed_cols, excl_indptr, excl_indices, (int)k, Apl, Bpl,
                                           a_rows, b_rows, amax, seed_scores, seed_vals, seed_idx, gthr, status, stream);
        if (src) return src;
      }
      static bool lds_set[64] = {};   // per device: the opt-in for > 64 KB of dynamic LDS is a per-device attribute
      int dev = 0;
      HSK_HIP(hipGetDevice(&dev));
      if (dev >= 0 && dev < 64 && !lds_set[dev]) {
#define HSK_TOPK_WIDE_LDS(IB, EX)                                                                                                     \
  HSK_HIP(hipFuncSetAttribute((const void*)k_score_topk_wide<IB, EX, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FGW_LDS_BYTES)); \
  HSK_HIP(hipFuncSetAttribute((const void*)k_score_topk_wide<IB, EX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FGW_H2_LDS_BYTES))
        HSK_TOPK_WIDE_LDS(true, false);
        HSK_TOPK_WIDE_LDS(true, true);
        HSK_TOPK_WIDE_LDS(false, false);
        HSK_TOPK_WIDE_LDS(false, true);
#undef HSK_TOPK_WIDE_LDS
        lds_set[dev] = true;
      }
      const unsigned grid = (unsigned)(hsk_ceil_div(row_blocks, 8) * 8 * SW);
#define HSK_TOPK_WIDE_K(IB, EX, H)                                                                                     \
  k_score_topk_wide<IB, EX, H><<<grid, 256, H ? FGW_H2_LDS_BYTES : FGW_LDS_BYTES, stream>>>(                            \
      item_bias, user_bias, global_bias, (int)n_users, Dp, u_idx, (int)n_rows, (long long)item_begin, (int)item_count, \
      tiles_per_split, excl_indptr, excl_indices, (int)k, SW, slab, SW == 1 ? out_vals : part_vals,                    \
      SW == 1 ? out_idx : part_idx, status, Apl, Bpl, a_rows, b_rows, gthr, dbg, pw, amax)
#define HSK_TOPK_WIDE(IB, EX) do { if (h2) HSK_TOPK_WIDE_K(IB, EX, true); else HSK_TOPK_WIDE_K(IB, EX, false); } while (0)
      static const int dbg = getenv("HSK_FUSED_DEBUG") ? atoi(getenv("HSK_FUSED_DEBUG")) : 0;   // timing experiments
      const bool extra = user_bias || global_bias;
      if (item_bias) { if (extra) HSK_TOPK_WIDE(true, true); else HSK_TOPK_WIDE(true, false); }
      else           { if (extra) HSK_TOPK_WIDE(false, true); else HSK_TOPK_WIDE(false, false); }
#undef HSK_TOPK_WIDE_K
#undef HSK_TOPK_WIDE
      HSK_LAUNCH_CHECK();
      if (SW > 1) {
        int npad = 1;
        while (npad < SW * pw) npad <<= 1;
        k_fused_merge<<<(unsigned)n_rows, 256, (size_t)npad * 8, stream>>>(part_vals, part_idx, SW, (int)n_rows, (int)k, npad,
                                                                           out_vals, out_idx, pw);
        HSK_LAUNCH_CHECK();
      }
      return HSK_OK;
    }
  }
  const int S = hsk_fused_splits(n_rows, item_count, k);
  const int64_t row_blocks = hsk_ceil_div(n_rows, FG_BM), n_tiles = hsk_ceil_div(item_count, FG_BN);
  const int tiles_per_split = (int)hsk_ceil_div(n_tiles, S);
  char* p = (char*)ws;
  unsigned long long* slab = (unsigned long long*)p;
  p += hsk_align_up(row_blocks * S * FG_BM * HSK_SEL_CAP * 8, 256);
  float* part_vals = (float*)p;
  int32_t* part_idx = (int32_t*)(p + (int64_t)S * n_rows * k * 4);
  const unsigned grid = (unsigned)(hsk_ceil_div(row_blocks, 8) * 8 * S);
  const bool vec4 = (dim % 4 == 0) && ((((uintptr_t)user_emb | (uintptr_t)item_emb) & 15) == 0);
  float* pv = S == 1 ? out_vals : part_vals;
  int32_t* pi = S == 1 ? out_idx : part_idx;
  // A workspace of hsk_mf_eval_fused_ws_bytes_dim bytes also holds the operands' bf16 pieces: split once, here, instead of
  // once per tile in every workgroup's k loop.  HSK_EVAL_PLANES: 0 off, 2 both operands, 3 items only, 1 (default) by rule.
  static const int planes_on = getenv("HSK_EVAL_PLANES") ? atoi(getenv("HSK_EVAL_PLANES")) : 1;
  const int64_t plane_bytes = hsk_fused_plane_bytes(n_rows, item_count, dim);
  const bool planes = hsk_eval_x3() && planes_on && vec4 && ws_bytes >= need + plane_bytes;
#define HSK_SCORE_TOPK(V4, MODE, ...)                                                                                   \
  k_score_topk<V4, MODE><<<grid, 256, 0, stream>>>(user_emb, item_emb, item_bias, user_bias, global_bias, (int)n_users,  \
                                                   (int)dim, u_idx, (int)n_rows, (long long)item_begin, (int)item_count, \
                                                   tiles_per_split, excl_indptr, excl_indices, (int)k, S, slab, pv, pi,   \
                                                   status, ##__VA_ARGS__)
  if (planes) {
    const int Dp = (int)hsk_align_up(dim, FG_BK);
    const int64_t a_rows = hsk_align_up(n_rows, GEMM_W_BM), b_rows = hsk_align_up(item_count, GEMM_W_BN);   // (as sized)
    __bf16* Apl = (__bf16*)((char*)ws + need);
    __bf16* Bpl = (__bf16*)((char*)Apl + hsk_align_up(3 * a_rows * Dp * 2, 256));
    const long long a_stride = a_rows * 96, b_stride = b_rows * 96;   // elements per k-tile
    const bool a_planes = planes_on != 3;
    if (a_planes)
      hsk_eval_split_planes(user_emb, u_idx, 0, n_users, (int)n_rows, (int)a_rows, (int)dim, Apl, stream);
    hsk_eval_split_planes(item_emb, nullptr, item_begin, n_items, (int)item_count, (int)b_rows, (int)dim, Bpl, stream);
    HSK_LAUNCH_CHECK();
    if (a_planes) HSK_SCORE_TOPK(true, 2, Apl, Bpl, a_stride, b_stride);
    else HSK_SCORE_TOPK(true, 3, Apl, Bpl, a_stride, b_stride);
  } else if (hsk_eval_x3()) {
    if (vec4) HSK_SCORE_TOPK(true, 1); else HSK_SCORE_TOPK(false, 1);
  } else {
    if (vec4) HSK_SCORE_TOPK(true, 0); else HSK_SCORE_TOPK(false, 0);
  }
#undef HSK_SCORE_TOPK
  HSK_LAUNCH_CHECK();
  if (S > 1) {
    int npad = 1;
    while (npad < S * (int)k) npad <<= 1;
    k_fused_merge<<<(unsigned)n_rows, 256, (size_t)npad * 8, stream>>>(part_vals, part_idx, S, (int)n_rows, (int)k, npad,
                                                                       out_vals, out_idx);
    HSK_LAUNCH_CHECK();
  }
  return HSK_OK;
}
