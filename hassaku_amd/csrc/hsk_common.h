// hsk_common.h -- shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/hassaku_hip.h"

#define HSK_WAVE 64

// ---------------------------------------------------------------------------------------------
// host-side error plumbing
// ---------------------------------------------------------------------------------------------
void hsk_set_error(const char* fmt, ...);

#define HSK_REQUIRE(cond, code, ...)   \
  do {                                 \
    if (!(cond)) {                     \
      hsk_set_error(__VA_ARGS__);      \
      return (code);                   \
    }                                  \
  } while (0)

#define HSK_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      hsk_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return HSK_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define HSK_LAUNCH_CHECK()                                                                 \
  do {                                                                                     \
    hipError_t e_ = hipGetLastError();                                                     \
    if (e_ != hipSuccess) {                                                                \
      hsk_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return HSK_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

static inline int64_t hsk_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t hsk_align_up(int64_t a, int64_t b) { return hsk_ceil_div(a, b) * b; }

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ int hsk_lane() { return (int)(threadIdx.x & (HSK_WAVE - 1)); }

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float hsk_dpp_add(float v) {
  // v + (DPP-permuted v); rows disabled by ROW_MASK contribute +0
  int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(moved);
}

// Sum over the 64 lanes of a wave; the result is wave-uniform (read from lane 63).
__device__ __forceinline__ float hsk_wave_sum(float v) {
  v = hsk_dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
  v = hsk_dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
  v = hsk_dpp_add<0x141>(v);       // row_half_mirror
  v = hsk_dpp_add<0x140>(v);       // row_mirror     -> every lane holds its 16-lane row sum
  v = hsk_dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
  v = hsk_dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ double hsk_wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, HSK_WAVE);
  return v;
}

__device__ __forceinline__ int hsk_wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, HSK_WAVE);
  return v;
}

__device__ __forceinline__ float hsk_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int hsk_readlane_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}
__device__ __forceinline__ int hsk_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- small fixed-width float vectors (1, 2 or 4 floats = one global_load_dword/x2/x4) ----------
template <int V> struct hsk_vec;
template <> struct hsk_vec<1> { float v[1]; };
template <> struct __attribute__((aligned(8))) hsk_vec<2> { float v[2]; };
template <> struct __attribute__((aligned(16))) hsk_vec<4> { float v[4]; };

template <int V>
__device__ __forceinline__ hsk_vec<V> hsk_ldg(const float* p) {
  return *reinterpret_cast<const hsk_vec<V>*>(p);
}
template <int V>
__device__ __forceinline__ void hsk_stg(float* p, const hsk_vec<V>& x) {
  *reinterpret_cast<hsk_vec<V>*>(p) = x;
}
// non-temporal store: a stream that nobody re-reads soon (AdamW moments of rows brought up to date) should not sit
// dirty in the Infinity Cache and drain into HBM under the next kernel's gathers
template <int V>
__device__ __forceinline__ void hsk_stg_nt(float* p, const hsk_vec<V>& x) {
  typedef float vf __attribute__((ext_vector_type(V)));
  vf t;
#pragma unroll
  for (int i = 0; i < V; ++i) t[i] = x.v[i];
  __builtin_nontemporal_store(t, reinterpret_cast<vf*>(p));
}
template <>
__device__ __forceinline__ void hsk_stg_nt<1>(float* p, const hsk_vec<1>& x) {
  __builtin_nontemporal_store(x.v[0], p);
}
template <int V>
__device__ __forceinline__ hsk_vec<V> hsk_zero() {
  hsk_vec<V> r;
#pragma unroll
  for (int i = 0; i < V; ++i) r.v[i] = 0.f;
  return r;
}

// ---- Philox4x32-10 counter RNG ---------------------------------------------------------------
struct hsk_u32x4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ hsk_u32x4 hsk_philox4x32_10(hsk_u32x4 ctr, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * ctr.x;
    uint64_t p1 = (uint64_t)M1 * ctr.z;
    hsk_u32x4 n;
    n.x = (uint32_t)(p1 >> 32) ^ ctr.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ ctr.w ^ k1;
    n.w = (uint32_t)p0;
    ctr = n;
    k0 += W0;
    k1 += W1;
  }
  return ctr;
}

// ---- optimiser scalars (host-computed exactly like torch.optim's single-tensor paths) ---------
// One struct serves the three optimisers train/trainer.py:48-53 of the reference selects by conf['optimizer']:
//   adamw    decoupled decay   p *= 1 - lr*wd;          then Adam on g
//   adam     L2 decay          g += wd * p;             then Adam on g           (decay = 1)
//   adagrad  L2 decay          g += wd * p;  v += g*g;  p -= lr * g / (sqrt(v) + eps)   (lr_decay = 0; m unused)
#define HSK_OPT_KIND_ADAM_FAMILY 0
#define HSK_OPT_KIND_ADAGRAD 1
struct hsk_adamw_consts {
  float decay;      // 1 - lr*wd (adamw) or 1
  float l2;         // wd (adam, adagrad) or 0
  int kind;         // HSK_OPT_KIND_*
  float w1;         // 1 - beta1   (lerp weight)
  float beta2;      // beta2
  float one_m_b2;   // 1 - beta2
  float step_size;  // lr / (1 - beta1^t)
  float bc2_sqrt;   // sqrt(1 - beta2^t)
  float rbc2_sqrt;  // 1 / sqrt(1 - beta2^t), correctly rounded from double
  float eps;
};

// One AdamW element update, torch's order of operations (see hsk_adamw_dense in the header).
// Default build: sqrt and the two divisions use the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 (the division by
// bc2_sqrt becomes a multiplication by its correctly-rounded reciprocal): <= 3 ulp on the update term, i.e.
// <= 2e-7 * lr on the parameter -- two orders below the 1e-5 parity bound -- and ~3.5x fewer VALU cycles,
// which is what the lazy zero-gradient replay is bound by.  -DHSK_ADAM_IEEE=1 builds the correctly-rounded
// form (IEEE sqrtf and '/'), bit-for-bit torch's arithmetic given the same gradient.
#ifndef HSK_ADAM_IEEE
#define HSK_ADAM_IEEE 0
#endif

// GEN = false: compile-time AdamW (no L2 term, no optimiser branch) for the kernels whose time is this function
// (the zero-gradient replay is VALU-bound: the generic form costs it +20 %); GEN = true serves all three optimisers.
template <bool GEN = true>
__device__ __forceinline__ void hsk_adamw_update(float& p, float& m, float& v, float g,
                                                 const hsk_adamw_consts& c) {
  if (GEN) {
    g = fmaf(c.l2, p, g);   // grad.add(param, alpha=weight_decay); exact no-op when l2 == 0
    if (c.kind == HSK_OPT_KIND_ADAGRAD) {   // uniform branch
      v = fmaf(g, g, v);    // state_sum.addcmul_(grad, grad)
#if HSK_ADAM_IEEE
      p = p - c.step_size * (g / (sqrtf(v) + c.eps));
#else
      p = fmaf(-c.step_size * g, __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + c.eps), p);
#endif
      return;
    }
  }
  p = p * c.decay;
  m = fmaf(c.w1, g - m, m);
  v = fmaf(c.one_m_b2 * g, g, v * c.beta2);
#if HSK_ADAM_IEEE
  float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p - c.step_size * (m / denom);
#else
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), c.rbc2_sqrt, c.eps);
  p = fmaf(-c.step_size * m, __builtin_amdgcn_rcpf(denom), p);
#endif
}


// The same update for g == 0 (the lazy replay): two operations fewer, bit-identical to hsk_adamw_update(..., 0.f, c)
// (0 - m and -m differ only in the sign of a zero that fma(w1, ., m) absorbs; (1-b2)*0*0 + v*b2 == v*b2 for v >= 0).
template <bool GEN = true>
__device__ __forceinline__ void hsk_adamw_replay(float& p, float& m, float& v, const hsk_adamw_consts& c) {
  if (GEN) {
    hsk_adamw_update<true>(p, m, v, 0.f, c);
    return;
  }
  p = p * c.decay;
  m = fmaf(c.w1, -m, m);
  v = v * c.beta2;
#if HSK_ADAM_IEEE
  float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p - c.step_size * (m / denom);
#else
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), c.rbc2_sqrt, c.eps);
  p = fmaf(-c.step_size * m, __builtin_amdgcn_rcpf(denom), p);
#endif
}

#endif  // __HIPCC__

#ifdef __cplusplus
#include <cmath>
// opt: HSK_OPT_ADAMW / HSK_OPT_ADAM / HSK_OPT_ADAGRAD of hassaku_hip.h (0 / 1 / 2)
static inline hsk_adamw_consts hsk_make_adamw_consts(double lr, double b1, double b2, double eps,
                                                     double wd, int64_t step, int opt = 0) {
  hsk_adamw_consts c;
  if (opt == 2) {  // adagrad: clr = lr / (1 + (step-1)*lr_decay) with lr_decay = 0
    c.decay = 1.f;
    c.l2 = (float)wd;
    c.kind = HSK_OPT_KIND_ADAGRAD;
    c.w1 = 0.f;
    c.beta2 = 1.f;
    c.one_m_b2 = 0.f;
    c.step_size = (float)lr;
    c.bc2_sqrt = 1.f;
    c.rbc2_sqrt = 1.f;
    c.eps = (float)eps;
    return c;
  }
  double bc1 = 1.0 - std::pow(b1, (double)step);
  double bc2 = 1.0 - std::pow(b2, (double)step);
  c.kind = HSK_OPT_KIND_ADAM_FAMILY;
  c.l2 = (opt == 1) ? (float)wd : 0.f;
  c.decay = (opt == 1) ? 1.f : (float)(1.0 - lr * wd);
  c.w1 = (float)(1.0 - b1);
  c.beta2 = (float)b2;
  c.one_m_b2 = (float)(1.0 - b2);
  c.step_size = (float)(lr / bc1);
  c.bc2_sqrt = (float)std::sqrt(bc2);
  c.rbc2_sqrt = (float)(1.0 / std::sqrt(bc2));
  c.eps = (float)eps;
  return c;
}
#endif
