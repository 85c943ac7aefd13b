// Common device/host helpers for the DRN-WSOD gfx950 kernels (CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRN_OK 0
#define DRN_ERR_ARG (-1)
#define DRN_ERR_LAUNCH (-2)
#define DRN_ERR_UNSUPPORTED (-3)

#define DRN_F32 0
#define DRN_BF16 1
#define DRN_FP8 2  /* OCP e4m3fn, one byte per element (gfx950's native fp8; NOT MI300X's fnuz); the conv trunk only */

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef unsigned short bf16_t;  // storage type

#define DRN_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return DRN_ERR_LAUNCH;           \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __builtin_bit_cast(float, (uint32_t)v << 16);
}
// round-to-nearest-even, NaN preserved (same as torch's float->bfloat16): gfx950's v_cvt_pk_bf16_f32
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}

// fp8 e4m3fn <-> f32.  v_cvt_pk_fp8_f32 rounds to nearest even; the clamp makes the conversion saturating (|x| > 448 would
// otherwise become NaN), like the quantisers the oracle emulates (x.clamp(-448, 448).to(torch.float8_e4m3fn)).
__device__ __forceinline__ float fp8_to_f32(uint8_t v) { return __builtin_amdgcn_cvt_f32_fp8((int)v, 0); }
__device__ __forceinline__ uint8_t f32_to_fp8(float f) {
  f = fminf(fmaxf(f, -448.f), 448.f);
  return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(f, 0.f, 0, false) & 0xff);
}

// Counter-based dropout multipliers (F.dropout's Bernoulli mask of DiscriminativeAdaptionNeck.forward,
// projects/WSL/wsl/modeling/roi_heads/box_head.py:89-91; the backward reads the mask off the saved output, so only the forward
// draws).  32-bit hashes (murmur3 finaliser) of the element index, shared by neighbouring elements:
//   p == 0.5 (box_head.py:90 hard-codes it): ONE hash per 32 consecutive indices, bit (idx & 31) decides element idx - a lane of a
//     GEMM epilogue that owns 16 elements of a 32-column block draws one hash for all of them;
//   any other p: a 24-bit uniform per element from a hash of its own.
// (Rounds 1-5 hashed every element with two 64-bit multiplies: ~35 VALU instructions per element - 12 us of a GEMM epilogue
// that owns 64 elements per lane.)
struct DrnDropRule { uint32_t s0; int half; float p, scale; };
__device__ __forceinline__ uint32_t drn_fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ DrnDropRule drn_drop_rule(unsigned long long seed, float p) {
  DrnDropRule r;
  r.s0 = drn_fmix32((uint32_t)seed ^ drn_fmix32((uint32_t)(seed >> 32) + 0x9E3779B9u));
  r.half = p == 0.5f;
  r.p = p;
  r.scale = 1.f / (1.f - p);
  return r;
}
__device__ __forceinline__ uint32_t drn_drop_hash(const DrnDropRule& r, unsigned long long key) {
  return drn_fmix32((uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x7FEB352Du + r.s0);
}
// p == 0.5: the 32 keep bits of the indices 32 g .. 32 g + 31
__device__ __forceinline__ uint32_t drn_drop_bits32(const DrnDropRule& r, unsigned long long g) { return drn_drop_hash(r, g); }
__device__ __forceinline__ float drn_drop_mult(const DrnDropRule& r, unsigned long long idx) {
  if (r.half) return ((drn_drop_bits32(r, idx >> 5) >> (int)(idx & 31)) & 1u) ? r.scale : 0.f;
  const float u = (drn_drop_hash(r, idx ^ 0x5bd1e99500000000ULL) >> 8) * (1.0f / 16777216.0f);
  return u < r.p ? 0.f : r.scale;
}
// multipliers of the four elements idx .. idx + 3 (idx % 4 == 0: they share a 32-index group)
__device__ __forceinline__ void drn_drop_mult4(const DrnDropRule& r, unsigned long long idx, float (&m)[4]) {
  if (r.half) {
    const uint32_t h = drn_drop_bits32(r, idx >> 5) >> (int)(idx & 31);
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = ((h >> e) & 1u) ? r.scale : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = drn_drop_mult(r, idx + e);
  }
}

// one row of the optimizer's segment table (device memory, refreshed in place when the LR schedule moves)
struct SgdSeg { long off; long cnt; float lr; float wd; };

template <int DT> struct ElemOf;
template <> struct ElemOf<DRN_F32> {
  using type = float;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemOf<DRN_BF16> {
  using type = bf16_t;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct ElemOf<DRN_FP8> {
  using type = uint8_t;
  static __device__ __forceinline__ float ld(const uint8_t* p) { return fp8_to_f32(*p); }
  static __device__ __forceinline__ void st(uint8_t* p, float v) { *p = f32_to_fp8(v); }
};

static inline int drn_esize(int dtype) { return dtype == DRN_FP8 ? 1 : dtype == DRN_BF16 ? 2 : 4; }
template <int DT> struct EsOf { static constexpr int value = DT == DRN_FP8 ? 1 : DT == DRN_BF16 ? 2 : 4; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a linear workgroup id: the dispatcher places block b on XCD b % 8;
// give every XCD a contiguous chunk of the logical id space (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
