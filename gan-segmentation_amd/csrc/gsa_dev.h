// Device-side helpers shared by the translation units of the HIP library (gsa_kernels.hip, gsa_wino_lean.hip): vector types, the
// fixed-point statistics conversions (part of the canonical arithmetic: ONE definition), quad DPP, packed fp32 operations with their
// hand-settled MFMA hazards, the XCD-aware block order.  Internal; included after gsa_kernels.h inside HIP sources only.
#pragma once
#include "gsa_kernels.h"

#ifndef GSA_NO_XCD_REMAP
#define GSA_NO_XCD_REMAP 0
#endif

namespace gsa {

// Diagnostic build only (make stamp): s_memtime stamps at the phase boundaries of the conv
// kernel, summed per phase into ConvParams::stamps.  No stamp executes in the product build.
#ifdef GSA_STAMP
#define GSA_DBG_HOOKS 1
#endif
#ifdef GSA_STAMP
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t[i]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_DECL unsigned long long stamp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP_FLUSH(nph) do { if (p.stamps && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < (nph); ++i_) atomicAdd(&p.stamps[i_], stamp_t[i_ + 1] - stamp_t[i_]); atomicAdd(&p.stamps[15], 1ull); } } while (0)
#define TICK(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TSUM(acc, a, b) acc += (b) - (a)
#define TFLUSH(idx, acc) do { if (p.stamps && (threadIdx.x & 63) == 0) atomicAdd(&p.stamps[idx], acc); } while (0)
#else
#define TICK(var) do {} while (0)
#define TSUM(acc, a, b) do {} while (0)
#define TFLUSH(idx, acc) do {} while (0)
#define STAMP(i) do {} while (0)
#define STAMP_DECL do {} while (0)
#define STAMP_FLUSH(nph) do {} while (0)
#endif


constexpr int kDirectRows = 64;               // accumulator rows per sample of the "direct statistics" form (power of two)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));


// LeakyReLU(0.2): max(v, 0.2*v) is the same value bit for bit as (v > 0 ? v : 0.2*v) -- v > 0 gives v > 0.2v,
// v < 0 gives 0.2v > v, +-0 and NaN map to themselves -- and is one VALU instruction shorter
// max(a, b) as ONE v_max_f32: fmaxf() compiles to three (both operands are first "canonicalised" by a v_max_f32 with themselves, the
// IEEE treatment of signalling NaNs) -- 48 instead of 16 instructions per 16 outputs in every LeakyReLU epilogue.  Same result for
// every non-NaN input; a NaN operand yields the other operand (as fmaxf), two NaNs a NaN.
__device__ __forceinline__ float max1(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float lrelu(float v) { return max1(v, 0.2f * v); }

// rint(v * scale) as a wrapping 64-bit integer (1.5*2^52 magic constant); order-independent sums
__device__ __forceinline__ unsigned long long to_fixed(float v, double scale) {
    const double magic = 6755399441055744.0;
    double t = fma((double)v, scale, magic);
    return (unsigned long long)__double_as_longlong(t) - (unsigned long long)__double_as_longlong(magic);
}

// Fixed-point sums of squares (round 4; gsa_kernels.h): the unit is 2^-S2 with S2 = clamp(40 - ceil(log2(H*W)), 20, 26), a static
// function of the plane size -- fine enough on small planes that E[x^2] - mean^2 survives the cancellation when a plane's values
// sit on a bias far above their spread, coarse enough at 1024^2 that the 64-bit sum holds rms(x) < 2.9e3.  A quad whose sum of
// squares would not fit the 1.5*2^52 conversion at that unit (q >= 2^(50-S2)) is rounded at 2^-20 instead and shifted into the
// unit: every term stays an integer multiple of 2^-S2 and a pure function of q, so the sum is order-independent as before, and the
// per-value range stays |x| < 2.3e4 at every plane size.  S2 is wave-uniform (scalar registers).
__device__ __forceinline__ int stat_s2(int HW) {
    const int s2 = 40 - (HW > 1 ? 32 - __builtin_clz((unsigned)HW - 1u) : 0);
    return s2 < 20 ? 20 : (s2 > 26 ? 26 : s2);
}
__device__ __forceinline__ double stat_scale2(int HW) { return __hiloint2double((1023 + stat_s2(HW)) << 20, 0); }
__device__ __forceinline__ unsigned long long to_fixed_sq(float q, int s2) {
    // pre-scaling a big q by the exact power of two 2^-(S2-20) and shifting the integer back IS rounding it at 2^-20; the
    // conversion itself keeps its wave-uniform scale (one scalar register pair), only the mask and two selects are per lane
    const bool big = q >= __int_as_float((127 + 50 - s2) << 23);
    const float qs = big ? q * __int_as_float((127 + 20 - s2) << 23) : q;
    const unsigned long long k = to_fixed(qs, __hiloint2double((1023 + s2) << 20, 0));
    return k << (big ? s2 - 20 : 0);
}

// value of another lane of the same aligned quad (DPP quad_perm, no LDS traffic)
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));   // quad_perm: every lane has a source
}

// 4x4 transpose across the four lanes of an aligned quad (two DPP butterfly steps, no LDS): lane j
// enters with M[r][j] in register r and leaves with M[j][r].  The MFMA C layout (lane = channel,
// register = x) becomes (lane = x, registers = 4 consecutive channels): one 16-byte access per lane
// instead of four dword accesses, and the 16 lanes of a patch row cover 4 whole pixels.
__device__ __forceinline__ f32x4 quad_transpose(float v0, float v1, float v2, float v3, int j) {
    const bool b0 = (j & 1) != 0, b1 = (j & 2) != 0;
    float pa = dpp_quad<0xB1>(v0), pb = dpp_quad<0xB1>(v1);
    const float a0 = b0 ? pb : v0, a1 = b0 ? v1 : pa;
    pa = dpp_quad<0xB1>(v2); pb = dpp_quad<0xB1>(v3);
    const float a2 = b0 ? pb : v2, a3 = b0 ? v3 : pa;
    pa = dpp_quad<0x4E>(a0); pb = dpp_quad<0x4E>(a2);
    const float c0 = b1 ? pb : a0, c2 = b1 ? a2 : pa;
    pa = dpp_quad<0x4E>(a1); pb = dpp_quad<0x4E>(a3);
    const float c1 = b1 ? pb : a1, c3 = b1 ? a3 : pa;
    return f32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return ((unsigned long long)hi << 32) | lo;
}

// Packed fp32 adds for the Winograd transforms: one v_pk_add_f32 per two additions (neg modifiers for a subtraction; the
// same IEEE results).  hipcc scalarises a <4 x float> add whose only users are element extracts -- each component feeds its
// own MFMA -- into four v_add_f32 / v_sub_f32 (and folds every other spelling of the subtraction back into one), so the
// instruction is written out.  f32 MFMAs and vector-ALU instructions share the SIMD's issue time:
// every instruction saved is MFMA time gained.
// Hazards: the compiler's hazard recognizer does not see through inline asm, so the two software-managed ones are handled
// here.  A vector-ALU result needs 2 wait states before an MFMA reads it as an operand: the transformed patch goes through
// valu_settle() (s_nop 1 with the four vectors of a frequency row as in/out operands) before the MFMAs of that row.  An MFMA RESULT needs 11 wait
// states after an 8-pass MFMA before a vector-ALU instruction reads it: the accumulators go through mfma_settle() before the
// first packed add.  (The waits for the LDS reads feeding an asm are inserted by the compiler from its register operands.)
__device__ __forceinline__ f32x2 pk_add2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 fma2v(f32x2 a, f32x2 m, f32x2 b) {      // fmaf(a, m, b) per component (IEEE fused: the same bits as fmaf)
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(m), "v"(b));
    return r;
}
__device__ __forceinline__ f32x4 add4(const f32x4& a, const f32x4& b) {
    const f32x2 lo = pk_add2(a.xy, b.xy), hi = pk_add2(a.zw, b.zw);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 sub4(const f32x4& a, const f32x4& b) {
    const f32x2 lo = pk_sub2(a.xy, b.xy), hi = pk_sub2(a.zw, b.zw);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void valu_settle(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));      // not volatile: ordered by its operands only
}
// 12 wait states with the sixteen accumulators of one output-channel tile as in/out operands: every later read of them is
// ordered behind it
__device__ __forceinline__ void mfma_settle(f32x4 (&a)[16]) {
    asm("s_nop 7\n\ts_nop 3"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                   "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}

// more packed fp32 operations (IEEE, the same bits as the scalar forms)
__device__ __forceinline__ f32x2 pk_mul2(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a * b.x and a * b.y (both halves of the product take ONE half of b: op_sel picks the dword of the 64-bit source)
__device__ __forceinline__ f32x2 pk_mul2_lo(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_mul2_hi(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 fma4(const f32x4& a, const f32x4& m, const f32x4& b) {
    const f32x2 lo = fma2v(a.xy, m.xy, b.xy), hi = fma2v(a.zw, m.zw, b.zw);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
// LeakyReLU(0.2) = max(v, 0.2 v) (gsa_kernels.hip lrelu: the same bits as the select form)
__device__ __forceinline__ f32x4 lrelu4(const f32x4& v, f32x2 k02) {      // k02 = {0.2f, 0.2f}
    const f32x2 lo = pk_mul2(v.xy, k02), hi = pk_mul2(v.zw, k02);
    return f32x4{max1(v.x, lo.x), max1(v.y, lo.y), max1(v.z, hi.x), max1(v.w, hi.y)};
}

// ---- activation tensors: fp32, or (bf16 mode, BF = true) bf16 in the same NHWC order: the element index is the same, the element size is not
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <bool BF>
__device__ __forceinline__ f32x4 act_load4(const float* base, size_t idx) {      // 4 consecutive channels at element idx
    if constexpr (BF) {
        const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(base) + idx);
        return f32x4{bf16_lo(r[0]), bf16_hi(r[0]), bf16_lo(r[1]), bf16_hi(r[1])};
    } else {
        return *reinterpret_cast<const f32x4*>(base + idx);
    }
}

template <bool BF>
__device__ __forceinline__ void act_store4(float* base, size_t idx, const f32x4& v) {
    if constexpr (BF) {
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(base) + idx) =
            u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2)),
                  __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2))};
    } else {
        *reinterpret_cast<f32x4*>(base + idx) = v;
    }
}


// Workgroups are dispatched round-robin over the 8 XCDs (workgroup i -> XCD i % 8) and each XCD has its own L2.
// Persistent kernels therefore hand XCD x the x-th contiguous eighth of the work: the vertical neighbours of a
// tile (one image row of tiles further on) are then processed on the same XCD at about the same time, and the
// halo rows they share are served by that L2 instead of being fetched from HBM once per XCD.
__device__ __forceinline__ int xcd_block(int b, int nb) {
    return (nb & 7) == 0 && !GSA_NO_XCD_REMAP ? (b & 7) * (nb >> 3) + (b >> 3) : b;
}


}  // namespace gsa
