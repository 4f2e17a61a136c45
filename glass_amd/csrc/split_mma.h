// fp32 products on the bf16 matrix cores (gfx950 has no xf32 / tf32 form and its f32-input MFMA runs at 1/16 of the bf16
// rate).  An fp32 value is cut into three bf16 pieces, x = hi + mid + lo, each piece the RNE rounding of what the pieces
// before it left.  The cut is EXACT for |x| >= 2^-100 (8 + 8 + 8 significant bits plus the two sign bits cover fp32's 24;
// below ~2^-109 the low pieces reach the denormal range; tests/test_split_products.py), and a product x*w is the six partial
// products whose weight is >= 2^-18:
//   mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi      (dropped: mid*lo, lo*mid, lo*lo < 2^-24 |x w| together)
// each one exact in the fp32 accumulator's input (8 x 8 significant bits), summed in fp32 by v_mfma_f32_32x32x16_bf16.
// The truncation is below one fp32 rounding of the product; measured against an fp64 product on this chip
// (tools/split_lab.hip, profiles/r04_split_product_accuracy.txt) the result is as close as the f32-input MFMA's or
// closer (the 16 k of one instruction are summed before the accumulator rounds), at 6/16 of its matrix-core cycles.
// Non-finite input: the pieces of +-Inf are (Inf, NaN, NaN) -> the product is NaN where the f32 form gives Inf or NaN.
#pragma once
#include "common.h"

namespace glass {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// (a, b) -> the three bf16 pieces of both, packed (a in bits 0..15, b in bits 16..31)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){a, b}, bf16x2));
    float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){ra, rb}, bf16x2));
    ra -= __builtin_bit_cast(float, mid << 16);
    rb -= __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){ra, rb}, bf16x2));
}

struct Split4 {
    uint2 hi, mid, lo;  // four consecutive k of one row, per piece
};
__device__ __forceinline__ Split4 split4(const float4& v) {
    Split4 s;
    split2(v.x, v.y, s.hi.x, s.mid.x, s.lo.x);
    split2(v.z, v.w, s.hi.y, s.mid.y, s.lo.y);
    return s;
}

// One 32x32 output tile += the six partial products of one 16-deep K block; a[p], b[p]: piece p (0 hi, 1 mid, 2 lo) of
// the operand fragments (lane (j, h): row / column j, k = 8h .. 8h+7).  Small terms first.
__device__ __forceinline__ void split_mma(f32x16& acc, const uint4 (&a)[3], const uint4 (&b)[3]) {
#define GLASS_SMMA(pa, pb)                                                                                            \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[pa]), __builtin_bit_cast(bf16x8, b[pb]), acc, 0, 0, 0)
    GLASS_SMMA(1, 1);
    GLASS_SMMA(2, 0);
    GLASS_SMMA(0, 2);
    GLASS_SMMA(1, 0);
    GLASS_SMMA(0, 1);
    GLASS_SMMA(0, 0);
#undef GLASS_SMMA
}

// LDS image of one operand of one 16-deep K block: [piece 3][h 2][R rows] 16-byte units (8 bf16 = k 8h .. 8h+7 of a row) —
// a fragment is one ds_read_b128, 32 consecutive lanes consecutive units.
template <int R>
struct SplitImg {
    static constexpr int kPlane = 2 * R;     // units per piece
    static constexpr int kUnits = 3 * kPlane;  // 16-byte units per image
    // the four k of k-quad kq (0..3) of row r: 8 bytes per piece
    static __device__ __forceinline__ void put4(float4* img, int r, int kq, const float4& v) {
        const Split4 s = split4(v);
        uint2* p = reinterpret_cast<uint2*>(img) + ((kq >> 1) * R + r) * 2 + (kq & 1);
        p[0] = s.hi;
        p[2 * kPlane] = s.mid;
        p[4 * kPlane] = s.lo;
    }
    // the eight k of half h of row r (two consecutive k-quads): 16 bytes per piece
    static __device__ __forceinline__ void put8(float4* img, int r, int h, const float4& v0, const float4& v1) {
        const Split4 s0 = split4(v0), s1 = split4(v1);
        uint4* p = reinterpret_cast<uint4*>(img) + h * R + r;
        p[0] = make_uint4(s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y);
        p[kPlane] = make_uint4(s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y);
        p[2 * kPlane] = make_uint4(s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y);
    }
    static __device__ __forceinline__ void frag(const float4* img, int r, int h, uint4 (&f)[3]) {
        const uint4* p = reinterpret_cast<const uint4*>(img) + h * R + r;
        f[0] = p[0];
        f[1] = p[kPlane];
        f[2] = p[2 * kPlane];
    }
};

}  // namespace glass
