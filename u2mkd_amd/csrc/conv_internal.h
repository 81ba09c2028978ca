// Internal declarations shared by the sparse-conv translation units (conv.hip, conv_tp.hip).
#pragma once
#include "common.h"

namespace u2mkd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the bf16 matrix instruction of every kernel of this library --------------------------------------------------------------
// D += A[16 x 32] * B[32 x 16] on bf16 operands, fp32 accumulate; a / b = the lane's 8 consecutive k of the 16x16x32 operand maps
// (lane l: row or column l & 15, k = 8 (l >> 4) + j).
//
// gfx950's own form, v_mfma_f32_16x16x32_bf16, is NOT used by default: while waves of it (or of v_mfma_f32_32x32x16_bf16 /
// v_mfma_f32_16x16x32_f16) execute, waves of OTHER kernels resident on the chip -- another HIP stream's -- can compute wrong
// results from correct inputs: tools/repro_concurrent_kernels.hip, a program without torch, shows u2mkd_ti_weights returning a
// weight of 0 for lanes 48..63 of a wave in ~50 % of its launches next to such a loop, and never next to the gfx942 forms, the
// fp32 matrix instructions or any non-matrix load (NOTES N9).  The product is the same set of 32 products either way: two
// v_mfma_f32_16x16x16_bf16 take k = 8 (l >> 4) + {0..3} and + {4..7} of the same operand registers (a dot product does not care
// which slot a k sits in as long as A and B agree), so fragments, LDS images and weight layouts are unchanged; only the order in
// which the 32 products reach the fp32 accumulator differs (still fixed, results stay bitwise reproducible).
// -DU2MKD_MFMA_GFX950_K32=1 builds the gfx950 form (tools/build_variant.sh: A/B runs only).
#ifndef U2MKD_MFMA_GFX950_K32
#define U2MKD_MFMA_GFX950_K32 0
#endif
typedef short u2_s16x8 __attribute__((ext_vector_type(8)));
typedef short u2_s16x4 __attribute__((ext_vector_type(4)));
// one half (H = 0: k slots 0..3 of every lane, H = 1: slots 4..7) of the K = 32 product: a caller with several independent
// products interleaves their halves so that no two consecutive matrix instructions accumulate into the same registers
template <int H, class V8>
__device__ __forceinline__ f32x4 mfma_bf16_k32_half(const V8 &a, const V8 &b, f32x4 c) {
    static_assert(sizeof(V8) == 16, "mfma_bf16_k32_half: 8 bf16 per lane");
    const u2_s16x8 a8 = __builtin_bit_cast(u2_s16x8, a), b8 = __builtin_bit_cast(u2_s16x8, b);
    return H == 0 ? __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a8.lo, b8.lo, c, 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a8.hi, b8.hi, c, 0, 0, 0);
}

template <class V8>
__device__ __forceinline__ f32x4 mfma_bf16_k32(const V8 &a, const V8 &b, f32x4 c, int = 0, int = 0, int = 0) {
    static_assert(sizeof(V8) == 16, "mfma_bf16_k32: 8 bf16 per lane");
#if U2MKD_MFMA_GFX950_K32
    typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), c, 0, 0, 0);
#else
    const u2_s16x8 a8 = __builtin_bit_cast(u2_s16x8, a), b8 = __builtin_bit_cast(u2_s16x8, b);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a8.lo, b8.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a8.hi, b8.hi, c, 0, 0, 0);
#endif
}

// ---- fp32 through TWO fp16 operands ("f16x2") ------------------------------------------------------------------------------------
// x * s = h + l (+ a rest below 2^-23 |x s|) with h = fp16(x s), l = fp16(x s - h), s a power of two that puts the largest |x| of
// the scaled set (a gathered row; a weight tensor) into [2^14, 2^15): both planes stay inside fp16's range whatever the magnitude
// of the fp32 data, and the result is scaled back exactly.  a * b = hh + hl + lh (+ ll, below 2^-24 |a b|, dropped): THREE
// products instead of bf16x3's six at the same accuracy class (round-to-nearest planes: ~2^-23 per product), on
// v_mfma_f32_16x16x16_f16 -- a gfx942 form, clean next to other streams (see above), same rate as the bf16 one.
typedef _Float16 u2_h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 u2_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 u2_h16x2 __attribute__((ext_vector_type(2)));
template <int H, class V8>
__device__ __forceinline__ f32x4 mfma_f16_k32_half(const V8 &a, const V8 &b, f32x4 c) {
    static_assert(sizeof(V8) == 16, "mfma_f16_k32_half: 8 fp16 per lane");
    const u2_h16x8 a8 = __builtin_bit_cast(u2_h16x8, a), b8 = __builtin_bit_cast(u2_h16x8, b);
    return H == 0 ? __builtin_amdgcn_mfma_f32_16x16x16f16(a8.lo, b8.lo, c, 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_16x16x16f16(a8.hi, b8.hi, c, 0, 0, 0);
}
// the power of two s (and 1 / s) that puts m >= 0 into [2^14, 2^15); exponents clamped so that both stay normal floats
__device__ __forceinline__ void f16x2_scale(float m, float &s, float &inv) {
    unsigned e = __float_as_uint(m) >> 23;
    e = e < 15u ? 15u : (e > 253u ? 253u : e);
    s = __uint_as_float((268u - e) << 23);
    inv = __uint_as_float((e - 14u) << 23);
}
// two scaled values -> their packed h and l planes
__device__ __forceinline__ void f16x2_split2(float a, float b, uint32_t &h, uint32_t &l) {
    const u2_h16x2 hv = {(_Float16)a, (_Float16)b};
    const u2_h16x2 lv = {(_Float16)(a - (float)hv[0]), (_Float16)(b - (float)hv[1])};
    h = __builtin_bit_cast(uint32_t, hv);
    l = __builtin_bit_cast(uint32_t, lv);
}

// Rows a table-walking forward launch covers: sorted rows [begin, end) of the neighbour table
// nbr[k * ld + row].
struct RowRange {
    int64_t ld, begin, end;
    const int32_t *tile_order;   // launch order of the 64-row tiles (heaviest first) or nullptr (offset-walking kernels)
    // optional per-column epilogue of the tile-pair kernel (inference: an eval-mode BatchNorm (+ residual, + ReLU) folded into
    // the convolution's store, u2mkd_conv_forward_tiles_ep): out = relu?(acc * ep_scale[col] + ep_shift[col] (+ ep_res[row][col]))
    const float *ep_scale = nullptr, *ep_shift = nullptr, *ep_res = nullptr;
    int ep_relu = 0;
};

// Tile-local pair schedule (conv_tp.hip); wf = weight fragments of launch_weight_fragments (same `arith`).
// Returns -1 when the shape has no instantiation, else the launch status.
int launch_conv_tp(const char *who, const float *in, int cin, const float *wf, int cout, const int32_t *nbr,
                   const int32_t *order, RowRange rr, const int32_t *items, const int32_t *n_items, int k, int kflip,
                   int arith, float *out, hipStream_t st, unsigned long long *stamps = nullptr);
bool conv_tp_supported(int cin, int cout, int k);
bool conv_tp_f16x2_supported(int cin);
int conv_tp_arith(int arith);
size_t weight_fragments_bytes(int k, int rows, int cols, int arith);
int launch_weight_fragments(const float *w, int k, int rows, int cols, int transpose, int arith, float *wf, hipStream_t st);
int launch_weight_fragments_batch(const int64_t *jobs, int n_jobs, int64_t total_units, hipStream_t st);

// Global pair schedule in bf16x3 arithmetic (conv_px3.hip); wf = arith-2 fragments.  -1 = shape not supported.
bool conv_px3_supported(int cin, int cout);
int launch_conv_px3(const char *who, const float *in, int cin, const float *wf, int cout, const int32_t *pair_idx,
                    const int32_t *tile_k, const int32_t *n_tiles, int64_t capacity, float *y, hipStream_t st, bool b16 = false,
                    int f16x2_k = 0 /* > 0: wf = the arith-4 (f16x2) fragments of a weight with this many offsets */);
// the same kernel with the identity pair list: y[n_rows, cout] = in x B (+ bias), one offset (nn.Linear)
int launch_linear_px3(const char *who, const float *in, int64_t n_rows, int cin, const float *wf, int cout,
                      const float *bias, float *y, hipStream_t st, bool b16 = false, bool f16x2 = false);

// Weight gradient in bf16x3 arithmetic (conv_wgrad_x3.hip): 64 x 64-channel tiles, slabs as the f32 kernel.
bool conv_wgrad_x3_supported(int ca, int cb, int k);
int launch_conv_wgrad_x3(const float *a, int ca, const float *b, int cb, const int32_t *pairs, const int32_t *plan,
                         int k, int swap, int g, int merge, float *slabs, hipStream_t st, bool b16);
int conv_wgrad_x3_merge(int ca, int cb);

}  // namespace u2mkd
