// Weight gradient over the compacted pair list in bf16x3 arithmetic (gfx950, v_mfma_f32_16x16x32_bf16 +
// ds_read_b64_tr_b16).
//
// dW[k] = sum over the pairs (i, j) of offset k of A_i^T B_j -- the dW half of torchsparse v1.4.0
// convolution_backward_cuda behind every spnn.Conv3d of core/models/build_blocks.py:25-80.
// conv_wgrad_pairs_kernel (conv.hip) does it on v_mfma_f32_16x16x4_f32 and sits at 82 % of the fp32 matrix pipe in
// steady state (in-kernel stamps): only fewer matrix cycles can make it faster.  Here the product
// runs in the bf16x3 arithmetic of conv_tp.hip: every fp32 element of BOTH gathered operands is split exactly into three
// bf16 (h + m + l = x) and the six partial products above 2^-24 relative are accumulated in fp32 -- 24 MFMAs of 16
// cycles per 32 pairs and 32 x 32 channels instead of 32 of 32 cycles.  What makes it pay this time (a first attempt,
// round 2, was slower than the f32 kernel):
//   * the split is by TRUNCATION (x & 0xffff0000, exact residuals): 2 and + 2 sub + 3 byte-permutes per two
//     elements and plane instead of the convert / widen / subtract chain (~30 % fewer VALU instructions);
//   * the pairs are the REDUCTION dimension of the MFMA, so both operands are needed channel-major while the
//     gathers arrive pair-major: the planes are stored pair-major ([plane][pair][channel], one 8-byte store per
//     plane and 4 channels) and read back with the transposing LDS read (ds_read_b64_tr_b16: a 16-lane group fetches
//     4 pairs x 16 channels and every lane receives its channel's 4 pairs) -- two reads per operand fragment and
//     plane, no shuffles;
//   * ONE LDS image per workgroup (27 KB: 4 workgroups per CU) filled between two barriers, the gathered rows of the
//     next two chunks in registers -- the other workgroups of the CU cover the fill.
// Same plan, slabs and fixed-order reduction as the f32 kernel (deterministic); 64 x 64-channel tiles.
//
// (Tried and rejected, round 5: this kernel in the f16x2 arithmetic of conv_tp.hip / conv_px3.hip -- two fp16 planes, three
// products.  The pairs are the reduction dimension, so a scale has to be common to a 32-pair chunk: row a_p scaled by its own
// power of two, b_p by the inverse, then ONE scale per chunk from the largest |b'|, exchanged between the four waves through
// LDS in front of the barrier that frees the image.  Half the matrix instructions, a third less LDS -- and SLOWER, with the
// row maxima before the multiply of the chunk in the image (every wave waits for its gathered rows at the point the pipeline
// was built to hide: 41 against 36 us at 64 x 64 on the 80k scene) and with them between that multiply and the barrier (same
// box, standalone: 43.2 against 38.9 us; 128 x 256 on a 12k scene 42.7 against 38.6; only 128 x 128 gains, 110 against 114):
// the chain per chunk is barrier - transposing reads - products - barrier - split - barrier, and the maxima lengthen the part
// of it that no other workgroup's products cover.  Accuracy against float64 was no better either (largest error / sum |x||g|
// on rows of 1e-10..1e10: 2^-18.3 against bf16x3's 2^-20.1 at 128 x 256).  bf16x3 stays.)
#include <type_traits>

#include "conv_internal.h"

namespace u2mkd {

typedef short wx_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wx_bf16x8 __attribute__((ext_vector_type(8)));

// bytes per (plane, pair) row.  128 = the 64 bf16 channels, no pad, the four 32-byte windows (16 channels) of pair-row p stored at
// window (c ^ h(p)), h(p) = bit 1 of p | bit 3 of p << 1: a ds_read_b64_tr_b16 is served in two 32-lane halves, each reading rows
// {0..3, 8..11} (+4, +16, +20) x one 32-byte window -- with that swizzle the 8 rows land on 8 different 8-bank groups (rows p and
// p+1 are the two halves of a 256-byte bank line).  (Rounds 2-3: 144-byte rows, every transposing read a 2-way conflict.)
constexpr int kWxRow = 128;
constexpr int kWxCP = 32;                      // pairs per step = the K of one MFMA
constexpr int kWxOperand = 3 * kWxCP * kWxRow; // bytes of one operand's image: [3 planes][32 pairs][row]

// x = h + m + l, each the upper 16 bits of an fp32 (a bf16), by truncation: the residuals are exact
struct WxPlanes { uint2 h, m, l; };
__device__ __forceinline__ WxPlanes wx_split4(const f32x4 &v) {
    uint32_t x[4], h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = __float_as_uint(v[i]);
        h[i] = x[i] & 0xffff0000u;
        const float r1 = v[i] - __uint_as_float(h[i]);
        m[i] = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(m[i]);
        l[i] = __float_as_uint(r2);            // <= 8 significant bits left: its upper half is exact
    }
    WxPlanes p;
    // two bf16 per dword: element i in the low half, i+1 in the high half
    p.h = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    p.m = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    p.l = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
    return p;
}

__device__ __forceinline__ wx_bf16x8 wx_frag(const char *base_lo, const char *base_hi) {
    // 8 consecutive pairs of this lane's channel: two transposing reads of 4 pairs each
    const wx_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wx_s16x4 __attribute__((address_space(3))) *)(base_lo));
    const wx_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wx_s16x4 __attribute__((address_space(3))) *)(base_hi));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(wx_bf16x8, v);
}

// B16: a and b are BF16 rows (bf16 storage): no split, ONE plane in the image, one MFMA per product.
template <bool B16>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
conv_wgrad_x3_kernel(const float *__restrict__ a, int ca, const float *__restrict__ b, int cb,
                     const int32_t *__restrict__ pairs, const int32_t *__restrict__ plan, int K, int swap,
                     int merge, float *__restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [A | B][3 (B16: 1)][32][kWxRow]
    constexpr int NPL = B16 ? 1 : 3;
    constexpr int OPB = NPL * kWxCP * kWxRow;                         // bytes of one operand's image
    // this workgroup walks `merge` consecutive workgroup slots of the plan (chunks of `ch` pairs, cut per offset): a run of
    // slots inside one offset is ONE contiguous pair range = one segment with one slab (written at the run's first slot;
    // wgrad_pairs_reduce_kernel reads exactly those).  merge = 1: a segment = a chunk, as the f32 kernel has it.  The wide
    // layers run merge = 8 / 16: a 64 x 64 tile of a 512 x 512 weight at 229-pair chunks wrote (and the reduce re-read) 523
    // slabs of 1 MB per launch.
    const int *wg = plan + 3 + K;
    const int ch = plan[1];
    const int lsel = min((int)(threadIdx.x & 63), K);
    const int wgv = wg[lsel], kof = plan[2 + lsel];
    const int total = __builtin_amdgcn_readlane(wgv, K);
    int wlo = blockIdx.x * merge;
    const int whi = min(wlo + merge, total);
    if (wlo >= whi) return;
    const int a0 = (blockIdx.y / ((cb + 63) / 64)) * 64, b0 = (blockIdx.y % ((cb + 63) / 64)) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gq = lane >> 4, li = lane & 15;                // 16-lane group = 8 pairs of the K dimension; lane = channel
    const int wy = wave >> 1, wx = wave & 1;

    f32x4 acc[2][2];
    int p_begin = 0, p_end = 0;          // the current segment's pair range

    // gather: 32 pairs x 16 chunks of 4 channels per operand = 512 chunks, 2 per thread and operand
    f32x4 ra[2][2], rb[2][2];
    int ia[2][2], ib[2][2];
    auto load_idx = [&](int p0, int (&xa)[2], int (&xb)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pr = (tid + 256 * i) >> 4;
            const int pp = min(p0 + pr, p_end - 1);
            xa[i] = pairs[2 * (size_t)pp + (swap ? 1 : 0)];
            xb[i] = pairs[2 * (size_t)pp + (swap ? 0 : 1)];
        }
    };
    auto load_rows = [&](const int (&xa)[2], const int (&xb)[2], f32x4 (&va)[2], f32x4 (&vb)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = ((tid + 256 * i) & 15) * 4;
            if (B16) {     // 4 bf16 channels = 8 bytes, kept in the first two dwords
                const uint2 ua = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(a) + ((size_t)xa[i] * ca + min(a0 + c, ca - 4)) * 2);
                const uint2 ub = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(b) + ((size_t)xb[i] * cb + min(b0 + c, cb - 4)) * 2);
                va[i] = (f32x4){__uint_as_float(ua.x), __uint_as_float(ua.y), 0.f, 0.f};
                vb[i] = (f32x4){__uint_as_float(ub.x), __uint_as_float(ub.y), 0.f, 0.f};
            } else {
                va[i] = *reinterpret_cast<const f32x4 *>(a + (size_t)xa[i] * ca + min(a0 + c, ca - 4));
                vb[i] = *reinterpret_cast<const f32x4 *>(b + (size_t)xb[i] * cb + min(b0 + c, cb - 4));
            }
        }
    };
    const int hs = ((tid >> 5) & 1) | (((tid >> 7) & 1) << 1);       // h(pair row) of this thread's stores (rows tid >> 4, + 16)
    auto store_impl = [&](auto FULL, int p0, const f32x4 (&va)[2], const f32x4 (&vb)[2]) __attribute__((always_inline)) {
        constexpr bool kFull = decltype(FULL)::value;          // a whole chunk inside the range and whole tiles: no masks
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            const int pr = f >> 4, c = (f & 15) * 4;
            const bool live = kFull || p0 + pr < p_end;
            const f32x4 xa = (kFull || (live && a0 + c < ca)) ? va[i] : zero, xb = (kFull || (live && b0 + c < cb)) ? vb[i] : zero;
            const int cofs = ((((f & 15) >> 2) ^ hs) << 5) | ((f & 3) << 3);
            char *da = smem + pr * kWxRow + cofs, *db = smem + OPB + pr * kWxRow + cofs;
            if (B16) {
                *reinterpret_cast<uint2 *>(da) = make_uint2(__float_as_uint(xa[0]), __float_as_uint(xa[1]));
                *reinterpret_cast<uint2 *>(db) = make_uint2(__float_as_uint(xb[0]), __float_as_uint(xb[1]));
                continue;
            }
            const WxPlanes pa = wx_split4(xa), pb = wx_split4(xb);
            *reinterpret_cast<uint2 *>(da) = pa.h;
            *reinterpret_cast<uint2 *>(da + kWxCP * kWxRow) = pa.m;
            *reinterpret_cast<uint2 *>(da + 2 * kWxCP * kWxRow) = pa.l;
            *reinterpret_cast<uint2 *>(db) = pb.h;
            *reinterpret_cast<uint2 *>(db + kWxCP * kWxRow) = pb.m;
            *reinterpret_cast<uint2 *>(db + 2 * kWxCP * kWxRow) = pb.l;
        }
    };
    auto store_chunk = [&](int p0, const f32x4 (&va)[2], const f32x4 (&vb)[2]) __attribute__((always_inline)) {
        if (p0 + kWxCP <= p_end && a0 + 64 <= ca && b0 + 64 <= cb) store_impl(std::true_type{}, p0, va, vb);      // (uniform)
        else store_impl(std::false_type{}, p0, va, vb);
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // transposing read: lane 4q+p of a 16-lane group supplies row q (of 4 pairs), channels 4p .. 4p+3; lane i receives
    // channel i's 4 pairs.  Group gq covers pairs 8 gq .. 8 gq + 7 (two reads).
    const int tq = li >> 2, tp = li & 3;
    const int hr = ((tq >> 1) & 1) | ((gq & 1) << 1);                // h(row) of this lane's reads (rows 8 gq + tq, + 4)
    auto multiply = [&]() __attribute__((always_inline)) {
        wx_bf16x8 fa[2][NPL], fb[2][NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const char *base = smem + (pl * kWxCP + 8 * gq + tq) * kWxRow + ((((2 * wy + m) ^ hr) << 5) | (tp << 3));
                fa[m][pl] = wx_frag(base, base + 4 * kWxRow);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const char *base = smem + OPB + (pl * kWxCP + 8 * gq + tq) * kWxRow + ((((2 * wx + n) ^ hr) << 5) | (tp << 3));
                fb[n][pl] = wx_frag(base, base + 4 * kWxRow);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 c = acc[m][n];
                if (B16) {
                    c = mfma_bf16_k32(fa[m][0], fb[n][0], c, 0, 0, 0);
                } else {
                    // the six partial products, low order first (plane 0 = h, 1 = m, 2 = l)
                    c = mfma_bf16_k32(fa[m][NPL - 1], fb[n][0], c, 0, 0, 0);
                    c = mfma_bf16_k32(fa[m][0], fb[n][NPL - 1], c, 0, 0, 0);
                    c = mfma_bf16_k32(fa[m][NPL / 2], fb[n][NPL / 2], c, 0, 0, 0);
                    c = mfma_bf16_k32(fa[m][NPL / 2], fb[n][0], c, 0, 0, 0);
                    c = mfma_bf16_k32(fa[m][0], fb[n][NPL / 2], c, 0, 0, 0);
                    c = mfma_bf16_k32(fa[m][0], fb[n][0], c, 0, 0, 0);
                }
                acc[m][n] = c;
            }
    };

    while (wlo < whi) {
        // the segment's offset = the number of prefix entries wg[1..K] <= wlo (empty offsets are skipped by construction)
        const unsigned long long le = __ballot(wgv <= wlo) & ((2ULL << K) - 2ULL);
        const int k = __builtin_amdgcn_readfirstlane(__popcll(le));
        const int wk = __builtin_amdgcn_readlane(wgv, k), seg_hi = min(whi, __builtin_amdgcn_readlane(wgv, k + 1));
        p_begin = __builtin_amdgcn_readlane(kof, k) + (wlo - wk) * ch;
        p_end = min(__builtin_amdgcn_readlane(kof, k) + (seg_hi - wk) * ch, __builtin_amdgcn_readlane(kof, k + 1));
        const int w = wlo;                   // the slab of this segment
        wlo = seg_hi;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // prologue: chunk 0 in the image, rows of chunk 1 and indices of chunk 2 in flight
        load_idx(p_begin, ia[0], ib[0]);
        load_idx(p_begin + kWxCP, ia[1], ib[1]);
        load_rows(ia[0], ib[0], ra[0], rb[0]);
        load_idx(p_begin + 2 * kWxCP, ia[0], ib[0]);
        load_rows(ia[1], ib[1], ra[1], rb[1]);
        store_chunk(p_begin, ra[0], rb[0]);
        lds_barrier();
        for (int p0 = p_begin; p0 < p_end; p0 += 2 * kWxCP) {
            // even chunk c: rows c+1 in set 1, indices c+2 in set 0
            load_idx(p0 + 3 * kWxCP, ia[1], ib[1]);
            load_rows(ia[0], ib[0], ra[0], rb[0]);                 // rows of chunk c+2
            __builtin_amdgcn_sched_barrier(0);
            multiply();
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();                                         // every wave has read chunk c
            store_chunk(p0 + kWxCP, ra[1], rb[1]);
            lds_barrier();
            // odd chunk c+1: rows c+2 in set 0, indices c+3 in set 1
            load_idx(p0 + 4 * kWxCP, ia[0], ib[0]);
            load_rows(ia[1], ib[1], ra[1], rb[1]);                 // rows of chunk c+3
            __builtin_amdgcn_sched_barrier(0);
            multiply();                                            // (a range with an odd number of chunks multiplies zeros)
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            store_chunk(p0 + 2 * kWxCP, ra[0], rb[0]);
            lds_barrier();
        }
        // D[i = a channel][j = b channel]: row = 4 gq + reg, col = li
        float *slab = slabs + (size_t)w * ca * cb;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ach = a0 + 16 * (2 * wy + m) + 4 * gq + reg;
                if (ach < ca) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int bch = b0 + 16 * (2 * wx + n) + li;
                        if (bch < cb) slab[(size_t)ach * cb + bch] = acc[m][n][reg];
                    }
                }
            }
    }
}

bool conv_wgrad_x3_supported(int ca, int cb, int k) { return ca % 4 == 0 && cb % 4 == 0 && ca >= 4 && cb >= 4 && k <= 63; }

int launch_conv_wgrad_x3(const float *a, int ca, const float *b, int cb, const int32_t *pairs, const int32_t *plan,
                         int k, int swap, int g, int merge, float *slabs, hipStream_t st, bool b16) {
    const int tiles_a = (ca + 63) / 64, tiles_b = (cb + 63) / 64;
    dim3 grid((g + merge - 1) / merge, tiles_a * tiles_b);
    if (b16)
        hipLaunchKernelGGL(conv_wgrad_x3_kernel<true>, grid, dim3(256), (size_t)2 * kWxCP * kWxRow, st, a, ca, b, cb, pairs,
                           plan, k, swap, merge, slabs);
    else
        hipLaunchKernelGGL(conv_wgrad_x3_kernel<false>, grid, dim3(256), (size_t)2 * kWxOperand, st, a, ca, b, cb, pairs, plan,
                           k, swap, merge, slabs);
    return check_launch("u2mkd_conv_wgrad_pairs");
}

// Slots merged per workgroup, by the number of 64 x 64 tiles of the weight: every tile is a workgroup of its own, so the wide
// layers have workgroups to spare, and their slabs (slots x ca x cb floats, written here and re-read by the reduce) were a
// large part of what a launch moved.  Measured (MI355X, 80k scene, tools/ab_wgrad_layers.py, merge 1 -> 4 / 6): 128 x 128 at
// stride 2 95 -> 84 us, 256 x 256 at stride 4 299 -> 238 us, 512 x 512 at stride 8 636 -> 518 us, 768 x 512 925 -> 756 us,
// 192 x 192 at stride 1 220 -> 193 us; merge 12-16 loses again (too few workgroups for an even last wave); 64 x 64 keeps 1
// (45 us against 35 us at merge 4: that shape needs every workgroup it can get).
int conv_wgrad_x3_merge(int ca, int cb) {
    const int tiles = ((ca + 63) / 64) * ((cb + 63) / 64);
    return tiles >= 9 ? 6 : (tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1));
}

}  // namespace u2mkd
