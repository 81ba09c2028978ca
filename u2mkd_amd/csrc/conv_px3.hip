// Sparse conv forward / input gradient of the WIDE layers: the global pair schedule in bf16x3 arithmetic
// (gfx950, v_mfma_f32_16x16x32_bf16).
//
// Replaces torchsparse v1.4.0 convolution_forward_cuda / the dX half of convolution_backward_cuda
// (gather -> cuBLAS mm -> scatter-add per kernel offset) behind the spnn.Conv3d
// calls of core/models/build_blocks.py:25-80 for layers with cin * cout >= 8192 (the 96..768-channel
// stages of SPVCNN cr 1.0 / 2.0), on the pair schedule of u2mkd_pairs_build: every 64-pair tile of the
// offset-grouped pair list is ONE dense [64 x cin] x [cin x cout] product into the scratch rows y.
//
// conv_pairs_kernel (conv.hip) does this product on v_mfma_f32_16x16x4_f32 with every wave gathering its own
// 16 pairs straight into operand registers: matrix-pipe bound at ~80 TF (256 x 256), half of the fp32 MFMA
// peak.  Here the product runs in the bf16x3 arithmetic of conv_tp.hip (each fp32 operand split exactly into
// three bf16, the six partial products above 2^-24 relative accumulated in fp32: fp32 GEMM accuracy at 2.7x
// fewer matrix-pipe cycles), organised so that the run-time split of the gathered rows is paid ONCE per
// workgroup and amortised over all of its output columns:
//   * a workgroup (NW waves) owns 64 pairs x 16*NW*NBW output columns; the waves split the COLUMNS, every
//     wave multiplies all 64 pairs (4 row blocks) by its NBW column blocks: 4*NBW accumulators, 24*NBW MFMAs
//     per 32-channel step, each weight fragment used 4 times from registers;
//   * the 64 gathered rows of a step (32 channels = one full 128-byte line per row, 8 lanes per row) are
//     split into the h | m | l planes at the LDS store and read back as ready MFMA operand fragments
//     (ds_read_b128 and the plane stores conflict-free: 192-byte rows, chunks swizzled by bit 3 of the row, see RS);
//   * weights come in the MFMA-fragment order of u2mkd_weight_fragments (arith = 2), 1 KiB contiguous per
//     wave load instruction, straight from L2 into registers;
//   * (tile, 32-channel step) pairs form ONE software pipeline over the workgroup's contiguous run of tiles:
//     gather of step i+2 and weight fragments of step i+1 in flight while step i multiplies, rows of step i+1
//     stored into the other LDS image, one LDS-only barrier per step; all loads in the loop are unconditional
//     so the compiler's s_waitcnt counters stay exact.
// Results differ from the f32-MFMA kernel by fp32 rounding only (tests hold both to the same gates).
//
// BF16 STORAGE (B16; BASELINE.json configs[4]: torchsparse runs its conv in half precision under autocast,
// custom_fwd(cast_inputs=half)): the same pipeline on bf16 rows -- one plane in the LDS
// image, no split, ONE v_mfma_f32_16x16x32_bf16 per 32 channels, fp32 accumulators -- with 64-channel steps where
// cin allows (a row's step is again one full 128-byte line), weights = the one-plane fragments of
// u2mkd_weight_fragments(arith 3), scratch rows y / outputs in bf16 (rounded once from the fp32 accumulator).
#include <stdlib.h>

#include <type_traits>

#include "conv_internal.h"

namespace u2mkd {

typedef __bf16 px_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 px_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void px_split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    float r1 = x - (float)h;
    m = (__bf16)r1;
    float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

__device__ __forceinline__ px_bf16x8 px_bf8(const float4 &x) {
    f32x4 v = (f32x4){x.x, x.y, x.z, x.w};
    return __builtin_bit_cast(px_bf16x8, v);
}

// DENSE (nn.Linear over point features, the learner / point_transforms MLPs of spvcnn.py:58-74 and tsd_full.py): the
// "pair list" is the identity -- tile t = rows 64 t .. 64 t + 63 of `in`, one offset -- the bias is added at the
// store and rows past n_rows are not written (y = the [n_rows, cout] output itself, no scratch rows).
// B16: bf16 rows in / out (see the header); SC = channels per pipeline step (fp32 rows: 32; bf16 rows: 64 or 32)
// TL: 64-pair tiles a workgroup multiplies per step against ONE set of weight fragments (a "unit" = TL consecutive tiles of
// one offset: the pair schedule pads every offset's group to 128 entries).  The wide layers are bound by the bytes a step pulls
// through the vector L1 (weight fragments from L2 / Infinity Cache + gathered rows: ~7 TB/s chip-wide at 512 -> 512), and
// per 128 pairs x 256 columns x 32 channels those are 2 x (48 KB weights + 8 KB rows) = 112 KB with TL = 1 and 256-column
// tiles (NBW = 4), 2 x (24 + 16) = 80 KB with TL = 2 and 128-column tiles (NBW = 2) at the same 64 accumulator registers.
// F2: fp32 rows in f16x2 arithmetic (conv_internal.h) instead of bf16x3: the 32 channels of a row's step are scaled by the power of
// two that puts their largest |x| into [2^14, 2^15) and split into two fp16 planes (the third plane's place in the LDS row holds
// 1 / (row scale x weight scale)); a step's three partial products start from zero and join the accumulators through one fma per
// register that takes the scales out again -- half the matrix instructions and two thirds of the weight-fragment bytes of bf16x3.
// wf = the arith-4 fragments, `k_total` = the weight's number of offsets (the scale trailer sits behind all of them).
template <int CTRL>
__device__ __forceinline__ float px_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int NW, int NBW, bool DENSE = false, bool B16 = false, int SC = 32, int TL = 1, bool F2 = false>
__global__ void __launch_bounds__(64 * NW)
conv_px3_kernel(const float *__restrict__ in, int cin, const float *__restrict__ wf, int cout,
                const int32_t *__restrict__ pair_idx, const int32_t *__restrict__ tile_k,
                const int32_t *__restrict__ n_tiles, float *__restrict__ y, const float *__restrict__ bias = nullptr,
                int n_rows = 0, int k_total = 1) {
    static_assert(B16 ? (SC == 32 || SC == 64) : SC == 32, "step width");
    static_assert(!(F2 && B16), "f16x2 is an arithmetic of fp32 rows");
    constexpr int NT = 64 * NW, TN = 16 * NW * NBW;
    // bytes per row of the LDS image: 3 planes x 32 bf16 (or up to 64 bf16), NO pad; the 16-byte chunk c of a 64-byte plane
    // segment of tile row `row` sits at chunk c ^ 2 * bit 3 of row.  A ds_read_b128 is served in four 16-lane groups (rows
    // 0-3 | 12-15 at chunk q with rows 4-11 at chunk q ^ 1, and the complement): 192-byte rows put rows r, r + 4, r + 8, r + 12
    // on the same 64-byte quarter of the 256-byte bank line, the swizzle spreads those four over its four chunks -- every
    // group hits 64 different banks; the 8-byte plane stores (16 lanes = 2 rows x 64 bytes) fall on two different halves of
    // the 128-byte store line.  (Rounds 2-3 ran 208-byte rows: every fragment read and every plane store a 2-way conflict;
    // removing them is worth 1-3 % -- the kernel is not LDS-bound, see DESIGN.md.)
    constexpr int RS = 192;
    constexpr int CH = B16 ? SC / 8 : 8;          // 16-byte chunks of a row's step (fp32: 4 channels each, bf16: 8)
    constexpr int RU = 64 * TL, RB = 4 * TL;      // rows / 16-row blocks of a unit
    constexpr int NCH = RU * CH;                  // chunks per step
    constexpr int LPT = (NCH + NT - 1) / NT;      // 16-byte chunks a thread gathers per step
    constexpr int NWF = B16 ? SC / 32 : F2 ? 2 : 3;        // weight / row fragments per (column block, step): k-steps or planes
    const float wsc = F2 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(wf) + (size_t)k_total * cin * cout * 4)[1] : 1.f;
    const int esz = B16 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][RU][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int ntile64 = DENSE ? (n_rows + 63) / 64 : *n_tiles;
    const int ntile = (ntile64 + TL - 1) / TL;                    // units
    // Work items (unit, column tile) -> workgroups.  Workgroups are dealt to the 8 XCDs round-robin by their linear index
    // (xcd = lin % 8, slot j = lin / 8 of WPX per XCD).  Items are cut into BLOCKS of UPB (= WPX, below) consecutive units x one column tile,
    // block b = (unit group b / gy, column tile b % gy); XCD x takes blocks x, x + 8, ...: at any time all WPX workgroups of an
    // XCD multiply the SAME column tile against WPX consecutive units = one or two offsets, whose fragments (cin x TN x 6 B per
    // offset: 0.4 MB at 512 x 128) stay in that XCD's 4 MB L2 -- with each workgroup on its own contiguous run of units
    // (rounds 2-3) every XCD streamed all 27 offsets at once from the Infinity Cache (L2 hit rate 40 %).
    const int gy = (cout + TN - 1) / TN;
    const int lin = (int)blockIdx.x, wpx = (int)gridDim.x >> 3;
    // short pair lists: halve the block (and split an XCD's slots into teams with blocks of their own) until every team has
    // about three blocks to walk -- otherwise a list of a few blocks would leave whole XCDs idle
    int upb = wpx;
    while ((upb & 1) == 0 && ((ntile + upb - 1) / upb) * gy < 24 * (wpx / upb)) upb >>= 1;
    const int teams = 8 * (wpx / upb);
    const int team = (lin & 7) + 8 * ((lin >> 3) / upb), slot = (lin >> 3) % upb;
    if (slot >= ntile) return;
    const int gmax = (ntile - 1 - slot) / upb;                 // last unit group that holds a unit for this slot
    const int bmax = gmax * gy + gy - 1;
    if (bmax < team) return;
    const int t0 = 0, t1 = (bmax - team) / teams + 1;          // this workgroup's blocks: team + teams t, t = 0 .. t1 - 1
    const int ns = cin / SC;
    const int nsteps = (t1 - t0) * ns;
    const int ncb = cout / 16;
    auto block_of = [&](int t) __attribute__((always_inline)) { return team + teams * min(t, t1 - 1); };
    auto unit_of = [&](int t) __attribute__((always_inline)) { return (block_of(t) / gy) * upb + slot; };
    auto cbw_of = [&](int t) __attribute__((always_inline)) { return (block_of(t) % gy) * (TN / 16) + NBW * wave; };   // the wave's first column block

    // chunk e of a step: row e / CH of the tile, 16-byte chunk e % CH of the row's step bytes
    int crow[LPT], cch[LPT];
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
        const int e = min(tid + l * NT, NCH - 1);
        crow[l] = e / CH;
        cch[l] = e % CH;
    }

    f32x4 acc[RB][NBW];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc[b][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 bw[2][NWF * NBW];
    f32x4 g[2][LPT];
    int gix[LPT];        // pair entries of the tile the NEXT gather reads
    int kw, cw;          // offset / first column block of the unit the NEXT weight issue reads

    auto load_idx = [&](int t, int (&ix)[LPT]) __attribute__((always_inline)) {
        const int tt = unit_of(t);
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            if (DENSE) {
                const int row = tt * RU + crow[l];
                ix[l] = row < n_rows ? row : 0;
            } else {
                ix[l] = pair_idx[(size_t)tt * RU + crow[l]];
            }
        }
    };
    auto issue_G = [&](const int (&ix)[LPT], int s, f32x4 (&gg)[LPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            const int row = ix[l] >= 0 ? ix[l] : 0;       // padding entries multiply row 0; their y rows are never read
            gg[l] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(in) +
                                                     ((size_t)row * cin + SC * s) * esz + 16 * cch[l]);
        }
    };
    auto issue_B = [&](int k, int cbw, int s, float4 (&bb)[NWF * NBW]) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
            const int cb = min(cbw + n, ncb - 1);         // column blocks past cout: computed, never stored
            // fragments of one (offset, column block): [32-channel step][plane][lane][16 B]; a step of this kernel reads
            // NWF consecutive ones (x3: the 3 planes of its 32 channels; bf16: SC / 32 one-plane k-steps)
            const float *pb = wf + ((((size_t)k * ncb + cb) * ns + s) * NWF * 64 + lane) * 4;
#pragma unroll
            for (int p = 0; p < NWF; ++p) bb[NWF * n + p] = *reinterpret_cast<const float4 *>(pb + p * 256);
        }
    };
    auto store_G = [&](const f32x4 (&gg)[LPT], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            if (B16) {
                if (tid + l * NT < NCH)
                    *reinterpret_cast<f32x4 *>(smem + (slot * RU + crow[l]) * RS + 16 * (cch[l] ^ ((crow[l] >> 2) & 2))) = gg[l];
            } else if (F2) {
                // the step's largest |x| of the row: its 8 lanes (every lane of the wave takes part: NCH is a multiple of 64)
                float m = fmaxf(fmaxf(fabsf(gg[l][0]), fabsf(gg[l][1])), fmaxf(fabsf(gg[l][2]), fabsf(gg[l][3])));
                m = fmaxf(m, px_dpp<0xB1>(m));                  // quad_perm [1, 0, 3, 2]
                m = fmaxf(m, px_dpp<0x4E>(m));                  // quad_perm [2, 3, 0, 1]
                m = fmaxf(m, px_dpp<0x141>(m));                 // row_half_mirror: the other quad of the 8
                float rs, rinv;
                f16x2_scale(m, rs, rinv);
                if (tid + l * NT < NCH) {
                    uint32_t h01, l01, h23, l23;
                    f16x2_split2(gg[l][0] * rs, gg[l][1] * rs, h01, l01);
                    f16x2_split2(gg[l][2] * rs, gg[l][3] * rs, h23, l23);
                    char *row = smem + (slot * RU + crow[l]) * RS + 8 * (cch[l] ^ ((crow[l] >> 1) & 4));
                    *reinterpret_cast<uint2 *>(row) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2 *>(row + 64) = make_uint2(l01, l23);
                    if (cch[l] == 0) *reinterpret_cast<float *>(smem + (slot * RU + crow[l]) * RS + 128) = rinv * wsc;
                }
            } else if (tid + l * NT < NCH) {
                // split by truncation (x & 0xffff0000; exact residuals), two bf16 packed per dword by a byte permute
                uint32_t hb[4], mb[4], lb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float x = gg[l][c];
                    hb[c] = __float_as_uint(x) & 0xffff0000u;
                    const float r1 = x - __uint_as_float(hb[c]);
                    mb[c] = __float_as_uint(r1) & 0xffff0000u;
                    lb[c] = __float_as_uint(r1 - __uint_as_float(mb[c]));
                }
                char *row = smem + (slot * RU + crow[l]) * RS + 8 * (cch[l] ^ ((crow[l] >> 1) & 4));
                *reinterpret_cast<uint2 *>(row) = make_uint2(__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u));
                *reinterpret_cast<uint2 *>(row + 64) = make_uint2(__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u));
                *reinterpret_cast<uint2 *>(row + 128) = make_uint2(__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u));
            }
        }
    };
    auto read_frag = [&](int slot, int rb, float4 (&aa)[NWF]) __attribute__((always_inline)) {
        const char *row = smem + (slot * RU + 16 * rb + r) * RS + 16 * (q ^ ((r >> 2) & 2));
#pragma unroll
        for (int p = 0; p < NWF; ++p) aa[p] = *reinterpret_cast<const float4 *>(row + 64 * p);
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto tile_offset = [&](int t) __attribute__((always_inline)) { return DENSE ? 0 : tile_k[TL * unit_of(t)]; };

    // positions (tile, step) of flattened step i + d; advance = next 32-channel step, then next tile
    // positions (block, step) of flattened step i + d; advance = next 32-channel step, then the workgroup's next block
    // (the unit / column tile of a block are re-derived by a division per use: carrying them incrementally was measured
    // 25 % slower -- the extra live scalars pushed hipcc into a vmcnt(0) at the top of the step)
    int tc = t0, sc = 0;                 // step i   (multiplied)
    int tw = t0, sw = 0;                 // step i+1 (weights issued)
    int tg = t0, sg = 0;                 // step i+2 (rows gathered)
    auto adv = [&](int &t, int &s) __attribute__((always_inline)) {
        if (++s == ns) { s = 0; ++t; }
    };

    // ---- prologue: rows of steps 0 and 1 in flight, step 0 stored, weights of step 0 in registers
    {
        int ix0[LPT];
        load_idx(tg, ix0);
        issue_G(ix0, sg, g[0]);
        issue_B(tile_offset(tw), cbw_of(tw), sw, bw[0]);
        adv(tg, sg);
        adv(tw, sw);
        load_idx(tg, ix0);
        issue_G(ix0, sg, g[1]);
        adv(tg, sg);
        load_idx(tg, gix);               // tile of step 2
        kw = tile_offset(tw);            // tile of step 1
        cw = cbw_of(tw);
        store_G(g[0], 0);
        lds_barrier();
    }

    auto step = [&](auto U, int i) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;      // i mod 2: ring positions are static register names
        // -- issue what later steps need (all unconditional)
        const int kcur = __builtin_amdgcn_readfirstlane(kw);
        issue_B(kcur, __builtin_amdgcn_readfirstlane(cw), sw, bw[u ^ 1]);   // weights of step i+1
        issue_G(gix, sg, g[u]);                    // rows of step i+2 (the rows of step i left g[u] at step i-1)
        adv(tw, sw);
        adv(tg, sg);
        kw = tile_offset(tw);                      // step i+2's tile
        cw = cbw_of(tw);
        load_idx(tg, gix);                         // step i+3's tile
        // (measured and rejected: the index loads issued BEFORE the fragment / row loads so that the next step's wait for them
        // leaves those in flight -- no vmcnt(0) left in the loop, and the 256- / 192-column instantiations 20 % slower)
        // -- multiply step i from LDS image u
        float4 a[2][NWF];
        read_frag(u, 0, a[0]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (rb < RB - 1) read_frag(u, rb + 1, a[(rb + 1) & 1]);
            if (B16) {      // one plane: one MFMA per 32 channels; weights = A operand: D[col][pair]
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    f32x4 c = acc[rb][n];
#pragma unroll
                    for (int p = 0; p < NWF; ++p)
                        c = mfma_bf16_k32(px_bf8(bw[u][NWF * n + p]), px_bf8(a[rb & 1][p]), c, 0, 0, 0);
                    acc[rb][n] = c;
                }
                continue;
            }
            if (F2) {      // three partial products from zero, the column blocks' chains interleaved; then the scales out
                const float osc = *reinterpret_cast<const float *>(smem + (u * RU + 16 * rb + r) * RS + 128);
                const float4 xh = a[rb & 1][0], xl = a[rb & 1][NWF - 1];
                f32x4 c[NBW];
#pragma unroll
                for (int n = 0; n < NBW; ++n) c[n] = mfma_f16_k32_half<0>(bw[u][NWF * n + NWF - 1], xh, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int n = 0; n < NBW; ++n) c[n] = mfma_f16_k32_half<1>(bw[u][NWF * n + NWF - 1], xh, c[n]);
#pragma unroll
                for (int n = 0; n < NBW; ++n) c[n] = mfma_f16_k32_half<0>(bw[u][NWF * n], xl, c[n]);
#pragma unroll
                for (int n = 0; n < NBW; ++n) c[n] = mfma_f16_k32_half<1>(bw[u][NWF * n], xl, c[n]);
#pragma unroll
                for (int n = 0; n < NBW; ++n) c[n] = mfma_f16_k32_half<0>(bw[u][NWF * n], xh, c[n]);
#pragma unroll
                for (int n = 0; n < NBW; ++n) c[n] = mfma_f16_k32_half<1>(bw[u][NWF * n], xh, c[n]);
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    acc[rb][n][0] = fmaf(c[n][0], osc, acc[rb][n][0]);
                    acc[rb][n][1] = fmaf(c[n][1], osc, acc[rb][n][1]);
                    acc[rb][n][2] = fmaf(c[n][2], osc, acc[rb][n][2]);
                    acc[rb][n][3] = fmaf(c[n][3], osc, acc[rb][n][3]);
                }
                continue;
            }
            const px_bf16x8 xh = px_bf8(a[rb & 1][0]), xm = px_bf8(a[rb & 1][NWF / 2]), xl = px_bf8(a[rb & 1][NWF - 1]);
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                const px_bf16x8 wh = px_bf8(bw[u][NWF * n]), wm = px_bf8(bw[u][NWF * n + NWF / 2]), wl = px_bf8(bw[u][NWF * n + NWF - 1]);
                // the six partial products, low order first; weights = A operand: D[col][pair]
                f32x4 c = acc[rb][n];
                c = mfma_bf16_k32(wl, xh, c, 0, 0, 0);
                c = mfma_bf16_k32(wh, xl, c, 0, 0, 0);
                c = mfma_bf16_k32(wm, xm, c, 0, 0, 0);
                c = mfma_bf16_k32(wm, xh, c, 0, 0, 0);
                c = mfma_bf16_k32(wh, xm, c, 0, 0, 0);
                c = mfma_bf16_k32(wh, xh, c, 0, 0, 0);
                acc[rb][n] = c;
            }
        }
        // -- last channel step of a tile: lane (r, q) holds columns 4q .. 4q+3 of pair r of every row block
        if (sc + 1 == ns) {
            const int cbw = cbw_of(tc);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    const int col = 16 * (cbw + n) + 4 * q;
                    const int row = unit_of(tc) * RU + 16 * rb + r;
                    if (cbw + n < ncb && (!DENSE || row < n_rows)) {
                        f32x4 o = acc[rb][n];
                        if (DENSE && bias) o += *reinterpret_cast<const f32x4 *>(bias + col);
                        if (B16) {
                            px_bf16x4 ob;
                            ob[0] = (__bf16)o[0]; ob[1] = (__bf16)o[1]; ob[2] = (__bf16)o[2]; ob[3] = (__bf16)o[3];
                            *reinterpret_cast<px_bf16x4 *>(reinterpret_cast<char *>(y) + ((size_t)row * cout + col) * 2) = ob;
                        } else {
                            *reinterpret_cast<f32x4 *>(y + (size_t)row * cout + col) = o;
                        }
                    }
                    acc[rb][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
        }
        adv(tc, sc);
        // -- rows of step i+1 (gathered at step i-1) into the other image (last read at step i-1, a barrier ago)
        store_G(g[u ^ 1], u ^ 1);
        (void)i;
        lds_barrier();
    };
    for (int i = 0; i < nsteps; i += 2) {
        step(std::integral_constant<int, 0>{}, i);
        if (i + 1 >= nsteps) break;
        step(std::integral_constant<int, 1>{}, i + 1);
    }
}

// (cin, cout) the kernel takes: whole 32-channel steps and fragment-layout weights (multiples of 32)
bool conv_px3_supported(int cin, int cout) { return cin >= 32 && cin % 32 == 0 && cout >= 32 && cout % 32 == 0; }

// column tile: 128 (4 waves x 2 column blocks), 96 (3 waves x 2: the 96- and 192-column layers exactly) or 256 (4 waves x 4:
// layers whose columns are a multiple of 256 -- every row fragment read from LDS feeds twice the MFMAs: 512 -> 512 at stride
// 8 498 -> 405 us, 256 -> 256 at stride 4 198 -> 184 us; 232 VGPRs = 2 workgroups per CU.  The same 4 blocks per wave on TWO waves for the 128-column layers was measured slower: 109 -> 129 us)
static bool px3_wide(int cout) { return cout % 256 == 0; }

// 192 columns per tile (3 waves x 4 blocks) for the 192-column layers: 192 -> 192 at stride 1 234 -> 225 us,
// 256 -> 192 at stride 2 215 -> 204 us
static bool px3_wide3(int cout) { return cout % 192 == 0; }

// Tile shape by layer.  Two 64-pair tiles per step on 128- / 96-column tiles (TL = 2) where the weight fragments dominate
// what a step pulls through the vector L1 -- measured (MI355X, 80k scene, tools/ab_px3.py, TL 1 -> 2): 512 -> 512 at stride 8
// 431 -> 406 us, at stride 16 231 -> 204 us, 512 -> 768 (the input gradient of 768 -> 512) 735 -> 617 us, 256 -> 384 206 ->
// 189 us; the narrower layers lose 3-10 % (256 -> 256 183 -> 189 us, 128 -> 128 85 -> 89 us, 96 -> 96 77 -> 86 us: more rows
// gathered per weight byte saved than their weights cost) and keep one tile per step on the 256- / 192- / 128- / 96-column tile.
struct Px3Shape { bool w3; int nbw, tl, tn; };
static Px3Shape px3_shape(int cin, int cout) {
    Px3Shape p;
    p.w3 = cout % 96 == 0 && cout % 128 != 0;
    p.tl = (cin >= 512 || cout >= 384) ? 2 : 1;
    if (p.tl == 2) p.nbw = 2;
    else p.nbw = (!p.w3 && px3_wide(cout)) || (p.w3 && px3_wide3(cout)) ? 4 : 2;
    p.tn = 16 * (p.w3 ? 3 : 4) * p.nbw;
    return p;
}

template <bool DENSE, bool B16, int SC, bool F2 = false>
static void px3_launch(const Px3Shape &p, dim3 grid, hipStream_t st, const float *in, int cin, const float *wf, int cout,
                       const int32_t *pair_idx, const int32_t *tile_k, const int32_t *n_tiles, float *y, const float *bias,
                       int n_rows, int k_total = 1) {
    const size_t lds = (size_t)2 * 64 * p.tl * 192;
#define U2_PX3(NW_, NBW_, TL_)                                                                                                  \
    hipLaunchKernelGGL((conv_px3_kernel<NW_, NBW_, DENSE, B16, SC, TL_, F2>), grid, dim3(64 * NW_), lds, st, in, cin, wf, cout, pair_idx, \
                       tile_k, n_tiles, y, bias, n_rows, k_total)
    if (p.tl == 2) {
        if (p.w3) U2_PX3(3, 2, 2); else U2_PX3(4, 2, 2);
    } else if (p.nbw == 4) {
        if (p.w3) U2_PX3(3, 4, 1); else U2_PX3(4, 4, 1);
    } else {
        if (p.w3) U2_PX3(3, 2, 1); else U2_PX3(4, 2, 1);
    }
#undef U2_PX3
}

// b16: `in` and `y` are bf16 rows, wf = the arith-3 (one bf16 plane) fragments
int launch_conv_px3(const char *who, const float *in, int cin, const float *wf, int cout, const int32_t *pair_idx,
                    const int32_t *tile_k, const int32_t *n_tiles, int64_t capacity, float *y, hipStream_t st, bool b16,
                    int f16x2_k) {
    if (!conv_px3_supported(cin, cout)) return -1;
    const Px3Shape p = px3_shape(cin, cout);
    // ONE resident wave of workgroups (registers: 2 or 3 per CU), a multiple of 8 = the same number per XCD; fewer when the
    // pair list cannot fill them (every workgroup owns at least one (unit, column tile) item of the capacity)
    const int64_t cap_x = (p.tl == 1 && p.nbw == 2 ? 3 : 2) * 256;
    const int gy = (int)ceil_div(cout, p.tn);
    int64_t gx = std::min<int64_t>(cap_x, ceil_div(capacity / (64 * p.tl) * gy, 8) * 8);
    if (gx < 8) gx = 8;
    dim3 grid((unsigned)gx);
    if (f16x2_k > 0) px3_launch<false, false, 32, true>(p, grid, st, in, cin, wf, cout, pair_idx, tile_k, n_tiles, y, nullptr, 0, f16x2_k);
    else if (!b16) px3_launch<false, false, 32>(p, grid, st, in, cin, wf, cout, pair_idx, tile_k, n_tiles, y, nullptr, 0);
    else if (cin % 64 == 0) px3_launch<false, true, 64>(p, grid, st, in, cin, wf, cout, pair_idx, tile_k, n_tiles, y, nullptr, 0);
    else px3_launch<false, true, 32>(p, grid, st, in, cin, wf, cout, pair_idx, tile_k, n_tiles, y, nullptr, 0);
    return check_launch(who);
}

// y[n_rows, cout] = in[n_rows, cin] x B (+ bias), B in the arith-2 (b16: arith-3) fragment order of ONE offset
int launch_linear_px3(const char *who, const float *in, int64_t n_rows, int cin, const float *wf, int cout,
                      const float *bias, float *y, hipStream_t st, bool b16, bool f16x2) {
    if (!conv_px3_supported(cin, cout)) return -1;
    const Px3Shape p = px3_shape(cin, cout);
    const int gy = (int)ceil_div(cout, p.tn);
    const int64_t cap_x = (p.tl == 1 && p.nbw == 2 ? 3 : 2) * 256;
    int64_t gx = std::min<int64_t>(cap_x, ceil_div(ceil_div(n_rows, 64 * p.tl) * gy, 8) * 8);
    if (gx < 8) gx = 8;
    dim3 grid((unsigned)gx);
    if (f16x2) px3_launch<true, false, 32, true>(p, grid, st, in, cin, wf, cout, nullptr, nullptr, nullptr, y, bias, (int)n_rows, 1);
    else if (!b16) px3_launch<true, false, 32>(p, grid, st, in, cin, wf, cout, nullptr, nullptr, nullptr, y, bias, (int)n_rows);
    else if (cin % 64 == 0) px3_launch<true, true, 64>(p, grid, st, in, cin, wf, cout, nullptr, nullptr, nullptr, y, bias, (int)n_rows);
    else px3_launch<true, true, 32>(p, grid, st, in, cin, wf, cout, nullptr, nullptr, nullptr, y, bias, (int)n_rows);
    return check_launch(who);
}

}  // namespace u2mkd
