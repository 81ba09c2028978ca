// Sparse conv forward / input gradient: the TILE-LOCAL PAIR schedule (gfx950, fp32 MFMA).
//
// Replaces torchsparse v1.4.0 convolution_forward_cuda / the dX half of convolution_backward_cuda
// (gather -> cuBLAS mm -> scatter-add per kernel offset) behind the
// spnn.Conv3d calls of core/models/build_blocks.py:25-80, for layers whose whole reduction
// (cin) fits the register file.
//
// Why: the offset-walking tile kernel (conv_os2_kernel) visits the union of its 64 rows' offsets
// one after the other, every stage behind a barrier, with a wave idle whenever its own 16 rows lack
// the offset -- on LiDAR maps (k-bar ~3.6 of 27) the kernel time is the critical path of the tiles
// made of rare neighbour masks (27 stages x ~4k cycles).  Here a workgroup owns a work item of <= 64
// (mask-sorted) output rows and
//   1. COMPACTS, per kernel offset, the rows that have a neighbour into dense lists in LDS
//      (wave ballot + prefix popcount; one wave per offset, 64 rows per ballot), and flattens them
//      into a list of 16-pair MFMA blocks (wave prefix sum);
//   2. walks the blocks in a software pipeline (details at the loop).  The waves split the output
//      COLUMNS (16*NBW each): every wave does the same work on every block and owns its columns of the
//      LDS-resident output tile exclusively.  The gathered rows of a block are read ONCE by the
//      workgroup, coalesced (16 lanes per row), through a 2-slot LDS image; the MFMA computes
//      D^T = B_k x rows^T (weights as the A operand), which leaves each lane with 4 CONSECUTIVE output
//      columns of ONE pair: the accumulate into the output tile is one ds_read_b128 + ds_write_b128;
//   3. weights come in MFMA-fragment order (u2mkd_weight_fragments), so a wave's fragment load is
//      1 KiB contiguous per instruction.
// Every output row is written once, no atomics, fixed summation order (bitwise reproducible).
// MFMA padding: sum over (tile, offset) of ceil(pairs / 16) blocks = 1.12 x dense on the 80k-voxel
// scene (the 16-row-block walk of conv_os2: 1.29 x).
//
// What was measured on the way (MI355X, 64 -> 64 at 80k voxels, in-kernel s_memtime stamps):
//   * gathering straight into MFMA operand registers (lane (r, q) loads 16 B of row r): the 4 column-split
//     waves repeat the gather and every wave instruction is 64 uncoalesced 16-byte accesses: the texture
//     addresser, not the MFMA pipe, bounds the kernel (83 -> 106 us when the fragments were also re-read
//     per block);
//   * a conditional load inside the pipelined loop makes hipcc's s_waitcnt counters inexact (vmcnt(0)
//     before every use = no prefetch): all loads in the loop are unconditional;
//   * __syncthreads() waits for vmcnt(0): the loop uses an LDS-only barrier;
//   * a tile is a serial chain of blocks (up to 62 at 64 rows, mean 16): tiles above 30 / 60 blocks run as
//     2 / 4 work items (TileSchedule in torchsparse/nn/functional.py), heaviest item first.
#include <stdlib.h>

#include <type_traits>

#include "conv_internal.h"

namespace u2mkd {

#ifndef U2MKD_TP_EXTRA_LDS
#define U2MKD_TP_EXTRA_LDS 0      // (occupancy probes only: tools/build_variant.sh _lds "-DU2MKD_TP_EXTRA_LDS=8192" conv_tp.hip)
#endif
constexpr int kTpPad = 8;             // row pad of the LDS row image (dwords), see AS below
constexpr int kTpRowShift = 25;       // s_idx word of a pair = input row | tile row << 25 (input rows < 2^25, launch_conv_tp checks)

// Weight fragments: B_k = the [ncol x nred] matrix whose row `col` holds the reduction channels of output
// column `col` (forward: B_k[col][ci] = kernel[k][ci][col], transpose = 1; input gradient: B_k[ci][co] =
// kernel[k][ci][co], transpose = 0).  Fragment layout: wf[k][cb][j][lane][4] = B_k[16 cb + r][16 j + 4 q .. +3],
// lane = r + 16 q -- exactly the registers of one MFMA operand fragment, so a wave's load instruction reads
// 1 KiB contiguously.
__global__ void weight_fragments_kernel(const float *__restrict__ w, int rows, int cols, int transpose,
                                        float *__restrict__ wf, int64_t total) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    if (transpose == 2) {      // both orientations in one launch: [transpose = 1 | transpose = 0]
        transpose = 1 - (int)blockIdx.y;
        wf += (size_t)blockIdx.y * total;
    }
    const int ncol = transpose ? cols : rows, nred = transpose ? rows : cols;
    const int c = (int)(t & 3);
    const int lane = (int)((t >> 2) & 63);
    int64_t u = t >> 8;
    const int nj = nred / 16, ncb = ncol / 16;
    const int j = (int)(u % nj);
    u /= nj;
    const int cb = (int)(u % ncb);
    const int64_t k = u / ncb;
    const int col = 16 * cb + (lane & 15), red = 16 * j + 4 * (lane >> 4) + c;
    wf[t] = transpose ? w[((size_t)k * rows + red) * cols + col] : w[((size_t)k * rows + col) * cols + red];
}

// ---- fp32 through three bf16 MFMA operands ("bf16x3") -------------------------------------------
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): the three 8-bit significands tile the
// 24-bit one, both subtractions are exact.  a*b = sum of the nine partial products, each of them exact in fp32;
// the six with (order_a + order_b) <= 2 are kept -- the dropped ones are below 2^-24 |a b|, half an fp32 ulp of the
// product -- and accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (8192 MACs per 16-cycle issue against 1024 per
// 32 cycles of v_mfma_f32_16x16x4_f32: 16x the rate, 6 products -> 2.7x fewer matrix-pipe cycles at fp32
// accuracy).  Result = fp32 GEMM accuracy, not the bitwise fma chain of the f32 MFMA; tests hold it to the same
// gates (tests/test_gpu_torchsparse_ops.py: <= 1e-4 relative on operators, measured ~1e-6; bitwise reproducible).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    float r1 = x - (float)h;
    m = (__bf16)r1;
    float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// Fragment layout, bf16x3: wf3[k][cb][f = 3 s + p][lane][8 bf16] = plane p (0 = h, 1 = m, 2 = l) of
// B_k[16 cb + r][32 s + 8 q .. +7], lane = r + 16 q: the operand fragment of v_mfma_f32_16x16x32_bf16
// (lane l holds A[row l & 15][k = 8 (l >> 4) + j]).  One thread reads 8 reduction channels of one column
// and writes the three 16-byte fragments.
template <int PLANES>
__global__ void weight_fragments_x3_kernel(const float *__restrict__ w, int rows, int cols, int transpose,
                                           bf16x8 *__restrict__ wf, int64_t total) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    if (transpose == 2) {      // both orientations in one launch: [transpose = 1 | transpose = 0]
        transpose = 1 - (int)blockIdx.y;
        wf += (size_t)blockIdx.y * (total * PLANES + (PLANES == 2 ? 1 : 0));     // (f16x2: + the 16-byte scale trailer)
    }
    const int ncol = transpose ? cols : rows, nred = transpose ? rows : cols;
    const int lane = (int)(t & 63);
    int64_t u = t >> 6;
    const int ns = nred / 32, ncb = ncol / 16;
    const int sstep = (int)(u % ns);
    u /= ns;
    const int cb = (int)(u % ncb);
    const int64_t k = u / ncb;
    const int col = 16 * cb + (lane & 15), red0 = 32 * sstep + 8 * (lane >> 4);
    float x[8];
    if (transpose) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = w[((size_t)k * rows + red0 + i) * cols + col];
    } else {
        const float4 *p = reinterpret_cast<const float4 *>(w + ((size_t)k * rows + col) * cols + red0);
        const float4 a = p[0], b = p[1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
    bf16x8 *o = wf + (((size_t)k * ncb + cb) * ns + sstep) * PLANES * 64 + lane;
    if (PLANES == 2) {      // f16x2: h | l planes of w * scale, scale = the tensor's (trailer, weight_absmax_kernel)
        const float sc = reinterpret_cast<const float *>(wf + total * PLANES)[0];
        uint4 vh4, vl4;
        f16x2_split2(x[0] * sc, x[1] * sc, vh4.x, vl4.x);
        f16x2_split2(x[2] * sc, x[3] * sc, vh4.y, vl4.y);
        f16x2_split2(x[4] * sc, x[5] * sc, vh4.z, vl4.z);
        f16x2_split2(x[6] * sc, x[7] * sc, vh4.w, vl4.w);
        o[0] = __builtin_bit_cast(bf16x8, vh4);
        o[64] = __builtin_bit_cast(bf16x8, vl4);
        return;
    }
    bf16x8 vh, vm, vl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 h, m, l;
        split3(x[i], h, m, l);
        vh[i] = h; vm[i] = m; vl[i] = l;
    }
    o[0] = vh;              // PLANES == 1 (bf16 storage): the weights rounded to bf16
    if (PLANES == 3) {
        o[64] = vm;
        o[128] = vl;
    }
}

// f16x2 fragments: the power-of-two scale of a whole weight tensor (largest |w| -> [2^14, 2^15)), written as {scale, 1 / scale, 0, 0}
// into the 16-byte trailer behind the fragments of each orientation.  One workgroup per weight: jobs == nullptr: the one tensor
// (w, elems, trailer0, trailer1); else job blockIdx.x of the batch table (only jobs with planes == 2 do anything).
constexpr int kAbsmaxSlices = 32;
__global__ void __launch_bounds__(1024)
weight_absmax_kernel(const int64_t *__restrict__ jobs, const float *w, int64_t elems, float *t0, float *t1) {
    // jobs == nullptr: ONE workgroup for the one tensor, scale written.  Batch: grid (jobs, kAbsmaxSlices), a weight's elements
    // cut into 32 slices whose maxima go to the first 32 floats of the job's FRAGMENT area (overwritten by the fragment launch
    // that follows; no initialised scratch needed) and are folded by weight_absmax_finish_kernel -- one workgroup per weight
    // read a 27 x 256 x 256 kernel at 30 GB/s: 231 us on the step's chain behind the optimizer.
    int nslice = 1, slice = 0;
    float *part = nullptr;
    if (jobs) {
        const int64_t *jb = jobs + (size_t)blockIdx.x * 8;
        if (jb[6] != 2) return;
        w = reinterpret_cast<const float *>(jb[0]);
        elems = jb[3] * jb[4] * jb[5];
        part = reinterpret_cast<float *>(jb[1]);
        nslice = kAbsmaxSlices;
        slice = blockIdx.y;
    }
    __shared__ float s_m[16];
    float m = 0.f;
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    const int64_t n4 = elems / 4, per = (n4 + nslice - 1) / nslice;          // (elems is a multiple of 1024: rows, cols multiples of 32)
    const int64_t e1 = min(n4, (slice + 1) * per);
    for (int64_t e = slice * per + threadIdx.x; e < e1; e += blockDim.x) {
        const float4 v = w4[e];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, s_m[i]);
        if (part) {
            part[slice] = m;
        } else {
            float sc, inv;
            f16x2_scale(m, sc, inv);
            t0[0] = sc; t0[1] = inv; t0[2] = 0.f; t0[3] = 0.f;
            t1[0] = sc; t1[1] = inv; t1[2] = 0.f; t1[3] = 0.f;
        }
    }
}

__global__ void __launch_bounds__(64)
weight_absmax_finish_kernel(const int64_t *__restrict__ jobs) {
    const int64_t *jb = jobs + (size_t)blockIdx.x * 8;
    if (jb[6] != 2) return;
    const int64_t elems = jb[3] * jb[4] * jb[5];
    char *base = reinterpret_cast<char *>(jb[1]);
    float m = threadIdx.x < kAbsmaxSlices ? reinterpret_cast<const float *>(base)[threadIdx.x] : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (threadIdx.x == 0) {
        float *t0 = reinterpret_cast<float *>(base + (size_t)elems * 4);          // 2 planes x 2 bytes per element
        float *t1 = reinterpret_cast<float *>(base + (size_t)elems * 8 + 16);
        float sc, inv;
        f16x2_scale(m, sc, inv);
        t0[0] = sc; t0[1] = inv; t0[2] = 0.f; t0[3] = 0.f;
        t1[0] = sc; t1[1] = inv; t1[2] = 0.f; t1[3] = 0.f;
    }
}

// The same re-layout for MANY weights in one launch (u2mkd_weight_fragments_batch): a training step changes every
// trainable weight at once (the optimizer step), and one latency-bound launch per weight -- ~5 us each, ~100 per step,
// each in front of the first convolution that needs it -- becomes one launch behind the optimizer.  jobs[j] =
// {w, wf, first unit, k, rows, cols, planes (3 = bf16x3, 1 = one bf16 plane), 0} as int64; a unit = one wave = 64 lanes
// x 8 reduction channels; job j owns units [first_j, first_j+1): 2 orientations x k rows cols / 512, transpose = 1 first
// (the layout of u2mkd_weight_fragments with transpose = 2).
__global__ void __launch_bounds__(64)
weight_fragments_batch_kernel(const int64_t *__restrict__ jobs, int n_jobs) {
    const int lane = threadIdx.x;
    const int64_t unit = blockIdx.x;
    int job = -1;                                            // last job whose first unit is <= unit
    for (int j0 = 0; j0 < n_jobs; j0 += 64) {
        const int j = j0 + lane;
        const bool le = j < n_jobs && jobs[(size_t)j * 8 + 2] <= unit;
        job += __popcll(__ballot(le));
    }
    job = __builtin_amdgcn_readfirstlane(job);
    const int64_t *jb = jobs + (size_t)job * 8;
    const float *w = reinterpret_cast<const float *>(jb[0]);
    bf16x8 *wf = reinterpret_cast<bf16x8 *>(jb[1]);
    const int rows = (int)jb[4], cols = (int)jb[5], planes = (int)jb[6];
    const int64_t per = jb[3] * rows * cols / 512;           // units per orientation
    int64_t u = unit - jb[2];
    const int transpose = u < per ? 1 : 0;
    if (!transpose) { u -= per; wf += (size_t)per * 64 * planes + (planes == 2 ? 1 : 0); }      // (f16x2: + the scale trailer)
    const int ncol = transpose ? cols : rows, nred = transpose ? rows : cols;
    const int ns = nred / 32, ncb = ncol / 16;
    const int sstep = (int)(u % ns);
    u /= ns;
    const int cb = (int)(u % ncb);
    const int64_t k = u / ncb;
    const int col = 16 * cb + (lane & 15), red0 = 32 * sstep + 8 * (lane >> 4);
    float x[8];
    if (transpose) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = w[((size_t)k * rows + red0 + i) * cols + col];
    } else {
        const float4 *p = reinterpret_cast<const float4 *>(w + ((size_t)k * rows + col) * cols + red0);
        const float4 a = p[0], b = p[1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
    bf16x8 *o = wf + (((size_t)k * ncb + cb) * ns + sstep) * planes * 64 + lane;
    if (planes == 2) {      // f16x2 (the scale: weight_absmax_kernel, launched in front of this kernel)
        const float sc = reinterpret_cast<const float *>(wf + (size_t)per * 64 * planes)[0];
        uint4 vh4, vl4;
        f16x2_split2(x[0] * sc, x[1] * sc, vh4.x, vl4.x);
        f16x2_split2(x[2] * sc, x[3] * sc, vh4.y, vl4.y);
        f16x2_split2(x[4] * sc, x[5] * sc, vh4.z, vl4.z);
        f16x2_split2(x[6] * sc, x[7] * sc, vh4.w, vl4.w);
        o[0] = __builtin_bit_cast(bf16x8, vh4);
        o[64] = __builtin_bit_cast(bf16x8, vl4);
        return;
    }
    bf16x8 vh, vm, vl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 h, m, l;
        split3(x[i], h, m, l);
        vh[i] = h; vm[i] = m; vl[i] = l;
    }
    o[0] = vh;
    if (planes == 3) {
        o[64] = vm;
        o[128] = vl;
    }
}

__device__ __forceinline__ bf16x8 as_bf8(const float4 &x) {
    f32x4 v = (f32x4){x.x, x.y, x.z, x.w};
    return __builtin_bit_cast(bf16x8, v);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int NW, int NBW, int CIN, bool STAMP, int AR, bool SB = true>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu((CIN * NBW <= 64 && (NW >= 3 || CIN <= 32)) ? 4 : 1)))
conv_tp_kernel(const float *__restrict__ in, const float *__restrict__ wf, int cout, const int32_t *__restrict__ nbr,
               const int32_t *__restrict__ order, RowRange rr_, const int32_t *__restrict__ items,
               const int32_t *__restrict__ n_items_dev, int n_tiles, int K, int kflip, float *__restrict__ out,
               unsigned long long *__restrict__ stamps) {
    // STAMP (tools/stamps_tp.py only): per workgroup {realtime start, cycles start, after compaction, after the block
    // walk, end, realtime end, blocks, tile}; the product instantiation has none of it
    unsigned long long t_rt0 = 0, t_c0 = 0, t_c1 = 0, t_c2 = 0;
    int n_blocks = 0;
    if (STAMP) { t_rt0 = __builtin_amdgcn_s_memrealtime(); t_c0 = __builtin_amdgcn_s_memtime(); }
    // AR: 1 = fp32 rows, f32 MFMA; 2 = fp32 rows, bf16x3; 3 = BF16 STORAGE: `in` and `out` are bf16 rows, wf = one bf16
    // plane (u2mkd_weight_fragments arith 3), one v_mfma_f32_16x16x32_bf16 per 32-channel step, fp32 accumulation in the
    // LDS tile, outputs rounded to bf16 once (BASELINE.json configs[4]: half the gather bytes, no run-time split)
    // 4 = fp32 rows, f16x2: two fp16 planes of every gathered row scaled by a per-row power of two, three products per 16 channels
    // (conv_internal.h), the row's and the weight tensor's scales taken out again when the block's products join the output tile
    constexpr bool X3 = AR == 2, B16 = AR == 3, F2 = AR == 4;
    constexpr int T = 64, NT = 64 * NW, TN = 16 * NW * NBW, NJ = CIN / 16;
    constexpr int OS = TN + 4;                    // output tile row stride (floats)
    // gathered-row image: fp32 rows, or (X3) three bf16 planes h | m | l of CIN elements per row; + 16 B pad
    // row stride (dwords) = data + 8: lane (r, q) of a ds_read_b128 reads dwords AS r + 4 q .. +3 (+ the fragment's offset), the
    // hardware serves the 16-lane groups {r in 0-3 | 12-15 at q, r in 4-11 at q + 1} and {the complement} in one cycle each only
    // if their 16 x 4 dwords hit 64 different banks: AS / 4 even and = 2 (mod 4).  (Rounds 2-3 ran data + 4: every fragment read
    // was a 2-way conflict, 96 of the ~400 LDS cycles of a block.)
    constexpr int AS = (X3 ? 6 * CIN / 4 : B16 ? 2 * CIN / 4 : CIN) + kTpPad;
    constexpr int NF = X3 ? CIN / 32 * 3 : F2 ? CIN / 32 * 2 : B16 ? CIN / 32 : CIN / 16;   // 16-byte operand fragments per lane per block
    constexpr int CPR = B16 ? CIN / 8 : CIN / 4;  // 16-byte chunks per gathered row
    constexpr int LPT = (16 * CPR + NT - 1) / NT; // chunks a thread moves per block
    constexpr int KPW = (32 + NW - 1) / NW;       // offsets a wave compacts (K <= 32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_out = reinterpret_cast<float *>(smem);                           // [T][OS]
    float *s_a = s_out + T * OS;                                              // [2][16][AS] gathered rows of blocks t, t+1
    int *s_idx = reinterpret_cast<int *>(s_a + 2 * 16 * AS);                  // [K+1][T] compacted (input row | tile row << 25) (+ sentinel row)
    int *s_cnt = s_idx + (K + 1) * T;                                         // [32] pairs per offset
    int *s_rid = s_cnt + 32;                                                  // [T] original output row
    int *s_blk = s_rid + T;                                                   // [4K + 8] block descriptors
    float *s_sc = reinterpret_cast<float *>(s_blk + 4 * K + 8);               // (F2) [2][16] 1 / (row scale x weight scale) of the row image's rows
    // (F2) 1 / scale of the weight tensor: the trailer behind this orientation's fragments
    const float wsc = F2 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(wf) + (size_t)K * cout * CIN * 4)[1] : 1.f;

    const int col0 = blockIdx.y * TN;
    const int64_t n_out = rr_.end, ld = rr_.ld;
    // Work items: a 64-row tile or, for tiles with many blocks, a half / a quarter of one (a tile is a serial
    // chain of blocks: the heaviest tiles would set the kernel time).  item = tile << 4 | sub << 2 | lg with
    // 64 >> lg rows starting at row 64 tile + (64 >> lg) sub, listed heaviest first.
    // The grid is at most ONE resident wave of workgroups (launch_conv_tp: workgroups per CU x CUs); with more
    // items than workgroups a workgroup takes several, dealt boustrophedon over the heaviest-first list (round r
    // gives workgroup b item r G + b, or r G + G-1-b when r is odd), and runs its LIGHTEST item first.  Measured
    // with one workgroup per item (MI355X, 64 -> 64, 80k voxels, 1415 items on 1024 slots): the 1024 resident
    // items all end at 25-30 us whatever their block count (the CU's issue slots go oldest wave first: 1.6k
    // cycles per block for the heaviest item of a CU, 4.8k for the lightest, 600 cycles per block per CU in
    // total), and the 391 light items (<= 4 blocks) then run as a second wave that is pure latency -- 3 us of
    // set-up + 5 us of pipeline fill each on a nearly empty chip: 9 of the kernel's 38 us.  A light item at the
    // START of a young workgroup costs nothing: it waits for memory while the older workgroups use the CU.
    const int n_it = items ? *n_items_dev : n_tiles;
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    const int n_rd = (n_it + G - 1) / G;
    // this workgroup's next item at or after round rq (-1: none)
    auto next_item = [&](int &rq) __attribute__((always_inline)) {
        for (; rq < n_rd; ++rq) {
            const int rd = n_rd - 1 - rq;             // (the workgroup's lightest item first)
            const int t = rd * G + ((rd & 1) ? G - 1 - wg : wg);
            if (t < n_it) return t;
        }
        return -1;
    };
    auto item_code = [&](int t) __attribute__((always_inline)) { return items ? items[t] : (t << 4); };
    // neighbour indices of an item's rows: one burst of independent loads, KPW per thread
    auto load_nbr = [&](int code_, int lane_, int wave_, int (&vv)[KPW]) __attribute__((always_inline)) {
        const int R_ = T >> (code_ & 3);
        const int64_t row = rr_.begin + (int64_t)(code_ >> 4) * T + ((code_ >> 2) & 3) * R_ + lane_;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int kk = wave_ + i * NW;
            vv[i] = (kk < K && lane_ < R_ && row < n_out) ? nbr[(int64_t)kk * ld + row] : -1;
        }
    };
    int rq = 0;
    int it = next_item(rq);
    if (it < 0) return;
    int code = item_code(it);
    constexpr bool PF = CIN * NBW <= 64;      // (the 8 extra registers cost the wide instantiations a wave per SIMD)
    int v[KPW];
    if (PF) {
        int t0_ = threadIdx.x;
        asm volatile("" : "+v"(t0_));
        load_nbr(code, t0_ & 63, __builtin_amdgcn_readfirstlane(t0_ >> 6), v);
    }
#pragma nounroll
    while (true) {
    // the item after this one: its code now (a scalar load), its neighbour indices once this item's compaction is
    // done -- they arrive during the block walk, so the next set-up starts from registers instead of a memory
    // round trip (set-up + pipeline fill of an item are ~4 us of pure latency for a workgroup with two items)
    int rq_n = rq + 1;
    const int it_n = next_item(rq_n);
    const int code_n = it_n >= 0 ? item_code(it_n) : 0;
    // the lane index is re-derived behind an opaque asm in every round: otherwise the compiler hoists every
    // lane-dependent address out of the item loop and keeps it live (180 VGPRs instead of 118: 2 waves per SIMD)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int R = T >> (code & 3);                        // rows of this item
    const int64_t row0 = rr_.begin + (int64_t)(code >> 4) * T + ((code >> 2) & 3) * R;
    if (STAMP) { t_rt0 = __builtin_amdgcn_s_memrealtime(); t_c0 = __builtin_amdgcn_s_memtime(); n_blocks = 0; }

    // ---- 1. compaction per offset of the tile's neighbour indices (in v: loaded before the loop / during the
    // previous item's walk)
    if (!PF) load_nbr(code, lane, wave, v);
    for (int e = tid; e < T * OS / 4; e += NT) reinterpret_cast<float4 *>(s_out)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = tid; e < T; e += NT) {
        int64_t row = row0 + e;
        s_rid[e] = (e < R && row < n_out) ? (order ? order[row] : (int)row) : -1;
    }
    if (tid >= K && tid < 32) s_cnt[tid] = 0;            // (entries < K are written by the compaction below)
    if (tid < T) s_idx[K * T + tid] = -1;
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        int kk = wave + i * NW;
        if (kk < K) {
            const bool valid = v[i] >= 0;
            const unsigned long long bal = __ballot(valid);
            const int cnt = __popcll(bal);
            const int rank = __popcll(bal & ((1ULL << lane) - 1ULL));
            if (valid) s_idx[kk * T + rank] = v[i] | (lane << kTpRowShift);
            if (lane >= cnt && lane < ((cnt + 15) & ~15)) s_idx[kk * T + lane] = -1;   // pad the last block
            if (lane == 0) s_cnt[kk] = cnt;
        }
    }
    __syncthreads();
    int vn[PF ? KPW : 1];
    if (PF && it_n >= 0) load_nbr(code_n, lane, wave, reinterpret_cast<int (&)[KPW]>(vn));

    if (STAMP) t_c1 = __builtin_amdgcn_s_memtime();
    // ---- 2. the tile's (offset, 16-pair block) sequence as a flat list: descriptor = offset << 8 | block.
    // Every wave writes the same values (a wave reads back its own writes, no barrier needed); the list is
    // followed by sentinel blocks (offset K: an all -1 index row) so the pipeline can run ahead of the end
    // without conditionals.
    const int myc = lane < 32 ? s_cnt[lane] : 0;
    int total;
    {
        const int nbk = (myc + 15) >> 4;
        int incl = nbk;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        total = __builtin_amdgcn_readlane(incl, 31);
        const int start = incl - nbk;
        if (lane < 32)
            for (int b = 0; b < nbk; ++b) s_blk[start + b] = (lane << 8) | b;
        if (lane < 8) s_blk[total + lane] = K << 8;          // desc(t + 6) is read at the last step
    }

    // ---- 3. software pipeline over the blocks.  Block b's data moves through four steps, all waves in
    // lockstep (one barrier per step; every wave does the same work in every step):
    //   step b-3  gather issue: 16 rows x CIN floats, one 16-byte chunk per thread, 16 consecutive lanes read
    //             one row's contiguous bytes (a gather straight into MFMA operand registers is 64 uncoalesced
    //             16-byte accesses per wave instruction -- measured texture-addresser-bound -- and repeats the
    //             gather in every column-split wave);
    //   step b-1  the rows are stored into one half of the LDS row image (2 slots: the slot stored at the end of
    //             step t was last read at the top of step t-1, one barrier earlier, and is read at step t+1, one
    //             barrier later; a third slot and a store one step earlier cost 6.4 KB of LDS = the fourth
    //             workgroup per CU: 42.9 -> 39.9 us at 64 -> 64);
    //   step b-1  this wave's weight fragment of block b's offset is issued (1 KiB contiguous per instruction; re-loading an unchanged
    //             offset is an L1 hit -- loads stay UNCONDITIONAL so that the compiler's s_waitcnt counters
    //             stay exact: with conditional loads in the loop they degrade to vmcnt(0) before every use);
    //   step b-1  the pair's (input row, output row) are read from LDS;
    //   step b    the MFMA operand fragments are read from LDS, multiplied, accumulated into the output tile.
    // Every LDS / global read a step needs was issued at least one step earlier, so a step is its 16*NBW*NJ/4
    // MFMAs + one barrier (in-kernel stamps of the straightforward order: 2.0k cycles per step, 0.5k of them MFMA).
    // The MFMA computes D^T = B_k (A operand) x rows^T (B operand): a lane ends up with 4 consecutive output
    // columns of ONE pair.  Lanes of a padded last block (idx < 0) multiply row 0 and drop the result.
    // weight fragments: two register sets filled one step ahead (every step re-loads, also an unchanged offset), or
    // (SB) ONE set re-loaded only when the next block's offset differs (64 % of the steps repeat the offset on the
    // 80k scene: 24 KB of the 28 KB a step pulls through the texture path are weights)
    float4 bw[SB ? 1 : 2][NF][NBW], a[NF];
    int kcur = -1;
    f32x4 g[3][LPT];      // (native vector type: the HIP float4 struct kept this ring in scratch memory)
    int gix[LPT];                                  // gather row of this thread's chunk(s), block t+3
    int pidx[2], prow[2];                          // (input row, output tile row) of pair r, blocks t / t+1
    const int kf = kflip & 1;
    const int ncb = cout / 16;
    const int cb0 = col0 / 16 + NBW * wave;

    auto desc = [&](int t) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(s_blk[t]); };
    auto dbase = [&](int d) __attribute__((always_inline)) { return (d >> 8) * T + 16 * (d & 255); };
    // (Tried, MI355X: switching the fragment load of block t+1 off when its offset equals block t's -- all lanes
    // reading one address, registers kept by a select -- to spare the texture addresser: no gain at 64 -> 64
    // (45.0 vs 42.9 us), a loss at 64 -> 128.  Buffer loads with an out-of-range offset cannot be used: hipcc 7.2
    // narrows element reads of __builtin_amdgcn_raw_buffer_load_b128 into dword loads at the wrong offset.)
    auto issue_B = [&](int d, float4 (&bb)[NF][NBW]) __attribute__((always_inline)) {
        int k = d >> 8;
        if (k >= K) k = K - 1;                        // sentinel block: any valid fragment
        if (kf) k = K - 1 - k;
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
            int cb = cb0 + n;
            if (cb >= ncb) cb = ncb - 1;              // column blocks past cout: computed, never stored
            const float *pb = wf + ((((size_t)k * ncb + cb) * NF) * 64 + lane) * 4;
#pragma unroll
            for (int j = 0; j < NF; ++j) bb[j][n] = *reinterpret_cast<const float4 *>(pb + (size_t)j * 256);
        }
    };
    auto read_gix = [&](int d) __attribute__((always_inline)) {
        const int base = dbase(d);
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int e = tid + i * NT;
            gix[i] = e < 16 * CPR ? s_idx[base + e / CPR] : 0;
        }
    };
    auto issue_G = [&](f32x4 (&gg)[LPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int e = tid + i * NT;
            const int ch = e % CPR;
            const int ix = gix[i] >= 0 ? (gix[i] & ((1 << kTpRowShift) - 1)) : 0;
            gg[i] = *reinterpret_cast<const f32x4 *>(in + (B16 ? (size_t)ix * (CIN / 2) : (size_t)ix * CIN) + 4 * ch);
        }
    };
    auto store_G = [&](const f32x4 (&gg)[LPT], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int e = tid + i * NT;
            const int pr = e / CPR, ch = e - pr * CPR;
            float rs = 1.f, rinv = 1.f;
            if (F2) {      // the row's largest |x| over its CPR lanes (all lanes of the wave take part: 16 * CPR is a multiple of 64)
                float m = fmaxf(fmaxf(fabsf(gg[i][0]), fabsf(gg[i][1])), fmaxf(fabsf(gg[i][2]), fabsf(gg[i][3])));
                m = fmaxf(m, dpp_f<0xB1>(m));                  // quad_perm [1, 0, 3, 2]
                m = fmaxf(m, dpp_f<0x4E>(m));                  // quad_perm [2, 3, 0, 1]
                m = fmaxf(m, dpp_f<0x141>(m));                 // row_half_mirror: the other quad of the 8
                if (CPR >= 16) m = fmaxf(m, dpp_f<0x140>(m));  // row_mirror: the other half of the 16
                if (CPR >= 32) m = fmaxf(m, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(m), 0x401F)));   // lane ^ 16
                f16x2_scale(m, rs, rinv);
            }
            if (e < 16 * CPR) {
                if (F2) {      // the two fp16 planes of the scaled row: h | l, CIN elements each
                    uint32_t h01, l01, h23, l23;
                    f16x2_split2(gg[i][0] * rs, gg[i][1] * rs, h01, l01);
                    f16x2_split2(gg[i][2] * rs, gg[i][3] * rs, h23, l23);
                    char *row = reinterpret_cast<char *>(s_a + (slot * 16 + pr) * AS) + 8 * ch;
                    *reinterpret_cast<uint2 *>(row) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2 *>(row + 2 * CIN) = make_uint2(l01, l23);
                    if (ch == 0) s_sc[slot * 16 + pr] = rinv * wsc;
                } else if (X3) {      // split the 4 channels into the three bf16 planes of the row image
                    // by TRUNCATION (x & 0xffff0000, exact residuals: h + m + l = x as with the rounding split of the
                    // weights, fewer instructions: 2 and + 2 sub per element, one byte-permute per two elements and plane)
                    uint32_t hb[4], mb[4], lb[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = gg[i][c];
                        hb[c] = __float_as_uint(x) & 0xffff0000u;
                        const float r1 = x - __uint_as_float(hb[c]);
                        mb[c] = __float_as_uint(r1) & 0xffff0000u;
                        lb[c] = __float_as_uint(r1 - __uint_as_float(mb[c]));
                    }
                    char *row = reinterpret_cast<char *>(s_a + (slot * 16 + pr) * AS) + 8 * ch;
                    *reinterpret_cast<uint2 *>(row) = make_uint2(__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u),
                                                                 __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u));
                    *reinterpret_cast<uint2 *>(row + 2 * CIN) = make_uint2(__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u),
                                                                           __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u));
                    *reinterpret_cast<uint2 *>(row + 4 * CIN) = make_uint2(__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u),
                                                                           __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u));
                } else {   // fp32 rows, or bf16 rows as they are: 16 bytes = 4 floats / 8 bf16 channels
                    *reinterpret_cast<f32x4 *>(s_a + (slot * 16 + pr) * AS + 4 * ch) = gg[i];
                }
            }
        }
    };
    auto read_frag = [&](int slot, float4 (&aa)[NF]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (X3)    // fragment j = 3 s + p: 8 consecutive channels 32 s + 8 q .. of plane p
                aa[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_a + (slot * 16 + r) * AS) +
                                                          (j % 3) * 2 * CIN + 64 * (j / 3) + 16 * q);
            else if (F2)   // fragment j = 2 s + p
                aa[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_a + (slot * 16 + r) * AS) +
                                                          (j % 2) * 2 * CIN + 64 * (j / 2) + 16 * q);
            else if (B16)   // fragment j: 8 consecutive bf16 channels 32 j + 8 q ..
                aa[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_a + (slot * 16 + r) * AS) + 64 * j + 16 * q);
            else
                aa[j] = *reinterpret_cast<const float4 *>(s_a + (slot * 16 + r) * AS + 16 * j + 4 * q);
        }
    };

    if (total > 0) {
        int d1 = desc(1), d2 = desc(2), d4 = desc(4);
        {
            const int d0 = desc(0);
            read_gix(d0); issue_G(g[0]);
            read_gix(d1); issue_G(g[1]);
            read_gix(d2); issue_G(g[2]);
            issue_B(d0, bw[0]);
            kcur = d0 >> 8;
            store_G(g[0], 0);
            read_gix(desc(3));
            pidx[0] = s_idx[dbase(d0) + r];
            prow[0] = (pidx[0] >> kTpRowShift) & 63;
            __syncthreads();
        }
        // one step, `u` = t mod 6 as a compile-time constant (ring positions are static register names)
        auto step = [&](auto U, int t) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
                // -- issue everything later steps need
            read_frag(u & 1, a);                                     // block t (stored at step t-1, barrier since)
            const float osc = F2 ? s_sc[(u & 1) * 16 + r] : 1.f;
            if (!SB) issue_B(d1, bw[SB ? 0 : ((u + 1) & 1)]);       // block t+1
            issue_G(g[u % 3]);                                       // block t+3 (block t left these registers at step t-1)
            read_gix(d4);                                            // block t+4
            pidx[(u + 1) & 1] = s_idx[dbase(d1) + r];                // block t+1
            prow[(u + 1) & 1] = (pidx[(u + 1) & 1] >> kTpRowShift) & 63;
            const int d5 = desc(t + 5);
            // -- block t: the old output values first (in flight under the MFMAs), then the products
            const bool live = pidx[u & 1] >= 0;
            float4 *po[NBW];
            float4 o[NBW];
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                po[n] = reinterpret_cast<float4 *>(s_out + prow[u & 1] * OS + 16 * (NBW * wave + n) + 4 * q);
                o[n] = *po[n];
            }
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (X3) {
                    // the six partial products per 32-channel step, low order first; weights = A operand
#pragma unroll
                    for (int sk = 0; sk < CIN / 32; ++sk) {
                        const bf16x8 wh = as_bf8(bw[SB ? 0 : (u & 1)][3 * sk][n]), wm = as_bf8(bw[SB ? 0 : (u & 1)][3 * sk + 1][n]),
                                     wl = as_bf8(bw[SB ? 0 : (u & 1)][3 * sk + 2][n]);
                        const bf16x8 xh = as_bf8(a[3 * sk]), xm = as_bf8(a[3 * sk + 1]), xl = as_bf8(a[3 * sk + 2]);
#if U2MKD_MFMA_GFX950_K32
                        acc0 = mfma_bf16_k32(wl, xh, acc0, 0, 0, 0);
                        acc1 = mfma_bf16_k32(wh, xl, acc1, 0, 0, 0);
                        acc0 = mfma_bf16_k32(wm, xm, acc0, 0, 0, 0);
                        acc1 = mfma_bf16_k32(wm, xh, acc1, 0, 0, 0);
                        acc0 = mfma_bf16_k32(wh, xm, acc0, 0, 0, 0);
                        acc1 = mfma_bf16_k32(wh, xh, acc1, 0, 0, 0);
#else
                        // the gfx942 form: twelve K = 16 instructions, the two accumulation chains alternating
                        acc0 = mfma_bf16_k32_half<0>(wl, xh, acc0);
                        acc1 = mfma_bf16_k32_half<0>(wh, xl, acc1);
                        acc0 = mfma_bf16_k32_half<1>(wl, xh, acc0);
                        acc1 = mfma_bf16_k32_half<1>(wh, xl, acc1);
                        acc0 = mfma_bf16_k32_half<0>(wm, xm, acc0);
                        acc1 = mfma_bf16_k32_half<0>(wm, xh, acc1);
                        acc0 = mfma_bf16_k32_half<1>(wm, xm, acc0);
                        acc1 = mfma_bf16_k32_half<1>(wm, xh, acc1);
                        acc0 = mfma_bf16_k32_half<0>(wh, xm, acc0);
                        acc1 = mfma_bf16_k32_half<0>(wh, xh, acc1);
                        acc0 = mfma_bf16_k32_half<1>(wh, xm, acc0);
                        acc1 = mfma_bf16_k32_half<1>(wh, xh, acc1);
#endif
                    }
                } else if (F2) {
                    // three partial products per 32-channel step (hl, lh, hh: low order first), six K = 16 instructions on
                    // two alternating accumulation chains
#pragma unroll
                    for (int sk = 0; sk < CIN / 32; ++sk) {
                        const float4 wh = bw[SB ? 0 : (u & 1)][2 * sk][n], wl = bw[SB ? 0 : (u & 1)][2 * sk + 1][n];
                        const float4 xh = a[2 * sk], xl = a[2 * sk + 1];
                        acc0 = mfma_f16_k32_half<0>(wl, xh, acc0);
                        acc1 = mfma_f16_k32_half<0>(wh, xl, acc1);
                        acc0 = mfma_f16_k32_half<1>(wl, xh, acc0);
                        acc1 = mfma_f16_k32_half<1>(wh, xl, acc1);
                        acc0 = mfma_f16_k32_half<0>(wh, xh, acc0);
                        acc1 = mfma_f16_k32_half<1>(wh, xh, acc1);
                    }
                } else if (B16) {
#pragma unroll
                    for (int sk = 0; sk < CIN / 32; ++sk) {
                        if (sk & 1) acc1 = mfma_bf16_k32(as_bf8(bw[SB ? 0 : (u & 1)][sk][n]), as_bf8(a[sk]), acc1, 0, 0, 0);
                        else acc0 = mfma_bf16_k32(as_bf8(bw[SB ? 0 : (u & 1)][sk][n]), as_bf8(a[sk]), acc0, 0, 0, 0);
                    }
                } else {
                    // two interleaved accumulation chains (16x16x4 f32: 40-cycle dependent latency, 32 issue)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[SB ? 0 : (u & 1)][j][n].x, a[j].x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[SB ? 0 : (u & 1)][j][n].y, a[j].y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[SB ? 0 : (u & 1)][j][n].z, a[j].z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[SB ? 0 : (u & 1)][j][n].w, a[j].w, acc1, 0, 0, 0);
                    }
                }
                // D^T: lane (r, q) holds columns 4q .. 4q+3 of pair r
                // (the opaque use keeps the read of the old values where it was written, in flight under the MFMAs: its only
                // other use is inside the branch, and hipcc sinks it there -- an LDS round trip between the last MFMA and
                // the store, every step)
                asm volatile("" : "+v"(o[n].x), "+v"(o[n].y), "+v"(o[n].z), "+v"(o[n].w));
                if (live) {
                    if (F2) {      // the row's and the weights' scales out again (a power of two: exact)
                        o[n].x = fmaf(acc0[0] + acc1[0], osc, o[n].x);
                        o[n].y = fmaf(acc0[1] + acc1[1], osc, o[n].y);
                        o[n].z = fmaf(acc0[2] + acc1[2], osc, o[n].z);
                        o[n].w = fmaf(acc0[3] + acc1[3], osc, o[n].w);
                    } else {
                        o[n].x += acc0[0] + acc1[0];
                        o[n].y += acc0[1] + acc1[1];
                        o[n].z += acc0[2] + acc1[2];
                        o[n].w += acc0[3] + acc1[3];
                    }
                    *po[n] = o[n];
                }
            }
            if (SB && (d1 >> 8) != kcur) {                           // block t+1 starts another offset: its fragments now,
                issue_B(d1, bw[0]);                                  // after the last MFMA that reads the old ones
                kcur = d1 >> 8;
            }
            store_G(g[(u + 1) % 3], (u + 1) & 1);                    // block t+1 (gathered at step t-2); the slot was last read at step t-1
            d1 = d2;
            d2 = desc(t + 3);
            d4 = d5;
            if (STAMP) ++n_blocks;
            // LDS-only barrier: __syncthreads() also waits for vmcnt(0), i.e. it would drain the gathers and
            // weight fragments in flight for the next steps at every step
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        for (int t0 = 0; t0 < total; t0 += 6) {
            step(std::integral_constant<int, 0>{}, t0);
            if (t0 + 1 >= total) break;
            step(std::integral_constant<int, 1>{}, t0 + 1);
            if (t0 + 2 >= total) break;
            step(std::integral_constant<int, 2>{}, t0 + 2);
            if (t0 + 3 >= total) break;
            step(std::integral_constant<int, 3>{}, t0 + 3);
            if (t0 + 4 >= total) break;
            step(std::integral_constant<int, 4>{}, t0 + 4);
            if (t0 + 5 >= total) break;
            step(std::integral_constant<int, 5>{}, t0 + 5);
        }
    }
    if (STAMP) t_c2 = __builtin_amdgcn_s_memtime();
    __syncthreads();

    // ---- 4. epilogue: whole rows of the tile, 16-byte coalesced, to their original positions
    constexpr int F4 = TN / 4;
    for (int e = tid; e < T * F4; e += NT) {
        const int row = e / F4, c4 = e - row * F4;
        const int rid = s_rid[row];
        const int col = col0 + 4 * c4;
        if (rid >= 0 && col < cout) {
            float4 v = *reinterpret_cast<const float4 *>(s_out + row * OS + 4 * c4);
            if (!B16 && rr_.ep_scale) {      // folded eval-mode BatchNorm (+ residual, + ReLU): fp32 rows only
                const float4 sc = *reinterpret_cast<const float4 *>(rr_.ep_scale + col), sh = *reinterpret_cast<const float4 *>(rr_.ep_shift + col);
                v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                if (rr_.ep_res) {
                    const float4 r = *reinterpret_cast<const float4 *>(rr_.ep_res + (size_t)rid * cout + col);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (rr_.ep_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            if (B16) {
                bf16x4 b;
                b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
                *reinterpret_cast<bf16x4 *>(reinterpret_cast<char *>(out) + ((size_t)rid * cout + col) * 2) = b;
            } else {
                *reinterpret_cast<float4 *>(out + (size_t)rid * cout + col) = v;
            }
        }
    }
    if (STAMP && tid == 0) {
        unsigned long long *o = stamps + (size_t)it * 8;
        o[0] = t_rt0; o[1] = t_c0; o[2] = t_c1; o[3] = t_c2; o[4] = __builtin_amdgcn_s_memtime();
        o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = (unsigned long long)n_blocks; o[7] = (unsigned long long)code;
    }
    __syncthreads();      // the next item re-initialises the LDS tile
    if (it_n < 0) break;
    rq = rq_n; it = it_n; code = code_n;
    if (PF) {
#pragma unroll
        for (int i = 0; i < KPW; ++i) v[i] = vn[PF ? i : 0];
    }
    }
}

static int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    return cus;
}

template <int NW, int NBW, int CIN, bool STAMP = false, int AR = 1, bool SB = true>
static void launch_tp(dim3 grid, int K, hipStream_t st, const float *in, const float *wt, int cout, const int32_t *nbr,
                      const int32_t *order, RowRange rr, const int32_t *items, const int32_t *n_items, int kflip,
                      float *out, unsigned long long *stamps = nullptr) {
    constexpr int TN = 16 * NW * NBW;
    const size_t lds = (size_t)64 * (TN + 4) * 4 + (size_t)2 * 16 * ((AR == 2 ? 6 * CIN : AR == 3 ? 2 * CIN : 4 * CIN) + 4 * kTpPad) + (size_t)(K + 1) * 64 * 4 + 32 * 4 + 64 * 4 +
                       (size_t)(4 * K + 8) * 4 + (AR == 4 ? 2 * 16 * 4 : 0) + U2MKD_TP_EXTRA_LDS;
    const int n_tiles = (int)ceil_div(rr.end - rr.begin, 64);
    // one resident wave of workgroups at most (they deal the items among themselves, lightest first)
    static int occ_by_k[33];                         // resident workgroups per CU of THIS instantiation at kernel volume K
    if (occ_by_k[K] == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_tp_kernel<NW, NBW, CIN, STAMP, AR, SB>, 64 * NW, lds) != hipSuccess || n <= 0) n = 1;
        occ_by_k[K] = n;
    }
    const int per_cu = occ_by_k[K];
    const unsigned slots = (unsigned)(per_cu * device_cus());
    if (grid.x > slots) grid.x = slots;
    hipLaunchKernelGGL((conv_tp_kernel<NW, NBW, CIN, STAMP, AR, SB>), grid, dim3(64 * NW), lds, st, in, wt, cout, nbr, order, rr,
                       items, n_items, n_tiles, K, kflip, out, stamps);
}

// column split: cout -> (waves, 16-column blocks per wave, columns per workgroup)
static bool tp_split(int cout, int &nw, int &nbw) {
    if (cout % 16 != 0) return false;
    switch (cout) {
        case 32: nw = 2; nbw = 1; return true;
        case 64: nw = 4; nbw = 1; return true;
        case 96: nw = 3; nbw = 2; return true;
        case 128: nw = 4; nbw = 2; return true;
        default: return false;
    }
}

bool conv_tp_supported(int cin, int cout, int k) {
    int nw, nbw;
    if (k < 1 || k > 32 || !tp_split(cout, nw, nbw)) return false;
    if (!(cin == 32 || cin == 64 || cin == 96 || cin == 128)) return false;
    return cin * nbw <= 128;      // weight fragment (double-buffered) + gathered rows stay in registers
}

// f16x2 (arith 4): a gathered row's 16-byte chunks must fill whole 16-lane groups (the in-register row maximum)
bool conv_tp_f16x2_supported(int cin) { return cin == 32 || cin == 64 || cin == 128; }

// arithmetic of the tile-pair kernel: 1 = f32 MFMA (bitwise fma chain), 2 = bf16x3 (fp32 accuracy, 2.7x fewer
// matrix-pipe cycles); 0 = the library default (U2MKD_CONV_ARITH=f32|bf16x3, default bf16x3)
int conv_tp_arith(int arith) {
    if (arith >= 1 && arith <= 4) return arith;
    static const int dflt = [] {
        const char *e = getenv("U2MKD_CONV_ARITH");
        return (e && e[0] == 'f' && e[1] == '3') ? 1 : 2;      // ("f16x2" concerns the tile kernel only: u2mkd_conv_tiles_arith)
    }();
    return dflt;
}

int launch_conv_tp(const char *who, const float *in, int cin, const float *wf, int cout, const int32_t *nbr,
                   const int32_t *order, RowRange rr, const int32_t *items, const int32_t *n_items, int k, int kflip,
                   int arith, float *out, hipStream_t st, unsigned long long *stamps) {
    if (!conv_tp_supported(cin, cout, k)) return -1;
    int nw = 0, nbw = 0;
    tp_split(cout, nw, nbw);
    const int ar = conv_tp_arith(arith);
    if (ar == 4 && !conv_tp_f16x2_supported(cin)) return -1;
    const bool x3 = ar == 2;
    const int64_t n_rows = rr.end - rr.begin;
    dim3 grid((unsigned)(ceil_div(n_rows, 64) * (items ? 4 : 1)), 1);     // <= 4 items per 64-row tile; launch_tp clamps it
#define U2_TP(NW_, NBW_, CIN_)                                                                                          \
    do {                                                                                                                \
        if (ar == 3) launch_tp<NW_, NBW_, CIN_, false, 3>(grid, k, st, in, wf, cout, nbr, order, rr, items, n_items, kflip, out);  \
        else if (ar == 4 && CIN_ != 96) launch_tp<NW_, NBW_, (CIN_ != 96 ? CIN_ : 64), false, 4>(grid, k, st, in, wf, cout, nbr, order, rr, items, n_items, kflip, out);  \
        else if (x3) launch_tp<NW_, NBW_, CIN_, false, 2>(grid, k, st, in, wf, cout, nbr, order, rr, items, n_items, kflip, out);  \
        else launch_tp<NW_, NBW_, CIN_, false, 1>(grid, k, st, in, wf, cout, nbr, order, rr, items, n_items, kflip, out);    \
    } while (0)
    if (nw == 2) {
        if (cin == 32) U2_TP(2, 1, 32); else if (cin == 64) U2_TP(2, 1, 64);
        else if (cin == 96) U2_TP(2, 1, 96); else U2_TP(2, 1, 128);
    } else if (nw == 4 && nbw == 1) {
        if (stamps && cin == 64 && x3) launch_tp<4, 1, 64, true, 2>(grid, k, st, in, wf, cout, nbr, order, rr, items, n_items, kflip, out, stamps);
        else if (stamps && cin == 64) launch_tp<4, 1, 64, true, 1>(grid, k, st, in, wf, cout, nbr, order, rr, items, n_items, kflip, out, stamps);
        else if (cin == 32) U2_TP(4, 1, 32); else if (cin == 64) U2_TP(4, 1, 64);
        else if (cin == 96) U2_TP(4, 1, 96); else U2_TP(4, 1, 128);
    } else if (nw == 3) {
        if (cin == 32) U2_TP(3, 2, 32); else U2_TP(3, 2, 64);
    } else {
        if (cin == 32) U2_TP(4, 2, 32); else U2_TP(4, 2, 64);
    }
#undef U2_TP
    return check_launch(who);
}

size_t weight_fragments_bytes(int k, int rows, int cols, int arith) {
    const int ar = conv_tp_arith(arith);
    return (size_t)k * rows * cols * (ar == 2 ? 6 : ar == 3 ? 2 : 4) + (ar == 4 ? 16 : 0);      // (f16x2: two fp16 planes + the scale trailer)
}

int launch_weight_fragments(const float *w, int k, int rows, int cols, int transpose, int arith, float *wf, hipStream_t st) {
    const int64_t elems = (int64_t)k * rows * cols;
    if (elems == 0) return 0;
    const int ar = conv_tp_arith(arith);
    if (ar == 2 || ar == 3 || ar == 4) {
        const int64_t total = elems / 8;          // one thread per (offset, column, 8 reduction channels)
        // one-wave workgroups: the kernel is a single round of loads and stores per thread (latency-bound), and 27 x 64 x 64
        // weights are only 432 waves -- as 256-thread workgroups they sat on 108 of the 256 CUs
        constexpr int bt = 64;
        const dim3 grid((unsigned)ceil_div(total, bt), transpose == 2 ? 2 : 1);
        if (ar == 4) {
            char *base = reinterpret_cast<char *>(wf);
            float *t0 = reinterpret_cast<float *>(base + (size_t)elems * 4);
            float *t1 = transpose == 2 ? reinterpret_cast<float *>(base + (size_t)elems * 8 + 16) : t0;
            hipLaunchKernelGGL(weight_absmax_kernel, dim3(1), dim3(1024), 0, st, (const int64_t *)nullptr, w, elems, t0, t1);
            hipLaunchKernelGGL(weight_fragments_x3_kernel<2>, grid, dim3(bt), 0, st, w, rows, cols, transpose,
                               reinterpret_cast<bf16x8 *>(wf), total);
        } else if (ar == 2)
            hipLaunchKernelGGL(weight_fragments_x3_kernel<3>, grid, dim3(bt), 0, st, w, rows, cols, transpose,
                               reinterpret_cast<bf16x8 *>(wf), total);
        else
            hipLaunchKernelGGL(weight_fragments_x3_kernel<1>, grid, dim3(bt), 0, st, w, rows, cols, transpose,
                               reinterpret_cast<bf16x8 *>(wf), total);
    } else {
        hipLaunchKernelGGL(weight_fragments_kernel, dim3((unsigned)ceil_div(elems, 256), transpose == 2 ? 2 : 1), dim3(256), 0, st, w, rows, cols,
                           transpose, wf, elems);
    }
    return check_launch("u2mkd_weight_fragments");
}

int launch_weight_fragments_batch(const int64_t *jobs, int n_jobs, int64_t total_units, hipStream_t st) {
    if (n_jobs == 0 || total_units == 0) return 0;
    // (jobs without f16x2 planes leave at once)
    hipLaunchKernelGGL(weight_absmax_kernel, dim3((unsigned)n_jobs, kAbsmaxSlices), dim3(256), 0, st, jobs, (const float *)nullptr,
                       (int64_t)0, (float *)nullptr, (float *)nullptr);
    hipLaunchKernelGGL(weight_absmax_finish_kernel, dim3((unsigned)n_jobs), dim3(64), 0, st, jobs);
    hipLaunchKernelGGL(weight_fragments_batch_kernel, dim3((unsigned)total_units), dim3(64), 0, st, jobs, n_jobs);
    return check_launch("u2mkd_weight_fragments_batch");
}

}  // namespace u2mkd
