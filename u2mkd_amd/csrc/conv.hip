// Sparse 3D convolution on gfx950: output-stationary implicit GEMM over the
// neighbour table, fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32 fma chain).
// Replaces torchsparse v1.4.0 convolution_forward_cuda/backward_cuda
// (gather -> cuBLAS mm -> scatter-add per kernel offset)
// behind every spnn.Conv3d of core/models/build_blocks.py:25-80.
//
// Design (MI355X-first, not a translation of gather-GEMM-scatter):
//   * one wave owns 16*MR output rows x 16*NB output columns and walks the
//     kernel offsets; rows are gathered straight into the MFMA A operand
//     (lane (r = l&15, q = l>>4) loads the 16 bytes in[nbr[k][row r]][c0+4q..+3]),
//     so each gathered row is read once per pass and outputs are written once,
//     with no atomics and a deterministic summation order;
//   * an offset whose 16*MR rows have no neighbour is skipped by a ballot;
//   * dgrad is the same kernel on the swapped-role table with wt = kernel;
//   * wgrad scans the table, compacts the valid (gather,row) pairs of 256 rows
//     with wave ballots + prefix sums into LDS and feeds them as the MFMA
//     reduction dimension; partial slabs + ordered reduce (bitwise reproducible).
#include <stdlib.h>

#include "conv_internal.h"

namespace u2mkd {

__global__ void transpose_weights_kernel(const float *__restrict__ w, int cin, int cout, float *__restrict__ wt,
                                         int64_t total) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    // t indexes wt[k][co][ci]
    int ci = (int)(t % cin);
    int64_t r = t / cin;
    int co = (int)(r % cout);
    int64_t k = r / cout;
    wt[t] = w[(k * cin + ci) * cout + co];
}

// ---- output-stationary kernel, LDS-staged weights -----------------------------------
// Workgroup = WAVES waves = 16*WAVES output rows (mask-sorted order) x 16*NB output columns;
// every wave owns one 16-row MFMA block and all NB column blocks.
//   * the tile's neighbour indices (K x TM) and row ids are loaded once into LDS; the OR
//     of the rows' neighbour masks gives the offsets this tile visits at all;
//   * per (offset, KC-channel chunk) stage: the B tile wt[k][col0..][c..c+KC) is staged
//     through LDS once for all waves ([col][ci] rows of KC+8 floats: a lane's
//     ds_read_b128 of 4 consecutive ci is bank-conflict free), A rows are gathered
//     straight into MFMA operand registers; the global loads of stage s+1 are issued
//     before the MFMAs of stage s (register double buffering), one barrier per stage;
//   * a wave whose 16 rows have no neighbour at the offset skips its MFMAs;
//   * consecutive MFMAs go to different accumulators (16x16x4 f32 has a 40-cycle
//     dependent latency against a 32-cycle issue interval).
template <int WAVES, int NB, int KC>
__global__ void __launch_bounds__(64 * WAVES)
conv_os2_kernel(const float *__restrict__ in, int cin, const float *__restrict__ wt, int cout,
                const int32_t *__restrict__ nbr, const int32_t *__restrict__ order, RowRange rr_, int K, int kflip,
                float *__restrict__ out) {
    constexpr int NT = 64 * WAVES;
    constexpr int TM = 16 * WAVES, TN = 16 * NB;
    constexpr int BS = KC + 8;                          // LDS row stride (floats)
    constexpr int F4ROW = KC / 4;                       // float4 per B row
    constexpr int BPASS = (TN * F4ROW + NT - 1) / NT;   // float4 B loads per thread per stage
    constexpr int NJ = KC / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Bs = reinterpret_cast<float *>(smem);                 // [2][TN][BS]
    int *s_idx = reinterpret_cast<int *>(Bs + 2 * TN * BS);      // [K][TM]
    int *s_rid = s_idx + K * TM;                                 // [TM]
    unsigned *s_mask = reinterpret_cast<unsigned *>(s_rid + TM); // [1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int col0 = blockIdx.y * TN;
    const int64_t row0 = rr_.begin + (int64_t)(rr_.tile_order ? rr_.tile_order[blockIdx.x] : (int)blockIdx.x) * TM;
    const int64_t n_out = rr_.end, ld = rr_.ld;

    if (tid == 0) *s_mask = 0u;
    for (int e = tid; e < TM; e += NT) {
        int64_t row = row0 + e;
        s_rid[e] = row < n_out ? (order ? order[row] : (int)row) : -1;
    }
    __syncthreads();
    unsigned mymask = 0u;
    {
        // all of the tile's neighbour indices in ONE burst of independent loads (K <= 32): a rolled
        // load -> LDS-store loop serialises ceil(K*TM/NT) global round trips per tile, which was the
        // dominant fixed cost of the kernel (tools/ab_kscan.py)
        constexpr int IT = (32 * TM + NT - 1) / NT;
        int v[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int e = tid + i * NT;
            int k = e / TM, rr = e - k * TM;
            int64_t row = row0 + rr;
            v[i] = (e < K * TM && row < n_out) ? nbr[(int64_t)k * ld + row] : -1;
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int e = tid + i * NT;
            if (e < K * TM) {
                s_idx[e] = v[i];
                if (v[i] >= 0) mymask |= 1u << (e / TM);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mymask |= __shfl_xor(mymask, off);
    if (lane == 0 && mymask) atomicOr(s_mask, mymask);
    __syncthreads();
    unsigned km = *s_mask;

    f32x4 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunk = (cin + KC - 1) / KC;
    const int wrow = 16 * wave;   // first tile-local row of this wave

    float4 a_cur[NJ], a_nxt[NJ], breg[BPASS];
    bool act_cur = false, act_nxt = false;

    auto load_stage = [&](int k, int c, float4 (&a)[NJ], bool &active) {
        const float *wk = wt + (size_t)((kflip & 1) ? K - 1 - k : k) * cout * cin;
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int f = tid + NT * p;
            int col = f / F4ROW, ci = c * KC + (f % F4ROW) * 4;
            breg[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < TN && col0 + col < cout && ci < cin)
                breg[p] = *reinterpret_cast<const float4 *>(wk + (size_t)(col0 + col) * cin + ci);
        }
        int idx = s_idx[k * TM + wrow + r];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int ci = c * KC + 16 * j + 4 * q;
            a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx >= 0 && ci < cin) a[j] = *reinterpret_cast<const float4 *>(in + (size_t)idx * cin + ci);
        }
        active = __ballot(idx >= 0) != 0ULL;
    };
    auto store_B = [&](int buf) {
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int f = tid + NT * p;
            int col = f / F4ROW;
            if (col < TN)
                *reinterpret_cast<float4 *>(Bs + ((size_t)buf * TN + col) * BS + (f % F4ROW) * 4) = breg[p];
        }
    };

    if (km) {
        int k = __builtin_ctz(km);
        km &= km - 1;
        int c = 0;
        load_stage(k, 0, a_cur, act_cur);
        store_B(0);
        __syncthreads();
        int buf = 0;
        while (true) {
            int kn = k, cn = c + 1;
            bool have_next = true;
            if (cn == nchunk) {
                cn = 0;
                if (km) { kn = __builtin_ctz(km); km &= km - 1; } else have_next = false;
            }
            if (have_next) load_stage(kn, cn, a_nxt, act_nxt);
            if (act_cur) {
                const float *bb = Bs + (size_t)buf * TN * BS + r * BS + 4 * q;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float4 b[NB];
#pragma unroll
                    for (int n = 0; n < NB; ++n) b[n] = *reinterpret_cast<const float4 *>(bb + 16 * n * BS + 16 * j);
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].x, b[n].x, acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].y, b[n].y, acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].z, b[n].z, acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].w, b[n].w, acc[n], 0, 0, 0);
                }
            }
            if (!have_next) break;
            store_B(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            k = kn;
            c = cn;
            act_cur = act_nxt;
#pragma unroll
            for (int j = 0; j < NJ; ++j) a_cur[j] = a_nxt[j];
        }
    }
    // epilogue: D row = 4q + reg, col = r; rows go to their original (unsorted) positions
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        int rid = s_rid[wrow + 4 * q + reg];
        if (rid >= 0) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
                if (col0 + 16 * n + r < cout) out[(size_t)rid * cout + col0 + 16 * n + r] = acc[n][reg];
        }
    }
}

// ---- output-stationary kernel, both operands through LDS, column-split waves -----------------
// Same contraction as conv_os2_kernel; different work split.  The 4 waves of a workgroup share
// the 64 (mask-sorted) output rows and split the 64*NBW output COLUMNS, so for every offset all
// waves do the same amount of work (in conv_os2 a wave idles whenever its own 16 rows lack the
// offset the workgroup is visiting).  Per (offset, KC-channel chunk) stage:
//   * the gathered A rows of the ACTIVE 16-row blocks (whole 16-byte-coalesced row segments)
//     and the weight tile are loaded into registers while the previous stage computes, then
//     written to LDS ([row][KC+8] images, conflict-free ds_read_b128 fragments);
//   * every wave multiplies each active row block with its NBW column blocks.
// Row blocks without any neighbour at the offset are neither loaded nor multiplied.
template <int RB, int NBW, int KC>
__global__ void __launch_bounds__(256)
conv_os3_kernel(const float *__restrict__ in, int cin, const float *__restrict__ wt, int cout,
                const int32_t *__restrict__ nbr, const int32_t *__restrict__ order, RowRange rr_, int K, int kflip,
                float *__restrict__ out) {
    constexpr int TM = 16 * RB, TN = 64 * NBW;   // RB 16-row blocks per workgroup tile
    constexpr int S = KC + 8;                 // LDS row stride (floats)
    constexpr int F4 = KC / 4;                // float4 per staged row
    constexpr int AP = TM * F4 / 256;         // A float4 per thread per stage
    constexpr int BP = TN * F4 / 256;         // B float4 per thread per stage
    constexpr int NJ = KC / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);            // [TM][S]
    float *Bs = As + TM * S;                                // [TN][S]
    int *s_idx = reinterpret_cast<int *>(Bs + TN * S);      // [K][TM]
    int *s_rid = s_idx + K * TM;                            // [TM]
    unsigned *s_act = reinterpret_cast<unsigned *>(s_rid + TM);   // [K] active-row-block bits

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int64_t row0 = rr_.begin + (int64_t)(rr_.tile_order ? rr_.tile_order[blockIdx.x] : (int)blockIdx.x) * TM;
    const int64_t n_out = rr_.end, ld = rr_.ld;
    const int col0 = blockIdx.y * TN;

    for (int e = tid; e < TM; e += 256) {
        int64_t row = row0 + e;
        s_rid[e] = row < n_out ? (order ? order[row] : (int)row) : -1;
    }
    // neighbour indices of the tile + per-offset active-row-block bits.  A wave covers 64 consecutive
    // rows of ONE offset, so a ballot gives the block bits with no LDS atomics.
    constexpr int IT = 32 * TM / 256;                // K <= 32
    int vv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {                   // one burst of independent loads (see conv_os2)
        int e = wave * 64 + i * 256 + lane;
        int k = e / TM, rr = e - k * TM;
        int64_t row = row0 + rr;
        vv[i] = (e < K * TM && row < n_out) ? nbr[(int64_t)k * ld + row] : -1;
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int e0 = wave * 64 + i * 256;
        if (e0 >= K * TM) break;
        int e = e0 + lane;
        int k = e0 / TM;                             // e0 is a multiple of 64 and TM is 64 or 128
        int v = vv[i];
        s_idx[e] = v;
        unsigned long long bal = __ballot(v >= 0);
        if (lane == 0) {
            unsigned bits = 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if ((bal >> (16 * g)) & 0xFFFFULL) bits |= 1u << g;
            if (TM == 64) s_act[k] = bits;
            else reinterpret_cast<unsigned char *>(s_act)[4 * k + ((e0 % TM) >> 6)] = (unsigned char)bits;
        }
    }
    __syncthreads();
    if (TM == 128) {   // fold the two half-tile nibbles into one 8-bit mask per offset
        for (int kk = tid; kk < K; kk += 256) {
            unsigned w = s_act[kk];
            s_act[kk] = (w & 0xFu) | (((w >> 8) & 0xFu) << 4);
        }
        __syncthreads();
    }

    f32x4 acc[RB][NBW];
#pragma unroll
    for (int m = 0; m < RB; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunk = (cin + KC - 1) / KC;
    float4 ra[AP], rb[BP];

    const int kf = kflip & 1;
    auto load_stage = [&](int k, int c, unsigned act) {
        const float *wk = wt + (size_t)(kf ? K - 1 - k : k) * cout * cin;
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            int f = tid + 256 * p;
            int col = f / F4, ci = c * KC + (f % F4) * 4;
            rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col0 + col < cout && ci < cin)
                rb[p] = *reinterpret_cast<const float4 *>(wk + (size_t)(col0 + col) * cin + ci);
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int f = tid + 256 * p;
            int row = f / F4, ci = c * KC + (f % F4) * 4;
            ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((act >> (row >> 4)) & 1u) {
                int idx = s_idx[k * TM + row];
                if (idx >= 0 && ci < cin) ra[p] = *reinterpret_cast<const float4 *>(in + (size_t)idx * cin + ci);
            }
        }
    };
    auto store_stage = [&](unsigned act) {
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            int f = tid + 256 * p;
            *reinterpret_cast<float4 *>(Bs + (f / F4) * S + (f % F4) * 4) = rb[p];
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int f = tid + 256 * p;
            int row = f / F4;
            if ((act >> (row >> 4)) & 1u) *reinterpret_cast<float4 *>(As + row * S + (f % F4) * 4) = ra[p];
        }
    };

    // first offset with any active block
    int k = 0;
    while (k < K && s_act[k] == 0u) ++k;
    if (k < K) {
        int c = 0;
        unsigned act = s_act[k];
        load_stage(k, 0, act);
        store_stage(act);
        __syncthreads();
        while (true) {
            int kn = k, cn = c + 1;
            unsigned actn = act;
            bool have_next = true;
            if (cn == nchunk) {
                cn = 0;
                kn = k + 1;
                while (kn < K && s_act[kn] == 0u) ++kn;
                if (kn < K) actn = s_act[kn]; else have_next = false;
            }
            if (have_next) load_stage(kn, cn, actn);
            // ---- compute stage (k, c): B fragments of this wave's column blocks, then every active row block
            {
                float4 b[NJ][NBW];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int n = 0; n < NBW; ++n)
                        b[j][n] = *reinterpret_cast<const float4 *>(Bs + (16 * (NBW * wave + n) + r) * S + 16 * j + 4 * q);
#pragma unroll
                for (int m = 0; m < RB; ++m) {
                    if (!((act >> m) & 1u)) continue;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float4 a = *reinterpret_cast<const float4 *>(As + (16 * m + r) * S + 16 * j + 4 * q);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[j][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[j][n].y, acc[m][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[j][n].z, acc[m][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[j][n].w, acc[m][n], 0, 0, 0);
                    }
                }
            }
            if (!have_next) break;
            __syncthreads();          // everyone finished reading the LDS images
            store_stage(actn);
            __syncthreads();          // images of the next stage complete
            k = kn;
            c = cn;
            act = actn;
        }
    }
    // epilogue: this wave's columns of all rows of the tile; D row = 4q + reg, col = r
#pragma unroll
    for (int m = 0; m < RB; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            int rid = s_rid[16 * m + 4 * q + reg];
            if (rid >= 0) {
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    int col = col0 + 16 * (NBW * wave + n) + r;
                    if (col < cout) out[(size_t)rid * cout + col] = acc[m][n][reg];
                }
            }
        }
}

template <int RB, int NBW, int KC>
static void launch_conv_os3(dim3 grid, int K, hipStream_t st, const float *in, int cin, const float *wt, int cout,
                            const int32_t *nbr, const int32_t *order, RowRange n_out, int kflip, float *out) {
    size_t lds = (size_t)(16 * RB + 64 * NBW) * (KC + 8) * 4 + (size_t)K * 16 * RB * 4 + 16 * RB * 4 + (size_t)K * 4 + 16;
    if (lds > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_os3_kernel<RB, NBW, KC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv_os3_kernel<RB, NBW, KC>), grid, dim3(256), lds, st, in, cin, wt, cout, nbr, order, n_out,
                       K, kflip, out);
}

// ---- weight gradient over the compacted pair list (rulebook) ----------------------------
// dW[k] = sum over the pairs of offset k of A[pa]^T B[pb].  The pairs are the MFMA
// reduction dimension: a workgroup stages CP pairs' rows (both operands gathered, full
// 16-byte coalesced segments) into LDS as [pair][channel] images whose row stride is
// == 16 (mod 32) floats, so the transposed operand reads (ds_read_b32, 16 channels x 4
// pairs per wave instruction) are bank-conflict free.  Work split: every offset's pair
// segment is cut into chunks of plan.ch pairs (computed on the device from the pair
// count, so no host sync); workgroup w finds its (offset, chunk) in the 27-entry prefix.
// Partial tiles go to slabs; wgrad_pairs_reduce_kernel sums them in a fixed order.
//
// plan layout (int32): [0] = P, [1] = CH, [2 .. 2+K] = pair offsets kofs[0..K],
//                      [3+K .. 3+2K] = workgroup prefix wg[0..K].
// 4 bf16 channels (8 bytes) -> f32x4: a bf16 is the upper half of the fp32 with the same value
__device__ __forceinline__ f32x4 widen_bf16x4(const char *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    f32x4 v;
    v[0] = __uint_as_float(u.x << 16);
    v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16);
    v[3] = __uint_as_float(u.y & 0xffff0000u);
    return v;
}

__global__ void wgrad_plan_kernel(const int32_t *__restrict__ nbsizes, int K, int g_target, int cp,
                                  int32_t *__restrict__ plan) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int P = 0;
    for (int k = 0; k < K; ++k) { plan[2 + k] = P; P += nbsizes[k]; }
    plan[2 + K] = P;
    int ch = (P + g_target - 1) / g_target;
    if (ch < 128) ch = 128;
    ch = (ch + cp - 1) / cp * cp;
    plan[0] = P;
    plan[1] = ch;
    int w = 0;
    for (int k = 0; k < K; ++k) { plan[3 + K + k] = w; w += (nbsizes[k] + ch - 1) / ch; }
    plan[3 + 2 * K] = w;
}

// B16: a and b are BF16 rows (bf16 storage, BASELINE.json configs[4]): 8-byte gathers of 4 channels, widened to fp32
// in registers; LDS images, MFMA arithmetic (f32) and slabs as in the fp32 form.
template <int WM, int WN, int CP, bool STAMP = false, bool B16 = false>
__global__ void __launch_bounds__(256)
conv_wgrad_pairs_kernel(const float *__restrict__ a, int ca, const float *__restrict__ b, int cb,
                        const int32_t *__restrict__ pairs, const int32_t *__restrict__ plan, int K, int swap,
                        int tiles_b, float *__restrict__ slabs, unsigned long long *__restrict__ stamps = nullptr) {
    // STAMP (tools/stamps_wgrad.py only): s_memtime of workgroup 0 at the phases of its first 32 chunks
    int st_i = 0;
#define U2_PH()                                                                              \
    do {                                                                                     \
        if (STAMP && blockIdx.x == 0 && blockIdx.y == 0 && st_i < 256) {                     \
            unsigned long long ts_ = __builtin_amdgcn_s_memtime();                           \
            if (threadIdx.x == 0) stamps[st_i] = ts_;                                        \
            ++st_i;                                                                          \
        }                                                                                    \
    } while (0)
    constexpr int TA = 32 * WM, TB = 32 * WN;
    constexpr int SA = TA + 16 - (TA % 32 == 16 ? 16 : 0);   // row stride == 16 (mod 32)
    constexpr int SB = TB + 16 - (TB % 32 == 16 ? 16 : 0);
    constexpr int FA = TA / 4, FB = TB / 4;                  // float4 per staged row
    constexpr int PA = (CP * FA + 255) / 256, PB = (CP * FB + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);             // [2][CP][SA]
    float *Bs = As + 2 * CP * SA;                            // [2][CP][SB]

    const int w = blockIdx.x;
    const int *wg = plan + 3 + K;
    // this workgroup's (offset, chunk): ONE round trip -- lane l holds wg[l] and kofs[l], the offset is the number of
    // prefix entries <= w (a serial scan of the prefix costs up to K dependent loads before the first pair index
    // can be fetched: ~2 us at the head of every workgroup)
    int k = 0, p_begin, p_end;
    const int ch = plan[1];
    if (K <= 63) {
        const int l = min((int)(threadIdx.x & 63), K);
        const int wgv = wg[l], kof = plan[2 + l];
        const unsigned long long le = __ballot(wgv <= w) & ((2ULL << K) - 2ULL);      // lanes 1..K
        k = __builtin_amdgcn_readfirstlane(__popcll(le));
        if (k >= K) return;                                  // w >= wg[K]: surplus workgroup
        p_begin = __builtin_amdgcn_readlane(kof, k) + (w - __builtin_amdgcn_readlane(wgv, k)) * ch;
        p_end = min(p_begin + ch, __builtin_amdgcn_readlane(kof, k + 1));
    } else {
        if (w >= wg[K]) return;
        while (w >= wg[k + 1]) ++k;
        p_begin = plan[2 + k] + (w - wg[k]) * ch;
        p_end = min(p_begin + ch, plan[2 + k + 1]);
    }

    const int ta = blockIdx.y / tiles_b, tb = blockIdx.y % tiles_b;
    const int a0 = ta * TA, b0 = tb * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wy = wave >> 1, wx = wave & 1;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the CP-pair chunks of this workgroup's pair range: in iteration c the pair
    // indices of chunk c+3 and the gathered rows of chunk c+2 are issued, chunk c multiplies from its LDS image,
    // the rows of chunk c+1 (issued one iteration earlier) are stored into the other image, one LDS-only
    // barrier.  Measured before (MI355X, 64x64 at 80k voxels): the kernel took the same 55 us on sequential
    // rows as on gathered ones -- not gather-bound but latency-bound: per chunk one dependent index -> row
    // round trip with a single chunk in flight, drained by __syncthreads() (which waits for vmcnt(0)).  All
    // loads are UNCONDITIONAL (positions clamped into the range, rows zeroed at the store) so the compiler's
    // s_waitcnt counters stay exact.
    const int ca_ok = ca - 4, cb_ok = cb - 4;      // last valid 16-byte column of a row
    f32x4 ra[2][PA], rb[2][PB];      // (native vectors: arrays of the HIP float4 struct end up in scratch memory)
    int ia[2][PA], ib[2][PB];
    auto load_idx = [&](int p0, int (&xa)[PA], int (&xb)[PB]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int pr = (tid + 256 * i) / FA;
            const int pp = min(p0 + min(pr, CP - 1), p_end - 1);
            xa[i] = pairs[2 * (size_t)pp + (swap ? 1 : 0)];
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int pr = (tid + 256 * i) / FB;
            const int pp = min(p0 + min(pr, CP - 1), p_end - 1);
            xb[i] = pairs[2 * (size_t)pp + (swap ? 0 : 1)];
        }
    };
    auto load_rows = [&](const int (&xa)[PA], const int (&xb)[PB], f32x4 (&va)[PA], f32x4 (&vb)[PB])
                         __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int c = min(a0 + ((tid + 256 * i) % FA) * 4, ca_ok);
            if (B16) va[i] = widen_bf16x4(reinterpret_cast<const char *>(a) + ((size_t)xa[i] * ca + c) * 2);
            else va[i] = *reinterpret_cast<const f32x4 *>(a + (size_t)xa[i] * ca + c);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int c = min(b0 + ((tid + 256 * i) % FB) * 4, cb_ok);
            if (B16) vb[i] = widen_bf16x4(reinterpret_cast<const char *>(b) + ((size_t)xb[i] * cb + c) * 2);
            else vb[i] = *reinterpret_cast<const f32x4 *>(b + (size_t)xb[i] * cb + c);
        }
    };
    auto store_chunk = [&](int p0, int buf, const f32x4 (&va)[PA], const f32x4 (&vb)[PB]) __attribute__((always_inline)) {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int f = tid + 256 * i;
            const int pr = f / FA, c = (f % FA) * 4;
            const bool ok = p0 + pr < p_end && a0 + c < ca;
            if (pr < CP) *reinterpret_cast<f32x4 *>(As + ((size_t)buf * CP + pr) * SA + c) = ok ? va[i] : zero;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int f = tid + 256 * i;
            const int pr = f / FB, c = (f % FB) * 4;
            const bool ok = p0 + pr < p_end && b0 + c < cb;
            if (pr < CP) *reinterpret_cast<f32x4 *>(Bs + ((size_t)buf * CP + pr) * SB + c) = ok ? vb[i] : zero;
        }
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto multiply = [&](int buf) __attribute__((always_inline)) {
        const float *ap = As + (size_t)buf * CP * SA + q * SA + 16 * WM * wy + r;
        const float *bp = Bs + (size_t)buf * CP * SB + q * SB + 16 * WN * wx + r;
        // operands are read LD steps ahead of their MFMAs (an LDS read issued right before its MFMAs exposes
        // the whole LDS latency; one step = WM*WN MFMAs = 128 cycles at 2x2 covers less than one LDS round trip)
        constexpr int NS4 = CP / 4, LD = 3;
        float av[LD + 1][WM], bv[LD + 1][WN];
#pragma unroll
        for (int s4 = 0; s4 < LD && s4 < NS4; ++s4) {
#pragma unroll
            for (int m = 0; m < WM; ++m) av[s4][m] = ap[s4 * 4 * SA + 16 * m];
#pragma unroll
            for (int n = 0; n < WN; ++n) bv[s4][n] = bp[s4 * 4 * SB + 16 * n];
        }
#pragma unroll
        for (int s4 = 0; s4 < NS4; ++s4) {
            if (s4 + LD < NS4) {
#pragma unroll
                for (int m = 0; m < WM; ++m) av[(s4 + LD) % (LD + 1)][m] = ap[(s4 + LD) * 4 * SA + 16 * m];
#pragma unroll
                for (int n = 0; n < WN; ++n) bv[(s4 + LD) % (LD + 1)][n] = bp[(s4 + LD) * 4 * SB + 16 * n];
            }
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4 % (LD + 1)][m], bv[s4 % (LD + 1)][n], acc[m][n], 0, 0, 0);
        }
    };

    // prologue: chunk 0 into image 0, rows of chunk 1 and indices of chunk 2 in flight
    load_idx(p_begin, ia[0], ib[0]);
    load_idx(p_begin + CP, ia[1], ib[1]);
    load_rows(ia[0], ib[0], ra[0], rb[0]);
    load_idx(p_begin + 2 * CP, ia[0], ib[0]);
    load_rows(ia[1], ib[1], ra[1], rb[1]);
    store_chunk(p_begin, 0, ra[0], rb[0]);
    lds_barrier();
    // iteration c (chunk at p0): sets are named by parity -- rows of chunk c+1 sit in set (c+1)&1, the
    // indices of chunk c+2 in set c&1
    // (sched_barrier: hipcc otherwise sinks the load issue below the MFMA block -- nothing orders them -- which
    // turns the prefetch into a wait; the loop body has no exit in the middle: a range with an odd number of
    // chunks multiplies one all-zero image)
    for (int p0 = p_begin; p0 < p_end; p0 += 2 * CP) {
        // even chunk c: image 0; rows c+1 in set 1; indices c+2 in set 0
        U2_PH();
        load_idx(p0 + 3 * CP, ia[1], ib[1]);
        load_rows(ia[0], ib[0], ra[0], rb[0]);                 // rows of chunk c+2
        __builtin_amdgcn_sched_barrier(0);
        U2_PH();
        multiply(0);
        __builtin_amdgcn_sched_barrier(0);
        U2_PH();
        store_chunk(p0 + CP, 1, ra[1], rb[1]);
        U2_PH();
        lds_barrier();
        // odd chunk c+1: image 1; rows c+2 in set 0; indices c+3 in set 1
        load_idx(p0 + 4 * CP, ia[0], ib[0]);
        load_rows(ia[1], ib[1], ra[1], rb[1]);                 // rows of chunk c+3
        __builtin_amdgcn_sched_barrier(0);
        multiply(1);
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(p0 + 2 * CP, 0, ra[0], rb[0]);
        lds_barrier();
    }
#undef U2_PH
    // D[i = a channel][j = b channel]: row = 4q + reg, col = r
    float *slab = slabs + (size_t)w * ca * cb;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            int ach = a0 + 16 * (WM * wy + m) + 4 * q + reg;
            if (ach < ca) {
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    int bch = b0 + 16 * (WN * wx + n) + r;
                    if (bch < cb) slab[(size_t)ach * cb + bch] = acc[m][n][reg];
                }
            }
        }
}

// (Tried and rejected, MI355X: this kernel in bf16x3 arithmetic -- both gathered operands split into three bf16
// planes at the LDS store, fragments read with ds_read_b64_tr_b16, 6 x v_mfma_f32_16x16x32_bf16 per 16x16x32 block.
// Bit-for-bit gate passed, but 50.6 us against 44.3 us at 64 x 64 and 610 against 438 us at 256 x 256: every element
// of BOTH operands needs the ~10-instruction split at run time and feeds only 64 MACs, which costs the vector ALU
// what the matrix pipe saves, and the fragments + planes push the kernel to 2 waves per SIMD.  The forward kernel
// is different: its weights are split once, off line, and a gathered row is split once for 4 column-split waves.)

// dw[k][e] = sum of the offset's slabs, fixed order: 16 slab lanes x 16 float4 lanes per
// block (64 elements); lane g sums slabs wg[k]+g, +16, ... and the 16 partials are added
// in lane order through LDS (deterministic, ~P/CH/16 dependent loads per thread).
__global__ void __launch_bounds__(256)
wgrad_pairs_reduce_kernel(const float *__restrict__ slabs, const int32_t *__restrict__ plan, int K,
                          int64_t tile_elems, float *__restrict__ dw, int merge = 1) {
    __shared__ float4 part[16][16];
    const int k = blockIdx.y;
    const int lx = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int64_t e = ((int64_t)blockIdx.x * 16 + lx) * 4;
    const int *wg = plan + 3 + K;
    const int w0 = wg[k], w1 = wg[k + 1];
    // the slabs that were written: slot w0 and every slot of the offset that is a multiple of `merge` (conv_wgrad_x3_kernel:
    // one slab per run of merged slots); merge = 1: every slot
    const int m0 = (w0 / merge + 1) * merge;
    const int nslab = w1 > w0 ? 1 + (w1 > m0 ? (w1 - m0 + merge - 1) / merge : 0) : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < tile_elems) {
        // four slab rows in flight per thread (the loads do not depend on the running sum; the order of the
        // additions is fixed: live slabs g, g + 16, g + 32, ... of the offset)
        for (int j = g; j < nslab; j += 64) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int jj = j + 16 * i;
                const int ww = jj == 0 ? w0 : m0 + (jj - 1) * merge;
                v[i] = jj < nslab ? *reinterpret_cast<const float4 *>(slabs + (size_t)ww * tile_elems + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
    }
    part[g][lx] = acc;
    __syncthreads();
    if (g == 0 && e < tile_elems) {
        float4 t = part[0][lx];
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            float4 v = part[i][lx];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        *reinterpret_cast<float4 *>(dw + (size_t)k * tile_elems + e) = t;
    }
}

static int wgrad_g_target(int64_t n_rows, int k) {
    int64_t g = (n_rows * (k < 8 ? k : 8) * 2 + 511) / 512;      // (x 2: measured 5-10 % on the <= 32k-voxel levels)
    if (g < 32) g = 32;
    if (g > 1024) g = 1024;                                       // 4 workgroups per CU
    return (int)g;
}

// ---- pair-schedule kernel: one offset per 64-pair tile, pipelined across tiles ------------
// The pair schedule (u2mkd_pairs_build) makes every tile a plain [64 x cin] x [cin x 16*NB]
// product with gathered A rows: no neighbour table, no offset walk.  A workgroup (4 waves, one
// 16-pair MFMA block each) strides over the tiles with a fixed grid; (tile, channel chunk)
// stages form ONE software pipeline -- the loads of the next stage, including the first
// stage of the NEXT tile, are in flight while the current stage multiplies, and the pair
// indices of the next tile are fetched a whole tile ahead.  B goes through a double-buffered
// LDS image ([col][KC+8], conflict-free ds_read_b128), A straight into MFMA operand registers.
template <int NB, int KC>
__global__ void __launch_bounds__(256)
conv_pairs_kernel(const float *__restrict__ in, int cin, const float *__restrict__ wt, int cout,
                  const int32_t *__restrict__ pair_idx, const int32_t *__restrict__ tile_k,
                  const int32_t *__restrict__ n_tiles, int64_t n_dense, const float *__restrict__ bias,
                  float *__restrict__ y) {
    constexpr int NT = 256, TN = 16 * NB, BS = KC + 8, F4ROW = KC / 4;
    constexpr int BPASS = (TN * F4ROW + NT - 1) / NT, NJ = KC / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Bs = reinterpret_cast<float *>(smem);   // [2][TN][BS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int col0 = blockIdx.y * TN;
    // dense mode (pair_idx == nullptr): y = in * B_0 (+ bias) over rows [0, n_dense) -- a plain
    // linear layer on the same pipeline (entry p is row p, every tile uses offset 0)
    const int ntile = pair_idx ? *n_tiles : (int)((n_dense + 63) / 64);
    const int nchunk = (cin + KC - 1) / KC;
    auto entry = [&](int t) -> int {
        int64_t p = (int64_t)t * 64 + 16 * wave + r;
        return pair_idx ? pair_idx[p] : (p < n_dense ? (int)p : -1);
    };
    // a contiguous run of tiles per workgroup: consecutive tiles mostly share their offset, and
    // with a single channel chunk the weight image then simply stays in LDS (no reload, no barrier)
    const int per = (ntile + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = blockIdx.x * per;
    const int tile_end = min(tile + per, ntile);
    if (tile >= tile_end) return;

    float4 a_cur[NJ], a_nxt[NJ], breg[BPASS];
    auto load_stage = [&](int k, int c, int idx, float4 (&a)[NJ], bool with_b) {
        if (with_b) {
            const float *wk = wt + (size_t)k * cout * cin;
#pragma unroll
            for (int p = 0; p < BPASS; ++p) {
                int f = tid + NT * p;
                int col = f / F4ROW, ci = c * KC + (f % F4ROW) * 4;
                breg[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col < TN && col0 + col < cout && ci < cin)
                    breg[p] = *reinterpret_cast<const float4 *>(wk + (size_t)(col0 + col) * cin + ci);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int ci = c * KC + 16 * j + 4 * q;
            a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx >= 0 && ci < cin) a[j] = *reinterpret_cast<const float4 *>(in + (size_t)idx * cin + ci);
        }
    };
    auto store_B = [&](int buf) {
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int f = tid + NT * p;
            int col = f / F4ROW;
            if (col < TN)
                *reinterpret_cast<float4 *>(Bs + ((size_t)buf * TN + col) * BS + (f % F4ROW) * 4) = breg[p];
        }
    };

    int idx = entry(tile);
    int k = tile_k ? tile_k[tile] : 0;
    int tile_n = tile + 1;
    int idx_n = -1, k_n = k;
    if (tile_n < tile_end) {
        idx_n = entry(tile_n);
        k_n = tile_k ? tile_k[tile_n] : 0;
    }
    float bv[NB];
    f32x4 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        bv[n] = (bias && col0 + 16 * n + r < cout) ? bias[col0 + 16 * n + r] : 0.f;
        acc[n] = (f32x4){bv[n], bv[n], bv[n], bv[n]};
    }
    load_stage(k, 0, idx, a_cur, true);
    store_B(0);
    __syncthreads();
    int buf = 0, c = 0;
    while (true) {
        const bool last_chunk = c + 1 == nchunk;
        const bool have_next = !last_chunk || tile_n < tile_end;
        const bool new_b = have_next && (nchunk > 1 || k_n != k);   // workgroup-uniform
        if (have_next) {
            if (!last_chunk) load_stage(k, c + 1, idx, a_nxt, new_b);
            else load_stage(k_n, 0, idx_n, a_nxt, new_b);
        }
        if (__ballot(idx >= 0) != 0ULL) {
            const float *bb = Bs + (size_t)buf * TN * BS + r * BS + 4 * q;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float4 b[NB];
#pragma unroll
                for (int n = 0; n < NB; ++n) b[n] = *reinterpret_cast<const float4 *>(bb + 16 * n * BS + 16 * j);
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].x, b[n].x, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].y, b[n].y, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].z, b[n].z, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].w, b[n].w, acc[n], 0, 0, 0);
            }
        }
        if (last_chunk) {   // D row = 4q + reg, col = r; y row = the pair's slot
            float *yr = y + ((size_t)tile * 64 + 16 * wave + 4 * q) * cout + col0 + r;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    if (col0 + 16 * n + r < cout) yr[(size_t)reg * cout + 16 * n] = acc[n][reg];
                    acc[n][reg] = bv[n];
                }
        }
        if (!have_next) break;
        if (new_b) {
            // the other buffer was last read before the previous flip's barrier: free to overwrite
            store_B(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        if (last_chunk) {
            tile = tile_n;
            idx = idx_n;
            k = k_n;
            c = 0;
            ++tile_n;
            idx_n = -1;
            if (tile_n < tile_end) {
                idx_n = entry(tile_n);
                k_n = tile_k ? tile_k[tile_n] : 0;
            }
        } else {
            ++c;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) a_cur[j] = a_nxt[j];
    }
}

template <int KC>
static int launch_conv_pairs(int nb, dim3 grid, hipStream_t st, const float *in, int cin, const float *wt, int cout,
                             const int32_t *pair_idx, const int32_t *tile_k, const int32_t *n_tiles, int64_t n_dense,
                             const float *bias, float *y) {
    size_t lds = (size_t)2 * 16 * nb * (KC + 8) * 4;
#define U2_CASE(N)                                                                                                 \
    case N:                                                                                                        \
        if (lds > 65536)                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_pairs_kernel<N, KC>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        hipLaunchKernelGGL((conv_pairs_kernel<N, KC>), grid, dim3(256), lds, st, in, cin, wt, cout, pair_idx,      \
                           tile_k, n_tiles, n_dense, bias, y);                                                     \
        break;
    switch (nb) {
        U2_CASE(1) U2_CASE(2) U2_CASE(3) U2_CASE(4) U2_CASE(5) U2_CASE(6) U2_CASE(7) U2_CASE(8)
        default: set_error("conv pairs: unsupported column block count %d", nb); return 2;
    }
#undef U2_CASE
    return 0;
}

// out[j] = sum over offsets (ascending) of y[pos[j][k]] (pos < 0: no pair); one float4 per
// thread, a row's K slots are read first (contiguous), then all of its y rows are in flight.
__global__ void __launch_bounds__(256)
pairs_gather_sum_kernel(const float *__restrict__ y, const int32_t *__restrict__ pos, int64_t n_rows, int K, int c4,
                        float *__restrict__ out, const float *__restrict__ ep_scale = nullptr,
                        const float *__restrict__ ep_shift = nullptr, const float *__restrict__ ep_res = nullptr, int ep_relu = 0) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_rows * c4) return;
    int64_t r = t / c4;
    int c = (int)(t - r * c4);
    const int32_t *pr = pos + r * K;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K; k0 += 8) {
        int p[8];
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = k0 + i < K ? pr[k0 + i] : -1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = p[i] >= 0 ? reinterpret_cast<const float4 *>(y)[(int64_t)p[i] * c4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    if (ep_scale) {      // folded eval-mode BatchNorm (+ residual, + ReLU), u2mkd_pairs_gather_sum_ep
        const float4 sc = reinterpret_cast<const float4 *>(ep_scale)[c], sh = reinterpret_cast<const float4 *>(ep_shift)[c];
        acc.x = fmaf(acc.x, sc.x, sh.x); acc.y = fmaf(acc.y, sc.y, sh.y); acc.z = fmaf(acc.z, sc.z, sh.z); acc.w = fmaf(acc.w, sc.w, sh.w);
        if (ep_res) {
            const float4 rs = reinterpret_cast<const float4 *>(ep_res)[r * c4 + c];
            acc.x += rs.x; acc.y += rs.y; acc.z += rs.z; acc.w += rs.w;
        }
        if (ep_relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    }
    reinterpret_cast<float4 *>(out)[r * c4 + c] = acc;
}

// The same sum with the BatchNorm statistics of its output taken in the store (round 6): a workgroup owns kGsSlabRows consecutive
// output rows -- rl = 256 / c4 row lanes x c4 float4 columns, 32 / rl rows per thread, their sums kept in registers -- and writes,
// next to the rows, the slab's per-channel (mean, centred second moment M2) in the layout of bn.hip's slab partials
// (`partial` [slabs][2][C]; rows of slab b = min(kGsSlabRows, n_rows - b * kGsSlabRows)): the BatchNorm that follows a wide
// convolution then starts at its merge step (u2mkd_bn_train_forward_from_partial) -- its statistics pass, one launch and two
// reads of the feature matrix per layer, is gone (no statistics pass at all: -2.3 ms of KD step, NOTES N10.13).  Two passes over
// the registers (mean first, then M2 around it): cancellation-safe like the pass it replaces, fixed order: reproducible.
constexpr int kGsSlabRows = 32, kGsMaxIt = 16;

__global__ void __launch_bounds__(256)
pairs_gather_sum_stats_kernel(const float *__restrict__ y, const int32_t *__restrict__ pos, int64_t n_rows, int K, int c4,
                              float *__restrict__ out, float *__restrict__ partial) {
    __shared__ float4 red[256];
    const int rl = 256 / c4;                          // row lanes (c4 <= 128: rl >= 2, at most kGsMaxIt rows per thread)
    const int ry = threadIdx.x / c4, j = threadIdx.x - ry * c4;
    const bool lane_on = ry < rl;
    const int64_t r0 = (int64_t)blockIdx.x * kGsSlabRows;
    const int rows = (int)min((int64_t)kGsSlabRows, n_rows - r0);
    float4 vals[kGsMaxIt];
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < kGsMaxIt; ++it) {
        const int rr = it * rl + ry;
        vals[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_on && rr < rows) {
            const int64_t r = r0 + rr;
            const int32_t *pr = pos + r * K;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k0 = 0; k0 < K; k0 += 8) {
                int p[8];
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = k0 + i < K ? pr[k0 + i] : -1;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    v[i] = p[i] >= 0 ? reinterpret_cast<const float4 *>(y)[(int64_t)p[i] * c4 + j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
            }
            reinterpret_cast<float4 *>(out)[r * c4 + j] = acc;
            vals[it] = acc;
            sum.x += acc.x; sum.y += acc.y; sum.z += acc.z; sum.w += acc.w;
        }
    }
    if (lane_on) red[ry * c4 + j] = sum;
    __syncthreads();
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane_on) {
        for (int g = 0; g < rl; ++g) {
            const float4 v = red[g * c4 + j];
            mean.x += v.x; mean.y += v.y; mean.z += v.z; mean.w += v.w;
        }
        const float inv = 1.f / (float)rows;
        mean.x *= inv; mean.y *= inv; mean.z *= inv; mean.w *= inv;
    }
    __syncthreads();
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < kGsMaxIt; ++it) {
        if (lane_on && it * rl + ry < rows) {
            const float dx = vals[it].x - mean.x, dy = vals[it].y - mean.y, dz = vals[it].z - mean.z, dw = vals[it].w - mean.w;
            m2.x += dx * dx; m2.y += dy * dy; m2.z += dz * dz; m2.w += dw * dw;
        }
    }
    if (lane_on) red[ry * c4 + j] = m2;
    __syncthreads();
    if (lane_on && ry == 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < rl; ++g) {
            const float4 v = red[g * c4 + j];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        float *p = partial + (size_t)blockIdx.x * 2 * (4 * c4);
        *reinterpret_cast<float4 *>(p + 4 * j) = mean;
        *reinterpret_cast<float4 *>(p + 4 * c4 + 4 * j) = t;
    }
}

// the same over bf16 rows (bf16 storage, BASELINE.json configs[4]): 8 channels = 16 bytes per thread, fp32 sum in ascending
// offset order, one rounding to bf16 at the store
__global__ void __launch_bounds__(256)
pairs_gather_sum_bf16_kernel(const uint4 *__restrict__ y, const int32_t *__restrict__ pos, int64_t n_rows, int K, int c8,
                             uint4 *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_rows * c8) return;
    int64_t r = t / c8;
    int c = (int)(t - r * c8);
    const int32_t *pr = pos + r * K;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 8) {
        int p[8];
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = k0 + i < K ? pr[k0 + i] : -1;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[i] >= 0 ? y[(int64_t)p[i] * c8 + c] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {      // a bf16 is the upper half of the fp32 with the same value
                acc[2 * j] += __uint_as_float(w[j] << 16);
                acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
            }
        }
    }
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 lo = (__bf16)acc[2 * j], hi = (__bf16)acc[2 * j + 1];
        o[j] = (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    out[r * c8 + c] = make_uint4(o[0], o[1], o[2], o[3]);
}

template <int WAVES, int KC>
static int launch_conv_os2(int nb, dim3 grid, int K, hipStream_t st, const float *in, int cin, const float *wt,
                           int cout, const int32_t *nbr, const int32_t *order, RowRange n_out, int kflip, float *out) {
    const int tm = 16 * WAVES, tn = 16 * nb;
    size_t lds = (size_t)2 * tn * (KC + 8) * 4 + (size_t)K * tm * 4 + (size_t)tm * 4 + 16;
#define U2_CASE(N)                                                                                                 \
    case N:                                                                                                        \
        if (lds > 65536)                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_os2_kernel<WAVES, N, KC>),              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        hipLaunchKernelGGL((conv_os2_kernel<WAVES, N, KC>), grid, dim3(64 * WAVES), lds, st, in, cin, wt, cout,    \
                           nbr, order, n_out, K, kflip, out);                                                      \
        break;
    switch (nb) {
        U2_CASE(1) U2_CASE(2) U2_CASE(3) U2_CASE(4) U2_CASE(5) U2_CASE(6) U2_CASE(7) U2_CASE(8)
        default: set_error("conv: unsupported column block count %d", nb); return 2;
    }
#undef U2_CASE
    return 0;
}

static int pick_nb(int cout16) {
    if (cout16 <= 8) return cout16;
    for (int nb = 8; nb >= 4; --nb)
        if (cout16 % nb == 0) return nb;
    return 8;  // ragged last tile is masked
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

int u2mkd_transpose_weights(const float *w, int32_t k, int32_t cin, int32_t cout, float *wt, u2mkd_stream_t s) {
    int64_t total = (int64_t)k * cin * cout;
    if (total == 0) return 0;
    U2_REQUIRE(w && wt, "u2mkd_transpose_weights: null pointer");
    hipLaunchKernelGGL(transpose_weights_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), w,
                       cin, cout, wt, total);
    return check_launch("u2mkd_transpose_weights");
}

static int conv_forward_impl(const char *who, const float *in, int64_t n_in, int32_t cin, const float *wt,
                             int32_t cout, const int32_t *nbr, const int32_t *order, RowRange rr, int32_t k,
                             int32_t kflip, int32_t variant, float *out, u2mkd_stream_t s) {
    const int64_t n_rows = rr.end - rr.begin;
    if (n_rows <= 0) return 0;
    U2_REQUIRE(in && wt && nbr && out, "%s: null pointer", who);
    U2_REQUIRE(cin > 0 && cin % 4 == 0, "%s: cin=%d must be a positive multiple of 4", who, cin);
    U2_REQUIRE(cout > 0, "%s: cout=%d must be positive", who, cout);
    U2_REQUIRE(k > 0 && k <= 32 && n_in >= 0, "%s: kernel volume %d not in 1..32", who, k);
    // variant (u2mkd_debug_conv_forward_sorted only; the product entries pass 0):
    //   0 / 1 = heuristic;
    //   waves * 100 + kc = conv_os2 (waves in {4,8,16}, kc in {32,64});  3000 + rb * 100 + kc = conv_os3
    //   (+10000: 64-column workgroups);  50000 + nb * 1000 + waves * 100 + kc = conv_os2 with nb column blocks.
    if (variant == 1) variant = 0;
    const int c16 = (cout + 15) / 16;
    int nb = pick_nb(c16);
    if (variant >= 50000) {   // 50000 + nb * 1000 + waves * 100 + kc: conv_os2 with nb 16-column blocks per workgroup
        nb = (variant - 50000) / 1000;
        variant = (variant - 50000) % 1000;
        U2_REQUIRE(nb >= 1 && nb <= 8, "%s: bad column block count %d", who, nb);
    }
    // variant: 0 = heuristic; 3000 + kc = column-split kernel (conv_os3); otherwise waves * 100 + kc
    if (variant == 0 && cout % 128 == 0 && cin >= 16) {   // (cout == 64: conv_os2 is ~8 % faster, ab_conv.py)
        variant = 3000 + (cin % 64 == 0 ? 64 : 32);
        // measured (tools/ab_conv.py): a 64-row tile walks its offsets serially at ~10k cycles per
        // stage, so the kernel is latency-bound unless >= ~4 workgroups per CU are resident; with
        // few row tiles use 64-column workgroups (more of them) even though A is gathered twice
        if (cout % 128 == 0 && ceil_div(n_rows, 64) * (cout / 128) < 1024) variant += 10000;
    }
    if (variant >= 3000) {
        // 3000 + RB*100 + KC (RB in {4, 8} row blocks per tile; plain 3000 + KC means RB = 4)
        const bool narrow = variant >= 13000;        // +10000: 64-column tiles even when cout % 128 == 0
        if (narrow) variant -= 10000;
        int rb3 = (variant - 3000) / 100, kc3 = (variant - 3000) % 100;
        if (rb3 == 0) rb3 = 4;
        U2_REQUIRE(cout % 64 == 0 && (kc3 == 32 || kc3 == 64) && (rb3 == 4 || (rb3 == 8 && !rr.tile_order)),
                   "%s: bad variant %d", who, variant);
        const int nbw = (cout % 128 == 0 && !narrow) ? 2 : 1;
        dim3 grid3((unsigned)ceil_div(n_rows, 16 * rb3), (unsigned)(cout / (64 * nbw)));
        hipStream_t st3 = as_stream(s);
#define U2_O3(RB_, NBW_, KC_) launch_conv_os3<RB_, NBW_, KC_>(grid3, k, st3, in, cin, wt, cout, nbr, order, rr, kflip, out)
        if (rb3 == 4) {
            if (nbw == 1 && kc3 == 32) U2_O3(4, 1, 32); else if (nbw == 1) U2_O3(4, 1, 64);
            else if (kc3 == 32) U2_O3(4, 2, 32); else U2_O3(4, 2, 64);
        } else {
            if (nbw == 1 && kc3 == 32) U2_O3(8, 1, 32); else if (nbw == 1) U2_O3(8, 1, 64);
            else if (kc3 == 32) U2_O3(8, 2, 32); else U2_O3(8, 2, 64);
        }
#undef U2_O3
        return check_launch(who);
    }
    int waves, kc;
    if (variant == 0) {
        kc = cin % 64 == 0 ? 64 : 32;   // measured: ab_conv.py (64/128/256/384 -> 64; 32/96 -> 32)
        waves = 4;                      // larger tiles lose more to per-wave offset imbalance than they save on B
    } else {
        waves = variant / 100;
        kc = variant % 100;
    }
    U2_REQUIRE((waves == 4 || ((waves == 8 || waves == 16) && !rr.tile_order)) && (kc == 32 || kc == 64),
               "%s: bad variant %d", who, variant);
    dim3 grid((unsigned)ceil_div(n_rows, 16 * waves), (unsigned)ceil_div(c16, nb));
    hipStream_t st = as_stream(s);
    int rc;
#define U2_V(W, KCV) rc = launch_conv_os2<W, KCV>(nb, grid, k, st, in, cin, wt, cout, nbr, order, rr, kflip, out)
    if (waves == 4 && kc == 32) U2_V(4, 32);
    else if (waves == 4) U2_V(4, 64);
    else if (waves == 8 && kc == 32) U2_V(8, 32);
    else if (waves == 8) U2_V(8, 64);
    else if (kc == 32) U2_V(16, 32);
    else U2_V(16, 64);
#undef U2_V
    if (rc) return rc;
    return check_launch(who);
}

int u2mkd_conv_forward_sorted(const float *in, int64_t n_in, int32_t cin, const float *wt, int32_t cout,
                               const int32_t *nbr_sorted, const int32_t *order, const int32_t *tile_order, int64_t n_out,
                               int32_t k, int32_t kflip, float *out, u2mkd_stream_t s) {
    U2_REQUIRE(kflip == 0 || kflip == 1, "u2mkd_conv_forward_sorted: kflip must be 0 or 1");
    return conv_forward_impl("u2mkd_conv_forward_sorted", in, n_in, cin, wt, cout, nbr_sorted, order,
                             RowRange{n_out, 0, n_out, tile_order}, k, kflip, 0, out, s);
}

int u2mkd_debug_conv_forward_sorted(const float *in, int64_t n_in, int32_t cin, const float *wt, int32_t cout,
                                     const int32_t *nbr_sorted, const int32_t *order, const int32_t *tile_order,
                                     int64_t n_out, int32_t k, int32_t kflip, int32_t variant, float *out,
                                     u2mkd_stream_t s) {
    U2_REQUIRE(kflip == 0 || kflip == 1, "u2mkd_debug_conv_forward_sorted: kflip must be 0 or 1");
    return conv_forward_impl("u2mkd_debug_conv_forward_sorted", in, n_in, cin, wt, cout, nbr_sorted, order,
                             RowRange{n_out, 0, n_out, tile_order}, k, kflip, variant, out, s);
}

int32_t u2mkd_conv_tiles_supported(int32_t cin, int32_t cout, int32_t k) { return conv_tp_supported(cin, cout, k) ? 1 : 0; }

/* the arithmetic (u2mkd_weight_fragments' codes) the tile kernel runs a cin -> cout layer in when asked for the default with
 * fp32 rows: 4 (f16x2) where its row scaling has an instantiation, else 2 (bf16x3); 1 under U2MKD_CONV_ARITH=f32; bf16x3
 * everywhere under U2MKD_CONV_ARITH=bf16x3.  0: the layer is not the tile kernel's. */
int32_t u2mkd_conv_tiles_arith(int32_t cin, int32_t cout, int32_t k) {
    if (!conv_tp_supported(cin, cout, k)) return 0;
    static const int pref = [] {
        const char *e = getenv("U2MKD_CONV_ARITH");
        return (e && e[0] == 'f' && e[1] == '3') ? 1 : (e && e[0] == 'b') ? 2 : 4;
    }();
    if (pref == 4) return conv_tp_f16x2_supported(cin) ? 4 : 2;
    return pref;
}

size_t u2mkd_weight_fragments_bytes(int32_t k, int32_t rows, int32_t cols, int32_t arith) {
    return weight_fragments_bytes(k, rows, cols, arith);
}

int u2mkd_weight_fragments(const float *w, int32_t k, int32_t rows, int32_t cols, int32_t transpose, int32_t arith,
                           void *wf, u2mkd_stream_t s) {
    U2_REQUIRE(w && wf, "u2mkd_weight_fragments: null pointer");
    U2_REQUIRE(arith >= 0 && arith <= 4, "u2mkd_weight_fragments: arith must be 0 (default), 1 (f32), 2 (bf16x3), 3 (bf16 storage) or 4 (f16x2)");
    U2_REQUIRE(k > 0 && rows > 0 && cols > 0 && rows % 32 == 0 && cols % 32 == 0,
               "u2mkd_weight_fragments: [%d, %d, %d]: rows and cols must be positive multiples of 32", k, rows, cols);
    U2_REQUIRE(transpose >= 0 && transpose <= 2, "u2mkd_weight_fragments: transpose must be 0, 1 or 2 (both)");
    return launch_weight_fragments(w, k, rows, cols, transpose, arith, reinterpret_cast<float *>(wf), as_stream(s));
}

int u2mkd_weight_fragments_batch(const int64_t *jobs, int32_t n_jobs, int64_t total_units, u2mkd_stream_t s) {
    if (n_jobs == 0) return 0;
    U2_REQUIRE(jobs && n_jobs > 0 && total_units > 0 && total_units < (1LL << 31), "u2mkd_weight_fragments_batch: bad job table");
    return launch_weight_fragments_batch(jobs, n_jobs, total_units, as_stream(s));
}

int u2mkd_conv_forward_tiles(const float *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                             const int32_t *nbr_sorted, const int32_t *order, const int32_t *items,
                             const int32_t *n_items, int64_t n_out, int32_t k, int32_t kflip, int32_t arith, float *out,
                             u2mkd_stream_t s) {
    if (n_out <= 0) return 0;
    U2_REQUIRE(in && wf && nbr_sorted && out, "u2mkd_conv_forward_tiles: null pointer");
    U2_REQUIRE(kflip == 0 || kflip == 1, "u2mkd_conv_forward_tiles: kflip must be 0 or 1");
    U2_REQUIRE((arith >= 0 && arith <= 2) || arith == 4, "u2mkd_conv_forward_tiles: arith must be 0 (default), 1 (f32), 2 (bf16x3) or 4 (f16x2)");
    U2_REQUIRE(n_in > 0, "u2mkd_conv_forward_tiles: empty input");
    U2_REQUIRE(n_in <= (1 << 25), "u2mkd_conv_forward_tiles: %lld input rows, the tile kernel packs row indices into 25 bits", (long long)n_in);
    U2_REQUIRE((items == nullptr) == (n_items == nullptr), "u2mkd_conv_forward_tiles: items and n_items go together");
    int rc = launch_conv_tp("u2mkd_conv_forward_tiles", in, cin, reinterpret_cast<const float *>(wf), cout, nbr_sorted, order,
                            RowRange{n_out, 0, n_out, nullptr}, items, n_items, k, kflip, arith, out, as_stream(s));
    U2_REQUIRE(rc >= 0, "u2mkd_conv_forward_tiles: no instantiation for %d -> %d channels, kernel volume %d "
               "(ask u2mkd_conv_tiles_supported first)", cin, cout, k);
    return rc;
}

int u2mkd_conv_forward_tiles_ep(const float *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                const int32_t *nbr_sorted, const int32_t *order, const int32_t *items,
                                const int32_t *n_items, int64_t n_out, int32_t k, int32_t kflip, int32_t arith,
                                const float *scale, const float *shift, const float *res, int32_t relu, float *out,
                                u2mkd_stream_t s) {
    if (n_out <= 0) return 0;
    U2_REQUIRE(in && wf && nbr_sorted && out && scale && shift, "u2mkd_conv_forward_tiles_ep: null pointer");
    U2_REQUIRE(kflip == 0 || kflip == 1, "u2mkd_conv_forward_tiles_ep: kflip must be 0 or 1");
    U2_REQUIRE((arith >= 0 && arith <= 2) || arith == 4, "u2mkd_conv_forward_tiles_ep: arith must be 0 (default), 1 (f32), 2 (bf16x3) or 4 (f16x2)");
    U2_REQUIRE(n_in > 0 && n_in <= (1 << 25), "u2mkd_conv_forward_tiles_ep: %lld input rows out of range", (long long)n_in);
    U2_REQUIRE((items == nullptr) == (n_items == nullptr), "u2mkd_conv_forward_tiles_ep: items and n_items go together");
    RowRange rr{n_out, 0, n_out, nullptr};
    rr.ep_scale = scale; rr.ep_shift = shift; rr.ep_res = res; rr.ep_relu = (int)relu;
    int rc = launch_conv_tp("u2mkd_conv_forward_tiles_ep", in, cin, reinterpret_cast<const float *>(wf), cout, nbr_sorted, order,
                            rr, items, n_items, k, kflip, arith, out, as_stream(s));
    U2_REQUIRE(rc >= 0, "u2mkd_conv_forward_tiles_ep: no instantiation for %d -> %d channels, kernel volume %d", cin, cout, k);
    return rc;
}

int u2mkd_debug_conv_tile_pairs_stamps(const float *in, int64_t n_in, const void *wf, const int32_t *nbr_sorted,
                                       const int32_t *order, const int32_t *items, const int32_t *n_items, int64_t n_out,
                                       int32_t k, int32_t arith, float *out, uint64_t *stamps, u2mkd_stream_t s) {
    U2_REQUIRE(in && wf && nbr_sorted && out && stamps && n_out > 0, "u2mkd_debug_conv_tile_pairs_stamps: null pointer");
    U2_REQUIRE(n_in <= (1 << 25), "u2mkd_debug_conv_tile_pairs_stamps: too many input rows");
    int rc = launch_conv_tp("u2mkd_debug_conv_tile_pairs_stamps", in, 64, reinterpret_cast<const float *>(wf), 64, nbr_sorted,
                            order, RowRange{n_out, 0, n_out, nullptr}, items, n_items, k, 0, arith, out, as_stream(s),
                            reinterpret_cast<unsigned long long *>(stamps));
    return rc < 0 ? 2 : rc;
}

int u2mkd_conv_forward(const float *in, int64_t n_in, int32_t cin, const float *wt, int32_t cout, const int32_t *nbr,
                       int64_t n_out, int32_t k, int32_t kflip, float *out, u2mkd_stream_t s) {
    return u2mkd_conv_forward_sorted(in, n_in, cin, wt, cout, nbr, nullptr, nullptr, n_out, k, kflip, out, s);
}

int u2mkd_conv_forward_pairs(const float *in, int64_t n_in, int32_t cin, const float *wt, int32_t cout,
                              const int32_t *pair_idx, const int32_t *tile_k, const int32_t *meta, int64_t capacity,
                              int32_t k, int32_t variant, float *y, u2mkd_stream_t s) {
    if (capacity == 0) return 0;
    U2_REQUIRE(in && wt && pair_idx && tile_k && meta && y, "u2mkd_conv_forward_pairs: null pointer");
    U2_REQUIRE(cin > 0 && cin % 4 == 0 && cout > 0 && k > 0 && n_in >= 0 && capacity % 64 == 0,
               "u2mkd_conv_forward_pairs: cin=%d must be a positive multiple of 4, the capacity a multiple of 64", cin);
    // variant: 0 = heuristic, 32 / 64 = channels per stage (measured: tools/ab_hybrid.py)
    if (variant == 0) variant = ((int64_t)cin * cout >= 16384 && cin % 64 == 0) ? 64 : 32;
    U2_REQUIRE(variant == 32 || variant == 64, "u2mkd_conv_forward_pairs: bad variant %d", variant);
    const int c16 = (cout + 15) / 16;
    const int nb = pick_nb(c16);
    int64_t gx = capacity / 64;
    if (gx > 2048) gx = 2048;   // 8 workgroups per CU, each a contiguous run of the device-side tile count
    dim3 grid((unsigned)gx, (unsigned)ceil_div(c16, nb));
    int rc = variant == 32 ? launch_conv_pairs<32>(nb, grid, as_stream(s), in, cin, wt, cout, pair_idx, tile_k, meta + 1, 0, nullptr, y)
                           : launch_conv_pairs<64>(nb, grid, as_stream(s), in, cin, wt, cout, pair_idx, tile_k, meta + 1, 0, nullptr, y);
    if (rc) return rc;
    return check_launch("u2mkd_conv_forward_pairs");
}

int32_t u2mkd_conv_pairs_x3_supported(int32_t cin, int32_t cout) {
    return (conv_px3_supported(cin, cout) && conv_tp_arith(0) == 2) ? 1 : 0;
}

int u2mkd_conv_forward_pairs_x3(const float *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                 const int32_t *pair_idx, const int32_t *tile_k, const int32_t *meta, int64_t capacity,
                                 int32_t k, float *y, u2mkd_stream_t s) {
    if (capacity == 0) return 0;
    U2_REQUIRE(in && wf && pair_idx && tile_k && meta && y, "u2mkd_conv_forward_pairs_x3: null pointer");
    U2_REQUIRE(k > 0 && n_in > 0 && capacity % 64 == 0, "u2mkd_conv_forward_pairs_x3: the capacity must be a multiple of 64");
    int rc = launch_conv_px3("u2mkd_conv_forward_pairs_x3", in, cin, reinterpret_cast<const float *>(wf), cout, pair_idx,
                             tile_k, meta + 1, capacity, y, as_stream(s));
    U2_REQUIRE(rc >= 0, "u2mkd_conv_forward_pairs_x3: cin=%d and cout=%d must be multiples of 32 "
               "(ask u2mkd_conv_pairs_x3_supported first)", cin, cout);
    return rc;
}

/* the pair-schedule product / the dense product in f16x2 arithmetic: wf = the arith-4 fragments (u2mkd_weight_fragments) */
int32_t u2mkd_conv_pairs_f16x2_supported(int32_t cin, int32_t cout) {
    static const int pref = [] {
        const char *e = getenv("U2MKD_CONV_ARITH");
        return (e && (e[0] == 'b' || (e[0] == 'f' && e[1] == '3'))) ? 0 : 1;      // bf16x3 / f32 asked for
    }();
    return (pref && conv_px3_supported(cin, cout)) ? 1 : 0;
}

int u2mkd_conv_forward_pairs_f16x2(const float *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                    const int32_t *pair_idx, const int32_t *tile_k, const int32_t *meta, int64_t capacity,
                                    int32_t k, float *y, u2mkd_stream_t s) {
    if (capacity == 0) return 0;
    U2_REQUIRE(in && wf && pair_idx && tile_k && meta && y, "u2mkd_conv_forward_pairs_f16x2: null pointer");
    U2_REQUIRE(k > 0 && n_in > 0 && capacity % 64 == 0, "u2mkd_conv_forward_pairs_f16x2: the capacity must be a multiple of 64");
    int rc = launch_conv_px3("u2mkd_conv_forward_pairs_f16x2", in, cin, reinterpret_cast<const float *>(wf), cout, pair_idx,
                             tile_k, meta + 1, capacity, y, as_stream(s), false, k);
    U2_REQUIRE(rc >= 0, "u2mkd_conv_forward_pairs_f16x2: cin=%d and cout=%d must be multiples of 32", cin, cout);
    return rc;
}

int u2mkd_linear_forward_f16x2(const float *x, int64_t n, int32_t cin, const void *wf, int32_t cout, const float *bias,
                               float *y, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(x && wf && y, "u2mkd_linear_forward_f16x2: null pointer");
    U2_REQUIRE(n > 0 && n < ((int64_t)1 << 31) - 64, "u2mkd_linear_forward_f16x2: %lld rows out of range", (long long)n);
    int rc = launch_linear_px3("u2mkd_linear_forward_f16x2", x, n, cin, reinterpret_cast<const float *>(wf), cout, bias, y,
                               as_stream(s), false, true);
    U2_REQUIRE(rc >= 0, "u2mkd_linear_forward_f16x2: cin=%d and cout=%d must be multiples of 32", cin, cout);
    return rc;
}

int u2mkd_linear_forward(const float *x, int64_t n, int32_t cin, const float *w, int32_t cout, const float *bias,
                         int32_t variant, float *y, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(x && w && y, "u2mkd_linear_forward: null pointer");
    U2_REQUIRE(cin > 0 && cin % 4 == 0 && cout > 0, "u2mkd_linear_forward: cin=%d must be a positive multiple of 4", cin);
    if (variant == 0) variant = ((int64_t)cin * cout >= 16384 && cin % 64 == 0) ? 64 : 32;
    U2_REQUIRE(variant == 32 || variant == 64, "u2mkd_linear_forward: bad variant %d", variant);
    const int c16 = (cout + 15) / 16;
    const int nb = pick_nb(c16);
    int64_t gx = ceil_div(n, 64);
    if (gx > 2048) gx = 2048;
    dim3 grid((unsigned)gx, (unsigned)ceil_div(c16, nb));
    int rc = variant == 32 ? launch_conv_pairs<32>(nb, grid, as_stream(s), x, cin, w, cout, nullptr, nullptr, nullptr, n, bias, y)
                           : launch_conv_pairs<64>(nb, grid, as_stream(s), x, cin, w, cout, nullptr, nullptr, nullptr, n, bias, y);
    if (rc) return rc;
    return check_launch("u2mkd_linear_forward");
}

int u2mkd_linear_forward_x3(const float *x, int64_t n, int32_t cin, const void *wf, int32_t cout, const float *bias,
                            float *y, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(x && wf && y, "u2mkd_linear_forward_x3: null pointer");
    U2_REQUIRE(n > 0 && n < ((int64_t)1 << 31) - 64, "u2mkd_linear_forward_x3: %lld rows out of range", (long long)n);
    int rc = launch_linear_px3("u2mkd_linear_forward_x3", x, n, cin, reinterpret_cast<const float *>(wf), cout, bias, y,
                               as_stream(s));
    U2_REQUIRE(rc >= 0, "u2mkd_linear_forward_x3: cin=%d and cout=%d must be multiples of 32 "
               "(ask u2mkd_conv_pairs_x3_supported first)", cin, cout);
    return rc;
}

int u2mkd_pairs_gather_sum(const float *y, const int32_t *pos, int64_t n_rows, int32_t k, int32_t cout, float *out,
                            u2mkd_stream_t s) {
    if (n_rows == 0) return 0;
    U2_REQUIRE(y && pos && out, "u2mkd_pairs_gather_sum: null pointer");
    U2_REQUIRE(cout > 0 && cout % 4 == 0 && k > 0, "u2mkd_pairs_gather_sum: cout=%d must be a positive multiple of 4", cout);
    const int c4 = cout / 4;
    const int64_t total = n_rows * c4;
    hipLaunchKernelGGL(pairs_gather_sum_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), y, pos,
                       n_rows, k, c4, out, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0);
    return check_launch("u2mkd_pairs_gather_sum");
}

int32_t u2mkd_pairs_gather_sum_stats_slab_rows(void) { return kGsSlabRows; }

int32_t u2mkd_pairs_gather_sum_stats_supported(int32_t cout) { return cout >= 8 && cout % 4 == 0 && cout <= 512; }

int u2mkd_pairs_gather_sum_stats(const float *y, const int32_t *pos, int64_t n_rows, int32_t k, int32_t cout, float *out,
                                 float *partial, u2mkd_stream_t s) {
    if (n_rows == 0) return 0;
    U2_REQUIRE(y && pos && out && partial, "u2mkd_pairs_gather_sum_stats: null pointer");
    U2_REQUIRE(k > 0 && u2mkd_pairs_gather_sum_stats_supported(cout), "u2mkd_pairs_gather_sum_stats: cout=%d must be a multiple of 4 in 8..512", cout);
    const int64_t slabs = ceil_div(n_rows, (int64_t)kGsSlabRows);
    U2_REQUIRE(slabs < ((int64_t)1 << 31), "u2mkd_pairs_gather_sum_stats: too many rows");
    hipLaunchKernelGGL(pairs_gather_sum_stats_kernel, dim3((unsigned)slabs), dim3(256), 0, as_stream(s), y, pos, n_rows, k, cout / 4,
                       out, partial);
    return check_launch("u2mkd_pairs_gather_sum_stats");
}

int u2mkd_pairs_gather_sum_ep(const float *y, const int32_t *pos, int64_t n_rows, int32_t k, int32_t cout, const float *scale,
                              const float *shift, const float *res, int32_t relu, float *out, u2mkd_stream_t s) {
    if (n_rows == 0) return 0;
    U2_REQUIRE(y && pos && out && scale && shift, "u2mkd_pairs_gather_sum_ep: null pointer");
    U2_REQUIRE(cout > 0 && cout % 4 == 0 && k > 0, "u2mkd_pairs_gather_sum_ep: cout=%d must be a positive multiple of 4", cout);
    const int c4 = cout / 4;
    const int64_t total = n_rows * c4;
    hipLaunchKernelGGL(pairs_gather_sum_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), y, pos,
                       n_rows, k, c4, out, scale, shift, res, (int)relu);
    return check_launch("u2mkd_pairs_gather_sum_ep");
}

int u2mkd_conv_forward_pairs_bf16(const void *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                   const int32_t *pair_idx, const int32_t *tile_k, const int32_t *meta, int64_t capacity,
                                   int32_t k, void *y, u2mkd_stream_t s) {
    if (capacity == 0) return 0;
    U2_REQUIRE(in && wf && pair_idx && tile_k && meta && y, "u2mkd_conv_forward_pairs_bf16: null pointer");
    U2_REQUIRE(k > 0 && n_in > 0 && capacity % 64 == 0, "u2mkd_conv_forward_pairs_bf16: the capacity must be a multiple of 64");
    int rc = launch_conv_px3("u2mkd_conv_forward_pairs_bf16", reinterpret_cast<const float *>(in), cin,
                             reinterpret_cast<const float *>(wf), cout, pair_idx, tile_k, meta + 1, capacity,
                             reinterpret_cast<float *>(y), as_stream(s), true);
    U2_REQUIRE(rc >= 0, "u2mkd_conv_forward_pairs_bf16: cin=%d and cout=%d must be multiples of 32", cin, cout);
    return rc;
}

int u2mkd_linear_forward_bf16(const void *x, int64_t n, int32_t cin, const void *wf, int32_t cout, const float *bias,
                              void *y, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(x && wf && y, "u2mkd_linear_forward_bf16: null pointer");
    U2_REQUIRE(n > 0 && n < ((int64_t)1 << 31) - 64, "u2mkd_linear_forward_bf16: %lld rows out of range", (long long)n);
    int rc = launch_linear_px3("u2mkd_linear_forward_bf16", reinterpret_cast<const float *>(x), n, cin,
                               reinterpret_cast<const float *>(wf), cout, bias, reinterpret_cast<float *>(y), as_stream(s), true);
    U2_REQUIRE(rc >= 0, "u2mkd_linear_forward_bf16: cin=%d and cout=%d must be multiples of 32", cin, cout);
    return rc;
}

int u2mkd_pairs_gather_sum_bf16(const void *y, const int32_t *pos, int64_t n_rows, int32_t k, int32_t cout, void *out,
                                u2mkd_stream_t s) {
    if (n_rows == 0) return 0;
    U2_REQUIRE(y && pos && out, "u2mkd_pairs_gather_sum_bf16: null pointer");
    U2_REQUIRE(cout > 0 && cout % 8 == 0 && k > 0, "u2mkd_pairs_gather_sum_bf16: cout=%d must be a positive multiple of 8", cout);
    const int c8 = cout / 8;
    const int64_t total = n_rows * c8;
    hipLaunchKernelGGL(pairs_gather_sum_bf16_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const uint4 *>(y), pos, n_rows, k, c8, reinterpret_cast<uint4 *>(out));
    return check_launch("u2mkd_pairs_gather_sum_bf16");
}

int u2mkd_debug_wgrad_stamps(const float *a, const float *b, const int32_t *pairs, const int32_t *plan, int64_t n_rows,
                             int32_t k, void *workspace, uint64_t *stamps, u2mkd_stream_t s) {
    // 64 x 64 channels, 32-pair chunks; slabs only (no reduction)
    constexpr int SA_ = 80;
    size_t lds = (size_t)2 * 32 * (SA_ + SA_) * 4;
    dim3 grid(wgrad_g_target(n_rows, k) + k, 1);
    hipLaunchKernelGGL((conv_wgrad_pairs_kernel<2, 2, 32, true>), grid, dim3(256), lds, as_stream(s), a, 64, b, 64, pairs, plan,
                       k, 0, 1, reinterpret_cast<float *>(workspace), reinterpret_cast<unsigned long long *>(stamps));
    return check_launch("u2mkd_debug_wgrad_stamps");
}

int32_t u2mkd_wgrad_plan_ints(int32_t k) { return 4 + 2 * k; }

int u2mkd_wgrad_plan(const int32_t *nbsizes, int32_t k, int64_t n_rows, int32_t *plan, u2mkd_stream_t s) {
    U2_REQUIRE(nbsizes && plan, "u2mkd_wgrad_plan: null pointer");
    U2_REQUIRE(k > 0 && k <= 64, "u2mkd_wgrad_plan: kernel volume %d not in 1..64", k);
    hipLaunchKernelGGL(wgrad_plan_kernel, dim3(1), dim3(64), 0, as_stream(s), nbsizes, k, wgrad_g_target(n_rows, k), 64,
                       plan);
    return check_launch("u2mkd_wgrad_plan");
}

size_t u2mkd_conv_wgrad_pairs_workspace_bytes(int64_t n_rows, int32_t ca, int32_t cb, int32_t k) {
    return ((size_t)wgrad_g_target(n_rows, k) + k) * (size_t)ca * cb * sizeof(float);
}

static int wgrad_pairs_impl(bool b16, const float *a, int32_t ca, const float *b, int32_t cb, const int32_t *pairs,
                            const int32_t *plan, int64_t n_rows, int32_t k, int32_t swap, void *workspace,
                            size_t workspace_bytes, float *dw, u2mkd_stream_t s) {
    U2_REQUIRE(dw, "u2mkd_conv_wgrad_pairs: null pointer");
    if (n_rows == 0) {   // empty map: the gradient is zero
        (void)hipMemsetAsync(dw, 0, (size_t)k * ca * cb * sizeof(float), as_stream(s));
        return check_launch("u2mkd_conv_wgrad_pairs");
    }
    U2_REQUIRE(a && b && pairs && plan && workspace, "u2mkd_conv_wgrad_pairs: null pointer");
    U2_REQUIRE(ca > 0 && cb > 0 && ca % 4 == 0 && cb % 4 == 0,
               "u2mkd_conv_wgrad_pairs: ca=%d cb=%d must be positive multiples of 4", ca, cb);
    U2_REQUIRE(k > 0 && k <= 64, "u2mkd_conv_wgrad_pairs: kernel volume %d not in 1..64", k);
    const int g = wgrad_g_target(n_rows, k) + k;
    U2_REQUIRE(workspace_bytes >= (size_t)g * ca * cb * sizeof(float), "u2mkd_conv_wgrad_pairs: workspace too small");
    hipStream_t st = as_stream(s);
    // bf16x3 form (conv_wgrad_x3.hip, 64 x 64-channel tiles) for every shape whose channel counts are multiples of 64
    const bool x3_shape = ca % 64 == 0 && cb % 64 == 0;
    if (x3_shape && (b16 || conv_tp_arith(0) == 2) && conv_wgrad_x3_supported(ca, cb, k)) {
        // (bf16 rows: the same kernel without the split -- one plane, one MFMA per product)
        const int merge = conv_wgrad_x3_merge(ca, cb);
        int rc = launch_conv_wgrad_x3(a, ca, b, cb, pairs, plan, k, swap, g, merge, reinterpret_cast<float *>(workspace), st, b16);
        if (rc) return rc;
        const int64_t te = (int64_t)ca * cb;
        hipLaunchKernelGGL(wgrad_pairs_reduce_kernel, dim3((unsigned)ceil_div(te, 64), k), dim3(256), 0, st,
                           reinterpret_cast<float *>(workspace), plan, k, te, dw, merge);
        return check_launch("u2mkd_conv_wgrad_pairs");
    }
    // 32*W-channel tiles per operand: 96-channel layers get exact 96-wide tiles (W = 3)
    auto pick = [](int c) { return c <= 32 ? 1 : (c <= 64 ? 2 : (c <= 96 ? 3 : 4)); };
    const int wm = pick(ca), wn = pick(cb);
    const int tiles_a = (ca + 32 * wm - 1) / (32 * wm), tiles_b = (cb + 32 * wn - 1) / (32 * wn);
    dim3 grid(g, tiles_a * tiles_b);
    float *slabs = reinterpret_cast<float *>(workspace);
    // pairs staged per step: fewer pairs = less LDS = more resident workgroups per CU to hide the
    // gather latency.  Measured (tools/ab_conv.py): 128-wide tiles 16 > 32 > 64 pairs per step
    // (256x256 at stride 8: 197 / 228 / 333 us), 32/64-wide tiles best at 32.
    const int cp = (wm >= 3 || wn >= 3) ? 16 : 32;
#define U2_WPC(WM_, WN_, CPV)                                                                                       \
    do {                                                                                                            \
        constexpr int CP_ = CPV;                                                                                    \
        constexpr int TA_ = 32 * WM_, TB_ = 32 * WN_;                                                               \
        constexpr int SA_ = TA_ + 16 - (TA_ % 32 == 16 ? 16 : 0), SB_ = TB_ + 16 - (TB_ % 32 == 16 ? 16 : 0);       \
        size_t lds = (size_t)2 * CP_ * (SA_ + SB_) * 4;                                                             \
        if (lds > 65536)                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_pairs_kernel<WM_, WN_, CP_>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        if (b16) {                                                                                                  \
            if (lds > 65536)                                                                                        \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_pairs_kernel<WM_, WN_, CP_, false, true>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
            hipLaunchKernelGGL((conv_wgrad_pairs_kernel<WM_, WN_, CP_, false, true>), grid, dim3(256), lds, st, a, ca, b, cb, \
                               pairs, plan, k, swap, tiles_b, slabs);                                               \
        } else                                                                                                      \
        hipLaunchKernelGGL((conv_wgrad_pairs_kernel<WM_, WN_, CP_>), grid, dim3(256), lds, st, a, ca, b, cb, pairs, \
                           plan, k, swap, tiles_b, slabs);                                                          \
    } while (0)
#define U2_WP(WM_, WN_)                                                     \
    do {                                                                    \
        if (cp == 16) U2_WPC(WM_, WN_, 16);                                 \
        else U2_WPC(WM_, WN_, 32);                                          \
    } while (0)
#define U2_WN(WM_)                                                          \
    do {                                                                    \
        if (wn == 1) U2_WP(WM_, 1);                                         \
        else if (wn == 2) U2_WP(WM_, 2);                                    \
        else if (wn == 3) U2_WP(WM_, 3);                                    \
        else U2_WP(WM_, 4);                                                 \
    } while (0)
    if (wm == 1) U2_WN(1);
    else if (wm == 2) U2_WN(2);
    else if (wm == 3) U2_WN(3);
    else U2_WN(4);
#undef U2_WN
#undef U2_WP
#undef U2_WPC
    int64_t tile_elems = (int64_t)ca * cb;
    hipLaunchKernelGGL(wgrad_pairs_reduce_kernel, dim3((unsigned)ceil_div(tile_elems, 64), k), dim3(256), 0, st,
                       slabs, plan, k, tile_elems, dw);
    return check_launch("u2mkd_conv_wgrad_pairs");
}

int u2mkd_conv_wgrad_pairs(const float *a, int32_t ca, const float *b, int32_t cb, const int32_t *pairs,
                           const int32_t *plan, int64_t n_rows, int32_t k, int32_t swap, void *workspace,
                           size_t workspace_bytes, float *dw, u2mkd_stream_t s) {
    return wgrad_pairs_impl(false, a, ca, b, cb, pairs, plan, n_rows, k, swap, workspace, workspace_bytes, dw, s);
}

int u2mkd_conv_wgrad_pairs_bf16(const void *a, int32_t ca, const void *b, int32_t cb, const int32_t *pairs,
                                const int32_t *plan, int64_t n_rows, int32_t k, int32_t swap, void *workspace,
                                size_t workspace_bytes, float *dw, u2mkd_stream_t s) {
    return wgrad_pairs_impl(true, reinterpret_cast<const float *>(a), ca, reinterpret_cast<const float *>(b), cb, pairs, plan,
                            n_rows, k, swap, workspace, workspace_bytes, dw, s);
}

int u2mkd_conv_forward_tiles_bf16(const void *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                  const int32_t *nbr_sorted, const int32_t *order, const int32_t *items,
                                  const int32_t *n_items, int64_t n_out, int32_t k, int32_t kflip, void *out,
                                  u2mkd_stream_t s) {
    if (n_out <= 0) return 0;
    U2_REQUIRE(in && wf && nbr_sorted && out, "u2mkd_conv_forward_tiles_bf16: null pointer");
    U2_REQUIRE(kflip == 0 || kflip == 1, "u2mkd_conv_forward_tiles_bf16: kflip must be 0 or 1");
    U2_REQUIRE(n_in > 0, "u2mkd_conv_forward_tiles_bf16: empty input");
    U2_REQUIRE(n_in <= (1 << 25), "u2mkd_conv_forward_tiles_bf16: %lld input rows, the tile kernel packs row indices into 25 bits", (long long)n_in);
    U2_REQUIRE((items == nullptr) == (n_items == nullptr), "u2mkd_conv_forward_tiles_bf16: items and n_items go together");
    int rc = launch_conv_tp("u2mkd_conv_forward_tiles_bf16", reinterpret_cast<const float *>(in), cin,
                            reinterpret_cast<const float *>(wf), cout, nbr_sorted, order, RowRange{n_out, 0, n_out, nullptr},
                            items, n_items, k, kflip, 3, reinterpret_cast<float *>(out), as_stream(s));
    U2_REQUIRE(rc >= 0, "u2mkd_conv_forward_tiles_bf16: no instantiation for %d -> %d channels, kernel volume %d "
               "(ask u2mkd_conv_tiles_supported first)", cin, cout, k);
    return rc;
}

}  // extern "C"
