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
#include "common.h"

namespace u2mkd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

template <int MR, int NB>
__global__ void __launch_bounds__(256)
conv_os_kernel(const float *__restrict__ in, int cin, const float *__restrict__ wt, int cout,
               const int32_t *__restrict__ nbr, int64_t n_out, int K, int kflip, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t row0 = tile * (16 * MR);
    if (row0 >= n_out) return;  // wave-uniform
    const int col0 = blockIdx.y * (16 * NB);

    f32x4 acc[MR][NB];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int64_t rows[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) rows[m] = row0 + 16 * m + r;

    int idx_next[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) idx_next[m] = rows[m] < n_out ? nbr[rows[m]] : -1;

    for (int k = 0; k < K; ++k) {
        int idx[MR];
        bool mine = false;
#pragma unroll
        for (int m = 0; m < MR; ++m) { idx[m] = idx_next[m]; mine |= idx[m] >= 0; }
        if (k + 1 < K) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
                idx_next[m] = rows[m] < n_out ? nbr[(int64_t)(k + 1) * n_out + rows[m]] : -1;
        }
        if (__ballot(mine) == 0ULL) continue;  // no row of this tile has a neighbour at offset k
        const float *wk = wt + (size_t)(kflip ? K - 1 - k : k) * cout * cin;
        for (int c0 = 0; c0 < cin; c0 += 16) {
            const int ci = c0 + 4 * q;
            const bool cok = ci < cin;
            float4 a[MR], b[NB];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                a[m] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx[m] >= 0 && cok) a[m] = *reinterpret_cast<const float4 *>(in + (size_t)idx[m] * cin + ci);
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                b[n] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cok && col0 + 16 * n + r < cout)
                    b[n] = *reinterpret_cast<const float4 *>(wk + (size_t)(col0 + 16 * n + r) * cin + ci);
            }
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b[n].x, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b[n].y, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b[n].z, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b[n].w, acc[m][n], 0, 0, 0);
                }
        }
    }
    // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            int64_t row = row0 + 16 * m + 4 * q + reg;
            if (row < n_out) {
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    if (col0 + 16 * n + r < cout) out[row * cout + col0 + 16 * n + r] = acc[m][n][reg];
            }
        }
}

// ---- weight gradient ---------------------------------------------------------
// grid = (S splits of the rows, K offsets (or 1), channel tiles).  256 threads:
// wave w owns the 16 A-channels [ta*64 + 16w, +16) x 16*NB B-channels.
template <int NB>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const float *__restrict__ a, int ca, const float *__restrict__ b, int cb,
                  const int32_t *__restrict__ nbr, int64_t n_rows, int K, int a_gathered, int k_only, int k_skip,
                  int S, int tiles_b, float *__restrict__ slabs) {
    __shared__ int s_gi[2][256];
    __shared__ int s_rj[2][256];
    __shared__ int s_wcnt[2][4];

    const int k = k_only >= 0 ? k_only : (int)blockIdx.y;
    if (k == k_skip) return;
    const int kslot = k_only >= 0 ? 0 : k;
    const int s = blockIdx.x;
    const int ta = blockIdx.z / tiles_b, tb = blockIdx.z % tiles_b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int a_ch = ta * 64 + 16 * wave + r;
    const bool wave_active = ta * 64 + 16 * wave < ca;
    const bool a_ok = a_ch < ca;
    const int b_ch0 = tb * 16 * NB + r;

    const int64_t chunks = (n_rows + 255) / 256;
    const int64_t per = (chunks + S - 1) / S;
    const int64_t c_begin = (int64_t)s * per;
    const int64_t c_end = c_begin + per < chunks ? c_begin + per : chunks;

    f32x4 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int32_t *nk = nbr + (int64_t)k * n_rows;
    int buf = 0;
    for (int64_t c = c_begin; c < c_end; ++c, buf ^= 1) {
        // --- compaction of this chunk's valid pairs (ballot + prefix sum) ---
        int64_t j = c * 256 + threadIdx.x;
        int gi = j < n_rows ? nk[j] : -1;
        bool valid = gi >= 0;
        unsigned long long m = __ballot(valid);
        if (lane == 0) s_wcnt[buf][wave] = __popcll(m);
        int rank = __popcll(m & ((1ULL << lane) - 1ULL));
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int cw = s_wcnt[buf][w];
            if (w < wave) base += cw;
            total += cw;
        }
        if (valid) {
            s_gi[buf][base + rank] = gi;
            s_rj[buf][base + rank] = (int)j;
        }
        __syncthreads();
        if (!wave_active) continue;
        // --- MFMA over the pairs: 4 pairs per step (reduction dim) ---
        for (int p0 = 0; p0 < total; p0 += 4) {
            int p = p0 + q;
            bool pv = p < total;
            int g = pv ? s_gi[buf][p] : 0;
            int jj = pv ? s_rj[buf][p] : 0;
            int64_t arow = a_gathered ? g : jj;
            int64_t brow = a_gathered ? jj : g;
            float av = (pv && a_ok) ? a[arow * ca + a_ch] : 0.f;
            float bv[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) bv[n] = (pv && b_ch0 + 16 * n < cb) ? b[brow * cb + b_ch0 + 16 * n] : 0.f;
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[n], acc[n], 0, 0, 0);
        }
    }
    if (!wave_active) return;
    // D[i = a channel][j = b channel]: row = 4q + reg, col = r
    float *slab = slabs + ((size_t)kslot * S + s) * (size_t)ca * cb;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        int ach = ta * 64 + 16 * wave + 4 * q + reg;
        if (ach < ca) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
                if (b_ch0 + 16 * n < cb) slab[(size_t)ach * cb + b_ch0 + 16 * n] = acc[n][reg];
        }
    }
}

// dw[k][e] = sum_s slab(k, s)[e]; two slab regions (all offsets with S0 splits,
// optional dense centre with S1 splits), fixed summation order.
__global__ void wgrad_reduce_kernel(const float *__restrict__ slabs0, int S0, const float *__restrict__ slabs1, int S1,
                                    int k_centre, int64_t tile_elems, int K, float *__restrict__ dw) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (e >= tile_elems) return;
    float acc = 0.f;
    if (k == k_centre) {
        for (int s = 0; s < S1; ++s) acc += slabs1[(size_t)s * tile_elems + e];
    } else {
        const float *p = slabs0 + (size_t)k * S0 * tile_elems + e;
        for (int s = 0; s < S0; ++s) acc += p[(size_t)s * tile_elems];
    }
    dw[(size_t)k * tile_elems + e] = acc;
}

struct WgradPlan {
    int S0, S1, k_centre;
    size_t bytes;
};

static WgradPlan wgrad_plan(int64_t n_rows, int ca, int cb, int K, int centre_dense) {
    WgradPlan p;
    int64_t chunks = (n_rows + 255) / 256;
    if (chunks < 1) chunks = 1;
    if (centre_dense) {
        p.k_centre = K / 2;
        int64_t s1 = chunks < 512 ? chunks : 512;
        int64_t s0 = (chunks + 23) / 24;
        if (s0 < 1) s0 = 1;
        if (s0 > 32) s0 = 32;
        p.S0 = (int)s0;
        p.S1 = (int)s1;
    } else {
        p.k_centre = -1;
        int64_t s0 = (chunks + 3) / 4;
        if (s0 < 1) s0 = 1;
        if (s0 > 64) s0 = 64;
        p.S0 = (int)s0;
        p.S1 = 0;
    }
    p.bytes = ((size_t)K * p.S0 + p.S1) * (size_t)ca * cb * sizeof(float);
    return p;
}

template <int MR>
static int launch_conv_os(int nb, dim3 grid, hipStream_t st, const float *in, int cin, const float *wt, int cout,
                          const int32_t *nbr, int64_t n_out, int K, int kflip, float *out) {
#define U2_CASE(N)                                                                                             \
    case N:                                                                                                    \
        hipLaunchKernelGGL((conv_os_kernel<MR, N>), grid, dim3(256), 0, st, in, cin, wt, cout, nbr, n_out, K, \
                           kflip, out);                                                                        \
        break;
    switch (nb) {
        U2_CASE(1) U2_CASE(2) U2_CASE(3) U2_CASE(4) U2_CASE(5) U2_CASE(6) U2_CASE(7) U2_CASE(8)
        default: set_error("conv: unsupported column block count %d", nb); return 2;
    }
#undef U2_CASE
    return 0;
}

// column blocks (of 16) per wave: all of them up to 8, else the largest divisor <= 8
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

int u2mkd_conv_forward(const float *in, int64_t n_in, int32_t cin, const float *wt, int32_t cout, const int32_t *nbr,
                       int64_t n_out, int32_t k, int32_t kflip, float *out, u2mkd_stream_t s) {
    if (n_out == 0) return 0;
    U2_REQUIRE(in && wt && nbr && out, "u2mkd_conv_forward: null pointer");
    U2_REQUIRE(cin > 0 && cin % 4 == 0, "u2mkd_conv_forward: cin=%d must be a positive multiple of 4", cin);
    U2_REQUIRE(cout > 0, "u2mkd_conv_forward: cout=%d must be positive", cout);
    U2_REQUIRE(k > 0 && n_in >= 0, "u2mkd_conv_forward: bad sizes");
    const int c16 = (cout + 15) / 16;
    const int nb = pick_nb(c16);
    constexpr int MR = 2;
    int64_t tiles = ceil_div(n_out, 16 * MR);
    dim3 grid((unsigned)ceil_div(tiles, 4), (unsigned)ceil_div(c16, nb));
    int rc = launch_conv_os<MR>(nb, grid, as_stream(s), in, cin, wt, cout, nbr, n_out, k, kflip, out);
    if (rc) return rc;
    return check_launch("u2mkd_conv_forward");
}

size_t u2mkd_conv_wgrad_workspace_bytes(int64_t n_rows, int32_t ca, int32_t cb, int32_t k) {
    // upper bound over both plans so callers need not know centre_dense
    size_t a = wgrad_plan(n_rows, ca, cb, k, 0).bytes, b = wgrad_plan(n_rows, ca, cb, k, 1).bytes;
    return a > b ? a : b;
}

int u2mkd_conv_wgrad(const float *a, int32_t ca, const float *b, int32_t cb, const int32_t *nbr, int64_t n_rows,
                     int32_t k, int32_t a_gathered, int32_t centre_dense, void *workspace, size_t workspace_bytes,
                     float *dw, u2mkd_stream_t s) {
    U2_REQUIRE(dw, "u2mkd_conv_wgrad: null dw");
    U2_REQUIRE(ca > 0 && cb > 0, "u2mkd_conv_wgrad: ca=%d cb=%d must be positive", ca, cb);
    hipStream_t st = as_stream(s);
    if (n_rows == 0) {
        hipError_t e = hipMemsetAsync(dw, 0, (size_t)k * ca * cb * sizeof(float), st);
        if (e != hipSuccess) { set_error("u2mkd_conv_wgrad: memset: %s", hipGetErrorString(e)); return 1; }
        return 0;
    }
    U2_REQUIRE(a && b && nbr && workspace, "u2mkd_conv_wgrad: null pointer");
    if (centre_dense) U2_REQUIRE(k % 2 == 1, "u2mkd_conv_wgrad: centre_dense needs an odd kernel volume");
    WgradPlan p = wgrad_plan(n_rows, ca, cb, k, centre_dense);
    U2_REQUIRE(workspace_bytes >= p.bytes, "u2mkd_conv_wgrad: workspace %zu < %zu bytes", workspace_bytes, p.bytes);
    const int c16 = (cb + 15) / 16;
    const int nb = pick_nb(c16);
    const int tiles_b = (int)ceil_div(c16, nb), tiles_a = (ca + 63) / 64;
    float *slabs0 = reinterpret_cast<float *>(workspace);
    float *slabs1 = slabs0 + (size_t)k * p.S0 * ca * cb;
#define U2_WG(N, GRID, KONLY, KSKIP, SS, SLAB)                                                                    \
    case N:                                                                                                       \
        hipLaunchKernelGGL((conv_wgrad_kernel<N>), GRID, dim3(256), 0, st, a, ca, b, cb, nbr, n_rows, k,          \
                           a_gathered, KONLY, KSKIP, SS, tiles_b, SLAB);                                          \
        break;
#define U2_WG_SWITCH(GRID, KONLY, KSKIP, SS, SLAB)                                                                \
    switch (nb) {                                                                                                 \
        U2_WG(1, GRID, KONLY, KSKIP, SS, SLAB) U2_WG(2, GRID, KONLY, KSKIP, SS, SLAB)                             \
        U2_WG(3, GRID, KONLY, KSKIP, SS, SLAB) U2_WG(4, GRID, KONLY, KSKIP, SS, SLAB)                             \
        U2_WG(5, GRID, KONLY, KSKIP, SS, SLAB) U2_WG(6, GRID, KONLY, KSKIP, SS, SLAB)                             \
        U2_WG(7, GRID, KONLY, KSKIP, SS, SLAB) U2_WG(8, GRID, KONLY, KSKIP, SS, SLAB)                             \
        default: set_error("wgrad: unsupported column block count %d", nb); return 2;                             \
    }
    {
        dim3 grid(p.S0, k, tiles_a * tiles_b);
        U2_WG_SWITCH(grid, -1, p.k_centre, p.S0, slabs0)
    }
    if (p.k_centre >= 0) {
        dim3 grid(p.S1, 1, tiles_a * tiles_b);
        U2_WG_SWITCH(grid, p.k_centre, -1, p.S1, slabs1)
    }
#undef U2_WG_SWITCH
#undef U2_WG
    int64_t tile_elems = (int64_t)ca * cb;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(tile_elems, 256), k), dim3(256), 0, st, slabs0,
                       p.S0, slabs1, p.S1, p.k_centre, tile_elems, k, dw);
    return check_launch("u2mkd_conv_wgrad");
}

}  // extern "C"
