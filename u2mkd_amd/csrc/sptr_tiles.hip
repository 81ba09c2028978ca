// sptr window attention, TILE form: 16 queries x 16 keys per step on the matrix pipe, relative-position terms by look-up.
//
// Same function as csrc/sptr.hip (the reference's third_party/SparseTransformer/src/sptr/rpe/relative_pos_encoding_cuda_kernel.cu
// :42-274 + attention/attention_cuda_kernel.cu:4-112 + the CSR softmax of sptr/utils.py:80-95, fused; sptr/modules.py:35-65):
//     s_ij  = q_i . k_j + q_i . Tq(r_ij) + k_j . Tk(r_ij),      Tx(r) = Tx[r0][0] + Tx[r1][1] + Tx[r2][2]
//     out_i = sum_j softmax_j(s_ij) (v_j + Tv(r_ij))
// csrc/sptr.hip walks a token's window with one thread (S lanes) per (token, head) and evaluates the three table terms per
// PAIR -- 96 multiply-adds and 96 adds next to the 16 of q . k: VALU-issue bound, 0.04 of the HBM roofline, no matrix
// instruction (profiles/r5_pmc_sptr.txt).  Here the table terms are taken per TOKEN first,
//     A_i[e] = q_i . Tq[r][ax],     B_j[e] = k_j . Tk[r][ax],     e = ax * 52 + r         (strips of 160 floats)
// as matrix products ([16 tokens x 16] x [16 x 160]: 40 v_mfma_f32_16x16x4_f32 per 16 tokens, fp32 operands), so that a pair
// costs six LDS look-ups, s_ij = q_i . k_j + sum_ax A_i[ax][r_ax] + B_j[ax][r_ax]; the value side keeps a histogram per
// query, H_i[e] = sum_j p_ij [r_ax(i, j) = r], and out_i = sum_j p_ij v_j + sum_e H_i[e] Tv[e] (40 more matrix instructions,
// once).  One wave owns 16 consecutive sorted tokens of one head and walks the keys of their windows 16 at a time:
//   S^T = K Q^T        4 matrix instructions; lane (i = l % 16, g = l / 16) holds the scores of keys 4 g .. 4 g + 3 for query i,
//                      which is P's A-operand layout for the next product (no transpose through LDS);
//   + bias             6 look-ups per pair (A strips of the 16 queries, B strips of the 16 keys, both in LDS);
//   online softmax     row maximum over 4 registers and 2 cross-lane steps; the histogram rows are rescaled only when a
//                      maximum moved (rare after the first tiles);
//   O += P V           4 matrix instructions; H += p by LDS float adds (one wave, program order: reproducible);
// software-pipelined: the next tile's keys / values / coordinates are loaded (from copies in SORTED order, a pre-pass) while the
// current tile is evaluated, and its B strips are multiplied between the current tile's look-ups and its softmax.
// Windows may end inside a tile: pairs of different windows are masked.  Nothing of size M = sum L_w^2 reaches HBM.
#include "common.h"
#include "sptr_internal.h"

namespace u2mkd {

constexpr int kAx = 52;                 // strip entries per axis (tables have <= 50 rows)
constexpr int kStrip = 160;             // floats per strip row: 3 * 52 = 156 entries, padded to 10 blocks of 16
constexpr int kBlk = kStrip / 16;
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define U2_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---- pre-pass: k and v rows in sorted order, [p][hh][16] ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sptr_sorted_kv_kernel(const float *__restrict__ k, const float *__restrict__ v, int64_t ld_qkv, const int32_t *__restrict__ sort_idx,
                      int64_t n, int h, float *__restrict__ ks, float *__restrict__ vs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one float4 of one row
    if (idx >= n * h * 4) return;
    const int c4 = (int)(idx & 3);
    const int64_t ph = idx >> 2;
    const int hh = (int)(ph % h);
    const int64_t p = ph / h;
    const size_t src = (size_t)sort_idx[p] * ld_qkv + hh * kHd;
    reinterpret_cast<float4 *>(ks)[idx] = reinterpret_cast<const float4 *>(k + src)[c4];
    reinterpret_cast<float4 *>(vs)[idx] = reinterpret_cast<const float4 *>(v + src)[c4];
}

// table [L][3][h][16] of head hh, flattened to [e = ax * kAx + r][16] in LDS (rows r >= L and the padding rows: zero)
__device__ __forceinline__ void load_flat_table(float *sT, const float *__restrict__ tab, int L, int h, int hh, int l) {
    for (int f = l; f < kStrip * 4; f += 64) {
        const int e = f >> 2, c4 = f & 3;
        const int ax = e / kAx, r = e - ax * kAx;
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ax < 3 && r < L) t4 = reinterpret_cast<const float4 *>(tab + (((size_t)r * 3 + ax) * h + hh) * kHd)[c4];
        reinterpret_cast<float4 *>(sT)[f] = t4;
    }
}

// strip[row][e] = sum_d X[row][d] T[e][d] for the 16 rows whose operand slices the lanes hold (lane: row l % 16, d = 4 g .. 4 g + 3)
__device__ __forceinline__ void strip_products(const float4 x, const float *sT, float *strip, int i, int g) {
#pragma unroll
    for (int blk = 0; blk < kBlk; ++blk) {
        const float4 t = *reinterpret_cast<const float4 *>(sT + (blk * 16 + i) * kHd + 4 * g);
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = U2_MFMA(x.x, t.x, c);
        c = U2_MFMA(x.y, t.y, c);
        c = U2_MFMA(x.z, t.z, c);
        c = U2_MFMA(x.w, t.w, c);
#pragma unroll
        for (int s = 0; s < 4; ++s) strip[(4 * g + s) * kStrip + blk * 16 + i] = c[s];
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
sptr_tile_fwd_kernel(const float *__restrict__ q, const float *__restrict__ ks, const float *__restrict__ vs,
                     const int32_t *__restrict__ sort_idx, const int32_t *__restrict__ wstart, const int32_t *__restrict__ wlen,
                     const int32_t *__restrict__ qc, const float *__restrict__ radial, const float *__restrict__ tq,
                     const float *__restrict__ tk, const float *__restrict__ tv, int L, RelCtx rc, int64_t n, int h,
                     float *__restrict__ out, float *__restrict__ lse, SptrLayout ly) {
    __shared__ __attribute__((aligned(16))) float sA[16 * kStrip];
    __shared__ __attribute__((aligned(16))) float sB[16 * kStrip];
    __shared__ __attribute__((aligned(16))) float sH[16 * kStrip];
    __shared__ __attribute__((aligned(16))) float sT[kStrip * kHd];     // Tq, then Tk, at the end Tv: flattened [e][16]
    // (per-row scalars live in the padding columns 156, 157 of the A strips: the four arrays are exactly a quarter of a CU's LDS)
#define sCorr(r) sA[(r) * kStrip + 156]
#define sInv(r) sA[(r) * kStrip + 157]
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const int hh = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * 16;

    // this lane's query (i): window, quantised coordinates, operand slice q[4 g .. 4 g + 3]
    const int64_t pi = p0 + i;
    const bool qlive = pi < n;
    const int64_t pic = qlive ? pi : n - 1;
    const int ws = wstart[pic], we = qlive ? ws + wlen[pic] : ws;
    int qci[3] = {qc[pic * 3], qc[pic * 3 + 1], qc[pic * 3 + 2]};
    const float ri = radial ? radial[pic] : 0.f;
    float4 qop = *reinterpret_cast<const float4 *>(q + (size_t)sort_idx[pic] * ly.ld_qkv + hh * kHd + 4 * g);
    qop.x *= ly.q_scale; qop.y *= ly.q_scale; qop.z *= ly.q_scale; qop.w *= ly.q_scale;

    // the keys of the 16 queries' windows are one contiguous range of sorted positions
    int lo = ws, hi = we;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    // operands of a key tile: k rows (A operand, row l % 16), v columns (B operand, rows 4 g + s), coordinates of keys 4 g + s
    const size_t hrow = (size_t)h * kHd;
    auto load_k = [&](int kb) {
        const int64_t pk = min((int64_t)kb + i, n - 1);
        return *reinterpret_cast<const float4 *>(ks + pk * hrow + hh * kHd + 4 * g);
    };
    float4 kop = load_k(lo);

    // A strips of the 16 queries (Tq), zeroed histogram; then Tk stays in sT for the key tiles
    load_flat_table(sT, tq, L, h, hh, l);
    for (int f = l; f < 16 * kStrip / 4; f += 64) reinterpret_cast<float4 *>(sH)[f] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    strip_products(qop, sT, sA, i, g);
    __syncthreads();
    load_flat_table(sT, tk, L, h, hh, l);
    __syncthreads();
    strip_products(kop, sT, sB, i, g);          // B strips of the first key tile

    float vop[4], rj[4];
    int qcj[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int64_t pj = min((int64_t)lo + 4 * g + s, n - 1);
        vop[s] = vs[pj * hrow + hh * kHd + i];
        qcj[s][0] = qc[pj * 3]; qcj[s][1] = qc[pj * 3 + 1]; qcj[s][2] = qc[pj * 3 + 2];
        rj[s] = radial ? radial[pj] : 0.f;
    }
    float m = -INFINITY, lsum = 0.f;
    f32x4 acc_o = {0.f, 0.f, 0.f, 0.f};          // O[row 4 g + v][d = i]
    bool first = true;
    for (int kb = lo; kb < hi; kb += 16) {
        // the next tile's operands: in flight while this tile is evaluated
        const bool more = kb + 16 < hi;
        float4 kop_n = kop;
        float vop_n[4], rj_n[4];
        int qcj_n[4][3];
        if (more) {
            kop_n = load_k(kb + 16);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int64_t pj = min((int64_t)kb + 16 + 4 * g + s, n - 1);
                vop_n[s] = vs[pj * hrow + hh * kHd + i];
                qcj_n[s][0] = qc[pj * 3]; qcj_n[s][1] = qc[pj * 3 + 1]; qcj_n[s][2] = qc[pj * 3 + 2];
                rj_n[s] = radial ? radial[pj] : 0.f;
            }
        }
        // S^T = K Q^T: lane holds s[v] = q_i . k_(4 g + v)
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = U2_MFMA(kop.x, qop.x, sc);
        sc = U2_MFMA(kop.y, qop.y, sc);
        sc = U2_MFMA(kop.z, qop.z, sc);
        sc = U2_MFMA(kop.w, qop.w, sc);
        __syncthreads();                          // (sB of this tile is complete)
        float s4[4];
        int rr[4][3];
        bool ok[4];
        float mt = -INFINITY;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pj = kb + 4 * g + s;
            ok[s] = pj >= ws && pj < we;
            rel_rows(rc, qci, ri, qcj[s], rj[s], rr[s]);
            const float *a = sA + i * kStrip, *b = sB + (4 * g + s) * kStrip;
            const float bias = ((a[rr[s][0]] + a[kAx + rr[s][1]]) + a[2 * kAx + rr[s][2]]) +
                               ((b[rr[s][0]] + b[kAx + rr[s][1]]) + b[2 * kAx + rr[s][2]]);
            s4[s] = sc[s] + bias;
            if (ok[s]) mt = fmaxf(mt, s4[s]);
        }
        __syncthreads();                          // (every look-up in sB is done: the next tile's strips may be written)
        if (more) strip_products(kop_n, sT, sB, i, g);
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mn = fmaxf(m, mt);
        const float corr = (m == -INFINITY) ? (mn == -INFINITY ? 1.f : 0.f) : __expf(m - mn);
        float pv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) pv[s] = ok[s] ? __expf(s4[s] - mn) : 0.f;
        lsum = lsum * corr + ((pv[0] + pv[1]) + (pv[2] + pv[3]));
        m = mn;
        if (g == 0) sCorr(i) = corr;
        const bool moved = !first && __any(corr != 1.f);
        __syncthreads();
        if (moved) {                              // a row maximum moved: its histogram row follows
            for (int row = 0; row < 16; ++row) {
                const float c = sCorr(row);
                if (c != 1.f)
                    for (int e = l; e < kStrip; e += 64) sH[row * kStrip + e] *= c;
            }
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc_o[s] *= sCorr(4 * g + s);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (ok[s]) {
                float *hrow_ = sH + i * kStrip;
                __hip_atomic_fetch_add(hrow_ + rr[s][0], pv[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(hrow_ + kAx + rr[s][1], pv[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(hrow_ + 2 * kAx + rr[s][2], pv[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // O += P V
        acc_o = U2_MFMA(pv[0], vop[0], acc_o);
        acc_o = U2_MFMA(pv[1], vop[1], acc_o);
        acc_o = U2_MFMA(pv[2], vop[2], acc_o);
        acc_o = U2_MFMA(pv[3], vop[3], acc_o);
        first = false;
        kop = kop_n;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            vop[s] = vop_n[s]; rj[s] = rj_n[s];
            qcj[s][0] = qcj_n[s][0]; qcj[s][1] = qcj_n[s][1]; qcj[s][2] = qcj_n[s][2];
        }
    }
    // row sums, log-sum-exp
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (g == 0) {
        sInv(i) = lsum > 0.f ? 1.f / lsum : 0.f;
        if (qlive) lse[pi * h + hh] = m + __logf(lsum);
    }
    __syncthreads();
    // O += H Tv
    load_flat_table(sT, tv, L, h, hh, l);
    __syncthreads();
#pragma unroll 4
    for (int t = 0; t < kStrip / 4; ++t)
        acc_o = U2_MFMA(sH[i * kStrip + 4 * t + g], sT[(4 * t + g) * kHd + i], acc_o);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int64_t pr = p0 + 4 * g + s;
        if (pr < n) out[(size_t)sort_idx[pr] * ly.ld_out + hh * kHd + i] = acc_o[s] * sInv(4 * g + s);
    }
}

#undef sCorr
#undef sInv

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

size_t u2mkd_sptr_tiles_workspace_bytes(int64_t n, int32_t h) { return (size_t)2 * n * h * kHd * sizeof(float); }

int u2mkd_sptr_attention_forward_tiles(const float *q, const float *k, const float *v, int64_t ld_qkv, float q_scale,
                                       const int32_t *sort_idx, const int32_t *wstart, const int32_t *wlen, const int32_t *qc,
                                       const float *radial, const float *tq, const float *tk, const float *tv, int32_t L,
                                       int32_t qgl, float split_a, int64_t n, int32_t h, int32_t hdim, float *out, int64_t ld_out,
                                       float *lse, void *workspace, size_t workspace_bytes, u2mkd_stream_t s) {
    if (n == 0 || h == 0) return 0;
    U2_REQUIRE(q && k && v && sort_idx && wstart && wlen && qc && tq && tk && tv && out && lse && workspace,
               "u2mkd_sptr_attention_forward_tiles: null pointer");
    U2_REQUIRE(hdim == kHd, "u2mkd_sptr_attention_forward_tiles: head dim %d != 16", hdim);
    U2_REQUIRE(h > 0 && h <= 65535 && L > 0 && L <= 50, "u2mkd_sptr_attention_forward_tiles: bad head count %d / table length %d", h, L);
    U2_REQUIRE(split_a > 0.f ? L >= 2 * qgl : L >= 2 * qgl - 1, "u2mkd_sptr_attention_forward_tiles: table length %d too short for grid length %d", L, qgl);
    U2_REQUIRE(split_a <= 0.f || radial, "u2mkd_sptr_attention_forward_tiles: spherical branch needs the radial coordinate");
    U2_REQUIRE(ld_qkv >= (int64_t)h * kHd && ld_out >= (int64_t)h * kHd && ld_qkv % 4 == 0 && ld_out % 4 == 0,
               "u2mkd_sptr_attention_forward_tiles: row strides %lld / %lld must be multiples of 4 floats and hold %d heads",
               (long long)ld_qkv, (long long)ld_out, h);
    U2_REQUIRE(workspace_bytes >= u2mkd_sptr_tiles_workspace_bytes(n, h), "u2mkd_sptr_attention_forward_tiles: workspace too small");
    U2_REQUIRE(n < (1LL << 31) - 64, "u2mkd_sptr_attention_forward_tiles: too many tokens");
    float *ks = reinterpret_cast<float *>(workspace), *vs = ks + (size_t)n * h * kHd;
    RelCtx rc{qgl, split_a};
    SptrLayout ly{ld_qkv, ld_out, 0, q_scale};
    const float *rad = split_a > 0.f ? radial : nullptr;
    hipLaunchKernelGGL(sptr_sorted_kv_kernel, dim3((unsigned)ceil_div(n * h * 4, 256)), dim3(256), 0, as_stream(s), k, v, ld_qkv,
                       sort_idx, n, h, ks, vs);
    hipLaunchKernelGGL(sptr_tile_fwd_kernel, dim3((unsigned)ceil_div(n, 16), h), dim3(64), 0, as_stream(s), q, ks, vs, sort_idx, wstart,
                       wlen, qc, rad, tq, tk, tv, L, rc, n, h, out, lse, ly);
    return check_launch("u2mkd_sptr_attention_forward_tiles");
}

}  // extern "C"
