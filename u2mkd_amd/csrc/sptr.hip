// SphereFormer / sptr variable-length window attention with contextual relative
// position tables, fused.  Replaces, for the path the reference actually runs
// (pe_type='contextual', rel_query=rel_key=rel_value=True):
//   third_party/SparseTransformer/src/sptr/precompute/precompute_cuda_kernel.cu:4-35
//   third_party/SparseTransformer/src/sptr/rpe/relative_pos_encoding_cuda_kernel.cu:42-149 (scores fwd/bwd)
//   third_party/SparseTransformer/src/sptr/rpe/relative_pos_encoding_cuda_kernel.cu:151-274 (values fwd/bwd)
//   third_party/SparseTransformer/src/sptr/attention/attention_cuda_kernel.cu:29-75   (q.k backward)
//   sptr/utils.py:80-95 (CSR softmax) and the index glue of sptr/modules.py:35-65.
//
// The reference materialises M = sum_w L_w^2 pairs (index_0, index_1, rel_idx [M,3],
// attn [M,h], softmax [M,h]) in HBM and walks them in five kernels.  Here nothing of
// size M exists: tokens are sorted by window, every (token, head) thread walks the keys
// of its window, derives the relative-position rows from per-token quantised
// coordinates, keeps the three tables of its head in LDS and runs an online softmax;
// backward recomputes the scores from the saved log-sum-exp.  Table gradients are
// accumulated in LDS and flushed once per workgroup.
#include "common.h"
#include "sptr_internal.h"

namespace u2mkd {

constexpr int kTabRow = 20;    // LDS floats per table row (16 + 4 pad against bank conflicts)
constexpr int kSptrThreads = 128;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// c10::div_floor_floating (torch.div(rounding_mode='floor') for floats)
__device__ __forceinline__ float div_floor(float a, float b) {
    if (b == 0.f) return a / b;
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if ((mod != 0.f) && ((b < 0.f) != (mod < 0.f))) div -= 1.f;
    float fl;
    if (div != 0.f) {
        fl = floorf(div);
        if (div - fl > 0.5f) fl += 1.f;
    } else {
        fl = copysignf(0.f, a / b);
    }
    return fl;
}

// torch_cluster grid_cluster over (x, y, z, batch): key = sum_d trunc((p_d - start_d)/size_d) * stride_d
__global__ void window_keys_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ batch, int64_t n,
                                   const float *__restrict__ lo, const float *__restrict__ hi,
                                   const int32_t *__restrict__ bmax, float sx, float sy, float sz,
                                   int64_t *__restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float size[3] = {sx, sy, sz};
    int64_t c = 0, k = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        c += (int64_t)((xyz[i * 3 + d] - lo[d]) / size[d]) * k;
        k *= (int64_t)((hi[d] - lo[d]) / size[d]) + 1;
    }
    // batch dimension: size 1, start = min(batch) (lo[3]) , end = max(batch)
    float bl = lo[3];
    c += (int64_t)(((float)batch[i] - bl) / 1.f) * k;
    (void)bmax;
    keys[i] = c;
}

// ---- both window plans of a SphereFormer block from one pass over the points (u2mkd_sptr_plan_prepare) -----------
// spherical_transformer.py:31-36 (cart2sphere) + the two grid_cluster calls (:206-213) issue, as torch operators, 16
// element-wise launches for the spherical coordinates and per branch a cat, two reductions and the key kernel: ~28
// launches per block, 8 blocks per KD step.  Here: kernel 1 writes the spherical coordinates and per-workgroup
// minima / maxima of (x, y, z, batch, theta, beta, r); kernel 2 merges those (<= 256 workgroups) and writes both key
// arrays.  The arithmetic is torch's, operation by operation (fp32, no contraction): theta = (atan2(y, x) + pi) * 180 *
// (1 / pi) -- a division by a Python scalar is a multiplication by its fp32 reciprocal in ATen --, beta =
// atan2(sqrt(x * x + y * y), z) * 180 * (1 / pi), r = sqrt(x * x + y * y + z * z).
constexpr int kPrepWg = 256;     // workgroups of kernel 1 at most (= partial rows kernel 2 merges)

__global__ void __launch_bounds__(256)
sptr_prep_sphere_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ batch, int64_t n,
                        float *__restrict__ sphere, float *__restrict__ partial /*[gridDim.x][14]*/) {
#pragma clang fp contract(off)
    const float pi = 3.14159265358979323846f, inv_pi = 1.0f / pi;
    float lo[7], hi[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) { lo[d] = INFINITY; hi[d] = -INFINITY; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
        const float x2 = x * x, y2 = y * y, z2 = z * z;
        float v[7];
        v[0] = x; v[1] = y; v[2] = z; v[3] = (float)batch[i];
        v[4] = ((atan2f(y, x) + pi) * 180.0f) * inv_pi;
        v[5] = (atan2f(sqrtf(x2 + y2), z) * 180.0f) * inv_pi;
        v[6] = sqrtf((x2 + y2) + z2);
        sphere[i * 3] = v[4]; sphere[i * 3 + 1] = v[5]; sphere[i * 3 + 2] = v[6];
#pragma unroll
        for (int d = 0; d < 7; ++d) { lo[d] = fminf(lo[d], v[d]); hi[d] = fmaxf(hi[d], v[d]); }
    }
    __shared__ float red[4][14];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 7; ++d) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off));
        }
        if (lane == 0) { red[wave][d] = lo[d]; red[wave][7 + d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 14) {
        const int d = threadIdx.x;
        float v = red[0][d];
        for (int w = 1; w < 4; ++w) v = d < 7 ? fminf(v, red[w][d]) : fmaxf(v, red[w][d]);
        partial[(size_t)blockIdx.x * 14 + d] = v;
    }
}

// bounds[16] = lo4 | hi4 of (x, y, z, batch), lo4 | hi4 of (theta, beta, r, batch); keys as window_keys_kernel
__global__ void __launch_bounds__(256)
sptr_prep_keys_kernel(const float *__restrict__ xyz, const float *__restrict__ sphere, const int32_t *__restrict__ batch,
                      int64_t n, const float *__restrict__ partial, int n_partial, float cx, float cy, float cz, float sx,
                      float sy, float sz, float *__restrict__ bounds, int64_t *__restrict__ keys_c,
                      int64_t *__restrict__ keys_s) {
    __shared__ float b[14];
    if (threadIdx.x < 14) {
        const int d = threadIdx.x;
        float v = partial[d];
        for (int w = 1; w < n_partial; ++w) v = d < 7 ? fminf(v, partial[(size_t)w * 14 + d]) : fmaxf(v, partial[(size_t)w * 14 + d]);
        b[d] = v;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 16) {
        const int t = threadIdx.x, d = t & 3, hi = (t >> 2) & 1, sph = t >> 3;
        bounds[t] = b[(hi ? 7 : 0) + (d == 3 ? 3 : (sph ? 4 + d : d))];
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float bf = ((float)batch[i] - b[3]) / 1.f;
    {
        const float size[3] = {cx, cy, cz};
        int64_t c = 0, k = 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            c += (int64_t)((xyz[i * 3 + d] - b[d]) / size[d]) * k;
            k *= (int64_t)((b[7 + d] - b[d]) / size[d]) + 1;
        }
        keys_c[i] = c + (int64_t)bf * k;
    }
    {
        const float size[3] = {sx, sy, sz};
        int64_t c = 0, k = 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            c += (int64_t)((sphere[i * 3 + d] - b[4 + d]) / size[d]) * k;
            k *= (int64_t)((b[11 + d] - b[4 + d]) / size[d]) + 1;
        }
        keys_s[i] = c + (int64_t)bf * k;
    }
}

// sorted keys -> (first sorted position, length) of the window of every sorted position
__global__ void window_ranges_kernel(const int64_t *__restrict__ keys, int64_t n, int32_t *__restrict__ wstart,
                                     int32_t *__restrict__ wlen) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int64_t key = keys[p];
    if (p > 0 && keys[p - 1] == key) return;       // not a window head
    int64_t e = p + 1;
    while (e < n && keys[e] == key) ++e;
    for (int64_t t = p; t < e; ++t) {
        wstart[t] = (int32_t)p;
        wlen[t] = (int32_t)(e - p);
    }
}

// floor(((xyz - min) % window) / quant) per sorted position (sptr/modules.py:40-43)
__global__ void quant_coords_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ sort_idx, int64_t n,
                                    const float *__restrict__ lo, float wx, float wy, float wz, float qx, float qy,
                                    float qz, int32_t *__restrict__ qc, float *__restrict__ radial) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int64_t t = sort_idx[p];
    const float w[3] = {wx, wy, wz}, qs[3] = {qx, qy, qz};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = xyz[t * 3 + d] - lo[d] + 0.0f;
        float m = fmodf(v, w[d]);
        if ((m != 0.f) && ((w[d] < 0.f) != (m < 0.f))) m += w[d];
        qc[p * 3 + d] = (int32_t)div_floor(m, qs[d]);
    }
    if (radial) radial[p] = xyz[t * 3 + 2];
}

// s_tab: [3 tables][L][3][kTabRow]
__device__ __forceinline__ void load_tables(float *s_tab, const float *tq, const float *tk, const float *tv, int L,
                                            int h, int hh) {
    const float *src[3] = {tq, tk, tv};
    const int rows = L * 3;
    for (int e = threadIdx.x; e < 3 * rows * kHd; e += blockDim.x) {
        int t = e / (rows * kHd);
        int rem = e - t * rows * kHd;
        int row = rem / kHd, d = rem - row * kHd;
        s_tab[(t * rows + row) * kTabRow + d] = src[t][((size_t)row * h + hh) * kHd + d];
    }
}

__device__ __forceinline__ void tab_sum(const float *tab, const int r[3], float out[kHd]) {
    const float4 *a = reinterpret_cast<const float4 *>(tab + (r[0] * 3 + 0) * kTabRow);
    const float4 *b = reinterpret_cast<const float4 *>(tab + (r[1] * 3 + 1) * kTabRow);
    const float4 *c = reinterpret_cast<const float4 *>(tab + (r[2] * 3 + 2) * kTabRow);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float4 x = a[v], y = b[v], z = c[v];
        out[4 * v + 0] = x.x + y.x + z.x;
        out[4 * v + 1] = x.y + y.y + z.y;
        out[4 * v + 2] = x.z + y.z + z.z;
        out[4 * v + 3] = x.w + y.w + z.w;
    }
}

__device__ __forceinline__ void load16(const float *p, float out[kHd]) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float4 x = reinterpret_cast<const float4 *>(p)[v];
        out[4 * v] = x.x; out[4 * v + 1] = x.y; out[4 * v + 2] = x.z; out[4 * v + 3] = x.w;
    }
}

// ---- forward: out[t,h,:] = softmax_j(s) . (v_j + Tv(rel)),  lse saved per (sorted pos, head)
__global__ void __launch_bounds__(kSptrThreads)
sptr_attn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                     const int32_t *__restrict__ sort_idx, const int32_t *__restrict__ wstart,
                     const int32_t *__restrict__ wlen, const int32_t *__restrict__ qc,
                     const float *__restrict__ radial, const float *__restrict__ tq, const float *__restrict__ tk,
                     const float *__restrict__ tv, int L, RelCtx rc, int64_t n, int h, float *__restrict__ out,
                     float *__restrict__ lse, SptrLayout ly) {
    extern __shared__ __attribute__((aligned(16))) float s_tab[];
    const int hh = blockIdx.y;
    load_tables(s_tab, tq, tk, tv, L, h, hh);
    __syncthreads();
    const float *Tq = s_tab, *Tk = s_tab + L * 3 * kTabRow, *Tv = s_tab + 2 * L * 3 * kTabRow;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t t = sort_idx[p];
    const size_t hc = ly.ld_qkv;
    float qi[kHd];
    load16(q + t * hc + hh * kHd, qi);
#pragma unroll
    for (int d = 0; d < kHd; ++d) qi[d] *= ly.q_scale;
    int qci[3] = {qc[p * 3], qc[p * 3 + 1], qc[p * 3 + 2]};
    float ri = radial ? radial[p] : 0.f;
    const int ws = wstart[p], wl = wlen[p];
    float m = -INFINITY, l = 0.f, acc[kHd];
#pragma unroll
    for (int d = 0; d < kHd; ++d) acc[d] = 0.f;
    for (int jj = 0; jj < wl; ++jj) {
        const int pj = ws + jj;
        const int64_t tj = sort_idx[pj];
        int qcj[3] = {qc[pj * 3], qc[pj * 3 + 1], qc[pj * 3 + 2]};
        float rj = radial ? radial[pj] : 0.f;
        int r[3];
        rel_rows(rc, qci, ri, qcj, rj, r);
        float kj[kHd], ts[kHd];
        load16(k + tj * hc + hh * kHd, kj);
        tab_sum(Tq, r, ts);
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < kHd; ++d) s += qi[d] * (kj[d] + ts[d]);
        tab_sum(Tk, r, ts);
#pragma unroll
        for (int d = 0; d < kHd; ++d) s += kj[d] * ts[d];
        float mn = fmaxf(m, s);
        float corr = __expf(m - mn), pe = __expf(s - mn);
        l = l * corr + pe;
        float vj[kHd];
        load16(v + tj * hc + hh * kHd, vj);
        tab_sum(Tv, r, ts);
#pragma unroll
        for (int d = 0; d < kHd; ++d) acc[d] = acc[d] * corr + pe * (vj[d] + ts[d]);
        m = mn;
    }
    float inv = 1.f / l;
    float *o = out + t * (size_t)ly.ld_out + hh * kHd;
#pragma unroll
    for (int v4 = 0; v4 < 4; ++v4)
        reinterpret_cast<float4 *>(o)[v4] =
            make_float4(acc[4 * v4] * inv, acc[4 * v4 + 1] * inv, acc[4 * v4 + 2] * inv, acc[4 * v4 + 3] * inv);
    lse[p * h + hh] = m + __logf(l);
}

// ---- forward, key-split form: S lanes per (token, head) ------------------------------------------------------------
// The spherical branch's windows grow with the stage (window_size_sphere x window_size_scale^stage): at the coarsest
// stage a few thousand tokens sit in windows of hundreds, so one thread per (token, head) walking its window serially
// is a handful of waves per CU in a long chain of dependent loads -- latency-bound, and the kernel lasts as long as
// the largest window (measured: 754 us forward for 7 552 tokens at stride 16 against 121 us for 57 216 at stride 2).
// Here S consecutive lanes share a token: lane s walks keys s, s + S, ... with its own online-softmax state, and the
// states are merged by a butterfly over the S lanes (a fixed tree: deterministic; every lane ends with the same
// value).  S x more waves hide the latency and the longest chain is S x shorter.
template <int S>
__global__ void __launch_bounds__(kSptrThreads)
sptr_attn_fwd_split_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                           const int32_t *__restrict__ sort_idx, const int32_t *__restrict__ wstart,
                           const int32_t *__restrict__ wlen, const int32_t *__restrict__ qc,
                           const float *__restrict__ radial, const float *__restrict__ tq, const float *__restrict__ tk,
                           const float *__restrict__ tv, int L, RelCtx rc, int64_t n, int h, float *__restrict__ out,
                           float *__restrict__ lse, SptrLayout ly) {
    extern __shared__ __attribute__((aligned(16))) float s_tab[];
    static_assert(S >= 2 && S <= 16 && (S & (S - 1)) == 0, "lanes per token: 2, 4, 8 or 16");
    constexpr int TPB = kSptrThreads / S;
    const int hh = blockIdx.y;
    load_tables(s_tab, tq, tk, tv, L, h, hh);
    __syncthreads();
    const float *Tq = s_tab, *Tk = s_tab + L * 3 * kTabRow, *Tv = s_tab + 2 * L * 3 * kTabRow;
    const int sub = threadIdx.x % S;
    const int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x / S;
    const bool live = p < n;                     // (no early return: the merge shuffles need every lane of the wave)
    const int64_t pp = live ? p : n - 1;
    const int64_t t = sort_idx[pp];
    const size_t hc = ly.ld_qkv;
    float qi[kHd];
    load16(q + t * hc + hh * kHd, qi);
#pragma unroll
    for (int d = 0; d < kHd; ++d) qi[d] *= ly.q_scale;
    int qci[3] = {qc[pp * 3], qc[pp * 3 + 1], qc[pp * 3 + 2]};
    float ri = radial ? radial[pp] : 0.f;
    const int ws = wstart[pp], wl = live ? wlen[pp] : 0;
    float m = -INFINITY, l = 0.f, acc[kHd];
#pragma unroll
    for (int d = 0; d < kHd; ++d) acc[d] = 0.f;
    for (int jj = sub; jj < wl; jj += S) {
        const int pj = ws + jj;
        const int64_t tj = sort_idx[pj];
        int qcj[3] = {qc[pj * 3], qc[pj * 3 + 1], qc[pj * 3 + 2]};
        float rj = radial ? radial[pj] : 0.f;
        int r[3];
        rel_rows(rc, qci, ri, qcj, rj, r);
        float kj[kHd], ts[kHd];
        load16(k + tj * hc + hh * kHd, kj);
        tab_sum(Tq, r, ts);
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < kHd; ++d) sc += qi[d] * (kj[d] + ts[d]);
        tab_sum(Tk, r, ts);
#pragma unroll
        for (int d = 0; d < kHd; ++d) sc += kj[d] * ts[d];
        float mn = fmaxf(m, sc);
        float corr = __expf(m - mn), pe = __expf(sc - mn);
        l = l * corr + pe;
        float vj[kHd];
        load16(v + tj * hc + hh * kHd, vj);
        tab_sum(Tv, r, ts);
#pragma unroll
        for (int d = 0; d < kHd; ++d) acc[d] = acc[d] * corr + pe * (vj[d] + ts[d]);
        m = mn;
    }
    // merge the S partial softmax states (a lane without keys carries m = -inf, l = 0)
#pragma unroll
    for (int off = S >> 1; off >= 1; off >>= 1) {
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        const float c1 = m == -INFINITY ? 0.f : __expf(m - mn), c2 = m2 == -INFINITY ? 0.f : __expf(m2 - mn);
        l = l * c1 + l2 * c2;
#pragma unroll
        for (int d = 0; d < kHd; ++d) acc[d] = acc[d] * c1 + __shfl_xor(acc[d], off) * c2;
        m = mn;
    }
    if (live && sub == 0) {
        const float inv = 1.f / l;
        float *o = out + t * (size_t)ly.ld_out + hh * kHd;
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4)
            reinterpret_cast<float4 *>(o)[v4] =
                make_float4(acc[4 * v4] * inv, acc[4 * v4 + 1] * inv, acc[4 * v4 + 2] * inv, acc[4 * v4 + 3] * inv);
        lse[p * h + hh] = m + __logf(l);
    }
}

// delta[p,h] = sum_d dout[t,h,d] * out[t,h,d]
__global__ void sptr_delta_kernel(const float *__restrict__ dout, const float *__restrict__ out,
                                  const int32_t *__restrict__ sort_idx, int64_t n, int h, float *__restrict__ delta,
                                  int64_t ld_out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * h) return;
    int64_t p = e / h;
    int hh = (int)(e - p * h);
    int64_t t = sort_idx[p];
    const float *a = dout + t * ld_out + hh * kHd, *b = out + t * ld_out + hh * kHd;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < kHd; ++d) s += a[d] * b[d];
    delta[e] = s;
}



// ---- backward, histogram form (the default) ----------------------------------------------------
// LDS float atomics run at a small fraction of the LDS rate on gfx950 (144 of them per (query, key)
// pair were 90 % of the backward above; tools/ab_sptr.py).  The table gradients are bilinear:
//     dTq[ax][r] = sum_i q_i  (x) Hq_i[ax][r],   Hq_i[ax][r] = sum_{j : rel_ax(i,j) = r} ds_ij
//     dTv[ax][r] = sum_i do_i (x) Hv_i[ax][r],   Hv_i = the same with p_ij
//     dTk[ax][r] = sum_j k_j  (x) Hk_j[ax][r],   Hk_j[ax][r] = sum_{i : rel_ax(i,j) = r} ds_ij
// i.e. the VECTOR is constant for the thread that owns the token and only a SCALAR goes to a row
// chosen per pair.  Each thread keeps its scalar histograms in a private strip of LDS (plain
// read-add-write, no atomics: for an affine axis the reachable rows of a token are the 24 values
// base..base+23, for the radial exponential split all 2*qgl), and after the window walk the wave
// contracts histograms with the token vectors on the MFMA unit: G[r][d] += sum_i P_i[r] * vec_i[d]
// (16x16x4 f32, m = row, k = token, n = channel), accumulating in registers over all the tokens
// the wave ever sees.  One slab of [3][L*3][16] per wave at the end, summed in a fixed order.
constexpr int kNB = 25;                       // quantised coordinates per affine axis the strips can hold (0..24: the
                                              // configs give window / quant = 24 up to float rounding)
constexpr int kHistAx = 2 * kNB;              // offset of axis 2's bins; axis 2 gets 2*kNB bins (radial split)
constexpr int kHistTab = 4 * kNB;             // bins of one table: 24 + 24 + 48
constexpr int kVecRow = kHd + 1;              // LDS stride of a staged token vector

__device__ __forceinline__ int hist_base(const RelCtx &c, int coord, bool as_query) {
    // smallest row a token with quantised coordinate `coord` can reach on an affine axis
    int b = as_query ? coord - (kNB - 1) + c.qgl - 1 : 0 - coord + c.qgl - 1;
    if (c.a > 0.f) b = min(max(b, 0), 2 * c.qgl - 1);
    return b;
}

// P_i[r] of one table/axis from a private histogram strip
__device__ __forceinline__ float hist_row(const float *hist, int ax, int base, int r, bool sphere) {
    if (ax == 2 && sphere) return r < 2 * kNB ? hist[kHistAx + r] : 0.f;
    int bin = r - base;
    return (bin >= 0 && bin < kNB) ? hist[ax * kNB + bin] : 0.f;
}

// wave-private contraction of the 64 tokens of this wave: acc[ax][rb] += P^T vec
template <int NT, int S>
__device__ __forceinline__ void hist_contract(const float *s_hist, int hs, const float *s_vec, const int *s_base,
                                              int wave, int lane, bool sphere, f32x4 (&acc)[NT][3][3]) {
    constexpr int TPB = kSptrThreads / S, TW = 64 / S;           // tokens per workgroup / per wave (S lanes per token)
    const int col = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tb = 0; tb < NT; ++tb) {
        for (int ks = 0; ks < TW / 4; ++ks) {
            const int i = TW * wave + 4 * ks + kq;               // token slot in the workgroup
            const float bvec = s_vec[(tb * TPB + i) * kVecRow + col];
            const float *hist = s_hist + (size_t)i * hs + tb * kHistTab;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const int base = s_base[i * 3 + ax];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    float a = hist_row(hist, ax, base, 16 * rb + col, sphere);
                    acc[tb][ax][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bvec, acc[tb][ax][rb], 0, 0, 0);
                }
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ void hist_store_slab(float *slab, const int (&table_of)[NT], int L, int lane,
                                                const f32x4 (&acc)[NT][3][3]) {
    const int col = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tb = 0; tb < NT; ++tb)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    int r = 16 * rb + 4 * kq + reg;
                    if (r < L) slab[((size_t)table_of[tb] * L * 3 + r * 3 + ax) * kHd + col] = acc[tb][ax][rb][reg];
                }
}

// Both backward kernels take S lanes per token like the forward (S = 1: one thread walks the window): lane s handles
// pairs s, s + S, ...; the token's histogram strips are shared by its S lanes (LDS float adds; the lanes of a token sit
// in ONE wave, whose instructions execute in program order, so the sums are reproducible), the per-lane partial row
// gradients are summed by a butterfly.  A workgroup then holds 128 / S tokens: S x fewer strips, more workgroups per
// CU, S x shorter chains.
// query role: dq_i, and the Tq / Tv table gradients
template <int S>
__device__ __forceinline__ void
sptr_bwd_query_body(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                      const float *__restrict__ dout, const float *__restrict__ lse, const float *__restrict__ delta,
                      const int32_t *__restrict__ sort_idx, const int32_t *__restrict__ wstart,
                      const int32_t *__restrict__ wlen, const int32_t *__restrict__ qc,
                      const float *__restrict__ radial, const float *__restrict__ tq, const float *__restrict__ tk,
                      const float *__restrict__ tv, int L, RelCtx rc, int64_t n, int h, float *__restrict__ dq,
                      float *__restrict__ slabs, SptrLayout ly) {
    extern __shared__ __attribute__((aligned(16))) float s_tab[];
    constexpr int NT = 2, HS = NT * kHistTab + 1, TPB = kSptrThreads / S, TW = 64 / S;
    const int hh = blockIdx.y;
    const int tabf = L * 3 * kTabRow;
    float *s_hist = s_tab + 3 * tabf;                              // [tokens][HS]
    float *s_vec = s_hist + TPB * HS;                              // [NT][tokens][kVecRow]
    int *s_base = reinterpret_cast<int *>(s_vec + NT * TPB * kVecRow);   // [tokens][3]
    load_tables(s_tab, tq, tk, tv, L, h, hh);
    __syncthreads();
    const float *Tq = s_tab, *Tk = s_tab + tabf, *Tv = s_tab + 2 * tabf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = tid % S, slot = tid / S;
    const bool sphere = rc.a > 0.f;
    const size_t hc = ly.ld_qkv;
    float *hist = s_hist + (size_t)slot * HS;
    f32x4 acc[NT][3][3];
#pragma unroll
    for (int tb = 0; tb < NT; ++tb)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[tb][ax][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t nblk = (n + TPB - 1) / TPB;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t p = blk * TPB + slot;
        if (S == 1) {
            for (int e = 0; e < HS; ++e) hist[e] = 0.f;
        } else {                                         // the wave clears the strips of its tokens
            for (int e = lane; e < TW * HS; e += 64) s_hist[(size_t)wave * TW * HS + e] = 0.f;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
        float qi[kHd], doi[kHd];
#pragma unroll
        for (int d = 0; d < kHd; ++d) { qi[d] = 0.f; doi[d] = 0.f; }
        int base[3] = {0, 0, 0};
        if (p < n) {
            const int64_t t = sort_idx[p];
            int qci[3] = {qc[p * 3], qc[p * 3 + 1], qc[p * 3 + 2]};
            float ri = radial ? radial[p] : 0.f;
            const int ws = wstart[p], wl = wlen[p];
            load16(q + t * hc + hh * kHd, qi);
#pragma unroll
            for (int d = 0; d < kHd; ++d) qi[d] *= ly.q_scale;
            load16(dout + t * (size_t)ly.ld_out + hh * kHd, doi);
            const float lse_i = lse[p * h + hh], del_i = delta[p * h + hh];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) base[ax] = hist_base(rc, qci[ax], true);
            float dqi[kHd];
#pragma unroll
            for (int d = 0; d < kHd; ++d) dqi[d] = 0.f;
            // the pair's index chain (sorted position -> token id, quantised coordinates) is loaded ONE PAIR AHEAD: with one or two
            // waves per SIMD nothing else hides the two dependent global latencies of a pair (index, then the token's rows)
            int64_t tj_n = 0;
            int qcj_n[3] = {0, 0, 0};
            float rj_n = 0.f;
            if (sub < wl) {
                const int pj = ws + sub;
                tj_n = sort_idx[pj];
                qcj_n[0] = qc[pj * 3]; qcj_n[1] = qc[pj * 3 + 1]; qcj_n[2] = qc[pj * 3 + 2];
                rj_n = radial ? radial[pj] : 0.f;
            }
            for (int jj = sub; jj < wl; jj += S) {
                const int64_t tj = tj_n;
                int qcj[3] = {qcj_n[0], qcj_n[1], qcj_n[2]};
                float rj = rj_n;
                float xj[kHd], ts[kHd], tks[kHd];
                float vj[kHd], tvs[kHd];
                load16(k + tj * hc + hh * kHd, xj);
                load16(v + tj * hc + hh * kHd, vj);
                if (jj + S < wl) {
                    const int pn = ws + jj + S;
                    tj_n = sort_idx[pn];
                    qcj_n[0] = qc[pn * 3]; qcj_n[1] = qc[pn * 3 + 1]; qcj_n[2] = qc[pn * 3 + 2];
                    rj_n = radial ? radial[pn] : 0.f;
                }
                int r[3];
                rel_rows(rc, qci, ri, qcj, rj, r);
                tab_sum(Tq, r, ts);
                tab_sum(Tk, r, tks);
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < kHd; ++d) s += qi[d] * (xj[d] + ts[d]) + xj[d] * tks[d];
                float pr = __expf(s - lse_i);
                tab_sum(Tv, r, tvs);
                float dp = 0.f;
#pragma unroll
                for (int d = 0; d < kHd; ++d) dp += doi[d] * (vj[d] + tvs[d]);
                float ds = pr * (dp - del_i);
#pragma unroll
                for (int d = 0; d < kHd; ++d) dqi[d] += ds * (xj[d] + ts[d]);
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const bool radial_ax = ax == 2 && sphere;
                    const int bin = radial_ax ? r[2] : r[ax] - base[ax];
                    if ((unsigned)bin < (unsigned)(radial_ax ? 2 * kNB : kNB)) {   // always true for qc_span <= kNB
                        const int off = radial_ax ? kHistAx + bin : ax * kNB + bin;
                        if (S == 1) {
                            hist[off] += ds;                    // Hq
                            hist[kHistTab + off] += pr;         // Hv
                        } else {
                            atomicAdd(&hist[off], ds);
                            atomicAdd(&hist[kHistTab + off], pr);
                        }
                    }
                }
            }
#pragma unroll
            for (int off = S >> 1; off >= 1; off >>= 1)
#pragma unroll
                for (int d = 0; d < kHd; ++d) dqi[d] += __shfl_xor(dqi[d], off);
            if (sub == 0) {
                float *o1 = dq + t * (size_t)ly.ld_grad + hh * kHd;      // d(unscaled q) = q_scale * d(q)
#pragma unroll
                for (int d = 0; d < kHd; ++d) o1[d] = dqi[d] * ly.q_scale;
            }
        }
        if (sub == 0) {
#pragma unroll
            for (int d = 0; d < kHd; ++d) {
                s_vec[(0 * TPB + slot) * kVecRow + d] = qi[d];
                s_vec[(1 * TPB + slot) * kVecRow + d] = doi[d];
            }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) s_base[slot * 3 + ax] = base[ax];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        hist_contract<NT, S>(s_hist, HS, s_vec, s_base, wave, lane, sphere, acc);
        __builtin_amdgcn_wave_barrier();
    }
    const int table_of[NT] = {0, 2};
    float *slab = slabs + (((size_t)blockIdx.x * 2 + wave) * h + hh) * 3 * L * 3 * kHd;
    hist_store_slab<NT>(slab, table_of, L, lane, acc);
}

// key role: dk_j, dv_j and the Tk table gradient
template <int S>
__device__ __forceinline__ void
sptr_bwd_key_body(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                    const float *__restrict__ dout, const float *__restrict__ lse, const float *__restrict__ delta,
                    const int32_t *__restrict__ sort_idx, const int32_t *__restrict__ wstart,
                    const int32_t *__restrict__ wlen, const int32_t *__restrict__ qc,
                    const float *__restrict__ radial, const float *__restrict__ tq, const float *__restrict__ tk,
                    const float *__restrict__ tv, int L, RelCtx rc, int64_t n, int h, float *__restrict__ dk,
                    float *__restrict__ dv, float *__restrict__ slabs, SptrLayout ly) {
    extern __shared__ __attribute__((aligned(16))) float s_tab[];
    constexpr int NT = 1, HS = NT * kHistTab + 1, TPB = kSptrThreads / S, TW = 64 / S;
    const int hh = blockIdx.y;
    const int tabf = L * 3 * kTabRow;
    float *s_hist = s_tab + 3 * tabf;
    float *s_vec = s_hist + TPB * HS;
    int *s_base = reinterpret_cast<int *>(s_vec + NT * TPB * kVecRow);
    load_tables(s_tab, tq, tk, tv, L, h, hh);
    __syncthreads();
    const float *Tq = s_tab, *Tk = s_tab + tabf, *Tv = s_tab + 2 * tabf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = tid % S, slot = tid / S;
    const bool sphere = rc.a > 0.f;
    const size_t hc = ly.ld_qkv;
    float *hist = s_hist + (size_t)slot * HS;
    f32x4 acc[NT][3][3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[0][ax][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t nblk = (n + TPB - 1) / TPB;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t p = blk * TPB + slot;
        if (S == 1) {
            for (int e = 0; e < HS; ++e) hist[e] = 0.f;
        } else {
            for (int e = lane; e < TW * HS; e += 64) s_hist[(size_t)wave * TW * HS + e] = 0.f;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
        float ki[kHd];
#pragma unroll
        for (int d = 0; d < kHd; ++d) ki[d] = 0.f;
        int base[3] = {0, 0, 0};
        if (p < n) {
            const int64_t t = sort_idx[p];
            int qci[3] = {qc[p * 3], qc[p * 3 + 1], qc[p * 3 + 2]};
            float ri = radial ? radial[p] : 0.f;
            const int ws = wstart[p], wl = wlen[p];
            float vi[kHd];
            load16(k + t * hc + hh * kHd, ki);
            load16(v + t * hc + hh * kHd, vi);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) base[ax] = hist_base(rc, qci[ax], false);
            float dki[kHd], dvi[kHd];
#pragma unroll
            for (int d = 0; d < kHd; ++d) { dki[d] = 0.f; dvi[d] = 0.f; }
            // (the query's index chain and its two scalars one pair ahead, as in the query role)
            int64_t tj_n = 0;
            int qcj_n[3] = {0, 0, 0};
            float rj_n = 0.f, lse_n = 0.f, del_n = 0.f;
            if (sub < wl) {
                const int pj = ws + sub;
                tj_n = sort_idx[pj];
                qcj_n[0] = qc[pj * 3]; qcj_n[1] = qc[pj * 3 + 1]; qcj_n[2] = qc[pj * 3 + 2];
                rj_n = radial ? radial[pj] : 0.f;
                lse_n = lse[pj * h + hh]; del_n = delta[pj * h + hh];
            }
            for (int jj = sub; jj < wl; jj += S) {
                const int64_t tj = tj_n;                      // the QUERY of this pair
                int qcj[3] = {qcj_n[0], qcj_n[1], qcj_n[2]};
                float rj = rj_n;
                const float lse_j = lse_n, del_j = del_n;
                float qj[kHd], doj[kHd], ts[kHd], tks[kHd], tvs[kHd];
                load16(q + tj * hc + hh * kHd, qj);
                load16(dout + tj * (size_t)ly.ld_out + hh * kHd, doj);
                if (jj + S < wl) {
                    const int pn = ws + jj + S;
                    tj_n = sort_idx[pn];
                    qcj_n[0] = qc[pn * 3]; qcj_n[1] = qc[pn * 3 + 1]; qcj_n[2] = qc[pn * 3 + 2];
                    rj_n = radial ? radial[pn] : 0.f;
                    lse_n = lse[pn * h + hh]; del_n = delta[pn * h + hh];
                }
                int r[3];
                rel_rows(rc, qcj, rj, qci, ri, r);
#pragma unroll
                for (int d = 0; d < kHd; ++d) qj[d] *= ly.q_scale;
                tab_sum(Tq, r, ts);
                tab_sum(Tk, r, tks);
                float s2 = 0.f;
#pragma unroll
                for (int d = 0; d < kHd; ++d) s2 += qj[d] * (ki[d] + ts[d]) + ki[d] * tks[d];
                float pr2 = __expf(s2 - lse_j);
                tab_sum(Tv, r, tvs);
                float dp2 = 0.f;
#pragma unroll
                for (int d = 0; d < kHd; ++d) dp2 += doj[d] * (vi[d] + tvs[d]);
                float ds2 = pr2 * (dp2 - del_j);
#pragma unroll
                for (int d = 0; d < kHd; ++d) {
                    dki[d] += ds2 * (qj[d] + tks[d]);
                    dvi[d] += pr2 * doj[d];
                }
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const bool radial_ax = ax == 2 && sphere;
                    const int bin = radial_ax ? r[2] : r[ax] - base[ax];
                    if ((unsigned)bin < (unsigned)(radial_ax ? 2 * kNB : kNB)) {
                        float *hk = &hist[(radial_ax ? kHistAx : ax * kNB) + bin];   // Hk
                        if (S == 1) *hk += ds2;
                        else atomicAdd(hk, ds2);
                    }
                }
            }
#pragma unroll
            for (int off = S >> 1; off >= 1; off >>= 1)
#pragma unroll
                for (int d = 0; d < kHd; ++d) {
                    dki[d] += __shfl_xor(dki[d], off);
                    dvi[d] += __shfl_xor(dvi[d], off);
                }
            if (sub == 0) {
                float *o2 = dk + t * (size_t)ly.ld_grad + hh * kHd, *o3 = dv + t * (size_t)ly.ld_grad + hh * kHd;
#pragma unroll
                for (int d = 0; d < kHd; ++d) { o2[d] = dki[d]; o3[d] = dvi[d]; }
            }
        }
        if (sub == 0) {
#pragma unroll
            for (int d = 0; d < kHd; ++d) s_vec[slot * kVecRow + d] = ki[d];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) s_base[slot * 3 + ax] = base[ax];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        hist_contract<NT, S>(s_hist, HS, s_vec, s_base, wave, lane, sphere, acc);
        __builtin_amdgcn_wave_barrier();
    }
    const int table_of[NT] = {1};
    float *slab = slabs + (((size_t)blockIdx.x * 2 + wave) * h + hh) * 3 * L * 3 * kHd;
    hist_store_slab<NT>(slab, table_of, L, lane, acc);
}

// Waves per SIMD the backward kernels are compiled for.  Everything in registers (256 + 88 accumulator registers) is one
// wave per SIMD; S = 16 -- the spherical branch at the coarsest stride: windows of hundreds of tokens, the longest launch of the
// student's backward -- gains 25 % from two waves (<= 256 registers, ~200 bytes of scratch per lane: 1.37-1.58 -> 1.04-1.07 ms
// in the step), the other splits do not (round 6, NOTES N10.7).  -DU2MKD_SPTR_BWD_WPE=n forces n everywhere (A/B builds).
#if defined(U2MKD_SPTR_BWD_WPE)
#define U2_SPTR_BWD_BOUNDS __launch_bounds__(kSptrThreads, U2MKD_SPTR_BWD_WPE)
#else
#define U2_SPTR_BWD_BOUNDS __launch_bounds__(kSptrThreads, (S == 16 ? 2 : 1))
#endif
#define U2_SPTR_BWD_IN                                                                                                   \
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v, const float *__restrict__ dout,  \
        const float *__restrict__ lse, const float *__restrict__ delta, const int32_t *__restrict__ sort_idx,             \
        const int32_t *__restrict__ wstart, const int32_t *__restrict__ wlen, const int32_t *__restrict__ qc,             \
        const float *__restrict__ radial, const float *__restrict__ tq, const float *__restrict__ tk,                     \
        const float *__restrict__ tv, int L, RelCtx rc, int64_t n, int h
#define U2_SPTR_BWD_PASS q, k, v, dout, lse, delta, sort_idx, wstart, wlen, qc, radial, tq, tk, tv, L, rc, n, h

template <int S>
__global__ void U2_SPTR_BWD_BOUNDS
sptr_bwd_query_kernel(U2_SPTR_BWD_IN, float *__restrict__ dq, float *__restrict__ slabs, SptrLayout ly) {
    sptr_bwd_query_body<S>(U2_SPTR_BWD_PASS, dq, slabs, ly);
}

template <int S>
__global__ void U2_SPTR_BWD_BOUNDS
sptr_bwd_key_kernel(U2_SPTR_BWD_IN, float *__restrict__ dk, float *__restrict__ dv, float *__restrict__ slabs, SptrLayout ly) {
    sptr_bwd_key_body<S>(U2_SPTR_BWD_PASS, dk, dv, slabs, ly);
}

// Both roles in ONE launch, grid (G, heads, 2): blockIdx.z = 0 the query role, 1 the key role.  The two are independent (they
// write different rows of the gradients and different tables of the same per-wave slab), and with S > 1 lanes per token
// (the spherical branch at the coarse strides: 128 workgroups x heads of 2 waves, ~40 KB of LDS each) one role alone leaves
// most of the chip idle: launched one behind the other they were the longest kernels of the student's backward chain.
template <int S>
__global__ void U2_SPTR_BWD_BOUNDS
sptr_bwd_both_kernel(U2_SPTR_BWD_IN, float *__restrict__ dq, float *__restrict__ dk, float *__restrict__ dv,
                     float *__restrict__ slabs, SptrLayout ly) {
    if (blockIdx.z == 0) sptr_bwd_query_body<S>(U2_SPTR_BWD_PASS, dq, slabs, ly);
    else sptr_bwd_key_body<S>(U2_SPTR_BWD_PASS, dk, dv, slabs, ly);
}
#undef U2_SPTR_BWD_IN
#undef U2_SPTR_BWD_PASS

// dt[tb][row][hh][d] = sum over the G slabs (ascending) -- deterministic
__global__ void sptr_table_reduce_kernel(const float *__restrict__ slabs, int G, int L, int h, float *__restrict__ dtq,
                                         float *__restrict__ dtk, float *__restrict__ dtv) {
    const int rows = L * 3;
    const int per = 3 * rows * kHd;
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    int hh = blockIdx.y;
    if (e >= per) return;
    // (ascending slab order, eight loads in flight: one dependent round trip per slab made this sum 100 us for 256 slabs)
    float acc = 0.f;
    int g = 0;
    for (; g + 8 <= G; g += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slabs[((size_t)(g + u) * h + hh) * per + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; g < G; ++g) acc += slabs[((size_t)g * h + hh) * per + e];
    int tb = e / (rows * kHd);
    int rem = e - tb * rows * kHd;
    int row = rem / kHd, d = rem - row * kHd;
    float *dst = tb == 0 ? dtq : (tb == 1 ? dtk : dtv);
    dst[((size_t)row * h + hh) * kHd + d] = acc;
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

int u2mkd_sptr_window_keys(const float *xyz, const int32_t *batch, int64_t n, const float *lo4, const float *hi4,
                           float sx, float sy, float sz, int64_t *keys, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(xyz && batch && lo4 && hi4 && keys, "u2mkd_sptr_window_keys: null pointer");
    U2_REQUIRE(sx > 0 && sy > 0 && sz > 0, "u2mkd_sptr_window_keys: window sizes must be positive");
    hipLaunchKernelGGL(window_keys_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), xyz, batch, n,
                       lo4, hi4, (const int32_t *)nullptr, sx, sy, sz, keys);
    return check_launch("u2mkd_sptr_window_keys");
}

size_t u2mkd_sptr_plan_prepare_workspace_bytes() { return (size_t)kPrepWg * 14 * sizeof(float); }

int u2mkd_sptr_plan_prepare(const float *xyz, const int32_t *batch, int64_t n, float cx, float cy, float cz, float sx,
                            float sy, float sz, float *sphere, float *bounds, int64_t *keys_cubic, int64_t *keys_sphere,
                            void *workspace, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(xyz && batch && sphere && bounds && keys_cubic && keys_sphere && workspace, "u2mkd_sptr_plan_prepare: null pointer");
    U2_REQUIRE(cx > 0 && cy > 0 && cz > 0 && sx > 0 && sy > 0 && sz > 0, "u2mkd_sptr_plan_prepare: window sizes must be positive");
    const int g1 = (int)(ceil_div(n, 256) < kPrepWg ? ceil_div(n, 256) : kPrepWg);
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(sptr_prep_sphere_kernel, dim3(g1), dim3(256), 0, as_stream(s), xyz, batch, n, sphere, partial);
    hipLaunchKernelGGL(sptr_prep_keys_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), xyz, sphere, batch, n,
                       partial, g1, cx, cy, cz, sx, sy, sz, bounds, keys_cubic, keys_sphere);
    return check_launch("u2mkd_sptr_plan_prepare");
}

int u2mkd_sptr_window_ranges(const int64_t *sorted_keys, int64_t n, int32_t *wstart, int32_t *wlen,
                             u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(sorted_keys && wstart && wlen, "u2mkd_sptr_window_ranges: null pointer");
    hipLaunchKernelGGL(window_ranges_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), sorted_keys,
                       n, wstart, wlen);
    return check_launch("u2mkd_sptr_window_ranges");
}

int u2mkd_sptr_quant_coords(const float *xyz, const int32_t *sort_idx, int64_t n, const float *lo, float wx, float wy,
                            float wz, float qx, float qy, float qz, int32_t *qc, float *radial, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(xyz && sort_idx && lo && qc, "u2mkd_sptr_quant_coords: null pointer");
    hipLaunchKernelGGL(quant_coords_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), xyz,
                       sort_idx, n, lo, wx, wy, wz, qx, qy, qz, qc, radial);
    return check_launch("u2mkd_sptr_quant_coords");
}

// Lanes per (token, head).  The window length is only known on the device; what the host knows is the token count
// and the branch: the cubic windows hold a few tokens at every stage, the spherical windows grow as the token count
// shrinks (coarser stages, wider cones).  U2MKD_SPTR_SPLIT = 1 / 2 / 4 / 8 / 16 overrides (A/B runs).
static int sptr_split(int64_t n, float split_a) {
    static const int forced = [] { const char *e = getenv("U2MKD_SPTR_SPLIT"); return e ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) return forced;
    if (split_a <= 0.f) return 1;
    return n < 12000 ? 16 : n < 24000 ? 8 : n < 48000 ? 4 : 2;
}

static int sptr_check(const char *who, int64_t n, int h, int hdim, int L, int qgl, float a) {
    U2_REQUIRE(hdim == kHd, "%s: head dim %d != 16 (the reference asserts hdim == 16)", who, hdim);
    U2_REQUIRE(h > 0 && h <= 65535, "%s: bad head count %d", who, h);
    U2_REQUIRE(L > 0 && L <= 50, "%s: table length %d not in 1..50 (the reference asserts L <= 50)", who, L);
    U2_REQUIRE(a > 0.f ? L >= 2 * qgl : L >= 2 * qgl - 1, "%s: table length %d too short for grid length %d", who, L,
               qgl);
    (void)n;
    return 0;
}

int u2mkd_sptr_attention_forward_strided(const float *q, const float *k, const float *v, int64_t ld_qkv, float q_scale,
                                         const int32_t *sort_idx, const int32_t *wstart, const int32_t *wlen,
                                         const int32_t *qc, const float *radial, const float *tq, const float *tk,
                                         const float *tv, int32_t L, int32_t qgl, float split_a, int64_t n, int32_t h,
                                         int32_t hdim, float *out, int64_t ld_out, float *lse, u2mkd_stream_t s) {
    if (n == 0 || h == 0) return 0;
    U2_REQUIRE(q && k && v && sort_idx && wstart && wlen && qc && tq && tk && tv && out && lse,
               "u2mkd_sptr_attention_forward: null pointer");
    if (int rc = sptr_check("u2mkd_sptr_attention_forward", n, h, hdim, L, qgl, split_a)) return rc;
    U2_REQUIRE(split_a <= 0.f || radial, "u2mkd_sptr_attention_forward: spherical branch needs the radial coordinate");
    U2_REQUIRE(ld_qkv >= (int64_t)h * kHd && ld_out >= (int64_t)h * kHd && ld_qkv % 4 == 0 && ld_out % 4 == 0,
               "u2mkd_sptr_attention_forward: row strides %lld / %lld must be multiples of 4 floats and hold %d heads",
               (long long)ld_qkv, (long long)ld_out, h);
    RelCtx rc{qgl, split_a};
    SptrLayout ly{ld_qkv, ld_out, 0, q_scale};
    size_t lds = (size_t)3 * L * 3 * kTabRow * sizeof(float);
    const float *rad = split_a > 0.f ? radial : nullptr;
    const int S = sptr_split(n, split_a);
#define U2_SPTR_FWD(SS)                                                                                                  \
    hipLaunchKernelGGL(sptr_attn_fwd_split_kernel<SS>, dim3((unsigned)ceil_div(n, kSptrThreads / SS), h),                \
                       dim3(kSptrThreads), lds, as_stream(s), q, k, v, sort_idx, wstart, wlen, qc, rad, tq, tk, tv, L,   \
                       rc, n, h, out, lse, ly)
    if (S == 16) U2_SPTR_FWD(16);
    else if (S == 8) U2_SPTR_FWD(8);
    else if (S == 4) U2_SPTR_FWD(4);
    else if (S == 2) U2_SPTR_FWD(2);
    else
        hipLaunchKernelGGL(sptr_attn_fwd_kernel, dim3((unsigned)ceil_div(n, kSptrThreads), h), dim3(kSptrThreads), lds,
                           as_stream(s), q, k, v, sort_idx, wstart, wlen, qc, rad, tq, tk, tv, L, rc, n, h, out, lse, ly);
#undef U2_SPTR_FWD
    return check_launch("u2mkd_sptr_attention_forward");
}

int u2mkd_sptr_attention_forward(const float *q, const float *k, const float *v, const int32_t *sort_idx,
                                 const int32_t *wstart, const int32_t *wlen, const int32_t *qc, const float *radial,
                                 const float *tq, const float *tk, const float *tv, int32_t L, int32_t qgl,
                                 float split_a, int64_t n, int32_t h, int32_t hdim, float *out, float *lse,
                                 u2mkd_stream_t s) {
    return u2mkd_sptr_attention_forward_strided(q, k, v, (int64_t)h * kHd, 1.f, sort_idx, wstart, wlen, qc, radial, tq, tk, tv,
                                                L, qgl, split_a, n, h, hdim, out, (int64_t)h * kHd, lse, s);
}

// persistent grid of the backward kernels: S lanes per token -> 128 / S tokens per workgroup pass
static int sptr_bwd_grid(int64_t n, int S) {
    int64_t nblk = ceil_div(n, kSptrThreads / S);
    const int64_t cap = S == 1 ? 512 : 128;            // (a slab per wave: the table reduce reads 2 x grid x heads of them)
    return (int)(nblk < cap ? nblk : cap);
}

static int sptr_bwd_grid_max(int64_t n) {
    int g = sptr_bwd_grid(n, 1);
    for (int S = 2; S <= 16; S *= 2) g = std::max(g, sptr_bwd_grid(n, S));
    return g;
}

size_t u2mkd_sptr_backward_workspace_bytes(int64_t n, int32_t h, int32_t L) {
    return (size_t)2 * sptr_bwd_grid_max(n) * h * 3 * L * 3 * kHd * sizeof(float);   // one slab per wave, any split
}

int u2mkd_sptr_attention_backward_strided(const float *q, const float *k, const float *v, int64_t ld_qkv, float q_scale,
                                          const float *out, const float *dout, int64_t ld_out,
                                  const float *lse, const int32_t *sort_idx, const int32_t *wstart,
                                  const int32_t *wlen, const int32_t *qc, const float *radial, const float *tq,
                                  const float *tk, const float *tv, int32_t L, int32_t qgl, float split_a,
                                  int32_t qc_span, int64_t n, int32_t h, int32_t hdim, float *delta /*[n,h] scratch*/,
                                  void *workspace, size_t workspace_bytes, float *dq, float *dk, float *dv,
                                  int64_t ld_grad, float *dtq, float *dtk, float *dtv, u2mkd_stream_t s) {
    if (n == 0 || h == 0) return 0;
    U2_REQUIRE(q && k && v && out && dout && lse && sort_idx && wstart && wlen && qc && tq && tk && tv && delta &&
                   workspace && dq && dk && dv,
               "u2mkd_sptr_attention_backward: null pointer");
    U2_REQUIRE((dtq && dtk && dtv) || (!dtq && !dtk && !dtv),
               "u2mkd_sptr_attention_backward: the three table gradients go together (all NULL: the caller sums the slabs itself, u2mkd_sptr_table_reduce)");
    if (int rc = sptr_check("u2mkd_sptr_attention_backward", n, h, hdim, L, qgl, split_a)) return rc;
    U2_REQUIRE(workspace_bytes >= u2mkd_sptr_backward_workspace_bytes(n, h, L),
               "u2mkd_sptr_attention_backward: workspace too small");
    U2_REQUIRE(ld_qkv >= (int64_t)h * kHd && ld_out >= (int64_t)h * kHd && ld_grad >= (int64_t)h * kHd && ld_qkv % 4 == 0 &&
                   ld_out % 4 == 0 && ld_grad % 4 == 0,
               "u2mkd_sptr_attention_backward: row strides must be multiples of 4 floats and hold %d heads", h);
    RelCtx rc{qgl, split_a};
    SptrLayout ly{ld_qkv, ld_out, ld_grad, q_scale};
    hipStream_t st = as_stream(s);
    hipLaunchKernelGGL(sptr_delta_kernel, dim3((unsigned)ceil_div(n * h, 256)), dim3(256), 0, st, dout, out, sort_idx,
                       n, h, delta, ld_out);
    const int S = sptr_split(n, split_a);
    const int G = sptr_bwd_grid(n, S);
    float *slabs = reinterpret_cast<float *>(workspace);
    const float *rad = split_a > 0.f ? radial : nullptr;
    const int per = 3 * L * 3 * kHd;
    // histogram form: every token reaches at most kNB rows per affine axis (quantised coordinates in
    // [0, qc_span), qc_span <= kNB) and 2*kNB rows on the radial axis
    U2_REQUIRE(qc_span > 0 && qc_span <= kNB && qgl < kNB && L <= 48,
               "u2mkd_sptr_attention_backward: qc_span=%d (window / quant size) must be in 1..%d, qgl=%d below %d, L=%d <= 48 "
               "(every U2MKD configuration has span = qgl = 24)", qc_span, kNB, qgl, kNB, L);
    const int tabf = L * 3 * kTabRow;
    const int tpb = kSptrThreads / S;
    size_t lds_q = ((size_t)3 * tabf + (size_t)tpb * (2 * kHistTab + 1) + 2 * tpb * kVecRow + tpb * 3) * sizeof(float);
    size_t lds_k = ((size_t)3 * tabf + (size_t)tpb * (kHistTab + 1) + tpb * kVecRow + tpb * 3) * sizeof(float);
#define U2_SPTR_BWD(SS)                                                                                                  \
    do {                                                                                                                 \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sptr_bwd_query_kernel<SS>),                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sptr_bwd_key_kernel<SS>),                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);                               \
        hipLaunchKernelGGL(sptr_bwd_query_kernel<SS>, dim3(G, h), dim3(kSptrThreads), lds_q, st, q, k, v, dout, lse,     \
                           delta, sort_idx, wstart, wlen, qc, rad, tq, tk, tv, L, rc, n, h, dq, slabs, ly);              \
        hipLaunchKernelGGL(sptr_bwd_key_kernel<SS>, dim3(G, h), dim3(kSptrThreads), lds_k, st, q, k, v, dout, lse,       \
                           delta, sort_idx, wstart, wlen, qc, rad, tq, tk, tv, L, rc, n, h, dk, dv, slabs, ly);          \
    } while (0)
    // S > 1: both roles in one launch (see sptr_bwd_both_kernel); S = 1 (128 tokens' strips per workgroup: 137 + 95 KB of LDS, the
    // grid already covers the chip) keeps the two launches.  U2MKD_SPTR_BWD_MERGE=0: two launches everywhere (A/B).
    static const bool merge = [] { const char *e = getenv("U2MKD_SPTR_BWD_MERGE"); return !(e && e[0] == '0'); }();
    const size_t lds_b = lds_q > lds_k ? lds_q : lds_k;
#define U2_SPTR_BOTH(SS)                                                                                                 \
    do {                                                                                                                 \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sptr_bwd_both_kernel<SS>),                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);                               \
        hipLaunchKernelGGL(sptr_bwd_both_kernel<SS>, dim3(G, h, 2), dim3(kSptrThreads), lds_b, st, q, k, v, dout, lse,   \
                           delta, sort_idx, wstart, wlen, qc, rad, tq, tk, tv, L, rc, n, h, dq, dk, dv, slabs, ly);      \
    } while (0)
    if (S > 1 && merge) {
        if (S == 16) U2_SPTR_BOTH(16);
        else if (S == 8) U2_SPTR_BOTH(8);
        else if (S == 4) U2_SPTR_BOTH(4);
        else U2_SPTR_BOTH(2);
    } else if (S == 16) U2_SPTR_BWD(16);
    else if (S == 8) U2_SPTR_BWD(8);
    else if (S == 4) U2_SPTR_BWD(4);
    else if (S == 2) U2_SPTR_BWD(2);
    else U2_SPTR_BWD(1);
#undef U2_SPTR_BWD
#undef U2_SPTR_BOTH
    if (dtq)
        hipLaunchKernelGGL(sptr_table_reduce_kernel, dim3((unsigned)ceil_div(per, 256), h), dim3(256), 0, st, slabs, 2 * G, L, h,
                           dtq, dtk, dtv);
    return check_launch("u2mkd_sptr_attention_backward");
}

/* The last launch of the backward on its own: sums the per-wave slabs u2mkd_sptr_attention_backward(_strided) left in
 * `workspace` when it was called with dtq = dtk = dtv = NULL (same n, h, L, split_a) -- on any stream ordered behind that call.
 * The tables are leaf parameters: nobody reads their gradients before the backward ends. */
int u2mkd_sptr_table_reduce(const void *workspace, int64_t n, int32_t h, int32_t L, float split_a, float *dtq, float *dtk,
                            float *dtv, u2mkd_stream_t s) {
    if (n == 0 || h == 0) return 0;
    U2_REQUIRE(workspace && dtq && dtk && dtv && L > 0 && L <= 50, "u2mkd_sptr_table_reduce: bad arguments");
    const int G = sptr_bwd_grid(n, sptr_split(n, split_a));
    const int per = 3 * L * 3 * kHd;
    hipLaunchKernelGGL(sptr_table_reduce_kernel, dim3((unsigned)ceil_div(per, 256), h), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const float *>(workspace), 2 * G, L, h, dtq, dtk, dtv);
    return check_launch("u2mkd_sptr_table_reduce");
}

int u2mkd_sptr_attention_backward(const float *q, const float *k, const float *v, const float *out, const float *dout,
                                  const float *lse, const int32_t *sort_idx, const int32_t *wstart,
                                  const int32_t *wlen, const int32_t *qc, const float *radial, const float *tq,
                                  const float *tk, const float *tv, int32_t L, int32_t qgl, float split_a,
                                  int32_t qc_span, int64_t n, int32_t h, int32_t hdim, float *delta /*[n,h] scratch*/,
                                  void *workspace, size_t workspace_bytes, float *dq, float *dk, float *dv, float *dtq,
                                  float *dtk, float *dtv, u2mkd_stream_t s) {
    const int64_t ld = (int64_t)h * kHd;
    return u2mkd_sptr_attention_backward_strided(q, k, v, ld, 1.f, out, dout, ld, lse, sort_idx, wstart, wlen, qc, radial, tq, tk,
                                                 tv, L, qgl, split_a, qc_span, n, h, hdim, delta, workspace, workspace_bytes,
                                                 dq, dk, dv, ld, dtq, dtk, dtv, s);
}

}  // extern "C"
