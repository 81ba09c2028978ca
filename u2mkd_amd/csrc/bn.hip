// BatchNorm over the rows of a [N, C] feature matrix, optionally fused with ReLU.
// Replaces spnn.BatchNorm + spnn.ReLU (= nn.BatchNorm1d / nn.ReLU applied to
// SparseTensor.feats through fapply; core/models/build_blocks.py:30-31,48-49,64-65,71,77;
// 49 instances per SPVCNN).
//
// HBM-bound row work.  Statistics are deterministic and cancellation-safe:
//   pass 1  each workgroup takes a slab of rows and computes, per channel, its local
//           mean and the centred second moment M2 around that mean (the slab is re-read
//           from L2), float4 columns x row lanes, LDS tree over the row lanes;
//   pass 2  64 lanes per channel merge the slabs with Chan's parallel update in a fixed
//           order (no atomics -> bitwise reproducible), write mean / invstd and update
//           the running statistics (unbiased variance, momentum) like nn.BatchNorm1d;
//   pass 3  y = (x - mean) * invstd * gamma + beta [, ReLU]  (float4 elementwise).
// Backward mirrors it: slab partial sums of dy' and dy'*xhat (dy' = dy masked by the fused
// ReLU, recomputed from x), ordered merge, then
//   dx = gamma * invstd * (dy' - mean(dy') - xhat * mean(dy' * xhat)).
#include "common.h"

#include <cstdlib>

namespace u2mkd {

constexpr int kBnThreads = 256;
constexpr int kBnSlabRows = 128;   // rows per workgroup in the partial passes (>= 600 workgroups at 80k rows)

// Rows are float or bf16 (common.h: bf16row, ld4, st4).  BF16 STORAGE (BASELINE.json configs[4]): under autocast the
// reference's BatchNorm1d takes and returns half rows with fp32 statistics; here bf16 rows, every sum, mean, invstd
// and gradient sum in fp32, one rounding per stored element.
// thread layout for a [rows, C4 float4] slab: j = float4 column, ry = row lane
struct BnLayout {
    int c4, rl;
};
__device__ __forceinline__ BnLayout bn_layout(int c) {
    BnLayout l;
    l.c4 = c >> 2;
    l.rl = kBnThreads / l.c4;
    if (l.rl < 1) l.rl = 1;
    return l;
}

// partial: [nslab][2][C] (mean_b, M2_b); rows of slab b = min(kBnSlabRows, n - b*kBnSlabRows)
// REG (c4 <= 32, i.e. C <= 128: at least 8 row lanes, at most 16 rows per thread): the thread's rows are loaded ONCE, all of
// them in flight together, and stay in registers for the second moment -- the same sums in the same order as the form
// below (bitwise the same partials), half the reads and no dependent load chain (the serial form: 12.5 us per launch on
// average in the KD step, 73 launches on the student's stream)
constexpr int kBnRegRows = 16;

template <typename T>
__global__ void __launch_bounds__(kBnThreads)
bn_stats_partial_reg_kernel(const T *__restrict__ x, int64_t n, int c, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];   // [rl][c4]
    const int c4 = c >> 2, rl = kBnThreads / c4;
    const int64_t r0 = (int64_t)blockIdx.x * kBnSlabRows;
    const int rows = (int)min((int64_t)kBnSlabRows, n - r0);
    const int j = threadIdx.x % c4, ry = threadIdx.x / c4;
    const bool live = ry < rl;
    float4 vals[kBnRegRows];
#pragma unroll
    for (int u = 0; u < kBnRegRows; ++u) {
        const int rr = ry + u * rl;
        vals[u] = (live && rr < rows) ? ld4(x, (r0 + rr) * c4 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < kBnRegRows; ++u) { s.x += vals[u].x; s.y += vals[u].y; s.z += vals[u].z; s.w += vals[u].w; }
    if (live) red[ry * c4 + j] = s;
    __syncthreads();
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int g = 0; g < rl; ++g) {
            float4 v = red[g * c4 + j];
            mean.x += v.x; mean.y += v.y; mean.z += v.z; mean.w += v.w;
        }
        float inv = 1.f / (float)rows;
        mean.x *= inv; mean.y *= inv; mean.z *= inv; mean.w *= inv;
    }
    __syncthreads();
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < kBnRegRows; ++u) {
        if (live && ry + u * rl < rows) {
            float dx = vals[u].x - mean.x, dy = vals[u].y - mean.y, dz = vals[u].z - mean.z, dw = vals[u].w - mean.w;
            m2.x += dx * dx; m2.y += dy * dy; m2.z += dz * dz; m2.w += dw * dw;
        }
    }
    if (live) red[ry * c4 + j] = m2;
    __syncthreads();
    if (live && ry == 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < rl; ++g) {
            float4 v = red[g * c4 + j];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        float *p = partial + (size_t)blockIdx.x * 2 * c;
        *reinterpret_cast<float4 *>(p + 4 * j) = mean;
        *reinterpret_cast<float4 *>(p + c + 4 * j) = t;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBnThreads)
bn_stats_partial_kernel(const T *__restrict__ x, int64_t n, int c, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];   // [rl][c4] (c4 <= 256)
    const int c4 = c >> 2;
    const int nloop = (c4 + kBnThreads - 1) / kBnThreads;           // > 1 only for C > 1024
    const int64_t r0 = (int64_t)blockIdx.x * kBnSlabRows;
    const int rows = (int)min((int64_t)kBnSlabRows, n - r0);
    for (int it = 0; it < nloop; ++it) {
        const int rl = c4 >= kBnThreads ? 1 : kBnThreads / c4;
        const int j = c4 >= kBnThreads ? it * kBnThreads + threadIdx.x : threadIdx.x % c4;
        const int ry = c4 >= kBnThreads ? 0 : threadIdx.x / c4;
        const bool live = j < c4 && ry < rl;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
            for (int rr0 = ry; rr0 < rows; rr0 += 8 * rl) {       // (eight rows' loads in flight per trip, summed in row order)
                float4 vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    vv[u] = rr0 + u * rl < rows ? ld4(x, (r0 + rr0 + u * rl) * c4 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) { s.x += vv[u].x; s.y += vv[u].y; s.z += vv[u].z; s.w += vv[u].w; }
            }
        if (live) red[ry * c4 + (j % c4)] = s;
        __syncthreads();
        float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            for (int g = 0; g < rl; ++g) {
                float4 v = red[g * c4 + (j % c4)];
                mean.x += v.x; mean.y += v.y; mean.z += v.z; mean.w += v.w;
            }
            float inv = 1.f / (float)rows;
            mean.x *= inv; mean.y *= inv; mean.z *= inv; mean.w *= inv;
        }
        __syncthreads();
        float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
            for (int rr0 = ry; rr0 < rows; rr0 += 8 * rl) {
                float4 vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    vv[u] = rr0 + u * rl < rows ? ld4(x, (r0 + rr0 + u * rl) * c4 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (rr0 + u * rl >= rows) break;
                    float dx = vv[u].x - mean.x, dy = vv[u].y - mean.y, dz = vv[u].z - mean.z, dw = vv[u].w - mean.w;
                    m2.x += dx * dx; m2.y += dy * dy; m2.z += dz * dz; m2.w += dw * dw;
                }
            }
        if (live) red[ry * c4 + (j % c4)] = m2;
        __syncthreads();
        if (live && ry == 0) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g = 0; g < rl; ++g) {
                float4 v = red[g * c4 + (j % c4)];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            float *p = partial + (size_t)blockIdx.x * 2 * c;
            *reinterpret_cast<float4 *>(p + 4 * j) = mean;
            *reinterpret_cast<float4 *>(p + c + 4 * j) = t;
        }
        __syncthreads();
    }
}

// 4 channels x 64 slab lanes per block: lane g merges slabs g, g+64, ... with Chan's update
// (the loads of a lane's slabs are independent of the update chain and issued four at a time --
// with 16 lanes the 40 dependent round trips to L2 made this tiny kernel cost 10 us), the 64
// partial (n, mean, M2) triples are then merged in lane order through LDS (fixed order).
constexpr int kBnFinLanes = 64, kBnFinCh = 4;

__global__ void __launch_bounds__(256)
bn_stats_finalize_kernel(const float *__restrict__ partial, int nslab, int64_t n, int c, float eps,
                         float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                         float *__restrict__ mean_out, float *__restrict__ invstd_out, float *__restrict__ m2_out,
                         int64_t *__restrict__ num_batches_tracked = nullptr, int slab_rows = kBnSlabRows) {
    // nn.BatchNorm's step counter (a separate one-element add_ launch per layer otherwise)
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    __shared__ float s_n[kBnFinLanes][kBnFinCh], s_m[kBnFinLanes][kBnFinCh], s_q[kBnFinLanes][kBnFinCh];
    const int cl = threadIdx.x & (kBnFinCh - 1), g = threadIdx.x / kBnFinCh;
    const int ch = blockIdx.x * kBnFinCh + cl;
    float na = 0.f, mean = 0.f, m2 = 0.f;
    if (ch < c) {
        for (int b0 = g; b0 < nslab; b0 += 4 * kBnFinLanes) {
            float mb[4], qb[4], nb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int b = b0 + u * kBnFinLanes;
                nb[u] = b < nslab ? (float)min((int64_t)slab_rows, n - (int64_t)b * slab_rows) : 0.f;
                mb[u] = b < nslab ? partial[(size_t)b * 2 * c + ch] : 0.f;
                qb[u] = b < nslab ? partial[(size_t)b * 2 * c + c + ch] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (nb[u] == 0.f) continue;
                float tot = na + nb[u];
                float delta = mb[u] - mean;
                mean += delta * (nb[u] / tot);
                m2 += qb[u] + delta * delta * (na * nb[u] / tot);
                na = tot;
            }
        }
    }
    s_n[g][cl] = na; s_m[g][cl] = mean; s_q[g][cl] = m2;
    __syncthreads();
    // fixed binary tree over the 64 lanes (a serial merge of 64 triples by one lane, two divisions
    // each, made this kernel slower than the statistics pass it finishes)
    for (int stride = kBnFinLanes / 2; stride >= 1; stride >>= 1) {
        if (g < stride) {
            float nb = s_n[g + stride][cl];
            if (nb != 0.f) {
                float na2 = s_n[g][cl], ma = s_m[g][cl];
                float tot = na2 + nb;
                float delta = s_m[g + stride][cl] - ma;
                s_m[g][cl] = ma + delta * (nb / tot);
                s_q[g][cl] = s_q[g][cl] + s_q[g + stride][cl] + delta * delta * (na2 * nb / tot);
                s_n[g][cl] = tot;
            }
        }
        __syncthreads();
    }
    if (g != 0 || ch >= c) return;
    mean = s_m[0][cl];
    m2 = s_q[0][cl];
    float var = m2 / (float)n;
    mean_out[ch] = mean;
    if (m2_out) {   // local statistics only (cross-rank merge follows): stats row = mean[c], M2[c], count
        m2_out[ch] = m2;
        if (ch == 0) m2_out[c] = (float)n;
        return;
    }
    invstd_out[ch] = 1.f / sqrtf(var + eps);
    if (running_mean) {
        float unbiased = n > 1 ? m2 / (float)(n - 1) : var;
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
    }
}

// y = (x - mean) * invstd * gamma + beta [, relu]; gamma / beta may be null (affine=False)
// res (may be null): the residual branch of a ResidualBlock, y = relu(bn(x) + res) in the same pass
template <typename T>
__global__ void bn_apply_kernel(const T *__restrict__ x, int64_t total4, int c4, const float *__restrict__ mean,
                                const float *__restrict__ invstd, const float *__restrict__ gamma,
                                const float *__restrict__ beta, int relu, T *__restrict__ y,
                                const T *__restrict__ res = nullptr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    int j = (int)(t % c4) * 4;
    float4 v = ld4(x, t);
    float4 m = *reinterpret_cast<const float4 *>(mean + j);
    float4 is = *reinterpret_cast<const float4 *>(invstd + j);
    float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + j) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 b = beta ? *reinterpret_cast<const float4 *>(beta + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o;
    o.x = (v.x - m.x) * is.x * g.x + b.x;
    o.y = (v.y - m.y) * is.y * g.y + b.y;
    o.z = (v.z - m.z) * is.z * g.z + b.z;
    o.w = (v.w - m.w) * is.w * g.w + b.w;
    if (res) {
        const float4 rv = ld4(res, t);
        o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4(y, t, o);
}

// eval mode: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta [, relu] in ONE launch (the separate
// invstd pass was a launch per layer of the frozen KD teacher); workgroup 0 also writes invstd for a backward pass
template <typename T>
__global__ void bn_apply_eval_kernel(const T *__restrict__ x, int64_t total4, int c4, const float *__restrict__ mean,
                                     const float *__restrict__ var, float eps, const float *__restrict__ gamma,
                                     const float *__restrict__ beta, int relu, float *__restrict__ invstd_out,
                                     T *__restrict__ y, const T *__restrict__ res = nullptr) {
    if (blockIdx.x == 0 && invstd_out)
        for (int ch = threadIdx.x; ch < 4 * c4; ch += blockDim.x) invstd_out[ch] = 1.f / sqrtf(var[ch] + eps);
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    int j = (int)(t % c4) * 4;
    float4 v = ld4(x, t);
    float4 m = *reinterpret_cast<const float4 *>(mean + j);
    float4 vr = *reinterpret_cast<const float4 *>(var + j);
    float4 is = make_float4(1.f / sqrtf(vr.x + eps), 1.f / sqrtf(vr.y + eps), 1.f / sqrtf(vr.z + eps), 1.f / sqrtf(vr.w + eps));
    float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + j) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 b = beta ? *reinterpret_cast<const float4 *>(beta + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o;
    o.x = (v.x - m.x) * is.x * g.x + b.x;
    o.y = (v.y - m.y) * is.y * g.y + b.y;
    o.z = (v.z - m.z) * is.z * g.z + b.z;
    o.w = (v.w - m.w) * is.w * g.w + b.w;
    if (res) {
        const float4 rv = ld4(res, t);
        o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4(y, t, o);
}

// partial: [nslab][2][C] (sum dy', sum dy' * xhat)
template <typename T>
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_partial_kernel(const T *__restrict__ dy, const T *__restrict__ x, int64_t n, int c,
                      const float *__restrict__ mean, const float *__restrict__ invstd,
                      const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                      float *__restrict__ partial, const T *__restrict__ res = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];   // [2][rl][c4]
    const int c4 = c >> 2;
    const int nloop = (c4 + kBnThreads - 1) / kBnThreads;
    const int64_t r0 = (int64_t)blockIdx.x * kBnSlabRows;
    const int rows = (int)min((int64_t)kBnSlabRows, n - r0);
    for (int it = 0; it < nloop; ++it) {
        const int rl = c4 >= kBnThreads ? 1 : kBnThreads / c4;
        const int j = c4 >= kBnThreads ? it * kBnThreads + threadIdx.x : threadIdx.x % c4;
        const int ry = c4 >= kBnThreads ? 0 : threadIdx.x / c4;
        const bool live = j < c4 && ry < rl;
        const int cw = c4 >= kBnThreads ? kBnThreads : c4;
        const int jl = c4 >= kBnThreads ? threadIdx.x : j;
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
        if (live) {
            float4 m = *reinterpret_cast<const float4 *>(mean + 4 * j);
            float4 is = *reinterpret_cast<const float4 *>(invstd + 4 * j);
            float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + 4 * j) : make_float4(1.f, 1.f, 1.f, 1.f);
            float4 b = beta ? *reinterpret_cast<const float4 *>(beta + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            // four rows' loads in flight per trip (the sums in the same row order: bitwise the same partials)
            for (int rr0 = ry; rr0 < rows; rr0 += 4 * rl) {
                float4 vv[4], dd[4], rr4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = rr0 + u * rl;
                    const bool ok = rr < rows;
                    vv[u] = ok ? ld4(x, (r0 + rr) * c4 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
                    dd[u] = ok ? ld4(dy, (r0 + rr) * c4 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
                    rr4[u] = (ok && relu && res) ? ld4(res, (r0 + rr) * c4 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (rr0 + u * rl >= rows) break;
                    const float4 v = vv[u], rv = rr4[u];
                    float4 d = dd[u];
                    float hx = (v.x - m.x) * is.x, hy = (v.y - m.y) * is.y, hz = (v.z - m.z) * is.z, hw = (v.w - m.w) * is.w;
                    if (relu) {
                        if (hx * g.x + b.x + rv.x <= 0.f) d.x = 0.f;
                        if (hy * g.y + b.y + rv.y <= 0.f) d.y = 0.f;
                        if (hz * g.z + b.z + rv.z <= 0.f) d.z = 0.f;
                        if (hw * g.w + b.w + rv.w <= 0.f) d.w = 0.f;
                    }
                    s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
                    s2.x += d.x * hx; s2.y += d.y * hy; s2.z += d.z * hz; s2.w += d.w * hw;
                }
            }
            red[ry * cw + jl] = s1;
            red[(rl + ry) * cw + jl] = s2;
        }
        __syncthreads();
        if (live && ry == 0) {
            float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
            for (int gq = 0; gq < rl; ++gq) {
                float4 a = red[gq * cw + jl], bq = red[(rl + gq) * cw + jl];
                t1.x += a.x; t1.y += a.y; t1.z += a.z; t1.w += a.w;
                t2.x += bq.x; t2.y += bq.y; t2.z += bq.z; t2.w += bq.w;
            }
            float *p = partial + (size_t)blockIdx.x * 2 * c;
            *reinterpret_cast<float4 *>(p + 4 * j) = t1;
            *reinterpret_cast<float4 *>(p + c + 4 * j) = t2;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(const float *__restrict__ partial, int nslab, int c, float *__restrict__ dbeta,
                       float *__restrict__ dgamma, float *__restrict__ keep = nullptr) {
    __shared__ float s_1[kBnFinLanes][kBnFinCh], s_2[kBnFinLanes][kBnFinCh];
    const int cl = threadIdx.x & (kBnFinCh - 1), g = threadIdx.x / kBnFinCh;
    const int ch = blockIdx.x * kBnFinCh + cl;
    float s1 = 0.f, s2 = 0.f;
    if (ch < c) {
        for (int b0 = g; b0 < nslab; b0 += 4 * kBnFinLanes) {
            float p1[4], p2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int b = b0 + u * kBnFinLanes;
                p1[u] = b < nslab ? partial[(size_t)b * 2 * c + ch] : 0.f;
                p2[u] = b < nslab ? partial[(size_t)b * 2 * c + c + ch] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s1 += p1[u]; s2 += p2[u]; }
        }
    }
    s_1[g][cl] = s1; s_2[g][cl] = s2;
    __syncthreads();
    if (g != 0 || ch >= c) return;
    for (int i = 1; i < kBnFinLanes; ++i) { s1 += s_1[i][cl]; s2 += s_2[i][cl]; }
    dbeta[ch] = s1;
    dgamma[ch] = s2;
    if (keep) { keep[ch] = s1; keep[c + ch] = s2; }      // a second copy: the first one is summed over the ranks in place
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T *__restrict__ dy, const T *__restrict__ x, int64_t total4, int c4,
                                    float inv_n_host, const float *__restrict__ total_n,
                                    const float *__restrict__ mean, const float *__restrict__ invstd,
                                    const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                    const float *__restrict__ dbeta, const float *__restrict__ dgamma,
                                    T *__restrict__ dx, const T *__restrict__ res = nullptr,
                                    T *__restrict__ dres = nullptr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const float inv_n = total_n ? 1.f / *total_n : inv_n_host;     // SyncBatchNorm: the count of all ranks
    int j = (int)(t % c4) * 4;
    float4 v = ld4(x, t);
    float4 d = ld4(dy, t);
    float4 m = *reinterpret_cast<const float4 *>(mean + j);
    float4 is = *reinterpret_cast<const float4 *>(invstd + j);
    float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + j) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 b = beta ? *reinterpret_cast<const float4 *>(beta + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 db = *reinterpret_cast<const float4 *>(dbeta + j);
    float4 dg = *reinterpret_cast<const float4 *>(dgamma + j);
    float hx = (v.x - m.x) * is.x, hy = (v.y - m.y) * is.y, hz = (v.z - m.z) * is.z, hw = (v.w - m.w) * is.w;
    if (relu) {
        const float4 rv = res ? ld4(res, t) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (hx * g.x + b.x + rv.x <= 0.f) d.x = 0.f;
        if (hy * g.y + b.y + rv.y <= 0.f) d.y = 0.f;
        if (hz * g.z + b.z + rv.z <= 0.f) d.z = 0.f;
        if (hw * g.w + b.w + rv.w <= 0.f) d.w = 0.f;
    }
    if (dres) st4(dres, t, d);          // gradient of the residual branch = the masked dy
    float4 o;
    o.x = g.x * is.x * (d.x - db.x * inv_n - hx * dg.x * inv_n);
    o.y = g.y * is.y * (d.y - db.y * inv_n - hy * dg.y * inv_n);
    o.z = g.z * is.z * (d.z - db.z * inv_n - hz * dg.z * inv_n);
    o.w = g.w * is.w * (d.w - db.w * inv_n - hw * dg.w * inv_n);
    st4(dx, t, o);
}

// eval-mode backward / plain affine: dx = dy' * gamma * invstd
template <typename T>
__global__ void bn_bwd_eval_kernel(const T *__restrict__ dy, const T *__restrict__ x, int64_t total4, int c4,
                                   const float *__restrict__ mean, const float *__restrict__ invstd,
                                   const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                   T *__restrict__ dx, const T *__restrict__ res = nullptr,
                                   T *__restrict__ dres = nullptr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    int j = (int)(t % c4) * 4;
    float4 v = ld4(x, t);
    float4 d = ld4(dy, t);
    float4 m = *reinterpret_cast<const float4 *>(mean + j);
    float4 is = *reinterpret_cast<const float4 *>(invstd + j);
    float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + j) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 b = beta ? *reinterpret_cast<const float4 *>(beta + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (relu) {
        const float4 rv = res ? ld4(res, t) : make_float4(0.f, 0.f, 0.f, 0.f);
        if ((v.x - m.x) * is.x * g.x + b.x + rv.x <= 0.f) d.x = 0.f;
        if ((v.y - m.y) * is.y * g.y + b.y + rv.y <= 0.f) d.y = 0.f;
        if ((v.z - m.z) * is.z * g.z + b.z + rv.z <= 0.f) d.z = 0.f;
        if ((v.w - m.w) * is.w * g.w + b.w + rv.w <= 0.f) d.w = 0.f;
    }
    if (dres) st4(dres, t, d);
    st4(dx, t, make_float4(d.x * g.x * is.x, d.y * g.y * is.y, d.z * g.z * is.z, d.w * g.w * is.w));
}

__global__ void bn_invstd_kernel(const float *__restrict__ var, int c, float eps, float *__restrict__ invstd) {
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < c) invstd[ch] = 1.f / sqrtf(var[ch] + eps);
}

// SyncBatchNorm: merge the per-rank (mean, M2, count) triples (rank order, Chan's update), write
// mean / invstd of the global batch and update the running statistics; stats rows are [2c+1].
__global__ void bn_sync_merge_kernel(const float *__restrict__ gathered, int world, int c, float eps, float momentum,
                                     float *__restrict__ running_mean, float *__restrict__ running_var,
                                     float *__restrict__ mean_out, float *__restrict__ invstd_out,
                                     float *__restrict__ total_out, int64_t *__restrict__ num_batches_tracked = nullptr) {
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (num_batches_tracked && ch == 0) *num_batches_tracked += 1;      // nn.BatchNorm's step counter (no launch of its own)
    if (ch >= c) return;
    const int row = 2 * c + 1;
    float na = 0.f, mean = 0.f, m2 = 0.f;
    for (int r = 0; r < world; ++r) {
        float nb = gathered[(size_t)r * row + 2 * c];
        if (nb == 0.f) continue;
        float mb = gathered[(size_t)r * row + ch], qb = gathered[(size_t)r * row + c + ch];
        float tot = na + nb;
        float delta = mb - mean;
        mean += delta * (nb / tot);
        m2 += qb + delta * delta * (na * nb / tot);
        na = tot;
    }
    float var = na > 0.f ? m2 / na : 0.f;
    mean_out[ch] = mean;
    invstd_out[ch] = 1.f / sqrtf(var + eps);
    if (ch == 0) *total_out = na;
    if (running_mean) {
        float unbiased = na > 1.f ? m2 / (na - 1.f) : var;
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
    }
}

static size_t bn_lds_bytes(int c, int arrays) {
    int c4 = c / 4;
    int rl = c4 >= kBnThreads ? 1 : kBnThreads / c4;
    int cw = c4 >= kBnThreads ? kBnThreads : c4;
    return (size_t)arrays * rl * cw * sizeof(float4);
}

// ---- launch sequences, shared by the fp32-row and the bf16-row entry points ------------------------------------
// U2MKD_BN_STATS_SERIAL=1: the statistics pass in its two-read form for every width (A/B; the partials are bitwise the same)
static bool bn_stats_serial() {
    static const bool on = [] { const char *e = getenv("U2MKD_BN_STATS_SERIAL"); return e && e[0] == '1'; }();
    return on;
}

template <typename T>
static void bn_launch_stats_partial(hipStream_t st, int nslab, const T *x, int64_t n, int c, float *partial) {
    const int c4 = c / 4;
    if (c4 <= 32 && kBnSlabRows / (kBnThreads / c4) <= kBnRegRows && !bn_stats_serial())
        hipLaunchKernelGGL(bn_stats_partial_reg_kernel<T>, dim3(nslab), dim3(kBnThreads), bn_lds_bytes(c, 1), st, x, n, c, partial);
    else
        hipLaunchKernelGGL(bn_stats_partial_kernel<T>, dim3(nslab), dim3(kBnThreads), bn_lds_bytes(c, 1), st, x, n, c, partial);
}

template <typename T>
static int bn_train_forward_impl(const T *x, const T *res, int64_t n, int32_t c, const float *gamma, const float *beta,
                                 float eps, float momentum, float *running_mean, float *running_var,
                                 int64_t *num_batches_tracked, int32_t relu, float *partial, float *mean, float *invstd, T *y,
                                 u2mkd_stream_t s) {
    U2_REQUIRE(c > 0 && c % 4 == 0 && c <= 1024, "u2mkd_bn_train_forward: c=%d must be a multiple of 4 in 4..1024", c);
    U2_REQUIRE(n > 0, "u2mkd_bn_train_forward: empty batch (n=%lld)", (long long)n);
    U2_REQUIRE(x && partial && mean && invstd && y, "u2mkd_bn_train_forward: null pointer");
    hipStream_t st = as_stream(s);
    int nslab = (int)u2mkd_bn_num_slabs(n);
#ifndef U2MKD_EXP_SKIP_BN_STATS      // (tools/build_variant.sh: an UPPER BOUND on what statistics taken in the producer's store could buy)
    bn_launch_stats_partial<T>(st, nslab, x, n, c, partial);
#endif
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((unsigned)ceil_div(c, kBnFinCh)), dim3(256), 0, st, partial, nslab, n, c,
                       eps, momentum, running_mean, running_var, mean, invstd, (float *)nullptr, num_batches_tracked);
    int64_t total4 = n * (c / 4);
    hipLaunchKernelGGL(bn_apply_kernel<T>, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, x, total4, c / 4, mean,
                       invstd, gamma, beta, relu, y, res);
    return check_launch("u2mkd_bn_train_forward");
}

// train-mode forward from slab partials somebody else computed (the gather-sum of the producing convolution:
// u2mkd_pairs_gather_sum_stats, slabs of `slab_rows` rows): merge + apply
static int bn_train_forward_from_partial_impl(const float *x, const float *res, int64_t n, int32_t c, const float *gamma,
                                              const float *beta, float eps, float momentum, float *running_mean,
                                              float *running_var, int64_t *num_batches_tracked, int32_t relu,
                                              const float *partial, int32_t slab_rows, float *mean, float *invstd, float *y,
                                              u2mkd_stream_t s) {
    U2_REQUIRE(c > 0 && c % 4 == 0 && c <= 1024, "u2mkd_bn_train_forward_from_partial: c=%d must be a multiple of 4 in 4..1024", c);
    U2_REQUIRE(n > 0 && slab_rows > 0, "u2mkd_bn_train_forward_from_partial: n=%lld rows in slabs of %d", (long long)n, slab_rows);
    U2_REQUIRE(x && partial && mean && invstd && y, "u2mkd_bn_train_forward_from_partial: null pointer");
    U2_REQUIRE(res == nullptr || relu, "u2mkd_bn_train_forward_from_partial: a residual input is only fused with the ReLU form");
    hipStream_t st = as_stream(s);
    const int64_t nslab = ceil_div(n, (int64_t)slab_rows);
    U2_REQUIRE(nslab < ((int64_t)1 << 31), "u2mkd_bn_train_forward_from_partial: too many slabs");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((unsigned)ceil_div(c, kBnFinCh)), dim3(256), 0, st, partial, (int)nslab, n, c,
                       eps, momentum, running_mean, running_var, mean, invstd, (float *)nullptr, num_batches_tracked, (int)slab_rows);
    int64_t total4 = n * (c / 4);
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, x, total4, c / 4, mean,
                       invstd, gamma, beta, relu, y, res);
    return check_launch("u2mkd_bn_train_forward_from_partial");
}

template <typename T>
static int bn_eval_forward_impl(const T *x, const T *res, int64_t n, int32_t c, const float *gamma, const float *beta,
                                float eps, const float *running_mean, const float *running_var, int32_t relu, float *invstd,
                                T *y, u2mkd_stream_t s) {
    U2_REQUIRE(c > 0 && c % 4 == 0, "u2mkd_bn_eval_forward: c=%d must be a positive multiple of 4", c);
    if (n == 0) return 0;
    U2_REQUIRE(x && running_mean && running_var && invstd && y, "u2mkd_bn_eval_forward: null pointer");
    hipStream_t st = as_stream(s);
    int64_t total4 = n * (c / 4);
    hipLaunchKernelGGL(bn_apply_eval_kernel<T>, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, x, total4, c / 4,
                       running_mean, running_var, eps, gamma, beta, relu, invstd, y, res);
    return check_launch("u2mkd_bn_eval_forward");
}

template <typename T>
static int bn_backward_impl(const T *dy, const T *x, const T *res, int64_t n, int32_t c, const float *mean,
                            const float *invstd, const float *gamma, const float *beta, int32_t relu, int32_t training,
                            float *partial, float *dgamma, float *dbeta, T *dx, T *dres, u2mkd_stream_t s) {
    U2_REQUIRE(c > 0 && c % 4 == 0 && c <= 1024, "u2mkd_bn_backward: c=%d must be a multiple of 4 in 4..1024", c);
    U2_REQUIRE((res == nullptr) == (dres == nullptr), "u2mkd_bn_backward_res: res and dres go together");
    U2_REQUIRE(res == nullptr || relu, "u2mkd_bn_backward_res: a residual input is only fused with the ReLU form");
    if (n == 0) return 0;
    U2_REQUIRE(dy && x && mean && invstd && partial && dgamma && dbeta && dx, "u2mkd_bn_backward: null pointer");
    hipStream_t st = as_stream(s);
    int nslab = (int)u2mkd_bn_num_slabs(n);
    hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, dim3(nslab), dim3(kBnThreads), bn_lds_bytes(c, 2), st, dy, x, n, c, mean,
                       invstd, gamma, beta, relu, partial, res);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(c, kBnFinCh)), dim3(256), 0, st, partial, nslab, c,
                       dbeta, dgamma);
    int64_t total4 = n * (c / 4);
    if (training)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, dy, x, total4,
                           c / 4, 1.f / (float)n, (const float *)nullptr, mean, invstd, gamma, beta, relu, dbeta, dgamma,
                           dx, res, dres);
    else
        hipLaunchKernelGGL(bn_bwd_eval_kernel<T>, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, dy, x, total4,
                           c / 4, mean, invstd, gamma, beta, relu, dx, res, dres);
    return check_launch("u2mkd_bn_backward");
}

template <typename T>
static int bn_local_stats_impl(const T *x, int64_t n, int32_t c, float *partial, float *stats, u2mkd_stream_t s) {
    U2_REQUIRE(c > 0 && c % 4 == 0 && c <= 1024, "u2mkd_bn_local_stats: c=%d must be a multiple of 4 in 4..1024", c);
    U2_REQUIRE(stats && (n == 0 || (x && partial)), "u2mkd_bn_local_stats: null pointer");
    hipStream_t st = as_stream(s);
    if (n == 0) {
        (void)hipMemsetAsync(stats, 0, (size_t)(2 * c + 1) * sizeof(float), st);
        return check_launch("u2mkd_bn_local_stats");
    }
    int nslab = (int)u2mkd_bn_num_slabs(n);
    bn_launch_stats_partial<T>(st, nslab, x, n, c, partial);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((unsigned)ceil_div(c, kBnFinCh)), dim3(256), 0, st, partial, nslab,
                       n, c, 0.f, 0.f, (float *)nullptr, (float *)nullptr, stats, (float *)nullptr, stats + c);
    return check_launch("u2mkd_bn_local_stats");
}

template <typename T>
static int bn_apply_impl(const T *x, int64_t n, int32_t c, const float *mean, const float *invstd, const float *gamma,
                         const float *beta, int32_t relu, T *y, u2mkd_stream_t s, const T *res = nullptr) {
    U2_REQUIRE(c > 0 && c % 4 == 0, "u2mkd_bn_apply: c=%d must be a positive multiple of 4", c);
    if (n == 0) return 0;
    U2_REQUIRE(x && mean && invstd && y, "u2mkd_bn_apply: null pointer");
    int64_t total4 = n * (c / 4);
    hipLaunchKernelGGL(bn_apply_kernel<T>, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, as_stream(s), x, total4,
                       c / 4, mean, invstd, gamma, beta, relu, y, res);
    return check_launch("u2mkd_bn_apply");
}

template <typename T>
static int bn_backward_local_impl(const T *dy, const T *x, int64_t n, int32_t c, const float *mean, const float *invstd,
                                  const float *gamma, const float *beta, int32_t relu, float *partial, float *sums,
                                  u2mkd_stream_t s, const T *res = nullptr, float *keep = nullptr) {
    U2_REQUIRE(c > 0 && c % 4 == 0 && c <= 1024, "u2mkd_bn_backward_local: c=%d must be a multiple of 4 in 4..1024", c);
    U2_REQUIRE(sums, "u2mkd_bn_backward_local: null pointer");
    hipStream_t st = as_stream(s);
    if (n == 0) {
        (void)hipMemsetAsync(sums, 0, (size_t)2 * c * sizeof(float), st);
        if (keep) (void)hipMemsetAsync(keep, 0, (size_t)2 * c * sizeof(float), st);
        return check_launch("u2mkd_bn_backward_local");
    }
    U2_REQUIRE(dy && x && mean && invstd && partial, "u2mkd_bn_backward_local: null pointer");
    int nslab = (int)u2mkd_bn_num_slabs(n);
    hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, dim3(nslab), dim3(kBnThreads), bn_lds_bytes(c, 2), st, dy, x, n, c, mean,
                       invstd, gamma, beta, relu, partial, res);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(c, kBnFinCh)), dim3(256), 0, st, partial, nslab, c,
                       sums, sums + c, keep);
    return check_launch("u2mkd_bn_backward_local");
}

template <typename T>
static int bn_backward_apply_impl(const T *dy, const T *x, int64_t n, int32_t c, const float *total_n, const float *mean,
                                  const float *invstd, const float *gamma, const float *beta, int32_t relu,
                                  const float *sums, T *dx, u2mkd_stream_t s, const T *res = nullptr, T *dres = nullptr) {
    U2_REQUIRE(c > 0 && c % 4 == 0, "u2mkd_bn_backward_apply: c=%d must be a positive multiple of 4", c);
    if (n == 0) return 0;
    U2_REQUIRE(dy && x && total_n && mean && invstd && sums && dx, "u2mkd_bn_backward_apply: null pointer");
    int64_t total4 = n * (c / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, as_stream(s), dy, x,
                       total4, c / 4, 0.f, total_n, mean, invstd, gamma, beta, relu, sums, sums + c, dx, res, dres);
    return check_launch("u2mkd_bn_backward_apply");
}

}  // namespace u2mkd

using namespace u2mkd;

#define BF(p) reinterpret_cast<const bf16row *>(p)
#define BFW(p) reinterpret_cast<bf16row *>(p)

extern "C" {

int64_t u2mkd_bn_num_slabs(int64_t n) { return n > 0 ? (n + kBnSlabRows - 1) / kBnSlabRows : 0; }

int u2mkd_bn_train_forward_res(const float *x, const float *res, int64_t n, int32_t c, const float *gamma, const float *beta,
                               float eps, float momentum, float *running_mean, float *running_var,
                               int64_t *num_batches_tracked, int32_t relu, float *partial, float *mean, float *invstd, float *y,
                               u2mkd_stream_t s) {
    return bn_train_forward_impl<float>(x, res, n, c, gamma, beta, eps, momentum, running_mean, running_var,
                                        num_batches_tracked, relu, partial, mean, invstd, y, s);
}

int u2mkd_bn_train_forward_from_partial(const float *x, const float *res, int64_t n, int32_t c, const float *gamma,
                                        const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                        int64_t *num_batches_tracked, int32_t relu, const float *partial, int32_t slab_rows,
                                        float *mean, float *invstd, float *y, u2mkd_stream_t s) {
    return bn_train_forward_from_partial_impl(x, res, n, c, gamma, beta, eps, momentum, running_mean, running_var,
                                              num_batches_tracked, relu, partial, slab_rows, mean, invstd, y, s);
}

int u2mkd_bn_train_forward_counted(const float *x, int64_t n, int32_t c, const float *gamma, const float *beta, float eps,
                                   float momentum, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                                   int32_t relu, float *partial /*[slabs,2,c]*/, float *mean /*[c]*/,
                                   float *invstd /*[c]*/, float *y, u2mkd_stream_t s) {
    return u2mkd_bn_train_forward_res(x, nullptr, n, c, gamma, beta, eps, momentum, running_mean, running_var,
                                      num_batches_tracked, relu, partial, mean, invstd, y, s);
}

int u2mkd_bn_train_forward(const float *x, int64_t n, int32_t c, const float *gamma, const float *beta, float eps,
                           float momentum, float *running_mean, float *running_var, int32_t relu,
                           float *partial /*[slabs,2,c]*/, float *mean /*[c]*/, float *invstd /*[c]*/, float *y,
                           u2mkd_stream_t s) {
    return u2mkd_bn_train_forward_counted(x, n, c, gamma, beta, eps, momentum, running_mean, running_var, nullptr, relu,
                                          partial, mean, invstd, y, s);
}

int u2mkd_bn_eval_forward_res(const float *x, const float *res, int64_t n, int32_t c, const float *gamma, const float *beta,
                              float eps, const float *running_mean, const float *running_var, int32_t relu,
                              float *invstd /*[c]*/, float *y, u2mkd_stream_t s) {
    return bn_eval_forward_impl<float>(x, res, n, c, gamma, beta, eps, running_mean, running_var, relu, invstd, y, s);
}

int u2mkd_bn_eval_forward(const float *x, int64_t n, int32_t c, const float *gamma, const float *beta, float eps,
                          const float *running_mean, const float *running_var, int32_t relu, float *invstd /*[c]*/,
                          float *y, u2mkd_stream_t s) {
    return u2mkd_bn_eval_forward_res(x, nullptr, n, c, gamma, beta, eps, running_mean, running_var, relu, invstd, y, s);
}

int u2mkd_bn_backward_res(const float *dy, const float *x, const float *res, int64_t n, int32_t c, const float *mean,
                          const float *invstd, const float *gamma, const float *beta, int32_t relu, int32_t training,
                          float *partial, float *dgamma, float *dbeta, float *dx, float *dres, u2mkd_stream_t s) {
    return bn_backward_impl<float>(dy, x, res, n, c, mean, invstd, gamma, beta, relu, training, partial, dgamma, dbeta, dx,
                                   dres, s);
}

int u2mkd_bn_backward(const float *dy, const float *x, int64_t n, int32_t c, const float *mean, const float *invstd,
                      const float *gamma, const float *beta, int32_t relu, int32_t training, float *partial,
                      float *dgamma /*[c]*/, float *dbeta /*[c]*/, float *dx, u2mkd_stream_t s) {
    return u2mkd_bn_backward_res(dy, x, nullptr, n, c, mean, invstd, gamma, beta, relu, training, partial, dgamma, dbeta, dx,
                                 nullptr, s);
}

/* ---- SyncBatchNorm pieces: local statistics | (all_gather by the caller) | merge | apply, and
 * local sums | (all_reduce by the caller) | apply in the backward ---- */
int u2mkd_bn_local_stats(const float *x, int64_t n, int32_t c, float *partial, float *stats /*[2c+1]*/,
                         u2mkd_stream_t s) {
    return bn_local_stats_impl<float>(x, n, c, partial, stats, s);
}

int u2mkd_bn_merge_stats(const float *gathered /*[world,2c+1]*/, int32_t world, int32_t c, float eps, float momentum,
                         float *running_mean, float *running_var, float *mean, float *invstd, float *total,
                         u2mkd_stream_t s) {
    U2_REQUIRE(gathered && mean && invstd && total && world > 0 && c > 0, "u2mkd_bn_merge_stats: bad arguments");
    hipLaunchKernelGGL(bn_sync_merge_kernel, dim3((unsigned)ceil_div(c, 64)), dim3(64), 0, as_stream(s), gathered, world, c,
                       eps, momentum, running_mean, running_var, mean, invstd, total);
    return check_launch("u2mkd_bn_merge_stats");
}

/* the same merge, and num_batches_tracked += 1 (may be NULL) in the same launch */
int u2mkd_bn_merge_stats_counted(const float *gathered, int32_t world, int32_t c, float eps, float momentum,
                                 float *running_mean, float *running_var, float *mean, float *invstd, float *total,
                                 int64_t *num_batches_tracked, u2mkd_stream_t s) {
    U2_REQUIRE(gathered && mean && invstd && total && world > 0 && c > 0, "u2mkd_bn_merge_stats_counted: bad arguments");
    hipLaunchKernelGGL(bn_sync_merge_kernel, dim3((unsigned)ceil_div(c, 64)), dim3(64), 0, as_stream(s), gathered, world, c,
                       eps, momentum, running_mean, running_var, mean, invstd, total, num_batches_tracked);
    return check_launch("u2mkd_bn_merge_stats_counted");
}

int u2mkd_bn_apply(const float *x, int64_t n, int32_t c, const float *mean, const float *invstd, const float *gamma,
                   const float *beta, int32_t relu, float *y, u2mkd_stream_t s) {
    return bn_apply_impl<float>(x, n, c, mean, invstd, gamma, beta, relu, y, s);
}

int u2mkd_bn_backward_local(const float *dy, const float *x, int64_t n, int32_t c, const float *mean,
                            const float *invstd, const float *gamma, const float *beta, int32_t relu, float *partial,
                            float *sums /*[2c]: dbeta, dgamma of this rank*/, u2mkd_stream_t s) {
    return bn_backward_local_impl<float>(dy, x, n, c, mean, invstd, gamma, beta, relu, partial, sums, s);
}

int u2mkd_bn_backward_apply(const float *dy, const float *x, int64_t n, int32_t c, const float *total_n,
                            const float *mean, const float *invstd, const float *gamma, const float *beta,
                            int32_t relu, const float *sums /*[2c] summed over ranks*/, float *dx, u2mkd_stream_t s) {
    return bn_backward_apply_impl<float>(dy, x, n, c, total_n, mean, invstd, gamma, beta, relu, sums, dx, s);
}

/* the three pieces with the residual branch of a ResidualBlock: y = relu(bn(x) + res); the backward recomputes the mask
 * from x and res and returns dres = the masked dy */
int u2mkd_bn_apply_res(const float *x, const float *res, int64_t n, int32_t c, const float *mean, const float *invstd,
                       const float *gamma, const float *beta, int32_t relu, float *y, u2mkd_stream_t s) {
    return bn_apply_impl<float>(x, n, c, mean, invstd, gamma, beta, relu, y, s, res);
}

int u2mkd_bn_backward_local_res(const float *dy, const float *x, const float *res, int64_t n, int32_t c, const float *mean,
                                const float *invstd, const float *gamma, const float *beta, int32_t relu, float *partial,
                                float *sums, u2mkd_stream_t s) {
    return bn_backward_local_impl<float>(dy, x, n, c, mean, invstd, gamma, beta, relu, partial, sums, s, res);
}

int u2mkd_bn_backward_apply_res(const float *dy, const float *x, const float *res, int64_t n, int32_t c, const float *total_n,
                                const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                                const float *sums, float *dx, float *dres, u2mkd_stream_t s) {
    return bn_backward_apply_impl<float>(dy, x, n, c, total_n, mean, invstd, gamma, beta, relu, sums, dx, s, res, dres);
}

/* ---- the same on BF16 rows (x, res, y, dy, dx, dres are bf16 [n, c]; every statistic, parameter and sum fp32) ---- */
int u2mkd_bn_train_forward_res_bf16(const void *x, const void *res, int64_t n, int32_t c, const float *gamma,
                                    const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                    int64_t *num_batches_tracked, int32_t relu, float *partial, float *mean, float *invstd,
                                    void *y, u2mkd_stream_t s) {
    return bn_train_forward_impl<bf16row>(BF(x), BF(res), n, c, gamma, beta, eps, momentum, running_mean, running_var,
                                          num_batches_tracked, relu, partial, mean, invstd, BFW(y), s);
}

int u2mkd_bn_eval_forward_res_bf16(const void *x, const void *res, int64_t n, int32_t c, const float *gamma, const float *beta,
                                   float eps, const float *running_mean, const float *running_var, int32_t relu,
                                   float *invstd, void *y, u2mkd_stream_t s) {
    return bn_eval_forward_impl<bf16row>(BF(x), BF(res), n, c, gamma, beta, eps, running_mean, running_var, relu, invstd,
                                         BFW(y), s);
}

int u2mkd_bn_backward_res_bf16(const void *dy, const void *x, const void *res, int64_t n, int32_t c, const float *mean,
                               const float *invstd, const float *gamma, const float *beta, int32_t relu, int32_t training,
                               float *partial, float *dgamma, float *dbeta, void *dx, void *dres, u2mkd_stream_t s) {
    return bn_backward_impl<bf16row>(BF(dy), BF(x), BF(res), n, c, mean, invstd, gamma, beta, relu, training, partial, dgamma,
                                     dbeta, BFW(dx), BFW(dres), s);
}

int u2mkd_bn_local_stats_bf16(const void *x, int64_t n, int32_t c, float *partial, float *stats, u2mkd_stream_t s) {
    return bn_local_stats_impl<bf16row>(BF(x), n, c, partial, stats, s);
}

int u2mkd_bn_apply_bf16(const void *x, int64_t n, int32_t c, const float *mean, const float *invstd, const float *gamma,
                        const float *beta, int32_t relu, void *y, u2mkd_stream_t s) {
    return bn_apply_impl<bf16row>(BF(x), n, c, mean, invstd, gamma, beta, relu, BFW(y), s);
}

int u2mkd_bn_backward_local_bf16(const void *dy, const void *x, int64_t n, int32_t c, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, int32_t relu, float *partial, float *sums,
                                 u2mkd_stream_t s) {
    return bn_backward_local_impl<bf16row>(BF(dy), BF(x), n, c, mean, invstd, gamma, beta, relu, partial, sums, s);
}

int u2mkd_bn_backward_apply_bf16(const void *dy, const void *x, int64_t n, int32_t c, const float *total_n, const float *mean,
                                 const float *invstd, const float *gamma, const float *beta, int32_t relu, const float *sums,
                                 void *dx, u2mkd_stream_t s) {
    return bn_backward_apply_impl<bf16row>(BF(dy), BF(x), n, c, total_n, mean, invstd, gamma, beta, relu, sums, BFW(dx), s);
}

int u2mkd_bn_apply_res_bf16(const void *x, const void *res, int64_t n, int32_t c, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, int32_t relu, void *y, u2mkd_stream_t s) {
    return bn_apply_impl<bf16row>(BF(x), n, c, mean, invstd, gamma, beta, relu, BFW(y), s, BF(res));
}

int u2mkd_bn_backward_local_res_bf16(const void *dy, const void *x, const void *res, int64_t n, int32_t c, const float *mean,
                                     const float *invstd, const float *gamma, const float *beta, int32_t relu, float *partial,
                                     float *sums, u2mkd_stream_t s) {
    return bn_backward_local_impl<bf16row>(BF(dy), BF(x), n, c, mean, invstd, gamma, beta, relu, partial, sums, s, BF(res));
}

/* u2mkd_bn_backward_local(_res)(_bf16) with a SECOND copy of the sums: `sums` goes into the all_reduce (in place), `keep` [2c]
 * stays this rank's (the parameter gradients, which DDP averages) -- the copy kernel in between is gone.  bf16_rows != 0: dy, x,
 * res are bf16 rows; res may be NULL. */
int u2mkd_bn_backward_local_keep(const void *dy, const void *x, const void *res, int32_t bf16_rows, int64_t n, int32_t c,
                                 const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                                 float *partial, float *sums, float *keep, u2mkd_stream_t s) {
    U2_REQUIRE(keep, "u2mkd_bn_backward_local_keep: null pointer");
    if (bf16_rows)
        return bn_backward_local_impl<bf16row>(BF(dy), BF(x), n, c, mean, invstd, gamma, beta, relu, partial, sums, s, BF(res), keep);
    return bn_backward_local_impl<float>(reinterpret_cast<const float *>(dy), reinterpret_cast<const float *>(x), n, c, mean, invstd,
                                         gamma, beta, relu, partial, sums, s, reinterpret_cast<const float *>(res), keep);
}

int u2mkd_bn_backward_apply_res_bf16(const void *dy, const void *x, const void *res, int64_t n, int32_t c, const float *total_n,
                                     const float *mean, const float *invstd, const float *gamma, const float *beta,
                                     int32_t relu, const float *sums, void *dx, void *dres, u2mkd_stream_t s) {
    return bn_backward_apply_impl<bf16row>(BF(dy), BF(x), n, c, total_n, mean, invstd, gamma, beta, relu, sums, BFW(dx), s,
                                           BF(res), BFW(dres));
}

}  // extern "C"
