// BatchNorm2d over NCHW feature maps, fused with the ReLU and the residual add that follow it (gfx950).
//
// Replaces the nn.BatchNorm2d -> nn.ReLU pairs (and bn2(conv2(.)) + identity -> ReLU of BasicBlock) of the SwiftNet-18
// camera branch and of the LiDAR->camera fusion convs (core/models/image_branch/swiftnet.py:20-50, 114-341;
// tsd_full.py L2C blocks) on 6 x C x H x W fp32 maps.  The 2-D convolutions themselves stay on MIOpen (the north
// star prescribes it); MIOpen's spatial BatchNorm launches ONE workgroup per channel -- 64 or 128 workgroups on 256
// CUs, 1.7-2 TB/s on the 88-700 MB maps of this branch -- and the ReLU / residual add around it are separate
// elementwise passes.  Here the map is cut into (plane, chunk) pieces, thousands of workgroups:
//   pass 1  per piece: count, mean and centred M2 from shifted sums (shift = the piece's first element: the sums
//           stay at the scale of the spread, no cancellation), one read of x;
//   pass 2  per channel: the pieces merged with Chan's update in a fixed order (64 lanes, then a fixed tree) ->
//           mean, invstd, running statistics (momentum, unbiased variance), num_batches_tracked;
//   pass 3  y = (x - mean) * invstd * gamma + beta [+ res] [ReLU], float4 over the piece.
// Backward: per piece sums of dy' and dy' * xhat (dy' = dy masked by the fused ReLU, recomputed from x [and res]),
// fixed-order merge -> dgamma / dbeta, then dx = gamma * invstd * (dy' - mean(dy') - xhat * mean(dy' * xhat)) and
// dres = dy'.  Deterministic (no atomics).  HBM-bound: 3 passes over the map forward, 5 backward.
#include "common.h"

namespace u2mkd {

constexpr int kB2Threads = 256;
constexpr int kB2Chunk = 8192;          // elements of one plane per statistics workgroup
constexpr int kB2ApplyChunk = 4096;     // elements of one plane per elementwise workgroup

__host__ __device__ inline int b2_splits(int hw, int chunk) { return (hw + chunk - 1) / chunk; }

// block-wide sum of two floats, result valid in thread 0 (fixed order: lanes by shuffle tree, waves in order)
__device__ __forceinline__ void b2_block_sum2(float &a, float &b) {
    __shared__ float s_a[kB2Threads / 64], s_b[kB2Threads / 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_a[wave] = a; s_b[wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = s_a[0]; b = s_b[0];
#pragma unroll
        for (int w = 1; w < kB2Threads / 64; ++w) { a += s_a[w]; b += s_b[w]; }
    }
}

// grid (splits * B, C): piece (b, c, s) -> partial[c][b * splits + s] = (mean, M2)
__global__ void __launch_bounds__(kB2Threads)
bn2d_stats_partial_kernel(const float *__restrict__ x, int C, int hw, int splits, float *__restrict__ partial) {
    const int c = blockIdx.y, b = blockIdx.x / splits, s = blockIdx.x - b * splits;
    const int lo = s * kB2Chunk, cnt = min(kB2Chunk, hw - lo);
    const float *p = x + ((size_t)b * C + c) * hw + lo;
    const float shift = p[0];
    float s1 = 0.f, s2 = 0.f;
    if ((hw & 3) == 0) {
        for (int e = threadIdx.x * 4; e < cnt; e += kB2Threads * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p + e);
            const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
            s1 += (d0 + d1) + (d2 + d3);
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    } else {
        for (int e = threadIdx.x; e < cnt; e += kB2Threads) {
            const float d = p[e] - shift;
            s1 += d;
            s2 += d * d;
        }
    }
    b2_block_sum2(s1, s2);
    if (threadIdx.x == 0) {
        const float n = (float)cnt;
        float *o = partial + ((size_t)c * (gridDim.x) + blockIdx.x) * 2;
        o[0] = shift + s1 / n;
        o[1] = fmaxf(s2 - s1 * s1 / n, 0.f);
    }
}

// one 64-lane workgroup per channel: Chan merge of the np = B * splits pieces in a fixed order
__global__ void __launch_bounds__(64)
bn2d_stats_finalize_kernel(const float *__restrict__ partial, int np, int hw, int splits, float eps, float momentum,
                           float *__restrict__ running_mean, float *__restrict__ running_var,
                           int64_t *__restrict__ num_batches_tracked, float *__restrict__ mean_out,
                           float *__restrict__ invstd_out, float *__restrict__ m2_out = nullptr) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (num_batches_tracked && c == 0 && lane == 0) *num_batches_tracked += 1;
    float na = 0.f, mean = 0.f, m2 = 0.f;
    for (int i = lane; i < np; i += 64) {
        const int s = i % splits;
        const float nb = (float)min(kB2Chunk, hw - s * kB2Chunk);
        const float mb = partial[((size_t)c * np + i) * 2], qb = partial[((size_t)c * np + i) * 2 + 1];
        const float tot = na + nb, delta = mb - mean;
        mean += delta * (nb / tot);
        m2 += qb + delta * delta * (na * nb / tot);
        na = tot;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float nb = __shfl_down(na, off, 64), mb = __shfl_down(mean, off, 64), qb = __shfl_down(m2, off, 64);
        if (nb != 0.f) {
            const float tot = na + nb, delta = mb - mean;
            mean += delta * (nb / tot);
            m2 += qb + delta * delta * (na * nb / tot);
            na = tot;
        }
    }
    if (lane != 0) return;
    const float var = m2 / na;
    mean_out[c] = mean;
    if (m2_out) {          // local statistics only (the cross-rank merge follows): stats row = mean[C], M2[C], count
        m2_out[c] = m2;
        if (c == 0) m2_out[gridDim.x] = na;
        return;
    }
    invstd_out[c] = 1.f / sqrtf(var + eps);
    if (running_mean) {
        const float unbiased = na > 1.f ? m2 / (na - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// grid (asplits * B, C).  var != nullptr: evaluation mode (mean = running mean, invstd from the running variance)
__global__ void __launch_bounds__(kB2Threads)
bn2d_apply_kernel(const float *__restrict__ x, const float *__restrict__ res, int C, int hw, int asplits,
                  const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ var,
                  float eps, const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                  float *__restrict__ y) {
    const int c = blockIdx.y, b = blockIdx.x / asplits, s = blockIdx.x - b * asplits;
    const int lo = s * kB2ApplyChunk, cnt = min(kB2ApplyChunk, hw - lo);
    const size_t base = ((size_t)b * C + c) * hw + lo;
    const float is = var ? 1.f / sqrtf(var[c] + eps) : invstd[c];
    const float m = mean[c], g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    // the expression of the backward's ReLU mask, operation for operation
    auto one = [&](float v, float r) {
        const float h = (v - m) * is;
        float o = h * g + bt + r;
        return relu ? fmaxf(o, 0.f) : o;
    };
    if ((hw & 3) == 0) {
        for (int e = threadIdx.x * 4; e < cnt; e += kB2Threads * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + base + e);
            const float4 r = res ? *reinterpret_cast<const float4 *>(res + base + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(y + base + e) = make_float4(one(v.x, r.x), one(v.y, r.y), one(v.z, r.z), one(v.w, r.w));
        }
    } else {
        for (int e = threadIdx.x; e < cnt; e += kB2Threads) y[base + e] = one(x[base + e], res ? res[base + e] : 0.f);
    }
}

// grid (splits * B, C): partial[c][piece] = (sum dy', sum dy' * xhat)
__global__ void __launch_bounds__(kB2Threads)
bn2d_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ res, int C,
                        int hw, int splits, const float *__restrict__ mean, const float *__restrict__ invstd,
                        const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                        float *__restrict__ partial) {
    const int c = blockIdx.y, b = blockIdx.x / splits, s = blockIdx.x - b * splits;
    const int lo = s * kB2Chunk, cnt = min(kB2Chunk, hw - lo);
    const size_t base = ((size_t)b * C + c) * hw + lo;
    const float m = mean[c], is = invstd[c], g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    auto one = [&](float v, float d, float r) {
        const float h = (v - m) * is;
        if (relu && h * g + bt + r <= 0.f) d = 0.f;
        s1 += d;
        s2 += d * h;
    };
    if ((hw & 3) == 0) {
        for (int e = threadIdx.x * 4; e < cnt; e += kB2Threads * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + base + e);
            const float4 d = *reinterpret_cast<const float4 *>(dy + base + e);
            const float4 r = res ? *reinterpret_cast<const float4 *>(res + base + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            one(v.x, d.x, r.x); one(v.y, d.y, r.y); one(v.z, d.z, r.z); one(v.w, d.w, r.w);
        }
    } else {
        for (int e = threadIdx.x; e < cnt; e += kB2Threads) one(x[base + e], dy[base + e], (relu && res) ? res[base + e] : 0.f);
    }
    b2_block_sum2(s1, s2);
    if (threadIdx.x == 0) {
        float *o = partial + ((size_t)c * gridDim.x + blockIdx.x) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

__global__ void __launch_bounds__(64)
bn2d_bwd_finalize_kernel(const float *__restrict__ partial, int np, float *__restrict__ dbeta, float *__restrict__ dgamma,
                         float *__restrict__ keep = nullptr) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < np; i += 64) {
        s1 += partial[((size_t)c * np + i) * 2];
        s2 += partial[((size_t)c * np + i) * 2 + 1];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if (lane == 0) {
        dbeta[c] = s1; dgamma[c] = s2;
        if (keep) { keep[c] = s1; keep[gridDim.x + c] = s2; }      // (grid = the channels)
    }
}

// grid (asplits * B, C); training: dx = gamma * invstd * (dy' - dbeta / n - xhat * dgamma / n); evaluation (dbeta
// == nullptr): dx = dy' * gamma * invstd.  dres (may be null) = dy'
__global__ void __launch_bounds__(kB2Threads)
bn2d_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ res, int C,
                      int hw, int asplits, float inv_n, const float *__restrict__ mean, const float *__restrict__ invstd,
                      const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                      const float *__restrict__ dbeta, const float *__restrict__ dgamma, float *__restrict__ dx,
                      float *__restrict__ dres, const float *__restrict__ total_n = nullptr) {
    if (total_n) inv_n = 1.f / *total_n;          // SyncBatchNorm: the element count of all ranks (device side)
    const int c = blockIdx.y, b = blockIdx.x / asplits, s = blockIdx.x - b * asplits;
    const int lo = s * kB2ApplyChunk, cnt = min(kB2ApplyChunk, hw - lo);
    const size_t base = ((size_t)b * C + c) * hw + lo;
    const float m = mean[c], is = invstd[c], g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float k1 = dbeta ? dbeta[c] * inv_n : 0.f, k2 = dbeta ? dgamma[c] * inv_n : 0.f, gs = g * is;
    auto one = [&](float v, float d, float r, float &dm) {
        const float h = (v - m) * is;
        if (relu && h * g + bt + r <= 0.f) d = 0.f;
        dm = d;
        return gs * (d - k1 - h * k2);
    };
    if ((hw & 3) == 0) {
        for (int e = threadIdx.x * 4; e < cnt; e += kB2Threads * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + base + e);
            const float4 d = *reinterpret_cast<const float4 *>(dy + base + e);
            const float4 r = (relu && res) ? *reinterpret_cast<const float4 *>(res + base + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o, dm;
            o.x = one(v.x, d.x, r.x, dm.x); o.y = one(v.y, d.y, r.y, dm.y);
            o.z = one(v.z, d.z, r.z, dm.z); o.w = one(v.w, d.w, r.w, dm.w);
            *reinterpret_cast<float4 *>(dx + base + e) = o;
            if (dres) *reinterpret_cast<float4 *>(dres + base + e) = dm;
        }
    } else {
        for (int e = threadIdx.x; e < cnt; e += kB2Threads) {
            float dm;
            dx[base + e] = one(x[base + e], dy[base + e], (relu && res) ? res[base + e] : 0.f, dm);
            if (dres) dres[base + e] = dm;
        }
    }
}

static bool b2_shape_ok(int64_t b, int32_t c, int64_t hw) {
    return b > 0 && c > 0 && hw > 0 && c <= 65535 && hw < ((int64_t)1 << 30) &&
           b * b2_splits((int)hw, kB2ApplyChunk) < ((int64_t)1 << 31);
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

size_t u2mkd_bn2d_workspace_bytes(int64_t b, int32_t c, int64_t hw) {
    if (b <= 0 || c <= 0 || hw <= 0) return 0;
    return (size_t)c * b * b2_splits((int)hw, kB2Chunk) * 2 * sizeof(float);
}

int u2mkd_bn2d_train_forward(const float *x, const float *res, int64_t b, int32_t c, int64_t hw, const float *gamma,
                             const float *beta, float eps, float momentum, int32_t relu, float *running_mean,
                             float *running_var, int64_t *num_batches_tracked, void *workspace, float *mean, float *invstd,
                             float *y, u2mkd_stream_t s) {
    U2_REQUIRE(x && workspace && mean && invstd && y, "u2mkd_bn2d_train_forward: null pointer");
    U2_REQUIRE(b2_shape_ok(b, c, hw), "u2mkd_bn2d_train_forward: shape [%lld, %d, %lld] out of range", (long long)b, c, (long long)hw);
    U2_REQUIRE(b * hw > 1, "u2mkd_bn2d_train_forward: more than one value per channel is needed in training mode");
    const int splits = b2_splits((int)hw, kB2Chunk), asplits = b2_splits((int)hw, kB2ApplyChunk);
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(bn2d_stats_partial_kernel, dim3((unsigned)(splits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), x,
                       c, (int)hw, splits, partial);
    hipLaunchKernelGGL(bn2d_stats_finalize_kernel, dim3((unsigned)c), dim3(64), 0, as_stream(s), partial, (int)(splits * b),
                       (int)hw, splits, eps, momentum, running_mean, running_var, num_batches_tracked, mean, invstd);
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3((unsigned)(asplits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), x, res, c,
                       (int)hw, asplits, mean, invstd, nullptr, eps, gamma, beta, relu, y);
    return check_launch("u2mkd_bn2d_train_forward");
}

int u2mkd_bn2d_eval_forward(const float *x, const float *res, int64_t b, int32_t c, int64_t hw, const float *gamma,
                            const float *beta, float eps, int32_t relu, const float *running_mean, const float *running_var,
                            float *y, u2mkd_stream_t s) {
    U2_REQUIRE(x && running_mean && running_var && y, "u2mkd_bn2d_eval_forward: null pointer");
    U2_REQUIRE(b2_shape_ok(b, c, hw), "u2mkd_bn2d_eval_forward: shape [%lld, %d, %lld] out of range", (long long)b, c, (long long)hw);
    const int asplits = b2_splits((int)hw, kB2ApplyChunk);
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3((unsigned)(asplits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), x, res, c,
                       (int)hw, asplits, running_mean, nullptr, running_var, eps, gamma, beta, relu, y);
    return check_launch("u2mkd_bn2d_eval_forward");
}

int u2mkd_bn2d_backward(const float *dy, const float *x, const float *res, int64_t b, int32_t c, int64_t hw,
                        const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                        int32_t batch_stats, void *workspace, float *dgamma, float *dbeta, float *dx, float *dres,
                        u2mkd_stream_t s) {
    U2_REQUIRE(dy && x && mean && invstd && workspace && dgamma && dbeta && dx, "u2mkd_bn2d_backward: null pointer");
    U2_REQUIRE(b2_shape_ok(b, c, hw), "u2mkd_bn2d_backward: shape [%lld, %d, %lld] out of range", (long long)b, c, (long long)hw);
    const int splits = b2_splits((int)hw, kB2Chunk), asplits = b2_splits((int)hw, kB2ApplyChunk);
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(bn2d_bwd_partial_kernel, dim3((unsigned)(splits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), dy, x,
                       res, c, (int)hw, splits, mean, invstd, gamma, beta, relu, partial);
    hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3((unsigned)c), dim3(64), 0, as_stream(s), partial, (int)(splits * b), dbeta,
                       dgamma);
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel, dim3((unsigned)(asplits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), dy, x,
                       res, c, (int)hw, asplits, 1.f / (float)(b * hw), mean, invstd, gamma, beta, relu, batch_stats ? dbeta : nullptr,
                       batch_stats ? dgamma : nullptr, dx, dres);
    return check_launch("u2mkd_bn2d_backward");
}

/* ---- SyncBatchNorm2d in pieces (as u2mkd_bn_local_stats / _merge_stats / _apply / _backward_local / _backward_apply for
 * feature rows): local (mean, M2, count) per channel -> the caller's all_gather -> u2mkd_bn_merge_stats -> normalise
 * (+ residual, ReLU); backward: local (sum dy', sum dy' xhat) -> all_reduce -> dx with the global count. ---- */
int u2mkd_bn2d_local_stats(const float *x, int64_t b, int32_t c, int64_t hw, void *workspace, float *stats /*[2c+1]*/,
                           u2mkd_stream_t s) {
    U2_REQUIRE(x && workspace && stats, "u2mkd_bn2d_local_stats: null pointer");
    U2_REQUIRE(b2_shape_ok(b, c, hw), "u2mkd_bn2d_local_stats: shape [%lld, %d, %lld] out of range", (long long)b, c, (long long)hw);
    const int splits = b2_splits((int)hw, kB2Chunk);
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(bn2d_stats_partial_kernel, dim3((unsigned)(splits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), x,
                       c, (int)hw, splits, partial);
    hipLaunchKernelGGL(bn2d_stats_finalize_kernel, dim3((unsigned)c), dim3(64), 0, as_stream(s), partial, (int)(splits * b),
                       (int)hw, splits, 0.f, 0.f, (float *)nullptr, (float *)nullptr, (int64_t *)nullptr, stats, (float *)nullptr,
                       stats + c);
    return check_launch("u2mkd_bn2d_local_stats");
}

int u2mkd_bn2d_apply(const float *x, const float *res, int64_t b, int32_t c, int64_t hw, const float *mean,
                     const float *invstd, const float *gamma, const float *beta, int32_t relu, float *y, u2mkd_stream_t s) {
    U2_REQUIRE(x && mean && invstd && y, "u2mkd_bn2d_apply: null pointer");
    U2_REQUIRE(b2_shape_ok(b, c, hw), "u2mkd_bn2d_apply: shape [%lld, %d, %lld] out of range", (long long)b, c, (long long)hw);
    const int asplits = b2_splits((int)hw, kB2ApplyChunk);
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3((unsigned)(asplits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), x, res, c,
                       (int)hw, asplits, mean, invstd, nullptr, 0.f, gamma, beta, relu, y);
    return check_launch("u2mkd_bn2d_apply");
}

int u2mkd_bn2d_backward_local(const float *dy, const float *x, const float *res, int64_t b, int32_t c, int64_t hw,
                              const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                              void *workspace, float *sums /*[2c]: dbeta, dgamma of this rank*/, u2mkd_stream_t s) {
    U2_REQUIRE(dy && x && mean && invstd && workspace && sums, "u2mkd_bn2d_backward_local: null pointer");
    U2_REQUIRE(b2_shape_ok(b, c, hw), "u2mkd_bn2d_backward_local: shape [%lld, %d, %lld] out of range", (long long)b, c, (long long)hw);
    const int splits = b2_splits((int)hw, kB2Chunk);
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(bn2d_bwd_partial_kernel, dim3((unsigned)(splits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), dy, x,
                       res, c, (int)hw, splits, mean, invstd, gamma, beta, relu, partial);
    hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3((unsigned)c), dim3(64), 0, as_stream(s), partial, (int)(splits * b), sums,
                       sums + c);
    return check_launch("u2mkd_bn2d_backward_local");
}

/* the same with a second copy `keep` [2c] of the sums (u2mkd_bn_backward_local_keep) */
int u2mkd_bn2d_backward_local_keep(const float *dy, const float *x, const float *res, int64_t b, int32_t c, int64_t hw,
                                   const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                                   void *workspace, float *sums, float *keep, u2mkd_stream_t s) {
    U2_REQUIRE(dy && x && mean && invstd && workspace && sums && keep, "u2mkd_bn2d_backward_local_keep: null pointer");
    U2_REQUIRE(b2_shape_ok(b, c, hw), "u2mkd_bn2d_backward_local_keep: shape [%lld, %d, %lld] out of range", (long long)b, c, (long long)hw);
    const int splits = b2_splits((int)hw, kB2Chunk);
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(bn2d_bwd_partial_kernel, dim3((unsigned)(splits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), dy, x,
                       res, c, (int)hw, splits, mean, invstd, gamma, beta, relu, partial);
    hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3((unsigned)c), dim3(64), 0, as_stream(s), partial, (int)(splits * b), sums,
                       sums + c, keep);
    return check_launch("u2mkd_bn2d_backward_local_keep");
}

int u2mkd_bn2d_backward_apply(const float *dy, const float *x, const float *res, int64_t b, int32_t c, int64_t hw,
                              const float *total_n, const float *mean, const float *invstd, const float *gamma,
                              const float *beta, int32_t relu, const float *sums /*[2c] over all ranks*/, float *dx,
                              float *dres, u2mkd_stream_t s) {
    U2_REQUIRE(dy && x && total_n && mean && invstd && sums && dx, "u2mkd_bn2d_backward_apply: null pointer");
    U2_REQUIRE(b2_shape_ok(b, c, hw), "u2mkd_bn2d_backward_apply: shape [%lld, %d, %lld] out of range", (long long)b, c, (long long)hw);
    const int asplits = b2_splits((int)hw, kB2ApplyChunk);
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel, dim3((unsigned)(asplits * b), (unsigned)c), dim3(kB2Threads), 0, as_stream(s), dy, x,
                       res, c, (int)hw, asplits, 0.f, mean, invstd, gamma, beta, relu, sums, sums + c, dx, dres, total_n);
    return check_launch("u2mkd_bn2d_backward_apply");
}

}  // extern "C"
