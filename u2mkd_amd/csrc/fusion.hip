// Point <-> pixel index plans of the LiDAR/camera fusion (one launch each instead of ~30 small torch launches).
//
// Replaces the index arithmetic of the reference's Python loops over (sample, camera, scale):
//   camera -> LiDAR  Feature_Gather + the per-camera masked overwrite (core/models/fusion_blocks.py:241-254,
//                    spvcnn_swiftnet18_spformer_tsd_full.py:482-495): grid_sample(bilinear, zeros, align_corners=True)
//                    of every camera map at every point, later cameras overwrite earlier ones;
//   LiDAR -> camera  the multi-scale pixel mean (tsd_full.py:448-478): uv = floor((co + 1) / 2 * (size - 1)), mean of
//                    the point features per pixel.
// Both reduce to index lists consumed by u2mkd_devoxelize_forward (<= 4 weighted corners per point) and
// u2mkd_segment_sum (entries grouped by pixel / by point); this file computes the lists.  The arithmetic is the torch
// formulation's, operation for operation in fp32 (no fused multiply-add), so the plans are bit-identical to it.
#include "common.h"

namespace u2mkd {

constexpr int kFuThreads = 256;

// camera -> LiDAR: idx [n, 8] / w [n, 8] (slots 4..7 unused) of the 4 bilinear corners of every point in the
// [ncam, h, w] maps of its sample, rows of the channel-last matrix [(b * ncam + cam) * h * w + y * w + x]
__global__ void __launch_bounds__(kFuThreads)
c2l_plan_kernel(const float *__restrict__ pc /*[ncam,n,2]*/, const uint8_t *__restrict__ mask /*[ncam,n]*/, int ncam,
                int64_t n, int sample, int h, int w, int32_t *__restrict__ idx8, float *__restrict__ w8) {
#pragma clang fp contract(off)      // hipcc contracts a * b - c into one fma by default (also through the __f*_rn
                                    // helpers, which are plain operators in a header): the torch ops round after each step
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cam = -1;
    for (int c = 0; c < ncam; ++c)
        if (mask[(int64_t)c * n + i]) cam = c;                       // the LAST camera that sees the point
    const bool seen = cam >= 0;
    if (cam < 0) cam = 0;
    const float cx = pc[((int64_t)cam * n + i) * 2], cy = pc[((int64_t)cam * n + i) * 2 + 1];
    // x = (cx + 1.0) * 0.5 * (w - 1)
    const float x = ((cx + 1.0f) * 0.5f) * (float)(w - 1);
    const float y = ((cy + 1.0f) * 0.5f) * (float)(h - 1);
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int64_t base = ((int64_t)sample * ncam + cam) * ((int64_t)h * w);
    int32_t *oi = idx8 + i * 8;
    float *ow = w8 + i * 8;
    int s = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const float wy = dy ? fy : 1.0f - fy;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx, ++s) {
            const float wx = dx ? fx : 1.0f - fx;
            const long long xi = (long long)x0 + dx, yi = (long long)y0 + dy;
            const bool ok = seen && xi >= 0 && xi < w && yi >= 0 && yi < h;
            oi[s] = ok ? (int32_t)(base + yi * w + xi) : -1;
            ow[s] = ok ? wx * wy : 0.f;
        }
    }
#pragma unroll
    for (; s < 8; ++s) { oi[s] = -1; ow[s] = 0.f; }
}

// LiDAR -> camera, one grid size: per (camera, point) entry e = e0 + cam * n + i its pixel id, its destination key
// (pixel, or -1 when the camera does not see the point), its source key (global point row or -1) and its point row
__global__ void __launch_bounds__(kFuThreads)
l2c_keys_kernel(const float *__restrict__ pc, const uint8_t *__restrict__ mask, int ncam, int64_t n, int sample,
                int64_t row0, int64_t e0, int ch, int cw, int32_t *__restrict__ pix, int32_t *__restrict__ key_d,
                int32_t *__restrict__ key_s /*or null*/, int32_t *__restrict__ row /*or null*/) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)ncam * n) return;
    const int cam = (int)(t / n);
    const int64_t i = t - (int64_t)cam * n;
    const float cx = pc[t * 2], cy = pc[t * 2 + 1];
    // u = floor((cx + 1.0) / 2 * (cw - 1.0)).long().clamp(0, cw - 1)
    float fu = floorf(((cx + 1.0f) / 2.0f) * ((float)cw - 1.0f));
    float fv = floorf(((cy + 1.0f) / 2.0f) * ((float)ch - 1.0f));
    // (float -> int64 of an out-of-range or NaN value is undefined in torch as well; the mask drops such entries)
    long long u = (fu >= -9.0e18f && fu <= 9.0e18f) ? (long long)fu : 0, v = (fv >= -9.0e18f && fv <= 9.0e18f) ? (long long)fv : 0;
    u = u < 0 ? 0 : (u > cw - 1 ? cw - 1 : u);
    v = v < 0 ? 0 : (v > ch - 1 ? ch - 1 : v);
    const int32_t p = (int32_t)((((int64_t)sample * ncam + cam) * ch + v) * cw + u);
    const bool m = mask[t] != 0;
    pix[e0 + t] = p;
    key_d[e0 + t] = m ? p : -1;
    if (key_s) key_s[e0 + t] = m ? (int32_t)(row0 + i) : -1;
    if (row) row[e0 + t] = (int32_t)(row0 + i);
}

// the two entry lists of one grid size from the grouped orders: forward (entries by pixel: source row + 1 / count of
// the pixel) and backward (entries by point: pixel + the same weight)
__global__ void __launch_bounds__(kFuThreads)
l2c_finish_kernel(const int32_t *__restrict__ order_d, const int32_t *__restrict__ seg_d, const int32_t *__restrict__ order_s,
                  const int32_t *__restrict__ pix, const int32_t *__restrict__ row, int64_t e, int32_t *__restrict__ fwd_row,
                  float *__restrict__ fwd_w, int32_t *__restrict__ bwd_pix, float *__restrict__ bwd_w) {
#pragma clang fp contract(off)
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= e) return;
    {
        const int ed = order_d[p], px = pix[ed];
        int cnt = seg_d[px + 1] - seg_d[px];
        cnt = cnt < 1 ? 1 : cnt;
        fwd_row[p] = row[ed];
        fwd_w[p] = 1.0f / (float)cnt;
    }
    {
        const int es = order_s[p], px = pix[es];
        int cnt = seg_d[px + 1] - seg_d[px];
        cnt = cnt < 1 ? 1 : cnt;
        bwd_pix[p] = px;
        bwd_w[p] = 1.0f / (float)cnt;
    }
}

}  // namespace u2mkd

using namespace u2mkd;

// ---- camera -> LiDAR select + pseudo-feature MSE of one fusion stage (tsd_full.py:489-498) -----------------------------
// img_feat = where(fov, gathered, pseudo); mse = MSELoss()(pseudo[fov], gathered[fov].detach()) -- in torch: a where, and
// sub / pow / mul / two sums / clamp / div for the loss, ~20 more element-wise launches in the backward, four stages per
// step, every one a pass over [N, C].  Here: one forward pass (the selected rows + per-workgroup partial sums of the
// squared differences, merged in workgroup order by a one-workgroup kernel: reproducible), one backward pass.
constexpr int kSelWg = 512;        // workgroups of the forward at most (= partial sums the finish kernel merges)

__global__ void __launch_bounds__(256)
select_mse_fwd_kernel(const float *__restrict__ gathered, const float *__restrict__ pseudo, const uint8_t *__restrict__ fov,
                      int64_t n, int c4, float *__restrict__ out, float *__restrict__ partial /*[grid][2]*/) {
    float sq = 0.f, cnt = 0.f;
    const int64_t total = n * c4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / c4;
        const bool m = fov[row] != 0;
        const float4 g = reinterpret_cast<const float4 *>(gathered)[e], p = reinterpret_cast<const float4 *>(pseudo)[e];
        reinterpret_cast<float4 *>(out)[e] = m ? g : p;
        if (m) {
            const float dx = p.x - g.x, dy = p.y - g.y, dz = p.z - g.z, dw = p.w - g.w;
            sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            if (e - row * c4 == 0) cnt += 1.f;
        }
    }
    __shared__ float red[4][2];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sq += __shfl_xor(sq, off); cnt += __shfl_xor(cnt, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sq; red[threadIdx.x >> 6][1] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        partial[2 * blockIdx.x + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

// loss = sum / max(count * C, 1);  stats = {loss, 2 / max(count * C, 1)} (the backward's factor)
__global__ void __launch_bounds__(64)
select_mse_finish_kernel(const float *__restrict__ partial, int g, int c, float *__restrict__ stats) {
    double sq = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < g; i += 64) { sq += partial[2 * i]; cnt += partial[2 * i + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sq += __shfl_xor(sq, off); cnt += __shfl_xor(cnt, off); }
    if (threadIdx.x == 0) {
        const double den = cnt * c < 1.0 ? 1.0 : cnt * c;
        stats[0] = (float)(sq / den);
        stats[1] = (float)(2.0 / den);
    }
}

// d_gathered = fov ? g_out : 0;  d_pseudo = fov ? g_loss * (2 / den) * (pseudo - gathered) : g_out
__global__ void __launch_bounds__(256)
select_mse_bwd_kernel(const float *__restrict__ g_out, const float *__restrict__ g_loss, const float *__restrict__ stats,
                      const float *__restrict__ gathered, const float *__restrict__ pseudo, const uint8_t *__restrict__ fov,
                      int64_t n, int c4, float *__restrict__ d_gathered, float *__restrict__ d_pseudo) {
    const int64_t total = n * c4;
    const float k = (g_loss ? g_loss[0] : 0.f) * stats[1];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const bool m = fov[e / c4] != 0;
        const float4 go = g_out ? reinterpret_cast<const float4 *>(g_out)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), dp = go;
        if (m) {
            const float4 g = reinterpret_cast<const float4 *>(gathered)[e], p = reinterpret_cast<const float4 *>(pseudo)[e];
            dg = go;
            dp = make_float4(k * (p.x - g.x), k * (p.y - g.y), k * (p.z - g.z), k * (p.w - g.w));
        }
        if (d_gathered) reinterpret_cast<float4 *>(d_gathered)[e] = dg;
        reinterpret_cast<float4 *>(d_pseudo)[e] = dp;
    }
}

extern "C" {

int u2mkd_c2l_plan(const float *pixel_coords, const uint8_t *mask, int32_t ncam, int64_t n, int32_t sample, int32_t h,
                   int32_t w, int32_t *idx8, float *w8, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(pixel_coords && mask && idx8 && w8, "u2mkd_c2l_plan: null pointer");
    U2_REQUIRE(ncam > 0 && h > 0 && w > 0 && sample >= 0, "u2mkd_c2l_plan: bad shape");
    U2_REQUIRE(((int64_t)sample + 1) * ncam * h * w < ((int64_t)1 << 31), "u2mkd_c2l_plan: the feature matrix has more than 2^31 rows");
    hipLaunchKernelGGL(c2l_plan_kernel, dim3((unsigned)ceil_div(n, kFuThreads)), dim3(kFuThreads), 0, as_stream(s), pixel_coords,
                       mask, ncam, n, sample, h, w, idx8, w8);
    return check_launch("u2mkd_c2l_plan");
}

int u2mkd_l2c_keys(const float *pixel_coords, const uint8_t *mask, int32_t ncam, int64_t n, int32_t sample, int64_t row0,
                   int64_t e0, int32_t ch, int32_t cw, int32_t *pix, int32_t *key_d, int32_t *key_s, int32_t *row,
                   u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(pixel_coords && mask && pix && key_d, "u2mkd_l2c_keys: null pointer");
    U2_REQUIRE(ncam > 0 && ch > 0 && cw > 0 && sample >= 0, "u2mkd_l2c_keys: bad shape");
    U2_REQUIRE(((int64_t)sample + 1) * ncam * ch * cw < ((int64_t)1 << 31) && row0 + n < ((int64_t)1 << 31),
               "u2mkd_l2c_keys: index range beyond int32");
    hipLaunchKernelGGL(l2c_keys_kernel, dim3((unsigned)ceil_div((int64_t)ncam * n, kFuThreads)), dim3(kFuThreads), 0, as_stream(s),
                       pixel_coords, mask, ncam, n, sample, row0, e0, ch, cw, pix, key_d, key_s, row);
    return check_launch("u2mkd_l2c_keys");
}

int u2mkd_l2c_finish(const int32_t *order_d, const int32_t *seg_d, const int32_t *order_s, const int32_t *pix,
                     const int32_t *row, int64_t n_entries, int32_t *fwd_row, float *fwd_w, int32_t *bwd_pix, float *bwd_w,
                     u2mkd_stream_t s) {
    if (n_entries == 0) return 0;
    U2_REQUIRE(order_d && seg_d && order_s && pix && row && fwd_row && fwd_w && bwd_pix && bwd_w, "u2mkd_l2c_finish: null pointer");
    hipLaunchKernelGGL(l2c_finish_kernel, dim3((unsigned)ceil_div(n_entries, kFuThreads)), dim3(kFuThreads), 0, as_stream(s),
                       order_d, seg_d, order_s, pix, row, n_entries, fwd_row, fwd_w, bwd_pix, bwd_w);
    return check_launch("u2mkd_l2c_finish");
}


int32_t u2mkd_select_mse_partials(void) { return kSelWg; }

int u2mkd_select_mse_forward(const float *gathered, const float *pseudo, const uint8_t *fov, int64_t n, int32_t c, float *out,
                             float *partial, float *stats, u2mkd_stream_t s) {
    U2_REQUIRE(c > 0 && c % 4 == 0, "u2mkd_select_mse_forward: %d channels, need a positive multiple of 4", c);
    U2_REQUIRE(stats && partial, "u2mkd_select_mse_forward: null pointer");
    int g = n > 0 ? (int)(ceil_div(n * (c / 4), 256) < kSelWg ? ceil_div(n * (c / 4), 256) : kSelWg) : 0;
    if (g > 0) {
        U2_REQUIRE(gathered && pseudo && fov && out, "u2mkd_select_mse_forward: null pointer");
        hipLaunchKernelGGL(select_mse_fwd_kernel, dim3(g), dim3(256), 0, as_stream(s), gathered, pseudo, fov, n, c / 4, out, partial);
    }
    hipLaunchKernelGGL(select_mse_finish_kernel, dim3(1), dim3(64), 0, as_stream(s), partial, g, c, stats);
    return check_launch("u2mkd_select_mse_forward");
}

int u2mkd_select_mse_backward(const float *g_out, const float *g_loss, const float *stats, const float *gathered,
                              const float *pseudo, const uint8_t *fov, int64_t n, int32_t c, float *d_gathered, float *d_pseudo,
                              u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(c > 0 && c % 4 == 0 && stats && gathered && pseudo && fov && d_pseudo, "u2mkd_select_mse_backward: bad arguments");
    const int g = (int)(ceil_div(n * (c / 4), 256) < 2048 ? ceil_div(n * (c / 4), 256) : 2048);
    hipLaunchKernelGGL(select_mse_bwd_kernel, dim3(g), dim3(256), 0, as_stream(s), g_out, g_loss, stats, gathered, pseudo, fov, n,
                       c / 4, d_gathered, d_pseudo);
    return check_launch("u2mkd_select_mse_backward");
}

}  // extern "C"
