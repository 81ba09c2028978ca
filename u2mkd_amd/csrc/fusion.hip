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

}  // extern "C"
