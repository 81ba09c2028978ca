// The full-resolution tail of the camera branch's pixel head, evaluated only where it is read.
//
// The reference up-samples the decoder's [b*ncam, 128, H/2, W/2] map to the image size, runs BatchNorm + ReLU + a 1x1
// classifier over all b*ncam*H*W pixels (core/models/image_branch/swiftnet.py forward_up + the `classifier_pix`
// BNReluConv of spvcnn_swiftnet18_spformer_tsd_full.py) and then reads the logits at the <= 4 bilinear corners of every
// LiDAR point (Feature_Fetch, core/models/fusion_blocks.py:257-278): 6 x 128 x 360 x 640 floats = 708 MB per tensor
// (4.4 GB at 900 x 1600), forward and backward, to produce 80 000 x 17 numbers.  Everything after the up-sampling is
// per pixel EXCEPT the BatchNorm's batch statistics, and those are sums over the up-sampled map U = Wy X Wx^T that can
// be taken on the low-resolution map X exactly:
//     sum U   = a^T X b,                a = Wy^T 1, b = Wx^T 1             (column sums of the interpolation matrices)
//     sum U^2 = < X, Ay X Ax >,         Ay = Wy^T Wy, Ax = Wx^T Wx         (tridiagonal: a 9-point stencil on X)
// and the dense part of the BatchNorm backward, dU = c0 + c1 U for every pixel nobody reads, folds back the same way:
//     Wy^T (c0 1 1^T + c1 U) Wx = c0 a b^T + c1 Ay X Ax.
// The kernels here: the corner composition (full-resolution corner -> 4 low-resolution corners), the two sums and the
// dense gradient term.  Gather, normalisation of the sampled rows and the classifier are the existing row operators.
#include "common.h"

namespace u2mkd {

constexpr int kPhThreads = 256;

// sample t = point * 4 + corner: the 4 low-resolution bilinear sources of full-resolution pixel idx8[point][corner]
// (F.interpolate(mode='bilinear', align_corners=True): src = scale * dst in fp32, lambda = src - floor)
__global__ void __launch_bounds__(kPhThreads)
up_plan_kernel(const int32_t *__restrict__ idx8, int64_t n, int H, int W, int h, int w, float rh, float rw,
               int32_t *__restrict__ idx_out, float *__restrict__ w_out) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 4) return;
    const int64_t f = idx8[(t >> 2) * 8 + (t & 3)];
    int32_t *oi = idx_out + t * 8;
    float *ow = w_out + t * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) { oi[s] = -1; ow[s] = 0.f; }
    if (f < 0) return;
    const int64_t hw_full = (int64_t)H * W;
    const int64_t img = f / hw_full;
    const int rem = (int)(f - img * hw_full);
    const int y = rem / W, x = rem - y * W;
    const float h1r = rh * (float)y, w1r = rw * (float)x;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < h - 1 ? 1 : 0, w1p = w1 < w - 1 ? 1 : 0;
    const float hl1 = h1r - (float)h1, wl1 = w1r - (float)w1;
    const float hl0 = 1.f - hl1, wl0 = 1.f - wl1;
    const int64_t base = img * (int64_t)h * w;
    oi[0] = (int32_t)(base + (int64_t)h1 * w + w1);                 ow[0] = hl0 * wl0;
    oi[1] = (int32_t)(base + (int64_t)h1 * w + w1 + w1p);           ow[1] = hl0 * wl1;
    oi[2] = (int32_t)(base + (int64_t)(h1 + h1p) * w + w1);         ow[2] = hl1 * wl0;
    oi[3] = (int32_t)(base + (int64_t)(h1 + h1p) * w + w1 + w1p);   ow[3] = hl1 * wl1;
}

// (Ay X Ax)[i][j] of one plane, X shifted by k0; ay / ax = the three diagonals [lower | main | upper]
__device__ __forceinline__ float stencil(const float *__restrict__ xp, int i, int j, int h, int w,
                                         const float *__restrict__ ay, const float *__restrict__ ax, float k0) {
    float t = 0.f;
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
        const int ii = i + di;
        if (ii < 0 || ii >= h) continue;
        const float cy = ay[(di + 1) * h + i];
        float r = 0.f;
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            const int jj = j + dj;
            if (jj < 0 || jj >= w) continue;
            r += ax[(dj + 1) * w + j] * (xp[(int64_t)ii * w + jj] - k0);
        }
        t += cy * r;
    }
    return t;
}

// partial[plane][chunk] = (sum a_i b_j (X - K), sum (X - K) (Ay (X - K) Ax)) over the chunk's rows; K = the channel's
// first element (image 0): the sums of the up-sampled map shifted by K, since every interpolation row sums to 1
__global__ void __launch_bounds__(kPhThreads)
upbn_stats_kernel(const float *__restrict__ X, int C, int h, int w, const float *__restrict__ a, const float *__restrict__ b,
                  const float *__restrict__ ay, const float *__restrict__ ax, int rows_per_chunk,
                  float *__restrict__ partial) {
    __shared__ float red[2][kPhThreads / 64];
    const int plane = blockIdx.x, chunk = blockIdx.y;
    const int c = plane % C;
    const float k0 = X[(int64_t)c * h * w];
    const float *xp = X + (int64_t)plane * h * w;
    const int r0 = chunk * rows_per_chunk, r1 = min(h, r0 + rows_per_chunk);
    float s1 = 0.f, s2 = 0.f;
    for (int e = threadIdx.x; e < (r1 - r0) * w; e += kPhThreads) {
        const int i = r0 + e / w, j = e % w;
        const float xc = xp[(int64_t)i * w + j] - k0;
        s1 += a[i] * b[j] * xc;
        s2 += xc * stencil(xp, i, j, h, w, ay, ax, k0);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *o = partial + ((int64_t)plane * gridDim.y + chunk) * 2;
        o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// dX = c0[c] a_i b_j + c1[c] (Ay X Ax)[i][j]
__global__ void __launch_bounds__(kPhThreads)
upbn_dense_grad_kernel(const float *__restrict__ X, int64_t total, int C, int h, int w, const float *__restrict__ a,
                       const float *__restrict__ b, const float *__restrict__ ay, const float *__restrict__ ax,
                       const float *__restrict__ c0, const float *__restrict__ c1, float *__restrict__ dX) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e % w);
    const int64_t r = e / w;
    const int i = (int)(r % h);
    const int64_t plane = r / h;
    const int c = (int)(plane % C);
    dX[e] = c0[c] * a[i] * b[j] + c1[c] * stencil(X + plane * (int64_t)h * w, i, j, h, w, ay, ax, 0.f);
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

int u2mkd_up_plan(const int32_t *idx8, int64_t n, int32_t H, int32_t W, int32_t h, int32_t w, float rh, float rw,
                  int32_t *idx_out, float *w_out, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(idx8 && idx_out && w_out, "u2mkd_up_plan: null pointer");
    U2_REQUIRE(H > 0 && W > 0 && h > 0 && w > 0 && h <= H && w <= W, "u2mkd_up_plan: bad sizes %dx%d <- %dx%d", H, W, h, w);
    U2_REQUIRE(n * 4 < ((int64_t)1 << 31), "u2mkd_up_plan: more than 2^31 samples");
    hipLaunchKernelGGL(up_plan_kernel, dim3((unsigned)ceil_div(n * 4, kPhThreads)), dim3(kPhThreads), 0, as_stream(s), idx8, n,
                       H, W, h, w, rh, rw, idx_out, w_out);
    return check_launch("u2mkd_up_plan");
}

int u2mkd_upbn_stats(const float *x, int32_t n_img, int32_t c, int32_t h, int32_t w, const float *a, const float *b,
                     const float *ay, const float *ax, int32_t rows_per_chunk, float *partial, u2mkd_stream_t s) {
    U2_REQUIRE(x && a && b && ay && ax && partial, "u2mkd_upbn_stats: null pointer");
    U2_REQUIRE(n_img > 0 && c > 0 && h > 0 && w > 0 && rows_per_chunk > 0, "u2mkd_upbn_stats: bad shape");
    const int chunks = (int)ceil_div(h, rows_per_chunk);
    U2_REQUIRE((int64_t)n_img * c < ((int64_t)1 << 31) && chunks <= 65535, "u2mkd_upbn_stats: grid too large");
    hipLaunchKernelGGL(upbn_stats_kernel, dim3((unsigned)(n_img * c), (unsigned)chunks), dim3(kPhThreads), 0, as_stream(s), x, c,
                       h, w, a, b, ay, ax, rows_per_chunk, partial);
    return check_launch("u2mkd_upbn_stats");
}

int u2mkd_upbn_dense_grad(const float *x, int32_t n_img, int32_t c, int32_t h, int32_t w, const float *a, const float *b,
                          const float *ay, const float *ax, const float *c0, const float *c1, float *dx,
                          u2mkd_stream_t s) {
    U2_REQUIRE(x && a && b && ay && ax && c0 && c1 && dx, "u2mkd_upbn_dense_grad: null pointer");
    U2_REQUIRE(n_img > 0 && c > 0 && h > 0 && w > 0, "u2mkd_upbn_dense_grad: bad shape");
    const int64_t total = (int64_t)n_img * c * h * w;
    hipLaunchKernelGGL(upbn_dense_grad_kernel, dim3((unsigned)ceil_div(total, kPhThreads)), dim3(kPhThreads), 0, as_stream(s), x,
                       total, c, h, w, a, b, ay, ax, c0, c1, dx);
    return check_launch("u2mkd_upbn_dense_grad");
}

}  // extern "C"
