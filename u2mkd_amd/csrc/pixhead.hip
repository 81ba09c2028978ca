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

// Both stencil kernels work on a chunk of rows of one plane staged in LDS with one halo row either side (coalesced
// loads; the 9 neighbours then come from LDS).  s_x[(r - (r0 - 1)) * w + j] = X[r][j] - k0 for r0 - 1 <= r <= r1, 0 outside.
__device__ __forceinline__ void stage_rows(const float *__restrict__ xp, int h, int w, int r0, int r1, float k0, float *s_x) {
    const int n = (r1 - r0 + 2) * w;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = r0 - 1 + e / w;
        s_x[e] = (r >= 0 && r < h) ? xp[(int64_t)r * w + (e % w)] - k0 : 0.f;
    }
    __syncthreads();
}

// (Ay X Ax)[i][j] from the staged rows; ay / ax = the three diagonals [lower | main | upper] (zero past the borders)
__device__ __forceinline__ float stencil_lds(const float *s_x, int li, int j, int i, int h, int w, const float *__restrict__ ay,
                                             const float *__restrict__ ax) {
    const float al = j > 0 ? ax[j] : 0.f, am = ax[w + j], au = j < w - 1 ? ax[2 * w + j] : 0.f;
    const int jl = j > 0 ? j - 1 : j, ju = j < w - 1 ? j + 1 : j;
    float t = 0.f;
#pragma unroll
    for (int di = 0; di < 3; ++di) {
        const float *row = s_x + (li + di) * w;             // rows i - 1, i, i + 1 (halo rows are zero outside the plane)
        t += ay[di * h + i] * (al * row[jl] + am * row[j] + au * row[ju]);
    }
    return t;
}

// partial[plane][chunk] = (sum a_i b_j (X - K), sum (X - K) (Ay (X - K) Ax)) over the chunk's rows; K = the channel's
// first element (image 0): the sums of the up-sampled map shifted by K, since every interpolation row sums to 1
__global__ void __launch_bounds__(kPhThreads)
upbn_stats_kernel(const float *__restrict__ X, int C, int h, int w, const float *__restrict__ a, const float *__restrict__ b,
                  const float *__restrict__ ay, const float *__restrict__ ax, int rows_per_chunk,
                  float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float s_x[];
    __shared__ float red[2][kPhThreads / 64];
    const int plane = blockIdx.x, chunk = blockIdx.y;
    const int c = plane % C;
    const float k0 = X[(int64_t)c * h * w];
    const float *xp = X + (int64_t)plane * h * w;
    const int r0 = chunk * rows_per_chunk, r1 = min(h, r0 + rows_per_chunk);
    stage_rows(xp, h, w, r0, r1, k0, s_x);
    float s1 = 0.f, s2 = 0.f;
    for (int e = threadIdx.x; e < (r1 - r0) * w; e += kPhThreads) {
        const int li = e / w, j = e - li * w, i = r0 + li;
        const float xc = s_x[(li + 1) * w + j];
        s1 += a[i] * b[j] * xc;
        s2 += xc * stencil_lds(s_x, li, j, i, h, w, ay, ax);
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
upbn_dense_grad_kernel(const float *__restrict__ X, int C, int h, int w, const float *__restrict__ a,
                       const float *__restrict__ b, const float *__restrict__ ay, const float *__restrict__ ax,
                       const float *__restrict__ c0, const float *__restrict__ c1, int rows_per_chunk,
                       float *__restrict__ dX) {
    extern __shared__ __attribute__((aligned(16))) float s_x[];
    const int plane = blockIdx.x, chunk = blockIdx.y;
    const int c = plane % C;
    const float *xp = X + (int64_t)plane * h * w;
    float *dp = dX + (int64_t)plane * h * w;
    const int r0 = chunk * rows_per_chunk, r1 = min(h, r0 + rows_per_chunk);
    stage_rows(xp, h, w, r0, r1, 0.f, s_x);
    const float k0c = c0[c], k1c = c1[c];
    for (int e = threadIdx.x; e < (r1 - r0) * w; e += kPhThreads) {
        const int li = e / w, j = e - li * w, i = r0 + li;
        dp[(int64_t)i * w + j] = k0c * a[i] * b[j] + k1c * stencil_lds(s_x, li, j, i, h, w, ay, ax);
    }
}

// out[b][j][i] = in[b][i][j]: batched 2-D transpose through a 64 x 64 LDS tile (reads and writes both coalesced).  The
// camera maps are NCHW and the point <-> pixel gathers read whole channel rows: [planes][C][h*w] <-> [planes][h*w][C].
// (torch's permute().contiguous() runs this copy at ~0.7 TB/s: 0.5 ms for the 177 MB decoder map.)
constexpr int kTrTile = 64;
__global__ void __launch_bounds__(256)
transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols, float scale = 1.f) {
    __shared__ float tile[kTrTile][kTrTile + 1];
    const size_t plane = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * kTrTile, r0 = blockIdx.y * kTrTile;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4 threads
#pragma unroll
    for (int k = 0; k < kTrTile; k += 4) {
        const int r = r0 + ty + k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + k][tx] = in[plane + (size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTrTile; k += 4) {
        const int c = c0 + ty + k, r = r0 + tx;
        if (c < cols && r < rows) out[plane + (size_t)c * rows + r] = tile[tx][ty + k] * scale;
    }
}

// ---- LiDAR -> camera: the multi-scale pixel-mean grids of one fusion point combined (tsd_full.py:448-478) ---------------------
// l2c_scatter averages n grids -- the points' pixel-mean map on the (H, W) feature grid and on (H/2, W/2), (H/4, W/4), ... --
// after up-sampling each to (H, W) (bilinear, align_corners=True).  As torch operations that was n - 1 up-samplings, n - 1
// additions, a division and a layout copy over maps the LiDAR and the camera branch both wait for (1.1 GB of traffic per KD step
// forward, 1.5 GB backward with an atomic up-sampling gradient).  Here: the grids are channel-last rows [img][y][x][C]
// (what the segment-mean produces); ONE kernel sums them at full resolution (rows out, coalesced over the channels), a scaled
// LDS transpose makes the NCHW map; backward: the scaled transpose of the gradient IS the full-resolution grid's gradient,
// and every coarse grid gathers its own (tent weights, fixed order: reproducible).
struct L2cGrids {
    const float *g[4];
    int ch[4], cw[4];
    float ry[4], rx[4];
    int n;
};

__device__ __forceinline__ void l2c_taps(int d, float r, int n_src, int &i0, int &i1, float &f) {
    const float s = r * (float)d;                 // align_corners: source index = dst * (in - 1) / (out - 1)
    i0 = (int)s;
    if (i0 > n_src - 1) i0 = n_src - 1;
    f = s - (float)i0;
    i1 = i0 + (i0 < n_src - 1 ? 1 : 0);
}

__global__ void __launch_bounds__(256)
l2c_combine_rows_kernel(L2cGrids G, int n_img, int H, int W, int c4, float4 *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n_img * H * W * c4) return;
    const int c = (int)(t % c4);
    const int64_t pix = t / c4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), img = (int)(pix / ((int64_t)W * H));
    float4 acc = reinterpret_cast<const float4 *>(G.g[0])[t];
#pragma unroll
    for (int s = 1; s < 4; ++s) {      // (static indices into the argument block)
        if (s >= G.n) break;
        int y0, y1, x0, x1;
        float fy, fx;
        l2c_taps(y, G.ry[s], G.ch[s], y0, y1, fy);
        l2c_taps(x, G.rx[s], G.cw[s], x0, x1, fx);
        const float4 *g = reinterpret_cast<const float4 *>(G.g[s]) + (int64_t)img * G.ch[s] * G.cw[s] * c4 + c;
        const float4 a = g[((int64_t)y0 * G.cw[s] + x0) * c4], b = g[((int64_t)y0 * G.cw[s] + x1) * c4];
        const float4 d = g[((int64_t)y1 * G.cw[s] + x0) * c4], e = g[((int64_t)y1 * G.cw[s] + x1) * c4];
        const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
        acc.x += w00 * a.x + w01 * b.x + w10 * d.x + w11 * e.x;
        acc.y += w00 * a.y + w01 * b.y + w10 * d.y + w11 * e.y;
        acc.z += w00 * a.z + w01 * b.z + w10 * d.z + w11 * e.z;
        acc.w += w00 * a.w + w01 * b.w + w10 * d.w + w11 * e.w;
    }
    out[t] = acc;
}

// d[img][i][j][c] = sum over the full-resolution pixels (y, x) that read cell (i, j): weight(y, i) * weight(x, j) * g[img][y][x][c],
// with the weights of the forward's taps (both taps of a pixel may be the same clamped cell)
__global__ void __launch_bounds__(256)
l2c_combine_rows_bwd_kernel(const float4 *__restrict__ g, int n_img, int H, int W, int c4, int ch, int cw, float ry, float rx,
                            float4 *__restrict__ d) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n_img * ch * cw * c4) return;
    const int c = (int)(t % c4);
    const int64_t cell = t / c4;
    const int j = (int)(cell % cw), i = (int)((cell / cw) % ch), img = (int)(cell / ((int64_t)cw * ch));
    // pixels whose source index lies in (i - 1, i + 1): a margin of one pixel either side, the exact test is inside
    int ylo = 0, yhi = H - 1, xlo = 0, xhi = W - 1;
    if (ry > 0.f) { ylo = max(0, (int)floorf((float)(i - 1) / ry) - 1); yhi = min(H - 1, (int)ceilf((float)(i + 1) / ry) + 1); }
    if (rx > 0.f) { xlo = max(0, (int)floorf((float)(j - 1) / rx) - 1); xhi = min(W - 1, (int)ceilf((float)(j + 1) / rx) + 1); }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int y = ylo; y <= yhi; ++y) {
        int y0, y1;
        float fy;
        l2c_taps(y, ry, ch, y0, y1, fy);
        const float wy = (y0 == i ? 1.f - fy : 0.f) + (y1 == i ? fy : 0.f);
        if (wy == 0.f) continue;
        const float4 *row = g + ((int64_t)img * H + y) * W * c4 + c;
        for (int x = xlo; x <= xhi; ++x) {
            int x0, x1;
            float fx;
            l2c_taps(x, rx, cw, x0, x1, fx);
            const float w = wy * ((x0 == j ? 1.f - fx : 0.f) + (x1 == j ? fx : 0.f));
            if (w == 0.f) continue;
            const float4 v = row[(int64_t)x * c4];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
    }
    d[t] = acc;
}

// ---- MaxPool2d(kernel 3, stride 2, padding 1) of the SwiftNet stem (swiftnet.py: self.maxpool) -------------------------
// torch keeps an int64 index per output (177 MB at 6 x 64 x 180 x 320) and its backward walks the candidate outputs of
// every input pixel comparing those indices: 0.7 ms, at the very end of the step's camera chain.  Here the forward
// stores the position of the maximum inside its window as one byte (row-major over the window clipped to the map, the
// first maximum wins -- torch's scan order and strict `>`, so ties between equal values, e.g. the zeros a ReLU leaves,
// go to the same element), and the backward gathers: every input pixel sums the gradients of the <= 4 windows that
// selected it.
__global__ void __launch_bounds__(256)
maxpool3s2_fwd_kernel(const float *__restrict__ x, int h, int w, int oh, int ow, float *__restrict__ y,
                      uint8_t *__restrict__ code) {
    // grid (pixels of a plane / 256, planes): 32-bit index arithmetic (a 64-bit division per thread cost more than the
    // kernel's memory traffic)
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (uint32_t)(oh * ow)) return;
    const int oy = (int)(pix / (uint32_t)ow), ox = (int)(pix - (uint32_t)oy * (uint32_t)ow);
    const int64_t plane = blockIdx.y;
    const int64_t e = plane * oh * ow + pix;
    const float *xp = x + plane * (int64_t)h * w;
    const int y0 = 2 * oy - 1, x0 = 2 * ox - 1;
    float best = -INFINITY;
    int bc = 0;
    bool first = true;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y0 + dy;
        if (yy < 0 || yy >= h) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = x0 + dx;
            if (xx < 0 || xx >= w) continue;
            const float v = xp[(int64_t)yy * w + xx];
            if (first || v > best || v != v) { best = v; bc = dy * 3 + dx; first = false; }
        }
    }
    y[e] = best;
    code[e] = (uint8_t)bc;
}

__global__ void __launch_bounds__(256)
maxpool3s2_bwd_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ code, int h, int w, int oh, int ow,
                      float *__restrict__ dx) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (uint32_t)(h * w)) return;
    const int yy = (int)(pix / (uint32_t)w), xx = (int)(pix - (uint32_t)yy * (uint32_t)w);
    const int64_t plane = blockIdx.y;
    const int64_t e = plane * h * w + pix;
    const float *gp = dy + plane * (int64_t)oh * ow;
    const uint8_t *cp = code + plane * (int64_t)oh * ow;
    // windows (oy, ox) with 2 oy - 1 <= yy <= 2 oy + 1: oy in {yy / 2, (yy + 1) / 2}
    float g = 0.f;
    const int oy0 = yy >> 1, oy1 = (yy + 1) >> 1, ox0 = xx >> 1, ox1 = (xx + 1) >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = a ? oy1 : oy0;
        if ((a && oy1 == oy0) || oy >= oh) continue;
        const int dyw = yy - (2 * oy - 1);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = b ? ox1 : ox0;
            if ((b && ox1 == ox0) || ox >= ow) continue;
            const int dxw = xx - (2 * ox - 1);
            if (cp[(int64_t)oy * ow + ox] == dyw * 3 + dxw) g += gp[(int64_t)oy * ow + ox];
        }
    }
    dx[e] = g;
}

// the same for 4 consecutive columns per thread (w % 4 == 0): the 2 x 3 windows that can have selected them are read once
// (6 code bytes, 6 gradients, unconditionally) and the four sums leave as one 16-byte store
__global__ void __launch_bounds__(256)
maxpool3s2_bwd4_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ code, int h, int w, int oh, int ow,
                       float *__restrict__ dx) {
    const int w4 = w >> 2;
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (uint32_t)(h * w4)) return;
    const int yy = (int)(q / (uint32_t)w4), x0 = 4 * (int)(q - (uint32_t)yy * (uint32_t)w4);
    const int64_t plane = blockIdx.y;
    const float *gp = dy + plane * (int64_t)oh * ow;
    const uint8_t *cp = code + plane * (int64_t)oh * ow;
    const int oy0 = yy >> 1, oy1 = (yy + 1) >> 1, oxb = x0 >> 1;          // windows oxb .. oxb + 2 touch columns x0 .. x0 + 3
    float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = a ? oy1 : oy0;
        if ((a && oy1 == oy0) || oy >= oh) continue;
        const int dyw = yy - (2 * oy - 1);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ox = oxb + b;
            if (ox >= ow) continue;
            const int sel = (int)cp[(int64_t)oy * ow + ox] - 3 * dyw;     // the selected column of the window's row dyw: 0, 1 or 2
            const float v = gp[(int64_t)oy * ow + ox];
            const int col = 2 * ox - 1 + sel - x0;                         // as an offset into this thread's four columns
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] += (sel >= 0 && sel < 3 && col == i) ? v : 0.f;
        }
    }
    *reinterpret_cast<float4 *>(dx + plane * (int64_t)h * w + (int64_t)yy * w + x0) = make_float4(g[0], g[1], g[2], g[3]);
}

// ---- F.interpolate(mode='bilinear', align_corners=True) of NCHW maps (+ the skip add that follows it in the decoder) ---
// torch's kernels run the decoder's up-samplings at ~0.5 TB/s and their backward scatters with float atomics (run-to-run
// differences in the camera branch's gradients).  Forward: one thread per output element, torch's fp32 index
// arithmetic (src = scale * dst, lambda = src - floor, h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)), optionally
// + skip in the same pass.  Backward: a GATHER per input element over the outputs that read it -- per-axis lists
// (taps[i] = first output, count; weight per (input, tap)) built once per size on the host from the same arithmetic --
// summed in a fixed order: deterministic.
// source index and fraction of one axis: the fraction is of the ROUNDED product scale * dst (torch's kernel; a fused
// multiply-add here moves the up-sampled map by ~1e-5)
__device__ __forceinline__ void up_source(float scale, int dst, int &i0, float &lambda1) {
#pragma clang fp contract(off)
    const float src = scale * (float)dst;
    i0 = (int)src;
    lambda1 = src - (float)i0;
}

__global__ void __launch_bounds__(256)
up_bilinear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ skip, int64_t total, int h, int w, int H, int W,
                       float rh, float rw, float *__restrict__ y) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int ox = (int)(e % W);
    const int64_t r = e / W;
    const int oy = (int)(r % H);
    const int64_t plane = r / H;
    int h1, w1;
    float h1l, w1l;
    up_source(rh, oy, h1, h1l);
    up_source(rw, ox, w1, w1l);
    const int h1p = h1 < h - 1 ? 1 : 0, w1p = w1 < w - 1 ? 1 : 0;
    const float h0l = 1.f - h1l, w0l = 1.f - w1l;
    const float *xp = x + plane * (int64_t)h * w + (int64_t)h1 * w + w1;
    float v = h0l * (w0l * xp[0] + w1l * xp[w1p]) + h1l * (w0l * xp[(int64_t)h1p * w] + w1l * xp[(int64_t)h1p * w + w1p]);
    if (skip) v += skip[e];
    y[e] = v;
}

// dx[plane][i][j] = sum over outputs (oy, ox) that read (i, j) of wy * wx * g[oy][ox]; ty / tx: per input index the first
// output index and the number of outputs (int2), wy / wx: [n_in][kUpTaps] weights (0 beyond the count)
constexpr int kUpTaps = 8;
__global__ void __launch_bounds__(256)
up_bilinear_bwd_kernel(const float *__restrict__ g, int64_t total_in, int h, int w, int H, int W, const int2 *__restrict__ ty,
                       const float *__restrict__ wy, const int2 *__restrict__ tx, const float *__restrict__ wx,
                       float *__restrict__ dx) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total_in) return;
    const int j = (int)(e % w);
    const int64_t r = e / w;
    const int i = (int)(r % h);
    const int64_t plane = r / h;
    const float *gp = g + plane * (int64_t)H * W;
    const int2 ry = ty[i], rx = tx[j];
    float acc = 0.f;
    for (int a = 0; a < ry.y; ++a) {
        const float *row = gp + (int64_t)(ry.x + a) * W + rx.x;
        float t = 0.f;
        for (int b = 0; b < rx.y; ++b) t += wx[j * kUpTaps + b] * row[b];
        acc += wy[i * kUpTaps + a] * t;
    }
    dx[e] = acc;
}

// the same sum with the block's window of g staged in LDS: a tile of kUpTi x kUpTj inputs reads the output rows
// [ty[i_first].x, ty[i_last].x + ty[i_last].y) x columns likewise (monotone lists); the caller has checked that every
// tile's window fits kUpLr x kUpLc.  g is read from HBM once, coalesced; 256 threads = 4 row groups x 64 columns
constexpr int kUpTi = 16, kUpTj = 64, kUpLr = 40, kUpLc = 136;
__global__ void __launch_bounds__(256)
up_bilinear_bwd_tiled_kernel(const float *__restrict__ g, int h, int w, int H, int W, const int2 *__restrict__ ty,
                             const float *__restrict__ wy, const int2 *__restrict__ tx, const float *__restrict__ wx,
                             float *__restrict__ dx) {
    __shared__ float win[kUpLr][kUpLc + 1];
    const int j0 = blockIdx.x * kUpTj, i0 = blockIdx.y * kUpTi;
    const int64_t plane = blockIdx.z;
    const int i1 = min(i0 + kUpTi, h) - 1, j1 = min(j0 + kUpTj, w) - 1;
    const int r_lo = ty[i0].x, r_hi = ty[i1].x + ty[i1].y, c_lo = tx[j0].x, c_hi = tx[j1].x + tx[j1].y;
    const int nr = r_hi - r_lo, nc = c_hi - c_lo;
    const float *gp = g + plane * (int64_t)H * W;
    for (int rr = threadIdx.x / 64; rr < nr; rr += 4)
        for (int cc = threadIdx.x % 64; cc < nc; cc += 64) win[rr][cc] = gp[(int64_t)(r_lo + rr) * W + c_lo + cc];
    __syncthreads();
    const int j = j0 + threadIdx.x % 64;
    if (j > j1) return;
    const int2 rx = tx[j];
    float wxj[kUpTaps];
#pragma unroll
    for (int b = 0; b < kUpTaps; ++b) wxj[b] = wx[j * kUpTaps + b];
    const int cb = rx.x - c_lo;
    for (int i = i0 + threadIdx.x / 64; i <= i1; i += 4) {
        const int2 ry = ty[i];
        float acc = 0.f;
        for (int a = 0; a < ry.y; ++a) {
            const float *row = &win[ry.x - r_lo + a][cb];
            float t = 0.f;
#pragma unroll
            for (int b = 0; b < kUpTaps; ++b)
                if (b < rx.y) t += wxj[b] * row[b];
            acc += wy[i * kUpTaps + a] * t;
        }
        dx[plane * (int64_t)h * w + (int64_t)i * w + j] = acc;
    }
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
    const size_t lds = (size_t)(rows_per_chunk + 2) * w * sizeof(float);
    U2_REQUIRE(lds <= 60 * 1024, "u2mkd_upbn_stats: %d rows of %d floats do not fit the LDS stage", rows_per_chunk + 2, w);
    hipLaunchKernelGGL(upbn_stats_kernel, dim3((unsigned)(n_img * c), (unsigned)chunks), dim3(kPhThreads), lds, as_stream(s), x,
                       c, h, w, a, b, ay, ax, rows_per_chunk, partial);
    return check_launch("u2mkd_upbn_stats");
}

int u2mkd_upbn_dense_grad(const float *x, int32_t n_img, int32_t c, int32_t h, int32_t w, const float *a, const float *b,
                          const float *ay, const float *ax, const float *c0, const float *c1, int32_t rows_per_chunk,
                          float *dx, u2mkd_stream_t s) {
    U2_REQUIRE(x && a && b && ay && ax && c0 && c1 && dx, "u2mkd_upbn_dense_grad: null pointer");
    U2_REQUIRE(n_img > 0 && c > 0 && h > 0 && w > 0 && rows_per_chunk > 0, "u2mkd_upbn_dense_grad: bad shape");
    const int chunks = (int)ceil_div(h, rows_per_chunk);
    const size_t lds = (size_t)(rows_per_chunk + 2) * w * sizeof(float);
    U2_REQUIRE(lds <= 60 * 1024 && chunks <= 65535, "u2mkd_upbn_dense_grad: %d rows of %d floats do not fit the LDS stage",
               rows_per_chunk + 2, w);
    hipLaunchKernelGGL(upbn_dense_grad_kernel, dim3((unsigned)(n_img * c), (unsigned)chunks), dim3(kPhThreads), lds, as_stream(s),
                       x, c, h, w, a, b, ay, ax, c0, c1, rows_per_chunk, dx);
    return check_launch("u2mkd_upbn_dense_grad");
}

int u2mkd_transpose_batched(const float *in, float *out, int32_t batch, int32_t rows, int32_t cols, u2mkd_stream_t s) {
    if (batch == 0 || rows == 0 || cols == 0) return 0;
    U2_REQUIRE(in && out && batch > 0 && rows > 0 && cols > 0, "u2mkd_transpose_batched: bad arguments");
    U2_REQUIRE(batch <= 65535 && ceil_div(rows, kTrTile) <= 65535, "u2mkd_transpose_batched: grid too large");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(cols, kTrTile), (unsigned)ceil_div(rows, kTrTile), (unsigned)batch),
                       dim3(256), 0, as_stream(s), in, out, rows, cols);
    return check_launch("u2mkd_transpose_batched");
}

int u2mkd_transpose_batched_scaled(const float *in, float *out, int32_t batch, int32_t rows, int32_t cols, float scale,
                                   u2mkd_stream_t s) {
    if (batch == 0 || rows == 0 || cols == 0) return 0;
    U2_REQUIRE(in && out && batch > 0 && rows > 0 && cols > 0, "u2mkd_transpose_batched_scaled: bad arguments");
    U2_REQUIRE(batch <= 65535 && ceil_div(rows, kTrTile) <= 65535, "u2mkd_transpose_batched_scaled: grid too large");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(cols, kTrTile), (unsigned)ceil_div(rows, kTrTile), (unsigned)batch),
                       dim3(256), 0, as_stream(s), in, out, rows, cols, scale);
    return check_launch("u2mkd_transpose_batched_scaled");
}

static float l2c_ratio(int n_src, int n_dst) { return n_dst > 1 ? (float)(n_src - 1) / (float)(n_dst - 1) : 0.f; }

int u2mkd_l2c_combine_forward(const float *g0, const float *g1, const float *g2, const float *g3, int32_t n_grids, int32_t n_img,
                              int32_t h, int32_t w, int32_t c, int32_t ch1, int32_t cw1, int32_t ch2, int32_t cw2, int32_t ch3,
                              int32_t cw3, float *out_rows, u2mkd_stream_t s) {
    U2_REQUIRE(n_grids >= 1 && n_grids <= 4 && c > 0 && c % 4 == 0 && n_img > 0 && h > 0 && w > 0,
               "u2mkd_l2c_combine_forward: %d grids of %d channels (1..4 grids, channels a multiple of 4)", n_grids, c);
    U2_REQUIRE(g0 && out_rows && (n_grids < 2 || g1) && (n_grids < 3 || g2) && (n_grids < 4 || g3), "u2mkd_l2c_combine_forward: null pointer");
    L2cGrids G;
    const float *gp[4] = {g0, g1, g2, g3};
    const int chs[4] = {h, ch1, ch2, ch3}, cws[4] = {w, cw1, cw2, cw3};
    for (int i = 0; i < 4; ++i) {
        G.g[i] = gp[i]; G.ch[i] = chs[i]; G.cw[i] = cws[i];
        G.ry[i] = l2c_ratio(chs[i], h); G.rx[i] = l2c_ratio(cws[i], w);
        U2_REQUIRE(i >= n_grids || (chs[i] > 0 && cws[i] > 0 && chs[i] <= h && cws[i] <= w), "u2mkd_l2c_combine_forward: grid %d is %dx%d", i, chs[i], cws[i]);
    }
    G.n = n_grids;
    const int64_t total = (int64_t)n_img * h * w * (c / 4);
    hipLaunchKernelGGL(l2c_combine_rows_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), G, n_img, h, w,
                       c / 4, reinterpret_cast<float4 *>(out_rows));
    return check_launch("u2mkd_l2c_combine_forward");
}

int u2mkd_l2c_combine_backward(const float *g_rows, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t ch, int32_t cw,
                               float *d_grid, u2mkd_stream_t s) {
    U2_REQUIRE(g_rows && d_grid && c > 0 && c % 4 == 0 && n_img > 0 && h > 0 && w > 0 && ch > 0 && cw > 0 && ch <= h && cw <= w,
               "u2mkd_l2c_combine_backward: bad arguments");
    const int64_t total = (int64_t)n_img * ch * cw * (c / 4);
    hipLaunchKernelGGL(l2c_combine_rows_bwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const float4 *>(g_rows), n_img, h, w, c / 4, ch, cw, l2c_ratio(ch, h), l2c_ratio(cw, w),
                       reinterpret_cast<float4 *>(d_grid));
    return check_launch("u2mkd_l2c_combine_backward");
}

int u2mkd_maxpool3s2_forward(const float *x, int64_t planes, int32_t h, int32_t w, float *y, uint8_t *code, u2mkd_stream_t s) {
    if (planes == 0) return 0;
    U2_REQUIRE(x && y && code && planes > 0 && h > 0 && w > 0, "u2mkd_maxpool3s2_forward: bad arguments");
    const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
    U2_REQUIRE(planes <= 65535 && (int64_t)h * w < ((int64_t)1 << 31), "u2mkd_maxpool3s2_forward: at most 65535 planes of < 2^31 pixels");
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3((unsigned)ceil_div((int64_t)oh * ow, 256), (unsigned)planes), dim3(256), 0,
                       as_stream(s), x, h, w, oh, ow, y, code);
    return check_launch("u2mkd_maxpool3s2_forward");
}

int u2mkd_maxpool3s2_backward(const float *dy, const uint8_t *code, int64_t planes, int32_t h, int32_t w, float *dx,
                              u2mkd_stream_t s) {
    if (planes == 0) return 0;
    U2_REQUIRE(dy && code && dx && planes > 0 && h > 0 && w > 0, "u2mkd_maxpool3s2_backward: bad arguments");
    const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
    U2_REQUIRE(planes <= 65535 && (int64_t)h * w < ((int64_t)1 << 31), "u2mkd_maxpool3s2_backward: at most 65535 planes of < 2^31 pixels");
    if (w % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0)
        hipLaunchKernelGGL(maxpool3s2_bwd4_kernel, dim3((unsigned)ceil_div((int64_t)h * (w / 4), 256), (unsigned)planes), dim3(256), 0,
                           as_stream(s), dy, code, h, w, oh, ow, dx);
    else
        hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((unsigned)ceil_div((int64_t)h * w, 256), (unsigned)planes), dim3(256), 0,
                           as_stream(s), dy, code, h, w, oh, ow, dx);
    return check_launch("u2mkd_maxpool3s2_backward");
}

int u2mkd_up_bilinear_forward(const float *x, const float *skip, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W,
                              float rh, float rw, float *y, u2mkd_stream_t s) {
    if (planes == 0) return 0;
    U2_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "u2mkd_up_bilinear_forward: bad arguments");
    const int64_t total = planes * H * W;
    hipLaunchKernelGGL(up_bilinear_fwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), x, skip, total, h,
                       w, H, W, rh, rw, y);
    return check_launch("u2mkd_up_bilinear_forward");
}

int u2mkd_up_bilinear_backward(const float *g, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W, const int32_t *taps_y,
                               const float *wy, const int32_t *taps_x, const float *wx, int32_t tiled, float *dx, u2mkd_stream_t s) {
    if (planes == 0) return 0;
    U2_REQUIRE(g && taps_y && wy && taps_x && wx && dx && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0,
               "u2mkd_up_bilinear_backward: bad arguments");
    U2_REQUIRE(!tiled || planes <= 65535, "u2mkd_up_bilinear_backward: the tiled form takes at most 65535 planes");
    const int64_t total = planes * h * w;
    if (tiled) {
        hipLaunchKernelGGL(up_bilinear_bwd_tiled_kernel, dim3((unsigned)ceil_div(w, kUpTj), (unsigned)ceil_div(h, kUpTi), (unsigned)planes),
                           dim3(256), 0, as_stream(s), g, h, w, H, W, reinterpret_cast<const int2 *>(taps_y), wy,
                           reinterpret_cast<const int2 *>(taps_x), wx, dx);
        return check_launch("u2mkd_up_bilinear_backward");
    }
    hipLaunchKernelGGL(up_bilinear_bwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), g, total, h, w, H,
                       W, reinterpret_cast<const int2 *>(taps_y), wy, reinterpret_cast<const int2 *>(taps_x), wx, dx);
    return check_launch("u2mkd_up_bilinear_backward");
}

}  // extern "C"
