// torchsparse v1.4.0 backend-format entry points: convolution_forward / convolution_backward on the
// RULEBOOK a v1.4.0 kernel map holds -- neighbor_map int32 [P,2] rows (in, out) grouped by kernel
// offset, neighbor_offset int32 [K] pairs per offset ON THE HOST (torchsparse keeps it on the CPU),
// transpose -- exactly the arguments of torchsparse.backend.convolution_forward_cuda /
// convolution_backward_cuda (SURVEY.md §8b last row, Appendix A-6).  A maintainer who already holds such
// a kmap (e.g. the reference's own F.conv3d python layer, core/models/build_blocks.py:25-80) calls these
// instead of the native neighbour-table entries.
//
// The rulebook is turned into the PAIR SCHEDULE of conv.hip (every offset's group padded to 64 slots,
// slot tables per output row) by one scatter kernel -- the group starts are host-known because
// neighbor_offset is a host array -- then the dense 64-pair MFMA tiles + the ordered gather-sum run
// (deterministic; torchsparse: K x (gather, cuBLAS mm, scatter-add)).
#include "common.h"

namespace u2mkd {

struct Groups {
    int32_t start[65];    // rulebook row of the first pair of offset k (start[K] = P)
    int32_t pstart[65];   // first padded slot of offset k
};

// slot tables + padded gather list from the rulebook.  gather_col: which rulebook column is gathered
// (0 = in: forward;  1 = out: transposed conv / input gradient), the other one owns the slot table row.
__global__ void rulebook_scatter_kernel(const int32_t *__restrict__ nbmaps, Groups g, int K, int gather_col,
                                        int32_t *__restrict__ pair_idx, int32_t *__restrict__ pos,
                                        int32_t *__restrict__ tile_k, int32_t *__restrict__ meta, int cap) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) {
        meta[0] = cap;
        meta[1] = cap / 64;
    }
    if (p >= g.start[K]) return;
    int k = 0;
    while (p >= g.start[k + 1]) ++k;
    const int slot = g.pstart[k] + (p - g.start[k]);
    const int a = nbmaps[2 * (size_t)p + gather_col], b = nbmaps[2 * (size_t)p + 1 - gather_col];
    pair_idx[slot] = a;
    pos[(size_t)b * K + k] = slot;
    if ((slot & 63) == 0) tile_k[slot >> 6] = k;
}

__global__ void fill_i32_kernel(int32_t *__restrict__ p, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct Layout {
    int64_t cap;
    size_t off_pair, off_pos, off_tile, off_meta, off_wt, off_y, off_sizes, off_plan, off_wgrad, total;
};

// worst case over the host-unknown pair count: every offset full (P <= k * min(rows))
static Layout layout_for(int64_t cap, int64_t n_scatter_rows, int32_t cin, int32_t cout, int32_t k, size_t wgrad_bytes) {
    Layout l;
    l.cap = cap;
    size_t o = 0;
    l.off_pair = o;  o += align256((size_t)cap * 4);
    l.off_pos = o;   o += align256((size_t)n_scatter_rows * k * 4);
    l.off_tile = o;  o += align256((size_t)(cap / 64 + 1) * 4);
    l.off_meta = o;  o += 256;
    l.off_wt = o;    o += align256((size_t)k * cin * cout * 4);
    l.off_y = o;     o += align256((size_t)cap * (cin > cout ? cin : cout) * 4);
    l.off_sizes = o; o += align256((size_t)k * 4);
    l.off_plan = o;  o += align256((size_t)u2mkd_wgrad_plan_ints(k) * 4);
    l.off_wgrad = o; o += align256(wgrad_bytes);
    l.total = o;
    return l;
}

static int64_t padded_capacity(const int32_t *nbsizes_host, int32_t k, Groups *g) {
    int64_t p = 0, s = 0;
    for (int i = 0; i < k; ++i) {
        if (g) { g->start[i] = (int32_t)p; g->pstart[i] = (int32_t)s; }
        p += nbsizes_host[i];
        s += ((int64_t)nbsizes_host[i] + 63) / 64 * 64;
    }
    if (g) { g->start[k] = (int32_t)p; g->pstart[k] = (int32_t)s; }
    return s;
}

// out[scatter rows] = sum over the rulebook of in[gather rows] * B_k, B_k = wt_src[k] as [ncol][nred]
static int run_pairs(const char *who, const float *in, int64_t n_gather_rows, int32_t nred, const float *bmat, int32_t ncol,
                     const int32_t *nbmaps, const Groups &g, int32_t k, int gather_col, int64_t n_scatter_rows,
                     char *ws, const Layout &l, float *out, u2mkd_stream_t s) {
    hipStream_t st = as_stream(s);
    int32_t *pair_idx = reinterpret_cast<int32_t *>(ws + l.off_pair), *pos = reinterpret_cast<int32_t *>(ws + l.off_pos);
    int32_t *tile_k = reinterpret_cast<int32_t *>(ws + l.off_tile), *meta = reinterpret_cast<int32_t *>(ws + l.off_meta);
    float *y = reinterpret_cast<float *>(ws + l.off_y);
    const int64_t npos = n_scatter_rows * k;
    if (l.cap == 0) {   // empty rulebook: the output is zero
        (void)hipMemsetAsync(out, 0, (size_t)n_scatter_rows * ncol * sizeof(float), st);
        return check_launch(who);
    }
    hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(l.cap, 256)), dim3(256), 0, st, pair_idx, l.cap, -1);
    hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(npos, 256)), dim3(256), 0, st, pos, npos, -1);
    const int P = g.start[k];
    hipLaunchKernelGGL(rulebook_scatter_kernel, dim3((unsigned)ceil_div(P > 0 ? P : 1, 256)), dim3(256), 0, st, nbmaps, g, k,
                       gather_col, pair_idx, pos, tile_k, meta, (int)l.cap);
    if (int rc = check_launch(who)) return rc;
    if (int rc = u2mkd_conv_forward_pairs(in, n_gather_rows, nred, bmat, ncol, pair_idx, tile_k, meta, l.cap, k, 0, y, s)) return rc;
    return u2mkd_pairs_gather_sum(y, pos, n_scatter_rows, k, ncol, out, s);
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

size_t u2mkd_convolution_workspace_bytes(int64_t n_in_rows, int64_t n_out_rows, int32_t cin, int32_t cout,
                                         const int32_t *nbsizes_host, int32_t k) {
    if (!nbsizes_host || k <= 0 || k > 64) return 0;
    const int64_t cap = padded_capacity(nbsizes_host, k, nullptr);
    const int64_t rows = n_in_rows > n_out_rows ? n_in_rows : n_out_rows;
    return layout_for(cap, rows, cin, cout, k, u2mkd_conv_wgrad_pairs_workspace_bytes(rows, cin, cout, k)).total;
}

int u2mkd_convolution_forward(const float *in_feat, int64_t n_in_rows, int32_t cin, float *out_feat, int64_t n_out_rows,
                              int32_t cout, const float *kernel, const int32_t *nbmaps, const int32_t *nbsizes_host,
                              int32_t k, int32_t transpose, void *workspace, size_t workspace_bytes, u2mkd_stream_t s) {
    U2_REQUIRE(in_feat && out_feat && kernel && nbsizes_host && workspace, "u2mkd_convolution_forward: null pointer");
    U2_REQUIRE(k > 0 && k <= 64, "u2mkd_convolution_forward: kernel volume %d not in 1..64", k);
    U2_REQUIRE(cin > 0 && cout > 0 && cin % 4 == 0 && cout % 4 == 0,
               "u2mkd_convolution_forward: cin=%d cout=%d must be positive multiples of 4", cin, cout);
    if (n_out_rows == 0) return 0;
    Groups g;
    const int64_t cap = padded_capacity(nbsizes_host, k, &g);
    U2_REQUIRE(g.start[k] == 0 || nbmaps, "u2mkd_convolution_forward: null neighbor_map");
    const int64_t rows = n_in_rows > n_out_rows ? n_in_rows : n_out_rows;
    const Layout l = layout_for(cap, rows, cin, cout, k, u2mkd_conv_wgrad_pairs_workspace_bytes(rows, cin, cout, k));
    U2_REQUIRE(workspace_bytes >= l.total, "u2mkd_convolution_forward: workspace %zu < %zu bytes", workspace_bytes, l.total);
    char *ws = reinterpret_cast<char *>(workspace);
    float *wt = reinterpret_cast<float *>(ws + l.off_wt);
    // B_k[col][ci] = kernel[k][ci][col]
    if (int rc = u2mkd_transpose_weights(kernel, k, cin, cout, wt, s)) return rc;
    // transpose = 0: out[out_idx] += in[in_idx] W_k;  transpose = 1 (v1.4.0): out[in_idx] += in[out_idx] W_k
    return run_pairs("u2mkd_convolution_forward", in_feat, n_in_rows, cin, wt, cout, nbmaps, g, k, transpose ? 1 : 0,
                     n_out_rows, ws, l, out_feat, s);
}

int u2mkd_convolution_backward(const float *in_feat, int64_t n_in_rows, int32_t cin, float *grad_in_feat,
                               const float *grad_out_feat, int64_t n_out_rows, int32_t cout, const float *kernel,
                               float *grad_kernel, const int32_t *nbmaps, const int32_t *nbsizes_host, int32_t k,
                               int32_t transpose, void *workspace, size_t workspace_bytes, u2mkd_stream_t s) {
    U2_REQUIRE(in_feat && grad_out_feat && kernel && nbsizes_host && workspace,
               "u2mkd_convolution_backward: null pointer");
    U2_REQUIRE(k > 0 && k <= 64, "u2mkd_convolution_backward: kernel volume %d not in 1..64", k);
    U2_REQUIRE(cin > 0 && cout > 0 && cin % 4 == 0 && cout % 4 == 0,
               "u2mkd_convolution_backward: cin=%d cout=%d must be positive multiples of 4", cin, cout);
    Groups g;
    const int64_t cap = padded_capacity(nbsizes_host, k, &g);
    U2_REQUIRE(g.start[k] == 0 || nbmaps, "u2mkd_convolution_backward: null neighbor_map");
    const int64_t rows = n_in_rows > n_out_rows ? n_in_rows : n_out_rows;
    const size_t wg_bytes = u2mkd_conv_wgrad_pairs_workspace_bytes(rows, cin, cout, k);
    const Layout l = layout_for(cap, rows, cin, cout, k, wg_bytes);
    U2_REQUIRE(workspace_bytes >= l.total, "u2mkd_convolution_backward: workspace %zu < %zu bytes", workspace_bytes, l.total);
    char *ws = reinterpret_cast<char *>(workspace);
    if (grad_in_feat && n_in_rows > 0) {
        // dX[in_idx] += dY[out_idx] W_k^T: gather grad_out rows, B_k[ci][co] = kernel[k][ci][co] as it is
        // (transposed conv: dX[out_idx] += dY[in_idx] W_k^T)
        if (int rc = run_pairs("u2mkd_convolution_backward", grad_out_feat, n_out_rows, cout, kernel, cin, nbmaps, g, k,
                               transpose ? 0 : 1, n_in_rows, ws, l, grad_in_feat, s))
            return rc;
    }
    if (grad_kernel) {
        int32_t *sizes = reinterpret_cast<int32_t *>(ws + l.off_sizes), *plan = reinterpret_cast<int32_t *>(ws + l.off_plan);
        hipError_t e = hipMemcpyAsync(sizes, nbsizes_host, (size_t)k * sizeof(int32_t), hipMemcpyHostToDevice, as_stream(s));
        U2_REQUIRE(e == hipSuccess, "u2mkd_convolution_backward: copy of neighbor_offset: %s", hipGetErrorString(e));
        if (int rc = u2mkd_wgrad_plan(sizes, k, rows, plan, s)) return rc;
        // dW[k] = sum X[in_idx]^T dY[out_idx]  (transposed: X[out_idx]^T dY[in_idx])
        return u2mkd_conv_wgrad_pairs(in_feat, cin, grad_out_feat, cout, nbmaps, plan, rows, k, transpose ? 1 : 0,
                                      ws + l.off_wgrad, wg_bytes, grad_kernel, s);
    }
    return 0;
}

}  // extern "C"
