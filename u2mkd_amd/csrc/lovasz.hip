// Lovasz-softmax (classes = 'present') around its sort: the element-wise chains of core/criterions.py:73-101 as three kernels.
//
// The reference evaluates, per class c, errors = |fg_c - p_c| over the valid points, sorts them descending, and dots them with
// the discrete gradient of the Jaccard index of the sorted foreground flags (lovasz_grad, criterions.py:40-52); the loss is the
// mean over the classes that occur.  u2mkd_amd/losses.py evaluates all classes at once on the [C, P] transpose with ONE radix
// sort of the composite key 4 c - error (float64: exact) -- but around that sort it queued ~45 element-wise torch launches per
// call in the forward and ~18 in the backward, twice per KD step, right between the forward and the backward of the critical
// stream.  Here:
//   lovasz_errors_kernel   probabilities [P, C] + labels -> errors [C, P] (-1 on ignored rows), sort keys [C, P] f64
//   (torch.sort of the keys: rocPRIM's radix sort, value-dependent, stays in torch)
//   lovasz_gather_kernel   sorted positions -> foreground flag of every sorted entry (int32, for the prefix sum)
//   (one flat cumsum: exact integers)
//   lovasz_terms_kernel    prefix sums -> Jaccard gradient of every sorted entry (kept for the backward) and the per-class sums
//                          of max(error, 0) * gradient, partial per workgroup, then
//   lovasz_finish_kernel   fixed-order sums per class, mean over the present classes (and the factor the backward needs)
//   lovasz_backward_kernel d loss / d probability of every (point, class): one pass, no atomics (a permutation per class)
// Same arithmetic per element as the torch formulation (1 - intersection / union in fp32 on exact integers, the difference of
// neighbours); only the order of the final sums differs.  Deterministic.
#include "common.h"

namespace u2mkd {

constexpr int kLvThreads = 256;

__global__ void lovasz_errors_kernel(const float *__restrict__ probas, const int64_t *__restrict__ labels, int ignore,
                                     int64_t P, int C, float *__restrict__ errors, double *__restrict__ keys) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int64_t lab = labels[p];
    const bool valid = lab != ignore;
    for (int c = 0; c < C; ++c) {
        const float fg = (valid && lab == c) ? 1.f : 0.f;
        const float e = valid ? fabsf(fg - probas[p * C + c]) : -1.f;
        errors[(int64_t)c * P + p] = e;
        keys[(int64_t)c * P + p] = (double)(4 * c) - (double)e;
    }
}

// perm[g] = position (c * P + p) of the g-th smallest key: block c of the result holds class c's points, errors descending
__global__ void lovasz_gather_kernel(const int64_t *__restrict__ perm, const int64_t *__restrict__ labels, int ignore,
                                     int64_t P, int C, int32_t *__restrict__ fg_sorted) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P * C) return;
    const int64_t pos = perm[g];
    const int c = (int)(g / P);
    const int64_t p = pos - (int64_t)c * P;
    const int64_t lab = labels[p];
    fg_sorted[g] = (lab != ignore && lab == c) ? 1 : 0;
}

// csum = inclusive prefix sums of fg_sorted over the whole [C * P] array (int64).  Sorted entry (c, i):
//   cs = csum[c P + i] - start_c, gts = csum[c P + P - 1] - start_c, start_c = csum[c P - 1] (0 for c = 0)
//   jaccard(i) = 1 - (gts - cs) / (gts + (i + 1) - cs);  grad(i) = jaccard(i) - jaccard(i - 1), grad(0) = jaccard(0)
__global__ void __launch_bounds__(kLvThreads)
lovasz_terms_kernel(const int64_t *__restrict__ perm, const float *__restrict__ errors, const int64_t *__restrict__ csum,
                    const int32_t *__restrict__ fg_sorted, int64_t P, int C, float *__restrict__ jgrad,
                    float *__restrict__ partial /*[C][gridDim.x]*/) {
    __shared__ float red[kLvThreads];
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float term = 0.f;
    if (i < P) {
        const int64_t g = (int64_t)c * P + i;
        const int64_t start = c ? csum[(int64_t)c * P - 1] : 0;
        const float gts = (float)(csum[(int64_t)c * P + P - 1] - start);
        const float cs = (float)(csum[g] - start);
        const float jac = 1.f - (gts - cs) / (gts + ((float)(i + 1) - cs));
        float grad = jac;
        if (i > 0) {
            const float cs1 = cs - (float)fg_sorted[g];
            grad = jac - (1.f - (gts - cs1) / (gts + ((float)i - cs1)));
        }
        jgrad[g] = grad;
        const float e = errors[perm[g]];
        term = fmaxf(e, 0.f) * grad;
    }
    red[threadIdx.x] = term;
    __syncthreads();
    for (int s = kLvThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.x + blockIdx.x] = red[0];
}

// out[0] = loss, out[1] = 1 / max(#present, 1), out[2 + c] = 1 if class c occurs among the valid points else 0
__global__ void __launch_bounds__(64)
lovasz_finish_kernel(const float *__restrict__ partial, int nblk, const int64_t *__restrict__ csum, int64_t P, int C,
                     float *__restrict__ out) {
    __shared__ float s_loss[64], s_present[64];
    float loss = 0.f, present = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) {
        float t = 0.f;
        for (int b = 0; b < nblk; ++b) t += partial[(size_t)c * nblk + b];
        const int64_t start = c ? csum[(int64_t)c * P - 1] : 0;
        const float pr = (csum[(int64_t)c * P + P - 1] - start) > 0 ? 1.f : 0.f;
        out[2 + c] = pr;
        loss += t * pr;
        present += pr;
    }
    s_loss[threadIdx.x] = loss;
    s_present[threadIdx.x] = present;
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = 0.f, n = 0.f;
        for (int t = 0; t < 64; ++t) { l += s_loss[t]; n += s_present[t]; }
        const float inv = 1.f / fmaxf(n, 1.f);
        out[0] = l * inv;
        out[1] = inv;
    }
}

// d loss / d probas[p, c] = g_out * present_c / #present * grad(i) * [error >= 0] * d|fg - p| / dp, (c, i) the sorted entry at p
__global__ void lovasz_backward_kernel(const float *__restrict__ g_out, const float *__restrict__ stats,
                                       const int64_t *__restrict__ perm, const float *__restrict__ jgrad,
                                       const float *__restrict__ probas, const int64_t *__restrict__ labels, int ignore,
                                       int64_t P, int C, float *__restrict__ d_probas) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P * C) return;
    const int c = (int)(g / P);
    const int64_t p = perm[g] - (int64_t)c * P;
    const int64_t lab = labels[p];
    float d = 0.f;
    if (lab != ignore) {
        const float fg = lab == c ? 1.f : 0.f;
        const float diff = fg - probas[p * C + c];
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        d = -sgn * jgrad[g] * (g_out[0] * stats[1] * stats[2 + c]);
    }
    d_probas[p * C + c] = d;
}

// ---- cross entropy with ignore_index, mean over the valid rows (nn.CrossEntropyLoss, core/criterions.py:167-174) -----------------
// torch runs log_softmax + nll_loss as four launches whose two reductions are single-workgroup kernels: 57 + 76 us per call at
// 80 000 x 17, twice per KD step, on the critical stream between the forward and the backward.  Here: one pass per direction.
//   ce_forward_kernel   per row lse = max + log sum exp(x - max), loss_row = lse - x[label] (0 on ignored rows); per-workgroup
//                       (sum, count) partials, fixed-order finish by the last launch -> stats = {mean loss, 1 / count}
//   ce_backward_kernel  dx = g / count * (exp(x - lse) - onehot) on valid rows, 0 on ignored ones
constexpr int kCeThreads = 256;

__global__ void __launch_bounds__(kCeThreads)
ce_forward_kernel(const float *__restrict__ x, const int64_t *__restrict__ labels, int ignore, int64_t P, int C,
                  float *__restrict__ lse, float *__restrict__ partial /*[grid][2]*/) {
    const int64_t p = (int64_t)blockIdx.x * kCeThreads + threadIdx.x;
    float loss = 0.f, cnt = 0.f;
    if (p < P) {
        const float *row = x + p * C;
        float m = row[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(row[c] - m);
        const float l = m + logf(s);
        lse[p] = l;
        const int64_t lab = labels[p];
        if (lab != ignore && lab >= 0 && lab < C) { loss = l - row[lab]; cnt = 1.f; }
    }
    __shared__ float s_l[kCeThreads / 64], s_c[kCeThreads / 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { loss += __shfl_xor(loss, off); cnt += __shfl_xor(cnt, off); }
    if ((threadIdx.x & 63) == 0) { s_l[threadIdx.x >> 6] = loss; s_c[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
        partial[2 * blockIdx.x + 1] = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
    }
}

__global__ void __launch_bounds__(256)
ce_finish_kernel(const float *__restrict__ partial, int n, float *__restrict__ stats) {
    __shared__ double s_l[256], s_c[256];
    double l = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { l += partial[2 * i]; c += partial[2 * i + 1]; }
    s_l[threadIdx.x] = l; s_c[threadIdx.x] = c;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) { s_l[threadIdx.x] += s_l[threadIdx.x + st]; s_c[threadIdx.x] += s_c[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (no valid row: torch returns nan = 0 / 0)
        stats[0] = (float)(s_l[0] / s_c[0]);
        stats[1] = s_c[0] > 0.0 ? (float)(1.0 / s_c[0]) : 0.f;
    }
}

__global__ void __launch_bounds__(kCeThreads)
ce_backward_kernel(const float *__restrict__ g, const float *__restrict__ stats, const float *__restrict__ x,
                   const float *__restrict__ lse, const int64_t *__restrict__ labels, int ignore, int64_t P, int C,
                   float *__restrict__ dx) {
    const int64_t t = (int64_t)blockIdx.x * kCeThreads + threadIdx.x;
    if (t >= P * C) return;
    const int64_t p = t / C;
    const int c = (int)(t - p * C);
    const int64_t lab = labels[p];
    float d = 0.f;
    if (lab != ignore && lab >= 0 && lab < C) d = (g[0] * stats[1]) * (expf(x[t] - lse[p]) - (c == lab ? 1.f : 0.f));
    dx[t] = d;
}

// ---- KL divergence of two logit matrices, reduction 'batchmean' (nn.KLDivLoss on log_softmax(student) / softmax(teacher),
// core/nusc_trainers.py:330-336) ---------------------------------------------------------------------------------------------
// As torch operations: an index_select of the teacher's rows, log_softmax, softmax, kl_div's point-wise kernel and a sum (then
// five more in the backward), all on the critical stream between the forward and the backward.  One pass per direction:
//   kl_forward_kernel   per row lse_s, lse_t, sum p_t (saved) and sum_c p_t ((t_c - lse_t) - (s_c - lse_s)); the teacher's row
//                       is t[idx[p]] when an index is given; per-workgroup partials, fixed-order finish -> stats[0] = sum / P
//   kl_backward_kernel  ds = g / P * (exp(s - lse_s) * sum p_t - p_t): kl_div's and log_softmax's backward, composed
__global__ void __launch_bounds__(kCeThreads)
kl_forward_kernel(const float *__restrict__ s, const float *__restrict__ t, const int64_t *__restrict__ idx, int64_t P, int C,
                  float *__restrict__ rows /*[P][3]*/, float *__restrict__ partial /*[grid][2]*/) {
    const int64_t p = (int64_t)blockIdx.x * kCeThreads + threadIdx.x;
    float loss = 0.f;
    if (p < P) {
        const float *sr = s + p * C, *tr = t + (idx ? idx[p] : p) * C;
        float ms = sr[0], mt = tr[0];
        for (int c = 1; c < C; ++c) { ms = fmaxf(ms, sr[c]); mt = fmaxf(mt, tr[c]); }
        float es = 0.f, et = 0.f;
        for (int c = 0; c < C; ++c) { es += expf(sr[c] - ms); et += expf(tr[c] - mt); }
        const float ls = ms + logf(es), lt = mt + logf(et);
        float sp = 0.f;
        for (int c = 0; c < C; ++c) {
            const float lpt = tr[c] - lt, pt = expf(lpt);
            sp += pt;
            if (pt > 0.f) loss += pt * (lpt - (sr[c] - ls));      // (xlogy: 0 where the target is 0)
        }
        rows[3 * p] = ls; rows[3 * p + 1] = lt; rows[3 * p + 2] = sp;
    }
    __shared__ float s_l[kCeThreads / 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) loss += __shfl_xor(loss, off);
    if ((threadIdx.x & 63) == 0) s_l[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
        partial[2 * blockIdx.x + 1] = 0.f;
    }
}

__global__ void __launch_bounds__(256)
kl_finish_kernel(const float *__restrict__ partial, int n, float inv_rows, float *__restrict__ stats) {
    __shared__ double s_l[256];
    double l = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) l += partial[2 * i];
    s_l[threadIdx.x] = l;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s_l[threadIdx.x] += s_l[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) stats[0] = (float)(s_l[0] * (double)inv_rows);
}

__global__ void __launch_bounds__(kCeThreads)
kl_backward_kernel(const float *__restrict__ g, const float *__restrict__ s, const float *__restrict__ t,
                   const int64_t *__restrict__ idx, const float *__restrict__ rows, int64_t P, int C, float inv_rows,
                   float *__restrict__ ds) {
    const int64_t e = (int64_t)blockIdx.x * kCeThreads + threadIdx.x;
    if (e >= P * C) return;
    const int64_t p = e / C;
    const int c = (int)(e - p * C);
    const float ls = rows[3 * p], lt = rows[3 * p + 1], sp = rows[3 * p + 2];
    const float pt = expf(t[(idx ? idx[p] : p) * C + c] - lt);
    ds[e] = (g[0] * inv_rows) * (expf(s[e] - ls) * sp - pt);
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

int u2mkd_lovasz_errors(const float *probas, const int64_t *labels, int32_t ignore_index, int64_t n, int32_t c, float *errors,
                        double *keys, u2mkd_stream_t s) {
    U2_REQUIRE(n >= 0 && c > 0 && c <= 4096, "u2mkd_lovasz_errors: n=%lld classes=%d", (long long)n, c);
    if (n == 0) return 0;
    U2_REQUIRE(probas && labels && errors && keys, "u2mkd_lovasz_errors: null pointer");
    hipLaunchKernelGGL(lovasz_errors_kernel, dim3((unsigned)ceil_div(n, kLvThreads)), dim3(kLvThreads), 0, as_stream(s), probas,
                       labels, ignore_index, n, c, errors, keys);
    return check_launch("u2mkd_lovasz_errors");
}

int u2mkd_lovasz_gather(const int64_t *perm, const int64_t *labels, int32_t ignore_index, int64_t n, int32_t c,
                        int32_t *fg_sorted, u2mkd_stream_t s) {
    U2_REQUIRE(n >= 0 && c > 0, "u2mkd_lovasz_gather: n=%lld classes=%d", (long long)n, c);
    if (n == 0) return 0;
    U2_REQUIRE(perm && labels && fg_sorted, "u2mkd_lovasz_gather: null pointer");
    hipLaunchKernelGGL(lovasz_gather_kernel, dim3((unsigned)ceil_div(n * c, kLvThreads)), dim3(kLvThreads), 0, as_stream(s), perm,
                       labels, ignore_index, n, c, fg_sorted);
    return check_launch("u2mkd_lovasz_gather");
}

int64_t u2mkd_lovasz_partials(int64_t n, int32_t c) { return (int64_t)c * ceil_div(n > 0 ? n : 1, kLvThreads); }

int u2mkd_lovasz_terms(const int64_t *perm, const float *errors, const int64_t *csum, const int32_t *fg_sorted, int64_t n,
                       int32_t c, float *jgrad, float *partial, float *stats, u2mkd_stream_t s) {
    U2_REQUIRE(n > 0 && c > 0 && c <= 65535, "u2mkd_lovasz_terms: n=%lld classes=%d", (long long)n, c);
    U2_REQUIRE(perm && errors && csum && fg_sorted && jgrad && partial && stats, "u2mkd_lovasz_terms: null pointer");
    const unsigned nblk = (unsigned)ceil_div(n, kLvThreads);
    hipLaunchKernelGGL(lovasz_terms_kernel, dim3(nblk, (unsigned)c), dim3(kLvThreads), 0, as_stream(s), perm, errors, csum,
                       fg_sorted, n, c, jgrad, partial);
    hipLaunchKernelGGL(lovasz_finish_kernel, dim3(1), dim3(64), 0, as_stream(s), partial, (int)nblk, csum, n, c, stats);
    return check_launch("u2mkd_lovasz_terms");
}

int u2mkd_lovasz_backward(const float *g_out, const float *stats, const int64_t *perm, const float *jgrad, const float *probas,
                          const int64_t *labels, int32_t ignore_index, int64_t n, int32_t c, float *d_probas, u2mkd_stream_t s) {
    U2_REQUIRE(n >= 0 && c > 0, "u2mkd_lovasz_backward: n=%lld classes=%d", (long long)n, c);
    if (n == 0) return 0;
    U2_REQUIRE(g_out && stats && perm && jgrad && probas && labels && d_probas, "u2mkd_lovasz_backward: null pointer");
    hipLaunchKernelGGL(lovasz_backward_kernel, dim3((unsigned)ceil_div(n * c, kLvThreads)), dim3(kLvThreads), 0, as_stream(s),
                       g_out, stats, perm, jgrad, probas, labels, ignore_index, n, c, d_probas);
    return check_launch("u2mkd_lovasz_backward");
}

int64_t u2mkd_ce_partials(int64_t n) { return 2 * ceil_div(n, kCeThreads); }

int u2mkd_ce_forward(const float *x, const int64_t *labels, int32_t ignore_index, int64_t n, int32_t c, float *lse, float *partial,
                     float *stats, u2mkd_stream_t s) {
    U2_REQUIRE(n > 0 && c > 0 && x && labels && lse && partial && stats, "u2mkd_ce_forward: bad arguments");
    const int grid = (int)ceil_div(n, kCeThreads);
    hipLaunchKernelGGL(ce_forward_kernel, dim3(grid), dim3(kCeThreads), 0, as_stream(s), x, labels, ignore_index, n, c, lse, partial);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, as_stream(s), partial, grid, stats);
    return check_launch("u2mkd_ce_forward");
}

int u2mkd_kl_forward(const float *s_logits, const float *t_logits, const int64_t *t_index, int64_t n, int32_t c, float *rows,
                     float *partial, float *stats, u2mkd_stream_t s) {
    U2_REQUIRE(n > 0 && c > 0 && s_logits && t_logits && rows && partial && stats, "u2mkd_kl_forward: bad arguments");
    const int grid = (int)ceil_div(n, kCeThreads);
    hipLaunchKernelGGL(kl_forward_kernel, dim3(grid), dim3(kCeThreads), 0, as_stream(s), s_logits, t_logits, t_index, n, c, rows, partial);
    hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(256), 0, as_stream(s), partial, grid, 1.f / (float)n, stats);
    return check_launch("u2mkd_kl_forward");
}

int u2mkd_kl_backward(const float *g_out, const float *s_logits, const float *t_logits, const int64_t *t_index, const float *rows,
                      int64_t n, int32_t c, float *ds, u2mkd_stream_t s) {
    U2_REQUIRE(n > 0 && c > 0 && g_out && s_logits && t_logits && rows && ds, "u2mkd_kl_backward: bad arguments");
    hipLaunchKernelGGL(kl_backward_kernel, dim3((unsigned)ceil_div(n * c, kCeThreads)), dim3(kCeThreads), 0, as_stream(s), g_out,
                       s_logits, t_logits, t_index, rows, n, c, 1.f / (float)n, ds);
    return check_launch("u2mkd_kl_backward");
}

int u2mkd_ce_backward(const float *g_out, const float *stats, const float *x, const float *lse, const int64_t *labels,
                      int32_t ignore_index, int64_t n, int32_t c, float *dx, u2mkd_stream_t s) {
    U2_REQUIRE(n > 0 && c > 0 && g_out && stats && x && lse && labels && dx, "u2mkd_ce_backward: bad arguments");
    hipLaunchKernelGGL(ce_backward_kernel, dim3((unsigned)ceil_div(n * c, kCeThreads)), dim3(kCeThreads), 0, as_stream(s), g_out, stats, x,
                       lse, labels, ignore_index, n, c, dx);
    return check_launch("u2mkd_ce_backward");
}

}  // extern "C"
