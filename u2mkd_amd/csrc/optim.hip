// SGD (momentum, Nesterov, weight decay) over ALL parameters of a group in ONE launch.
//
// The reference trains with torch.optim.SGD(nesterov=True) (core/builder.py:663-669, configs/nuscenes/default.yaml:18-23).
// On the device torch runs it as its multi-tensor ("foreach") path: five passes over parameters / gradients / momentum
// buffers in ~40 launches per step behind ~3.5 ms of host-side list handling (torch/optim/sgd.py:_multi_tensor_sgd) -- in a
// step that is bound by the host (NOTES N10).  Here: a job table (parameter, gradient, momentum buffer, elements, first-use
// flag per tensor) + one kernel that applies, element by element, the SAME five operations in the same order and with the same
// roundings as the foreach kernels (each `a + alpha * b` of ATen's BinaryOpListAlphaFunctor is one fused multiply-add; `b *
// momentum` and `b + g` round separately):
//     g1 = fma(wd, p, g)                      torch._foreach_add(grads, params, alpha=weight_decay)
//     b  = first ? g1 : (b * mom) + g1        clone(grad)  |  _foreach_mul_(bufs, momentum); _foreach_add_(bufs, grads, alpha=1)
//     g2 = nesterov ? fma(mom, b, g1) : b     _foreach_add_(grads, bufs, alpha=momentum)
//     p  = fma(-lr, g2, p)                    _foreach_add_(params, grads, alpha=-lr)
// bit-identical to torch's result (tests/test_gpu_optim.py), one read of p / g / b and one write of p / b per element.
#include "common.h"

namespace u2mkd {

constexpr int kSgdChunk = 4096;          // elements per workgroup (256 threads x 4 x float4)

struct SgdJob {                          // one row of the job table (int64 x 6)
    float *p;
    const float *g;                      // nullptr: this parameter received no gradient (skipped, as torch does)
    float *b;
    int64_t numel;
    int64_t first_chunk;                 // index of this tensor's first chunk in the launch
    int64_t first;                       // 1: no momentum buffer yet (torch: buf = clone(grad))
};

template <bool FMA>
__device__ __forceinline__ void sgd_one(float &p, float g, float &b, bool first, float lr_neg, float mom, float wd,
                                        bool use_wd, bool use_mom, bool nesterov) {
#pragma clang fp contract(off)
    float g1 = g;
    if (use_wd) g1 = FMA ? __builtin_fmaf(wd, p, g) : g + wd * p;
    float g2 = g1;
    if (use_mom) {
        float nb;
        if (first) {
            nb = g1;
        } else {
            nb = b * mom;
            nb = nb + g1;
        }
        b = nb;
        g2 = nesterov ? (FMA ? __builtin_fmaf(mom, nb, g1) : g1 + mom * nb) : nb;
    }
    p = FMA ? __builtin_fmaf(lr_neg, g2, p) : p + lr_neg * g2;
}

template <bool FMA>
__global__ void __launch_bounds__(256)
sgd_batch_kernel(const SgdJob *__restrict__ jobs, int n_jobs, float lr_neg, float mom, float wd, int use_wd, int use_mom,
                 int nesterov) {
    // which tensor owns this chunk: binary search over first_chunk (ascending)
    __shared__ int s_job;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n_jobs - 1;
        const int64_t c = blockIdx.x;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first_chunk <= c) lo = mid; else hi = mid - 1;
        }
        s_job = lo;
    }
    __syncthreads();
    const SgdJob j = jobs[s_job];
    if (j.g == nullptr) return;
    const int64_t base = ((int64_t)blockIdx.x - j.first_chunk) * kSgdChunk;
    if (base >= j.numel) return;
    const bool first = j.first != 0;
    const bool aligned = (((uintptr_t)j.p | (uintptr_t)j.g | (uintptr_t)j.b) & 15) == 0;
    const int64_t end = base + kSgdChunk < j.numel ? base + kSgdChunk : j.numel;
    if (aligned) {
        const int64_t end4 = base + ((end - base) & ~(int64_t)3);
        for (int64_t i = base + threadIdx.x * 4; i < end4; i += 256 * 4) {
            float4 p = *reinterpret_cast<const float4 *>(j.p + i);
            const float4 g = *reinterpret_cast<const float4 *>(j.g + i);
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (use_mom && !first) b = *reinterpret_cast<const float4 *>(j.b + i);
            sgd_one<FMA>(p.x, g.x, b.x, first, lr_neg, mom, wd, use_wd, use_mom, nesterov);
            sgd_one<FMA>(p.y, g.y, b.y, first, lr_neg, mom, wd, use_wd, use_mom, nesterov);
            sgd_one<FMA>(p.z, g.z, b.z, first, lr_neg, mom, wd, use_wd, use_mom, nesterov);
            sgd_one<FMA>(p.w, g.w, b.w, first, lr_neg, mom, wd, use_wd, use_mom, nesterov);
            *reinterpret_cast<float4 *>(j.p + i) = p;
            if (use_mom) *reinterpret_cast<float4 *>(j.b + i) = b;
        }
        for (int64_t i = end4 + threadIdx.x; i < end; i += 256) {
            float p = j.p[i], b = (use_mom && !first) ? j.b[i] : 0.f;
            sgd_one<FMA>(p, j.g[i], b, first, lr_neg, mom, wd, use_wd, use_mom, nesterov);
            j.p[i] = p;
            if (use_mom) j.b[i] = b;
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < end; i += 256) {
            float p = j.p[i], b = (use_mom && !first) ? j.b[i] : 0.f;
            sgd_one<FMA>(p, j.g[i], b, first, lr_neg, mom, wd, use_wd, use_mom, nesterov);
            j.p[i] = p;
            if (use_mom) j.b[i] = b;
        }
    }
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

int32_t u2mkd_sgd_chunk_elements(void) { return kSgdChunk; }

int u2mkd_sgd_batch(const int64_t *jobs, int32_t n_jobs, int64_t total_chunks, float lr, float momentum, float weight_decay,
                    int32_t nesterov, int32_t contract, u2mkd_stream_t s) {
    if (n_jobs == 0 || total_chunks == 0) return 0;
    U2_REQUIRE(jobs && n_jobs > 0 && total_chunks > 0 && total_chunks < (1LL << 31), "u2mkd_sgd_batch: bad job table");
    static_assert(sizeof(SgdJob) == 6 * sizeof(int64_t), "job table row = 6 x int64");
    const SgdJob *j = reinterpret_cast<const SgdJob *>(jobs);
    const float lr_neg = -lr;
    if (contract)
        hipLaunchKernelGGL(sgd_batch_kernel<true>, dim3((unsigned)total_chunks), dim3(256), 0, as_stream(s), j, n_jobs, lr_neg,
                           momentum, weight_decay, weight_decay != 0.f, momentum != 0.f, nesterov);
    else
        hipLaunchKernelGGL(sgd_batch_kernel<false>, dim3((unsigned)total_chunks), dim3(256), 0, as_stream(s), j, n_jobs, lr_neg,
                           momentum, weight_decay, weight_decay != 0.f, momentum != 0.f, nesterov);
    return check_launch("u2mkd_sgd_batch");
}

}  // extern "C"
