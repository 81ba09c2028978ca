// Torch-free reproducer for the round-4 "stale read" item (NOTES N9): does a small kernel compute DIFFERENT RESULTS from the same
// inputs when kernels of another HIP stream run next to it on the chip?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/repro_concurrent_kernels.hip -o /tmp/repro -L u2mkd_amd/lib -lu2mkd_hip \
//         -Wl,-rpath,$PWD/u2mkd_amd/lib && /tmp/repro [rounds=4000]
//
// Victims (stream A), each checked against its own result from a run with nothing else on the GPU:
//   V1  u2mkd_ti_weights (the library's kernel, through the C ABI) on 74 232 points, scale 2
//   V2  div_kernel: q = x / y with y a power of two (IEEE division sequence v_div_scale / v_rcp / v_div_fmas / v_div_fixup),
//       compared IN the kernel with x * (1 / y), which is exact
// Aggressors (stream B), one per phase: nothing | the same kernels | exp / rcp / sqrt loop | LDS + barriers | MFMA loop | streaming copy
// Prints, per phase, the number of victim launches whose output differed and the number of wrong elements.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/u2mkd_hip.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } \
    } while (0)

__global__ void div_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ ry,
                           const int *__restrict__ idx, int n, int iters, float *__restrict__ out, unsigned *__restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i];
    const float b = y[i], rb = ry[i];
    const int id = idx[i];
    unsigned wrong = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float w = a;
        if (id == -1) w = 0.f;
        const float q = w / b;
        const float e = w * rb;
        wrong += (q != e);
        acc += q;
        a += 1.0f;
    }
    out[i] = acc;
    if (wrong) atomicAdd(bad, wrong);
}

__global__ void trans_kernel(float *__restrict__ p, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int it = 0; it < iters; ++it) v = __expf(-v) + __frcp_rn(1.0f + v * v) + sqrtf(v + 1.0f);
    p[i] = v;
}

__global__ void lds_kernel(float *__restrict__ p, int n, int iters) {
    __shared__ float s[1024];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    for (int it = 0; it < iters; ++it) {
        s[threadIdx.x] = v;
        __syncthreads();
        v += s[(threadIdx.x * 7 + it) & 255] * 0.5f;
        __syncthreads();
    }
    if (i < n) p[i] = v;
}

typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
__global__ void mfma_kernel(float *__restrict__ p, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    s8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (short)(0x3f80 + (threadIdx.x & 3)); b[k] = (short)0x3f80; }
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    if (i < n) p[i] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ void copy_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}

// further victims: pure VALU chain, compare / select / integer, load-store only, transcendental
__global__ void fma_kernel(const float *__restrict__ x, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i], a = 0.f;
    for (int it = 0; it < 16; ++it) { a = fmaf(v, 1.0009765625f, a); v = v * 0.99951171875f + 0.125f; }
    out[i] = a;
}
__global__ void select_kernel(const float *__restrict__ x, const int *__restrict__ idx, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    const float v = x[i];
    const int id = idx[i];
    for (int it = 0; it < 16; ++it) a += ((id + it) % 3 == 0 || id == -1) ? 0.f : v + (float)it;
    out[i] = a;
}
__global__ void copy_small_kernel(const float *__restrict__ x, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i];
}
__global__ void rcp_kernel(const float *__restrict__ x, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i], a = 0.f;
    for (int it = 0; it < 8; ++it) { a += __frcp_rn(v) + sqrtf(v); v += 1.0f; }
    out[i] = a;
}
// further aggressors: other MFMA shapes
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void mfma_f32_kernel(float *__restrict__ p, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    const float a = 1.0f + (threadIdx.x & 3), b = 1.0f;
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    if (i < n) p[i] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void mfma_32_kernel(float *__restrict__ p, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    f16v acc;
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    s8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (short)(0x3f80 + (threadIdx.x & 3)); b[k] = (short)0x3f80; }
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    float sum = 0.f;
    for (int k = 0; k < 16; ++k) sum += acc[k];
    if (i < n) p[i] = sum;
}

// The instruction pattern of ti_weights' corner 4, hand-written: a VALU instruction READS an SGPR pair as a scalar operand, a VALU
// compare then WRITES the pair, and after WAIT wait states a v_cndmask reads it as its lane mask.  The compiler's hazard recogniser
// puts `s_nop 1` (2 wait states) there.  out = (idx != -1) ? 1.0f : 0.0f; a stale pair (0x0f0f...) gives a recognisable pattern.
#define HAZARD_KERNEL(NAME, PRIOR_READ, NOPS)                                                                              \
    __global__ void NAME(const int *__restrict__ idx, int n, float *__restrict__ out) {                                      \
        const int i = blockIdx.x * blockDim.x + threadIdx.x;                                                                 \
        if (i >= n) return;                                                                                                  \
        const int v = idx[i];                                                                                                \
        float r;                                                                                                             \
        unsigned long long tmp;                                                                                              \
        const unsigned long long zero64 = (unsigned long long)v;                                                             \
        const float one = 1.0f;                                                                                              \
        asm volatile("s_mov_b32 s20, 0x0f0f0f0f\n\ts_mov_b32 s21, 0x0f0f0f0f\n\ts_nop 4\n\t" PRIOR_READ                      \
                     "s_nop 4\n\tv_cmp_ne_u32_e64 s[20:21], -1, %[v]\n\t" NOPS "v_cndmask_b32_e64 %[r], 0, %[one], s[20:21]"   \
                     : [r] "=v"(r), [tmp] "=&v"(tmp)                                                                         \
                     : [v] "v"(v), [one] "v"(one), [z] "v"(zero64)                                                           \
                     : "s20", "s21");                                                                                        \
        out[i] = r + (float)(tmp & 0);                                                                                       \
    }
#define PRIOR "v_lshl_add_u64 %[tmp], %[z], 0, s[20:21]\n\t"
#define NOPRIOR "v_lshl_add_u64 %[tmp], %[z], 0, 0\n\t"
HAZARD_KERNEL(hazard_r_w0, PRIOR, "")
HAZARD_KERNEL(hazard_r_w1, PRIOR, "s_nop 0\n\t")
HAZARD_KERNEL(hazard_r_w2, PRIOR, "s_nop 1\n\t")
HAZARD_KERNEL(hazard_r_w4, PRIOR, "s_nop 3\n\t")
HAZARD_KERNEL(hazard_r_w8, PRIOR, "s_nop 7\n\t")
HAZARD_KERNEL(hazard_r_w16, PRIOR, "s_nop 7\n\ts_nop 7\n\t")
HAZARD_KERNEL(hazard_n_w0, NOPRIOR, "")
HAZARD_KERNEL(hazard_n_w2, NOPRIOR, "s_nop 1\n\t")

// aggressors: long-running waves of one instruction kind, 64 workgroups x 256 threads
typedef short s4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void ag_bf16_16x16x32(float *__restrict__ p, int iters) {
    f4 acc = {0, 0, 0, 0};
    s8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = 0x3f80; b[k] = 0x3f80; }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
__global__ void ag_bf16_16x16x16_1k(float *__restrict__ p, int iters) {
    f4 acc = {0, 0, 0, 0};
    s4v a, b;
    for (int k = 0; k < 4; ++k) { a[k] = 0x3f80; b[k] = 0x3f80; }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
__global__ void ag_f32_16x16x4(float *__restrict__ p, int iters) {
    f4 acc = {0, 0, 0, 0};
    float a = 1.f, b = 1.f;
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
__global__ void ag_f32_32x32x2(float *__restrict__ p, int iters) {
    f16v acc;
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    float a = 1.f, b = 1.f;
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
__global__ void ag_f16_16x16x32(float *__restrict__ p, int iters) {
    f4 acc = {0, 0, 0, 0};
    h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)1.0f; b[k] = (_Float16)1.0f; }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
__global__ void ag_bf16_16x16x32_spaced(float *__restrict__ p, int iters) {
    f4 acc = {0, 0, 0, 0};
    s8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = 0x3f80; b[k] = 0x3f80; }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        asm volatile("s_nop 7");
        asm volatile("s_nop 7");
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
// the gfx950 form with its accumulator in the ACCUMULATION register file (a[...]) instead of the vector register file
__global__ void ag_bf16_16x16x32_agpr(float *__restrict__ p, int iters) {
    f4 acc = {0, 0, 0, 0};
    s8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = 0x3f80; b[k] = 0x3f80; }
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
// ... with four independent accumulators in the accumulation file (no dependent chain)
__global__ void ag_bf16_16x16x32_agpr4(float *__restrict__ p, int iters) {
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    s8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = 0x3f80; b[k] = 0x3f80; }
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                     "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %5, %3"
                     : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0];
}
// the gfx950 form, four independent accumulators in vector registers (the compiler's default placement)
__global__ void ag_bf16_16x16x32_vgpr4(float *__restrict__ p, int iters) {
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    s8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = 0x3f80; b[k] = 0x3f80; }
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                     "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %5, %3"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0];
}
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ void ag_f16_16x16x16(float *__restrict__ p, int iters) {
    f4 acc = {0, 0, 0, 0};
    h4 a, b;
    for (int k = 0; k < 4; ++k) { a[k] = (_Float16)1.0f; b[k] = (_Float16)1.0f; }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
__global__ void ag_valu_fma(float *__restrict__ p, int iters) {
    float v = (float)threadIdx.x, a = 0.f;
    for (int it = 0; it < iters; ++it) {
        a = fmaf(v, 1.0009765625f, a);
        v = v * 0.99951171875f + 0.125f;
    }
    p[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

__global__ void compare_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, int64_t n, unsigned *__restrict__ diff) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(diff, 1u);
}

struct Victim {
    int64_t n;
    float scale;
    float *coords, *w, *w_ref;
    int64_t *idx_kn;
    int32_t *i8, *i8_ref;
};

static Victim make_victim(int64_t n, float scale, unsigned seed) {
    Victim v{};
    v.n = n;
    v.scale = scale;
    std::vector<float> c(4 * n);
    std::vector<int64_t> k(8 * n);
    srand(seed);
    for (int64_t i = 0; i < n; ++i) {
        for (int d = 0; d < 3; ++d) c[4 * i + d] = ((float)(rand() % 1000) * 0.05f) / 0.05f;     // the reference's (c * pres) / vres
        c[4 * i + 3] = 0.f;
    }
    for (int64_t j = 0; j < 8 * n; ++j) k[j] = (rand() % 3 == 0) ? -1 : rand() % n;
    CK(hipMalloc(&v.coords, 16 * n));
    CK(hipMalloc(&v.idx_kn, 64 * n));
    CK(hipMalloc(&v.w, 32 * n));
    CK(hipMalloc(&v.w_ref, 32 * n));
    CK(hipMalloc(&v.i8, 32 * n));
    CK(hipMalloc(&v.i8_ref, 32 * n));
    CK(hipMemcpy(v.coords, c.data(), 16 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(v.idx_kn, k.data(), 64 * n, hipMemcpyHostToDevice));
    return v;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 4000;
    const int only_aggr = argc > 2 ? atoi(argv[2]) : -1;
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    Victim va = make_victim(74232, 2.f, 1), vb = make_victim(74267, 4.f, 2);
    const int n = 74232;
    float *x, *y, *ry, *scratch;
    int *idx;
    unsigned *bad;
    std::vector<float> hx(n), hy(n), hr(n);
    std::vector<int> hi(n);
    for (int i = 0; i < n; ++i) {
        hx[i] = (float)(i % 977) * 0.37f + 1.f;
        const int e = i % 4;
        hy[i] = (float)(1 << e) * (e == 3 ? 8.f : 1.f);
        hr[i] = 1.f / hy[i];
        hi[i] = (i % 5 == 0) ? -1 : i;
    }
    CK(hipMalloc(&x, 4 * n)); CK(hipMalloc(&y, 4 * n)); CK(hipMalloc(&ry, 4 * n)); CK(hipMalloc(&idx, 4 * n)); CK(hipMalloc(&bad, 4));
    const int64_t big = 64 << 20;
    CK(hipMalloc(&scratch, big * 2));
    CK(hipMemset(scratch, 0, big * 2));
    CK(hipMemcpy(x, hx.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(y, hy.data(), 4 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(ry, hr.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(idx, hi.data(), 4 * n, hipMemcpyHostToDevice));
    CK(hipMemset(bad, 0, 4));

    // victims: name, output words, launch(out, stream)
    const int NV = 15;
    const char *vname[NV] = {"ti_weights w", "ti_weights idx", "div (IEEE x / y)", "fma chain", "compare / select", "load-store copy", "rcp + sqrt",
                             "asm: VALU read, VALU cmp write, 0 wait states, cndmask", "asm: ... 1 wait state", "asm: ... 2 wait states (the compiler's)",
                             "asm: ... 4 wait states", "asm: ... 8 wait states", "asm: ... 16 wait states", "asm: no prior VALU read, 0 wait states",
                             "asm: no prior VALU read, 2 wait states"};
    const int64_t vwords[NV] = {8 * va.n, 8 * va.n, n, n, n, n, n, n, n, n, n, n, n, n, n};
    uint32_t *vout[NV], *vref[NV];
    unsigned *vdiff;
    CK(hipMalloc(&vdiff, 4 * NV));
    for (int v = 0; v < NV; ++v) { CK(hipMalloc(&vout[v], 4 * vwords[v])); CK(hipMalloc(&vref[v], 4 * vwords[v])); }
    auto launch_victims = [&](uint32_t **o, hipStream_t s) {
        if (u2mkd_ti_weights(va.coords, va.idx_kn, va.n, va.scale, (float *)o[0], (int32_t *)o[1], s)) { fprintf(stderr, "%s\n", u2mkd_last_error()); exit(3); }
        const dim3 g((n + 255) / 256), b(256);
        hipLaunchKernelGGL(div_kernel, g, b, 0, s, x, y, ry, idx, n, 8, (float *)o[2], bad);
        hipLaunchKernelGGL(fma_kernel, g, b, 0, s, x, n, (float *)o[3]);
        hipLaunchKernelGGL(select_kernel, g, b, 0, s, x, idx, n, (float *)o[4]);
        hipLaunchKernelGGL(copy_small_kernel, g, b, 0, s, x, n, (float *)o[5]);
        hipLaunchKernelGGL(rcp_kernel, g, b, 0, s, x, n, (float *)o[6]);
        hipLaunchKernelGGL(hazard_r_w0, g, b, 0, s, idx, n, (float *)o[7]);
        hipLaunchKernelGGL(hazard_r_w1, g, b, 0, s, idx, n, (float *)o[8]);
        hipLaunchKernelGGL(hazard_r_w2, g, b, 0, s, idx, n, (float *)o[9]);
        hipLaunchKernelGGL(hazard_r_w4, g, b, 0, s, idx, n, (float *)o[10]);
        hipLaunchKernelGGL(hazard_r_w8, g, b, 0, s, idx, n, (float *)o[11]);
        hipLaunchKernelGGL(hazard_r_w16, g, b, 0, s, idx, n, (float *)o[12]);
        hipLaunchKernelGGL(hazard_n_w0, g, b, 0, s, idx, n, (float *)o[13]);
        hipLaunchKernelGGL(hazard_n_w2, g, b, 0, s, idx, n, (float *)o[14]);
    };
    launch_victims(vref, sa);            // references: nothing else on the GPU
    CK(hipDeviceSynchronize());
    {   // the hand-written pattern's expectation comes from the host (its 0-wait-state form may fail on its own)
        std::vector<float> e(n);
        for (int i = 0; i < n; ++i) e[i] = hi[i] != -1 ? 1.f : 0.f;
        for (int v = 7; v < NV; ++v) CK(hipMemcpy(vref[v], e.data(), 4 * n, hipMemcpyHostToDevice));
    }
    unsigned h = 0;
    CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    printf("alone: div_kernel in-kernel mismatches %u\n", h);

    {   // rate of the matrix instructions used as aggressors, one launch filling the chip (1024 workgroups x 4 waves)
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        struct { const char *name; void (*k)(float *, int); double flop; int iters; } rates[] = {
            {"v_mfma_f32_16x16x32_bf16", ag_bf16_16x16x32, 2.0 * 16 * 16 * 32, 4096}, {"v_mfma_f32_16x16x16_bf16", ag_bf16_16x16x16_1k, 2.0 * 16 * 16 * 16, 8192},
            {"v_mfma_f32_16x16x4_f32", ag_f32_16x16x4, 2.0 * 16 * 16 * 4, 4096}, {"v_mfma_f32_16x16x32_f16", ag_f16_16x16x32, 2.0 * 16 * 16 * 32, 4096},
            {"v_mfma_f32_16x16x16_f16", ag_f16_16x16x16, 2.0 * 16 * 16 * 16, 8192}};
        for (auto &r : rates) {
            hipLaunchKernelGGL(r.k, dim3(1024), dim3(256), 0, sa, scratch, r.iters);
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(r.k, dim3(1024), dim3(256), 0, sa, scratch, r.iters);
            CK(hipEventRecord(e1, sa));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("rate: %-28s %.1f TFLOP/s (dependent chain per wave, 4096 waves)\n", r.name, 1024.0 * 4 * r.iters * r.flop / (ms * 1e-3) / 1e12);
        }
    }
    const int NA = 20;
    const char *aname[NA] = {"nothing", "the same victim kernels", "exp/rcp/sqrt loop", "LDS + barriers", "MFMA 16x16x32 bf16 loop", "streaming copy",
                             "MFMA 16x16x4 f32 loop", "MFMA 32x32x16 bf16 loop", "MFMA 16x16x32 bf16, 64 workgroups only",
                             "64 wg: v_mfma_f32_16x16x32_bf16", "64 wg: v_mfma_f32_16x16x16_bf16 (gfx942 form)", "64 wg: v_mfma_f32_16x16x4_f32",
                             "64 wg: v_mfma_f32_32x32x2_f32", "64 wg: v_mfma_f32_16x16x32_f16", "64 wg: v_mfma_f32_16x16x32_bf16 + 16 idle cycles each",
                             "64 wg: plain VALU fma loop (no MFMA)", "64 wg: v_mfma_f32_16x16x32_bf16, accumulator in a[...] (AGPR)",
                             "64 wg: v_mfma_f32_16x16x32_bf16, 4 independent accumulators in a[...]",
                             "64 wg: v_mfma_f32_16x16x32_bf16, 4 independent accumulators in v[...]",
                             "64 wg: v_mfma_f32_16x16x16_f16 (gfx942 form)"};
    std::vector<uint32_t> hout, href;
    for (int phase = 0; phase < NA; ++phase) {
        if (only_aggr >= 0 && phase < only_aggr) continue;
        unsigned rounds_bad[NV] = {0}, elems_bad[NV] = {0}, inkernel = 0;
        bool sampled[NV] = {false};
        CK(hipMemset(bad, 0, 4));
        for (int r = 0; r < rounds; ++r) {
            for (int rep = 0; rep < 3; ++rep) {      // aggressor launches before, next to and behind the victims
                switch (phase) {
                case 1: launch_victims(vout, sb); break;        // (writes the same values to the same outputs)
                case 2: hipLaunchKernelGGL(trans_kernel, dim3(600), dim3(256), 0, sb, scratch, 600 * 256, 64); break;
                case 3: hipLaunchKernelGGL(lds_kernel, dim3(600), dim3(256), 0, sb, scratch, 600 * 256, 32); break;
                case 4: hipLaunchKernelGGL(mfma_kernel, dim3(600), dim3(256), 0, sb, scratch, 600 * 256, 256); break;
                case 5: hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, sb, (const float4 *)scratch, (float4 *)(scratch + big / 4), big / 16); break;
                case 6: hipLaunchKernelGGL(mfma_f32_kernel, dim3(600), dim3(256), 0, sb, scratch, 600 * 256, 256); break;
                case 7: hipLaunchKernelGGL(mfma_32_kernel, dim3(600), dim3(256), 0, sb, scratch, 600 * 256, 128); break;
                case 8: hipLaunchKernelGGL(mfma_kernel, dim3(64), dim3(256), 0, sb, scratch, 64 * 256, 2048); break;
                case 9: hipLaunchKernelGGL(ag_bf16_16x16x32, dim3(64), dim3(256), 0, sb, scratch, 2048); break;
                case 10: hipLaunchKernelGGL(ag_bf16_16x16x16_1k, dim3(64), dim3(256), 0, sb, scratch, 4096); break;
                case 11: hipLaunchKernelGGL(ag_f32_16x16x4, dim3(64), dim3(256), 0, sb, scratch, 2048); break;
                case 12: hipLaunchKernelGGL(ag_f32_32x32x2, dim3(64), dim3(256), 0, sb, scratch, 1024); break;
                case 13: hipLaunchKernelGGL(ag_f16_16x16x32, dim3(64), dim3(256), 0, sb, scratch, 2048); break;
                case 14: hipLaunchKernelGGL(ag_bf16_16x16x32_spaced, dim3(64), dim3(256), 0, sb, scratch, 1024); break;
                case 15: hipLaunchKernelGGL(ag_valu_fma, dim3(64), dim3(256), 0, sb, scratch, 16384); break;
                case 16: hipLaunchKernelGGL(ag_bf16_16x16x32_agpr, dim3(64), dim3(256), 0, sb, scratch, 2048); break;
                case 17: hipLaunchKernelGGL(ag_bf16_16x16x32_agpr4, dim3(64), dim3(256), 0, sb, scratch, 512); break;
                case 18: hipLaunchKernelGGL(ag_bf16_16x16x32_vgpr4, dim3(64), dim3(256), 0, sb, scratch, 512); break;
                case 19: hipLaunchKernelGGL(ag_f16_16x16x16, dim3(64), dim3(256), 0, sb, scratch, 4096); break;
                default: break;
                }
                if (rep == 0) {
                    launch_victims(vout, sa);
                    CK(hipMemsetAsync(vdiff, 0, 4 * NV, sa));
                    for (int v = 0; v < NV; ++v)
                        hipLaunchKernelGGL(compare_kernel, dim3((unsigned)((vwords[v] + 255) / 256)), dim3(256), 0, sa, vout[v], vref[v], vwords[v], vdiff + v);
                }
            }
            unsigned d[NV];
            CK(hipMemcpyAsync(d, vdiff, 4 * NV, hipMemcpyDeviceToHost, sa));
            CK(hipStreamSynchronize(sa));
            for (int v = 0; v < NV; ++v)
                if (d[v]) {
                    ++rounds_bad[v];
                    elems_bad[v] += d[v];
                    if (!sampled[v]) {       // show what a wrong element looks like (first occurrence per victim and phase)
                        sampled[v] = true;
                        hout.resize(vwords[v]); href.resize(vwords[v]);
                        CK(hipMemcpy(hout.data(), vout[v], 4 * vwords[v], hipMemcpyDeviceToHost));
                        CK(hipMemcpy(href.data(), vref[v], 4 * vwords[v], hipMemcpyDeviceToHost));
                        int shown = 0;
                        for (int64_t j = 0; j < vwords[v] && shown < 6; ++j)
                            if (hout[j] != href[j]) {
                                float a, b;
                                memcpy(&a, &hout[j], 4); memcpy(&b, &href[j], 4);
                                printf("      %s: round %d element %lld (thread %lld, wave %lld): got %#x (%g), alone %#x (%g)\n", vname[v], r, (long long)j,
                                       (long long)(v < 2 ? j / 8 : j), (long long)((v < 2 ? j / 8 : j) / 64), hout[j], a, href[j], b);
                                ++shown;
                            }
                    }
                }
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&inkernel, bad, 4, hipMemcpyDeviceToHost));
        printf("aggressor on the other stream: %s (%d rounds)\n", aname[phase], rounds);
        bool any = false;
        for (int v = 0; v < NV; ++v)
            if (rounds_bad[v]) { any = true; printf("      %-60s rounds with a different result: %u (%u elements)\n", vname[v], rounds_bad[v], elems_bad[v]); }
        if (!any) printf("      every victim identical to its reference in every round\n");
        printf("      div_kernel's in-kernel check (q != x * (1 / y)): %u\n", inkernel);
        fflush(stdout);
    }
    return 0;
}
