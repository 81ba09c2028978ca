// Point <-> voxel scatter/gather: count, voxelize (scatter-mean), devoxelize
// (8-corner trilinear gather) and their backward passes, trilinear weights.
// Replaces torchsparse v1.4.0 count_cuda / voxelize_*_cuda / devoxelize_*_cuda
// and F.calc_ti_weights (call sites core/models/utils.py).
//
// HBM-bound row work: one thread moves 16 B (float4) of a row, so a row of C
// floats is covered by C/4 adjacent lanes and every wave instruction touches
// whole contiguous row segments.
#include <stdlib.h>

#include "common.h"

namespace u2mkd {

__global__ void count_kernel(const int32_t *__restrict__ idx, int64_t n, int32_t *__restrict__ counts,
                             int64_t num) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int p = idx[i];
        if (p >= 0 && p < num) atomicAdd(&counts[p], 1);
    }
}

// out[idx[i]] += feats[i] / counts[idx[i]]  (float atomics, as the reference)
template <int VEC>
__global__ void voxelize_fwd_kernel(const float *__restrict__ feats, const int32_t *__restrict__ idx,
                                    const int32_t *__restrict__ counts, int64_t n, int64_t nv, int c,
                                    float *__restrict__ out) {
    int cv = c / VEC;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / cv;
    int j = (int)(t - i * cv) * VEC;
    if (i >= n) return;
    int p = idx[i];
    if (p < 0 || p >= nv) return;
    int cnt = counts[p];
    if (cnt == 0) return;
    float inv = (float)cnt;
    const float *src = feats + i * c + j;
    float *dst = out + (int64_t)p * c + j;
    if (VEC == 4) {
        float4 v = *reinterpret_cast<const float4 *>(src);
        atomicAdd(dst + 0, v.x / inv);
        atomicAdd(dst + 1, v.y / inv);
        atomicAdd(dst + 2, v.z / inv);
        atomicAdd(dst + 3, v.w / inv);
    } else {
        atomicAdd(dst, src[0] / inv);
    }
}

template <int VEC>
__global__ void voxelize_bwd_kernel(const float *__restrict__ gout, const int32_t *__restrict__ idx,
                                    const int32_t *__restrict__ counts, int64_t n, int64_t nv, int c,
                                    float *__restrict__ gin) {
    int cv = c / VEC;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / cv;
    int j = (int)(t - i * cv) * VEC;
    if (i >= n) return;
    int p = idx[i];
    float *dst = gin + i * c + j;
    bool ok = p >= 0 && p < nv;
    int cnt = ok ? counts[p] : 0;
    if (VEC == 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cnt > 0) {
            float4 g = *reinterpret_cast<const float4 *>(gout + (int64_t)p * c + j);
            float inv = (float)cnt;
            v = make_float4(g.x / inv, g.y / inv, g.z / inv, g.w / inv);
        }
        *reinterpret_cast<float4 *>(dst) = v;
    } else {
        dst[0] = cnt > 0 ? gout[(int64_t)p * c + j] / (float)cnt : 0.f;
    }
}

// out[i] = sum_k w[i,k] * feats[idx[i,k]]   (k ascending, idx < 0 skipped)
template <int VEC>
__global__ void devoxelize_fwd_kernel(const float *__restrict__ feats, const int32_t *__restrict__ idx,
                                      const float *__restrict__ w, int64_t n, int c, float *__restrict__ out) {
    int cv = c / VEC;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / cv;
    int j = (int)(t - i * cv) * VEC;
    if (i >= n) return;
    const int32_t *ii = idx + i * 8;
    const float *ww = w + i * 8;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int p = ii[k];
        float wk = ww[k];
        if (p >= 0) {
            const float *src = feats + (int64_t)p * c + j;
            if (VEC == 4) {
                float4 f = *reinterpret_cast<const float4 *>(src);
                acc[0] += wk * f.x; acc[1] += wk * f.y; acc[2] += wk * f.z; acc[3] += wk * f.w;
            } else {
                acc[0] += wk * src[0];
            }
        }
    }
    float *dst = out + i * c + j;
    if (VEC == 4) *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else dst[0] = acc[0];
}

// g_feats[idx[i,k]] += w[i,k] * g_out[i]   (float atomics, as the reference)
template <int VEC>
__global__ void devoxelize_bwd_kernel(const float *__restrict__ gout, const int32_t *__restrict__ idx,
                                      const float *__restrict__ w, int64_t n, int64_t nv, int c,
                                      float *__restrict__ gin) {
    int cv = c / VEC;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / cv;
    int j = (int)(t - i * cv) * VEC;
    if (i >= n) return;
    const int32_t *ii = idx + i * 8;
    const float *ww = w + i * 8;
    float g[VEC];
    if (VEC == 4) {
        float4 v = *reinterpret_cast<const float4 *>(gout + i * c + j);
        g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
    } else {
        g[0] = gout[i * c + j];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int p = ii[k];
        float wk = ww[k];
        if (p >= 0 && p < nv && wk != 0.f) {
            float *dst = gin + (int64_t)p * c + j;
#pragma unroll
            for (int v = 0; v < VEC; ++v) atomicAdd(dst + v, wk * g[v]);
        }
    }
}

// Deterministic scatter replacement: out[v] = sum over the entries e of segment v of
// w[e] * src[row[e]] (optionally / segment length).  Entries are pre-sorted by destination
// (CSR built once per point<->voxel map), so there are no atomics, every output row is
// written once, and the summation order is fixed.  One thread = 16 B of one output row.
__global__ void segment_sum_kernel(const float *__restrict__ src, int c4, const int32_t *__restrict__ erow,
                                   const float *__restrict__ ew, const int32_t *__restrict__ seg, int64_t nv,
                                   int mean, float *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t v = t / c4;
    if (v >= nv) return;
    int j = (int)(t - v * c4);
    const int e0 = seg[v], e1 = seg[v + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = e0;
    // mean: every term is divided by the segment length BEFORE it is added, in entry order -- the arithmetic
    // of torchsparse's voxelize (out[idx[i]] += feats[i] / counts[idx[i]]), so voxel means
    // (and with them the metric coordinates SphereFormer quantises into windows and relative-position bins)
    // are bit-identical to the CPU path instead of differing in the last place
    const float cnt = (float)(e1 - e0);
    // four entries in flight (segments reach ~100 entries at stride 16: a one-at-a-time walk is a
    // chain of dependent L2 round trips); the accumulation order stays e0, e0+1, ...
    for (; e + 4 <= e1; e += 4) {
        int r[4];
        float w[4];
        float4 f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { r[u] = erow[e + u]; w[u] = ew ? ew[e + u] : 1.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = reinterpret_cast<const float4 *>(src)[(int64_t)r[u] * c4 + j];
        if (mean) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc.x += f[u].x / cnt; acc.y += f[u].y / cnt; acc.z += f[u].z / cnt; acc.w += f[u].w / cnt;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc.x += w[u] * f[u].x; acc.y += w[u] * f[u].y; acc.z += w[u] * f[u].z; acc.w += w[u] * f[u].w;
            }
        }
    }
    for (; e < e1; ++e) {
        float w = ew ? ew[e] : 1.f;
        float4 f = reinterpret_cast<const float4 *>(src)[(int64_t)erow[e] * c4 + j];
        if (mean) {
            acc.x += f.x / cnt; acc.y += f.y / cnt; acc.z += f.z / cnt; acc.w += f.w / cnt;
        } else {
            acc.x += w * f.x; acc.y += w * f.y; acc.z += w * f.z; acc.w += w * f.w;
        }
    }
    reinterpret_cast<float4 *>(out)[t] = acc;
}

// ---- the row movers on float OR bf16 rows (common.h ld4 / st4), 4 channels per thread; the fp32 entries keep the
// kernels above (bit-for-bit the round-2 arithmetic), the bf16-storage entries use these with T = bf16row: sums in fp32
// in the same fixed order, one rounding at the store
template <typename T>
__global__ void voxelize_bwd_rows_kernel(const T *__restrict__ gout, const int32_t *__restrict__ idx,
                                         const int32_t *__restrict__ counts, int64_t n, int64_t nv, int c4,
                                         T *__restrict__ gin) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / c4;
    int j = (int)(t - i * c4);
    if (i >= n) return;
    int p = idx[i];
    bool ok = p >= 0 && p < nv;
    int cnt = ok ? counts[p] : 0;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cnt > 0) {
        float4 g = ld4(gout, (int64_t)p * c4 + j);
        float inv = (float)cnt;
        v = make_float4(g.x / inv, g.y / inv, g.z / inv, g.w / inv);
    }
    st4(gin, t, v);
}

template <typename T>
__global__ void devoxelize_fwd_rows_kernel(const T *__restrict__ feats, const int32_t *__restrict__ idx,
                                           const float *__restrict__ w, int64_t n, int c4, T *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / c4;
    int j = (int)(t - i * c4);
    if (i >= n) return;
    const int32_t *ii = idx + i * 8;
    const float *ww = w + i * 8;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int p = ii[k];
        float wk = ww[k];
        if (p >= 0) {
            float4 f = ld4(feats, (int64_t)p * c4 + j);
            acc.x += wk * f.x; acc.y += wk * f.y; acc.z += wk * f.z; acc.w += wk * f.w;
        }
    }
    st4(out, t, acc);
}

template <typename T>
__global__ void segment_sum_rows_kernel(const T *__restrict__ src, int c4, const int32_t *__restrict__ erow,
                                        const float *__restrict__ ew, const int32_t *__restrict__ seg, int64_t nv,
                                        int mean, T *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t v = t / c4;
    if (v >= nv) return;
    int j = (int)(t - v * c4);
    const int e0 = seg[v], e1 = seg[v + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = e0;
    const float cnt = (float)(e1 - e0);
    for (; e + 4 <= e1; e += 4) {
        int r[4];
        float w[4];
        float4 f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { r[u] = erow[e + u]; w[u] = ew ? ew[e + u] : 1.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = ld4(src, (int64_t)r[u] * c4 + j);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (mean) { acc.x += f[u].x / cnt; acc.y += f[u].y / cnt; acc.z += f[u].z / cnt; acc.w += f[u].w / cnt; }
            else { acc.x += w[u] * f[u].x; acc.y += w[u] * f[u].y; acc.z += w[u] * f[u].z; acc.w += w[u] * f[u].w; }
        }
    }
    for (; e < e1; ++e) {
        float w = ew ? ew[e] : 1.f;
        float4 f = ld4(src, (int64_t)erow[e] * c4 + j);
        if (mean) { acc.x += f.x / cnt; acc.y += f.y / cnt; acc.z += f.z / cnt; acc.w += f.w / cnt; }
        else { acc.x += w * f.x; acc.y += w * f.y; acc.z += w * f.z; acc.w += w * f.w; }
    }
    st4(out, t, acc);
}

// F.calc_ti_weights fused with the [8,N] -> [N,8] transposes of
// core/models/utils.py:94-95.  Same operation order as the reference:
// products of differences, / scale^3, zero where idx == -1, / (sum + 1e-8).
__global__ void ti_weights_kernel(const float4 *__restrict__ coords, const int64_t *__restrict__ idx_kn, int64_t n,
                                  float scale, float *__restrict__ w_n8, int32_t *__restrict__ idx_n8) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = coords[i];
    float xf, yf, zf;
    if (scale != 1.f) {
        xf = floorf(p.x / scale) * scale; yf = floorf(p.y / scale) * scale; zf = floorf(p.z / scale) * scale;
    } else {
        xf = floorf(p.x); yf = floorf(p.y); zf = floorf(p.z);
    }
    float xc = xf + scale, yc = yf + scale, zc = zf + scale;
    float w[8];
    w[0] = (xc - p.x) * (yc - p.y) * (zc - p.z);
    w[1] = (xc - p.x) * (yc - p.y) * (p.z - zf);
    w[2] = (xc - p.x) * (p.y - yf) * (zc - p.z);
    w[3] = (xc - p.x) * (p.y - yf) * (p.z - zf);
    w[4] = (p.x - xf) * (yc - p.y) * (zc - p.z);
    w[5] = (p.x - xf) * (yc - p.y) * (p.z - zf);
    w[6] = (p.x - xf) * (p.y - yf) * (zc - p.z);
    w[7] = (p.x - xf) * (p.y - yf) * (p.z - zf);
    float s3 = scale * scale * scale;
    float sum = 0.f;
    int id[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (scale != 1.f) w[k] = w[k] / s3;
        id[k] = (int)idx_kn[(int64_t)k * n + i];
        if (id[k] == -1) w[k] = 0.f;
        sum += w[k];
    }
    sum += 1e-8f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        w_n8[i * 8 + k] = w[k] / sum;
        idx_n8[i * 8 + k] = id[k];
    }
}

#ifdef U2MKD_DEBUG_PROBE     // (tools/build_variant.sh _probe "-DU2MKD_DEBUG_PROBE" voxel.hip; not in the shipped library)
// ---- debug: coherence probe (tools/dbg_stale_probe.py; U2MKD_DEBUG_TI_PROBE=1) -------------------------------------
// The same arithmetic as ti_weights_kernel on the values an ORDINARY load returns, but every input word is read a second
// and third time past the caches (agent-scope and system-scope atomic loads) and a fourth time with an ordinary load
// behind an explicit cache invalidate; a thread whose reads disagree appends a record to a device-side log.  A record
// means: this kernel, launched behind the producer of `idx_kn` / `coords` on the same stream, saw two different values
// of one address that nothing writes while it runs.
struct ProbeEntry {
    int64_t i;          // point
    int32_t k;          // corner (0..7), or 8 + j for the j-th 64-bit half of the coordinate row
    uint32_t xcc;       // HW_REG_XCC_ID of the reading wave
    uint32_t hwid;      // HW_REG_HW_ID
    uint32_t launch;    // probe launch counter
    int64_t v_plain, v_agent, v_sys, v_after_inv;
    uint64_t t;         // wall_clock64() at the read
};
constexpr int kProbeCap = 4096;
__device__ unsigned int g_probe_n;
__device__ unsigned int g_probe_launch;
__device__ ProbeEntry g_probe_log[kProbeCap];

__device__ __forceinline__ int64_t plain_load_after_inv(const int64_t *p) {
    int64_t v;
    asm volatile("buffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)"
                 : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ int64_t probe_word(const int64_t *p, int64_t i, int k, unsigned launch) {
    const int64_t v0 = *p;
    const int64_t va = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t vs = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v0 != vs || va != vs) {
        const int64_t vi = plain_load_after_inv(p);
        const unsigned slot = atomicAdd(&g_probe_n, 1u);
        if (slot < (unsigned)kProbeCap) {
            ProbeEntry e;
            e.i = i; e.k = k;
            e.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
            e.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
            e.launch = launch;
            e.v_plain = v0; e.v_agent = va; e.v_sys = vs; e.v_after_inv = vi;
            e.t = wall_clock64();
            g_probe_log[slot] = e;
        }
    }
    return v0;
}

// every workgroup reports the launch number and the output pointer IT received as kernel arguments (the array is addressed
// through its symbol, not through an argument): a workgroup that ran on another launch's arguments shows up as a record missing
// from its launch's row and as a record in a row that is not of this step
constexpr int kWgRing = 64, kWgMax = 2048;
__device__ uint2 g_wg_seen[kWgRing][kWgMax];
// what every thread of a probe launch SAW and COMPUTED, kept apart from its outputs: {bit k = (idx of corner k == -1) | xcc << 8,
// bits of its normalised w[0], the three cell fractions as small integers, launch}
constexpr int kRowRing = 16, kRowMax = 81920;
__device__ uint4 g_row_dbg[kRowRing][kRowMax];
__device__ uint4 g_row_dbg2[kRowRing][kRowMax];     // {wall-clock ticks between the thread's first load and its stores, TRAPSTS, STATUS, launch}

__global__ void ti_weights_probe_kernel(const float4 *__restrict__ coords, const int64_t *__restrict__ idx_kn, int64_t n,
                                        float scale, float *__restrict__ w_n8, int32_t *__restrict__ idx_n8, unsigned launch) {
    if (threadIdx.x == 0 && blockIdx.x < kWgMax)
        g_wg_seen[launch % kWgRing][blockIdx.x] = make_uint2(launch + 1, (unsigned)(uintptr_t)w_n8);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t t_begin = wall_clock64();
    const int64_t *cw = reinterpret_cast<const int64_t *>(coords + i);
    const int64_t c0 = probe_word(cw, i, 8, launch), c1 = probe_word(cw + 1, i, 9, launch);
    float4 p;
    p.x = __int_as_float((int)(c0 & 0xffffffff)); p.y = __int_as_float((int)(c0 >> 32));
    p.z = __int_as_float((int)(c1 & 0xffffffff)); p.w = __int_as_float((int)(c1 >> 32));
    float xf, yf, zf;
    if (scale != 1.f) {
        xf = floorf(p.x / scale) * scale; yf = floorf(p.y / scale) * scale; zf = floorf(p.z / scale) * scale;
    } else {
        xf = floorf(p.x); yf = floorf(p.y); zf = floorf(p.z);
    }
    float xc = xf + scale, yc = yf + scale, zc = zf + scale;
    float w[8];
    w[0] = (xc - p.x) * (yc - p.y) * (zc - p.z);
    w[1] = (xc - p.x) * (yc - p.y) * (p.z - zf);
    w[2] = (xc - p.x) * (p.y - yf) * (zc - p.z);
    w[3] = (xc - p.x) * (p.y - yf) * (p.z - zf);
    w[4] = (p.x - xf) * (yc - p.y) * (zc - p.z);
    w[5] = (p.x - xf) * (yc - p.y) * (p.z - zf);
    w[6] = (p.x - xf) * (p.y - yf) * (zc - p.z);
    w[7] = (p.x - xf) * (p.y - yf) * (p.z - zf);
    float s3 = scale * scale * scale;
    float sum = 0.f;
    int id[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (scale != 1.f) w[k] = w[k] / s3;
        id[k] = (int)probe_word(idx_kn + (int64_t)k * n + i, i, k, launch);
        if (id[k] == -1) w[k] = 0.f;
        sum += w[k];
    }
    sum += 1e-8f;
    if (i < kRowMax) {
        unsigned miss = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) miss |= (id[k] == -1 ? 1u : 0u) << k;
        const unsigned frac = (unsigned)(int)(p.x - xf) | (unsigned)(int)(p.y - yf) << 8 | (unsigned)(int)(p.z - zf) << 16;
        g_row_dbg[launch % kRowRing][i] = make_uint4(miss | (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) << 8,
                                                     __float_as_uint(w[0] / sum), frac, launch);
        g_row_dbg2[launch % kRowRing][i] = make_uint4((unsigned)(wall_clock64() - t_begin), __builtin_amdgcn_s_getreg((31 << 11) | 3),
                                                      __builtin_amdgcn_s_getreg((31 << 11) | 2), launch);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        w_n8[i * 8 + k] = w[k] / sum;
        idx_n8[i * 8 + k] = id[k];
    }
}

#endif  // U2MKD_DEBUG_PROBE

}  // namespace u2mkd


using namespace u2mkd;

#ifdef U2MKD_DEBUG_PROBE
static unsigned g_probe_launches_host = 0;      // (debug probe: launches so far; the host thread that launches is the only writer)
#endif

#define LAUNCH_ROWS(kernel, n, c, ...)                                                                  \
    do {                                                                                                \
        if ((c) % 4 == 0) {                                                                             \
            int64_t total = (n) * ((c) / 4);                                                            \
            hipLaunchKernelGGL(kernel<4>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), \
                               __VA_ARGS__);                                                            \
        } else {                                                                                        \
            int64_t total = (n) * (int64_t)(c);                                                         \
            hipLaunchKernelGGL(kernel<1>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), \
                               __VA_ARGS__);                                                            \
        }                                                                                               \
    } while (0)

extern "C" {

int u2mkd_count(const int32_t *idx, int64_t n, int32_t *counts, int64_t num, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(idx && counts, "u2mkd_count: null pointer");
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), idx, n, counts,
                       num);
    return check_launch("u2mkd_count");
}

int u2mkd_voxelize_forward(const float *feats, const int32_t *idx, const int32_t *counts, int64_t n, int64_t nv,
                           int32_t c, float *out, u2mkd_stream_t s) {
    if (n == 0 || c == 0) return 0;
    U2_REQUIRE(feats && idx && counts && out, "u2mkd_voxelize_forward: null pointer");
    LAUNCH_ROWS(voxelize_fwd_kernel, n, c, feats, idx, counts, n, nv, c, out);
    return check_launch("u2mkd_voxelize_forward");
}

int u2mkd_voxelize_backward(const float *grad_out, const int32_t *idx, const int32_t *counts, int64_t n, int64_t nv,
                            int32_t c, float *grad_feats, u2mkd_stream_t s) {
    if (n == 0 || c == 0) return 0;
    U2_REQUIRE(grad_out && idx && counts && grad_feats, "u2mkd_voxelize_backward: null pointer");
    LAUNCH_ROWS(voxelize_bwd_kernel, n, c, grad_out, idx, counts, n, nv, c, grad_feats);
    return check_launch("u2mkd_voxelize_backward");
}

int u2mkd_devoxelize_forward(const float *feats, const int32_t *idx, const float *w, int64_t n, int32_t c, float *out,
                             u2mkd_stream_t s) {
    if (n == 0 || c == 0) return 0;
    U2_REQUIRE(feats && idx && w && out, "u2mkd_devoxelize_forward: null pointer");
    LAUNCH_ROWS(devoxelize_fwd_kernel, n, c, feats, idx, w, n, c, out);
    return check_launch("u2mkd_devoxelize_forward");
}

int u2mkd_devoxelize_backward(const float *grad_out, const int32_t *idx, const float *w, int64_t n, int64_t nv,
                              int32_t c, float *grad_feats, u2mkd_stream_t s) {
    if (n == 0 || c == 0) return 0;
    U2_REQUIRE(grad_out && idx && w && grad_feats, "u2mkd_devoxelize_backward: null pointer");
    LAUNCH_ROWS(devoxelize_bwd_kernel, n, c, grad_out, idx, w, n, nv, c, grad_feats);
    return check_launch("u2mkd_devoxelize_backward");
}

int u2mkd_segment_sum(const float *src, int32_t c, const int32_t *entry_row, const float *entry_w,
                      const int32_t *seg_offsets, int64_t nv, int32_t mean, float *out, u2mkd_stream_t s) {
    if (nv == 0 || c == 0) return 0;
    U2_REQUIRE(src && entry_row && seg_offsets && out, "u2mkd_segment_sum: null pointer");
    U2_REQUIRE(c % 4 == 0, "u2mkd_segment_sum: c=%d must be a multiple of 4", c);
    int64_t total = nv * (c / 4);
    hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s), src, c / 4,
                       entry_row, entry_w, seg_offsets, nv, mean, out);
    return check_launch("u2mkd_segment_sum");
}

/* ---- bf16 rows (feature rows in and out are bf16 [., c], c a multiple of 4; indices, weights, counts as above) ---- */
int u2mkd_voxelize_backward_bf16(const void *grad_out, const int32_t *idx, const int32_t *counts, int64_t n, int64_t nv,
                                 int32_t c, void *grad_feats, u2mkd_stream_t s) {
    if (n == 0 || c == 0) return 0;
    U2_REQUIRE(grad_out && idx && counts && grad_feats, "u2mkd_voxelize_backward_bf16: null pointer");
    U2_REQUIRE(c % 4 == 0, "u2mkd_voxelize_backward_bf16: c=%d must be a multiple of 4", c);
    int64_t total = n * (c / 4);
    hipLaunchKernelGGL(voxelize_bwd_rows_kernel<bf16row>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const bf16row *>(grad_out), idx, counts, n, nv, c / 4,
                       reinterpret_cast<bf16row *>(grad_feats));
    return check_launch("u2mkd_voxelize_backward_bf16");
}

int u2mkd_devoxelize_forward_bf16(const void *feats, const int32_t *idx, const float *w, int64_t n, int32_t c, void *out,
                                  u2mkd_stream_t s) {
    if (n == 0 || c == 0) return 0;
    U2_REQUIRE(feats && idx && w && out, "u2mkd_devoxelize_forward_bf16: null pointer");
    U2_REQUIRE(c % 4 == 0, "u2mkd_devoxelize_forward_bf16: c=%d must be a multiple of 4", c);
    int64_t total = n * (c / 4);
    hipLaunchKernelGGL(devoxelize_fwd_rows_kernel<bf16row>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const bf16row *>(feats), idx, w, n, c / 4, reinterpret_cast<bf16row *>(out));
    return check_launch("u2mkd_devoxelize_forward_bf16");
}

int u2mkd_segment_sum_bf16(const void *src, int32_t c, const int32_t *entry_row, const float *entry_w,
                           const int32_t *seg_offsets, int64_t nv, int32_t mean, void *out, u2mkd_stream_t s) {
    if (nv == 0 || c == 0) return 0;
    U2_REQUIRE(src && entry_row && seg_offsets && out, "u2mkd_segment_sum_bf16: null pointer");
    U2_REQUIRE(c % 4 == 0, "u2mkd_segment_sum_bf16: c=%d must be a multiple of 4", c);
    int64_t total = nv * (c / 4);
    hipLaunchKernelGGL(segment_sum_rows_kernel<bf16row>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const bf16row *>(src), c / 4, entry_row, entry_w, seg_offsets, nv, mean,
                       reinterpret_cast<bf16row *>(out));
    return check_launch("u2mkd_segment_sum_bf16");
}

int u2mkd_ti_weights(const float *coords, const int64_t *idx_kn, int64_t n, float scale, float *w_n8,
                     int32_t *idx_n8, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(coords && idx_kn && w_n8 && idx_n8, "u2mkd_ti_weights: null pointer");
#ifdef U2MKD_DEBUG_PROBE
    static const bool probe = [] { const char *e = getenv("U2MKD_DEBUG_TI_PROBE"); return e && e[0] == '1'; }();
    if (probe) {
        hipLaunchKernelGGL(ti_weights_probe_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s),
                           reinterpret_cast<const float4 *>(coords), idx_kn, n, scale, w_n8, idx_n8, g_probe_launches_host++);
        return check_launch("u2mkd_ti_weights(probe)");
    }
#endif
    hipLaunchKernelGGL(ti_weights_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const float4 *>(coords), idx_kn, n, scale, w_n8, idx_n8);
    return check_launch("u2mkd_ti_weights");
}

#ifdef U2MKD_DEBUG_PROBE
// debug: copies the coherence probe's log to the host (synchronises the device); layout = ProbeEntry of csrc/voxel.hip
int u2mkd_debug_probe_read(void *dst, int64_t max_entries, int32_t *n_total, int32_t reset) {
    U2_REQUIRE(dst && n_total, "u2mkd_debug_probe_read: null pointer");
    unsigned int n = 0;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_probe_n), sizeof(n));
    const int64_t take = n < (unsigned)kProbeCap ? n : kProbeCap;
    if (e == hipSuccess && take > 0)
        e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_probe_log), (size_t)(take < max_entries ? take : max_entries) * sizeof(ProbeEntry));
    if (e == hipSuccess && reset) { unsigned int z = 0; e = hipMemcpyToSymbol(HIP_SYMBOL(g_probe_n), &z, sizeof(z)); }
    if (e != hipSuccess) { set_error("u2mkd_debug_probe_read: %s", hipGetErrorString(e)); return 1; }
    *n_total = (int32_t)n;
    return 0;
}
int32_t u2mkd_debug_probe_entry_bytes(void) { return (int32_t)sizeof(ProbeEntry); }
// debug: one launch's per-thread records ([81920] x uint4, see g_row_dbg) of ring slot `slot`
int u2mkd_debug_probe_rows_read(void *dst, int32_t slot) {
    U2_REQUIRE(dst && slot >= 0 && slot < kRowRing, "u2mkd_debug_probe_rows_read: bad argument");
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_row_dbg), sizeof(uint4) * kRowMax, sizeof(uint4) * kRowMax * (size_t)slot);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(reinterpret_cast<char *>(dst) + sizeof(uint4) * kRowMax, HIP_SYMBOL(g_row_dbg2), sizeof(uint4) * kRowMax,
                                                 sizeof(uint4) * kRowMax * (size_t)slot);
    if (e != hipSuccess) { set_error("u2mkd_debug_probe_rows_read: %s", hipGetErrorString(e)); return 1; }
    return 0;
}
// debug: the per-workgroup argument records of the probe launches ([64 launches mod 64][2048 workgroups] x {launch + 1, low 32
// bits of the output pointer}), optionally cleared; *launches = number of probe launches so far
int u2mkd_debug_probe_wg_read(void *dst, int32_t *launches, int32_t reset) {
    U2_REQUIRE(dst && launches, "u2mkd_debug_probe_wg_read: null pointer");
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wg_seen), sizeof(uint2) * kWgRing * kWgMax);
    if (e == hipSuccess && reset) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_wg_seen));
        if (e == hipSuccess) e = hipMemset(p, 0, sizeof(uint2) * kWgRing * kWgMax);
    }
    if (e != hipSuccess) { set_error("u2mkd_debug_probe_wg_read: %s", hipGetErrorString(e)); return 1; }
    *launches = (int32_t)g_probe_launches_host;
    return 0;
}

#endif  // U2MKD_DEBUG_PROBE

}  // extern "C"
