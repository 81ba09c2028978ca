// Destination-grouped entry lists (CSR) for the deterministic scatter replacements: voxelise forward, devoxelise
// backward, LiDAR -> camera pixel means, camera -> LiDAR backward (u2mkd_segment_sum walks them).  torchsparse v1.4.0
// scatters with float atomics (voxelize_forward_cuda / devoxelize_backward_cuda); here every
// destination row sums its entries in ascending entry order, which needs the entries grouped by destination.
//
// Round 2 built the grouping with torch: count + zeros + cumsum + where + a stable argsort (rocPRIM's merge sort below
// its radix threshold: ~13 launches) + a cast -- ~20 launches and 6 Python-level ops per list, ~40 lists per KD step, on
// a step that is bound by the host's launch rate.  The keys are small integers (destination ids < nv), so a counting
// sort does it in ONE call: histogram (integer atomics: exact), exclusive scan, placement with an atomic cursor (the
// order inside a segment is then arbitrary), and a sort of every segment's entry ids (segments are short: a few to a
// few hundred entries) that restores the ascending-entry order -- the result is bit-identical to the stable argsort.
#include "common.h"

namespace u2mkd {

constexpr int kCsrThreads = 256;
constexpr int kScanBlock = 2048;            // elements per scan block (256 threads x 8)

__global__ void __launch_bounds__(kCsrThreads)
csr_count_kernel(const int32_t *__restrict__ keys, int64_t e, int64_t nv, int32_t *__restrict__ counts,
                 int32_t *__restrict__ order) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e) return;
    order[i] = 0;                            // entries past seg[nv] stay a valid index (the placement fills the live ones)
    const int k = keys[i];
    if (k >= 0 && k < nv) atomicAdd(&counts[k], 1);
}

// exclusive scan, two small launches: per-block sums, per-block scan behind the sum of the preceding blocks' sums
__global__ void __launch_bounds__(kCsrThreads)
csr_block_sums_kernel(const int32_t *__restrict__ counts, int64_t nv, int32_t *__restrict__ block_sums) {
    __shared__ int s[kCsrThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanBlock;
    int t = 0;
#pragma unroll
    for (int j = 0; j < kScanBlock / kCsrThreads; ++j) {
        const int64_t i = base + threadIdx.x + j * kCsrThreads;
        t += i < nv ? counts[i] : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_down(t, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(kCsrThreads)
csr_scan_blocks_kernel(const int32_t *__restrict__ counts, int64_t nv, const int32_t *__restrict__ block_sums, int nb,
                       int32_t *__restrict__ seg /*[nv+1]*/, int32_t *__restrict__ cursor) {
    // this block's offset = the sums of the blocks before it (nb <= a few hundred: summed here instead of by a one-workgroup
    // scan launch in between); block 0 also takes the grand total = seg[nv]
    __shared__ int s_pre[kCsrThreads / 64], s_tot[kCsrThreads / 64];
    {
        int pre = 0, tot = 0;
        for (int b = threadIdx.x; b < nb; b += kCsrThreads) {
            const int v = block_sums[b];
            tot += v;
            pre += b < (int)blockIdx.x ? v : 0;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            pre += __shfl_down(pre, off);
            tot += __shfl_down(tot, off);
        }
        if ((threadIdx.x & 63) == 0) { s_pre[threadIdx.x >> 6] = pre; s_tot[threadIdx.x >> 6] = tot; }
        __syncthreads();
    }
    const int block_off = s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
    // thread t owns 8 consecutive elements: local prefix, wave scan, workgroup scan
    __shared__ int s[kCsrThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * 8;
    int v[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = base + j < nv ? counts[base + j] : 0;
        sum += v[j];
    }
    int incl = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s[w];
    int run = block_off + woff + incl - sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < nv) {
            seg[base + j] = run;
            cursor[base + j] = run;          // the placement kernel bumps this copy
        }
        run += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) seg[nv] = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
}

__global__ void __launch_bounds__(kCsrThreads)
csr_place_kernel(const int32_t *__restrict__ keys, int64_t e, int64_t nv, int32_t *__restrict__ cursor,
                 int32_t *__restrict__ order) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e) return;
    const int k = keys[i];
    if (k >= 0 && k < nv) order[atomicAdd(&cursor[k], 1)] = (int32_t)i;
}

// ascending entry ids inside every segment.  One thread per segment for short segments (insertion sort in place), one
// workgroup per long segment (bitonic sort in LDS).
constexpr int kShortSeg = 24;

// One launch: the first `short_blocks` workgroups sort the short segments (a thread per segment), the rest walk the long ones
// (a workgroup per candidate, grid-stride).
constexpr int kLongChunk = 2048;
constexpr int kWaveSeg = 1024;          // entries a single wave sorts in its quarter of the LDS array
__global__ void __launch_bounds__(kCsrThreads)
csr_sort_kernel(const int32_t *__restrict__ seg, int64_t nv, int32_t *__restrict__ order, int32_t *__restrict__ scratch,
                int short_blocks) {
    constexpr int kShared = kShortSeg * kCsrThreads;      // 6 144 ints: the short path's columns; 4 wave strips of kWaveSeg; one chunk of kLongChunk
    static_assert(kShared >= 4 * kWaveSeg && kShared >= kLongChunk, "LDS array too small");
    __shared__ int s_in[kShared];
    if ((int)blockIdx.x < short_blocks) {
        int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (v >= nv) return;
        const int b = seg[v], n = seg[v + 1] - b;
        if (n < 2 || n > kShortSeg) return;
        int32_t *o = order + b;
        // the segment in a private LDS column (entry i of thread t at s_in[i * 256 + t]: conflict-free), insertion sort there:
        // in place in global memory every shift was a dependent load + store through L2 (48 us per launch, 1.6 ms per KD step)
        int *col = s_in + threadIdx.x;
        for (int i = 0; i < n; ++i) col[i * kCsrThreads] = o[i];
        for (int i = 1; i < n; ++i) {
            const int x = col[i * kCsrThreads];
            int j = i - 1;
            while (j >= 0 && col[j * kCsrThreads] > x) { col[(j + 1) * kCsrThreads] = col[j * kCsrThreads]; --j; }
            col[(j + 1) * kCsrThreads] = x;
        }
        for (int i = 0; i < n; ++i) o[i] = col[i * kCsrThreads];
        return;
    }
    const int first = (int)blockIdx.x - short_blocks, stride = (int)gridDim.x - short_blocks;
    // medium segments (25 .. kWaveSeg entries: the voxels of a coarse level in the devoxelisation's backward grouping hold
    // 20-100 corners each): ONE WAVE per segment, bitonic in the wave's own LDS strip -- no workgroup barrier between the
    // stages (a wave's LDS operations are served in issue order), four segments per workgroup at a time.  A workgroup per
    // segment spent its time in 21-28 __syncthreads per 64-128 entries: 1.6 ms per KD step.
    {
        const int wid = threadIdx.x >> 6, wl = threadIdx.x & 63;
        volatile int *ws = s_in + wid * kWaveSeg;
        // (64 segments are LOOKED AT per step, one per lane; walking them one by one -- two dependent loads per segment from a
        // workgroup that mostly finds nothing to do -- took 120 us on the 345 600 pixel segments of the full-resolution grid)
        // wave g of tw looks at segments g, g + tw, g + 2 tw, ...: 64 of them per step (one per lane), so that a run of
        // consecutive medium segments -- a whole coarse level -- spreads over all waves of the launch
        const int64_t gw = (int64_t)first * 4 + wid, tw = (int64_t)stride * 4;
        for (int64_t base = gw; base < nv; base += tw * 64) {
            const int64_t vl = base + (int64_t)wl * tw;
            const int bl = vl < nv ? seg[vl] : 0;
            const int nl = vl < nv ? seg[vl + 1] - bl : 0;
            unsigned long long todo = __ballot(nl > kShortSeg && nl <= kWaveSeg);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int b = __shfl(bl, src), n = __shfl(nl, src);
                int32_t *o = order + b;
                int m = 64;
                while (m < n) m <<= 1;
                for (int i = wl; i < m; i += 64) ws[i] = i < n ? o[i] : 0x7fffffff;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int k = 2; k <= m; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int t = wl; t < (m >> 1); t += 64) {
                            const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                            const int a = ws[lo], b2 = ws[hi];
                            const bool up = (lo & k) == 0;
                            if ((a > b2) == up) { ws[lo] = b2; ws[hi] = a; }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                for (int i = wl; i < n; i += 64) o[i] = ws[i];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x, nl = kCsrThreads;
    __shared__ int s_list[kCsrThreads], s_cnt;
    // the long segments (> kWaveSeg entries), a workgroup each: found 256 at a time, then sorted one after the other
    for (int64_t base = (int64_t)first * kCsrThreads; base < nv; base += (int64_t)stride * kCsrThreads) {
        if (lane == 0) s_cnt = 0;
        __syncthreads();
        {
            const int64_t vl = base + lane;
            if (vl < nv && seg[vl + 1] - seg[vl] > kWaveSeg) s_list[atomicAdd(&s_cnt, 1)] = (int)(vl - base);
        }
        __syncthreads();
        const int cnt = s_cnt;
        for (int qi = 0; qi < cnt; ++qi) {
        const int64_t v = base + s_list[qi];
        const int b = seg[v], n = seg[v + 1] - b;
        int32_t *o = order + b;
        if (n <= kLongChunk) {
            // bitonic sort of the (distinct) entry ids in LDS, padded to a power of two with INT_MAX: n log^2 n / 2
            // compare-exchanges against the n^2 comparisons of ranking by counting (a 1000-entry pixel of a coarse
            // LiDAR -> camera grid: 55 k against 1 M)
            int m = 64;
            while (m < n) m <<= 1;
            for (int i = lane; i < m; i += nl) s_in[i] = i < n ? o[i] : 0x7fffffff;
            __syncthreads();
            for (int k = 2; k <= m; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = lane; t < (m >> 1); t += nl) {
                        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;      // the t-th pair at distance j
                        const int a = s_in[lo], b2 = s_in[hi];
                        const bool up = (lo & k) == 0;
                        if ((a > b2) == up) { s_in[lo] = b2; s_in[hi] = a; }
                    }
                    __syncthreads();
                }
            }
            for (int i = lane; i < n; i += nl) o[i] = s_in[i];
            __syncthreads();
        } else {      // (never on the U2MKD scenes: thousands of entries on one destination) rank against global memory
            int32_t *tmp = scratch + b;
            for (int i = lane; i < n; i += nl) tmp[i] = o[i];
            __syncthreads();
            for (int i = lane; i < n; i += nl) {
                const int x = tmp[i];
                int r = 0;
                for (int j = 0; j < n; ++j) r += tmp[j] < x;
                o[r] = x;
            }
            __syncthreads();
        }
        }
        __syncthreads();
    }
}

}  // namespace u2mkd

namespace u2mkd {

// keys of the trilinear devoxelisation's backward grouping: entry 8 i + j = corner j of point i, keyed by its voxel row or
// dropped (-1) where the corner has no voxel or weight 0
__global__ void devox_keys_kernel(const int32_t *__restrict__ idx, const float *__restrict__ w, int64_t e,
                                  int32_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < e) keys[i] = w[i] != 0.f ? idx[i] : -1;
}

// the grouped entries as (point row, weight): entry order[g] = 8 i + j
__global__ void devox_finish_kernel(const int32_t *__restrict__ order, const float *__restrict__ w, int64_t e,
                                    int32_t *__restrict__ erow, float *__restrict__ ew) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < e) {
        const int32_t o = order[g];
        erow[g] = o >> 3;
        ew[g] = w[o];
    }
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

size_t u2mkd_csr_workspace_bytes(int64_t n_entries, int64_t nv) {
    const int64_t nb = ceil_div(nv > 0 ? nv : 1, kScanBlock);
    return (size_t)(2 * (nv + 1) + nb + 2 + n_entries) * sizeof(int32_t);
}

int u2mkd_csr_build(const int32_t *keys, int64_t n_entries, int64_t nv, void *workspace, int32_t *order, int32_t *seg,
                    u2mkd_stream_t s) {
    U2_REQUIRE(nv >= 0 && n_entries >= 0, "u2mkd_csr_build: negative sizes");
    U2_REQUIRE(seg && workspace && (n_entries == 0 || (keys && order)), "u2mkd_csr_build: null pointer");
    U2_REQUIRE(nv < ((int64_t)1 << 31) - kScanBlock && n_entries < ((int64_t)1 << 31), "u2mkd_csr_build: sizes beyond int32");
    hipStream_t st = as_stream(s);
    if (nv == 0 || n_entries == 0) {
        (void)hipMemsetAsync(seg, 0, (size_t)(nv + 1) * sizeof(int32_t), st);
        return check_launch("u2mkd_csr_build");
    }
    const int nb = (int)ceil_div(nv, kScanBlock);
    int32_t *counts = reinterpret_cast<int32_t *>(workspace);
    int32_t *cursor = counts + (nv + 1);
    int32_t *block_sums = cursor + (nv + 1);
    int32_t *total = block_sums + nb;
    int32_t *scratch = total + 2;
    (void)hipMemsetAsync(counts, 0, (size_t)nv * sizeof(int32_t), st);
    const unsigned ge = (unsigned)ceil_div(n_entries, kCsrThreads);
    hipLaunchKernelGGL(csr_count_kernel, dim3(ge), dim3(kCsrThreads), 0, st, keys, n_entries, nv, counts, order);
    hipLaunchKernelGGL(csr_block_sums_kernel, dim3(nb), dim3(kCsrThreads), 0, st, counts, nv, block_sums);
    hipLaunchKernelGGL(csr_scan_blocks_kernel, dim3(nb), dim3(kCsrThreads), 0, st, counts, nv, block_sums, nb, seg, cursor);
    hipLaunchKernelGGL(csr_place_kernel, dim3(ge), dim3(kCsrThreads), 0, st, keys, n_entries, nv, cursor, order);
    const int gs = (int)ceil_div(nv, kCsrThreads);
    const int gl = (int)std::min<int64_t>(nv, 512);
    hipLaunchKernelGGL(csr_sort_kernel, dim3((unsigned)(gs + gl)), dim3(kCsrThreads), 0, st, seg, nv, order, scratch, gs);
    return check_launch("u2mkd_csr_build");
}

size_t u2mkd_devoxelize_plan_workspace_bytes(int64_t n, int64_t nv) {
    return u2mkd_csr_workspace_bytes(8 * n, nv) + (size_t)16 * n * sizeof(int32_t);
}

int u2mkd_devoxelize_plan(const int32_t *idx8, const float *w8, int64_t n, int64_t nv, void *workspace, int32_t *entry_row,
                          float *entry_w, int32_t *seg, u2mkd_stream_t s) {
    U2_REQUIRE(n >= 0 && nv >= 0, "u2mkd_devoxelize_plan: negative sizes");
    U2_REQUIRE(seg && workspace && (n == 0 || (idx8 && w8 && entry_row && entry_w)), "u2mkd_devoxelize_plan: null pointer");
    const int64_t e = 8 * n;
    if (nv == 0 || e == 0) {     // no voxel rows (or no points): every segment is empty, `order` is never written -- nothing to gather
        (void)hipMemsetAsync(seg, 0, (size_t)(nv + 1) * sizeof(int32_t), as_stream(s));
        if (e) {                 // (the entry arrays are defined: row 0, weight 0; no segment addresses them)
            (void)hipMemsetAsync(entry_row, 0, (size_t)e * sizeof(int32_t), as_stream(s));
            (void)hipMemsetAsync(entry_w, 0, (size_t)e * sizeof(float), as_stream(s));
        }
        return check_launch("u2mkd_devoxelize_plan");
    }
    int32_t *keys = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(workspace) + u2mkd_csr_workspace_bytes(e, nv));
    int32_t *order = keys + e;
    hipStream_t st = as_stream(s);
    if (e) hipLaunchKernelGGL(devox_keys_kernel, dim3((unsigned)ceil_div(e, 256)), dim3(256), 0, st, idx8, w8, e, keys);
    int rc = u2mkd_csr_build(keys, e, nv, workspace, order, seg, s);
    if (rc) return rc;
    if (e) hipLaunchKernelGGL(devox_finish_kernel, dim3((unsigned)ceil_div(e, 256)), dim3(256), 0, st, order, w8, e, entry_row, entry_w);
    return check_launch("u2mkd_devoxelize_plan");
}

}  // extern "C"
