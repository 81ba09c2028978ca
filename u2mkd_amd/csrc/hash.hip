// Coordinate hashing, device hash table, kernel-map (rulebook) construction.
// Replaces torchsparse v1.4.0 hash_cuda / kernel_hash_cuda / hash_query_cuda
// and the python kmap build of F.conv3d.
#include <mutex>
#include <map>
#include <unordered_map>
#include <stdarg.h>

#include <stdlib.h>

#include "common.h"

namespace u2mkd {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ void hash_kernel(const int4 *__restrict__ coords, int64_t n, int64_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int4 c = coords[i];
        out[i] = fnv_hash4(c.x, c.y, c.z, c.w);
    }
}

// grid.y = kernel offset (wave-uniform), threads over voxels: coalesced int4
// reads, coalesced int64 writes into out[k][i].
__global__ void kernel_hash_kernel(const int4 *__restrict__ coords, const int32_t *__restrict__ offsets,
                                   int64_t n, int64_t *__restrict__ out) {
    int k = blockIdx.y;
    int ox = offsets[3 * k], oy = offsets[3 * k + 1], oz = offsets[3 * k + 2];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int4 c = coords[i];
        out[(int64_t)k * n + i] = fnv_hash4(c.x + ox, c.y + oy, c.z + oz, c.w);
    }
}

__global__ void table_insert_kernel(TableView t, const int64_t *__restrict__ refs, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t key = refs[i];
    uint32_t slot = table_slot(t, key);
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&t.keys[slot]),
                                            (unsigned long long)(-1LL), (unsigned long long)key);
        if (prev == (unsigned long long)(-1LL) || prev == (unsigned long long)key) {
            atomicMin(&t.vals[slot], (int)i);   // duplicates: smallest index wins
            return;
        }
        slot = (slot + 1) & t.mask;
    }
}

__global__ void table_query_kernel(TableView t, const int64_t *__restrict__ q, int64_t n,
                                   int64_t *__restrict__ out, int32_t *__restrict__ out32 = nullptr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t v = (int64_t)table_lookup(t, q[i]);
        out[i] = v;
        if (out32) out32[i] = (int32_t)v;
    }
}

// Fused kernel_hash + query: nbr[k][j] = index of (out_coords[j] + offsets[k]) or -1.
__global__ void kmap_table_kernel(TableView t, const int4 *__restrict__ out_coords,
                                  const int32_t *__restrict__ offsets, int64_t n_out,
                                  int32_t *__restrict__ nbr) {
    int k = blockIdx.y;
    int ox = offsets[3 * k], oy = offsets[3 * k + 1], oz = offsets[3 * k + 2];
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_out) {
        int4 c = out_coords[j];
        int64_t key = fnv_hash4(c.x + ox, c.y + oy, c.z + oz, c.w);
        nbr[(int64_t)k * n_out + j] = table_lookup(t, key);
    }
}

__global__ void kmap_invert_kernel(const int32_t *__restrict__ nbr, int64_t n_out, int64_t n_in,
                                   int32_t *__restrict__ inv) {
    int k = blockIdx.y;
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_out) {
        int i = nbr[(int64_t)k * n_out + j];
        if (i >= 0 && i < n_in) inv[(int64_t)k * n_in + i] = (int)j;
    }
}

__global__ void kmap_rowmask_kernel(const int32_t *__restrict__ nbr, int64_t n_out, int k_total,
                                    int32_t *__restrict__ mask) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    unsigned m = 0u;
    for (int k = 0; k < k_total; ++k)
        if (nbr[(int64_t)k * n_out + j] >= 0) m |= 1u << k;
    mask[j] = (int32_t)m;
}

// ---- rulebook compaction (torchsparse nbmaps order: by k, ascending out) ----
// 1024 rows per block; wave ballots give the in-block rank of every valid pair.
constexpr int kCompactBlock = 1024;

__global__ void __launch_bounds__(kCompactBlock)
kmap_sizes_kernel(const int32_t *__restrict__ nbr, int64_t n_out, int32_t *__restrict__ nbsizes,
                  int32_t *__restrict__ block_counts) {
    __shared__ int wave_cnt[kCompactBlock / kWave];
    int k = blockIdx.y;
    int64_t j = (int64_t)blockIdx.x * kCompactBlock + threadIdx.x;
    bool valid = (j < n_out) && nbr[(int64_t)k * n_out + j] >= 0;
    unsigned long long m = __ballot(valid);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < kCompactBlock / kWave; ++w) tot += wave_cnt[w];
        block_counts[(int64_t)k * gridDim.x + blockIdx.x] = tot;
        if (tot) atomicAdd(&nbsizes[k], tot);
    }
}

// One block: exclusive scan of block_counts in (k, block) order, offset by the
// prefix of nbsizes over k.  K * nblocks is small (27 * 300 for 300k voxels).
__global__ void kmap_scan_kernel(const int32_t *__restrict__ nbsizes, int32_t *__restrict__ block_counts,
                                 int k_total, int nblocks) {
    __shared__ int kbase[256];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < k_total; ++k) { kbase[k] = acc; acc += nbsizes[k]; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < k_total; k += blockDim.x) {
        int acc = kbase[k];
        int32_t *row = block_counts + (int64_t)k * nblocks;
        for (int b = 0; b < nblocks; ++b) { int c = row[b]; row[b] = acc; acc += c; }
    }
}

__global__ void __launch_bounds__(kCompactBlock)
kmap_compact_kernel(const int32_t *__restrict__ nbr, int64_t n_out, const int32_t *__restrict__ block_base,
                    int32_t *__restrict__ nbmaps) {
    __shared__ int wave_cnt[kCompactBlock / kWave];
    int k = blockIdx.y;
    int64_t j = (int64_t)blockIdx.x * kCompactBlock + threadIdx.x;
    int i = (j < n_out) ? nbr[(int64_t)k * n_out + j] : -1;
    bool valid = i >= 0;
    unsigned long long m = __ballot(valid);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    if (valid) {
        int base = block_base[(int64_t)k * gridDim.x + blockIdx.x];
        for (int w = 0; w < wave; ++w) base += wave_cnt[w];
        int rank = __popcll(m & ((1ULL << lane) - 1ULL));
        int64_t p = (int64_t)base + rank;
        nbmaps[2 * p] = i;
        nbmaps[2 * p + 1] = (int)j;
    }
}


// ---- pair schedule: offset-grouped, 128-padded pair list + slot tables -----------------
// (128 = two 64-pair tiles: conv_px3_kernel multiplies two tiles of ONE offset per step, conv_px3.hip TL)
// One block: kbase[k] = padded prefix (each offset's pair count rounded up to 128), block bases
// in (k, block) order, tile_k for every 64-entry tile, -1 into the padding entries, and
// meta = {P_pad, n_tiles}.  Nothing of this is read back by the host.
__global__ void pairs_scan_kernel(const int32_t *__restrict__ nbsizes, int32_t *__restrict__ block_counts, int k_total,
                                  int nblocks, int32_t *__restrict__ pair_in, int32_t *__restrict__ pair_out,
                                  int32_t *__restrict__ tile_k, int32_t *__restrict__ meta) {
    __shared__ int kbase[257];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < k_total; ++k) { kbase[k] = acc; acc += (nbsizes[k] + 127) / 128 * 128; }
        kbase[k_total] = acc;
        meta[0] = acc;
        meta[1] = acc / 64;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < k_total; k += blockDim.x) {
        int acc = kbase[k];
        int32_t *row = block_counts + (int64_t)k * nblocks;
        for (int b = 0; b < nblocks; ++b) { int c = row[b]; row[b] = acc; acc += c; }
    }
    for (int k = 0; k < k_total; ++k) {
        for (int t = kbase[k] / 64 + threadIdx.x; t < kbase[k + 1] / 64; t += blockDim.x) tile_k[t] = k;
        for (int p = kbase[k] + nbsizes[k] + threadIdx.x; p < kbase[k + 1]; p += blockDim.x) {
            pair_in[p] = -1;
            pair_out[p] = -1;
        }
    }
}

__global__ void __launch_bounds__(kCompactBlock)
pairs_build_kernel(const int32_t *__restrict__ nbr, int64_t n_out, int K, const int32_t *__restrict__ block_base,
                   int32_t *__restrict__ pair_in, int32_t *__restrict__ pair_out, int32_t *__restrict__ pos_out,
                   int32_t *__restrict__ pos_in) {
    __shared__ int wave_cnt[kCompactBlock / kWave];
    int k = blockIdx.y;
    int64_t j = (int64_t)blockIdx.x * kCompactBlock + threadIdx.x;
    int i = (j < n_out) ? nbr[(int64_t)k * n_out + j] : -1;
    bool valid = i >= 0;
    unsigned long long m = __ballot(valid);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int p = -1;
    if (valid) {
        int base = block_base[(int64_t)k * gridDim.x + blockIdx.x];
        for (int w = 0; w < wave; ++w) base += wave_cnt[w];
        p = base + __popcll(m & ((1ULL << lane) - 1ULL));
        pair_in[p] = i;
        pair_out[p] = (int)j;
        pos_in[(int64_t)i * K + k] = p;
    }
    if (j < n_out) pos_out[j * K + k] = p;
}

// ---- downsample keys: order-preserving (b,x,y,z) pack -----------------------
// b: 10 bits, x/y/z: 18 bits each with bias 2^17 (|coord| < 131072).
// A coordinate outside the packed range (the reference's torch.unique(dim=0) has none) would alias into another
// voxel: such a row gets the key INT64_MAX and raises *range_flag, which the caller reads after its sort.
__global__ void downsample_keys_kernel(const int4 *__restrict__ coords, int64_t n, int sx, int sy, int sz,
                                       int64_t *__restrict__ keys, int32_t *__restrict__ range_flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    auto fl = [](int v, int s) { int q = v / s; if ((v % s != 0) && ((v < 0) != (s < 0))) --q; return q * s; };
    int x = fl(c.x, sx), y = fl(c.y, sy), z = fl(c.z, sz);
    const int64_t bias = 1 << 17;
    const int lim = 1 << 17;
    const bool ok = x >= -lim && x < lim && y >= -lim && y < lim && z >= -lim && z < lim && c.w >= 0 && c.w < 512;
    if (!ok) {
        keys[i] = INT64_MAX;
        if (range_flag) *range_flag = 1;
        return;
    }
    keys[i] = ((int64_t)c.w << 54) | ((int64_t)(x + bias) << 36) | ((int64_t)(y + bias) << 18) | (int64_t)(z + bias);
}

__global__ void table_clear_kernel(TableView t, int64_t cap) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    t.keys[i] = -1;
    t.vals[i] = 0x7F7F7F7F;
}

// int32 (floor(x / s) * s, floor(y / s) * s, floor(z / s) * s, (int)b) of float point coordinates (x, y, z, b):
// the voxel a point falls into at tensor stride s (core/models/utils.py:43-47,86-90: torch.floor(z.C[:, :3] / s) * s
// concatenated with the batch column and cast to int -- seven element-wise launches in torch)
__global__ void floor_coords_kernel(const float4 *__restrict__ pc, int64_t n, int stride, int4 *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = pc[i];
    const float s = (float)stride;
    int4 o;
    o.x = (int)floorf(c.x / s) * stride;
    o.y = (int)floorf(c.y / s) * stride;
    o.z = (int)floorf(c.z / s) * stride;
    o.w = (int)c.w;
    out[i] = o;
}

__global__ void unpack_keys_kernel(const int64_t *__restrict__ keys, int64_t n, int4 *__restrict__ coords) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t key = keys[i];
    const int64_t bias = 1 << 17, m18 = (1 << 18) - 1;
    int4 c;
    c.w = (int)(key >> 54);
    c.x = (int)(((key >> 36) & m18) - bias);
    c.y = (int)(((key >> 18) & m18) - bias);
    c.z = (int)((key & m18) - bias);
    coords[i] = c;
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

int u2mkd_version(void) { return 100; }
const char *u2mkd_last_error(void) { return g_err; }

/* `waiter` continues only behind everything queued on `signaler` so far: one event record + one stream wait (what
 * torch.cuda.Stream.wait_stream does through three Python calls and a new event object each time -- the trainers order a side
 * stream behind the caller's ~130 times per backward pass).  The event objects are made once per waiter stream and kept; a later
 * record on the same event does not disturb a wait already queued (hipStreamWaitEvent takes the record that precedes it). */
int u2mkd_stream_wait_stream(u2mkd_stream_t waiter, u2mkd_stream_t signaler) {
    static std::mutex mu;
    // one event per (waiter stream, device of the signaler): an event records only on streams of the device it was made on
    static std::map<std::pair<hipStream_t, int>, hipEvent_t> events;
    hipStream_t w = as_stream(waiter), sg = as_stream(signaler);
    if (w == sg) return 0;
    int dev = 0, cur = 0;
    if (hipStreamGetDevice(sg, &dev) != hipSuccess || hipGetDevice(&cur) != hipSuccess) {
        set_error("u2mkd_stream_wait_stream: %s", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    hipStreamCaptureStatus cap_w = hipStreamCaptureStatusNone, cap_s = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(w, &cap_w);
    (void)hipStreamIsCapturing(sg, &cap_s);
    const bool capturing = cap_w != hipStreamCaptureStatusNone || cap_s != hipStreamCaptureStatusNone;
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = capturing ? events.end() : events.find({w, dev});
        if (it == events.end()) {
            // (a cached event must not be re-recorded inside a stream capture: a capture gets an event of its own)
            if (cur != dev) (void)hipSetDevice(dev);
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (cur != dev) (void)hipSetDevice(cur);
            if (e != hipSuccess) {
                set_error("u2mkd_stream_wait_stream: hipEventCreateWithFlags failed: %s", hipGetErrorString(e));
                return 1;
            }
            if (!capturing) events.emplace(std::make_pair(w, dev), ev);
        } else {
            ev = it->second;
        }
        // (record + wait under the lock: two threads ordering the same waiter must not interleave their record / wait pairs)
        hipError_t e = hipEventRecord(ev, sg);
        if (e == hipSuccess) e = hipStreamWaitEvent(w, ev, 0);
        if (capturing) (void)hipEventDestroy(ev);      // (released once the recorded work has completed)
        if (e != hipSuccess) {
            set_error("u2mkd_stream_wait_stream: %s", hipGetErrorString(e));
            return 1;
        }
    }
    return 0;
}

int u2mkd_hash(const int32_t *coords, int64_t n, int64_t *out, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(coords && out, "u2mkd_hash: null pointer");
    hipLaunchKernelGGL(hash_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const int4 *>(coords), n, out);
    return check_launch("u2mkd_hash");
}

int u2mkd_kernel_hash(const int32_t *coords, const int32_t *offsets, int64_t n, int32_t k, int64_t *out,
                      u2mkd_stream_t s) {
    if (n == 0 || k == 0) return 0;
    U2_REQUIRE(coords && offsets && out, "u2mkd_kernel_hash: null pointer");
    U2_REQUIRE(k <= 65535, "u2mkd_kernel_hash: k=%d too large", k);
    hipLaunchKernelGGL(kernel_hash_kernel, dim3((unsigned)ceil_div(n, 256), k), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const int4 *>(coords), offsets, n, out);
    return check_launch("u2mkd_kernel_hash");
}

size_t u2mkd_hash_table_bytes(int64_t n_refs) {
    return (size_t)table_capacity(n_refs) * (sizeof(int64_t) + sizeof(int32_t));
}

int u2mkd_hash_table_build(const int64_t *refs, int64_t n_refs, void *table, u2mkd_stream_t s) {
    U2_REQUIRE(table, "u2mkd_hash_table_build: null table");
    U2_REQUIRE(n_refs < (1LL << 31) - 1, "u2mkd_hash_table_build: too many refs");
    TableView t = make_table_view(table, n_refs);
    int64_t cap = table_capacity(n_refs);
    // (debug, NOTES N9: U2MKD_DEBUG_HASH_FILL=1 clears the table with a kernel of this library instead of the runtime's
    // two memset commands -- one of the discriminators of the stale-read item)
    static const bool fill_kernel = [] { const char *e = getenv("U2MKD_DEBUG_HASH_FILL"); return e && e[0] == '1'; }();
    if (fill_kernel) {
        hipLaunchKernelGGL(table_clear_kernel, dim3((unsigned)ceil_div(cap, 256)), dim3(256), 0, as_stream(s), t, cap);
        if (check_launch("u2mkd_hash_table_build(clear)")) return 1;
    } else {
        hipError_t e = hipMemsetAsync(t.keys, 0xFF, cap * sizeof(int64_t), as_stream(s));
        if (e == hipSuccess) e = hipMemsetAsync(t.vals, 0x7F, cap * sizeof(int32_t), as_stream(s));
        if (e != hipSuccess) { set_error("u2mkd_hash_table_build: memset: %s", hipGetErrorString(e)); return 1; }
    }
    if (n_refs == 0) return 0;
    U2_REQUIRE(refs, "u2mkd_hash_table_build: null refs");
    hipLaunchKernelGGL(table_insert_kernel, dim3((unsigned)ceil_div(n_refs, 256)), dim3(256), 0, as_stream(s), t,
                       refs, n_refs);
    return check_launch("u2mkd_hash_table_build");
}

int u2mkd_hash_table_query(const void *table, int64_t n_refs, const int64_t *queries, int64_t n_q, int64_t *out,
                           u2mkd_stream_t s) {
    if (n_q == 0) return 0;
    U2_REQUIRE(table && queries && out, "u2mkd_hash_table_query: null pointer");
    TableView t = make_table_view(const_cast<void *>(table), n_refs);
    hipLaunchKernelGGL(table_query_kernel, dim3((unsigned)ceil_div(n_q, 256)), dim3(256), 0, as_stream(s), t,
                       queries, n_q, out);
    return check_launch("u2mkd_hash_table_query");
}

int u2mkd_hash_table_query2(const void *table, int64_t n_refs, const int64_t *queries, int64_t n_q, int64_t *out,
                            int32_t *out32, u2mkd_stream_t s) {
    if (n_q == 0) return 0;
    U2_REQUIRE(table && queries && out && out32, "u2mkd_hash_table_query2: null pointer");
    TableView t = make_table_view(const_cast<void *>(table), n_refs);
    hipLaunchKernelGGL(table_query_kernel, dim3((unsigned)ceil_div(n_q, 256)), dim3(256), 0, as_stream(s), t,
                       queries, n_q, out, out32);
    return check_launch("u2mkd_hash_table_query2");
}

int u2mkd_kmap_build_table(const void *table, int64_t n_refs, const int32_t *out_coords, int64_t n_out,
                           const int32_t *offsets, int32_t k, int32_t *nbr, u2mkd_stream_t s) {
    if (n_out == 0 || k == 0) return 0;
    U2_REQUIRE(table && out_coords && offsets && nbr, "u2mkd_kmap_build_table: null pointer");
    U2_REQUIRE(k <= 65535, "u2mkd_kmap_build_table: k=%d too large", k);
    TableView t = make_table_view(const_cast<void *>(table), n_refs);
    hipLaunchKernelGGL(kmap_table_kernel, dim3((unsigned)ceil_div(n_out, 256), k), dim3(256), 0, as_stream(s), t,
                       reinterpret_cast<const int4 *>(out_coords), offsets, n_out, nbr);
    return check_launch("u2mkd_kmap_build_table");
}

int u2mkd_kmap_invert(const int32_t *nbr, int64_t n_out, int32_t k, int64_t n_in, int32_t *nbr_inv,
                      u2mkd_stream_t s) {
    if (n_out == 0 || k == 0) return 0;
    U2_REQUIRE(nbr && nbr_inv, "u2mkd_kmap_invert: null pointer");
    hipLaunchKernelGGL(kmap_invert_kernel, dim3((unsigned)ceil_div(n_out, 256), k), dim3(256), 0, as_stream(s), nbr,
                       n_out, n_in, nbr_inv);
    return check_launch("u2mkd_kmap_invert");
}

int u2mkd_kmap_rowmask(const int32_t *nbr, int64_t n_out, int32_t k, int32_t *mask, u2mkd_stream_t s) {
    if (n_out == 0) return 0;
    U2_REQUIRE(nbr && mask, "u2mkd_kmap_rowmask: null pointer");
    U2_REQUIRE(k > 0 && k <= 31, "u2mkd_kmap_rowmask: kernel volume %d not in 1..31", k);
    hipLaunchKernelGGL(kmap_rowmask_kernel, dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0, as_stream(s), nbr,
                       n_out, k, mask);
    return check_launch("u2mkd_kmap_rowmask");
}

int u2mkd_kmap_sizes(const int32_t *nbr, int64_t n_out, int32_t k, int32_t *nbsizes, int32_t *block_counts,
                     u2mkd_stream_t s) {
    if (n_out == 0 || k == 0) return 0;
    U2_REQUIRE(nbr && nbsizes && block_counts, "u2mkd_kmap_sizes: null pointer");
    hipLaunchKernelGGL(kmap_sizes_kernel, dim3((unsigned)ceil_div(n_out, kCompactBlock), k), dim3(kCompactBlock), 0,
                       as_stream(s), nbr, n_out, nbsizes, block_counts);
    return check_launch("u2mkd_kmap_sizes");
}

int u2mkd_kmap_compact(const int32_t *nbr, int64_t n_out, int32_t k, const int32_t *nbsizes, int32_t *block_counts,
                       int32_t *nbmaps, u2mkd_stream_t s) {
    if (n_out == 0 || k == 0) return 0;
    U2_REQUIRE(nbr && nbsizes && block_counts && nbmaps, "u2mkd_kmap_compact: null pointer");
    U2_REQUIRE(k <= 256, "u2mkd_kmap_compact: k=%d > 256", k);
    int nblocks = (int)ceil_div(n_out, kCompactBlock);
    hipLaunchKernelGGL(kmap_scan_kernel, dim3(1), dim3(64), 0, as_stream(s), nbsizes, block_counts, k, nblocks);
    hipLaunchKernelGGL(kmap_compact_kernel, dim3(nblocks, k), dim3(kCompactBlock), 0, as_stream(s), nbr, n_out,
                       block_counts, nbmaps);
    return check_launch("u2mkd_kmap_compact");
}

int64_t u2mkd_pairs_capacity(int64_t n_in, int64_t n_out, int32_t k) {
    int64_t m = n_in < n_out ? n_in : n_out;
    return ((int64_t)k * m + 127) / 128 * 128 + 128 * (int64_t)k;
}

int u2mkd_pairs_build(const int32_t *nbr, int64_t n_out, int64_t n_in, int32_t k, const int32_t *nbsizes,
                      int32_t *block_counts, int32_t *pair_in, int32_t *pair_out, int32_t *pos_out, int32_t *pos_in,
                      int32_t *tile_k, int32_t *meta, u2mkd_stream_t s) {
    U2_REQUIRE(meta, "u2mkd_pairs_build: null pointer");
    if (n_out == 0 || k == 0) {
        (void)hipMemsetAsync(meta, 0, 2 * sizeof(int32_t), as_stream(s));
        return 0;
    }
    U2_REQUIRE(nbr && nbsizes && block_counts && pair_in && pair_out && pos_out && pos_in && tile_k,
               "u2mkd_pairs_build: null pointer");
    U2_REQUIRE(k <= 256 && n_in > 0, "u2mkd_pairs_build: k=%d > 256 or no input rows", k);
    int nblocks = (int)ceil_div(n_out, kCompactBlock);
    hipLaunchKernelGGL(pairs_scan_kernel, dim3(1), dim3(256), 0, as_stream(s), nbsizes, block_counts, k, nblocks, pair_in,
                       pair_out, tile_k, meta);
    hipLaunchKernelGGL(pairs_build_kernel, dim3(nblocks, k), dim3(kCompactBlock), 0, as_stream(s), nbr, n_out, k,
                       block_counts, pair_in, pair_out, pos_out, pos_in);
    return check_launch("u2mkd_pairs_build");
}

int u2mkd_downsample_keys_checked(const int32_t *coords, int64_t n, int32_t sx, int32_t sy, int32_t sz, int64_t *keys,
                                  int32_t *range_flag, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(coords && keys, "u2mkd_downsample_keys: null pointer");
    U2_REQUIRE(sx > 0 && sy > 0 && sz > 0, "u2mkd_downsample_keys: strides must be positive");
    hipLaunchKernelGGL(downsample_keys_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const int4 *>(coords), n, sx, sy, sz, keys, range_flag);
    return check_launch("u2mkd_downsample_keys");
}

int u2mkd_downsample_keys(const int32_t *coords, int64_t n, int32_t sx, int32_t sy, int32_t sz, int64_t *keys,
                          u2mkd_stream_t s) {
    return u2mkd_downsample_keys_checked(coords, n, sx, sy, sz, keys, nullptr, s);
}

int u2mkd_floor_coords(const float *pc, int64_t n, int32_t stride, int32_t *out, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(pc && out, "u2mkd_floor_coords: null pointer");
    U2_REQUIRE(stride > 0, "u2mkd_floor_coords: stride must be positive");
    hipLaunchKernelGGL(floor_coords_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<const float4 *>(pc), n, stride, reinterpret_cast<int4 *>(out));
    return check_launch("u2mkd_floor_coords");
}

int u2mkd_unpack_keys(const int64_t *keys, int64_t n, int32_t *coords, u2mkd_stream_t s) {
    if (n == 0) return 0;
    U2_REQUIRE(coords && keys, "u2mkd_unpack_keys: null pointer");
    hipLaunchKernelGGL(unpack_keys_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), keys, n,
                       reinterpret_cast<int4 *>(coords));
    return check_launch("u2mkd_unpack_keys");
}

}  // extern "C"
