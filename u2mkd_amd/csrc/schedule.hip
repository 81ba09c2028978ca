// Tile schedule of a kernel map, built on the device in two launches (gfx950).
//
// The sparse-conv tile kernels (conv_tp.hip, conv.hip) walk 64-row tiles of the MASK-SORTED neighbour table
// heaviest first, tiles with many MFMA blocks cut into halves / quarters (work items).  The schedule belongs to
// the kernel map (the nbmaps / nbsizes cache of torchsparse v1.4.0, built once per stride and shared by the 8-9
// convs of a stage, core/models/utils.py:60-61), but a training step builds ~30 of them (two models, five
// strides, forward and inverse tables) and composing one from torch operators cost ~35 tiny launches each:
// 20 % of the step's launches.  Here:
//   tile_weights_kernel  one wave per tile: 27 ballots over the rows' neighbour masks -> MFMA blocks of the
//                        whole tile, its halves and its quarters;
//   tile_sort_kernel     ONE workgroup: two stable counting sorts (keys = block counts, <= 129 values) --
//                        the tiles by descending blocks, and the 7 item candidates per tile (whole, 2 halves,
//                        4 quarters; the ones not chosen sort to the end) -- every wave owns a contiguous
//                        segment of the list, ranks its 64-element groups with ballots, and scatters without
//                        inter-wave synchronisation; the number of live items stays on the device.
// Same order as the torch formulation it replaces (stable argsort by descending weight of the candidate list
// [wholes | halves | quarters]), no host synchronisation.
#include "common.h"

namespace u2mkd {

constexpr int kSortThreads = 1024, kSortWaves = kSortThreads / 64, kSortKeys = 132;

// tw[t][8] = {b1, b2[0], b2[1], b4[0..3], 0}: 16-pair MFMA blocks (sum over offsets of ceil(pairs / 16)) of the
// tile, its 32-row halves and its 16-row quarters
__global__ void __launch_bounds__(64)
tile_weights_kernel(const int32_t *__restrict__ mask, const int32_t *__restrict__ order, int64_t n, int K,
                    int32_t *__restrict__ tw) {
    const int64_t t = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t row = t * 64 + lane;
    const unsigned m = row < n ? (unsigned)mask[order ? order[row] : row] : 0u;
    int b1 = 0, b2[2] = {0, 0}, b4[4] = {0, 0, 0, 0};
    for (int k = 0; k < K; ++k) {
        const unsigned long long bal = __ballot((m >> k) & 1u);
        int c[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) c[qd] = __popcll(bal & (0xFFFFULL << (16 * qd)));
        b1 += (c[0] + c[1] + c[2] + c[3] + 15) >> 4;
        b2[0] += (c[0] + c[1] + 15) >> 4;
        b2[1] += (c[2] + c[3] + 15) >> 4;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) b4[qd] += c[qd] > 0;
    }
    if (lane < 8) {
        const int v = lane == 0 ? b1 : lane < 3 ? b2[lane - 1] : lane < 7 ? b4[lane - 3] : 0;
        tw[t * 8 + lane] = v;
    }
}

// stable counting sort of `count` elements (key(e) in [0, kSortKeys), value(e)) by ascending key into out[];
// one workgroup, the waves own contiguous segments of the element range
template <typename KeyFn, typename ValFn>
__device__ __forceinline__ void block_counting_sort(int count, KeyFn key, ValFn val, int32_t *__restrict__ out,
                                                    int (*hist)[kSortKeys], int *start) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < kSortWaves * kSortKeys; e += kSortThreads) (&hist[0][0])[e] = 0;
    __syncthreads();
    const int seg = ((count + kSortWaves - 1) / kSortWaves + 63) / 64 * 64;     // elements per wave, whole groups
    const int lo = wave * seg, hi = min(lo + seg, count);
    for (int e = lo + lane; e < hi; e += 64) atomicAdd(&hist[wave][key(e)], 1);
    __syncthreads();
    // start[k] = elements with a smaller key; hist[w][k] becomes wave w's first slot of key k
    if (tid < kSortKeys) {
        int tot = 0;
        for (int w = 0; w < kSortWaves; ++w) tot += hist[w][tid];
        start[tid] = tot;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k < kSortKeys; ++k) { const int c = start[k]; start[k] = run; run += c; }
    }
    __syncthreads();
    if (tid < kSortKeys) {
        int run = start[tid];
        for (int w = 0; w < kSortWaves; ++w) { const int c = hist[w][tid]; hist[w][tid] = run; run += c; }
    }
    __syncthreads();
    // every wave scatters its segment in order; the rank inside a 64-element group comes from 8 ballots
    for (int e0 = lo; e0 < hi; e0 += 64) {
        const int e = e0 + lane;
        const bool live = e < hi;
        const int k = live ? key(e) : 255;
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((k >> b) & 1);
            same &= ((k >> b) & 1) ? bal : ~bal;
        }
        if (live) {
            const int rank = __popcll(same & ((1ULL << lane) - 1ULL));
            out[hist[wave][k] + rank] = val(e);
        }
        // the group's last element of each key advances the wave's slot (LDS operations of one wave are ordered)
        if (live && (same >> lane) == 1ULL) hist[wave][k] += __popcll(same);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kSortThreads)
tile_sort_kernel(const int32_t *__restrict__ tw, int64_t n, int t, int split0, int split1,
                 int32_t *__restrict__ tile_order, int32_t *__restrict__ items, int32_t *__restrict__ n_items) {
    __shared__ int hist[kSortWaves][kSortKeys];
    __shared__ int start[kSortKeys];
    constexpr int KMAX = 128;          // blocks of a tile <= 32 offsets x 4
    // 1. tiles by descending block count
    block_counting_sort(t, [&](int e) { return KMAX - min(tw[(int64_t)e * 8], KMAX); }, [&](int e) { return e; },
                        tile_order, hist, start);
    // 2. item candidates [t wholes | 2t halves | 4t quarters]; a tile's split level lg = (b1 > split0) + (b1 > split1)
    // picks which of its candidates live; dead ones (and parts without rows) take the last key
    auto cand = [&](int e, int &tile, int &sub, int &l) {
        if (e < t) { tile = e; sub = 0; l = 0; }
        else if (e < 3 * t) { tile = (e - t) >> 1; sub = (e - t) & 1; l = 1; }
        else { tile = (e - 3 * t) >> 2; sub = (e - 3 * t) & 3; l = 2; }
    };
    auto ckey = [&](int e) {
        int tile, sub, l;
        cand(e, tile, sub, l);
        const int b1 = tw[(int64_t)tile * 8];
        const int lg = (b1 > split0) + (b1 > split1);
        const int64_t rows_left = n - (int64_t)tile * 64;
        const bool live = lg == l && (int64_t)sub * (64 >> l) < rows_left;
        const int w = l == 0 ? b1 : l == 1 ? tw[(int64_t)tile * 8 + 1 + sub] : tw[(int64_t)tile * 8 + 3 + sub];
        return live ? KMAX - min(w, KMAX) : KMAX + 1;
    };
    auto cval = [&](int e) {
        int tile, sub, l;
        cand(e, tile, sub, l);
        return (tile << 4) | (sub << 2) | l;
    };
    block_counting_sort(7 * t, ckey, cval, items, hist, start);
    if (threadIdx.x == 0) *n_items = start[KMAX + 1];      // elements with a key below the dead one
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

size_t u2mkd_tile_schedule_workspace_bytes(int64_t n_rows) { return (size_t)ceil_div(n_rows, 64) * 8 * sizeof(int32_t); }

int u2mkd_tile_schedule(const int32_t *mask, const int32_t *order, int64_t n_rows, int32_t k, int32_t split0,
                        int32_t split1, void *workspace, int32_t *tile_order, int32_t *items, int32_t *n_items,
                        u2mkd_stream_t s) {
    U2_REQUIRE(n_items, "u2mkd_tile_schedule: null pointer");
    if (n_rows <= 0) {
        (void)hipMemsetAsync(n_items, 0, sizeof(int32_t), as_stream(s));
        return check_launch("u2mkd_tile_schedule");
    }
    U2_REQUIRE(mask && workspace && tile_order && items, "u2mkd_tile_schedule: null pointer");
    U2_REQUIRE(k > 0 && k <= 32, "u2mkd_tile_schedule: kernel volume %d not in 1..32", k);
    U2_REQUIRE(split0 >= 0 && split1 >= split0, "u2mkd_tile_schedule: need 0 <= split0 <= split1");
    const int64_t t = ceil_div(n_rows, 64);
    U2_REQUIRE(7 * t < (int64_t)1 << 27, "u2mkd_tile_schedule: %lld rows are too many for 28-bit tile ids", (long long)n_rows);
    int32_t *tw = reinterpret_cast<int32_t *>(workspace);
    hipLaunchKernelGGL(tile_weights_kernel, dim3((unsigned)t), dim3(64), 0, as_stream(s), mask, order, n_rows, k, tw);
    hipLaunchKernelGGL(tile_sort_kernel, dim3(1), dim3(kSortThreads), 0, as_stream(s), tw, n_rows, (int)t, split0, split1,
                       tile_order, items, n_items);
    return check_launch("u2mkd_tile_schedule");
}

}  // extern "C"
