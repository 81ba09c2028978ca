// Shared helpers of libu2mkd_hip (gfx950 only; 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/u2mkd_hip.h"

namespace u2mkd {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}

#define U2_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            set_error(__VA_ARGS__);  \
            return 2;                \
        }                            \
    } while (0)

inline hipStream_t as_stream(u2mkd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// FNV-1a-64 over 4 int32 words folded to 60 bits (torchsparse v1.4.0 hash kernel).
__device__ __forceinline__ int64_t fnv_hash4(int x, int y, int z, int b) {
    uint64_t h = 14695981039346656037ULL;
    h ^= (uint32_t)x; h *= 1099511628211ULL;
    h ^= (uint32_t)y; h *= 1099511628211ULL;
    h ^= (uint32_t)z; h *= 1099511628211ULL;
    h ^= (uint32_t)b; h *= 1099511628211ULL;
    h = (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFULL);
    return (int64_t)h;
}

// ---- feature-row element types: float, or bf16 (BF16 STORAGE, BASELINE.json configs[4]) --------------------------
// ld4 / st4 move 4 consecutive channels of a row (index in units of 4 elements); bf16 values are widened exactly on
// load and rounded to nearest-even once at the store, arithmetic in between is fp32.
struct bf16row { unsigned short v; };
template <typename T> __device__ __forceinline__ float4 ld4(const T *p, int64_t i4);
template <> __device__ __forceinline__ float4 ld4<float>(const float *p, int64_t i4) { return reinterpret_cast<const float4 *>(p)[i4]; }
template <> __device__ __forceinline__ float4 ld4<bf16row>(const bf16row *p, int64_t i4) {
    const uint2 w = reinterpret_cast<const uint2 *>(p)[i4];     // a bf16 is the upper half of the fp32 with the same value
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                       __uint_as_float(w.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T *p, int64_t i4, const float4 &v);
template <> __device__ __forceinline__ void st4<float>(float *p, int64_t i4, const float4 &v) { reinterpret_cast<float4 *>(p)[i4] = v; }
template <> __device__ __forceinline__ void st4<bf16row>(bf16row *p, int64_t i4, const float4 &v) {
    const __bf16 a = (__bf16)v.x, b = (__bf16)v.y, c = (__bf16)v.z, d = (__bf16)v.w;      // round to nearest even, NaN stays NaN
    uint2 w;
    w.x = (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
    w.y = (uint32_t)__builtin_bit_cast(unsigned short, c) | ((uint32_t)__builtin_bit_cast(unsigned short, d) << 16);
    reinterpret_cast<uint2 *>(p)[i4] = w;
}


// ---- hash table view (keys then values in one caller-owned buffer) --------
struct TableView {
    int64_t *keys;   // cap entries, -1 = empty
    int32_t *vals;   // cap entries
    uint32_t mask;   // cap - 1
    int shift;       // 64 - log2(cap)
};

inline int64_t table_capacity(int64_t n_refs) {
    int64_t cap = 1024;
    while (cap < 2 * n_refs) cap <<= 1;
    return cap;
}

inline TableView make_table_view(void *buf, int64_t n_refs) {
    TableView t;
    int64_t cap = table_capacity(n_refs);
    t.keys = reinterpret_cast<int64_t *>(buf);
    t.vals = reinterpret_cast<int32_t *>(t.keys + cap);
    t.mask = (uint32_t)(cap - 1);
    int lg = 0;
    while ((1LL << lg) < cap) ++lg;
    t.shift = 64 - lg;
    return t;
}

__device__ __forceinline__ uint32_t table_slot(const TableView &t, int64_t key) {
    uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ULL;
    return (uint32_t)(h >> t.shift) & t.mask;
}

__device__ __forceinline__ int table_lookup(const TableView &t, int64_t key) {
    uint32_t slot = table_slot(t, key);
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        int64_t k = t.keys[slot];
        if (k == key) return t.vals[slot];
        if (k == -1) return -1;
        slot = (slot + 1) & t.mask;
    }
    return -1;
}

}  // namespace u2mkd
