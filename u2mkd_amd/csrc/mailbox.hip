// Host mailbox for the voxel-set sizes of the geometry pre-pass.
//
// The reference reads every such size with a blocking device-to-host copy (`torch.unique` inside initial_voxelize and
// spdownsample: core/models/utils.py:20, torchsparse v1.4.0 `spdownsample`).  A stream-ordered copy is a packet in a HARDWARE
// queue: with five HIP streams on four hardware queues it sits behind whatever a neighbour stream queued earlier -- the whole
// backward pass -- and the host that waits for it gives up its lead over the GPU once per step (NOTES N10).  Here the LAST
// kernel of the producing slice stores the numbers into fine-grained, coherent host memory itself and the host polls them:
// what the host waits for is the producer, not the queue.
#include <string.h>

#include "common.h"

namespace u2mkd {

constexpr int kMailMax = 32;
struct MailSources {
    const void *p[kMailMax];
};

// One 8-byte word per value: (seq << 32) | value.  An aligned 8-byte store is indivisible, so the host needs no ordering
// BETWEEN the words (measured: with a separate sequence word the flag reached host memory before a value stored ahead of it
// behind a system-scope fence -- writes to host memory over the fabric are not delivered in program order): a word whose upper
// half is this post's sequence number carries this post's value.  Values outside [0, 2^32 - 2] are sent as 0xffffffff.
__global__ void __launch_bounds__(64)
mailbox_post_kernel(MailSources src, uint32_t is64, int n, uint64_t *__restrict__ slot, uint32_t seq) {
    const int i = threadIdx.x;
    if (i < n) {
        const int64_t v = (is64 >> i) & 1u ? *reinterpret_cast<const int64_t *>(src.p[i])
                                           : (int64_t)*reinterpret_cast<const int32_t *>(src.p[i]);
        const uint64_t lo = (v < 0 || v >= 0xffffffffLL) ? 0xffffffffull : (uint64_t)v;
        __hip_atomic_store(slot + i, ((uint64_t)seq << 32) | lo, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace u2mkd

using namespace u2mkd;

extern "C" {

void *u2mkd_mailbox_alloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) {
        set_error("u2mkd_mailbox_alloc: hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    memset(p, 0, bytes);
    return p;
}

int u2mkd_mailbox_free(void *p) {
    if (p && hipHostFree(p) != hipSuccess) {
        set_error("u2mkd_mailbox_free: hipHostFree failed");
        return 1;
    }
    return 0;
}

int u2mkd_mailbox_post(const void *const *srcs, uint32_t is64, int32_t n, int64_t *slot, int64_t seq, u2mkd_stream_t s) {
    U2_REQUIRE(srcs && slot, "u2mkd_mailbox_post: null pointer");
    U2_REQUIRE(n >= 0 && n <= kMailMax, "u2mkd_mailbox_post: n=%d outside [0, %d]", n, kMailMax);
    MailSources src;
    for (int i = 0; i < kMailMax; ++i) src.p[i] = i < n ? srcs[i] : nullptr;
    for (int i = 0; i < n; ++i) U2_REQUIRE(src.p[i], "u2mkd_mailbox_post: source %d is null", i);
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, slot, 0) != hipSuccess || !dev) {
        set_error("u2mkd_mailbox_post: slot is not mapped host memory (u2mkd_mailbox_alloc)");
        return 1;
    }
    hipLaunchKernelGGL(mailbox_post_kernel, dim3(1), dim3(64), 0, as_stream(s), src, is64, n, reinterpret_cast<uint64_t *>(dev), (uint32_t)seq);
    return check_launch("u2mkd_mailbox_post");
}

}  // extern "C"
