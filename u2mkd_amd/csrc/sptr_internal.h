// Shared by csrc/sptr.hip (one thread per (token, head)) and csrc/sptr_tiles.hip (16 x 16 tiles on the matrix pipe): the
// relative-position row of a (query, key) pair, sptr/modules.py:40-63 + spherical_transformer.py:39-64.
#pragma once
#include "common.h"

namespace u2mkd {

constexpr int kHd = 16;        // head dim (asserted by the reference, sptr/functional.py:355)

// spherical_transformer.py:39-64 exponential_split on d = r_query - r_key
__device__ __forceinline__ int exp_split(float d, float a) {
    float da = fabsf(d);
    float flag = d >= 0.f ? 1.f : 0.f;
    float idx = 2.f * floorf(logf((da + 2.f * a) / a) / 0.6931471805599453f) - 2.f;
    float half = floorf(idx / 2.f);
    idx = idx + (((3.f * exp2f(half) - 2.f) * a <= da) ? 1.f : 0.f);
    idx = idx * (2.f * flag - 1.f) + (flag - 1.f);
    return (int)idx + 24;
}

// row strides (floats) of the token-major operands and the scale applied to q on load: lets the kernels read q, k, v
// straight out of the packed [N, 3, H, 16] output of the qkv projection (one branch = a range of heads), write the
// heads of a branch into their columns of the [N, H * 16] attention output, and the gradients into a packed
// [N, 3, H, 16] buffer -- without the slice / scale / concatenate copies around them
struct SptrLayout {
    int64_t ld_qkv, ld_out, ld_grad;
    float q_scale;
};

struct RelCtx {
    int qgl;        // quant_grid_length
    float a;        // > 0: spherical branch (exponential radial split + clamp)
};

__device__ __forceinline__ void rel_rows(const RelCtx &c, const int qi[3], float ri, const int qj[3], float rj,
                                         int r[3]) {
    r[0] = qi[0] - qj[0] + c.qgl - 1;
    r[1] = qi[1] - qj[1] + c.qgl - 1;
    r[2] = qi[2] - qj[2] + c.qgl - 1;
    if (c.a > 0.f) {
        r[2] = exp_split(ri - rj, c.a);
        const int hi = 2 * c.qgl - 1;
        r[0] = min(max(r[0], 0), hi);
        r[1] = min(max(r[1], 0), hi);
        r[2] = min(max(r[2], 0), hi);
    }
}


}  // namespace u2mkd
