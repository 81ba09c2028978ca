// Internal declarations shared by the sparse-conv translation units (conv.hip, conv_tp.hip).
#pragma once
#include "common.h"

namespace u2mkd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Rows a table-walking forward launch covers: sorted rows [begin, end) of the neighbour table
// nbr[k * ld + row].
struct RowRange {
    int64_t ld, begin, end;
    const int32_t *tile_order;   // launch order of the 64-row tiles (heaviest first) or nullptr (offset-walking kernels)
};

// Tile-local pair schedule (conv_tp.hip); wf = weight fragments of launch_weight_fragments (same `arith`).
// Returns -1 when the shape has no instantiation, else the launch status.
int launch_conv_tp(const char *who, const float *in, int cin, const float *wf, int cout, const int32_t *nbr,
                   const int32_t *order, RowRange rr, const int32_t *items, const int32_t *n_items, int k, int kflip,
                   int arith, float *out, hipStream_t st, unsigned long long *stamps = nullptr);
bool conv_tp_supported(int cin, int cout, int k);
int conv_tp_arith(int arith);
size_t weight_fragments_bytes(int k, int rows, int cols, int arith);
int launch_weight_fragments(const float *w, int k, int rows, int cols, int transpose, int arith, float *wf, hipStream_t st);
int launch_weight_fragments_batch(const int64_t *jobs, int n_jobs, int64_t total_units, hipStream_t st);

// Global pair schedule in bf16x3 arithmetic (conv_px3.hip); wf = arith-2 fragments.  -1 = shape not supported.
bool conv_px3_supported(int cin, int cout);
int launch_conv_px3(const char *who, const float *in, int cin, const float *wf, int cout, const int32_t *pair_idx,
                    const int32_t *tile_k, const int32_t *n_tiles, int64_t capacity, float *y, hipStream_t st, bool b16 = false);
// the same kernel with the identity pair list: y[n_rows, cout] = in x B (+ bias), one offset (nn.Linear)
int launch_linear_px3(const char *who, const float *in, int64_t n_rows, int cin, const float *wf, int cout,
                      const float *bias, float *y, hipStream_t st, bool b16 = false);

// Weight gradient in bf16x3 arithmetic (conv_wgrad_x3.hip): 64 x 64-channel tiles, slabs as the f32 kernel.
bool conv_wgrad_x3_supported(int ca, int cb, int k);
int launch_conv_wgrad_x3(const float *a, int ca, const float *b, int cb, const int32_t *pairs, const int32_t *plan,
                         int k, int swap, int g, int merge, float *slabs, hipStream_t st, bool b16);
int conv_wgrad_x3_merge(int ca, int cb);

}  // namespace u2mkd
