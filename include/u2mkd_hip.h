/*
 * u2mkd_hip.h -- C ABI of libu2mkd_hip.so, the MI355X (gfx950) native
 * implementation of the U2MKD training hot path.
 *
 * Boundary rules
 *   - plain C: device pointers + sizes + a HIP stream handle, no torch types;
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - every function returns 0 on success, non-zero on error (message via
 *     u2mkd_last_error()); nothing is allocated or synchronised inside, so
 *     every call is hipGraph-capturable;
 *   - outputs documented "pre-zeroed" must be zero-filled by the caller, the
 *     convention of the reference's native layer (SURVEY.md section 8b:
 *     "Python allocates outputs (zero-filled) and passes them in").
 *
 * Each entry cites the reference interface it replaces.  torchsparse v1.4.0
 * (reference README.md:44-48) is the un-vendored native backend behind
 * core/models/build_blocks.py:25-80 and core/models/utils.py:15-135; its
 * pybind names are torchsparse.backend.<name>_cuda.  The sptr entries replace
 * the extern "C" launchers declared in
 * third_party/SparseTransformer/src/sptr/{precompute,attention,rpe}/..._kernel.h.
 */
#ifndef U2MKD_HIP_H
#define U2MKD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *u2mkd_stream_t; /* hipStream_t */

/* ---- library ---------------------------------------------------------- */
int u2mkd_version(void);
const char *u2mkd_last_error(void);
/* Stream ordering for the host side (no reference counterpart: the reference is single-stream): `waiter` continues only behind
 * what is queued on `signaler` at the time of the call -- one event record + one stream wait on an event object kept per waiter
 * stream (the one exception to "nothing is allocated inside": a hipEvent per stream, once).  torch.cuda.Stream.wait_stream's
 * job in one call; not for use under hipGraph capture of the signaler. */
int u2mkd_stream_wait_stream(u2mkd_stream_t waiter, u2mkd_stream_t signaler);
/* Host mailbox (csrc/mailbox.hip) for the sizes a geometry pre-pass has to know on the host -- the values the reference reads
 * with blocking copies inside `torch.unique` (core/models/utils.py:20) and torchsparse v1.4.0's spdownsample.
 * u2mkd_mailbox_alloc: `bytes` of mapped, coherent host memory, zeroed (the second exception to "nothing is allocated
 * inside": once per process).  u2mkd_mailbox_post queues ONE small kernel on `s` that stores the n <= 32 device integers
 * srcs[i] (int64 where bit i of is64 is set, else int32; `srcs` is a HOST array of device pointers) as
 * slot[i] = (seq mod 2^32) << 32 | value, one indivisible 8-byte system-scope store each (values outside [0, 2^32 - 2] arrive
 * as 0xffffffff): a host that polls until the upper half of every word equals `seq` has the values without any stream-ordered
 * copy or stream synchronisation, and without relying on the order in which two device stores reach host memory. */
void *u2mkd_mailbox_alloc(size_t bytes);
int u2mkd_mailbox_free(void *p);
int u2mkd_mailbox_post(const void *const *srcs, uint32_t is64, int32_t n, int64_t *slot, int64_t seq, u2mkd_stream_t s);

/* ---- coordinate hashing ------------------------------------------------
 * replaces torchsparse.backend.hash_cuda / kernel_hash_cuda
 * (F.sphash, called core/models/utils.py:19,43-49,86-92,133-134).
 * FNV-1a-64 over the int32 words (x,y,z,b), folded to 60 bits.            */
int u2mkd_hash(const int32_t *coords /*[n,4]*/, int64_t n, int64_t *out /*[n]*/, u2mkd_stream_t s);
int u2mkd_kernel_hash(const int32_t *coords /*[n,4]*/, const int32_t *offsets /*[k,3]*/, int64_t n,
                      int32_t k, int64_t *out /*[k,n]*/, u2mkd_stream_t s);

/* ---- hash table --------------------------------------------------------
 * replaces torchsparse.backend.hash_query_cuda (F.sphashquery,
 * core/models/utils.py:21,50,93,135).  Open addressing, 64-bit keys;
 * duplicates resolve to the smallest index (== dense_hash_map::insert).
 * The table lives in caller memory of u2mkd_hash_table_bytes(n_refs) bytes. */
size_t u2mkd_hash_table_bytes(int64_t n_refs);
int u2mkd_hash_table_build(const int64_t *refs, int64_t n_refs, void *table, u2mkd_stream_t s);
int u2mkd_hash_table_query(const void *table, int64_t n_refs, const int64_t *queries, int64_t n_q,
                           int64_t *out /*[n_q] index or -1*/, u2mkd_stream_t s);
/* the same with an int32 copy of the result (indices fit: n_refs < 2^31): what the voxelise / count kernels read */
int u2mkd_hash_table_query2(const void *table, int64_t n_refs, const int64_t *queries, int64_t n_q, int64_t *out,
                            int32_t *out32, u2mkd_stream_t s);

/* ---- kernel map (rulebook) ---------------------------------------------
 * replaces the kmap build of torchsparse F.conv3d (python: sphash(offsets) +
 * sphashquery + nonzero).  The native form is the neighbour table
 * nbr[k][j] = index of the input voxel at out_coords[j] + offsets[k], or -1.
 * u2mkd_kmap_invert gives the table of the swapped roles (strided maps);
 * u2mkd_kmap_sizes / u2mkd_kmap_compact produce torchsparse's
 * (nbmaps [P,2] rows (in,out) grouped by k with ascending out, nbsizes [K]). */
int u2mkd_kmap_build_table(const void *table, int64_t n_refs, const int32_t *out_coords /*[n_out,4]*/,
                           int64_t n_out, const int32_t *offsets /*[k,3]*/, int32_t k,
                           int32_t *nbr /*[k,n_out]*/, u2mkd_stream_t s);
int u2mkd_kmap_invert(const int32_t *nbr /*[k,n_out]*/, int64_t n_out, int32_t k, int64_t n_in,
                      int32_t *nbr_inv /*[k,n_in], pre-filled with -1*/, u2mkd_stream_t s);
int u2mkd_kmap_sizes(const int32_t *nbr, int64_t n_out, int32_t k, int32_t *nbsizes /*[k] pre-zeroed*/,
                     int32_t *block_counts /*[k, ceil(n_out/1024)]*/, u2mkd_stream_t s);
int u2mkd_kmap_compact(const int32_t *nbr, int64_t n_out, int32_t k, const int32_t *nbsizes /*[k]*/,
                       int32_t *block_counts /*[k, ceil(n_out/1024)] from u2mkd_kmap_sizes (scanned in place)*/,
                       int32_t *nbmaps /*[P,2]*/, u2mkd_stream_t s);

/* The PAIR SCHEDULE of a map (what torchsparse keeps as nbmaps/nbsizes, laid out for dense
 * MFMA tiles): all (input i, output j) pairs grouped by offset, every offset's group padded
 * with -1 to a multiple of 128 entries so that TWO consecutive 64-entry tiles belong to ONE offset (conv_px3.hip multiplies
 * two tiles per step by one set of weight fragments).
 *   pair_in / pair_out [u2mkd_pairs_capacity]  rows of the pair in slot p (or -1)
 *   pos_out [n_out, k]   slot of the pair (k, j), -1 if none       (written completely)
 *   pos_in  [n_in,  k]   slot of the pair (k, i), -1 if none       (caller pre-fills -1)
 *   tile_k  [capacity / 64]  offset of tile t;  meta[0] = padded pair count, meta[1] = tiles
 * nbsizes / block_counts come from u2mkd_kmap_sizes (block_counts is overwritten).  Entirely
 * device-side: the host never learns the pair count (buffers are sized by the capacity).   */
int64_t u2mkd_pairs_capacity(int64_t n_in, int64_t n_out, int32_t k);
int u2mkd_pairs_build(const int32_t *nbr /*[k,n_out]*/, int64_t n_out, int64_t n_in, int32_t k,
                      const int32_t *nbsizes, int32_t *block_counts, int32_t *pair_in, int32_t *pair_out,
                      int32_t *pos_out, int32_t *pos_in, int32_t *tile_k, int32_t *meta, u2mkd_stream_t s);
/* downsample keys: pack floor(xyz / s) * s and b into an order-preserving
 * int64 ((b,x,y,z) lexicographic, as torch.unique(dim=0) sorts) and back.
 * replaces the arithmetic of F.spdownsample (Appendix A-4).                 */
int u2mkd_downsample_keys(const int32_t *coords /*[n,4]*/, int64_t n, int32_t sx, int32_t sy, int32_t sz,
                          int64_t *keys /*[n]*/, u2mkd_stream_t s);
/* The packed key holds x, y, z in [-131072, 131072) and b in [0, 512).  A row outside that range cannot be packed
 * (torch.unique(dim=0) of the reference has no such limit): it gets the key INT64_MAX and *range_flag (device int32,
 * caller-zeroed, may be NULL) is set to 1, so the caller can raise instead of merging voxels silently.            */
int u2mkd_downsample_keys_checked(const int32_t *coords /*[n,4]*/, int64_t n, int32_t sx, int32_t sy, int32_t sz,
                                  int64_t *keys /*[n]*/, int32_t *range_flag, u2mkd_stream_t s);
int u2mkd_unpack_keys(const int64_t *keys, int64_t n, int32_t *coords /*[n,4]*/, u2mkd_stream_t s);
/* voxel of every point at tensor stride s: (floor(xyz / s) * s, (int)b) from float (x, y, z, b) rows -- the hash
 * input of point_to_voxel / voxel_to_point (core/models/utils.py:43-47,86-90), one launch instead of seven.    */
int u2mkd_floor_coords(const float *pc /*[n,4]*/, int64_t n, int32_t stride, int32_t *out /*[n,4]*/, u2mkd_stream_t s);

/* ---- sparse convolution ------------------------------------------------
 * replaces torchsparse.backend.convolution_forward_cuda /
 * convolution_backward_cuda (ConvolutionFunction; every spnn.Conv3d of
 * core/models/build_blocks.py:25-80).  Output-stationary: no atomics, every
 * output row is written exactly once.
 *   out[j] = sum_k in[nbr[k][j]] * B_k,  B_k = wt[kflip ? K-1-k : k] given
 *   as [cout][cin] (reduction dim contiguous).
 * forward:  wt = transpose of `kernel` (u2mkd_transpose_weights)
 * dgrad:    wt = `kernel` itself with in := grad_out, nbr := table of the
 *           swapped roles (kflip = 1 on the same table for submanifold maps). */
int u2mkd_transpose_weights(const float *w /*[k,cin,cout]*/, int32_t k, int32_t cin, int32_t cout,
                            float *wt /*[k,cout,cin]*/, u2mkd_stream_t s);
int u2mkd_conv_forward(const float *in /*[n_in,cin]*/, int64_t n_in, int32_t cin, const float *wt /*[k,cout,cin]*/,
                       int32_t cout, const int32_t *nbr /*[k,n_out]*/, int64_t n_out, int32_t k, int32_t kflip,
                       float *out /*[n_out,cout]*/, u2mkd_stream_t s);
/* Same contraction on a MASK-SORTED table: row p of nbr_sorted / of the launch grid is
 * original row order[p] (order == NULL: identity).  Sorting rows by their 27-bit
 * neighbour mask makes 64-row tiles mask-homogeneous.  A tile visits the union of its rows'
 * offsets one after the other (conv_os2 / conv_os3); tile_order: launch order of the 64-row
 * tiles -- heaviest first (NULL: ascending).  kflip in {0,1}.                               */
int u2mkd_conv_forward_sorted(const float *in, int64_t n_in, int32_t cin, const float *wt, int32_t cout,
                              const int32_t *nbr_sorted /*[k,n_out]*/, const int32_t *order /*[n_out] or NULL*/,
                              const int32_t *tile_order /*[ceil(n_out/64)] or NULL*/, int64_t n_out, int32_t k,
                              int32_t kflip, float *out /*[n_out,cout]*/, u2mkd_stream_t s);
/* The TILE-LOCAL PAIR schedule (csrc/conv_tp.hip), for layers whose reduction fits the register
 * file (u2mkd_conv_tiles_supported: cin in {32,64,96,128}, cout in {32,64,96,128}): the (row,
 * neighbour) pairs of a 64-row tile are compacted per offset in LDS with wave ballots + prefix
 * popcounts into dense 16-pair MFMA blocks, the waves split the output columns, the output tile is
 * LDS-resident, gathered rows go through a coalesced LDS image.  Same arguments as
 * u2mkd_conv_forward_sorted except the weights: wf = FRAGMENT layout of
 * u2mkd_weight_fragments(w [k,rows,cols], transpose):
 *   forward:        w = kernel [k,cin,cout], transpose = 1   (B_k[col][ci] = kernel[k][ci][col])
 *   input gradient: w = kernel [k,cin,cout], transpose = 0   (B_k[ci][co]  = kernel[k][ci][co])
 * wf has u2mkd_weight_fragments_bytes(k, rows, cols, arith) bytes; rows and cols must be multiples of 32.
 * transpose = 2 writes BOTH layouts in one launch (a training step needs both, the launch is latency-bound):
 * wf = [transpose = 1 layout | transpose = 0 layout], 2 x u2mkd_weight_fragments_bytes bytes.
 * arith selects the arithmetic of the MFMA tiles (the same value for the fragments and the convolution):
 *   1 = f32 MFMA (v_mfma_f32_16x16x4_f32: the exact fp32 fma chain);
 *   2 = bf16x3: every fp32 operand is split exactly into three bf16 (8 + 8 + 8 significand bits), the six partial
 *       products above 2^-24 relative are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 -- fp32 GEMM accuracy
 *       (not the bitwise fma chain) at 2.7x fewer matrix-pipe cycles; inputs and outputs stay fp32;
 *   4 = f16x2 (tile kernel, fp32 rows, cin in {32, 64, 128}): every gathered row is scaled by the power of two that puts its
 *       largest |x| into [2^14, 2^15) and split into two fp16 planes (11 + 11 significand bits, round to nearest), the weights
 *       likewise with one scale per tensor (16-byte trailer {scale, 1/scale, 0, 0} behind each orientation's fragments); the
 *       three partial products above 2^-24 relative are accumulated in fp32 by v_mfma_f32_16x16x16_f16 and the scales taken
 *       out exactly -- the accuracy class of bf16x3 at half its matrix instructions; inputs and outputs stay fp32;
 *   0 = the library default (bf16x3; environment U2MKD_CONV_ARITH=f32 selects 1).
 * u2mkd_conv_tiles_arith(cin, cout, k): the code the host side should pass for a layer of fp32 rows -- 4 where the f16x2
 * form exists, else 2; U2MKD_CONV_ARITH=f32 / bf16x3 force 1 / 2; 0 = not a tile-kernel layer.
 * Work items (optional): a tile is a serial chain of MFMA blocks, so tiles with many blocks are cut
 * into halves / quarters: items[i] = tile << 4 | sub << 2 | lg  covers the 64 >> lg sorted rows from
 * 64 tile + (64 >> lg) sub, every row exactly once, listed heaviest first; *n_items (device memory,
 * never read by the host) of them.  NULL = one item per 64-row tile.                              */
int32_t u2mkd_conv_tiles_supported(int32_t cin, int32_t cout, int32_t k);
int32_t u2mkd_conv_tiles_arith(int32_t cin, int32_t cout, int32_t k);
size_t u2mkd_weight_fragments_bytes(int32_t k, int32_t rows, int32_t cols, int32_t arith);
int u2mkd_weight_fragments(const float *w, int32_t k, int32_t rows, int32_t cols, int32_t transpose, int32_t arith,
                           void *wf, u2mkd_stream_t s);
/* The same re-layout for MANY weights in ONE launch: a training step changes every trainable weight at once, so the
 * host side (torchsparse/nn/functional.py: refresh_weight_fragments, an optimizer-step post hook) re-lays all of them
 * behind the optimizer instead of one latency-bound launch per weight in front of its first convolution.
 * jobs: DEVICE int64 [n_jobs][8] = {w (device address of the fp32 [k,rows,cols] weight), wf (device address of
 * 2 x u2mkd_weight_fragments_bytes(k, rows, cols, arith) bytes: [transpose = 1 | transpose = 0]), first unit, k, rows,
 * cols, planes (3 for arith 2 = bf16x3, 1 for arith 3 = one bf16 plane, 2 for arith 4 = f16x2), 0}; job j owns the units
 * [first_j, first_j + 2 k rows cols / 512), consecutive from 0; total_units = their sum.  Replaces nothing in the
 * reference (torchsparse reads `kernel` as it is); it is the price of the MFMA fragment order.            */
int u2mkd_weight_fragments_batch(const int64_t *jobs, int32_t n_jobs, int64_t total_units, u2mkd_stream_t s);
int u2mkd_conv_forward_tiles(const float *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                             const int32_t *nbr_sorted /*[k,n_out]*/, const int32_t *order /*[n_out] or NULL*/,
                             const int32_t *items /*[<= 4 ceil(n_out/64)] or NULL*/,
                             const int32_t *n_items /*[1] device, or NULL*/, int64_t n_out, int32_t k,
                             int32_t kflip, int32_t arith, float *out /*[n_out,cout]*/, u2mkd_stream_t s);
/* Inference form of the two conv schedules with an eval-mode BatchNorm (+ residual add, + ReLU) FOLDED into the store:
 * out = relu?(conv * scale[col] + shift[col] (+ res[row][col])), scale = gamma / sqrt(running_var + eps), shift = beta -
 * running_mean * scale -- spnn.Conv3d -> spnn.BatchNorm (eval) -> spnn.ReLU of core/models/build_blocks.py:25-31,59-71 as ONE
 * launch (the frozen teacher of the KD step, tsd_full.py:590-596, and every evaluation pass).  fp32 rows only.               */
int u2mkd_conv_forward_tiles_ep(const float *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                const int32_t *nbr_sorted, const int32_t *order, const int32_t *items,
                                const int32_t *n_items, int64_t n_out, int32_t k, int32_t kflip, int32_t arith,
                                const float *scale /*[cout]*/, const float *shift /*[cout]*/, const float *res /*[n_out,cout] or NULL*/,
                                int32_t relu, float *out, u2mkd_stream_t s);
/* u2mkd_pairs_gather_sum with the BatchNorm statistics of its output taken in the store (spnn.Conv3d -> spnn.BatchNorm in
 * training mode, core/models/build_blocks.py:25-31,59-70): partial [ceil(n_rows / slab_rows)][2][cout] = (mean, M2) per slab of
 * u2mkd_pairs_gather_sum_stats_slab_rows() consecutive output rows, for u2mkd_bn_train_forward_from_partial.  cout: a multiple
 * of 4 in 8..512 (u2mkd_pairs_gather_sum_stats_supported).                                                                */
int32_t u2mkd_pairs_gather_sum_stats_slab_rows(void);
int32_t u2mkd_pairs_gather_sum_stats_supported(int32_t cout);
int u2mkd_pairs_gather_sum_stats(const float *y, const int32_t *pos, int64_t n_rows, int32_t k, int32_t cout, float *out,
                                 float *partial, u2mkd_stream_t s);
int u2mkd_pairs_gather_sum_ep(const float *y, const int32_t *pos, int64_t n_rows, int32_t k, int32_t cout, const float *scale,
                              const float *shift, const float *res, int32_t relu, float *out, u2mkd_stream_t s);
/* Tuning / A-B experiments only (tools/ab_*.py), NOT part of the drop-in boundary:
 * u2mkd_conv_forward_sorted with an explicit kernel choice.  variant 0 = the product heuristic,
 * waves*100 + kc = conv_os2, 3000 + rb*100 + kc = conv_os3 (see conv.hip).                   */
int u2mkd_debug_conv_forward_sorted(const float *in, int64_t n_in, int32_t cin, const float *wt, int32_t cout,
                                    const int32_t *nbr_sorted, const int32_t *order, const int32_t *tile_order,
                                    int64_t n_out, int32_t k, int32_t kflip, int32_t variant, float *out,
                                    u2mkd_stream_t s);
/* Profiling only: the 64 -> 64 tile-pair kernel with per-workgroup timestamps; stamps [tiles, 8] uint64 =
 * {realtime start, cycles start, after compaction, after the block walk, end, realtime end, blocks, tile}. */
int u2mkd_debug_conv_tile_pairs_stamps(const float *in, int64_t n_in, const void *wf /*fragments of [k,64,64]*/,
                                       const int32_t *nbr_sorted, const int32_t *order, const int32_t *items,
                                       const int32_t *n_items, int64_t n_out, int32_t k, int32_t arith, float *out,
                                       uint64_t *stamps, u2mkd_stream_t s);
/* Profiling only: the 64 x 64 weight-gradient kernel with s_memtime stamps of workgroup 0 (stamps [256]). */
int u2mkd_debug_wgrad_stamps(const float *a, const float *b, const int32_t *pairs, const int32_t *plan, int64_t n_rows,
                             int32_t k, void *workspace, uint64_t *stamps, u2mkd_stream_t s);
/* y[p] = in[pair_idx[p]] * B_{tile_k[p / 64]} over the pair schedule of u2mkd_pairs_build
 * (pair_idx = pair_in for a normal conv, pair_out for a transposed conv / the input gradient):
 * one dense MFMA stage per 64-pair tile, no serial walk over offsets.  meta (device) holds
 * the tile count; a fixed grid splits the tiles into contiguous runs, so nothing is read back
 * by the host.  kc = reduction channels per pipeline stage: 32 or 64, 0 = chosen from the shape.
 * y must hold `capacity` rows (only the first meta[0] are written; padding entries give 0).  */
int u2mkd_conv_forward_pairs(const float *in, int64_t n_in, int32_t cin, const float *wt, int32_t cout,
                             const int32_t *pair_idx, const int32_t *tile_k, const int32_t *meta, int64_t capacity,
                             int32_t k, int32_t kc, float *y /*[capacity,cout]*/, u2mkd_stream_t s);
/* The same product in bf16x3 arithmetic (csrc/conv_px3.hip) for cin, cout multiples of 32
 * (u2mkd_conv_pairs_x3_supported; 0 when U2MKD_CONV_ARITH=f32): wf = the arith-2 FRAGMENT layout of
 * u2mkd_weight_fragments (forward: transpose = 1; input gradient / transposed roles: transpose = 0).  A workgroup
 * multiplies 64 pairs by 96 / 128 output columns, the gathered rows are split into their three bf16 planes once
 * per workgroup (LDS image) and every weight fragment is used for 4 row blocks from registers: fp32 GEMM accuracy
 * at 2.7x fewer matrix-pipe cycles than the f32-MFMA kernel above.  The y rows of padding entries are NOT written
 * (u2mkd_pairs_gather_sum never reads them).                                                                    */
int32_t u2mkd_conv_pairs_x3_supported(int32_t cin, int32_t cout);
int u2mkd_conv_forward_pairs_x3(const float *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                const int32_t *pair_idx, const int32_t *tile_k, const int32_t *meta, int64_t capacity,
                                int32_t k, float *y /*[capacity,cout]*/, u2mkd_stream_t s);
/* y = x * w^T (+ bias): nn.Linear on the rows of a feature matrix (the point-branch MLPs of
 * core/models/semantickitti/spvcnn.py:58-74 and the 1x1x1 convs of ResidualBlock.downsample,
 * build_blocks.py:69-72), on the pair kernel's pipeline with the identity schedule.  w is
 * nn.Linear's [cout, cin]; y must hold ceil(n / 64) * 64 rows (rows >= n receive the bias). */
int u2mkd_linear_forward(const float *x /*[n,cin]*/, int64_t n, int32_t cin, const float *w /*[cout,cin]*/, int32_t cout,
                         const float *bias /*[cout] or NULL*/, int32_t kc, float *y, u2mkd_stream_t s);
/* The same product in bf16x3 arithmetic (csrc/conv_px3.hip, identity pair list): wf = the fragment-order weights
 * of u2mkd_weight_fragments(w, 1, cout_l, cin_l, ...) for nn.Linear's w [cout_l, cin_l] -- orientation 1
 * (transpose = 0) for the forward y = x w^T, orientation 0 (transpose = 1) for the input gradient dx = dy w; cin
 * and cout as u2mkd_conv_pairs_x3_supported asks.  y holds exactly n rows.                                      */
int u2mkd_linear_forward_x3(const float *x /*[n,cin]*/, int64_t n, int32_t cin, const void *wf, int32_t cout,
                            const float *bias /*[cout] or NULL*/, float *y /*[n,cout]*/, u2mkd_stream_t s);
/* The pair-schedule product and the dense product in f16x2 arithmetic (csrc/conv_px3.hip, F2): the same kernels, schedules and
 * arguments as u2mkd_conv_forward_pairs_x3 / u2mkd_linear_forward_x3 with wf = the arith-4 fragments of u2mkd_weight_fragments
 * (two fp16 planes + the tensor's scale trailer); every 32-channel step of a gathered row carries its own power-of-two scale.
 * Half the matrix instructions and two thirds of the weight-fragment bytes of bf16x3 at the same accuracy class.
 * u2mkd_conv_pairs_f16x2_supported: cin, cout multiples of 32 and no U2MKD_CONV_ARITH=bf16x3 / f32 override.  Replace the same
 * torchsparse v1.4.0 gather -> mm -> scatter-add and torch.nn.functional.linear as the _x3 entries. */
int32_t u2mkd_conv_pairs_f16x2_supported(int32_t cin, int32_t cout);
int u2mkd_conv_forward_pairs_f16x2(const float *in, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                    const int32_t *pair_idx, const int32_t *tile_k, const int32_t *meta, int64_t capacity,
                                    int32_t k, float *y, u2mkd_stream_t s);
int u2mkd_linear_forward_f16x2(const float *x, int64_t n, int32_t cin, const void *wf, int32_t cout, const float *bias,
                               float *y, u2mkd_stream_t s);
/* out[j] = sum_k y[pos[j][k]] (pos < 0: no pair), offsets in ascending order: the
 * deterministic replacement of torchsparse's scatter-add for the pair schedule.             */
int u2mkd_pairs_gather_sum(const float *y, const int32_t *pos /*[n_rows,k]*/, int64_t n_rows, int32_t k, int32_t cout,
                           float *out /*[n_rows,cout]*/, u2mkd_stream_t s);
/* neighbour mask of every output row: bit k set iff nbr[k][j] >= 0 (k <= 32). */
int u2mkd_kmap_rowmask(const int32_t *nbr /*[k,n_out]*/, int64_t n_out, int32_t k, int32_t *mask /*[n_out]*/,
                       u2mkd_stream_t s);
/* BF16 STORAGE variants (BASELINE.json configs[4]; torchsparse runs its conv in half precision under autocast,
 * custom_fwd(cast_inputs=half)): feature rows in and out are bf16 [n, c] (2 bytes per
 * channel: half the gather bytes), weights = ONE bf16 plane in fragment order (u2mkd_weight_fragments arith = 3,
 * u2mkd_weight_fragments_bytes(.., 3) = 2 bytes per weight), products on v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation, every output rounded to bf16 once.  The weight gradient reads bf16 rows and returns fp32 (the
 * optimizer's master dtype).  Parity statements stay in fp32; these are held against the fp32 kernels on
 * bf16-rounded inputs (tests/test_gpu_torchsparse_ops.py).                                                     */
int u2mkd_conv_forward_tiles_bf16(const void *in /*bf16 [n_in,cin]*/, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                  const int32_t *nbr_sorted, const int32_t *order, const int32_t *items,
                                  const int32_t *n_items, int64_t n_out, int32_t k, int32_t kflip,
                                  void *out /*bf16 [n_out,cout]*/, u2mkd_stream_t s);
int u2mkd_conv_wgrad_pairs_bf16(const void *a /*bf16 [.,ca]*/, int32_t ca, const void *b /*bf16 [.,cb]*/, int32_t cb,
                                const int32_t *pairs, const int32_t *plan, int64_t n_rows, int32_t k, int32_t swap,
                                void *workspace, size_t workspace_bytes, float *dw /*[k,ca,cb] fp32*/, u2mkd_stream_t s);
/* The wide layers (cin * cout >= 8192), nn.Linear and the pair schedule's gather-sum on bf16 rows (csrc/conv_px3.hip B16):
 * u2mkd_conv_forward_pairs_x3 / u2mkd_linear_forward_x3 / u2mkd_pairs_gather_sum with bf16 rows in and out (the scratch
 * rows y are bf16 too), wf = the arith-3 fragments (u2mkd_weight_fragments: conv forward transpose = 1, input gradient /
 * transposed roles transpose = 0; Linear: w [cout_l, cin_l] with k = 1, forward orientation 1 = transpose 0), bias fp32,
 * cin and cout multiples of 32 (gather-sum: cout a multiple of 8), fp32 accumulation, ONE rounding per stored value.   */
int u2mkd_conv_forward_pairs_bf16(const void *in /*bf16 [n_in,cin]*/, int64_t n_in, int32_t cin, const void *wf, int32_t cout,
                                   const int32_t *pair_idx, const int32_t *tile_k, const int32_t *meta, int64_t capacity,
                                   int32_t k, void *y /*bf16 [capacity,cout]*/, u2mkd_stream_t s);
int u2mkd_linear_forward_bf16(const void *x /*bf16 [n,cin]*/, int64_t n, int32_t cin, const void *wf, int32_t cout,
                              const float *bias /*[cout] or NULL*/, void *y /*bf16 [n,cout]*/, u2mkd_stream_t s);
int u2mkd_pairs_gather_sum_bf16(const void *y /*bf16*/, const int32_t *pos /*[n_rows,k]*/, int64_t n_rows, int32_t k,
                                int32_t cout, void *out /*bf16 [n_rows,cout]*/, u2mkd_stream_t s);
/* The TILE SCHEDULE of a (mask-sorted) neighbour table, built on the device (csrc/schedule.hip): `mask` = the
 * rows' neighbour masks (u2mkd_kmap_rowmask), `order` = the row permutation that sorts them (NULL: identity);
 * a tile = 64 consecutive sorted rows, its weight = its 16-pair MFMA blocks (sum over offsets of ceil(pairs / 16)).
 *   tile_order [ceil(n/64)]   tiles by descending weight (stable)                (offset-walking kernels)
 *   items [7 ceil(n/64)]      work items of u2mkd_conv_forward_tiles, heaviest first: tile << 4 | sub << 2 | lg,
 *                             a tile above split0 blocks as 2 halves, above split1 as 4 quarters; the first
 *                             *n_items entries are live (the count never leaves the device)
 * workspace: u2mkd_tile_schedule_workspace_bytes(n_rows).  Belongs to the kernel-map cache (torchsparse keeps
 * nbmaps / nbsizes per (stride, kernel) key, core/models/utils.py:60-61).                                     */
size_t u2mkd_tile_schedule_workspace_bytes(int64_t n_rows);
int u2mkd_tile_schedule(const int32_t *mask /*[n_rows]*/, const int32_t *order /*[n_rows] or NULL*/, int64_t n_rows,
                        int32_t k, int32_t split0, int32_t split1, void *workspace, int32_t *tile_order,
                        int32_t *items, int32_t *n_items, u2mkd_stream_t s);
/* dW[k] = sum over the pairs (i, j) of offset k of A_i^T B_j  (the dW half of torchsparse
 * convolution_backward_cuda), over the compacted pair list (the rulebook of u2mkd_kmap_compact; the
 * buffer may be over-allocated, only plan[0] = P pairs are read).  u2mkd_wgrad_plan turns
 * nbsizes into the device-side work split (no host sync); plan has u2mkd_wgrad_plan_ints(k)
 * int32 entries.  swap = 0: A rows = pairs[:,0], B rows = pairs[:,1] (normal conv: A = the
 * layer input, B = grad_output); swap = 1: the transposed conv.  Deterministic.           */
int32_t u2mkd_wgrad_plan_ints(int32_t k);
int u2mkd_wgrad_plan(const int32_t *nbsizes /*[k]*/, int32_t k, int64_t n_rows, int32_t *plan, u2mkd_stream_t s);
size_t u2mkd_conv_wgrad_pairs_workspace_bytes(int64_t n_rows, int32_t ca, int32_t cb, int32_t k);
int u2mkd_conv_wgrad_pairs(const float *a, int32_t ca, const float *b, int32_t cb, const int32_t *pairs /*[>=P,2]*/,
                           const int32_t *plan, int64_t n_rows, int32_t k, int32_t swap, void *workspace,
                           size_t workspace_bytes, float *dw /*[k,ca,cb]*/, u2mkd_stream_t s);

/* ---- torchsparse v1.4.0 backend format ---------------------------------------
 * replace torchsparse.backend.convolution_forward_cuda(in_feat, out_feat, kernel, neighbor_map,
 * neighbor_offset, transpose) and convolution_backward_cuda(in_feat, grad_in_feat, grad_out_feat, kernel,
 * grad_kernel, neighbor_map, neighbor_offset, transpose) argument for argument (ConvolutionFunction of
 * torchsparse/nn/functional/conv.py, behind every spnn.Conv3d of core/models/build_blocks.py:25-80), for a
 * caller that already holds a v1.4.0 kernel map: neighbor_map = nbmaps int32 [P,2] rows (in, out) grouped
 * by kernel offset (DEVICE), neighbor_offset = nbsizes int32 [k] pairs per offset ON THE HOST (v1.4.0 keeps
 * it on the CPU).  transpose = 0: out[out_idx] += in[in_idx] W_k;  transpose = 1: out[in_idx] += in[out_idx] W_k
 * (roles of the two map columns swapped, as the transposed conv of build_blocks.py:39-52 calls it).
 * kernel / grad_kernel: [k, cin, cout].  Outputs are fully written (the reference pre-zeroes and accumulates:
 * same result).  Everything else lives in `workspace` (u2mkd_convolution_workspace_bytes); cin and cout must be
 * multiples of 4.  The rulebook is re-laid into the pair schedule on the fly (one scatter kernel), then runs
 * on u2mkd_conv_forward_pairs + u2mkd_pairs_gather_sum / u2mkd_conv_wgrad_pairs (deterministic).  grad_in_feat
 * or grad_kernel may be NULL (skipped).                                                                   */
size_t u2mkd_convolution_workspace_bytes(int64_t n_in_rows, int64_t n_out_rows, int32_t cin, int32_t cout,
                                         const int32_t *nbsizes_host /*[k] HOST*/, int32_t k);
int u2mkd_convolution_forward(const float *in_feat /*[n_in_rows,cin]*/, int64_t n_in_rows, int32_t cin,
                              float *out_feat /*[n_out_rows,cout]*/, int64_t n_out_rows, int32_t cout,
                              const float *kernel /*[k,cin,cout]*/, const int32_t *nbmaps /*[P,2] device*/,
                              const int32_t *nbsizes_host /*[k] HOST*/, int32_t k, int32_t transpose, void *workspace,
                              size_t workspace_bytes, u2mkd_stream_t s);
int u2mkd_convolution_backward(const float *in_feat /*[n_in_rows,cin]*/, int64_t n_in_rows, int32_t cin,
                               float *grad_in_feat /*[n_in_rows,cin] or NULL*/, const float *grad_out_feat /*[n_out_rows,cout]*/,
                               int64_t n_out_rows, int32_t cout, const float *kernel /*[k,cin,cout]*/,
                               float *grad_kernel /*[k,cin,cout] or NULL*/, const int32_t *nbmaps, const int32_t *nbsizes_host,
                               int32_t k, int32_t transpose, void *workspace, size_t workspace_bytes, u2mkd_stream_t s);

/* ---- point <-> voxel ---------------------------------------------------
 * replace torchsparse.backend.count_cuda, voxelize_forward/backward_cuda,
 * devoxelize_forward/backward_cuda (F.spcount / spvoxelize / spdevoxelize,
 * core/models/utils.py:22-26,51,58,99,111) and F.calc_ti_weights (:94).    */
int u2mkd_count(const int32_t *idx /*[n]*/, int64_t n, int32_t *counts /*[num] pre-zeroed*/, int64_t num,
                u2mkd_stream_t s);
int u2mkd_voxelize_forward(const float *feats /*[n,c]*/, const int32_t *idx /*[n]*/, const int32_t *counts /*[nv]*/,
                           int64_t n, int64_t nv, int32_t c, float *out /*[nv,c] pre-zeroed*/, u2mkd_stream_t s);
int u2mkd_voxelize_backward(const float *grad_out /*[nv,c]*/, const int32_t *idx, const int32_t *counts, int64_t n,
                            int64_t nv, int32_t c, float *grad_feats /*[n,c]*/, u2mkd_stream_t s);
int u2mkd_devoxelize_forward(const float *feats /*[nv,c]*/, const int32_t *idx /*[n,8]*/, const float *w /*[n,8]*/,
                             int64_t n, int32_t c, float *out /*[n,c]*/, u2mkd_stream_t s);
int u2mkd_devoxelize_backward(const float *grad_out /*[n,c]*/, const int32_t *idx /*[n,8]*/, const float *w /*[n,8]*/,
                              int64_t n, int64_t nv, int32_t c, float *grad_feats /*[nv,c] pre-zeroed*/,
                              u2mkd_stream_t s);
/* Deterministic form of the two scatters (voxelize forward, devoxelize backward): entries
 * sorted by destination row (CSR: seg_offsets [nv+1]); out[v] = sum_e w[e] * src[row[e]]
 * (entry_w NULL = 1); mean != 0: out[v] = sum_e src[row[e]] / len(v), each term divided before it is
 * added, in entry order -- torchsparse's voxelize arithmetic, bit for bit.  No atomics.            */
int u2mkd_segment_sum(const float *src /*[*,c]*/, int32_t c, const int32_t *entry_row /*[E]*/,
                      const float *entry_w /*[E] or NULL*/, const int32_t *seg_offsets /*[nv+1]*/, int64_t nv,
                      int32_t mean, float *out /*[nv,c]*/, u2mkd_stream_t s);
/* The row movers of the point <-> voxel transfers on BF16 rows (BASELINE.json configs[4]; torchsparse's voxelize /
 * devoxelize functions cast to half under autocast like its conv): feature rows in and out are bf16 [., c], c a multiple
 * of 4, sums in fp32 in the same fixed order, one rounding at the store.  u2mkd_segment_sum_bf16 serves voxelize forward
 * and devoxelize backward exactly as u2mkd_segment_sum does.                                                        */
int u2mkd_voxelize_backward_bf16(const void *grad_out /*bf16 [nv,c]*/, const int32_t *idx, const int32_t *counts, int64_t n,
                                 int64_t nv, int32_t c, void *grad_feats /*bf16 [n,c]*/, u2mkd_stream_t s);
int u2mkd_devoxelize_forward_bf16(const void *feats /*bf16 [nv,c]*/, const int32_t *idx /*[n,8]*/, const float *w /*[n,8]*/,
                                  int64_t n, int32_t c, void *out /*bf16 [n,c]*/, u2mkd_stream_t s);
int u2mkd_segment_sum_bf16(const void *src /*bf16 [*,c]*/, int32_t c, const int32_t *entry_row /*[E]*/,
                           const float *entry_w /*[E] or NULL*/, const int32_t *seg_offsets /*[nv+1]*/, int64_t nv,
                           int32_t mean, void *out /*bf16 [nv,c]*/, u2mkd_stream_t s);
/* The entry lists u2mkd_segment_sum walks, built in one call (csrc/csr.hip): entries e with key[e] in [0, nv) grouped by
 * key, ascending e inside a group -- bit-identical to a stable argsort by key, by a counting sort (histogram, scan,
 * placement, per-segment sort of the entry ids).  order [n_entries] (the first seg[nv] entries are live, the rest 0),
 * seg [nv + 1]; keys outside [0, nv) are dropped.  workspace: u2mkd_csr_workspace_bytes(n_entries, nv) bytes.      */
size_t u2mkd_csr_workspace_bytes(int64_t n_entries, int64_t nv);
int u2mkd_csr_build(const int32_t *keys /*[n_entries]*/, int64_t n_entries, int64_t nv, void *workspace,
                    int32_t *order /*[n_entries]*/, int32_t *seg /*[nv+1]*/, u2mkd_stream_t s);
/* The grouping the backward of u2mkd_devoxelize_forward sums over (core/models/utils.py:70-118 -> torchsparse v1.4.0
 * devoxelize_backward_cuda, float atomics there): the 8 n (point, corner) entries with a voxel and a non-zero weight grouped by
 * voxel row, ascending entry id inside a voxel, as (entry_row = point, entry_w = weight) [8 n] + seg [nv + 1] for
 * u2mkd_segment_sum -- keys, u2mkd_csr_build and the gather of rows / weights in one call.                               */
size_t u2mkd_devoxelize_plan_workspace_bytes(int64_t n, int64_t nv);
int u2mkd_devoxelize_plan(const int32_t *idx8 /*[n,8]*/, const float *w8 /*[n,8]*/, int64_t n, int64_t nv, void *workspace,
                          int32_t *entry_row /*[8n]*/, float *entry_w /*[8n]*/, int32_t *seg /*[nv+1]*/, u2mkd_stream_t s);
int u2mkd_ti_weights(const float *coords /*[n,4] float (x,y,z,b)*/, const int64_t *idx_kn /*[8,n]*/, int64_t n,
                     float scale, float *w_n8 /*[n,8]*/, int32_t *idx_n8 /*[n,8]*/, u2mkd_stream_t s);
/* (The coherence probe of round 5 -- tools/dbg_stale_probe.py, U2MKD_DEBUG_TI_PROBE=1 -- lives behind -DU2MKD_DEBUG_PROBE in
 * csrc/voxel.hip and is built by tools/build_variant.sh into a library of its own; the shipped library carries neither its
 * device-side logs nor its four read-out entries.) */

/* ---- the pixel head's full-resolution tail at the pixels that are read (csrc/pixhead.hip) -----------------------------
 * Replaces the dense evaluation of  Feature_Fetch(classifier_pix(upsample(x, image size)))  (the final F.interpolate of
 * core/models/image_branch/swiftnet.py forward_up, the BNReluConv `classifier_pix` of
 * core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py and Feature_Fetch, core/models/fusion_blocks.py:257-278):
 *   u2mkd_up_plan          sample point * 4 + corner: the 4 low-resolution bilinear sources (align_corners; slots 4..7
 *                          unused) of full-resolution pixel idx8[point][corner] (rows (image * H + y) * W + x, -1 = none)
 *                          as rows (image * h + i) * w + j; rh = (h - 1) / (H - 1), rw likewise, in fp32
 *   u2mkd_upbn_stats       partial [n_img * c][chunks][2]: per plane and row chunk the sums of the UP-SAMPLED map shifted
 *                          by the channel's first element, taken on the low-resolution map x [n_img, c, h, w]:
 *                          sum a_i b_j (x - K) and sum (x - K) (Ay (x - K) Ax); a [h], b [w] = column sums of the two
 *                          interpolation matrices, ay [3][h], ax [3][w] = lower / main / upper diagonal of Wy^T Wy, Wx^T Wx
 *   u2mkd_upbn_dense_grad  dx = c0[ch] a_i b_j + c1[ch] (Ay x Ax)[i][j]: what the BatchNorm backward's dense terms
 *                          (c0 + c1 U on every up-sampled pixel) send back through the up-sampling                      */
int u2mkd_up_plan(const int32_t *idx8 /*[n,8]*/, int64_t n, int32_t H, int32_t W, int32_t h, int32_t w, float rh, float rw,
                  int32_t *idx_out /*[4n,8]*/, float *w_out /*[4n,8]*/, u2mkd_stream_t s);
int u2mkd_upbn_stats(const float *x, int32_t n_img, int32_t c, int32_t h, int32_t w, const float *a, const float *b,
                     const float *ay, const float *ax, int32_t rows_per_chunk, float *partial, u2mkd_stream_t s);
int u2mkd_upbn_dense_grad(const float *x, int32_t n_img, int32_t c, int32_t h, int32_t w, const float *a, const float *b,
                          const float *ay, const float *ax, const float *c0 /*[c]*/, const float *c1 /*[c]*/,
                          int32_t rows_per_chunk, float *dx, u2mkd_stream_t s);

/* F.interpolate(mode='bilinear', align_corners=True) of NCHW maps, optionally + skip (the decoder's `_up(x) + skip`,
 * core/models/image_branch/swiftnet.py _Upsample.forward): x [planes, h, w] -> y [planes, H, W], rh = (h - 1) / (H - 1) in
 * fp32 (rw likewise).  Backward: a deterministic gather; taps_y [h][2] = (first output row, number of output rows) that
 * read input row i, wy [h][8] their weights (0 beyond the count), taps_x / wx likewise for columns -- built by the
 * caller from the same index arithmetic.  tiled != 0: the window of g of a 16 x 64 tile of inputs is staged in LDS
 * (the caller has checked that every tile's window fits 40 rows x 136 columns and planes <= 65535)                                                                              */
int u2mkd_up_bilinear_forward(const float *x, const float *skip /*or NULL*/, int64_t planes, int32_t h, int32_t w, int32_t H,
                              int32_t W, float rh, float rw, float *y, u2mkd_stream_t s);
int u2mkd_up_bilinear_backward(const float *g, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W, const int32_t *taps_y,
                               const float *wy, const int32_t *taps_x, const float *wx, int32_t tiled, float *dx, u2mkd_stream_t s);

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the SwiftNet stem (core/models/image_branch/swiftnet.py): x
 * [planes, h, w] -> y [planes, oh, ow] with oh = (h - 1) / 2 + 1; `code` = one byte per output, the position of the
 * maximum inside its 3 x 3 window (first maximum in row-major order, as torch); backward: dx [planes, h, w], a gather
 * (no atomics).  planes <= 65535 (one grid row per plane), h * w < 2^31.                                              */
int u2mkd_maxpool3s2_forward(const float *x, int64_t planes, int32_t h, int32_t w, float *y, uint8_t *code, u2mkd_stream_t s);
int u2mkd_maxpool3s2_backward(const float *dy, const uint8_t *code, int64_t planes, int32_t h, int32_t w, float *dx,
                              u2mkd_stream_t s);

/* out[b][j][i] = in[b][i][j] (fp32): the NCHW <-> channel-last-rows copies around the point <-> pixel gathers
 * (`features.permute(...)` of core/models/fusion_blocks.py:241-254, tsd_full.py:482-495) as an LDS-tiled transpose      */
int u2mkd_transpose_batched(const float *in, float *out, int32_t batch, int32_t rows, int32_t cols, u2mkd_stream_t s);
/* the same with every element multiplied by `scale` */
int u2mkd_transpose_batched_scaled(const float *in, float *out, int32_t batch, int32_t rows, int32_t cols, float scale,
                                   u2mkd_stream_t s);

/* LiDAR -> camera scatter, the combination of its grids (spvcnn_swiftnet18_spformer_tsd_full.py:448-478: the per-scale pixel
 * means are up-sampled to the feature map's size -- F.interpolate(bilinear, align_corners=True) -- and averaged).
 *   u2mkd_l2c_combine_forward   out_rows[img][y][x][:] = g0[img][y][x][:] + sum_{s >= 1} bilinear(g_s[img])(y, x)[:]
 *                               g0 [n_img, h, w, c] and g_s [n_img, ch_s, cw_s, c] channel-last rows (the segment means),
 *                               out_rows [n_img*h*w, c]; the caller's scaled transpose (1 / n_grids) makes the NCHW map.
 *   u2mkd_l2c_combine_backward  d_grid[img][i][j][:] = sum over the (y, x) that read cell (i, j) of weight * g_rows[img][y][x][:]
 *                               (the exact adjoint of the forward's taps, a gather in fixed order: reproducible; torch's
 *                               up-sampling gradient adds atomically); the full-resolution grid's gradient is g_rows itself. */
int u2mkd_l2c_combine_forward(const float *g0, const float *g1, const float *g2, const float *g3, int32_t n_grids, int32_t n_img,
                              int32_t h, int32_t w, int32_t c, int32_t ch1, int32_t cw1, int32_t ch2, int32_t cw2, int32_t ch3,
                              int32_t cw3, float *out_rows, u2mkd_stream_t s);
int u2mkd_l2c_combine_backward(const float *g_rows, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t ch, int32_t cw,
                               float *d_grid, u2mkd_stream_t s);

/* One fusion stage's camera -> LiDAR select and pseudo-feature loss (spvcnn_swiftnet18_spformer_tsd_full.py:489-498):
 * out = fov ? gathered : pseudo (rows), loss = mean over the fov rows of (pseudo - gathered)^2 (nn.MSELoss on the
 * boolean-indexed rows, the target detached) -- one pass forward (+ a one-workgroup merge of the per-workgroup partial
 * sums in a fixed order), one pass backward instead of ~30 element-wise torch launches per stage.
 * fov: uint8 [n]; partial: 2 * u2mkd_select_mse_partials() floats; stats [2] = {loss, 2 / max(count * c, 1)}.
 * backward: d_gathered = fov ? g_out : 0 (NULL: not wanted), d_pseudo = fov ? g_loss * stats[1] * (pseudo - gathered) : g_out;
 * g_out / g_loss NULL = zero.                                                                                         */
int32_t u2mkd_select_mse_partials(void);
int u2mkd_select_mse_forward(const float *gathered, const float *pseudo, const uint8_t *fov, int64_t n, int32_t c, float *out,
                             float *partial, float *stats, u2mkd_stream_t s);
int u2mkd_select_mse_backward(const float *g_out, const float *g_loss, const float *stats, const float *gathered,
                              const float *pseudo, const uint8_t *fov, int64_t n, int32_t c, float *d_gathered, float *d_pseudo,
                              u2mkd_stream_t s);
/* ---- Lovasz-softmax, classes = 'present' (csrc/lovasz.hip; core/criterions.py:40-52, 73-101) ------------------------------
 * The element-wise chains around the sort of the per-class errors (the sort itself stays the caller's: any ascending sort of
 * `keys` [c * n] returning positions).  probas [n, c] f32 (softmax output), labels [n] int64, rows with label == ignore_index
 * do not count.  Order of calls: errors -> (sort keys) -> gather -> (inclusive prefix sum of fg_sorted over all c * n entries,
 * int64) -> terms -> ... -> backward.
 *   u2mkd_lovasz_errors    errors [c, n] = |fg - p| (-1 on ignored rows), keys [c, n] f64 = 4 class - error
 *   u2mkd_lovasz_gather    fg_sorted [c * n] int32: the foreground flag of every sorted entry (perm = sorted positions)
 *   u2mkd_lovasz_terms     jgrad [c * n] = the Jaccard gradient of every sorted entry (lovasz_grad), stats [2 + c] = (loss,
 *                          1 / max(#present classes, 1), present flag per class); partial: u2mkd_lovasz_partials(n, c) floats
 *   u2mkd_lovasz_backward  d_probas [n, c] = g_out * d loss / d probas (every element written, no atomics)                 */
int u2mkd_lovasz_errors(const float *probas, const int64_t *labels, int32_t ignore_index, int64_t n, int32_t c, float *errors,
                        double *keys, u2mkd_stream_t s);
int u2mkd_lovasz_gather(const int64_t *perm, const int64_t *labels, int32_t ignore_index, int64_t n, int32_t c,
                        int32_t *fg_sorted, u2mkd_stream_t s);
int64_t u2mkd_lovasz_partials(int64_t n, int32_t c);
int u2mkd_lovasz_terms(const int64_t *perm, const float *errors, const int64_t *csum, const int32_t *fg_sorted, int64_t n,
                       int32_t c, float *jgrad, float *partial, float *stats, u2mkd_stream_t s);
int u2mkd_lovasz_backward(const float *g_out, const float *stats, const int64_t *perm, const float *jgrad, const float *probas,
                          const int64_t *labels, int32_t ignore_index, int64_t n, int32_t c, float *d_probas, u2mkd_stream_t s);
/* nn.CrossEntropyLoss(ignore_index, mean over the valid rows) on [n, c] logits (core/criterions.py:167-174: the `ce` half of
 * MixLovaszCrossEntropy; torch: log_softmax + nll_loss, two single-workgroup reductions of 57 + 76 us at 80 000 x 17).
 * u2mkd_ce_forward: lse [n] (kept for the backward), stats = {mean loss over the valid rows, 1 / #valid}; partial:
 * u2mkd_ce_partials(n) floats.  u2mkd_ce_backward: dx [n, c] = g_out[0] / #valid * (softmax(x) - onehot), 0 on ignored rows. */
int64_t u2mkd_ce_partials(int64_t n);
int u2mkd_ce_forward(const float *x, const int64_t *labels, int32_t ignore_index, int64_t n, int32_t c, float *lse, float *partial,
                     float *stats, u2mkd_stream_t s);
int u2mkd_ce_backward(const float *g_out, const float *stats, const float *x, const float *lse, const int64_t *labels,
                      int32_t ignore_index, int64_t n, int32_t c, float *dx, u2mkd_stream_t s);
/* nn.KLDivLoss(reduction='batchmean') on log_softmax(s_logits) against softmax(t_logits[t_index]) (the `kl` term of the KD step,
 * core/nusc_trainers.py:330-336, with the teacher -> student row re-indexing of :288-324 folded in: t_index [n] int64 or NULL =
 * identity).  u2mkd_kl_forward: rows [n, 3] = (lse_s, lse_t, sum p_t) kept for the backward, partial: u2mkd_ce_partials(n) floats,
 * stats[0] = the loss.  u2mkd_kl_backward: ds [n, c] = g_out[0] / n * (softmax(s) * sum p_t - p_t).                          */
int u2mkd_kl_forward(const float *s_logits, const float *t_logits, const int64_t *t_index, int64_t n, int32_t c, float *rows,
                     float *partial, float *stats, u2mkd_stream_t s);
int u2mkd_kl_backward(const float *g_out, const float *s_logits, const float *t_logits, const int64_t *t_index, const float *rows,
                      int64_t n, int32_t c, float *ds, u2mkd_stream_t s);
/* ---- point <-> pixel index plans of the LiDAR / camera fusion (csrc/fusion.hip) --------------------------------------
 * replace the index arithmetic of the reference's Python loops over (sample, camera, scale): Feature_Gather + the
 * per-camera masked overwrite (core/models/fusion_blocks.py:241-254, spvcnn_swiftnet18_spformer_tsd_full.py:482-495) and
 * the multi-scale pixel mean (tsd_full.py:448-478).  One sample per call (pixel_coords [ncam, n, 2] f32 in [-1, 1],
 * mask [ncam, n] bytes); the outputs feed u2mkd_devoxelize_forward (4 weighted corners per point) and u2mkd_csr_build /
 * u2mkd_segment_sum (entries grouped by pixel and by point).
 *   u2mkd_c2l_plan   idx8 / w8 [n, 8]: the bilinear corners (align_corners, zero padding) of every point in the [h, w]
 *                    map of the LAST camera that sees it, as rows (sample * ncam + cam) * h * w + y * w + x; -1 / 0 else
 *   u2mkd_l2c_keys   entry e0 + cam * n + i: pix = its pixel of a [ch, cw] grid, key_d = pix or -1 (not seen),
 *                    key_s = row0 + i or -1 (may be NULL), row = row0 + i (may be NULL)
 *   u2mkd_l2c_finish the forward list (by pixel: source row, 1 / points in the pixel) and the backward list (by point:
 *                    pixel, the same weight) from the two grouped orders of u2mkd_csr_build                          */
int u2mkd_c2l_plan(const float *pixel_coords, const uint8_t *mask, int32_t ncam, int64_t n, int32_t sample, int32_t h,
                   int32_t w, int32_t *idx8 /*[n,8]*/, float *w8 /*[n,8]*/, u2mkd_stream_t s);
int u2mkd_l2c_keys(const float *pixel_coords, const uint8_t *mask, int32_t ncam, int64_t n, int32_t sample, int64_t row0,
                   int64_t e0, int32_t ch, int32_t cw, int32_t *pix, int32_t *key_d, int32_t *key_s, int32_t *row,
                   u2mkd_stream_t s);
int u2mkd_l2c_finish(const int32_t *order_d, const int32_t *seg_d, const int32_t *order_s, const int32_t *pix,
                     const int32_t *row, int64_t n_entries, int32_t *fwd_row, float *fwd_w, int32_t *bwd_pix, float *bwd_w,
                     u2mkd_stream_t s);

/* ---- BatchNorm over feature rows (+ fused ReLU) ------------------------------
 * replaces spnn.BatchNorm + spnn.ReLU (nn.BatchNorm1d / nn.ReLU over SparseTensor.feats,
 * core/models/build_blocks.py:30-31,48-49,64-65,71,77).  Same statistics as
 * nn.BatchNorm1d: biased variance for normalisation, unbiased for running_var,
 * running = (1 - momentum) * running + momentum * batch.  Deterministic (slab partials in
 * `partial`, [u2mkd_bn_num_slabs(n), 2, c] floats, merged in slab order).
 * gamma / beta / running_* may be NULL.  relu != 0 fuses max(., 0) (backward re-derives the
 * mask from x).                                                                          */
int64_t u2mkd_bn_num_slabs(int64_t n);
int u2mkd_bn_train_forward(const float *x /*[n,c]*/, int64_t n, int32_t c, const float *gamma, const float *beta,
                           float eps, float momentum, float *running_mean, float *running_var, int32_t relu,
                           float *partial, float *mean /*[c] out*/, float *invstd /*[c] out*/, float *y /*[n,c]*/,
                           u2mkd_stream_t s);
/* the same with nn.BatchNorm's step counter: *num_batches_tracked (device int64, may be NULL) += 1 in the
 * statistics kernel instead of a one-element launch of its own per layer and step                        */
/* Train-mode forward FROM slab partials computed elsewhere: `partial` [ceil(n / slab_rows)][2][c] = per slab of slab_rows
 * consecutive rows the per-channel (mean, centred second moment), as u2mkd_pairs_gather_sum_stats writes them in the producing
 * convolution's store.  Merge (running statistics, step counter) + normalise (+ residual, + ReLU): the statistics pass of
 * u2mkd_bn_train_forward_res is not run.  fp32 rows.                                                                        */
int u2mkd_bn_train_forward_from_partial(const float *x, const float *res, int64_t n, int32_t c, const float *gamma,
                                        const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                        int64_t *num_batches_tracked, int32_t relu, const float *partial, int32_t slab_rows,
                                        float *mean /*[c] out*/, float *invstd /*[c] out*/, float *y, u2mkd_stream_t s);
int u2mkd_bn_train_forward_counted(const float *x /*[n,c]*/, int64_t n, int32_t c, const float *gamma, const float *beta,
                                   float eps, float momentum, float *running_mean, float *running_var,
                                   int64_t *num_batches_tracked, int32_t relu, float *partial, float *mean /*[c] out*/,
                                   float *invstd /*[c] out*/, float *y /*[n,c]*/, u2mkd_stream_t s);
int u2mkd_bn_eval_forward(const float *x, int64_t n, int32_t c, const float *gamma, const float *beta, float eps,
                          const float *running_mean, const float *running_var, int32_t relu, float *invstd /*[c] out*/,
                          float *y, u2mkd_stream_t s);
/* The tail of a ResidualBlock (core/models/build_blocks.py:80-83: relu(net(x) + downsample(x))) in the BatchNorm
 * passes: y = relu(bn(x) + res) forward; backward masks dy with (bn(x) + res > 0), returns it as dres (the gradient
 * of the residual branch) and continues with the BatchNorm gradients -- one read of res instead of an add kernel,
 * a ReLU kernel and their two backward kernels.  res / dres [n, c]; relu must be set.                         */
int u2mkd_bn_train_forward_res(const float *x, const float *res, int64_t n, int32_t c, const float *gamma, const float *beta,
                               float eps, float momentum, float *running_mean, float *running_var,
                               int64_t *num_batches_tracked, int32_t relu, float *partial, float *mean, float *invstd,
                               float *y, u2mkd_stream_t s);
int u2mkd_bn_eval_forward_res(const float *x, const float *res, int64_t n, int32_t c, const float *gamma, const float *beta,
                              float eps, const float *running_mean, const float *running_var, int32_t relu,
                              float *invstd /*[c] out*/, float *y, u2mkd_stream_t s);
int u2mkd_bn_backward_res(const float *dy, const float *x, const float *res, int64_t n, int32_t c, const float *mean,
                          const float *invstd, const float *gamma, const float *beta, int32_t relu, int32_t training,
                          float *partial, float *dgamma, float *dbeta, float *dx, float *dres, u2mkd_stream_t s);
int u2mkd_bn_backward(const float *dy, const float *x, int64_t n, int32_t c, const float *mean, const float *invstd,
                      const float *gamma, const float *beta, int32_t relu, int32_t training, float *partial,
                      float *dgamma /*[c]*/, float *dbeta /*[c]*/, float *dx /*[n,c]*/, u2mkd_stream_t s);

/* SyncBatchNorm (utils.py:138-220, train_spformer.py:79) in pieces, so the caller can put ONE small
 * collective between them: local (mean, M2, count) -> all_gather -> merge in rank order (Chan) ->
 * normalise(+ReLU); backward: local (sum dy', sum dy'*xhat) -> all_reduce -> apply with the global
 * count.  stats rows are [2c+1] = mean[c], M2[c], count.                                        */
int u2mkd_bn_local_stats(const float *x, int64_t n, int32_t c, float *partial /*[slabs,2,c]*/, float *stats /*[2c+1]*/,
                         u2mkd_stream_t s);
int u2mkd_bn_merge_stats(const float *gathered /*[world,2c+1]*/, int32_t world, int32_t c, float eps, float momentum,
                         float *running_mean, float *running_var, float *mean /*[c]*/, float *invstd /*[c]*/,
                         float *total /*[1]*/, u2mkd_stream_t s);
int u2mkd_bn_apply(const float *x, int64_t n, int32_t c, const float *mean, const float *invstd, const float *gamma,
                   const float *beta, int32_t relu, float *y, u2mkd_stream_t s);
int u2mkd_bn_backward_local(const float *dy, const float *x, int64_t n, int32_t c, const float *mean,
                            const float *invstd, const float *gamma, const float *beta, int32_t relu,
                            float *partial /*[slabs,2,c]*/, float *sums /*[2c]: dbeta, dgamma*/, u2mkd_stream_t s);
int u2mkd_bn_backward_apply(const float *dy, const float *x, int64_t n, int32_t c, const float *total_n /*[1] device*/,
                            const float *mean, const float *invstd, const float *gamma, const float *beta,
                            int32_t relu, const float *sums /*[2c] over all ranks*/, float *dx, u2mkd_stream_t s);

/* the same three pieces with the residual branch of a ResidualBlock (build_blocks.py:80-83 under SyncBatchNorm):
 * y = relu(bn(x) + res) in the apply pass; the backward recomputes the mask from x and res, dres = the masked dy */
int u2mkd_bn_apply_res(const float *x, const float *res, int64_t n, int32_t c, const float *mean, const float *invstd,
                       const float *gamma, const float *beta, int32_t relu, float *y, u2mkd_stream_t s);
int u2mkd_bn_backward_local_res(const float *dy, const float *x, const float *res, int64_t n, int32_t c, const float *mean,
                                const float *invstd, const float *gamma, const float *beta, int32_t relu, float *partial,
                                float *sums, u2mkd_stream_t s);
int u2mkd_bn_backward_apply_res(const float *dy, const float *x, const float *res, int64_t n, int32_t c, const float *total_n,
                                const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                                const float *sums, float *dx, float *dres, u2mkd_stream_t s);
int u2mkd_bn_apply_res_bf16(const void *x, const void *res, int64_t n, int32_t c, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, int32_t relu, void *y, u2mkd_stream_t s);
int u2mkd_bn_backward_local_res_bf16(const void *dy, const void *x, const void *res, int64_t n, int32_t c, const float *mean,
                                     const float *invstd, const float *gamma, const float *beta, int32_t relu, float *partial,
                                     float *sums, u2mkd_stream_t s);
int u2mkd_bn_backward_apply_res_bf16(const void *dy, const void *x, const void *res, int64_t n, int32_t c, const float *total_n,
                                     const float *mean, const float *invstd, const float *gamma, const float *beta,
                                     int32_t relu, const float *sums, void *dx, void *dres, u2mkd_stream_t s);

/* Two launches fewer per synchronising BatchNorm and pass (113 such layers per KD step at N > 1):
 * u2mkd_bn_merge_stats_counted = u2mkd_bn_merge_stats + nn.BatchNorm's `num_batches_tracked += 1` (int64 device scalar, may be
 * NULL) in the same launch (torch.nn.SyncBatchNorm bumps it with an add_ of its own, torch/nn/modules/batchnorm.py);
 * u2mkd_bn_backward_local_keep = u2mkd_bn_backward_local(_res)(_bf16) writing the local sums TWICE: `sums` [2c] goes into the
 * all_reduce in place, `keep` [2c] stays this rank's (the parameter gradients, which DDP averages) -- replaces the copy
 * between the two.  bf16_rows != 0: dy, x, res are bf16 rows; res may be NULL (no residual branch). */
int u2mkd_bn_merge_stats_counted(const float *gathered /*[world,2c+1]*/, int32_t world, int32_t c, float eps, float momentum,
                                 float *running_mean, float *running_var, float *mean, float *invstd, float *total,
                                 int64_t *num_batches_tracked, u2mkd_stream_t s);
int u2mkd_bn_backward_local_keep(const void *dy, const void *x, const void *res, int32_t bf16_rows, int64_t n, int32_t c,
                                 const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                                 float *partial, float *sums /*[2c]*/, float *keep /*[2c]*/, u2mkd_stream_t s);

/* The BatchNorm entries above on BF16 ROWS (BASELINE.json configs[4]; under autocast the reference's nn.BatchNorm1d takes
 * and returns half rows while its statistics stay fp32): x, res, y, dy, dx, dres are bf16 [n, c]; gamma, beta, running
 * statistics, mean, invstd, partial, dgamma, dbeta, stats and sums are fp32 exactly as above; every value is rounded to
 * bf16 once, at its store.  The fused ReLU mask is recomputed in the backward from the bf16 x with the forward's fp32
 * expression, so forward and backward agree on it.                                                                  */
int u2mkd_bn_train_forward_res_bf16(const void *x, const void *res, int64_t n, int32_t c, const float *gamma,
                                    const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                    int64_t *num_batches_tracked, int32_t relu, float *partial, float *mean, float *invstd,
                                    void *y, u2mkd_stream_t s);
int u2mkd_bn_eval_forward_res_bf16(const void *x, const void *res, int64_t n, int32_t c, const float *gamma, const float *beta,
                                   float eps, const float *running_mean, const float *running_var, int32_t relu,
                                   float *invstd, void *y, u2mkd_stream_t s);
int u2mkd_bn_backward_res_bf16(const void *dy, const void *x, const void *res, int64_t n, int32_t c, const float *mean,
                               const float *invstd, const float *gamma, const float *beta, int32_t relu, int32_t training,
                               float *partial, float *dgamma, float *dbeta, void *dx, void *dres, u2mkd_stream_t s);
int u2mkd_bn_local_stats_bf16(const void *x, int64_t n, int32_t c, float *partial, float *stats, u2mkd_stream_t s);
int u2mkd_bn_apply_bf16(const void *x, int64_t n, int32_t c, const float *mean, const float *invstd, const float *gamma,
                        const float *beta, int32_t relu, void *y, u2mkd_stream_t s);
int u2mkd_bn_backward_local_bf16(const void *dy, const void *x, int64_t n, int32_t c, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, int32_t relu, float *partial, float *sums,
                                 u2mkd_stream_t s);
int u2mkd_bn_backward_apply_bf16(const void *dy, const void *x, int64_t n, int32_t c, const float *total_n, const float *mean,
                                 const float *invstd, const float *gamma, const float *beta, int32_t relu, const float *sums,
                                 void *dx, u2mkd_stream_t s);

/* ---- SphereFormer / sptr window attention ---------------------------------------
 * replaces the extern "C" launchers of third_party/SparseTransformer/src/sptr:
 *   precompute_all_cuda_launcher            (precompute/precompute_cuda_kernel.h)
 *   dot_prod_with_idx_all_forward/_backward (rpe/relative_pos_encoding_cuda_kernel.h)
 *   attention_step2_with_rel_pos_value_forward/_backward (same header)
 *   attention_step1_backward_cuda_launcher  (attention/attention_cuda_kernel.h)
 * plus torch_cluster grid_cluster, torch_scatter segment_csr softmax and the index glue of
 * sptr/utils.py:49-95, sptr/modules.py:35-65.  No M-sized (pair) array exists: tokens are
 * sorted by window key and each (token, head) walks its window.
 *   keys     = grid_cluster key over (x,y,z,batch); lo4/hi4 = device min/max of (x,y,z,batch)
 *   ranges   = per SORTED position: first position and length of its window
 *   qc       = floor(((xyz - lo) % window) / quant) per sorted position (+ radial = xyz[:,2])
 * q,k,v,out,dq,dk,dv: [n,h,16] in ORIGINAL token order (q pre-scaled); lse: [n,h] in sorted
 * order; tables [L,3,h,16]; split_a > 0 selects the spherical branch (exponential radial
 * split + clamp to [0, 2*qgl-1]).                                                         */
int u2mkd_sptr_window_keys(const float *xyz /*[n,3]*/, const int32_t *batch /*[n]*/, int64_t n, const float *lo4,
                           const float *hi4, float sx, float sy, float sz, int64_t *keys /*[n]*/, u2mkd_stream_t s);
/* Both window plans of one SphereFormer block from one pass: the spherical coordinates of cart2sphere
 * (spherical_transformer.py:31-36, torch's fp32 arithmetic operation by operation), the bounds of both coordinate systems
 * and the grid_cluster keys of the cubic (windows cx, cy, cz) and the spherical (sx, sy, sz) branch (:206-213), two launches
 * instead of ~28 torch operators.  sphere [n,3]; bounds [16] = lo4 | hi4 of (x, y, z, batch), lo4 | hi4 of
 * (theta, beta, r, batch); keys as u2mkd_sptr_window_keys; workspace of u2mkd_sptr_plan_prepare_workspace_bytes().   */
size_t u2mkd_sptr_plan_prepare_workspace_bytes(void);
int u2mkd_sptr_plan_prepare(const float *xyz /*[n,3]*/, const int32_t *batch /*[n]*/, int64_t n, float cx, float cy, float cz,
                            float sx, float sy, float sz, float *sphere, float *bounds, int64_t *keys_cubic,
                            int64_t *keys_sphere, void *workspace, u2mkd_stream_t s);
int u2mkd_sptr_window_ranges(const int64_t *sorted_keys, int64_t n, int32_t *wstart /*[n]*/, int32_t *wlen /*[n]*/,
                             u2mkd_stream_t s);
int u2mkd_sptr_quant_coords(const float *xyz /*[n,3]*/, const int32_t *sort_idx /*[n]*/, int64_t n, const float *lo,
                            float wx, float wy, float wz, float qx, float qy, float qz, int32_t *qc /*[n,3]*/,
                            float *radial /*[n] or NULL*/, u2mkd_stream_t s);
int u2mkd_sptr_attention_forward(const float *q, const float *k, const float *v, const int32_t *sort_idx,
                                 const int32_t *wstart, const int32_t *wlen, const int32_t *qc, const float *radial,
                                 const float *tq, const float *tk, const float *tv, int32_t L, int32_t qgl,
                                 float split_a, int64_t n, int32_t h, int32_t hdim, float *out, float *lse,
                                 u2mkd_stream_t s);
/* Table gradients: per-token scalar histograms in private LDS strips, contracted with the token
 * vectors on the MFMA unit, one slab per wave in `workspace`, summed in a fixed order
 * (deterministic, no atomics; dtq/dtk/dtv are fully written, no pre-zeroing).  qc_span = host-known
 * bound on the quantised in-window coordinates of the affine axes (ceil(window / quant_size));
 * spans above 24 (never on the U2MKD configs: window / quant size = 24) are an error.          */
size_t u2mkd_sptr_backward_workspace_bytes(int64_t n, int32_t h, int32_t L);
int u2mkd_sptr_attention_backward(const float *q, const float *k, const float *v, const float *out, const float *dout,
                                  const float *lse, const int32_t *sort_idx, const int32_t *wstart,
                                  const int32_t *wlen, const int32_t *qc, const float *radial, const float *tq,
                                  const float *tk, const float *tv, int32_t L, int32_t qgl, float split_a,
                                  int32_t qc_span, int64_t n, int32_t h, int32_t hdim, float *delta /*[n,h] scratch*/,
                                  void *workspace, size_t workspace_bytes, float *dq, float *dk, float *dv, float *dtq,
                                  float *dtk, float *dtv, u2mkd_stream_t s);
/* The same two with row strides (in floats) and the query scale as arguments: q, k, v are read straight out of the
 * packed [N, 3, H, 16] output of the qkv projection (`qkv(feats).reshape(N, 3, H, C // H)`, q = qkv[:, 0] * scale:
 * core/models/sphereformer/spherical_transformer.py:192-205) -- a branch is a range of heads, pass the pointer of its
 * first head and ld_qkv = 3 * H * 16 -- the heads of a branch go to their columns of the [N, H * 16] attention
 * output (ld_out = H * 16: the torch.cat of the two branches, :228), and the backward writes d(qkv) in the packed
 * layout (ld_grad = 3 * H * 16; dq already multiplied by q_scale).  The contiguous entries above are these with
 * ld = h * 16 and q_scale = 1.                                                                                      */
int u2mkd_sptr_attention_forward_strided(const float *q, const float *k, const float *v, int64_t ld_qkv, float q_scale,
                                         const int32_t *sort_idx, const int32_t *wstart, const int32_t *wlen,
                                         const int32_t *qc, const float *radial, const float *tq, const float *tk,
                                         const float *tv, int32_t L, int32_t qgl, float split_a, int64_t n, int32_t h,
                                         int32_t hdim, float *out, int64_t ld_out, float *lse, u2mkd_stream_t s);
int u2mkd_sptr_attention_backward_strided(const float *q, const float *k, const float *v, int64_t ld_qkv, float q_scale,
                                          const float *out, const float *dout, int64_t ld_out, const float *lse,
                                          const int32_t *sort_idx, const int32_t *wstart, const int32_t *wlen,
                                          const int32_t *qc, const float *radial, const float *tq, const float *tk,
                                          const float *tv, int32_t L, int32_t qgl, float split_a, int32_t qc_span,
                                          int64_t n, int32_t h, int32_t hdim, float *delta /*[n,h] scratch*/,
                                          void *workspace, size_t workspace_bytes, float *dq, float *dk, float *dv,
                                          int64_t ld_grad, float *dtq, float *dtk, float *dtv, u2mkd_stream_t s);

/* dtq = dtk = dtv = NULL in the two backward entries: the slab sum (their last launch) is left to the caller --
 * u2mkd_sptr_table_reduce(workspace, the same n, h, L, split_a, ...) on any stream ordered behind the backward call.  The
 * relative-position tables are leaf parameters (spherical_transformer.py:126-134): the host side queues the sum on its
 * weight-gradient side stream and joins it at the end of the backward (u2mkd_amd/sptr/functional.py). */
int u2mkd_sptr_table_reduce(const void *workspace, int64_t n, int32_t h, int32_t L, float split_a, float *dtq, float *dtk,
                            float *dtv, u2mkd_stream_t s);

/* ---- the `sptr_cuda` extension, function for function (csrc/sptr_ops.hip) ----------------------------------------
 * The ten functions third_party/SparseTransformer/src/sptr/pointops_api.cpp:9-20 exports, for a caller that keeps
 * sptr's Python layer (sptr/functional.py) and its M = sum_w L_w^2 pair arrays: u2mkd_sptr_<name> replaces
 * sptr_cuda.<name>_cuda with the same arguments in the same order (at::Tensor -> its data pointer, + the stream).
 * Layouts are what the reference's launchers consume (i.e. AFTER the permutes sptr/functional.py applies), all
 * float32 / int32, contiguous; outputs the reference pre-zeroes must be pre-zeroed (table gradients are accumulated
 * with atomics, one per table entry and workgroup).  hdim <= 64, L <= 50 (the reference asserts L <= 50).
 * The product path (u2mkd_sptr_attention_forward/_backward above) does not call these.                            */
/* precompute/precompute.cpp:7-17 -- counts/offsets/sq_offsets [n](+1); index_0_offsets, index_1_offsets [N];
 * index_0, index_1 [M]: pair m = sq_offsets[w] + i L_w + t  <->  (query offsets[w] + i, key offsets[w] + t)        */
int u2mkd_sptr_precompute_all(int32_t N, int32_t n, uint32_t n_max, const int32_t *counts, const int32_t *offsets,
                              const int32_t *sq_offsets, int32_t *index_0_offsets, int32_t *index_1_offsets,
                              int32_t *index_0, int32_t *index_1, u2mkd_stream_t s);
/* attention/attention_cuda.cpp:7-33 -- q [h,d,N_q], k [h,d,N_k] -> attn [h,M];  grad_out [M,h], q/k [N,h,d] ->
 * grad_q, grad_k [N,h,d]                                                                                          */
int u2mkd_sptr_attention_step1_forward(int32_t N_q, int32_t N_k, int32_t M, int32_t h, int32_t hdim, uint32_t n_max,
                                       const float *q, const float *k, const int32_t *index0, const int32_t *index1,
                                       float *attn, u2mkd_stream_t s);
int u2mkd_sptr_attention_step1_backward(int32_t N, int32_t M, int32_t h, int32_t hdim, uint32_t n_max,
                                        const float *grad_out, const int32_t *index0, const int32_t *index0_offsets,
                                        const int32_t *index1, const int32_t *index1_offsets, const float *q,
                                        const float *k, float *grad_q, float *grad_k, u2mkd_stream_t s);
/* attention/attention_cuda.cpp:35-60 -- attn [M,h], v [N,h,d] -> output [N,h,d];  backward: grad_out [N,h,d],
 * v [h,d,N] -> grad_attn [M,h], grad_v [N,h,d] (see the note on the reference's layout slip in csrc/sptr_ops.hip) */
int u2mkd_sptr_attention_step2_forward(int32_t N, int32_t M, int32_t h, int32_t hdim, int32_t n_max, const float *attn,
                                       const float *v, const int32_t *index0_offsets, const int32_t *index1,
                                       float *output, u2mkd_stream_t s);
int u2mkd_sptr_attention_step2_backward(int32_t N, int32_t M, int32_t h, int32_t hdim, int32_t n_max,
                                        const float *grad_out, const int32_t *index0, const int32_t *index0_offsets,
                                        const int32_t *index1, const int32_t *index1_offsets, const float *attn,
                                        const float *v, float *grad_attn, float *grad_v, u2mkd_stream_t s);
/* rpe/relative_pos_encoding_cuda.cpp -- forward: q, k [h,d,N], tables [h,d,3,L], rel_idx [3,M] -> output [h,M]
 * (`_all` adds the q.k term); backward: grad_out [M,h], q, k [N,h,d], tables [L,3,h,d], rel_idx [M,3] ->
 * grad_q, grad_k [N,h,d], grad_table_q, grad_table_k [L,3,h,d]                                                    */
int u2mkd_sptr_dot_prod_with_idx_forward(int32_t N, int32_t M, int32_t h, int32_t hdim, int32_t n_max, int32_t L,
                                         const float *q, const int32_t *index_q, const int32_t *index_q_offsets,
                                         const float *k, const int32_t *index_k, const float *table_q,
                                         const float *table_k, const int32_t *rel_idx, float *output,
                                         u2mkd_stream_t s);
int u2mkd_sptr_dot_prod_with_idx_all_forward(int32_t N, int32_t M, int32_t h, int32_t hdim, int32_t n_max, int32_t L,
                                             const float *q, const int32_t *index_q, const int32_t *index_q_offsets,
                                             const float *k, const int32_t *index_k, const float *table_q,
                                             const float *table_k, const int32_t *rel_idx, float *output,
                                             u2mkd_stream_t s);
int u2mkd_sptr_dot_prod_with_idx_backward(int32_t N, int32_t M, int32_t h, int32_t hdim, int32_t n_max, int32_t L,
                                          const float *grad_out, const float *q, const int32_t *index_q_offsets,
                                          const float *k, const int32_t *index_k_offsets, const int32_t *index_k,
                                          const float *table_q, const float *table_k, const int32_t *rel_idx,
                                          float *grad_q, float *grad_k, float *grad_table_q, float *grad_table_k,
                                          u2mkd_stream_t s);
/* forward: attn [M,h], v [N,h,d], table [L,3,h,d], rel_idx [M,3] -> output [N,h,d];  backward: grad_out [N,h,d],
 * attn [M,h], v [h,d,N], table [h,d,3,L], rel_idx [3,M] -> grad_attn [M,h], grad_v [N,h,d], grad_table [L,3,h,d]  */
int u2mkd_sptr_attention_step2_with_rel_pos_value_forward(int32_t N, int32_t M, int32_t h, int32_t hdim, int32_t n_max,
                                                          const float *attn, const float *v,
                                                          const int32_t *index0_offsets, const int32_t *index1,
                                                          const float *table, const int32_t *rel_idx, float *output,
                                                          u2mkd_stream_t s);
int u2mkd_sptr_attention_step2_with_rel_pos_value_backward(int32_t N, int32_t M, int32_t h, int32_t hdim, int32_t L,
                                                           int32_t n_max, const float *grad_out, const int32_t *index0,
                                                           const int32_t *index0_offsets, const int32_t *index1,
                                                           const int32_t *index1_offsets, const float *attn,
                                                           const float *v, const float *table, const int32_t *rel_idx,
                                                           float *grad_attn, float *grad_v, float *grad_table,
                                                           u2mkd_stream_t s);

/* ---- BatchNorm2d over NCHW maps, fused with the ReLU / residual add that follow it (csrc/bn2d.hip) -----------------
 * Replaces nn.BatchNorm2d -> nn.ReLU (and bn2(conv2(.)) + identity -> ReLU of BasicBlock) in the SwiftNet-18 camera
 * branch and the LiDAR->camera fusion convs (core/models/image_branch/swiftnet.py:20-50,114-341; the L2C blocks of
 * core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py), i.e. torch.nn.functional.batch_norm +
 * relu / add on [b, c, h*w] fp32 maps.  y = relu?((x - mean) * invstd * gamma + beta [+ res]); training mode uses
 * the batch statistics (biased variance) and updates running_mean / running_var (momentum, unbiased variance) and
 * num_batches_tracked when they are given; gamma / beta / res / running_* may be NULL.  Deterministic.            */
size_t u2mkd_bn2d_workspace_bytes(int64_t b, int32_t c, int64_t hw);
int u2mkd_bn2d_train_forward(const float *x, const float *res, int64_t b, int32_t c, int64_t hw, const float *gamma,
                             const float *beta, float eps, float momentum, int32_t relu, float *running_mean,
                             float *running_var, int64_t *num_batches_tracked, void *workspace,
                             float *mean /*[c] out*/, float *invstd /*[c] out*/, float *y, u2mkd_stream_t s);
int u2mkd_bn2d_eval_forward(const float *x, const float *res, int64_t b, int32_t c, int64_t hw, const float *gamma,
                            const float *beta, float eps, int32_t relu, const float *running_mean,
                            const float *running_var, float *y, u2mkd_stream_t s);
/* dx, dgamma = sum dy' * xhat, dbeta = sum dy', dres = dy' (dy' = dy masked by the fused ReLU, recomputed from x and
 * res); batch_stats = 0: the statistics were constants (evaluation mode), dx = dy' * gamma * invstd.              */
int u2mkd_bn2d_backward(const float *dy, const float *x, const float *res, int64_t b, int32_t c, int64_t hw,
                        const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                        int32_t batch_stats, void *workspace, float *dgamma, float *dbeta, float *dx,
                        float *dres /*or NULL*/, u2mkd_stream_t s);

/* SyncBatchNorm2d in pieces (SparseSyncBatchNorm.convert_sync_batchnorm(model.model_s), train_lc_nusc_tsd_full.py:80, turns
 * the camera branch's BatchNorm2d layers into torch.nn.SyncBatchNorm): the caller puts ONE small collective between
 * them, exactly as for the feature-row pieces above -- local (mean, M2, count) -> all_gather [2c+1] ->
 * u2mkd_bn_merge_stats -> u2mkd_bn2d_apply (+ residual, ReLU fused); backward: local sums [2c] -> all_reduce ->
 * u2mkd_bn2d_backward_apply with the global element count (device float).  workspace: u2mkd_bn2d_workspace_bytes.   */
int u2mkd_bn2d_local_stats(const float *x, int64_t b, int32_t c, int64_t hw, void *workspace, float *stats /*[2c+1]*/,
                           u2mkd_stream_t s);
int u2mkd_bn2d_apply(const float *x, const float *res, int64_t b, int32_t c, int64_t hw, const float *mean,
                     const float *invstd, const float *gamma, const float *beta, int32_t relu, float *y, u2mkd_stream_t s);
int u2mkd_bn2d_backward_local(const float *dy, const float *x, const float *res, int64_t b, int32_t c, int64_t hw,
                              const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                              void *workspace, float *sums /*[2c]*/, u2mkd_stream_t s);
/* u2mkd_bn2d_backward_local with a second copy `keep` [2c] of the sums (as u2mkd_bn_backward_local_keep) */
int u2mkd_bn2d_backward_local_keep(const float *dy, const float *x, const float *res, int64_t b, int32_t c, int64_t hw,
                                   const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t relu,
                                   void *workspace, float *sums /*[2c]*/, float *keep /*[2c]*/, u2mkd_stream_t s);
int u2mkd_bn2d_backward_apply(const float *dy, const float *x, const float *res, int64_t b, int32_t c, int64_t hw,
                              const float *total_n /*[1] device*/, const float *mean, const float *invstd, const float *gamma,
                              const float *beta, int32_t relu, const float *sums /*[2c] over all ranks*/, float *dx,
                              float *dres /*or NULL*/, u2mkd_stream_t s);

/* ---- sptr window attention, tile form (csrc/sptr_tiles.hip) ---------------------------------------------------------------
 * u2mkd_sptr_attention_forward_strided's function and arguments (same reference kernels replaced: rpe/
 * relative_pos_encoding_cuda_kernel.cu:42-274, attention/attention_cuda_kernel.cu:4-112, sptr/utils.py:80-95) evaluated on
 * 16 x 16 (query, key) tiles: q . k and P V on v_mfma_f32_16x16x4_f32, the relative-position terms from per-token strips
 * q_i . Tq[r][ax], k_j . Tk[r][ax] (a pre-pass into `workspace`, u2mkd_sptr_tiles_workspace_bytes(n, h) bytes) by look-up,
 * the value tables through a per-query histogram.  Same out / lse; results differ from the per-pair kernels by rounding
 * (the table terms are summed in a different order). */
size_t u2mkd_sptr_tiles_workspace_bytes(int64_t n, int32_t h);
int u2mkd_sptr_attention_forward_tiles(const float *q, const float *k, const float *v, int64_t ld_qkv, float q_scale,
                                       const int32_t *sort_idx, const int32_t *wstart, const int32_t *wlen, const int32_t *qc,
                                       const float *radial, const float *tq, const float *tk, const float *tv, int32_t L,
                                       int32_t qgl, float split_a, int64_t n, int32_t h, int32_t hdim, float *out, int64_t ld_out,
                                       float *lse, void *workspace, size_t workspace_bytes, u2mkd_stream_t s);

/* ---- optimizer (csrc/optim.hip) --------------------------------------------------------------------------------------
 * torch.optim.SGD(momentum, nesterov, weight_decay) -- core/builder.py:663-669 -- over all parameters of a group in one launch,
 * element for element the operations and roundings of torch's multi-tensor path (torch/optim/sgd.py:_multi_tensor_sgd).
 * jobs: device int64 [n_jobs, 6] = (parameter, gradient or 0 = skip, momentum buffer, elements, index of the tensor's first
 * chunk, 1 = the buffer does not exist yet: b = g + wd p); a chunk = u2mkd_sgd_chunk_elements() elements, one workgroup each;
 * contract = 1: every `a + alpha * b` is one fused multiply-add (what ATen's kernels compile to), 0: two roundings. */
int32_t u2mkd_sgd_chunk_elements(void);
int u2mkd_sgd_batch(const int64_t *jobs, int32_t n_jobs, int64_t total_chunks, float lr, float momentum, float weight_decay,
                    int32_t nesterov, int32_t contract, u2mkd_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* U2MKD_HIP_H */
