/*
 * toda.h — C ABI of libtoda_hip.so, the MI355X (gfx950) implementation of the
 * sparse LiDAR-detection hot path of rasd3/TODA (an OpenPCDet fork).
 *
 * The reference reaches this arithmetic through the third-party `spconv`
 * package (not vendored, not pinned).  Every entry point below names the
 * reference call site it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *     small geometry arrays (range, vsize, shape, ksize, ...) are HOST arrays
 *     read during the call.
 *   - the caller owns all memory, including workspaces; nothing here allocates
 *     or frees device memory and nothing synchronises the device.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).
 *   - return value: 0 on success, a negative TODA_E* code on failure; the
 *     message is available from toda_last_error() (thread local).
 *   - sizes that live on the device (`*_dev`) are int32 scalars; a kernel that
 *     takes both `n` and `n_dev` uses min(n, *n_dev) rows when n_dev != NULL,
 *     so index-building phases can be chained without a host round trip.
 *   - indices are int32 rows of (b, z, y, x); spatial shapes are (D, H, W).
 *   - neighbour tables are k-major: nbr[k * n_rows + row], -1 = no neighbour,
 *     k = (kz * KY + ky) * KX + kx.
 *   - weights use the spconv-2 layout [Cout][kz][ky][kx][Cin]
 *     (pcdet/models/detectors/detector3d_template.py:337-348 converts v1 layouts).
 */
#ifndef TODA_H_
#define TODA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TODA_OK 0
#define TODA_EINVAL (-1)   /* bad argument / unsupported shape            */
#define TODA_EWORKSPACE (-2) /* workspace too small                       */
#define TODA_ELAUNCH (-3)  /* HIP launch or runtime error                 */
#define TODA_EFAULT (-4)   /* a kernel reported a fault (toda_device_fault) */

/* Bits of the device fault word: the kernels whose workgroups wait for each other inside one launch bound their spins;
 * a wait that gives up raises its bit and the launch finishes with invalid numbers instead of hanging the GPU. */
#define TODA_FAULT_BN2D 1u /* bn2d_fwd/bwd_split_kernel: partner never published */
#define TODA_FAULT_WINO 2u /* wino_fwd_ws_kernel: stream-K contributor never published */

const char* toda_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int toda_abi_version(void);
/* Reads and clears the fault word (host-mapped memory: no device synchronisation).  TODA_OK, or TODA_EFAULT with the
 * kernels named in toda_last_error().  toda_bn2d_* and toda_conv3x3_fwd poll it on entry as well, so a fault raised by
 * one launch is reported by the next call of that family at the latest; call it after a synchronisation point to learn
 * about the launches before it.  (No reference counterpart: the reference's cuDNN / spconv kernels never wait on each other.) */
int toda_device_fault(void);

/* ------------------------------------------------------------------------
 * Hard voxelisation.  Replaces spconv Point2VoxelCPU3d.point_to_voxel /
 * VoxelGenerator.generate as called from
 * pcdet/datasets/processor/data_processor.py:44-60,115-143.
 * Sequential semantics (voxel ids in order of first appearance, first
 * `max_pts` points kept per voxel, voxels past `max_voxels` dropped) are
 * reproduced exactly.  `points` is [n, c] fp32, xyz in columns 0..2.
 * Outputs sized for max_voxels: voxels [max_voxels, max_pts, c] (zero padded
 * for rows < *m_dev), coords_zyx [max_voxels, 3], num_pts [max_voxels].
 * ---------------------------------------------------------------------- */
size_t toda_voxelize_workspace_bytes(int n_points, int max_voxels);
int toda_voxelize_hard(const float* points, int n, int c,
                       const float* range_host /*[6] x0 y0 z0 x1 y1 z1*/,
                       const float* vsize_host /*[3] xyz*/,
                       const int32_t* grid_host /*[3] xyz cells*/,
                       int max_pts, int max_voxels,
                       float* voxels, int32_t* coords_zyx, int32_t* num_pts,
                       int32_t* m_dev, void* ws, size_t ws_bytes, void* stream);
/* The same for a whole batch in three launches: the per-sample voxelisation of the DataLoader workers
 * (data_processor.py:115-143) AND collate_batch's concatenation with the batch column prepended
 * (pcdet/datasets/dataset.py:161-178).  points_host[b] is the DEVICE address of sample b's first feature
 * column, rows `row_stride` floats apart (a [sum N, 1 + C] batch tensor is read in place: address of column 1,
 * stride 1 + C); n_points_host[b] <= n_cap, the per-sample capacity the workspace is laid out for.  Outputs
 * are the collated tensors, samples back to back, each clipped at max_voxels: voxels [sum M_b, max_pts, c],
 * coords_bzyx [sum M_b, 4], num_pts [sum M_b]; room for batch * min(max_voxels, n_cap) rows.
 * counts_dev [batch + 1] = M_0 .. M_{batch-1}, sum M_b.  ws_clean != 0: the workspace was used by this entry
 * point before with the same (batch, n_cap) - every call leaves its hash table empty - and nothing else wrote it;
 * 0: the call clears the table first.  After a failed call pass 0. */
size_t toda_voxelize_batch_workspace_bytes(int batch, int n_cap);
int toda_voxelize_batch(const float* const* points_host, const int32_t* n_points_host, int batch, int n_cap,
                        int c, int row_stride,
                        const float* range_host, const float* vsize_host, const int32_t* grid_host,
                        int max_pts, int max_voxels,
                        float* voxels, int32_t* coords_bzyx, int32_t* num_pts, int32_t* counts_dev,
                        void* ws, size_t ws_bytes, int ws_clean, void* stream);

/* MeanVFE: pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31.
 * out[v, :] = sum_p voxels[v, p, :] / max(num_pts[v], 1); num_pts is fp32
 * because load_data_to_gpu (pcdet/models/__init__.py:23-34) casts it. */
int toda_mean_vfe_fwd(const float* voxels, const float* num_pts, int m, int p, int c,
                      float* out, void* stream);
int toda_mean_vfe_bwd(const float* grad_out, const float* num_pts, int m, int p, int c,
                      float* grad_voxels, void* stream);

/* ------------------------------------------------------------------------
 * Grid index: a bitmap + popcount-rank dictionary over the (b, z, y, x)
 * lattice of one sparse level.  rank(coord) is the position of the site in
 * ascending ((b*D+z)*H+y)*W+x order, which is the canonical row order of
 * every index set this library generates.  Replaces spconv's hash table
 * (SparseConvTensor.indice_dict users, pcdet/models/backbones_3d/
 * spconv_backbone.py:77-117).
 * ---------------------------------------------------------------------- */
size_t toda_gridindex_bytes(int batch, const int32_t* shape_host /*[3] D H W*/);
/* Build the index of an existing coordinate list (any row order).
 * rowof[rank] = row.  Coordinates must be unique and in range. */
int toda_gridindex_from_coords(const int32_t* idx, int n, const int32_t* n_dev,
                               int batch, const int32_t* shape_host,
                               void* gi, int32_t* rowof, void* stream);
/* The same index for the VOXEL level, in O(sites): rank(coord) is NOT canonical here (ranks are handed out per
 * occupied 32-cell word by the site that set the word's lowest bit), which is all a level whose rows are in
 * caller order needs - rowof[rank] = row, toda_rulebook_subm / toda_rulebook_conv read it as before.  No sweep over
 * the lattice: the bitmap must be all zero on entry (gi_clean != 0: the caller guarantees it, e.g. through
 * toda_gridindex_clear after the previous use; 0: the call clears all of it first). */
int toda_gridindex_from_coords_unordered(const int32_t* idx, int n, const int32_t* n_dev,
                                         int batch, const int32_t* shape_host,
                                         void* gi, int32_t* rowof, int gi_clean, void* stream);
/* Put the bitmap words of the listed sites back to zero (O(sites) instead of a memset of the lattice). */
int toda_gridindex_clear(const int32_t* idx, int n, const int32_t* n_dev, int batch,
                         const int32_t* shape_host, void* gi, void* stream);
/* Build the index of the OUTPUT set of a strided sparse convolution
 * (spconv.SparseConv3d, spconv_backbone.py:14-15,113-114) and emit its
 * coordinates in canonical order.  idx_out has room for out_cap rows. */
int toda_gridindex_from_conv(const int32_t* idx_in, int n_in, const int32_t* n_in_dev,
                             int batch, const int32_t* shape_in_host,
                             const int32_t* ksize_host, const int32_t* stride_host,
                             const int32_t* pad_host, const int32_t* shape_out_host,
                             void* gi_out, int32_t* idx_out, int32_t* n_out_dev,
                             int out_cap, void* stream);
/* The same from the INPUT level's grid index instead of its coordinate list: output stationary (one thread per
 * 32-cell output word ORs the input rows that reach it), no atomics, no clearing of gi_out, two launches; the input
 * row count never enters, so the levels of a backbone chain without a host round trip.  in_rows_marked != 0: gi_in was
 * built by toda_gridindex_from_coords_unordered, which also keeps one byte per lattice row (b, z, y) - input rows without
 * a site are then skipped after one byte load (the voxel level: most of its 123 k rows). */
int toda_gridindex_from_bitmap(const void* gi_in, int batch, const int32_t* shape_in_host,
                               const int32_t* ksize_host, const int32_t* stride_host,
                               const int32_t* pad_host, const int32_t* shape_out_host,
                               void* gi_out, int32_t* idx_out, int32_t* n_out_dev,
                               int out_cap, int in_rows_marked, void* stream);

/* Rulebook of spconv.SubMConv3d (spconv_backbone.py:12,78): out sites = in
 * sites.  nbr[k*n + o] = input row at o + (k - centre) * dilation or -1.
 * rowof may be NULL when rows are already in canonical order.
 * pair_cnt[K] (device) receives the number of valid pairs per offset (cnt_zeroed != 0: the caller has zeroed it,
 * e.g. every table's counters of a plan with one memset). */
int toda_rulebook_subm(const int32_t* idx, int n, int batch, const int32_t* shape_host,
                       const int32_t* ksize_host, const int32_t* dilation_host,
                       const void* gi, const int32_t* rowof,
                       int32_t* nbr, int32_t* pair_cnt, int cnt_zeroed, void* stream);
/* Rulebook of spconv.SparseConv3d: nbr_o2i[k*n_out + o] = input row feeding
 * output o through offset k; nbr_i2o[k*n_in + i] = output row fed by input i
 * through offset k (used by dgrad).  idx_out / gi_in / rowof_in (optional, NULL = unknown): the output coordinates
 * and the INPUT level's grid index (+ its rowof when the input rows are not canonical); with them the o2i table is
 * written output-stationary (complete coalesced rows) instead of a 0xFF fill + scattered stores. */
int toda_rulebook_conv(const int32_t* idx_in, int n_in, int batch,
                       const int32_t* shape_in_host, const int32_t* ksize_host,
                       const int32_t* stride_host, const int32_t* pad_host,
                       const int32_t* shape_out_host, const void* gi_out, int n_out,
                       int32_t* nbr_o2i, int32_t* nbr_i2o, int32_t* pair_cnt,
                       const int32_t* idx_out, const void* gi_in, const int32_t* rowof_in,
                       int cnt_zeroed, void* stream);

/* ------------------------------------------------------------------------
 * Sparse convolution arithmetic (spconv SubMConv3d / SparseConv3d forward and
 * the autograd backward reached from tools/train_utils/train_utils.py:55).
 * All three are gather -> fp32 MFMA GEMM -> store, output stationary, no
 * atomics in fwd/dgrad.
 * ---------------------------------------------------------------------- */
/* Re-order weights [Cout][K][Cin] into the MFMA fragment order the kernels
 * read.  transpose=0: operand for fwd (gathers Cin, produces Cout).
 * transpose=1: operand for dgrad (gathers Cout, produces Cin); flip_k=1
 * additionally reverses the offset order (SubM dgrad reuses the forward
 * table through the point symmetry of the stencil). */
size_t toda_spconv_packed_weight_floats(int k_vol, int c_gather, int c_produce);
/* Matrix path of the gather-GEMMs (forward and data gradient of spconv SubMConv3d / SparseConv3d, reference
 * pcdet/models/backbones_3d/spconv_backbone.py:77-125,191-240), process-wide: 0 = native - fp32 operands on
 * v_mfma_f32_16x16x4_f32; 1 = split - every fp32 operand taken apart EXACTLY into three bf16 values (hi + mid + lo = x),
 * six of the nine cross products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the three dropped ones are below
 * 2^-22 of the product each, 2^-25 on average: the size of an fp32 product's own rounding), 6/16 of the matrix cycles, for the channel pairs toda_spconv_split_supported names; every other
 * pair runs native under either path.  Initial value: environment TODA_MM = native | split (default split: it holds the
 * gates of tests/test_gpu_split.py - oracle parity at the unchanged tolerance, error against fp64 <= 1.5 x native, bit-reproducible).  The packed operand of a
 * supported pair is written in the format of the path current at PACK time (toda_spconv_packed_weight_floats covers both)
 * and must be multiplied under the same path: set the path before packing, re-pack after changing it.  The same switch selects the
 * arithmetic of toda_spconv_wgrad (32 / 64-channel pairs and 128 x 128) and of toda_conv3x3s2_* / toda_deconv_* (operands split when
 * they are staged; no packed operand involved; TODA_PG_SPLIT=0 keeps those on the fp32 instructions). */
int toda_matrix_path(void);
int toda_set_matrix_path(int mode);
int toda_spconv_split_supported(int c_gather, int c_produce);
int toda_spconv_pack_weight(const float* w, int cout, int k_vol, int cin,
                            int transpose, int flip_k, float* wp, void* stream);
/* The same for n weights in one launch (the forward and the data-gradient operand of every sparse convolution of a
 * backbone, once per optimizer step).  All array arguments are HOST arrays of length n; wp_host[i] receives
 * toda_spconv_packed_weight_floats(k_vol[i], c_gather, c_produce) floats. */
int toda_spconv_pack_weights(int n, const float* const* w_host, const int32_t* cout_host, const int32_t* k_vol_host,
                             const int32_t* cin_host, const int32_t* transpose_host, const int32_t* flip_k_host,
                             float* const* wp_host, void* stream);
/* out[o, :] = bias + sum_k Wp[k] . in[nbr[k*n_out + o], :]   (rows with nbr<0 skipped).
 * `in` has n_in rows of c_gather floats (the table is read through a bounds-checked buffer
 * descriptor, so n_in * c_gather * 4 must be < 4 GiB). */
int toda_spconv_gather_gemm(const float* in, int n_in, int c_gather, const float* wp,
                            const int32_t* nbr, int n_out, int k_vol, int c_produce,
                            const float* bias /*nullable*/, float* out, void* stream);
/* Same, visiting the output rows in the order `order[n_out]` (a permutation, nullable = canonical): results are
 * identical, only the assignment of rows to waves changes (the class-sorted data gradient below is built on it). */
int toda_spconv_gather_gemm_ordered(const float* in, int n_in, int c_gather, const float* wp,
                                    const int32_t* nbr, int n_out, int k_vol, int c_produce,
                                    const float* bias, float* out, const int32_t* order, void* stream);

/* Data gradient of a STRIDED SparseConv3d (autograd backward of spconv_backbone.py:118-160's spconv2/3/4 and conv_out):
 * an input site reaches an output only through the kernel offsets congruent to (coordinate + padding) modulo the
 * stride on every axis, i.e. 1..8 of the 27 offsets, all of them populated.  toda_rulebook_class_order regroups the
 * rows (= the conv's INPUT sites, in_coords [n_in][4] = b,z,y,x) by residue class inside 8192-row blocks:
 * order[pos] = row, cls_sorted[pos] = class.  toda_spconv_gather_gemm_classed is toda_spconv_gather_gemm_ordered whose
 * waves walk only the candidate offsets of their class (ksize / stride / padding: HOST int[3], z,y,x).  Results are
 * bit-identical to the plain call. */
int toda_rulebook_class_order(const int32_t* in_coords, int n_in, const int32_t* stride_host, const int32_t* padding_host,
                              int32_t* order, unsigned char* cls_sorted, void* stream);
int toda_spconv_gather_gemm_classed(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr, int n_out,
                                    int k_vol, int c_produce, const float* bias, float* out, const int32_t* order,
                                    const unsigned char* cls_sorted, const int32_t* ksize_host, const int32_t* stride_host,
                                    const int32_t* padding_host, void* stream);
/* The same contraction for the narrow K = 27 layers (<= 32 gathered, <= 32 produced channels: conv_input, the 16-channel SubM
 * level, the strided 16 -> 32 and their data gradients - reference pcdet/models/backbones_3d/spconv_backbone.py:92-115): per
 * offset a wave compacts the rows that have a neighbour, so only real pairs are gathered and multiplied.  w is the PLAIN weight
 * [w_cout][27][w_cin]; transpose / flip_k as in toda_spconv_pack_weight (data gradient: transpose = 1, flip_k = 1 for SubM). */
int toda_spconv_gather_gemm_compact_supported(int c_gather, int c_produce, int k_vol);
int toda_spconv_gather_gemm_compact(const float* in, int n_in, int c_gather, const float* w, int w_cout, int w_cin,
                                    int transpose, int flip_k, const int32_t* nbr, int n_out, int k_vol, int c_produce,
                                    const float* bias /*nullable*/, float* out, void* stream);
/* The same launch with the following BatchNorm1d's moments from the epilogue (the <= 16-channel layers of the backbone,
 * reference spconv_backbone.py:21-25): sums = 2 c results + [2 c][workgroups] scratch as toda_spconv_gather_gemm_stats;
 * blocks_out == NULL: folded by the call, else left for toda_bn_finalize_partials with *blocks_out (host) partials per column. */
size_t toda_spconv_gather_gemm_compact_stats_doubles(int n_out, int c_produce);
int toda_spconv_gather_gemm_compact_stats(const float* in, int n_in, int c_gather, const float* w, int w_cout, int w_cin,
                                          int transpose, int flip_k, const int32_t* nbr, int n_out, int k_vol,
                                          int c_produce, const float* bias /*nullable*/, float* out, double* sums,
                                          size_t sums_doubles, int* blocks_out /*nullable*/, void* stream);

/* dw[co][k][ci] = sum_o in[nbr[k*n_out+o], ci] * dout[o, co] */
size_t toda_spconv_wgrad_workspace_bytes(int n_out, int k_vol, int cin, int cout);
int toda_spconv_wgrad(const float* in, int n_in, const float* dout, const int32_t* nbr,
                      int n_out, int k_vol, int cin, int cout, float* dw,
                      void* ws, size_t ws_bytes, void* stream);
/* SparseConvTensor.dense() as used by HeightCompression
 * (pcdet/models/backbones_2d/map_to_bev/height_compression.py:21-23):
 * dense[b, c, z, y, x] = feat[row, c]; the caller views it as [B, C*D, H, W].
 * fwd zero-fills `dense` itself. */
int toda_sparse_to_dense_fwd(const float* feat, const int32_t* idx, int n, int c,
                             int batch, const int32_t* shape_host, float* dense, void* stream);
int toda_sparse_to_dense_bwd(const float* grad_dense, const int32_t* idx, int n, int c,
                             int batch, const int32_t* shape_host, float* grad_feat, void* stream);

/* PointPillarScatter (pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37):
 * canvas[b, c, y, x] = pillar[row, c] for coords (b, z=0, y, x). */
int toda_pillar_scatter_fwd(const float* feat, const int32_t* idx, int n, int c,
                            int batch, int ny, int nx, float* canvas, void* stream);
int toda_pillar_scatter_bwd(const float* grad_canvas, const int32_t* idx, int n, int c,
                            int batch, int ny, int nx, float* grad_feat, void* stream);

/* ------------------------------------------------------------------------
 * Fused per-channel statistics / normalise+ReLU over the active rows of a
 * sparse level: the nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU pair of
 * post_act_block (spconv_backbone.py:8-27).
 * ---------------------------------------------------------------------- */
/* sums[0:c] = sum_rows x, sums[c:2c] = sum_rows x*x  (fp32 inside a thread, fp64 across threads and blocks, fixed
 * order: bit-reproducible).  `sums` must hold toda_rows_reduce_doubles(n, c) doubles: the first 2c are the result,
 * the rest is per-block scratch (also for toda_rows_bn_bwd). */
size_t toda_rows_reduce_doubles(int n, int c);
int toda_rows_moments(const float* x, int n, int c, double* sums /*[toda_rows_reduce_doubles(n, c)]*/,
                      void* stream);
/* y = relu?(x * scale[c] + shift[c] (+ residual)) */
int toda_rows_affine_act(const float* x, const float* scale, const float* shift,
                         const float* residual /*nullable*/, int n, int c, int relu,
                         float* y, void* stream);

/* Batch statistics -> per-channel mean / invstd / scale / shift, plus nn.BatchNorm1d's
 * running-statistic update (momentum, unbiased running_var).  training == 0: running stats. */
int toda_bn_finalize(const double* sums /*[2c] from toda_rows_moments*/, int n, int c,
                     const float* gamma, const float* beta, float* running_mean /*nullable*/,
                     float* running_var /*nullable*/, float momentum, float eps, int training,
                     float* mean, float* invstd, float* scale, float* shift, void* stream);
/* The same with the fold of the per-workgroup partial sums inside (training mode): toda_spconv_gather_gemm_stats_partials leaves
 * `blocks` partials per column behind the 2c result slots of `sums`; one launch folds them in the fixed order of the separate
 * fold, writes the totals to sums[0:2c] and finalises.  Saves one launch per BatchNorm1d of the sparse backbone. */
int toda_bn_finalize_partials(double* sums, int blocks, int n, int c, const float* gamma, const float* beta,
                              float* running_mean /*nullable*/, float* running_var /*nullable*/, float momentum, float eps,
                              float* mean, float* invstd, float* scale, float* shift, void* stream);
/* Backward of the same pair.  stats = [mean | invstd | scale | shift] (4*c floats, as written by
 * toda_bn_finalize into one buffer).  dz = dy * (x*scale+shift > 0) (relu != 0) or dy;
 * sums[0:c] = sum dz = d(beta), sums[c:2c] = sum dz*xhat = d(gamma) (zeroed and filled by the call); the same 2c values rounded
 * to float32 follow at ((float*)(sums + 2c))[0:2c] (what an optimizer wants: no conversion pass);
 * dx = gamma * invstd * (dz - sums[0:c]/n - xhat * sums[c:2c]/n), xhat = (x - mean) * invstd. */
int toda_rows_bn_bwd(const float* dy, const float* x, const float* stats, const float* gamma,
                     int n, int c, int relu, double* sums /*[toda_rows_reduce_doubles(n, c)]*/, float* dx,
                     void* stream);

/* The same with a shortcut branch (SparseBasicBlock, spconv_backbone.py:30-66: y = relu(bn2(x) + identity)): the ReLU
 * mask is recomputed from x*scale+shift + residual, dres (nullable) receives dz = the gradient of the shortcut. */
int toda_rows_bn_bwd_res(const float* dy, const float* x, const float* residual /*nullable*/, const float* stats,
                         const float* gamma, int n, int c, int relu, double* sums, float* dx,
                         float* dres /*nullable*/, void* stream);
/* ... and with the column sums of dx, dx_colsum[0:c] = sum over rows of dx: the bias gradient of the convolution that produced
 * x (SparseBasicBlock's convolutions have a bias, spconv_backbone.py:37-40; autograd's `grad_output.sum(0)` pass over dx
 * disappears).  Taken while dx is written; fixed-order fold (deterministic).  colsum_ws: toda_rows_bn_bwd_colsum_doubles(n, c)
 * doubles; both pointers NULL: exactly toda_rows_bn_bwd_res. */
size_t toda_rows_bn_bwd_colsum_doubles(int n, int c);
int toda_rows_bn_bwd_res_colsum(const float* dy, const float* x, const float* residual /*nullable*/, const float* stats,
                                const float* gamma, int n, int c, int relu, double* sums, float* dx,
                                float* dres /*nullable*/, double* colsum_ws /*nullable*/, float* dx_colsum /*nullable*/,
                                void* stream);

/* Single-pass training-mode nn.BatchNorm2d (+ nn.ReLU) of the BEV neck and heads (base_bev_backbone.py:37-58,
 * center_head.py:20-28, 73-80; replaces torch's batch_norm + relu_ pair and their backward): the values of a channel sit in
 * registers between the statistics and the normalisation, so forward reads x once and writes y once and backward reads x
 * and dy once and writes dx once.  Two forms: one workgroup per channel (batch 1 / 2: hw <= 36864, batch 4: hw <= 16384;
 * backward then reads x a second time), or - batch 2 / 4 with a `sync` workspace, hw <= 36864 - one workgroup per (channel,
 * sample) plane, the workgroups of a channel exchanging two fp64 partial results through `sync`; the library takes the
 * per-plane form where the per-channel one does not fit or measured slower (large backward planes).  hw % 4 == 0 uses 16-byte
 * accesses, any other hw dword accesses.
 * sync: toda_bn2d_sync_bytes() bytes, zeroed ONCE by the caller, used by one stream at a time, channels <= 4096; epoch:
 * a non-zero number the caller does not repeat on that workspace (increment per call) - nothing is reset between launches.
 * save = [2][c] floats (mean, 1/sqrt(var + eps)) from forward for backward.  running_mean / running_var (nullable
 * together) are updated in place like nn.BatchNorm2d does (momentum, unbiased variance).  relu != 0: y = max(bn(x), 0);
 * backward recomputes the mask from x with the forward's expression.  Deterministic (fixed summation order). */
int toda_bn2d_supported(int batch, int c, int hw);
size_t toda_bn2d_sync_bytes(void);
int toda_bn2d_fwd(const float* x, int batch, int c, int hw, const float* gamma, const float* beta,
                  float* running_mean /*nullable*/, float* running_var /*nullable*/, float momentum, float eps, int relu,
                  float* y, float* save, void* sync /*nullable*/, unsigned epoch, void* stream);
int toda_bn2d_bwd(const float* x, const float* dy, int batch, int c, int hw, const float* gamma, const float* beta,
                  const float* save, int relu, float* dx, float* dgamma, float* dbeta, void* sync /*nullable*/,
                  unsigned epoch, void* stream);
/* The same pair with y / dy a CHANNEL SLICE [channel0, channel0 + c) of a wider [batch][channels][hw] tensor: the up-sampling
 * deblocks of the BEV neck write straight into the concatenated map and read its gradient in place, so torch.cat and the copies
 * of its backward (reference pcdet/models/backbones_2d/base_bev_backbone.py:104-107) never run. */
int toda_bn2d_fwd_into(const float* x, int batch, int c, int hw, const float* gamma, const float* beta,
                       float* running_mean /*nullable*/, float* running_var /*nullable*/, float momentum, float eps, int relu,
                       float* y, int y_channels, int y_channel0, float* save, void* sync /*nullable*/, unsigned epoch,
                       void* stream);
int toda_bn2d_bwd_from(const float* x, const float* dy, int dy_channels, int dy_channel0, int batch, int c, int hw,
                       const float* gamma, const float* beta, const float* save, int relu, float* dx, float* dgamma,
                       float* dbeta, void* sync /*nullable*/, unsigned epoch, void* stream);

/* ------------------------------------------------------------------------
 * CenterHead target assignment (pcdet/models/dense_heads/center_head.py:103-219,
 * pcdet/models/model_utils/centernet_utils.py:9-69): gaussian heat-maps,
 * regression targets, flat indices and masks for one head group.
 * gt_boxes [B, G, 8] = (x y z dx dy dz heading cls_in_head(1-based, 0 = pad)).
 * ---------------------------------------------------------------------- */
int toda_center_assign(const float* gt_boxes, int batch, int n_gt, int code_size,
                       int num_classes, int fm_w, int fm_h,
                       const float* range_host /*[6]*/, const float* vsize_host /*[3]*/,
                       int fm_stride, int max_objs, double gaussian_overlap, int min_radius,
                       float* heatmap /*[B, num_classes, fm_h, fm_w], zero-filled by the call*/,
                       float* ret_boxes /*[B, max_objs, code_size]*/,
                       int64_t* inds /*[B, max_objs]*/, int64_t* mask /*[B, max_objs]*/,
                       void* stream);

/* ------------------------------------------------------------------------
 * Rotated BEV IoU and greedy NMS (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:236-326 and the host
 * sweep of iou3d_nms.cpp:100-135, reached through model_nms_utils.class_agnostic_nms from
 * CenterHead.generate_predicted_boxes, center_head.py:291-300).  Boxes are rows of 7 floats
 * (x, y, z, dx, dy, dz, heading).  toda_nms_rotated expects the boxes sorted by descending score
 * and leaves the kept indices (ascending = score order) and their count on the device.
 * ---------------------------------------------------------------------- */
int toda_boxes_iou_bev(const float* boxes_a, int na, const float* boxes_b, int nb,
                       float* iou /*[na, nb]*/, void* stream);
/* Intersection AREA of the rotated BEV rectangles (iou3d_nms_kernel.cu boxes_overlap_kernel, used by
 * iou3d_nms_utils.boxes_iou3d_gpu :52-82 for the recall record of detector3d_template.py:287-328). */
int toda_boxes_overlap_bev(const float* boxes_a, int na, const float* boxes_b, int nb,
                           float* overlap /*[na, nb]*/, void* stream);
size_t toda_nms_workspace_bytes(int n);
int toda_nms_rotated(const float* boxes_sorted, int n, float thresh, int64_t* keep /*[n]*/,
                     int32_t* n_keep_dev, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Point-table primitives of the TODA mixing processors and the data processor's range mask.
 * They replace, on the device, the numpy / single-thread C++ work the reference does per scene in
 * DataLoader workers:
 *   - roiaware_pool3d_utils.points_in_boxes_cpu (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168),
 *     box_utils.remove_points_in_boxes3d (pcdet/utils/box_utils.py:75-89), augmentor_utils.get_points_in_box
 *     (pcdet/datasets/augmentor/augmentor_utils.py:474-491)                       -> toda_points_in_boxes
 *   - the azimuth-sector test of PolarMix swap (inter_domain_point_polarmix.py:76-98) -> toda_points_sector
 *   - the crop test of CutMix (inter_domain_point_cutmix.py:44-54) and mask_points_by_range
 *     (pcdet/utils/common_utils.py:60-63)                                         -> toda_points_rect
 *   - the cylinder cells of LaserMix (inter_domain_point_lasermix.py:89-165)       -> toda_points_polar_cell
 *   - the elevation bands of spherical LaserMix (inter_domain_point_lasermix.py:40-47,62-80) -> toda_points_pitch_band
 *   - PolarMix's sector cut at a random range (swap_with_range, inter_domain_point_polarmix.py:101-123) and the
 *     elevation test of swap(use_pitch=True) (:81-93)                  -> toda_points_polar_select, toda_points_pitch_range
 *   - boolean-mask indexing / np.delete / np.concatenate of point arrays           -> toda_rows_select_append
 *   - rotate_copy's point rotation (inter_domain_point_polarmix.py:160-188)        -> toda_points_rotate_z
 * points: [n, c] fp32 rows (x, y, z, ...).  n_dev (nullable): device int32, rows = min(n, *n_dev), so
 * chains run without host syncs.  flags / cells are int32 per row.
 * ---------------------------------------------------------------------- */
/* flags[j] = 1 iff some box (rows of box_stride >= 7 floats: x y z dx dy dz heading) contains point j.
 * mode 0: roiaware test (margin 1e-2, strict), mode 1: get_points_in_box test (margin 1e-1, inclusive).
 * mode 2: points_in_boxes_gpu (roiaware_pool3d_kernel.cu:23-36, 313-336; margin 1e-5): flags[j] = index of the FIRST
 * box that holds point j, -1 for none (used to cut the ground-truth database out of the scenes). */
int toda_points_in_boxes(const float* points, int n, const int32_t* n_dev, int c, const float* boxes,
                         int k, int box_stride, int mode, int32_t* flags, void* stream);
/* flags[j] = lo < -atan2(y, x) < hi   (yaw as an fp32 value, compared in fp64) */
int toda_points_sector(const float* points, int n, const int32_t* n_dev, int c, double lo, double hi,
                       int32_t* flags, void* stream);
/* closed == 0: lo < (x, y) < hi;  closed == 1: lo <= (x, y) <= hi */
int toda_points_rect(const float* points, int n, const int32_t* n_dev, int c, const double* lo_xy_host,
                     const double* hi_xy_host, int closed, int32_t* flags, void* stream);
/* cell[j] = i * n_dis + j' for yaw' in (yaw_edges[i], yaw_edges[i+1]] and range in (dis_edges[j'], dis_edges[j'+1]],
 * yaw' = wrap(-atan2(y, x) + phase), range = clip(sqrt(x^2 + y^2), dis_lo, dis_hi); -1 when outside every cell */
int toda_points_polar_cell(const float* points, int n, const int32_t* n_dev, int c, float phase,
                           const double* yaw_edges_host, int n_yaw, const double* dis_edges_host, int n_dis,
                           float dis_lo, float dis_hi, int32_t* cell, void* stream);
/* flags[j] = [yaw_mode 1: lo < yaw < hi | 2: yaw < lo or yaw > hi]  and  [dis_mode 0: - | 1: range < dis_th | 2: range > dis_th]
 * and, with pitch_range_dev != NULL, [range > 1 and -atan2(z, range) outside [pitch_range_dev[0], pitch_range_dev[1]]];
 * yaw = -atan2(y, x) and range = sqrt(x^2 + y^2) as fp32 values, thresholds compared in fp64 (round them to fp32 where numpy would). */
int toda_points_polar_select(const float* points, int n, const int32_t* n_dev, int c, double lo, double hi, int yaw_mode,
                             int dis_mode, double dis_th, const float* pitch_range_dev, int32_t* flags, void* stream);
/* range_dev[0:2] = min, max of -atan2(z, range) over the rows with range > 1 (+inf, -inf when there is none); deterministic. */
size_t toda_points_pitch_range_workspace_bytes(void);
int toda_points_pitch_range(const float* points, int n, const int32_t* n_dev, int c, float* range_dev, void* ws,
                            size_t ws_bytes, void* stream);
/* band[j] = i for edges[i + 1] < e <= edges[i] (n_bands + 1 descending edges, radians, host array), e = clip(atan2(z_offset + z,
 * range), clip_lo, clip_hi) in fp32; -1 when no band holds it */
int toda_points_pitch_band(const float* points, int n, const int32_t* n_dev, int c, float z_offset, float clip_lo,
                           float clip_hi, const double* edges_host, int n_bands, int32_t* band, void* stream);
/* Stable compaction: rows with (keys[j] == match) != invert (all rows when keys == NULL) are appended, in
 * order, at dst[*cursor_dev ...]; *cursor_dev += count.  Rows beyond cap_rows are dropped (the cursor still
 * counts them, so the caller sees the overflow). */
size_t toda_rows_select_workspace_bytes(int n);
int toda_rows_select_append(const float* src, int n, const int32_t* n_dev, int c, const int32_t* keys,
                            int match, int invert, float* dst, int cap_rows, int32_t* cursor_dev,
                            void* ws, size_t ws_bytes, void* stream);
/* dst[:, 0:2] = fp32(rotation of (x, y) in fp64 by (cosv, sinv)), z and column 3 copied, columns >= 4 zeroed */
int toda_points_rotate_z(const float* src, int n, const int32_t* n_dev, int c, double cosv, double sinv,
                         float* dst, void* stream);
/* Global augmentations in one pass, in the reference's order and fp32 arithmetic (pcdet/datasets/augmentor/
 * augmentor_utils.py:8-81 random_flip_along_x / _y, global_rotation, global_scaling): flip_x: y -> -y; flip_y: x -> -x;
 * rotate: (x, y) -> (x c - y s, x s + y c); rescale: xyz *= scale.  dst may alias src. */
int toda_points_world_transform(const float* src, int n, const int32_t* n_dev, int c, int flip_x, int flip_y,
                                int rotate, float cosv, float sinv, int rescale, float scale, float* dst,
                                void* stream);

/* Forward convolution that also returns the BatchNorm statistics of its output (reference
 * pcdet/models/backbones_3d/spconv_backbone.py:21-25,54-64: every sparse conv is followed by BatchNorm1d): the
 * per-channel sum and sum of squares are taken from the accumulators in the kernel's epilogue, so the separate
 * statistics pass over the output (toda_rows_moments) disappears.  sums: toda_spconv_gather_gemm_stats_doubles()
 * doubles; on return sums[0:c] = sum, sums[c:2c] = sum of squares (the layout toda_bn_finalize reads; fixed-order
 * fold, deterministic).  Only for channel pairs toda_spconv_gather_gemm_stats_supported() accepts. */
int toda_spconv_gather_gemm_stats_supported(int c_gather, int c_produce);
size_t toda_spconv_gather_gemm_stats_doubles(int n_out, int c_produce);
int toda_spconv_gather_gemm_stats(const float* in, int n_in, int c_gather, const float* packed_w, const int32_t* nbr,
                                  int n_out, int k_vol, int c_produce, const float* bias, float* out, double* sums,
                                  size_t sums_doubles, void* stream);
/* toda_spconv_gather_gemm_stats without the fold of the partial sums: *blocks_out (host memory, written before the call returns)
 * = number of partials per column, to be handed to toda_bn_finalize_partials. */
int toda_spconv_gather_gemm_stats_partials(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr,
                                           int n_out, int k_vol, int c_produce, const float* bias /*nullable*/, float* out,
                                           double* sums, size_t sums_doubles, int* blocks_out, void* stream);

/* ------------------------------------------------------------------------
 * Dense 3x3 / stride 1 / pad 1 fp32 convolution of the BEV neck and the dense heads (NCHW), replacing
 * torch.nn.Conv2d -> cuDNN / MIOpen at pcdet/models/backbones_2d/base_bev_backbone.py:37-58,81-112 and
 * pcdet/models/dense_heads/center_head.py:20-28,73-80 together with autograd's backward of those layers
 * (tools/train_utils/train_utils.py:55).  Fused Winograd F(4x4,3x3) on the fp32 matrix cores; nothing of the
 * Winograd domain is written to memory.  w is torch's [Cout][Cin][3][3].
 *   toda_conv3x3_supported        1 when (batch, cin, cout, H, W) can run here in all three directions
 *                                 (channels multiples of 32, W even, tensors below 4 GiB), else 0
 *   toda_conv3x3_transform_weight u = G w G^T in the kernel's operand order; mode 0: forward,
 *                                 mode 1: data gradient (filters rotated by 180 degrees, channel roles swapped),
 *                                 mode 2: both, forward operand first (2 x toda_conv3x3_weight_floats floats)
 *   toda_conv3x3_fwd              y[B][cout][H][W] = conv(x[B][cin][H][W]) (+ bias[cout], nullable).  With the
 *                                 mode-1 operand and (cin, cout) = (Cout, Cin) of the layer it computes dX from dY.
 *                                 ws: toda_conv3x3_workspace_bytes() bytes, ZERO before the first call and used by
 *                                 one call at a time (hand-off flags, which every call leaves zero again, +
 *                                 partial-sum slabs of the stream-K work split)
 *   toda_conv3x3_wgrad            dw[Cout][Cin][3][3] from x and dy (workspace: per-split partial sums, folded
 *                                 in fixed order - deterministic, no float atomics)
 * ---------------------------------------------------------------------- */
int toda_conv3x3_supported(int batch, int cin, int cout, int H, int W);
size_t toda_conv3x3_weight_floats(int cout, int cin);
int toda_conv3x3_transform_weight(const float* w, int cout, int cin, int mode, float* u, void* stream);
size_t toda_conv3x3_workspace_bytes(void);
size_t toda_conv3x3_wgrad_workspace_bytes(int batch, int cin, int cout, int H, int W);
int toda_conv3x3_wgrad(const float* x, const float* dy, int batch, int cin, int cout, int H, int W, float* dw,
                       void* ws, size_t ws_bytes, void* stream);
int toda_conv3x3_fwd(const float* x, const float* u, const float* bias, int batch, int cin, int cout, int H,
                     int W, float* y, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * The other convolutions of BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:32-36: ZeroPad2d(1) +
 * Conv2d(c_in, c, 3, stride 2, padding 0) at the head of a block; :47-66: the deblocks ConvTranspose2d(c, c_up, k = s,
 * stride = s) with s = 1 and s = 2), forward and autograd backward on the fp32 matrix cores, NCHW in and out (torch runs them
 * on MIOpen / rocBLAS behind layout transposes and im2col / col2im passes).
 *   toda_conv3x3s2_*   x [B][cin][H][W] (H, W even) -> y [B][cout][H/2][W/2]; w [cout][cin][3][3] (nn.Conv2d layout).
 *                      _dgrad covers the four input-pixel parities in one launch (an input pixel is reached by 1, 2, 2 or 4 of
 *                      the 9 taps); _wgrad contracts over pixels in splits, folded in fixed order (deterministic).
 *   toda_deconv_*      x [B][cin][H][W] -> y [B][cout][s H][s W]; w [cin][cout][s][s] (nn.ConvTranspose2d layout), s in {1, 2};
 *                      the pixel shuffle of s = 2 is the store of the forward GEMM / the gather of the backward ones.
 * No bias (the reference builds these layers with bias=False; a caller with a bias adds it afterwards).
 * ---------------------------------------------------------------------- */
int toda_conv3x3s2_supported(int batch, int cin, int cout, int H, int W);
int toda_conv3x3s2_fwd(const float* x, const float* w, int batch, int cin, int cout, int H, int W, float* y, void* stream);
int toda_conv3x3s2_dgrad(const float* dy, const float* w, int batch, int cin, int cout, int H, int W, float* dx, void* stream);
size_t toda_conv3x3s2_wgrad_workspace_bytes(int batch, int cin, int cout, int H, int W);
int toda_conv3x3s2_wgrad(const float* x, const float* dy, int batch, int cin, int cout, int H, int W, float* dw, void* ws,
                         size_t ws_bytes, void* stream);
int toda_deconv_supported(int batch, int cin, int cout, int H, int W, int s);
int toda_deconv_fwd(const float* x, const float* w, int batch, int cin, int cout, int H, int W, int s, float* y, void* stream);
int toda_deconv_dgrad(const float* dy, const float* w, int batch, int cin, int cout, int H, int W, int s, float* dx, void* stream);
size_t toda_deconv_wgrad_workspace_bytes(int batch, int cin, int cout, int H, int W, int s);
int toda_deconv_wgrad(const float* x, const float* dy, int batch, int cin, int cout, int H, int W, int s, float* dw, void* ws,
                      size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * CenterHead.get_loss of one head group, value and gradient (pcdet/models/dense_heads/center_head.py:229-262:
 * sigmoid + clamp(1e-4, 1 - 1e-4) of the heat-map logits; pcdet/utils/loss_utils.py:264-297 neg_loss_cornernet /
 * FocalLossCenterNet; :300-385 _gather_feat, _transpose_and_gather_feat, _reg_loss / RegLossCenterNet; the
 * code_weights / loc_weight / cls_weight products of center_head.py:250-257).
 *   hm_logits, heatmap [B][classes][H][W]; reg_maps_host[] = n_branch device pointers [B][c_j][H][W] in HEAD_ORDER
 *   (sum c_j = code_size <= 16); inds, mask [B][max_objs] int64; target_boxes [B][max_objs][code_size];
 *   code_weights: HOST array.
 *   _fwd  hm_prob = clamped sigmoid; out4 = {hm_loss * cls_weight, loc_loss * loc_weight, 1 / max(num_pos, 1),
 *         1 / max(num_objs, 1)} (device); ws keeps what _bwd needs (the same ws must be passed on)
 *   _bwd  up_hm / up_loc: device scalars dL/d(out4[0]), dL/d(out4[1]); hm_grad [B][classes][H][W];
 *         reg_grads_host[]: gradient maps of the branches (zero-filled here, then scattered into)
 * Sums are folded in a fixed order (fp64 partials): no atomics, bitwise reproducible.
 * ---------------------------------------------------------------------- */
size_t toda_center_loss_workspace_bytes(int batch, int classes, int H, int W, int max_objs, int code_size);
int toda_center_loss_fwd(const float* hm_logits, const float* heatmap, int batch, int classes, int H, int W, int n_branch,
                         const float* const* reg_maps_host, const int32_t* reg_channels_host, const int64_t* inds,
                         const int64_t* mask, const float* target_boxes, int max_objs, int code_size,
                         const float* code_weights_host, float cls_weight, float loc_weight, float* hm_prob, float* out4,
                         void* ws, size_t ws_bytes, void* stream);
int toda_center_loss_bwd(const float* out4, const float* up_hm, const float* up_loc, int batch, int classes, int H, int W,
                         int n_branch, float* const* reg_grads_host, const int32_t* reg_channels_host, const int64_t* inds,
                         const int64_t* mask, int max_objs, int code_size, const float* code_weights_host, float cls_weight, float loc_weight,
                         float* hm_grad, const void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolutions with 1..4 OUTPUT channels: the last layer of every CenterHead branch
 * (pcdet/models/dense_heads/center_head.py:20-28: Conv2d(64, out_channels, 3, padding=1, bias=True) of center /
 * center_z / dim / rot / hm), forward and autograd backward.  All n <= 8 branches of a head go through ONE launch
 * per direction.  Pointer arguments ending in [] are HOST arrays of n device pointers, cout_host holds the n
 * output-channel counts; every branch has the same (batch, cin, H, W), W % 4 == 0, W <= 256.  x_image_stride /
 * dx_image_stride: floats between two images of a branch input (its gradient): 0 = dense (cin * H * W); larger when
 * the branch inputs are channel slices of one [B][n * cin][H][W] tensor (the fused hidden layer of the head).
 *   _fwd     y_i[B][cout_i][H][W] = conv(x_i[B][cin][H][W], w_i[cout_i][cin][3][3]) + bias_i (bias_host may be NULL)
 *   _dgrad   dx_i[B][cin][H][W] from dy_i and w_i
 *   _wgrad   out = for i in order: dw_i [cout_i][cin][3][3] then db_i [cout_i]  (band slabs in ws, fixed-order fold)
 * ---------------------------------------------------------------------- */
int toda_conv3x3_narrow_supported(int batch, int cin, int cout, int H, int W);
int toda_conv3x3_narrow_fwd(int n, const float* const* x_host, const float* const* w_host, const float* const* bias_host,
                            const int32_t* cout_host, int batch, int cin, int H, int W, long long x_image_stride,
                            float* const* y_host, void* stream);
int toda_conv3x3_narrow_dgrad(int n, const float* const* dy_host, const float* const* w_host, const int32_t* cout_host,
                              int batch, int cin, int H, int W, long long dx_image_stride, float* const* dx_host, void* stream);
size_t toda_conv3x3_narrow_wgrad_workspace_bytes(int n, int batch, int cin, int H);
int toda_conv3x3_narrow_wgrad(int n, const float* const* x_host, const float* const* dy_host, const int32_t* cout_host,
                              int batch, int cin, int H, int W, long long x_image_stride, float* out, void* ws, size_t ws_bytes,
                              void* stream);

/* ------------------------------------------------------------------------
 * Instrumentation (no counterpart in the reference): per-launch durations of the gather-GEMM kernels, taken
 * from start / stop events stamped on the kernel dispatch itself (hipExtLaunchKernelGGL), i.e. the number
 * rocprofv3 --kernel-trace reports.  toda_timing_begin arms up to `capacity` launches of
 * toda_spconv_gather_gemm (process-wide, one stream at a time); toda_timing_end waits for them and returns the
 * milliseconds in launch order (*n_out = launches seen).
 * ---------------------------------------------------------------------- */
int toda_timing_begin(int capacity);
int toda_timing_end(float* ms_out_host, int cap, int* n_out_host);

/* ---------------------------------------------------------------------------------------------
 * The optimizer step of the reference trainers in two launches (reference tools/train_utils/train_utils.py:55-59:
 * clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP) + optimizer.step(), the optimizer being
 * tools/train_utils/optimization/fastai_optim.py:104-236 OptimWrapper(Adam, true_wd) = p *= 1 - lr * wd, then torch's Adam):
 *   norm = || all gradients ||_2 (returned in norm_out[0]); g *= min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no clipping);
 *   p *= 1 - lr * weight_decay; m += (g - m)(1 - beta1); v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps).
 * The n tensors (fp32, contiguous) are given as device arrays of device addresses + element counts; chunk_tensor / chunk_off list
 * the toda_clip_adam_chunk()-element pieces of all tensors (one workgroup each); partial: n_chunks doubles of scratch.  The partial
 * sums are folded in index order: deterministic. */
int toda_clip_adam_chunk(void);
int toda_clip_adam_step(const unsigned long long* param, const unsigned long long* grad, const unsigned long long* exp_avg,
                        const unsigned long long* exp_avg_sq, const long long* numel, const int* chunk_tensor,
                        const long long* chunk_off, int n_chunks, double* partial, float* norm_out, float max_norm,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TODA_H_ */
