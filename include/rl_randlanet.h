/*
 * rl_randlanet.h - C ABI of librandla_hip.so: the MI355X (gfx950) kernels of the RandLA-Net
 * segmentation hot path, a drop-in for the compiled / ATen pieces of
 * matthiasverstraete/3d_recognizer's `randlanet` package.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every pointer is a DEVICE pointer owned by the
 *     caller unless said otherwise; nothing is allocated, freed or synchronised inside.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and is
 *     capturable into a hipGraph (no host synchronisation, no allocation).
 *   - return 0 on success, a negative RL_ERR_* otherwise; rl_last_error() gives the text.
 *     Nothing throws across the boundary.
 *   - activations are POINT-MAJOR / CHANNEL-LAST fp32: a (B, n, C) tensor is B*n rows of C
 *     floats.  A "batch-strided" operand addresses row (b, i) at base + (b*bstride + i)*ld,
 *     which lets a kernel read the first n rows of every cloud of a larger (B, n_parent, C)
 *     tensor - the reference's "random sampling" prefix slice (modules.py:587-589).
 *   - a "lazy" operand is a raw pre-BatchNorm tensor plus per-channel scale/shift and an
 *     activation: value = act(raw*scale[c] + shift[c]).  Consumers apply it while loading, so
 *     BatchNorm+activation never costs a pass over memory (SharedMLP.forward, modules.py:93-104).
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   rl_knn_f32            knn_tpk.knn(support, querry, k)   randlanet/utils/src/bindings.cpp:5-7,
 *                         knn.cpp:43-61; also knn_naive / knn_approximate, utils/knn.py:7-117
 *   rl_gemm / rl_wgrad    SharedMLP conv / conv-transpose 1x1 (modules.py:82-84,99), fc_start
 *                         Linear (modules.py:494), AttentivePooling score Linear (modules.py:235),
 *                         with RelativePositionEncoding (modules.py:173-186) as an A-operand source
 *   rl_bn_*               BatchNorm2d(eps 1e-6, momentum 0.99) (modules.py:85-89, 496-499)
 *   rl_copy_rows / rl_scatter_add_rows
 *                         PointFeatureAugmentation gather+concat (modules.py:213-221), permutation /
 *                         prefix slicing (modules.py:571-573, 608), nearest-neighbour interpolation
 *                         gather + skip concat (modules.py:359-364, 600-602) and their backward
 *   rl_csr_build / rl_segment_sum_rows
 *                         the backward of those gathers (torch's scatter_add_) in a fixed summation order
 *   rl_pool_fwd/_bwd      PointFeatureAugmentation + AttentivePooling fused (modules.py:213-253)
 *   rl_attpool_*          AttentivePooling softmax over K + weighted sum (modules.py:246-253)
 *   rl_add_act_*          LocalFeatureAggregation residual + LeakyReLU (modules.py:325)
 *   rl_resid_bn_bwd_*     its backward fused with the two BatchNorm backwards behind it
 *   rl_rpe_build          RelativePositionEncoding (modules.py:173-186), materialised
 *   rl_scale_mask         Dropout of fc_end (modules.py:528)
 *   rl_batch_assemble     PointCloudPreprocessor.preprocess + augmentation + collation (dataset.py:61-131)
 *   rl_upsample_cf        UpSampler nni / nna / idw / isdw (modules.py:343-456)
 *   rl_logits_*           un-permute + (B,C,N) layout of the logits (modules.py:608-611)
 *   rl_loss_*             FocalTverskyLoss / FocalLoss / cross entropy (utils/losses.py:17-87,
 *                         trainer.py:244-269) and accuracy / iou (utils/metrics.py:8-59)
 *   rl_adam_step          torch.optim.Adam step of Trainer.train (utils/trainer.py:78,119)
 */
#ifndef RL_RANDLANET_H
#define RL_RANDLANET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RL_VERSION 110     /* round 6: rl_gemm_stat_slots (64-row tiles of the wide GEMM), rl_bn_finalize pivot vector; round 5: shifted BatchNorm statistics (stats_pivot_* fields, rl_bn_finalize pivoted), rl_head_*; round 4: rl_launch_count */

#define RL_OK 0
#define RL_ERR_ARGS (-1)         /* bad shape / null pointer / unsupported size            */
#define RL_ERR_FEW_SUPPORT (-2)  /* knn: support has fewer than k points (knn.cpp:15-17)   */
#define RL_ERR_LAUNCH (-3)       /* hip launch error                                        */
#define RL_ERR_UNSUPPORTED (-4)  /* valid in the reference, not implemented by this build   */

#define RL_MAX_SLOTS 1024        /* partial-statistics slots written by row-streaming kernels */
#define RL_KNN_MAX_K 64

const char* rl_last_error(void);
/* name of the main kernel function the last entry point dispatched to on this thread (profiling aid:
 * matches the kernel names of a rocprofv3 trace) */
const char* rl_last_kernel(void);
int rl_version(void);
/* kernel launches issued by this library in the calling process so far (every entry point counts the kernels it launches;
 * a measurement aid: the difference around an eager step = the kernels a rocprofv3 trace shows for it) */
int64_t rl_launch_count(void);
/* measurement aid: one idle wavefront that occupies `stream` for `us` microseconds (<= 20000) - stands in for a collective of
 * known length when a schedule around it is timed on a single GPU (tools/allreduce_standin.py); touches no memory */
int rl_spin_us(int us, void* stream);

/* Number of partial-statistics slots a row-streaming kernel writes for `rows` rows:
 * rl_gemm uses rows_per_tile = 128, rl_loss 256; rl_bn_bwd_reduce has its own rl_bn_bwd_slots. */
int rl_row_blocks(int64_t rows, int rows_per_tile);

/* ------------------------------------------------------------------------------------------
 * Exact K nearest neighbours.  d2 = ((dx*dx)+(dy*dy))+(dz*dz) in IEEE fp32 without FMA
 * (nanoflann.hpp:488-497), rows ascending by (d2, index).  support (B,Ns,3), query (B,Nq,3)
 * contiguous fp32; idx_out (B,Nq,k) int64, d2_out (B,Nq,k) fp32.  k <= RL_KNN_MAX_K.
 * Replaces knn_tpk.knn (bindings.cpp:5-7).
 * workspace: device scratch of rl_knn_workspace_bytes(B,Ns,Nq,k) bytes, 256-byte aligned; with it
 * supports of >= 512 points are searched through a uniform grid (same answer, ~100x fewer
 * distance evaluations); NULL selects the tiled brute-force scan.                           */
int64_t rl_knn_workspace_bytes(int B, int Ns, int Nq, int k);
int rl_knn_f32(const float* support, const float* query, int B, int Ns, int Nq, int k,
               int64_t* idx_out, float* d2_out, void* workspace, int64_t workspace_bytes,
               void* stream);

/* Host twin: the same search on the calling host, HOST pointers, no GPU involved, synchronous (queries are split over
 * the hardware threads).  Same contract and error codes; backs the package's CPU device (reference model.py:38-40
 * falls back to the CPU when there is no GPU: config P, predict.py on a GPU-less box).                         */
int rl_knn_f32_cpu(const float* support, const float* query, int B, int Ns, int Nq, int k,
                   int64_t* idx_out, float* d2_out);

/* Same search with int32 indices and batch strides (in points), used inside the network:
 * cloud b of the support starts at support + b*support_bstride*3.                          */
int rl_knn_i32(const float* support, int64_t support_bstride, const float* query,
               int64_t query_bstride, int B, int Ns, int Nq, int k, int32_t* idx_out,
               float* d2_out, void* workspace, int64_t workspace_bytes, void* stream);

/* Several searches over the same B clouds in one launch set (a forward pass needs eight that depend
 * only on the coordinates: encoder K-NN on every level, decoder 1-NN between levels).  Each task is
 * one rl_knn_i32 call; the small ones run beside the large one instead of after it.  At most 8 tasks. */
typedef struct rl_knn_task {
    const float* support;
    int64_t support_bstride;
    const float* query;
    int64_t query_bstride;
    int32_t Ns, Nq, k;
    int32_t* idx_out;
    float* d2_out;
} rl_knn_task;

int64_t rl_knn_multi_workspace_bytes(const rl_knn_task* tasks, int ntasks, int B);
int rl_knn_multi(const rl_knn_task* tasks, int ntasks, int B, void* workspace, int64_t workspace_bytes,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-point linear layer (1x1 conv / conv-transpose / Linear):   Y = A' . W (+ bias)
 *   A' [M x K] : A-operand, M = B*n rows.
 *       a_mode 0: row (b,i) = A + (b*a_bstride + i)*lda, lazily transformed by in_* if
 *                 in_scale != NULL.
 *       a_mode 1: relative position encoding rows built on the fly (modules.py:173-186):
 *                 M = B*n*nbr_k, row (b,i,j) = [xyz_i, xyz_nb, xyz_i - xyz_nb, sqrt(d2)],
 *                 K must be 10; xyz cloud b starts at xyz + b*xyz_bstride*3; nbr_idx/nbr_d2 are
 *                 (B,n,nbr_k) contiguous.
 *   W element (k, c) = W[k*w_ks + c*w_ns]   (Conv2d/Linear weight (N,K): w_ks=1, w_ns=K;
 *                                           ConvTranspose2d weight (K,N): w_ks=N, w_ns=1)
 *   Y row (b,i) = Y + (b*y_bstride + i)*ldy ; accumulate != 0 adds into Y.
 *   stats != NULL: per-block column sums of Y and Y*Y are written as doubles to
 *       stats[slot][0][c], stats[slot][1][c] for slot < rl_gemm_stat_slots(M,N,K) (BatchNorm batch
 *       statistics, finished by rl_bn_finalize with that slot count).                        */
typedef struct rl_gemm_desc {
    const float* A;
    int64_t lda, a_bstride;
    int32_t a_mode;
    int32_t in_act;
    float in_slope;
    const float* in_scale;
    const float* in_shift;
    const float* xyz;
    int64_t xyz_bstride;
    const int32_t* nbr_idx;
    const float* nbr_d2;
    int32_t nbr_k;
    int32_t B, n, N, K;
    const float* W;
    int64_t w_ks, w_ns;
    const float* bias;
    float* Y;
    int64_t ldy, y_bstride;
    int32_t accumulate;
    double* stats;
    /* optional scratch of rl_gemm_kslab_floats(M,N,K) floats: lets a wide layer with few rows split K
     * over workgroups (deterministic two-pass reduction); NULL or too small = single pass */
    float* kslab;
    int64_t kslab_floats;
    /* optional "split-scatter" epilogue (the dX of PointFeatureAugmentation's concat, modules.py:213-221):
     *   addend  != NULL : v += addend[R*N + c] before anything else (row stride N, row R = global row)
     *   out2    != NULL : columns c >= split_col are not stored to Y.  With out2_index == NULL they are stored to
     *                     out2[R*(N - split_col) + c - split_col] (a dense (M, N - split_col) tensor: the gradient of
     *                     every gathered row, summed per destination afterwards by rl_segment_sum_rows in a fixed
     *                     order - the deterministic path the network uses).  With out2_index they are atomically
     *                     added to out2[(b*out2_bstride + out2_index[R])*(N - split_col) + c - split_col], b = R / rows
     *                     per cloud (order-dependent in the last bits; out2 zeroed by the caller).
     *                     Columns c < split_col go to Y as usual (ldy, y_bstride, accumulate apply to them only).
     * Needs the LDS-tiled kernel (K or N > 64, 16-byte aligned operands) and no statistics; else RL_ERR_UNSUPPORTED. */
    const float* addend;
    float* out2;
    const int32_t* out2_index;
    int64_t out2_bstride;
    int32_t split_col;
    /* optional, wide layers in the bf16 arithmetic modes: the weight already split into bf16 head / tail planes in the
     * orientation THIS product reads it - plane[n*K + k] = W(k, n), tails N*K elements after the heads (rl_split_weights;
     * 16-byte aligned, K % 8 == 0).  Selects the 8-wavefront kernel: no conversion of the weight per tile, four
     * wavefronts per SIMD instead of two.  Same arithmetic, same results as without it. */
    const void* W_split;
    /* optional, with stats: SHIFTED statistics.  The partial sums are those of (y - pivot) and (y - pivot)^2 with
     * pivot[c] = stats_pivot_mean[c] - (stats_pivot_bias ? stats_pivot_bias[c] : 0) - the layer's running mean (what torch's
     * BatchNorm2d keeps, modules.py:85-89), minus the conv bias this product leaves out (rl_bn_finalize folded_bias) -
     * subtracted per element BEFORE squaring, so that a channel whose spread is a few 1e-4 of its mean keeps its variance
     * (the reference's ATen BatchNorm is two-pass; E[y^2] - E[y]^2 on fp32 partial sums is not).  rl_bn_finalize must then be
     * told `pivoted`.  NULL = plain sums (pivot 0). */
    const float* stats_pivot_mean;
    const float* stats_pivot_bias;
    /* optional (round 6), streaming kernel only (K, N <= 64: rl_gemm_streams(d) == 1): this product is the gradient G w.r.t. the
     * ACTIVATED output of a BatchNorm layer - the input gradient of the layer behind it - and `stats` receives, instead of the
     * forward sums, what rl_bn_bwd_reduce would leave for that layer: sum g' and sum g' xhat per column, g' = G act'(Yl scale +
     * shift), xhat = (Yl - mean) invstd, with Yl = bnb_Y the layer's raw output (same rows and row layout as Y), bnb_scale /
     * bnb_shift its folded BatchNorm, bnb_mean / bnb_invstd its saved batch statistics, bnb_act / bnb_slope its activation.
     * rl_gemm_stat_slots(M, N, K) slots; then rl_bn_bwd_finalize + rl_bn_bwd_apply - no reduce sweep over G and Yl.
     * Y must be complete after this call (the only or the last accumulating writer).  No pivot. */
    const float* bnb_Y;
    const float* bnb_scale;
    const float* bnb_shift;
    const float* bnb_mean;
    const float* bnb_invstd;
    int32_t bnb_act;
    float bnb_slope;
} rl_gemm_desc;

/* TWO products over one A' in ONE launch of the LDS-DMA wide GEMM (round 6): Y1 = A'.W1, Y2 = A'.W2 - mlp1 and shortcut of an
 * encoder level (modules.py:314, 325), which both read the level's input: the rows are fetched once and the narrow product's
 * few tiles ride along with the wide one's.  Both descriptors must name the same A' (pointer, strides, lazy transform; a_mode 0,
 * K % 32 == 0, K <= 1024) and carry W_split (bf16 arithmetic modes); neither product narrow enough for the exact-fp32 streaming
 * kernel (K <= 64 and N <= 64: its arithmetic would change); dense outputs (ldy == N, y_bstride == n);
 * statistics (with their pivots) for both or for none - each into its own buffer, rl_row_blocks(M, 128) slots; no bias, no
 * accumulate, no split epilogue.  Bitwise the results of two rl_gemm calls on 128 x 128 tiles.
 * rl_gemm_pair_supported: 1 if the pair can go out as one launch, else the caller issues two rl_gemm calls. */
int rl_gemm_streams(const rl_gemm_desc* d);
int rl_gemm_pair_supported(const rl_gemm_desc* a, const rl_gemm_desc* b);
int rl_gemm_pair(const rl_gemm_desc* a, const rl_gemm_desc* b, void* stream);

/* Splits weights for rl_gemm_desc.W_split: out (2*N*K bf16) <- heads, then tails, of W(k, n) = W[k*w_ks + n*w_ns].
 * One launch for all items (every wide layer of a step, in both orientations: forward and dgrad). */
typedef struct rl_wsplit_item {
    const float* W;
    int64_t w_ks, w_ns;
    int32_t K, N;
    void* out;
} rl_wsplit_item;

int rl_split_weights(const rl_wsplit_item* items, int count, void* stream);

int64_t rl_gemm_kslab_floats(int64_t M, int N, int K);
/* slots of `stats` an (M, N, K) product fills (<= RL_MAX_SLOTS): one per 128-row block, or one per 64-row block where the
 * wide GEMM may run on 64-row output tiles (few rows, N > 64) - every kernel zero-fills the slots it does not use, so this
 * is the count to hand to rl_bn_finalize whichever kernel took the launch. */
int64_t rl_gemm_stat_slots(int64_t M, int N, int K);

/* Arithmetic of the wide kernels (K or N > 64: rl_gemm's LDS-tiled kernel, rl_wgrad's 128x128 kernel):
 *   "bf16x3" (default)  every fp32 operand is split into a bf16 head and tail on its way into LDS and a product is
 *                       a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate: ~2^-16
 *                       relative product error; network logits within 2.5e-6 of the fp32 CPU forward
 *   "fp32"              v_mfma_f32_16x16x4_f32, bitwise an fp32 FMA chain
 *   "bf16"              heads only: plain bf16 operands, fp32 accumulate (does NOT meet the 1e-3 logits bound)
 * The environment variable RL_WIDE_GEMM sets the initial mode.  Storage, accumulation, statistics and the
 * narrow (streaming) kernels are fp32 in every mode.  Not thread-safe against concurrent launches.           */
int rl_set_wide_gemm(const char* mode);
const char* rl_get_wide_gemm(void);
/* How the wide forward / input-gradient GEMM of the bf16 modes stages its operands:
 *   "dma" (default)  wgemm2_kernel: one persistent 12-wavefront workgroup per CU; four loader wavefronts bring the raw fp32
 *                    rows and the pre-split weight planes into LDS rings by LDS-DMA (global_load_lds_dwordx4), four chunks
 *                    deep and across the workgroup's tiles; eight compute wavefronts convert on the fragment
 *   "registers"      wgemm_kernel: global loads converted on their way into LDS, one chunk ahead, two workgroups per CU
 * Same products in the same order: Y is bitwise the same (the BatchNorm partial sums are grouped differently).
 * K % 32 != 0 or K > 1024 always takes "registers".  RL_WGEMM_STAGING sets the initial choice.                        */
int rl_set_wgemm_staging(const char* how);
/* Output tile of the "dma" kernel: "auto" (default) = 128 x 128, or 64 x 128 / 64 x 64 where 128 x 128 tiles would leave
 * at most half / a quarter of the CUs with a tile (the deep levels: few rows) - more workgroups in one pass, mostly no K split;
 * "128" = always 128 x 128 (round 5).  Y is bitwise the same; the BatchNorm partial sums are grouped per 64 rows
 * (rl_gemm_stat_slots).  RL_WGEMM_TILE sets the initial choice.                                                         */
int rl_set_wgemm_tile(const char* how);
/* Diagnostics: enable = 0 switches the K split of wide products with few output tiles off (one summation order whatever the
 * kernel and its tile: what the bitwise kernel-against-kernel tests compare on); 1 (default) restores it.               */
int rl_set_gemm_ksplit(int enable);
/* Diagnostics: the streaming GEMM (K, N <= 64) on 1/div of its workgroups (1 = default).  Y is bitwise the same; the rows a
 * lane adds into its BatchNorm partial sums change - the regression knob for "a re-grouping of the statistics must not move
 * a gradient" (tests/test_net_gpu.py; round 4 met 0.5 - 3 % before the sums were shifted).                              */
int rl_set_sgemm_grid_div(int div);
int rl_gemm(const rl_gemm_desc* d, void* stream);

/* Weight / bias gradient of the same layer:  dW(k,c) = sum_r A'[r][k] * dY[r][c],
 * db[c] = sum_r dY[r][c].  The A-operand fields have the meaning they have in rl_gemm_desc.
 * dY row (b,i) = dY + (b*dy_bstride + i)*lddy.  dW is written with the strides (w_ks, w_ns) of
 * the weight it belongs to.  Deterministic: per-block partial slabs in `slab` (at least
 * rl_wgrad_slab_floats() floats) are summed in a fixed order by a second kernel.            */
typedef struct rl_wgrad_desc {
    const float* A;
    int64_t lda, a_bstride;
    int32_t a_mode;
    int32_t in_act;
    float in_slope;
    const float* in_scale;
    const float* in_shift;
    const float* xyz;
    int64_t xyz_bstride;
    const int32_t* nbr_idx;
    const float* nbr_d2;
    int32_t nbr_k;
    int32_t B, n, N, K;
    const float* dY;
    int64_t lddy, dy_bstride;
    float* dW;
    int64_t w_ks, w_ns;
    float* dbias; /* nullable */
    float* slab;
    int64_t slab_floats;
    /* != 0: only the partial slabs are written; the caller sums them later with
     * rl_wgrad_reduce_batch (one launch for all layers of a backward pass), so `slab` must be
     * private to this layer until then */
    int32_t defer_reduce;
    /* bf16-storage mode: A and dY are bf16 rows (lda / lddy count elements).  Wide layers only (a 128 x 128 tile of dW,
     * i.e. N or K > 64), plain A operand, outside the fp32 arithmetic mode; the products are then exact bf16 x bf16. */
    int32_t rows_bf16;
} rl_wgrad_desc;

int64_t rl_wgrad_slab_floats(int64_t M, int N, int K);
int rl_wgrad(const rl_wgrad_desc* d, void* stream);

/* Deferred second pass of rl_wgrad for `count` layers in one launch per 48 items: item i sums
 * the nsplit = slab_floats_used / (N*K + N) partial slabs of its layer in the same fixed order
 * as the per-layer reducer and writes dW (strides w_ks, w_ns) and dbias (nullable).
 * `items` is a HOST array (copied into the kernel arguments).                                 */
typedef struct rl_wgrad_reduce_item {
    const float* slab;
    float* dW;
    float* dbias;
    int64_t w_ks, w_ns;
    int32_t nsplit, N, K;
    int32_t reserved;
} rl_wgrad_reduce_item;

int rl_wgrad_nsplit(int64_t M, int N, int K);
/* Several layers' weight gradients in ONE launch: the layers of a backward pass are independent of each other and mostly
 * small (the wide ones 80 - 216 workgroups each on a chip that holds 512, the narrow ones 5 - 46 us each), so one by one
 * they leave most CUs idle and pay a launch each.  rl_wgrad_batchable: 0 = must go through rl_wgrad, 1 = joins the grouped
 * launch of the wide (128 x 128-tile) kernel (fp32 rows, bf16x3 / bf16 arithmetic), 2 = joins the grouped launch of the
 * narrow (streaming) kernel (K, N <= 64, fp32 rows, tensor operand).  rl_wgrad_batch takes a queue of either or both kinds
 * (one launch per kind and per 24 layers); every descriptor must have defer_reduce set; the partial slabs are exactly
 * those of rl_wgrad (same split, same per-workgroup arithmetic: bitwise equal results) and are summed by
 * rl_wgrad_reduce_batch as before.                                                                                    */
int rl_wgrad_batchable(const rl_wgrad_desc* d);
int rl_wgrad_batch(const rl_wgrad_desc* descs, int count, void* stream);
int rl_wgrad_reduce_batch(const rl_wgrad_reduce_item* items, int count, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d(eps, momentum) folded into per-channel (scale, shift).
 * training != 0: mean/var from the `nslots` partial sums written by a producer over `count`
 *   rows; running stats updated as torch does (running = (1-m)*running + m*batch, unbiased
 *   variance; num_batches_tracked += 1 when nbt != NULL); saves mean and invstd for backward.
 * training == 0: scale/shift from the running statistics; stats may be NULL.
 * folded_bias (C floats or NULL): the bias of the layer in front (SharedMLP's conv bias, modules.py:93-104), when the
 *   producer left it OUT of the tensor - it cancels in (y - mean), so the GEMM epilogue need not add it.  The statistics
 *   and the saved mean are then those of (y - bias), (scale, shift) apply to that tensor, and the running mean is kept
 *   as the reference keeps it: that of y (mean + bias); in eval mode the running mean is read as (running_mean - bias).
 * pivoted != 0 (training): the partial sums are SHIFTED - those of (t - p) and (t - p)^2 of the stored tensor t,
 *   p[c] = m[c] - folded_bias[c] with m = `pivot` when given, else running_mean (either as it stands BEFORE this call's
 *   update): mean = p + S/n, var = Q/n - (S/n)^2.  The producer must have been given the same two vectors
 *   (rl_gemm_desc.stats_pivot_*, rl_pool_desc.pivot_mean*).
 * pivot (C floats or NULL; training): the caller's pivot vector of this layer (round 6).  After the fold it holds THIS
 *   batch's mean of y (mean + folded bias) - the next step's pivot: always within the batch-to-batch drift of the mean,
 *   wherever a loaded checkpoint's running mean sits (a pivot ten standard deviations off costs the variance two digits);
 *   zero it for a fresh / freshly loaded model (pivot 0 = plain sums for one step). */
int rl_bn_finalize(const double* stats, int nslots, int64_t count, int C, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                   float momentum, float eps, int training, float* scale, float* shift,
                   float* save_mean, float* save_invstd, const float* folded_bias, int pivoted, float* pivot, void* stream);

/* Several independent layers' folds in one launch (same arithmetic per layer as rl_bn_finalize: same results).  The folds of
 * the layers at one dependency depth of an encoder level - mlp1 / shortcut / mlp_rpe1, then pool1.mlp / mlp_rpe2 - are wanted
 * at the same moment, and a fold is a launch that costs ten times its work.                                              */
typedef struct rl_bn_finalize_item {
    const double* stats;
    int64_t count;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    int64_t* num_batches_tracked;
    float* scale;
    float* shift;
    float* save_mean;
    float* save_invstd;
    const float* folded_bias;
    int32_t nslots, C, training;
    float momentum, eps;
    int32_t pivoted;
    float* pivot;
} rl_bn_finalize_item;
int rl_bn_finalize_batch(const rl_bn_finalize_item* items, int count, void* stream);

/* BatchNorm + activation backward for a lazy tensor Y (rows x C, row (b,i) at (b*bstride+i)*ld)
 * whose activated value received gradient G (same addressing):
 *   g = G * act'(Y*scale+shift);  xhat = (Y-mean)*invstd
 *   rl_bn_bwd_reduce : partial sums of g and g*xhat -> stats[slot][0|1][c], slot < rl_bn_bwd_slots(M)
 *   rl_bn_bwd_finalize: dgamma = sum g*xhat, dbeta = sum g, coef[0][c] = mean g, coef[1][c] = mean g*xhat
 *   rl_bn_bwd_apply  : G <- scale * (g - coef0 - xhat*coef1)      (training)
 *                      G <- scale * g                              (coef == NULL: eval / no BN)  */
typedef struct rl_bn_bwd_desc {
    float* G;
    const float* Y;
    int64_t ld, bstride;
    int32_t B, n, C;
    int32_t act;
    float slope;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* invstd;
    double* stats;      /* reduce: out */
    const float* coef;  /* apply: in, 2*C floats, or NULL */
} rl_bn_bwd_desc;

/* (nslots, 2, C) per-workgroup partials -> (2, C) totals in slot order.  SyncBN / equivalence mode: the caller
 * all-reduces the totals over the ranks and passes them on as ONE slot with the global row count
 * (rl_bn_finalize / rl_bn_bwd_finalize with nslots = 1).                                             */
int rl_bn_reduce_slots(const double* stats, int nslots, int C, double* out, void* stream);
int rl_bn_bwd_slots(int64_t rows);
int rl_bn_bwd_reduce(const rl_bn_bwd_desc* d, void* stream);
int rl_bn_bwd_finalize(const double* stats, int nslots, int64_t count, int C, float* dgamma,
                       float* dbeta, float* coef, void* stream);
/* rl_bn_bwd_finalize for the two BatchNorms behind a residual junction (same row and channel counts) in one launch. */
int rl_bn_bwd_finalize_pair(const double* stats0, const double* stats1, int nslots, int64_t count, int C, float* dgamma0,
                            float* dbeta0, float* coef0, float* dgamma1, float* dbeta1, float* coef1, void* stream);
/* rl_bn_bwd_finalize for up to 8 unrelated layers in one launch (round 6): e.g. a virtual rpe stage (sums left by its pooling
 * kernel) together with the per-point layer whose reduce sweep ran right behind it.  Same arithmetic per layer. */
typedef struct rl_bn_bwd_finalize_item {
    const double* stats;
    int64_t count;
    float* dgamma;
    float* dbeta;
    float* coef;
    int32_t nslots, C;
} rl_bn_bwd_finalize_item;
int rl_bn_bwd_finalize_batch(const rl_bn_bwd_finalize_item* items, int count, void* stream);
int rl_bn_bwd_apply(const rl_bn_bwd_desc* d, void* stream);
/* The three steps in ONE launch for a small tensor (rows <= 2048, C % 4 == 0, ld % 4 == 0, 16-byte aligned; ask
 * rl_bn_bwd_fused_supported): a workgroup owns a channel quad and all its rows, so the batch sums never leave it.  G becomes
 * the gradient w.r.t. Y in place, dgamma / dbeta (and, when coef_out != NULL, the 2*C means rl_bn_bwd_finalize would
 * leave) are written.  Same expressions per element; the sums are grouped per wavefront (another fixed order).
 * Replaces the same reference lines as the three (BatchNorm2d backward of modules.py:85-89).                          */
int rl_bn_bwd_fused_supported(int64_t rows, int C, int64_t ld);
int rl_bn_bwd_fused(const rl_bn_bwd_desc* d, int64_t count, float* dgamma, float* dbeta, float* coef_out, void* stream);

/* Backward of the residual junction of LocalFeatureAggregation (modules.py:325):
 *     O = LeakyReLU_slope( BN1(Y1) + BN2(Y2) ),   Y1 = mlp2 output, Y2 = shortcut output, no activation of their own.
 * Both branches receive the same g = G * (O > 0 ? 1 : slope); one sweep reduces both BatchNorms' sums
 * (rl_resid_bn_bwd_reduce fills stats1 / stats2 in the slot layout of rl_bn_bwd_reduce, to be finished by
 * rl_bn_bwd_finalize each) and one sweep writes both results (rl_resid_bn_bwd_apply):
 *     G  <- scale1 * (g - coef1[0] - xhat1 * coef1[1]),   G2 <- scale2 * (g - coef2[0] - xhat2 * coef2[1]).
 * All tensors are dense rows x C (C % 4 == 0, C/4 a power of two <= 256, 16-byte aligned); this replaces
 * rl_add_act_bwd + a copy + two rl_bn_bwd_reduce + two rl_bn_bwd_apply (15 passes over the tensor -> 10).       */
typedef struct rl_resid_bn_bwd_desc {
    float* G;                /* in: dL/dO;  apply: out, gradient w.r.t. Y1 */
    float* G2;               /* apply: out, gradient w.r.t. Y2 */
    const float* O;
    float slope;
    int64_t rows;
    int32_t C;
    const float* Y1; const float* scale1; const float* mean1; const float* invstd1;
    const float* Y2; const float* scale2; const float* mean2; const float* invstd2;
    double* stats1; double* stats2;          /* reduce: out */
    const float* coef1; const float* coef2;  /* apply: in (2*C floats each) */
} rl_resid_bn_bwd_desc;

int rl_resid_bn_bwd_supported(int64_t rows, int C);
int rl_resid_bn_bwd_reduce(const rl_resid_bn_bwd_desc* d, void* stream);
int rl_resid_bn_bwd_apply(const rl_resid_bn_bwd_desc* d, void* stream);
/* ... and the junction's two sweeps + rl_bn_bwd_finalize_pair as ONE launch for rows <= 2048 (stats / coef fields unused). */
int rl_resid_bn_bwd_fused_supported(int64_t rows, int C);
int rl_resid_bn_bwd_fused(const rl_resid_bn_bwd_desc* d, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row movement.  dst row r of `rows` rows (r = b*rows_per_batch + i) receives `C` channels:
 *   src row = b*src_bstride + (index ? (index_shared ? index[i] : index[r]) : i)
 *   dst[r*ldd + c] (=|+=) lazy(src[src_row*lds + c]),  c < C
 * index may be int32 (index32) or int64 (index64); at most one is non-NULL.
 * Covers: input permutation, PointFeatureAugmentation gather+concat, decoder interpolation
 * gather + skip concat, gradient splits.                                                    */
typedef struct rl_rows_desc {
    const float* src;
    int64_t lds, src_bstride;
    float* dst;
    int64_t ldd;
    int64_t rows, rows_per_batch;
    int32_t C;
    const int32_t* index32;
    const int64_t* index64;
    int32_t index_shared;
    int32_t accumulate;
    int32_t act;
    float slope;
    const float* scale;
    const float* shift;
} rl_rows_desc;

int rl_copy_rows(const rl_rows_desc* d, void* stream);
/* Two independent copies (e.g. the two halves of a concat, modules.py:183 / 362) in one launch; same results as two calls. */
int rl_copy_rows_pair(const rl_rows_desc* d0, const rl_rows_desc* d1, void* stream);

/* Transposed movement (gather backward): dst[(b*src_bstride + index[r])*ldd + c] += src[r*lds + c]
 * (src_bstride names the batch stride of the INDEXED tensor, here dst) with fp32 atomics.  NON-DETERMINISTIC (the order of
 * the adds is not fixed; dst must be zeroed by the caller) and NOT used by the network's schedule, whose gather backward is
 * rl_csr_build + rl_segment_sum_rows: refused with RL_ERR_UNSUPPORTED unless RL_ALLOW_FLOAT_ATOMICS=1 is in the environment
 * (the same holds for rl_gemm's out2_index).                                                                          */
int rl_scatter_add_rows(const rl_rows_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Deterministic gather backward.  The reference's backward of `torch.gather(features, idx)`
 * (PointFeatureAugmentation, modules.py:213-221; nearest_neighbor_interpolation, modules.py:359-364)
 * is a scatter_add_ whose summation order is undefined.  Here the neighbour graph is transposed once
 * per step and every destination row adds its contributions in ascending source-row order.
 *
 * rl_csr_build: for each task (a graph idx (B, n_src, k) int32 with values in [0, n_dst)) writes
 *   offsets (B, n_dst + 1): segment bounds of destination j of cloud b, local to the cloud
 *   entries (B, n_src*k)  : local source rows r = i*k + kk with idx[b][i][kk] == j, ascending per segment
 * All tasks (<= 8, same B) run in one launch set; workspace of rl_csr_workspace_bytes bytes, 256-byte
 * aligned.  Out-of-range indices are ignored.                                                   */
typedef struct rl_csr_task {
    const int32_t* idx;
    int32_t n_src, k, n_dst;
    int32_t* offsets;
    int32_t* entries;
} rl_csr_task;

int64_t rl_csr_workspace_bytes(const rl_csr_task* tasks, int ntasks, int B);
int rl_csr_build(const rl_csr_task* tasks, int ntasks, int B, void* workspace, int64_t workspace_bytes,
                 void* stream);

/* dst[(b*dst_bstride + j)*ldd + c] (=|+=) sum over e in [offsets[b][j], offsets[b][j+1]) of
 * src[(b*src_bstride + entries[b][e])*lds + c],  c < C, added in the order of the segment.
 * `src` may point at a column offset inside wider rows (lds is the row stride).                 */
typedef struct rl_segsum_desc {
    const float* src;
    int64_t lds, src_bstride;
    float* dst;
    int64_t ldd, dst_bstride;
    const int32_t* offsets;
    const int32_t* entries;
    int64_t entries_per_cloud;
    int32_t B, n_dst, C;
    int32_t accumulate;
    int32_t src_bf16;       /* bf16-storage mode: the source rows are bf16 (lds counts elements); sums and dst stay fp32 */
} rl_segsum_desc;

int rl_segment_sum_rows(const rl_segsum_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attentive pooling core (modules.py:246-253) on X, S of shape (P*K) x C (K consecutive rows
 * per point): A = softmax over the K rows of S, per channel;  Pout[p][c] = sum_k A*X.
 * Backward: dS = A * dP * (X - Pout);  dXa = dP * A  (the direct path of X).               */
int rl_attpool_fwd(const float* X, const float* S, int64_t P, int K, int C, float* Pout,
                   void* stream);
int rl_attpool_bwd(const float* X, const float* S, const float* Pout, const float* dP, int64_t P,
                   int K, int C, float* dS, float* dXa, void* stream);

/* Fused attentive pooling (d = 16 / 32 / 64, and 128 in the bf16x3 / bf16 arithmetic modes; 16 neighbours): gather + concat
 * (modules.py:213-221), score Linear + softmax over K + weighted sum (modules.py:246-253) without
 * materialising the (points*K) x d tensors.
 *   U   (points*16) x d/2 lazy rpe-branch features;  G  lazy per-point features, row (b,i) at
 *   (b*g_bstride + i)*(d/2), gathered through idx (points,16) int32;  W (d,d) score weight.
 * rl_pool_fwd : Pout (points, d).
 * rl_pool_bwd : given dP (points, d) recomputes the block and produces GU (gradient w.r.t. the
 *   activated U, stored or accumulated), DG ((points*16) x d/2, plain stores: the gradient w.r.t. the
 *   activated gathered row of every neighbourhood slot - rl_segment_sum_rows then sums it per gathered
 *   point in a fixed order) and dW (d,d) through per-workgroup slabs (rl_pool_slab_floats floats).  */
typedef struct rl_pool_desc {
    const float* U;
    const float* u_scale;
    const float* u_shift;
    int32_t u_act;
    float u_slope;
    const float* G;
    int64_t g_bstride;
    const float* g_scale;
    const float* g_shift;
    int32_t g_act;
    float g_slope;
    const int32_t* idx;
    const float* W;
    int64_t points;
    int32_t n, d, nbr_k;
    float* Pout;
    const float* dP;
    float* GU;
    int32_t gu_accumulate;
    float* DG;
    float* dW;
    float* slab;
    int64_t slab_floats;
    /* d = 128 (supported outside the fp32 arithmetic mode): rl_pool_bwd does not produce dW; it writes the block's
     * X = [rpe, gathered] and dS, (points*16) x 128 floats each, and the caller computes dW = dS^T.X with rl_wgrad
     * (dY = dS_out, A = X_out).  dW / slab are ignored then. */
    float* X_out;
    float* dS_out;
    /* u_source 1 / 2 (d = 16 / 32 / 64): the rpe-branch half of X is not read from U but RECOMPUTED per point from the
     * coordinates - the outputs of mlp_rpe1 / mlp_rpe2 (modules.py:287-291, 313-320) never exist in memory:
     *   1: relu(bn1(rpe . W1^T + b1))                          rpe = [x_i, x_nbr, x_i - x_nbr, sqrt(d2)] (modules.py:173-186)
     *   2: relu(bn2(relu(bn1(rpe . W1^T + b1)) . W2^T + b2))
     * xyz (B, xyz_bstride, xyz_width) with xyz_width 3 (0 means 3) or 4 (x, y, z and one unused float per point,
     * 16-byte aligned: one gather per neighbour instead of three); nbr_d2 (points,16); W1 (d/2, 10), W2 (d/2, d/2)
     * reference conv layouts [out][in];
     * scale / shift = the folded BatchNorms (rl_bn_finalize on rl_rpe_stats' partials), mean / invstd = their saved
     * batch statistics (backward entry points only).  U is ignored.                                         */
    int32_t u_source;
    const float* xyz;
    int64_t xyz_bstride;
    const float* nbr_d2;
    const float* W1;
    const float* b1;
    const float* scale1;
    const float* shift1;
    const float* W2;
    const float* b2;
    const float* scale2;
    const float* shift2;
    const float* mean1;
    const float* invstd1;
    const float* mean2;
    const float* invstd2;
    int32_t xyz_width;
    /* rl_pool_bwd with a virtual stage, optional: when this launch COMPLETES the gradient of the stage's activated
     * output (GU after it holds the total), it also leaves the batch-statistics partials of that stage's BatchNorm
     * backward - what rl_rpe_bn_reduce would compute from GU - in bn_bwd_stats[slot][2][d/2], slot <
     * rl_pool_bwd_slots(points, d), saving that pass. */
    double* bn_bwd_stats;
    /* rl_pool_fwd with u_source 1, optional: the batch statistics of the NEXT stage's raw output (mlp_rpe2 applied to the
     * tile this launch has in registers) - what rl_rpe_stats with u_source 2 would compute - as partials
     * bn_fwd_stats2[slot][2][d/2], slot < rl_pool_fwd_slots(points, d).  Needs W2 / b2. */
    double* bn_fwd_stats2;
    /* bf16-storage mode (backward entry points): the (points*16)-row gradient tensors this block WRITES are bf16 (2 bytes
     * per element, round to nearest even) instead of fp32: DG, X_out and dS_out always; GU - and the G / GU1 arguments
     * of rl_rpe_bn_reduce / rl_rpe_wgrad - when the rpe branch is virtual (u_source > 0).  With a real U tensor GU stays
     * fp32 (it continues into the fp32 GEMM chain).  Needs the bf16x3 arithmetic mode.  0: everything fp32.        */
    int32_t rows_bf16;
    /* SHIFTED BatchNorm statistics (see rl_gemm_desc.stats_pivot_*): the running mean of mlp_rpe1's / mlp_rpe2's BatchNorm
     * (modules.py:288-291).  With pivot_mean1, rl_rpe_stats (u_source 1) leaves the sums of (y - pivot_mean1) and its square,
     * y = raw + b1; with pivot_mean2 so do rl_rpe_stats (u_source 2) and rl_pool_fwd's bn_fwd_stats2 for the second stage.
     * rl_bn_finalize must then be called with `pivoted` (and no folded bias: these sums are those of y).  NULL: plain sums. */
    const float* pivot_mean1;
    const float* pivot_mean2;
} rl_pool_desc;

int rl_pool_supported(int d, int nbr_k);
int64_t rl_pool_slab_floats(int64_t points, int d);
int rl_pool_bwd_slots(int64_t points, int d);   /* workgroups (= partial slots) of an rl_pool_bwd launch with a virtual stage */
/* workgroups of an rl_pool_bwd launch = partial dW slabs it leaves.  With rl_pool_desc.dW == NULL the call does not sum them: the
 * caller queues (slab, nsplit = this, N = K = d, stride d*d + d) for rl_wgrad_reduce_batch.  rl_pool_slab_floats covers both. */
int rl_pool_bwd_grid(int64_t points, int d, int virtual_stage);
int rl_pool_fwd_slots(int64_t points, int d);   /* workgroups (= partial slots) of an rl_pool_fwd launch */
int rl_pool_fwd(const rl_pool_desc* d, void* stream);
int rl_pool_bwd(const rl_pool_desc* d, void* stream);

/* BatchNorm batch statistics of the RAW output of stage u_source (1: mlp_rpe1, 2: mlp_rpe2) of the virtual rpe branch
 * described by d (xyz, idx, nbr_d2, W1, b1 [, scale1, shift1, W2, b2]): per-workgroup partials (sum, sum of squares;
 * doubles) stats[slot][2][d/2], slot < rl_rpe_stats_slots(points) - what the GEMM epilogue of that layer would have left
 * for rl_bn_finalize, without the (points*16) x d/2 tensor.                                                    */
int rl_rpe_stats_slots(int64_t points);
int rl_rpe_stats(const rl_pool_desc* d, double* stats, void* stream);

/* Backward of virtual stage u_source (BatchNorm + ReLU + the stage's Linear).  G (points*16, d/2) is the gradient w.r.t.
 * the ACTIVATED stage output, as rl_pool_bwd writes it to GU; the raw tile is recomputed (scale / shift / mean / invstd
 * of the stage's BatchNorm - and of stage 1 for stage 2 - must be set in d).
 *   rl_rpe_bn_reduce : partials of sum g and sum g*xhat, g = G*[output > 0]: stats[slot][2][d/2], slot <
 *                      rl_rpe_stats_slots(points), for rl_bn_bwd_finalize (-> dgamma, dbeta, coef).
 *   rl_rpe_wgrad     : dY = scale*(g - coef[c] - xhat*coef[d/2 + c]); per-workgroup partial slabs [slot][h*Kin + h]
 *                      (dW[n][k], then db[n]; Kin = 10 for stage 1, d/2 for stage 2), nsplit = rl_rpe_stats_slots(points),
 *                      to be summed by rl_wgrad_reduce_batch; stage 2 also writes GU1 (points*16, d/2) = dY . W2, the
 *                      gradient w.r.t. the activated stage-1 output.                                                  */
int rl_rpe_bn_reduce(const rl_pool_desc* d, const float* G, double* stats, void* stream);
int64_t rl_rpe_wgrad_slab_floats(int64_t points, int d, int stage);
int rl_rpe_wgrad(const rl_pool_desc* d, const float* G, const float* coef, float* slab, int64_t slab_floats,
                 float* GU1, void* stream);

/* Residual sum of two lazy tensors + LeakyReLU (modules.py:325):
 *   O = lrelu(Y1*s1+b1 + Y2*s2+b2);  backward (in place): G <- G * (O > 0 ? 1 : slope)       */
int rl_add_act_fwd(const float* Y1, const float* s1, const float* b1, const float* Y2,
                   const float* s2, const float* b2, int64_t rows, int C, float slope, float* O,
                   void* stream);
int rl_add_act_bwd(float* G, const float* O, int64_t rows, int C, float slope, void* stream);

/* RelativePositionEncoding (modules.py:173-186) written out once per level: row (b, i, j) of out, 12 floats
 * apart, holds [x_i, x_nbr, x_i - x_nbr, sqrt(d2)] (10 channels) and two zeros of padding, so that the row
 * is 16-byte aligned and the tensor can be the A operand of rl_gemm / rl_wgrad with K = 10, lda = 12.
 * The a_mode-1 operand source computes the same values inside the kernels; this tensor is the faster choice
 * when it is read more than once (forward + weight gradient).  xyz (B, xyz_bstride, 3), idx/d2 (B,n,k).   */
int rl_rpe_build(const float* xyz, int64_t xyz_bstride, const int32_t* nbr_idx, const float* nbr_d2, int B,
                 int n, int k, float* out, void* stream);
/* the same with the DISTANCES given (RelativePositionEncoding.forward's own signature, modules.py:159-186) */
int rl_rpe_build_dist(const float* xyz, int64_t xyz_bstride, const int32_t* nbr_idx, const float* nbr_dist, int B,
                      int n, int k, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input pipeline on the device: PointCloudPreprocessor.preprocess
 * (randlanet/utils/dataset.py:61-97) + perturbate_point_cloud (randlanet/utils/augmentation.py:147-167)
 * + the DataLoader collation, for a batch of clouds that are resident in HBM.  The caller draws the
 * random numbers (sample indices, jitter noise, scale, rotation matrix, shift draws), this call applies
 * them in the reference's order with float64 arithmetic and rounds to float32 at the end, exactly
 * where the reference does (dataset.py:51).  One rl_cloud_job per cloud, array in DEVICE memory.
 *   indices (B,n) int64: rows of the source cloud to take (preprocessing.sample_points, :35-62)
 *   noise   (B,n,3) float64 standard-normal draws for the jitter, or NULL (no jitter)
 *   scratch rl_batch_assemble_scratch_doubles(B, n) float64 of work space ((B,n,3) coordinates + the records / arrival
 *           counters of the cloud-wide sums; any content - the call resets what it needs; one scratch per concurrent call)
 * Up to 16 workgroups share a cloud and meet at an arrival-counter barrier per cloud-wide sum; their number is sized so that
 * a launch fits the device at once (occupancy calculator, per device), and the wait is BOUNDED: a workgroup whose peers do not
 * arrive within ~1 s (a CU mask or partition mode the attribute does not show) gives up, sets the cloud's error word in
 * `scratch` (32-bit word rl_batch_assemble_flag_u32(B, n, b): 0 = fine) and writes the finite coordinates it has - never NaN -
 * instead of hanging the GPU; the caller reads the B words back (asynchronously) and treats a non-zero one as a failed
 * launch.  RL_ASSEMBLE_ONE_WG=1 forces one workgroup per cloud (no rendezvous).
 *   out_input (B,n,3+F) float32 = [xyz, features], out_labels (B,n) int64                       */
typedef struct rl_cloud_job {
    const void* xyz;          /* (n_points,3) float32, or float64 when xyz_f64 != 0 */
    const float* features;    /* (n_points,F) */
    const int64_t* labels;    /* (n_points,) */
    int64_t n_points;
    int32_t xyz_f64;
    int32_t normalization;    /* 0 none, 1 "mean", 2 "max", 3 "stdev", 4 any other string (centre only) */
    int32_t augment;          /* 0: the fields below are ignored */
    int32_t reserved;
    double jitter_variance, jitter_limit;
    double scale;             /* np.random.uniform(1 - scale_limit, 1 + scale_limit) */
    double R[9];              /* Rz.Ry.Rx, row-major */
    double shift[3];          /* np.random.uniform(-shift_limit, shift_limit, 3), before the radius factor */
} rl_cloud_job;

int64_t rl_batch_assemble_scratch_doubles(int B, int n);
int64_t rl_batch_assemble_flag_u32(int B, int n, int b);
int rl_batch_assemble(const rl_cloud_job* jobs_dev, int B, int n, int F, const int64_t* indices,
                      const double* noise, double* scratch, float* out_input, int64_t* out_labels,
                      void* stream);
/* The random draws rl_batch_assemble consumes, made on the device in ONE launch (the device loader's fast mode; the
 * reference draws them from numpy's global stream - preprocessing.py:35-62 np.random.choice, augmentation.py:147 np.random.randn -
 * which is the loader's "numpy" mode): indices (B,n) = n rows of each cloud without replacement (a keyed permutation of
 * [0, n_points), no sort; with replacement past n_points), noise (B,n,3) = standard-normal float64; either may be NULL.  Pure
 * functions of (seed, cloud, position). */
int rl_batch_draw(const rl_cloud_job* jobs_dev, int B, int n, uint64_t seed, int64_t* indices, double* noise, void* stream);

/* ------------------------------------------------------------------------------------------
 * The head of the network, fused for the training step (round 5): Dropout(p) -> fc_end.3 (Conv2d 32 -> C without BatchNorm,
 * modules.py:525-530) -> un-permute (modules.py:608) -> loss + metric counts (losses.py:17-87, metrics.py:8-59) as ONE kernel,
 * and its whole backward - loss derivative, permute, input gradient of fc_end.3, Dropout backward - as ONE more, which also
 * leaves the BatchNorm-backward sums of fc_end.1 and fc_end.3's weight / bias gradient slabs.  The step's logits are never stored.
 *   X (B*N, 32): raw output of fc_end.1 in PERMUTED row order, with its folded BatchNorm (scale, shift, act, slope) applied on
 *     load; perm (N) int64: row r of cloud b is point perm[r] (its label: labels[b][perm[r]]); W (C, 32), bias (C): fc_end.3.
 *   Dropout: the mask of rl_dropout_fwd (Philox on (element quad, *drop_key | drop_seed), rows offset by drop_first_row);
 *     drop_p == 0: none.
 *   work: rl_loss_work_doubles(B*N, C) doubles; rl_head_fwd fills the slots and the totals record and writes `out`
 *     (1 + 4*C doubles) exactly as rl_loss_forward does; rl_head_bwd reads the totals record.
 *   rl_head_bwd: G (B*N, 32) = gradient w.r.t. fc_end.1's ACTIVATED output (input of rl_bn_bwd_apply / rl_bn_backward);
 *     bn_bwd_stats (optional): [rl_head_grid(rows)][2][32] doubles, the partials rl_bn_bwd_reduce would leave for fc_end.1
 *     (needs mean / invstd); slab: rl_head_grid(rows) partial records of C*32 + C floats (dW[C][32], then db[C]) for
 *     rl_wgrad_reduce_batch (nsplit = rl_head_grid(rows), N = C, K = 32).
 * rl_head_supported(C, K): 1 for K == 32 and 1 <= C <= 32 (up to 8 classes: a class' weights and sums in registers; 9 .. 32,
 * round 6: the three 32-row products of a trip on exact-fp32 MFMA with their operands in LDS); otherwise the caller runs the
 * separate entry points. */
typedef struct rl_head_desc {
    const float* X;
    const float* scale;
    const float* shift;
    int32_t act;
    float slope;
    const float* mean;
    const float* invstd;
    const float* W;
    const float* bias;
    const int64_t* perm;
    const int64_t* labels;
    int32_t B, N, C;
    int32_t loss_kind;
    float alpha, gamma;
    int32_t neglect_background;
    float drop_p;
    const int64_t* drop_key;
    uint64_t drop_seed;
    int64_t drop_first_row;
    double* work;
    float* G;
    double* bn_bwd_stats;
    float* slab;
    int64_t slab_floats;
    float grad_scale;
    int32_t reserved;
    /* optional (drop_p > 0): B*N uint32 - the forward stores each row's 32 keep bits, the backward reads them instead of running
     * the generator a second time (a Philox call is ~40 quarter-rate integer multiplies: the dominant cost of both kernels) */
    void* drop_mask;
    int64_t perm_bstride;           /* 0: one permutation for every cloud; N: cloud b reads perm + b * N (rl_band_sort) */
} rl_head_desc;
int rl_head_supported(int C, int K);
int rl_head_grid(int64_t rows);
int rl_head_fwd(const rl_head_desc* d, double* out, void* stream);
int rl_head_bwd(const rl_head_desc* d, void* stream);

/* Dropout (fc_end, modules.py:528) with a keep-mask drawn by the caller (uint8, 1 = keep):
 * x[i] = mask[i] ? x[i]*scale : 0, in place; the same call is its own backward.             */
int rl_scale_mask(float* x, const uint8_t* mask, float scale, int64_t count, void* stream);

/* Dropout (fc_end's nn.Dropout, modules.py:528) with a counter-based generator (Philox4x32-10): element e of the
 * (rows, C) tensor is kept iff philox(seed, key[0], e) >= p*2^32; kept values are scaled by 1/(1-p).  `key` is a
 * device scalar, so a captured graph draws a fresh mask per replay: rl_dropout_tick does counter[0] += 1 and copies
 * the new value to key_out[0]; forward and backward of one pass read the same key and regenerate the same mask (no
 * mask tensor).  rl_dropout_fwd also applies the producer's lazy BatchNorm + activation; dense rows (ld = C), C % 4 == 0.
 * `first_row`: index of row 0 in the tensor of the WHOLE batch - e = (first_row + r) * C + c - so that ranks holding
 * shards of one batch draw the slices of ONE mask (0 for a single process). */
int rl_dropout_tick(int64_t* counter, int64_t* key_out, void* stream);
int rl_dropout_fwd(const float* src, const float* scale, const float* shift, int act, float slope, float* dst,
                   int64_t rows, int64_t first_row, int C, const int64_t* key, uint64_t seed, float p, void* stream);
int rl_dropout_bwd(float* G, int64_t rows, int64_t first_row, int C, const int64_t* key, uint64_t seed, float p, void* stream);

/* UpSampler (modules.py:343-456) on channel-first features feat (B,F,N1) with neighbours
 * idx/d2 (B,N2,k) from rl_knn_i32: power 0 = nearest-neighbour interpolation (k = 1),
 * power 1 / 2 = inverse (squared) distance weighting, w = (1+1e-7)/(dist^power + 1e-7)
 * normalised over the k neighbours.  out (B,F,N2).                                          */
int rl_upsample_cf(const float* feat, const int32_t* idx, const float* d2, int B, int F, int N1,
                   int N2, int k, int power, float* out, void* stream);

/* logits (B,N,C) channel-last in permuted order  <->  (B,C,N) in original order
 * (modules.py:608-611): out[b][c][perm[i]] = in[b][i][c]; backward is the gather.           */
int rl_logits_unpermute(const float* in, const int64_t* perm, int B, int N, int C, float* out,
                        void* stream);
int rl_logits_permute_grad(const float* dout, const int64_t* perm, int B, int N, int C,
                           float* din, void* stream);
/* the same with one permutation PER CLOUD: cloud b reads perm + b * perm_bstride (0: the shared permutation above) */
int rl_logits_unpermute_b(const float* in, const int64_t* perm, int64_t perm_bstride, int B, int N, int C, float* out,
                          void* stream);
int rl_logits_permute_grad_b(const float* dout, const int64_t* perm, int64_t perm_bstride, int B, int N, int C,
                             float* din, void* stream);

/* A spatial order INSIDE the sampling bands of the forward's permutation (modules.py:571, 587-598).  The reference sub-samples
 * by prefixes of ONE random permutation, so only the band [edges[k], edges[k+1]) a point falls in matters (edges = 0, N/dec^L,
 * ..., N/dec, N); inside a band the order is free, and the kernels run faster when it follows space.  perm_out[b] (N) holds, band
 * by band, the entries of `perm` in that band ordered by (4096-cell Morton code of cloud b's point, position) - a stable counting
 * sort, deterministic.  rows: (B, N, row_stride) fp32 with x, y, z first.  workspace: rl_band_sort_workspace_bytes(...) bytes,
 * 256-byte aligned.  Four launches, no memset, no global atomics.                                                              */
int64_t rl_band_sort_workspace_bytes(int B, int N, const int* edges, int nbands);
int rl_band_sort(const float* rows, int64_t row_stride, const int64_t* perm, int B, int N, const int* edges, int nbands,
                 int64_t* perm_out, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss + metrics on logits (B,C,N) fp32 and labels (B,N) int64, C <= 32.
 * kind: 0 cross_entropy, 1 focal(gamma), 2 focal Tversky(alpha, gamma, neglect background)
 * (trainer.py:244-269 maps dice -> (0.5, 1), tversky -> (0.7, 1), focal_tversky -> (0.7, 4/3)).
 * rl_loss_forward fills out[0] = loss and the metric counts
 *   out[1 + 0*C + c] = #(pred==c & label==c), out[1 + 1*C + c] = #(label==c),
 *   out[1 + 2*C + c] = #(pred==c),            out[1 + 3*C + c] = sum_n softmax_c (diagnostic)
 * (accuracy / iou of metrics.py:8-59 are ratios of these counts), all as doubles, and keeps
 * what rl_loss_backward needs in `work` (rl_loss_work_doubles(B*N, C) doubles).
 * rl_loss_backward writes dlogits (B,C,N) = dloss/dlogits * grad_scale.                      */
int64_t rl_loss_work_doubles(int64_t points, int C);
int rl_loss_forward(const float* logits, const int64_t* labels, int B, int C, int N, int kind,
                    float alpha, float gamma, int neglect_background, double* work, double* out,
                    void* stream);
int rl_loss_backward(const float* logits, const int64_t* labels, int B, int C, int N, int kind,
                     float alpha, float gamma, int neglect_background, const double* work,
                     float grad_scale, float* dlogits, void* stream);

/* The same loss in two steps, for the data-parallel EQUIVALENCE mode (the dice ratio of the global
 * batch is not the mean of per-rank dice ratios): rl_loss_partials runs the pass over the logits and leaves this
 * rank's sums in the totals record, 5*C+1 doubles at work + rl_loss_totals_offset(C); the caller all-reduces that
 * record (SUM) over the ranks; rl_loss_from_totals forms the loss and the metric counts from it with the GLOBAL
 * point count; rl_loss_backward_global is rl_loss_backward normalised by the global point count.               */
int rl_loss_partials(const float* logits, const int64_t* labels, int B, int C, int N, int kind, float gamma,
                     double* work, void* stream);
int64_t rl_loss_totals_offset(int C);
int rl_loss_from_totals(int64_t points_total, int C, int kind, float alpha, float gamma, int neglect_background,
                        double* work, double* out, void* stream);
int rl_loss_backward_global(const float* logits, const int64_t* labels, int B, int C, int N, int kind, float alpha,
                            float gamma, int neglect_background, const double* work, float grad_scale,
                            int64_t points_total, float* dlogits, void* stream);

/* Softmax over the class axis of (B,C,N) logits -> confidences (model.py:137, 229).         */
int rl_softmax_cf(const float* logits, int B, int C, int N, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam over a flat parameter buffer (torch.optim.Adam defaults, trainer.py:78): step[0] is
 * incremented on the device, lr is read from device memory, grads are multiplied by
 * grad_scale first (1/world_size after a gradient all-reduce).                              */
int rl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 const float* lr, float beta1, float beta2, float eps, float grad_scale,
                 int64_t* step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RL_RANDLANET_H */
