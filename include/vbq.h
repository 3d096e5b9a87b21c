/*
 * vbq.h -- C-ABI of the MI355X-native VBQ hot path (libvbq_hip.so).
 *
 * Boundary rules
 *   - plain `extern "C"` functions, plain pointers and sizes; no torch / C++ types.
 *   - every pointer named `d_*` is DEVICE memory (borrowed; e.g. tensor.data_ptr() of a
 *     PyTorch-ROCm tensor), every pointer named `h_*` is HOST memory.
 *   - all work is enqueued on the caller's `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream) and returns without synchronising; nothing is
 *     allocated inside a call -- scratch comes from a caller-provided workspace.
 *   - return value: 0 = ok, negative = VBQ_ERR_*; vbq_last_error() gives the text of
 *     the calling thread's most recent failure.
 *   - no global state besides that thread-local error string (launch policies are per-call arguments; environment
 *     presets are read once and never written).
 *
 * The reference (mandt-lab/vbq) has no FFI of its own: the boundary it offers is the
 * Python call surface listed in SURVEY.md 8(b).  Each entry point below names the
 * reference code it replaces; INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Table layout ("level-major"), shared by every entry point that takes `d_table_lm`:
 *   float table[C][T], T = 2^(N+1)-1; the 2^n code points of bit length n occupy slots
 *   [2^n-1, 2^(n+1)-1) in increasing order.  This is exactly
 *   ChannelwisePriorCDFQuantizer.all_code_points (img-compression/quantizer.py:30-36)
 *   and the notebook's `codepoints` (word-embeddings/...ipynb:383-389).
 * Quantization index ("rank index"), produced / consumed as uint16:
 *   slot (n, i) has rank k = (2i+1) * 2^(N-n) in the merged sorted table; the index is
 *   k-1, i.e. the position in `code_points_by_channel` (quantizer.py:37) -- the value
 *   the reference calls `qidx` / `I` (quantizer.py:135,223) whenever the sorted table
 *   is strictly increasing.  Bit length of index q is N - ctz(q+1).
 * Element layout:
 *   VBQ_LAYOUT_BC  channel-last  [n_rows][n_ch]   (quantizer.py:90-91,196-197)
 *   VBQ_LAYOUT_CB  channel-major [n_ch][n_rows]   (the reference's own C x B view, :73)
 *   Outputs with a lambda axis are [n_lambda][...same layout as the input...].
 */
#ifndef VBQ_H_
#define VBQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBQ_ABI_VERSION 5

enum {
    VBQ_OK = 0,
    VBQ_ERR_INVALID_ARGUMENT = -1,
    VBQ_ERR_UNSUPPORTED = -2,
    VBQ_ERR_LAUNCH = -3,
    VBQ_ERR_WORKSPACE = -4
};

enum {
    VBQ_LAYOUT_BC = 0,
    VBQ_LAYOUT_CB = 1,
    /* vbq_quantize_f32 only: inputs channel-last [n_rows][n_ch] as the latents arrive, outputs channel-major
     * planes [n_lambda][n_ch][n_rows] -- the solve without the two input transposes (VBQ_MODE_F32; three or more
     * lambdas must lie in the fast kernel's range [1.9e-12, 1.8e19], otherwise VBQ_ERR_UNSUPPORTED; one or two
     * lambdas take the pruned descent, which accepts any value). */
    VBQ_LAYOUT_BC_TO_CB = 2
};

/* Tie-break / arithmetic of the solve. */
enum {
    /* TF-eager path of the image pipeline: every op rounded to f32, candidate order
     * [L_0..L_N, R_1..R_N], first maximum wins (utils.py:319-320,388-401;
     * quantizer.py:183). */
    VBQ_MODE_F32 = 0,
    /* NumPy backend exactly as written: f32 distortion, lambda*len and the score in
     * f64 (utils.py:388 leaves integer lengths uncast for backend=np). */
    VBQ_MODE_F64_SCORE = 1
};

int vbq_abi_version(void);
const char *vbq_last_error(void);

/* Launch policy: `reserved_workgroups` (vbq_quantize_rows_f32, vbq_level_counts_f32, vbq_build_entropy_models_f32).
 * For builds that overlap a collective with the kernels (the rank histogram's all-reduce of step i runs while step
 * i + 1 computes).  The solve kernels run as RESIDENT grids sized to every workgroup slot of the chip; a collective's kernel (RCCL:
 * a few dozen workgroups) takes some of those slots, and a resident workgroup that finds its slot taken starts only after another
 * one has finished all its iterations -- up to twice the kernel time.  With n > 0 the grids of THAT CALL are sized to (slots - n), or
 * launched as short-lived workgroups when the (workgroups x channels) grid shape would give up more than a tenth of the chip.
 * n = 0: every slot.  n < 0: the default, 0 unless the environment variable VBQ_RESERVED_WORKGROUPS presets it (read once).  A
 * per-call argument: the library keeps no mutable launch state, two builds with different policies may run side by side. */

/* The launch shape a solve call would take on the current device -- nothing is launched: h_grid (HOST, int64 [3]) = { workgroups
 * per channel, channels, 1 when every workgroup is resident from start to end (0: short-lived workgroups) } for
 * VBQ_GRID_K1 (the fused per-lambda kernel of vbq_quantize_rows_f32) / VBQ_GRID_K1T (the threshold kernel of vbq_level_counts_f32)
 * on n_rows rows per channel.  A pure function of its arguments and the device's CU count: what makes the launch policy testable. */
enum { VBQ_GRID_K1 = 0, VBQ_GRID_K1T = 1 };
int vbq_solve_grid(int32_t kernel, int64_t n_rows, int32_t n_ch, int32_t workgroups_per_cu, int32_t reserved_workgroups,
                   int64_t *h_grid);

/* Number of GPUs visible / name of device `dev` (for harness output only). */
int vbq_device_count(void);
int vbq_device_name(int dev, char *buf, size_t buflen);

/* ----------------------------------------------------------------------------------
 * K1  R-D solve.  Replaces, in one pass over (mu, sigma):
 *       ChannelwisePriorCDFQuantizer.get_all_N_bit_intervals   quantizer.py:65-80
 *       ChannelwisePriorCDFQuantizer.compress_batch_channel_latents  quantizer.py:156-188
 *       utils.curry_normal_logpdf(ignore_const=True)           utils.py:307-321
 *       utils.batch_quantize_indep_dims                        utils.py:363-423
 *       the qidx lookup                                        quantizer.py:135,223
 *     for all `n_lambda` trade-offs at once (the reference also shares the distortions
 *     across lambdas, utils.py:387,392).
 *
 *   d_mu, d_sigma   f32, n_rows*n_ch elements in `layout`; sigma > 0, all finite.
 *   d_table_lm      f32 [n_ch][T] level-major, non-decreasing in xi for every channel.
 *   d_level_len     f32 [n_lambda][n_ch][N+1] code length of bit-level n (the
 *                   "n + overhead" of quantizer.py:171-175), or NULL for the raw
 *                   lengths n (quantizer.py:167-169).  Lengths are non-negative; when some
 *                   lambda*len falls outside {0} U [2^-39, 2^70] the solve detects it on the
 *                   device and takes its literal (slower) scan, with the same answers.
 *   h_lambdas       HOST doubles [n_lambda]; VBQ_MODE_F32 rounds each to f32 first
 *                   (TF casts the Python scalar to the tensor dtype).
 *   d_out_idx       u16 [n_lambda][n_rows*n_ch] rank index of the winner.
 *   d_out_zhat      optional f32, same shape: the winning code point (Z_hat).
 *   d_out_bits      optional f32, same shape: its code length (num_bits).
 *   d_workspace     vbq_quantize_workspace_bytes() bytes of device scratch.
 *   Which kernel serves a call is an implementation detail (same answers): one or two lambdas with indices as the only
 *   output take a descent that stops as soon as no deeper bit level can win (literal comparisons only: any lambda, any
 *   lengths); sweeps of 16-32 lambdas with raw lengths are solved from ten thresholds per element; everything else by
 *   the fused per-lambda kernel.
 *   N               max_bits_per_coord; kernels are built for 4 <= N <= 12 (the reference uses 10,
 *                   post_process.py:117); N = 11, 12 for channel-major planes / one code book only
 *                   (16 channel tables no longer fit the LDS); the notebook solve and the coder stop at
 *                   10.  Other values return VBQ_ERR_UNSUPPORTED.
 * ---------------------------------------------------------------------------------- */
size_t vbq_quantize_workspace_bytes(int32_t n_ch, int32_t n_lambda, int32_t N);

/* The solve's precondition, checkable: d_bad[0] += #{mu not finite}, d_bad[1] += #{sigma not finite or <= 0} (u32[2],
 * device, zero it first).  The reference lets such values flow through tf.argmax (NaN scores, whatever index results);
 * here they are outside the contract of vbq_quantize_f32, and this is how a caller finds out beforehand. */
int vbq_check_inputs_f32(const float *d_mu, const float *d_sigma, int64_t n, uint32_t *d_bad, void *stream);

int vbq_quantize_f32(const float *d_mu, const float *d_sigma, int64_t n_rows, int32_t n_ch,
                     int32_t layout, const float *d_table_lm, const float *d_level_len,
                     const double *h_lambdas, int32_t n_lambda, int32_t N, int32_t mode,
                     uint16_t *d_out_idx, float *d_out_zhat, float *d_out_bits,
                     void *d_workspace, size_t workspace_bytes, void *stream);

/* ChannelwisePriorCDFQuantizer.get_all_N_bit_intervals (quantizer.py:65-80) as a result of its own (API compatibility; the
 * solve never materialises it): d_left / d_right f32 [n_ch][N+1][n_rows] = the left / right n-bit neighbours of every
 * z on every level, edge padding of the per-level grids included (:54-57,75-76).  d_z_cb: planes [n_ch][n_rows]. */
int vbq_n_bit_intervals_f32(const float *d_z_cb, int64_t n_rows, int32_t n_ch, const float *d_table_lm, int32_t N,
                            float *d_left, float *d_right, void *stream);

/* The same solve on rows [row_begin, row_end) of the full arrays (pointers, n_rows and output addressing are those
 * of the whole tensor): lets the caller cut one pass into chunks and run K2 on chunk j (another stream) while K1
 * works on chunk j + 1.  workgroups_per_cu = 0: default grid (the 4 workgroups per CU that fit, resident from start to
 * end, issue priority rotating over them); 1..5: a resident grid of that many workgroups per CU (at most the 4 that
 * fit; 3 leaves wave slots and LDS for a concurrently running K2).  reserved_workgroups: see "Launch policy" above
 * (vbq_quantize_f32 takes the default).  For planes (VBQ_LAYOUT_CB) the vector path wants row_begin % 8 == 0. */
int vbq_quantize_rows_f32(const float *d_mu, const float *d_sigma, int64_t n_rows, int32_t n_ch,
                          int32_t layout, const float *d_table_lm, const float *d_level_len,
                          const double *h_lambdas, int32_t n_lambda, int32_t N, int32_t mode,
                          uint16_t *d_out_idx, float *d_out_zhat, float *d_out_bits,
                          void *d_workspace, size_t workspace_bytes, int64_t row_begin, int64_t row_end,
                          int32_t workgroups_per_cu, int32_t reserved_workgroups, void *stream);

/* ----------------------------------------------------------------------------------
 * K1t / K1h  Solve + bit-length histogram in one kernel, nothing written per element (K1t: raw lengths at N = 10 without a
 *      per-lambda loop -- ten thresholds per element; K1h: the dense form for everything else).  Replaces the FIRST pass of
 *      ChannelwisePriorCDFQuantizer.build_entropy_models (quantizer.py:96-105): compress_batch_channel_latents
 *      followed by np.bincount(raw_num_bits[:, c], minlength=N+1) per lambda and channel -- which needs the bit
 *      level of every winner and nothing else.  Same arithmetic and tie rules as vbq_quantize_f32 (VBQ_MODE_F32).
 *   d_level_counts  int64 [n_lambda][n_ch][N+1], ADDED to (zero it first).
 *   layout          VBQ_LAYOUT_CB, VBQ_LAYOUT_BC_TO_CB, or any layout with n_ch == 1.
 *   reserved_workgroups  see "Launch policy" above (0: every slot).
 *   Returns VBQ_ERR_UNSUPPORTED for lambdas outside [1.9e-12, 1.8e19] (use vbq_quantize_f32 + vbq_histogram_u16).
 * ---------------------------------------------------------------------------------- */
int vbq_level_counts_f32(const float *d_mu, const float *d_sigma, int64_t n_rows, int32_t n_ch,
                         int32_t layout, const float *d_table_lm, const float *d_level_len,
                         const double *h_lambdas, int32_t n_lambda, int32_t N, int64_t *d_level_counts,
                         void *d_workspace, size_t workspace_bytes, int32_t reserved_workgroups, void *stream);

/* Code lengths from counts through a table: out[i] = (level_period ? i % level_period : 0) + lut[counts[i]].
 * Replaces the float32 arithmetic of quantizer.py:105-110 / 141-146 (counts + n -> / sum -> -log2) when the caller
 * can tabulate it: with B elements per (lambda, channel) row and add-n smoothing, every row's sum is the same
 * exact integer B + K n (< 2^24), so -log2(f32(k + n) / f32(B + K n)) is a function of the count k alone; the
 * caller builds lut[0..B] on the HOST with the reference's own NumPy operations (bit-identical by construction)
 * and the table lookup keeps the whole alternation on the device.  level_period = N + 1 adds the bit level n to
 * every entry (the "n + overhead" of quantizer.py:171-175, one f32 add).  Counts outside [0, lut_n) clamp.
 *   d_counts      int64 (counts_are_i32 = 0) or int32 (!= 0), n entries
 *   d_out_len     optional f32 [n]: level + lut[count];  d_out_model optional f32 [n]: lut[count] */
int vbq_code_lengths_from_counts(const void *d_counts, int32_t counts_are_i32, int64_t n,
                                 const float *d_lut, int64_t lut_n, int32_t level_period,
                                 float *d_out_len, float *d_out_model, void *stream);

/* The same arithmetic on the HOST, for the count tables that cannot be tabulated (2^24 samples per histogram row or more, fractional
 * smoothing): model[r][k] = -log2(f32(count[r][k] + n) / sum_k f32(count[r][k] + n)) exactly as quantizer.py:105-110 / 141-146
 * evaluates it in NumPy float32 (the row sum in NumPy's pairwise order); h_out_len[r][k] = k + model[r][k] when add_level != 0
 * (the "n + overhead" of :171-175).  h_* are HOST pointers (pinned or not); K <= 8192; either output may be NULL.
 *   log2_loop / log2_data   the float32 log2 as a NumPy ufunc inner loop (numpy/ufuncobject.h PyUFuncGenericFunction: args[0] in,
 *                           args[1] out, dimensions[0] elements, byte steps): NumPy's float32 log2 is not libm's, and the
 *                           reference's numbers are NumPy's -- the Python host hands over np.log2's own loop (vbq_amd/pipeline.py).
 *                           NULL: libm's log2f.
 * vbq_host_stage_run(stage) runs it on a filled-in descriptor: the function a caller gives hipLaunchHostFunc to put the step
 * BETWEEN two device stages of a stream without a synchronisation -- plain C on the runtime's callback thread, no interpreter
 * lock involved; `status` receives the return code, `runs` counts the executions. */
typedef void (*vbq_f32_loop)(char **args, const intptr_t *dimensions, const intptr_t *steps, void *data);
typedef struct vbq_host_stage {
    const void *h_counts;
    int32_t counts_are_i32;
    int32_t add_level;
    int64_t n_rows;
    int64_t K;
    float add_n_smoothing;
    int32_t status;
    vbq_f32_loop log2_loop;
    void *log2_data;
    float *h_out_model;
    float *h_out_len;
    int64_t runs;
} vbq_host_stage;
int vbq_host_neg_log2_freq_f32(const void *h_counts, int32_t counts_are_i32, int64_t n_rows, int64_t K, float add_n_smoothing,
                               int32_t add_level, vbq_f32_loop log2_loop, void *log2_data, float *h_out_model,
                               float *h_out_len);
void vbq_host_stage_run(void *stage);

/* ----------------------------------------------------------------------------------
 * K1c  Generic candidate solve.  Replaces utils.batch_quantize_indep_dims
 *      (img-compression/utils.py:363-423) for caller-built candidates, i.e. its
 *      "3D tensor of M x B x K" form (:367-370), and -- with n_lambda = 1 and the sorted
 *      table broadcast by the caller -- utils.quantize_indep_dims (:330-360).
 *   d_P     f32 [M][n_elems] candidate code points (n_elems = B*K flattened).
 *   d_len   f32 [M][n_elems], or [n_lambda][M][n_elems] when len_per_lambda != 0
 *           (the 4-D stack of utils.py:393-394).
 *   d_out_j   optional i32 [n_lambda][n_elems] winning candidate (first maximum);
 *   d_out_zhat / d_out_bits   optional f32 [n_lambda][n_elems]   (utils.py:414-415).
 * ---------------------------------------------------------------------------------- */
int vbq_argmax_candidates_f32(const float *d_P, const float *d_len, int32_t len_per_lambda,
                              const float *d_mu, const float *d_sigma, int64_t n_elems,
                              const double *h_lambdas, int32_t n_lambda, int32_t M, int32_t mode,
                              int32_t *d_out_j, float *d_out_zhat, float *d_out_bits, void *stream);

/* ----------------------------------------------------------------------------------
 * The reference's first, xi-space encoder (img-compression/utils.py:215-304), float64 as there.
 *   vbq_xi_intervals_f64  utils.get_all_N_bit_intervals (:215-260): d_left / d_right [N+1][K] = the two n-bit grid
 *                         points around xi[k] for every bit budget n (n = 0: both 0.5; outside the grid: both the rim).
 *   vbq_xi_select_f64     utils.encode_vectorized after its three callables (:291-303).  d_F, d_endpoints,
 *                         d_unsquashed are [2][N+1][K] (left first, np.stack order): picks the better endpoint per budget
 *                         (first maximum), subtracts lamb * n, picks the best budget (first maximum) and gathers
 *                         z_hat, xi_hat, num_bits (int64, as NumPy's argmax) and the per-coordinate objective f_z_hat
 *                         (the caller sums it: result['score'] = np.sum(f_z_hat)).
 * squash / unsquash / fun are the caller's functions and stay where the caller evaluates them.
 * ---------------------------------------------------------------------------------- */
int vbq_xi_intervals_f64(const double *d_xi, int64_t K, int32_t N, double *d_left, double *d_right, void *stream);
int vbq_xi_select_f64(const double *d_F, const double *d_endpoints, const double *d_unsquashed, int64_t K, int32_t N,
                      double lamb, double *d_z_hat, int64_t *d_num_bits, double *d_xi_hat, double *d_f_z_hat, void *stream);

/* ----------------------------------------------------------------------------------
 * K1n  Notebook solve.  Replaces compress_coordinates(means, stds, beta, bitlengths)
 *      (word-embeddings/compress-trained-word-embeddings.ipynb:429-443): f64 squared
 *      error against an f64 code book, penalty (2*beta)*sigma^2 rounded to f32 and
 *      multiplied by the integer length, first minimum in level-major order; one
 *      shared code book.  Output values are the code points rounded to f32 (the
 *      notebook stores into empty_like(means)).
 *   d_codebook_lm   f64 [T] level-major (ipynb:383-389), non-decreasing in xi.
 *   h_betas         HOST doubles [n_beta].
 *   d_out_idx       u16 [n_beta][n] rank index;  d_out_val optional f32 [n_beta][n].
 * ---------------------------------------------------------------------------------- */
int vbq_quantize_notebook_f64(const float *d_means, const float *d_stds, int64_t n,
                              const double *d_codebook_lm, const double *h_betas, int32_t n_beta,
                              int32_t N, uint16_t *d_out_idx, float *d_out_val, void *stream);

/* ----------------------------------------------------------------------------------
 * K2  Histogram pass.  Replaces the per-channel np.bincount of quantizer.py:104-105
 *     and :138-140 and the Counter of ipynb:453.  counts[l][c][q] += #{elements of
 *     channel c with index q under lambda l}.  Counts are ADDED to d_counts (zero it
 *     first for a fresh histogram); int64 so that 1e9-element shards cannot overflow.
 *     The bit-length histogram of :104 is the same array summed over the ranks of each
 *     level (level = N - ctz(q+1)).
 * ---------------------------------------------------------------------------------- */
int vbq_histogram_u16(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                      int32_t n_lambda, int32_t N, int64_t *d_counts, void *stream);

/* Same pass with 32-bit counters: half the bytes for the cross-GPU all-reduce.  The caller
 * guarantees that no bin can reach 2^31 -- i.e. the GLOBAL number of rows per channel (over
 * all ranks whose histograms will be summed into this buffer) is below 2^31. */
int vbq_histogram_u16_i32(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                          int32_t n_lambda, int32_t N, int32_t *d_counts, void *stream);

/* K2 on rows [row_begin, row_end) of the full index array (same addressing rules as vbq_quantize_rows_f32);
 * d_counts is int32 when counts_are_i32 != 0, else int64. */
int vbq_histogram_rows_u16(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                           int32_t n_lambda, int32_t N, void *d_counts, int32_t counts_are_i32,
                           int64_t row_begin, int64_t row_end, void *stream);

/* K2 as the LAST stage of the build: counts := histogram of the planes d_idx [n_lambda][n_ch][n_rows] (ASSIGNED, not added:
 * no zeroing beforehand) and, when d_models is given, models[l][c][q] = lut[counts[l][c][q]] -- the code-length table of
 * quantizer.py:141-146 in its tabulated form (see vbq_code_lengths_from_counts).  With at least 2048 (lambda, channel) rows
 * one workgroup owns each row, stores its bins and looks the lengths up in the same flush (no memset of the count array, no
 * atomics, no separate pass over it); smaller problems are composed from the plain entry points.  Single-GPU form: sharded
 * builds all-reduce the counts first and then call vbq_code_lengths_from_counts. */
int vbq_histogram_models_u16(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t n_lambda, int32_t N,
                             void *d_counts, int32_t counts_are_i32, const float *d_lut, int64_t lut_n,
                             float *d_models, void *stream);

/* Largest index of a u16 index array (d_max: u32, device, MAX-ed into; zero it first).  K2 and the gather are
 * memory-safe for any u16 input (indices >= T are counted in a wrapped bin / read the last table entry); a
 * caller holding indices that did not come from K1 (a file, a decoder) checks max < T with this first. */
int vbq_index_max_u16(const uint16_t *d_idx, int64_t n, uint32_t *d_max, void *stream);

/* ----------------------------------------------------------------------------------
 * K3  Moment pass.  Replaces empirical_std = sqrt(mean(mu^2)) (ipynb:373-374) and the
 *     per-channel mean/std a FactoredGaussianPrior needs (vae_models.py:32-35).
 *     d_out[c] = { sum x, sum x^2 } accumulated in f64 (ADDED to d_out).
 * ---------------------------------------------------------------------------------- */
int vbq_moments_f32(const float *d_x, int64_t n_rows, int32_t n_ch, int32_t layout,
                    double *d_out /* [n_ch][2] */, void *stream);

/* The notebook's moment in NumPy's own float32 order: d_out[0] = np.sum(x.ravel()**2) bit for bit (blocks of 8192
 * elements, pairwise inside a block, block results chained; see vbq_hist.hip).  Replaces the reduction inside
 * empirical_std = np.sqrt(np.mean(vecs_u.ravel()**2)) (ipynb:374); the caller divides by n and takes the root in
 * float32.  d_x 16-byte aligned; workspace: vbq_numpy_sum_sq_workspace_bytes(n) bytes of device memory. */
size_t vbq_numpy_sum_sq_workspace_bytes(int64_t n);
int vbq_numpy_sum_sq_f32(const float *d_x, int64_t n, float *d_out, void *d_workspace, size_t workspace_bytes,
                         void *stream);

/* np.sum of every row of a float32 batch [n_rows][n] in NumPy's own float32 order, bit for bit: d_out[r] = np.sum(x[r]) for a
 * contiguous x[r] of any shape with n elements.  Replaces the reductions of the evaluation loop, utils.py:547-552
 * (`nbits = np.sum(num_bits)`, `np.sum(num_bits_cl)` per image and lambda): with the per-lambda code lengths of
 * vbq_compress_latents_f32 still on the device, L floats cross PCIe instead of L x [B, C] arrays.  n_rows <= 65535;
 * workspace: vbq_numpy_row_sums_workspace_bytes(n_rows, n) bytes of device memory. */
size_t vbq_numpy_row_sums_workspace_bytes(int64_t n_rows, int64_t n);
int vbq_numpy_row_sums_f32(const float *d_x, int64_t n_rows, int64_t n, float *d_out, void *d_workspace,
                           size_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------------------------
 * Table lookup by rank index: out[l][e] = tab[(l)][c(e)][idx[l][e]].  Replaces
 *   tf.gather(entropy_model, I, batch_dims=1)            quantizer.py:226-228
 *   tf.gather(code_points_by_channel, qidx, batch_dims=1) quantizer.py:136
 *   d_tab   f32 [n_lambda][n_ch][T] when tab_per_lambda != 0, else [n_ch][T];
 *           indexed by rank (i.e. a SORTED table such as code_points_by_channel).
 *   layout / out_layout: element layout of d_idx and of d_out (they may differ: the
 *           Z_hat / num_bits handed back to the caller are channel-last, :228,237).
 * ---------------------------------------------------------------------------------- */
int vbq_gather_f32(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                   int32_t n_lambda, int32_t N, const float *d_tab, int32_t tab_per_lambda,
                   float *d_out, int32_t out_layout, void *stream);

/* ----------------------------------------------------------------------------------
 * Rate-distortion report of a solve: for every lambda l,
 *     d_out[2l]     += sum_e (z_e - mu_e)^2 / (2 sigma_e^2),   z_e = d_tab_sorted[c(e)][idx[l][e]]   (utils.py:319-320
 *                      without the sign and the constant: the distortion term the solve minimises)
 *     d_out[2l + 1] += sum_e d_rate[(l)][c(e)][idx[l][e]]                                          (the bits of
 *                      quantizer.py:226-228 summed as utils.py:547-549 sums them; skipped when d_rate is NULL)
 * in f64.  d_tab_sorted is the SORTED table (code_points_by_channel); d_rate is [n_lambda][n_ch][T] when
 * rate_per_lambda != 0 (entropy_models), else [n_ch][T].  d_out: f64 [n_lambda][2], zero it first.  A report -- the
 * Lagrangian of BASELINE's R-D gate is d_out[2l] + lambda_l * d_out[2l + 1] -- not an input of any kernel.
 * ---------------------------------------------------------------------------------- */
int vbq_rd_sums_u16(const float *d_mu, const float *d_sigma, const uint16_t *d_idx, int64_t n_rows, int32_t n_ch,
                    int32_t layout, int32_t n_lambda, int32_t N, const float *d_tab_sorted, const float *d_rate,
                    int32_t rate_per_lambda, double *d_out, void *stream);

/* ----------------------------------------------------------------------------------
 * Layout change between channel-last and channel-major planes: out[c][r] = in[r][c]
 * (f32, LDS-tiled).  Replaces the tf.transpose calls around the solve
 * (quantizer.py:73,163-164,223,228): the hot kernels want [C][B] planes so that a
 * workgroup works on ONE channel (one 8 KB table in LDS, wave-uniform penalties).
 * ---------------------------------------------------------------------------------- */
int vbq_transpose_f32(const float *d_in, int64_t n_rows, int64_t n_cols, float *d_out, void *stream);

/* The same for a stack of planes of 2- or 4-byte elements: out[b][c][r] = in[b][r][c] for b < n_batch.  With
 * elem_bytes = 2 it hands the u16 rank indices of the plane kernels ([n_lambda][n_ch][n_rows]) back in the caller's
 * channel-last layout ([n_lambda][n_rows][n_ch], the shape of compress_batch_channel_latents' results,
 * quantizer.py:186-188); with 4, Z_hat / num_bits planes.  n_batch <= 65535. */
int vbq_transpose_planes(const void *d_in, int64_t n_batch, int64_t n_rows, int64_t n_cols, int32_t elem_bytes,
                         void *d_out, void *stream);

/* ----------------------------------------------------------------------------------
 * The per-image call of the evaluation loop: ChannelwisePriorCDFQuantizer.compress_latents
 * (quantizer.py:190-240, called once per image by utils.py:542-554) as ONE C call =
 * three launches (planes, solve, fused lookups).  The three pieces are entry points of
 * their own as well.
 *
 *   vbq_prep_planes_f32     channel-last latents [n_rows][n_ch] -> channel-major planes [n_ch][n_rows]
 *                           (the tf.transpose of quantizer.py:163-164,223), means and spreads in one launch;
 *                           spread_kind says what d_spread_bc holds: VBQ_SPREAD_SIGMA the standard deviations,
 *                           VBQ_SPREAD_VARIANCE exp(logvar) (sigma = sqrt(.), IEEE), VBQ_SPREAD_LOGVAR the encoder's
 *                           log-variances themselves (sigma = sqrt(exp(.)): quantizer.py:197,202
 *                           `tf.exp(posterior_logvars) ** 0.5` inside the same launch).
 *   vbq_gather_latents_u16  ONE pass over rank indices in planes [n_lambda][n_ch][n_rows] writing, channel-last
 *                           [n_lambda][n_rows][n_ch] (any subset; NULL skips an output):
 *                             d_out_zhat      f32  d_table_sorted[c][q]                      quantizer.py:224-225
 *                             d_out_raw_bits  the bit length n(q) = N - ctz(q + 1) of the winner as int32 when
 *                                             d_level_len is NULL (quantizer.py:167-169), else f32
 *                                             d_level_len[l][c][n(q)] = "n + overhead"        quantizer.py:171-175
 *                             d_out_num_bits  f32  d_models[l][c][q] (entropy_models, by rank) quantizer.py:226-228
 *                             d_out_idx       u16  q itself (the index planes transposed)
 *   vbq_compress_latents_f32   prep -> vbq_quantize_f32 on planes (raw lengths when d_level_len is NULL) ->
 *                           gather.  d_spread_bc as for vbq_prep_planes_f32 (spread_kind).
 *                           Workspace: vbq_compress_latents_workspace_bytes() bytes of device memory, 256-byte
 *                           aligned (the planes, the index planes and the solve's own workspace live there).
 * ---------------------------------------------------------------------------------- */
enum { VBQ_SPREAD_SIGMA = 0, VBQ_SPREAD_VARIANCE = 1, VBQ_SPREAD_LOGVAR = 2 };
int vbq_prep_planes_f32(const float *d_means_bc, const float *d_spread_bc, int32_t spread_kind, int64_t n_rows,
                        int32_t n_ch, float *d_mu_cb, float *d_sigma_cb, void *stream);
int vbq_gather_latents_u16(const uint16_t *d_idx_planes, int64_t n_rows, int32_t n_ch, int32_t n_lambda, int32_t N,
                           const float *d_table_sorted, const float *d_level_len, const float *d_models,
                           float *d_out_zhat, void *d_out_raw_bits, float *d_out_num_bits, uint16_t *d_out_idx,
                           void *stream);
size_t vbq_compress_latents_workspace_bytes(int64_t n_rows, int32_t n_ch, int32_t n_lambda, int32_t N);
int vbq_compress_latents_f32(const float *d_means_bc, const float *d_spread_bc, int32_t spread_kind,
                             int64_t n_rows, int32_t n_ch, const float *d_table_lm, const float *d_table_sorted,
                             const float *d_level_len, const float *d_models, const double *h_lambdas,
                             int32_t n_lambda, int32_t N, float *d_out_zhat, void *d_out_raw_bits,
                             float *d_out_num_bits, void *d_workspace, size_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------------------------
 * The whole two-pass build of ChannelwisePriorCDFQuantizer.build_entropy_models (quantizer.py:82-150) on ONE GPU as one C call
 * = six stream-ordered launches, nothing waits for the host (sharded builds all-reduce between the pieces and call them one by
 * one: INTEGRATION.md):
 *     planes (d_spread_bc / spread_kind as for vbq_prep_planes_f32)                                  quantizer.py:87-92
 *     pass 1   vbq_level_counts_f32 with raw lengths -> d_level_counts [n_lambda][n_ch][N+1] (int64, assigned)   :96-105
 *     lengths  d_raw_models = lut_levels[count], d_level_len = n + lut_levels[count]   (f32 [n_lambda][n_ch][N+1])   :105-112, 171-175
 *     pass 2   vbq_quantize_f32 with d_level_len; vbq_histogram_models_u16 -> d_counts [n_lambda][n_ch][T] (int64, or int32 with
 *              counts_are_i32; assigned) and d_models = lut_ranks[count] (f32 [n_lambda][n_ch][T]; both NULL: no model table)   :119-146
 *   d_lut_levels / d_lut_ranks: the tabulated -log2 of the smoothed frequencies for every possible count 0 .. n_rows, built by the
 *   caller with the reference's own NumPy float32 operations (vbq_code_lengths_from_counts): n_lut >= n_rows + 1 entries each.
 *   Workspace: vbq_build_entropy_models_workspace_bytes() bytes of device memory, 256-byte aligned (planes, index planes, solve).
 *   reserved_workgroups: for both solve passes, see "Launch policy" above (0: every slot -- a single-GPU build has no neighbour).
 * ---------------------------------------------------------------------------------- */
size_t vbq_build_entropy_models_workspace_bytes(int64_t n_rows, int32_t n_ch, int32_t n_lambda, int32_t N);
int vbq_build_entropy_models_f32(const float *d_means_bc, const float *d_spread_bc, int32_t spread_kind, int64_t n_rows,
                                 int32_t n_ch, const float *d_table_lm, const double *h_lambdas, int32_t n_lambda,
                                 int32_t N, const float *d_lut_levels, int64_t n_lut_levels, const float *d_lut_ranks,
                                 int64_t n_lut_ranks, int64_t *d_level_counts, float *d_level_len, float *d_raw_models,
                                 void *d_counts, int32_t counts_are_i32, float *d_models, void *d_workspace,
                                 size_t workspace_bytes, int32_t reserved_workgroups, void *stream);

/* ----------------------------------------------------------------------------------
 * K4  BMSHJ2018 prior (learned_prior.py).  Parameters are the EFFECTIVE ones --
 *     softplus(matrix_i), bias_i, tanh(factor_i) -- packed per channel as
 *       [ M0(3x1) b0(3) f0(3) | M1(3x3) b1(3) f1(3) | M2(3x3) b2(3) f2(3) | M3(1x3) b3(1) ]
 *     = 43 floats (dims = (3,3,3), learned_prior.py:11,30-57), row-major matrices.
 *   vbq_bmshj_cdf_pdf_f32   cdf / analytic pdf / log(pdf+1e-10) at x
 *                           (learned_prior.py:70-148, 235-242, 244-334); any of the
 *                           three outputs may be NULL.  x is [n_rows][n_ch] channel-last.
 *   vbq_bmshj_icdf_step_f32 one masked bisection update of learned_prior.py:199-209 on
 *                           [n_rows][n_ch] brackets; writes mid points and accumulates
 *                           d_flags[0] = #(f(mid) != 0), d_flags[1] = bits of the minimum
 *                           bracket width (as u32 of a non-negative float) so that the
 *                           host can apply the global stopping rule of :210-211.
 * ---------------------------------------------------------------------------------- */
#define VBQ_BMSHJ_PARAMS_PER_CHANNEL 43

int vbq_bmshj_cdf_pdf_f32(const float *d_params, const float *d_x, int64_t n_rows, int32_t n_ch,
                          float *d_cdf, float *d_pdf, float *d_logpdf, void *stream);

int vbq_bmshj_icdf_step_f32(const float *d_params, const float *d_xi, int64_t n_rows, int32_t n_ch,
                            float *d_left, float *d_right, float *d_mid, uint32_t *d_flags,
                            void *stream);

/* n_steps of those updates enqueued at once, the stopping rule of learned_prior.py:210-211 applied ON THE DEVICE between them
 * (no host read per bisection step): step j runs only while the pair step j - 1 accumulated does not meet the rule (no
 * f(mid) != 0 left, or the minimum bracket width <= tol).  d_flags: u32 [n_steps + 1][2], written here; after a synchronisation
 * the caller reads it: the first j >= 1 whose pair meets the rule is the number of steps the reference's loop would have run
 * (d_mid holds that step's mid points); none does: call again with first = 0, which continues the chain from flags[n_steps]
 * of the previous call (copied by the caller to flags[0]). */
int vbq_bmshj_icdf_chain_f32(const float *d_params, const float *d_xi, int64_t n_rows, int32_t n_ch,
                             float *d_left, float *d_right, float *d_mid, uint32_t *d_flags, int32_t n_steps,
                             float tol, int32_t first, void *stream);

/* Fit step of the prior (learned_prior.py:402-430: loss = -mean(log(pdf + 1e-10)), full batch).
 * d_x_cb is f32 [n_ch][n_rows] (channel-major planes).  ADDS to d_out[c][0..42] the gradient of
 * sum_rows -log(pdf+1e-10) with respect to the 43 effective parameters of channel c (same
 * packing as above) and to d_out[c][43] that sum itself; f64.  The caller applies the
 * softplus / tanh chain rule, the 1/(n_rows*n_ch) of reduce_mean and the optimiser. */
int vbq_bmshj_nll_grad_f32(const float *d_params, const float *d_x_cb, int64_t n_rows, int32_t n_ch,
                           double *d_out /* [n_ch][44] */, void *stream);

/* ----------------------------------------------------------------------------------
 * Entropy coder for the rank indices.  The reference only ESTIMATES rates
 * as sum(-log2 freq) (quantizer.py:144,226-228); this turns them into bits.  Static-model rANS
 * (32-bit state, 16-bit renormalisation, 15 probability bits).
 *   d_idx    u16 [n_streams][n]   one stream per (lambda, channel): the [L][C][B] planes K1 writes
 *   d_freq   u16 [n_streams][T]   quantised frequencies, every entry >= 1, each row sums to 2^15
 *   seg      symbols per independently coded segment (one GPU thread each)
 *   d_words  u16 [n_streams][nseg][seg+2], nseg = ceil(n/seg): per segment the renormalisation
 *            words in emission order followed by the final state (low half, high half)
 *   d_sizes  u32 [n_streams][nseg] number of valid words of every segment
 * Symbols are coded last-to-first so that decoding runs first-to-last.
 * ---------------------------------------------------------------------------------- */
int vbq_rans_encode_u16(const uint16_t *d_idx, int64_t n_streams, int64_t n, int32_t N, int32_t seg,
                        const uint16_t *d_freq, uint16_t *d_words, uint32_t *d_sizes, void *stream);
/* Decoding treats words / sizes as UNTRUSTED: no read leaves a segment's seg + 2 words, every decoded index is
 * below T, and d_status (u32, device, may be NULL; OR-ed into, zero it first) reports what was wrong --
 * bit 0 a segment size outside [2, seg + 2], bit 1 a segment that ran out of words, bit 2 words left over or a
 * wrong final state (a damaged stream), bit 3 a frequency row that does not sum to 2^15.  Segments with bits 0 / 3
 * decode to zeros. */
int vbq_rans_decode_u16(const uint16_t *d_words, const uint32_t *d_sizes, int64_t n_streams, int64_t n,
                        int32_t N, int32_t seg, const uint16_t *d_freq, uint16_t *d_idx, uint32_t *d_status,
                        void *stream);

/* ----------------------------------------------------------------------------------
 * Packed counters for the histogram all-reduce: three 21-bit fields per int64 word.
 * An integer SUM all-reduce of the words adds the fields independently while every GLOBAL count is
 * below 2^21, at 2.67 instead of 4 bytes per bin on the wire.  n bins <-> (n + 2) / 3 words.
 * Guard: *d_overflow (u32, device, may be NULL; OR-ed into, zero it first) is set to 1 when a LOCAL count is
 * negative or >= 2^21 / n_ranks -- while it stays 0 on every rank the sum over n_ranks ranks cannot carry from one
 * field into the next.  The caller checks the flag before trusting the unpacked sums.
 * ---------------------------------------------------------------------------------- */
int vbq_pack_counts_3x21(const int32_t *d_counts, int64_t n, int64_t *d_words, int32_t n_ranks,
                         uint32_t *d_overflow, void *stream);
int vbq_unpack_counts_3x21(const int64_t *d_words, int64_t n, int32_t *d_counts, void *stream);

/* ----------------------------------------------------------------------------------
 * The one collective of the path: SUM all-reduce of a histogram over the ranks of a node, RCCL over xGMI.
 * Replaces nothing in the reference (it is single-process); it is what makes quantizer.py:104-105 / 138-140 global
 * when the rows are sharded over GPUs: every rank ends up with the counts of ALL rows, integer sums, so the
 * entropy models do not depend on the number of ranks.  One communicator per (process, GPU):
 *   vbq_comm_unique_id   rank 0 fills VBQ_COMM_ID_BYTES host bytes and hands them to the other ranks by any means
 *   vbq_comm_init        collective: every rank calls it with the same id (the current HIP device is the rank's GPU)
 *   vbq_allreduce_hist   in place on d_counts (int64, or int32 when counts_are_i32 != 0), asynchronous on `stream`;
 *                        works for the level histogram of K1t / K1h, the rank histogram of K2 and the packed words of
 *                        vbq_pack_counts_3x21 (int64) alike
 *   vbq_comm_destroy
 * RCCL is bound at run time (dlopen): VBQ_ERR_UNSUPPORTED when no librccl.so can be loaded.
 * ---------------------------------------------------------------------------------- */
#define VBQ_COMM_ID_BYTES 128
int vbq_comm_unique_id(void *h_id);
int vbq_comm_init(void **comm, int32_t n_ranks, const void *h_id, int32_t rank);
int vbq_allreduce_hist(void *comm, void *d_counts, int64_t n, int32_t counts_are_i32, void *stream);
int vbq_comm_destroy(void *comm);

/* ----------------------------------------------------------------------------------
 * Comparison quantizers (img-compression/quantizer.py:259-333).
 *   vbq_uniform_quantize_f32  I = clip(floor((x - min) / delta), 0, levels-1) in f32 (:280,295),
 *                             value = offset + delta * I (:297); I is returned as f32 like the
 *                             reference does.  UniformQuantizer.quantize and the index pass of .fit.
 *   vbq_nearest_code_f64      scipy.cluster.vq.vq(samples, code_points) for 1-D data (:329): f64
 *                             squared distance, first minimum in code-book order; value = the code.
 * Any output may be NULL; d_counts (int64 [levels] / [n_codes]) is ADDED to when given: the
 * np.bincount of :281 / :314.
 * ---------------------------------------------------------------------------------- */
int vbq_uniform_quantize_f32(const float *d_x, int64_t n, float min, float delta, float offset,
                             int32_t levels, float *d_out_index, float *d_out_value, int64_t *d_counts,
                             void *stream);
int vbq_nearest_code_f64(const float *d_x, int64_t n, const double *d_codes, int32_t n_codes,
                         int32_t *d_out_index, double *d_out_value, int64_t *d_counts, void *stream);

/* ----------------------------------------------------------------------------------
 * Analogy evaluator (compress-trained-word-embeddings.ipynb cell 14, ipynb:199-209):
 * prediction_ranks(emb) for Q questions (a, b, c, d) given as int32 [Q][4] word ids.
 *   normed = emb / (1e-8 + |emb|_2);  pred = normed[b] - normed[a] + normed[c];
 *   rank   = V - #{v : pred . normed[v] < pred . normed[d]} - 1        (int64 [Q])
 * One fused f32 MFMA GEMM [Q x K] x [K x V]; the score matrix never reaches HBM.  Scores are fma
 * chains over ascending k (the oracle restates exactly that); NumPy's BLAS order differs in the last
 * bits, which can move a rank only where another word's score is within rounding of the ground truth.
 * Workspace: vbq_analogy_ranks_workspace_bytes(V, K, Q) bytes of device memory.
 * ---------------------------------------------------------------------------------- */
size_t vbq_analogy_ranks_workspace_bytes(int64_t V, int32_t K, int64_t Q);
int vbq_analogy_ranks_f32(const float *d_emb, int64_t V, int32_t K, const int32_t *d_analogies, int64_t Q,
                          int64_t *d_out_ranks, void *d_workspace, size_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------------------------
 * Image metrics (img-compression/img_comparison_metrics.py:6-220), batches
 * [B][H][W][C], float64 arithmetic as in the reference.
 *   vbq_image_sqerr_u8   per-image integer sum of (a - b)^2 (mse :6-16 = sum / n, psnr :19-33 from it)
 *   vbq_u8_to_f64        widening copy (img.astype(float64), :122-123)
 *   vbq_unit_to_u8_f32   np.clip(np.round(X_hat * 255), 0, 255).astype(np.uint8) of the evaluation loop (utils.py:555) on the
 *                        device: reconstructions in [0, 1] -> the uint8 images the metrics compare (a quarter of the bytes, should
 *                        they go to the host for PIL's colour conversion)
 *   vbq_ssim_scale_f64   one scale of _SSIMForMultiScale (:84-157): 'valid' Gaussian window given as its
 *                        separable factor d_window[size] (size <= 11), constants c1 = (k1 max_val)^2,
 *                        c2 = (k2 max_val)^2; writes mean ssim and mean cs per image.  Direct sums where the
 *                        reference uses fftconvolve: agreement to ~1e-10, not bit-exact.
 *   vbq_downsample2_f64  scipy.ndimage.convolve(im, ones(1,2,2,1)/4, mode='reflect')[:, ::2, ::2, :] (:214-216)
 * ---------------------------------------------------------------------------------- */
int vbq_image_sqerr_u8(const uint8_t *d_img1, const uint8_t *d_img2, int64_t n_images, int64_t n_per_image,
                       int64_t *d_out_sum, void *stream);
int vbq_u8_to_f64(const uint8_t *d_in, int64_t n, double *d_out, void *stream);
int vbq_unit_to_u8_f32(const float *d_in, int64_t n, uint8_t *d_out, void *stream);
size_t vbq_ssim_scale_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t size);
int vbq_ssim_scale_f64(const double *d_im1, const double *d_im2, int32_t B, int32_t H, int32_t W, int32_t C,
                       const double *d_window, int32_t size, double c1, double c2, double *d_out_ssim,
                       double *d_out_cs, void *d_workspace, size_t workspace_bytes, void *stream);
int vbq_downsample2_f64(const double *d_in, int32_t B, int32_t H, int32_t W, int32_t C, double *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VBQ_H_ */
