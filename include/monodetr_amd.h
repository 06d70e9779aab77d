/*
 * include/monodetr_amd.h -- C ABI of libmonodetr_amd.so (MI355X / gfx950 hot-path kernels).
 *
 * This is the drop-in boundary for MonoDETR's native extension.  The reference binds its
 * CUDA op through pybind11/ATen:
 *     lib/models/monodetr/ops/src/vision.cpp:13-16         module `MultiScaleDeformableAttention`
 *     lib/models/monodetr/ops/src/ms_deform_attn.h:20-60   ms_deform_attn_forward / _backward
 *     lib/models/monodetr/ops/src/cuda/ms_deform_attn_cuda.cu:20-80, :83-153   shapes, asserts, alloc
 * Here the same two operations -- and the other hand-written kernels of the training step (dense
 * attention, Hungarian matching, the criterion's losses, column sums, AdamW) -- are plain `extern "C"`
 * functions taking raw device pointers and sizes.  No torch/ATen types cross this boundary; the host shim (monodetr_amd/_capi.py) owns
 * allocation, contiguity checks and the current stream, mirroring what the ATen wrapper did.
 *
 * Conventions (all entry points)
 *   - return 0 on success, a negative MDETR_E_* code on failure; mdetr_last_error() returns a
 *     thread-local message (the reference only printf'd launch errors, .cuh:948-952).
 *   - never allocate, free or synchronise; work is enqueued on `stream` (a hipStream_t, may be
 *     NULL for the default stream) of device ordinal `device`.
 *   - re-entrant: the compute entry points keep no mutable state of their own (forward runs on the main
 *     thread, backward on autograd worker threads); scratch they need is passed in as `workspace`.
 *     The only process-wide state is the opt-in profiling store (mdetr_profile_*), guarded by a mutex.
 *   - tensors are dense row-major ("contiguous"), 16-byte aligned base pointers.
 */
#ifndef MONODETR_AMD_H
#define MONODETR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDETR_ABI_VERSION 12

/* element types of the floating-point tensors */
#define MDETR_F32 0
#define MDETR_F64 1
#define MDETR_BF16 2   /* attention, column-sum and AdamW entry points */

#define MDETR_OK            0
#define MDETR_E_ARG        -1   /* bad size / null pointer / unsupported dtype */
#define MDETR_E_HIP        -2   /* HIP runtime or launch error (message has the hipError string) */
#define MDETR_E_ALIGN      -3   /* pointer not 16-byte aligned */

int mdetr_abi_version(void);
const char *mdetr_last_error(void);

/*
 * Multi-scale deformable attention, forward.
 * Replaces ms_deform_attn_cuda_forward (ms_deform_attn_cuda.cu:20-80) and
 * ms_deformable_im2col_gpu_kernel (ms_deform_im2col_cuda.cuh:237-299).
 *
 *   value           [B, S, M, D]            dtype
 *   spatial_shapes  [L, 2] (H_l, W_l)       int64, DEVICE memory (as in the reference, .cu:67)
 *   level_start     [L]                     int64, DEVICE memory
 *   loc             [B, Lq, M, L, P, 2]     dtype, (x, y) normalised to [0,1]
 *   attn            [B, Lq, M, L, P]        dtype
 *   out             [B, Lq, M*D]            dtype, fully overwritten (no pre-zeroing needed)
 *
 * The reference's im2col_step batch chunking (.cu:50-52) is a launch-size artefact and has no
 * numerical effect; one launch covers the whole batch here.
 */
int mdetr_msda_forward(int dtype,
                       const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                       const void *loc, const void *attn, void *out,
                       int B, int S, int M, int D, int L, int Lq, int P,
                       int device, void *stream);

/*
 * Multi-scale deformable attention, backward.
 * Replaces ms_deform_attn_cuda_backward (ms_deform_attn_cuda.cu:83-153) and the col2im kernel
 * family (ms_deform_im2col_cuda.cuh:301-920).
 *
 *   grad_out    [B, Lq, M*D]
 *   grad_value  [B, S, M, D]           } all three are zero-filled by this call on `stream`
 *   grad_loc    [B, Lq, M, L, P, 2]    } before accumulation (the reference's at::zeros_like,
 *   grad_attn   [B, Lq, M, L, P]       } .cu:121-123); the caller may pass uninitialised memory.
 *
 * grad_value is accumulated with floating-point atomics: summation order (hence the last bits)
 * is not deterministic, exactly as in the reference (.cuh:125-152).
 */
int mdetr_msda_backward(int dtype,
                        const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                        const void *loc, const void *attn, const void *grad_out,
                        void *grad_value, void *grad_loc, void *grad_attn,
                        int B, int S, int M, int D, int L, int Lq, int P,
                        int device, void *stream);

/*
 * Host (CPU) twins of mdetr_msda_forward / mdetr_msda_backward: EVERY pointer is a host pointer, including
 * spatial_shapes and level_start; no device, no stream; outputs are written completely.  They replace the reference's
 * CPU entry points ms_deform_attn_cpu_forward / _backward (ops/src/cpu/ms_deform_attn_cpu.cpp:17-40), which only raise
 * "Not implement on cpu" -- so that BASELINE configs[0] (the yaml on a CPU, one training iteration) can run.  Same
 * arithmetic as the CUDA kernels (ms_deform_im2col_cuda.cuh:33-159, 237-403); one (image, head) per task on a pool of
 * host threads; deterministic.  Explicit entry points, not a fallback: the device entry points never route here.
 */
int mdetr_msda_forward_cpu(int dtype,
                           const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                           const void *loc, const void *attn, void *out,
                           int B, int S, int M, int D, int L, int Lq, int P);
int mdetr_msda_backward_cpu(int dtype,
                            const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                            const void *loc, const void *attn, const void *grad_out,
                            void *grad_value, void *grad_loc, void *grad_attn,
                            int B, int S, int M, int D, int L, int Lq, int P);

/*
 * Backward with the tile-privatised grad_value path.  Same contract as mdetr_msda_backward plus
 *   spatial_shapes_host / level_start_host   HOST copies of the two int64 arrays (the launch
 *                                            geometry is planned on the host)
 *   workspace / workspace_bytes             device scratch of at least
 *                                            mdetr_msda_backward_workspace_bytes(...) bytes, 256-B aligned
 * For fp32, D == 32 and Lq == S (self-attention over the pyramid: query q sits at pixel q) grad_value
 * is accumulated per query tile in LDS as 64-bit fixed point (exact, order-independent) and reduced
 * without atomics; samples that leave the tile's window, and every other geometry, take the
 * fp32-atomic path of mdetr_msda_backward.  Passing NULL host arrays or workspace == NULL is allowed
 * and equivalent to mdetr_msda_backward.  mdetr_msda_backward_workspace_bytes returns 0 when the
 * geometry does not qualify.
 */
int mdetr_msda_backward_ex(int dtype,
                           const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                           const void *loc, const void *attn, const void *grad_out,
                           void *grad_value, void *grad_loc, void *grad_attn,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           const int64_t *spatial_shapes_host, const int64_t *level_start_host,
                           void *workspace, int64_t workspace_bytes,
                           int device, void *stream);

int64_t mdetr_msda_backward_workspace_bytes(int dtype, const int64_t *spatial_shapes_host,
                                            const int64_t *level_start_host,
                                            int B, int S, int M, int D, int L, int Lq, int P);

/*
 * Gather indices the kernels use, for bit-exact index parity checks.
 *   idx [B, Lq, M, L, P, 4] int32 = (in_window, h_low, w_low, corner_mask); zeros when the sample
 *   is outside the window (.cuh:288).  corner_mask bit0..3 = validity of (low,low) (low,high)
 *   (high,low) (high,high) per .cuh:56,62,68,74.
 */
int mdetr_msda_indices(int dtype, const int64_t *spatial_shapes, const void *loc, int32_t *idx,
                       int B, int M, int L, int Lq, int P, int device, void *stream);

/*
 * Which kernel variant a given problem dispatches to: 1 = gfx950 fast path (f32, D == 32),
 * 0 = generic path (any D, f32/f64).  Pure host logic, no GPU needed.
 */
int mdetr_msda_variant(int dtype, int M, int D, int L, int P);

/*
 * Fused dense multi-head attention, head_dim = 32:  out = dropout(softmax(q k^T * scale + mask)) v.
 * Replaces the QK^T / softmax / dropout / PV core of the three torch.nn.MultiheadAttention calls on
 * the hot path (depthaware_transformer.py:456-459, :496; depth_predictor/transformer.py:59), which
 * the reference evaluates with need_weights=True, i.e. with the full [B*H, Lq, Lk] score matrix
 * (and its head average) in memory.  Scores stay in registers here (bf16 MFMA, fp32 accumulate).
 *
 *   dtype            MDETR_F32 or MDETR_BF16: element type of q, k, v, out (and of d_out, dq, dk, dv)
 *   q                [B, Lq, H*32]  batch stride q_bs, row stride q_rs (elements), innermost contiguous
 *   k, v             [B, Lk, H*32]  likewise (so slices of a packed in-projection output work)
 *   key_padding_mask [B, Lk] uint8, nonzero = ignore the key; may be NULL
 *   out              [B, Lq, H*32]  contiguous
 *   lse              [B, H, Lq] fp32: log2-domain log-sum-exp of the scaled scores (for backward)
 *   dropout_p, seed  dropout on the probabilities; the mask is a stateless hash of (seed, b, h, q, k),
 *                    regenerated identically by the backward
 *   seed_dev         optional DEVICE pointer to a uint64 added to `seed` when the kernel runs: lets a
 *                    captured hipGraph draw a fresh mask on every replay (the host scalar is frozen at
 *                    capture time); NULL = use `seed` alone
 * Backward additionally takes out, d_out [B, Lq, H*32] contiguous and a scratch dsum [B, H, Lq] fp32,
 * and writes dq [B, Lq, H*32], dk, dv [B, Lk, H*32] (contiguous, fully overwritten).
 */
int mdetr_attn_forward(int dtype, const void *q, const void *k, const void *v, const uint8_t *key_padding_mask,
                       void *out, float *lse, int B, int H, int Lq, int Lk,
                       int64_t q_bs, int64_t k_bs, int64_t v_bs, int q_rs, int k_rs, int v_rs,
                       float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev, int device, void *stream);

int mdetr_attn_backward(int dtype, const void *q, const void *k, const void *v, const uint8_t *key_padding_mask,
                        const void *out, const void *d_out, const float *lse, float *dsum,
                        void *dq, void *dk, void *dv, int B, int H, int Lq, int Lk,
                        int64_t q_bs, int64_t k_bs, int64_t v_bs, int q_rs, int k_rs, int v_rs,
                        float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev, int device, void *stream);

/*
 * MonoDETR's matched-pair losses for all decoder levels at once (lib/models/monodetr/monodetr.py:320-458:
 * loss_labels / loss_cardinality / loss_3dcenter / loss_boxes / loss_depths / loss_dims / loss_angles, called per
 * level by SetCriterion.forward :490-532), one launch forward and one backward.
 *   predictions, level-stacked fp32:  logits [L,B,Q,C], boxes [L,B,Q,6] (cx,cy,l,r,t,b), dims [L,B,Q,3],
 *                                     depths [L,B,Q,2] (mean, log-variance), angles [L,B,Q,24]
 *   assign  int32 [L,B,G,K]   matched query of target slot k in query group g, or -1 (mdetr_lsa_forward's output)
 *   ground truth padded to K slots per image:  labels, heading_bin int64 [B,K]; boxes3d [B,K,6]; depth [B,K];
 *                                     size3d [B,K,3]; heading_res [B,K]; valid uint8 [B,K]; num int32 [B]
 *   num_boxes / num_boxes_dev         the normaliser (:503-508), host value or device scalar (non-NULL wins)
 *   out     fp32 [9, L]: loss_ce, loss_center, loss_bbox, loss_giou, loss_depth, loss_dim, loss_angle,
 *           class_error, cardinality_error per level;   comp fp32 [L]: the dimension-aware L1's detached factor
 *   workspace  >= mdetr_pair_losses_workspace_bytes(L, B) bytes, ZERO before the first call; every forward
 *           call leaves it zero again
 * backward: grad_out fp32 [9, L] (rows 0-6 are read) -> gradients of the five prediction tensors, every element
 * written (zeros for unmatched queries).  Q % G == 0, C <= 8, K <= 64.
 */
int64_t mdetr_pair_losses_workspace_bytes(int L, int B);
int mdetr_pair_losses_forward(const float *logits, const float *boxes, const float *dims, const float *depths,
                              const float *angles, const int32_t *assign, const int64_t *labels, const float *boxes3d,
                              const float *depth, const float *size3d, const int64_t *heading_bin,
                              const float *heading_res, const uint8_t *valid, const int32_t *num,
                              int L, int B, int Q, int C, int G, int K, float focal_alpha,
                              float num_boxes, const float *num_boxes_dev, float *out, float *comp, void *workspace,
                              int device, void *stream);
int mdetr_pair_losses_backward(const float *logits, const float *boxes, const float *dims, const float *depths,
                               const float *angles, const int32_t *assign, const int64_t *labels, const float *boxes3d,
                               const float *depth, const float *size3d, const int64_t *heading_bin,
                               const float *heading_res, const uint8_t *valid,
                               int L, int B, int Q, int C, int G, int K, float focal_alpha,
                               float num_boxes, const float *num_boxes_dev, const float *grad_out, const float *comp,
                               float *g_logits, float *g_boxes, float *g_dims, float *g_depths, float *g_angles,
                               int device, void *stream);

/*
 * Mixed-precision variant of mdetr_msda_forward / mdetr_msda_backward_ex for a bf16 model body: `value`, `out` and
 * `grad_out` are bf16 (64-byte rows), sampling locations, attention weights and all three gradient outputs are
 * fp32, accumulation is fp32 throughout.  The reference's operator is fp32/fp64 only (ms_deform_attn_cuda.cu:64) and
 * a bf16 model has to widen `value` (42 -> 84 MB at the encoder shape) and narrow `out` around every call; this
 * variant reads and writes the model's tensors directly.  D = 32, L = P = 4 only.  Workspace and host geometry as
 * for mdetr_msda_backward_ex (mdetr_msda_backward_workspace_bytes with MDETR_F32).
 */
int mdetr_msda_forward_bf16(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                            const float *loc, const float *attn, void *out,
                            int B, int S, int M, int D, int L, int Lq, int P, int device, void *stream);
int mdetr_msda_backward_bf16(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                             const float *loc, const float *attn, const void *grad_out,
                             float *grad_value, float *grad_loc, float *grad_attn,
                             int B, int S, int M, int D, int L, int Lq, int P,
                             const int64_t *spatial_shapes_host, const int64_t *level_start_host,
                             void *workspace, int64_t workspace_bytes, int device, void *stream);

/*
 * The arithmetic MSDeformAttn.forward performs between its projections and the sampling operator
 * (lib/models/monodetr/ops/modules/ms_deform_attn.py:139-160), one launch each way, evaluated in fp32:
 *     attn_weight  = softmax over the L*P samples of a head (logits [B, Lq, M, L*P])
 *     sampling_loc = ref + offsets / (W_l, H_l)                          (R == 2 reference components)
 *                  = ref[:2] + offsets / P * (ref[2]+ref[3], ref[4]+ref[5]) * 0.5      (R == 6)
 *   io_dtype   MDETR_F32 or MDETR_BF16: dtype of offsets [B,Lq,M,L,P,2], logits and of grad_offsets / grad_logits
 *   ref_dtype  MDETR_F32 or MDETR_BF16: dtype of ref (a bf16 model body keeps its reference points in fp32)
 *   ref        [B, Lq, L, R] with element strides (ref_sb, ref_sq, ref_sl) -- 0 for a broadcast axis -- and a
 *              contiguous last dimension;  spatial_shapes int64 [L, 2] (H, W) on the device
 *   outputs    sampling_loc fp32 [B,Lq,M,L,P,2], attn_weight fp32 [B,Lq,M,L,P]: what mdetr_msda_forward consumes
 * backward: grad_loc / grad_attn (fp32, from mdetr_msda_backward) -> grad_offsets, grad_logits (io_dtype) and, if
 * grad_ref != NULL, grad_ref fp32 [B, Lq, L, R] dense (zero-filled here, then accumulated over the M heads).
 */
int mdetr_msda_prologue_forward(int io_dtype, int ref_dtype, const void *offsets, const void *logits, const void *ref,
                                const int64_t *spatial_shapes, float *sampling_loc, float *attn_weight,
                                int B, int Lq, int M, int L, int P, int R, int64_t ref_sb, int64_t ref_sq, int64_t ref_sl,
                                int device, void *stream);
int mdetr_msda_prologue_backward(int io_dtype, int ref_dtype, const void *offsets, const void *ref, const int64_t *spatial_shapes,
                                 const float *attn_weight, const float *grad_loc, const float *grad_attn,
                                 void *grad_offsets, void *grad_logits, float *grad_ref,
                                 int B, int Lq, int M, int L, int P, int R, int64_t ref_sb, int64_t ref_sq, int64_t ref_sl,
                                 int device, void *stream);
/*
 * The same pair on the output of ONE projection (the reference's two nn.Linear of the query, ms_deform_attn.py:138-139, with
 * their weights stacked: sampling_offsets rows, then attention_weights rows): packed [B, Lq, M L P 3] of io_dtype holds per
 * query the M L P 2 offsets followed by the M L P logits; grad_packed has the same layout.  L = P = 4 and 16-byte aligned
 * pointers only (MDETR_E_ARG / MDETR_E_HIP otherwise).
 */
int mdetr_msda_prologue_forward_packed(int io_dtype, int ref_dtype, const void *packed, const void *ref, const int64_t *spatial_shapes,
                                       float *sampling_loc, float *attn_weight, int B, int Lq, int M, int L, int P, int R,
                                       int64_t ref_sb, int64_t ref_sq, int64_t ref_sl, int device, void *stream);
int mdetr_msda_prologue_backward_packed(int io_dtype, int ref_dtype, const void *packed, const void *ref, const int64_t *spatial_shapes,
                                        const float *attn_weight, const float *grad_loc, const float *grad_attn, void *grad_packed,
                                        float *grad_ref, int B, int Lq, int M, int L, int P, int R,
                                        int64_t ref_sb, int64_t ref_sq, int64_t ref_sl, int device, void *stream);

/*
 * mdetr_lsa_forward with the matching cost evaluated inside the kernel (matcher.py:55-84) instead of read from a
 * cost matrix:  cost(q, t) = w_bbox L1(l,r,t,b) + w_center L1(cx,cy) + w_class (pos - neg focal cost at label_t)
 * - w_giou GIoU, from level-stacked predictions logits fp32 [layers, images, groups*n, num_classes], boxes fp32
 * [layers, images, groups*n, 6] and the padded ground truth labels int64 [images, kmax], boxes3d fp32
 * [images, kmax, 6].  Same output as mdetr_lsa_forward.
 */
int mdetr_lsa_forward_fused(const float *logits, const float *boxes, const int64_t *labels, const float *boxes3d,
                            const int32_t *num_targets, int32_t *assign, int layers, int images, int groups, int n,
                            int kmax, int num_classes, float w_class, float w_bbox, float w_center, float w_giou,
                            float focal_alpha, int device, void *stream);

/*
 * MonoDETR's depth-map loss (lib/models/monodetr/depth_predictor/ddn_loss/ddn_loss.py:103-127 with
 * balancer.py:26-50 and focalloss.py:58-129; called from monodetr.py:443-458) in one launch each way.
 *   logits   fp32 [B, C, H, W] depth-bin logits (C = bins + 1) in any dense layout: element strides sb, sc, sh, sw
 *   boxes    fp32 [B, K, 4] ground-truth 2D boxes (cx, cy, w, h) normalised to the image, padded to K slots
 *   depth    fp32 [B, K] object centre depths;   valid uint8 [B, K]
 *   out      fp32 [1]: sum over pixels of weight * focal / (B H W), weight = fg_weight inside any valid box
 *            (nearest object gives the target bin, linear-increasing discretisation between depth_min and
 *            depth_max), bg_weight elsewhere (target = the extra last bin)
 *   workspace  16 bytes, ZERO before the first call; every forward call leaves it zero again
 * backward: grad_out fp32 [1] -> grad_logits, same layout as logits, every element written.
 */
int mdetr_ddn_loss_forward(const float *logits, const float *boxes, const float *depth, const uint8_t *valid,
                           int B, int C, int H, int W, int K, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                           float alpha, float fg_weight, float bg_weight, float depth_min, float depth_max,
                           float *out, void *workspace, int device, void *stream);
int mdetr_ddn_loss_backward(const float *logits, const float *boxes, const float *depth, const uint8_t *valid,
                            int B, int C, int H, int W, int K, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                            float alpha, float fg_weight, float bg_weight, float depth_min, float depth_max,
                            const float *grad_out, float *grad_logits, int device, void *stream);

/*
 * One step of the reference's AdamW (lib/helpers/optimizer_helper.py:69-129 -- not torch.optim.AdamW: the
 * decoupled decay is scaled by the bias-corrected step, eps is added outside the bias correction) over a
 * FLAT parameter group, in one launch:
 *     m <- b1 m + (1-b1) g ;  v <- b2 v + (1-b2) g^2 ;  p <- p - step (wd p + m / (sqrt(v) + eps))
 *   param_dtype   MDETR_F32: param == master, fp32 gradient;  MDETR_BF16: bf16 param and gradient, the
 *                 update runs on the fp32 `master` copy and the rounded result is written to `param`
 *   n_no_decay    elements [0, n_no_decay) use weight decay 0 (the reference gives 'bias' parameters
 *                 wd = 0, :8-16); the caller lays the group out with those first
 *   step_size     lr * sqrt(1 - b2^t) / (1 - b1^t), computed by the caller;  step_size_dev, if not
 *                 NULL, is a device fp32 scalar used instead (graph-replay-safe step count)
 * All five arrays hold n elements and are 16-byte aligned.  exp_avg / exp_avg_sq / master are fp32.
 */
int mdetr_adamw_step(int param_dtype, void *param, float *master, const void *grad, float *exp_avg, float *exp_avg_sq,
                     int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps, float weight_decay,
                     float step_size, const float *step_size_dev, int device, void *stream);
/* The same update with the bias-corrected step size computed ON THE DEVICE: step = lr * sqrt(1 - beta2^t) / (1 - beta1^t) with the
 * step count t read from `step_count_dev` (a double the caller has already advanced) and the learning rate from `lr_dev` (a
 * double, or NULL: the host value `lr`) -- a captured optimizer step then needs no scalar launches of its own. */
int mdetr_adamw_step_counted(int param_dtype, void *param, float *master, const void *grad, float *exp_avg, float *exp_avg_sq,
                             int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps, float weight_decay,
                             const double *step_count_dev, float lr, const double *lr_dev, int device, void *stream);
/* The same update with the gradients WHERE AUTOGRAD LEFT THEM -- no flat gradient buffer, no copy into one (the reference's loop reads
 * p.grad tensor by tensor, lib/helpers/optimizer_helper.py:96-129; so does this).  grad_ptrs[i] (host array) = base address of the
 * gradient of tensor i, same dtype and memory layout as its parameter; flat_offsets[i] / nbytes[i] (device, int64) = byte offset of
 * tensor i in `param` (x 2 in the fp32 arrays of a bf16 group: they are indexed by ELEMENT) and its size in bytes; a workgroup
 * takes up to chunk_bytes of one tensor: block_tensor / block_start (device) name tensor and byte start per workgroup,
 * tensor_block_begin (host, ntensors + 1) the first workgroup of every tensor -- the tables of mdetr_gather_flat.  Step size:
 * step_count_dev (as mdetr_adamw_step_counted) if not NULL, else step_size_dev if not NULL, else step_size. */
int mdetr_adamw_step_gathered(int param_dtype, void *param, float *master, const void *const *grad_ptrs, int ntensors,
                              const int *tensor_block_begin, const int64_t *flat_offsets, const int64_t *nbytes, const int *block_tensor,
                              const int64_t *block_start, int chunk_bytes, float *exp_avg, float *exp_avg_sq, int64_t n_no_decay,
                              float beta1, float beta2, float eps, float weight_decay, float step_size, const float *step_size_dev,
                              const double *step_count_dev, float lr, const double *lr_dev, int device, void *stream);

/*
 * y[T, N] = dropout(relu(a[T, K] op(w) + bias + res)) in bf16 on the matrix cores, every part of the tail optional: the general
 * token-wise product of the training iteration with its elementwise tail inside (csrc/tgemm.hip).  Replaces, per call site, the
 * library GEMM + separate passes of the reference: torchvision Bottleneck.conv1 / conv3 (+ identity + ReLU) / downsample behind
 * lib/models/monodetr/backbone.py:93-106, the input projections monodetr.py:77-99, nn.Linear in ops/modules/ms_deform_attn.py:94-102
 * and depthaware_transformer.py:328-353 (linear1 + ReLU + Dropout), and their input gradients.
 *   a      bf16 [T, K], row stride lda (elements; % 8 == 0), 16-byte aligned
 *   w      bf16, 16-byte aligned: [N, K] with row stride ldw (y = a w^T: the forward of a linear layer / 1x1 convolution), or with
 *          MDETR_TGEMM_NN [K, N] with row stride ldw (y = a w: the input gradient dX = dY W with the parameter as it lies in memory)
 *   bias   [N] bf16 (fp32 with MDETR_TGEMM_BIAS_F32) or NULL
 *   res    bf16 [T, N] with row stride ldr, added in fp32 before the ReLU, or NULL.  res == y is allowed: y += a op(w) (the gradient
 *          arriving over a residual connection summed inside the input-gradient product)
 *   y      bf16 [T, N] (fp32 with MDETR_TGEMM_OUT_F32), row stride ldy (% 8 == 0; % 4 for fp32), 16-byte aligned
 *   dropout_p > 0: inverted dropout behind the ReLU, kept iff ln_hash(seed [+ *seed_dev], t * N + n) >= p * 2^32 (csrc/add_ln_math.h: the
 *          decision mdetr_bias_act_forward makes for the same element, so mdetr_bias_act_backward serves as its backward)
 * K % 8 == 0, N % 8 == 0; fp32 accumulation, ONE rounding at the store.
 */
#define MDETR_TGEMM_RELU 1
#define MDETR_TGEMM_NN 2
#define MDETR_TGEMM_BIAS_F32 4
#define MDETR_TGEMM_OUT_F32 8
int mdetr_tgemm(const void *a, const void *w, const void *bias, const void *res, void *y, int64_t T, int N, int K,
                int64_t lda, int64_t ldw, int64_t ldr, int64_t ldy, int flags, float dropout_p, uint64_t seed,
                const void *seed_dev, int device, void *stream);
/*
 * The input gradient of a layer whose INPUT is the output of a ReLU, with that ReLU's backward inside (ABI 8):
 *     y[T, N] = mask[T, N] <= 0 ? 0 : a[T, K] w[K, N] + res[T, N]       (bf16; w as with MDETR_TGEMM_NN; res may be NULL)
 * mask is the layer's forward input (= the ReLU's output: positive exactly where the ReLU passed its input).  Replaces the
 * input-gradient GEMM followed by autograd's threshold_backward of the preceding nn.ReLU (torchvision Bottleneck.forward behind
 * lib/models/monodetr/backbone.py:93-106: relu -> conv1 of the next block, relu -> conv3).  Applying the mask here is safe whatever
 * else consumes the ReLU's output: mask (g1 + g2) = mask g1 + mask g2, and mask mask = mask.  Alignment rules as mdetr_tgemm; ldm = the
 * mask's row stride.
 */
int mdetr_tgemm_masked(const void *a, const void *w, const void *res, const void *mask, void *y, int64_t T, int N, int K,
                       int64_t lda, int64_t ldw, int64_t ldr, int64_t ldm, int64_t ldy, int device, void *stream);

/*
 * Grouped fp32 products on the f32-input matrix instruction (ABI 9; csrc/sgemm.hip): up to MDETR_SGEMM_MAX_PROBLEMS independent
 * products per launch, each
 *     C[m, n] = mask( relu( sum_t A_t op(B_t) + bias + res ) )
 * in exact fp32 arithmetic (v_mfma_f32_32x32x2_f32: the k-ordered fmaf chain -- no reduced-precision path exists on gfx950 and none
 * is used).  Replaces, for the prediction heads of lib/models/monodetr/monodetr.py:222-262 (class_embed, bbox_embed, dim_embed_3d,
 * angle_embed, depth_embed: nn.Linear / MLP on the [B x queries, 256] decoder output of every level -> ATen addmm / mm -> rocBLAS /
 * hipBLASLt sgemm, one launch per layer and direction) the 20 library launches per decoder level by 7 grouped ones.
 *   mode MDETR_SGEMM_NT   C[M, N] = A[M, K] B[N, K]^T     forward of a linear layer (B = the weight as it lies in memory)
 *        MDETR_SGEMM_NN   C[M, N] = A[M, K] B[K, N]       input gradient dX = dY W
 *        MDETR_SGEMM_TN   C[M, N] = A[K, M]^T B[K, N]     weight gradient dW = dY^T X (k = the token count); colsum[m] = sum_k A[k, m]
 *                                                          (the bias gradient) rides along; one term, no bias / relu / mask / res
 *   term   A_t, B_t with their row strides (elements), contraction length k, element types MDETR_F32 or MDETR_BF16 (widened on load).
 *          Several terms = one product over a contraction axis that is split over several tensors (dX of a packed first layer:
 *          the per-head gradient slices against the per-head weights -- no concatenated copy of either exists).
 *   bias   [N] fp32 or NULL;  relu_cols: columns [0, relu_cols) pass through max(., 0) (0 = none);
 *   mask   fp32 [M, N] (row stride ldm) or NULL: the result is zeroed where mask <= 0 (the backward of the ReLU whose OUTPUT is mask);
 *   res    [M, N] (row stride ldr, res_dtype) or NULL: added before relu / mask (a gradient arriving over another path);
 *   c      [M, N] (row stride ldc), c_dtype MDETR_F32 or MDETR_BF16 (one rounding at the store).
 * Any shape: ragged edges are guarded element-wise; rows whose stride and base allow it are read with 16-byte loads.  Deterministic
 * (fixed summation order, no atomics).
 */
#define MDETR_SGEMM_NT 0
#define MDETR_SGEMM_NN 1
#define MDETR_SGEMM_TN 2
#define MDETR_SGEMM_MAX_PROBLEMS 10
#define MDETR_SGEMM_MAX_TERMS 5
typedef struct {
    const void *a, *b;
    int64_t lda, ldb;
    int32_t k, a_dtype, b_dtype, pad_;
} mdetr_sgemm_term;
typedef struct {
    mdetr_sgemm_term term[MDETR_SGEMM_MAX_TERMS];
    void *c;
    const float *bias;
    float *colsum;
    const float *mask;
    const void *res;
    int64_t ldc, ldm, ldr;
    int32_t nterm, m, n, relu_cols, c_dtype, res_dtype;
} mdetr_sgemm_problem;
/* workspace: a TN group cuts its contraction into parts (one workgroup per tile and part: 100 tiles alone would leave most of the chip
 * idle) whose partial results a second kernel sums in order; mdetr_sgemm_workspace_bytes gives the scratch that needs (0 for NT / NN). */
int64_t mdetr_sgemm_workspace_bytes(int mode, const mdetr_sgemm_problem *problems, int nprob);
int mdetr_sgemm_grouped(int mode, const mdetr_sgemm_problem *problems, int nprob, void *workspace, int64_t workspace_bytes, int device, void *stream);

/*
 * Training image path of the input pipeline on the device: what
 * lib/datasets/kitti/kitti_dataset.py:127-163 does per sample on a CPU worker -- photometric distortion
 * (lib/datasets/kitti/pd.py:376-398) with its uint8 cast, horizontal flip, PIL's affine warp with bilinear
 * resampling to out_w x out_h, / 255, (x - mean) / std, HWC -> CHW -- computed for a whole batch in one launch
 * from the decoded RGB8 images, bit-identical to that chain.  The random decisions are made on the host by the
 * caller (in the reference's order, monodetr_amd/datasets/kitti_dataset.py) and arrive in the descriptors.
 *   pixels   device, uint8: the batch's decoded images back to back, each H x W x 3 (RGB, rows of 3 W bytes)
 *   images   device, MdetrKittiImage[n_images] (8-byte aligned)
 *   out      device, [n_images, 3, out_h, out_w] (channels_last = 0) or [n_images, out_h, out_w, 3] (= 1: the
 *            memory of a torch channels_last tensor), MDETR_F32 or MDETR_BF16 (bf16 = the float32 result rounded to
 *            nearest even), 16-byte aligned, out_w % 4 == 0
 *   mean, std   host pointers to 3 floats each (kitti_dataset.py:79-80)
 */
#define MDETR_KITTI_FLIP 1u            /* Image.FLIP_LEFT_RIGHT before the warp */
#define MDETR_KITTI_DISTORT 2u         /* the photometric chain runs (even with every stage idle it re-quantises) */
#define MDETR_KITTI_CONTRAST_FIRST 4u  /* contrast before the HSV stages (pd.py:392-395), else after */
#define MDETR_KITTI_BRIGHTNESS 8u
#define MDETR_KITTI_CONTRAST 16u
#define MDETR_KITTI_SATURATION 32u
#define MDETR_KITTI_HUE 64u
typedef struct MdetrKittiImage {
    int64_t pixel_offset;   /* byte offset of this image in `pixels` */
    int32_t width, height;
    uint32_t flags;         /* MDETR_KITTI_* */
    uint32_t perm;          /* lighting noise: output channel c = distorted channel (perm >> 2c) & 3; identity 0x24 */
    float brightness, contrast, saturation, hue;
    double inv[6];          /* PIL's AFFINE data: source = inv * (output pixel centre, 1) */
} MdetrKittiImage;
int mdetr_kitti_preprocess(const uint8_t *pixels, const MdetrKittiImage *images, int n_images, void *out,
                           int out_dtype, int out_h, int out_w, int channels_last, const float *mean, const float *std,
                           int device, void *stream);

/*
 * KITTI evaluation: rotated-box overlaps on the device, the serial statistics on the host.
 *
 * mdetr_rotate_iou_eval replaces rotate_iou_gpu_eval (lib/datasets/kitti/kitti_eval_python/rotate_iou.py:262-330);
 * mdetr_box3d_overlap_eval additionally folds in d3_box_overlap_kernel (eval.py:195-228).  Both are SEGMENTED: frame f
 * owns boxes [box_start[f], box_start[f+1]) and query boxes [qbox_start[f], qbox_start[f+1]) and its row-major
 * [n_f, k_f] block of `out` starts at out_start[f] (out_start[n_frames] = total number of pairs), so one launch covers
 * a whole split and only within-frame pairs are computed (the reference evaluates all pairs of 50-frame parts).
 *   boxes / qboxes   device; float [., 5] (x, y, dx, dy, angle)  resp.  double [., 7] (x, y, z, l, h, w, ry; camera)
 *   *_start          device int64 [n_frames + 1]
 *   criterion        -1 IoU, 0 / 1 intersection over the query box's / the box's area (volume), 2 intersection
 *   out[n, k]        overlap of box n with query box k, evaluated as devRotateIoUEval(query box, box) (:293-296)
 */
int mdetr_rotate_iou_eval(const float *boxes, const float *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                          const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, float *out,
                          int device, void *stream);
int mdetr_box3d_overlap_eval(const double *boxes, const double *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                             const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, double *out,
                             int device, void *stream);
/*
 * HOST function (no device work): precision / recall accumulation of one (class, difficulty, metric, overlap) cell of
 * eval_class (eval.py:563-620) -- compute_statistics_jit over all frames, get_thresholds, fused_compute_statistics.
 *   overlaps     frame f: [nd_f, ng_f] row-major at ov_start[f] (detections x ground truths)
 *   gt_datas     [sum ng, 5] (bbox, alpha);  dt_datas [sum nd, 6] (bbox, alpha, score);  *_start int64 [n_frames + 1]
 *   ignored_*    int64 per box: 0 counts, 1 neutral, -1 other class (clean_data, eval.py:30-82)
 *   dontcares    [sum ndc, 4]
 *   pr           out, [n_thresholds, 4] = tp, fp, fn, orientation similarity per recall threshold
 *   thresholds   out, [n_thresholds] (<= max_thresholds, 41 in the reference)
 */
int mdetr_kitti_pr_curve(const double *overlaps, const int64_t *ov_start, const double *gt_datas, const double *dt_datas,
                         const int64_t *gt_start, const int64_t *dt_start, const int64_t *ignored_gt,
                         const int64_t *ignored_det, const double *dontcares, const int64_t *dc_start, int n_frames,
                         int metric, double min_overlap, int64_t num_valid_gt, int compute_aos, int max_thresholds,
                         double *pr, double *thresholds, int *n_thresholds);

/*
 * y = LayerNorm(a + dropout(b)) over the last dimension, one pass each way -- the ten residual sites of the model
 * (lib/models/monodetr/depthaware_transformer.py:331-353, :431-435, :456-510; depth_predictor/transformer.py:57-65),
 * which the reference runs as nn.Dropout, an addition and nn.LayerNorm.
 *   io_dtype   MDETR_F32 or MDETR_BF16: a, b, y, s (and dy, da, db); stats fp32; fp32 arithmetic
 *   param_dtype  MDETR_F32 or MDETR_BF16: gamma, beta (a bf16 model body keeps them in bf16)
 *   a, b       [rows, cols] contiguous, cols in {128, 256, 512}, 16-byte aligned; b may be NULL (plain LayerNorm)
 *   s          out, a + dropout(b) as the backward reads it (may be NULL when b is NULL: the backward then takes a)
 *   stats      out, fp32 [rows, 2] = (mean, 1 / sqrt(var + eps))
 *   dropout    element (row, col) of b is kept iff a hash of (seed [+ *seed_dev], row * cols + col) >= p * 2^32 and
 *              scaled by 1 / (1 - p); no mask is stored, the backward recomputes it (same seed)
 * backward: da = d(loss)/da, db = d(loss)/db (NULL when b was NULL), and per-workgroup partial sums
 *   partial    fp32 [mdetr_add_layernorm_partial_rows(rows), 2 * cols]: columns [0, cols) sum dy * xhat (d gamma),
 *              [cols, 2 cols) sum dy (d beta); the caller finishes them with mdetr_column_sum
 */
int mdetr_add_layernorm_forward(int io_dtype, int param_dtype, const void *a, const void *b, const void *gamma, const void *beta, void *y,
                                void *s, float *stats, int64_t rows, int cols, float eps, float dropout_p, uint64_t seed,
                                const uint64_t *seed_dev, int device, void *stream);
int64_t mdetr_add_layernorm_partial_rows(int64_t rows);
int mdetr_add_layernorm_backward(int io_dtype, int param_dtype, const void *dy, const void *s, const void *gamma, const float *stats,
                                 void *da, void *db, float *partial, int64_t rows, int cols, float dropout_p,
                                 uint64_t seed, const uint64_t *seed_dev, int device, void *stream);

/*
 * Column sums of a tall row-major matrix, accumulated in fp32: out[j] = sum_i x[i * ld + j].
 * Not an entry point of the reference's extension: it is the bias gradient of the model's token-wise
 * linear layers (db = sum over the 81 600 tokens of dY; torch's autograd computes it with a generic
 * reduction -- reference call sites: every nn.Linear applied to the flattened feature pyramid, e.g.
 * lib/models/monodetr/ops/modules/ms_deform_attn.py:80-83 and depthaware_transformer.py:331-333).
 *   dtype      MDETR_F32 (cols % 4 == 0) or MDETR_BF16 (cols % 8 == 0); x and ld * sizeof(T) 16-byte aligned
 *   out        fp32 [cols] (overwritten);  workspace  >= mdetr_column_sum_workspace_bytes(rows, cols)
 * Deterministic (two passes in a fixed order, no atomics).
 */
int64_t mdetr_column_sum_workspace_bytes(int64_t rows, int cols);
int mdetr_column_sum(int dtype, const void *x, float *out, void *workspace, int64_t workspace_bytes,
                     int64_t rows, int cols, int64_t ld, int device, void *stream);
/* The same reduction with the result written in `out_dtype` (MDETR_F32, or MDETR_BF16: the fp32 sum rounded once) -- the
 * gradient of a bf16 parameter then needs no separate cast launch. */
/*
 * What follows the prediction heads, per (decoder level, image, query), in one launch each way (ABI 9; csrc/head_tail.hip).  Replaces
 * the elementwise chain of lib/models/monodetr/monodetr.py:226-253 -- inverse_sigmoid of the references, box = sigmoid(delta +
 * reference), the geometric depth f H3d / h2d with its clamp, F.grid_sample of the weighted depth map at the (detached) 3-D centre, the
 * three-way depth average -- and its backward nodes (~100 framework launches per iteration on [3, 8, 550, <= 6] tensors).
 *   delta [L,B,Q,6], init_ref [B,Q,nd0] (nd0 = 2 or 6: refines the first nd0 components of level 0), inter_refs [L-1,B,Q,6] (levels >= 1),
 *   size3d [L,B,Q,3] (component 0 = height), depth_reg [L,B,Q,2], depth_map [B,H,W], img_h [B], focal [B]   -- all fp32, dense
 *   -> coord [L,B,Q,6], depth_ave [L,B,Q,2]
 * backward: g_coord / g_depth (either may be NULL = zeros) -> g_delta, g_init_ref, g_size3d, g_depth_reg, g_map (NULL = not wanted).
 * The map's gradient is computed per cell in a fixed order (no atomics).  mdetr_box_refine is the decoder's reference update between
 * layers (depthaware_transformer.py:602-613, no gradient): out [rows,6] = sigmoid(delta [rows,6] + inverse_sigmoid(ref [rows,nd]) on
 * the first nd components).
 */
int mdetr_box_refine(const float *delta, const float *ref, float *out, int64_t rows, int nd, int device, void *stream);
int mdetr_head_tail_forward(const float *delta, const float *init_ref, const float *inter_refs, const float *size3d, const float *depth_reg,
                            const float *depth_map, const float *img_h, const float *focal, float *coord, float *depth_ave,
                            int L, int B, int Q, int nd0, int H, int W, int device, void *stream);
int mdetr_head_tail_backward(const float *init_ref, const float *size3d, const float *depth_reg, const float *img_h, const float *focal,
                             const float *coord, const float *g_coord, const float *g_depth, float *g_delta, float *g_init_ref,
                             float *g_size3d, float *g_depth_reg, float *g_map, int L, int B, int Q, int nd0, int H, int W, int device, void *stream);

/* Several chunk sums in ONE launch (ABI 9): out_i[cols_i] = sum over the chunks of part_i[chunks_i][cols_i] (fp32 partials of the
 * split weight-gradient kernels), added in chunk order, rounded once to out_dtype -- what mdetr_column_sum_to computes for one such
 * matrix, for any number of them (159 launches of the round-5 step become 4).  cols % 4 == 0, part 16-byte aligned, out 16-byte (fp32)
 * / 8-byte (bf16) aligned. */
typedef struct {
    const float *part;
    void *out;
    int64_t cols;
    int32_t chunks, out_dtype;
} mdetr_chunk_job;
int mdetr_chunk_sums(const mdetr_chunk_job *jobs, int njobs, int device, void *stream);
int mdetr_column_sum_to(int dtype, const void *x, void *out, int out_dtype, void *workspace, int64_t workspace_bytes,
                        int64_t rows, int cols, int64_t ld, int device, void *stream);

/*
 * 3x3 / stride 1 / pad 1 convolution of a channels-last bf16 activation on the matrix cores (implicit GEMM, im2col in LDS),
 * with a per-channel shift (the folded frozen BatchNorm) and an optional ReLU in the epilogue -- Bottleneck.conv2 -> bn2 -> relu
 * of the ResNet body (lib/models/monodetr/backbone.py:100-102 -> torchvision resnet), which the reference hands to cuDNN.
 *   x      bf16 [B, H, W, C], C % 64 == 0 (a channels_last [B, C, H, W] tensor), 16-byte aligned
 *   w      bf16 [N, 3, 3, C], N % 32 == 0 (a channels_last [N, C, 3, 3] weight), 16-byte aligned
 *   shift  fp32 [N] or NULL;  relu: bit 0 = ReLU in the epilogue, bit 1 = taps read MIRRORED (w[n, 2 - t, 2 - s, :])
 *   y      bf16 [B, H, W, N], 8-byte aligned;  y = act(conv(x, w) + shift), fp32 accumulation, one rounding
 * The input gradient of the same convolution is this entry point on grad_y with the weight's channel axes swapped,
 * w'[c, t, s, n] = w[n, t, s, c], the taps mirrored by bit 1 of `relu`, shift = NULL.
 */
int mdetr_conv3x3_forward(const void *x, const void *w, const float *shift, void *y, int B, int H, int W, int C, int N,
                          int relu, int device, void *stream);
/* The launch geometry mdetr_conv3x3_forward takes for a shape (host only, no device work): 100 x WC + 10 x GC + NB -- a wave's 32 pixels
 * are a (32 / WC) x WC block, GC blocks side by side and 4 / GC one below the other form the workgroup's tile, NB x 32 output channels per
 * workgroup.  Chosen for the fewest ROUNDS of workgroups on 256 CUs, then the cheapest round, then the fewest blocks on empty columns. */
int mdetr_conv3x3_plan(int B, int H, int W, int N);
/* The same with a MASK (bf16 [B, H, W, N], 8-byte aligned): y = mask <= 0 ? 0 : act(conv(x, w) + shift).  The input gradient of
 * Bottleneck.conv2 with autograd's threshold_backward of the ReLU behind conv1 -> bn1 (torchvision Bottleneck.forward, `out =
 * self.relu(out)` between conv1 and conv2) applied where the gradient leaves the chip: mask = conv2's input. */
int mdetr_conv3x3_masked(const void *x, const void *w, const float *shift, const void *mask, void *y, int B, int H, int W, int C, int N,
                         int relu, int device, void *stream);

/*
 * The strided convolutions of the ResNet body, the fourth pyramid level and the depth predictor, and their input gradients, as
 * one implicit-GEMM kernel over a rectangular set of taps (csrc/conv_taps.hip) -- torchvision Bottleneck.conv2 / downsample of
 * each stage's first block behind lib/models/monodetr/backbone.py:93-106, monodetr.py:87-92, depth_predictor.py:29-31, which
 * the reference hands to cuDNN:
 *   y[b, r, c, n] = act(shift[n] + sum_{a < TR, e < TS, k < C} x[b, SI r + a - PT, SI c + e - PL, k]
 *                                                              * w[n, ta0 + a ta_step, te0 + e te_step, k]),  r < OH, c < OW
 *   x      bf16 [B, H, W, C], C % 64 == 0, 16-byte aligned; zero outside the map
 *   w      bf16; element (n, tap row, tap column, k) at w + n w_sn + row w_sa + column w_se + k, 16-byte aligned rows
 *   y      bf16; pixel (b, r, c) at y + y_off + b y_sb + r y_sr + c y_sc (elements, multiples of 4), N % 32 == 0 channels
 *   dims   23 values: B H W C  OH OW N  SI TR TS PT PL  ta0 ta_step te0 te_step  y_off y_sb y_sr y_sc  w_sn w_sa w_se
 * Supported tap sets: SI = 2 with 3x3 or 1x1 taps (forward of the stride-2 convolutions); SI = 1 with 1x1, 1x2, 2x1, 2x2 taps
 * (the four pixel-parity classes of a stride-2 convolution's input gradient, written at y_sr = 2 W C, y_sc = 2 C).
 */
int mdetr_conv_taps(const void *x, const void *w, const float *shift, void *y, const int64_t *dims, int relu, int device, void *stream);

/*
 * The 3x3 / stride-2 / pad-1 case of mdetr_conv_taps with the CONTRACTION split over ksplit groups of input channels -- for layers
 * with few output pixels against a long contraction (the fourth pyramid level, lib/models/monodetr/monodetr.py:87-92: 2048 channels
 * x 9 taps into 8 x 6 x 20 pixels).  Writes fp32 partials [ksplit][B][OH][OW][N] (split 0 carries the shift, no activation); the
 * caller adds the splits in a fixed order (mdetr_column_sum_to over the split axis) and rounds once.  dims as mdetr_conv_taps (the
 * output strides dims[16..19] are ignored); C % (32 ksplit) == 0.
 */
int mdetr_conv_taps_split(const void *x, const void *w, const float *shift, float *partial, int64_t partial_floats, const int64_t *dims,
                          int ksplit, int device, void *stream);

/*
 * Input gradient of a stride-2 convolution (3x3 / pad 1 or 1x1 / pad 0), the four pixel-parity classes in ONE launch; every element
 * of dx is written exactly once (the odd pixels of a 1x1 convolution's gradient as zeros):
 *   dx[b, i, j, c] = sum_{t, s, n : (i + P - t) and (j + P - s) even} dy[b, (i + P - t) / 2, (j + P - s) / 2, n] * wt[c, t, s, n]
 *   dy  bf16 [B, OH, OW, N], N % 64 == 0;  wt  bf16 [C, K, K, N] (the weight with its channel axes swapped, taps not mirrored);
 *   dx  bf16 [B, H, W, C], C % 32 == 0;  OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1
 */
int mdetr_conv_dgrad_s2(const void *dy, const void *wt, void *dx, int B, int OH, int OW, int N, int H, int W, int C, int K, int device, void *stream);

/*
 * The ResNet stem (torchvision ResNet.conv1 -> bn1 -> relu behind lib/models/monodetr/backbone.py:93-106; frozen: forward only):
 *   y[b, r, c, n] = relu(shift[n] + sum_{t, e < 7, ch < 3} x[b, 2r + t - 3, 2c + e - 3, ch] * w[n, ch, t, e])      (csrc/conv_stem.hip)
 *   x         bf16 [B, H, W, 3] (a channels_last [B, 3, H, W] image batch)
 *   w_packed  bf16 [64][176]: element t * 24 + e * 3 + ch of row n = w[n, ch, t, e], every other element zero; 16-byte aligned
 *   shift     fp32 [64] or NULL
 *   y         bf16 [B, (H - 1) / 2 + 1, (W - 1) / 2 + 1, 64], 8-byte aligned
 */
int mdetr_conv_stem(const void *x, const void *w_packed, const float *shift, void *y, int B, int H, int W, int device, void *stream);

/*
 * Weight gradient of the 3x3 (stride 1 / 2, pad 1) and 1x1 (stride 2) convolutions on the matrix cores (csrc/conv_wgrad.hip):
 *   dW[n, t, e, c] = sum_{b, r, q} dy[b, r, q, n] * x[b, SI r + t - P, SI q + e - P, c]        (P = 1 for K = 3, 0 for K = 1)
 * -- autograd of the convolutions named above, cuDNN's in the reference.  Split-K over pixel tiles: the kernel writes
 * mdetr_conv_wgrad_chunks(...) partial gradients, fp32 [chunks][N][K][K][C], into `partial`; the caller adds them (in a fixed
 * order: mdetr_column_sum_to over the chunk axis) and rounds once.
 *   x   bf16 [B, H, W, C], C % 64 == 0;  dy  bf16 [B, OH, OW, N], N % 32 == 0; both 16-byte aligned
 */
int mdetr_conv_wgrad_chunks(int B, int H, int W, int C, int OH, int OW, int N, int K, int SI);
int mdetr_conv_wgrad(const void *x, const void *dy, float *partial, int64_t partial_floats, int B, int H, int W, int C, int OH, int OW, int N,
                     int K, int SI, int device, void *stream);

/*
 * Weight AND bias gradient of a token-wise linear layer y = x W^T + b over T token rows (autograd of the nn.Linear / 1x1
 * convolutions of lib/models/monodetr/depthaware_transformer.py:328-331, 409-423, ops/modules/ms_deform_attn.py:94-102,
 * backbone.py:100-102), the 1x1 case of the kernel above with the column sums of dy riding along on the same operand:
 *   dW[n, c] = sum_t dy[t, n] x[t, c],    db[n] = sum_t dy[t, n]
 * The kernel writes mdetr_token_wgrad_chunks(T, C, N) partials, fp32 [chunks][N * C + (with_bias ? N : 0)] (a chunk's dW block,
 * then its db); the caller adds the chunks in a fixed order (mdetr_column_sum_to) and rounds once.
 *   x   bf16 [T, C] contiguous, C % 64 == 0;  dy  bf16 [T, N] contiguous, N % 32 == 0;  T % 8 == 0;  both 16-byte aligned
 */
int mdetr_token_wgrad_chunks(int64_t T, int C, int N);
int mdetr_token_wgrad(const void *x, const void *dy, float *partial, int64_t partial_floats, int64_t T, int C, int N, int with_bias,
                      int device, void *stream);

/*
 * y = dropout(relu(x + bias[col] + skip)) over a [rows, cols] channels-last activation in one pass, and its backward --
 * the tails the reference runs as separate operators after a convolution / linear layer: frozen-BN shift + ReLU after the
 * bottleneck's 3x3 convolution and "out += identity; relu(out)" after its expansion (lib/models/monodetr/backbone.py:100-102
 * -> torchvision resnet Bottleneck.forward), ReLU + Dropout between the two FFN layers (depthaware_transformer.py:334-337,
 * :431-435; depth_predictor/transformer.py:57-65).
 *   io_dtype   MDETR_F32 (cols % 4 == 0) or MDETR_BF16 (cols % 8 == 0): x, skip, y (and dy, dx); fp32 arithmetic, one rounding
 *   bias       [cols] fp32, or bf16 with a bf16 activation (bias_dtype); NULL = none.  skip [rows, cols]; NULL = none
 *   y          out; may be x itself (in place).  All tensors contiguous and 16-byte aligned
 *   relu       0 / 1 (NaN passes through, as clamp_min)
 *   dropout    p in [0, 1): element (row, col) is kept iff a hash of (seed [+ *seed_dev], row * cols + col) >= p * 2^32 and
 *              scaled by 1 / (1 - p) (the decision function of mdetr_add_layernorm_forward); no mask is stored
 * backward (relu = 1): dx = y > 0 ? dy * scale : 0 with scale = 1 / (1 - p) -- y > 0 holds exactly where the
 * pre-activation was positive and the element was kept, so neither the mask nor the hash is needed; dx may be dy itself.
 */
int mdetr_bias_act_forward(int io_dtype, int bias_dtype, const void *x, const void *bias, const void *skip, void *y,
                           int64_t rows, int cols, int relu, float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                           int device, void *stream);
int mdetr_bias_act_backward(int io_dtype, const void *dy, const void *y, void *dx, int64_t rows, int cols, float scale,
                            int device, void *stream);

/*
 * Every second pixel of a channels-last activation, and the adjoint: a 1x1 convolution of stride 2 (the projection shortcut
 * of a ResNet stage's first block, torchvision Bottleneck.downsample behind lib/models/monodetr/backbone.py:93-106) is a token
 * GEMM over x[:, ::2, ::2].
 *   backward = 0: src = x [B, H, W, C] -> dst = y [B, (H + 1) / 2, (W + 1) / 2, C]
 *   backward = 1: src = dy [B, (H + 1) / 2, (W + 1) / 2, C] -> dst = dx [B, H, W, C], zero at the skipped pixels
 * pixel_bytes = C * element size (any element type), a multiple of 16; pointers 16-byte aligned.
 */
/*
 * The frozen-BatchNorm fold of many trainable convolution weights in one launch each way (ABI 8; csrc/wfold.hip).  The reference
 * applies FrozenBatchNorm2d (lib/models/monodetr/backbone.py:27-64, y = x * scale + shift) as a pass over every activation behind
 * each convolution; here conv(x, W * scale) + shift, so with trainable weights every iteration folds
 *     folded_i[o][t][c] = bf16(w_i[o][t][c] * scale_i[o])        (fp32 [O][taps][C]: a channels-last weight as it lies in memory)
 * and, where folded_t[i] != NULL, writes the same values as folded_t_i[c][t][o] (the input-gradient kernels' operand), and unfolds
 *     dw_i[o][t][c] = float(dfolded_i[o][t][c]) * scale_i[o]     (bf16 in, fp32 out).
 * HOST arrays of n entries each: device pointers (they travel as kernel arguments, 48 tensors per launch) and the dimensions.
 * O_i and C_i multiples of 8, pointers 16-byte aligned.  folded_t itself may be NULL (no transposed copies at all).
 */
int mdetr_fold_weights(int n, const void *const *w, const void *const *scale, void *const *folded, void *const *folded_t,
                       const int *O, const int *C, const int *taps, int device, void *stream);
int mdetr_unfold_grads(int n, const void *const *dfolded, const void *const *scale, void *const *dw,
                       const int *O, const int *C, const int *taps, int device, void *stream);

/*
 * Gather of many dense device tensors into one flat device buffer (the optimizer's flat gradient buffer; the reference's AdamW
 * walks parameters one by one, lib/helpers/optimizer_helper.py:69-129).  HOST arrays: src_ptrs[ntensors] (device addresses; they
 * travel as kernel arguments, 256 per launch) and tensor_block_begin[ntensors + 1] (first workgroup of each tensor).  DEVICE
 * arrays: dst_offsets[i] / nbytes[i], the place and size of tensor i in bytes, and one entry per workgroup -- block_tensor[b],
 * block_start[b]: workgroup b copies bytes [block_start[b], + chunk_bytes) of its tensor (clipped to the tensor's size).
 */
int mdetr_gather_flat(const void *const *src_ptrs, int ntensors, const int *tensor_block_begin, void *dst, const int64_t *dst_offsets,
                      const int64_t *nbytes, const int *block_tensor, const int64_t *block_start, int chunk_bytes, int device, void *stream);

/* torchvision ResNet.maxpool (3x3 / stride 2 / pad 1) on a channels-last bf16 activation, forward only (the stem is frozen):
 * x [B, H, W, C] -> y [B, (H - 1) / 2 + 1, (W - 1) / 2 + 1, C]; C a multiple of 8, 16-byte aligned pointers. */
int mdetr_maxpool3x3s2_bf16(const void *x, void *y, int B, int H, int W, int C, int device, void *stream);
int mdetr_decimate2(int backward, const void *src, void *dst, int B, int H, int W, int64_t pixel_bytes, int device, void *stream);

/*
 * Weight and bias gradient of a linear layer y = x W^T + b over a few thousand token rows (the decoder's B x 550 query rows;
 * reference depthaware_transformer.py:399-456, monodetr.py:222-262): dW[n, k] = sum_t dY[t, n] X[t, k], db[n] = sum_t dY[t, n]
 * in one launch plus one chunk sum (deterministic, fp32 accumulation, one rounding into out_dtype).
 *   dy [rows, n] (row stride ldy >= n), x [rows, k] (row stride ldx, a multiple of 8 elements, 16-byte aligned): io_dtype
 *   MDETR_F32 / MDETR_BF16; rows <= 8 192 (65 536 for n <= 64); k a multiple of 64, any n >= 1, n * k <= 524 288.  n a multiple
 *   of 64 with 16-byte aligned dy rows takes vector loads, anything else guarded scalar loads of dy.
 *   out [round_up(n * k + n, 4)] in out_dtype: dW row-major, then db (then padding).  workspace:
 *   mdetr_small_wgrad_workspace_bytes(rows, n, k) bytes (0 = shape not supported)
 */
int64_t mdetr_small_wgrad_workspace_bytes(int64_t rows, int n, int k);
int mdetr_small_wgrad(int io_dtype, const void *dy, const void *x, void *out, int out_dtype, void *workspace, int64_t workspace_bytes,
                      int64_t rows, int n, int k, int64_t ldy, int64_t ldx, int device, void *stream);

/*
 * GroupNorm (+ ReLU) of a channels-last activation with 8 channels per group -- nn.GroupNorm(32, 256) after every input
 * projection (lib/models/monodetr/monodetr.py:77-99) and in the depth predictor's conv + GN (+ ReLU) stages
 * (depth_predictor/depth_predictor.py:30-56).  x, y, dy, dx: [n, hw, c] (NCHW tensors in channels_last memory format),
 * contiguous, 16-byte aligned, io_dtype MDETR_F32 / MDETR_BF16; gamma, beta: [c] in param_dtype (bf16 only with a bf16
 * activation); c == 8 * groups, c / 8 a power of two <= 256.  fp32 arithmetic; moments by pairwise (Chan) combination;
 * deterministic (no atomics).
 *   forward    y = (x - mean_g) * rstd_g * gamma + beta, then max(0, .) if relu; stats [n, groups, 2] fp32 (mean, rstd) out
 *   backward   dx; dparams [2, c] in param_dtype = (dgamma row, dbeta row).  The ReLU mask is recomputed from x, gamma, beta
 *              and stats with the forward's expression -- no output needs to be kept
 *   workspace  mdetr_group_norm_workspace_bytes(n, hw, c, groups) bytes of device memory, contents irrelevant (0 = shape not
 *              supported)
 */
int64_t mdetr_group_norm_workspace_bytes(int n, int64_t hw, int c, int groups);
int mdetr_group_norm_forward(int io_dtype, int param_dtype, const void *x, const void *gamma, const void *beta, void *y, float *stats,
                             void *workspace, int64_t workspace_bytes, int n, int64_t hw, int c, int groups, float eps, int relu,
                             int device, void *stream);
int mdetr_group_norm_backward(int io_dtype, int param_dtype, const void *dy, const void *x, const void *gamma, const void *beta,
                              const float *stats, void *dx, void *dparams, void *workspace, int64_t workspace_bytes,
                              int n, int64_t hw, int c, int groups, int relu, int device, void *stream);

/*
 * Batched linear sum assignment (Hungarian matching) on the device.  Replaces the host loop of
 * scipy.optimize.linear_sum_assignment calls in HungarianMatcher.forward (matcher.py:87-103: one
 * device->host copy and 3 x B x 11 solver calls per iteration).
 *
 *   cost         fp32; element (problem-image li = layer*images + b, query q, target t) at
 *                cost[li*img_stride + q*q_stride + t*t_stride]           (device)
 *   num_targets  [images] int32: valid targets of each image, <= kmax    (device)
 *   assign       [layers, images, groups, kmax] int32 (device): for problem (l, b, g) and target t <
 *                num_targets[b], the index in [g*n, (g+1)*n) of the query matched to it; -1 elsewhere
 *   n            queries per group (<= 128: one or two columns per lane of the solving wave), kmax <= min(n, 64)
 * Each (l, b, g) is an independent min-cost matching of all targets of image b to distinct queries
 * of group g, solved exactly in float64 (shortest augmenting paths, one wave64 per problem).
 */
int mdetr_lsa_forward(const float *cost, const int32_t *num_targets, int32_t *assign,
                      int layers, int images, int groups, int n, int kmax,
                      int64_t img_stride, int64_t q_stride, int64_t t_stride, int device, void *stream);

/*
 * Optional per-launch kernel timing with HIP events recorded on the launch stream (bench.py's
 * `roofline` block).  When enabled, every fast-path/generic MSDA kernel launch (the kernel only,
 * not the zero-fill memsets of the backward) is bracketed by an event pair.
 *   mdetr_profile_enable(1|0)      start / stop recording (clears previous records when enabling)
 *   mdetr_profile_read(...)        synchronises the recorded events and aggregates them by
 *                                  (kind, key): kind 0 = msda forward, 1 = msda backward gather kernel
 *                                  (msda_bwd_d32 / generic), 2 = msda_scatter_tiles, 3 = msda_reduce_tiles
 *                                  with key = Lq; kind 4 = attention forward, 5 = attention backward
 *                                  (prep + dq + dkv) with key = Lq * 4096 + Lk.
 *                                  Fills up to `cap` rows of {kind, Lq, launches, total_ms} (as
 *                                  doubles, 4 per row) and returns the number of rows, or < 0.
 * Recording costs two hipEventRecord calls per launch; it is off by default.
 */
int mdetr_profile_enable(int on);
int mdetr_profile_read(double *rows, int cap);
/* The same rows with the useful work the launches declared: (kind, key, launches, total_ms, MFLOP, algorithmic KB), 6 doubles per row
 * (0 where a kind leaves the accounting to the caller: MSDA and attention, whose bytes / flops follow from the key). */
int mdetr_profile_read_work(double *rows, int cap);

#ifdef __cplusplus
}
#endif
#endif /* MONODETR_AMD_H */
