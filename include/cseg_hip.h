/*
 * cseg_hip.h -- C-ABI of libcseg_hip.so: the MI355X (gfx950) kernels behind the contrastive-training
 * hot path of tfzhou/ContrastiveSeg.
 *
 * Convention (the reference's own native style, lib/extensions/cc_attention/src/ca.cu:188-205 and
 * lib_cffi.cpp:24-37): every entry point is `extern "C" int f(..., stream)` returning 1 = ok, 0 = error;
 * the caller pre-allocates every output and workspace on the device, passes plain pointers + sizes and the
 * stream to launch on; the callee allocates nothing, owns nothing and never synchronises. After a 0 return
 * cseg_last_error() gives a thread-local message. All floats are fp32, all indices int32 unless noted.
 * No torch types appear here; the Python host side (contrastiveseg_amd/_hip.py) binds these with ctypes.
 *
 * Each entry cites the reference code (paths relative to the reference root) that it replaces.
 */
#ifndef CSEG_HIP_H
#define CSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cseg_stream_t; /* hipStream_t */

int cseg_abi_version(void);
const char* cseg_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Anchor mining, device part.  Replaces the tensor work of
 *   lib/loss/loss_contrast.py:131-134 (labels -> float -> nearest resize -> long),
 *   lib/loss/loss_contrast.py:183     (torch.max(seg, 1)),
 *   lib/loss/loss_contrast.py:35-42, 60-64 (unique / per-class nonzero of hard and easy pixels).
 * P = h*w. Classes are 0..K-1; `ignore_label` pixels and labels outside [0,K) are dropped (the latter are
 * counted in status[0] so the host can refuse them).
 *   seg      [B,K,h,w] f32 (or NULL: then pred_in [B,P] i64 supplies the prediction)   target [B,H,W] i64
 *   lab,pred [B,P] i32 (nullable)          key [B,P] i16 workspace (2*c + (pred==c), -1 = dropped)
 *   counts   [B,K,2] i32  (hard, easy)     seg_off [B,K,2] i32 exclusive offsets inside image b's slice
 *   part_idx [B,P] i32: for image b, part_idx[b*P + seg_off[b,c,e] + r] = r-th smallest pixel of class c,
 *            e = 0 hard (lab==c, pred!=c), e = 1 easy (lab==c, pred==c) -- the order .nonzero() returns.
 *   status   [4] i32: [0] = number of labels outside [0,K) that are not ignore_label.
 * ------------------------------------------------------------------------------------------------ */
int cseg_classify_partition(const float* seg, const int64_t* pred_in, const int64_t* target, int B, int K, int h, int w, int H, int W,
                            int ignore_label, int32_t* lab, int32_t* pred, int16_t* key, int32_t* counts,
                            int32_t* seg_off, int32_t* part_idx, int32_t* status, cseg_stream_t stream);

/* Gather of the mined pixels.  Replaces lib/loss/loss_contrast.py:141-142 (NHWC copy of all embeddings) and
 * :85-87 (fancy-index gather).  embed stays NCHW [B,D,P].
 *   sel_pos [N] i32: b*P + seg_off + rank (position inside part_idx), rows already in contrast order
 *   anchors [N,D] f32, sel_pix [N] i32 = b*P + pixel (kept for the backward scatter and for tests) */
int cseg_gather_anchors(const float* embed, int B, int D, int P, const int32_t* part_idx,
                        const int32_t* sel_pos, int N, float* anchors, int32_t* sel_pix, cseg_stream_t stream);

/* Backward of the gather (autograd of X[ii, indices, :], loss_contrast.py:85): d_embed[b, :, pix] = scale *
 * sum_s d_anchor_parts[s, r, :]. d_embed [B,D,P] must be zero-filled by the caller; rows of sel_pix are
 * duplicate-free within one step, so no atomics. n_parts = cseg_contrast_bwd_parts(). */
int cseg_scatter_anchor_grad(const float* d_anchor_parts, int n_parts, const int32_t* sel_pix, int N, int D,
                             int P, float scale, float* d_embed, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Contrastive term.  Replaces PixelContrastLoss._contrastive:
 *   self mode   lib/loss/loss_contrast.py:91-128      (contrast set = the anchors, view-major rows)
 *   bank mode   lib/loss/loss_contrast_mem.py:91-152  (contrast set = cat(segment_queue, pixel_queue, 1)
 *               read in place: columns are classes 1..K-1, 2*ms rows each, then 2*ms zero rows labelled 0)
 *   plain mode  any [M,D] contrast matrix with labels (used for the cross-rank gathered set)
 * S = A.C^T / temperature runs on the fp32 MFMA (v_mfma_f32_32x32x2_f32); row statistics, the
 * "pair + all negatives" denominator, the column-index self mask and the mean over positives follow the
 * reference formulas exactly (see oracle/cseg_oracle.py:_contrast_core).
 *   anchors [N,D], a_lab [N] i32
 *   mode 0 self : contrast/c_lab/queues ignored, M = N
 *   mode 1 plain: contrast [M,D], c_lab [M]
 *   mode 2 bank : segment_queue, pixel_queue [K,ms,D]; M = K*2*ms
 *   S_ws [N,M] f32 workspace (kept for backward), row_stats [N,4] f32, row_loss [N] f32, loss [1] f32
 * Requires N <= M (the reference's scatter_ raises otherwise) and D % 8 == 0.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int mode;               /* 0 self, 1 plain, 2 bank */
    int N, M, D;
    const float* anchors;   /* [N,D] */
    const int32_t* a_lab;   /* [N] */
    const float* contrast;  /* mode 1: [M,D] */
    const int32_t* c_lab;   /* mode 1: [M] */
    const float* segment_queue; /* mode 2: [K,ms,D] */
    const float* pixel_queue;   /* mode 2: [K,ms,D] */
    int bank_classes;       /* mode 2: K */
    int bank_size;          /* mode 2: ms */
    float temperature;
    float base_temperature;
} cseg_contrast_desc;

size_t cseg_contrast_ws_bytes(int N, int M);
int cseg_contrast_fwd(const cseg_contrast_desc* d, float* S_ws, float* row_stats, float* row_loss, float* loss,
                      cseg_stream_t stream);
/* The same forward as ONE launch (round 5): S tiles on the fp32 MFMA, online (max, negative sum, positive count) per row by
 * wavefront reductions, the positives' sweep from LDS after an in-launch hand-off between the blocks of a row strip. No S round
 * trip through HBM inside the forward.
 *   fused_ws  scratch of cseg_contrast_fused_ws_bytes(N, M) bytes whose LAST 2 * strips + 1 ints (at byte offset
 *             cseg_contrast_fused_counter_offset(N, M)) are ZERO at the first launch that uses the buffer; the kernel leaves
 *             them zero, so a buffer kept per (device, stream) needs no fill between launches. Not shared by concurrent launches.
 *   S_out     [N, round32(M)] f32 or NULL: the similarity tiles stored once for cseg_contrast_bwd (which reads S); NULL = the
 *             N x M array never exists.
 * Same outputs as cseg_contrast_fwd (row_stats [N,4], row_loss [N], loss [1]), equal to ~1e-7 relative (online rescaling of the
 * log-sum-exp instead of max-then-sum), run-to-run deterministic. Reference: lib/loss/loss_contrast.py:91-128,
 * lib/loss/loss_contrast_mem.py:107-152. */
size_t cseg_contrast_fused_ws_bytes(int N, int M);
size_t cseg_contrast_fused_counter_offset(int N, int M);
int cseg_contrast_fwd_fused(const cseg_contrast_desc* d, float* fused_ws, float* S_out, float* row_stats, float* row_loss,
                            float* loss, cseg_stream_t stream);
/* d_loss [1] f32 on the device (upstream gradient). Writes d_anchor_parts [n_parts, N, D] (sum over parts =
 * dLoss/dAnchors); in self mode this already contains both the row and the column role of every anchor. */
int cseg_contrast_bwd_parts(int N, int M, int D);
int cseg_contrast_bwd(const cseg_contrast_desc* d, const float* S_ws, const float* row_stats,
                      const float* d_loss, float* d_anchor_parts, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * HRNet head input: bilinear(align_corners=True) upsample of the low-resolution maps to the first map's size
 * fused with the channel concat.  Replaces lib/models/nets/hrnet.py:86-91 (3x F.interpolate + torch.cat).
 *   n_maps <= 4; xs[i] [B,C[i],hs[i],ws[i]] NCHW; out [B,sum C,hs[0],ws[0]].
 * ------------------------------------------------------------------------------------------------ */
int cseg_upcat_fwd(const float* const* xs, const int* C, const int* hs, const int* ws, int n_maps, int B,
                   float* out, cseg_stream_t stream);
/* the same, with max|out| accumulated into `amax` while the values are stored (a zeroed CSEG_AMAX_WORDS record, the format of
 * cseg_amax_f32): the 720-channel tensor is the input of two split-operand convolutions. Width of the finest map % 4 == 0. */
int cseg_upcat_fwd_amax(const float* const* xs, const int* C, const int* hs, const int* ws, int n_maps, int B,
                        float* out, unsigned* amax, cseg_stream_t stream);
int cseg_upcat_bwd(const float* d_out, const int* C, const int* hs, const int* ws, int n_maps, int B,
                   float* const* d_xs, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * HRNet exchange unit: out = relu(sum_s same[s] + sum_l bilinear_up(low[l])), all terms with C channels.
 * Replaces the chain of F.interpolate + add + ReLU in HighResolutionModule.forward
 * (lib/models/backbones/hrnet/hrnet_backbone.py:271-286).  n_same <= 4, n_low <= 3.
 *   bwd: g_same [B,C,h,w] = d_out * (out > 0) is the gradient of every same-resolution term (NULL to skip; pass
 *   out_act = NULL when the forward ran without ReLU, then the same-resolution gradient is d_out itself);
 *   d_low[l] [B,C,low_h[l],low_w[l]] = exact adjoint of the upsample applied to the masked gradient.
 * ------------------------------------------------------------------------------------------------ */
int cseg_fuse_sum_fwd(const float* const* same, int n_same, const float* const* low, const int* low_h,
                      const int* low_w, int n_low, int B, int C, int h, int w, int relu, float* out,
                      cseg_stream_t stream);
/* the same, with max|out| accumulated into `amax` while the values are stored (a zeroed CSEG_AMAX_WORDS record, the format of
 * cseg_amax_f32): the outputs of an exchange unit are the inputs of the next unit's split-operand convolutions. */
int cseg_fuse_sum_fwd_amax(const float* const* same, int n_same, const float* const* low, const int* low_h,
                           const int* low_w, int n_low, int B, int C, int h, int w, int relu, float* out, unsigned* amax,
                           cseg_stream_t stream);
int cseg_fuse_sum_bwd(const float* d_out, const float* out_act, const int* low_h, const int* low_w, int n_low, int B,
                      int C, int h, int w, float* g_same, float* const* d_low, cseg_stream_t stream);
/* out[b][c][p] = a[c] + b[c] * u[b][c][p] over [B,C,P] (P % 4 == 0, 16-byte aligned), with max|out| accumulated into `amax` when it is not
 * NULL (a zeroed CSEG_AMAX_WORDS record): the dense part of a BatchNorm input gradient whose output gradient lives on a few pixels --
 * the row-sparse backward of the projection head (reference lib/models/modules/projection.py:8-24 runs the dense adjoint). */
int cseg_affine_channels(const float* u, const float* a, const float* b, int B, int C, long P, float* out, unsigned* amax,
                         cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Segmentation term: bilinear(align_corners=True) upsample of the logits to the label size fused with the
 * weighted cross entropy.  Replaces lib/loss/loss_contrast.py:180-181 + lib/loss/loss_helper.py:169-206
 * (nn.CrossEntropyLoss(weight, ignore_index, reduction mean)); the [B,K,H,W] tensor is never materialised.
 *   seg [B,K,h,w], target [B,H,W] i64, weight [K] or NULL
 *   partial [2*n_blocks] f32 workspace, n_blocks = cseg_upsample_ce_blocks(B,H,W)
 *   out [2] f32: out[0] = loss, out[1] = sum of weights of valid pixels; status[1] counts bad targets
 *   lse [B,H,W] f32: per-label-pixel log-sum-exp of the upsampled logits, written by fwd (nullable there) and read
 *       by bwd, so that the backward evaluates every exp once and needs no second pass over the classes
 *   bwd: d_loss [1] device scalar; d_seg [B,K,h,w] is overwritten
 * Upsampling factors up to 16x per axis (H >= h, W >= w).
 * ------------------------------------------------------------------------------------------------ */
int cseg_upsample_ce_blocks(int B, int H, int W);
int cseg_upsample_ce_fwd(const float* seg, const int64_t* target, const float* weight, int ignore_label, int B,
                         int K, int h, int w, int H, int W, float* partial, float* out, int32_t* status,
                         float* lse, cseg_stream_t stream);
int cseg_upsample_ce_bwd(const float* seg, const int64_t* target, const float* weight, int ignore_label, int B,
                         int K, int h, int w, int H, int W, const float* out, const float* d_loss,
                         const float* lse, float* d_seg, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Memory bank update, device part.  Replaces the tensor work of
 * segmentor/trainer_contrastive.py:102-138 (_dequeue_and_enqueue).
 *   cseg_queue_count:      histogram of labels[:, ::stride, ::stride] per image -> counts [B,K] i32 (labels
 *                          outside [0,K) are skipped; the host applies the reference's `> 0` filter).
 *   cseg_queue_class_sums: sums[b,c,:] = sum over positions q with strided label c of keys[b,:,q]; q indexes
 *                          the strided label map and is used as a raw position into the key map exactly as the
 *                          reference does (trainer_contrastive.py:111-120); needs Hs*Ws <= Pk.
 *   cseg_queue_write_segments: segment_queue[job_cls, job_dst_row, :] = L2-normalise(sums[img,cls,:] / count).
 *   cseg_queue_write_pixels:   pixel_queue[dst_cls, dst_row, :] = L2-normalise(keys[src_img, :, src_pos]).
 *   keys [B,D,Pk] f32, labels [B,H,W] i64, queues [K,ms,D] f32. The host resolves the pointer arithmetic and
 *   write-after-write order (later rows win) before building the job lists.
 * ------------------------------------------------------------------------------------------------ */
int cseg_queue_count(const int64_t* labels, int B, int H, int W, int stride, int K, int32_t* counts,
                     cseg_stream_t stream);
int cseg_queue_class_sums(const float* keys, const int64_t* labels, int B, int D, int Pk, int H, int W, int stride,
                          int K, float* sums, cseg_stream_t stream);
int cseg_queue_write_segments(const float* sums, const int32_t* counts, const int32_t* job_img,
                              const int32_t* job_cls, const int32_t* job_dst_row, int n_jobs, int K, int D,
                              float* segment_queue, int ms, cseg_stream_t stream);
int cseg_queue_write_pixels(const float* keys, int B, int D, int Pk, const int32_t* src_img,
                            const int32_t* src_pos, const int32_t* dst_cls, const int32_t* dst_row, int n_rows,
                            float* pixel_queue, int ms, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused (Sync)BatchNorm + residual add + ReLU, NCHW fp32.  Replaces what
 *   lib/models/tools/module_helper.py:29-68 (BatchNorm2d / BNReLU -> nn.SyncBatchNorm + nn.ReLU) and the
 *   relu(bn(conv(x)) [+ residual]) chains of lib/models/backbones/hrnet/hrnet_backbone.py:49-105 and
 *   lib/models/backbones/resnet/resnet_models.py run as separate BN / add / clamp / threshold kernels.
 * x, y, dy, dx, residual, out: [B,C,HW] f32.  mean_invstd [C,2] f32 = (mean, 1/sqrt(var+eps)).
 * moments / sums [C+1,2] f64: the ONLY data a SyncBN exchange needs -- the host all-reduces them (RCCL) between
 * cseg_bn_stats and cseg_bn_finalize, and between cseg_bn_bwd_reduce and cseg_bn_bwd_apply. Row C (ABI 4) carries this
 * rank's element count per channel (B*HW, 0): summed by the same all-reduce it becomes the GLOBAL count, which
 * cseg_bn_finalize / cseg_bn_bwd_apply read on the device when their `count` argument is 0 -- ranks may then contribute
 * different batch sizes (torch.nn.SyncBatchNorm semantics) without a second collective or a host round trip.
 * ws: cseg_bn_ws_floats(B,C,HW) floats of scratch.  weight / bias may be NULL (affine=False).
 * ------------------------------------------------------------------------------------------------ */
size_t cseg_bn_ws_floats(int B, int C, int HW);
/* moments[c] = (sum x, sum x^2) over this rank's B*HW values of channel c; moments[C] = (B*HW, 0) */
int cseg_bn_stats(const float* x, int B, int C, int HW, float* ws, double* moments, cseg_stream_t stream);
/* (globally summed) moments + total count (0 = moments[C][0]) -> mean_invstd; running_mean/var (nullable pair) updated in place with
 * `momentum` and the unbiased variance, *num_batches_tracked (nullable, i64) incremented: nn.BatchNorm2d semantics */
int cseg_bn_finalize(const double* moments, int C, double count, float eps, float momentum, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float* mean_invstd, cseg_stream_t stream);
/* single-rank shortcut: cseg_bn_stats + cseg_bn_finalize without the fp64 round trip */
int cseg_bn_stats_finalize(const float* x, int B, int C, int HW, float* ws, float eps, float momentum,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean_invstd,
                           cseg_stream_t stream);
/* single-rank training step in two launches each: cseg_bn_fwd = statistics + (finalise, running statistics, apply);
 * cseg_bn_bwd = reduce + (sums, d_weight/d_bias, dx); arguments as in the split entry points below; dx may be NULL
 * (parameter gradients only); training = 0 freezes the statistics (eval-mode backward: dx = w*invstd*dy'). */
int cseg_bn_fwd(const float* x, const float* residual, const float* weight, const float* bias, int relu, int B, int C,
                int HW, float* ws, float eps, float momentum, float* running_mean, float* running_var,
                int64_t* num_batches_tracked, float* mean_invstd, float* y, cseg_stream_t stream);
int cseg_bn_bwd(const float* dy, const float* x, const float* out, const float* mean_invstd, const float* weight,
                const float* bias, int mode, int training, int B, int C, int HW, float* ws, float* g_masked,
                float* d_weight, float* d_bias, float* dx, cseg_stream_t stream);
/* y = relu?((x - mean) * invstd * weight + bias [+ residual]) */
int cseg_bn_apply(const float* x, const float* residual, const float* mean_invstd, const float* weight,
                  const float* bias, int relu, int B, int C, int HW, float* y, cseg_stream_t stream);
/* sums[c] = (sum dy', sum dy' * (x - mean)), sums[C] = (B*HW, 0); d_weight[c] = sums[c][1] * invstd, d_bias[c] = sums[c][0] (nullable,
 * rank-local like torch's SyncBatchNorm).  dy' = dy (mode 0) | dy * [(x-mean)*invstd*w+b > 0] (mode 1: ReLU mask
 * recomputed from x) | dy * [out > 0] (mode 2: residual case; dy' is written to g_masked = gradient of the residual) */
int cseg_bn_bwd_reduce(const float* dy, const float* x, const float* out, const float* mean_invstd, const float* weight,
                       const float* bias, int mode, int B, int C, int HW, float* ws, float* g_masked, double* sums,
                       float* d_weight, float* d_bias, cseg_stream_t stream);
/* dx = w*invstd * (dy' - sums0/count - (x-mean) * invstd^2 * sums1/count), count 0 = sums[C][0]; sums == NULL: frozen statistics (eval),
 * dx = w*invstd*dy'.  mask_from_x: apply the mode-1 mask to dy here too (pass the already masked g for mode 2). */
int cseg_bn_bwd_apply(const float* dy, const float* x, const float* mean_invstd, const float* weight, const float* bias,
                      const double* sums, double count, int mask_from_x, int B, int C, int HW, float* dx,
                      cseg_stream_t stream);

/* Round 3: the four apply-type calls above with max|output| accumulated into *amax_out (uint32 bit pattern of the float,
 * atomicMax; the caller zeroes the word): the tensor they write is the operand of the next split-operand convolution
 * (CSEG_ARITH_F16X3 scales it by a power of two derived from this word), so the maximum comes for free instead of from an
 * extra pass over the tensor (cseg_amax_f32).  amax_out == NULL: identical to the plain call. */
int cseg_bn_fwd_amax(const float* x, const float* residual, const float* weight, const float* bias, int relu, int B, int C,
                     int HW, float* ws, float eps, float momentum, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float* mean_invstd, float* y, unsigned* amax_out, cseg_stream_t stream);
int cseg_bn_bwd_amax(const float* dy, const float* x, const float* out, const float* mean_invstd, const float* weight,
                     const float* bias, int mode, int training, int B, int C, int HW, float* ws, float* g_masked,
                     float* d_weight, float* d_bias, float* dx, unsigned* amax_out, cseg_stream_t stream);
int cseg_bn_apply_amax(const float* x, const float* residual, const float* mean_invstd, const float* weight,
                       const float* bias, int relu, int B, int C, int HW, float* y, unsigned* amax_out, cseg_stream_t stream);
int cseg_bn_bwd_apply_amax(const float* dy, const float* x, const float* mean_invstd, const float* weight, const float* bias,
                           const double* sums, double count, int mask_from_x, int B, int C, int HW, float* dx,
                           unsigned* amax_out, cseg_stream_t stream);

/* Round 4 (ABI 4): BatchNorm statistics from the epilogue of the convolution that produces the tensor (SURVEY.md section 8 f2;
 * the conv -> bn chains of lib/models/backbones/hrnet/hrnet_backbone.py:49-65, lib/models/tools/module_helper.py:35-39).
 * The `_st` forms of the split-operand forward convolutions are the plain calls plus `stats`: [Cout][T] float4 = (count, mean,
 * M2 = sum (y - mean)^2) of every 64-pixel segment of the output, T = cseg_conv_stat_segments(kind, B, H, W) (kind 0: 3x3
 * kernels, H x W of the OUTPUT; kind 1: 1x1 kernels).  cseg_bn_tiles_finalize replaces cseg_bn_stats_finalize (no pass over
 * y), cseg_bn_tiles_moments replaces cseg_bn_stats (moments [C+1,2] f64 for the SyncBN exchange). */
size_t cseg_conv_stat_segments(int kind, int B, int H, int W);
int cseg_conv3x3_split_fwd_st(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H, int W, int nt,
                              int arith, const unsigned* amax_x, const unsigned* amax_w, float* y, float* stats, cseg_stream_t stream);
int cseg_conv1x1_split_fwd_st(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int HW, int arith,
                              const unsigned* amax_x, const unsigned* amax_w, float* y, float* stats, cseg_stream_t stream);
int cseg_conv3x3_s2_split_fwd_st(const float* x, const void* wp, int B, int Cin, int Cout, int Ho, int Wo, int nt,
                                 const unsigned* amax_x, const unsigned* amax_w, float* y, float* stats, cseg_stream_t stream);
int cseg_bn_tiles_finalize(const float* stats, int C, long T, float eps, float momentum, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float* mean_invstd, cseg_stream_t stream);
int cseg_bn_tiles_moments(const float* stats, int C, long T, double* moments, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution, NCHW fp32, forward and backward-data, for the narrow HRNet branches
 * (lib/models/backbones/hrnet/hrnet_backbone.py:35-66 BasicBlock conv1/conv2 -> nn.Conv2d -> MIOpen in the reference;
 * MIOpen's best solver for the 48-channel shape reaches 45-50 TFLOP/s of the 157 TFLOP/s fp32 MFMA peak).
 * Implicit GEMM on v_mfma_f32_16x16x4_f32, exact fp32 FMA chain.  Cin % 8 == 0, Cout % 48 == 0, W % 4 == 0.
 *   cseg_conv3x3_pack_weights: w [Cout,Cin,3,3] -> wp (cseg_conv3x3_packed_floats floats) in MFMA lane order;
 *       transpose_flip = 1 packs the backward-data operator (maps Cout -> Cin channels, taps mirrored);
 *       cseg_conv3x3_packed_floats(conv_in, conv_out) takes the channel counts of the PACKED operator.
 *   cseg_conv3x3_fwd: y [B,Cout,H,W] = conv(x [B,Cin,H,W], wp); for backward-data call it with x = dy,
 *       Cin = forward Cout, Cout = forward Cin and the transpose_flip packing.
 *   cseg_conv3x3_wrw: the weight gradient (MIOpen's solver for these shapes is an NHWC implicit-GEMM kernel wrapped in
 *       three layout transposes).
 * ------------------------------------------------------------------------------------------------ */
size_t cseg_conv3x3_packed_floats(int Cin, int Cout);
int cseg_conv3x3_pack_weights(const float* w, int Cout, int Cin, int transpose_flip, float* wp, cseg_stream_t stream);
int cseg_conv3x3_fwd(const float* x, const float* wp, int B, int Cin, int Cout, int H, int W, float* y,
                     cseg_stream_t stream);
/* weight gradient dw [Cout,Cin,3,3] = sum_{b,y,x} dy * shifted x; Cin % 48 == 0, Cout % 48 == 0, W % 4 == 0;
 * ws: cseg_conv3x3_wrw_ws_floats(...) floats of scratch (per-split partials, summed in a fixed order: deterministic) */
size_t cseg_conv3x3_wrw_ws_floats(int B, int Cin, int Cout, int H, int W);
int cseg_conv3x3_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int H, int W, float* ws, float* dw,
                     cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The same convolution (3x3 / stride 1 / pad 1, NCHW fp32 in and out) on the BF16 matrix cores with split operands
 * ("bf16x6"): every fp32 operand = hi + mid + lo bf16 pieces (exact), the six piece products down to 2^-16 of the leading
 * one are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 -- fp32-class accuracy (tools/split_bf16_probe.py) at up to
 * 2500/6 = 417 TFLOP/s of fp32-equivalent work instead of the 157 TFLOP/s fp32 MFMA rate.  Replaces nn.Conv2d -> MIOpen
 * for the 720 -> 720 head convolution (lib/models/nets/hrnet.py:72-77 of the reference) and the HRNet branch
 * convolutions.  Cin % 16 == 0, Cout % 48 == 0, W % 4 == 0.  The host side uses it by default
 * (CSEG_CONV3X3_SPLIT_BF16=0 switches back to fp32 MFMA / MIOpen).
 *   cseg_conv3x3_sb_pack_weights: w [Cout,Cin,3,3] -> wp (cseg_conv3x3_sb_packed_bytes bytes, 16-byte aligned): split and
 *       laid out in MFMA lane order; transpose_flip = 1 packs the backward-data operator (maps Cout -> Cin channels);
 *       cseg_conv3x3_sb_packed_bytes(conv_in, conv_out) takes the channel counts of the PACKED operator.
 *   cseg_conv3x3_sb_fwd: y [B,Cout,H,W] = conv(x [B,Cin,H,W], wp) (+ bias[Cout] when not NULL).
 * ------------------------------------------------------------------------------------------------ */
size_t cseg_conv3x3_sb_packed_bytes(int Cin, int Cout);
int cseg_conv3x3_sb_pack_weights(const float* w, int Cout, int Cin, int transpose_flip, void* wp, cseg_stream_t stream);
int cseg_conv3x3_sb_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H, int W,
                        float* y, cseg_stream_t stream);
/* the same two calls with an explicit number of 16-channel tiles per block (nt = 3, 6 or 9; must divide conv_out / 16;
 * pack and forward must use the same nt): smaller nt = more blocks for small feature maps (192 channels at 8x32x64:
 * nt 3 -> 256 blocks, 81 us; nt 6 -> 128 blocks, 113 us). */
int cseg_conv3x3_sb_pack_weights_nt(const float* w, int Cout, int Cin, int transpose_flip, int nt, void* wp,
                                    cseg_stream_t stream);
int cseg_conv3x3_sb_fwd_nt(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H, int W, int nt,
                           float* y, cseg_stream_t stream);

/* weight gradient of the same convolution on the split-bf16 path: dw [Cout,Cin,3,3] = sum_{b,y,x} dy * shifted x.
 * Cin % 16 == 0, Cout % 48 == 0, W % 64 == 0.  ws: cseg_conv3x3_sb_wrw_ws_floats(...) floats of scratch (per-split
 * partials, summed in a fixed order: deterministic).  Replaces MIOpen's NHWC implicit-GEMM weight-gradient kernels
 * (+ 3 layout transposes) behind nn.Conv2d: 48 ch 108 vs 201 us, 96 ch 112 vs 148 us, 720 ch 17.6 vs 19.5 ms.  The host
 * side keeps it opt-in (kernels.CONV3X3_SB_WRW) until the one-SGD-step goldens have run on it. */
size_t cseg_conv3x3_sb_wrw_ws_floats(int B, int Cin, int Cout, int H, int W);
int cseg_conv3x3_sb_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int H, int W, float* ws, float* dw,
                        cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 1x1 / stride 1 convolution (NCHW fp32 in and out) on the BF16 matrix cores with split operands (bf16x6, fp32-class
 * accuracy): the projection head of lib/models/modules/projection.py:8-24 (720 -> 720 -> 256) and the pointwise layers of
 * the encoder; nn.Conv2d(k=1) -> rocBLAS / MIOpen fp32 in the reference.  Cin % 16 == 0, Cout % 48 == 0 or % 64 == 0,
 * H*W % 4 == 0.  transpose = 1 packs the backward-data operator (maps Cout -> Cin channels).  Written and index-checked in
 * round 2, first hardware run pending: the host side keeps it opt-in (kernels.CONV1X1_SPLIT_BF16).
 * ------------------------------------------------------------------------------------------------ */
size_t cseg_conv1x1_sb_packed_bytes(int Cin, int Cout);
int cseg_conv1x1_sb_pack_weights(const float* w, int Cout, int Cin, int transpose, void* wp, cseg_stream_t stream);
int cseg_conv1x1_sb_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int HW, float* y,
                        cseg_stream_t stream);
/* weight gradient of the 1x1 convolution on the same path: dw [Cout,Cin] = sum_{b,p} dy[b][co][p] * x[b][ci][p].
 * Cin % 16 == 0, Cout % 16 == 0, H*W % 32 == 0; ws: cseg_conv1x1_sb_wrw_ws_floats(...) floats (per-split partials, fixed
 * order of summation).  Opt-in on the host side (kernels.CONV1X1_SB_WRW), first hardware run pending. */
size_t cseg_conv1x1_sb_wrw_ws_floats(int B, int Cin, int Cout, int HW);
int cseg_conv1x1_sb_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int HW, float* ws, float* dw,
                        cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Round 3 (ABI 3): the split-operand convolutions with a selectable arithmetic.  Same operators, same layouts, same reference
 * sites as the cseg_conv3x3_sb_* / cseg_conv1x1_sb_* entry points above (nn.Conv2d -> cuDNN/MIOpen fp32 in the reference:
 * lib/models/nets/hrnet.py:72-77, lib/models/backbones/hrnet/hrnet_backbone.py:35-66, lib/models/modules/projection.py:18-20);
 * those are now wrappers with arith = CSEG_ARITH_BF16X6.
 *   CSEG_ARITH_BF16X6  three bf16 pieces per operand, six MFMAs per product (roof 2500/6 = 417 TFLOP/s fp32-equivalent)
 *   CSEG_ARITH_F16X3   two fp16 pieces of the operand scaled by a power of two, three MFMAs per product (roof 833 TFLOP/s);
 *                      same fp32-class accuracy (csrc/cseg_split.h).  The scale comes from max|tensor|, which the caller
 *                      supplies as a DEVICE pointer to a max|tensor| record (CSEG_AMAX_WORDS uint32, below; cseg_amax_f32
 *                      accumulates into a record the caller has zeroed).  With
 *                      BF16X6 the amax pointers may be NULL.
 * pack and forward / weight gradient of one operator must use the same arith (and nt).  nt = 0: the library's tiling.
 * ------------------------------------------------------------------------------------------------ */
#define CSEG_ARITH_BF16X6 0
#define CSEG_ARITH_F16X3 1
/* A max|tensor| RECORD is CSEG_AMAX_WORDS uint32 (4 KB): CSEG_AMAX_SLOTS words, CSEG_AMAX_STRIDE words (128 bytes) apart, each
 * the bit pattern of a non-negative float; the value is the maximum over the slots.  Producers (cseg_amax_f32, the
 * cseg_bn_*_amax calls) accumulate with atomicMax into slot (block index mod CSEG_AMAX_SLOTS): thousands of blocks hitting ONE
 * word serialise in the L2 atomic unit (measured: 15 -> 33 us for a 48-channel BN apply kernel with a single word), 32 words on
 * 32 cache lines do not.  The caller zeroes the record; several tensors may share one. */
#define CSEG_AMAX_SLOTS 32
#define CSEG_AMAX_STRIDE 32
#define CSEG_AMAX_WORDS (CSEG_AMAX_SLOTS * CSEG_AMAX_STRIDE)
int cseg_amax_f32(const float* x, long n, unsigned* amax_bits, cseg_stream_t stream);
/* nt of the split 3x3 entry points: 0 = the library's channel tiling, 3 / 6 / 9 = explicit 16-channel tiles per block, or
 * CSEG_NT_SB8 = the 8 x 64-pixel kernel for wide layers (f16x3 only, output channels % 144: the 720 -> 720 classifier-head convolution,
 * lib/models/nets/hrnet.py:72-77): half the packed-weight bytes per MFMA of the 4 x 64 kernel. Pack and forward must use the same nt;
 * cseg_conv3x3_split_packed_bytes already covers either format. */
#define CSEG_NT_SB8 0x109
size_t cseg_conv3x3_split_packed_bytes(int arith, int Cin, int Cout);
int cseg_conv3x3_split_pack(const float* w, int Cout, int Cin, int transpose_flip, int nt, int arith, const unsigned* amax_w,
                            void* wp, cseg_stream_t stream);
int cseg_conv3x3_split_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H, int W, int nt,
                           int arith, const unsigned* amax_x, const unsigned* amax_w, float* y, cseg_stream_t stream);
/* y = conv(x) + bias + addend: `addend` (nullable) has the output's shape and is added in the epilogue -- the residual gradient that
 * meets the backward-data result of the first convolution of a BasicBlock (reference hrnet_backbone.py:49-65: `out += residual`
 * forward means two gradients arrive at the block input backward), saving the elementwise add autograd would launch. nt: 0 / 3 / 6 / 9. */
int cseg_conv3x3_split_fwd_add(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout,
                               int H, int W, int nt, int arith, const unsigned* amax_x, const unsigned* amax_w, float* y,
                               cseg_stream_t stream);
/* Round 5 (ABI 5): the same operator with a dilation of 2 or 4 (padding = dilation): the 3x3 convolutions of the dilated ResNet
 * stages of DeepLab-V3-R101-d8 (reference lib/models/backbones/resnet/resnet_backbone.py:88-101; layer3 rate 2, layer4 rate 4).
 * f16x3 only; Cout must be packed in the 16-channel-chunk form (multiples of 64 that are not multiples of 48, or 48 / 192: what
 * cseg_conv3x3_split_plan reports as kind 1); wp = the packed weights of cseg_conv3x3_split_pack (nt 0) for the same (Cout, Cin) --
 * transposed packing for backward-data. addend / stats nullable, not both. Any width. */
int cseg_conv3x3_split_dil_fwd(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H,
                               int W, int dil, int arith, const unsigned* amax_x, const unsigned* amax_w, float* y, float* stats,
                               cseg_stream_t stream);
/* ws: cseg_conv3x3_sb_wrw_ws_floats(...) */
int cseg_conv3x3_split_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int H, int W, int arith,
                           const unsigned* amax_x, const unsigned* amax_dy, float* ws, float* dw, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Round 6 (ABI 6): GROUPED launches -- the work of several independent layers in ONE kernel launch each.  Replaces the Python loop
 * over the parallel branches of an HRNet exchange unit (reference lib/models/backbones/hrnet/hrnet_backbone.py:262-288,
 * `for i in range(self.num_branches): x[i] = self.branches[i](x[i])`): at every depth of the residual chains (BasicBlock.forward,
 * :49-65) the 2-4 branches run convolutions / BatchNorms of equal flops and very different shapes that do not depend on each other.
 * A group = n <= CSEG_GROUP_MAX members described by HOST arrays of the structs below (read during the call; device pointers inside).
 * Results are bit-identical to the one-layer entry points named with each call, member by member.
 *
 * cseg_conv3x3_split_group_fwd: y_i = conv2d(x_i, w_i, bias_i, stride 1, padding 1) [+ addend_i] [+ BatchNorm statistics records of
 *   y_i as cseg_conv3x3_split_fwd_st writes them]; f16x3 only; Cin % 16 == 0, Cin >= 32, Cout % 48 == 0, any width. wp_i = weights
 *   packed with nt = CSEG_NT_GROUP (16-channel-chunk format, three 16-channel tiles per unit; transpose_flip for backward-data).
 *   One persistent launch: every XCD pulls (member, 4 x 64-pixel tile, 48-channel group) units from its own queue, heaviest members
 *   first. == cseg_conv3x3_split_fwd / _fwd_st / _fwd_add with nt = CSEG_NT_GROUP per member.
 *   sched: CSEG_GROUP_SCHED_INTS int32 on the device (best 128-byte aligned: one counter per cache line), ZERO before the first launch that uses it; every launch
 *   leaves it zero again, so one record serves all launches of a stream (not two launches that may run concurrently).
 * ------------------------------------------------------------------------------------------------ */
#define CSEG_GROUP_MAX 8
#define CSEG_GROUP_SCHED_INTS 320
#define CSEG_NT_GROUP 0x203
typedef struct cseg_conv_group_member {
    const float* x;           /* [B, Cin, H, W] */
    const void* wp;           /* packed weights (cseg_conv3x3_split_pack / cseg_split_pack_batch, nt = CSEG_NT_GROUP) */
    const float* bias;        /* [Cout] or NULL */
    const float* addend;      /* [B, Cout, H, W] or NULL */
    float* y;                 /* [B, Cout, H, W] */
    float* stats;             /* [Cout][cseg_conv_stat_segments(0, B, H, W)][4] or NULL (not together with addend) */
    const unsigned* amax_x;   /* max|x| record */
    const unsigned* amax_w;   /* max|w| record */
    int B, Cin, Cout, H, W;
    int reserved[3];
} cseg_conv_group_member;
int cseg_conv3x3_split_group_fwd(const cseg_conv_group_member* members, int n, int arith, int* sched, cseg_stream_t stream);
/* The BatchNorm passes of the members in one launch per pass (grid = the members' one-layer grids back to back). Fields a call does
 * not use may be NULL / 0. Members whose plane size is not a multiple of 4 floats (or unaligned) make the call fall back to one launch
 * per member -- same results.
 *   cseg_bn_group_tiles_finalize  == cseg_bn_tiles_finalize per member: stats [C][T][4] -> mean_invstd [C,2], running statistics,
 *                                    num_batches_tracked.
 *   cseg_bn_group_apply           == cseg_bn_apply_amax per member: y = act(bn(x) [+ residual]), max|y| into amax_out.
 *   cseg_bn_group_bwd             == cseg_bn_bwd_amax per member (mode / training as there; ws = cseg_bn_ws_floats(B, C, HW) floats
 *                                    per member): g_masked (mode 2), d_weight, d_bias, dx, max|dx| into amax_out. Two launches. */
typedef struct cseg_bn_group_member {
    const float* x;            /* [B, C, HW] the BatchNorm's input (the convolution's output) */
    const float* residual;     /* apply: [B, C, HW] or NULL */
    float* y;                  /* apply: output */
    const float* stats;        /* tiles_finalize: the convolution epilogue's records */
    float* mean_invstd;        /* [C, 2]: written by tiles_finalize, read by apply / bwd */
    const float* weight;       /* [C] or NULL */
    const float* bias;         /* [C] or NULL */
    float* running_mean;       /* [C] or NULL (with running_var) */
    float* running_var;
    int64_t* num_batches_tracked;
    unsigned* amax_out;        /* max|.| record of what apply / bwd write (y / dx), or NULL */
    const float* dy;           /* bwd: gradient of the output */
    const float* out;          /* bwd mode 2: the forward's output (ReLU mask) */
    float* g_masked;           /* bwd mode 2: masked gradient (= the residual's gradient) */
    float* d_weight;           /* bwd: [C] */
    float* d_bias;             /* bwd: [C] */
    float* dx;                 /* bwd: [B, C, HW] */
    float* ws;                 /* bwd: scratch */
    int B, C, HW;
    long T;                    /* tiles_finalize: records per channel */
    float eps, momentum;
} cseg_bn_group_member;
int cseg_bn_group_tiles_finalize(const cseg_bn_group_member* members, int n, cseg_stream_t stream);
int cseg_bn_group_apply(const cseg_bn_group_member* members, int n, int relu, cseg_stream_t stream);
int cseg_bn_group_bwd(const cseg_bn_group_member* members, int n, int mode, int training, cseg_stream_t stream);
/* SyncBN forms (statistics summed over the ranks between two launches; reference nn.SyncBatchNorm behind
 * lib/models/tools/module_helper.py:35-39). `packed` = fp64 [sum_i (C_i + 1), 2] on the device: member i's rows start at
 * sum_{j<i} (C_j + 1), rows 0 .. C_i - 1 = per-channel pairs, row C_i = (this rank's element count, 0) -- the ONE tensor the host
 * all-reduces per depth and direction.
 *   cseg_bn_group_tiles_moments == cseg_bn_tiles_moments per member (epilogue records -> raw moments) into packed
 *   cseg_bn_group_finalize      == cseg_bn_finalize(count 0) per member from the all-reduced packed moments
 *   cseg_bn_group_bwd_reduce    == cseg_bn_bwd_reduce per member: g_masked (mode 2), d_weight, d_bias, sums into packed (two launches)
 *   cseg_bn_group_bwd_apply     == cseg_bn_bwd_apply_amax(count 0) per member from the all-reduced packed sums: dx, max|dx| */
int cseg_bn_group_tiles_moments(const cseg_bn_group_member* members, int n, double* packed, cseg_stream_t stream);
int cseg_bn_group_finalize(const cseg_bn_group_member* members, int n, const double* packed, cseg_stream_t stream);
int cseg_bn_group_bwd_reduce(const cseg_bn_group_member* members, int n, int mode, double* packed, cseg_stream_t stream);
int cseg_bn_group_bwd_apply(const cseg_bn_group_member* members, int n, int mode, const double* packed, cseg_stream_t stream);
/* The weight gradients of the members' convolutions (3x3 / stride 1 / pad 1, f16x3): == cseg_conv3x3_split_wrw per member (same split
 * counts, same fixed-order reductions: bit-identical), two launches for the group. ws = cseg_conv3x3_sb_wrw_ws_floats(B, Cin, Cout, H, W)
 * floats per member. */
typedef struct cseg_wrw_group_member {
    const float* x;            /* [B, Cin, H, W] the convolution's input */
    const float* dy;           /* [B, Cout, H, W] gradient of its output */
    const unsigned* amax_x;    /* max|x| record */
    const unsigned* amax_dy;   /* max|dy| record */
    float* ws;                 /* scratch */
    float* dw;                 /* [Cout, Cin, 3, 3] */
    int B, Cin, Cout, H, W;
    int reserved[3];
} cseg_wrw_group_member;
int cseg_conv3x3_split_group_wrw(const cseg_wrw_group_member* members, int n, int arith, cseg_stream_t stream);
/* ---- stride 2 (round 3): the 3x3 / stride 2 / pad 1 convolutions of HRNet's fuse and transition layers (reference:
 * lib/models/backbones/hrnet/hrnet_backbone.py:230-250 fuse layers, :652-660 transition layers -- nn.Conv2d(.., 3, 2, 1, bias=False)),
 * f16x3 arithmetic only. x is [B, Cin, 2 Ho, 2 Wo], y / dy are [B, Cout, Ho, Wo].
 *   forward:        cseg_conv3x3_s2_split_fwd, weights packed by cseg_conv3x3_s2_split_pack(transposed = 0) (format CSEG_PACK_C3_16);
 *   backward-data:  cseg_conv3x3_s2_split_bwd, weights packed with transposed = 1 (format CSEG_PACK_C3_S2T): dx [B, Cin, 2 Ho, 2 Wo];
 *   weight gradient: cseg_conv3x3_s2_split_wrw (ws: cseg_conv3x3_s2_wrw_ws_floats floats), deterministic.
 * nt = 16-channel tiles of the OUTPUT of the packed operator per block (3, 4 or 6; must divide its channel count / 16; the same value
 * for pack and run). Shapes: input channels of the operator % 16, output channels % 48 (or % 64 with nt = 4: the 256 input channels of
 * transition 1 in the backward-data direction), Wo % 32 for the weight gradient; 0 / error otherwise. Replaces miopenSp3AsmConv_*_stride2 / _dilation2 and the NHWC implicit-GEMM weight gradient with its layout transposes. */
size_t cseg_conv3x3_s2_split_packed_bytes(int conv_in, int conv_out);
int cseg_conv3x3_s2_split_plan(int conv_in, int conv_out, int transposed, int nt, int* kind, long* threads);
int cseg_conv3x3_s2_split_pack(const float* w, int Cout, int Cin, int transposed, int nt, const unsigned* amax_w, void* wp,
                               cseg_stream_t stream);
int cseg_conv3x3_s2_split_fwd(const float* x, const void* wp, int B, int Cin, int Cout, int Ho, int Wo, int nt,
                              const unsigned* amax_x, const unsigned* amax_w, float* y, cseg_stream_t stream);
int cseg_conv3x3_s2_split_bwd(const float* dy, const void* wp, int B, int Cin, int Cout, int Ho, int Wo, int nt,
                              const unsigned* amax_dy, const unsigned* amax_w, float* dx, cseg_stream_t stream);
size_t cseg_conv3x3_s2_wrw_ws_floats(int B, int Cin, int Cout, int Ho, int Wo);
int cseg_conv3x3_s2_split_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int Ho, int Wo, int arith,
                              const unsigned* amax_x, const unsigned* amax_dy, float* ws, float* dw, cseg_stream_t stream);
/* plan of one pack call, for callers that batch many of them (cseg_split_pack_batch below): kind = CSEG_PACK_* (which packed
 * format the library uses for these channel counts), nt = effective channel tiles per block, threads = one per packed uint4
 * group; conv_in / conv_out are the channel counts of the PACKED operator; returns 0 when the shape is not covered. */
#define CSEG_PACK_C3 0
#define CSEG_PACK_C3_16 1
#define CSEG_PACK_C1 2
#define CSEG_PACK_C3_S2T 3   /* backward-data operator of the stride-2 convolution, see cseg_pack.h */
int cseg_conv3x3_split_plan(int conv_in, int conv_out, int nt_request, int* kind, int* nt, long* threads);
int cseg_conv1x1_split_plan(int conv_in, int conv_out, int* nt, long* threads);
/* Round 5: the tiling depends on the arithmetic (f16x3: 15 / 16 channel tiles per block for multiples of 240 / 256 from 512 on); the
 * call above answers for bf16x6. Pack with the nt this reports (cseg_split_pack_batch) or let cseg_conv1x1_split_pack pick it. */
int cseg_conv1x1_split_plan_arith(int arith, int conv_in, int conv_out, int* nt, long* threads);
/* Batched form: jobs_dev = DEVICE array of n_jobs records sorted by block0 (block0 of job 0 = 0; job i owns blocks
 * [block0_i, block0_{i+1}) of a grid of total_blocks 256-thread blocks).
 *   cseg_amax_batch:       src = float tensor, total = its element count, amax = its (zeroed) max|.| record; any block count >= 1
 *   cseg_split_pack_batch: src = w (the forward's layout), dst = packed buffer, amax = max|w| record (F16X3), cout / cin = the
 *                          forward's channel counts, flag = transpose(_flip), nt / kind / total = the plan (blocks = ceil(total / 256))
 * One launch each, for every layer of the network. */
typedef struct cseg_split_job {
    const void* src;
    void* dst;
    unsigned* amax;
    int cout, cin, flag, nt, kind, total, block0, reserved;
} cseg_split_job;
int cseg_amax_batch(const cseg_split_job* jobs_dev, int n_jobs, int total_blocks, cseg_stream_t stream);
int cseg_split_pack_batch(const cseg_split_job* jobs_dev, int n_jobs, int total_blocks, int arith, cseg_stream_t stream);
size_t cseg_conv1x1_split_packed_bytes(int arith, int Cin, int Cout);
int cseg_conv1x1_split_pack(const float* w, int Cout, int Cin, int transpose, int arith, const unsigned* amax_w, void* wp,
                            cseg_stream_t stream);
int cseg_conv1x1_split_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int HW, int arith,
                           const unsigned* amax_x, const unsigned* amax_w, float* y, cseg_stream_t stream);
/* y = conv1x1(x) + bias + addend: addend [B,Cout,HW] in y's layout, 16-byte aligned (round 6: the input gradient of a residual block's
 * first 1x1 convolution plus the gradient over the skip connection -- reference lib/models/backbones/hrnet/hrnet_backbone.py:68-105 sums
 * them with a separate add). The same kernels as cseg_conv1x1_split_fwd. */
int cseg_conv1x1_split_fwd_add(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int HW,
                               int arith, const unsigned* amax_x, const unsigned* amax_w, float* y, cseg_stream_t stream);
/* ws: cseg_conv1x1_sb_wrw_ws_floats(...) */
int cseg_conv1x1_split_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int HW, int arith, const unsigned* amax_x,
                           const unsigned* amax_dy, float* ws, float* dw, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Round 6: the FIRST convolution of the stem, nn.Conv2d(3, 64, 3, 2, 1, bias=False) on the image (reference
 * lib/models/backbones/hrnet/hrnet_backbone.py:516-517, forward :664): three input channels, so both directions are streams with K = 27 --
 * plain fp32 FMA kernels on NCHW tensors, no packed weights, no max|.| records, fixed summation order (csrc/conv3x3_stem.hip).
 *   x [B,3,H,W] (H, W even), w [64,3,3,3], y / dy [B,64,H/2,W/2], dw [64,3,3,3]; Cout must be 64. No backward-data operator (the image
 *   needs no gradient). ws: cseg_conv3x3_s2_rgb_wrw_ws_floats(B, Cout, H, W) floats (0 = unsupported shape).
 * ------------------------------------------------------------------------------------------------ */
int cseg_conv3x3_s2_rgb_fwd(const float* x, const float* w, int B, int Cout, int H, int W, float* y, cseg_stream_t stream);
size_t cseg_conv3x3_s2_rgb_wrw_ws_floats(int B, int Cout, int H, int W);
int cseg_conv3x3_s2_rgb_wrw(const float* x, const float* dy, int B, int Cout, int H, int W, float* ws, float* dw, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Round 6 (ABI 6): the CLASSIFIER convolution -- a 1x1 convolution onto K <= 32 output channels (num_classes) from a wide activation:
 * the last layer of `cls_head`, reference lib/models/nets/hrnet.py:73-80 (nn.Conv2d(720, num_classes, 1) behind BNReLU +
 * nn.Dropout2d(0.10)) and :113-131 (the OCR classifier), which the reference runs as a library GEMM. Three fp32 streams over the wide
 * tensor (csrc/cls1x1.hip), fixed summation order. The weights are PER IMAGE, transposed and padded:
 *   wt [B][C][KP], KP = 20 or 32 >= K, wt[b][c][k] = w[k][c] * m[b][c] for k < K and 0 for k >= K,
 * m = the channel mask of the Dropout2d in front of the classifier (1 without dropout): the mask is folded into 19 x 720 numbers per
 * image instead of a pass over the 755 MB activation. P = H * W; x / dx [B,C,P], y / dy [B,K,P] (NCHW, contiguous), bias [K] or NULL.
 *   fwd  y[b][k][p]   = bias[k] + sum_c wt[b][c][k] x[b][c][p]
 *   bwd  dx[b][c][p]  = sum_k wt[b][c][k] dy[b][k][p]
 *   wrw  dwt[b][c][k] = sum_p x[b][c][p] dy[b][k][p]       (dwt [B][C][KP]; ws: cseg_cls1x1_wrw_ws_floats(B, C, KP, P) floats, 0 = unsupported)
 * ------------------------------------------------------------------------------------------------ */
int cseg_cls1x1_fwd(const float* x, const float* wt, const float* bias, int B, int C, int K, int KP, long P, float* y,
                    cseg_stream_t stream);
int cseg_cls1x1_bwd(const float* dy, const float* wt, int B, int C, int K, int KP, long P, float* dx, cseg_stream_t stream);
size_t cseg_cls1x1_wrw_ws_floats(int B, int C, int KP, long P);
int cseg_cls1x1_wrw(const float* x, const float* dy, int B, int C, int K, int KP, long P, float* ws, float* dwt, cseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GPU data pipeline: random resize (cv2 INTER_CUBIC image / INTER_NEAREST label) -> random
 * crop -> horizontal flip -> brightness shift -> ToTensor + Normalize(div, mean, std) + label look-up + ReLabel(255,-1)
 * -> collate padding to the fixed input size, as ONE kernel over the output batch.  Replaces the per-sample CPU chain
 * lib/datasets/tools/cv2_aug_transforms.py:143-209, 305-443, 504-603 + lib/datasets/tools/transforms.py:15-103 +
 * lib/datasets/tools/collate.py:37-175.  The random decisions are drawn by the host in the reference's order and passed
 * as CSEG_AUG_PARAM_INTS int32 per image:
 *   [0] Wr [1] Hr resized size; [2] x_off [3] y_off crop origin in the resized image; [4] tw [5] th crop size;
 *   [6] flip (0/1); [7] brightness shift (0 = skipped); [8] left_pad [9] up_pad (collate); [10..11] reserved.
 *   img [B,Hs,Ws,3] u8 (channel order preserved), lab [B,Hs,Ws] u8 or NULL, lut [256] i16 or NULL (raw id -> train id,
 *   255 = ignore), out_img [B,3,Ht,Wt] f32, out_lab [B,Ht,Wt] i64 (255 -> -1) or NULL.
 * ------------------------------------------------------------------------------------------------ */
#define CSEG_AUG_PARAM_INTS 12
int cseg_augment_batch(const uint8_t* img, const uint8_t* lab, const int16_t* lut, const int32_t* params, int B, int Hs,
                       int Ws, int Ht, int Wt, float div_value, const float* mean3, const float* std3, float* out_img,
                       int64_t* out_lab, cseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CSEG_HIP_H */
