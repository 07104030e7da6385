/*
 * sgan_hip.h -- C ABI of libsgan_hip.so: hand-written CDNA4 (gfx950) kernels for the conv
 * generator / discriminator training hot path of phymhan/supervised-gan.
 *
 * The reference has no FFI: its hot path is the stock torch.nn modules built in
 * models/networks.py.  Each entry point below replaces the ATen operator family that one of
 * those modules resolves to; the citation names the reference line that instantiates it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - tensors are NHWC, batch 1, fp32; the channel count `C` passed here is the STORED count,
 *     a multiple of 4 (logical channels are zero-padded up: 1,2,3 -> 4); `*_ld` is the element
 *     stride between pixels (>= C: lets a tensor be a channel slice of a wider concat buffer);
 *   - conv weights ("master layout") are [kh*kw][Cout][Cin] with Cout,Cin the STORED counts, for
 *     Conv2d and ConvTranspose2d alike (the Python side exposes them to state_dict() as strided
 *     views with the reference's logical shapes [Cout,Cin,kh,kw] / [Cin,Cout,kh,kw]);
 *   - per-channel statistics are double[2*C]: sum then sum-of-squares, accumulated with atomics
 *     into a buffer the caller zeroed;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - no entry point allocates, frees or synchronises: safe inside hipGraph capture;
 *   - return value: 0 = ok, <0 = error (see sgan_last_error()); nothing throws across the ABI.
 */
#ifndef SGAN_HIP_H
#define SGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGAN_OK 0
#define SGAN_ERR_INVALID (-1)
#define SGAN_ERR_UNSUPPORTED (-2)
#define SGAN_ERR_HIP (-3)

/* activation codes */
#define SGAN_ACT_NONE 0
#define SGAN_ACT_RELU 1
#define SGAN_ACT_LRELU 2
#define SGAN_ACT_TANH 3

/* conv kinds */
#define SGAN_CONV 0  /* nn.Conv2d            */
#define SGAN_CONVT 1 /* nn.ConvTranspose2d   */

/* Per-channel normalisation + activation applied to a tensor as it is READ by a kernel
 * ("normalise-on-load").  y = act( gamma * (x - mean) * rstd + beta ), mean/rstd derived in-kernel
 * from `stats` (biased variance, eps inside the sqrt).  stats == NULL: no normalisation, only act.
 * Replaces nn.InstanceNorm2d(affine=False) (models/networks.py:46-47), nn.BatchNorm2d in train
 * mode with batch 1 (models/networks.py:87,507,517), nn.ReLU / nn.LeakyReLU(0.2)
 * (models/networks.py:508,525,816,826,833). */
typedef struct sgan_norm_desc {
    const double* stats; /* [2*C] sum, sumsq of the tensor being read, or NULL */
    const float* gamma;  /* [C] or NULL (=1) */
    const float* beta;   /* [C] or NULL (=0) */
    int32_t count;       /* H*W of the tensor being read (statistics population) */
    float eps;
    int32_t act;         /* SGAN_ACT_NONE / RELU / LRELU */
    float slope;         /* LeakyReLU slope */
    int32_t sq_stride;   /* distance (in doubles) from sum[c] to sumsq[c]; 0 = C.  Lets `stats` point into the
                          * statistics of a wider concat buffer (U-Net skip: [up half | skip half]) */
    int32_t rep_stride;  /* 0: one copy of the statistics.  > 0: SGAN_STAT_REPLICAS copies, `rep_stride` doubles apart, whose SUM
                          * is the statistic: the producers spread their same-address fp64 atomics over the copies (workgroup b adds
                          * to copy b % SGAN_STAT_REPLICAS), every reader adds the copies up.  All copies start at zero. */
} sgan_norm_desc;
#define SGAN_STAT_REPLICAS 8

/* Geometry of one Conv2d / ConvTranspose2d layer (square kernel, symmetric stride/pad). */
typedef struct sgan_conv_desc {
    int32_t kind;   /* SGAN_CONV or SGAN_CONVT */
    int32_t k, stride, pad;
    int32_t Hin, Win, Cin;    /* forward input  (stored channels) */
    int32_t Hout, Wout, Cout; /* forward output (stored channels) */
    /* Optional hint, 0 = unknown: how many of the stored channels carry data (the rest is the zero padding to a
     * multiple of 4 -- a 1-channel logits map or a 2-channel image is stored with 4).  Kernels may skip arithmetic on
     * channels beyond these counts: the padding channels of a result are then written as if their weights were zero
     * (which they are in every buffer the host mirror builds). */
    int32_t Cin_logical, Cout_logical;
    /* Arithmetic of the MFMA kernels (the dtype enum of the boundary).  Storage is fp32 in every mode.
     *   SGAN_MATH_F32    : v_mfma_f32_16x16x4_f32, bit-for-bit an fp32 fma chain (the parity mode);
     *   SGAN_MATH_BF16X3 : split 16-bit planes -- every fp32 operand x is cut into hi = rne16(x), lo = rne16(x - hi) and a
     *                      product is a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulation,
     *                      at 16/3 of the fp32 matrix rate.  The forward pass uses fp16 planes (11 + 11 bits; the packed
     *                      weights hold w * 2^10): measured 2e-7 .. 1e-6 per layer against fp64, the same as SGAN_MATH_F32.
     *                      Backward-data and backward-weight use bf16 planes (8 + 8 bits, the fp32 exponent range of
     *                      gradients): ~5e-6 per layer, inside the 1e-3 contract.  Needs the job's `w_packed` copy of the
     *                      weights (sgan_pack_weights) and fails without it; layers the split kernels do not cover (stored
     *                      channels not a multiple of 8 or under 16, maps under 256 pixels, 4-channel heads) run the fp32
     *                      kernels whatever this field says. */
    int32_t math;
} sgan_conv_desc;
#define SGAN_MATH_F32 0
#define SGAN_MATH_BF16X3 1

const char* sgan_version(void);
const char* sgan_last_error(void);
/* name of the kernel template instantiation the calling thread's last sgan_conv_* call launched
 * (matches the name rocprofv3 reports) -- lets a benchmark attribute time and flops per kernel */
const char* sgan_last_kernel(void);
int sgan_stat_replicas(void);     /* SGAN_STAT_REPLICAS of the built library: statistics arenas must hold this many copies */
/* The explicit device of the boundary (SURVEY 8b "Threading": backward is called from autograd worker threads that did not
 * choose a device).  Launches go to the device their stream belongs to and HIP wants the calling thread's current device to be
 * that one: a host thread that has not set it calls sgan_set_device(sgan_stream_device(stream)) once before its first entry
 * point.  (torch's autograd engine does this for its own workers; the Python binding therefore never calls these.) */
int sgan_set_device(int32_t device_id);
int sgan_stream_device(void* stream, int32_t* device_id);

/* ---- optional per-launch timing (diagnostics; single-threaded; do not enable during graph capture) ----
 * While enabled, every main conv kernel launch (implicit-GEMM / small-N / backward-weight; not the split-K
 * epilogue) is bracketed by two HIP events recorded on the launch stream from inside the library.  After a
 * device synchronise, record i (0 <= i < sgan_profile_count(), launch order) yields the kernel's name as
 * rocprofv3 reports it and its duration.  sgan_profile_enable() also clears the records. */
int sgan_profile_enable(int on);
int sgan_profile_count(void);
int sgan_profile_mark(void* stream); /* records an empty bracket named "null": the events' own cost, to subtract */
int sgan_profile_read(int i, const char** name, float* ms);

/* ---- Conv2d / ConvTranspose2d: forward ------------------------------------------------------
 * out = conv(act(norm(in)), w) + bias ; optionally out = tanh(out) ; optionally accumulates the
 * per-channel sum / sumsq of the (pre-tanh) result into out_stats.
 * Replaces: nn.ConvTranspose2d k4 s2 p1 (models/networks.py:502,516,523,529),
 *           nn.Conv2d k4 s2 p2 / k4 s1 p2 (models/networks.py:815,824,831,835),
 *           nn.Conv2d k4 s2 p1 (models/networks.py:356,385), nn.Conv2d k3 s1 p1 (:686,752,774),
 *           nn.Tanh (models/networks.py:540).
 *
 * Workspace (forward and backward-data): deep reductions on small grids are split over K; the fp32
 * partial tiles live in a caller-owned scratch buffer.  Call with workspace_bytes == -1 (nothing is
 * launched) to get the recommended size in KiB as the return value (0 = none); passing NULL / a smaller
 * buffer is allowed and simply disables the split. */
int sgan_conv_fwd(const sgan_conv_desc* d, const float* in, int32_t in_ld, const sgan_norm_desc* in_norm,
                  const float* w, const float* bias, float* out, int32_t out_ld, int32_t out_act,
                  double* out_stats, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- backward-data ---------------------------------------------------------------------------
 * din = conv_bwd_data(dout, w), then multiplied by act'(norm(x)) of the forward tensor `x` at the
 * same positions (x_norm describes how the forward consumer read x; x == NULL: plain dgrad).
 * With x_norm->stats != NULL it also accumulates the two InstanceNorm/BatchNorm backward sums
 * bwd_sums[0..C) = sum(dY), bwd_sums[C..2C) = sum(dY * xhat) (caller-zeroed).  The result is dY
 * (gradient w.r.t. the normalised-affine value); sgan_norm_bwd_apply() turns it into dX.
 * Replaces: convolution_backward (input grad) + LeakyReLU/ReLU backward of the autograd graph
 * built by FCGANModel.backward_D / backward_G (models/fcgan_model.py:146-176). */
int sgan_conv_dgrad(const sgan_conv_desc* d, const float* dout, int32_t dout_ld, const float* w,
                    float* din, int32_t din_ld, const float* x, int32_t x_ld, const sgan_norm_desc* x_norm,
                    double* bwd_sums, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- grouped launches -----------------------------------------------------------------------------
 * Up to 8 independent problems of the SAME layer type (kind, k, stride, pad, stored channels) in ONE kernel
 * launch, each with its own tensors, weights and spatial size -- the matching layer of the three
 * multi-scale discriminators on the fake and on the real batch (models/fcgan_model.py:150-159) is one
 * launch instead of six.  Semantics per job are exactly those of the single-problem entry points (which
 * are the n == 1 case).  All jobs must agree on the activation codes of their norm descriptors. */
typedef struct sgan_conv_fwd_job {
    const sgan_conv_desc* d;
    const float* in; int32_t in_ld; const sgan_norm_desc* in_norm;
    const float* w; const float* bias;
    float* out; int32_t out_ld;
    double* out_stats;
    int32_t out_stats_sq_stride;  /* distance from sum[n] to sumsq[n] in out_stats; 0 = Cout */
    const void* w_packed;         /* SGAN_MATH_BF16X3: the `packed_fwd` copy of `w` (sgan_pack_weights), same element offset; else NULL */
    int32_t out_stats_rep_stride; /* > 0: out_stats is the first of SGAN_STAT_REPLICAS copies this far apart (sgan_norm_desc.rep_stride) */
} sgan_conv_fwd_job;
typedef struct sgan_conv_dgrad_job {
    const sgan_conv_desc* d;
    const float* dout; int32_t dout_ld;
    const float* w;
    float* din; int32_t din_ld;
    const float* x; int32_t x_ld; const sgan_norm_desc* x_norm;
    double* bwd_sums;
    int32_t bwd_sums_sq_stride;   /* distance from s1[n] to s2[n] in bwd_sums; 0 = Cin */
    int32_t accumulate;           /* 1: din += result (a tensor with two forward consumers, e.g. a U-Net skip) */
    int32_t w_transposed;         /* 1: `w` is the transposed master copy [kh*kw][Cin_s][Cout_s] (sgan_transpose_weights): the
                                   * reduction channel (Cout) is then contiguous and backward-data stages its weights with
                                   * 16-byte LDS stores like the forward pass */
    const void* w_packed;         /* SGAN_MATH_BF16X3: the `packed_bwd` copy of the weights (sgan_pack_weights); else NULL */
    int32_t bwd_sums_rep_stride;  /* > 0: bwd_sums is the first of SGAN_STAT_REPLICAS copies this far apart */
    const float* dout_amax;       /* device scalar max|dout| (sgan_norm_bwd_apply_multi publishes it) or NULL.  With it (and w_packed_f16)
                                   * SGAN_MATH_BF16X3 runs backward-data on fp16 planes of dout * 2^s, s from the exponent of the maximum:
                                   * 11 + 11 significant bits, an fp32-equivalent product like the forward pass.  Without: bf16 planes (8 + 8) */
    const void* w_packed_f16;     /* the `packed_bwd_f16` copy of the weights (sgan_pack_weights); NULL: bf16 planes */
} sgan_conv_dgrad_job;
typedef struct sgan_conv_wgrad_job {
    const sgan_conv_desc* d;
    const float* in; int32_t in_ld; const sgan_norm_desc* in_norm;
    const float* dout; int32_t dout_ld;
    float* dw; float* dbias;
    const float* dout_amax;       /* as sgan_conv_dgrad_job.dout_amax: with it backward-weight runs on fp16 planes (dout scaled), else bf16 */
} sgan_conv_wgrad_job;
int sgan_conv_fwd_grouped(const sgan_conv_fwd_job* jobs, int32_t n, int32_t out_act, void* workspace,
                          int64_t workspace_bytes, void* stream);
int sgan_conv_dgrad_grouped(const sgan_conv_dgrad_job* jobs, int32_t n, void* workspace, int64_t workspace_bytes,
                            void* stream);
int sgan_conv_wgrad_grouped(const sgan_conv_wgrad_job* jobs, int32_t n, void* workspace, int64_t workspace_bytes,
                            void* stream);
/* One launch for a layer's backward-data AND backward-weight (split-bf16 mode): the two halves of aten::convolution_backward
 * the reference's autograd runs for every conv in netD / netG backward (models/networks.py:502-529, 815-835 are the layers;
 * models/fcgan_model.py:131-152 the backward calls).  Same jobs as the two grouped entry points above, same results; the
 * workgroups of both share one grid so the chip is not left half idle twice.  Returns SGAN_OK when launched, 1 when this layer
 * is not covered by the fused kernel (the caller then issues the two grouped calls; nothing was written), < 0 on error.
 * dgrad_math: SGAN_MATH_* of the backward-data half, or -1 for the descriptors' own (the two job lists usually point at the
 * same descriptors, and backward-data into a layer without a normalisation runs exact fp32 next to a split-bf16
 * backward-weight: DESIGN.md R2.3); the backward-weight half must be SGAN_MATH_BF16X3. */
int sgan_conv_bwd_fused(const sgan_conv_dgrad_job* djobs, int32_t nd, const sgan_conv_wgrad_job* wjobs, int32_t nw,
                        int32_t dgrad_math, void* stream);
/* The same with a workspace: a pair whose backward-data half is a deep reduction on a small map (generator 256 -> 128 at 32 x 32, the
 * inner U-Net levels) keeps its split-K inside the fused launch -- the partial tiles go to `workspace`, a second kernel sums them and
 * runs the backward-data epilogue.  workspace_bytes == -1: query, returns the KiB the pair wants (0: none) and launches nothing.
 * Without enough workspace such a pair answers 1 (the caller issues the two grouped calls), as sgan_conv_bwd_fused does. */
int sgan_conv_bwd_fused_ws(const sgan_conv_dgrad_job* djobs, int32_t nd, const sgan_conv_wgrad_job* wjobs, int32_t nw,
                           int32_t dgrad_math, void* workspace, int64_t workspace_bytes, void* stream);

/* A layer with a 4-channel side (the generator's output ConvTranspose2d(ngf, 2), models/networks.py:528-533; the first PatchGAN conv
 * Conv2d(2, ndf) in a generator update, :815-817): its backward-data and backward-weight launches in ONE grid
 * (sg_bwd_thin_pair_kernel).  Same job lists as sgan_conv_dgrad_grouped / sgan_conv_wgrad_grouped, same results.  Returns 1 when the
 * pair is not one of the two shapes (the caller issues the two grouped calls). */
int sgan_conv_bwd_thin_pair(const sgan_conv_dgrad_job* djobs, int32_t nd, const sgan_conv_wgrad_job* wjobs, int32_t nw, void* stream);

/* ---- class-weighted cross-entropy on logits, softmax over channels (NHWC maps, <= 16 classes) ---------------------------
 * sgan_ce_fwd:  loss = sum_p w[y_p] (logsumexp(z_p) - z_p[y_p]) / sum_p w[y_p];  y_p = label[p] (int64 map) or const_label when label
 *               is NULL; class_w NULL = unit weights; labels outside [0, C) are skipped (torch's ignore_index).  `acc` (2 doubles) and
 *               `ticket` (1 uint32) are caller-owned scratch that must be ZERO on entry; acc keeps the two sums for sgan_ce_bwd.
 * sgan_ce_bwd:  dlogits = gout[0] * w[y_p] * (softmax(z_p) - onehot(y_p)) / sum w   (padding channels of dlogits: zeros).
 * Replaces: nn.CrossEntropyLoss in GANLossMultiClass (models/networks.py:188-202), CrossEntropyLoss2d = NLLLoss2d(log_softmax)
 * (models/loss.py:6-12) of the segmentation trainers.
 * sgan_softmax_fwd / _bwd: p = softmax over the C logical channels; dz = p * (dp - sum_c dp_c p_c).  Replaces F.softmax(logit, 1)
 * (models/segm_model.py:155-160). */
int sgan_ce_fwd(const float* logits, int32_t ld, int32_t npix, int32_t C, const int64_t* label, int32_t const_label,
                const float* class_w, double* acc, uint32_t* ticket, float* loss_out, void* stream);
int sgan_ce_bwd(const float* logits, int32_t ld, int32_t npix, int32_t C, const int64_t* label, int32_t const_label,
                const float* class_w, const double* acc, const float* gout, float* dlogits, int32_t dld, void* stream);
int sgan_softmax_fwd(const float* z, int32_t ld, int32_t npix, int32_t C, float* p, int32_t pld, void* stream);
int sgan_softmax_bwd(const float* dp, int32_t dpld, const float* p, int32_t pld, int32_t npix, int32_t C, float* dz, int32_t dzld,
                     void* stream);

/* ---- backward pass of a one-channel stride-1 head (the PatchGAN logits conv, models/networks.py:832-835) in one launch: the job
 * lists of sgan_conv_dgrad_grouped and sgan_conv_wgrad_grouped for the SAME pass (wjobs may be NULL: input gradient only).  Returns 1
 * when the layer is not of that type (Conv2d, stride 1, k <= 4, stored Cout 4 / logical 1, Cin >= 64, no `accumulate`): the caller
 * then issues the two generic calls.  Exact fp32. */
int sgan_conv_head_bwd(const sgan_conv_dgrad_job* djobs, const sgan_conv_wgrad_job* wjobs, int32_t n, void* stream);

/* ---- transposed weight copy for backward-data ---------------------------------------------------
 * flat_t[off + tap][ci][co] = flat[off + tap][co][ci] for every conv segment (bias / affine ranges of the flat
 * parameter buffer are not touched).  Run after each optimizer step on the nets whose backward-data is needed. */
typedef struct sgan_wt_seg { int64_t off; int32_t taps, cout, cin; } sgan_wt_seg;
int sgan_transpose_weights(const float* flat, float* flat_t, const sgan_wt_seg* segs, int32_t n /* <= 64 */, void* stream);

/* ---- packed weight copies (the reference keeps one weight tensor per layer, models/networks.py:502-529,815-835; the
 * kernels read it in three derived forms) ------------------------------------------------------------------------
 * For every conv segment of the flat parameter buffer (same element offsets in every copy):
 *   flat_t     [tap][ci][co] fp32                          transposed master copy (as sgan_transpose_weights), or NULL
 *   packed_fwd [tap][co][ci/8]{8 x bf16 hi | 8 x bf16 lo}  split-bf16 rows for the forward pass   (cin  % 8 == 0), or NULL
 *   packed_bwd [tap][ci][co/8]{8 x bf16 hi | 8 x bf16 lo}  split-bf16 rows for backward-data      (cout % 8 == 0), or NULL
 *   packed_bwd_f16: the same rows as fp16 planes of w * 2^10 (like packed_fwd, which holds fp16 planes of w * 2^10 too): what
 *                   backward-data reads when the gradient's maximum is known (sgan_conv_dgrad_job.dout_amax), or NULL
 * hi = bf16(x) (round to nearest even), lo = bf16(x - hi).  A copy occupies 4 bytes per weight like the master.  Segments
 * whose channel count does not divide are left untouched in that copy (such layers run the fp32 kernels).  Run after each
 * optimizer step / checkpoint load, before the next forward. */
int sgan_pack_weights(const float* flat, float* flat_t, void* packed_fwd, void* packed_bwd, void* packed_bwd_f16,
                      const sgan_wt_seg* segs, int32_t n /* <= 64 */, void* stream);

/* ---- backward-weight ---------------------------------------------------------------------------
 * dw += act(norm(in))^T (x) dout over all pixels (master layout), dbias += sum_pixels dout.
 * Accumulates (atomics) into caller-owned gradient buffers.
 * `workspace` (same protocol as sgan_conv_fwd: workspace_bytes == -1 returns the KiB needed) lets the thin-layer
 * kernel (stored Cin == 4 or Cout == 4) combine its pixel splits in two stages instead of with contended atomics;
 * optional -- without it the result is the same, only slower.
 * Replaces: convolution_backward (weight / bias grad). */
int sgan_conv_wgrad(const sgan_conv_desc* d, const float* in, int32_t in_ld, const sgan_norm_desc* in_norm,
                    const float* dout, int32_t dout_ld, float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
                    void* stream);

/* ---- InstanceNorm / BatchNorm(batch 1) backward, in place ------------------------------------
 * dy[p][c] <- gamma_c * rstd_c * ( dy - s1_c/M - xhat * s2_c/M ), xhat = (x - mean_c) * rstd_c.
 * dgamma += s2, dbeta += s1 when non-NULL.  M = npix.
 * Replaces: native_batch_norm_backward of nn.InstanceNorm2d / nn.BatchNorm2d. */
int sgan_norm_bwd_apply(float* dy, int32_t dy_ld, const float* x, int32_t x_ld, int32_t npix, int32_t C,
                        const sgan_norm_desc* x_norm, const double* bwd_sums, int32_t bwd_sums_sq_stride /* 0 = C */,
                        float* dgamma, float* dbeta, void* stream);

/* Several independent tensors in one launch (the matching layer of grouped discriminator chains). */
typedef struct sgan_norm_bwd_job {
    float* dy; int32_t dy_ld;
    const float* x; int32_t x_ld;
    int32_t npix, C;
    const sgan_norm_desc* x_norm;
    const double* bwd_sums; int32_t bwd_sums_sq_stride;
    float* dgamma; float* dbeta;
    int32_t bwd_sums_rep_stride;   /* > 0: bwd_sums is the first of SGAN_STAT_REPLICAS copies this far apart (summed on read) */
    float* amax_out;               /* NULL, or a device scalar that starts at 0: max|dy| of the result is stored there (an atomic max on the
                                    * bit pattern) -- the `dout_amax` of the backward-data / backward-weight calls that read dy next */
} sgan_norm_bwd_job;
int sgan_norm_bwd_apply_multi(const sgan_norm_bwd_job* jobs, int32_t n /* 1..8 */, void* stream);

/* ---- BatchNorm running statistics (momentum update, unbiased variance), n layers per launch --
 * Replaces the running_mean / running_var side effect of nn.BatchNorm2d.forward in train mode. */
typedef struct sgan_bn_running_desc {
    const double* stats;
    float* running_mean;
    float* running_var;
    int64_t* num_batches_tracked; /* device int64 incremented by one, or NULL */
    int32_t C;
    int32_t count;     /* H*W */
    int32_t sq_stride; /* distance from a channel's sum to its sum of squares inside `stats`; 0 = C (wider when `stats`
                          is a slice of a concatenated tensor's statistics, or C is not a multiple of 4) */
    int32_t rep_stride; /* > 0: `stats` is the first of SGAN_STAT_REPLICAS copies this far apart (summed on read) */
} sgan_bn_running_desc;
int sgan_bn_running_update(const sgan_bn_running_desc* layers, int32_t n, float momentum, void* stream);

/* ---- Gaussian pre-filter + stride pick of the multi-scale discriminators ----------------------
 * out[y][x][c] = sum_{ky,kx} g[c][ky][kx] * in[y*s + ky - pad][x*s + kx - pad][c], where g is read
 * from the diagonal of the reference's dense [C,C,k,k] weight (w_diag_stride = element stride
 * between g[c] and g[c+1]).  Only strided outputs and diagonal channels are computed.
 * Replaces: gauss_filter = Conv2d(C,C,4*sigma+1,pad 2*sigma) + AvgPool2d(1, stride s)
 * (models/networks.py:807-813). */
int sgan_gauss_down_fwd(const float* in, int32_t in_ld, int32_t H, int32_t W, int32_t C, int32_t Creal,
                        const float* g, int32_t g_chan_stride, int32_t k, int32_t pad, int32_t s,
                        float* out, int32_t out_ld, int32_t Ho, int32_t Wo, void* stream);
int sgan_gauss_down_bwd(const float* dout, int32_t dout_ld, int32_t Ho, int32_t Wo, int32_t C, int32_t Creal,
                        const float* g, int32_t g_chan_stride, int32_t k, int32_t pad, int32_t s,
                        float* din, int32_t din_ld, int32_t H, int32_t W, int32_t accumulate /* din += instead of = */,
                        void* stream);

/* Several pre-filters in one launch (the multi-scale discriminators filter one image at scale 2 and at scale 4;
 * models/fcgan_model.py:86-93 builds one NLayerDiscriminator per --scale_factor entry).  Forward: n <= 4 independent jobs.
 * Backward: the jobs must share `image` (the image gradient): it receives the SUM of their contributions
 * (accumulate != 0: added to what it holds). */
typedef struct sgan_gauss_job {
    float* image; int32_t image_ld, H, W;    /* full-resolution side: read by fwd, written by bwd */
    float* down; int32_t down_ld, Ho, Wo;    /* filtered + strided side: written by fwd, read by bwd */
    const float* g; int32_t g_chan_stride, k, pad, s;
} sgan_gauss_job;
int sgan_gauss_down_multi_fwd(const sgan_gauss_job* jobs, int32_t n, int32_t C, int32_t Creal, void* stream);
int sgan_gauss_down_multi_bwd(const sgan_gauss_job* jobs, int32_t n, int32_t C, int32_t Creal, int32_t accumulate, void* stream);

/* ---- GAN loss on a logits map (channel 0 of an NHWC-4 tensor) --------------------------------
 * mode 0: BCE(sigmoid(x), t) with torch's log clamp at -100 (GANLoss, --no_lsgan);
 * mode 1: MSE(x, t) (lsgan, no sigmoid).  loss_out[0] = mean loss; p_out (optional) = sigmoid(x).
 * The backward writes dlogits[p][0] = gout[0] * dloss/dx, other stored channels 0.
 * Replaces: nn.Sigmoid (models/networks.py:836-837) + nn.BCELoss / nn.MSELoss inside GANLoss
 * (models/networks.py:160-163,183-185). */
int sgan_gan_loss_fwd(const float* logits, int32_t ld, int32_t npix, float target, int32_t mode,
                      float* loss_out, float* p_out, void* stream);
int sgan_gan_loss_bwd(const float* logits, int32_t ld, int32_t npix, float target, int32_t mode,
                      const float* gout, float* dlogits, int32_t dld, void* stream);

/* ---- every GAN-loss term of one backward pass at once ----------------------------------------------
 * total = sum_i weight_i * loss_i over up to 8 logits maps, loss_i as sgan_gan_loss_fwd computes it;
 * each_out[i] = loss_i (for logging).  A forward job with a `dlogits` buffer also gets dlogits_i = weight_i * dloss_i/dx, the
 * gradient for an upstream gradient of 1 (the trainers call backward() on `total` itself); the backward entry point writes
 * dlogits_i = gout * weight_i * dloss_i/dx for any other upstream gradient.
 * Replaces the per-discriminator GANLoss calls plus the scalar arithmetic the trainers wrap around them:
 * (loss_D_fake + loss_D_real) * 0.5 and sum_i lambda_i * loss_i (models/fcgan_model.py:150-176). */
typedef struct sgan_gan_loss_job {
    const float* logits; int32_t ld; int32_t npix;
    float target; float weight;
    float* dlogits; int32_t dld;   /* forward: optional (unit-gradient result); backward: required */
} sgan_gan_loss_job;
/* `workspace`: SGAN_GAN_LOSS_WS_BYTES of 8-byte aligned device scratch, ZERO on first use and owned by one stream at a time: the
 * terms are reduced by several workgroups each, the last one to arrive (a ticket counter at the end of the scratch) finishes
 * and leaves the counter at zero again -- one launch, no fill per call. */
#define SGAN_GAN_LOSS_WS_BYTES 2048
int sgan_gan_loss_multi_fwd(const sgan_gan_loss_job* jobs, int32_t n, int32_t mode, float* each_out, float* total_out,
                            void* workspace, int64_t workspace_bytes, void* stream);
int sgan_gan_loss_multi_bwd(const sgan_gan_loss_job* jobs, int32_t n, int32_t mode, const float* gout, void* stream);

/* ---- standalone nn.Sigmoid on channel 0 of a logits map (models/networks.py:836-837) --------
 * Only needed when a caller wants the probability map itself; the GAN loss above consumes logits. */
int sgan_sigmoid_fwd(const float* x, int32_t ld, int32_t npix, float* p, int32_t pld, void* stream);
int sgan_sigmoid_bwd(const float* dp, int32_t dpld, const float* p, int32_t pld, int32_t npix, float* dx,
                     int32_t dxld, void* stream);

/* ---- U-Net up path: normalise (+ dropout) (+ additive Gaussian noise) into a concat slice --------------
 * t[p][c] = ((u[p][c] - mean_c) * rstd_c * gamma_c + beta_c) * (mask ? mask[p][c] : 1) + (noise ? sigma * noise[p][c] : 0)
 * (gamma = 1, beta = 0 without an affine: InstanceNorm; with one: norm_layer = BatchNorm2d -> nn.Dropout, networks.py:516-521)
 * mask holds 0 or 1/(1-p) (nn.Dropout(0.5) => 0 or 2).  Replaces norm_layer + nn.Dropout + the `y + noise` of
 * UnetSkipConnectionBlock (models/networks.py:387-403,414-419); `t` is written straight into the up half
 * of the concat buffer the next ConvTranspose2d reads (torch.cat, :417-419, never materialises).
 * Backward, first pass: dt <- dt * mask in place and bwd_sums += (sum dt, sum dt * uhat); then
 * sgan_norm_bwd_apply(dt, u, ...) finishes the normalisation backward. */
int sgan_norm_apply_fwd(const float* u, int32_t u_ld, const sgan_norm_desc* u_norm, const float* mask, const float* noise,
                        float sigma, float* t, int32_t t_ld, int32_t npix, int32_t C, void* stream);
int sgan_norm_apply_bwd_sums(float* dt, int32_t dt_ld, const float* mask, const float* u, int32_t u_ld,
                             const sgan_norm_desc* u_norm, double* bwd_sums, int32_t npix, int32_t C, void* stream);
/* mask[i] = uniform(Philox(seed, *offset_dev + i)) < p ? 0 : 1/(1-p); advances *offset_dev (nn.Dropout). */
int sgan_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, uint64_t* offset_dev, int32_t advance, void* stream);

/* ---- nn.ReflectionPad2d in front of a conv (resnet generators: models/networks.py:232,258 ReflectionPad2d(3) + Conv k7;
 * :282-300 ReflectionPad2d(1) + Conv k3 inside every ResnetBlock) -------------------------------------------------------
 * out [H + 2 pad][W + 2 pad][C] = mask * act(norm(x)) gathered at the reflected positions (what the reference normalises,
 * activates and drops out BEFORE it pads); the conv that follows takes `out` with pad 0 and no in_norm.  pad == 0 just
 * materialises mask * act(norm(x)).  `mask` is dense [H][W][C] (nn.Dropout: 0 or 1 / (1 - p)) or NULL.
 * Backward: din [H][W][C] = act'(norm(x)) * mask * (sum of dout over the <= 4 padded positions that mirror the pixel);
 * with x_norm->stats it also accumulates bwd_sums (sum din, sum din * xhat) for sgan_norm_bwd_apply, like sgan_conv_dgrad.
 * x == NULL: a plain fold (the padded tensor was the input itself). */
int sgan_pad_reflect_fwd(const float* x, int32_t x_ld, int32_t H, int32_t W, int32_t C, const sgan_norm_desc* x_norm,
                         const float* mask, int32_t pad, float* out, int32_t out_ld, void* stream);
int sgan_pad_reflect_bwd(const float* dout, int32_t dout_ld, int32_t H, int32_t W, int32_t C, int32_t pad, const float* x,
                         int32_t x_ld, const sgan_norm_desc* x_norm, const float* mask, float* din, int32_t din_ld,
                         double* bwd_sums, int32_t bwd_sums_sq_stride, void* stream);

/* ---- BCELoss on rescaled tanh outputs (two-stage trainers): loss = mean BCE((x + 1) / 2, (t + 1) / 2) over npix * C
 * with torch's -100 log clamp; g = dloss/dx for a unit upstream gradient (backward: dx = gout * g, sgan_scale).
 * Replaces: torch.nn.BCELoss()((x + 1) / 2, (t + 1) / 2) at models/twostage_cycle_model.py:398-403. */
/* `workspace` (this function and sgan_l1w_fwd): SGAN_IMAGE_LOSS_WS_BYTES of 8-byte aligned device scratch (uninitialised is
 * fine) for the per-workgroup partial sums a second kernel finishes. */
#define SGAN_IMAGE_LOSS_WS_BYTES 2048
int sgan_bce01_fwd(const float* x, int32_t x_ld, const float* t, int32_t t_ld, int32_t npix, int32_t C, float* loss_out,
                   float* g, int32_t g_ld, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- CRN building blocks (models/networks.py:642-794) --------------------------------------------
 * sgan_bilinear_up2_fwd: nn.Upsample(scale_factor=2, mode='bilinear') (align_corners = False) of an [H, W, C] tensor
 * into [2H, 2W, C], accumulating the (sum, sumsq) statistics of the result for the InstanceNorm that follows it
 * (CrnUpsampleBlock :741-746).  sgan_bilinear_up2_bwd is its adjoint (din [H, W, C] from dout [2H, 2W, C]).
 * sgan_avgpool_pyramid_fwd: the six label maps AvgPool2d(2^(s+1)) (label), s = 0..5, of
 * CascadedRefinementNetwork.forward (:709-731) in one launch (4 stored channels; H, W divisible by 64);
 * sgan_avgpool_pyramid_bwd: dlabel (+)= sum_s upsample(dlevels[s]) / 4^(s+1) (NULL levels are skipped). */
int sgan_bilinear_up2_fwd(const float* in, int32_t in_ld, int32_t H, int32_t W, int32_t C, float* out, int32_t out_ld,
                          double* out_stats, int32_t out_stats_sq_stride, void* stream);
int sgan_bilinear_up2_bwd(const float* dout, int32_t dout_ld, int32_t H, int32_t W, int32_t C, float* din, int32_t din_ld,
                          void* stream);
int sgan_avgpool_pyramid_fwd(const float* label, int32_t ld, int32_t H, int32_t W, float* const* levels,
                             const int32_t* level_ld, void* stream);
int sgan_avgpool_pyramid_bwd(const float* const* dlevels, const int32_t* level_ld, int32_t H, int32_t W, float* dlabel,
                             int32_t ld, int32_t accumulate, void* stream);

/* ---- weighted L1 (cgan): loss = lambda * mean(|x - y| * w), w = 1 + sum_i ((a_i + 1)/2) * (weights_i - 1) over the
 * first nweights channels of the label image `a` (w = 1 when a == NULL; with nweights == 0, `a` is the weight map
 * itself, one value per pixel).  Also writes g = dloss/dx (unscaled by
 * the incoming gradient); the backward is dx = gout * g.  Replaces WeightedL1Loss (models/networks.py:205-214)
 * and the weight-map construction in CGANModel.backward_G (models/cgan_model.py:196-207). */
int sgan_l1w_fwd(const float* x, int32_t x_ld, const float* y, int32_t y_ld, int32_t npix, int32_t C,
                 const float* a, int32_t a_ld, const float* weights_dev, int32_t nweights, float lambda,
                 float* loss_out, float* g, int32_t g_ld, void* workspace, int64_t workspace_bytes, void* stream);
int sgan_scale(const float* gout, const float* g, float* dx, int64_t n, void* stream);   /* dx = gout[0] * g */

/* ---- elementwise helpers ---------------------------------------------------------------------
 * tanh backward: dx = dy * (1 - y*y)                                   (nn.Tanh, networks.py:540)
 * strided gather into NHWC with zero channel padding (layout boundary of the module API).  `src` of sgan_to_nhwc may also
 * be PINNED (page-locked, device-mapped) HOST memory: the gather kernel is then the host-to-device copy of the batch
 * (the transforms.ToTensor() -> .cuda() step of the reference's loop, train.py:25-29), queued with the step's kernels. */
int sgan_tanh_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* out = act(a + b) over n contiguous floats (n % 4 == 0), act = SGAN_ACT_TANH or SGAN_ACT_NONE: `nn.Tanh()(x + y)`, the
 * --use_residual tail of ResnetGenerator.forward / UnetGenerator.forward (models/networks.py:268, :367).  Backward:
 * sgan_tanh_bwd(dout, out) is the gradient of both addends. */
int sgan_add_act_fwd(const float* a, const float* b, float* out, int64_t n, int32_t act, void* stream);
int sgan_to_nhwc(const float* src, int64_t sc, int64_t sh, int64_t sw, int32_t H, int32_t W, int32_t Creal,
                 float* dst, int32_t dst_ld, int32_t Cstore, void* stream);

/* ---- the (label, image) pair of the conditional discriminators -------------------------------------
 * torch.cat((real_A, fake_B), 1) (models/cgan_model.py:162,172,187; twostage_cycle_model.py) of two NHWC buffers, written as the
 * padded NHWC buffer the discriminator reads (channels Ca + Cb .. Cstore-1 zero), and its backward: channels [c0, c0 + C) of the
 * pair's gradient as a padded NHWC buffer of their own. */
int sgan_concat_nhwc(const float* a, int32_t a_ld, int32_t Ca, const float* b, int32_t b_ld, int32_t Cb, int64_t npix,
                     float* dst, int32_t dst_ld, int32_t Cstore, void* stream);
int sgan_slice_nhwc(const float* src, int32_t src_ld, int32_t c0, int32_t C, int64_t npix, float* dst, int32_t dst_ld,
                    int32_t Cstore, void* stream);

/* ---- input pipeline tail (data/base_dataset.py:17-55, data/aligned_dataset.py:31-42) ------------
 * From a decoded (and, if asked, resized: sgan_image_resize) RGB image `img` [H0][W0][3] uint8 already in device memory: crop the n x n window at
 * (x0, y0) (transforms.RandomCrop / the aligned dataset's offsets) -> horizontal flip (RandomHorizontalFlip) -> rotate by
 * 90 deg * rot counter-clockwise (__rotate: PIL's exact transpose path for square images) -> ToTensor (/255) -> Normalize(0.5, 0.5),
 * written as an NHWC fp32 buffer [n][n][Cstore >= 3] (extra channels zero).  The random draws stay with the caller. */
int sgan_image_prep(const unsigned char* img, int32_t H0, int32_t W0, int32_t x0, int32_t y0, int32_t n, int32_t flip, int32_t rot,
                    float* dst, int32_t dst_ld, int32_t Cstore, void* stream);

/* ---- Image.resize in front of the crop (data/base_dataset.py:19-21 transforms.Scale(.., BILINEAR), :43-50 __scale_width;
 * data/aligned_dataset.py:25 AB.resize((2 loadSize, loadSize), BICUBIC)) --------------------------
 * `src` [H][W][C] uint8 (C <= 4, interleaved) -> `dst` [Ho][Wo][C] uint8, both in device memory, bit-exact with Pillow's 8-bit
 * two-pass resampler (Resample.c): horizontal pass then vertical pass, filter support stretched by the down-scale factor, taps
 * normalised in double and rounded to 22-bit fixed point, each pass rounded and clipped to 8 bits.  `filter` uses Pillow's numbers.
 * `ws` is scratch of at least sgan_image_resize_workspace() bytes (the intermediate image and the two tap tables). */
#define SGAN_RESAMPLE_BILINEAR 2
#define SGAN_RESAMPLE_BICUBIC 3
int64_t sgan_image_resize_workspace(int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t filter);
int sgan_image_resize(const unsigned char* src, int32_t H, int32_t W, int32_t C, unsigned char* dst, int32_t Ho, int32_t Wo,
                      int32_t filter, void* ws, int64_t ws_bytes, void* stream);

/* ---- Adam over up to 64 contiguous fp32 segments in one launch --------------------------------
 * torch.optim.Adam default form (models/fcgan_model.py:98-109):
 *   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * `state_dev` points at 16 bytes of device memory owned by the optimizer: int32 step counter t
 * (starts at 0) followed by two kernel-private floats.  A 1-thread prep launch does t += 1 and
 * derives the bias corrections in fp64, the streaming launch applies them -- so a captured
 * hipGraph replays with the right t.  `lr_dev` points at a device float (the LR schedule writes it). */
typedef struct sgan_adam_seg {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
} sgan_adam_seg;
int sgan_adam_multi(const sgan_adam_seg* segs, int32_t nseg, const float* lr_dev, float beta1, float beta2,
                    float eps, int32_t* state_dev, void* stream);

/* ---- The whole optimizer step of ONE flat arena segment in one launch: Adam (as sgan_adam_multi, same `state_dev` layout plus a ticket
 * word state_dev[3] that must start at 0) and the three derived weight copies of sgan_pack_weights for the conv weight ranges
 * `segs` inside the segment (offsets relative to `p`, sorted, disjoint; flat_t / packed_fwd / packed_bwd use the same offsets; any may be
 * NULL).  zero_grads != 0: the consumed gradients are overwritten with zeros.
 * Replaces: torch.optim.Adam.step() + zero_grad() (models/fcgan_model.py:98-109,182-191) and the weight re-layout that follows it. */
int sgan_adam_pack(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2, float eps,
                   int32_t* state_dev, float* flat_t, void* packed_fwd, void* packed_bwd, void* packed_bwd_f16, const sgan_wt_seg* segs,
                   int32_t nseg, int32_t zero_grads, void* stream);

/* ---- Zero up to 64 device buffers in one launch (the statistics arenas of a training step).  bytes[i] and ptrs[i]: multiples of 16.
 * Replaces: the aten fill launches behind torch.zeros / Tensor.zero_(). */
int sgan_zero_multi(void* const* ptrs, const int64_t* bytes, int32_t n, void* stream);

/* ---- SGD over the same segment table: buf = momentum * buf + g ; p -= lr * buf  (torch.optim.SGD with dampening 0, no Nesterov,
 * no weight decay; `m` of a segment is the momentum buffer, NULL / momentum 0 = plain gradient descent; `v` is ignored).
 * The reference's trainers are hard-wired to Adam (options/train_options.py:30 parses --optimizer and nobody reads it); this
 * is the second update rule of the boundary for callers that do read it. */
int sgan_sgd_multi(const sgan_adam_seg* segs, int32_t nseg, const float* lr_dev, float momentum, void* stream);

/* ---- N(0,1) fill (Philox4x32-10 + Box-Muller), counter-based -----------------------------------
 * Replaces: noise_.normal_(0, 1) (models/fcgan_model.py:126-127).  `offset_dev` is a device uint64
 * the kernels read; with advance != 0 it is moved on by ceil(n / 4) afterwards (graph-replay safe).  Fills with different
 * seeds are independent streams, so a net that draws several tensors per forward pass reads one offset (advance = 0) and
 * moves it once with sgan_rng_advance. */
int sgan_normal_fill(float* dst, int64_t n, uint64_t seed, uint64_t* offset_dev, int32_t advance, void* stream);
/* The same values as sgan_normal_fill on a contiguous [C][H][W] tensor, written where the generator reads them: element (c, h, w)
 * at dst[(h * W + w) * Cs + c] of a padded NHWC buffer (channels C .. Cs-1 are not touched: zero them once). */
int sgan_normal_fill_nhwc(float* dst, int32_t C, int32_t H, int32_t W, int32_t Cs, uint64_t seed, uint64_t* offset_dev,
                          int32_t advance, void* stream);
/* Two latents of the same shape drawn back to back (dst_a first; the values two sgan_normal_fill_nhwc calls give, the stream offset
 * advanced past both) by one single-block launch that also zeroes `zero` (zero_bytes, a multiple of 16, may be 0): the generator pass
 * that opens with two latents (FCGANModel.sample_noise_and_prefetch: the reference's sample_noise() of one step, fcgan_model.py:192-193,
 * and forward() of the next, :179) needs its statistics arena cleared at the same point.  <= 65536 values per latent. */
int sgan_normal_fill_nhwc_pair(float* dst_a, float* dst_b, int32_t C, int32_t H, int32_t W, int32_t Cs, uint64_t seed,
                               uint64_t* offset_dev, void* zero, int64_t zero_bytes, void* stream);
int sgan_rng_advance(uint64_t* offset_dev, uint64_t by, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGAN_HIP_H */
