/*
 * octa_hip.h -- C ABI of libocta_hip.so: the MI355X (gfx950) kernels behind the OCTAve
 * segmentor + discriminator training hot path.
 *
 * The reference (IoBT-VISTEC/OCTAve, /root/reference/architectures) has no native code: every
 * entry point below replaces a chain of ATen calls issued by one of its nn.Module.forward
 * methods (and the autograd backward of that chain).  The reference line(s) replaced are cited
 * per function as  file:line  relative to  architectures/ .
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types.  All pointers are DEVICE pointers
 *     unless the name ends in _host.
 *   - activations are NHWC ("channels-last"): element (b,h,w,c) of a tensor with per-pixel
 *     stride ld lives at ((b*H + h)*W + w)*ld + off + c.  ld/off let a kernel read or write a
 *     channel slice of a wider buffer (U-Net skip concatenation without a copy).
 *   - dtype is OCTA_F32 or OCTA_BF16 for activations / packed weights.  Parameters, their
 *     gradients, BatchNorm statistics and all loss arithmetic are fp32.
 *   - the caller owns every buffer (incl. workspaces); the library never allocates, frees or
 *     retains device memory, never synchronises, and launches only on the stream passed in.
 *   - return value: 0 = OK, negative = octa_status.  octa_last_error() gives a thread-local
 *     message.  No C++ exception crosses the boundary.
 */
#ifndef OCTA_HIP_H
#define OCTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* octa_stream_t; /* hipStream_t */

enum octa_status { OCTA_OK = 0, OCTA_ERR_BAD_ARG = -1, OCTA_ERR_UNSUPPORTED = -2, OCTA_ERR_LAUNCH = -3 };
enum octa_dtype { OCTA_F32 = 0, OCTA_BF16 = 1, OCTA_F16 = 2 };
enum octa_act { OCTA_ACT_NONE = 0, OCTA_ACT_RELU = 1, OCTA_ACT_LEAKY02 = 2, OCTA_ACT_SIGMOID = 3, OCTA_ACT_TANH = 4 };

/* ABI revision: bumped whenever a struct layout or a signature below changes.  octa_version() returns the value the library
 * was BUILT with; the loader (octave_amd/_lib.py) refuses a library whose value differs from this header's. */
#define OCTA_HIP_ABI_VERSION 317
int octa_version(void);
const char* octa_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Convolution engine (implicit GEMM on MFMA).  Replaces every nn.Conv2d / nn.ConvTranspose2d
 * call of the path: extra/resnest.py:24,33,50,83,89,92,181,222,325-334,388;
 * segmentor/blocks.py:40; segmentor/compose.py:79,181; discriminator/blocks.py:46,91,97,69.
 * ---------------------------------------------------------------------------------------- */
typedef struct octa_conv_desc {
    int32_t B, H, W;       /* input  image: B x H x W                                         */
    int32_t OH, OW;        /* output image (OH = (H + 2*pad - KH)/stride + 1)                  */
    int32_t Cin, Cout;     /* logical channel counts (all groups)                             */
    int32_t KH, KW, stride, pad, groups;
    int32_t cin_g_pad;     /* Cin/groups rounded up to a multiple of 8: K granularity of the
                              packed weight; the input buffer must be readable (finite) there  */
    int32_t cout_g_pad;    /* Cout/groups rounded up to a multiple of 8 (dgrad packed weight)  */
    int32_t ldx, xoff;     /* input  per-pixel stride / channel offset (elements, % 8 == 0)    */
    int32_t ldy, yoff;     /* output per-pixel stride / channel offset                         */
    int32_t dtype;         /* octa_dtype of x, y and the packed weights                        */
    int32_t act;           /* octa_act fused into the forward epilogue                         */
    int32_t upshuffle;     /* 1: ConvTranspose2d k2 s2 as a 1x1 GEMM whose output channel
                              n = (di*2+dj)*Cout_t + co is scattered to pixel (2h+di, 2w+dj)   */
    int32_t algo;          /* fwd / dgrad kernel choice: 0 = library heuristic, 1 = 4-wave kernels (3x3 halo
                              / generic tiles), 2 / 3 = 8-wave LDS-DMA kernel with 256x128 / 128x256
                              (pixels x channels) output slabs, 4 / 5 / 6 = generic 4-wave kernel with 128x128 /
                              64x64 / 128x64 tiles, 8 = the LDS-DMA kernel as a 4-wave 128x128 slab (two workgroups
                              per CU), 7 = resident-weight persistent kernel (Cin/group 32 or 64,
                              <= 256 (1x1) / 64 (3x3 s1 p1) output channels).  A choice the shape does not allow falls back to the
                              heuristic.  Results are identical up to fp32 summation order.            */
    int32_t zero_pad;      /* 1 (groups == 1, no upshuffle): the kernel also stores zeros into the output's
                              padding channels [C, round8(C)) (C = Cout for fwd, Cin for dgrad), so the
                              caller need not pre-clear the buffer; needs off + round8(C) <= ld          */
    void* ws;              /* optional fp32 scratch of THIS call (caller-owned, 16-byte aligned, ws_bytes long; NULL / 0 = none),
                              used only by the launches the call issues and never remembered (the library retains no
                              pointer past return, so calls on different streams simply pass different buffers):
                              octa_conv2d_fwd / _dgrad / _dgrad_add: the 8-wave kernels' TAIL SPLIT (algo 2 / 3 / 12) -- a launch
                              whose tile count leaves the last round of CUs at most half full runs those last tiles as 2..8
                              workgroups each over disjoint input-channel ranges; the partial fp32 tiles go through ws and a second
                              small launch finishes them (64 MB covers every layer of the path; same results up to fp32 summation
                              order; without ws nothing splits).
                              octa_conv2d_wgrad: the M-splits store raw fp32 tiles into private slices of ws instead of adding into
                              dw with float atomics, and one fold launch sums the slices in a fixed order (see
                              octa_conv2d_wgrad_batch, which takes the scratch of a whole batch as an argument).            */
    int64_t ws_bytes;
} octa_conv_desc;

/* OIHW-logical fp32 weight (any strides, given in elements) -> packed forward operand
 * [groups][Cout/g][KH][KW][cin_g_pad] of `dtype` (zero padded). */
int octa_pack_weight_fwd(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w,
                         void* packed, int Cout, int Cin_g, int KH, int KW, int groups,
                         int cin_g_pad, int dtype, octa_stream_t stream);
/* ... -> packed data-gradient operand [groups][Cin/g][KH][KW][cout_g_pad] (zero padded). */
int octa_pack_weight_dgrad(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w,
                           void* packed, int Cout, int Cin_g, int KH, int KW, int groups,
                           int cout_g_pad, int dtype, octa_stream_t stream);

/* One launch for many packs (all operands of a network after the optimiser step).  kind 0: forward
 * operand, 1: data-gradient operand, 2: ConvTranspose up-shuffle operand (Cout_g = Cout_t, Cin_g = Cin_t,
 * strides = (s_ci, s_co, s_h, s_w)), 3 / 4: a GROUPED weight with SETS of adjacent groups merged into dense
 * block-diagonal blocks: forward operand [Cout][KH][KW][pad_to = set * Cin_g] / data-gradient operand
 * [Cin][KH][KW][pad_to = set * Cout_g] (zeros off the diagonal blocks; set = groups: one dense conv, set < groups:
 * a conv with groups / set wider groups), for small-channel grouped 3x3 layers that run faster on the halo /
 * resident-weight kernels with 32 or 64 channels per group;
 * 5: the tap-major data-gradient operand of octa_pack_weight_dgrad_taps ([KH*KW][round8(Cin_g)][pad_to >= Cout_g], groups 1).
 * `prefix` = exclusive prefix sum of the operands' TILE counts (octa_pack_tile_count; `total` = their sum): 2048
 * consecutive elements per tile for kinds 0/3/4/5, one 32x32 LDS-transposed tile for kinds 1/2. */
typedef struct octa_pack_desc {
    const float* src;
    void* dst;
    int64_t s_o, s_i, s_h, s_w;
    int32_t kind, dtype, Cout_g, Cin_g, KH, KW, groups, pad_to;
} octa_pack_desc;
size_t octa_pack_tile_count(const octa_pack_desc* desc_host);
int octa_pack_many(const octa_pack_desc* desc_dev, const int64_t* prefix_dev, int n, int64_t total,
                   octa_stream_t stream);

/* ConvTranspose2d k2 s2 weight (Cin_t, Cout_t, 2, 2), any strides -> up-shuffle GEMM operand
 * [(di*2+dj)*Cout_t + co][ci < cin_pad]  (extra/resnest.py:50). */
int octa_pack_weight_convT(const float* w, int64_t s_ci, int64_t s_co, int64_t s_h, int64_t s_w,
                           void* packed, int CinT, int CoutT, int cin_pad, int dtype,
                           octa_stream_t stream);

/* y = act(conv(x, w) + bias).  bias: fp32 [Cout] or NULL. */
int octa_conv2d_fwd(const octa_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                    void* y, octa_stream_t stream);
/* The same conv, and in its epilogue the BatchNorm statistics of y (extra/resnest.py:25,34,86,90,182,224: every conv of the
 * path is followed by a train-mode BatchNorm): stats[r][0][c] += sum (y - shift[c]), stats[r][1][c] += sum (y - shift[c])^2 over
 * the pixels, r = one of `replicas` accumulators (float atomics; the caller zero-fills stats[replicas][2][Cout] and merges the
 * replicas: octa_bn_train_fwd_sums).  shift (optional, [Cout]): the BatchNorm's running mean.  16-bit dtypes, act NONE only.
 * *fused_host = 1 when the kernel that ran supports it, 0 when the conv ran on a kernel that does not (3x3 halo / resident
 * weights): y is complete either way, stats untouched in the latter case and the caller runs the ordinary statistics pass. */
int octa_conv2d_fwd_stats(const octa_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                          float* stats, const float* shift, int replicas, int* fused_host, octa_stream_t stream);
/* dx = conv^T(dy, w): dy has the forward OUTPUT geometry (OH,OW,Cout,ldy,yoff), dx the input's. */
int octa_conv2d_dgrad(const octa_conv_desc* d, const void* dy, const void* w_packed_t, void* dx,
                      octa_stream_t stream);
/* Same, with dx = conv^T(dy, w) + addend: `addend` has dx's shape and dtype, per-pixel stride ldadd.  The sum a network's
 * fan-out needs (a bottleneck's input feeds conv1 AND the shortcut) then costs one extra read in this epilogue instead of a
 * separate 3-pass add kernel.  Runs on the generic and LDS-DMA kernels (not the 3x3 halo / resident-weight ones). */
int octa_conv2d_dgrad_add(const octa_conv_desc* d, const void* dy, const void* w_packed_t,
                          const void* addend, int ldadd, void* dx, octa_stream_t stream);
/* Data gradient of a conv whose INPUT was the output of a fused activation (the discriminator's LeakyReLU / sigmoid / tanh layers,
 * reference architectures/discriminator/blocks.py:46-51, 91-110): dx = conv^T(dy) * f'(gate), gate = that activation's output (the
 * tensor the conv read in forward, NHWC, ldgate elements per pixel, same dtype), gate_act an octa_act code.  The producing layer's
 * backward then skips its derivative kernel (octa_act_bwd).  Runs on the generic 4-wave kernel (d->algo 4 / 5 / 6, else by size). */
int octa_conv2d_dgrad_gated(const octa_conv_desc* d, const void* dy, const void* w_packed_t,
                            const void* gate, int ldgate, int gate_act, void* dx, octa_stream_t stream);
/* Strided data gradient as GEMM + col2im: Z[(b,oh,ow)][(ci*KH+kh)*KW+kw] = dy x W^T comes from
 * octa_conv2d_fwd (1x1, operand = the data-grad packed weight); this folds the overlapping taps:
 * dx[b,ih,iw,ci] = sum_{kh,kw : ih+pad-kh = stride*oh, ...} Z[...].  (discriminator/blocks.py:46,97)
 * The padding channels [Cin, min(round8(Cin), lddx)) of every dx pixel are stored as zeros. */
int octa_col2im(const void* z, int ldz, void* dx, int lddx, int B, int H, int W, int OH, int OW, int Cin,
                int KH, int KW, int stride, int pad, int dtype, octa_stream_t stream);
/* Tap-major form of the same path for small Cin (the 15-channel discriminator inputs): the packed operand's rows are
 * n = (kh*KW + kw) * cin_pad + ci (octa_pack_weight_dgrad_taps), the GEMM output Z has KH*KW*cin_pad columns, and the
 * fold reads one 16-byte channel vector per (pixel, tap).  dx receives cin_pad channels per pixel (the padding ones are
 * zeros because their operand rows are).  groups == 1. */
int octa_pack_weight_dgrad_taps(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w,
                                void* packed, int Cout, int Cin, int KH, int KW, int cin_pad,
                                int cout_pad, int dtype, octa_stream_t stream);
int octa_col2im_taps(const void* z, int ldz, void* dx, int lddx, int B, int H, int W, int OH, int OW,
                     int cin_pad, int KH, int KW, int stride, int pad, int dtype, octa_stream_t stream);
/* The same fold with the activation derivative of channels [0, gate_channels): dx[.., c] *= f'(gate[.., c]) (gate: the conv's own
 * input, whose first gate_channels channels are an activation's output -- the squeeze conv's sigmoid under torch.cat, reference
 * architectures/discriminator/blocks.py:91-95, 124). */
int octa_col2im_taps_gated(const void* z, int ldz, void* dx, int lddx, int B, int H, int W, int OH, int OW,
                           int cin_pad, int KH, int KW, int stride, int pad, int dtype, const void* gate, int ldgate,
                           int gate_act, int gate_channels, octa_stream_t stream);
/* dw[o,i,kh,kw] += sum_pixels dy * x   (fp32 gradient of the OIHW-logical parameter, addressed
 * through its element strides so OIHW-dense and channels-last storage both work; accumulated
 * with atomics, so the caller zeroes it once per step). */
int octa_conv2d_wgrad(const octa_conv_desc* d, const void* x, const void* dy, float* dw,
                      const int64_t* dw_strides /* o,i,h,w element strides of dw */,
                      float* dbias /* optional: dbias[Cout] += sum_pixels dy (fused bias gradient) */,
                      octa_stream_t stream);
/* MANY weight gradients in one launch.  The weight gradients of a backward pass are consumed only by the
 * optimiser, so the host queues them (per network stage) and hands the queue over here: every bf16 / f16 job
 * with >= 128 output channels per group runs on the batched 8-wave kernel (wgrad8.hip: 256x128 / 128x256 output
 * slabs, LDS-DMA ring), whose M-split is chosen over ALL its jobs; the rest falls through to octa_conv2d_wgrad,
 * job by job.  `jobs_host` is a HOST array, read before the call returns (the descriptors travel as kernel
 * arguments); the device buffers it names must stay valid until the stream has run the launch.
 * Same += semantics as octa_conv2d_wgrad (extra/resnest.py:24,33,83,181,222; discriminator/blocks.py:46,97). */
typedef struct octa_wgrad_job {
    octa_conv_desc d;
    const void* x;
    const void* dy;
    float* dw;
    float* dbias;            /* optional */
    int64_t dw_strides[4];   /* o,i,h,w element strides of dw */
} octa_wgrad_job;
/* ws / ws_bytes (optional; caller-owned fp32 scratch, 16-byte aligned, used by this call's launches only; 128 MB covers a batch of
 * every layer of the path): the M-splits of the single-problem kernels (and, with octa_tuning_set(4, 1), of the batched ones) store
 * raw fp32 tiles into private slices of ws instead of adding into dw with float atomics (384-512 workgroups adding into the same few
 * KB were half of the few-channel layers' time), and ONE fold launch per call sums the slices in a fixed order and adds the result
 * to dw / dbias: same += semantics, deterministic.  A job whose slices do not fit what is left of ws keeps the atomic epilogue
 * (deterministic mode: runs unsplit instead).  The jobs' own d.ws fields are ignored here. */
int octa_conv2d_wgrad_batch(const octa_wgrad_job* jobs_host, int n, float* ws, int64_t ws_bytes, octa_stream_t stream);
/* Which kernel family octa_conv2d_wgrad_batch runs this job on: 1 = batched 256(N)x128(K) slabs, 2 = batched 128x256
 * slabs, 3 = batched 256x256 tiles (wgrad9), 0 = the single-problem kernels (host-side query; lets the caller group a queue so
 * that one call = one family).
 * Not a status code. */
size_t octa_wgrad_job_class(const octa_wgrad_job* job_host);
/* Name of the kernel template instance the calling thread's last octa_conv2d_fwd / _dgrad / _wgrad
 * call dispatched, e.g. "conv_igemm_kernel<bf16,128x128>", "conv3x3_halo_kernel<bf16,128x64>",
 * "conv_wgrad_kernel<bf16,128>", "conv3x3_wgrad_halo_kernel<4>" (measurement aid: bench.py's roofline
 * leg keys its per-kernel timings with it, so they line up with the rocprofv3 kernel names). */
const char* octa_last_conv_kernel(void);
/* out[c] += sum over rows of src[row*ld + off + c]  (bias gradients; fp32 accumulate).  workspace: optional,
 * octa_colsum_workspace_floats(C) floats, uninitialised: per-block partials + a fold launch instead of
 * thousands of blocks adding into the one or two cache lines of out[] with float atomics. */
size_t octa_colsum_workspace_floats(int C);
int octa_colsum(const void* src, int64_t rows, int C, int ld, int off, int dtype, float* out,
                float* workspace, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Layout / copy helpers (compose.py:125-130 pad, 141-147 cat+crop, 155,162,169 cat).
 * ---------------------------------------------------------------------------------------- */
/* NCHW (strided, fp32) -> NHWC [B,H,W,ld] of dtype, channels [0,C) written, [C,cpad) zeroed. */
int octa_nchw_to_nhwc(const float* src, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                      void* dst, int B, int C, int H, int W, int ld, int off, int cpad, int dtype,
                      octa_stream_t stream);
/* NHWC of dtype -> NCHW fp32 (dense), optionally accumulate (+=). */
int octa_nhwc_to_nchw(const void* src, int ld, int off, int dtype, float* dst, int B, int C, int H,
                      int W, int accumulate, octa_stream_t stream);
/* dst[b, h, w, doff + c] = src[b, h, w, soff + c] for h < Hd, w < Wd (crop when smaller than the
 * source image, zero-fill when larger = F.pad bottom/right); C % 8 == 0.  accumulate: += */
int octa_copy_channels(const void* src, int Hs, int Ws, int lds, int soff, void* dst, int Hd, int Wd,
                       int ldd, int doff, int B, int C, int dtype, int accumulate, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d, training mode (extra/resnest.py:25,34,86,90,182,224,338,393): batch statistics,
 * running-stat update (momentum, unbiased variance), fused ReLU and residual add.
 * ---------------------------------------------------------------------------------------- */
/* workspace floats needed by octa_bn_stats / octa_bn_bwd_reduce for C channels, rows rows */
size_t octa_bn_workspace_floats(int64_t rows, int C);
/* mean[c], invstd[c] over rows = B*H*W; updates running_mean/var in place when non-NULL. */
int octa_bn_stats(const void* x, int64_t rows, int C, int ld, int off, int dtype, float eps,
                  float momentum, float* mean, float* invstd, float* running_mean,
                  float* running_var, float* workspace, octa_stream_t stream);
/* y = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ).
 * relu_mask (optional, relu != 0): rows * C / (8 bf16 | 4 fp32) bytes, one per 16-byte chunk of y in
 * row-major (row, chunk) order, bit e = output e of the chunk is > 0.  The backward pass reads it
 * instead of y (1/16 of the bytes). */
int octa_bn_apply(const void* x, int ldx, int xoff, const float* mean, const float* invstd,
                  const float* gamma, const float* beta, const void* residual, int ldr, int roff,
                  void* y, int ldy, int yoff, int64_t rows, int C, int dtype, int relu,
                  uint8_t* relu_mask, octa_stream_t stream);
/* statistics + apply in one call (training forward): same results as octa_bn_stats followed by octa_bn_apply.
 * Small tensors (<= 16 rows per thread over 32 row slabs) run as TWO launches (the apply merges the
 * per-slab partials itself: no finalize launch); larger ones as the three launches of the two calls above.  mean / invstd receive the batch
 * statistics (the backward pass needs them); relu_mask as in octa_bn_apply. */
int octa_bn_train_fwd(const void* x, int ldx, int xoff, const float* gamma, const float* beta,
                      const void* residual, int ldr, int roff, void* y, int ldy, int yoff, int64_t rows,
                      int C, int dtype, float eps, float momentum, int relu, float* mean, float* invstd,
                      float* running_mean, float* running_var, uint8_t* relu_mask, float* workspace,
                      octa_stream_t stream);
/* Training forward from the sums octa_conv2d_fwd_stats left (shift = running_mean before this call's update, NULL = 0): a
 * one-thread-per-channel merge of the replicas (double) + the apply launch; same outputs as octa_bn_train_fwd. */
int octa_bn_train_fwd_sums(const void* x, int ldx, int xoff, const float* sums, int replicas, const float* gamma,
                           const float* beta, const void* residual, int ldr, int roff, void* y, int ldy, int yoff,
                           int64_t rows, int C, int dtype, float eps, float momentum, int relu, float* mean, float* invstd,
                           float* running_mean, float* running_var, uint8_t* relu_mask, octa_stream_t stream);
/* backward.  With relu != 0 the ReLU mask comes from relu_mask (as written by octa_bn_apply) when it
 * is given, else from the forward output y (y may be NULL when relu_mask is given).
 * dgamma/dbeta are ACCUMULATED (+=).  dres (optional) receives the masked dy. */
int octa_bn_bwd(const void* dy, int lddy, int dyoff, const void* x, int ldx, int xoff, const void* y,
                int ldy, int yoff, const float* mean, const float* invstd, const float* gamma,
                void* dx, int lddx, int dxoff, void* dres, int lddr, int droff, float* dgamma,
                float* dbeta, int64_t rows, int C, int dtype, int relu, const uint8_t* relu_mask,
                float* workspace, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pooling (compose.py:45 maxpool 3/2/1; resnest.py:189 avgpool 3/s/1 count_include_pad;
 * resnest.py:383 avgpool s/s ceil_mode count_include_pad=False).
 * ---------------------------------------------------------------------------------------- */
int octa_maxpool3s2_fwd(const void* x, void* y, uint8_t* argmax, int B, int H, int W, int C, int OH,
                        int OW, int dtype, octa_stream_t stream);
int octa_maxpool3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int B, int H, int W, int C,
                        int OH, int OW, int dtype, octa_stream_t stream);
int octa_avgpool_fwd(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int k,
                     int stride, int pad, int count_include_pad, int dtype, octa_stream_t stream);
int octa_avgpool_bwd(const void* dy, void* dx, int B, int H, int W, int C, int OH, int OW, int k,
                     int stride, int pad, int count_include_pad, int dtype, octa_stream_t stream);
/* The same backward passes with a FAN-OUT ADDEND: dx = pool^T(dy) + addend, where addend (NULL = none) is the gradient another
 * consumer of the pooled tensor produced -- the U-Net skip connections (compose.py:141-147: x_k feeds the decoder's cat AND the
 * next encoder stage, whose first op on it is the avg_down shortcut's pool, resnest.py:383 / the stem's max-pool, compose.py:45) --
 * read with its own per-pixel stride ld_addend (elements, % 8 == 0, >= C; a channel slice of the cat's gradient), summed in fp32
 * before the one rounding of dx.  Replaces autograd's separate add kernel over the two gradients. */
int octa_maxpool3s2_bwd_add(const void* dy, const uint8_t* argmax, void* dx, const void* addend, int ld_addend, int B, int H, int W,
                            int C, int OH, int OW, int dtype, octa_stream_t stream);
int octa_avgpool_bwd_add(const void* dy, void* dx, const void* addend, int ld_addend, int B, int H, int W, int C, int OH, int OW,
                         int k, int stride, int pad, int count_include_pad, int dtype, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The rest of the reference's public surface around the hot path, extras.hip.
 * ---------------------------------------------------------------------------------------- */
/* InterlayerDivergence, JSD branch (segmentor/losses.py:154-169): mean_q = mean_j w_j resize(Q_j), M = (P + mean_q)/2,
 * loss = mean_pix sum_k [P/2 (log(P+1e-12) - log(M+eps)) + mean_q/2 (log(mean_q+1e-12) - log(M+eps))].  out[0] = loss,
 * out[1] = NaN flag; ws: 1024 floats.  bwd: gq_ws is a B*K*H*W fp32 scratch, dbasis may be NULL (stop_gradient). */
int octa_interlayer_jsd_fwd(const float* basis, const float* const* maps_host, const int* shifts_host,
                            const float* weights_host, int n_maps, float eps, int B, int K, int H, int W,
                            float* out, float* ws, octa_stream_t stream);
int octa_interlayer_jsd_bwd(const float* basis, const float* const* maps_host, const int* shifts_host,
                            const float* weights_host, int n_maps, float eps, int B, int K, int H, int W,
                            const float* g, float* dbasis, float* gq_ws, float* const* dmaps_host,
                            octa_stream_t stream);
/* WeightedPartialCE's non-manual branches (segmentor/losses.py:40-60) on z = y_hat * ys (or y_hat when full):
 * mode 0 (2 classes): nn.CrossEntropyLoss()(z, long(ys[:, 1])); mode 1 (1 class): nn.BCEWithLogitsLoss()(z, ys).
 * out[0] = loss; ws: 1024 floats; din is dense NCHW. */
int octa_pixel_ce_fwd(const float* in, const int64_t* in_strides, const float* ys, const int64_t* ys_strides, int B,
                      int K, int H, int W, int full, int mode, float* out, float* ws, octa_stream_t stream);
int octa_pixel_ce_bwd(const float* in, const int64_t* in_strides, const float* ys, const int64_t* ys_strides, int B,
                      int K, int H, int W, int full, int mode, const float* g, float* din, octa_stream_t stream);
/* LabelNoise mode 'label' (discriminator/blocks.py:172-177): out = |1 - x|; with dy: out = d|1-x|/dx * dy. */
int octa_abs1m(const float* x, const float* dy, float* out, int64_t n, octa_stream_t stream);
/* ResnestUNet.predict post-processing (segmentor/compose.py:189-199) on (B,K,H,W) logits (any strides):
 * mode 0: fp32 sigmoid map; mode 1: int64 one-hot of the argmax (first maximum wins, like torch.argmax), and
 * maxclass[0] (int, zeroed by the caller) = largest argmax seen (F.one_hot sizes its output by it). */
int octa_predict_map(const float* logits, const int64_t* strides, int B, int K, int H, int W, int mode, void* out,
                     int* maxclass_zeroed, octa_stream_t stream);
/* Dice coefficient terms per (sample, class): out[b][k][0] = sum pred*target, out[b][k][1] = sum (pred + target). */
int octa_dice_terms(const float* pred, const int64_t* pred_strides, const float* target, const int64_t* target_strides,
                    int B, int K, int H, int W, float* out, octa_stream_t stream);
/* nn.AdaptiveAvgPool2d on dense NCHW fp32 (classification head, segmentor/compose.py:89): forward (dy == NULL,
 * out = [BC][OH][OW]) or backward (x may be NULL, out = dx [BC][H][W]). */
int octa_adaptive_avgpool(const float* x, const float* dy, float* out, int64_t BC, int H, int W, int OH, int OW,
                          octa_stream_t stream);
/* Device-side synthetic batch: x (B,3,H,W) grayscale plane replicated to 3 channels, ys (B,2,H,W)
 * scribbles (~5 % per class, rest unlabelled), real (B,2,H,W) dense one-hot mask; counter-based generator keyed on
 * (seed, element index).  vessel != 0: curvilinear vessel field instead of white noise. */
int octa_synth_octa(int64_t seed, int B, int H, int W, int vessel, float* x, float* ys, float* real,
                    octa_stream_t stream);
/* Levels 1 .. n-1 of the discriminator's real-mask pyramid (nearest down-sampling by 2^l; the contract of
 * discriminator/blocks.py:114-125) in one launch; levels_host[0] is ignored (level 0 is `src` itself). */
int octa_mask_pyramid(const float* src, float* const* levels_host, int n_levels, int64_t BC, int H, int W,
                      octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Split attention (extra/resnest.py:106-138), radix 2.
 * ---------------------------------------------------------------------------------------- */
/* gap[b][c] = mean_hw( x[b,hw,c] + x[b,hw,C+c] )   (fp32 out; resnest.py:108-116).
 * prezeroed != 0 (here and below): the caller hands over an accumulator that is already zero (e.g. a slice of a scratch
 * slab cleared once per training step), so the entry point skips its own zero-fill launch. */
int octa_splat_gap(const void* x, float* gap, int B, int HW, int C, int dtype, int prezeroed, octa_stream_t stream);
/* out[b,hw,c] = a0*x[b,hw,c] + a1*x[b,hw,C+c], (a0,a1) = softmax(logit[b][c], logit[b][C+c])
 * (resnest.py:125-138; the view(B, radix, C) has no cardinality transpose).  relu: fuse the
 * ReLU that follows SplAt in ResNestDecoder (resnest.py:29). */
int octa_splat_apply(const void* x, const float* logits, void* out, int B, int HW, int C, int dtype,
                     int relu, octa_stream_t stream);
/* backward of gap+apply: given dout, x, logits, out (for the relu mask) and dgap (gradient
 * arriving at gap through fc1), produce dx [B,HW,2C] and dlogits [B,2C] (fp32). */
int octa_splat_bwd(const void* dout, const void* x, const float* logits, const void* out,
                   const float* dgap, void* dx, float* dlogits, int B, int HW, int C, int dtype,
                   int relu, int phase, int prezeroed /* dlogits, phase 0 */, octa_stream_t stream);

/* Split attention with the PRECEDING BatchNorm + ReLU (bn0, resnest.py:100-103) recomputed on the fly: x is the raw conv output
 * [B, HW, 2C]; mean / invstd (batch statistics: octa_bn_stats) / gamma / beta are bn0's, [2C] each; every kernel evaluates
 * y = max(x * gamma * invstd + (beta - mean * gamma * invstd), 0) in registers, so y and its gradient are never stored.
 *   octa_splat_bn_gap        = octa_splat_gap(y);   octa_splat_bn_apply = octa_splat_apply(y)
 *   octa_splat_bn_bwd_logits = octa_splat_bwd(phase 0) on y
 *   octa_splat_bn_bwd_dx     = octa_splat_bwd(phase 1) followed by bn0's backward: dx [B, HW, 2C] is the gradient of the RAW conv
 *                              output, dgamma / dbeta (optional) are accumulated (+=); ws: octa_bn_workspace_floats(B * HW, 2 C). */
int octa_splat_bn_gap(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      float* gap, int B, int HW, int C, int dtype, int prezeroed, octa_stream_t stream);
int octa_splat_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                        const float* logits, void* out, int B, int HW, int C, int dtype, int relu, octa_stream_t stream);
int octa_splat_bn_bwd_logits(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, const float* logits, const void* out, float* dlogits, int B, int HW, int C,
                             int dtype, int relu, int prezeroed, octa_stream_t stream);
int octa_splat_bn_bwd_dx(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const float* logits, const void* out, const float* dgap, void* dx,
                         float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int dtype, int relu,
                         octa_stream_t stream);
/* Round 4: the same backward in TWO passes over (dout, out, x) instead of three.  octa_splat_bn_bwd_logits2 also accumulates, per
 * (sample, channel) and radix half, aux[b][8][C] = {P, Px, M, Mx} x 2 (sums over the pixels of [y > 0] dout', [y > 0] dout' xhat,
 * [y > 0], [y > 0] xhat; zero-filled by the caller or here when prezeroed = 0); after the micro-net's backward octa_splat_bn_bwd_dx2
 * assembles bn0's backward sums from them (a_r P + (dgap / HW) M, ...), folds them and runs the dx pass: no statistics pass. */
int octa_splat_bn_bwd_logits2(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* logits, const void* out, float* dlogits, float* aux, int B, int HW, int C,
                              int dtype, int relu, int prezeroed, octa_stream_t stream);
int octa_splat_bn_bwd_dx2(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, const float* logits, const void* out, const float* dgap, const float* aux, void* dx,
                          float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int dtype, int relu, octa_stream_t stream);
/* Round 5: octa_splat_bn_bwd_logits2 WITHOUT its second launch (the radix-2 softmax backward of resnest.py:126-127 applied in place to the B x 2C
 * sums): `da` leaves with the raw attention gradients and octa_splat_mlp_bwd_da (below) applies the softmax backward where it reads them. */
int octa_splat_bn_bwd_da2(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, const float* logits, const void* out, float* da, float* aux, int B, int HW, int C,
                          int dtype, int relu, int prezeroed, octa_stream_t stream);


/* The attention micro-net on (B, C) vectors (resnest.py:118-125), exact fp32, 2 <= B <= 32:
 * h1 = fc1(gap) [grouped 1x1, + bias]; h2 = relu(bn1(h1)) [batch statistics when training, running stats
 * updated]; logits = fc2(h2).  w1: [inter][C/groups], w2: [2C][inter/groups] dense.
 * The backward consumes dlogits and produces dgap plus ACCUMULATED (+=) parameter gradients. */
int octa_splat_mlp_fwd(const float* gap, const float* w1, const float* b1, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                       int training, const float* w2, const float* b2, float* h1, float* h2, float* mean,
                       float* invstd, float* logits, int B, int C, int inter, int groups, octa_stream_t stream);
int octa_splat_mlp_bwd(const float* dlogits, const float* gap, const float* w1, const float* w2,
                       const float* h1, const float* h2, const float* mean, const float* invstd,
                       const float* gamma, float* dh1_workspace /* B*inter */, float* dgap, float* dw1,
                       float* db1, float* dgamma, float* dbeta, float* dw2, float* db2, int B, int C,
                       int inter, int groups, int prezeroed /* dgap */, octa_stream_t stream);
/* The same from the RAW attention gradients da[b][2C] and the logits (dlogits = rSoftmax backward of da, radix 2, computed on the fly):
 * one launch fewer per split-attention block and direction (21 per training step). */
int octa_splat_mlp_bwd_da(const float* da, const float* logits, const float* gap, const float* w1, const float* w2,
                          const float* h1, const float* h2, const float* mean, const float* invstd,
                          const float* gamma, float* dh1_workspace /* B*inter */, float* dgap, float* dw1,
                          float* db1, float* dgamma, float* dbeta, float* dw2, float* db2, int B, int C,
                          int inter, int groups, int prezeroed /* dgap */, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Attention gate / head (segmentor/blocks.py:38-46; compose.py:79,181): per-pixel K-class linear
 * (K <= 8), channel softmax, x * sum_{c>=1} y_c.  Class maps are fp32 NCHW (user-facing).
 *   mode 0: gate  -> masked (NHWC dtype) and y = softmax(logits) (NCHW fp32)
 *   mode 1: head  -> y = logits (NCHW fp32), masked unused
 * ---------------------------------------------------------------------------------------- */
int octa_aag_fwd(const void* x, const float* w, const float* bias, void* masked, float* y, int64_t B,
                 int HW, int C, int K, int dtype, int mode, octa_stream_t stream);
/* dx = dmasked*mask + W^T dlogits; dw/dbias ACCUMULATED (fp32). dy may be NULL (no grad).
 * workspace (optional, octa_aag_workspace_floats(C, K) floats, no initialisation needed): the blocks then
 * leave per-block partials that a second small launch folds into dw/dbias, instead of ~1000 blocks adding
 * into the same few cache lines with float atomics (50-100 us per launch). */
size_t octa_aag_workspace_floats(int C, int K);
int octa_aag_bwd(const void* x, const float* w, const float* y, const void* dmasked, const float* dy,
                 void* dx, float* dw, float* dbias, int64_t B, int HW, int C, int K, int dtype,
                 int mode, float* workspace, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise activations (discriminator/blocks.py:51,94,110): backward from the OUTPUT.
 * ---------------------------------------------------------------------------------------- */
int octa_act_bwd(const void* y, const void* dy, void* dx, int64_t n, int act, int dtype,
                 octa_stream_t stream);
int octa_relu_fwd(const void* x, void* y, int64_t n, int dtype, octa_stream_t stream);
int octa_add(const void* a, const void* b, void* out, int64_t n, int dtype, octa_stream_t stream);
int octa_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Losses.  Class-probability maps are fp32 with arbitrary (b,c,h,w) element strides.
 * ---------------------------------------------------------------------------------------- */
/* WeightedPartialCE (manual=True) + DiceLoss on p = softmax(logits) or on given probabilities:
 * segmentor/losses.py:26-61 and 70-74.  from_logits=1 fuses the nn.Softmax(dim=1) the caller
 * applies (compose.py:192).  out[0] = wpce, out[1] = dice.  ws: >= octa_loss_workspace_floats. */
size_t octa_loss_workspace_floats(int B, int K);
int octa_wpce_dice_fwd(const float* in, const int64_t* in_strides, const float* ys,
                       const int64_t* ys_strides, int B, int K, int H, int W, int from_logits,
                       int full, int reduction_sum, float* out, float* ws, octa_stream_t stream);
/* d(in) = g_wpce * dWPCE/d(in) + g_dice * dDice/d(in); g_* are device scalars (or NULL = 0). */
int octa_wpce_dice_bwd(const float* in, const int64_t* in_strides, const float* ys,
                       const int64_t* ys_strides, int B, int K, int H, int W, int from_logits,
                       int full, int reduction_sum, const float* g_wpce, const float* g_dice,
                       const float* ws, float* din, const int64_t* din_strides, octa_stream_t stream);

/* nn.Softmax(dim=1) over the class axis of the (B,K,H,W) logits (segmentor/compose.py:192):
 * strided fp32 in, dense NCHW fp32 out; backward from the saved probabilities. */
int octa_class_softmax_fwd(const float* in, const int64_t* in_strides, float* out, int B, int K, int H,
                           int W, octa_stream_t stream);
int octa_class_softmax_bwd(const float* p, const float* dp, const int64_t* dp_strides, float* din,
                           int B, int K, int H, int W, octa_stream_t stream);

/* InterlayerDivergence, KLD/mean (segmentor/losses.py:111-147), nearest up-sampling fused.
 * basis: [B,K,H,W]; maps[i]: [B,K,H>>shift[i],W>>shift[i]] (dense NCHW fp32), weight[i] != 0.
 * out[0] = loss, out[1] = NaN flag. */
int octa_interlayer_kl_fwd(const float* basis, const float* const* maps_host, const int* shifts_host,
                           const float* weights_host, int n_maps, float wsum, int B, int K, int H,
                           int W, float* out, float* ws, octa_stream_t stream);
/* dbasis (may be NULL when stop_gradient) and dmaps[i] (zero-initialised by the caller,
 * accumulated with atomics) scaled by the device scalar g. */
int octa_interlayer_kl_bwd(const float* basis, const float* const* maps_host, const int* shifts_host,
                           const float* weights_host, int n_maps, float wsum, int B, int K, int H,
                           int W, const float* g, float* dbasis, float* const* dmaps_host,
                           octa_stream_t stream);
/* The same with FAN-OUT ADDENDS: dbasis += basis_addend, dmaps[i] += map_addends_host[i] (each NULL = none; dense fp32, the
 * map's own shape): the gradient the other consumer of an attention map produced -- the discriminator's generator pass
 * (models/octa.py training step: the maps feed InterlayerDivergence AND the discriminator) -- added where the KL gradient
 * is written instead of by autograd's separate add launch per map. */
int octa_interlayer_kl_bwd_add(const float* basis, const float* const* maps_host, const int* shifts_host,
                               const float* weights_host, int n_maps, float wsum, int B, int K, int H,
                               int W, const float* g, float* dbasis, float* const* dmaps_host,
                               const float* basis_addend, const float* const* map_addends_host,
                               octa_stream_t stream);

/* LS-GAN losses (discriminator/losses.py:11-14, 22-24).  mode 0: 0.5*mean((f-1)^2) (generator);
 * mode 1: 0.5*mean((r-1)^2) + 0.5*mean((f+1)^2).  Forward writes out[0]; backward writes
 * d_real / d_fake scaled by device scalar g. */
int octa_lsgan_fwd(const float* real, const float* fake, int n_real, int n_fake, int mode, float* out,
                   octa_stream_t stream);
int octa_lsgan_bwd(const float* real, const float* fake, int n_real, int n_fake, int mode,
                   const float* g, float* d_real, float* d_fake, octa_stream_t stream);

/* The segmentor's loss from its terms (train.py; reference: the sums in segmentor/compose.py's training step): a = (2,) tensor
 * [wpce, dice], b = (2,) tensor [divergence, its NaN flag], c = the generator's adversarial term; any of them may be NULL.
 * total[0] = ((a0 wa0 + a1 wa1) + b0 wb0 + b1 wb1) + c0 wc0 with zero-weight terms skipped; scaled[0] = total x (scale_dev ? scale_dev[0] :
 * scale_host).  Backward: da / db / dc = g[0] x scale x weight.  One launch each way instead of ~18 scalar ATen launches. */
int octa_loss_combine_fwd(const float* a, const float* b, const float* c, float wa0, float wa1, float wb0, float wb1, float wc0,
                          const float* scale_dev, float scale_host, float* total, float* scaled, octa_stream_t stream);
int octa_loss_combine_bwd(const float* g, float wa0, float wa1, float wb0, float wb1, float wc0, const float* scale_dev,
                          float scale_host, float* da, float* db, float* dc, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Discriminator pieces.
 * ---------------------------------------------------------------------------------------- */
/* InstanceNoise + clip (discriminator/blocks.py:149-154): dst[b,h,w,c] = clip(src[b,c,h,w] +
 * noise[h,w], 0, 1) written NHWC of dtype with channels [C,cpad) zeroed; mask[b,c,h,w] = 1 where
 * the clip passed the gradient.  noise may be NULL (is_training False). */
int octa_noise_clip_fwd(const float* src, const int64_t* src_strides, const float* noise, void* dst,
                        uint8_t* mask, int B, int C, int H, int W, int ld, int cpad, int dtype,
                        int clip /* 0: InstanceNoise(clipping=False), mask all ones */, octa_stream_t stream);
/* dsrc[b,c,h,w] (dense NCHW fp32) = mask ? ddst[b,h,w,c] : 0 */
int octa_noise_clip_bwd(const void* ddst, int ld, const uint8_t* mask, float* dsrc, int B, int C,
                        int H, int W, int dtype, octa_stream_t stream);
/* The same, written space-to-depth for the k4 s2 p1 conv behind it (blocks.py:46): dst is NHWC [B][H/2+1][W/2+1][ld >= 4C],
 * dst[b,Y,X,(dy*2+dx)*C+c] = clip(src[b,c,2Y+dy-1,2X+dx-1] + noise, 0, 1) (0 outside the image; channels [4C, ld) zeroed); that conv is
 * then a k2 s1 p0 conv with weight w2[o,(dy*2+dx)*C+c,i,j] = w[o,c,2i+dy,2j+dx] -- same sums, no padded channels (2 -> 8 before).
 * mask as above (NCHW of the ORIGINAL geometry).  H, W even. */
int octa_noise_clip_s2d_fwd(const float* src, const int64_t* src_strides, const float* noise, void* dst,
                            uint8_t* mask, int B, int C, int H, int W, int ld, int dtype, int clip, octa_stream_t stream);
/* dsrc[b,c,h,w] (dense NCHW fp32) = mask ? ddst[b,(h+1)/2,(w+1)/2,(((h+1)&1)*2+((w+1)&1))*C+c] : 0 */
int octa_noise_clip_s2d_bwd(const void* ddst, int ld, const uint8_t* mask, float* dsrc, int B, int C,
                            int H, int W, int dtype, octa_stream_t stream);
/* Spectral norm (torch.nn.utils.spectral_norm, 1 power iteration; blocks.py:105-108).
 * w: fp32 [Cout][K] dense (OIHW flattened).  Updates u,v in place when do_power_iter, writes
 * sigma[0] and w_sn = w / sigma.  uv_saved (optional, Cout + K floats) receives the u then v this call
 * used, for the backward pass (u and v keep changing with every later forward).  ws_prezeroed: the first K
 * floats of ws are already zero (the caller's step-wide zero slab), so no clearing launch is issued. */
int octa_spectral_norm_fwd(const float* w, float* u, float* v, int Cout, int K, int do_power_iter,
                           float eps, float* sigma, float* w_sn, float* ws /* K + Cout floats */,
                           float* uv_saved, int ws_prezeroed, octa_stream_t stream);
/* dw (+)= (dw_sn - (sum(dw_sn * w_sn)) u v^T) / sigma      (ws: 1 float; accumulate 0 overwrites dw).
 * dwsn_khw: 0 = dw_sn is dense OIHW like w_sn; KH*KW = dw_sn is channels-last [Cout][KH][KW][Cin], the layout
 * octa_conv2d_wgrad's atomics prefer. */
int octa_spectral_norm_bwd(const float* dw_sn, const float* w_sn, const float* u, const float* v,
                           const float* sigma, int Cout, int K, float* dw, float* ws, int accumulate, int ws_prezeroed, int dwsn_khw,
                           octa_stream_t stream);
/* The same for up to 8 independent layers in one launch per kernel (a discriminator call normalises four conv weights:
 * 12 launches -> 3 forward, 8 -> 2 backward).  Job arrays live in HOST memory (they travel as kernel arguments). */
typedef struct octa_sn_job {
    const float* w; float* u; float* v; float* sigma; float* w_sn; float* ws /* K + Cout floats */; float* uv_saved /* optional */;
    int32_t Cout, K;
    /* optional (round 5): the conv operands of w_sn in `pack_dtype`, written by the same launch that writes w_sn -- the forward operand
     * [Cout][KH][KW][round8(Cin)] (octa_pack_weight_fwd's layout, groups 1) and / or the tap-major data-gradient operand
     * [KH*KW][round8(Cin)][round8(Cout)] (octa_pack_weight_dgrad_taps's); needs K == Cin*KH*KW.  NULL: not written. */
    void* packed_fwd; void* packed_dgrad_taps;
    int32_t KH, KW, Cin, pack_dtype;
} octa_sn_job;
typedef struct octa_sn_bwd_job {
    const float* dw_sn; const float* w_sn; const float* u; const float* v; const float* sigma; float* dw; float* ws /* 1 float */;
    int32_t Cout, K, accumulate, dwsn_khw;
} octa_sn_bwd_job;
int octa_spectral_norm_fwd_batch(const octa_sn_job* jobs_host, int n, int do_power_iter, float eps, int ws_prezeroed,
                                 octa_stream_t stream);
int octa_spectral_norm_bwd_batch(const octa_sn_bwd_job* jobs_host, int n, int ws_prezeroed, octa_stream_t stream);
/* Full-extent conv = per-sample dot product (blocks.py:68-72): out[b] = x[b,:].w + bias.
 * x NHWC [B, n] of dtype, w fp32 [n] in the same (h,w,c) order. */
int octa_fullconv_fwd(const void* x, const float* w, const float* bias, float* out, int B, int64_t n,
                      int dtype, float sign, const float* sign_dev /* optional device scalar, multiplies sign */,
                      int out_prezeroed /* 1: out is already zero (no init launch) */, octa_stream_t stream);
/* dx = sign * dout[b] * w;  dw += sign * sum_b dout[b] x[b];  dbias += sign * sum_b dout[b].  dw_c = 0: dw in x's (h,w,c) order;
 * dw_c = C: dw is [C][n / C], the (1, C, H, W) parameter's own order (a gradient-sink target). */
int octa_fullconv_bwd(const void* x, const float* w, const float* dout, void* dx, float* dw,
                      float* dbias, int B, int64_t n, int dtype, float sign, const float* sign_dev,
                      int dw_c, octa_stream_t stream);
/* ... with dx multiplied by f'(x): x is the output of activation gate_act (the last tanh) and dx the gradient that reaches it. */
int octa_fullconv_bwd_gated(const void* x, const float* w, const float* dout, void* dx, float* dw,
                            float* dbias, int B, int64_t n, int dtype, float sign, const float* sign_dev,
                            int dw_c, int gate_act, octa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser: fused Adam over a flat fp32 parameter arena (train step a17).
 * ---------------------------------------------------------------------------------------- */
/* step_dev (optional, device int32): the number of updates APPLIED so far.  The kernel then takes t = *step_dev + 1 for
 * the bias corrections instead of the host's `step`, so a captured hipGraph advances by itself and an update skipped for
 * a non-finite gradient does not count (torch.optim.Adam under GradScaler behaves the same); octa_step_end commits it. */
int octa_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale,
                   const int32_t* step_dev, const float* ls_state /* optional */, int ls_flag, octa_stream_t stream);

/* Dynamic loss scaling for fp16 training, all on the device (capturable, no host sync).  state (8 floats): [0] loss scale,
 * [1] clean steps since the last change, [2..7] found-inf flags, one per optimiser.  octa_nonfinite_flag sets *flag = 1 if
 * the (already all-reduced) gradient arena holds an inf / nan; octa_adam_step with ls_state divides its grad_scale by
 * state[0] and does nothing when state[ls_flag] != 0; octa_loss_scale_update (once per step, after the optimisers) halves
 * the scale on a flagged step (x backoff), multiplies it by `growth` after `interval` clean steps, and clears the flags. */
int octa_nonfinite_flag(const float* g, int64_t n, float* flag, octa_stream_t stream);
int octa_loss_scale_update(float* state, int nflags, float growth, float backoff, int interval,
                           octa_stream_t stream);
/* The end of an optimiser step as ONE launch: step_dev{0,1} (optional; Adam's applied-update counters of optimiser 0 / 1)
 * advance unless ls_state[2 + k] flags a skipped update, then octa_loss_scale_update's work on ls_state (optional), then
 * octa_host_tick's (optional: tick_dev += 1, published to tick_host). */
int octa_step_end(float* ls_state, int nflags, float growth, float backoff, int interval, int32_t* step_dev0,
                  int32_t* step_dev1, int32_t* tick_dev, int32_t* tick_host, octa_stream_t stream);

/* Tuning switches (process-wide; benchmarks and A/B tests).  key 1: which tile families octa_conv2d_wgrad_batch may use,
 * bit 0 = 256x128 / 128x256 (wgrad8), bit 1 = 256x256 (wgrad9); default 3.  key 2: timing-only ablation builds of wgrad9.
 * key 3: smallest Cout / groups the batched kernels accept (default 128).  key 4: 1 = the batched kernels' M-splits also store
 * partial tiles for the fold launch of octa_conv2d_wgrad_batch's scratch (default 0: their atomics are not contended, measured).
 * key 5: DETERMINISTIC MODE (0 / 1; also OCTA_DETERMINISTIC=1 in the environment; SURVEY 8b "deterministic variant required for parity
 * tests"): every sum that crosses workgroups runs in a fixed order -- weight gradients through private partial tiles + the ordered fold
 * (fp32 jobs too) or unsplit when no scratch was passed, the split-attention GAP / logit-gradient / dgap sums, the attention gate's
 * dw, column sums, the spectral-norm power iteration and dot product and the full-extent conv as ONE workgroup per output address,
 * no BatchNorm statistics in conv epilogues -- so two runs on the same inputs are bit-identical.  Slower (parity tests; never the
 * benchmark).  key 6: LDS image of the 2-D patch kernel's input patch (algo 12): 1 (default) = lines placed so that fragment reads
 * are bank-conflict-free, 0 = the linear image of round 4 (A/B runs).  key 7: 1 = the first-pass reductions (BatchNorm
 * statistics / backward sums, split-attention backward sums) walk their tensor END FIRST, i.e. start on what the producing kernel wrote
 * last and the 256 MB memory-side cache still holds; 0 (default: no gain measured) = forward.  Results are identical (same partial slots).
 * key 8: schedule of the 256 x 256 weight-gradient kernel: 0 (default) = rounds of workgroups with one split length for the whole
 * batch (wgrad9); 1 = every class of problems its own split, blocks dealt to the XCDs in eighths, XCDs started at different
 * positions of their sequences (wgrad9x); 2 = the same with one persistent workgroup per CU.  Same sums, different atomic order;
 * no gain measured in situ (the launch is power-limited: DESIGN.md 3.12).  key 9: MFMA shape of the 256 x 256 weight-gradient kernel:
 * 0 = v_mfma_f32_32x32x16 (wgrad9), 1 = v_mfma_f32_16x16x32 at the same wave tile (wgrad9s), 2 = 32x32x16 with FOUR waves of 128 x 128 per
 * workgroup, 256 accumulator registers per lane (wgrad9a: a quarter fewer LDS fragment reads per FLOP).  key 10: bias-free 3x3 stride-1
 * layers with H % 5 == 0, W % 25 == 0, Cin / groups % 32 == 0 on the 2-D patch weight-gradient kernel (wgrad2d: one input patch shared by the
 * nine taps): 0 = never, 1 = every such layer, 2 (default) = the ungrouped ones with >= 256 input and >= 256 output channels (one launch for up to
 * four of them), where taking them out of the batched wgrad9 launch is measured to pay. */
int octa_tuning_set(int key, int value);

/* Debug / self-test: raw MFMA + transposed LDS read layout probes (tests only). */
int octa_probe_mfma(int which, const void* a, const void* b, float* d, octa_stream_t stream);
/* Measurement aid (tools/comm_pressure.py): `nblocks` 256-thread workgroups stream over buf[0, bytes) (read, add 0, write back: the
 * contents are unchanged) `reps` times -- a stand-in for the CU footprint and HBM traffic of a collective's kernels on a communication
 * stream when only one GPU is available. */
int octa_probe_stream_load(float* buf, int64_t bytes, int nblocks, int reps, octa_stream_t stream);

/* dst <- slot (*dev_counter % slots) of a ring of `slots` x slot_bytes in pinned, device-addressable HOST memory, copied
 * by a kernel (no hipMemcpy).  dev_counter is the step counter octa_host_tick advances. */
int octa_ring_fetch(const void* ring_host, int64_t slot_bytes, int slots, const int* dev_counter, void* dst,
                    octa_stream_t stream);
/* Step tick: *dev_counter += 1 and the new value is stored (system scope) to host_flag, a word of pinned host memory
 * that the device can address.  Lets the host pace queued hipGraph replays with plain loads (train.py). */
int octa_host_tick(int* dev_counter, int* host_flag, octa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OCTA_HIP_H */
