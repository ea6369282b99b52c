/*
 * libmfcnet_hip.so -- C ABI of the MI355X-native MFCNet hot path (gfx950 only).
 *
 * The reference (shadowfax11/mfcnet-tracker) has NO native interface for this
 * path: all of its arithmetic runs inside stock PyTorch/cuDNN operators called
 * from models/hrnet.py and models/multiframe_model.py.  Each entry point below
 * therefore cites the reference *operator call* it replaces; the only native
 * precedent in the reference is the 8-function pybind surface of
 * models/sync_bn/inplace_abn/src/inplace_abn.cpp:66-75 (mean_var / forward /
 * edz_eydz / backward), whose MI355X equivalents are mfc_bn_finalize,
 * mfc_combine_fwd, mfc_bnbwd_reduce/_finalize/_apply.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()); no torch types;
 *   - activations are NHWC with a padded channel pitch Cp (multiple of 8);
 *     element type T is float (MFC_F32, parity mode), bf16 (MFC_BF16,
 *     throughput mode) or IEEE half (MFC_F16: the same kernels, layouts and
 *     fusions as bf16; training in it needs a loss scale); statistics, coefficients, parameters, parameter
 *     gradients and optimizer state are always fp32;
 *   - a "granule" is 16 bytes = 4 floats or 8 bf16 / fp16 of one pixel;
 *   - "groups": the T frames of a clip are batched as N = T*B images with
 *     G = T statistic groups (images_per_group = B), because the reference runs
 *     base_model once per frame with separate BatchNorm batches
 *     (models/multiframe_model.py:459-461);
 *   - functions return 0 on success, a negative mfc_status otherwise; they never
 *     throw, never allocate device memory, never synchronise (mfc_prof_collect / mfc_prof_dump / mfc_program_profile excepted);
 *     all work is enqueued on `stream` (a hipStream_t; pass torch.cuda.current_stream().cuda_stream);
 *   - the single-operation entry points keep no state between calls.  Interpreter state (side streams, events) lives in an mfc_ctx handle
 *     (mfc_ctx_create / mfc_program_run_ctx; one per model in mfcnet_amd); the handle-less mfc_program_run / _ex use a per-device default
 *     context and are therefore limited to one host thread per device at a time.  Per-call behaviour is an argument (run_flags), not a
 *     switch.  The only process-wide state left is the tuning switches, which are NOT declared here (mfcnet_hip_tuning.h, tools only) and
 *     the event profiler's log (mfc_prof_*: a measurement aid).  The model is not wrappable by nn.DataParallel (one process per GPU: INTEGRATION.md).
 *
 * SURVEY.md 8(b) lists a minimum set of entry points by working names; the exports that serve them:
 *   mfc_conv2d_fwd                  -> mfc_conv2d_fwd (k in {1,2,3,11} x any, stride 1/2, bias, fused input transform, fused output statistics)
 *   mfc_conv2d_dgrad                -> mfc_conv2d_fwd on the data-gradient weight image (mfc_pack_weights mode 1; stride 2 = four parity-class
 *                                      launches; optional epilogue fusions acc_src / bn_y of mfc_conv_desc)
 *   mfc_conv2d_wgrad                -> mfc_conv2d_wgrad (+ mfc_conv2d_wgrad_parts / _batch, mfc_unpack_wgrad for the partial-sum slices)
 *   mfc_bn_finalize                 -> mfc_bn_finalize, mfc_bn_finalize_batch (eval-mode table)
 *   mfc_bn_bwd_reduce / _apply      -> mfc_bnbwd_reduce, mfc_bnbwd_finalize, mfc_bnbwd_apply (finalize optionally fused into apply)
 *   mfc_bilinear_up_fwd             -> mfc_combine_fwd (a source whose view is smaller than the output is up-sampled, align_corners=False,
 *                                      explicit size; sum / BN-apply / ReLU / channel-slice concat in the same pass)
 *   mfc_bilinear_up_bwd             -> mfc_mask_add (dst smaller than g: adjoint of the up-sampling, optional ReLU mask, accumulate-into)
 *   mfc_concat_head_gather          -> mfc_head_gather_fwd / mfc_head_gather_bwd (x4 up-sampling + temporal concat + flow / depth, Basic warp)
 *   mfc_logsoftmax_nll_jaccard_fwd  -> mfc_loss_fwd (= mfc_loss_partial + mfc_loss_finalize; the split form for data-parallel sums)
 *   mfc_logsoftmax_nll_jaccard_bwd  -> mfc_loss_bwd
 *   mfc_adam_multi                  -> mfc_adam_step / mfc_adam_step_guarded (one launch per contiguous learning-rate segment of the flat arena)
 *   *_workspace_bytes()             -> mfc_conv2d_layout (packed-weight bytes, LDS), mfc_conv2d_wgrad_parts (partial-sum slices)
 */
#ifndef MFCNET_HIP_H
#define MFCNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MFC_F32 = 0, MFC_BF16 = 1, MFC_F16 = 2 } mfc_dtype;     /* F16: IEEE half storage, same kernels and layouts as BF16 */

typedef enum {
    MFC_OK = 0,
    MFC_ERR_INVALID_ARG = -1,
    MFC_ERR_UNSUPPORTED = -2,
    MFC_ERR_LAUNCH = -3
} mfc_status;

#ifndef MFC_STAT_REPLICAS      /* (an A/B build may define another count: make REPLICAS=4 -- the host side must then agree, MFC_STAT_REPLICAS in the environment) */
#define MFC_STAT_REPLICAS 8    /* rows the per-channel sum atomics are spread over (8 / 16 / 32 measured on the full step: 39.3 / 39.5 / 39.7 ms -- every finalize reads all rows) */
#endif
/* Statistic sums (BatchNorm / GroupNorm forward sums, BatchNorm-backward sums) are fp64 cells added to with fp64 atomics: the fp32 partial
 * sums of the workgroups then add up to the same value whatever order the workgroups arrive in.  With fp32 cells the order noise of the
 * atomics (1e-7 relative) is amplified by var = E[x^2] - mean^2 on channels whose variance is small against their mean, and a bf16 training-mode
 * forward then has run-to-run differences of 15 % of the logit scale at low resolutions (tools/train_noise.py); with fp64 cells it repeats. */
typedef double mfc_stat_t;
#define MFC_LOSS_ACC_FLOATS 96

/* ---- BatchNorm coefficient block: fp32 [G][4][Cp] = scale, shift, mean, rstd ---- */
#define MFC_COEF_SCALE 0
#define MFC_COEF_SHIFT 1
#define MFC_COEF_MEAN  2
#define MFC_COEF_RSTD  3

/* ------------------------------------------------------------------------------------
 * Convolution as an implicit GEMM on MFMA (forward, and data-gradient by re-use).
 * Replaces: every nn.Conv2d call of models/hrnet.py (conv3x3 :39-42, Bottleneck 1x1 :82-88,
 * fuse 1x1 / strided 3x3 :200-230, transition :361-386, last_layer :334-351) and of the
 * temporal head models/multiframe_model.py:191-201 (11x11, 3x3, 1x1); in backward, the
 * cuDNN dgrad those calls trigger under loss.backward() (src/engine.py:70).
 *
 * Logical problem: out[n, i, j, co] = sum_{a<TA, b<TB, ci} Wp[a,b][ci][co] *
 *                     f(in[n, i*in_stride + dh0 + a, j*in_stride + dw0 + b, ci])
 * where f is the optional fused input transform relu?(x*scale[g,ci] + shift[g,ci])
 * (BatchNorm-apply of the producer layer; zero padding is applied AFTER f), and the
 * logical output pixel (i,j), i<Hl, j<Wl is stored at tensor pixel
 * (i*out_sh + out_oh, j*out_sw + out_ow) of out[N, Hout, Wout, Cout_p].
 *   forward conv k,s,p :  TA=TB=k, dh0=dw0=-p, in_stride=s, out stride 1
 *   dgrad of stride 1   :  flipped taps (done by the weight packer), dh0=-(k-1-p)
 *   dgrad of stride 2   :  four launches, one per output parity class
 * Wp is the packed weight image produced by mfc_pack_weights.
 * bf16 1x1 / stride-1 launches with >= 128 input and output channels (or 64 -> >= 256) and a pixel count that is a
 * multiple of 256 (last_layer[0], hrnet.py:334-351, and its data gradient) run as a plain GEMM with 256x256 tiles
 * (conv_gemm1x1.hip); mfc_conv2d_layout reports the weight blocking of whichever kernel will run.
 * Optional epilogue: + bias[co]; out = acc + out (accumulate); per-(group,channel)
 * sum / sum-of-squares of the fp32 results added to out_stats[replica][G][2][Cs].
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const void* in;          /* T [N, Hin, Win, Cin_p] */
    const void* wp;          /* T packed weights, layout of mfc_conv2d_layout(d) */
    void* out;               /* T [N, Hout, Wout, Cout_p] */
    const float* bias;       /* [>=Cout] or NULL */
    const float* in_coef;    /* [G][4][Cin_p] or NULL (no input transform) */
    mfc_stat_t* out_stats;   /* [MFC_STAT_REPLICAS][G][2][Cout_p] or NULL */
    int32_t dtype;
    int32_t N, Hin, Win, Cin_p, Cin;       /* Cin = channels actually reduced over (<= Cin_p) */
    int32_t Hout, Wout, Cout_p, Cout;      /* physical output tensor */
    int32_t Hl, Wl;                        /* logical output grid of this launch */
    int32_t TA, TB, dh0, dw0, in_stride;
    int32_t out_sh, out_sw, out_oh, out_ow;
    int32_t in_relu, images_per_group, accumulate;
    int32_t TH, TW;                        /* pixel tile (TH*TW <= 128); 0 = choose */
    /* ---- optional epilogue fusions of a DATA-GRADIENT launch (bf16, launches whose mfc_conv_layout reports fa = 1) ----
     * acc_src: with `accumulate`, the running sum is read from THIS tensor (same geometry as out) instead of from out itself:
     *          out = acc + acc_src.  (The residual branch of a BasicBlock, hrnet.py:71: the block input's gradient is the masked block-output
     *          gradient plus conv1's data gradient -- no copy of the former is made.)
     * bn_y:    BatchNorm / ReLU backward statistics of the tensor this launch completes (precedent: inplace_abn_cuda.cu:174-256 edz_eydz):
     *          with v = the final value of an output element, m its mask (bn_mask_mode 0: 1; 2: bn_y*scale+shift > 0; 3: bit of bn_bits, the
     *          1-bit image mfc_combine_fwd wrote) the launch stores v*m instead of v and adds sum v*m and sum v*m*yhat (yhat = (bn_y - mean) * rstd,
     *          coefficients from bn_coef [G][4][Cout_p]) to out_stats [MFC_STAT_REPLICAS][G][2][Cout_p] -- i.e. it does the work of
     *          mfc_bnbwd_reduce (with its masked-gradient output) in the epilogue, and mfc_bnbwd_apply then runs with mask_mode 0. */
    const void* acc_src;     /* T [N, Hout, Wout, Cout_p] or NULL */
    const void* bn_y;        /* T [N, Hout, Wout, Cout_p] (pre-BatchNorm conv output) or NULL */
    const float* bn_coef;    /* [G][4][Cout_p] coefficient block of that BatchNorm */
    const void* bn_bits;     /* uint8 [N*Hout*Wout*Cout_p/8] (bn_mask_mode 3) */
    int32_t bn_mask_mode;
    int32_t flags;           /* bit 0 (MFC_CONV_WANT_FA): restrict the geometry search to launches that support the fusions above (fa = 1) when one
                              * exists -- set it BEFORE mfc_conv2d_layout / packing, so that the packed weight image matches the launch */
} mfc_conv_desc;
#define MFC_CONV_WANT_FA 1
/* flags bit 1 (round 4): this launch is the data gradient of a 3x3 / stride-2 / pad-1 convolution over ALL FOUR output parity classes at once
 * (instead of four launches): TA = TB = 2, dh0 = dw0 = 0, in_stride = 1, out_sh = out_sw = 2, out_oh = out_ow = 0, Hl = ceil(Hout / 2),
 * Wl = ceil(Wout / 2).  Class c = 2 ph + pw computes, on the dy grid, the 2x2-tap problem of output pixels (2i + ph, 2j + pw) from ITS weight
 * image: the packed image (mfc_conv2d_layout(d).bytes) holds the four class images one after the other, each packed by an mfc_pack_job with
 * TA = TB = 2, kh_step = kw_step = -2 and (kh0, kw0) = (1 + ph, 1 + pw) -- taps that fall outside the filter are written as zeros.  The
 * epilogue options (accumulate, acc_src, bn_y, out_stats) apply to every class. */
#define MFC_CONV_S2_CLASSES 2
/* flags bit 2 (round 4): the caller guarantees that no launch of this descriptor accumulates (accumulate = 0, acc_src = NULL) -- the planner sets it on data
 * gradients whose result has exactly one producer (the input of the forward convolution is a never-materialised conv -> BN -> ReLU tensor).  It lets
 * mfc_conv2d_layout choose launches that have no accumulating form (the 64-channel ring launch on large images); a launch that accumulates anyway is
 * refused with MFC_ERR_INVALID_ARG.  Like bit 0 it must be set BEFORE mfc_conv2d_layout / packing: the packed weight image follows the launch. */
#define MFC_CONV_NEVER_ACC 4
int mfc_conv2d_fwd(const mfc_conv_desc* d, void* stream);
/* The packed weight image a launch of `d` reads is laid out [TA/TAS][nchunks][Yblocks][nslots][NT16][granule]
 * (slot = (row-in-group, tap column, granule-in-chunk)): the weights of one (tap-row group, channel chunk, cout block) stage are one contiguous block that is DMA-copied
 * straight into LDS.  The blocking depends on the launch geometry, so the packer asks for it here.      */
typedef struct { int32_t KG, nchunks, NT16, Yblocks, nslots, TA, TB, lds_bytes; int64_t bytes;
                 int32_t MT, TH, TW, grid, per_block, TAS, NW, fa; } mfc_conv_layout;   /* last row: launch geometry (informational; NW = waves per workgroup;
                                                                                          * fa = 1: the launch supports the acc_src / bn_y epilogue fusions) */
int mfc_conv2d_layout(const mfc_conv_desc* d, mfc_conv_layout* out);
/* LDS bytes a launch of `d` needs (for tests / planners); <0 on invalid desc. */
int mfc_conv2d_lds_bytes(const mfc_conv_desc* d);

/* ------------------------------------------------------------------------------------
 * Weight packer: fp32 reference layout [Cout][Cin][KH][KW] (nn.Conv2d.weight, the
 * state_dict layout of SURVEY.md 3.4) -> the packed image(s) the conv kernel reads.
 * One launch handles a whole table of jobs (all ~311 convolutions of the net).
 *   mode 0: forward image          n = cout, k = cin, tap(a,b) = (kh0 + a*kh_step, kw0 + b*kw_step)
 *   mode 1: data-gradient image    n = cin,  k = cout (roles swapped), same tap rule
 * ------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t src;            /* const float* [Cout][Cin][KH][KW] */
    uint64_t dst;            /* T* [TA][nchunks][Yblocks][nslots][NT16][granule] (mfc_conv_layout of the consuming launch) */
    int32_t Cout, Cin, KH, KW;
    int32_t TA, TB, kh0, kh_step, kw0, kw_step;
    int32_t mode;            /* 0 fwd, 1 dgrad */
    int32_t KG, nchunks, NT16, Yblocks, nslots, TAS;
    int32_t block0;          /* first block of this job in the launch (prefix sum) */
    int32_t nblocks;
    int32_t pad_;
} mfc_pack_job;
int mfc_pack_weights(const mfc_pack_job* jobs_dev, int32_t njobs, int32_t total_blocks, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Weight gradient.  Replaces the cuDNN wgrad of every conv above under loss.backward().
 *   dWp[a,b][co][ci] = sum_{n,i,j} dy[n,i,j,co] * f(x[n, i*s+dh0+a, j*s+dw0+b, ci])
 * The pixel axis is split over `parts` workgroup sets; set s WRITES (plain stores, no atomics:
 * deterministic, and ~10x cheaper than fp32 atomics on a 36 KiB image shared by 300 workgroups)
 * its partial sum into slice s of dwp = fp32 [parts][TA*TB][Cout16][Cin16].  mfc_conv2d_wgrad_parts
 * returns the number of slices a launch of `d` writes (size dwp with it; the buffer must have
 * been zeroed once, ever: cells outside the channel blocks are never written).
 * mfc_unpack_wgrad sums the slices and converts all images to the reference layout.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const void* x;           /* T [N, Hin, Win, Cin_p] conv input (pre input-transform) */
    const void* dy;          /* T [N, Hout, Wout, Cout_p] */
    float* dwp;              /* fp32 [parts][TA*TB][Co16][Ci16] partial sums (see above) */
    const float* in_coef;    /* as in mfc_conv_desc */
    int32_t dtype;
    int32_t N, Hin, Win, Cin_p, Cin;
    int32_t Hout, Wout, Cout_p, Cout;
    int32_t TA, TB, dh0, dw0, in_stride;
    int32_t in_relu, images_per_group;
    int32_t TH, TW;
    int32_t splits;          /* pixel splits (= parts); 0 = choose */
    int32_t batch;           /* 0/1: alone.  n > 1: one of n gradients of identical geometry launched together by
                              * mfc_conv2d_wgrad_batch (each gets 1/n of the workgroups: fewer, longer-running workgroups pay the
                              * per-workgroup prologue / reduction / partial-sum store n times less often) */
    int32_t pad_;
} mfc_wgrad_desc;
int mfc_conv2d_wgrad(const mfc_wgrad_desc* d, void* stream);
int mfc_conv2d_wgrad_parts(const mfc_wgrad_desc* d);      /* > 0: slices written (depends on `batch`); < 0: error */
/* n <= 8 descriptors that differ only in x / dy / dwp / in_coef / in_relu, each with batch = n; bf16 3x3 / 11x11 only
 * (MFC_ERR_UNSUPPORTED otherwise -- launch them one by one) */
int mfc_conv2d_wgrad_batch(const mfc_wgrad_desc* descs, int32_t n, void* stream);

typedef struct {
    uint64_t src;            /* const float* packed partial sums [nparts][TA*TB][Co16][Ci16] */
    uint64_t dst;            /* float* [Cout][Cin][KH][KW] gradient (overwritten) */
    int32_t Cout, Cin, KH, KW, Co16, Ci16;
    int32_t block0, nblocks; /* nblocks = ceil(KH*KW*Co16*Ci16 / 256): one thread per packed cell */
    int32_t nparts, pad_;
} mfc_unpack_job;
int mfc_unpack_wgrad(const mfc_unpack_job* jobs_dev, int32_t njobs, int32_t total_blocks, void* stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm statistics -> coefficients.  Replaces the statistics half of every
 * nn.SyncBatchNorm / nn.BatchNorm2d call (hrnet.py:31, bn_helper.py:10,
 * multiframe_model.py:193-199; kernel precedent inplace_abn_cuda.cu:76-111 mean_var).
 * training=1: batch mean / biased var from the sums the conv epilogue accumulated, per
 * group; running stats updated group after group with momentum 0.1 and unbiased var,
 * num_batches_tracked += G.  training=0: coefficients from the running stats.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const mfc_stat_t* stats; /* [R][G][2][Cp] (training) */
    float* coef;             /* out [G][4][Cp] */
    const float* gamma;      /* [C] */
    const float* beta;       /* [C] */
    float* running_mean;     /* [C] */
    float* running_var;      /* [C] */
    int64_t* num_batches_tracked; /* scalar or NULL */
    int32_t C, Cp, G, training;
    float count;             /* elements per (group, channel) */
    float eps, momentum;
} mfc_bnfin_desc;
int mfc_bn_finalize(const mfc_bnfin_desc* d, void* stream);
/* the same for a DEVICE-resident table of n descriptors in one launch (max_Cp = largest Cp in the table).  The rows must not
 * depend on each other or on work still in flight; the plan uses it for eval-mode BatchNorms (coefficients from the running
 * statistics, hrnet.py:31 in `.eval()`), whose ~300 one-block launches otherwise sit on the forward critical path. */
/* (every row needs 1 <= G <= 8 like mfc_bn_finalize; the table lives on the device, so the CALLER validates it -- rows that violate it are skipped) */
int mfc_bn_finalize_batch(const mfc_bnfin_desc* table_dev, int32_t n, int32_t max_Cp, void* stream);

/* ------------------------------------------------------------------------------------
 * Tensor views used by the element-wise kernels.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t ptr;            /* T* base of [N, H, W, Cp] */
    uint64_t coef;           /* const float* [G][4][Cp_coef] BN coefficients or 0 */
    int32_t H, W, Cp, c_off; /* c_off: first channel of the slice used (multiple of 8) */
} mfc_view;

/* out[:, :, :, slice] = relu?( sum_k  affine_k?( bilinear_k?( src_k ) ) )
 * Replaces the BN-apply / ReLU / residual add / fuse-layer sum / F.interpolate element-wise
 * tail: hrnet.py:62-63,66,71-72 (BasicBlock), :99-113 (Bottleneck), :245-260 (fuse sum with
 * bilinear up-sampling, align_corners=False), :464-469 (4-way up-sample + concat).      */
typedef struct {
    mfc_view out;
    mfc_view src[4];
    int32_t nsrc, relu, dtype;   /* relu: 0 none, 1 ReLU, 2 SiLU (x * sigmoid(x), the ResUnet_VB activation) */
    int32_t N, C;            /* C: channels in the slice (multiple of granule size or padded) */
    int32_t images_per_group;
    uint64_t maskbits;       /* optional (bf16): uint8 [N*H*W*out.Cp/8], bit e of byte (pixel*Cp + c)/8 = out[pixel][c + e] > 0 -- the
                              * ReLU mask the BatchNorm backward of the summed terms reads (mask_mode 3) instead of the whole tensor */
} mfc_combine_desc;
int mfc_combine_fwd(const mfc_combine_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * ResUnet_VB building blocks (models/resunet.py:46-76: WeightStandardizedConv2d -> GroupNorm -> SiLU), inference.
 *   mfc_ws_normalize:  w'[o] = (w[o] - mean_o) * rsqrt(var_o + eps) over the cin*kh*kw weights of output channel o (biased variance,
 *                      resunet.py:55-60); the standardised weights then go through mfc_pack_weights / mfc_conv2d_fwd like any others.
 *   mfc_gn_finalize:   GroupNorm coefficients from the per-(image, channel) sums a convolution launched with images_per_group = 1
 *                      accumulated: per image n and channel group g, mean / rstd over C/groups channels x H x W; coef[n][0][c] = gamma_c * rstd,
 *                      coef[n][1][c] = beta_c - mean * scale, [2] = mean, [3] = rstd -- the layout of the BatchNorm coefficient block, so the
 *                      consumers apply it the same way (mfc_combine_desc.relu = 2 adds SiLU: x * sigmoid(x), resunet.py:67).
 *   mfc_upsample_nearest2x: nn.Upsample(scale_factor = 2, mode = 'nearest') (resunet.py:34-38), NHWC.
 * ------------------------------------------------------------------------------------ */
int mfc_ws_normalize(const float* w, float* w_out, int32_t Cout, int32_t per_out, float eps, void* stream);
typedef struct {
    const mfc_stat_t* stats; /* [MFC_STAT_REPLICAS][N][2][Cp] */
    float* coef;             /* out [N][4][Cp] */
    const float* gamma;      /* [C] */
    const float* beta;       /* [C] */
    int32_t C, Cp, N, groups;
    float count;             /* H * W */
    float eps;
} mfc_gnfin_desc;
int mfc_gn_finalize(const mfc_gnfin_desc* d, void* stream);
int mfc_upsample_nearest2x(const void* src, void* dst, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t Cp, void* stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm / ReLU backward (precedent: inplace_abn_cuda.cu:174-292 edz_eydz + backward).
 *   m     = mask (mode 0: 1; mode 1: mask_src > 0; mode 2: y*scale+shift > 0; mode 3: bit image; mode 4: the SiLU derivative at y*scale+shift)
 *   reduce:   bstats[r][g][0][c] += sum g*m ;  bstats[r][g][1][c] += sum g*m*yhat ;
 *             if dy.ptr is set, the masked gradient is also written out: dy (+)= g*m (`accumulate`) -- the residual /
 *             identity branch of the same sum receives exactly g*m, so one pass serves both (and later passes can
 *             read g*m back instead of g and the mask)
 *   finalize: c1 = mean(g*m), c2 = mean(g*m*yhat)  (0 in eval mode);
 *             dgamma[c] = sum_g sum g*m*yhat, dbeta[c] = sum_g sum g*m
 *   apply:    dy = scale * (g*m - c1 - yhat*c2)
 * `g` may be a bilinear-up-sampled view's gradient already reduced to y's resolution.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    mfc_view g;              /* gradient wrt the BN output (post-activation if mask) */
    mfc_view y;              /* conv output (pre-BN); y.coef = this BN's coefficient block */
    mfc_view mask;           /* mode 1: tensor whose sign gives the ReLU mask; mode 3 (bf16): its 1-bit image written by mfc_combine_fwd (ptr = bits) */
    mfc_view dy;             /* apply: destination (may alias g); reduce: optional g*m output */
    mfc_stat_t* bstats;      /* [R][G][2][Cp] */
    const float* bcoef;      /* [G][2][Cp] c1, c2 (apply) */
    int32_t mask_mode, dtype, N, C, images_per_group, accumulate;
    /* apply only, optional: finalize fused into the apply launch (fin_dgamma set).  Every workgroup first sums the replica
     * rows of `bstats` itself (c1, c2 into LDS; `bcoef` is then unused) and workgroup 0 writes dgamma / dbeta -- which takes
     * the tiny dependent mfc_bnbwd_finalize launch (~4 us of launch latency) out of the BatchNorm's backward chain.  For
     * BatchNorms of at most 128 channels (the redundant prologue grows with C). */
    float* fin_dgamma;       /* [fin_C] */
    float* fin_dbeta;        /* [fin_C] */
    int32_t fin_C, fin_training;
    float fin_count;
    int32_t gn_mode;         /* apply only: 1 = GroupNorm (nn.GroupNorm of resunet.py:64; statistics per image): bcoef = A, B of mfc_gnbwd_finalize and
                              * dy = scale * g*m - rstd * (A + yhat * B); the fused finalize is not available in this mode */
} mfc_bnbwd_desc;
int mfc_bnbwd_reduce(const mfc_bnbwd_desc* d, void* stream);
int mfc_bnbwd_apply(const mfc_bnbwd_desc* d, void* stream);

typedef struct {
    const mfc_stat_t* bstats; /* [R][G][2][Cp] */
    float* bcoef;            /* out [G][2][Cp] */
    float* dgamma;           /* [C] (overwritten) */
    float* dbeta;            /* [C] (overwritten) */
    int32_t C, Cp, G, training;
    float count;
} mfc_bnbwdfin_desc;
int mfc_bnbwd_finalize(const mfc_bnbwdfin_desc* d, void* stream);

/* ---- ResUnet_VB backward (models/resunet.py:46-76, 97-180 under loss.backward(); precedent for the normalisation half: inplace_abn_cuda.cu:174-292).
 *   SiLU:       mask_mode 4 of mfc_bnbwd_reduce / mfc_bnbwd_apply multiplies the incoming gradient by d silu(z) / dz, z = y*scale + shift
 *   GroupNorm:  mfc_bnbwd_reduce with images_per_group = 1 leaves per-(image, channel) sums S1 = sum g, S2 = sum g*yhat;
 *               mfc_gnbwd_finalize folds them into A = mean_{group channels, pixels}(gamma g), B = mean(gamma g yhat) (bcoef [N][2][Cp], per channel)
 *               and dgamma[c] = sum_n S2, dbeta[c] = sum_n S1 (overwritten); mfc_bnbwd_apply with gn_mode = 1 writes dy = scale g - rstd (A + yhat B)
 *   weight standardisation: mfc_ws_backward turns the gradient w.r.t. the standardised weights into the gradient w.r.t. the raw ones
 *   nearest x2: mfc_upsample_nearest2x_bwd sums the four copies (ddst [N, H, W, Cp] (+)= from dsrc [N, 2H, 2W, Cp])                          */
typedef struct {
    const mfc_stat_t* bstats; /* [MFC_STAT_REPLICAS][N][2][Cp] */
    float* bcoef;            /* out [N][2][Cp] */
    const float* gamma;      /* [C] */
    float* dgamma;           /* [C] (overwritten) */
    float* dbeta;            /* [C] (overwritten) */
    int32_t C, Cp, N, groups;
    float count;             /* H * W */
    int32_t pad_;
} mfc_gnbwdfin_desc;
int mfc_gnbwd_finalize(const mfc_gnbwdfin_desc* d, void* stream);
int mfc_ws_backward(const float* w, const float* dws, float* dw, int32_t Cout, int32_t per_out, float eps, void* stream);
int mfc_upsample_nearest2x_bwd(const void* dsrc, void* ddst, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t Cp, int32_t accumulate, void* stream);

/* dst = (accumulate ? dst : 0) + adjoint_bilinear?( g * m )   -- gradient of identity /
 * up-sampled terms of a combine (residual adds hrnet.py:71,112; fuse sums :250-259).      */
typedef struct {
    mfc_view g;              /* high-resolution gradient */
    mfc_view mask;           /* same resolution as g: the tensor whose sign is the mask (mode 1) or its 1-bit image (mode 3, bf16; ptr = bits) */
    mfc_view dst;            /* resolution of the term's source */
    int32_t mask_mode, dtype, N, C, accumulate;
    int32_t pad_;
    float* scratch;          /* dst smaller than g (adjoint of an up-sampling): fp32 [N, g.H, dst.W, C] workspace of the
                              * separable form (a W pass over the high-resolution gradient, then an H pass); NULL selects the
                              * one-pass gather */
} mfc_maskadd_desc;
int mfc_mask_add(const mfc_maskadd_desc* d, void* stream);

/* per-channel sum over pixels of dy (bias gradient of last_layer.0 / last_layer.3, hrnet.py:335-350) */
int mfc_bias_grad(const void* dy, float* db, int32_t dtype, int64_t npix, int32_t Cp, int32_t C, void* stream);
/* the same without atomics: `nparts` workgroups each write the partial sums of their pixel range to slices[part][Cp] (fp32); the caller adds the
 * slices in a fixed order -- the plan does it with an mfc_unpack_job {KH = KW = 1, Cin = Ci16 = 1, Co16 = Cp}, next to the weight gradients' slices --
 * so that the bias gradient, like every other gradient, does not depend on the order workgroups finish in (MFC_OP_BIAS_GRAD with raw.i[3] = nparts) */
int mfc_bias_grad_slices(const void* dy, float* slices, int32_t dtype, int64_t npix, int32_t Cp, int32_t C, int32_t nparts, void* stream);

/* ------------------------------------------------------------------------------------
 * Layout bridges at the boundary (the reference API is NCHW fp32 lists of tensors,
 * multiframe_model.py:457-471).
 * ------------------------------------------------------------------------------------ */
/* dst[n, h, w, c_off + c] = src[n, c, h, w], c < C; other channels of the granules touched are zeroed
 * when zero_pad != 0 */
int mfc_nchw_to_nhwc(const float* src, void* dst, int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W,
                     int32_t Cp, int32_t c_off, int32_t zero_pad, void* stream);
int mfc_nhwc_to_nchw(const void* src, float* dst, int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W,
                     int32_t Cp, void* stream);

/* Head input gather = x4 bilinear up-sampling of the T per-frame logit maps (hrnet.py:473-474)
 * + channel concat with optical flow and depth (multiframe_model.py:462-469), optionally with the
 * MultiFrameNetBasic flow warp (multiframe_model.py:89-170, incl. the 576x720 grid quirk).
 *   xh[b, h, w, t*nc + k] = up4(logits[t*B + b])[h, w, k]      (warped by flow_{t-1} if warp)
 *   then 2(T-1) flow channels (Large only) and T depth channels (warped if warp).          */
typedef struct {
    const void* logits;      /* T [T*B, Hs, Ws, Lp] */
    const float* flow[8];    /* T-1 NCHW fp32 [B,2,H,W] or all NULL */
    const float* depth[8];   /* T NCHW fp32 [B,1,H,W] or all NULL */
    void* xh;                /* T [B, H, W, Cp] */
    int32_t dtype, B, T, nc, Hs, Ws, Lp, H, W, Cp;
    int32_t warp;            /* 1: MultiFrameNetBasic warp (flow consumed, not concatenated) */
} mfc_headgather_desc;
int mfc_head_gather_fwd(const mfc_headgather_desc* d, void* stream);
/* dlogits[t*B+b] (overwritten) = adjoint of the above wrt logits; `xh` is d(xh).  With warp != 0 the caller passes an
 * fp32 scratch [B, H, W, (T-1)*nc] in depth[7] (the adjoint of grid_sample is a scatter; the call zeroes it). */
int mfc_head_gather_bwd(const mfc_headgather_desc* d, void* dlogits, void* stream);

/* ------------------------------------------------------------------------------------
 * Loss: F.log_softmax(dim=1) + class-weighted NLL + soft-Jaccard, forward and gradient
 * wrt logits (src/engine.py:65-66, src/loss.py:31-63).  acc: DEVICE fp32[MFC_LOSS_ACC_FLOATS], 8-byte aligned; [0..31] hold the results
 * below, [32..95] are scratch of mfc_loss_partial (the 26 sums as fp64 cells, so that the totals do not depend on the order the workgroups add
 * in; mfc_loss_finalize and mfc_loss_bwd read [0..31] only):
 *   [0] sum w_t*(-logp_t)  [1] sum w_t  [2+c] I_c = sum p_c*[t==c]  [10+c] sum p_c  [18+c] sum [t==c]   (c < 8)
 *   [26] nll  [27] soft-jaccard  [28] w_nll*nll + w_jac*jaccard
 *   [29] number of targets outside [0, nc): such pixels contribute nothing to the NLL term (and count as "no class" in the Jaccard
 *        sums) instead of indexing out of bounds; nn.NLLLoss raises on them -- callers that want that check acc[29] (mfc_loss(check_labels=True))
 * mfc_loss_fwd = mfc_loss_partial (zeroes acc, accumulates [0..25] over this call's B clips) + mfc_loss_finalize
 * ([26..28] from the sums).  Data-parallel ranks all-reduce (SUM) acc[0..25] between the two calls: the reference
 * evaluates both terms over the whole gathered batch (engine.py:64-66 runs after DataParallel's gather; the Jaccard
 * I/U sums span the batch, loss.py:57-58).  mfc_loss_bwd reads the (global) sums from acc.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const float* logits;     /* NCHW fp32 [B, nc, H, W] */
    const int64_t* target;   /* [B, H, W] */
    const float* class_w;    /* [nc] */
    float* acc;              /* fp32 [32] workspace / result block */
    float* dlogits;          /* NCHW fp32 [B, nc, H, W] (bwd) */
    int32_t B, nc, H, W;
    float w_nll, w_jac, grad_scale;
    int32_t pad_;
    const float* grad_scale_dev; /* optional (bwd): DEVICE scalar multiplied into the gradient on top of grad_scale -- the upstream gradient autograd hands to
                                  * the loss node (1, or the loss scale of an fp16 step) without a host read or a separate tensor pass */
} mfc_loss_desc;
int mfc_loss_fwd(const mfc_loss_desc* d, void* stream);
int mfc_loss_partial(const mfc_loss_desc* d, void* stream);
int mfc_loss_finalize(const mfc_loss_desc* d, void* stream);
int mfc_loss_bwd(const mfc_loss_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * Validation metrics on device.  Replaces `outputs.data.cpu().numpy().argmax(axis=1)` +
 * calculate_confusion_matrix_from_arrays (src/metrics.py:6-8,60-67) of every validation batch
 * (src/engine.py:139): per-sample confusion counts conf[b][truth][prediction] (int64, overwritten);
 * argmax takes the FIRST maximum like numpy.  IoU (first sample only -- get_jaccard's `[0]`,
 * metrics.py:41-45) and Dice (whole batch, :47-48) follow from the counts on the host.
 * ------------------------------------------------------------------------------------ */
int mfc_confusion_counts(const float* outputs /* NCHW fp32 [B, nc, H, W] */, const int64_t* target /* [B, H, W] */,
                         int64_t* conf /* [B][nc][nc] */, int32_t B, int32_t nc, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------------------------
 * Adam over flat fp32 arenas (torch.optim.Adam semantics, no weight decay, no amsgrad;
 * scripts/train_multiframe_detection.py:128-151 uses two lr groups = two segments).
 * ------------------------------------------------------------------------------------ */
int mfc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int32_t step, float grad_scale, void* stream);
/* Overflow guard of a loss-scaled (fp16) step, without a host round trip: mfc_grad_check sets flag[0] = 1 (after clearing it on the
 * stream) iff some element of g is Inf or NaN and counts such steps in flag[1]; mfc_adam_step_guarded is mfc_adam_step that leaves
 * p, m, v untouched when *skip_flag != 0 (what torch.cuda.amp.GradScaler.step does on the host).  flag: DEVICE int32[2]. */
int mfc_grad_check(const float* g, int64_t n, int32_t* flag, void* stream);
int mfc_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int32_t step, float grad_scale, const int32_t* skip_flag, void* stream);

/* ------------------------------------------------------------------------------------
 * Program interpreter: runs a whole forward or backward pass (hundreds of the calls above)
 * from ONE host call, so the step is not bound by Python launch overhead and can be captured
 * in a hipGraph.  Each record is {kind, payload}; payload is the matching *_desc.
 * ------------------------------------------------------------------------------------ */
typedef enum {
    MFC_OP_CONV = 1, MFC_OP_WGRAD = 2, MFC_OP_BNFIN = 3, MFC_OP_COMBINE = 4, MFC_OP_BNBWD_REDUCE = 5,
    MFC_OP_BNBWD_FIN = 6, MFC_OP_BNBWD_APPLY = 7, MFC_OP_MASK_ADD = 8, MFC_OP_HEAD_FWD = 9, MFC_OP_HEAD_BWD = 10,
    MFC_OP_BIAS_GRAD = 11, MFC_OP_MEMSET = 12, MFC_OP_PACK = 13, MFC_OP_UNPACK = 14, MFC_OP_NCHW2NHWC = 15,
    MFC_OP_NHWC2NCHW = 16,
    MFC_OP_WGRAD_BATCH = 17,     /* raw.a = HOST pointer to an mfc_wgrad_desc array (kept alive by the caller), raw.i[0] = n */
    MFC_OP_BNFIN_BATCH = 18,     /* raw.a = DEVICE pointer to an mfc_bnfin_desc array, raw.i[0] = n, raw.i[1] = max Cp */
    MFC_OP_WSNORM = 19,          /* raw.a = w, raw.b = w_out, raw.i[0] = Cout, raw.i[1] = per_out, raw.i[2] = eps (float bits) */
    MFC_OP_GNFIN = 20,           /* gnfin */
    MFC_OP_UPNEAR = 21,          /* raw.a = src, raw.b = dst, raw.i = dtype, N, H, W, Cp */
    MFC_OP_GNBWD_FIN = 22,       /* gnbwdfin */
    MFC_OP_WSBWD = 23,           /* raw.a = w, raw.b = dws, raw.c = dw, raw.i[0] = Cout, raw.i[1] = per_out, raw.i[2] = eps (float bits) */
    MFC_OP_UPNEAR_BWD = 24,      /* raw.a = dsrc, raw.b = ddst, raw.i = dtype, N, H, W, Cp, accumulate */
    MFC_OP_JOIN = 25             /* no launch: a lane-0 record that ends a parallel section (all side lanes are joined into the main stream) where the next
                                  * records, on lanes again, depend on several lanes of the section before (module fuse sums, hrnet.py:245-260) */
} mfc_op_kind;

#define MFC_LANE_ASYNC 0x100
typedef struct {
    int32_t kind;
    int32_t lane;            /* 0: serial.  >= 1: member of a parallel section (a maximal run of such records): lane 1 runs on
                              * the caller's stream, lanes 2..8 on side streams forked / joined with events; records of
                              * different lanes of one section must be independent (e.g. the branches of one HRNet module).
                              * | MFC_LANE_ASYNC: detached -- ordered after everything issued so far on its stream, runs on an
                              * extra stream, joined before the next MFC_OP_UNPACK / at the program end (weight gradients) */
    union {
        mfc_conv_desc conv;
        mfc_wgrad_desc wgrad;
        mfc_bnfin_desc bnfin;
        mfc_gnfin_desc gnfin;
        mfc_combine_desc combine;
        mfc_bnbwd_desc bnbwd;
        mfc_bnbwdfin_desc bnbwdfin;
        mfc_gnbwdfin_desc gnbwdfin;
        mfc_maskadd_desc maskadd;
        mfc_headgather_desc head;
        struct { mfc_headgather_desc d; uint64_t dlogits; } headbwd;
        struct { uint64_t a, b, c; int64_t n; int32_t i[12]; } raw;   /* generic small ops */
        uint8_t bytes[248];
    } u;
} mfc_op;
/* runs ops[0..n); returns 0 or (-(1000*index) + status) of the first failing record */
int mfc_program_run(const mfc_op* ops, int32_t n, void* stream);
/* per-call options (nothing process-wide is touched): MFC_RUN_DEFER_JOIN -- the program does not join the detached stream at its end; the next
 * program of the same step continues on it and the last one joins everything (backward segments of a data-parallel step, see mfc_wait_detached) */
#define MFC_RUN_DEFER_JOIN 1u
int mfc_program_run_ex(const mfc_op* ops, int32_t n, void* stream, uint32_t run_flags);
/* Interpreter state as a HANDLE (round 4; SURVEY.md 8(b) "re-entrant per stream; no hidden globals"): the side streams, fork / join events and
 * the pending-join state of the detached stream live in an mfc_ctx instead of the per-device default the two calls above use.  One context per
 * model (mfcnet_amd creates one per module instance): two models never share interpreter state, and two host threads may each drive their
 * own context -- the calling thread must have the context's device current (MFC_ERR_INVALID_ARG otherwise).  A context is not thread-safe
 * itself: one thread at a time per context.  Streams / events are created on the first run. */
int mfc_ctx_create(int32_t device, void** ctx_out);
int mfc_ctx_destroy(void* ctx);
int mfc_program_run_ctx(void* ctx, const mfc_op* ops, int32_t n, void* stream, uint32_t run_flags);
int mfc_wait_detached_ctx(void* ctx, void* stream);      /* mfc_wait_detached for a context's detached stream */
/* hipGraph form of a program: capture once (the program must have run once before; `stream` must not be the null
 * stream; section lanes / detached records become parallel graph branches), replay with mfc_graph_launch.  All pointers in
 * the records are baked into the graph, which is what the static plan guarantees.  Measured on MI355X / ROCm 7.2: no faster
 * than mfc_program_run (the step is GPU-bound and the host already runs ahead), so the Python layer does not use it. */
/* Backward segments (data parallel): with mfc_set_flag(28, 1) a program does not join the detached stream at its end -- the next
 * program of the step continues on it and the last one joins everything.  mfc_wait_detached makes `stream` wait for the detached
 * records issued so far (e.g. the stream an all-reduce of freshly unpacked gradients is ordered after). */
int mfc_wait_detached(void* stream);
int mfc_graph_capture(const mfc_op* ops, int32_t n, void* stream, void** exec_out);
int mfc_graph_launch(void* exec, void* stream);
int mfc_graph_destroy(void* exec);
/* tuning aid: the same, with a HIP event between records (each record launched `reps` times back to back);
 * synchronises the stream and returns the stream time per record and repetition in ms_out[n] */
int mfc_program_profile(const mfc_op* ops, int32_t n, int32_t reps, float* ms_out, void* stream);
/* In-process kernel timing with HIP events on the launch stream (bench.py's `roofline` object).
 * While enabled, every kernel launch of the entry points above is bracketed by two events recorded on ITS stream;
 * mfc_prof_collect synchronises them and sums elapsed time, launches, algorithmic FLOPs and algorithmic bytes (the tensors a launch
 * must touch once) per kernel.  Rows are keyed by the kernel's NAME as rocprofv3 prints it (demangled, e.g.
 * "conv_wgrad_wave_kernel<3, 3, 2, 2, 4, 1>", "bnbwd_reduce_kernel<__bf16>"), so the rows are those of the rocprofv3 kernel-stats tables
 * under profiles/.  Only meaningful when every record runs on one stream (mfc_set_flag(9, 0)): with lanes a launch shares the GPU. */
typedef struct { char name[120]; double ms, flops, bytes; int64_t launches; } mfc_prof_entry;
int mfc_prof_enable(int on);
int mfc_prof_dump(const char* csv_path);                     /* tuning aid: the recorded launches as a timeline (name, stream, start_us, end_us); clears the log */
int mfc_prof_collect(mfc_prof_entry* out, int32_t cap);     /* synchronises the recorded events, fills out[0..n), clears the log; returns n (< 0: error) */

/* The tuning switches (mfc_set_flag and its table) are NOT part of this interface: they are declared in mfcnet_hip_tuning.h, which only the
 * measurement scripts under tools/ and the tests include.  No entry point of the product path changes a switch. */
int mfc_op_size(void);      /* sizeof(mfc_op), so the host side can check its mirror */
const char* mfc_version(void);

#ifdef __cplusplus
}
#endif
#endif
