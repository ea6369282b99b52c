/*
 * Tuning switches of libmfcnet_hip.so -- NOT part of the product interface (include/mfcnet_hip.h): process-global, not thread-safe,
 * read by the launch-geometry searches and the program interpreter.  The defaults are the measured optima; only the measurement
 * scripts under tools/, bench.py's MFC_* environment overrides (A/B runs) and tests that compare two settings include this header.
 */
#ifndef MFCNET_HIP_TUNING_H
#define MFCNET_HIP_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif
/* mfc_set_flag(id, value): 
 *   1  wgrad: use ds_read_b64_tr_b16 (1)            2  conv: force pixel-tile MT (0 = search)     3  wgrad: K-split mode (0)
 *   4  conv: persistent workgroups per launch (512)  5  conv: ablation mask (0)                    6  conv: LDS budget KiB (80)
 *   7  wgrad: ablation mask (0)                      8  conv: cout-block-fastest unit order (-1 auto)
 *   9  lanes: bit 0 parallel-section lanes, bit 1 detached records (3); 0 = every record on the caller's stream
 *  10  detached streams in use (1)                  11  wgrad: target workgroups per launch (256; sizes the partial-sum slices)
 *  12  lane -> stream folding (n streams, or a 4-digit map such as 1221)   13  run detached records on side lane k (0 = own stream)
 *  14  program main stream = interpreter's own (0)  15  what-if: skip record kinds (bit mask, timing only)
 *  16  detached stream priority (0; read at stream creation)              17  wgrad: prefetch-distance-2 variant (0)
 *  18  conv: exponent (%) of the under-filled-launch penalty (100)         19  conv: score weight (%) of the 8-wave geometries (90)
 *  20  conv: single-stage launches keep every cout block's weights in LDS and stage each pixel tile once (1)
 *  21  wgrad: output pixels per workgroup above which the pixel axis is split further than switch 11 asks (6000; 0 = never)
 *  22  lanes: measure which side streams really overlap with the caller's stream before choosing them (1; see runtime.hip)
 *  23  conv: big 1x1 / stride-1 convolutions without input transform run as a plain GEMM (conv_gemm1x1.hip) (1)
 *  24  conv: smallest Cin and Cout sent to that GEMM (128)
 *  25  wgrad: 1x1 weight gradients without input transform as a split-K GEMM (wgrad_gemm1x1.hip) (1)
 *  26  wgrad: smallest Cin and Cout sent to that GEMM (64)           27  BN-backward reduce: workgroups per launch (1024; 512-2048 measured equal)
 *  28  program: defer the final join of the detached stream to the next program (0; tuning only -- the product path passes
 *      MFC_RUN_DEFER_JOIN to mfc_program_run_ex instead)
 *  29  wgrad: 3x3 / stride-1 weight gradients of 32-channel-multiple layers through the LDS-DMA ring kernel (conv_wgrad_dma.hip) (1)
 *  30  conv: 32 -> 32 / 64 -> 64 3x3 stride-1 convolutions through the register-resident-weight ring kernel (conv3x3_ring.hip) (1)
 *  31  ring kernel: 16-pixel rows per wave (4; 2)    32  ring kernel: ablation mask (0)             33  ring kernel: workgroups per CU (2)
 *  34-36  (round 3's conv3x3_stream.hip: removed in round 4 -- measured slower)
 *  37  ring kernel: start delay of the workgroup in the odd wave slot of a CU (0; no effect measured)
 *  38  wgrad DMA kernel: 8 waves per workgroup for launches with an input transform (0: faster alone, slower in the step)
 *  39  fused BatchNorm-backward finalize + apply: workgroups per launch (1024; 2048 / 4096 measured slower)
 *  40  element-wise ablation mask, timing only (0)   41  BN-backward reduce: threads per workgroup (256; 512 / 1024 slower)
 *  42  BN-backward reduce: fewest pixels per thread (8)
 *  46  wgrad: 3x3 / stride-2 weight gradients through the LDS-DMA ring kernel (conv_wgrad_dma_s2.hip) (1)
 *  47  wgrad: 3x3 / stride-1 weight gradients with channel counts that are multiples of 48 but not of 32 through the shared-ring kernel
 *      (conv_wgrad_dma48.hip) (1)                    48  that kernel: twice the workgroups and partial-sum slices per launch (0)
 *  50-51  (round 3's conv3x3_ring48.hip: removed in round 4 -- measured slower)
 *  52  ring kernel: workgroups per launch (0 = 256 x switch 33; probes of partitioned grids, tools/probe_group.py)
 *  53  descriptor hardening (0): 1 = every non-null pointer of a convolution / weight-gradient / combine / BatchNorm-backward / mask-add descriptor is checked with
 *      hipPointerGetAttributes before the launch (MFC_ERR_INVALID_ARG for host or unmapped addresses instead of a GPU fault); the -m gpu tests run with it on
 *  54  write-through (sc1) 16-byte output stores from this many MB of output per launch (12; 0 = plain stores everywhere): mfcnet-tracker_amd/csrc/common.h, mfc_st16
 *  55  conv: score weight (%) of the 8-wave geometries for launches that want the fused data-gradient epilogue (90; 0: 4-wave forms; -1 = switch 19)
 *  56  lanes: in a program without detached records (the forward pass) lane 4 runs on the detached stream's hardware queue (1)
 *  57  ring kernel: thousands of pixels (N*H*W) from which a 64-channel data gradient takes the unfused ring launch + a reduce pass (100) */
int mfc_set_flag(int id, int value);
#ifdef __cplusplus
}
#endif
#endif
