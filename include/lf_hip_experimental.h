/* liblf_hip.so -- EXPERIMENTAL entry points: kernels and switches that exist for in-process A/B measurements (tools/,
 * profiles/) or that a measured alternative has superseded.  They are exported by the same shared object and bound by
 * latentfusion_amd/experimental.py; nothing on a default path or a documented preset calls them, and a maintainer binding
 * the drop-in does not need this header.  Conventions as in lf_hip.h.
 */
#ifndef LF_HIP_EXPERIMENTAL_H
#define LF_HIP_EXPERIMENTAL_H

#include "lf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-variant switch for in-process A/B measurements (tools/, profiles/); results are equivalent within the test
 * tolerances for every value.  key 1: 3-D resampler kernels, 1 = generic (64-bit addressing, any C), 2 = lean
 * (32-bit buffer addressing, C % 4 == 0, volumes < 4 GB per sample; other shapes take the generic ones),
 * 3 = lean + the 16-channel specialisation of the gather (default), 4 = LDS-staged source footprint (16 channels; measured
 * slower, profiles/r03_resample_staged_ab.txt), 5 = gather with the map evaluated once per voxel (bit-identical to 3, 6 % slower).  key 2: lean coefficient-gradient kernel, sub-tiles in
 * flight per workgroup iteration: 1 = one, 2 = two (round-2 default), 3-5 = register-capped forms of 2 / 1, 6-11 = forms that do the
 * per-voxel arithmetic once per voxel instead of once per lane (10 = default: two gather passes in flight, gradient records
 * requested one sub-tile ahead).  key 3: workgroup
 * shape of lf_wino_fused_gemm (output channels x Winograd tiles): 0 = 64 x 64, 1 = 128 x 64, 2 = 64 x 128, 3 = 128 x 128,
 * 4 = 64 x 256 (3, 4: 2-D only), -1 = chosen from the problem shape (default).  key 4: lf_resample3d_bwd_vol_det, 1 = global
 * 64-bit atomics, 2 = source tiles accumulated in LDS (default; C == 16; bit-identical results; lf_resample3d_bwd_vol_det_io bins
 * the output voxels per tile when every sample has its own volume), 3 = as 2 without the binned form, 4 = as 2 with the tile pass that
 * gives a list entry 16 lanes instead of a lane quad (round-6 A/B: slower, profiles/r06_splat_ab.txt).  key 5: resident workgroups
 * per CU of lf_conv3d_c16_ring_bf16, 2 (default) or 3.  key 6: samples per pass of the binned splat at most (0 = default: as many
 * as 512 MB of lists hold; tests use 1 or 2 to walk the multi-pass path at small sizes).
 * Returns the previous value or LF_EINVAL. */
int lf_set_tuning(int key, int value);

/* d(loss)/d(sampled volume) of lf_resample3d_fwd with fp32 ATOMICS: order-dependent rounding (not bit-reproducible);
 * the product path uses lf_resample3d_bwd_vol_det (ops.DETERMINISTIC_SPLAT).  gvol must be zeroed by the caller. */
int lf_resample3d_bwd_vol(const float* gout, const float* coef, int kind, float* gvol, int vol_n,
                          int N, int D, int H, int W, int C, void* stream);

/* Factor projection backward fused into the first data-gradient convolution (measured slower than the two launches it
 * replaces: 1.45 vs 0.40 + 0.95 ms, profiles/r04_proj_fuse_ab.txt):
 * lf_conv3d_c16_wino_projbwd = lf_conv1x1_bwd_data(gp (N, H*W, 16), ..., prev = (act, act_norm, act_flags)) followed by
 *   lf_conv3d_c16_wino(data-gradient form on that volume, prev = (prev_y, prev_norm, prev_flags))
 * in ONE launch: the (N, 16, D, H, W) gradient volume between the two never exists; its halo planes are formed on chip
 * from `act` (the last camera block's saved output), `act_norm`, gp and the depth slices of the transposed projection.
 * proj_wtA: [D][64 lanes l][4 i] = Wp[cout = (l >> 4)*4 + i][k = d*16 + (l & 15)] (lf_conv3d_c16_wino_proj_pack_floats(D)
 * floats).  upack = the TRANSPOSED Winograd pack of the block's convolution.  Differs from the two-launch form only by the
 * reciprocal used for 1 / act_norm (<= 1 ulp). */
int lf_conv3d_c16_wino_projbwd(const float* gp, const float* proj_wtA, float proj_he, const float* act,
                               const float* act_norm, unsigned act_flags, const float* upack, float* y,
                               int N, int D, int H, int W, float he, float slope, const float* prev_y,
                               const float* prev_norm, unsigned prev_flags, void* stream);

/* Winograd F(2x2x2,3x3x3) with split-precision products: transforms in fp32, every Winograd-domain product
 * from three v_mfma_f32_16x16x16_f16 (U_hi.V_hi + U_hi.V_lo + U_lo.V_hi, fp32 accumulate).  Same semantics
 * as lf_conv3d_c16_wino; amax_in / amax_out as in lf_conv3d_c16_split.
 * upack: lf_conv3d_c16_wino_split_upack_halfs() f16 values, [4 a][16 b*4+c][hi, lo][64 lanes l][4 j]
 *        = split of U[a][b][c][cout = l & 15][cin = (l >> 4) * 4 + j]. */
size_t lf_conv3d_c16_wino_split_upack_halfs(void);
int lf_conv3d_c16_wino_split(const float* x, const void* upack, const float* bias, float* y, float* norm_out,
                             int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                             const float* prev_y, const float* prev_norm, unsigned prev_flags,
                             const float* amax_in, float* amax_out, void* stream);

/* bf16-autocast form of the fused 16 -> 16 conv3d block step, for the training step (BASELINE cfg 5; the reference wraps
 * Sculptor / Photographer.forward in `autocast(enabled=self.training)`, recon/models.py:199,405):
 * operands rounded to bf16 (RNE) while the halo is staged, products on v_mfma_f32_16x16x16_bf16 with fp32 accumulation,
 * DIRECT convolution (a Winograd transform of bf16 data is not bf16-exact).  round_out: 0 = fp32 epilogue on the fp32
 * accumulator; 1 = autocast forward, y = epilogue(bf16(bf16(acc) * he) + bias) -- the convolution returns a half tensor and
 * `* he` stays in half, the fp32 bias promotes the rest (modules/equalized.py:57-64); 2 = autocast data gradient
 * (transposed / flipped pack, flags = 0, bias = NULL): the result is additionally rounded to bf16.
 * wpack: lf_conv3d_c16_bf16_wpack_elems() bf16 values, [27 taps][64 lanes l][4]: W[cout = l & 15][cin = (l >> 4)*4 + i][tap].
 * Superseded by lf_conv3d_c16_ring_bf16 (lf_hip.h), which is what the training step runs. */
size_t lf_conv3d_c16_bf16_wpack_elems(void);
int lf_conv3d_c16_bf16(const float* x, const void* wpack, const float* bias, float* y, float* norm_out, int N, int D, int H,
                       int W, float he, unsigned flags, float slope, float eps, int round_out, void* stream);

/* Stage 3 of the THREE-stage wide Winograd convolution (input transform, per-frequency library GEMMs M[f] = V[f] @ U[f]
 * on the host side, this output transform with the fused epilogue): the A/B reference of lf_wino_fused_gemm
 * (experimental.WIDE_CONV_MODE = 'bmm').  M [F][T][Cout]; PixelNorm is fused for Cout <= 256. */
int lf_wino3d_output_transform(const float* M, const float* bias, float* y, float* norm_out, int N, int D, int H, int W,
                               int C, float he, unsigned flags, float slope, float eps, void* stream);

int lf_wino2d_output_transform(const float* M, const float* bias, float* y, float* norm_out, int N, int H, int W,
                               int C, float he, unsigned flags, float slope, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
