/*
 * lf_hip.h -- C ABI of liblf_hip.so: the MI355X (gfx950) kernels behind the LatentFusion
 * reconstruct-and-render hot path.
 *
 * The reference (NVlabs/latentfusion) has no FFI layer: its hot path is a chain of ATen calls
 * issued from nn.Module.forward.  Each entry point below replaces one such chain; the
 * reference call sites are cited per function as latentfusion/<file>:<lines>.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise;
 *   - activations are channels-last: volumes [N][D][H][W][C], images [N][H][W][C]
 *     (the Python host mirror converts from/to the reference's NCDHW at its API boundary);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *     LF_E* code for rejected arguments (nothing is launched in that case);
 *   - no function keeps a pointer past its return; no hidden allocations: scratch space is
 *     passed in explicitly (see lf_*_scratch_bytes).
 */
#ifndef LF_HIP_H
#define LF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_ABI_VERSION 1

#define LF_EINVAL   (-1)   /* bad size / flag combination            */
#define LF_EALIGN   (-2)   /* pointer or channel count not aligned   */
#define LF_ENOSPC   (-3)   /* scratch buffer too small               */

/* epilogue flags of the conv entry points */
#define LF_EPI_LRELU     1u   /* y = max(y, slope*y)                                         */
#define LF_EPI_PIXELNORM 2u   /* y = y / sqrt(mean_c(y^2) + eps); also writes the norm       */
#define LF_EPI_ADD       4u   /* (prev_flags of lf_conv3d_c16_wino only) prev_y is an addend  */
#define LF_EPI_DOT       8u   /* (prev_flags of lf_conv1x1_bwd_data only, alone) gx is stored as it is; prev_norm is an OUTPUT:
                                 prev_norm[record] = sum_c gx[record][c] * prev_y[record][c] per 16-channel record            */

/* grid-map kinds of the 3-D resampler */
#define LF_MAP_O2C 0   /* bilinear polynomial map (ObjectToCameraTransform)                  */
#define LF_MAP_C2O 1   /* projective map (CameraToObjectTransform)                           */
#define LF_MAP_COEFS 20 /* floats per sample in the coefficient block of either kind         */

int lf_abi_version(void);

/* Name of the device the library is running on (for logs); returns 0 / hip error. */
int lf_device_name(char* buf, int buflen);

/* ------------------------------------------------------------------------------------------
 * 3-D resampling: out[n,z,y,x,:] = trilinear(vol[n or 0], g(n; x,y,z)), padding=border,
 * align_corners=False, i.e. F.grid_sample as used by
 *   ObjectToCameraTransform.forward   modules/geometry.py:669-690  (+ camera_coords :515-531)
 *   CameraToObjectTransform.forward   modules/geometry.py:625-657
 * The sampling grid is never materialised; it is evaluated per voxel from `coef`:
 *   LF_MAP_O2C: g = c0 + c1*a + c2*b + c3*k + c4*a*k + c5*b*k, (a,b,k) = (x/(W-1), y/(H-1),
 *               z/(D-1)), c_j = coef[n][3*j .. 3*j+2] (x,y,z components)        -> 18 floats
 *   LF_MAP_C2O: l = (lx,ly,lz,1) with lx = -1 + 2x/(W-1) (lattice in [-1,1]),
 *               gx = (A0.l)/(A3.l), gy = (A1.l)/(A3.l), gz = A2.l, A_r = coef[n][4*r..] -> 16 floats
 * vol_n == 1 broadcasts one volume to all N samples (Photographer.decode, recon/models.py:489-494).
 * Output spatial size == input spatial size (D,H,W), channels C.
 */
int lf_resample3d_fwd(const float* vol, int vol_n, const float* coef, int kind,
                      float* out, int N, int D, int H, int W, int C, void* stream);

/* d(loss)/d(coef) for LF_MAP_O2C given gout = d(loss)/d(out): gcoef[n][18] (fixed-order,
 * deterministic two-stage reduction).  Replaces grid_sampler_3d_backward (grid part) plus the
 * autograd chain grid -> camera of modules/geometry.py:469-531,669-686.
 * scratch must hold lf_resample3d_bwd_coef_scratch_bytes(N,D,H,W) bytes. */
size_t lf_resample3d_bwd_coef_scratch_bytes(int N, int D, int H, int W);
int lf_resample3d_bwd_coef(const float* gout, const float* vol, int vol_n, const float* coef,
                           float* gcoef, void* scratch, size_t scratch_bytes,
                           int N, int D, int H, int W, int C, void* stream);

/* d(loss)/d(vol) (trilinear splat, fp32 atomics; training / encoder backward only).
 * gvol must be zero-initialised by the caller; with vol_n == 1 all samples accumulate into one
 * volume. */

/* The same gradient, DETERMINISTIC: contributions are accumulated as round(value * 2^K) with 64-bit integer atomics
 * (integer addition is associative, so the result does not depend on the order the hardware retires them) and
 * converted back in a second pass; K follows max|gout| (found by an order-independent atomic max) so that 2^24
 * contributions to one voxel cannot overflow -- the quantum is 2^-38 of the largest gradient.  gvol is overwritten.
 * scratch: lf_resample3d_bwd_vol_det_scratch_bytes(vol_n, D, H, W, C) bytes, 8-byte aligned. */
size_t lf_resample3d_bwd_vol_det_scratch_bytes(int vol_n, int D, int H, int W, int C);
int lf_resample3d_bwd_vol_det(const float* gout, const float* coef, int kind, float* gvol, int vol_n, void* scratch,
                              size_t scratch_bytes, int N, int D, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * He-equalised 3x3(x3) convolution with fused epilogue, stride 1, zero padding 1:
 *   y = conv(x, W) * he + bias ; [LeakyReLU(slope)] ; [PixelNorm over channels (eps)]
 * = Equalized.forward + nn.LeakyReLU + PixelNorm of one half of Block.forward
 *   modules/equalized.py:57-64, modules/blocks.py:152-158, modules/__init__.py:14-15.
 * dims = 3: x [N][D][H][W][Cin];  dims = 2: D must be 1.
 * wpack: weights re-laid-out by the host as [tap][CoutP][CinP], tap = (kz*3+ky)*3+kx (dims=2:
 *        ky*3+kx), CinP = Cin rounded up to 16, CoutP = lf_conv3x3_cout_padded(Cout), zero padded.
 * bias: [Cout] or NULL.  norm_out: [N*D*H*W] receives sqrt(mean_c(y^2)+eps) when
 * LF_EPI_PIXELNORM is set (needed by the backward), may be NULL otherwise.
 * LF_EPI_PIXELNORM requires Cout <= 64 here; wider layers use lf_pixelnorm_fwd afterwards.
 */
/* Padded cout count the host packer must use for wpack (multiple of the cout-tile group). */
int lf_conv3x3_cout_padded(int Cout);
int lf_conv1x1_cout_padded(int Cout);

int lf_conv3x3_fwd(const float* x, const float* wpack, const float* bias, float* y, float* norm_out,
                   int dims, int N, int D, int H, int W, int Cin, int Cout,
                   float he, unsigned flags, float slope, float eps, void* stream);

/* Pointwise (1x1 / 1x1x1) He-equalised convolution with the same epilogue, i.e. a GEMM over
 * pixels.  Used for InputBlock/OutputBlock (modules/blocks.py:78-133), FactorProjection2d3d
 * and FactorProjection3d2d (modules/geometry.py:711-749).
 * Input addressing is generalised so that the depth axis of a channels-last volume can be
 * folded into K without a copy: K = ksl * Cin; element (pixel p, k = s*Cin + c) is read from
 * x[n*x_batch_stride + s*x_slice_stride + p*Cin + c].  wpack: [CoutP][Kp] (K ordered s-major,
 * Kp = K rounded up to 16, CoutP = lf_conv1x1_cout_padded(Cout), zero padded).
 * Output addressing is generalised the same way: channel co of pixel p of sample n is written to
 *   y[n*y_batch_stride + (co / y_slice_channels)*y_slice_stride + p*y_row_stride + co % y_slice_channels]
 * (y_slice_channels >= Cout: plain channels-last rows; y_slice_channels = C: the output is
 * unfolded into Cout/C depth slices of a channels-last volume -- used by the projection backward).
 * LF_EPI_PIXELNORM requires Cout <= 128 here. */
int lf_conv1x1_fwd(const float* x, const float* wpack, const float* bias, float* y, float* norm_out,
                   int N, int P, int Cin, int ksl, long x_batch_stride, long x_slice_stride,
                   int Cout, long y_batch_stride, int y_row_stride, int y_slice_channels,
                   long y_slice_stride,
                   float he, unsigned flags, float slope, float eps, void* stream);
/* The same with every input slice scaled by a per-pixel factor first: element (p, k = s*Cin + c) is
 * x[...] * xscale[(n*ksl + s)*P + p], the product rounded to fp32 as a separate scaling pass would round it (Cin % 4 == 0).
 * Replaces `z = z * depth_weights_resized` ahead of the factor projection (reference recon/models.py:427-430 -> modules/
 * geometry.py:744-749) without the scaled volume (round 6; was lf_column_scale_fwd + lf_conv1x1_fwd). */
int lf_conv1x1_fwd_scaled(const float* x, const float* xscale, const float* wpack, const float* bias, float* y, float* norm_out,
                          int N, int P, int Cin, int ksl, long x_batch_stride, long x_slice_stride,
                          int Cout, long y_batch_stride, int y_row_stride, int y_slice_channels,
                          long y_slice_stride,
                          float he, unsigned flags, float slope, float eps, void* stream);

/* Data gradients of the two convolutions: gx = conv^T(gy) * he with the host-packed transposed
 * weights (taps flipped / in-out swapped).  When prev_y != NULL the epilogue backward of the layer
 * that PRODUCED this convolution's input (whose saved output is prev_y, norm prev_norm, epilogue
 * prev_flags) is folded into the store, i.e. the result is already dL/d(pre-activation) of that
 * layer and no separate lf_epilogue_bwd pass over the volume is needed.  Requires 16-channel
 * records (Cout == 16 for 3x3; y_slice_channels == y_row_stride == 16 for 1x1). */
int lf_conv3x3_bwd_data(const float* gy, const float* wpack_t, float* gx, int dims, int N, int D, int H, int W,
                        int Cin, int Cout, float he, const float* prev_y, const float* prev_norm,
                        unsigned prev_flags, float slope, void* stream);
int lf_conv1x1_bwd_data(const float* gy, const float* wpack_t, float* gx, int N, int P, int Cin, int Cout,
                        long y_batch_stride, int y_row_stride, int y_slice_channels, long y_slice_stride,
                        float he, const float* prev_y, const float* prev_norm, unsigned prev_flags,
                        float slope, float* amax_out, void* stream);
/* (amax_out, optional: zero-initialised LF_AMAX_FLOATS-float buffer that receives max|gx| when prev_y != NULL;
 *  consumed by lf_conv3d_c16_split as amax_in.) */

/* Split-precision ("f16x3") variant of the fused conv3d 16->16 block and of its data gradient:
 * every fp32 operand is split on-chip into f16 hi + lo (22 mantissa bits) and the product is formed
 * from three f16 MFMAs accumulating in fp32 (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi; dropped term <= 2^-22).
 * Same semantics / epilogue / fused previous-layer backward as lf_conv3x3_fwd / lf_conv3x3_bwd_data
 * for dims = 3, Cin = Cout = 16.  wsplit: [14 tap pairs][hi,lo][16 cout][32 = 2 taps x 16 cin] f16 (pairing from
 * lf_conv3d_c16_split_pairs),
 * lf_conv3d_c16_split_wpack_halfs() elements (host-packed).  amax_in (device scalar, may be NULL):
 * max-abs of x, used to pre-scale tiny gradient tensors by a power of two (undone exactly);
 * amax_out (may be NULL, zero-initialised by the caller): receives max-abs of the output.
 * Both are LF_AMAX_FLOATS-float buffers of partial maxima (one slot per 128-byte line; the maximum over the
 * buffer is the value): producers spread their atomic maxima over the slots, consumers reduce them. */
#define LF_AMAX_FLOATS 2048
size_t lf_conv3d_c16_split_wpack_halfs(void);
void lf_conv3d_c16_split_pairs(int* taps28);   /* tap indices (kz*9+ky*3+kx; -1 = zero) of the 14 K-slot pairs */
int lf_conv3d_c16_split(const float* x, const void* wsplit, const float* bias, float* y, float* norm_out,
                        int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                        const float* prev_y, const float* prev_norm, unsigned prev_flags,
                        const float* amax_in, float* amax_out, void* stream);

/* Winograd F(2x2x2, 3x3x3) variant of the same fused 16 -> 16 conv3d step, all-fp32 arithmetic
 * (v_mfma_f32_16x16x4_f32 on G-transformed weights, B/A transforms on the fp32 VALU): 64 instead of
 * 216 multiplies per 2x2x2 outputs.  Same semantics / epilogue / fused previous-layer backward as
 * lf_conv3x3_fwd / lf_conv3x3_bwd_data for dims = 3, Cin = Cout = 16 (modules/blocks.py:152-158).
 * upack: lf_conv3d_c16_wino_upack_floats() floats, [4 z-freq a][16 (y,x)-freq b*4+c][4 k-chunks i][64 lanes l]
 *        = U[a][b][c][cout = l & 15][cin = (l >> 4) * 4 + i],  U = (G (x) G (x) G) w  with
 *        G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]  (host-packed, once per weight update).
 * amax_out (may be NULL, zero-initialised by the caller): receives max-abs of the output.
 * prev_flags == LF_EPI_ADD: forward form with an addend, y = epilogue(conv(x) * he + bias + prev_y), prev_y in
 * y's layout -- a convolution over concatenated inputs evaluated as a sum of 16-channel pieces (the ConvGRU
 * gates, modules/gru.py:37-42). */
size_t lf_conv3d_c16_wino_upack_floats(void);
int lf_conv3d_c16_wino(const float* x, const float* upack, const float* bias, float* y, float* norm_out,
                       int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                       const float* prev_y, const float* prev_norm, unsigned prev_flags,
                       float* amax_out, void* stream);

/* The renderer's factor projection fused into the Winograd kernel of the LAST camera block (round 4).
 * Reference: Photographer.decode runs camera_blocks then FactorProjection3d2d (recon/models.py:417-437,
 * modules/geometry.py:731-749: view (N, C*D, H, W) -> 1x1 conv -> LeakyReLU -> PixelNorm); a workgroup of
 * lf_conv3d_c16_wino walks exactly the axis that projection contracts.
 *
 * lf_conv3d_c16_wino_projfwd = lf_conv3d_c16_wino (forward form, prev_y = NULL) followed by
 *   lf_conv1x1_fwd(y as (N, H*W, K = D*16), ..., Cout = 16, proj_he, proj_flags)  ->  zp (N, H, W, 16), pnorm (N*H*W or NULL)
 * in ONE launch: same operands, same accumulation order, bit-identical zp; y / norm_out are still written (the backward
 * pass needs them).  proj_wA: lf_conv3d_c16_wino_proj_pack_floats(D) floats, [D][64 lanes l][4 i]
 *   = Wp[cout = l & 15][k = d*16 + (l >> 4)*4 + i]   (Wp = the (16, D*16) matrix lf_conv1x1_fwd takes, depth-major K).
 */
size_t lf_conv3d_c16_wino_proj_pack_floats(int D);
int lf_conv3d_c16_wino_projfwd(const float* x, const float* upack, const float* bias, float* y, float* norm_out,
                               int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                               const float* proj_wA, const float* proj_bias, float* zp, float* pnorm,
                               float proj_he, unsigned proj_flags, void* stream);

/* bf16 autocast policy of the training step (BASELINE cfg 5; the reference wraps Sculptor / Photographer.forward in
 * `autocast(enabled=self.training)`, recon/models.py:199,405).
 * lf_round_bf16: y = bf16(x) kept in fp32 containers (the autocast cast of an operand of the fp32-MFMA kernels: bf16 x bf16
 * products are exact in fp32, so those kernels then compute what a bf16 MFMA would). */
int lf_round_bf16(const float* x, float* y, long n, void* stream);
/* The fused 16 -> 16 conv3d block step under that policy: operands rounded to bf16 (RNE) while the halo is staged, products on
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation, DIRECT convolution (a Winograd transform of bf16 data is not bf16-exact),
 * on the ring organisation of lf_conv3d_c16_split (column walk over 2 x 8 x 16 tiles, six z-plane slots
 * of bf16 halo in LDS, tap pairs per MFMA, the epilogue under the next tile's MFMAs).  round_out: 0 = fp32 epilogue on the
 * accumulator, 1 = y = epilogue(bf16(bf16(acc) * he) + bias) -- autocast's convolution returns a half tensor and `* he` stays in
 * half, the fp32 bias promotes the rest (modules/equalized.py:57-64); with flags = 0 and bias = NULL also its data gradient.  addend != NULL (bias = NULL, flags = 0,
 * round_out = 0): y = conv(x) * he + addend, the running sum of the ConvGRU gates evaluated without the channel
 * concatenation (modules/gru.py:31-40).
 * wpack: lf_conv3d_c16_ring_bf16_wpack_elems() bf16 values, [14 pairs][16 cout][32 = 2 taps x 16 cin], the tap pairs of
 * lf_conv3d_c16_split_pairs (a missing second tap = zeros). */
size_t lf_conv3d_c16_ring_bf16_wpack_elems(void);
int lf_conv3d_c16_ring_bf16(const float* x, const void* wpack, const float* bias, float* y, float* norm_out,
                            int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                            const float* addend, int round_out, void* stream);
/* The same launch with the STORAGE type of its three volumes selectable (round 5: the training step keeps the 16-channel
 * tensors that only half-precision convolutions produce and consume as bf16 channels-last records, 32 B per voxel --
 * half the HBM bytes of the same arithmetic: the staging rounds an fp32 input to bf16 anyway).  io = OR of LF_IO_*:
 * LF_IO_IN_BF16: x is bf16; LF_IO_OUT_BF16: y is written as bf16 (RNE of the fp32 epilogue result);
 * LF_IO_ADDEND_BF16: addend is bf16 (needs addend != NULL).  norm_out stays fp32.  io = 0 is lf_conv3d_c16_ring_bf16. */
#define LF_IO_IN_BF16 1
#define LF_IO_OUT_BF16 2
#define LF_IO_ADDEND_BF16 4
int lf_conv3d_c16_ring_bf16_io(const void* x, const void* wpack, const float* bias, void* y, float* norm_out,
                               int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                               const void* addend, int round_out, int io, void* stream);

/* MULTI-OUTPUT form of the bf16 ring convolution for the ConvGRU recurrence of the training step (round 6; reference
 * modules/gru.py:30-43 under recon/fusion.py:188-197, autograd of the same lines): ONE staged input volume x (bf16 records,
 * or fp32 with x_bf16 = 0) is convolved with `ngroups` (1 or 2) weight packs (wpack = the packs of
 * lf_conv3d_c16_ring_bf16 back to back) into `ngroups` outputs; per group g the epilogue is
 *     y_g = fma(conv_g, he, add_g)                       add_g fp32 or (LF_RING_ADD_BF16) bf16, NULL = no addend
 *     y_g = bf16(bf16(conv_g) * he)                      LF_RING_ROUND: what autocast's half-precision convolution returns
 * stored as fp32 or (LF_RING_OUT_BF16) bf16 records; addend_per_sample = 0: one addend volume for all N samples.
 * `extra` folds an element-wise stage of the recurrence into the epilogue of the convolution that owns the same voxels:
 *   LF_RING_EX_RH    (x = h: fp32, or its bf16 copy with e0 = the fp32 h; the LAST group = reset gate stored as bf16):
 *                    o2 = bf16(h * sigmoid(y_last as stored))
 *   LF_RING_EX_BLEND (ngroups 1, y_0 = candidate stored as bf16; e0 = h fp32, e1 = update pre-activation bf16):
 *                    o2 (fp32) = h (1 - u) + y_0 u,  u = sigmoid(e1)
 *   LF_RING_EX_ABWD  (group 0 = LF_RING_ROUND | LF_RING_ADD_BF16 with add_0 = reset pre-activation, NOT added;
 *                    e0 = h fp32, e1 = gh1 fp32): g = group 0's result is not stored; y_0 (bf16) = g h r (1 - r),
 *                    o2 (fp32) = e1 + g r,  r = sigmoid(add_0).  y_0 may alias add_0.
 *   LF_RING_EX_BLOCK (ngroups 1, flags LF_RING_ROUND | LF_RING_OUT_BF16; e0 = bias[16] fp32 or NULL, o2 = norms fp32 per voxel):
 *                    y_0 = PixelNorm(LeakyReLU(bf16(bf16(conv) * he) + bias)), slope 0.2, eps 1e-8: the Block step of
 *                    lf_conv3d_c16_ring_bf16_io (flags LRELU | PIXELNORM, round_out 1) with its epilogue at compile time.
 *   LF_RING_EX_PREV  (ngroups 1, bf16 x, flags LF_RING_ROUND | LF_RING_OUT_BF16: the data gradient of a 16 -> 16 layer) e0 = the
 *                    bf16 activation of the layer that PRODUCED this layer's input, e1 = its PixelNorm norms (fp32 per voxel):
 *                    y_0 = LeakyReLU'(PixelNorm'(g; e0, e1)) of the rounded result g, slope 0.2 -- that producer's pre-activation
 *                    gradient; o2 = fp32 scratch of 16 * (1 + 2 * #CUs) floats, zero-filled by the caller: o2[0..15] receives the
 *                    producer's bias gradient (sums of the un-rounded y_0, fixed order).  Autograd of modules/blocks.py:152-158.
 * In-place addends (y_g == add_g) are allowed.  Only the combinations the recurrence launches are instantiated (LF_EINVAL
 * otherwise): no extra with (bf16 x, 1 or 2 groups) or (fp32 x, 1 group); EX_RH and EX_ABWD with 1 or 2 groups; EX_BLEND with 1.
 * D*H*W*64 < 2^31. */
#define LF_RING_ADD_BF16 1u
#define LF_RING_OUT_BF16 2u
#define LF_RING_ROUND 4u
#define LF_RING_EX_NONE 0
#define LF_RING_EX_RH 1
#define LF_RING_EX_BLEND 2
#define LF_RING_EX_ABWD 3
#define LF_RING_EX_BLOCK 4
#define LF_RING_EX_PREV 5
int lf_conv3d_c16_ring_multi(const void* x, int x_bf16, const void* wpack, int ngroups,
                             void* y0, const void* add0, unsigned flags0, void* y1, const void* add1, unsigned flags1,
                             int extra, const void* e0, const void* e1, void* o2,
                             int N, int D, int H, int W, float he, int addend_per_sample, void* stream);
/* The LF_RING_EX_BLEND launch spelled out, with an optional second output: cand (bf16, may alias `addend`) = conv(rh) * he +
 * addend (bf16); h_new (fp32) = h (1 - u) + cand u, u = sigmoid(upre as stored, bf16); h_new_bf16 (or NULL) = the same state
 * rounded to bf16 -- what the next step's gate convolutions and the weight gradients stage, at half the bytes. */
int lf_conv3d_c16_ring_blend(const void* rh, const void* wpack, void* cand, const void* addend, const float* h, const void* upre,
                             float* h_new, void* h_new_bf16, int N, int D, int H, int W, float he, void* stream);

/* Winograd F(2x2x2,3x3x3) for wide (>= 64-channel) 3-D convolutions, stage 1: the input transform.  x channels-last;
 * V [64][T][Cin], T = lf_wino3d_tiles(N, D, H, W), frequency f = (a*4 + b)*4 + c (z, y, x).  Cin a multiple of 4.
 * Stages 2 + 3: lf_wino_fused_gemm below. */
long lf_wino3d_tiles(int N, int D, int H, int W);
int lf_wino3d_input_transform(const float* x, float* V, int N, int D, int H, int W, int C, void* stream);

/* Stages 2 + 3 of the wide Winograd convolution in ONE launch, on this library's own fp32-MFMA GEMM (no library GEMM on
 * the hot path): for every frequency f the product V[f] (T x Cin) . U2[f]^T is accumulated on v_mfma_f32_16x16x4_f32 and
 * folded straight into the 2^dims outputs of each tile (output transform A^T M A), then He scale, bias and LeakyReLU are
 * applied in the store -- M (8x / 4x the output) never reaches memory.  dims = 3: V [64][T][Cin] from
 * lf_wino3d_input_transform; dims = 2 (D = 1): V [16][T][Cin] from lf_wino2d_input_transform.
 * U2: [F][CoutP][Cin], CoutP = lf_wino_fused_cout_padded(Cout) (zero padded), U2[f][co][ci] = ((G (x) ..) w)[co][ci][f]
 * -- output-channel major so that both GEMM operands stream along the contraction axis.  Cin, Cout multiples of 4.
 * flags: LF_EPI_LRELU only; for PixelNorm run lf_pixelnorm_fwd on y afterwards (y is 1/8 resp. 1/4 of V);
 * | LF_OUT_DEPTH_INNER (dims = 3): y is written as [N][H][W][D][Cout] instead of [N][D][H][W][Cout] -- a pixel's depth column
 * of the last camera block becomes ONE contiguous row of D*Cout floats, so the factor projection 3-D -> 2-D that follows
 * (modules/geometry.py:744-749; K = D*Cout = 4096 in the released model) is a plain row-major GEMM.
 * Also the data gradient of the same convolution (transposed / flipped U2, flags = 0, bias = NULL).
 * Replaces Equalized.forward + LeakyReLU of modules/equalized.py:57-64, blocks.py:152-158 for >= 64-channel layers.
 * Problems with few tile / channel blocks are split over the frequencies (more workgroups): each part writes its raw
 * partial outputs to `scratch` (lf_wino_fused_scratch_bytes(...) bytes, 0 when no split is used) and a second launch
 * adds them in a fixed order and applies the epilogue -- deterministic, no atomics. */
#define LF_OUT_DEPTH_INNER 0x100u
int lf_wino_fused_cout_padded(int Cout);
size_t lf_wino_fused_scratch_bytes(int dims, int N, int D, int H, int W, int Cout);
int lf_wino_fused_gemm(const float* V, const float* U2, const float* bias, float* y, void* scratch, size_t scratch_bytes,
                       int dims, int N, int D, int H, int W, int Cin, int Cout, float he, unsigned flags, float slope,
                       void* stream);

/* 2-D counterpart, F(2x2,3x3): V [16][T][Cin], T = lf_wino2d_tiles(N, H, W), f = b*4 + c (y, x). */
long lf_wino2d_tiles(int N, int H, int W);
int lf_wino2d_input_transform(const float* x, float* V, int N, int H, int W, int C, void* stream);

/* Gate arithmetic of the convolutional GRU fuser, inference path (modules/gru.py:30-43; no tanh on the
 * candidate).  `rec` is the channels-last record [x | state] (rec_stride floats per voxel, state at
 * rec_off) that the gate convolutions read; upre / rpre = update / reset pre-activations, pre_stride floats
 * per voxel (one merged [update | reset] convolution: rpre = upre + Ch, pre_stride = 2*Ch; two 16-channel
 * Winograd convolutions: separate tensors, pre_stride = Ch).
 *   stage A: u = sigmoid(upre);  rec.state = h * sigmoid(rpre)
 *   stage B: h_out = h * (1 - u) + cand * u;  rec.state = h_out (rec may be NULL) */
int lf_gru_stage_a(const float* upre, const float* rpre, int pre_stride, const float* h, float* u, float* rec,
                   long nvox, int Ch, int rec_stride, int rec_off, void* stream);
int lf_gru_stage_b(const float* h, const float* u, const float* cand, float* h_out, float* rec, long nvox, int Ch,
                   int rec_stride, int rec_off, void* stream);
/* Backward of the two stages for the training path (autograd of modules/gru.py:37-43), plain arrays of n floats
 * (n % 4 == 0, 16-byte aligned):
 *   stage B: gh = g (1 - u), gu = g (cand - h), gc = g u
 *   stage A: gupre = gu u (1 - u), grpre = grh h r (1 - r), gh = grh r,  r = sigmoid(rpre) */
int lf_gru_stage_b_bwd(const float* g, const float* h, const float* u, const float* cand, float* gh, float* gu, float* gc,
                       long n, void* stream);
int lf_gru_stage_a_bwd(const float* gu, const float* grh, const float* u, const float* rpre, const float* h, float* gupre,
                       float* grpre, float* gh, long n, void* stream);
/* The ConvGRU recurrence of the TRAINING step sequenced explicitly (ops._GruFuse; autograd of recon/fusion.py:188-197 over
 * modules/gru.py:30-43): plain arrays of n values (n % 4 == 0, 16-byte aligned).  The state h, the gradient chain g / gh1 /
 * gh12 and the accumulators are fp32; rpre, upre, cand, rh and the gate gradients are fp32 (bf16 = 0) or bf16 (bf16 = 1).
 *   stage_a     rh = h sigmoid(rpre)
 *   stage_b     h_out = h (1 - u) + cand u,  u = sigmoid(upre)
 *   stage_b_bwd gh1 = g (1 - u); gupre = g (cand - h) u (1 - u); gc = g u; acc_u += gupre; acc_o += gc   (acc_* may be NULL)
 *   stage_a_bwd grpre = grh h r (1 - r); gh12 = gh1 + grh r, r = sigmoid(rpre); acc_r += grpre            (acc_r may be NULL) */
int lf_gru_train_stage_a(const void* rpre, const float* h, void* rh, long n, int bf16, void* stream);
int lf_gru_train_stage_b(const float* h, const void* upre, const void* cand, float* h_out, long n, int bf16, void* stream);
int lf_gru_train_stage_b_bwd(const float* g, const float* h, const void* upre, const void* cand, float* gh1, void* gupre,
                             void* gc, float* acc_u, float* acc_o, long n, int bf16, void* stream);
int lf_gru_train_stage_a_bwd(const void* grh, const void* rpre, const float* h, const float* gh1, void* grpre, float* gh12,
                             float* acc_r, long n, int bf16, void* stream);
/* dst[i] (+= if accumulate) sum over `views` bf16 arrays of n elements each (src = the arrays back to back), in order:
 * the sum over the recurrence's steps of a stored gate gradient (bias / coordinate-channel weight gradients). n % 4 == 0. */
int lf_sum_views_bf16(const void* src, float* dst, long n, int views, int accumulate, void* stream);

/* Storage-type variants of the 16-channel resampler for the training step (io bit 0: the source, bit 1: the destination is a
 * bf16 channels-last volume; same fp32 interpolation / same fixed-point sums as lf_resample3d_fwd / lf_resample3d_bwd_vol_det
 * with C = 16).  The splat runs in BINNED form: the output voxels are first listed under the source tiles their corners touch
 * (count / scan / fill, <= 8 entries per voxel), then one workgroup per tile adds exactly its lists into 64-bit LDS accumulators
 * -- same integers, bit-identical to the tile / atomic forms of lf_resample3d_bwd_vol_det (3.7 -> 2.5 ms at 8 x 128^3).  A volume
 * per sample (vol_n == N) is processed in passes of as many samples as 512 MB of lists hold; one shared volume (vol_n == 1)
 * needs all samples in one pass and otherwise takes the tile form.  scratch: lf_resample3d_bwd_vol_det_io_scratch_bytes. */
int lf_resample3d_fwd_io(const void* vol, int vol_n, const float* coef, int kind, void* out, int N, int D, int H, int W,
                         int io, void* stream);
size_t lf_resample3d_bwd_vol_det_io_scratch_bytes(int vol_n, int N, int D, int H, int W);
int lf_resample3d_bwd_vol_det_io(const void* gout, const float* coef, int kind, void* gvol, int vol_n, void* scratch,
                                 size_t scratch_bytes, int N, int D, int H, int W, int io, void* stream);

/* ------------------------------------------------------------------------------------------
 * Depth-column composites of the renderer (channels-last volumes [N][D][H][W][C], P = H*W pixels).
 * A depth column is strided by P*C floats: lanes run along the contiguous (pixel, channel) axis, the depth axis is
 * split over the waves of a workgroup (and over 16-lane groups combined with wave shuffles for the single-channel
 * softmax), partials meet in LDS in a fixed order -> deterministic, no atomics.
 *   lf_column_reduce_sum_fwd   y[n][p][c] = sum_d x[n][d][p][c]        Photographer 'sum' projection,
 *                                                                       recon/models.py:436-437
 *   lf_column_reduce_sum_bwd   gx[n][d][p][c] = gy[n][p][c]
 *   lf_column_softmax_fwd      w = softmax_d(logits[n][d][p]) ; zdepth[n][p] = sum_d w * linspace(-1,1,D)[d]
 *                              (_compute_depth_weights + _depth_from_weight, recon/models.py:378-395);
 *                              weights or zdepth may be NULL
 *   lf_column_softmax_bwd      glogits = w * (t - sum_d w t),  t = gweights + gzdepth * linspace(-1,1,D)
 *                              (either gradient may be NULL)
 *   lf_column_scale_fwd/bwd    out[row][:] = z[row][:] * w[row]  (z * depth_weights_resized, models.py:427-430);
 *                              bwd: gz = gout * w, gw[row] = sum_c gout * z  (gz or gw may be NULL); C % 4 == 0 */
/* The 17th channel of the OCCLUSION module for the render-loop engine (recon/models.py:378-395 over modules/unet.py's input
 * block, blocks.py:78-91, and the U-Net's first 3x3x3 convolution, blocks.py:152-158), 16 latent channels: the 17-channel tensors
 * cat(z, depth coordinate) / input-block output are never built.
 *   lf_occ_input_fwd   t = LeakyReLU(W1 [z; d] + b1) for the 17 outputs, d = linspace(-1, 1, D)[plane]: ta (outputs 0..15,
 *                      channels-last [rows][16]) and t16 (output 16, [rows]).
 *                      w: [17][20] = W1 * he (row = output; 16 latent inputs, depth input at [16]), b: [20]
 *   lf_occ_conv17_fwd  pre[v][co] = sum_taps w27[tap][co] * t16[v + tap - 1], zero padding; w27: [kz*9 + ky*3 + kx][16] =
 *                      W2[co][16][kz][ky][kx] * he.  The addend (LF_EPI_ADD) of the lf_conv3d_c16_wino launch over ta.
 *   lf_occ_conv17_bwd  gp16[u] = LeakyReLU'(t16[u]) * sum_taps sum_co w27[tap][co] * g[u - (tap - 1)][co]
 *                      (both on v_mfma_f32_16x16x4_f32: the forward as a 16 x 27 by 27 x voxels product, the backward as
 *                      h[tap][voxel] = W g per z plane into LDS followed by the shifted sums)
 *   lf_occ_input_bwd   gz = g_zs * wocc[row] + W1[:16]^T (gta * LeakyReLU'(ta)) + W1[16] * gp16   (ta == NULL: gta already carries
 *                      LeakyReLU'(ta) -- the data gradient that produced it applied it, prev_y = ta; g_zs / wocc: the direct term of
 *                      the occlusion scaling z * wocc, both NULL to leave it out); prev_y != NULL: followed by the epilogue
 *                      backward of the layer that produced z (saved output prev_y, norm prev_norm, prev_flags), as in
 *                      lf_conv3x3_bwd_data */
int lf_occ_input_fwd(const float* z, const float* w, const float* b, float* ta, float* t16, int N, int D, long P, float slope,
                     void* stream);
int lf_occ_input_bwd(const float* gta, const float* ta, const float* gp16, const float* w, const float* g_zs, const float* wocc,
                     float* gz, long rows, float slope, const float* prev_y, const float* prev_norm, unsigned prev_flags,
                     void* stream);
/* lf_occ_input_bwd with the factor projection's data gradient recomputed instead of read (round 6): the direct term of the scaling is
 * ((Wp_d^T gp2d[n][pixel]) * he_p) * wocc[voxel] with wpack_t = the transposed projection pack lf_conv1x1_bwd_data takes
 * ([D * 16 rows (d, c)][16 cout]) and gp2d the (N, H*W, 16) gradient of the projection's pre-activation; gta must already carry
 * LeakyReLU'(ta) (the data gradient that produced it ran with prev_y = ta).  P % 16 == 0, N*D*P < 2^31. */
int lf_occ_input_bwd_proj(const float* gta, const float* gp16, const float* w, const float* gp2d, const float* wpack_t, float he_p,
                          const float* wocc, float* gz, int N, int D, long P, float slope,
                          const float* prev_y, const float* prev_norm, unsigned prev_flags, void* stream);
/* Gradient of the occlusion weights through `z * w` -> factor projection: gw[n][d][p] = sum_c z[n][d][p][c] * g_zs[n][d][p][c],
 * g_zs = (Wp_d^T gp2d[n][p]) * he_p recomputed on the matrix pipe (wpack_t, gp2d as for lf_occ_input_bwd_proj); one read of z. */
int lf_occ_weight_grad(const float* z, const float* gp2d, const float* wpack_t, float he_p, float* gw, int N, int D, long P, void* stream);
/* lf_occ_weight_grad followed by the depth softmax's backward in one launch: glogits[d] = w[d] * (gw[d] - sum_e w[e] gw[e]) per pixel
 * column (weights = the softmax output lf_column_softmax(_head)_fwd stored); no gw volume.  D <= 256, P % 16 == 0. */
int lf_occ_weight_grad_softmax_bwd(const float* z, const float* gp2d, const float* wpack_t, float he_p, const float* weights,
                                   float* glogits, int N, int D, long P, void* stream);
/* Data gradient of the occlusion module's 16 -> 1 output block (no activation) onto the 16-channel activation y that fed it, with
 * the LeakyReLU' / PixelNorm' of the layer that produced y (its norm, flags) applied in the store:
 *   g[v][c] = epilogue'( (gl[v] * w16[c]) * he )          (rows voxels; what lf_conv1x1_bwd_data(Cin = 1, prev_y = y) computes) */
int lf_occ_head_bwd(const float* gl, const float* w16, float he, const float* y, const float* norm, unsigned flags, float slope,
                    float* g, long rows, void* stream);
int lf_occ_conv17_fwd(const float* t16, const float* w27, float* pre, int N, int D, int H, int W, void* stream);
int lf_occ_conv17_bwd(const float* g, const float* t16, const float* w27, float* gp16, int N, int D, int H, int W, float slope,
                      void* stream);
int lf_column_reduce_sum_fwd(const float* x, float* y, int N, int D, long P, int C, void* stream);
int lf_column_reduce_sum_bwd(const float* gy, float* gx, int N, int D, long P, int C, void* stream);
int lf_column_softmax_fwd(const float* logits, float* weights, float* zdepth, int N, int D, long P, void* stream);
/* The same with the logits formed on the way in: logit[n][d][p] = (sum_c y[n][d][p][c] * w16[c]) * he + bias[0] over a channels-last
 * 16-channel volume y (the occlusion module's 16 -> 1 output block without activation, reference recon/models.py:378-388 over
 * modules/unet.py / blocks.py OutputBlock); the logits volume is not written.  D <= 256; y and w16 16-byte aligned. */
int lf_column_softmax_head_fwd(const float* y, const float* w16, const float* bias, float he, float* weights, float* zdepth,
                               int N, int D, long P, void* stream);
int lf_column_softmax_bwd(const float* weights, const float* gweights, const float* gzdepth, float* glogits,
                          int N, int D, long P, void* stream);
int lf_column_scale_fwd(const float* z, const float* w, float* out, long rows, int C, void* stream);
int lf_column_scale_bwd(const float* gout, const float* z, const float* w, float* gz, float* gw, long rows, int C,
                        void* stream);

/* View reductions of the fusers: z holds V per-view volumes of n floats each, `view_stride` floats apart (any
 * element order -- the output uses the same one).
 *   lf_fuse_views_fwd  kind = LF_FUSE_MEAN | MAX | ABSMAX (signed value of largest magnitude) | MEDIAN (lower median,
 *                      torch.median; V <= 64): PoolFuser / pool_tensor, recon/fusion.py:45-57, functional.py:47-49.
 *                      idx (may be NULL; unused for MEAN) receives the selected view per element.
 *   lf_fuse_views_bwd  gz[v][i] = g[i] / V (MEAN) or g[i] * (v == idx[i]).
 *   lf_fuse_blend_fwd  BlendFuser.forward, recon/fusion.py:139-148: weights[v][row] = softmax_v(logits[v][row]),
 *                      out[row][:] = sum_v z[v][row][:] * weights[v][row]  (rows voxels, C % 4 == 0 channels-last;
 *                      logits / weights views `logit_view_stride` floats apart); weights may be NULL.
 *   lf_fuse_blend_bwd  gz[v] = g * w_v ; glogits_v = w_v (a_v - sum_u w_u a_u), a_v = sum_c g z_v (either may be NULL). */
#define LF_FUSE_MEAN   0
#define LF_FUSE_MAX    1
#define LF_FUSE_ABSMAX 2
#define LF_FUSE_MEDIAN 3
int lf_fuse_views_fwd(const float* z, float* out, int* idx, int kind, int V, long n, long view_stride, void* stream);
int lf_fuse_views_bwd(const float* g, const int* idx, float* gz, int kind, int V, long n, long view_stride, void* stream);
int lf_fuse_blend_fwd(const float* z, const float* logits, float* weights, float* out, int V, long rows, int C,
                      long z_view_stride, long logit_view_stride, void* stream);
int lf_fuse_blend_bwd(const float* g, const float* z, const float* weights, float* gz, float* glogits, int V, long rows,
                      int C, long z_view_stride, long logit_view_stride, void* stream);

/* Cell arithmetic of the convolutional LSTM fuser (modules/lstm.py:41-56): cc = conv([x, h]) holds the gate
 * pre-activations of a voxel as channel blocks [i | f | o | g] (channels-last record of 4*Ch floats);
 *   c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c').
 * bwd: given gh = dL/dh' and gcn = dL/dc' (either may be NULL) -> gcc [nvox][4*Ch] and gc = dL/dc. */
int lf_lstm_cell_fwd(const float* cc, const float* c_cur, float* h_next, float* c_next, long nvox, int Ch, void* stream);
int lf_lstm_cell_bwd(const float* cc, const float* c_cur, const float* gh, const float* gcn, float* gcc, float* gc,
                     long nvox, int Ch, void* stream);

/* 2-D grid sampling of planar images, F.grid_sample(align_corners=False) semantics: the crop / zoom
 * (geometry.py:20-44,287-354; zeros padding), Camera.uncrop (geometry.py:261-285; border padding) and the
 * image-based-rendering warps (ibr.py:52-93).  img [N][C][H][W], grid [N][Ho][Wo][2] = (x, y) in [-1,1],
 * out / gout [N][C][Ho][Wo]; bilinear = 1 | 0 (nearest, round-half-even); border = 1 (replicate) | 0 (zeros).
 * bwd: gimg (zero-initialised by the caller, accumulated with float atomics) and/or ggrid may be NULL. */
int lf_grid_sample2d_fwd(const float* img, const float* grid, float* out, int N, int C, int H, int W, int Ho, int Wo,
                         int bilinear, int border, void* stream);
int lf_grid_sample2d_bwd(const float* img, const float* grid, const float* gout, float* gimg, float* ggrid,
                         int N, int C, int H, int W, int Ho, int Wo, int bilinear, int border, void* stream);

/* Weight gradient of the He-equalised conv (training step; autograd of equalized.py:57-64):
 *   gw[tap][co][ci] = scale * sum_v gpre[v][co] * x[v + tap][ci],   tap = (kz*3 + ky)*3 + kx, zero padding,
 * dims = 3 (27 taps), 2 (9 taps, D = 1) or 0 (pointwise: one tap, rows = N*D*H*W).  x, gpre channels-last
 * [N][D][H][W][C]; gpre is the gradient w.r.t. the pre-activation (after lf_epilogue_bwd); scale = he.
 * x == NULL: all-ones single-channel input, i.e. gw[0][co][0] = sum_v gpre[v][co] (bias gradient, scale 1).
 * Deterministic: per-block partials in `scratch` (lf_conv_bwd_weight_scratch_bytes), fixed-order fp64 sum. */
size_t lf_conv_bwd_weight_scratch_bytes(int dims, int N, int D, int H, int W, int Cin, int Cout);
int lf_conv_bwd_weight(const float* x, const float* gpre, float* gw, void* scratch, size_t scratch_bytes,
                       int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, void* stream);
/* The same weight gradient on the bf16 MFMA (autocast policy of the training step; recon/models.py:199,405): x and gpre
 * are rounded to bf16 (RNE) as they are staged -- the identity on operands the policy has already rounded -- products
 * are exact, accumulation fp32, partials summed in a fixed order in fp64 like lf_conv_bwd_weight (same scratch size).
 * 3-D 16 -> 16 layers with N*D*H*W >= 8192, D*H*W*64 < 2^31 and (D+3)*H*W*64 < 2^32 only: LF_EINVAL otherwise (the caller
 * keeps lf_conv_bwd_weight for the rest). */
int lf_conv_bwd_weight_bf16(const float* x, const float* gpre, float* gw, void* scratch, size_t scratch_bytes,
                            int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, void* stream);
/* ... with x (io & 1) and / or gpre (io & 2) stored as bf16 channels-last records (see lf_conv3d_c16_ring_bf16_io). */
int lf_conv_bwd_weight_bf16_io(const void* x, const void* gpre, float* gw, void* scratch, size_t scratch_bytes,
                               int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, int io, void* stream);

/* Epilogue backward of a 16-CHANNEL layer for the training step: gp = LeakyReLU'(y) PixelNorm'(gy; y, norm) (flags says
 * which; flags = 0: gp = gy) with the bias gradient gbias[16] = column sums of the un-rounded gp folded into the same pass
 * (gbias = NULL: no sums; else scratch of lf_epilogue_bwd_c16_scratch_bytes(rows) bytes), fixed-order reduction.
 * io: bit 0 -- gy, bit 1 -- y, bit 2 -- gp are bf16 [rows][16] arrays instead of fp32 (see lf_conv3d_c16_ring_bf16_io).
 * Autograd of modules/blocks.py:152-158 + equalized.py:57-64. */
size_t lf_epilogue_bwd_c16_scratch_bytes(long rows);
int lf_epilogue_bwd_c16(const void* gy, const void* y, const float* norm, void* gp, float* gbias, void* scratch,
                        size_t scratch_bytes, long rows, unsigned flags, float slope, int io, void* stream);

/* The 2-D -> 3-D lift of the training step for 16 image channels and C0 = 16 volume channels as ONE kernel each way (round 6;
 * reference modules/geometry.py:711-731 FactorProjection2d3d + autograd, bf16 autocast policy): rows x [R = V*P][16] fp32
 * (bf16 values) -> pointwise conv to 16*S channels (channel = c*S + d), He scale, bias, LeakyReLU, PixelNorm over all 16*S
 * channels -> bf16 channels-last volume records (v, d, p) of 16 channels + norm[R]; the 16*S-channel row is never stored.
 * wtab: bf16 [S][16 c][16 cin] = W[c*S + d][cin]; btab: fp32 [S][16 c] or NULL.  R % 16 == 0, P % 16 == 0, R % P == 0.
 * Backward: gvol / yvol = gradient and saved volume (bf16 records), wtab_t: bf16 [S][16 cin][16 c]; gx [R][16] fp32 (rounded to
 * bf16 values if round_gx), gw [16*S][16] (scaled by he), gb [16*S]; S in {16, 32, 64, 128}; scratch of
 * lf_lift16_bwd_scratch_bytes(S) bytes (per-workgroup partials, summed in a fixed order: deterministic). */
int lf_lift16_fwd(const float* x, const void* wtab, const float* btab, void* vol, float* norm, long R, long P, int S,
                  float he, float slope, float eps, void* stream);
size_t lf_lift16_bwd_scratch_bytes(int S);
int lf_lift16_bwd(const void* gvol, const void* yvol, const float* norm, const float* x, const void* wtab_t, float* gx, float* gw,
                  float* gb, void* scratch, size_t scratch_bytes, long R, long P, int S, float he, float slope, int round_gx,
                  void* stream);

/* The 3-D -> 2-D factor projection of the training step's renderer for 16 volume channels and 16 output channels on the same
 * idea (reference modules/geometry.py:731-749 FactorProjection3d2d + autograd): vol = bf16 records (v, d, p); y fp32 rows
 * [R][16] = PixelNorm(LeakyReLU(he * sum_{c,d} W[o][c*S + d] vol + b)), norm [R].  wtab: bf16 [S][16 o][16 c].
 * Backward from gp (fp32 rows [R][16], rounded to bf16 inside): gxvol = bf16 records of he * W^T gp; gw [16][16*S] (scaled by
 * he); wtab_t: bf16 [S][16 c][16 o]; S in {16, 32, 64, 128}; deterministic (per-workgroup partials, fixed order). */
int lf_proj16_fwd(const void* vol, const void* wtab, const float* bias, float* y, float* norm, long R, long P, int S, float he,
                  float slope, float eps, void* stream);
size_t lf_proj16_bwd_scratch_bytes(int S);
int lf_proj16_bwd(const float* gp, const void* vol, const void* wtab_t, void* gxvol, float* gw, void* scratch, size_t scratch_bytes,
                  long R, long P, int S, float he, void* stream);

/* ---- pointwise 16 -> 16 layers of the training step (round 6; csrc/pw16.hip) ------------------------------------------------------------
 * 1x1x1 convolutions under the bf16 autocast + storage policy (the sculptor's output block, reference recon/models.py:143,222 over
 * modules/blocks.py:108-119): one v_mfma_f32_16x16x16_bf16 per 16 voxels instead of the 3x3x3 ring kernels with the weights on the
 * centre tap.  rows = N*D*H*W voxels, channels-last 16-channel records.
 *   lf_pw16_fwd   y = act( bf16(bf16(W x) * he) + bias ) stored as bf16; x in bf16 (x_bf16 = 1) or fp32 storage (rounded on load);
 *                 w_bf16 = bf16 [cout][cin]; flags: 0 or LF_EPI_LRELU
 *   lf_pw16_bwd   gx = bf16(bf16(W^T gy) * he) stored as bf16 (gx_bf16 = 1) or fp32; wt_bf16 = bf16 [cin][cout]; gy in bf16 storage;
 *                 gbias != NULL: also gbias[c] = sum over the voxels of gy[.][c] (per-workgroup partial sums in `scratch`, reduced in
 *                 a fixed order: deterministic) */
int lf_pw16_fwd(const void* x, int x_bf16, const void* w_bf16, const float* bias, float he, unsigned flags, float slope, void* y,
                long rows, void* stream);
size_t lf_pw16_bwd_scratch_bytes(long rows);
int lf_pw16_bwd(const void* gy_bf16, const void* wt_bf16, float he, void* gx, int gx_bf16, float* gbias, void* scratch,
                size_t scratch_bytes, long rows, void* stream);

/* Standalone PixelNorm over the last (channel) axis of [rows][C], in place allowed.
 * norm_out[rows] receives sqrt(mean+eps).  modules/__init__.py:14-15. */
int lf_pixelnorm_fwd(const float* x, float* y, float* norm_out, long rows, int C, float eps, void* stream);

/* Backward of the fused epilogue: given gy = dL/dy, the saved output y and norm, produce
 * dL/d(conv*he + bias) (the caller then runs the transposed convolution and applies he):
 *   flags = LRELU|PIXELNORM: gp = lrelu'(y) * (gy - y * mean_c(gy*y)) / norm
 *   flags = LRELU           : gp = lrelu'(y) * gy
 *   flags = PIXELNORM       : gp = (gy - y*mean_c(gy*y)) / norm
 * rows x C, channels-last; in place (gp == gy) allowed. */
int lf_epilogue_bwd(const float* gy, const float* y, const float* norm, float* gp,
                    long rows, int C, unsigned flags, float slope, void* stream);

/* Block-end rescale by exactly x2 (up = 1) or x0.5 (up = 0): nearest (linear = 0) or bi-/tri-linear
 * with align_corners=False (linear = 1) = Interpolate / F.interpolate(scale_factor=...) of
 * modules/__init__.py:18-36, blocks.py:160-162.  x: [N][D][H][W][C] (D = 1 for dims = 2); the output
 * extent is 2*in or floor(in/2) per spatial axis.  lf_resize_bwd is the exact adjoint (gather form). */
int lf_resize_fwd(const float* x, float* y, int dims, int N, int D, int H, int W, int C, int linear, int up,
                  void* stream);
int lf_resize_bwd(const float* gy, float* gx, int dims, int N, int D, int H, int W, int C, int linear, int up,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Camera -> coefficient blocks, with Jacobian (fp64 forward-mode duals inside).
 * Replaces the chain of tiny ops log_quaternion -> qexp -> normalize -> quat_to_mat -> cam_to_obj ->
 * camera_coords (three/quaternion.py:39-93,287-311; modules/geometry.py:106-153,207-213,469-531)
 * plus the viewport / depth-range algebra of Camera.uncrop and denormalize_depth (:261-285,555-558).
 *   params     [N][10] = (log_quaternion 3, translation 3, viewport xmin,ymin,xmax,ymax)
 *   intrinsics [N][4]  = (fu, fv, u0, v0)
 *   coefs      [N][24] : [0..17] LF_MAP_O2C block; [18..21] (ax,bx,ay,by): crop sample position of
 *                         frame pixel (x,y) is (ax*x+bx, ay*y+by); [22..23] (a,b): depth = d*a + b
 *   jac        [N][24][10] = d coefs / d params
 * lf_camera_coefs_bwd: gparams[N][10] = sum_j gcoefs[N][j] * jac[N][j][:]. */
#define LF_CAM_PARAMS 10
#define LF_CAM_COEFS  24
int lf_camera_coefs(const float* params, const float* intrinsics, float cube_size, float z_span,
                    int crop_h, int crop_w, float* coefs, float* jac, int N, void* stream);
int lf_camera_coefs_bwd(const float* gcoefs, const float* jac, float* gparams, int N, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused pose loss = default_pose_loss (pose/estimation.py:70-118, pose/utils.py:81-117) over
 * Photographer.interpret_logits(apply_mask=True) (recon/models.py:455-484), Camera.uncrop
 * (nearest for depth, bilinear for mask logits, border padding) and denormalize_depth.
 *   logits  [N][h*w][2]  head outputs (depth_logit, mask_logit), channels-last
 *   coefs   [N][24]      from lf_camera_coefs (entries 18..23 are used)
 *   target_depth/mask [H*W] of the (single) target frame; weights[4] = (depth, ov_depth, iou, mask)
 *   sums    [N][8]  S0..S6 (sum l1, sum l1*sig*mt, sum sig*mt, sum sig, intersection, sum mt*valid, sum bce)
 *   losses  [N][8]  (depth, ov_depth, iou, mask, weighted total, 0,0,0)
 *   gsums   [N][8]  d(mean_n total)/d(sums), consumed by lf_pose_loss_bwd
 * lf_pose_loss_bwd writes glogits [N][h*w][2] and gcoefs[N][24] entries 18..23 (0..17 untouched).
 * All reductions are fixed-order (no float atomics).  scratch: lf_pose_loss_scratch_bytes. */
size_t lf_pose_loss_scratch_bytes(int N, int h, int w, int H, int W);
int lf_pose_loss_fwd(const float* logits, const float* coefs, const float* target_depth,
                     const float* target_mask, const float* weights, float* sums, float* losses,
                     float* gsums, void* scratch, size_t scratch_bytes,
                     int N, int h, int w, int H, int W, void* stream);
int lf_pose_loss_bwd(const float* logits, const float* coefs, const float* target_depth,
                     const float* target_mask, const float* gsums, float* glogits, float* gcoefs,
                     void* scratch, size_t scratch_bytes, int N, int h, int w, int H, int W, void* stream);
/* The loss as the cross-entropy and Metropolis estimators evaluate it (PoseEstimator._render_observation,
 * pose/estimation.py:207-216): the de-normalised crop depth is multiplied by the crop's own sigmoid mask before it is
 * uncropped, z_depth = denormalize(depth) * mask; everything else as lf_pose_loss_fwd.  Forward only (those estimators
 * rank hypotheses, they do not differentiate); same scratch size. */
int lf_pose_loss_fwd_masked(const float* logits, const float* coefs, const float* target_depth,
                            const float* target_mask, const float* weights, float* sums, float* losses,
                            void* scratch, size_t scratch_bytes, int N, int h, int w, int H, int W, void* stream);
/* The loss as the MODULE path evaluates it (default_pose_loss handed the predicted metric depth crop and the mask-logit
 * crop, pose/estimation.py:70-118 called from :703-713 with the caller's own de-normalisation): channel 0 of
 * depth_and_logits [N][h*w][2] is the depth itself (no tanh / apply_mask / denormalize_depth inside), channel 1 the mask
 * logit; coefs entries 18..21 (the uncrop map) are used.  lf_pose_loss_bwd_depth writes d/d(depth crop, logit crop) and
 * gcoefs entries 18..21 (22, 23 = 0).  Same sums / losses / gsums / scratch as lf_pose_loss_fwd / _bwd. */
int lf_pose_loss_fwd_depth(const float* depth_and_logits, const float* coefs, const float* target_depth,
                           const float* target_mask, const float* weights, float* sums, float* losses,
                           float* gsums, void* scratch, size_t scratch_bytes,
                           int N, int h, int w, int H, int W, void* stream);
int lf_pose_loss_bwd_depth(const float* depth_and_logits, const float* coefs, const float* target_depth,
                           const float* target_mask, const float* gsums, float* gcrop, float* gcoefs,
                           void* scratch, size_t scratch_bytes, int N, int h, int w, int H, int W, void* stream);

/* Batched Adam / AdamW step over N independent rows of P parameters (pose/estimation.py:579-594,
 * 664-666).  step_size[n] = lr[n] / (1 - beta1^t) and bias_correction2_sqrt = sqrt(1 - beta2^t)
 * are formed on the host in double like torch.optim does. */
int lf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                 const float* step_size, const float* lr, float bias_correction2_sqrt,
                 float beta1, float beta2, float eps, float weight_decay, int N, int P, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout helpers (the reference is NCDHW; kernels are channels-last). */
int lf_nchw_to_nhwc(const float* src, float* dst, int N, int C, long P, void* stream);
int lf_nhwc_to_nchw(const float* src, float* dst, int N, int C, long P, void* stream);
/* FactorProjection2d3d.view (geometry.py:728): src [N][P][C0*S] with channel = c*S + d
 * -> dst [N][S][P][C0], optionally scaling row (n,p) by 1/norm[n*P+p] (deferred PixelNorm). */
int lf_lift_unfold(const float* src, const float* norm_or_null, float* dst,
                   int N, long P, int C0, int S, void* stream);
/* The same permutation for the training step, either way (autograd of the .view above and of
 * FactorProjection3d2d's reshape, geometry.py:745): fold = 0: src [N][P][C0*S] -> dst [N][S][P][C0];
 * fold = 1: src [N][S][P][C0] -> dst [N][P][C0*S].  4 * C0 * (S + 1) * 4 bytes of LDS must fit 64 KB (LF_EINVAL). */
int lf_lift_permute(const float* src, float* dst, int N, long P, int C0, int S, int fold, void* stream);
/* FactorProjection2d3d of the TRAINING step in two passes instead of four (modules/geometry.py:711-728 and its autograd):
 * lf_lift_norm_unfold: src [N][P][C0*S] = LeakyReLU(conv * he + b) rows (channel = c*S + d) -> PixelNorm over the C0*S channels
 *   of every pixel, written as the (N,C0,S,H,W) channels-last volume dst [N][S][P][C0] (fp32, or bf16 with out_bf16) and
 *   norm_out [N*P] = sqrt(mean + eps);
 * lf_lift_bwd: gradient volume + saved volume (io bit 0 / bit 1: stored as bf16) + norm -> gp [N][P][C0*S], the gradient
 *   w.r.t. the convolution's pre-activation (PixelNorm' then LeakyReLU'), as rows for the weight / data gradient products;
 *   round_bf16: written as bf16 values in the fp32 container (the autocast policy's operand rounding).
 * C0 % 4 == 0, S % 4 == 0, 2 * 4 * C0 * (S + 1) floats of LDS (<= 150 KB). */
int lf_lift_norm_unfold(const float* src, void* dst, float* norm_out, int N, long P, int C0, int S, float eps, int out_bf16,
                        void* stream);
int lf_lift_bwd(const void* gvol, const void* yvol, const float* norm, float* gp, int N, long P, int C0, int S, float slope,
                int round_bf16, int io, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LF_HIP_H */
