/*
 * saragan_hip.h -- C ABI of libsaragan_hip.so: the MI355X (gfx950) kernels behind the SURFGAN_3D
 * `pgan` generator + discriminator training step.
 *
 * The reference (sara-nl/saraGAN) has no FFI on this path: its boundary is Python calling TensorFlow-1
 * ops (SURFGAN_3D/networks/ops.py).  Each entry point below replaces the TF op(s) named in its
 * comment; INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C: pointers, sizes and small POD structs only; no torch / HIP types in signatures
 *    (`sg_stream_t` is a hipStream_t passed as void*; NULL = the default stream);
 *  - every buffer is caller-owned DEVICE memory; no entry point allocates, frees or synchronises;
 *    kernels are enqueued on the stream given; entry points are re-entrant.  Process-wide state is limited to the
 *    opt-in profiler (sg_prof_*), an immutable snapshot of the SG_* diagnostic environment switches taken at
 *    the first launch (sg_config_reload) and the one-time registration of each kernel's LDS size;
 *  - activations are NDHWC (channels-last 3-D): element (n,d,h,w,c) at (((n*D+d)*H+h)*W+w)*C+c.
 *    A 2-D image batch is D == 1.  dtype SG_F32 or SG_BF16 (storage AND MFMA input type; accumulation
 *    is always f32);
 *  - weights stay in the reference's layouts: conv DHWIO [kD][kH][kW][Cin][Cout] f32
 *    (networks/ops.py:148), dense [in][out] f32 (networks/ops.py:142); the equalised-LR runtime
 *    coefficient (networks/ops.py:111-122) is an argument, applied when packing;
 *  - return value: 0 on success, a negative SG_E* code for bad arguments, or a positive hipError_t.
 */
#ifndef SARAGAN_HIP_H
#define SARAGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sg_stream_t;

typedef enum { SG_F32 = 0, SG_BF16 = 1 } sg_dtype;

enum {
  SG_OK = 0,
  SG_EINVAL = -1,      /* bad shape / null pointer / unsupported combination */
  SG_EWORKSPACE = -2,  /* workspace too small */
  SG_EALIGN = -3,      /* pointer not 16-byte aligned */
  SG_EUNSUPPORTED = -4 /* valid request that no kernel of this build covers (sub-pixel epilogue on a small layer): the
                          caller falls back to the plain formulation */
};

/* Geometry of one stride-1 'SAME' convolution (tf.nn.conv3d at networks/ops.py:150).
 * (d,h,w) is the OUTPUT extent.  If `upsample_in` != 0 the input tensor is half resolution
 * ((d/2,h/2,w/2), all even) and is read through a nearest-neighbour x2 gather, i.e. the kernel
 * computes conv3d(upscale3d(x)) (networks/ops.py:276-289 + :147-150) without materialising it. */
typedef struct {
  int32_t n, d, h, w;
  int32_t cin, cout;
  int32_t kd, kh, kw; /* odd */
  int32_t upsample_in;
} sg_conv_shape;

/* Epilogue fused into sg_conv3d_fwd (apply_bias + act + pixel_norm: networks/ops.py:130-136,167-192,308-310).
 * struct_size MUST be sizeof(sg_conv_epilogue) of the header the caller was built against: the library rejects any
 * other value with SG_EINVAL instead of reading past a shorter (older) struct. */
typedef struct {
  uint32_t struct_size;
  const float* bias;   /* [cout] or NULL */
  int32_t act;         /* 0 = linear, 1 = leaky_relu */
  float slope;         /* leaky_relu negative slope */
  int32_t pixel_norm;  /* 1: y *= rsqrt(mean_c(y^2)+eps); requires cout <= 128 */
  float eps;
  float* pn_scale;     /* optional [n*d*h*w] f32: the per-voxel rsqrt factor (saved for backward) */
  const void* mask_bits; /* optional sign words (layout below) of a tensor shaped like y: y *= (bit ? mask_slope : 1),
                            applied last.  Fuses the LeakyReLU backward of the layer whose output gradient this
                            data-gradient convolution produces (networks/ops.py:175-178: dx = where(y >= 0, dy,
                            dy*alpha)) into the convolution. */
  float mask_slope;
  void* sign_out;        /* optional: receives the sign words of y (after bias + act) for a later mask_bits */
  /* Sub-pixel form of conv3d(upscale3d(x)) (networks/ops.py:276-289 + :147-150): the output voxels of one parity class
   * (2i+a, 2j+b, 2k+c) are a 2x2x2-tap convolution of the LOW-resolution input with summed weights (27 -> 8 taps,
   * 3.4x fewer FLOPs).  A launch with kd = kh = kw = 2 computes one class: out_scale = 2 makes y the full-resolution
   * tensor [n,2d,2h,2w,cout] (pn_scale / sign words likewise) written at the voxels of class out_off; tap t of a
   * dimension reads input offset t - 1 + tap_off (tap_off = parity: taps {-1,0} for even, {0,+1} for odd outputs).
   * All zero: the ordinary convolution. */
  int32_t out_scale;
  int32_t out_off[3];
  int32_t tap_off[3];
  /* pool = 1: y is NOT the full-resolution output but its mean over 2 (D) x 1 (H) x 2 (W) blocks, [n, d/2, h, w/2, cout]
   * -- the first stage of downscale3d(act(conv3d(x) + b)) (pgan/discriminator.py:39-44), finished by
   * sg_downscale_sum(1,2,1, gain 1/2).  sign_out still receives the FULL-resolution sign words (all the backward
   * needs).  Only the sliding-halo kernel implements it (bf16, 3x3x3, cin <= 32, w % 32 == 0, cout % 32 == 0, even
   * d and h, no pixel-norm; mask_bits only without bias / activation / sign_out and with 32 input channels: the block means
   * of M * conv(x), the double backward of such a layer): anything else returns SG_EUNSUPPORTED and the caller runs conv +
   * downscale.
   * pool = 2: the mean over 1 (D) x 2 (H) x 2 (W) blocks instead, [n, d, h/2, w/2, cout], finished by
   * sg_downscale_sum(2,1,1, gain 1/2): the streamed ping-pong kernel's tile (bf16, 3x3x3, cin % 16 == 0, cout % 64 == 0,
   * even h and w, no pixel-norm / mask), for the layers with more than 32 input channels.  With act = 0 and no bias
   * either mode is the block mean of a plain convolution (the gradient of conv3d(upscale3d(x)), 8 x the mean).
   * pool = 3: the mean over whole 2 x 2 x 2 blocks, [n, d/2, h/2, w/2, cout] -- downscale3d finished inside the epilogue, no
   * second pass (32 input channels, even d / h / w, otherwise as pool = 1; SG_EUNSUPPORTED elsewhere: use pool = 1). */
  int32_t pool;
  /* Optional scratch for layers the library computes in two passes over halves of the input channels (64 -> 32 at
   * 3x3x3, bf16: f32 partial sums of the first half, n*d*h*w*cout*4 bytes, 16-byte aligned, contents irrelevant on
   * entry).  NULL: such layers run as one pass of the streamed kernel.  sg_conv3d_fwd_workspace() gives the size. */
  void* workspace;
  size_t workspace_bytes;
  /* 0: x is one NDHWC tensor of cin channels.  32: x is cin / 32 separate NDHWC tensors of 32 channels each, one after
   * the other (what sg_upscale_nn_planes writes).  The two passes of a 64 -> 32 layer (see workspace) then read whole
   * 64-byte rows instead of one half of every 128-byte row -- a half row costs the whole line from HBM.  Only that
   * path reads the layout: anything else returns SG_EUNSUPPORTED. */
  int32_t x_plane_channels;
  /* Pixel-norm backward in the epilogue of a DATA-GRADIENT convolution: the convolution's result is the gradient g for the
   * output y = pixel_norm(leaky_relu(z + b)) of a generator stage (pgan/generator.py:33-45); with pn_bwd_y = y [n,d,h,w,cout]
   * and pn_bwd_scale = the stage's rsqrt factor [n*d*h*w] the epilogue turns it into the gradient for z,
   * mask_bits(y's sign words) * pn_bwd_scale * (g - y * mean_c(g * y)), before it is written (what sg_pixel_norm_act_bwd
   * computes in a pass of its own).  Needs mask_bits, cout == 32, the sliding-halo kernel's shapes (bf16, 3x3x3,
   * cin <= 32, w % 32 == 0), no bias / activation / pooling; otherwise SG_EUNSUPPORTED.  Both NULL: off. */
  const void* pn_bwd_y;
  const float* pn_bwd_scale;
  /* Masked nearest up-scale of the INPUT in the gather (with s->upsample_in = 1): the convolution reads
   * in_gain * where(bit, in_mask_slope, 1) * upscale3d(x) instead of upscale3d(x); in_mask_bits are the sign words of a
   * tensor of the FINE shape [n,d,h,w,cin] (ceil(cin/32) words per voxel).  This is the gradient of
   * downscale3d(leaky_relu(.)) (pgan/discriminator.py:39-44 backward: networks/ops.py:265-273 then :175-178, gain 1/8)
   * formed while the halo is staged, so the full-resolution gradient -- 8 x the bytes of x -- is never written.  Values are
   * rounded to the storage type exactly as sg_upscale2x_masked rounds them; in_gain must be a power of two (1/8 for downscale3d).  Implemented by the two-pass 64 -> 32 path
   * (see workspace; bf16, 3x3x3, even d/h/w, w % 32 == 0): anything else returns SG_EUNSUPPORTED.  NULL: off. */
  const void* in_mask_bits;
  float in_mask_slope;
  float in_gain;
  /* to_rgb (networks/ops.py:239-240, pgan/generator.py:96-97: a 1x1x1 convolution to ONE image channel) of this convolution's
   * output inside its epilogue: rgb_out[n,d,h,w,1] = sum_c y[..,c] * rgb_w[c] + rgb_bias[0] with y as it is stored (rounded to dt)
   * and rgb_w the [cout] f32 values to_rgb's forward multiplies with -- the full-resolution stage output is not read again for the
   * image.  With pixel_norm + act + sign_out, cout == 32, the 32-input-channel 3x3x3 kernel's shapes (bf16, w % 32 == 0);
   * otherwise SG_EUNSUPPORTED (run sg_conv3d_fwd on y).  rgb_out NULL: off. */
  const float* rgb_w;
  const float* rgb_bias;
  void* rgb_out;
  /* from_rgb's WHOLE backward (networks/ops.py:243-247, pgan/discriminator.py:9-12: 1x1x1 from ONE image channel, bias, LeakyReLU)
   * in the epilogue of the data-gradient convolution that produces the gradient g of its output (mask_bits = from_rgb's sign
   * words): pw_dx[n,d,h,w,1] = sum_c g_c * pw_wmat[c] (optional), pw_dw[c] = pw_coef * sum_v pw_x[v] * g[v][c], pw_dbias[c] =
   * sum_v g[v][c] (each optional), g as it would have been stored (rounded to dt) -- and y is NOT written (y may be NULL): the
   * 32-channel gradient, the largest tensor of the discriminator's backward, is needed by nobody else.  workspace must hold
   * sg_conv3d_pw_epilogue_workspace() bytes (per-wave partial sums, added in a fixed order).  Same kernel and shapes as rgb_*
   * (cout == 32, no bias / activation / pooling); otherwise SG_EUNSUPPORTED (run the convolution, then sg_conv3d_pw_bwd).
   * pw_x NULL: off. */
  const void* pw_x;
  const float* pw_wmat;
  void* pw_dx;
  float* pw_dw;
  float* pw_dbias;
  float pw_coef;
} sg_conv_epilogue;

/* Sign words of an NDHWC tensor t[nvox][c]: uint32 words[nvox][ceil(c/32)], bit j of word (v, k) = (t[v][32k+j] < 0),
 * bits of channels >= c are 0.  One bit per element instead of re-reading the bf16/f32 activation in backward. */
size_t sg_sign_words_bytes(int64_t nvox, int32_t c);
int sg_sign_words(const void* t, void* words, int64_t nvox, int32_t c, sg_dtype dt, sg_stream_t st);

const char* sg_version(void);
const char* sg_error_string(int code);

/* ---- conv3d / dense (networks/ops.py:139-150) -------------------------------------------------- */
/* Bytes of the MFMA-fragment-ordered weight image sg_conv3d_pack_weights writes. */
size_t sg_conv3d_packed_bytes(const sg_conv_shape* s, sg_dtype dt);
/* wp <- coef * w, cast to dt, in fragment order.  transpose_flip != 0 packs the weights of the
 * data-gradient convolution instead: w is then [kD][kH][kW][s->cout][s->cin] and is mirrored in
 * (kD,kH,kW) and transposed in (I,O) (what tf's Conv3DBackpropInputV2 computes for stride 1). */
int sg_conv3d_pack_weights(const float* w_dhwio, float coef, int transpose_flip, void* wp,
                           const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
/* The images of n layers at once -- what n calls of sg_conv3d_pack_weights write, the fragment images in one launch per 56
 * layers.  After an optimiser step (optimization.py:16-73: every variable of the network moves) all of a network's images are
 * stale together; w / coef / transpose_flip / wp / shapes are host arrays of n entries. */
int sg_conv3d_pack_weights_batch(int n, const float* const* w_dhwio, const float* coef, const int* transpose_flip,
                                 void* const* wp, const sg_conv_shape* shapes, sg_dtype dt, sg_stream_t st);
/* Bytes of sg_conv_epilogue.workspace that let this shape take its fastest path (0: none needed). */
size_t sg_conv3d_fwd_workspace(const sg_conv_shape* s, sg_dtype dt);
size_t sg_conv3d_pw_epilogue_workspace(void);     /* bytes of sg_conv_epilogue.workspace for the pw_* epilogue */
/* y = epilogue(conv3d(x, wp)).  x: [n,d,h,w,cin] (or half-res if upsample_in), y: [n,d,h,w,cout]. */
int sg_conv3d_fwd(const void* x, const void* wp, void* y, const sg_conv_shape* s,
                  const sg_conv_epilogue* ep, sg_dtype dt, sg_stream_t st);
/* conv3d(upscale3d(x)) (networks/ops.py:276-289 followed by :147-150; pgan/generator.py:49-57) in sub-pixel form, ONE
 * launch for all eight parity classes: 64 instead of 216 tap products per low-resolution voxel.  `s` is the LOW-resolution
 * shape (n, d, h, w, cin, cout; kd = kh = kw = 3 of the original convolution, upsample_in ignored); x: [n,d,h,w,cin],
 * y: [n,2d,2h,2w,cout].  wp comes from sg_upconv3d_subpixel_pack (the summed weights of the 8 classes, coef applied, in
 * fragment order; sg_upconv3d_subpixel_packed_bytes bytes).  Epilogue: bias, act / slope, pixel_norm (cout 32 or 64) with
 * pn_scale, sign_out; anything else, f32, or a shape that does not tile (256 low-resolution voxels per tile; cin % 16,
 * cout % 32): SG_EUNSUPPORTED -- run sg_conv3d_fwd with upsample_in = 1. */
int sg_upconv3d_subpixel_supported(const sg_conv_shape* s, sg_dtype dt);   /* 1 if the shape tiles (no launch, no GPU needed) */
size_t sg_upconv3d_subpixel_packed_bytes(const sg_conv_shape* s, sg_dtype dt);
int sg_upconv3d_subpixel_pack(const float* w_dhwio, float coef, void* wp, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
int sg_upconv3d_subpixel_fwd(const void* x, const void* wp, void* y, const sg_conv_shape* s, const sg_conv_epilogue* ep,
                             sg_dtype dt, sg_stream_t st);
/* Weight (and bias) gradient of conv3d(upscale3d(x)) in the same sub-pixel form: 64 accumulator tiles per (cin, cout) tile
 * pair instead of 8 x 27 tap products per low-resolution voxel, folded to the 27-tap gradient at the end.  `s`: the
 * LOW-resolution shape; x: [n,d,h,w,cin], gy: [n,2d,2h,2w,cout]; dw: [3][3][3][cin][cout] f32 = coef * gradient, dbias: [cout]
 * or NULL.  bf16, w % 32 == 0, even h, cin and cout multiples of 32; otherwise SG_EUNSUPPORTED (run sg_conv3d_wgrad_bias
 * with upsample_in = 1). */
int sg_upconv3d_subpixel_wgrad_supported(const sg_conv_shape* s, sg_dtype dt);
size_t sg_upconv3d_subpixel_wgrad_workspace(const sg_conv_shape* s, sg_dtype dt);
int sg_upconv3d_subpixel_wgrad(const void* x, const void* gy, float* dw_dhwio, float* dbias, float coef, void* workspace,
                               size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
/* Data gradient of conv3d(upscale3d(x)) (what tf.gradients delivers for x at pgan/generator.py:33-34) in the same form: a
 * stride-2 convolution of the fine gradient with 4 x 4 x 4 taps whose weights are the forward's summed sets transposed, 64 tap
 * products per low-resolution voxel instead of the 216 of the 27-tap data gradient followed by the 2x2x2 block sum, rounded
 * once.  `s`: the LOW-resolution shape (cin = channels of x and gx, cout = channels of gy); gy: [n,2d,2h,2w,cout],
 * gx: [n,d,h,w,cin]; w_dhwio is the FORWARD weight [3][3][3][cin][cout].  bf16, cin % 64 == 0, cout % 16 == 0, the tiles of the
 * forward kernel with 32- or 16-wide rows; otherwise SG_EUNSUPPORTED (run sg_conv3d_fwd with transposed weights and
 * sg_conv_epilogue.pool, then sg_downscale_sum). */
int sg_upconv3d_subpixel_dgrad_supported(const sg_conv_shape* s, sg_dtype dt);
size_t sg_upconv3d_subpixel_dgrad_packed_bytes(const sg_conv_shape* s, sg_dtype dt);
int sg_upconv3d_subpixel_dgrad_pack(const float* w_dhwio, float coef, void* wp, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
int sg_upconv3d_subpixel_dgrad(const void* gy, const void* wp, void* gx, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
/* dw[kD][kH][kW][cin][cout] (f32) = coef * sum_v x[v+tap] (x) dy[v]   (tf Conv3DBackpropFilterV2).
 * workspace: sg_conv3d_wgrad_workspace() bytes, contents irrelevant on entry. */
size_t sg_conv3d_wgrad_workspace(const sg_conv_shape* s, sg_dtype dt);
int sg_conv3d_wgrad(const void* x, const void* dy, float* dw_dhwio, float coef, void* workspace,
                    size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
/* Same, and dbias[cout] (f32, overwritten) = sum_v dy[v]: the gradient of apply_bias (networks/ops.py:130-136)
 * from the pass that reads dy anyway. */
int sg_conv3d_wgrad_bias(const void* x, const void* dy, float* dw_dhwio, float* dbias, float coef, void* workspace,
                         size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
/* The same for one discriminator block's pooled tail, downscale3d(leaky_relu(conv3d(x) + b)) (pgan/discriminator.py:39-44):
 * dy_half [n, d/2, h/2, w/2, cout] is the gradient of the POOLED output; the gradients are taken against
 * dy_gain * where(bit, mask_slope, 1) * upscale3d(dy_half) -- networks/ops.py:265-273 backward (dy_gain = 1/8) followed by
 * :175-178 with the layer's sign words mask_bits [n*d*h*w][cout/32] -- formed while each tile is staged, with the rounding
 * of sg_upscale2x_masked (dy_gain: a power of two), so the full-resolution gradient is never written.  s is the FINE shape.  bf16, 3x3x3,
 * cout % 32 == 0, even d/h/w, w % 32 == 0 and the sliding-halo kernel's tile: otherwise SG_EUNSUPPORTED.  Same workspace
 * as sg_conv3d_wgrad_bias. */
int sg_conv3d_wgrad_bias_up_masked(const void* x, const void* dy_half, const void* mask_bits, float mask_slope, float dy_gain,
                                   float* dw, float* dbias, float coef, void* workspace, size_t workspace_bytes,
                                   const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
/* sg_conv3d_wgrad_bias (mask_bits == NULL) / sg_conv3d_wgrad_bias_up_masked (mask_bits != NULL) with options:
 * SG_WGRAD_ACCUMULATE: dw_dhwio += coef * sum -- the weights are used more than once in the graph a training step
 *   differentiates (the discriminator under WGAN-GP: networks/loss.py:136-140 differentiates D a second time,
 *   optimization.py:128-163 asks for one gradient per variable) and the parameter's gradient buffer already holds the other
 *   contribution.  The sum is rounded to f32 before it is added: the value tf.gradients' add_n of the two finished gradients
 *   has.  dbias (optional) is WRITTEN, not added to.
 * SG_WGRAD_CLEAN_WORKSPACE: the caller keeps the workspace between calls; its first sg_conv3d_wgrad_clean_bytes(s, dt) bytes are
 *   zero on entry and are left zero (the finalize pass clears what it reads), and the call launches no memset.  (A failed
 *   call leaves the workspace undefined: clear it before the next use.)
 * SG_EUNSUPPORTED, with nothing touched, where the layer runs on the pointwise (<= 4 channels on one side) or the small-channel
 * (<= 16, 2-D top levels) kernels. */
#define SG_WGRAD_ACCUMULATE 1u
#define SG_WGRAD_CLEAN_WORKSPACE 2u
int sg_conv3d_wgrad_bias_ex(const void* x, const void* dy, const void* mask_bits, float mask_slope, float dy_gain,
                            float* dw_dhwio, float* dbias, float coef, unsigned flags, void* workspace, size_t workspace_bytes,
                            const sg_conv_shape* s, sg_dtype dt, sg_stream_t st);
size_t sg_conv3d_wgrad_clean_bytes(const sg_conv_shape* s, sg_dtype dt);

/* Whole backward of a pointwise convolution FROM cin <= 4 channels (from_rgb, pgan/discriminator.py:9-12) in one pass over
 * dy: dw, dbias as sg_conv3d_wgrad_bias, and dx[n,d,h,w,cin] = sum_c dy[..,c] * w_mat[j][c] (w_mat: [cin][cout] f32, the
 * values the forward multiplied with).  Other shapes: SG_EUNSUPPORTED (run sg_conv3d_fwd + sg_conv3d_wgrad_bias). */
int sg_conv3d_pw_bwd(const void* x, const void* dy, const float* w_mat, float* dw_dhwio, float* dbias, void* dx,
                     float coef, void* workspace, size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt,
                     sg_stream_t st);

/* ---- elementwise / reductions over NDHWC -------------------------------------------------------- */
/* y = act(x + bias[c])                       (apply_bias + act, networks/ops.py:130-136,185-192) */
int sg_bias_act_fwd(const void* x, const float* bias, void* y, int64_t nvox, int32_t c, int32_t act,
                    float slope, sg_dtype dt, sg_stream_t st);
/* dx = (y >= 0 ? dy : slope*dy) (mask from the OUTPUT, networks/ops.py:177); y == NULL: dx = dy.
 * dbias (optional, [c] f32, overwritten) = sum_v dx.  workspace >= sg_bias_act_bwd_workspace(c). */
size_t sg_bias_act_bwd_workspace(int32_t c);
int sg_bias_act_bwd(const void* dy, const void* y, void* dx, float* dbias, void* workspace, int64_t nvox,
                    int32_t c, float slope, sg_dtype dt, sg_stream_t st);
/* Same with the mask taken from the sign words of y (sg_sign_words / sg_conv_epilogue.sign_out). */
int sg_bias_act_bwd_bits(const void* dy, const void* y_sign_words, void* dx, float* dbias, void* workspace,
                         int64_t nvox, int32_t c, float slope, sg_dtype dt, sg_stream_t st);
/* y = x * rsqrt(mean_c(x^2) + eps); scale (optional, [nvox] f32) receives the rsqrt factor.
 * (pixel_norm, networks/ops.py:308-310) */
int sg_pixel_norm_fwd(const void* x, void* y, float* scale, int64_t nvox, int32_t c, float eps,
                      sg_dtype dt, sg_stream_t st);
/* dx = scale * (dy - y * mean_c(dy*y)),  y = forward OUTPUT, scale = forward rsqrt factor. */
int sg_pixel_norm_bwd(const void* dy, const void* y, const float* scale, void* dx, int64_t nvox,
                      int32_t c, sg_dtype dt, sg_stream_t st);
/* Gradient of y = pixel_norm(leaky_relu(z + b)) (one generator stage, pgan/generator.py:49-71) in one pass:
 * dz = mask(sign words of y) * pixel_norm_bwd(dy), dbias (optional, [c] f32) = sum_v dz.
 * workspace >= sg_bias_act_bwd_workspace(c) when dbias is given. */
int sg_pixel_norm_act_bwd(const void* dy, const void* y, const float* scale, const void* y_sign_words, float slope,
                          void* dz, float* dbias, void* workspace, int64_t nvox, int32_t c, sg_dtype dt,
                          sg_stream_t st);
/* The same when dy is the data gradient of a pointwise convolution to cs <= 4 channels (to_rgb of the stage's output,
 * pgan/generator.py:13-16,96-97): dy[v][c] = sum_j g_small[v][j] * w_small[j][c] is formed in registers from
 * g_small [nvox][cs] and w_small [cs][c] (f32, the values the forward multiplied with) instead of being written and read
 * back.  c a multiple of the 16-byte piece, c <= 512 (bf16) / 256 (f32); otherwise SG_EUNSUPPORTED. */
int sg_pixel_norm_act_bwd_pw(const void* g_small, int32_t cs, const float* w_small, const void* y, const float* scale,
                             const void* y_sign_words, float slope, void* dz, float* dbias, void* workspace,
                             int64_t nvox, int32_t c, sg_dtype dt, sg_stream_t st);
/* The same pass also returning the pointwise convolution's OWN gradients (to_rgb's filter and bias, pgan/generator.py:13-16), which
 * need exactly the two tensors it reads: dw_small [c][cs] (a [1,1,1,c,cs] filter) = coef_small * sum_v y[v][ch] * g_small[v][j],
 * db_small [cs] = sum_v g_small[v][j] -- instead of another pass over y (sg_conv3d_wgrad_bias: 1.07 GB at the benchmarked size).
 * dbias, dw_small, db_small: each optional.  cs == 1; otherwise SG_EUNSUPPORTED.  workspace >= sg_pixel_norm_act_bwd_pw_wg_workspace(c, cs). */
size_t sg_pixel_norm_act_bwd_pw_wg_workspace(int32_t c, int32_t cs);
int sg_pixel_norm_act_bwd_pw_wg(const void* g_small, int32_t cs, const float* w_small, const void* y, const float* scale,
                                const void* y_sign_words, float slope, void* dz, float* dbias, float* dw_small, float* db_small,
                                float coef_small, void* workspace, size_t workspace_bytes, int64_t nvox, int32_t c, sg_dtype dt,
                                sg_stream_t st);
/* y[n,2d,2h,2w,c] = gain * x[n,d,h,w,c] nearest-neighbour    (upscale3d / avg_unpool3d, ops.py:250-262) */
int sg_upscale2x(const void* x, void* y, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                 float gain, sg_dtype dt, sg_stream_t st);
/* Same with a fused LeakyReLU backward: y *= (bit ? mask_slope : 1), mask_bits = sign words of a tensor shaped like
 * y.  This is the gradient of `downscale3d(leaky_relu(.))` (pgan/discriminator.py:39-44) in one pass. */
int sg_upscale2x_masked(const void* x, void* y, const void* mask_bits, float mask_slope, int32_t n, int32_t d,
                        int32_t h, int32_t w, int32_t c, float gain, sg_dtype dt, sg_stream_t st);
/* y[n,d/2,h/2,w/2,c] = gain * sum of the 2x2x2 block; gain = 1/8 is downscale3d (ops.py:265-273),
 * gain = 1 is the gradient of upscale3d (ops.py:284).  (d,h,w) = INPUT extent, all even. */
int sg_downscale2x(const void* x, void* y, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                   float gain, sg_dtype dt, sg_stream_t st);
/* The same two operations with a factor of 1 or 2 PER DIMENSION (fd, fh, fw): (1,2,2) are upscale2d / downscale2d of the
 * 2-D tree on D == 1 tensors (SURFGAN_2D/networks/ops.py:176-231); (1,2,1) finishes a D x W-pooled convolution output.
 * sg_upscale_nn: x [n,d,h,w,c] -> y [n,d*fd,h*fh,w*fw,c], optional sign-word mask of a tensor shaped like y.
 * sg_upscale_nn_planes: the same values, y written as c / plane_channels separate NDHWC tensors of plane_channels
 * channels each, back to back (sg_conv_epilogue.x_plane_channels; 16-byte channel pieces, rows of >= 128 pieces).
 * sg_downscale_sum: x [n,d,h,w,c] (INPUT extent, divisible by the factors) -> y [n,d/fd,h/fh,w/fw,c] = gain * block sum. */
int sg_upscale_nn(const void* x, void* y, const void* mask_bits, float mask_slope, int32_t n, int32_t d, int32_t h,
                  int32_t w, int32_t c, int32_t fd, int32_t fh, int32_t fw, float gain, sg_dtype dt, sg_stream_t st);
int sg_upscale_nn_planes(const void* x, void* y, const void* mask_bits, float mask_slope, int32_t n, int32_t d, int32_t h,
                         int32_t w, int32_t c, int32_t fd, int32_t fh, int32_t fw, float gain, int32_t plane_channels,
                         sg_dtype dt, sg_stream_t st);
int sg_downscale_sum(const void* x, void* y, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c, int32_t fd,
                     int32_t fh, int32_t fw, float gain, sg_dtype dt, sg_stream_t st);
/* The same over M * x, M = where(sign bit, mask_slope, 1) from sign words shaped like x [n*d*h*w][ceil(c/32)]: the
 * gradient of sg_upscale_nn with a mask (networks/ops.py:175-178 composed with :265-273) in one pass. */
int sg_downscale_sum_masked(const void* x, const void* mask_bits, float mask_slope, void* y, int32_t n, int32_t d, int32_t h,
                            int32_t w, int32_t c, int32_t fd, int32_t fh, int32_t fw, float gain, sg_dtype dt, sg_stream_t st);
/* Trilinear x2 up-sampling with half-pixel centres (`align_corners=False`; BASELINE north_star names a trilinear
 * resampler, the reference itself only has the nearest one): adjoint == 0: x [n,d,h,w,c] -> y [n,2d,2h,2w,c];
 * adjoint != 0: the gradient, x = dy [n,2d,2h,2w,c] -> y = dx [n,d,h,w,c] ((d,h,w) is always the LOW-resolution
 * extent).  Trilinear x2 down-sampling with half-pixel centres is the 2x2x2 mean: sg_downscale2x(gain 1/8). */
int sg_trilinear_up2x(const void* x, void* y, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c, int32_t adjoint,
                      sg_dtype dt, sg_stream_t st);
/* out = wa*a + wb*b     (fade-in lerp pgan/generator.py:100-101, pgan/discriminator.py:105; b may be NULL) */
int sg_axpby(const void* a, const void* b, void* out, float wa, float wb, int64_t numel, sg_dtype dt,
             sg_stream_t st);
/* The same with the coefficients read from DEVICE memory, out = w[0]*a + w[1]*b (b NULL: w[0]*a, w[1] is not read): the
 * fade-in of a step captured as a hipGraph, whose alpha moves every step (networks/ops.py:4-23, optuna_objective.py:446-467:
 * the reference feeds alpha as a graph variable too).  Same f32 arithmetic as sg_axpby. */
int sg_axpby_dev(const void* a, const void* b, void* out, const float* w, int64_t numel, sg_dtype dt, sg_stream_t st);
/* out[s][i] = gamma[s]*a[s][i] + (1-gamma[s])*b[s][i], one f32 DEVICE weight per batch sample, f32 arithmetic, one rounding:
 * the gradient penalty's interpolates `gamma*real + (1-gamma)*fake` (networks/loss.py:70-71, :133-134). */
int sg_lerp_rows(const void* a, const void* b, const float* gamma, void* out, int32_t n, int64_t per_sample, sg_dtype dt,
                 sg_stream_t st);
/* out = x + stddev * N(0,1), counter-based Philox4x32-10 keyed by (seed, element index)
 * (instance noise, networks/loss.py:122-123). */
int sg_add_noise(const void* x, void* out, float stddev, uint64_t seed, uint64_t offset, int64_t numel,
                 sg_dtype dt, sg_stream_t st);
/* The same with the Philox offset read from DEVICE memory (one uint64) and, after the launch, advanced by `bump` on the
 * device: a captured hipGraph of the training step (SARAGAN_HIPGRAPH=1) draws fresh instance noise at every replay. */
int sg_add_noise_dev(const void* x, void* out, float stddev, uint64_t seed, uint64_t* offset_dev, uint64_t bump,
                     int64_t numel, sg_dtype dt, sg_stream_t st);
/* out[n*w_extent + w] = sum_{d,h,c} g[n,d,h,w,c]^2   (the reduce_sum of networks/loss.py:140, quirk Q1:
 * axes (1,2,3) of NCDHW = c,d,h).  out f32 [n*w], overwritten. */
int sg_sumsq_ndhwc_keep_w(const void* g, float* out, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                          sg_dtype dt, sg_stream_t st);
/* minibatch_stddev_layer (networks/ops.py:313-325): y[n,d,h,w,c+1] = concat(x, stat[n % (n/group)]) with
 * stat[m] = mean_{c,d,h,w} sqrt(var_group(x) + 1e-8), group = min(group_size, n), n % group == 0.
 * workspace: (n/group) floats. */
int sg_minibatch_stddev_fwd(const void* x, void* y, float* workspace, int32_t n, int64_t vox_per_sample,
                            int32_t c, int32_t group_size, sg_dtype dt, sg_stream_t st);
/* Its gradient: dx[n,d,h,w,c] from dy[n,d,h,w,c+1] and the forward input x (same workspace size). */
int sg_minibatch_stddev_bwd(const void* dy, const void* x, void* dx, float* workspace, int32_t n,
                            int64_t vox_per_sample, int32_t c, int32_t group_size, sg_dtype dt, sg_stream_t st);
/* dtype conversion between f32 and bf16 buffers (dst dtype = dt_dst). */
int sg_cast(const void* src, sg_dtype dt_src, void* dst, sg_dtype dt_dst, int64_t numel, sg_stream_t st);

/* ---- optimiser (optimization.py:16,28 tf.train.AdamOptimizer; ExtendedEMA.py:56-59) ------------ */
/* Fused TF-formulation Adam + EMA over one contiguous f32 range:
 *   m = b1*m + (1-b1)*g*gscale; v = b2*v + (1-b2)*(g*gscale)^2; p -= lr_t * m / (sqrt(v) + eps);
 *   if (ema) ema -= (1-ema_decay)*(ema - p)
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller.  g NULL => EMA only. */
int sg_adam_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t numel, float lr_t,
                float b1, float b2, float eps, float gscale, float ema_decay, sg_stream_t st);
/* The other optimisers optimization.py:17-22,29-35 can create, fused with the EMA update in the same way
 * (TF training_ops rules; g is scaled by gscale first; ema may be NULL):
 *   SG_OPT_SGD       tf.train.GradientDescentOptimizer      p -= lr*g
 *   SG_OPT_MOMENTUM  tf.train.MomentumOptimizer(momentum=h) s1 = h*s1 + g; p -= nesterov ? lr*g + lr*h*s1 : lr*s1
 *   SG_OPT_ADADELTA  tf.train.AdadeltaOptimizer(rho=h, eps) s1 = h*s1 + (1-h)*g^2; u = sqrt(s2+eps)*rsqrt(s1+eps)*g;
 *                                                          p -= lr*u; s2 = h*s2 + (1-h)*u^2
 * s1 / s2: f32 state ranges like p (unused ones may be NULL). */
enum { SG_OPT_SGD = 0, SG_OPT_MOMENTUM = 1, SG_OPT_ADADELTA = 2 };
int sg_optim_step(int kind, float* p, const float* g, float* s1, float* s2, float* ema, int64_t numel, float lr,
                  float h, float eps, int nesterov, float gscale, float ema_decay, sg_stream_t st);
/* sg_adam_ema / sg_optim_step with the step size read from DEVICE memory (one float: lr_t resp. lr): the optimiser launches
 * of a captured step, whose learning-rate schedule (optimization.py:227-296) and Adam bias correction stay host arithmetic
 * -- the host writes the value before each replay. */
int sg_adam_ema_dev(float* p, const float* g, float* m, float* v, float* ema, int64_t numel, const float* lr_t,
                    float b1, float b2, float eps, float gscale, float ema_decay, sg_stream_t st);
int sg_optim_step_dev(int kind, float* p, const float* g, float* s1, float* s2, float* ema, int64_t numel, const float* lr,
                      float h, float eps, int nesterov, float gscale, float ema_decay, sg_stream_t st);
/* out[i] = sum of squares of segment i (offsets[i]..offsets[i+1]) of a flat f32 buffer
 * (tf.norm per gradient + tf.clip_by_global_norm, optimization.py:66-71).  offsets: DEVICE int64[nseg+1]. */
int sg_segment_sumsq(const float* flat, const int64_t* offsets, float* out, int32_t nseg, sg_stream_t st);

/* ---- validation metrics on device tensors (metrics/swd.py:13-123, metrics/skim_metrics.py:8-45) ---------------- */
/* One axis of a separable FIR filter over x viewed as [outer, n, inner] (f32, or f64 when `f64` != 0; accumulated in
 * f64 either way): y = alpha * value + add (add may be NULL; it is shaped like y).
 *   mode 0: value[o] = sum_t taps[t] x[B(o + t - r)]                    extent n      (scipy.ndimage.gaussian_filter1d)
 *   mode 1: value[o] = sum_t taps[t] x[B(2o + t - r)]                   extent (n+1)/2  (pyr_down, swd.py:63-66:
 *           the 5x5x5 binomial filter is separable, then [::2])
 *   mode 2: value[o] = sum_t taps[t] z[B(o + t - r)], z[2k] = x[k], z[odd] = 0, extent 2n  (pyr_up, swd.py:69-74)
 * r = ntaps / 2, ntaps odd <= 15, taps on the HOST.  border B: 0 scipy 'mirror', 1 scipy 'reflect'.  x != y. */
int sg_filter_axis(const void* x, void* y, const void* add, int64_t outer, int32_t n, int64_t inner, const double* taps,
                   int32_t ntaps, int32_t mode, int32_t border, double alpha, int32_t f64, sg_stream_t st);
/* get_descriptors_for_minibatch (swd.py:13-26): x f32 [n_img, c, d, h, w] (NCDHW, as the metrics receive it);
 * out[nh, c, i, j, k] = x[nh / per_image, c, d0[nh] + i - rd, h0[nh] + k - rw, w0[nh] + j - rh], out shaped
 * [n_img * per_image, c, 2rd+1, 2rh+1, 2rw+1] -- the reference adds its fourth-axis offsets to the W centre and its
 * fifth-axis offsets to the H centre, and so does this.  d0 / h0 / w0: DEVICE int32[n_img * per_image], centres the
 * caller drew inside [r, extent - r). */
int sg_swd_gather(const float* x, float* out, const int32_t* d0, const int32_t* h0, const int32_t* w0, int32_t n_img,
                  int32_t c, int32_t d, int32_t h, int32_t w, int32_t per_image, int32_t rd, int32_t rh, int32_t rw,
                  sg_stream_t st);
/* finalize_descriptors (swd.py:31-39), in place: desc [n, c, inner] -> (desc - mean_c) / std_c over (n, inner). */
size_t sg_desc_normalize_workspace(int32_t c);
int sg_desc_normalize(float* desc, int64_t n, int32_t c, int64_t inner, void* workspace, size_t workspace_bytes,
                      sg_stream_t st);
/* sliced_wasserstein (swd.py:44-58) in three steps.  sg_swd_project: pt[dir][row] = sum_f a[row][f] * dirs[f][dir],
 * a [n, f], dirs [f, ndirs], pt [ndirs, npad] with npad = sg_swd_padded_rows(n) (a power of two >= 64; rows n.. are
 * +inf).  sg_sort_rows: every row of [rows, npad] ascending, in place (np.sort(axis=0) of the reference's layout).
 * sg_swd_distance: out[0] = mean_{row, i < n} |pa - pb|; out: 1 + rows doubles (out[1..] the row sums). */
int32_t sg_swd_padded_rows(int32_t n);
int sg_swd_project(const float* a, const float* dirs, float* pt, int32_t n, int32_t f, int32_t ndirs, int32_t npad,
                   sg_stream_t st);
int sg_sort_rows(float* data, int32_t rows, int32_t npad, sg_stream_t st);
int sg_swd_distance(const float* pa, const float* pb, double* out, int32_t rows, int32_t n, int32_t npad, sg_stream_t st);
/* skimage.metrics restated on f64 device buffers (skim_metrics.py:8-45); workspace: sg_metric_workspace() bytes.
 * sg_sqdiff_mean: out[0] = mean((a - b)^2).  sg_minmax: out[0] = min, out[1] = max.
 * sg_ssim_products: xx = x*x, yy = y*y, xy = x*y.  sg_ssim_mean: out[0] = mean over the channels-last map
 * [s0, s1, s2, c], cropped by crop0 on the first and crop on the other two spatial extents, of
 * (2 ux uy + c1)(2 vxy + c2) / ((ux^2 + uy^2 + c1)(vx + vy + c2)), v* = cov_norm * (u** - u* u*). */
size_t sg_metric_workspace(void);
int sg_sqdiff_mean(const double* a, const double* b, double* out, int64_t numel, void* workspace, size_t workspace_bytes,
                   sg_stream_t st);
int sg_minmax(const double* a, double* out, int64_t numel, void* workspace, size_t workspace_bytes, sg_stream_t st);
int sg_ssim_products(const double* x, const double* y, double* xx, double* yy, double* xy, int64_t numel, sg_stream_t st);
int sg_ssim_mean(const double* ux, const double* uy, const double* uxx, const double* uyy, const double* uxy, double* out,
                 int32_t s0, int32_t s1, int32_t s2, int32_t c, int32_t crop0, int32_t crop, double cov_norm, double c1,
                 double c2, void* workspace, size_t workspace_bytes, sg_stream_t st);

/* ---- opt-in kernel timing (used by bench.py's roofline leg) ------------------------------------- */
/* When enabled, every sg_conv3d_fwd / sg_conv3d_wgrad launch is bracketed by hipEvents on its stream.
 * sg_prof_collect synchronises those events and returns, per kind (0 = conv fwd, 1 = wgrad) and per
 * distinct shape, the launch count, total milliseconds and algorithmic FLOPs. */
typedef struct {
  int32_t kind;
  sg_conv_shape shape;
  int32_t dtype;
  int64_t launches;
  double total_ms;
  double flops_per_launch;
  char kernel[64];     /* kernel family + template arguments the dispatcher chose for this shape, e.g. "conv_fwd4<bf16,2,1,3,3,3>" */
} sg_prof_entry;
int sg_prof_enable(int on);
/* 1 while the timing is enabled.  A caller that captures launches into a hipGraph (optimization.StepGraph) must not capture
 * the profiler's event records: it checks this and runs that step eagerly. */
int sg_prof_enabled(void);
/* Restrict the timing to ONE (kind, shape) (NULL: every launch again): two event records per launch are not free, and a
 * throughput measurement that wants the dominant kernel's duration from inside its own timed region brackets only
 * that kernel's launches. */
int sg_prof_set_filter(int kind, const sg_conv_shape* shape);
int sg_prof_collect(sg_prof_entry* out, int32_t max_entries, int32_t* n_entries);

/* The SG_* environment switches (kernel-selection overrides for diagnosis, e.g. SG_FWD_NO_V4=1) are read once, at the
 * first launch.  Tools that change one between launches call this to take a new snapshot. */
int sg_config_reload(void);

#ifdef __cplusplus
}
#endif
#endif /* SARAGAN_HIP_H */
