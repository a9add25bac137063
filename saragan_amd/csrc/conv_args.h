// Launch arguments shared by the forward / data-gradient convolution kernels (conv3d.hip, conv3p.hip).
#pragma once
#include "common.h"

struct ConvFwdArgs {
  const void* x;
  const void* wp;
  void* y;
  const float* bias;
  float* pn_scale;
  const uint32_t* mask_bits;   // optional sign words [voxel][ntile]: output *= (bit ? mask_slope : 1)  (fused LeakyReLU backward)
  uint32_t* sign_out;          // optional sign words of THIS output (after bias/act), same layout, for a later mask_bits
  float mask_slope;
  sg_tile_geom g;
  int cin, cout, taps, kh, kw;
  int nchunk, ntile;    // K chunks of 32 B, output-channel tiles of 32
  int G, TG;            // chunks staged per K phase, taps staged per weight phase
  int rs;               // LDS row stride of the halo image (bytes)
  int xbytes;           // LDS bytes of the halo image
  sg_fastdiv fnp;       // fastdiv by pieces per halo row (G*2)
  int sshift, rshift;   // v2: log2(slots per row), log2(rows per 256-byte bank row)
  int wbytes, ntiles;   // v3r: resident weight image bytes, number of spatial tiles
  int wres;             // v4: all weight slabs resident in LDS
  int lean;             // v4: input below 2 GiB and no fused up-sampling: buffer addressing for the halo
  int tap_d, tap_h, tap_w;   // added to the tap index when addressing the halo (sub-pixel classes)
  int xcs, xco;              // sliding-halo kernel: channels per voxel of the x TENSOR and first channel of the slice convolved (K split)
  const float* addend;       // sliding-halo kernel, second K-split pass: the first pass's f32 partial sums [voxel][cout]
  const void* pnb_y;         // sliding-halo kernel, pixel-norm backward epilogue: the stage's output y [voxel][cout] ...
  const float* pnb_scale;    // ... and its per-voxel rsqrt factor (sg_conv_epilogue.pn_bwd_y / pn_bwd_scale)
  const uint32_t* in_mask;   // sliding-halo kernel with the fused nearest-x2 gather: sign words of the FINE input [voxel][in_mask_nw] ...
  float in_mask_slope, in_gain;   // ... the staged halo is in_gain * where(bit, in_mask_slope, 1) * x (sg_conv_epilogue.in_mask_bits)
  int in_mask_nw;
  const float* rgb_w;        // conv_fwd3w: to_rgb (one image channel) of the stored output in the epilogue: [cout] f32 ...
  const float* rgb_bias;     // ... + rgb_bias[0] (or NULL) ...
  void* rgb_out;             // ... -> [n,d,h,w,1]
  const void* pw_x;          // conv_fwd3w: from_rgb's backward in the epilogue of the data gradient for its output: its input image [n,d,h,w,1],
  const float* pw_wmat;      // the [cout] values its forward multiplied with,
  void* pw_dx;               // -> image gradient [n,d,h,w,1] (or NULL),
  float* pw_part;            // -> per-wave partial sums [rows][2][32] f32 (sum x * g, sum g): pw_wgrad_final_kernel adds them
  int pool;                  // 1: y is the D x W mean-pooled output [n, D/2, H, W/2, cout] (sliding-halo kernel only); 3: the 2 x 2 x 2 means (conv_fwd3w)
  int os, oa, ob, oc;        // output scatter: os == 2 writes voxel (2d+oa, 2h+ob, 2w+oc) of a [n,2D,2H,2W,cout] tensor
  unsigned long long* dbg;  // diagnostic time stamps (NULL in production)
  int dbg_flags;            // diagnostic ablations (0 in production): 1 = no re-staging, 2 = no epilogue
  int vec_in, vec_out;
  int act, pixel_norm;
  float slope, eps;
};

// Epilogue features of the sliding-halo / sliding-accumulator kernels (compile-time bits)
enum : int { SG_EP_SIGN = 1, SG_EP_MASK = 2, SG_EP_PN = 4, SG_EP_POOL = 8, SG_EP_PNB = 16,
              SG_EP_RGB = 32, SG_EP_PWB = 128, SG_EP_POOL3 = 256 /* with SG_EP_POOL: whole 2 x 2 x 2 blocks */ /* conv_fwd3w only: to_rgb of the output / from_rgb's backward in the epilogue */ };

// conv3p.hip: one-pass 64 -> 32 sliding-accumulator kernel (replaces the two-pass K split where it applies)
int sg_launch_fwd3p(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used);
// conv3w.hip: 32 -> 32k layers with wave-private halo planes and sliding accumulators on v_mfma_f32_16x16x32_bf16 (replaces the
// sliding-halo kernel where it applies; SG_FWD3S_16=0 switches it off)
// pw_rows: with ConvFwdArgs.pw_x the number of partial-sum rows the kernel wrote to pw_part (sg_pw_wgrad_finalize adds them)
int sg_launch_fwd3w(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used, int* pw_rows);
constexpr int SG_PW_PART_ROWS = 2048;      // rows of [2][32] f32 in sg_conv_epilogue.workspace for the fused from_rgb backward
// conv3w.hip: dw[c] = coef * sum_rows part[row][0][c], dbias[c] = sum_rows part[row][1][c], c < 32, rows added in order (either may be NULL)
int sg_pw_wgrad_finalize(const float* part, int rows, float* dw, float* dbias, float coef, hipStream_t st);
// The v_mfma_f32_16x16x32_bf16 kernels read a fragment image of their own, packed behind the standard one (conv3w.hip):
// bytes of that image (0: the layer has none) and the packer of n such images in one launch per 32 layers
size_t sg_pack16_bytes(const sg_conv_shape* s, sg_dtype dt);
int sg_pack16_batch(int n, const float* const* w, const float* coef, const int* flip, void* const* dst, const sg_conv_shape* shapes,
                    hipStream_t st);
