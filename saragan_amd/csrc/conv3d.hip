// Implicit-GEMM 3-D convolution (stride 1, 'SAME') on MFMA for gfx950: replaces tf.nn.conv3d as called by
// SURFGAN_3D/networks/ops.py:147-150 (and, with transposed+mirrored weights, its data gradient).
//
// Formulation ("im2col in LDS"): a 256-thread block owns a TN x TD x TH x TW tile of output voxels and ALL
// taps.  The NDHWC input halo of the tile is staged once per channel group into LDS as [halo voxel][channels];
// the im2col matrix is never built: the B operand of tap (i,j,l) is the same LDS image read at a row offset.
// GEMM view per tap: D[cout][voxel] += W_tap[cout][cin] * X_tap[cin][voxel]  (A = weights, B = activations),
// so that after the MFMA each lane owns ONE voxel and 16 output channels of it: bias, LeakyReLU and the
// pixel-norm channel reduction are lane-local (+ one cross-half shuffle).
//
// K is cut into 32-byte chunks (16 bf16 / 8 f32 input channels).  Packed weights are stored in fragment order
//   wp[chunk][tap][ntile][lane 0..63][16 B]      (1 KiB per fragment, lane = (cin-half h)*32 + (cout & 31))
// so a wave fetches a fragment with one lane-linear 1-KiB access (LDS-DMA friendly, conflict-free ds_read_b128).
#include "common.h"
#include "prof.h"
#include "conv_args.h"

// ------------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------------
struct PackArgs {
  const float* w;
  void* wp;
  float coef;
  int taps, kd, kh, kw, cin, cout, nchunk, ntile, flip;
};

template <typename T>
__global__ void pack_weights_kernel(PackArgs a) {
  constexpr int CH = sg_traits<T>::CH;
  constexpr int EPL = CH / 2;  // elements per lane per fragment
  const int64_t total = (int64_t)a.nchunk * a.taps * a.ntile * 64 * EPL;
  T* out = reinterpret_cast<T*>(a.wp);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int e = (int)(i % EPL);
    int64_t q = i / EPL;
    int lane = (int)(q % 64); q /= 64;
    int nt = (int)(q % a.ntile); q /= a.ntile;
    int tap = (int)(q % a.taps);
    int chunk = (int)(q / a.taps);
    int ci = chunk * CH + (lane >> 5) * EPL + e;
    int co = nt * 32 + (lane & 31);
    float v = 0.f;
    if (ci < a.cin && co < a.cout) {
      if (!a.flip) {
        v = a.w[((int64_t)tap * a.cin + ci) * a.cout + co];
      } else {  // source is [taps][cout][cin] mirrored in the taps
        v = a.w[((int64_t)(a.taps - 1 - tap) * a.cout + co) * a.cin + ci];
      }
      v *= a.coef;
    }
    out[i] = sg_traits<T>::from_f(v);
  }
}

// Many images in one launch (sg_conv3d_pack_weights_batch): blockIdx.y picks the item.  The table travels in the kernel
// arguments (<= 4 KiB): no device-side table to keep alive, and a captured launch carries its own copy.
constexpr int SG_PACK_BATCH = 56;
struct PackBatchArgs {
  PackArgs item[SG_PACK_BATCH];
};

template <typename T>
__global__ void pack_weights_batch_kernel(PackBatchArgs b) {
  const PackArgs& a = b.item[blockIdx.y];
  constexpr int CH = sg_traits<T>::CH;
  constexpr int EPL = CH / 2;
  const int64_t total = (int64_t)a.nchunk * a.taps * a.ntile * 64 * EPL;
  T* out = reinterpret_cast<T*>(a.wp);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int e = (int)(i % EPL);
    int64_t q = i / EPL;
    int lane = (int)(q % 64); q /= 64;
    int nt = (int)(q % a.ntile); q /= a.ntile;
    int tap = (int)(q % a.taps);
    int chunk = (int)(q / a.taps);
    int ci = chunk * CH + (lane >> 5) * EPL + e;
    int co = nt * 32 + (lane & 31);
    float v = 0.f;
    if (ci < a.cin && co < a.cout) {
      v = a.flip ? a.w[((int64_t)(a.taps - 1 - tap) * a.cout + co) * a.cin + ci] : a.w[((int64_t)tap * a.cin + ci) * a.cout + co];
      v *= a.coef;
    }
    out[i] = sg_traits<T>::from_f(v);
  }
}

static inline int conv_nchunk(const sg_conv_shape* s, sg_dtype dt) {
  int ch = dt == SG_BF16 ? 16 : 8;
  return sg_cdiv(s->cin, ch);
}
static inline int conv_ntile(const sg_conv_shape* s) { return sg_cdiv(s->cout, 32); }

static int conv_shape_ok(const sg_conv_shape* s) {
  if (!s) return 0;
  if (s->n < 1 || s->d < 1 || s->h < 1 || s->w < 1 || s->cin < 1 || s->cout < 1) return 0;
  const bool sub = s->kd == 2 && s->kh == 2 && s->kw == 2 && !s->upsample_in;   // one sub-pixel parity class
  if (s->kd < 1 || s->kh < 1 || s->kw < 1 || (!sub && (!(s->kd & 1) || !(s->kh & 1) || !(s->kw & 1)))) return 0;
  if (s->kd > 7 || s->kh > 7 || s->kw > 7) return 0;
  if (s->upsample_in && ((s->d | s->h | s->w) & 1)) return 0;
  return 1;
}

extern "C" size_t sg_conv3d_packed_bytes(const sg_conv_shape* s, sg_dtype dt) {
  if (!conv_shape_ok(s)) return 0;
  // [MFMA fragment image][plain f32 [taps][cin][cout] copy for the small-channel VALU kernels, where they take the layer]
  return (size_t)conv_nchunk(s, dt) * (s->kd * s->kh * s->kw) * conv_ntile(s) * 1024 + sg_small_tail_bytes(s) + sg_pack16_bytes(s, dt);
}

extern "C" int sg_conv3d_pack_weights(const float* w, float coef, int transpose_flip, void* wp,
                                      const sg_conv_shape* s, sg_dtype dt, sg_stream_t st) {
  if (!conv_shape_ok(s) || !w || !wp) return SG_EINVAL;
  PackArgs a;
  a.w = w; a.wp = wp; a.coef = coef;
  a.kd = s->kd; a.kh = s->kh; a.kw = s->kw; a.taps = s->kd * s->kh * s->kw;
  a.cin = s->cin; a.cout = s->cout; a.nchunk = conv_nchunk(s, dt); a.ntile = conv_ntile(s);
  a.flip = transpose_flip ? 1 : 0;
  int64_t total = (int64_t)a.nchunk * a.taps * a.ntile * 64 * (dt == SG_BF16 ? 8 : 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dt == SG_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(blocks), dim3(256), 0, sg_st(st), a);
  else
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, sg_st(st), a);
  SG_LAUNCH_CHECK();
  if (sg_pack16_bytes(s, dt)) {      // (these shapes have no small-channel tail: the second image follows the first directly)
    void* tail = reinterpret_cast<char*>(wp) + (size_t)a.nchunk * a.taps * a.ntile * 1024;
    return sg_pack16_batch(1, &w, &coef, &a.flip, &tail, s, sg_st(st));
  }
  if (sg_small_tail_bytes(s))
    return sg_small_pack(w, coef, a.flip, reinterpret_cast<char*>(wp) + (size_t)a.nchunk * a.taps * a.ntile * 1024, s, dt, sg_st(st));
  return SG_OK;
}

// The images of n layers (what n calls of sg_conv3d_pack_weights write), the MFMA-fragment images in one launch per 56 layers:
// after an optimiser step every layer's images are stale at once (44 pack launches per step at the benchmarked configuration).
extern "C" int sg_conv3d_pack_weights_batch(int n, const float* const* w, const float* coef, const int* transpose_flip,
                                            void* const* wp, const sg_conv_shape* shapes, sg_dtype dt, sg_stream_t st) {
  if (n < 0 || (n > 0 && (!w || !coef || !transpose_flip || !wp || !shapes))) return SG_EINVAL;
  if (dt != SG_BF16 && dt != SG_F32) return SG_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!conv_shape_ok(&shapes[i]) || !w[i] || !wp[i]) return SG_EINVAL;
  for (int i0 = 0; i0 < n; i0 += SG_PACK_BATCH) {
    const int m = n - i0 < SG_PACK_BATCH ? n - i0 : SG_PACK_BATCH;
    PackBatchArgs b;
    int64_t most = 0;
    for (int j = 0; j < SG_PACK_BATCH; ++j) {
      const int i = i0 + (j < m ? j : 0);         // (unused slots repeat the first item; their blocks are not launched)
      const sg_conv_shape* s = &shapes[i];
      PackArgs& a = b.item[j];
      a.w = w[i]; a.wp = wp[i]; a.coef = coef[i];
      a.kd = s->kd; a.kh = s->kh; a.kw = s->kw; a.taps = s->kd * s->kh * s->kw;
      a.cin = s->cin; a.cout = s->cout; a.nchunk = conv_nchunk(s, dt); a.ntile = conv_ntile(s);
      a.flip = transpose_flip[i] ? 1 : 0;
      const int64_t total = (int64_t)a.nchunk * a.taps * a.ntile * 64 * (dt == SG_BF16 ? 8 : 4);
      if (j < m && total > most) most = total;
    }
    int blocks = (int)((most + 255) / 256);
    if (blocks > 256) blocks = 256;               // (grid-stride: m x 256 blocks fill the chip several times over)
    if (blocks < 1) blocks = 1;
    if (dt == SG_BF16)
      hipLaunchKernelGGL(pack_weights_batch_kernel<bf16_t>, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, sg_st(st), b);
    else
      hipLaunchKernelGGL(pack_weights_batch_kernel<float>, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, sg_st(st), b);
    SG_LAUNCH_CHECK();
  }
  // the second images of the layers that have one: the 16x16x32 fragment images in one launch per 32 layers
  constexpr int M16 = 64;
  const float* w16[M16]; float c16[M16]; int f16[M16]; void* d16[M16]; sg_conv_shape s16[M16];
  int n16 = 0;
  for (int i = 0; i < n; ++i) {
    const sg_conv_shape* s = &shapes[i];
    char* tail = reinterpret_cast<char*>(wp[i]) + (size_t)conv_nchunk(s, dt) * (s->kd * s->kh * s->kw) * conv_ntile(s) * 1024;
    int rc = SG_OK;
    if (sg_pack16_bytes(s, dt)) {
      w16[n16] = w[i]; c16[n16] = coef[i]; f16[n16] = transpose_flip[i] ? 1 : 0; d16[n16] = tail; s16[n16] = *s;
      if (++n16 == M16) {
        rc = sg_pack16_batch(n16, w16, c16, f16, d16, s16, sg_st(st));
        n16 = 0;
      }
    } else if (sg_small_tail_bytes(s)) rc = sg_small_pack(w[i], coef[i], transpose_flip[i] ? 1 : 0, tail, s, dt, sg_st(st));
    if (rc != SG_OK) return rc;
  }
  if (n16) return sg_pack16_batch(n16, w16, c16, f16, d16, s16, sg_st(st));
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------

template <typename T> struct sg_vec4 { typedef u32x4 type; };
template <> struct sg_vec4<bf16_t> { typedef u32x2 type; };

// (sg_sign_word, sg_apply_sign_word, sg_lrelu, sg_pack_bf16, sg_store_tile_row_bf16: common.h, shared with subpix.hip)
// ---- epilogue shared by both forward kernels: lane owns voxel r of each M tile and output channels
// (i&3) + 8*(i>>2) + 4*hh of each N tile: bias, LeakyReLU and pixel-norm are lane-local (+1 shuffle).
template <typename T, int MTW, int NTB>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MTW][NTB], const int64_t (&ooff)[MTW],
                                              const ConvFwdArgs& a, int nt0, int hh) {
  T* y = reinterpret_cast<T*>(a.y);
  const float inv_c = 1.f / (float)a.cout;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    float ss = 0.f;
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = (nt0 + nt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        float v = acc[mt][nt][i];
        if (a.bias != nullptr && co < a.cout) v += a.bias[co];
        if (a.act) v = fmaxf(v, v * a.slope);
        acc[mt][nt][i] = v;
        ss += v * v;
      }
    }
    if (a.pixel_norm) {
      ss += __shfl_xor(ss, 32);
      const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
      for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][nt][i] *= sc;
      if (a.pn_scale != nullptr && hh == 0 && ooff[mt] >= 0) a.pn_scale[ooff[mt]] = sc;
    }
    if (a.sign_out != nullptr) {   // uniform branch: the shuffle inside needs every lane
#pragma unroll
      for (int nt = 0; nt < NTB; ++nt) {
        const uint32_t sw = sg_sign_word(acc[mt][nt], hh);
        if (hh == 0 && ooff[mt] >= 0 && nt0 + nt < a.ntile) a.sign_out[ooff[mt] * a.ntile + nt0 + nt] = sw;
      }
    }
    if (ooff[mt] >= 0) {
      T* yrow = y + ooff[mt] * (int64_t)a.cout;
      if (a.mask_bits != nullptr) {   // fused LeakyReLU backward of the layer that produced this conv's input gradient
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
          if (nt0 + nt < a.ntile)
            sg_apply_sign_word(acc[mt][nt], a.mask_bits[ooff[mt] * a.ntile + nt0 + nt], hh, a.mask_slope);
      }
#pragma unroll
      for (int nt = 0; nt < NTB; ++nt) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int co = (nt0 + nt) * 32 + 8 * qd + 4 * hh;
          if (a.vec_out && co + 4 <= a.cout) {
            T tmp[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) tmp[e] = sg_traits<T>::from_f(acc[mt][nt][qd * 4 + e]);
            if (sizeof(T) == 2)
              *reinterpret_cast<u32x2*>(yrow + co) = *reinterpret_cast<u32x2*>(tmp);
            else
              { const u32x4 t16 = *reinterpret_cast<u32x4*>(tmp); *reinterpret_cast<u32x4*>(yrow + co) = t16; SG_STORE16_GUARD(t16); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (co + e < a.cout) yrow[co + e] = sg_traits<T>::from_f(acc[mt][nt][qd * 4 + e]);
          }
        }
      }
    }
  }
}

template <typename T, int MTW, int NTB>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(ConvFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = sg_traits<T>::CH;
  constexpr int BM = MTW * 4 * 32;
  const sg_tile_geom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  char* xlds = smem;
  char* wlds = smem + a.xbytes;
  const T* x = reinterpret_cast<const T*>(a.x);
  const char* wp = reinterpret_cast<const char*>(a.wp);

  const sg_tile_origin o = sg_tile_of(g, blockIdx.x);
  const int nt0 = blockIdx.y * NTB;
  const int tvox = g.TN * g.TD * g.TH * g.TW;

  // per-lane rows (voxels) of this wave's M tiles
  int lbase[MTW];      // LDS byte offset of the voxel's halo row at tap (0,0,0)
  int64_t ooff[MTW];   // output voxel linear index, -1 if masked
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const int m = (wave * MTW + mt) * 32 + r;
    uint32_t q = sg_div((uint32_t)m, g.fTW);
    int tw = m - (int)q * g.TW;
    uint32_t q2 = sg_div(q, g.fTH);
    int th = (int)(q - q2 * g.TH);
    uint32_t q3 = sg_div(q2, g.fTD);
    int td = (int)(q2 - q3 * g.TD);
    int tn = (int)q3;
    const int n = o.n0 + tn, d = o.d0 + td, h = o.h0 + th, w = o.w0 + tw;
    const bool ok = (m < tvox) && n < g.N && d < g.D && h < g.H && w < g.W;
    lbase[mt] = ok ? ((((tn * g.HD + td) * g.HH + th) * g.HW + tw) * a.rs) : 0;
    ooff[mt] = ok ? ((((int64_t)n * g.D + d) * g.H + h) * g.W + w) : -1;
  }

  f32x16 acc[MTW][NTB];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  const int ntb = min(NTB, a.ntile - nt0);  // live N tiles of this block

  for (int c0 = 0; c0 < a.nchunk; c0 += a.G) {
    const int gcur = min(a.G, a.nchunk - c0);
    for (int t0 = 0; t0 < a.taps; t0 += a.TG) {
      const int tcur = min(a.TG, a.taps - t0);
      __syncthreads();  // everyone is done reading the previous phase's LDS
      if (t0 == 0)
        sg_stage_halo<T>(xlds, a.rs, x, g, o, a.cin, c0 * CH, a.G * 2, a.fnp, a.vec_in != 0, tid, 256);
      // weights of (chunks c0.., taps t0.., N tiles nt0..): LDS order [g][t][nt] fragments of 1 KiB
      {
        const int nfrag = gcur * tcur * NTB;
        for (int f = wave; f < nfrag; f += 4) {
          int nt = f % NTB;
          int q = f / NTB;
          int t = q % tcur;
          int gi = q / tcur;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (nt < ntb)
            v = *reinterpret_cast<const u32x4*>(
                wp + ((((int64_t)(c0 + gi) * a.taps + (t0 + t)) * a.ntile + (nt0 + nt)) << 10) + lane * 16);
          *reinterpret_cast<u32x4*>(wlds + ((size_t)f << 10) + lane * 16) = v;
        }
      }
      __syncthreads();
      for (int t = 0; t < tcur; ++t) {
        const int tap = t0 + t;
        const int kw_i = tap % a.kw;
        const int q = tap / a.kw;
        const int kh_i = q % a.kh;
        const int kd_i = q / a.kh;
        const int tapoff = ((kd_i * g.HH + kh_i) * g.HW + kw_i) * a.rs + hh * 16;
        for (int gi = 0; gi < gcur; ++gi) {
          u32x4 wf[NTB];
#pragma unroll
          for (int nt = 0; nt < NTB; ++nt)
            wf[nt] = *reinterpret_cast<const u32x4*>(wlds + ((size_t)((gi * tcur + t) * NTB + nt) << 10) + lane * 16);
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            const u32x4 xf = *reinterpret_cast<const u32x4*>(xlds + lbase[mt] + tapoff + gi * 32);
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) acc[mt][nt] = sg_mfma_chunk<T>(wf[nt], xf, acc[mt][nt]);
          }
        }
      }
    }
  }

  conv_epilogue<T, MTW, NTB>(acc, ooff, a, nt0, hh);
}

template <typename T, int MTW, int NTB>
static int launch_fwd(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st) {
  constexpr int BM = MTW * 128;
  constexpr int ES = (int)sizeof(T);
  a.g = sg_make_geom(s, BM);
  const sg_tile_geom& g = a.g;
  const int64_t ntiles = (int64_t)g.nTn * g.nTd * g.nTh * g.nTw;
  if (ntiles >= (1 << 24)) return SG_EINVAL;
  const int hv = g.TN * g.HD * g.HH * g.HW;
  // K phase sizing: G chunks (32 B each) of the halo + TG taps of weights must fit the LDS budget.
  const int budget = 64 * 1024;  // two blocks per CU
  int G = a.nchunk < 2 ? a.nchunk : 2;
  if (a.taps == 1) {  // 1x1x1 / dense: deepen K per phase instead
    while (G * 2 <= a.nchunk && G < 16 && (int64_t)hv * ((G * 2) * 32 + 16) + (int64_t)G * 2 * NTB * 1024 <= budget) G *= 2;
  }
  while (G > 1 && (int64_t)hv * (G * 32 + 16) + (int64_t)G * NTB * 1024 > 156 * 1024) G >>= 1;
  a.G = G;
  a.rs = G * 32 + 16;
  a.xbytes = hv * a.rs;
  a.xbytes = (a.xbytes + 1023) & ~1023;
  int64_t left = budget - a.xbytes;
  int TG = (int)(left / ((int64_t)G * NTB * 1024));
  if (TG < 1) {
    left = 160 * 1024 - a.xbytes;
    TG = (int)(left / ((int64_t)G * NTB * 1024));
    if (TG > 3) TG = 3;
  }
  if (TG < 1) return SG_EINVAL;
  if (TG > a.taps) TG = a.taps;
  a.TG = TG;
  a.fnp = sg_make_fastdiv(G * 2);
  a.vec_in = ((s->cin * ES) % 16 == 0) ? 1 : 0;
  a.vec_out = (s->cout % 4 == 0) ? 1 : 0;
  const size_t lds = (size_t)a.xbytes + (size_t)TG * G * NTB * 1024;
  if (lds > 160 * 1024) return SG_EINVAL;
  auto kern = conv_fwd_kernel<T, MTW, NTB>;
  SG_ALLOW_160K_LDS(kern);
  dim3 grid((unsigned)ntiles, (unsigned)sg_cdiv(a.ntile, NTB));
  SG_KNAME("conv_fwd<%s,%d,%d>", sg_tname<T>(), MTW, NTB);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  SG_LAUNCH_CHECK();
  return SG_OK;
}


// ------------------------------------------------------------------------------------------------------
// forward, v2: the tuned path for 16-byte-aligned channel counts.
//  * halo image rows are exactly G*32 bytes, no padding; 16-byte slot s of row r lives at slot s ^ f(r),
//    f(r) = (r / R) % S with S = slots per row, R = 16 / S rows per 256-byte bank row: a wave's ds_read_b128
//    of 32 consecutive rows is bank-conflict free for every tap shift (tiles are 32 voxels wide in W);
//  * halo and weights are staged by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip): the DMA
//    writes LDS lane-linearly, so the swizzle is applied on the SOURCE address; voxels outside the volume
//    read from a zero page;
//  * weights are double buffered: the slab of phase p+1 is in flight while phase p runs on the MFMAs; one
//    __syncthreads() per phase.
// ------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(1024))) uint32_t sg_zero_page[256] = {0};   // 1 KiB of zeros

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void sg_glds16(const void* g, char* lds_uniform_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)lds_uniform_base, 16, 0, 0);
}

template <typename T>
__device__ __forceinline__ void stage_halo_dma(char* xlds, const T* __restrict__ x, const sg_tile_geom& g,
                                               const sg_tile_origin& o, int cin, int c0, int S, int sshift,
                                               int rshift, int wave, int lane) {
  constexpr int EPP = 16 / (int)sizeof(T);
  const int hv = g.TN * g.HD * g.HH * g.HW;
  const int items = hv * S;
  const int Di = g.ups ? (g.D >> 1) : g.D, Hi = g.ups ? (g.H >> 1) : g.H, Wi = g.ups ? (g.W >> 1) : g.W;
  for (int base = wave * 64; base < items; base += 256) {
    const int it = base + lane;
    const int row = it >> sshift;
    const int p = it & (S - 1);
    const int sl = p ^ ((row >> rshift) & (S - 1));  // logical slot stored at physical slot p
    uint32_t v = (uint32_t)row;
    uint32_t q = sg_div(v, g.fHW);
    int hw = (int)(v - q * g.HW);
    uint32_t q2 = sg_div(q, g.fHH);
    int hh_ = (int)(q - q2 * g.HH);
    uint32_t q3 = sg_div(q2, g.fHD);
    int hd = (int)(q2 - q3 * g.HD);
    int n = o.n0 + (int)q3;
    int d = o.d0 + hd - g.PD, h = o.h0 + hh_ - g.PH, w = o.w0 + hw - g.PW;
    const int c = c0 + sl * EPP;
    const void* src = sg_zero_page;
    if (row < hv && n < g.N && (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H &&
        (unsigned)w < (unsigned)g.W && c < cin) {
      if (g.ups) { d >>= 1; h >>= 1; w >>= 1; }
      src = x + ((((int64_t)n * Di + d) * Hi + h) * Wi + w) * (int64_t)cin + c;
    }
    sg_glds16(src, xlds + (size_t)base * 16);
  }
}

template <typename T, int MTW, int NTB, int GC>
__global__ __launch_bounds__(256, 2) void conv_fwd2_kernel(ConvFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = sg_traits<T>::CH;
  const sg_tile_geom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  char* xlds = smem;
  char* wlds = smem + a.xbytes;
  const int wbuf_bytes = a.TG * a.G * NTB * 1024;
  const T* x = reinterpret_cast<const T*>(a.x);
  const char* wp = reinterpret_cast<const char*>(a.wp);
  const int S = a.G * 2;                      // 16-byte slots per halo row (power of two)
  const int sshift = a.sshift, rshift = a.rshift;
  const int rb = a.G * 32;

  const sg_tile_origin o = sg_tile_of(g, blockIdx.x);
  const int nt0 = blockIdx.y * NTB;
  const int tvox = g.TN * g.TD * g.TH * g.TW;

  int lrow[MTW];
  int64_t ooff[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const int m = (wave * MTW + mt) * 32 + r;
    uint32_t q = sg_div((uint32_t)m, g.fTW);
    int tw = m - (int)q * g.TW;
    uint32_t q2 = sg_div(q, g.fTH);
    int th = (int)(q - q2 * g.TH);
    uint32_t q3 = sg_div(q2, g.fTD);
    int td = (int)(q2 - q3 * g.TD);
    int tn = (int)q3;
    const int n = o.n0 + tn, d = o.d0 + td, h = o.h0 + th, w = o.w0 + tw;
    const bool ok = (m < tvox) && n < g.N && d < g.D && h < g.H && w < g.W;
    lrow[mt] = ok ? (((tn * g.HD + td) * g.HH + th) * g.HW + tw) : 0;
    ooff[mt] = ok ? ((((int64_t)n * g.D + d) * g.H + h) * g.W + w) : -1;
  }

  f32x16 acc[MTW][NTB];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  const int ntb = min(NTB, a.ntile - nt0);
  const int nphase_t = (a.taps + a.TG - 1) / a.TG;          // tap phases per channel group
  const int ngroup = (a.nchunk + a.G - 1) / a.G;
  const int nphase = nphase_t * ngroup;

  auto stage_w = [&](int p, char* dst) {
    const int grp = p / nphase_t, tp = p - grp * nphase_t;
    const int c0 = grp * a.G, t0 = tp * a.TG;
    const int gcur = min(a.G, a.nchunk - c0), tcur = min(a.TG, a.taps - t0);
    const int nfrag = gcur * tcur * NTB;
    for (int f = wave; f < nfrag; f += 4) {
      const int nt = f % NTB;
      const int q = f / NTB;
      const int t = q % tcur;
      const int gi = q / tcur;
      const char* src = nt < ntb ? wp + ((((int64_t)(c0 + gi) * a.taps + (t0 + t)) * a.ntile + (nt0 + nt)) << 10)
                                 : reinterpret_cast<const char*>(sg_zero_page);   // dead N tile: zero weights
      sg_glds16(src + lane * 16, dst + ((size_t)f << 10));
    }
  };

  stage_halo_dma<T>(xlds, x, g, o, a.cin, 0, S, sshift, rshift, wave, lane);
  stage_w(0, wlds);
  __syncthreads();

  for (int p = 0; p < nphase; ++p) {
    const int grp = p / nphase_t, tp = p - grp * nphase_t;
    const int c0 = grp * a.G, t0 = tp * a.TG;
    const int gcur = min(a.G, a.nchunk - c0), tcur = min(a.TG, a.taps - t0);
    const char* wcur = wlds + (p & 1) * wbuf_bytes;
    const bool group_ends = (tp == nphase_t - 1);
    if (p + 1 < nphase && !group_ends) stage_w(p + 1, wlds + ((p + 1) & 1) * wbuf_bytes);

    // one step = one tap (all chunks of the group); fragments of step t+1 are fetched before the MFMAs of step t
    int kw_i = t0 % a.kw, kh_i = (t0 / a.kw) % a.kh, kd_i = t0 / (a.kw * a.kh);
    auto load_step = [&](int t, u32x4 (&wf)[GC][NTB], u32x4 (&xf)[GC][MTW]) {
      const int taprow = (kd_i * g.HH + kh_i) * g.HW + kw_i;
      if (++kw_i == a.kw) { kw_i = 0; if (++kh_i == a.kh) { kh_i = 0; ++kd_i; } }
#pragma unroll
      for (int gi = 0; gi < GC; ++gi) {
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
          wf[gi][nt] = *reinterpret_cast<const u32x4*>(wcur + ((size_t)((gi * tcur + t) * NTB + nt) << 10) + lane * 16);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int row = lrow[mt] + taprow;
          const int sl = (gi * 2 + hh) ^ ((row >> rshift) & (S - 1));
          xf[gi][mt] = *reinterpret_cast<const u32x4*>(xlds + row * rb + (sl << 4));
        }
      }
    };
    auto mma_step = [&](const u32x4 (&wf)[GC][NTB], const u32x4 (&xf)[GC][MTW]) {
#pragma unroll
      for (int gi = 0; gi < GC; ++gi) {
        if (gi < gcur) {
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) acc[mt][nt] = sg_mfma_chunk<T>(wf[gi][nt], xf[gi][mt], acc[mt][nt]);
        }
      }
    };
    {
      u32x4 wA[GC][NTB], xA[GC][MTW], wB[GC][NTB], xB[GC][MTW];
      load_step(0, wA, xA);
      int t = 0;
      for (; t + 2 <= tcur; t += 2) {
        load_step(t + 1, wB, xB);
        mma_step(wA, xA);
        if (t + 2 < tcur) load_step(t + 2, wA, xA);
        mma_step(wB, xB);
      }
      if (t < tcur) mma_step(wA, xA);
    }
    __syncthreads();
    if (group_ends && p + 1 < nphase) {   // next channel group: restage the halo (single buffer) + its first slab
      stage_halo_dma<T>(xlds, x, g, o, a.cin, (grp + 1) * a.G * CH, S, sshift, rshift, wave, lane);
      stage_w(p + 1, wlds + ((p + 1) & 1) * wbuf_bytes);
      __syncthreads();
    }
  }
  conv_epilogue<T, MTW, NTB>(acc, ooff, a, nt0, hh);
}

template <typename T, int MTW, int NTB, int GC>
static int launch_fwd2(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st) {
  constexpr int BM = MTW * 128;
  a.g = sg_make_geom(s, BM, /*prefer_w32=*/true);
  const sg_tile_geom& g = a.g;
  const int64_t ntiles = (int64_t)g.nTn * g.nTd * g.nTh * g.nTw;
  if (ntiles >= (1 << 24)) return SG_EINVAL;
  const int hv = g.TN * g.HD * g.HH * g.HW;
  const int lds_cap = sg_cfg().fwd_lds;   // per block; 80 KiB = two blocks per CU
  const int G = GC;
  a.G = G;
  a.rs = G * 32;
  a.sshift = G == 1 ? 1 : (G == 2 ? 2 : 3);
  a.rshift = G == 1 ? 3 : (G == 2 ? 2 : 1);
  a.xbytes = ((hv * a.rs) + 1023) & ~1023;
  int cap = lds_cap;
  if (a.xbytes + 2 * G * NTB * 1024 > cap) cap = 160 * 1024;
  int TG = (cap - a.xbytes) / (2 * G * NTB * 1024);
  if (TG < 1) return SG_EINVAL;
  if (TG > a.taps) TG = a.taps;
  // even out the phases (e.g. 27 taps with room for 10 -> 3 phases of 9)
  const int nph = sg_cdiv(a.taps, TG);
  TG = sg_cdiv(a.taps, nph);
  if (sg_cfg().fwd_tg > 0) TG = sg_cfg().fwd_tg;
  a.TG = TG;
  a.vec_in = 1;
  a.vec_out = (s->cout % 4 == 0) ? 1 : 0;
  const size_t lds = (size_t)a.xbytes + 2ull * TG * G * NTB * 1024;
  if (lds > 160 * 1024) return SG_EINVAL;
  auto kern = conv_fwd2_kernel<T, MTW, NTB, GC>;
  SG_ALLOW_160K_LDS(kern);
  SG_KNAME("conv_fwd2<%s,%d,%d,%d>", sg_tname<T>(), MTW, NTB, GC);
  dim3 grid((unsigned)ntiles, (unsigned)sg_cdiv(a.ntile, NTB));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  SG_LAUNCH_CHECK();
  return SG_OK;
}



// Fully unrolled, software-pipelined K loop of the weight-stationary kernel (see the call site).  Template
// recursion keeps every LDS offset and wait count an immediate.
// Address-table index of operand fragment (tap, M tile).  With HS ("H shift") the wave's MTW M tiles are consecutive
// H rows of the tile, so the fragment of (kd, kh, kw) for M tile mt IS the fragment of (kd, kh + mt, kw) for M tile 0:
// the table holds KD x (KH + MTW - 1) x KW addresses instead of KD x KH x KW x MTW (36 instead of 54 registers at 3x3x3
// with two M tiles, which is what kept <2,2,3,3,3> from fitting 256 VGPRs next to its 64 accumulators; 54 instead of
// 108 with four).
// RS: H rows between the wave's consecutive M tiles -- 1 with 32-wide tile rows (an M tile is one row), 2 with 16-wide
// rows (an M tile is two rows): KD x (KH + RS (MTW - 1)) x KW addresses.
template <int MTW, int KH, int KW, bool HS, int RS = 1>
__host__ __device__ constexpr int sg_xa_index(int tap, int mt) {
  return HS ? ((tap / (KH * KW)) * (KH + RS * (MTW - 1)) + (tap / KW) % KH + RS * mt) * KW + tap % KW : tap * MTW + mt;
}
template <int MTW, int KD, int KH, int KW, bool HS, int RS = 1>
struct sg_xa_size { static constexpr int value = HS ? KD * (KH + RS * (MTW - 1)) * KW : KD * KH * KW * MTW; };

template <typename T, int MTW, int GC, int TAPS, int RING, int KH, int KW, bool HS, int NA>
struct sg_unrolled_k {
  static constexpr int NS = TAPS * GC, PF = RING - 1, RPS = 1 + MTW;
  template <int ST>
  static __device__ __forceinline__ void load(u32x4 (&wfr)[RING], u32x4 (&xfr)[RING][MTW], const int (&xaddr)[NA],
                                              int wl_off) {
    constexpr int SL = ST % RING, tap = ST / GC, gi = ST % GC;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wfr[SL]) : "v"(wl_off), "n"(ST << 10));
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      int xa = xaddr[sg_xa_index<MTW, KH, KW, HS>(tap, mt)];
      // chunk gi sits in slot 2*gi + hh (XOR-swizzled): toggled here, opaque to the compiler, which would otherwise
      // keep a second copy of the whole address table in registers
      if constexpr (gi != 0) asm volatile("v_xor_b32_e32 %0, %1, %2" : "=v"(xa) : "n"(gi << 5), "v"(xaddr[sg_xa_index<MTW, KH, KW, HS>(tap, mt)]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(xfr[SL][mt]) : "v"(xa));
    }
  }
  template <int ST>
  static __device__ __forceinline__ void step(f32x16 (&acc)[MTW], u32x4 (&wfr)[RING], u32x4 (&xfr)[RING][MTW],
                                              const int (&xaddr)[NA], int wl_off) {
    if constexpr (ST < NS) {
      if constexpr (ST + PF < NS) load<ST + PF>(wfr, xfr, xaddr, wl_off);
      constexpr int younger = (NS - 1 - ST < PF ? NS - 1 - ST : PF) * RPS;   // reads issued after step ST's
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[mt] = sg_mfma_chunk<T>(wfr[ST % RING], xfr[ST % RING][mt], acc[mt]);
      __builtin_amdgcn_sched_barrier(0);
      step<ST + 1>(acc, wfr, xfr, xaddr, wl_off);
    }
  }
  template <int ST>
  static __device__ __forceinline__ void prologue(u32x4 (&wfr)[RING], u32x4 (&xfr)[RING][MTW],
                                                  const int (&xaddr)[NA], int wl_off) {
    if constexpr (ST < PF && ST < NS) {
      load<ST>(wfr, xfr, xaddr, wl_off);
      prologue<ST + 1>(wfr, xfr, xaddr, wl_off);
    }
  }
  static __device__ __forceinline__ void run(f32x16 (&acc)[MTW], u32x4 (&wfr)[RING], u32x4 (&xfr)[RING][MTW],
                                             const int (&xaddr)[NA], int wl_off) {
    SG_KLOOP_BEGIN();
    prologue<0>(wfr, xfr, xaddr, wl_off);
    step<0>(acc, wfr, xfr, xaddr, wl_off);
    SG_KLOOP_END();
  }
};


// Same for the streamed-weight kernel (v4): one step = one tap of the current 32-byte channel chunk:
// NTB weight fragments + MTW activation fragments, MTW*NTB MFMAs.
template <typename T, int MTW, int NTB, int TAPS, int RING, int KH, int KW, bool HS, int NA, int RS = 1>
struct sg_unrolled_k4 {
  static constexpr int PF = RING - 1, RPS = NTB + MTW;
  template <int ST>
  static __device__ __forceinline__ void load(u32x4 (&wfr)[RING][NTB], u32x4 (&xfr)[RING][MTW],
                                              const int (&xaddr)[NA], int wl_off) {
    constexpr int SL = ST % RING;
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wfr[SL][nt]) : "v"(wl_off), "n"((ST * NTB + nt) << 10));
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
      asm volatile("ds_read_b128 %0, %1" : "=v"(xfr[SL][mt]) : "v"(xaddr[sg_xa_index<MTW, KH, KW, HS, RS>(ST, mt)]));
  }
  template <int ST>
  static __device__ __forceinline__ void step(f32x16 (&acc)[MTW][NTB], u32x4 (&wfr)[RING][NTB], u32x4 (&xfr)[RING][MTW],
                                              const int (&xaddr)[NA], int wl_off) {
    if constexpr (ST < TAPS) {
      if constexpr (ST + PF < TAPS) load<ST + PF>(wfr, xfr, xaddr, wl_off);
      constexpr int younger = (TAPS - 1 - ST < PF ? TAPS - 1 - ST : PF) * RPS;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
          acc[mt][nt] = sg_mfma_chunk<T>(wfr[ST % RING][nt], xfr[ST % RING][mt], acc[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
      step<ST + 1>(acc, wfr, xfr, xaddr, wl_off);
    }
  }
  template <int ST>
  static __device__ __forceinline__ void prologue(u32x4 (&wfr)[RING][NTB], u32x4 (&xfr)[RING][MTW],
                                                  const int (&xaddr)[NA], int wl_off) {
    if constexpr (ST < PF && ST < TAPS) {
      load<ST>(wfr, xfr, xaddr, wl_off);
      prologue<ST + 1>(wfr, xfr, xaddr, wl_off);
    }
  }
  static __device__ __forceinline__ void run(f32x16 (&acc)[MTW][NTB], const int (&xaddr)[NA], int wl_off) {
    u32x4 wfr[RING][NTB], xfr[RING][MTW];
    SG_KLOOP_BEGIN();
    prologue<0>(wfr, xfr, xaddr, wl_off);
    step<0>(acc, wfr, xfr, xaddr, wl_off);
    SG_KLOOP_END();
  }
};

// ------------------------------------------------------------------------------------------------------
// forward, v3r: persistent, weight-stationary variant for Cin <= one channel group (32 bf16 / 16 f32 channels).
// One block per CU owns a 32-wide slice of output channels (blockIdx.y) and walks spatial tiles.  ALL weights of
// the slice stay resident in LDS (27 taps x GC KiB); the halo image is double buffered and the next tile's halo
// is fetched by LDS-DMA while the MFMAs run on the current one, so the only per-tile synchronisation is one
// barrier and HBM/L2 traffic per tile is the halo alone.  Tiles are dealt so that the blocks of one XCD
// (blockIdx.x % 8, the round-robin dispatch group) work on spatially adjacent tiles: halo overlap hits that
// XCD's L2 (placement affects speed only).
// ------------------------------------------------------------------------------------------------------
template <typename T, int MTW, int GC, int KD, int KH, int KW>
__global__ __launch_bounds__(512) void conv_fwd3r_kernel(ConvFwdArgs a) {
  // 8 waves = two groups of 4 (one wave of each group per SIMD).  Phase p: group (p & 1) runs the MFMA loop of
  // tile p out of ITS halo buffer; the other group stores tile p-1 (accumulators stay in registers across the
  // barrier) and fetches tile p+1 into its own buffer.  One workgroup barrier per phase.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTB = 1;
  constexpr int TAPS = KD * KH * KW;
  constexpr int EPP = 16 / (int)sizeof(T);
  const sg_tile_geom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int r = lane & 31, hh = lane >> 5;
  // LDS map: [halo buffer of group 0][halo buffer of group 1][resident weights: [tap][gi] fragments of 1 KiB]
  char* xmine = smem + grp * a.xbytes;
  char* wlds = smem + 2 * a.xbytes;
  const T* x = reinterpret_cast<const T*>(a.x);
  const char* wp = reinterpret_cast<const char*>(a.wp);
  constexpr int S = GC * 2;
  constexpr int sshift = GC == 1 ? 1 : (GC == 2 ? 2 : 3);
  constexpr int rshift = GC == 1 ? 3 : (GC == 2 ? 2 : 1);
  constexpr int rb = GC * 32;
  const int nt0 = blockIdx.y;
  const int tvox = g.TN * g.TD * g.TH * g.TW;

  // tile schedule: XCD group xg = blockIdx.x % 8 owns a contiguous chunk of the tile list
  const int xg = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (a.ntiles + 7) >> 3;
  const int t_begin = xg * cpx;
  const int t_end = min(a.ntiles, t_begin + cpx);
  const int first = t_begin + slot;
  const int K = first < t_end ? (t_end - first + per_x - 1) / per_x : 0;   // tiles of this block

  // ---- tile-invariant per-lane state ----------------------------------------------------------------
  // LDS byte address of every (tap, M tile) operand fragment, chunk 0, own buffer (HS: see sg_xa_index; the host
  // only launches an HS instantiation on tiles where M tile 1 is M tile 0 moved by one H row)
  constexpr bool HS = (MTW == 2 && GC == 2);
  constexpr int NA = sg_xa_size<MTW, KD, KH, KW, HS>::value;
  int xaddr[NA];
  int tcoord[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const int m = (wave * MTW + mt) * 32 + r;
    uint32_t q = sg_div((uint32_t)m, g.fTW);
    int tw = m - (int)q * g.TW;
    uint32_t q2 = sg_div(q, g.fTH);
    int th = (int)(q - q2 * g.TH);
    uint32_t q3 = sg_div(q2, g.fTD);
    int td = (int)(q2 - q3 * g.TD);
    int tn = (int)q3;
    const int lrow = (m < tvox) ? (((tn * g.HD + td) * g.HH + th) * g.HW + tw) : 0;
    tcoord[mt] = (m < tvox) ? (tw | (th << 8) | (td << 16) | (tn << 24)) : -1;
    if (HS && mt > 0) continue;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd)
#pragma unroll
      for (int kh = 0; kh < KH + (HS ? MTW - 1 : 0); ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const int row = lrow + (kd * g.HH + kh) * g.HW + kw;
          const int idx = HS ? (kd * (KH + MTW - 1) + kh) * KW + kw : ((kd * KH + kh) * KW + kw) * MTW + mt;
          xaddr[idx] = grp * a.xbytes + row * rb + (((hh) ^ ((row >> rshift) & (S - 1))) << 4);
        }
  }
  // halo staging items of this lane: element offset relative to the tile's first halo voxel + packed coords
  const int hv = g.TN * g.HD * g.HH * g.HW;
  const int items = hv * S;
  constexpr int MAXIT = 14;   // LDS-DMA pieces per wave per tile (host checks)
  int it_rel[MAXIT], it_crd[MAXIT];
  const int Di = g.ups ? (g.D >> 1) : g.D, Hi = g.ups ? (g.H >> 1) : g.H, Wi = g.ups ? (g.W >> 1) : g.W;
#pragma unroll
  for (int k = 0; k < MAXIT; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> sshift;
    const int p = it & (S - 1);
    const int sl = p ^ ((row >> rshift) & (S - 1));
    uint32_t v = (uint32_t)row;
    uint32_t q = sg_div(v, g.fHW);
    int hw = (int)(v - q * g.HW);
    uint32_t q2 = sg_div(q, g.fHH);
    int hh_ = (int)(q - q2 * g.HH);
    uint32_t q3 = sg_div(q2, g.fHD);
    int hd = (int)(q2 - q3 * g.HD);
    const int c = sl * EPP;
    const bool live = row < hv && c < a.cin;
    it_crd[k] = live ? (hw | (hh_ << 8) | (hd << 16) | ((int)q3 << 24)) : 0x7F7F7F7F;   // dead: fails every range
    it_rel[k] = live ? ((((int)q3 * g.D + hd) * g.H + hh_) * g.W + hw) * a.cin + c : -1;
  }

  // Halo staging by LDS-DMA: asynchronous (lands while this group runs its epilogue), no VGPRs held.  A piece
  // costs 100-200 issue cycles on THIS wave only; the MFMA group on the same SIMDs keeps running.
  // a.lean: buffer addressing -- a scalar resource rebased to the tile's first sample, a scalar tile offset and the
  // per-lane 32-bit offset computed once; pieces outside the volume carry an out-of-range offset and read zeros
  // (64-bit per-piece pointers and zero-page selects had the compiler keep 28 more registers live: spills).
  constexpr uint32_t DEAD = 0x80000000u;
  const int64_t sample_bytes = (int64_t)Di * Hi * Wi * a.cin * (int)sizeof(T);
  auto stage_tile = [&](const sg_tile_origin& o) {
    if (a.lean) {
      const bool interior = o.d0 >= g.PD && o.h0 >= g.PH && o.w0 >= g.PW && o.d0 + g.TD + g.PD <= g.D &&
                            o.h0 + g.TH + g.PH <= g.H && o.w0 + g.TW + g.PW <= g.W && o.n0 + g.TN <= g.N;
      // boundary tile: a piece is valid iff its packed halo coordinate (w,h,d,n bytes, all < 128) lies inside the
      // per-tile range [lo, hi] in every byte; two byte-parallel subtractions with the borrow guard bit 7 test it
      const int lo_w = max(0, g.PW - o.w0), hi_w = min(g.HW, g.W + g.PW - o.w0) - 1;
      const int lo_h = max(0, g.PH - o.h0), hi_h = min(g.HH, g.H + g.PH - o.h0) - 1;
      const int lo_d = max(0, g.PD - o.d0), hi_d = min(g.HD, g.D + g.PD - o.d0) - 1;
      const int hi_n = min(g.TN, g.N - o.n0) - 1;
      const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8) | (lo_d << 16));
      const uint32_t hi = (uint32_t)(hi_w | (hi_h << 8) | (hi_d << 16) | (hi_n << 24)) | 0x80808080u;
      const int64_t left = (int64_t)(g.N - o.n0) * sample_bytes;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.x)) + (int64_t)o.n0 * sample_bytes, 0,
          (int)(left < 0x7FFFFFFFll ? left : 0x7FFFFFFFll), 0x00020000);
      const int tile_off = (int)((((((int64_t)(o.d0 - g.PD)) * g.H + (o.h0 - g.PH)) * g.W + (o.w0 - g.PW)) *
                                  (int64_t)a.cin) * (int)sizeof(T));
#pragma unroll
      for (int k = 0; k < MAXIT; ++k) {
        if ((wave + 4 * k) * 64 < items) {
          bool ok = it_rel[k] >= 0;
          if (!interior) {
            const uint32_t c_ = (uint32_t)it_crd[k];
            const uint32_t t1 = (c_ | 0x80808080u) - lo, t2 = hi - c_;
            ok = ok && ((t1 & t2 & 0x80808080u) == 0x80808080u);
          }
          const uint32_t vo = ok ? (uint32_t)(it_rel[k] * (int)sizeof(T) + tile_off) : DEAD;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xmine + (size_t)(wave + 4 * k) * 1024), 16, vo, 0, 0, 0);
        }
      }
    } else {   // fused nearest-x2 gather / tensors beyond the 31-bit offset range: per-piece coordinates
#pragma unroll 1
      for (int k = 0; k < MAXIT; ++k) {
        if ((wave + 4 * k) * 64 < items) {
          const int crd = it_crd[k];
          int n = o.n0 + (crd >> 24);
          int d = o.d0 + ((crd >> 16) & 255) - g.PD, h = o.h0 + ((crd >> 8) & 255) - g.PH,
              w = o.w0 + (crd & 255) - g.PW;
          const void* src = sg_zero_page;
          if (it_rel[k] >= 0 && n < g.N && (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H &&
              (unsigned)w < (unsigned)g.W) {
            if (g.ups) { d >>= 1; h >>= 1; w >>= 1; }
            const int row = ((wave + 4 * k) * 64 + lane) >> sshift;
            const int sl_c = ((lane & (S - 1)) ^ ((row >> rshift) & (S - 1))) * EPP;
            src = x + ((((int64_t)n * Di + d) * Hi + h) * Wi + w) * (int64_t)a.cin + sl_c;
          }
          sg_glds16(src, xmine + (size_t)(wave + 4 * k) * 1024);
        }
      }
    }
  };

  // resident weights (all 8 waves)
  {
    constexpr int nfrag = TAPS * GC;
    for (int f = wave8; f < nfrag; f += 8) {
      const int tap = f / GC, gi = f - tap * GC;
      const char* src = gi < a.nchunk ? wp + ((((int64_t)gi * TAPS + tap) * a.ntile + nt0) << 10)
                                      : reinterpret_cast<const char*>(sg_zero_page);
      sg_glds16(src + lane * 16, wlds + ((size_t)f << 10));
    }
  }
  float* bias_lds = reinterpret_cast<float*>(wlds + a.wbytes);   // 32 floats: bias of this block's channel slice
  if (tid < 32) {
    const int co = nt0 * 32 + tid;
    bias_lds[tid] = (a.bias != nullptr && co < a.cout) ? a.bias[co] : 0.f;
  }
  if (grp == 0 && K > 0) stage_tile(sg_tile_of(g, (uint32_t)first));
  __syncthreads();

  const char* wl = wlds + lane * 16;
  T* y = reinterpret_cast<T*>(a.y);
  const float inv_c = 1.f / (float)a.cout;
  f32x16 acc[MTW];

  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && lane == 0 && wave == 0 && dbgi < 120)
      a.dbg[grp * 128 + dbgi] = __builtin_amdgcn_s_memtime();
    ++dbgi;
  };
  for (int p = 0; p <= K; ++p) {
    stamp();
    if ((p & 1) == grp) {
      if (p < K) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mt][i] = bias_lds[(i & 3) + 8 * (i >> 2) + 4 * hh];   // bias rides in C
        // K loop: NS = TAPS*GC steps of (1 weight + MTW activation) fragment reads and MTW MFMAs.  The reads are
        // inline asm so that the waits can be COUNTED: hipcc only emits lgkmcnt(0) here, which also waits for the
        // fragments just requested PF steps ahead and exposes a full LDS round trip every few MFMAs.  LDS returns
        // in order, so before the MFMAs of step s at most PF*(1+MTW) younger reads may still be in flight.
        constexpr int NS = TAPS * GC;
        constexpr int RING = 3, PF = RING - 1;
        constexpr int RPS = 1 + MTW;                    // reads per step
        u32x4 wfr[RING], xfr[RING][MTW];
        const int wl_off = (int)(wl - smem);            // LDS byte address of this lane's weight column
        (void)RPS; (void)NS; (void)PF;
        sg_unrolled_k<T, MTW, GC, TAPS, RING, KH, KW, HS, NA>::run(acc, wfr, xfr, xaddr, wl_off);
      }
    } else {
      if (a.dbg_flags & 4) __builtin_amdgcn_s_setprio(3);
      // fused LeakyReLU-backward mask: request the sign words of the tile I am about to store BEFORE the halo DMA
      // is queued, so that they have landed by the time the epilogue needs them (one VGPR per M tile)
      uint32_t mb[MTW];
      const bool use_mask = a.mask_bits != nullptr && p >= 1 && !(a.dbg_flags & 2);
      if (use_mask) {
        const sg_tile_origin om = sg_tile_of(g, (uint32_t)(first + (p - 1) * per_x));
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int tc = tcoord[mt];
          const int n = om.n0 + (tc >> 24), d = om.d0 + ((tc >> 16) & 255), h = om.h0 + ((tc >> 8) & 255),
                    w = om.w0 + (tc & 255);
          const bool ok = tc >= 0 && n < g.N && d < g.D && h < g.H && w < g.W;
          mb[mt] = ok ? a.mask_bits[((((int64_t)n * g.D + d) * g.H + h) * g.W + w) * a.ntile + nt0] : 0u;
        }
      }
      if (p + 1 < K && !((a.dbg_flags & 1) && p >= 2)) stage_tile(sg_tile_of(g, (uint32_t)(first + (p + 1) * per_x)));   // into my (now idle) buffer
      stamp();
      if (p >= 1 && !(a.dbg_flags & 2)) {
        const sg_tile_origin o = sg_tile_of(g, (uint32_t)(first + (p - 1) * per_x));
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int tc = tcoord[mt];
          const int n = o.n0 + (tc >> 24), d = o.d0 + ((tc >> 16) & 255), h = o.h0 + ((tc >> 8) & 255),
                    w = o.w0 + (tc & 255);
          const bool ok = tc >= 0 && n < g.N && d < g.D && h < g.H && w < g.W;
          if (a.act) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = fmaxf(acc[mt][i], acc[mt][i] * a.slope);
          }
          if (a.pixel_norm) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) ss += acc[mt][i] * acc[mt][i];
            ss += __shfl_xor(ss, 32);
            const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] *= sc;
            if (a.pn_scale != nullptr && hh == 0 && ok)
              a.pn_scale[(((int64_t)n * g.D + d) * g.H + h) * g.W + w] = sc;
          }
          if (a.sign_out != nullptr) {
            const uint32_t sw = sg_sign_word(acc[mt], hh);
            if (hh == 0 && ok) a.sign_out[((((int64_t)n * g.D + d) * g.H + h) * g.W + w) * a.ntile + nt0] = sw;
          }
          if (use_mask) sg_apply_sign_word(acc[mt], mb[mt], hh, a.mask_slope);
          if (ok) {
            T* yrow = y + ((((int64_t)n * g.D + d) * g.H + h) * g.W + w) * (int64_t)a.cout;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const int co = nt0 * 32 + 8 * qd + 4 * hh;
              if (a.vec_out && co + 4 <= a.cout) {
                T tmp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) tmp[e] = sg_traits<T>::from_f(acc[mt][qd * 4 + e]);
                if (sizeof(T) == 2) *reinterpret_cast<u32x2*>(yrow + co) = *reinterpret_cast<u32x2*>(tmp);
                else { const u32x4 t16 = *reinterpret_cast<u32x4*>(tmp); *reinterpret_cast<u32x4*>(yrow + co) = t16; SG_STORE16_GUARD(t16); }
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (co + e < a.cout) yrow[co + e] = sg_traits<T>::from_f(acc[mt][qd * 4 + e]);
              }
            }
          }
        }
      }
    }
    if (a.dbg_flags & 4) __builtin_amdgcn_s_setprio(0);
    stamp();
    __syncthreads();
  }
  // the group that computed the last tile (K-1) still holds it: phase K stored it iff ((K & 1) != grp), i.e. the
  // group (K-1)&1 ran the else-branch in phase K.  Nothing is left over.
}

template <typename T, int MTW, int GC, int KD, int KH, int KW>
static int launch_fwd3r(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  constexpr int BM = MTW * 128;
  a.g = sg_make_geom(s, BM, /*prefer_w32=*/true);
  const sg_tile_geom& g = a.g;
  if (g.TW > 255 || g.TH > 255 || g.TD > 255 || g.TN > 127) return SG_OK;
  const int64_t ntiles = (int64_t)g.nTn * g.nTd * g.nTh * g.nTw;
  if (ntiles >= (1 << 24) || ntiles < 512) return SG_OK;   // persistence only pays with many tiles per CU
  const int hv = g.TN * g.HD * g.HH * g.HW;
  a.G = GC;
  a.rs = GC * 32;
  a.xbytes = ((hv * a.rs) + 1023) & ~1023;
  a.wbytes = a.taps * GC * 1024;
  const size_t lds = (size_t)a.wbytes + 2ull * a.xbytes + 128;
  if (lds > 160 * 1024) return SG_OK;
  a.ntiles = (int)ntiles;
  a.vec_in = 1;
  a.vec_out = (s->cout % 4 == 0) ? 1 : 0;
  if (sg_cdiv(hv * GC * 2, 64) > 56) return SG_OK;   // more halo DMA pieces than the kernel's per-lane table holds
  if (MTW == 2 && GC == 2 && !(g.TW == 32 && (g.TH & 1) == 0)) return SG_OK;   // shared address table (sg_xa_index)
  // buffer addressing of the halo: per-lane offsets are relative to the tile's first sample
  a.lean = (!s->upsample_in && (int64_t)g.TN * s->d * s->h * s->w * (int64_t)s->cin * (int64_t)sizeof(T) < (1ll << 31) &&
            !sg_cfg().fwd3_no_lean) ? 1 : 0;
  auto kern = conv_fwd3r_kernel<T, MTW, GC, KD, KH, KW>;
  SG_ALLOW_160K_LDS(kern);
  int gx = 256 / a.ntile;            // one block per CU in total
  gx = (gx / 8) * 8;
  if (gx < 8) gx = 8;
  if (sg_cfg().fwd3_gx > 0) gx = sg_cfg().fwd3_gx;
  SG_KNAME("conv_fwd3r<%s,%d,%d,%d,%d,%d>", sg_tname<T>(), MTW, GC, KD, KH, KW);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)a.ntile), dim3(512), lds, st, a);
  SG_LAUNCH_CHECK();
  *used = true;
  return SG_OK;
}


// ------------------------------------------------------------------------------------------------------
// forward, v3s: v3r with a SLIDING halo (3x3x3 taps, Cin <= one channel group, tile fixed at 2 x 4 x 32 voxels).
// In-kernel stamps of v3r show its off-phase (13 LDS-DMA pieces per wave ~3.3-5k cycles + epilogue 2.5k) longer
// than the 4.5k-cycle MFMA phase: the MFMA group idles 2-4k cycles per tile.  Here a wave group walks a column of
// tiles along D and keeps the halo as a ring of four D planes (slot = plane & 3): a step fetches only the TWO new
// planes (6.5 pieces per wave).  Wave w owns H row w of the tile and its two M tiles are the tile's two D planes,
// so the ring slot of every fragment read, (kd + mt + ROT) & 3 with ROT = 3 (even step) / 1 (odd step), is a
// compile-time ds_read offset and a lane needs 9*GC halo addresses instead of 27*2.
// ------------------------------------------------------------------------------------------------------
template <typename T, int GC, int ROT, int RING>
struct sg_unrolled_ks {
  static constexpr int NS = 27 * GC, PF = RING - 1, RPS = 3;
  static constexpr int PB = (GC == 1 ? 224 : 208) * GC * 32;   // bytes per plane slot: 204 live rows padded to whole 1-KiB pieces
  template <int ST>
  static __device__ __forceinline__ void load(u32x4 (&wfr)[RING], u32x4 (&xfr)[RING][2], const int (&xa)[9][GC],
                                              int wl_off) {
    constexpr int SL = ST % RING, tap = ST / GC, gi = ST % GC, kd = tap / 9, khw = tap % 9;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wfr[SL]) : "v"(wl_off), "n"(ST << 10));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xfr[SL][0]) : "v"(xa[khw][gi]), "n"(((kd + 0 + ROT) & 3) * PB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xfr[SL][1]) : "v"(xa[khw][gi]), "n"(((kd + 1 + ROT) & 3) * PB));
  }
  template <int ST>
  static __device__ __forceinline__ void step(f32x16 (&acc)[2], u32x4 (&wfr)[RING], u32x4 (&xfr)[RING][2],
                                              const int (&xa)[9][GC], int wl_off) {
    if constexpr (ST < NS) {
      if constexpr (ST + PF < NS) load<ST + PF>(wfr, xfr, xa, wl_off);
      constexpr int younger = (NS - 1 - ST < PF ? NS - 1 - ST : PF) * RPS;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = sg_mfma_chunk<T>(wfr[ST % RING], xfr[ST % RING][0], acc[0]);
      acc[1] = sg_mfma_chunk<T>(wfr[ST % RING], xfr[ST % RING][1], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      step<ST + 1>(acc, wfr, xfr, xa, wl_off);
    }
  }
  template <int ST>
  static __device__ __forceinline__ void prologue(u32x4 (&wfr)[RING], u32x4 (&xfr)[RING][2], const int (&xa)[9][GC],
                                                  int wl_off) {
    if constexpr (ST < PF && ST < NS) {
      load<ST>(wfr, xfr, xa, wl_off);
      prologue<ST + 1>(wfr, xfr, xa, wl_off);
    }
  }
  static __device__ __forceinline__ void run(f32x16 (&acc)[2], const int (&xa)[9][GC], int wl_off) {
    u32x4 wfr[RING], xfr[RING][2];
    SG_KLOOP_BEGIN();
    prologue<0>(wfr, xfr, xa, wl_off);
    step<0>(acc, wfr, xfr, xa, wl_off);
    SG_KLOOP_END();
  }
};

// Same K loop with each halo fragment read ONCE for both M tiles: the tile's two M tiles are consecutive D planes,
// so halo plane p (0..3) is tap kd = p of M tile 0 and tap kd = p - 1 of M tile 1.  Step s = (kh, kw, chunk) * 4 + p:
// reads X[p] and, for p < 3, W[kd = p]; MFMAs acc0 += W[p] X[p] (p <= 2) and acc1 += W[p-1] X[p] (p >= 1).
// 7 fragment reads per 6 MFMAs instead of 9: the MFMA phases run at the board's power cap, and LDS reads are a
// large part of what they burn.
template <typename T, int GC, int ROT, int RING>
struct sg_unrolled_ks2 {
  static constexpr int NG = 9 * GC, NS = NG * 4, PF = RING - 2;
  static constexpr int PB = (GC == 1 ? 224 : 208) * GC * 32;
  static constexpr int nloads(int st) { return st >= NS ? 0 : ((st & 3) < 3 ? 2 : 1); }
  static constexpr int younger(int st) {   // reads issued after step st's own loads at the time step st computes
    int n = 0;
    for (int t = st + 1; t <= st + PF; ++t) n += nloads(t);
    return n;
  }
  template <int ST>
  static __device__ __forceinline__ void load(u32x4 (&wfr)[RING], u32x4 (&xfr)[RING], const int (&xa)[9][GC], int wl_off) {
    constexpr int SL = ST % RING, grp_ = ST >> 2, pl = ST & 3, khw = grp_ / GC, gi = grp_ % GC;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xfr[SL]) : "v"(xa[khw][gi]), "n"(((pl + ROT) & 3) * PB));
    if constexpr (pl < 3) {   // weight fragment of tap (kd = pl, kh, kw), chunk gi: image order [tap][gi]
      constexpr int frag = (pl * 9 + khw) * GC + gi;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wfr[SL]) : "v"(wl_off), "n"(frag << 10));
    }
  }
  template <int ST>
  static __device__ __forceinline__ void step(f32x16 (&acc)[2], u32x4 (&wfr)[RING], u32x4 (&xfr)[RING],
                                              const int (&xa)[9][GC], int wl_off) {
    if constexpr (ST < NS) {
      if constexpr (ST + PF < NS) load<ST + PF>(wfr, xfr, xa, wl_off);
      SG_WAIT_LGKM(younger(ST));
      __builtin_amdgcn_sched_barrier(0);
      constexpr int pl = ST & 3;
      if constexpr (pl <= 2) sg_mfma_bf16_acc(acc[0], wfr[ST % RING], xfr[ST % RING]);
      if constexpr (pl >= 1) sg_mfma_bf16_acc(acc[1], wfr[(ST - 1) % RING], xfr[ST % RING]);
      __builtin_amdgcn_sched_barrier(0);
      step<ST + 1>(acc, wfr, xfr, xa, wl_off);
    }
  }
  template <int ST>
  static __device__ __forceinline__ void prologue(u32x4 (&wfr)[RING], u32x4 (&xfr)[RING], const int (&xa)[9][GC], int wl_off) {
    if constexpr (ST < PF && ST < NS) {
      load<ST>(wfr, xfr, xa, wl_off);
      prologue<ST + 1>(wfr, xfr, xa, wl_off);
    }
  }
  static __device__ __forceinline__ void run(f32x16 (&acc)[2], const int (&xa)[9][GC], int wl_off) {
    static_assert(sizeof(T) == 2, "bf16 only");
    u32x4 wfr[RING], xfr[RING];
    SG_KLOOP_BEGIN();
    prologue<0>(wfr, xfr, xa, wl_off);
    step<0>(acc, wfr, xfr, xa, wl_off);
    sg_mfma_drain(acc);
    SG_KLOOP_END();
  }
};

// Epilogue features of the sliding-halo kernel, compile-time: the off-phase of the generic (run-time flags) version
// spent most of its ~4.3k cycles on scalar bookkeeping -- 200 spilled SGPRs (v_readlane), kernel arguments re-read
// from memory behind s_waitcnt lgkmcnt(0), branches around features the launch did not use.
// (SG_EP_*: conv_args.h)
#ifndef SG_V3S_RING
#define SG_V3S_RING 6   // fragment ring of the MFMA phase: reads run RING - 2 steps (of 1-2 MFMAs) ahead of their use
#endif

template <int GC, int KS, int EPI, bool UPS, bool INM = false>   // KS: 0 whole layer; 1 / 2: first / second pass of a layer split
                                              // over its input channels; UPS: x is the half-resolution tensor, gathered
                                              // nearest-x2; INM: ... times in_gain * where(sign bit of the fine voxel, slope, 1)
__global__ __launch_bounds__(512) void conv_fwd3s_kernel(ConvFwdArgs a) {
  static_assert(!INM || (UPS && GC == 2), "the input mask rides on the fused gather");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int TAPS = 27;
  constexpr int ES = 2, EPP = 8;
  constexpr int S = GC * 2;                          // 16-byte slots per halo row
  constexpr int sshift = GC == 1 ? 1 : 2;
  constexpr int rshift = GC == 1 ? 3 : 2;            // rows per 256-byte bank row
  constexpr int rb = GC * 32;
  constexpr int PR = GC == 1 ? 224 : 208, PB = PR * rb, PPIECES = PB / 1024;   // 13 (GC=2) / 7 (GC=1) pieces per plane
  static_assert(PB % 1024 == 0 && PB == sg_unrolled_ks2<T, GC, 1, 6>::PB, "plane slot must be whole pieces");
  constexpr int RINGB = 4 * PB;
  constexpr uint32_t DEAD = 0x80000000u;             // byte offset beyond every buffer: loads return 0, stores drop
  constexpr bool SIGN = (EPI & SG_EP_SIGN) != 0, MASK = (EPI & SG_EP_MASK) != 0, PN = (EPI & SG_EP_PN) != 0,
                 POOL = (EPI & SG_EP_POOL) != 0, PNB = (EPI & SG_EP_PNB) != 0;
  static_assert(!(PN && POOL) && !(KS == 1 && EPI != 0) && !(PNB && (PN || POOL || SIGN || KS != 0)), "unsupported epilogue combination");
  const sg_tile_geom& g = a.g;                       // TN=1, TD=2, TH=4, TW=32, HD=4, HH=6, HW=34 (host-checked)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int r = lane & 31, hh = lane >> 5;
  // LDS map: [ring of group 0][ring of group 1][resident weights: [tap][gi] 1-KiB fragments][bias: 32 floats]
  char* xmine = smem + grp * RINGB;
  char* wlds = smem + 2 * RINGB;
  const char* wp = reinterpret_cast<const char*>(a.wp);
  const int nt0 = blockIdx.y;
  const int H = g.H, W = g.W, D = g.D, nTd = g.nTd, cout = a.cout, ntile = a.ntile;
  // Everything global goes through buffer resources: a scalar base (rebased per batch sample, so only ONE SAMPLE of a
  // tensor has to stay below 2 GiB), a scalar per-tile / per-plane offset and a 32-bit per-lane offset computed once;
  // dead lanes carry DEAD.  A tile column (fixed sample, H, W origin; 2 planes further along D per step) keeps its
  // resources and per-lane offsets: per step only the scalar offsets advance.
  const int64_t svox = (int64_t)D * H * W;           // voxels per sample
  auto rsrc_of = [&](const void* base, int64_t sample_bytes, int n0) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + n0 * sample_bytes, 0,
                                             (int)sample_bytes, 0x00020000);
  };
  const int64_t xsb = (UPS ? svox >> 3 : svox) * a.xcs * ES, ysb = svox * cout * (KS == 1 ? 4 : ES), wsb = svox * ntile * 4, psb = svox * 4;

  // column schedule: a block walks PAIRS of H-adjacent tile columns along D, wave group g taking the column with
  // tile row 2*k + g, one phase apart: the two halo rows the pair shares are fetched twice within ~2 us on the same
  // CU and the second fetch is an L2 hit (dealt independently, H-neighbours drift apart and every overlap row came
  // from HBM again: fetch 1.55x the input, now 1.02x, PMC).  XCD group xg owns a contiguous chunk of the pair list.
  const int nTh2 = (g.nTh + 1) >> 1;
  const int npair = g.nTn * nTh2 * g.nTw;
  const int xg = blockIdx.x & 7, bslot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (npair + 7) >> 3;
  const int c_begin = xg * cpx, c_end = min(npair, c_begin + cpx);
  const int cfirst = c_begin + bslot;
  const int ncols_blk = cfirst < c_end ? (c_end - cfirst + per_x - 1) / per_x : 0;
  const int items_mine = ncols_blk * nTd;

  // fragment addresses inside a plane slot: voxel (h = wave + kh, w = r + kw), channel chunk gi, half hh
  int xa[9][GC];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int row = (wave + kh) * 34 + r + kw;
      const int f = (row >> rshift) & (S - 1);
#pragma unroll
      for (int gi = 0; gi < GC; ++gi)
        xa[kh * 3 + kw][gi] = grp * RINGB + row * rb + (((2 * gi + hh) ^ f) << 4);
    }
  // plane-local staging table: this lane's 16-byte pieces of 1-KiB blocks wave, wave+4, ... of a plane.  The halo
  // goes global -> registers -> LDS.
  constexpr int MAXP = (PPIECES + 3) / 4;
  uint32_t relb[MAXP];   // byte offset of my piece relative to the plane's first halo voxel (h0-1, w0-1); dead: huge
  uint32_t relm[INM ? MAXP : 1];   // INM: byte offset of my piece's 8 sign bits relative to that voxel's first sign word
  int crdp[MAXP];        // packed (hw, hh) for the boundary test; dead pieces fail every range
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> sshift, p = it & (S - 1);
    const int hh_ = row / 34, hw = row - hh_ * 34;
    const int c = p * EPP;
    const bool live = row < 204 && c < a.cin && (wave + 4 * k) < PPIECES;
    // UPS: tile origins are even, so a halo voxel's halved coordinate is a per-lane constant relative to the tile's
    // half-resolution origin: ((h0 - 1 + hh) >> 1) = h0 / 2 + ((hh - 1) >> 1), likewise along W
    const int rel = UPS ? ((((hh_ - 1) >> 1) * (W >> 1) + ((hw - 1) >> 1)) * a.xcs + a.xco + c) * ES : ((hh_ * W + hw) * a.xcs + a.xco + c) * ES;
    relb[k] = live ? (uint32_t)rel : 0xC0000000u;   // dead: stays >= DEAD after + tile offset
    crdp[k] = live ? (hw | (hh_ << 8)) : 0x7F7F;
    if constexpr (INM)   // the byte of word (xco + c) / 32 that holds channels c .. c + 7 of FINE voxel (hh_, hw)
      relm[k] = live ? (uint32_t)(((hh_ * W + hw) * a.in_mask_nw + ((a.xco + c) >> 5)) * 4 + (((a.xco + c) & 31) >> 3)) : 0xC0000000u;
  }
  // LDS position of my piece inside its 1-KiB block: slot p of row lands at p ^ f(row), and f(row) only depends on
  // the lane because a block is a whole number of 256-byte bank rows
  const int wofs = (lane >> sshift) * rb + (((lane & (S - 1)) ^ ((lane >> (sshift + rshift)) & (S - 1))) << 4);
  // my two output voxels (d = mt, h = wave, w = r of the tile): byte / word offsets relative to the tile origin
  uint32_t yvo[2], svo[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int v = (mt * H + wave) * W + r;
    yvo[mt] = (uint32_t)((v * cout + nt0 * 32) * (KS == 1 ? 4 : ES));
    svo[mt] = (uint32_t)(v * ntile + nt0) * 4u;
  }
  const uint32_t plane_bytes = (uint32_t)((UPS ? (H >> 1) * (W >> 1) : H * W) * a.xcs * ES);
  const int plane_vox = H * W;

  // A cursor walks my tiles in order: column cj of my list, step di along D.  Three run one behind the other: the
  // tile whose halo planes were requested last (P), the tile computed next (same as P until P advances at the end of
  // an off-phase) and the tile whose results are stored (E).
  struct Cur { int cj, di, n0, h0, w0; };
  auto enter_column = [&](Cur& c) {
    const int pr = cfirst + c.cj * per_x;
    const int c1 = (int)sg_div((uint32_t)pr, g.fnTw);
    c.w0 = (pr - c1 * g.nTw) * 32;
    const int c2 = c1 / nTh2;
    c.h0 = (2 * (c1 - c2 * nTh2) + grp) * 4;         // may lie beyond H for the last odd row: a dead column
    c.n0 = c2;
  };
  // ---- halo side (cursor P): per-column resource and per-lane offsets
  Cur P{0, 0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rxP, rmP;
  uint32_t vk[MAXP], vkm[INM ? MAXP : 1];
  const int64_t msb = svox * a.in_mask_nw * 4;     // (INM) sign words of one sample of the fine input
  auto enter_column_P = [&]() {
    enter_column(P);
    rxP = rsrc_of(a.x, xsb, P.n0);
    if constexpr (INM) rmP = rsrc_of(a.in_mask, msb, P.n0);
    const int tile_off = UPS ? ((P.h0 >> 1) * (W >> 1) + (P.w0 >> 1)) * a.xcs * ES
                             : ((P.h0 - 1) * W + (P.w0 - 1)) * a.xcs * ES;   // may be negative: only dead lanes go below 0
    const int lo_w = max(0, 1 - P.w0), hi_w = min(34, W + 1 - P.w0) - 1;
    const int lo_h = max(0, 1 - P.h0), hi_h = min(6, H + 1 - P.h0) - 1;   // hi_h < 0 for a dead column
    const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8));
    const uint32_t hi = (uint32_t)(hi_w | ((hi_h & 0x7F) << 8)) | 0x8080u;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const uint32_t c_ = (uint32_t)crdp[k];
      const uint32_t t1 = (c_ | 0x8080u) - lo, t2 = hi - c_;
      const bool in = (t1 & t2 & 0x8080u) == 0x8080u && hi_h >= 0;
      vk[k] = in ? relb[k] + (uint32_t)tile_off : DEAD;
      if constexpr (INM) vkm[k] = in ? relm[k] + (uint32_t)(((P.h0 - 1) * W + (P.w0 - 1)) * a.in_mask_nw * 4) : DEAD;
    }
  };
  u32x4 stg[2][MAXP];
  uint32_t mstg[INM ? 2 : 1][INM ? MAXP : 1];      // (INM) the 8 sign bits of each piece in flight
  auto load_planes = [&](int d0, int hd0) __attribute__((always_inline)) {   // planes d0 - 1 + hd0 + {0, 1} of P's column
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gp = d0 - 1 + hd0 + j;               // global D plane
      const bool plane_ok = gp >= 0 && gp < D;
      const uint32_t soff = plane_ok ? (uint32_t)(UPS ? gp >> 1 : gp) * plane_bytes : 0u;
#pragma unroll
      for (int k = 0; k < MAXP; ++k) stg[j][k] = __builtin_amdgcn_raw_buffer_load_b128(rxP, plane_ok ? vk[k] : DEAD, soff, 0);
      if constexpr (INM) {
        const uint32_t moff = plane_ok ? (uint32_t)gp * (uint32_t)(plane_vox * a.in_mask_nw * 4) : 0u;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) mstg[j][k] = __builtin_amdgcn_raw_buffer_load_b8(rmP, plane_ok ? vkm[k] : DEAD, moff, 0);
      }
    }
  };
  const float gain_in = a.in_gain, slope_in = a.in_mask_slope;   // (INM) sg_mask_piece_bf16: the arithmetic of sg_upscale2x_masked
  // Ring slot of a halo plane: by its D index, shifted by two slots for the columns that START at an odd running tile
  // index (qP - P.di odd).  The tiles of a group then alternate strictly between the two ring rotations whatever the
  // number of tiles per column -- all four planes are rewritten at a column start anyway -- and the two rotations
  // of the unrolled MFMA code follow each other in straight-line code instead of behind a branch on the tile's parity
  // (at whose join the compiler shuffled all 32 accumulator registers, every phase).
  int qP = 0;                                        // index of P's tile in my list
  auto store_planes = [&](int d0, int hd0) __attribute__((always_inline)) {   // (left to itself the compiler calls the INM variant: the cursor then lives in scratch)
    const int rot2 = 2 * ((qP - P.di) & 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gp = d0 - 1 + hd0 + j;
      char* dst = xmine + ((gp + rot2 + 8) & 3) * PB + wofs;
#pragma unroll
      for (int k = 0; k < MAXP; ++k)
        if (wave + 4 * k < PPIECES) {
          if constexpr (INM) *reinterpret_cast<u32x4*>(dst + (wave + 4 * k) * 1024) = sg_mask_piece_bf16(stg[j][k], mstg[j][k], gain_in, slope_in);
          else *reinterpret_cast<u32x4*>(dst + (wave + 4 * k) * 1024) = stg[j][k];
        }
    }
  };
  // ---- output side (cursor E): per-column resources, validity and voxel origin
  Cur E{0, 0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t ryE, rsE, rmE, rpE, rbE, rqE;
  int colvoxE = 0;
  bool row_okE = false;
  auto enter_column_E = [&]() {
    enter_column(E);
    row_okE = E.h0 + wave < H;
    colvoxE = E.h0 * W + E.w0;
    ryE = rsrc_of(a.y, POOL ? ysb / 4 : ysb, E.n0);
    if (SIGN) rsE = rsrc_of(a.sign_out, wsb, E.n0);
    if (MASK) rmE = rsrc_of(a.mask_bits, wsb, E.n0);
    if (PN) rpE = rsrc_of(a.pn_scale, psb, E.n0);
    if (PNB) {
      rbE = rsrc_of(a.pnb_y, ysb, E.n0);
      rqE = rsrc_of(a.pnb_scale, psb, E.n0);
    }
  };
  // LeakyReLU sign words of E's tile (masked epilogue): requested at the END of an off-phase, like the halo planes, and
  // used in the next one.  Requested at the top of the off-phase that applies them, their latency (2-3k cycles under
  // load) sat in front of the epilogue: rocprofv3 SQ_VALU_MFMA_BUSY 0.52 for the masked variant against 0.78.
  int qE = 0;                                        // index of E's tile in my list
  uint32_t mbn[2] = {0u, 0u};
  auto request_mask = [&]() {
    if constexpr (MASK) {
      const int d0 = 2 * E.di;
      const bool live = qE < items_mine && row_okE;
      const uint32_t tv = (uint32_t)(d0 * plane_vox + colvoxE);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        mbn[mt] = __builtin_amdgcn_raw_buffer_load_b32(rmE, (live && d0 + mt < D) ? svo[mt] : DEAD, tv * (uint32_t)(ntile * 4), 0);
    }
  };

  // resident weights (all 8 waves) and bias
  {
    constexpr int nfrag = TAPS * GC;
    for (int f = wave8; f < nfrag; f += 8) {
      const int tap = f / GC, gi = f - tap * GC;
      const char* src = gi < a.nchunk ? wp + ((((int64_t)gi * TAPS + tap) * ntile + nt0) << 10)
                                      : reinterpret_cast<const char*>(sg_zero_page);
      sg_glds16(src + (gi < a.nchunk ? lane * 16 : 0), wlds + ((size_t)f << 10));
    }
  }
  float* bias_lds = reinterpret_cast<float*>(wlds + TAPS * GC * 1024);
  if (tid < 32) {
    const int co = nt0 * 32 + tid;
    bias_lds[tid] = (a.bias != nullptr && co < cout) ? a.bias[co] : 0.f;
  }
  f32x16 acc[2];
  // The upper two halo planes of a group's NEXT tile are requested at the END of an off-phase and written to the ring
  // at the start of the following one: they are in flight during the group's whole MFMA phase.  (Requested at the
  // start of the off-phase that needs them, their ~3 us under load made the off-phase longer than the MFMA phase:
  // in-kernel stamps, 5.1k against 4.6k cycles.)  `stg` is therefore live across the MFMA phase.
  const bool no_stage = (a.dbg_flags & 1) != 0, no_epi = (a.dbg_flags & 2) != 0;
  if ((a.dbg_flags & 2048) && grp == 1) __builtin_amdgcn_s_setprio(1);      // diagnostic: static priority for the younger wave group
  if (items_mine > 0) {
    enter_column_P();
    enter_column_E();
    if (grp == 0) {   // the very first tile of group 0: all four planes, latency exposed once
      load_planes(0, 0);
      store_planes(0, 0);
      load_planes(0, 2);
      store_planes(0, 2);
      if (items_mine > 1) {
        qP = 1;
        P.di = 1;     // nTd >= 2
        load_planes(2, 2);
      }
    } else {
      load_planes(0, 2);
    }
  }
  __syncthreads();

  const int wl_off = (int)(wlds - smem) + lane * 16;
  const float inv_c = 1.f / (float)cout;
  const float slope = a.act ? a.slope : 1.f;         // max(x, 1 * x) = x: no branch for "no activation"

  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && lane == 0 && wave == 0 && dbgi < 120)
      a.dbg[grp * 128 + dbgi] = __builtin_amdgcn_s_memtime();
    ++dbgi;
  };
  auto init_acc = [&]() {   // at the end of an off-phase (the accumulators are free once the epilogue has run)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][i] = bias_lds[(i & 3) + 8 * (i >> 2) + 4 * hh];   // bias rides in C
    // two separate register tuples from here on (equal values: left to itself the compiler keeps ONE copy, lets the
    // first MFMA of each chain write elsewhere and shuffles 32 registers per phase to get back)
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
  };
  auto mfma_even = [&]() { sg_unrolled_ks2<T, GC, 3, SG_V3S_RING>::run(acc, xa, wl_off); };   // tiles 0, 2, ... of my list
  auto mfma_odd = [&]() { sg_unrolled_ks2<T, GC, 1, SG_V3S_RING>::run(acc, xa, wl_off); };

  // ---- off-phase: P's planes (requested one phase ago) into the ring, epilogue of E's tile (if `closes`), request
  // the planes of the tile after P
  auto off_phase = [&](bool closes, bool stage) {
    const int d0E = 2 * E.di;
    const uint32_t tile_vox = (uint32_t)(d0E * plane_vox + colvoxE);   // within sample E.n0
    const bool okE[2] = {closes && row_okE, closes && row_okE && d0E + 1 < D};
    // sign words of the tile about to be stored (requested one phase ago).  Issued and consumed UNCONDITIONALLY (DEAD
    // offset: no memory access): with the request under one `if` and the use under another the compiler sees a path on
    // which the load is never waited for and puts s_waitcnt vmcnt(0) -- which also waits for the halo planes in flight --
    // in front of the next MFMA that reuses the register.
    uint32_t mb[2] = {mbn[0], mbn[1]};
    if (MASK) asm volatile("" : "+v"(mb[0]), "+v"(mb[1]));
    // K-split second pass: the first pass's f32 partial sums of the tile about to be stored (held across the MFMA
    // phase they would not fit beside the accumulators, the planes in flight and the fragment ring)
    // pixel-norm backward epilogue: the stage's output at my two voxels (2 x 16 bytes each: the row pieces the plain
    // epilogue would store) and its rsqrt factor, requested here and used after the planes have been written
    u32x4 yraw[PNB ? 2 : 1][PNB ? 2 : 1];
    float pscale[2] = {0.f, 0.f};
    if constexpr (PNB) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          yraw[mt][j] = __builtin_amdgcn_raw_buffer_load_b128(rbE, okE[mt] ? yvo[mt] + (uint32_t)((16 * j + 8 * hh) * 2) : DEAD,
                                                              tile_vox * (uint32_t)(cout * ES), 0);
        pscale[mt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rqE, okE[mt] ? svo[mt] : DEAD, tile_vox * 4u, 0));
      }
    }
    f32x4 part[KS == 2 ? 2 : 1][KS == 2 ? 4 : 1];
    if constexpr (KS == 2) {
      const __amdgpu_buffer_rsrc_t ra = rsrc_of(a.addend, svox * cout * 4, E.n0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          part[mt][qd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              ra, okE[mt] ? 2 * yvo[mt] + (uint32_t)((8 * qd + 4 * hh) * 4) : DEAD, tile_vox * (uint32_t)(cout * 4), 0));
    }
    if (stage && !no_stage) {
      const int d0P = 2 * P.di;
      store_planes(d0P, 2);
      if (P.di == 0) {   // bottom of a column: the two lower planes too, straight through
        load_planes(0, 0);
        store_planes(0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    if (!no_epi && closes) {
      if (KS == 1) {     // K-split first pass: the accumulators as they stand, 4 x 16 bytes per lane
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            f32x4 v4 = {acc[mt][4 * qd], acc[mt][4 * qd + 1], acc[mt][4 * qd + 2], acc[mt][4 * qd + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), ryE,
                                                   okE[mt] ? yvo[mt] + (uint32_t)((8 * qd + 4 * hh) * 4) : DEAD,
                                                   tile_vox * (uint32_t)(cout * 4), 0);
            SG_STORE16_GUARD(v4);
          }
      } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if constexpr (KS == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] += part[mt][i >> 2][i & 3];
          }
          if (slope != 1.f) {   // uniform
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = sg_lrelu(acc[mt][i], slope);
          }
          if (PN) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) ss += acc[mt][i] * acc[mt][i];
            const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
            ss = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);   // own half + partner lane ^ 32's
            const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] *= sc;
            if (a.pn_scale != nullptr)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc), rpE,
                                                    (okE[mt] && hh == 0) ? (svo[mt] - nt0 * 4u) / (uint32_t)ntile : DEAD,
                                                    tile_vox * 4u, 0);
          }
          if (SIGN) {
            uint32_t b = 0u;
#pragma unroll
            for (int i = 0; i < 16; ++i) b |= (__float_as_uint(acc[mt][i]) >> 31) << ((i & 3) + 8 * (i >> 2));
            b <<= 4 * hh;
            const auto sw2 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
            __builtin_amdgcn_raw_buffer_store_b32(sw2[0] | sw2[1], rsE, (okE[mt] && hh == 0) ? svo[mt] : DEAD,
                                                  tile_vox * (uint32_t)(ntile * 4), 0);
          }
          if constexpr (PNB) {
            // d/dx of y = x * s, s = rsqrt(mean_c(x^2) + eps):  s * (g - y * mean_c(g * y)); the accumulator holds g.
            // y arrives as the 16-byte row pieces of the store layout: undo the half-wave exchange of the store path
            float yv[16];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const auto t0 = __builtin_amdgcn_permlane32_swap(yraw[mt][j][0], yraw[mt][j][2], false, false);
              const auto t1 = __builtin_amdgcn_permlane32_swap(yraw[mt][j][1], yraw[mt][j][3], false, false);
              const uint32_t pk[4] = {t0[0], t1[0], t0[1], t1[1]};     // channel pairs of acc[8j + 0..7]
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                yv[8 * j + 2 * k] = __uint_as_float(pk[k] << 16);
                yv[8 * j + 2 * k + 1] = __uint_as_float(pk[k] & 0xFFFF0000u);
              }
            }
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) dot = fmaf(acc[mt][i], yv[i], dot);
            const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(dot), __float_as_uint(dot), false, false);
            const float mean = (__uint_as_float(sw2[0]) + __uint_as_float(sw2[1])) * inv_c;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = pscale[mt] * fmaf(-yv[i], mean, acc[mt][i]);
          }
          if (MASK && !(a.dbg_flags & 256)) sg_apply_sign_word(acc[mt], mb[mt], hh, a.mask_slope);
          if (!POOL) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {   // 16 contiguous bytes per lane (see sg_store_tile_row_bf16)
              const uint32_t a0 = sg_pack_bf16(acc[mt][8 * j + 0], acc[mt][8 * j + 1]), a1 = sg_pack_bf16(acc[mt][8 * j + 2], acc[mt][8 * j + 3]);
              const uint32_t b0 = sg_pack_bf16(acc[mt][8 * j + 4], acc[mt][8 * j + 5]), b1 = sg_pack_bf16(acc[mt][8 * j + 6], acc[mt][8 * j + 7]);
              const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
              u32x4 out;
              out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
              __builtin_amdgcn_raw_buffer_store_b128(out, ryE, okE[mt] ? yvo[mt] + (uint32_t)((16 * j + 8 * hh) * 2) : DEAD,
                                                     tile_vox * (uint32_t)(cout * ES), 0);
              SG_STORE16_GUARD(out);
            }
          }
        }
        if (POOL) {
          // fused downscale3d, first stage (pgan/discriminator.py:44 after conv_2 + bias + LeakyReLU): the tile's two D
          // planes are this wave's two M tiles (a lane-local add) and W neighbours are adjacent lanes (one cross-lane
          // exchange), so the mean over the 2 x 1 x 2 block costs 32 VALU ops and the full-resolution activation --
          // the largest tensor of the network, needed by nobody else: the backward only wants its sign words -- is never
          // written.  Output [n, D/2, H, W/2, cout]; the H pairs (two different waves) are pooled by sg_downscale_sum(1,2,1).
          const uint32_t psoff = (uint32_t)(((E.di * H + E.h0) * (W >> 1) + (E.w0 >> 1)) * cout * ES);
          const uint32_t pvo = ((r & 1) || !okE[0]) ? DEAD : (uint32_t)(((wave * (W >> 1) + (r >> 1)) * cout + nt0 * 32) * ES);
          float sp[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float t = acc[0][i] + acc[1][i];
            // lane r <-> r ^ 1 (the W neighbour) by DPP quad_perm [1,0,3,2]: one VALU op, no LDS round trip
            const float u = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0xB1, 0xF, 0xF, true));
            sp[i] = (t + u) * 0.25f;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t a0 = sg_pack_bf16(sp[8 * j + 0], sp[8 * j + 1]), a1 = sg_pack_bf16(sp[8 * j + 2], sp[8 * j + 3]);
            const uint32_t b0 = sg_pack_bf16(sp[8 * j + 4], sp[8 * j + 5]), b1 = sg_pack_bf16(sp[8 * j + 6], sp[8 * j + 7]);
            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            u32x4 out;
            out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
            __builtin_amdgcn_raw_buffer_store_b128(out, ryE, pvo == DEAD ? DEAD : pvo + (uint32_t)((16 * j + 8 * hh) * 2), psoff, 0);
            SG_STORE16_GUARD(out);
          }
        }
      }
    }
    if (a.dbg_flags & 128) stamp();   // fine-grained stamps: end of the epilogue
    if (closes) {   // E moves on (its column's resources follow at a column change)
      ++qE;
      if (++E.di == nTd) {
        E.di = 0;
        ++E.cj;
        if (E.cj < ncols_blk) enter_column_E();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (a.dbg_flags & 128) stamp();   // cursor advanced
    request_mask();
    if (stage) {   // request the upper planes of the tile after P
      ++qP;
      if (++P.di == nTd) {
        P.di = 0;
        ++P.cj;
        if (qP < items_mine) enter_column_P();
      }
      if (qP < items_mine && !no_stage) load_planes(2 * P.di, 2);
    }
    if (a.dbg_flags & 128) stamp();   // requests issued
    init_acc();
  };
  // Each wave group runs its own straight loop (one phase apart, paced by the block barrier), two tiles per trip, so
  // that the registers that live across phases -- the accumulators and the halo planes in flight -- are plain
  // loop-carried values: in one loop over phases with a group-dependent branch the compiler copied all 64 of them at
  // the loop's end, behind an s_waitcnt vmcnt(0) that undid the prefetch.
  if (grp == 0) {
    init_acc();
    request_mask();
    for (int q = 0; q < items_mine; q += 2) {
      const bool second = q + 1 < items_mine;
      stamp();
      mfma_even();
      stamp();
      __syncthreads();
      stamp();
      off_phase(true, second);
      stamp();
      __syncthreads();
      if (second) mfma_odd();
      __syncthreads();
      off_phase(second, q + 2 < items_mine);
      __syncthreads();
    }
  } else {
    for (int q = 0; q < items_mine; q += 2) {
      const bool second = q + 1 < items_mine;
      stamp();
      off_phase(q > 0, true);
      stamp();
      __syncthreads();
      stamp();
      mfma_even();
      stamp();
      __syncthreads();
      off_phase(true, second);
      __syncthreads();
      if (second) mfma_odd();
      __syncthreads();
    }
    if (items_mine > 0 && (items_mine & 1) == 0) off_phase(true, false);
  }
}

template <int GC, int KS, int EPI, bool UPS = false, bool INM = false>
static int launch_fwd3s_inst(const ConvFwdArgs& a, unsigned gx, size_t lds, hipStream_t st) {
  auto kern = conv_fwd3s_kernel<GC, KS, EPI, UPS, INM>;
  SG_ALLOW_160K_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(gx, (unsigned)a.ntile), dim3(512), lds, st, a);
  return SG_OK;
}

// The sliding-halo kernel exists for bf16 with whole 32-channel output tiles; KS as in the kernel.
template <int GC, int KS = 0>
static int launch_fwd3s(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  if (s->kd != 3 || s->kh != 3 || s->kw != 3) return SG_OK;
  // the fused nearest-x2 gather exists for the two passes of a split layer only (G's 64 -> 32 after upscale3d)
  if (s->upsample_in && (KS == 0 || GC != 2 || ((s->d | s->h | s->w) & 1))) return SG_OK;
  if (a.in_mask && !s->upsample_in) return SG_OK;
  if (s->d < 4 || (s->w % 32) != 0 || (s->cout % 32) != 0) return SG_OK;   // >= 2 steps per column; full 32-wide rows and tiles
  if (a.pool && ((s->d & 1) || (s->h & 1) || a.pixel_norm || (a.mask_bits && (a.sign_out || a.bias || a.act)) || KS != 0)) return SG_OK;
  if (a.pixel_norm && (a.mask_bits || KS == 1 || a.ntile != 1)) return SG_OK;
  if (a.mask_bits && a.sign_out) return SG_OK;
  if (a.pnb_y && (KS != 0 || GC != 2 || a.ntile != 1 || !a.mask_bits || a.pixel_norm || a.pool || a.sign_out || a.bias || a.act ||
                  !a.pnb_scale)) return SG_OK;
  a.g = sg_make_geom(s, 256, /*prefer_w32=*/true, /*td=*/2, /*th=*/4);
  const sg_tile_geom& g = a.g;
  if (g.TN != 1 || g.TD != 2 || g.TH != 4 || g.TW != 32 || g.HD != 4 || g.HH != 6 || g.HW != 34 || g.nTd < 2) return SG_OK;
  {   // buffer addressing (rebased per sample): one sample of every tensor this kernel touches stays below 2 GiB
    const int64_t svox = (int64_t)s->d * s->h * s->w;
    if (svox * a.xcs * 2 >= (1ll << 31) || svox * s->cout * 4 >= (1ll << 31) || svox * a.ntile * 4 >= (1ll << 31)) return SG_OK;
  }
  int gx = (256 / a.ntile) / 8 * 8;
  if (gx < 8) gx = 8;
  if (g.nTn * ((g.nTh + 1) / 2) * g.nTw < gx) return SG_OK;   // at least one column pair per block
  constexpr int PB = (GC == 1 ? 224 : 208) * GC * 32;
  const size_t lds = 8ull * PB + 27ull * GC * 1024 + 128;
  if (lds > 160 * 1024) return SG_OK;
  a.G = GC;
  a.rs = GC * 32;
  a.vec_in = 1;
  a.vec_out = 1;
  const int epi = (a.sign_out ? SG_EP_SIGN : 0) | (a.mask_bits ? SG_EP_MASK : 0) | (a.pixel_norm ? SG_EP_PN : 0) |
                  (a.pool ? SG_EP_POOL : 0) | (a.pnb_y ? SG_EP_PNB : 0);
  int rc = SG_OK;
  if constexpr (KS == 1) {
    if constexpr (GC == 2) {
      if (s->upsample_in && a.in_mask) rc = launch_fwd3s_inst<GC, 1, 0, true, true>(a, (unsigned)gx, lds, st);
      else if (s->upsample_in) rc = launch_fwd3s_inst<GC, 1, 0, true>(a, (unsigned)gx, lds, st);
      else rc = launch_fwd3s_inst<GC, 1, 0>(a, (unsigned)gx, lds, st);
    }
    else rc = launch_fwd3s_inst<GC, 1, 0>(a, (unsigned)gx, lds, st);
  } else if constexpr (KS == 2) {
    if (s->upsample_in && a.in_mask) {
      if constexpr (GC == 2) {
        switch (epi) {
          case 0: rc = launch_fwd3s_inst<GC, 2, 0, true, true>(a, (unsigned)gx, lds, st); break;
          case SG_EP_MASK: rc = launch_fwd3s_inst<GC, 2, SG_EP_MASK, true, true>(a, (unsigned)gx, lds, st); break;
          default: return SG_OK;
        }
      } else return SG_OK;
    } else if (s->upsample_in) {
      if constexpr (GC == 2) {
        switch (epi) {
          case 0: rc = launch_fwd3s_inst<GC, 2, 0, true>(a, (unsigned)gx, lds, st); break;
          case SG_EP_SIGN: rc = launch_fwd3s_inst<GC, 2, SG_EP_SIGN, true>(a, (unsigned)gx, lds, st); break;
          case SG_EP_PN: rc = launch_fwd3s_inst<GC, 2, SG_EP_PN, true>(a, (unsigned)gx, lds, st); break;
          case SG_EP_PN | SG_EP_SIGN: rc = launch_fwd3s_inst<GC, 2, SG_EP_PN | SG_EP_SIGN, true>(a, (unsigned)gx, lds, st); break;
          default: return SG_OK;
        }
      } else return SG_OK;
    } else
    switch (epi) {
      case 0: rc = launch_fwd3s_inst<GC, 2, 0>(a, (unsigned)gx, lds, st); break;
      case SG_EP_SIGN: rc = launch_fwd3s_inst<GC, 2, SG_EP_SIGN>(a, (unsigned)gx, lds, st); break;
      case SG_EP_MASK: rc = launch_fwd3s_inst<GC, 2, SG_EP_MASK>(a, (unsigned)gx, lds, st); break;
      case SG_EP_PN: rc = launch_fwd3s_inst<GC, 2, SG_EP_PN>(a, (unsigned)gx, lds, st); break;
      case SG_EP_PN | SG_EP_SIGN: rc = launch_fwd3s_inst<GC, 2, SG_EP_PN | SG_EP_SIGN>(a, (unsigned)gx, lds, st); break;
      default: return SG_OK;
    }
  } else {
    switch (epi) {
      case 0: rc = launch_fwd3s_inst<GC, 0, 0>(a, (unsigned)gx, lds, st); break;
      case SG_EP_SIGN: rc = launch_fwd3s_inst<GC, 0, SG_EP_SIGN>(a, (unsigned)gx, lds, st); break;
      case SG_EP_MASK: rc = launch_fwd3s_inst<GC, 0, SG_EP_MASK>(a, (unsigned)gx, lds, st); break;
      case SG_EP_PN: rc = launch_fwd3s_inst<GC, 0, SG_EP_PN>(a, (unsigned)gx, lds, st); break;
      case SG_EP_PN | SG_EP_SIGN: rc = launch_fwd3s_inst<GC, 0, SG_EP_PN | SG_EP_SIGN>(a, (unsigned)gx, lds, st); break;
      case SG_EP_SIGN | SG_EP_POOL: rc = launch_fwd3s_inst<GC, 0, SG_EP_SIGN | SG_EP_POOL>(a, (unsigned)gx, lds, st); break;
      case SG_EP_POOL: rc = launch_fwd3s_inst<GC, 0, SG_EP_POOL>(a, (unsigned)gx, lds, st); break;
      case SG_EP_MASK | SG_EP_POOL:
      case SG_EP_MASK | SG_EP_POOL | SG_EP_SIGN:      // block means of M * conv(x): the double backward of a pooled LeakyReLU layer
        if constexpr (GC == 2) {                        // (a caller's sign_out then receives the signs of M * conv(x): nobody asks)
          if (epi & SG_EP_SIGN) return SG_OK;
          rc = launch_fwd3s_inst<GC, 0, SG_EP_MASK | SG_EP_POOL>(a, (unsigned)gx, lds, st);
          break;
        } else return SG_OK;
      case SG_EP_MASK | SG_EP_PNB:
        if constexpr (GC == 2) { rc = launch_fwd3s_inst<GC, 0, SG_EP_MASK | SG_EP_PNB>(a, (unsigned)gx, lds, st); break; }
        else return SG_OK;
      default: return SG_OK;
    }
  }
  if (rc != SG_OK) return rc;
  SG_KNAME(KS ? "conv_fwd3s<bf16,%d> K-split" : "conv_fwd3s<bf16,%d>", GC);
  SG_LAUNCH_CHECK();
  *used = true;
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------------
// forward, v4: persistent 8-wave ping-pong for ANY channel count (multiples of the 16-byte piece): the K loop
// runs in 32-byte channel chunks (16 bf16 / 8 f32 channels).  A work item is (tile, chunk); group g = item
// parity owns halo buffer g and weight buffer g.  While group g runs the 27 x MTW x NTB MFMAs of its item, the
// other group fetches ITS next halo chunk and weight slab (all taps of that chunk, NTB output tiles) by
// LDS-DMA and, when its previous item closed a tile, runs that tile's epilogue.  One barrier per item.
// ------------------------------------------------------------------------------------------------------
template <typename T, int MTW, int NTB, int KD, int KH, int KW, int RS = 1>
__global__ __launch_bounds__(512) void conv_fwd4_kernel(ConvFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TAPS = KD * KH * KW;
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int CH = sg_traits<T>::CH;
  const sg_tile_geom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int r = lane & 31, hh = lane >> 5;
  // LDS map: [halo 0][halo 1][weights 0][weights 1][bias: NTB*32 floats]
  char* xmine = smem + grp * a.xbytes;
  // a.wres: every chunk's weight slab stays resident ([chunk][tap][nt]).  Otherwise two slab buffers SHARED by the
  // wave groups: both groups walk the chunks in lockstep (item j of either group uses chunk j % ncg, group 1 one
  // phase after group 0), so slab j sits in buffer j & 1 for two phases while each of the two off-phases fetches
  // one half of slab j+1 into the other buffer -- half the weight DMA of a private slab per group and item, which
  // at 54 KiB per item was more LDS-DMA than the MFMA phase could cover.
  char* wmine = smem + 2 * a.xbytes;
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * a.xbytes + (a.wres ? a.nchunk : 2) * a.wbytes);
  const T* x = reinterpret_cast<const T*>(a.x);
  const char* wp = reinterpret_cast<const char*>(a.wp);
  constexpr int rb = 32;                      // halo row bytes: one 32-byte chunk, 2 slots, f(row) = (row>>3)&1
  const int nt0 = blockIdx.y * NTB;
  const int ntb = min(NTB, a.ntile - nt0);
  const int tvox = g.TN * g.TD * g.TH * g.TW;
  const int ncg = a.nchunk;                   // chunks per tile

  const int xg = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (a.ntiles + 7) >> 3;
  const int t_begin = xg * cpx;
  const int t_end = min(a.ntiles, t_begin + cpx);
  const int first = t_begin + slot;
  const int K = first < t_end ? (t_end - first + per_x - 1) / per_x : 0;   // tiles of this block
  const int kmine = (K + 1 - grp) >> 1;                                     // tiles of my group
  const int items_mine = kmine * ncg;
  const int items_max = ((K + 1) >> 1) * ncg;                               // group 0 has the most

  // HS (64 accumulators per lane: two M x two N tiles, or four M tiles): the address table is shared between the
  // wave's M tiles (sg_xa_index); the host launches it only where M tile mt is M tile 0 moved by mt H rows
  constexpr bool HS = (MTW * NTB >= 4);
  constexpr int NA = sg_xa_size<MTW, KD, KH, KW, HS, RS>::value;
  int xaddr[NA];
  int tcoord[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const int m = (wave * MTW + mt) * 32 + r;
    uint32_t q = sg_div((uint32_t)m, g.fTW);
    int tw = m - (int)q * g.TW;
    uint32_t q2 = sg_div(q, g.fTH);
    int th = (int)(q - q2 * g.TH);
    uint32_t q3 = sg_div(q2, g.fTD);
    int td = (int)(q2 - q3 * g.TD);
    int tn = (int)q3;
    const int lrow = (m < tvox) ? (((tn * g.HD + td) * g.HH + th) * g.HW + tw) : 0;
    tcoord[mt] = (m < tvox) ? (tw | (th << 8) | (td << 16) | (tn << 24)) : -1;
    if (HS && mt > 0) continue;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd)
#pragma unroll
      for (int kh = 0; kh < KH + (HS ? RS * (MTW - 1) : 0); ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const int row = lrow + ((kd + a.tap_d) * g.HH + (kh + a.tap_h)) * g.HW + (kw + a.tap_w);
          const int idx = HS ? (kd * (KH + RS * (MTW - 1)) + kh) * KW + kw : ((kd * KH + kh) * KW + kw) * MTW + mt;
          xaddr[idx] = grp * a.xbytes + row * rb + ((hh ^ ((row >> 3) & 1)) << 4);
        }
  }
  const int hv = g.TN * g.HD * g.HH * g.HW;
  const int items = hv * 2;
  constexpr int MAXIT = MTW > 2 ? 10 : (RS == 2 ? 6 : 8);   // LDS-DMA pieces per wave per halo chunk (host-checked)
  int it_rel[MAXIT];   // element offset (chunk 0) relative to the tile's first halo voxel, -1 dead
  int it_crd[MAXIT];
  const int Di = g.ups ? (g.D >> 1) : g.D, Hi = g.ups ? (g.H >> 1) : g.H, Wi = g.ups ? (g.W >> 1) : g.W;
#pragma unroll
  for (int k = 0; k < MAXIT; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> 1;
    const int sl = (it & 1) ^ ((row >> 3) & 1);
    uint32_t q = sg_div((uint32_t)row, g.fHW);
    int hw = (int)(row - q * g.HW);
    uint32_t q2 = sg_div(q, g.fHH);
    int hh_ = (int)(q - q2 * g.HH);
    uint32_t q3 = sg_div(q2, g.fHD);
    int hd = (int)(q2 - q3 * g.HD);
    it_rel[k] = row < hv ? ((((int)q3 * g.D + hd) * g.H + hh_) * g.W + hw) * a.cin + sl * EPP : -2;   // -2: beyond the image, lane idle
    if (g.ups && a.lean && row < hv)   // fused nearest x2 gather: tile origins are even (host-checked), so the halved
      // coordinate of halo voxel hd is d0/2 + ((hd - PD) >> 1): an offset relative to the tile's low-resolution origin
      it_rel[k] = ((((int)q3 * Di + ((hd - g.PD) >> 1)) * Hi + ((hh_ - g.PH) >> 1)) * Wi + ((hw - g.PW) >> 1)) * a.cin + sl * EPP;
    // packed halo coordinate for the boundary test (bytes w,h,d,n, all < 128); bit 31 = second 16-byte slot
    it_crd[k] = row < hv ? (hw | (hh_ << 8) | (hd << 16) | ((int)q3 << 24) | (sl << 31)) : 0x7F7F7F7F;
  }
  constexpr uint32_t DEAD = 0x80000000u;        // byte offset beyond the buffer: the load returns zeros
  const int64_t sample_bytes = (int64_t)Di * Hi * Wi * a.cin * (int)sizeof(T);   // of the (possibly half-resolution) input

  auto tile_of_item = [&](int q) { return first + (2 * (q / ncg) + grp) * per_x; };

  auto stage_item = [&](int q) {
    const int t = tile_of_item(q);
    const int cg = q % ncg;
    const sg_tile_origin o = sg_tile_of(g, (uint32_t)t);
    const int c0 = cg * CH;
    const bool interior = !g.ups && o.d0 >= g.PD && o.h0 >= g.PH && o.w0 >= g.PW && o.d0 + g.TD + g.PD <= g.D &&
                          o.h0 + g.TH + g.PH <= g.H && o.w0 + g.TW + g.PW <= g.W && o.n0 + g.TN <= g.N &&
                          c0 + CH <= a.cin;
    if (interior) {
      const T* base = x + ((((int64_t)o.n0 * g.D + (o.d0 - g.PD)) * g.H + (o.h0 - g.PH)) * g.W + (o.w0 - g.PW)) *
                              (int64_t)a.cin + c0;
#pragma unroll
      for (int k = 0; k < MAXIT; ++k)
        if ((wave + 4 * k) * 64 < items && it_rel[k] != -2)   // idle lanes write nothing: the image ends at hv rows
          sg_glds16(it_rel[k] >= 0 ? (const void*)(base + it_rel[k]) : (const void*)sg_zero_page,
                    xmine + (size_t)(wave + 4 * k) * 1024);
    } else if (a.lean) {
      // boundary tile: a piece is valid iff its packed halo coordinate lies inside the per-tile range in every byte
      // (two byte-parallel subtractions, guard bit 7) and its channels exist; everything else reads zeros through
      // the buffer's bounds check.  (At <= 64^2 every tile is a W-boundary tile: this IS the common path there.)
      const int lo_w = max(0, g.PW - o.w0), hi_w = min(g.HW, g.W + g.PW - o.w0) - 1;
      const int lo_h = max(0, g.PH - o.h0), hi_h = min(g.HH, g.H + g.PH - o.h0) - 1;
      const int lo_d = max(0, g.PD - o.d0), hi_d = min(g.HD, g.D + g.PD - o.d0) - 1;
      const int hi_n = min(g.TN, g.N - o.n0) - 1;
      const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8) | (lo_d << 16));
      const uint32_t hi = (uint32_t)(hi_w | (hi_h << 8) | (hi_d << 16) | (hi_n << 24)) | 0x80808080u;
      // the resource starts at the tile's first sample (a 64-bit scalar add): offsets stay below 2 GiB for any batch
      const int64_t left = (int64_t)(g.N - o.n0) * sample_bytes;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.x)) + (int64_t)o.n0 * sample_bytes, 0,
          (int)(left < 0x7FFFFFFFll ? left : 0x7FFFFFFFll), 0x00020000);
      const int tile_off = g.ups ? (int)(((((int64_t)(o.d0 >> 1) * Hi + (o.h0 >> 1)) * Wi + (o.w0 >> 1)) * (int64_t)a.cin + c0) *
                                         (int)sizeof(T))
                                 : (int)((((((int64_t)(o.d0 - g.PD)) * g.H + (o.h0 - g.PH)) * g.W + (o.w0 - g.PW)) *
                                          (int64_t)a.cin + c0) * (int)sizeof(T));
      const bool tail = c0 + CH > a.cin;                 // the second slot's channels may not exist
#pragma unroll
      for (int k = 0; k < MAXIT; ++k) {
        if ((wave + 4 * k) * 64 < items && it_crd[k] != 0x7F7F7F7F) {   // idle lanes: beyond the halo image
          uint32_t craw = (uint32_t)it_crd[k];
          asm volatile("" : "+v"(craw));   // opaque: or the compiler keeps three derived copies of the table in registers
          const uint32_t c_ = craw & 0x7FFFFFFFu;
          const uint32_t t1 = (c_ | 0x80808080u) - lo, t2 = hi - c_;
          bool ok = (t1 & t2 & 0x80808080u) == 0x80808080u;
          if (tail) ok = ok && (c0 + (int)(craw >> 31) * EPP < a.cin);
          const uint32_t vo = ok ? (uint32_t)(it_rel[k] * (int)sizeof(T) + tile_off) : DEAD;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xmine + (size_t)(wave + 4 * k) * 1024), 16, vo, 0, 0, 0);
        }
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < MAXIT; ++k) {
        if ((wave + 4 * k) * 64 < items) {
          const int it = (wave + 4 * k) * 64 + lane;
          const int row = it >> 1;
          const int sl = (it & 1) ^ ((row >> 3) & 1);
          uint32_t q_ = sg_div((uint32_t)row, g.fHW);
          int hw = (int)(row - q_ * g.HW);
          uint32_t q2 = sg_div(q_, g.fHH);
          int hh_ = (int)(q_ - q2 * g.HH);
          uint32_t q3 = sg_div(q2, g.fHD);
          int hd = (int)(q2 - q3 * g.HD);
          int n = o.n0 + (int)q3, d = o.d0 + hd - g.PD, h = o.h0 + hh_ - g.PH, w = o.w0 + hw - g.PW;
          const int c = c0 + sl * EPP;
          const void* src = sg_zero_page;
          if (row >= hv) continue;
          if (c < a.cin && n < g.N && (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H &&
              (unsigned)w < (unsigned)g.W) {
            if (g.ups) { d >>= 1; h >>= 1; w >>= 1; }
            src = x + ((((int64_t)n * Di + d) * Hi + h) * Wi + w) * (int64_t)a.cin + c;
          }
          sg_glds16(src, xmine + (size_t)(wave + 4 * k) * 1024);
        }
      }
    }
  };
  // fragments [f0, f1) of the slab of item j ([tap][nt] order) into buffer j & 1, by the 4 waves of my group
  auto stage_slab = [&](int j, int f0, int f1, int w, int nw) {
    const int cg = j % ncg;
    char* dst = wmine + (j & 1) * a.wbytes;
    for (int f = f0 + w; f < f1; f += nw) {
      const int tap = f / NTB, nt = f - tap * NTB;
      const char* src = nt < ntb ? wp + ((((int64_t)cg * TAPS + tap) * a.ntile + (nt0 + nt)) << 10)
                                 : reinterpret_cast<const char*>(sg_zero_page);
      sg_glds16(src + lane * 16, dst + ((size_t)f << 10));
    }
  };

  if (tid < NTB * 32) {
    const int co = nt0 * 32 + tid;
    bias_lds[tid] = (a.bias != nullptr && co < a.cout) ? a.bias[co] : 0.f;
  }
  if (a.wres) {   // all chunks' slabs, once, by all 8 waves
    const int nfrag_all = ncg * TAPS * NTB;
    for (int f = wave8; f < nfrag_all; f += 8) {
      const int nt = f % NTB;
      const int q_ = f / NTB;
      const int tap = q_ % TAPS, cg = q_ / TAPS;
      const char* src = nt < ntb ? wp + ((((int64_t)cg * TAPS + tap) * a.ntile + (nt0 + nt)) << 10)
                                 : reinterpret_cast<const char*>(sg_zero_page);
      sg_glds16(src + lane * 16, wmine + ((size_t)f << 10));
    }
  }
  if (!a.wres && items_max > 0) stage_slab(0, 0, TAPS * NTB, wave8, 8);
  if (grp == 0 && items_mine > 0) stage_item(0);
  __syncthreads();

  const char* wl = wmine + lane * 16;
  T* y = reinterpret_cast<T*>(a.y);
  const float inv_c = 1.f / (float)a.cout;
  const bool wide_store = (a.cout % 32 == 0) && !(a.dbg_flags & 16);   // whole 32-channel tiles, 16-byte aligned rows
  // output voxel index: the conv's own grid, or (sub-pixel class) voxel (2d+oa, 2h+ob, 2w+oc) of the x2 tensor
  auto ovox = [&](int n, int d, int h, int w) -> int64_t {
    if (a.os == 2) return ((((int64_t)n * (2 * g.D) + 2 * d + a.oa) * (2 * g.H) + 2 * h + a.ob) * (2 * g.W) + 2 * w + a.oc);
    return (((int64_t)n * g.D + d) * g.H + h) * g.W + w;
  };
  f32x16 acc[MTW][NTB];

  // diagnostic time stamps (tools/ts_conv.py): per phase [top, after staging, end].  Measured on 64->32 at 128^2:
  // MFMA phase 2.4k cycles, staging of one 26-KiB halo chunk 3.3k (the CU's ~20 B/clk share of the memory system),
  // epilogue 6k every fourth item: this layer is bound by the bytes staged per tile (4 chunk passes, no sliding
  // halo -- a ring of 64-channel planes does not fit next to 108 KiB of resident weights).
  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && lane == 0 && wave == 0 && dbgi < 120)
      a.dbg[grp * 128 + dbgi] = __builtin_amdgcn_s_memtime();
    ++dbgi;
  };
  const int nphase = 2 * items_max + 1;
  for (int p = 0; p < nphase; ++p) {
    const int q = p >> 1;                      // item index within the running group
    stamp();
    if ((p & 1) == grp) {
      if (q < items_mine) {
        if (q % ncg == 0) {
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
              for (int i = 0; i < 16; ++i) acc[mt][nt][i] = bias_lds[nt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];
        }
        // (four M tiles: a step already carries 4 MFMAs = 128 cycles, one step of fragment prefetch covers the LDS latency)
        sg_unrolled_k4<T, MTW, NTB, TAPS, (MTW > 2 || RS == 2 ? 2 : 3) /* RS = 2: 45 table registers, ring of 2 */, KH, KW, HS, NA, RS>::run(acc, xaddr, (int)(wl - smem) + (a.wres ? (q % ncg) : (q & 1)) * a.wbytes);
      }
    } else {
      // my next item is q' = (p + 1) >> 1; my previous one q' - 1 (ran in phase p - 1)
      const int qn = (p + 1) >> 1;
      const int qp = qn - 1;
      const bool closes = qp >= 0 && qp < items_mine && (qp % ncg) == ncg - 1;   // my previous item finished a tile
      uint32_t mb[MTW][NTB];
      const bool use_mask = a.mask_bits != nullptr && closes;
      if (use_mask) {   // sign words of the tile about to be stored, requested ahead of the halo DMA
        const sg_tile_origin om = sg_tile_of(g, (uint32_t)tile_of_item(qp));
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int tc = tcoord[mt];
          const int n = om.n0 + (tc >> 24), d = om.d0 + ((tc >> 16) & 255), h = om.h0 + ((tc >> 8) & 255),
                    w = om.w0 + (tc & 255);
          const bool ok = tc >= 0 && n < g.N && d < g.D && h < g.H && w < g.W;
          const uint32_t* mrow = a.mask_bits + ovox(n, d, h, w) * a.ntile + nt0;
#pragma unroll
          for (int nt = 0; nt < NTB; ++nt) mb[mt][nt] = (ok && nt < ntb) ? mrow[nt] : 0u;
        }
      }
      if (qn < items_mine) stage_item(qn);
      stamp();
      if (!a.wres && (p >> 1) + 1 < items_max) {   // my half of the next item's slab (phase parity picks the half)
        constexpr int nfrag = TAPS * NTB, hf = (nfrag + 1) / 2;
        stage_slab((p >> 1) + 1, (p & 1) ? hf : 0, (p & 1) ? nfrag : hf, wave, 4);
      }
      if (closes) {
        const sg_tile_origin o = sg_tile_of(g, (uint32_t)tile_of_item(qp));
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int tc = tcoord[mt];
          const int n = o.n0 + (tc >> 24), d = o.d0 + ((tc >> 16) & 255), h = o.h0 + ((tc >> 8) & 255),
                    w = o.w0 + (tc & 255);
          const bool ok = tc >= 0 && n < g.N && d < g.D && h < g.H && w < g.W;
          if (a.act) {
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
              for (int i = 0; i < 16; ++i) acc[mt][nt][i] = fmaxf(acc[mt][nt][i], acc[mt][nt][i] * a.slope);
          }
          if (a.pixel_norm) {
            float ss = 0.f;
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
              for (int i = 0; i < 16; ++i) ss += acc[mt][nt][i] * acc[mt][nt][i];
            ss += __shfl_xor(ss, 32);
            const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
              for (int i = 0; i < 16; ++i) acc[mt][nt][i] *= sc;
            if (a.pn_scale != nullptr && hh == 0 && ok)
              a.pn_scale[ovox(n, d, h, w)] = sc;
          }
          if (a.sign_out != nullptr) {
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) {
              const uint32_t sw = sg_sign_word(acc[mt][nt], hh);
              if (hh == 0 && ok && nt < ntb)
                a.sign_out[ovox(n, d, h, w) * a.ntile + nt0 + nt] = sw;
            }
          }
          if (use_mask) {
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) sg_apply_sign_word(acc[mt][nt], mb[mt][nt], hh, a.mask_slope);
          }
          if (sizeof(T) == 2 && wide_store) {   // uniform: 16 contiguous bytes per lane after a half-wave exchange
            bf16_t* yrow = reinterpret_cast<bf16_t*>(y) + ovox(n, d, h, w) * (int64_t)a.cout + nt0 * 32;
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) sg_store_tile_row_bf16(yrow + nt * 32, acc[mt][nt], hh, ok && nt < ntb);
          } else if (ok) {
            T* yrow = y + ovox(n, d, h, w) * (int64_t)a.cout;
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                const int co = (nt0 + nt) * 32 + 8 * qd + 4 * hh;
                if (a.vec_out && co + 4 <= a.cout) {
                  T tmp[4];
#pragma unroll
                  for (int e = 0; e < 4; ++e) tmp[e] = sg_traits<T>::from_f(acc[mt][nt][qd * 4 + e]);
                  if (sizeof(T) == 2) *reinterpret_cast<u32x2*>(yrow + co) = *reinterpret_cast<u32x2*>(tmp);
                  else { const u32x4 t16 = *reinterpret_cast<u32x4*>(tmp); *reinterpret_cast<u32x4*>(yrow + co) = t16; SG_STORE16_GUARD(t16); }
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (co + e < a.cout) yrow[co + e] = sg_traits<T>::from_f(acc[mt][nt][qd * 4 + e]);
                }
              }
          }
        }
      }
    }
    stamp();
    __syncthreads();
  }
}

template <typename T, int MTW, int NTB, int KD, int KH, int KW, int RS = 1>
static int launch_fwd4(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  constexpr int BM = MTW * 128;
  a.g = sg_make_geom(s, BM, /*prefer_w32=*/true);
  const sg_tile_geom& g = a.g;
  if (g.TW > 255 || g.TH > 255 || g.TD > 255 || g.TN > 127) return SG_OK;
  const int64_t ntiles = (int64_t)g.nTn * g.nTd * g.nTh * g.nTw;
  const int ny = sg_cdiv(a.ntile, NTB);
  int gx = (256 / ny) / 8 * 8;
  if (gx < 8) gx = 8;
  if (sg_cfg().fwd4_gx > 0) gx = sg_cfg().fwd4_gx;   // tools shrink the grid to reach this kernel with small tensors
  if (gx < 8 || (gx & 7)) return SG_EINVAL;
  if (ntiles >= (1 << 24) || ntiles < 2 * gx) return SG_OK;
  // one tile per wave group leaves nothing to overlap: on the 512 -> 512 (1,3,3) layers of the 2x8x8 level at batch 64
  // (32 tiles, 16 blocks in x) the weight-stationary kernel is faster, 92 against 123 us
  if (KD == 1 && ntiles < 4 * gx) return SG_OK;
  // per-lane halo offsets are relative to the tile's first sample: only TN samples have to fit 31 bits
  if ((int64_t)g.TN * s->d * s->h * s->w * (int64_t)s->cin * (int64_t)sizeof(T) >= (1ll << 31)) return SG_OK;
  const int hv = g.TN * g.HD * g.HH * g.HW;
  a.G = 1;
  a.rs = 32;
  a.xbytes = hv * 32;   // exact: lanes beyond the last row are masked off in the LDS-DMA
  // buffer resource rebased per tile sample; with the fused x2 gather the tile origins must be even
  a.lean = ((!s->upsample_in || (g.TD % 2 == 0 && g.TH % 2 == 0 && g.TW % 2 == 0 && g.TN == 1)) &&
            !sg_cfg().fwd4_no_lean) ? 1 : 0;
  a.wbytes = a.taps * NTB * 1024;
  if (sg_cdiv(hv * 2, 64) > (MTW > 2 ? 40 : (RS == 2 ? 24 : 32))) return SG_OK;
  // shared address table (sg_xa_index): M tile mt must be M tile 0 moved by RS * mt H rows of the same (n, d) plane
  if (MTW * NTB >= 4 && !(RS == 1 ? (g.TW == 32 && g.TH % MTW == 0) : (g.TW == 16 && g.TH % (2 * MTW) == 0))) return SG_OK;
  size_t lds = 2ull * a.xbytes + (size_t)a.nchunk * a.wbytes + NTB * 128;
  a.wres = (lds <= 160 * 1024 && !sg_cfg().fwd4_no_wres) ? 1 : 0;
  if (!a.wres) lds = 2ull * a.xbytes + 2ull * a.wbytes + NTB * 128;
  if (lds > 160 * 1024) return SG_OK;
  a.ntiles = (int)ntiles;
  a.vec_in = 1;
  a.vec_out = (s->cout % 4 == 0) ? 1 : 0;
  auto kern = conv_fwd4_kernel<T, MTW, NTB, KD, KH, KW, RS>;
  SG_ALLOW_160K_LDS(kern);
  SG_KNAME("conv_fwd4<%s,%d,%d,%d,%d,%d>", sg_tname<T>(), MTW, NTB, KD, KH, KW);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ny), dim3(512), lds, st, a);
  SG_LAUNCH_CHECK();
  *used = true;
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------------
// forward, v5: the streamed kernel (v4) for its hot class -- bf16, 3x3x3, no fused up-sampling, two M x two N tiles
// per wave (64 accumulators), whole 64-channel output slices, one sample per tile -- with the lessons of the
// sliding-halo kernel applied to its off-phase, which at 3.7k cycles of issue per item (5.3k with the DMA wait, 8.6k when
// the item closes a tile) against a 4.3k-cycle MFMA phase bounded these layers (in-kernel stamps, tools/ts_conv.py):
// per-TILE lane offsets and resources (a tile's ncg channel chunks differ by a scalar offset), buffer-addressed LDS-DMA
// for halo and weight slab, compile-time epilogue variants, per-group loops, in-place MFMAs.
// Work item = (tile, 32-byte channel chunk); group g owns halo buffer g; the weight slabs (all taps of a chunk, both
// N tiles) are shared by the two groups, slab j + 1 being fetched half by each group's off-phase (see v4).
// ------------------------------------------------------------------------------------------------------
template <int RING>
struct sg_unrolled_k5 {   // one step = one tap: 2 weight + 2 activation fragments, 4 MFMAs
  static constexpr int TAPS = 27, PF = RING - 1, RPS = 4, NA = 36;
  template <int ST>
  static __device__ __forceinline__ void load(u32x4 (&wfr)[RING][2], u32x4 (&xfr)[RING][2], const int (&xaddr)[NA], int wl_off) {
    constexpr int SL = ST % RING;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wfr[SL][nt]) : "v"(wl_off), "n"((ST * 2 + nt) << 10));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      asm volatile("ds_read_b128 %0, %1" : "=v"(xfr[SL][mt]) : "v"(xaddr[sg_xa_index<2, 3, 3, true>(ST, mt)]));
  }
  template <int ST>
  static __device__ __forceinline__ void step(f32x16 (&acc)[2][2], u32x4 (&wfr)[RING][2], u32x4 (&xfr)[RING][2],
                                              const int (&xaddr)[NA], int wl_off) {
    if constexpr (ST < TAPS) {
      if constexpr (ST + PF < TAPS) load<ST + PF>(wfr, xfr, xaddr, wl_off);
      constexpr int younger = (TAPS - 1 - ST < PF ? TAPS - 1 - ST : PF) * RPS;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) sg_mfma_bf16_acc(acc[mt][nt], wfr[ST % RING][nt], xfr[ST % RING][mt]);
      __builtin_amdgcn_sched_barrier(0);
      step<ST + 1>(acc, wfr, xfr, xaddr, wl_off);
    }
  }
  template <int ST>
  static __device__ __forceinline__ void prologue(u32x4 (&wfr)[RING][2], u32x4 (&xfr)[RING][2], const int (&xaddr)[NA], int wl_off) {
    if constexpr (ST < PF) {
      load<ST>(wfr, xfr, xaddr, wl_off);
      prologue<ST + 1>(wfr, xfr, xaddr, wl_off);
    }
  }
  static __device__ __forceinline__ void run(f32x16 (&acc)[2][2], const int (&xaddr)[NA], int wl_off) {
    u32x4 wfr[RING][2], xfr[RING][2];
    SG_KLOOP_BEGIN();
    prologue<0>(wfr, xfr, xaddr, wl_off);
    step<0>(acc, wfr, xfr, xaddr, wl_off);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));   // MFMA result hazard
    SG_KLOOP_END();
  }
};

template <int EPI>
__global__ __launch_bounds__(512) void conv_fwd5_kernel(ConvFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int TAPS = 27, ES = 2, EPP = 8, NFRAG = TAPS * 2, HF = (NFRAG + 1) / 2;
  constexpr uint32_t DEAD = 0x80000000u;             // byte offset beyond every buffer: loads return 0, stores drop
  constexpr bool SIGN = (EPI & SG_EP_SIGN) != 0, MASK = (EPI & SG_EP_MASK) != 0, PN = (EPI & SG_EP_PN) != 0;
  constexpr bool POOL = (EPI & SG_EP_POOL) != 0;     // y: the 1 x 2 x 2 (H x W) block means, [n, D, H/2, W/2, cout] (pool == 2)
  const sg_tile_geom& g = a.g;                       // TN = 1, TW = 32, 256 voxels, TH even (host-checked)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int r = lane & 31, hh = lane >> 5;
  // LDS map: [halo 0][halo 1][slab 0][slab 1][bias: 64 floats]
  const int xmine = grp * a.xbytes, wbase = 2 * a.xbytes;
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * a.xbytes + 2 * a.wbytes);
  const int nt0 = blockIdx.y * 2;
  const int ncg = a.nchunk, cout = a.cout, ntile = a.ntile, cin = a.cin;
  const int D = g.D, H = g.H, W = g.W;

  const int xg = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (a.ntiles + 7) >> 3;
  const int t_begin = xg * cpx;
  const int t_end = min(a.ntiles, t_begin + cpx);
  const int first = t_begin + slot;
  const int K = first < t_end ? (t_end - first + per_x - 1) / per_x : 0;   // tiles of this block
  const int kmine = (K + 1 - grp) >> 1;                                     // tiles of my group
  const int items_mine = kmine * ncg;
  const int items_max = ((K + 1) >> 1) * ncg;                               // group 0 has the most

  // fragment addresses (shared between the wave's two M tiles: M tile 1 is M tile 0 one H row further) and the
  // tile-relative coordinates of my two output voxels
  int xaddr[36];
  int tcoord[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = (wave * 2 + mt) * 32 + r;
    const int tw = m & 31, q = m >> 5;               // TW = 32
    const int th = q % g.TH, td = q / g.TH;
    tcoord[mt] = tw | (th << 8) | (td << 16);
    if (mt > 0) continue;
    const int lrow = (td * g.HH + th) * g.HW + tw;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 4; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int row = lrow + (kd * g.HH + kh) * g.HW + kw;
          xaddr[(kd * 4 + kh) * 3 + kw] = xmine + row * 32 + ((hh ^ ((row >> 3) & 1)) << 4);
        }
  }
  // halo staging table: this lane's 16-byte pieces of 1-KiB blocks wave, wave+4, ... of the halo image (32-byte rows)
  const int hv = g.HD * g.HH * g.HW;
  constexpr int MAXIT = 8;                           // host-checked: <= 32 pieces
  uint32_t relb[MAXIT];
  int crd[MAXIT];
#pragma unroll
  for (int k = 0; k < MAXIT; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> 1;
    const int sl = (it & 1) ^ ((row >> 3) & 1);
    uint32_t q = sg_div((uint32_t)row, g.fHW);
    const int hw = (int)(row - q * g.HW);
    uint32_t q2 = sg_div(q, g.fHH);
    const int hh_ = (int)(q - q2 * g.HH), hd = (int)q2;
    relb[k] = row < hv ? (uint32_t)((((hd * H + hh_) * W + hw) * cin + sl * EPP) * ES) : 0xC0000000u;
    crd[k] = row < hv ? (hw | (hh_ << 8) | (hd << 16)) : 0x7F7F7F7F;   // 0x7F7F7F7F: beyond the image, the lane stays idle
  }
  const int64_t svox = (int64_t)D * H * W;
  const int64_t xsb = svox * cin * ES, ysb = (POOL ? svox >> 2 : svox) * cout * ES, wsb = svox * ntile * 4, psb = svox * 4;
  auto rsrc_of = [&](const void* base, int64_t sample_bytes, int n0) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + n0 * sample_bytes, 0,
                                             (int)sample_bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, 0x7FFFFFFF, 0x00020000);

  // ---- staging cursor S: tile kS of my list, chunk cgS
  int kS = 0, cgS = 0;
  __amdgpu_buffer_rsrc_t rxS;
  uint32_t vk[MAXIT];
  auto enter_tile_S = [&]() {
    const sg_tile_origin o = sg_tile_of(g, (uint32_t)(first + (2 * kS + grp) * per_x));
    rxS = rsrc_of(a.x, xsb, o.n0);
    const int tile_off = (((o.d0 - 1) * H + (o.h0 - 1)) * W + (o.w0 - 1)) * cin * ES;   // may be negative: dead lanes only
    const int lo_w = max(0, 1 - o.w0), hi_w = min(g.HW, W + 1 - o.w0) - 1;
    const int lo_h = max(0, 1 - o.h0), hi_h = min(g.HH, H + 1 - o.h0) - 1;
    const int lo_d = max(0, 1 - o.d0), hi_d = min(g.HD, D + 1 - o.d0) - 1;
    const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8) | (lo_d << 16));
    const uint32_t hi = (uint32_t)(hi_w | (hi_h << 8) | (hi_d << 16)) | 0x808080u;
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
      const uint32_t c_ = (uint32_t)crd[k] & 0xFFFFFFu;
      const uint32_t t1 = (c_ | 0x808080u) - lo, t2 = hi - c_;
      vk[k] = (t1 & t2 & 0x808080u) == 0x808080u ? relb[k] + (uint32_t)tile_off : DEAD;
    }
  };
  auto stage_halo = [&]() {                                        // chunk cgS of tile kS into my halo buffer
    const uint32_t soff = (uint32_t)(cgS * 32);
#pragma unroll
    for (int k = 0; k < MAXIT; ++k)
      if ((wave + 4 * k) * 64 < hv * 2 && crd[k] != 0x7F7F7F7F)    // idle lanes write nothing: the image ends at hv rows
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxS, (lds_ptr_t)(smem + xmine + (wave + 4 * k) * 1024), 16, vk[k], soff, 0, 0);
  };
  // fragments [f0, f1) ([tap][nt] order) of the slab of item j into slab buffer j & 1, by nw waves of which I am w
  auto stage_slab = [&](int j, int f0, int f1, int w, int nw) {
    const int cg = j % ncg;
    char* dst = smem + wbase + (j & 1) * a.wbytes;
    const uint32_t cbase = (uint32_t)(cg * TAPS * ntile + nt0) << 10;
    for (int f = f0 + w; f < f1; f += nw) {
      const int tap = f >> 1, nt = f & 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + (f << 10)), 16, (uint32_t)(lane * 16),
                                               cbase + ((uint32_t)(tap * ntile + nt) << 10), 0, 0);
    }
  };
  // ---- output cursor E: tile kE of my list
  int kE = 0;
  __amdgpu_buffer_rsrc_t ryE, rsE, rmE, rpE;
  uint32_t vox[2];                                                 // my two output voxels within the sample; DEAD: outside
  uint32_t pvox = DEAD;                                            // POOL: the pooled voxel my lane writes (even w only)
  auto enter_tile_E = [&]() {
    const sg_tile_origin o = sg_tile_of(g, (uint32_t)(first + (2 * kE + grp) * per_x));
    ryE = rsrc_of(a.y, ysb, o.n0);
    if (SIGN) rsE = rsrc_of(a.sign_out, wsb, o.n0);
    if (MASK) rmE = rsrc_of(a.mask_bits, wsb, o.n0);
    if (PN) rpE = rsrc_of(a.pn_scale, psb, o.n0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int tc = tcoord[mt];
      const int d = o.d0 + (tc >> 16), h = o.h0 + ((tc >> 8) & 255), w = o.w0 + (tc & 255);
      vox[mt] = (d < D && h < H && w < W) ? (uint32_t)((d * H + h) * W + w) : DEAD;
      // H and W are even (host-checked) and M tile 0 sits on an even row: an even-w lane inside the volume has all
      // four voxels of its block inside
      if (POOL && mt == 0)
        pvox = (vox[0] != DEAD && !(r & 1)) ? (uint32_t)((d * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) : DEAD;
    }
  };
  // LeakyReLU sign words of E's tile (masked epilogue), requested when E enters the tile: ncg phases before their use
  uint32_t mbn[2][2] = {{0u, 0u}, {0u, 0u}};
  auto request_mask = [&]() {
    if constexpr (MASK) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          mbn[mt][nt] = __builtin_amdgcn_raw_buffer_load_b32(rmE, (kE < kmine && vox[mt] != DEAD) ? (vox[mt] * (uint32_t)ntile + nt0 + nt) * 4u : DEAD, 0, 0);
    }
  };

  if (tid < 64) {
    const int co = nt0 * 32 + tid;
    bias_lds[tid] = (a.bias != nullptr && co < cout) ? a.bias[co] : 0.f;
  }
  if (items_max > 0) stage_slab(0, 0, NFRAG, wave8, 8);
  if (items_mine > 0) {
    enter_tile_S();
    enter_tile_E();
    request_mask();
    if (grp == 0) stage_halo();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int wl0 = wbase + lane * 16;
  const float inv_c = 1.f / (float)cout;
  const float slope = a.act ? a.slope : 1.f;         // max(x, 1 * x) = x: no branch for "no activation"
  f32x16 acc[2][2];
  auto init_acc = [&]() {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][nt][i] = bias_lds[nt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];   // bias rides in C
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));   // four separate tuples (see v3s)
  };
  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && lane == 0 && wave == 0 && dbgi < 120)
      a.dbg[grp * 128 + dbgi] = __builtin_amdgcn_s_memtime();
    ++dbgi;
  };
  auto mfma_phase = [&](int q) { sg_unrolled_k5<3>::run(acc, xaddr, wl0 + (q & 1) * a.wbytes); };

  // off-phase: halo chunk of my item qn (cursor S), my half of slab js, epilogue of tile kE if `closes`
  auto off_phase = [&](bool closes, int qn, int js, int f0, int f1) {
    if (qn < items_mine) stage_halo();
    if (js < items_max) stage_slab(js, f0, f1, wave, 4);
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    if (closes) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bool ok = vox[mt] != DEAD;
        if (slope != 1.f) {   // uniform
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = sg_lrelu(acc[mt][nt][i], slope);
        }
        if (PN) {
          float ss = 0.f;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) ss += acc[mt][nt][i] * acc[mt][nt][i];
          const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
          ss = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);   // own half + partner lane ^ 32's
          const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] *= sc;
          if (a.pn_scale != nullptr)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc), rpE, (ok && hh == 0) ? vox[mt] * 4u : DEAD, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (SIGN) {
            uint32_t b = 0u;
#pragma unroll
            for (int i = 0; i < 16; ++i) b |= (__float_as_uint(acc[mt][nt][i]) >> 31) << ((i & 3) + 8 * (i >> 2));
            b <<= 4 * hh;
            const auto sw2 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
            __builtin_amdgcn_raw_buffer_store_b32(sw2[0] | sw2[1], rsE, (ok && hh == 0) ? (vox[mt] * (uint32_t)ntile + nt0 + nt) * 4u : DEAD, 0, 0);
          }
          if (MASK) sg_apply_sign_word(acc[mt][nt], mbn[mt][nt], hh, a.mask_slope);
          if constexpr (!POOL) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {   // 16 contiguous bytes per lane (see sg_store_tile_row_bf16)
              const uint32_t a0 = sg_pack_bf16(acc[mt][nt][8 * j + 0], acc[mt][nt][8 * j + 1]), a1 = sg_pack_bf16(acc[mt][nt][8 * j + 2], acc[mt][nt][8 * j + 3]);
              const uint32_t b0 = sg_pack_bf16(acc[mt][nt][8 * j + 4], acc[mt][nt][8 * j + 5]), b1 = sg_pack_bf16(acc[mt][nt][8 * j + 6], acc[mt][nt][8 * j + 7]);
              const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
              u32x4 out;
              out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
              __builtin_amdgcn_raw_buffer_store_b128(out, ryE, ok ? (vox[mt] * (uint32_t)cout + (uint32_t)((nt0 + nt) * 32 + 16 * j + 8 * hh)) * 2u : DEAD, 0, 0);
              SG_STORE16_GUARD(out);
            }
          }
        }
      }
      if constexpr (POOL) {
        // fused downscale3d, first stage, for this kernel's tile: the wave's two M tiles are H neighbours (a lane-local
        // add), W neighbours are adjacent lanes (one DPP exchange); the D pairs (other waves) are pooled by
        // sg_downscale_sum(2,1,1) on the quarter-size tensor.  See the sliding-halo kernel's pool epilogue.
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float sp[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float t = acc[0][nt][i] + acc[1][nt][i];
            const float u = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0xB1, 0xF, 0xF, true));
            sp[i] = (t + u) * 0.25f;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t a0 = sg_pack_bf16(sp[8 * j + 0], sp[8 * j + 1]), a1 = sg_pack_bf16(sp[8 * j + 2], sp[8 * j + 3]);
            const uint32_t b0 = sg_pack_bf16(sp[8 * j + 4], sp[8 * j + 5]), b1 = sg_pack_bf16(sp[8 * j + 6], sp[8 * j + 7]);
            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            u32x4 out;
            out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
            __builtin_amdgcn_raw_buffer_store_b128(out, ryE, pvox != DEAD ? (pvox * (uint32_t)cout + (uint32_t)((nt0 + nt) * 32 + 16 * j + 8 * hh)) * 2u : DEAD, 0, 0);
            SG_STORE16_GUARD(out);
          }
        }
      }
      ++kE;
      if (kE < kmine) enter_tile_E();
      request_mask();
      init_acc();
    }
    if (qn < items_mine) {   // S moves on to my item qn + 1
      if (++cgS == ncg) {
        cgS = 0;
        ++kS;
        if (kS < kmine) enter_tile_S();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // halo chunk and slab half have landed
  };

  // group 0: [MFMA(q) | barrier | off: halo of item q+1, second half of slab q+1, epilogue if item q closed a tile | barrier]
  // group 1: [off: halo of item q, first half of slab q+1, epilogue if item q-1 closed a tile | barrier | MFMA(q) | barrier]
  if (grp == 0) {
    init_acc();
    // the cursor points at the item whose halo is staged NEXT: item 0 is already there
    if (items_mine > 0 && ++cgS == ncg) { cgS = 0; ++kS; if (kS < kmine) enter_tile_S(); }
    for (int q = 0; q < items_max; ++q) {
      stamp();
      if (q < items_mine) mfma_phase(q);
      stamp();
      __syncthreads();
      stamp();
      off_phase(q < items_mine && (q % ncg) == ncg - 1, q + 1, q + 1, HF, NFRAG);
      stamp();
      __syncthreads();
    }
  } else {
    init_acc();
    for (int q = 0; q < items_max; ++q) {
      stamp();
      off_phase(q > 0 && q - 1 < items_mine && ((q - 1) % ncg) == ncg - 1, q, q + 1, 0, HF);
      stamp();
      __syncthreads();
      stamp();
      if (q < items_mine) mfma_phase(q);
      stamp();
      __syncthreads();
    }
    if (items_mine > 0 && items_mine == items_max) off_phase(true, items_mine, items_max, 0, 0);   // my last tile
    else if (items_mine > 0) { /* closed inside the loop: group 0 had more items */ }
  }
}

template <int EPI>
static int launch_fwd5_inst(const ConvFwdArgs& a, unsigned gx, unsigned gy, size_t lds, hipStream_t st) {
  auto kern = conv_fwd5_kernel<EPI>;
  SG_ALLOW_160K_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(512), lds, st, a);
  return SG_OK;
}

static int launch_fwd5(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  if (s->upsample_in || s->kd != 3 || s->kh != 3 || s->kw != 3 || a.os == 2 || a.tap_d || a.tap_h || a.tap_w) return SG_OK;
  if ((s->cin % 16) != 0 || (s->cout % 64) != 0 || a.pool == 1) return SG_OK;
  if (a.pool && ((s->h | s->w) & 1 || a.pixel_norm || a.mask_bits)) return SG_OK;   // whole 1 x 2 x 2 blocks, plain epilogue
  if (a.pixel_norm && (a.mask_bits || a.ntile != 2)) return SG_OK;   // pixel norm needs all channels in one block
  if (a.mask_bits && a.sign_out) return SG_OK;
  a.g = sg_make_geom(s, 256, /*prefer_w32=*/true);
  const sg_tile_geom& g = a.g;
  if (g.TN != 1 || g.TW != 32 || (g.TH & 1) || g.TD * g.TH * g.TW != 256 || g.HD > 127 || g.HH > 127 || g.HW > 127) return SG_OK;
  const int64_t ntiles = (int64_t)g.nTn * g.nTd * g.nTh * g.nTw;
  const int ny = a.ntile / 2;
  int gx = (256 / ny) / 8 * 8;
  if (gx < 8) gx = 8;
  if (ntiles >= (1 << 24) || ntiles < 2 * gx) return SG_OK;
  {   // buffer addressing, rebased per sample
    const int64_t svox = (int64_t)s->d * s->h * s->w;
    if (svox * s->cin * 2 >= (1ll << 31) || svox * s->cout * 2 >= (1ll << 31) || svox * a.ntile * 4 >= (1ll << 31)) return SG_OK;
  }
  const int hv = g.HD * g.HH * g.HW;
  if (sg_cdiv(hv * 2, 64) > 32) return SG_OK;
  a.G = 1;
  a.rs = 32;
  a.xbytes = hv * 32;
  a.wbytes = 27 * 2 * 1024;
  a.wres = 0;
  const size_t lds = 2ull * a.xbytes + 2ull * a.wbytes + 256;
  if (lds > 160 * 1024) return SG_OK;
  a.ntiles = (int)ntiles;
  a.vec_in = 1;
  a.vec_out = 1;
  const int epi = (a.sign_out ? SG_EP_SIGN : 0) | (a.mask_bits ? SG_EP_MASK : 0) | (a.pixel_norm ? SG_EP_PN : 0) |
                  (a.pool ? SG_EP_POOL : 0);
  int rc = SG_OK;
  switch (epi) {
    case 0: rc = launch_fwd5_inst<0>(a, (unsigned)gx, (unsigned)ny, lds, st); break;
    case SG_EP_POOL: rc = launch_fwd5_inst<SG_EP_POOL>(a, (unsigned)gx, (unsigned)ny, lds, st); break;
    case SG_EP_SIGN | SG_EP_POOL: rc = launch_fwd5_inst<SG_EP_SIGN | SG_EP_POOL>(a, (unsigned)gx, (unsigned)ny, lds, st); break;
    case SG_EP_SIGN: rc = launch_fwd5_inst<SG_EP_SIGN>(a, (unsigned)gx, (unsigned)ny, lds, st); break;
    case SG_EP_MASK: rc = launch_fwd5_inst<SG_EP_MASK>(a, (unsigned)gx, (unsigned)ny, lds, st); break;
    case SG_EP_PN: rc = launch_fwd5_inst<SG_EP_PN>(a, (unsigned)gx, (unsigned)ny, lds, st); break;
    case SG_EP_PN | SG_EP_SIGN: rc = launch_fwd5_inst<SG_EP_PN | SG_EP_SIGN>(a, (unsigned)gx, (unsigned)ny, lds, st); break;
    default: return SG_OK;
  }
  if (rc != SG_OK) return rc;
  SG_KNAME("conv_fwd5<bf16,2,2,3,3,3>");
  SG_LAUNCH_CHECK();
  *used = true;
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------------
// pointwise (1x1x1) convolutions with <= 4 channels on one side: from_rgb / to_rgb and their data gradients
// (pgan/generator.py:13-16, pgan/discriminator.py:9-12).  One side of the product is the whole HBM traffic
// (32-64 channels per voxel against 1), so these are streaming kernels, not GEMMs: 16 bytes per lane on the wide
// side, weights (read from the packed MFMA image, i.e. the same rounded values) in registers.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float sg_packed_w(const char* wp, int ntile, int cin_j, int cout_c) {
  constexpr int CH = sg_traits<T>::CH, ES = (int)sizeof(T);
  const int chunk = cin_j / CH, hh = (cin_j % CH) / (CH / 2), e = cin_j % (CH / 2);
  const char* frag = wp + ((int64_t)(chunk * ntile + (cout_c >> 5)) << 10);
  return sg_traits<T>::to_f(*reinterpret_cast<const T*>(frag + ((cout_c & 31) + 32 * hh) * 16 + e * ES));
}

// cin <= 4: y[v][c] = epilogue(sum_j x[v][j] * w[j][c] + b[c]); thread owns one 16-byte piece of couts (fixed).
// The stream is write-dominated (cout x the input bytes), so what matters is that nothing sits between two stores:
// CIN is a template parameter (no per-channel branches around the loads), loads are clamped instead of predicated,
// the next trip's input and mask word are requested before this trip's arithmetic, and the sign word of a 32-channel
// group is assembled by DPP exchanges within the quad instead of LDS shuffles.
template <typename T, int CIN>
__global__ __launch_bounds__(256) void pw_fwd_small_cin_kernel(ConvFwdArgs a, int64_t nvox) {
  constexpr int E = 16 / (int)sizeof(T);
  const int P = a.cout / E;                     // pieces per voxel, divides 256
  const int p = threadIdx.x % P;
  const int c0 = p * E;
  const char* wp = reinterpret_cast<const char*>(a.wp);
  float w[CIN][E], b[E];
#pragma unroll
  for (int j = 0; j < CIN; ++j)
#pragma unroll
    for (int e = 0; e < E; ++e) w[j][e] = sg_packed_w<T>(wp, a.ntile, j, c0 + e);
#pragma unroll
  for (int e = 0; e < E; ++e) b[e] = a.bias ? a.bias[c0 + e] : 0.f;
  const T* x = reinterpret_cast<const T*>(a.x);
  T* y = reinterpret_cast<T*>(a.y);
  const int rows = 256 / P;
  const int tpw = P < 32 / E ? P : 32 / E;      // threads per 32-channel sign word (adjacent lanes; powers of two)
  const int64_t nv_pad = (nvox + rows - 1) / rows * rows;   // whole waves stay in the loop for the exchanges
  const int64_t stride = (int64_t)gridDim.x * rows;
  const float slope = a.act ? a.slope : 1.f;    // max(t, 1 * t) = t
  const bool masked = a.mask_bits != nullptr, signs = a.sign_out != nullptr;
  const int wsel = c0 >> 5, wsh = c0 & 31, psh = (p % tpw) * E;
  auto fetch = [&](int64_t v, float (&xv)[CIN], uint32_t& mw) {
    const int64_t vc = v < nvox ? v : nvox - 1;               // clamped: dead lanes read the last voxel, store nothing
#pragma unroll
    for (int j = 0; j < CIN; ++j) xv[j] = sg_traits<T>::to_f(x[vc * CIN + j]);
    mw = masked ? a.mask_bits[vc * a.ntile + wsel] >> wsh : 0u;
  };
  int64_t v = (int64_t)blockIdx.x * rows + threadIdx.x / P;
  float xn[CIN];
  uint32_t mwn = 0u;
  if (v < nv_pad) fetch(v, xn, mwn);
  for (; v < nv_pad; v += stride) {
    float xv[CIN];
#pragma unroll
    for (int j = 0; j < CIN; ++j) xv[j] = xn[j];
    const uint32_t mw = mwn;
    if (v + stride < nv_pad) fetch(v + stride, xn, mwn);
    const bool live = v < nvox;
    float o[E];
    uint32_t neg = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float t = b[e];
#pragma unroll
      for (int j = 0; j < CIN; ++j) t = fmaf(xv[j], w[j][e], t);
      t = fmaxf(t, t * slope);
      neg |= (t < 0.f ? 1u : 0u) << e;
      o[e] = t;
    }
    if (signs) {                                // the TPW threads of a word are adjacent lanes
      uint32_t wbits = neg << psh;
      if (tpw == 4) {
        wbits |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wbits, 0xB1, 0xF, 0xF, true);   // lane ^ 1
        wbits |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wbits, 0x4E, 0xF, 0xF, true);   // lane ^ 2
      } else {
        for (int sh = 1; sh < tpw; sh <<= 1) wbits |= (uint32_t)__shfl_xor((int)wbits, sh);
      }
      if (live && p % tpw == 0) a.sign_out[v * a.ntile + wsel] = wbits;
    }
    if (!live) continue;
    if (masked) {
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = ((mw >> e) & 1u) ? o[e] * a.mask_slope : o[e];
    }
    u32x4 raw;
    T* t = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int e = 0; e < E; ++e) t[e] = sg_traits<T>::from_f(o[e]);
    __builtin_nontemporal_store(raw, reinterpret_cast<u32x4*>(y + v * a.cout + c0));   // write-dominated stream
    SG_STORE16_GUARD(raw);
  }
}

// cout <= 4: y[v][c] = epilogue(sum_k x[v][k] * w[k][c] + b[c]); the P = cin/E threads of a voxel are adjacent lanes
template <typename T>
__global__ __launch_bounds__(256) void pw_fwd_small_cout_kernel(ConvFwdArgs a, int64_t nvox) {
  constexpr int E = 16 / (int)sizeof(T);
  const int P = a.cin / E;                      // power of two <= 64
  const int p = threadIdx.x % P;
  const char* wp = reinterpret_cast<const char*>(a.wp);
  float w[E][4];
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int c = 0; c < 4; ++c) w[e][c] = c < a.cout ? sg_packed_w<T>(wp, a.ntile, p * E + e, c) : 0.f;
  const T* x = reinterpret_cast<const T*>(a.x);
  T* y = reinterpret_cast<T*>(a.y);
  const int rows = 256 / P;
  const int64_t nv_pad = (nvox + rows - 1) / rows * rows;   // whole waves stay in the loop for the shuffles
  // four ADJACENT voxel groups per block and trip, every load issued before the first reduction (one 16-byte load per
  // lane and trip: 4.9 TB/s; four groups a grid stride apart: slower than one)
  constexpr int U = 4;
  const int64_t nv_pad4 = (nvox + U * rows - 1) / (U * rows) * (U * rows);
  for (int64_t vb = (int64_t)blockIdx.x * U * rows + threadIdx.x / P; vb < nv_pad4; vb += (int64_t)gridDim.x * U * rows) {
    u32x4 rawu[U];
    uint32_t mwu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = vb + u * rows;
      rawu[u] = v < nvox ? *reinterpret_cast<const u32x4*>(x + v * a.cin + p * E) : u32x4{0u, 0u, 0u, 0u};
      mwu[u] = (v < nvox && p == 0 && a.mask_bits) ? a.mask_bits[v] : 0u;   // cout <= 4: one word per voxel
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int64_t v = vb + u * rows;
    if (v >= nv_pad) break;                     // uniform per block: nv_pad is a multiple of the block's rows
    const bool live = v < nvox;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    {
      const u32x4 raw = rawu[u];
      const T* t = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float xv = sg_traits<T>::to_f(t[e]);
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = fmaf(xv, w[e][c], s[c]);
      }
    }
    for (int sh = 1; sh < P; sh <<= 1)
#pragma unroll
      for (int c = 0; c < 4; ++c) s[c] += __shfl_xor(s[c], sh);
    if (live && p == 0) {
      uint32_t neg = 0u;
      const uint32_t mw = mwu[u];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < a.cout) {
          float t = s[c] + (a.bias ? a.bias[c] : 0.f);
          if (a.act) t = fmaxf(t, t * a.slope);
          neg |= (t < 0.f ? 1u : 0u) << c;
          if ((mw >> c) & 1u) t *= a.mask_slope;
          y[v * a.cout + c] = sg_traits<T>::from_f(t);
        }
      }
      if (a.sign_out != nullptr) a.sign_out[v] = neg;
    }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// dense layers on <= 128 "voxels" (the batch; blocks of 32 rows): y[v][c] = epilogue(sum_k x[v][k] w[k][c] + b[c]), e.g. D's
// 8192 -> 512 (pgan/discriminator.py:60-63).  The whole cost is streaming the weight image once: one block of 16
// waves per 32-channel output tile, wave w takes K chunks w, w+16, ... (8 loads in flight), 1-KiB weight fragments
// straight from the packed image into the MFMA A operand, partial tiles summed through LDS.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void dense_small_m_kernel(ConvFwdArgs a, int nvox) {
  __shared__ float red[16][16][64];             // [wave][acc element][lane]
  constexpr int CH = sg_traits<T>::CH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nt0 = blockIdx.x;
  const char* wp = reinterpret_cast<const char*>(a.wp) + ((int64_t)nt0 << 10) + lane * 16;
  const T* x = reinterpret_cast<const T*>(a.x);
  const int vrow = blockIdx.y * 32 + r;          // blockIdx.y: block of 32 rows (the weight slice is re-read from L2)
  const bool vlive = vrow < nvox;
  const T* xrow = x + (int64_t)(vlive ? vrow : 0) * a.cin + hh * (CH / 2);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  constexpr int U = 8;
  for (int q0 = wave; q0 < a.nchunk; q0 += 16 * U) {
    u32x4 wf[U], xf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + 16 * u;
      const bool ok = q < a.nchunk;
      wf[u] = ok ? *reinterpret_cast<const u32x4*>(wp + ((int64_t)q * a.ntile << 10)) : u32x4{0u, 0u, 0u, 0u};
      xf[u] = (ok && vlive) ? *reinterpret_cast<const u32x4*>(xrow + q * CH) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = sg_mfma_chunk<T>(wf[u], xf[u], acc);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += red[w][i][lane];
      const int co = nt0 * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (a.bias != nullptr && co < a.cout) t += a.bias[co];
      if (a.act) t = fmaxf(t, t * a.slope);
      acc[i] = t;
    }
    if (a.sign_out != nullptr) {
      const uint32_t sw = sg_sign_word(acc, hh);
      if (hh == 0 && vlive) a.sign_out[(int64_t)vrow * a.ntile + nt0] = sw;
    }
    if (a.mask_bits != nullptr && vlive) sg_apply_sign_word(acc, a.mask_bits[(int64_t)vrow * a.ntile + nt0], hh, a.mask_slope);
    if (vlive) {
      T* yrow = reinterpret_cast<T*>(a.y) + (int64_t)vrow * a.cout;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = nt0 * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (co < a.cout) yrow[co] = sg_traits<T>::from_f(acc[i]);
      }
    }
  }
}

template <typename T>
static int launch_pw_fwd(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  constexpr int E = 16 / (int)sizeof(T);
  if (a.taps != 1 || s->upsample_in || a.pixel_norm) return SG_OK;
  const int64_t nvox = (int64_t)s->n * s->d * s->h * s->w;
  if (nvox <= 128 && s->cin % sg_traits<T>::CH == 0 && a.nchunk >= 64 && !sg_cfg().fwd_no_dense) {
    SG_KNAME("dense_small_m<%s>", sg_tname<T>());
    hipLaunchKernelGGL(dense_small_m_kernel<T>, dim3((unsigned)a.ntile, (unsigned)((nvox + 31) / 32)), dim3(1024), 0, st, a, (int)nvox);
    SG_LAUNCH_CHECK();
    *used = true;
    return SG_OK;
  }
  if (s->cin <= 4 && s->cout % E == 0 && 256 % (s->cout / E) == 0 && s->cout / E <= 256 &&
      ((s->cout / E) & (s->cout / E - 1)) == 0) {
    const int rows = 256 / (s->cout / E);
    // Grid: EIGHT trips per block (never fewer blocks than the 2048 a full chip holds).  This write stream is sensitive to
    // how far apart the blocks in flight write: at 32 x 128 x 128, n32, 8192 blocks of 32 trips ran at 4.4 TB/s, 2048 of 128
    // (all resident, in lockstep) at 5.1-5.6, 32768 of 8 at 6.1-6.2, 131072 of 2 at 4.0 (tools/archive/pw_probe.py) -- blocks that
    // live for a few trips are dispatched in order and sweep memory nearly sequentially.
    const int64_t full = (nvox + rows - 1) / rows;
    int64_t nb = (full + 7) / 8;
    if (nb < 2048) nb = full < 2048 ? full : 2048;
    SG_KNAME("pw_fwd_small_cin<%s>", sg_tname<T>());
    switch (s->cin) {
      case 1: hipLaunchKernelGGL((pw_fwd_small_cin_kernel<T, 1>), dim3((unsigned)nb), dim3(256), 0, st, a, nvox); break;
      case 2: hipLaunchKernelGGL((pw_fwd_small_cin_kernel<T, 2>), dim3((unsigned)nb), dim3(256), 0, st, a, nvox); break;
      case 3: hipLaunchKernelGGL((pw_fwd_small_cin_kernel<T, 3>), dim3((unsigned)nb), dim3(256), 0, st, a, nvox); break;
      default: hipLaunchKernelGGL((pw_fwd_small_cin_kernel<T, 4>), dim3((unsigned)nb), dim3(256), 0, st, a, nvox); break;
    }
    SG_LAUNCH_CHECK();
    *used = true;
  } else if (s->cout <= 4 && s->cin % E == 0 && (s->cin / E) <= 64 && ((s->cin / E) & (s->cin / E - 1)) == 0) {
    const int rows = 256 / (s->cin / E);
    int64_t nb = (nvox + rows - 1) / rows;
    if (nb > 8192) nb = 8192;
    SG_KNAME("pw_fwd_small_cout<%s>", sg_tname<T>());
    hipLaunchKernelGGL(pw_fwd_small_cout_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, a, nvox);
    SG_LAUNCH_CHECK();
    *used = true;
  }
  return SG_OK;
}

unsigned long long* g_dbg_ts = nullptr;   // in-kernel phase stamps (tools/ts_conv.py, tools/ts_wgrad.py); also read by wgrad.hip
extern "C" __attribute__((visibility("default"))) void sg_debug_set_ts_buffer(void* p) { g_dbg_ts = (unsigned long long*)p; }

extern "C" size_t sg_conv3d_pw_epilogue_workspace(void) { return (size_t)SG_PW_PART_ROWS * 64 * sizeof(float); }

extern "C" size_t sg_conv3d_fwd_workspace(const sg_conv_shape* s, sg_dtype dt) {
  if (!conv_shape_ok(s) || dt != SG_BF16) return 0;
  if (!sg_cfg().no_gemm && sg_gemm_conv_eligible(s, dt)) return sg_gemm_conv_workspace(s, dt);   // K-split partial tiles
  if (s->kd == 3 && s->kh == 3 && s->kw == 3 && s->cin == 64 && s->cout == 32 && s->w % 32 == 0 && s->d >= 4 &&
      (!s->upsample_in || ((s->d | s->h | s->w) & 1) == 0))
    return (size_t)s->n * s->d * s->h * s->w * (size_t)s->cout * 4;
  return 0;
}

extern "C" int sg_conv3d_fwd(const void* x, const void* wp, void* y, const sg_conv_shape* s,
                             const sg_conv_epilogue* ep, sg_dtype dt, sg_stream_t st) {
  if (!conv_shape_ok(s) || !x || !wp || (!y && !(ep && ep->pw_x))) return SG_EINVAL;
  if (ep && ep->struct_size != (uint32_t)sizeof(sg_conv_epilogue)) return SG_EINVAL;   // caller built against another header
  if (!sg_aligned16(x) || !sg_aligned16(wp) || !sg_aligned16(y)) return SG_EALIGN;
  ConvFwdArgs a;
  a.x = x; a.wp = wp; a.y = y;
  a.bias = ep ? ep->bias : nullptr;
  a.pn_scale = ep ? ep->pn_scale : nullptr;
  a.mask_bits = ep ? reinterpret_cast<const uint32_t*>(ep->mask_bits) : nullptr;
  a.sign_out = ep ? reinterpret_cast<uint32_t*>(ep->sign_out) : nullptr;
  a.mask_slope = ep ? ep->mask_slope : 0.f;
  a.act = ep ? ep->act : 0;
  a.slope = ep ? ep->slope : 0.f;
  a.pixel_norm = ep ? ep->pixel_norm : 0;
  a.eps = ep ? ep->eps : 0.f;
  a.tap_d = ep ? ep->tap_off[0] : 0; a.tap_h = ep ? ep->tap_off[1] : 0; a.tap_w = ep ? ep->tap_off[2] : 0;
  a.pool = ep ? ep->pool : 0;
  if (a.pool < 0 || a.pool > 3) return SG_EINVAL;
  a.pnb_y = ep ? ep->pn_bwd_y : nullptr;
  a.pnb_scale = ep ? ep->pn_bwd_scale : nullptr;
  if ((a.pnb_y != nullptr) != (a.pnb_scale != nullptr) || (a.pnb_y && !sg_aligned16(a.pnb_y))) return SG_EINVAL;
  a.xcs = s->cin; a.xco = 0; a.addend = nullptr;
  a.in_mask = ep ? reinterpret_cast<const uint32_t*>(ep->in_mask_bits) : nullptr;
  a.in_mask_slope = ep ? ep->in_mask_slope : 0.f;
  a.in_gain = ep ? ep->in_gain : 1.f;
  a.in_mask_nw = (s->cin + 31) >> 5;
  if (a.in_mask && (dt != SG_BF16 || !s->upsample_in || s->kd != 3 || s->kh != 3 || s->kw != 3 || s->cin != 64 || s->cout != 32 ||
                    !sg_aligned16(a.in_mask) || !sg_is_pow2f(a.in_gain)))
    return a.in_mask && !sg_aligned16(a.in_mask) ? SG_EALIGN : SG_EUNSUPPORTED;     // only the two-pass 64 -> 32 path masks its gather;
                                                                                   // the gain must be a power of two (sg_mask_piece_bf16)
  a.rgb_w = ep ? ep->rgb_w : nullptr; a.rgb_bias = ep ? ep->rgb_bias : nullptr; a.rgb_out = ep ? ep->rgb_out : nullptr;
  a.pw_x = ep ? ep->pw_x : nullptr; a.pw_wmat = ep ? ep->pw_wmat : nullptr; a.pw_dx = ep ? ep->pw_dx : nullptr;
  a.pw_part = (ep && ep->pw_x) ? reinterpret_cast<float*>(ep->workspace) : nullptr;
  if (a.rgb_out && (!a.rgb_w || !sg_aligned16(a.rgb_out))) return SG_EINVAL;
  if (a.pw_x && (!a.pw_wmat || !ep->workspace || ep->workspace_bytes < sg_conv3d_pw_epilogue_workspace() || !sg_aligned16(ep->workspace)))
    return SG_EINVAL;
  if ((a.rgb_out || a.pw_x) && (dt != SG_BF16 || s->kd != 3 || s->kh != 3 || s->kw != 3 || s->cin != 32 || s->cout != 32 || s->upsample_in ||
                                !sg_cfg().fwd3s_16 || sg_cfg().fwd_no_v3 || sg_cfg().fwd_no_v3s || sg_cfg().fwd_v1 || ep->x_plane_channels))
    return SG_EUNSUPPORTED;      // these epilogues exist in conv_fwd3w only
  a.os = (ep && ep->out_scale == 2) ? 2 : 1;
  a.oa = ep ? ep->out_off[0] : 0; a.ob = ep ? ep->out_off[1] : 0; a.oc = ep ? ep->out_off[2] : 0;
  const bool subpixel = a.os == 2 || a.tap_d || a.tap_h || a.tap_w || s->kd == 2;
  if (subpixel && ((a.tap_d | a.tap_h | a.tap_w | a.oa | a.ob | a.oc) & ~1)) return SG_EINVAL;
  a.dbg = g_dbg_ts;
  a.dbg_flags = sg_cfg().dbg_flags;
  a.cin = s->cin; a.cout = s->cout;
  a.taps = s->kd * s->kh * s->kw; a.kh = s->kh; a.kw = s->kw;
  a.nchunk = conv_nchunk(s, dt);
  a.ntile = conv_ntile(s);
  if (a.pixel_norm && a.ntile > 4) return SG_EINVAL;
  sg_prof_scope prof(0, s, dt, sg_st(st));
  int rc;
  hipStream_t hs = sg_st(st);
  const int es = dt == SG_BF16 ? 2 : 4;
  if (subpixel) {   // one sub-pixel parity class: only the streamed ping-pong kernel implements the scatter epilogue
    const int es_ = dt == SG_BF16 ? 2 : 4;
    bool used = false;
    rc = SG_OK;
    if (s->kd == 2 && s->kh == 2 && s->kw == 2 && (s->cin * es_) % 16 == 0 && (!a.pixel_norm || a.ntile <= 2)) {
      const bool n1 = a.ntile == 1;
      if (dt == SG_BF16) rc = n1 ? launch_fwd4<bf16_t, 2, 1, 2, 2, 2>(a, s, hs, &used) : launch_fwd4<bf16_t, 2, 2, 2, 2, 2>(a, s, hs, &used);
      else rc = n1 ? launch_fwd4<float, 2, 1, 2, 2, 2>(a, s, hs, &used) : launch_fwd4<float, 2, 2, 2, 2, 2>(a, s, hs, &used);
    }
    if (rc == SG_OK && !used) rc = SG_EUNSUPPORTED;
    prof.done(rc);
    return rc;
  }
  if (a.pool && (subpixel || dt != SG_BF16 || sg_cfg().fwd_v1 || (s->cin * 2) % 16 != 0 || a.pixel_norm ||
                 (a.pool == 1 && (sg_cfg().fwd_no_v3 || sg_cfg().fwd_no_v3s)) || (a.pool == 2 && sg_cfg().fwd_no_v5) ||
                 (a.pool == 3 && (s->cin != 32 || s->kd != 3 || s->kh != 3 || s->kw != 3 || s->upsample_in || !sg_cfg().fwd3s_16 ||
                                  sg_cfg().fwd_no_v3 || sg_cfg().fwd_no_v3s || (ep && ep->x_plane_channels)))))
    return SG_EUNSUPPORTED;
  // the pixel-norm backward epilogue exists in the sliding-halo kernel only
  if (a.pnb_y && (subpixel || dt != SG_BF16 || sg_cfg().fwd_v1 || sg_cfg().fwd_no_v3 || sg_cfg().fwd_no_v3s || a.pool ||
                  a.pixel_norm || (s->cin * 2) % 16 != 0 || s->kd != 3 || s->kh != 3 || s->kw != 3 || s->upsample_in ||
                  (ep && ep->x_plane_channels)))
    return SG_EUNSUPPORTED;
  if (sg_small_eligible(s) && !sg_cfg().no_small && !a.pool && !a.pnb_y && !(ep && ep->x_plane_channels)) {
    // 4 / 8 / 16-channel 1x3x3 layers (the 2-D pgan's top levels): bandwidth-bound VALU kernel, small.hip
    rc = sg_small_fwd(x, reinterpret_cast<const char*>(wp) + (size_t)a.nchunk * a.taps * a.ntile * 1024, y, s, a.bias, a.act, a.slope,
                      a.pixel_norm, a.eps, a.pn_scale, a.mask_bits, a.mask_slope, a.sign_out, dt, hs);
    prof.done(rc);
    return rc;
  }
  if (dt == SG_BF16 && !sg_cfg().no_gemm && !a.pool && !a.pnb_y && !a.pixel_norm && !(ep && ep->x_plane_channels) &&
      sg_gemm_conv_eligible(s, dt)) {   // low-resolution levels: 256-voxel x 128-channel GEMM tiles over the folded batch (gemm.hip)
    bool used = false;
    rc = sg_gemm_conv_fwd(x, wp, y, s, a.bias, a.act, a.slope, a.mask_bits, a.mask_slope, a.sign_out, ep ? ep->workspace : nullptr,
                          ep ? ep->workspace_bytes : 0, hs, &used);
    if (rc != SG_OK || used) { prof.done(rc); return rc; }
  }
  if (a.pool == 2) {   // H x W pooling: the streamed kernel's tile (two H rows per wave)
    bool used = false;
    rc = launch_fwd5(a, s, hs, &used);
    if (rc == SG_OK && !used) rc = SG_EUNSUPPORTED;
    prof.done(rc);
    return rc;
  }
  if (!sg_cfg().fwd_no_pw && !a.pool && !a.pnb_y) {   // streaming kernels for the 1x1x1 layers with <= 4 channels on one side
    bool used = false;
    rc = dt == SG_BF16 ? launch_pw_fwd<bf16_t>(a, s, hs, &used) : launch_pw_fwd<float>(a, s, hs, &used);
    if (rc != SG_OK || used) { prof.done(rc); return rc; }
  }
  const bool v2 = ((s->cin * es) % 16 == 0) && !sg_cfg().fwd_v1;
  const int xpl = ep ? ep->x_plane_channels : 0;      // x as separate 32-channel tensors: the K-split path only
  if (xpl != 0 && (xpl != 32 || s->upsample_in || a.pool || !(v2 && !sg_cfg().fwd_no_v3 && (!a.pixel_norm || a.ntile == 1)))) {
    prof.done(xpl != 32 ? SG_EINVAL : SG_EUNSUPPORTED);
    return xpl != 32 ? SG_EINVAL : SG_EUNSUPPORTED;
  }
  if (v2 && !sg_cfg().fwd_no_v3 && (!a.pixel_norm || a.ntile == 1)) {
    bool used = false;
    rc = SG_OK;
    const bool k333 = s->kd == 3 && s->kh == 3 && s->kw == 3, k133 = s->kd == 1 && s->kh == 3 && s->kw == 3;
    if (dt == SG_BF16 && k333 && !sg_cfg().fwd_no_v3s && !sg_cfg().fwd_no_ksplit && a.nchunk == 4 && a.ntile == 1 && s->cin == 64 &&
        !a.pool && !a.pnb_y && !(a.mask_bits && a.sign_out) && !(a.pixel_norm && a.mask_bits) &&
        !(s->upsample_in && a.mask_bits && !a.in_mask) &&      // (the plain fused-gather variants carry no output mask)
        !(a.in_mask && (!s->upsample_in || a.sign_out || a.pixel_norm || xpl)) &&
        ep && ep->workspace &&
        ep->workspace_bytes >= (size_t)s->n * s->d * s->h * s->w * (size_t)s->cout * 4 && sg_aligned16(ep->workspace)) {
      // one pass with sliding accumulators (conv3p.hip) where its tile applies; else the K split below
      if (!sg_cfg().fwd_no_3p && !xpl) {
        bool u3 = false;
        rc = sg_launch_fwd3p(a, s, hs, &u3);
        if (rc != SG_OK || u3) { prof.done(rc); return rc; }
      }
      // K split: 64 input channels as two sliding-halo passes over 32 channels each (resident weights, a 64-byte
      // half of every 128-byte channel row fetched ONCE per pass) with the f32 partial sums in the caller's workspace.
      // The streamed kernel fetches every row as four 32-byte slivers in four passes too far apart for L2: each
      // sliver costs a whole line (tools/probe/sliver_probe.hip), 3.1x the algorithmic traffic, 0.27 of MFMA peak.
      ConvFwdArgs p1 = a, p2 = a;
      sg_conv_shape sh = *s;
      sh.cin = 32;
      p1.cin = p2.cin = 32; p1.nchunk = p2.nchunk = 2; p1.xcs = p2.xcs = 64;
      p1.xco = 0; p1.y = ep->workspace; p1.bias = nullptr; p1.act = 0; p1.mask_bits = nullptr; p1.sign_out = nullptr;
      p1.pixel_norm = 0; p1.pn_scale = nullptr;
      p2.xco = 32; p2.addend = reinterpret_cast<const float*>(ep->workspace);
      if (xpl) {   // the halves are two tensors of 32 channels: whole rows
        p1.xcs = p2.xcs = 32;
        p2.xco = 0;
        p2.x = reinterpret_cast<const char*>(a.x) + (size_t)s->n * s->d * s->h * s->w * 64;
      }
      p2.wp = reinterpret_cast<const char*>(a.wp) + (size_t)2 * a.taps * a.ntile * 1024;   // chunks 2, 3 of the packed image
      bool u1 = false, u2 = false;
      rc = launch_fwd3s<2, 1>(p1, &sh, hs, &u1);
      if (rc == SG_OK && u1) {
        rc = launch_fwd3s<2, 2>(p2, &sh, hs, &u2);
        if (rc == SG_OK && !u2) rc = SG_EINVAL;     // same geometry as pass 1: cannot decline
        SG_KNAME("conv_fwd3s<bf16,2> x2 (K split)");
        prof.done(rc);
        return rc;
      }
      if (rc != SG_OK) { prof.done(rc); return rc; }
    }
    if (xpl || a.in_mask) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }
    if (dt == SG_BF16 && k333 && !sg_cfg().fwd_no_v3s) {   // sliding-halo variant where its tile fits
      if (a.nchunk == 2 && sg_cfg().fwd3s_16) {              // ... as wave-private planes + sliding accumulators on 16x16x32 (conv3w.hip)
        int pw_rows = 0;
        rc = sg_launch_fwd3w(a, s, hs, &used, &pw_rows);
        if (rc == SG_OK && used && a.pw_x)      // the per-wave rows of the fused from_rgb backward, added in order
          rc = sg_pw_wgrad_finalize(a.pw_part, pw_rows, ep->pw_dw, ep->pw_dbias, ep->pw_coef, hs);
        if (rc != SG_OK || used) { prof.done(rc); return rc; }
      }
      if (a.rgb_out || a.pw_x || a.pool == 3) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }      // (only that kernel has these epilogues)
      if (a.nchunk == 2) rc = launch_fwd3s<2>(a, s, hs, &used);
      else if (a.nchunk == 1) rc = launch_fwd3s<1>(a, s, hs, &used);
      if (rc != SG_OK || used) { prof.done(rc); return rc; }
    }
    if (a.pool || a.pnb_y) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }   // only the sliding-halo kernel has these
    if (dt == SG_BF16) {
      if (k333 && a.nchunk == 2) rc = launch_fwd3r<bf16_t, 2, 2, 3, 3, 3>(a, s, hs, &used);
      else if (k333 && a.nchunk == 1) rc = launch_fwd3r<bf16_t, 2, 1, 3, 3, 3>(a, s, hs, &used);
      else if (k133 && a.nchunk == 2) rc = launch_fwd3r<bf16_t, 2, 2, 1, 3, 3>(a, s, hs, &used);
      else if (k133 && a.nchunk == 1) rc = launch_fwd3r<bf16_t, 2, 1, 1, 3, 3>(a, s, hs, &used);
    } else {
      if (k333 && a.nchunk == 2) rc = launch_fwd3r<float, 2, 2, 3, 3, 3>(a, s, hs, &used);
      else if (k333 && a.nchunk == 1) rc = launch_fwd3r<float, 2, 1, 3, 3, 3>(a, s, hs, &used);
      else if (k133 && a.nchunk == 2) rc = launch_fwd3r<float, 2, 2, 1, 3, 3>(a, s, hs, &used);
      else if (k133 && a.nchunk == 1) rc = launch_fwd3r<float, 2, 1, 1, 3, 3>(a, s, hs, &used);
    }
    if (rc != SG_OK || used) { prof.done(rc); return rc; }
  }
  if (a.in_mask) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }
  if (v2 && dt == SG_BF16 && !sg_cfg().fwd_no_v5) {   // the streamed kernel's hot class, lean off-phase
    bool used = false;
    rc = launch_fwd5(a, s, hs, &used);
    if (rc != SG_OK || used) { prof.done(rc); return rc; }
  }
  if (v2 && !sg_cfg().fwd_no_v4 && (!a.pixel_norm || a.ntile <= 2)) {
    bool used = false;
    rc = SG_OK;
    const bool k333 = s->kd == 3 && s->kh == 3 && s->kw == 3, k133 = s->kd == 1 && s->kh == 3 && s->kw == 3;
    // two output-channel tiles per wave need the shared address table (TW = 32, even TH); other tiles run one
    // 32-channel slice per block (pixel-norm over 64 channels then falls through to the v2 kernels)
    const sg_tile_geom g4 = sg_make_geom(s, 256, /*prefer_w32=*/true);
    const bool n1 = a.ntile == 1 || (!(g4.TW == 32 && (g4.TH & 1) == 0) && !a.pixel_norm);
    // 16-wide tile rows (the 16^2 level): two output-channel tiles per wave through the address table with a two-row shift
    if (dt == SG_BF16 && k333 && a.ntile >= 2 && g4.TW == 16 && (g4.TH & 3) == 0 && !a.pixel_norm && !sg_cfg().fwd4_no_lean) {
      rc = launch_fwd4<bf16_t, 2, 2, 3, 3, 3, 2>(a, s, hs, &used);
      if (rc != SG_OK || used) { prof.done(rc); return rc; }
    }
    // (A 4 x 4 x 32 tile with four M tiles per wave was measured for the one-slice deep-K layers, 64 -> 32 at 128^2:
    // 682 against 732 TFLOP/s.  It doubles the MFMAs per staged halo chunk but no longer fits next to resident weights,
    // so the streamed slab halves put back the bytes per phase it saved: those layers are bound by the LDS-DMA issue ->
    // landed latency of one phase's staging, which only a two-phase-deep halo pipeline -- LDS the kernel does not have
    // -- would hide.)
    if (dt == SG_BF16) {
      if (k333) rc = n1 ? launch_fwd4<bf16_t, 2, 1, 3, 3, 3>(a, s, hs, &used) : launch_fwd4<bf16_t, 2, 2, 3, 3, 3>(a, s, hs, &used);
      else if (k133) rc = n1 ? launch_fwd4<bf16_t, 2, 1, 1, 3, 3>(a, s, hs, &used) : launch_fwd4<bf16_t, 2, 2, 1, 3, 3>(a, s, hs, &used);
    } else {
      if (k333) rc = n1 ? launch_fwd4<float, 2, 1, 3, 3, 3>(a, s, hs, &used) : launch_fwd4<float, 2, 2, 3, 3, 3>(a, s, hs, &used);
      else if (k133) rc = n1 ? launch_fwd4<float, 2, 1, 1, 3, 3>(a, s, hs, &used) : launch_fwd4<float, 2, 2, 1, 3, 3>(a, s, hs, &used);
    }
    if (rc != SG_OK || used) { prof.done(rc); return rc; }
  }
  // v2 / v1: pick the largest register tile (MTW x NTB) that still fills the chip; low-resolution layers with few
  // voxels get 128-voxel tiles and single 32-channel output slices so that the grid has enough blocks.
  {
    const int cand[5][2] = {{2, 4}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
    int best = -1;
    int64_t best_blocks = -1;
    for (int i = 0; i < 5; ++i) {
      const int mtw = cand[i][0], ntb = cand[i][1];
      if (ntb > 1 && ntb / 2 >= a.ntile) continue;               // wider than the layer
      if (a.pixel_norm && ntb < a.ntile) continue;                // pixel-norm needs every channel in the block
      if (!v2 && mtw == 1) continue;
      sg_tile_geom gg = sg_make_geom(s, mtw * 128, v2);
      const int64_t blocks = (int64_t)gg.nTn * gg.nTd * gg.nTh * gg.nTw * sg_cdiv(a.ntile, ntb);
      if (blocks >= 384) { best = i; break; }
      if (blocks > best_blocks) { best_blocks = blocks; best = i; }
    }
    if (best < 0) best = 0;
    const int mtw = cand[best][0], ntb = cand[best][1];
    // (2 x 4 register tile with 4 chunks per step would need 320 VGPRs: it stays at 2 chunks per step)
    const int gc = !v2 ? 0 : (a.nchunk >= 4 && a.taps <= 9 && !(mtw == 2 && ntb == 4) ? 4 : (a.nchunk >= 2 ? 2 : 1));
#define SG_FWD2(T, M, N)                                                                                   \
  (gc == 4 ? launch_fwd2<T, M, (M == 2 && N == 4 ? 2 : N), 4>(a, s, hs)                                     \
           : gc == 2 ? launch_fwd2<T, M, N, 2>(a, s, hs) : launch_fwd2<T, M, N, 1>(a, s, hs))
#define SG_FWD_PICK(T)                                                                                     \
  do {                                                                                                     \
    if (!v2) {                                                                                             \
      rc = ntb == 4 ? launch_fwd<T, 2, 4>(a, s, hs) : ntb == 2 ? launch_fwd<T, 2, 2>(a, s, hs)              \
                                                              : launch_fwd<T, 2, 1>(a, s, hs);              \
    } else if (mtw == 2) {                                                                                 \
      rc = ntb == 4 ? SG_FWD2(T, 2, 4) : ntb == 2 ? SG_FWD2(T, 2, 2) : SG_FWD2(T, 2, 1);                    \
    } else {                                                                                               \
      rc = ntb == 2 ? SG_FWD2(T, 1, 2) : SG_FWD2(T, 1, 1);                                                  \
    }                                                                                                      \
  } while (0)
    if (dt == SG_BF16) SG_FWD_PICK(bf16_t);
    else if (dt == SG_F32) SG_FWD_PICK(float);
    else rc = SG_EINVAL;
#undef SG_FWD_PICK
#undef SG_FWD2
  }
  prof.done(rc);
  return rc;
}
