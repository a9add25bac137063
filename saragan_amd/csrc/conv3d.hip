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

static inline int conv_nchunk(const sg_conv_shape* s, sg_dtype dt) {
  int ch = dt == SG_BF16 ? 16 : 8;
  return sg_cdiv(s->cin, ch);
}
static inline int conv_ntile(const sg_conv_shape* s) { return sg_cdiv(s->cout, 32); }

static int conv_shape_ok(const sg_conv_shape* s) {
  if (!s) return 0;
  if (s->n < 1 || s->d < 1 || s->h < 1 || s->w < 1 || s->cin < 1 || s->cout < 1) return 0;
  if (s->kd < 1 || s->kh < 1 || s->kw < 1 || !(s->kd & 1) || !(s->kh & 1) || !(s->kw & 1)) return 0;
  if (s->kd > 7 || s->kh > 7 || s->kw > 7) return 0;
  if (s->upsample_in && ((s->d | s->h | s->w) & 1)) return 0;
  return 1;
}

extern "C" size_t sg_conv3d_packed_bytes(const sg_conv_shape* s, sg_dtype dt) {
  if (!conv_shape_ok(s)) return 0;
  return (size_t)conv_nchunk(s, dt) * (s->kd * s->kh * s->kw) * conv_ntile(s) * 1024;
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
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
struct ConvFwdArgs {
  const void* x;
  const void* wp;
  void* y;
  const float* bias;
  float* pn_scale;
  sg_tile_geom g;
  int cin, cout, taps, kh, kw;
  int nchunk, ntile;    // K chunks of 32 B, output-channel tiles of 32
  int G, TG;            // chunks staged per K phase, taps staged per weight phase
  int rs;               // LDS row stride of the halo image (bytes)
  int xbytes;           // LDS bytes of the halo image
  sg_fastdiv fnp;       // fastdiv by pieces per halo row (G*2)
  int vec_in, vec_out;
  int act, pixel_norm;
  float slope, eps;
};

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

  // ---- epilogue: lane owns voxel r of each M tile and couts (i&3) + 8*(i>>2) + 4*hh of each N tile ----
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
    if (ooff[mt] >= 0) {
      T* yrow = y + ooff[mt] * (int64_t)a.cout;
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
              *reinterpret_cast<u32x4*>(yrow + co) = *reinterpret_cast<u32x4*>(tmp);
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
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid((unsigned)ntiles, (unsigned)sg_cdiv(a.ntile, NTB));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_conv3d_fwd(const void* x, const void* wp, void* y, const sg_conv_shape* s,
                             const sg_conv_epilogue* ep, sg_dtype dt, sg_stream_t st) {
  if (!conv_shape_ok(s) || !x || !wp || !y) return SG_EINVAL;
  if (!sg_aligned16(x) || !sg_aligned16(wp) || !sg_aligned16(y)) return SG_EALIGN;
  ConvFwdArgs a;
  a.x = x; a.wp = wp; a.y = y;
  a.bias = ep ? ep->bias : nullptr;
  a.pn_scale = ep ? ep->pn_scale : nullptr;
  a.act = ep ? ep->act : 0;
  a.slope = ep ? ep->slope : 0.f;
  a.pixel_norm = ep ? ep->pixel_norm : 0;
  a.eps = ep ? ep->eps : 0.f;
  a.cin = s->cin; a.cout = s->cout;
  a.taps = s->kd * s->kh * s->kw; a.kh = s->kh; a.kw = s->kw;
  a.nchunk = conv_nchunk(s, dt);
  a.ntile = conv_ntile(s);
  if (a.pixel_norm && a.ntile > 4) return SG_EINVAL;
  sg_prof_scope prof(0, s, dt, sg_st(st));
  int rc;
  hipStream_t hs = sg_st(st);
  if (dt == SG_BF16) {
    if (a.ntile == 1) rc = launch_fwd<bf16_t, 2, 1>(a, s, hs);
    else if (a.ntile == 2) rc = launch_fwd<bf16_t, 2, 2>(a, s, hs);
    else rc = launch_fwd<bf16_t, 2, 4>(a, s, hs);
  } else if (dt == SG_F32) {
    if (a.ntile == 1) rc = launch_fwd<float, 2, 1>(a, s, hs);
    else if (a.ntile == 2) rc = launch_fwd<float, 2, 2>(a, s, hs);
    else rc = launch_fwd<float, 2, 4>(a, s, hs);
  } else {
    rc = SG_EINVAL;
  }
  prof.done(rc);
  return rc;
}
