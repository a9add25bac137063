// Weight gradient of the stride-1 'SAME' 3-D convolution on MFMA (gfx950): replaces what tf.gradients derives
// for tf.nn.conv3d at SURFGAN_3D/networks/ops.py:150 (Conv3DBackpropFilterV2):
//     dw[tap][ci][co] = sum_{voxels v} x[v + tap - pad][ci] * dy[v][co]
// GEMM view per tap: M = ci (32-row tile), N = co (32-col tile), K = voxels.  A block owns one (ci tile, co tile)
// pair and sweeps spatial tiles persistently; the x halo and the dy tile are staged in LDS as [voxel][32 ch].
// Both MFMA operands need K (= voxel) along the fragment's register axis while memory has channels
// contiguous, so the bf16 path reads them with the gfx950 transposing LDS read (ds_read_b64_tr_b16); the
// f32 path (32x32x2, one element per lane) needs no transpose.  Each of the 4 waves keeps the accumulators of
// up to 7 taps (27 taps / 4 waves); partial sums of different blocks are combined with f32 atomics into a
// tile-ordered workspace, then scaled and re-laid out to DHWIO by a finalize kernel.
#include "common.h"
#include "prof.h"

struct WgradArgs {
  const void* x;
  const void* dy;
  float* dwt;  // [taps][ciT][coT][32][32]
  sg_tile_geom g;   // x halo geometry
  sg_tile_geom gy;  // same tiles, no halo (dy image)
  int cin, cout, taps, kh, kw, tap0, taps_blk;
  int ciT, coT;
  int rs;  // LDS row stride (bytes), same for both images
  int xbytes, ybytes;
  int ntiles;
  sg_fastdiv fnp;
  int vec_x, vec_y;
};

constexpr int WG_MAXT = 7;  // taps per wave

template <typename T, int BM>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = (int)sizeof(T);
  const sg_tile_geom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  char* xlds = smem;
  char* ylds = smem + a.xbytes;
  int* rowtab = reinterpret_cast<int*>(smem + a.xbytes + a.ybytes);  // [BM] halo row byte offset of tile voxel m
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* dy = reinterpret_cast<const T*>(a.dy);
  const int ci_t = blockIdx.y / a.coT, co_t = blockIdx.y % a.coT;
  const int tvox = g.TN * g.TD * g.TH * g.TW;

  for (int m = tid; m < BM; m += 256) {
    int off = 0;
    if (m < tvox) {
      uint32_t q = sg_div((uint32_t)m, g.fTW);
      int tw = m - (int)q * g.TW;
      uint32_t q2 = sg_div(q, g.fTH);
      int th = (int)(q - q2 * g.TH);
      uint32_t q3 = sg_div(q2, g.fTD);
      int td = (int)(q2 - q3 * g.TD);
      off = ((((int)q3 * g.HD + td) * g.HH + th) * g.HW + tw) * a.rs;
    }
    rowtab[m] = off;
  }
  // dy rows in [tvox, BM) are never staged: keep them zero so that the padded K range adds nothing
  for (int i = tid; i < (BM - tvox) * (a.rs / 4); i += 256) reinterpret_cast<int*>(ylds + (size_t)tvox * a.rs)[i] = 0;

  f32x16 acc[WG_MAXT];
#pragma unroll
  for (int j = 0; j < WG_MAXT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  int tapoff[WG_MAXT];
#pragma unroll
  for (int j = 0; j < WG_MAXT; ++j) {
    const int tl = wave + 4 * j;
    int off = 0;
    if (tl < a.taps_blk) {
      const int tap = a.tap0 + tl;
      const int kw_i = tap % a.kw;
      const int q = tap / a.kw;
      off = (((q / a.kh) * g.HH + (q % a.kh)) * g.HW + kw_i) * a.rs;
    }
    tapoff[j] = off;
  }

  const int kend = (tvox + 15) & ~15;
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const sg_tile_origin o = sg_tile_of(g, (uint32_t)t);
    __syncthreads();
    sg_stage_halo<T>(xlds, a.rs, x, g, o, a.cin, ci_t * 32, 32 * ES / 16, a.fnp, a.vec_x != 0, tid, 256);
    sg_stage_halo<T>(ylds, a.rs, dy, a.gy, o, a.cout, co_t * 32, 32 * ES / 16, a.fnp, a.vec_y != 0, tid, 256);
    __syncthreads();
    if constexpr (sizeof(T) == 2) {
      // transposing reads: 16-lane group grp, lane (q, p) supplies row q, columns 4p..4p+3 of a 4 x 16 block
      const int i16 = lane & 15, grp = lane >> 4;
      const int qd = i16 >> 2, pp = i16 & 3;
      const int colb = (16 * (grp & 1) + 4 * pp) * 2;
      const int kb = 8 * (grp >> 1) + qd;
      for (int k0 = 0; k0 < kend; k0 += 16) {
        const int m0 = k0 + kb, m1 = k0 + kb + 4;
        const int xr0 = rowtab[m0] + colb, xr1 = rowtab[m1] + colb;
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(ylds + m0 * a.rs + colb));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(ylds + m1 * a.rs + colb));
        u32x4 bf;
        bf[0] = __builtin_bit_cast(u32x2, b0)[0]; bf[1] = __builtin_bit_cast(u32x2, b0)[1];
        bf[2] = __builtin_bit_cast(u32x2, b1)[0]; bf[3] = __builtin_bit_cast(u32x2, b1)[1];
#pragma unroll
        for (int j = 0; j < WG_MAXT; ++j) {
          if (wave + 4 * j < a.taps_blk) {
            s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(xlds + xr0 + tapoff[j]));
            s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(xlds + xr1 + tapoff[j]));
            u32x4 af;
            af[0] = __builtin_bit_cast(u32x2, a0)[0]; af[1] = __builtin_bit_cast(u32x2, a0)[1];
            af[2] = __builtin_bit_cast(u32x2, a1)[0]; af[3] = __builtin_bit_cast(u32x2, a1)[1];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af),
                                                            __builtin_bit_cast(bf16x8, bf), acc[j], 0, 0, 0);
          }
        }
      }
    } else {
      for (int k0 = 0; k0 < kend; k0 += 2) {
        const int m = k0 + hh;
        const float bv = *reinterpret_cast<const float*>(ylds + m * a.rs + r * 4);
        const int xr = rowtab[m] + r * 4;
#pragma unroll
        for (int j = 0; j < WG_MAXT; ++j) {
          if (wave + 4 * j < a.taps_blk) {
            const float av = *reinterpret_cast<const float*>(xlds + xr + tapoff[j]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }
  // D[row = ci][col = co]: lane holds col r, rows (i&3) + 8*(i>>2) + 4*hh
#pragma unroll
  for (int j = 0; j < WG_MAXT; ++j) {
    const int tl = wave + 4 * j;
    if (tl < a.taps_blk) {
      float* dst = a.dwt + ((((int64_t)(a.tap0 + tl) * a.ciT + ci_t) * a.coT + co_t) << 10);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
        unsafeAtomicAdd(dst + row * 32 + r, acc[j][i]);
      }
    }
  }
}

__global__ void wgrad_finalize_kernel(const float* __restrict__ dwt, float* __restrict__ dw, float coef, int taps,
                                      int cin, int cout, int ciT, int coT) {
  const int64_t total = (int64_t)taps * cin * cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int co = (int)(i % cout);
    int64_t q = i / cout;
    int ci = (int)(q % cin);
    int tap = (int)(q / cin);
    dw[i] = coef * dwt[((((int64_t)tap * ciT + (ci >> 5)) * coT + (co >> 5)) << 10) + (ci & 31) * 32 + (co & 31)];
  }
}

static int conv_shape_ok_w(const sg_conv_shape* s) {
  if (!s) return 0;
  if (s->n < 1 || s->d < 1 || s->h < 1 || s->w < 1 || s->cin < 1 || s->cout < 1) return 0;
  if (s->kd < 1 || s->kh < 1 || s->kw < 1 || !(s->kd & 1) || !(s->kh & 1) || !(s->kw & 1)) return 0;
  if (s->kd > 7 || s->kh > 7 || s->kw > 7) return 0;
  if (s->upsample_in && ((s->d | s->h | s->w) & 1)) return 0;
  return 1;
}

extern "C" size_t sg_conv3d_wgrad_workspace(const sg_conv_shape* s, sg_dtype dt) {
  (void)dt;
  if (!conv_shape_ok_w(s)) return 0;
  return (size_t)(s->kd * s->kh * s->kw) * sg_cdiv(s->cin, 32) * sg_cdiv(s->cout, 32) * 4096;
}

template <typename T, int BM>
static int launch_wgrad(WgradArgs& a, const sg_conv_shape* s, hipStream_t st) {
  constexpr int ES = (int)sizeof(T);
  a.g = sg_make_geom(s, BM);
  sg_conv_shape sy = *s;
  sy.kd = sy.kh = sy.kw = 1; sy.upsample_in = 0;
  a.gy = a.g;
  a.gy.PD = a.gy.PH = a.gy.PW = 0;
  a.gy.HD = a.g.TD; a.gy.HH = a.g.TH; a.gy.HW = a.g.TW;
  a.gy.fHW = sg_make_fastdiv(a.gy.HW); a.gy.fHH = sg_make_fastdiv(a.gy.HH); a.gy.fHD = sg_make_fastdiv(a.gy.HD);
  a.gy.ups = 0;
  const sg_tile_geom& g = a.g;
  const int64_t ntiles = (int64_t)g.nTn * g.nTd * g.nTh * g.nTw;
  if (ntiles >= (1 << 24)) return SG_EINVAL;
  a.ntiles = (int)ntiles;
  a.rs = 32 * ES + 16;
  const int hv = g.TN * g.HD * g.HH * g.HW;
  a.xbytes = (hv * a.rs + 15) & ~15;
  a.ybytes = BM * a.rs;
  a.fnp = sg_make_fastdiv(32 * ES / 16);
  a.vec_x = ((s->cin * ES) % 16 == 0) ? 1 : 0;
  a.vec_y = ((s->cout * ES) % 16 == 0) ? 1 : 0;
  const size_t lds = (size_t)a.xbytes + a.ybytes + BM * 4;
  if (lds > 160 * 1024) return SG_EINVAL;
  auto kern = conv_wgrad_kernel<T, BM>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int pairs = a.ciT * a.coT;
  int P = sg_cdiv(768, pairs);
  if (P > a.ntiles) P = a.ntiles;
  if (P < 1) P = 1;
  for (int tap0 = 0; tap0 < a.taps; tap0 += 4 * WG_MAXT) {
    a.tap0 = tap0;
    a.taps_blk = a.taps - tap0 < 4 * WG_MAXT ? a.taps - tap0 : 4 * WG_MAXT;
    hipLaunchKernelGGL(kern, dim3((unsigned)P, (unsigned)pairs), dim3(256), lds, st, a);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

extern "C" int sg_conv3d_wgrad(const void* x, const void* dy, float* dw, float coef, void* workspace,
                               size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st) {
  if (!conv_shape_ok_w(s) || !x || !dy || !dw || !workspace) return SG_EINVAL;
  if (!sg_aligned16(x) || !sg_aligned16(dy) || !sg_aligned16(workspace)) return SG_EALIGN;
  const size_t need = sg_conv3d_wgrad_workspace(s, dt);
  if (workspace_bytes < need) return SG_EWORKSPACE;
  hipStream_t hs = sg_st(st);
  sg_prof_scope prof(1, s, dt, hs);
  hipError_t e = hipMemsetAsync(workspace, 0, need, hs);
  if (e != hipSuccess) { prof.done((int)e); return (int)e; }
  WgradArgs a;
  a.x = x; a.dy = dy; a.dwt = reinterpret_cast<float*>(workspace);
  a.cin = s->cin; a.cout = s->cout;
  a.taps = s->kd * s->kh * s->kw; a.kh = s->kh; a.kw = s->kw;
  a.ciT = sg_cdiv(s->cin, 32); a.coT = sg_cdiv(s->cout, 32);
  int rc;
  if (dt == SG_BF16) rc = launch_wgrad<bf16_t, 256>(a, s, hs);
  else if (dt == SG_F32) rc = launch_wgrad<float, 128>(a, s, hs);
  else rc = SG_EINVAL;
  if (rc == SG_OK) {
    const int64_t total = (int64_t)a.taps * s->cin * s->cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, hs, a.dwt, dw, coef, a.taps, s->cin,
                       s->cout, a.ciT, a.coT);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) rc = (int)e2;
  }
  prof.done(rc);
  return rc;
}
