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
#include <stdlib.h>
#include <type_traits>

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
  int plane_rows;  // wgrad3: LDS rows per halo plane slot
  float* dbias;    // optional [cout]: column sums of dy (bias gradient), accumulated by the ci_t == 0 blocks
  // Reproducible mode (SG_DETERMINISTIC=1): block x of the grid STORES its partial sums into slab x of the workspace
  // (dwt + x * slab, dbias + x * bslab) instead of adding them with f32 atomics; the finalize kernels add the slabs in
  // order.  slab == 0: one buffer, atomics (the sums then depend on the order in which blocks retire).
  int64_t slab, bslab;
  int nslab;       // slabs written = gridDim.x of the launch (set by the launcher)
  int dbg_flags;   // diagnostic ablations (0 in production): 1 = stage only the first items
  unsigned long long* dbg;   // in-kernel phase stamps (sg_debug_set_ts_buffer), nullptr in production
  // lean sliding-halo kernel, DYM: dy is the HALF-resolution tensor [n, D/2, H/2, W/2, cout]; the staged tile is
  // dy_gain * where(sign bit of the fine voxel, dy_mask_slope, 1) * nearest-x2(dy) (sg_conv3d_wgrad_bias_up_masked)
  const uint32_t* dy_mask;   // sign words of the fine tensor [voxel][coT]
  float dy_mask_slope, dy_gain;
};
extern unsigned long long* g_dbg_ts;

constexpr int WG_MAXT = 7;  // taps per wave

__device__ __forceinline__ void sg_wg_out(float* dst, float v, bool slab) {
  if (slab) *dst = v;
  else unsafeAtomicAdd(dst, v);
}

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
      float* dst = a.dwt + (int64_t)blockIdx.x * a.slab + ((((int64_t)(a.tap0 + tl) * a.ciT + ci_t) * a.coT + co_t) << 10);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
        sg_wg_out(dst + row * 32 + r, acc[j][i], a.slab != 0);
      }
    }
  }
}

__global__ void wgrad_finalize_kernel(float* __restrict__ dwt, float* __restrict__ dw, float coef, int taps,
                                      int cin, int cout, int ciT, int coT, int nslab, int64_t slab,
                                      float* __restrict__ bias_staged = nullptr, float* __restrict__ dbias = nullptr,
                                      int accumulate = 0, int clean = 0) {
  const int64_t total = (int64_t)taps * cin * cout;
  // the bias gradient the kernel accumulated beside the tile (one memset clears both): out to the caller's buffer
  // clean: every word read here is set back to zero -- the next call on this workspace starts without a memset
  if (bias_staged != nullptr && blockIdx.x == 0)
    for (int c = threadIdx.x; c < cout; c += blockDim.x) {
      dbias[c] = bias_staged[c];
      if (clean) bias_staged[c] = 0.f;
    }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int co = (int)(i % cout);
    int64_t q = i / cout;
    int ci = (int)(q % cin);
    int tap = (int)(q / cin);
    float* src = dwt + ((((int64_t)tap * ciT + (ci >> 5)) * coT + (co >> 5)) << 10) + (ci & 31) * 32 + (co & 31);
    float v = src[0];
    if (clean) src[0] = 0.f;
    for (int b = 1; b < nslab; ++b) v += src[(int64_t)b * slab];      // reproducible mode: the slabs in order
    const float t = __fmul_rn(coef, v);      // (accumulate: the same two roundings as a separate add of the finished gradient)
    dw[i] = accumulate ? __fadd_rn(dw[i], t) : t;
  }
  // clean: the padding rows / columns of partial 32 x 32 tiles too (some kernels add what their staging left there; nobody
  // reads it, but the next layer on this workspace may have more channels)
  if (clean && ((cin & 31) || (cout & 31))) {
    const int64_t padded = ((int64_t)taps * ciT * coT) << 10;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += (int64_t)gridDim.x * blockDim.x) {
      const int co = (int)((i >> 10) % coT) * 32 + (int)(i & 31);
      const int ci = (int)(((i >> 10) / coT) % ciT) * 32 + (int)((i >> 5) & 31);
      if (ci >= cin || co >= cout) dwt[i] = 0.f;
    }
  }
}

__global__ void wgrad_bias_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ db, int cout, int nslab, int64_t bslab) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cout) return;
  float v = 0.f;
  for (int b = 0; b < nslab; ++b) v += slabs[(int64_t)b * bslab + c];
  db[c] = v;
}

static int conv_shape_ok_w(const sg_conv_shape* s) {
  if (!s) return 0;
  if (s->n < 1 || s->d < 1 || s->h < 1 || s->w < 1 || s->cin < 1 || s->cout < 1) return 0;
  if (s->kd < 1 || s->kh < 1 || s->kw < 1 || !(s->kd & 1) || !(s->kh & 1) || !(s->kw & 1)) return 0;
  if (s->kd > 7 || s->kh > 7 || s->kw > 7) return 0;
  if (s->upsample_in && ((s->d | s->h | s->w) & 1)) return 0;
  return 1;
}


// ------------------------------------------------------------------------------------------------------
// 1x1x1 weight gradient with a tiny channel count on one side (from_rgb: cin = image channels, to_rgb: cout =
// image channels; networks/ops.py:239-247): dw[ci][co] = sum_v x[v][ci] * dy[v][co] is a column reduction,
// HBM-bound.  "small" has cs <= 4 channels, "big" has cb channels (16-byte pieces, pieces | 256).
// ------------------------------------------------------------------------------------------------------
template <typename T, int CS = 0, int U = 4>      // CS: the small side's channel count at compile time (1: the image layers), 0: run time, <= 4
__global__ __launch_bounds__(256) void pw_wgrad_partial_kernel(const T* __restrict__ small, const T* __restrict__ big,
                                                               float* __restrict__ part, int64_t nvox, int cs_rt, int cb,
                                                               int ones_extra, T* __restrict__ dsmall = nullptr,
                                                               const float* __restrict__ wsm = nullptr) {
  // dsmall (optional, [nvox][cs]): the data gradient for the small side from the same read of `big` (= dy),
  // dsmall[v][j] = sum_c big[v][c] * wsm[j][c] (sg_conv3d_pw_bwd)
  constexpr int E = 16 / (int)sizeof(T);
  constexpr int CM = CS ? CS : 4;                // rows of sums kept in registers (+ 1: the ones row)
  const int cs = CS ? CS : cs_rt;
  __shared__ float red[256 * E];
  const int P = cb / E, rows = 256 / P;
  const int p = threadIdx.x % P, rr = threadIdx.x / P;
  // One-channel images are the case the training step runs: with the four-channel arrays (72 registers of sums and
  // weights) the kernel held 123 VGPRs -- four blocks per CU, 48 KiB in flight per CU where the memory system wants ~60:
  // 3.8-4.2 TB/s.  Sized for ONE channel it fits eight.
  float wreg[CM][E];
#pragma unroll
  for (int j = 0; j < CM; ++j)
#pragma unroll
    for (int e = 0; e < E; ++e) wreg[j][e] = (dsmall != nullptr && j < cs) ? wsm[j * cb + p * E + e] : 0.f;
  float s[CM + 1][E];   // row CM (when ones_extra): the big side's plain column sums (bias gradient)
#pragma unroll
  for (int j = 0; j < CM + 1; ++j)
#pragma unroll
    for (int e = 0; e < E; ++e) s[j][e] = 0.f;
  const int csx = cs + (ones_extra ? 1 : 0);
  // four voxel rows per trip: the loads are independent, and one 16-byte load per lane per trip left the memory
  // system mostly idle (2.4 TB/s)
  const int64_t step = rows;                     // the U rows of a trip are adjacent voxel groups (one contiguous stretch)
  for (int64_t v0 = (int64_t)blockIdx.x * U * rows + rr; v0 < nvox; v0 += (int64_t)gridDim.x * U * rows) {
    u32x4 raw[U];
    float sv[U][CM];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * step;
      const bool live = v < nvox;
      raw[u] = live ? *reinterpret_cast<const u32x4*>(big + v * cb + (int64_t)p * E) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < CM; ++j) sv[u][j] = (live && j < cs) ? sg_traits<T>::to_f(small[v * cs + j]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const T* bt = reinterpret_cast<const T*>(&raw[u]);
      float bv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) bv[e] = sg_traits<T>::to_f(bt[e]);
#pragma unroll
      for (int j = 0; j < CM; ++j) {
        if (j < cs) {
#pragma unroll
          for (int e = 0; e < E; ++e) s[j][e] += sv[u][j] * bv[e];
        }
      }
      if (ones_extra) {   // dead rows were loaded as zeros
#pragma unroll
        for (int e = 0; e < E; ++e) s[CM][e] += bv[e];
      }
      if (dsmall != nullptr) {   // uniform; the P lanes of a voxel are adjacent and run the same trips
        const int64_t v = v0 + u * step;
#pragma unroll
        for (int j = 0; j < CM; ++j) {
          if (j < cs) {
            float dsum = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) dsum = fmaf(bv[e], wreg[j][e], dsum);
            if (P == 4) {      // the voxel's four lanes are one quad: two DPP exchanges, no LDS round trip
              dsum += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(dsum), 0xB1, 0xF, 0xF, true));
              dsum += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(dsum), 0x4E, 0xF, 0xF, true));
            } else {
              for (int sh = 1; sh < P; sh <<= 1) dsum += __shfl_xor(dsum, sh);
            }
            if (p == 0 && v < nvox) dsmall[v * cs + j] = sg_traits<T>::from_f(dsum);
          }
        }
      }
    }
  }
  for (int j = 0; j < csx; ++j) {
    const int js = j < cs ? j : CM;    // the ones row lives in slot CM
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float val = s[CM][e];
#pragma unroll
      for (int jj = 0; jj < CM; ++jj) val = js == jj ? s[jj][e] : val;
      red[threadIdx.x * E + e] = val;
    }
    __syncthreads();
    // column e of piece q summed over the block's `rows` voxel rows: thread (q, e-group) instead of P threads doing it all
    for (int idx = threadIdx.x; idx < P * E; idx += 256) {
      const int q = idx / E, e = idx - q * E;
      float t = 0.f;
      for (int k = 0; k < rows; ++k) t += red[(k * P + q) * E + e];
      part[((int64_t)blockIdx.x * csx + j) * cb + q * E + e] = t;
    }
  }
}

// dw[ci][co] = coef * sum_b part[b][j][i]; small_is_cin: (j, i) = (ci, co) else (j, i) = (co, ci)
__global__ __launch_bounds__(256) void pw_wgrad_final_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                             float coef, int nb, int cs, int cb, int small_is_cin,
                                                             float* __restrict__ dbias) {
  __shared__ float red[8][32];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  const int c = (cs + (dbias ? 1 : 0)) * cb;
  float sum = 0.f;
  if (col < c) {      // four independent sums: the loads of one thread are nb / 32 deep instead of nb / 8 (18 -> ~6 us at 1 024 rows)
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    int b = rg;
    for (; b + 24 < nb; b += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += part[(int64_t)(b + 8 * u) * c + col];
    }
    for (int u = 0; b < nb; b += 8, ++u) s4[u & 3] += part[(int64_t)b * c + col];
    sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  red[rg][threadIdx.x & 31] = sum;
  __syncthreads();
  if (rg == 0 && col < c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    const int j = col / cb, i = col - j * cb;
    if (j < cs) dw[small_is_cin ? (j * cb + i) : (i * cs + j)] = coef * t;
    else dbias[i] = t;
  }
}

static size_t wgrad_tile_bytes(const sg_conv_shape* s);
static constexpr int PW_WGRAD_MAX_BLOCKS = 1024;   // per-block partial sums of the pointwise weight gradient (pw_wgrad_partial)
// upper bound of the slabs a launcher writes (generic kernel: gridDim.x <= cdiv(512, pairs); ping-pong kernels: two wave
// groups per block, gridDim.x <= 256 / pairs and >= 8)
static int wgrad_slab_count(const sg_conv_shape* s) {
  const int pairs = sg_cdiv(s->cin, 32) * sg_cdiv(s->cout, 32);
  const int p = sg_cdiv(512, pairs);
  return p > 16 ? p : 16;
}
static size_t wgrad_bias_slab_bytes(const sg_conv_shape* s) { return ((size_t)sg_cdiv(s->cout, 32) * 32 * 4 + 255) & ~(size_t)255; }

extern "C" size_t sg_conv3d_wgrad_workspace(const sg_conv_shape* s, sg_dtype dt) {
  (void)dt;
  if (!conv_shape_ok_w(s)) return 0;
  // [tile / partial sums (x slabs in reproducible mode)][bias-gradient fallback][bias slabs]
  const size_t ns = sg_cfg().deterministic ? (size_t)wgrad_slab_count(s) : 1;
  // (+ one bias row: with SG_WGRAD_CLEAN_WORKSPACE the bias fallback's scratch starts behind the staged bias row, which stays clean)
  const size_t gen = ns * wgrad_tile_bytes(s) + sg_bias_act_bwd_workspace(s->cout) + (ns > 1 ? ns * wgrad_bias_slab_bytes(s) : 0) +
                     wgrad_bias_slab_bytes(s);
  const size_t small = sg_small_wgrad_workspace(s);                              // per-block slabs of the small-channel kernel
  return gen > small ? gen : small;
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
  SG_ALLOW_160K_LDS(kern);
  const int pairs = a.ciT * a.coT;
  // blocks per (ci, co) pair: two blocks per CU in ONE wave of blocks.  768 (three per CU) was measured slower on the
  // 512 -> 512 layers of the 2x8x8 level, 229 against 179 us at batch 64: the third block of a CU runs as a tail.
  int P = sg_cdiv(sg_cfg().wgrad_v1_blocks > 0 ? sg_cfg().wgrad_v1_blocks : 512, pairs);
  if (P > a.ntiles) P = a.ntiles;
  // reproducible mode: every block stores a whole slab, and the workspace holds wgrad_slab_count() of them -- a block target
  // raised by SG_WGRAD_V1_BLOCKS must not write past it
  if (a.slab != 0 && P > wgrad_slab_count(s)) P = wgrad_slab_count(s);
  if (P < 1) P = 1;
  for (int tap0 = 0; tap0 < a.taps; tap0 += 4 * WG_MAXT) {
    a.tap0 = tap0;
    a.taps_blk = a.taps - tap0 < 4 * WG_MAXT ? a.taps - tap0 : 4 * WG_MAXT;
    SG_KNAME("conv_wgrad<%s,%d>", sg_tname<T>(), BM);
    a.nslab = P;
    hipLaunchKernelGGL(kern, dim3((unsigned)P, (unsigned)pairs), dim3(256), lds, st, a);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}


// ------------------------------------------------------------------------------------------------------
// wgrad v2 (bf16, tiles 32 voxels wide): persistent 8-wave ping-pong.  Two groups of 4 waves alternate: while
// group g runs the MFMAs of its tile out of its own LDS images (x halo + dy tile, plain 64-byte rows: the
// transposing reads of 4 consecutive rows are bank-conflict free), the other group fetches its next tile by
// LDS-DMA.  Accumulators persist over all tiles of the block; one f32 atomic pass at the end.
// ------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(1024))) uint32_t sg_zero_page_w[256] = {0};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void sg_glds16w(const void* gp, char* lds_uniform_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)gp, (lds_ptr_t)lds_uniform_base, 16, 0, 0);
}

template <int KD, int KH, int KW>
__global__ __launch_bounds__(512) void conv_wgrad2_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int TAPS = KD * KH * KW;
  constexpr int MAXT = (TAPS + 3) / 4;            // taps per wave
  constexpr int BM = 256, KSTEPS = BM / 16;
  const sg_tile_geom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int bufbytes = a.xbytes + a.ybytes;
  char* xmine = smem + grp * bufbytes;
  char* ymine = xmine + a.xbytes;
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* dy = reinterpret_cast<const T*>(a.dy);
  const int ci_t = blockIdx.y / a.coT, co_t = blockIdx.y % a.coT;

  // tile schedule (same XCD-chunked dealing as the forward kernel)
  const int xg = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (a.ntiles + 7) >> 3;
  const int t_begin = xg * cpx;
  const int t_end = min(a.ntiles, t_begin + cpx);
  const int first = t_begin + slot;
  const int K = first < t_end ? (t_end - first + per_x - 1) / per_x : 0;

  // transposing-read lane geometry: 16-lane group q16, lane (qd, pp) supplies row qd, columns 4pp..4pp+3
  const int i16 = lane & 15, q16 = lane >> 4;
  const int qd = i16 >> 2, pp = i16 & 3;
  const int colb = (16 * (q16 & 1) + 4 * pp) * 2;
  const int kb = 8 * (q16 >> 1) + qd;
  // lane-constant byte offsets of the two reads of a K step (voxels kb, kb+4 of the 16-voxel run).  In the halo
  // image a run narrower than 16 voxels (tiles with TW = 4 or 8: the low-resolution layers) continues on the next
  // H row, HW rows further; the dy image is voxel-linear.
  const int twr = g.TW < 16 ? g.TW : 16;
  auto xrow = [&](int kk) { return ((kk / twr) * g.HW + (kk % twr)) * 64; };
  const int xl0 = (int)(xmine - smem) + xrow(kb) + colb, xl1 = (int)(xmine - smem) + xrow(kb + 4) + colb;
  const int yl0 = (int)(ymine - smem) + kb * 64 + colb, yl1 = yl0 + 4 * 64;

  // halo staging tables: element offset relative to the tile's first halo voxel (-1: dead piece)
  const int hvx = g.TN * g.HD * g.HH * g.HW, itx = hvx * 4;
  const int hvy = g.TN * g.TD * g.TH * g.TW, ity = hvy * 4;
  constexpr int MAXX = 14, MAXY = 4;
  int relx[MAXX], rely[MAXY];
  // packed halo / tile coordinates (w | h << 8 | d << 16 | n << 24, every extent < 128: host-checked) of my pieces: a tile
  // at the volume's boundary tests them against the tile's valid range with one packed compare per piece.  (It used to
  // recompute the coordinates per piece with three divisions and 64-bit address arithmetic in a rolled loop -- and on the
  // 16^2 level, where a tile spans the whole H x W plane, EVERY tile is a boundary tile: 7.6 us per tile against 2 us of MFMAs.)
  uint32_t crdx[MAXX], crdy[MAXY];
#pragma unroll
  for (int k = 0; k < MAXX; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> 2, c = ci_t * 32 + (it & 3) * 8;
    uint32_t q = sg_div((uint32_t)row, g.fHW);
    int hw = (int)(row - q * g.HW);
    uint32_t q2 = sg_div(q, g.fHH);
    int hh_ = (int)(q - q2 * g.HH);
    uint32_t q3 = sg_div(q2, g.fHD);
    int hd = (int)(q2 - q3 * g.HD);
    relx[k] = (row < hvx && c < a.cin) ? ((((int)q3 * g.D + hd) * g.H + hh_) * g.W + hw) * a.cin + c : -1;
    crdx[k] = (row < hvx && c < a.cin) ? ((uint32_t)hw | ((uint32_t)hh_ << 8) | ((uint32_t)hd << 16) | (q3 << 24)) : 0x7F7F7F7Fu;
  }
#pragma unroll
  for (int k = 0; k < MAXY; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> 2, c = co_t * 32 + (it & 3) * 8;
    uint32_t q = sg_div((uint32_t)row, g.fTW);
    int tw = (int)(row - q * g.TW);
    uint32_t q2 = sg_div(q, g.fTH);
    int th = (int)(q - q2 * g.TH);
    uint32_t q3 = sg_div(q2, g.fTD);
    int td = (int)(q2 - q3 * g.TD);
    rely[k] = (row < hvy && c < a.cout) ? ((((int)q3 * g.D + td) * g.H + th) * g.W + tw) * a.cout + c : -1;
    crdy[k] = (row < hvy && c < a.cout) ? ((uint32_t)tw | ((uint32_t)th << 8) | ((uint32_t)td << 16) | (q3 << 24)) : 0x7F7F7F7Fu;
  }

  auto stage_tile = [&](const sg_tile_origin& o) {
    const bool interior = !g.ups && o.d0 >= g.PD && o.h0 >= g.PH && o.w0 >= g.PW && o.d0 + g.TD + g.PD <= g.D &&
                          o.h0 + g.TH + g.PH <= g.H && o.w0 + g.TW + g.PW <= g.W && o.n0 + g.TN <= g.N;
    const int64_t v0 = (((int64_t)o.n0 * g.D + o.d0) * g.H + o.h0) * g.W + o.w0;   // first tile voxel
    const T* ybase = dy + v0 * a.cout;
    const T* xbase = x + (v0 - ((int64_t)g.PD * g.H + g.PH) * g.W - g.PW) * a.cin;
    if (interior) {
#pragma unroll
      for (int k = 0; k < MAXX; ++k)
        if ((wave + 4 * k) * 64 < itx)
          sg_glds16w(relx[k] >= 0 ? (const void*)(xbase + relx[k]) : (const void*)sg_zero_page_w,
                     xmine + (size_t)(wave + 4 * k) * 1024);
#pragma unroll
      for (int k = 0; k < MAXY; ++k)
        if ((wave + 4 * k) * 64 < ity)
          sg_glds16w(rely[k] >= 0 ? (const void*)(ybase + rely[k]) : (const void*)sg_zero_page_w,
                     ymine + (size_t)(wave + 4 * k) * 1024);
    } else if (!g.ups) {   // boundary tile: the same offsets, pieces outside the volume read the zero page
      const int lo_w = max(0, g.PW - o.w0), hi_w = min(g.HW, g.W + g.PW - o.w0) - 1;
      const int lo_h = max(0, g.PH - o.h0), hi_h = min(g.HH, g.H + g.PH - o.h0) - 1;
      const int lo_d = max(0, g.PD - o.d0), hi_d = min(g.HD, g.D + g.PD - o.d0) - 1;
      const int hi_n = min(g.TN, g.N - o.n0) - 1;
      const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8) | (lo_d << 16));
      const uint32_t hi = (uint32_t)(hi_w | (hi_h << 8) | (hi_d << 16) | (hi_n << 24)) | 0x80808080u;
      const uint32_t hiy = (uint32_t)((min(g.TW, g.W - o.w0) - 1) | ((min(g.TH, g.H - o.h0) - 1) << 8) |
                                      ((min(g.TD, g.D - o.d0) - 1) << 16) | (hi_n << 24)) | 0x80808080u;
#pragma unroll
      for (int k = 0; k < MAXX; ++k)
        if ((wave + 4 * k) * 64 < itx) {
          const uint32_t c_ = crdx[k];
          const bool in = ((((c_ | 0x80808080u) - lo) & (hi - c_) & 0x80808080u) == 0x80808080u);
          sg_glds16w(in ? (const void*)(xbase + relx[k]) : (const void*)sg_zero_page_w, xmine + (size_t)(wave + 4 * k) * 1024);
        }
#pragma unroll
      for (int k = 0; k < MAXY; ++k)
        if ((wave + 4 * k) * 64 < ity) {
          const uint32_t c_ = crdy[k];
          const bool in = (((c_ | 0x80808080u) & (hiy - c_) & 0x80808080u) == 0x80808080u);
          sg_glds16w(in ? (const void*)(ybase + rely[k]) : (const void*)sg_zero_page_w, ymine + (size_t)(wave + 4 * k) * 1024);
        }
    } else {   // fused nearest-x2 gather: coordinates per piece (kept rolled: it must not cost registers)
#pragma unroll 1
      for (int k = 0; k < MAXX; ++k) {
        if ((wave + 4 * k) * 64 < itx) {
          const int it = (wave + 4 * k) * 64 + lane;
          const int row = it >> 2, c = ci_t * 32 + (it & 3) * 8;
          uint32_t q = sg_div((uint32_t)row, g.fHW);
          int hw = (int)(row - q * g.HW);
          uint32_t q2 = sg_div(q, g.fHH);
          int hh_ = (int)(q - q2 * g.HH);
          uint32_t q3 = sg_div(q2, g.fHD);
          int hd = (int)(q2 - q3 * g.HD);
          const int n = o.n0 + (int)q3, d = o.d0 + hd - g.PD, h = o.h0 + hh_ - g.PH, w = o.w0 + hw - g.PW;
          const bool ok = row < hvx && c < a.cin && n < g.N && (unsigned)d < (unsigned)g.D &&
                          (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
          const void* src = sg_zero_page_w;
          if (ok) {
            if (g.ups)   // x is half resolution: nearest-neighbour x2 gather (conv3d(upscale3d(x)))
              src = x + ((((int64_t)n * (g.D >> 1) + (d >> 1)) * (g.H >> 1) + (h >> 1)) * (g.W >> 1) + (w >> 1)) *
                            (int64_t)a.cin + c;
            else
              src = xbase + (((((int)q3 * g.D + hd) * g.H + hh_) * g.W + hw) * a.cin + c);
          }
          sg_glds16w(src, xmine + (size_t)(wave + 4 * k) * 1024);
        }
      }
#pragma unroll 1
      for (int k = 0; k < MAXY; ++k) {
        if ((wave + 4 * k) * 64 < ity) {
          const int it = (wave + 4 * k) * 64 + lane;
          const int row = it >> 2, c = co_t * 32 + (it & 3) * 8;
          uint32_t q = sg_div((uint32_t)row, g.fTW);
          int tw = (int)(row - q * g.TW);
          uint32_t q2 = sg_div(q, g.fTH);
          int th = (int)(q - q2 * g.TH);
          uint32_t q3 = sg_div(q2, g.fTD);
          int td = (int)(q2 - q3 * g.TD);
          const int n = o.n0 + (int)q3, d = o.d0 + td, h = o.h0 + th, w = o.w0 + tw;
          const bool ok = row < hvy && c < a.cout && n < g.N && d < g.D && h < g.H && w < g.W;
          const int rel = ((((int)q3 * g.D + td) * g.H + th) * g.W + tw) * a.cout + c;
          sg_glds16w(ok ? (const void*)(ybase + rel) : (const void*)sg_zero_page_w,
                     ymine + (size_t)(wave + 4 * k) * 1024);
        }
      }
    }
  };

  // per-wave taps: tap = wave + 4*j; byte offset of the tap's row shift in the halo image
  int tapoff[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int tap = wave + 4 * j;
    const int kw_i = tap % KW, kh_i = (tap / KW) % KH, kd_i = tap / (KW * KH);
    tapoff[j] = tap < TAPS ? ((kd_i * g.HH + kh_i) * g.HW + kw_i) * 64 : 0;
  }
  // byte offset of the (td, th) line of K step pair j in the halo image / dy image
  // (tile is TD x TH x 32: K steps 2j, 2j+1 cover line j)
  f32x16 acc[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  if (grp == 0 && K > 0) stage_tile(sg_tile_of(g, (uint32_t)first));
  __syncthreads();

  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
  for (int p = 0; p <= K; ++p) {
    if ((p & 1) == grp) {
      if (p < K) {
#pragma unroll
        for (int run = 0; run < KSTEPS; ++run) {               // 16 consecutive tile voxels = one K step
          const uint32_t m0 = (uint32_t)run * 16u;
          const uint32_t q1 = sg_div(m0, g.fTW);
          const int tw0 = (int)(m0 - q1 * g.TW);
          const uint32_t q2 = sg_div(q1, g.fTH);
          const int th = (int)(q1 - q2 * g.TH);
          const uint32_t q3 = sg_div(q2, g.fTD);
          const int td = (int)(q2 - q3 * g.TD);
          const int xline = ((((int)q3 * g.HD + td) * g.HH + th) * g.HW + tw0) * 64;
          const int yline = run * 16 * 64;
          s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + yl0 + yline));
          s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + yl1 + yline));
          u32x4 bf;
          bf[0] = __builtin_bit_cast(u32x2, b0)[0]; bf[1] = __builtin_bit_cast(u32x2, b0)[1];
          bf[2] = __builtin_bit_cast(u32x2, b1)[0]; bf[3] = __builtin_bit_cast(u32x2, b1)[1];
#pragma unroll
          for (int j = 0; j < MAXT; ++j) {
            if (wave + 4 * j < TAPS) {
              s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + xl0 + xline + tapoff[j]));
              s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + xl1 + xline + tapoff[j]));
              u32x4 af;
              af[0] = __builtin_bit_cast(u32x2, a0)[0]; af[1] = __builtin_bit_cast(u32x2, a0)[1];
              af[2] = __builtin_bit_cast(u32x2, a1)[0]; af[3] = __builtin_bit_cast(u32x2, a1)[1];
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af),
                                                              __builtin_bit_cast(bf16x8, bf), acc[j], 0, 0, 0);
            }
          }
        }
      }
    } else {
      if (p + 1 < K) stage_tile(sg_tile_of(g, (uint32_t)(first + (p + 1) * per_x)));
    }
    __syncthreads();
  }
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int tap = wave + 4 * j;
    if (tap < TAPS && (K > 0 || a.slab != 0)) {
      float* dst = a.dwt + (int64_t)(blockIdx.x * 2 + grp) * a.slab + ((((int64_t)tap * a.ciT + ci_t) * a.coT + co_t) << 10);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
        sg_wg_out(dst + row * 32 + r, acc[j][i], a.slab != 0);
      }
    }
  }
}

template <int KD, int KH, int KW>
static int launch_wgrad2(WgradArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  a.g = sg_make_geom(s, 256, /*prefer_w32=*/true);
  const sg_tile_geom& g = a.g;
  // 256-voxel tiles whose 16-voxel K steps are whole W rows (TW = 16) or half rows (TW = 32).  The kernel also
  // runs 8-wide tiles, but measured on the 2x8x8 layers (256 cin/cout pairs, 8-16 tiles) it loses to the v1 kernel.
  if (g.TW < 16 || g.TN * g.TD * g.TH * g.TW != 256) return SG_OK;
  if (g.TN > 127 || g.HD > 127 || g.HH > 127 || g.HW > 127) return SG_OK;
  if ((s->cin % 8) || (s->cout % 8)) return SG_OK;
  // staging offsets are relative to the tile's first sample (64-bit tile bases): TN samples must fit 31 bits
  if ((int64_t)g.TN * s->d * s->h * s->w * (int64_t)(s->cin > s->cout ? s->cin : s->cout) >= (1ll << 31)) return SG_OK;
  const int64_t ntiles = (int64_t)g.nTn * g.nTd * g.nTh * g.nTw;
  const int pairs = a.ciT * a.coT;
  int gx = (256 / pairs) / 8 * 8;
  if (gx < 8) gx = 8;
  if (ntiles >= (1 << 24) || ntiles < 2 * gx) return SG_OK;
  a.gy = a.g;
  a.ntiles = (int)ntiles;
  a.rs = 64;
  const int hv = g.TN * g.HD * g.HH * g.HW;
  a.xbytes = (hv * 64 + 1023) & ~1023;
  a.ybytes = 256 * 64;
  if (sg_cdiv(hv * 4, 64) > 56) return SG_OK;
  const size_t lds = 2ull * (a.xbytes + a.ybytes);
  if (lds > 160 * 1024) return SG_OK;
  auto kern = conv_wgrad2_kernel<KD, KH, KW>;
  SG_ALLOW_160K_LDS(kern);
  a.tap0 = 0; a.taps_blk = a.taps;
  SG_KNAME("conv_wgrad2<%d,%d,%d>", KD, KH, KW);
  a.nslab = 2 * gx;      // (both wave groups of a block keep sums of their own)
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)pairs), dim3(512), lds, st, a);
  SG_LAUNCH_CHECK();
  *used = true;
  return SG_OK;
}



// Fully unrolled, software-pipelined K loop of one 256-voxel tile for the sliding-halo kernel: a stream of
// 16 K steps x (1 dy item + MAXT x items); every item is two transposing reads, issued PF items ahead of use with
// COUNTED waits (LDS returns in order; hipcc would emit lgkmcnt(0) per K step and expose a full LDS round trip
// every 7 MFMAs).  dy fragments are double buffered across K steps, x fragments live in a ring of PF + 1.
template <int MAXT, int TH, int PF>
struct sg_wgrad_tile {
  static constexpr int IPS = 1 + MAXT;          // items per K step
  static constexpr int NI = 16 * IPS;           // items per tile
  static constexpr int RING = PF + 1;
  typedef s16x4 frag_t;
  struct Ctx {
    int xl0, xl1, yl0, yl1;                     // lane parts (VGPR)
    int sslot[2][MAXT];                         // per (td, tap): ring-slot base + tap row shift (uniform)
    int ones_last;                              // 1: this wave's last slot multiplies dy by ones (bias gradient)
    int throw_[TH];                             // per th: row offset (uniform)
  };
  template <int I>
  static __device__ __forceinline__ void load(const Ctx& c, frag_t (&a0)[RING], frag_t (&a1)[RING], frag_t (&b0)[2],
                                              frag_t (&b1)[2]) {
    constexpr int ks = I / IPS, r = I % IPS, line = ks / 2, half = ks % 2, td = line / TH, th = line % TH;
    if constexpr (r == 0) {
      const int q0 = c.yl0 + ks * 16 * 64, q1 = c.yl1 + ks * 16 * 64;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0[ks & 1]) : "v"(q0));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1[ks & 1]) : "v"(q1));
    } else {
      constexpr int j = r - 1, SL = (ks * MAXT + j) % RING;
      const int off = c.sslot[td][j] + c.throw_[th] + half * 16 * 64;
      const int p0 = c.xl0 + off, p1 = c.xl1 + off;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0[SL]) : "v"(p0));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a1[SL]) : "v"(p1));
    }
  }
  template <int I>
  static __device__ __forceinline__ void step(const Ctx& c, f32x16 (&acc)[MAXT], frag_t (&a0)[RING], frag_t (&a1)[RING],
                                              frag_t (&b0)[2], frag_t (&b1)[2]) {
    if constexpr (I < NI) {
      if constexpr (I + PF < NI) load<I + PF>(c, a0, a1, b0, b1);
      constexpr int ks = I / IPS, r = I % IPS;
      if constexpr (r != 0) {
        constexpr int younger = (NI - 1 - I < PF ? NI - 1 - I : PF) * 2;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
        __builtin_amdgcn_sched_barrier(0);
        constexpr int j = r - 1, SL = (ks * MAXT + j) % RING;
        u32x4 af, bf;
        af[0] = __builtin_bit_cast(u32x2, a0[SL])[0]; af[1] = __builtin_bit_cast(u32x2, a0[SL])[1];
        af[2] = __builtin_bit_cast(u32x2, a1[SL])[0]; af[3] = __builtin_bit_cast(u32x2, a1[SL])[1];
        if constexpr (j == MAXT - 1) {
          if (c.ones_last) af = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};   // bf16 1.0 x 8
        }
        bf[0] = __builtin_bit_cast(u32x2, b0[ks & 1])[0]; bf[1] = __builtin_bit_cast(u32x2, b0[ks & 1])[1];
        bf[2] = __builtin_bit_cast(u32x2, b1[ks & 1])[0]; bf[3] = __builtin_bit_cast(u32x2, b1[ks & 1])[1];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf),
                                                        acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      step<I + 1>(c, acc, a0, a1, b0, b1);
    }
  }
  template <int I>
  static __device__ __forceinline__ void prologue(const Ctx& c, frag_t (&a0)[RING], frag_t (&a1)[RING], frag_t (&b0)[2],
                                                  frag_t (&b1)[2]) {
    if constexpr (I < PF && I < NI) {
      load<I>(c, a0, a1, b0, b1);
      prologue<I + 1>(c, a0, a1, b0, b1);
    }
  }
  static __device__ __forceinline__ void run(const Ctx& c, f32x16 (&acc)[MAXT]) {
    frag_t a0[RING], a1[RING], b0[2], b1[2];
    SG_KLOOP_BEGIN();
    prologue<0>(c, a0, a1, b0, b1);
    step<0>(c, acc, a0, a1, b0, b1);
    SG_KLOOP_END();
  }
};

// ------------------------------------------------------------------------------------------------------
// wgrad v3 (bf16, 3x3x3 / 1x3x3, tiles TD x TH x 32): wgrad2 with a SLIDING halo.  Tiles are walked along D
// (columns of tiles at fixed (n, h, w)); the x halo lives in a ring of HD = TD + 2*PD plane slots, so a step
// along D fetches only the TD new planes (half the halo for 3x3x3) instead of all HD: LDS-DMA pieces per tile
// drop from 13 + 4 to 7 + 4 per wave, which is what bounded wgrad2 (its staging phase was longer than its
// MFMA phase).  Group g of the ping-pong walks columns g, g+2, ... of the block's column list.
// ------------------------------------------------------------------------------------------------------
template <int KD, int KH, int KW>
__global__ __launch_bounds__(512) void conv_wgrad3_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int TAPS = KD * KH * KW;
  constexpr int MAXT = (TAPS + 3) / 4;
  constexpr int BM = 256, KSTEPS = BM / 16;
  const sg_tile_geom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int PR = a.plane_rows;                       // LDS rows per plane slot (multiple of 16)
  const int pbytes = PR * 64;
  const int bufbytes = a.xbytes + a.ybytes;
  char* xmine = smem + grp * bufbytes;
  char* ymine = xmine + a.xbytes;
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* dy = reinterpret_cast<const T*>(a.dy);
  const int ci_t = blockIdx.y / a.coT, co_t = blockIdx.y % a.coT;
  const int HDm = g.HD - 1;                          // HD is a power of two (2 or 4): slot = plane & HDm

  // column schedule: XCD group xg owns a contiguous chunk of the (n, h, w) column list
  const int ncol = g.nTn * g.nTh * g.nTw;
  const int xg = blockIdx.x & 7, bslot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (ncol + 7) >> 3;
  const int c_begin = xg * cpx, c_end = min(ncol, c_begin + cpx);
  const int cfirst = c_begin + bslot;
  const int ncols_blk = cfirst < c_end ? (c_end - cfirst + per_x - 1) / per_x : 0;
  const int ncols_mine = (ncols_blk + 1 - grp) >> 1;          // columns of my group: grp, grp+2, ...
  const int items_mine = ncols_mine * g.nTd;
  const int items_max = ((ncols_blk + 1) >> 1) * g.nTd;

  const int i16 = lane & 15, q16 = lane >> 4;
  const int qd = i16 >> 2, pp = i16 & 3;
  const int colb = (16 * (q16 & 1) + 4 * pp) * 2;
  const int kb = 8 * (q16 >> 1) + qd;
  const int xl0 = (int)(xmine - smem) + kb * 64 + colb, xl1 = xl0 + 4 * 64;
  const int yl0 = (int)(ymine - smem) + kb * 64 + colb, yl1 = yl0 + 4 * 64;

  // plane-local staging tables (identical for every plane): this lane's pieces w, w+4, w+8, w+12 of a plane
  const int prow = g.HH * g.HW;                      // live rows per plane
  const int ppieces = PR / 16;                       // 1-KiB pieces per plane
  constexpr int MAXP = 4, MAXY = 4;
  int relp[MAXP];   // element offset relative to the plane's first halo voxel (h0-PH, w0-PW); -1 dead
  int crdp[MAXP];   // packed (hw, hh) for the boundary test; dead pieces fail every range
  int rely[MAXY];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> 2, c = ci_t * 32 + (it & 3) * 8;
    uint32_t q = sg_div((uint32_t)row, g.fHW);
    const int hw = (int)(row - q * g.HW), hh_ = (int)q;
    const bool live = row < prow && c < a.cin;
    relp[k] = live ? (hh_ * g.W + hw) * a.cin + c : -1;
    crdp[k] = live ? (hw | (hh_ << 8)) : 0x7F7F;
  }
  const int hvy = g.TD * g.TH * g.TW, ity = hvy * 4;
#pragma unroll
  for (int k = 0; k < MAXY; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> 2, c = co_t * 32 + (it & 3) * 8;
    uint32_t q = sg_div((uint32_t)row, g.fTW);
    int tw = (int)(row - q * g.TW);
    uint32_t q2 = sg_div(q, g.fTH);
    int th = (int)(q - q2 * g.TH);
    int td = (int)q2;
    rely[k] = (row < hvy && c < a.cout) ? ((td * g.H + th) * g.W + tw) * a.cout + c : -1;
  }

  // item q of my group -> (column, d index)
  auto origin_of = [&](int q, int& di) {
    const int cj = q / g.nTd;
    di = q - cj * g.nTd;
    const int col = cfirst + (2 * cj + grp) * per_x;
    sg_tile_origin o;
    uint32_t c1 = sg_div((uint32_t)col, g.fnTw);
    o.w0 = (int)(col - c1 * g.nTw) * g.TW;
    uint32_t c2 = sg_div(c1, g.fnTh);
    o.h0 = (int)(c1 - c2 * g.nTh) * g.TH;
    o.n0 = (int)c2;
    o.d0 = di * g.TD;
    return o;
  };

  auto stage_item = [&](int q) {
    int di;
    const sg_tile_origin o = origin_of(q, di);
    // ---- x: halo planes hd in [hd_begin, HD) are new (all of them at the bottom of a column) ----
    const int hd_begin = di == 0 ? 0 : g.HD - g.TD;
    const bool hw_interior = !g.ups && o.h0 >= g.PH && o.w0 >= g.PW && o.h0 + g.TH + g.PH <= g.H &&
                             o.w0 + g.TW + g.PW <= g.W;
    const int lo_w = max(0, g.PW - o.w0), hi_w = min(g.HW, g.W + g.PW - o.w0) - 1;
    const int lo_h = max(0, g.PH - o.h0), hi_h = min(g.HH, g.H + g.PH - o.h0) - 1;
    const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8));
    const uint32_t hi = (uint32_t)(hi_w | (hi_h << 8)) | 0x8080u;
    for (int hd = hd_begin; hd < g.HD; ++hd) {
      const int gp = o.d0 - g.PD + hd;                       // global plane (d coordinate)
      const int slot_ = (gp + 2 * g.HD) & HDm;
      char* dst = xmine + slot_ * pbytes;
      const bool plane_ok = gp >= 0 && gp < g.D;
      const T* base = g.ups ? x
                            : x + ((((int64_t)o.n0 * g.D + gp) * g.H + (o.h0 - g.PH)) * g.W + (o.w0 - g.PW)) * (int64_t)a.cin;
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        if (wave + 4 * k < ppieces) {
          const void* src = sg_zero_page_w;
          if (plane_ok) {
            if (hw_interior) {
              if (relp[k] >= 0) src = base + relp[k];
            } else if (!g.ups) {
              const uint32_t c_ = (uint32_t)crdp[k];
              const uint32_t t1 = (c_ | 0x8080u) - lo, t2 = hi - c_;
              if ((t1 & t2 & 0x8080u) == 0x8080u) src = base + relp[k];
            } else {   // fused nearest-x2 gather
              const int hw = crdp[k] & 255, hh_ = (crdp[k] >> 8) & 255;
              const int h = o.h0 + hh_ - g.PH, w = o.w0 + hw - g.PW;
              if (relp[k] >= 0 && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W) {
                const int c = ci_t * 32 + (lane & 3) * 8;
                src = x + ((((int64_t)o.n0 * (g.D >> 1) + (gp >> 1)) * (g.H >> 1) + (h >> 1)) * (g.W >> 1) + (w >> 1)) *
                              (int64_t)a.cin + c;
              }
            }
          }
          sg_glds16w(src, dst + (size_t)(wave + 4 * k) * 1024);
        }
      }
    }
    // ---- dy tile ----
    const bool y_interior = o.d0 + g.TD <= g.D && o.h0 + g.TH <= g.H && o.w0 + g.TW <= g.W;
    const T* ybase = dy + ((((int64_t)o.n0 * g.D + o.d0) * g.H + o.h0) * g.W + o.w0) * (int64_t)a.cout;
#pragma unroll
    for (int k = 0; k < MAXY; ++k) {
      if ((wave + 4 * k) * 64 < ity) {
        const void* src = sg_zero_page_w;
        if (rely[k] >= 0) {
          bool ok = y_interior;
          if (!ok) {
            const int row = ((wave + 4 * k) * 64 + lane) >> 2;
            uint32_t q_ = sg_div((uint32_t)row, g.fTW);
            int tw = (int)(row - q_ * g.TW);
            uint32_t q2 = sg_div(q_, g.fTH);
            int th = (int)(q_ - q2 * g.TH);
            int td = (int)q2;
            ok = o.d0 + td < g.D && o.h0 + th < g.H && o.w0 + tw < g.W;
          }
          if (ok) src = ybase + rely[k];
        }
        sg_glds16w(src, ymine + (size_t)(wave + 4 * k) * 1024);
      }
    }
  };

  // per-wave taps
  int tap_hw[MAXT], tap_kd[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int tap = wave + 4 * j;
    const int kw_i = tap % KW, kh_i = (tap / KW) % KH, kd_i = tap / (KW * KH);
    tap_hw[j] = tap < TAPS ? (kh_i * g.HW + kw_i) * 64 : 0;
    tap_kd[j] = tap < TAPS ? kd_i : 0;
  }
  // bias gradient: wave 3's last slot is spare (27 = 4*7 - 1 taps, 9 = 4*3 - 3); it multiplies dy by ones
  static_assert(3 + 4 * (MAXT - 1) >= TAPS, "no spare tap slot for the bias gradient");
  const int ones_last = (a.dbias != nullptr && ci_t == 0 && wave == 3) ? 1 : 0;
  f32x16 acc[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  if (grp == 0 && items_mine > 0) stage_item(0);
  __syncthreads();

  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && lane == 0 && wave == 0 && dbgi < 120)
      a.dbg[grp * 128 + dbgi] = __builtin_amdgcn_s_memtime();
    ++dbgi;
  };
  const int nphase = 2 * items_max + 1;
  for (int p = 0; p < nphase; ++p) {
    const int q = p >> 1;
    stamp();
    if ((p & 1) == grp) {
      if (q < items_mine) {
        const int di = q % g.nTd;
        const int pbase = di * g.TD - g.PD + 2 * g.HD;          // plane of halo index 0 (kept non-negative)
        typedef sg_wgrad_tile<MAXT, 4, 5> KT;
        typename KT::Ctx c;
        c.xl0 = xl0; c.xl1 = xl1; c.yl0 = yl0; c.yl1 = yl1;
        c.ones_last = ones_last;
#pragma unroll
        for (int td = 0; td < 2; ++td)
#pragma unroll
          for (int j = 0; j < MAXT; ++j) c.sslot[td][j] = ((pbase + td + tap_kd[j]) & HDm) * pbytes + tap_hw[j];
#pragma unroll
        for (int th = 0; th < 4; ++th) c.throw_[th] = th * g.HW * 64;
        KT::run(c, acc);
      }
    } else {
      const int qn = (p + 1) >> 1;
      if (qn < items_mine && !((a.dbg_flags & 1) && p >= 2)) stage_item(qn);
      stamp();
      if (a.dbg != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stamps only: issue vs landed
    }
    stamp();
    __syncthreads();
  }
  const int r = lane & 31, hh = lane >> 5;
  if (ones_last && (items_mine > 0 || a.slab != 0) && hh == 0 && co_t * 32 + r < a.cout)   // row 0 of the ones product = column sums
    sg_wg_out(a.dbias + (int64_t)(blockIdx.x * 2 + grp) * a.bslab + co_t * 32 + r, acc[MAXT - 1][0], a.slab != 0);
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int tap = wave + 4 * j;
    if (tap < TAPS && (items_mine > 0 || a.slab != 0)) {
      float* dst = a.dwt + (int64_t)(blockIdx.x * 2 + grp) * a.slab + ((((int64_t)tap * a.ciT + ci_t) * a.coT + co_t) << 10);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
        sg_wg_out(dst + row * 32 + r, acc[j][i], a.slab != 0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// wgrad v3 "lean" (bf16, 3x3x3, no fused up-sampling; tiles 2 x 4 x 32, halo plane slots of 208 64-byte rows):
// the sliding-halo kernel above with (a) the fragment addresses of the unrolled K loop folded into 14 per-tile base
// registers + instruction offsets -- the loop above spends three of its seven instructions per MFMA on address
// arithmetic, and one wave issues roughly one instruction per 5 cycles: in-kernel stamps, 49 cycles per MFMA --
// and (b) staging through buffer resources with per-column lane offsets and scalar per-tile offsets (LDS-DMA,
// hardware zero fill) instead of 64-bit pointer selects per piece: the staging of one tile took 4.8-5.8k cycles
// to ISSUE, as long as the MFMA phase it should hide behind.
// ------------------------------------------------------------------------------------------------------
template <int PF, int TW = 32>   // TW = 32: tiles 2 x 4 x 32 (a W row is two 16-voxel K steps); TW = 16: tiles 2 x 8 x 16 (one K step per row)
struct sg_wgrad_tile_lean {
  static constexpr int MAXT = 7, TH = 128 / TW, HW = TW + 2, KPL = TW / 16;
  static constexpr int IPS = 1 + MAXT, NI = 16 * IPS, RING = PF + 1;
  typedef s16x4 frag_t;
  struct Ctx {
    int xb[2][MAXT];   // per (td, tap slot): lane part + ring slot + tap shift (the bias slot: the block's image of bf16 ones)
    int yb;            // lane part of the dy image
  };
  template <int I>
  static __device__ __forceinline__ void load(const Ctx& c, frag_t (&a0)[RING], frag_t (&a1)[RING], frag_t (&b0)[2],
                                              frag_t (&b1)[2]) {
    constexpr int ks = I / IPS, r = I % IPS, line = ks / KPL, half = ks % KPL, td = line / TH, th = line % TH;
    if constexpr (r == 0) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b0[ks & 1]) : "v"(c.yb), "n"(ks * 1024));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b1[ks & 1]) : "v"(c.yb), "n"(ks * 1024 + 256));
    } else {
      constexpr int j = r - 1, SL = (ks * MAXT + j) % RING;
      constexpr int off = th * HW * 64 + half * 1024;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a0[SL]) : "v"(c.xb[td][j]), "n"(off));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a1[SL]) : "v"(c.xb[td][j]), "n"(off + 256));
    }
  }
  template <int I>
  static __device__ __forceinline__ void step(const Ctx& c, f32x16 (&acc)[MAXT], frag_t (&a0)[RING], frag_t (&a1)[RING],
                                              frag_t (&b0)[2], frag_t (&b1)[2]) {
    if constexpr (I < NI) {
      if constexpr (I + PF < NI) load<I + PF>(c, a0, a1, b0, b1);
      constexpr int ks = I / IPS, r = I % IPS;
      if constexpr (r != 0) {
        constexpr int younger = (NI - 1 - I < PF ? NI - 1 - I : PF) * 2;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
        __builtin_amdgcn_sched_barrier(0);
        constexpr int j = r - 1, SL = (ks * MAXT + j) % RING;
        u32x4 af, bf;
        af[0] = __builtin_bit_cast(u32x2, a0[SL])[0]; af[1] = __builtin_bit_cast(u32x2, a0[SL])[1];
        af[2] = __builtin_bit_cast(u32x2, a1[SL])[0]; af[3] = __builtin_bit_cast(u32x2, a1[SL])[1];
        bf[0] = __builtin_bit_cast(u32x2, b0[ks & 1])[0]; bf[1] = __builtin_bit_cast(u32x2, b0[ks & 1])[1];
        bf[2] = __builtin_bit_cast(u32x2, b1[ks & 1])[0]; bf[3] = __builtin_bit_cast(u32x2, b1[ks & 1])[1];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf),
                                                        acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      step<I + 1>(c, acc, a0, a1, b0, b1);
    }
  }
  template <int I>
  static __device__ __forceinline__ void prologue(const Ctx& c, frag_t (&a0)[RING], frag_t (&a1)[RING], frag_t (&b0)[2],
                                                  frag_t (&b1)[2]) {
    if constexpr (I < PF && I < NI) {
      load<I>(c, a0, a1, b0, b1);
      prologue<I + 1>(c, a0, a1, b0, b1);
    }
  }
  static __device__ __forceinline__ void run(const Ctx& c, f32x16 (&acc)[MAXT]) {
    frag_t a0[RING], a1[RING], b0[2], b1[2];
    SG_KLOOP_BEGIN();
    prologue<0>(c, a0, a1, b0, b1);
    step<0>(c, acc, a0, a1, b0, b1);
    SG_KLOOP_END();
  }
};

// The same tile on v_mfma_f32_16x16x32_bf16 (the board sustains a higher clock on this shape: DESIGN.md section 5).  A K step is
// 32 voxels -- a whole 32-wide row, or two 16-wide rows --: operand rows 16 channels, lane l holds channel l & 15 at voxels
// 8 (l >> 4) .. + 7 of the step (two transposing reads of 4 voxels each; a 32-lane half reads blocks 8 rows apart in the same
// columns: conflict-free).  Per step and tap: the two 16-channel halves of x (4 reads) against the two halves of dy (4 reads,
// shared by the wave's 7 taps): 4 MFMAs into the tap's four 16 x 16 accumulator tiles [ci half][co half] -- the same LDS
// bytes and accumulator registers per FLOP as the 32x32x16 form.
template <int PF, int TW = 32>
struct sg_wgrad_tile_lean16 {
  static constexpr int MAXT = 7, TH = 128 / TW, HW = TW + 2, RPS = 32 / TW;      // RPS: tile rows per K step
  // instruction slots of a K step: the two halves of dy, then (tap, ci half) -- two reads and (tap slots) two MFMAs of 16 cycles
  // each: the rhythm of the 32x32x16 loop (two reads per 32-cycle MFMA), so the same prefetch depth fits lgkmcnt's 4 bits
  static constexpr int NKS = 8, IPS = 2 + 2 * MAXT, NI = NKS * IPS, RING = PF + 1;
  static_assert(PF * 2 + 2 <= 15, "lgkmcnt is a 4-bit counter");
  typedef s16x4 frag_t;
  struct Ctx {
    int xb[2][MAXT];   // per (td, tap slot): lane part + ring slot + tap shift (the bias slot: the block's image of bf16 ones)
    int yb;            // lane part of the dy image
  };
  struct HF { frag_t p[2]; };       // one 16-channel half of an operand: voxels 0-3 and 4-7 of the lane's eight
  template <int I>
  static __device__ __forceinline__ void load(const Ctx& c, HF (&a)[RING], HF (&b)[2][2]) {
    constexpr int ks = I / IPS, r = I % IPS, td = ks / (NKS / 2), th = (ks % (NKS / 2)) * RPS;
    if constexpr (r < 2) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b[ks & 1][r].p[0]) : "v"(c.yb), "n"(ks * 2048 + r * 32));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b[ks & 1][r].p[1]) : "v"(c.yb), "n"(ks * 2048 + r * 32 + 256));
    } else {
      constexpr int t = r - 2, j = t / 2, h = t % 2, SL = (ks * 2 * MAXT + t) % RING;
      constexpr int off = th * HW * 64 + h * 32;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a[SL].p[0]) : "v"(c.xb[td][j]), "n"(off));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a[SL].p[1]) : "v"(c.xb[td][j]), "n"(off + 256));
    }
  }
  static __device__ __forceinline__ u32x4 cat(const HF& f) {
    u32x4 o;
    o[0] = __builtin_bit_cast(u32x2, f.p[0])[0]; o[1] = __builtin_bit_cast(u32x2, f.p[0])[1];
    o[2] = __builtin_bit_cast(u32x2, f.p[1])[0]; o[3] = __builtin_bit_cast(u32x2, f.p[1])[1];
    return o;
  }
  template <int I>
  static __device__ __forceinline__ void step(const Ctx& c, f32x4 (&acc)[MAXT][4], HF (&a)[RING], HF (&b)[2][2]) {
    if constexpr (I < NI) {
      if constexpr (I + PF < NI) load<I + PF>(c, a, b);
      constexpr int ks = I / IPS, r = I % IPS;
      if constexpr (r >= 2) {
        constexpr int younger = (NI - 1 - I < PF ? NI - 1 - I : PF) * 2;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
        __builtin_amdgcn_sched_barrier(0);
        constexpr int t = r - 2, j = t / 2, h = t % 2, SL = (ks * 2 * MAXT + t) % RING;
        const u32x4 af = cat(a[SL]);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          acc[j][h * 2 + hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, cat(b[ks & 1][hb])),
                                                                       acc[j][h * 2 + hb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      step<I + 1>(c, acc, a, b);
    }
  }
  template <int I>
  static __device__ __forceinline__ void prologue(const Ctx& c, HF (&a)[RING], HF (&b)[2][2]) {
    if constexpr (I < PF && I < NI) {
      load<I>(c, a, b);
      prologue<I + 1>(c, a, b);
    }
  }
  static __device__ __forceinline__ void run(const Ctx& c, f32x4 (&acc)[MAXT][4]) {
    HF a[RING], b[2][2];
    SG_KLOOP_BEGIN();
    prologue<0>(c, a, b);
    step<0>(c, acc, a, b);
    SG_KLOOP_END();
  }
};

template <bool UPS, bool DYM = false, int TW = 32, bool M16 = false>   // UPS: x is the half-resolution tensor, gathered nearest-x2 (upscale3d fused into
                                        // the layer); DYM: dy is the half-resolution gradient of a pooled layer, gathered
                                        // nearest-x2, scaled and LeakyReLU-masked while it is staged; TW = 16: the 16-wide levels
                                        // (4 x 16 x 16) in tiles of 2 x 8 x 16 voxels, halo planes of 10 x 18 rows; M16: the
                                        // K loop on v_mfma_f32_16x16x32_bf16 (sg_wgrad_tile_lean16)
__global__ __launch_bounds__(512) void conv_wgrad3l_kernel(WgradArgs a) {
  static_assert(!(UPS && DYM), "one gathered operand at a time");
  static_assert(TW == 32 || TW == 16, "tile width");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TAPS = 27, MAXT = 7;
  constexpr int TH = 128 / TW, HH = TH + 2, HW = TW + 2, PB = 208 * 64, XB = 4 * PB, YB = 256 * 64, BUF = XB + YB;
  static_assert(HH * HW <= 208, "halo plane slot");
  constexpr int XPIECES = (HH * HW + 15) / 16;       // 13 (12) 1-KiB pieces per halo plane
  // An image of bf16 ones behind the staging buffers: the bias gradient's tap slot (wave 3's spare one) reads its "x" operand there,
  // with the offsets of a real tap -- a select per operand register in the K loop (64 v_cndmask per tile and wave, in a loop
  // that is short of issue slots) cost more than these 12 KiB.
  constexpr int ONES_OFF = 2 * BUF, ONES_BYTES = 12288;
  static_assert((TH - 1) * HW * 64 + 2304 + 1024 + 12 * 64 + 64 <= ONES_BYTES, "ones image covers every fragment offset");
  for (int i = threadIdx.x * 16; i < ONES_BYTES; i += 512 * 16)
    *reinterpret_cast<u32x4*>(smem + ONES_OFF + i) = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  constexpr uint32_t DEAD = 0x80000000u;             // byte offset beyond every buffer: the DMA writes zeros
  const sg_tile_geom& g = a.g;                       // TN=1, TD=2, TH x TW = 4 x 32 | 8 x 16, HD=4, HH=TH+2, HW=TW+2 (host-checked)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int xmine = grp * BUF, ymine = xmine + XB;   // byte offsets into smem
  const int ci_t = blockIdx.y / a.coT, co_t = blockIdx.y % a.coT;
  const int D = g.D, H = g.H, W = g.W, nTd = g.nTd, cin = a.cin, cout = a.cout;

  // column schedule: XCD group xg owns a contiguous chunk of the (n, h, w) column list; group g takes columns g, g+2, ...
  const int ncol = g.nTn * g.nTh * g.nTw;
  const int xg = blockIdx.x & 7, bslot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (ncol + 7) >> 3;
  const int c_begin = xg * cpx, c_end = min(ncol, c_begin + cpx);
  const int cfirst = c_begin + bslot;
  const int ncols_blk = cfirst < c_end ? (c_end - cfirst + per_x - 1) / per_x : 0;
  const int ncols_mine = (ncols_blk + 1 - grp) >> 1;
  const int items_mine = ncols_mine * nTd;
  const int items_max = ((ncols_blk + 1) >> 1) * nTd;

  const int i16 = lane & 15, q16 = lane >> 4;
  const int qd = i16 >> 2, pp = i16 & 3;
  // 32x32x16: lane group g reads voxels 8 (g >> 1) .. + 3 of channels 16 (g & 1) ..; 16x16x32: voxels 8 g .. + 3 of the K step's 32
  // (its second 16 voxels are the next tile row for 16-wide tiles), the channel half is an instruction offset
  const int colb = M16 ? 4 * pp * 2 : (16 * (q16 & 1) + 4 * pp) * 2;
  const int kb = M16 ? 8 * (q16 & 1) + qd : 8 * (q16 >> 1) + qd;
  const int xl0 = xmine + kb * 64 + colb + (M16 ? (q16 >> 1) * (TW == 32 ? 1024 : HW * 64) : 0);
  const int yl0 = ymine + kb * 64 + colb + (M16 ? (q16 >> 1) * 1024 : 0);

  // plane-local staging tables: this lane's 16-byte pieces of 1-KiB blocks wave, wave+4, ... of a halo plane / the dy tile
  constexpr int MAXP = 4, MAXY = 4;
  uint32_t relx[MAXP], rely[MAXY], relm[DYM ? MAXY : 1];
  int crdx[MAXP], crdy[MAXY];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> 2, c = ci_t * 32 + (it & 3) * 8;
    const int hh_ = row / HW, hw = row - hh_ * HW;
    const bool live = row < HH * HW && c < cin && (wave + 4 * k) < XPIECES;
    // tile origins are even, so a halo voxel's halved coordinate is a per-lane constant relative to the tile's
    // half-resolution origin: ((h0 - 1 + hh) >> 1) = h0 / 2 + ((hh - 1) >> 1), likewise along W
    const int rel = UPS ? ((((hh_ - 1) >> 1) * (W >> 1) + ((hw - 1) >> 1)) * cin + c) * 2 : ((hh_ * W + hw) * cin + c) * 2;
    relx[k] = live ? (uint32_t)rel : 0xC0000000u;   // dead: stays >= DEAD after + tile offset
    crdx[k] = live ? (hw | (hh_ << 8)) : 0x7F7F;
  }
#pragma unroll
  for (int k = 0; k < MAXY; ++k) {
    const int it = (wave + 4 * k) * 64 + lane;
    const int row = it >> 2, c = co_t * 32 + (it & 3) * 8;   // row = (td * TH + th) * TW + tw
    const int tw = row & (TW - 1), th = (row / TW) & (TH - 1), td = row >> 7;
    if constexpr (DYM) {   // (tile origins are even: the halved coordinates are per-lane constants relative to the halved origin)
      rely[k] = c < cout ? (uint32_t)(((((th >> 1) * (W >> 1)) + (tw >> 1)) * cout + c) * 2) : 0xC0000000u;
      relm[k] = c < cout ? (uint32_t)((((td * H + th) * W + tw) * a.coT + co_t) * 4 + (it & 3)) : 0xC0000000u;
    } else {
      rely[k] = c < cout ? (uint32_t)((((td * H + th) * W + tw) * cout + c) * 2) : 0xC0000000u;
    }
    crdy[k] = th;
  }
  const int64_t svox = (int64_t)D * H * W;
  const int64_t xsb = (UPS ? svox >> 3 : svox) * cin * 2, ysb = (DYM ? svox >> 3 : svox) * cout * 2, msb = svox * a.coT * 4;
  const uint32_t xplane = (uint32_t)((UPS ? (H >> 1) * (W >> 1) : H * W) * cin * 2),
                 yplane = (uint32_t)((DYM ? (H >> 1) * (W >> 1) : H * W) * cout * 2), mplane = (uint32_t)(H * W * a.coT * 4);

  // cursor over my tiles: column cj of my list, step di along D; per column: resources and lane offsets
  int cj = 0, di = 0;
  __amdgpu_buffer_rsrc_t rx, ry, rm;
  uint32_t vkx[MAXP], vky[MAXY], vkm[DYM ? MAXY : 1];
  auto enter_column = [&]() {
    const int col = cfirst + (2 * cj + grp) * per_x;
    const int c1 = (int)sg_div((uint32_t)col, g.fnTw);
    const int w0 = (col - c1 * g.nTw) * TW;
    const int c2 = (int)sg_div((uint32_t)c1, g.fnTh);
    const int h0 = (c1 - c2 * g.nTh) * TH;
    const int n0 = c2;
    rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + n0 * xsb, 0, (int)xsb, 0x00020000);
    ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.dy)) + n0 * ysb, 0, (int)ysb, 0x00020000);
    if constexpr (DYM)
      rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.dy_mask)) + n0 * msb, 0, (int)msb, 0x00020000);
    const int tile_off = UPS ? ((h0 >> 1) * (W >> 1) + (w0 >> 1)) * cin * 2
                             : ((h0 - 1) * W + (w0 - 1)) * cin * 2;   // may be negative: only dead lanes go below 0
    const int lo_w = max(0, 1 - w0), hi_w = min(HW, W + 1 - w0) - 1;
    const int lo_h = max(0, 1 - h0), hi_h = min(HH, H + 1 - h0) - 1;
    const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8));
    const uint32_t hi = (uint32_t)(hi_w | (hi_h << 8)) | 0x8080u;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const uint32_t c_ = (uint32_t)crdx[k];
      const uint32_t t1 = (c_ | 0x8080u) - lo, t2 = hi - c_;
      vkx[k] = (t1 & t2 & 0x8080u) == 0x8080u ? relx[k] + (uint32_t)tile_off : DEAD;
    }
    const int col_off = DYM ? ((h0 >> 1) * (W >> 1) + (w0 >> 1)) * cout * 2 : (h0 * W + w0) * cout * 2;
    const int rows_left = H - h0;                                // W is a multiple of TW (host-checked): every tw is inside
#pragma unroll
    for (int k = 0; k < MAXY; ++k) {
      const bool in = crdy[k] < rows_left && rely[k] < DEAD;
      vky[k] = in ? rely[k] + (uint32_t)col_off : DEAD;
      if constexpr (DYM) vkm[k] = in ? relm[k] + (uint32_t)((h0 * W + w0) * a.coT * 4) : DEAD;
    }
  };
  const float gain_y = a.dy_gain, slope_y = a.dy_mask_slope;     // (DYM) sg_mask_piece_bf16: the arithmetic of sg_upscale2x_masked
  // staging of the cursor's tile: the new halo planes (all four at the bottom of a column) and the dy tile
  auto stage = [&]() __attribute__((always_inline)) {
    const int d0 = 2 * di;
    // (DYM) the dy tile goes through registers and is requested first: its latency runs beside the halo DMA issue.  Both D
    // planes of the tile (d0 is even) read half-resolution plane d0 / 2.
    u32x4 sy[DYM ? MAXY : 1];
    uint32_t my[DYM ? MAXY : 1];
    if constexpr (DYM) {
      const uint32_t ysoff = (uint32_t)(d0 >> 1) * yplane, msoff = (uint32_t)d0 * mplane;
#pragma unroll
      for (int k = 0; k < MAXY; ++k) {
        const bool ok = d0 + (k >> 1) < D;
        sy[k] = __builtin_amdgcn_raw_buffer_load_b128(ry, ok ? vky[k] : DEAD, ysoff, 0);
        my[k] = __builtin_amdgcn_raw_buffer_load_b8(rm, ok ? vkm[k] : DEAD, msoff, 0);
      }
    }
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
      if (hd < 2 && di != 0) continue;                           // uniform
      const int gp = d0 - 1 + hd;
      const bool plane_ok = gp >= 0 && gp < D;
      const uint32_t soff = plane_ok ? (uint32_t)(UPS ? gp >> 1 : gp) * xplane : 0u;
      char* dst = smem + xmine + ((gp + 8) & 3) * PB;
#pragma unroll
      for (int k = 0; k < MAXP; ++k)
        if (wave + 4 * k < XPIECES)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(dst + (wave + 4 * k) * 1024), 16, plane_ok ? vkx[k] : DEAD, soff, 0, 0);
    }
    if constexpr (DYM) {
#pragma unroll
      for (int k = 0; k < MAXY; ++k)
        *reinterpret_cast<u32x4*>(smem + ymine + (wave + 4 * k) * 1024 + lane * 16) = sg_mask_piece_bf16(sy[k], my[k], gain_y, slope_y);
    } else {
      const uint32_t ysoff = (uint32_t)d0 * yplane;
#pragma unroll
      for (int k = 0; k < MAXY; ++k) {
        const bool ok = d0 + (k >> 1) < D;                          // pieces 8..15 are the tile's second D plane
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_ptr_t)(smem + ymine + (wave + 4 * k) * 1024), 16, ok ? vky[k] : DEAD, ysoff, 0, 0);
      }
    }
  };
  auto advance = [&]() {
    if (++di == nTd) {
      di = 0;
      ++cj;
      if (cj < ncols_mine) enter_column();
    }
  };

  // per-wave taps
  int tap_hw[MAXT], tap_kd[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int tap = wave + 4 * j;
    const int kw_i = tap % 3, kh_i = (tap / 3) % 3, kd_i = tap / 9;
    tap_hw[j] = tap < TAPS ? (kh_i * HW + kw_i) * 64 : 0;
    tap_kd[j] = tap < TAPS ? kd_i : 0;
  }
  // bias gradient: wave 3's last slot is spare (27 = 4*7 - 1 taps); it multiplies dy by ones
  const int ones_last = (a.dbias != nullptr && ci_t == 0 && wave == 3) ? 1 : 0;
  // 16 accumulator registers per tap: one 32 x 32 tile, or four 16 x 16 tiles [ci half * 2 + co half]
  typename std::conditional<M16, f32x4[MAXT][4], f32x16[MAXT]>::type acc;
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (M16) acc[j][i >> 2][i & 3] = 0.f;
      else acc[j][i] = 0.f;
    }

  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && lane == 0 && wave == 0 && dbgi < 120)
      a.dbg[grp * 128 + dbgi] = __builtin_amdgcn_s_memtime();
    ++dbgi;
  };
  typedef typename std::conditional<M16, sg_wgrad_tile_lean16<5, TW>, sg_wgrad_tile_lean<5, TW>>::type KT;
  auto mfma_phase = [&](int q) {
    const int dq = q % nTd;
    const int pbase = 2 * dq - 1 + 8;                             // plane of halo index 0 (kept non-negative)
    typename KT::Ctx c;
    c.yb = yl0;
#pragma unroll
    for (int td = 0; td < 2; ++td)
#pragma unroll
      for (int j = 0; j < MAXT; ++j) c.xb[td][j] = xl0 + ((pbase + td + tap_kd[j]) & 3) * PB + tap_hw[j];
    if (ones_last) c.xb[0][MAXT - 1] = c.xb[1][MAXT - 1] = ONES_OFF + (xl0 - xmine);
    KT::run(c, acc);
  };
  const bool stage_first_only = (a.dbg_flags & 1) != 0;
  auto off_phase = [&](int qn) {                                  // stage my tile qn (the cursor's)
    if (qn < items_mine && !(stage_first_only && qn >= 1)) stage();
    stamp();
    if (qn < items_mine) advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the DMA has landed before the barrier releases the readers
  };

  if (items_mine > 0) enter_column();
  if (grp == 0) {
    off_phase(0);
    __syncthreads();
    for (int q = 0; q < items_max; ++q) {
      stamp();
      if (q < items_mine) mfma_phase(q);
      stamp();
      __syncthreads();
      stamp();
      off_phase(q + 1);
      stamp();
      __syncthreads();
    }
  } else {
    __syncthreads();
    for (int q = 0; q < items_max; ++q) {
      stamp();
      off_phase(q);
      stamp();
      __syncthreads();
      stamp();
      if (q < items_mine) mfma_phase(q);
      stamp();
      __syncthreads();
    }
  }
  // The two groups' sums are added through LDS (every tile is done and the loop ended on a barrier: the staging buffers are
  // free), so a block sends ONE set of partial sums to memory.  Measured at 4 x 16 x 16, 128 -> 128, batch 32 (256 blocks): the
  // f32 atomics of 2 x 256 groups were 32 of the kernel's 68 us (tools/w16_probe.py, SG_DETERMINISTIC A/B).
  {
    char* cbuf = smem + ((wave * MAXT) << 12) + lane * 16;
    if (grp == 1) {
#pragma unroll
      for (int j = 0; j < MAXT; ++j)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          if constexpr (M16) *reinterpret_cast<f32x4*>(cbuf + ((j * 4 + i4) << 10)) = acc[j][i4];
          else *reinterpret_cast<f32x4*>(cbuf + ((j * 4 + i4) << 10)) = f32x4{acc[j][4 * i4], acc[j][4 * i4 + 1], acc[j][4 * i4 + 2], acc[j][4 * i4 + 3]};
        }
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(cbuf + ((j * 4 + i4) << 10));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (M16) acc[j][i4][e] += o[e];
          else acc[j][4 * i4 + e] += o[e];
        }
      }
  }
  if constexpr (M16) {
    // D tile [ci half][co half]: lane l holds ci = 16 ha + 4 (l >> 4) + e, co = 16 hb + (l & 15)
    const int c16 = lane & 15, q4 = lane >> 4;
    if (ones_last && (items_mine > 0 || a.slab != 0) && q4 == 0) {      // row 0 of the ones product = column sums
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
        if (co_t * 32 + 16 * hb + c16 < cout)
          sg_wg_out(a.dbias + (int64_t)blockIdx.x * a.bslab + co_t * 32 + 16 * hb + c16, acc[MAXT - 1][hb][0], a.slab != 0);
    }
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int tap = wave + 4 * j;
      if (tap < TAPS && (items_mine > 0 || a.slab != 0)) {
        float* dst = a.dwt + (int64_t)blockIdx.x * a.slab + ((((int64_t)tap * a.ciT + ci_t) * a.coT + co_t) << 10);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            sg_wg_out(dst + (16 * (t >> 1) + 4 * q4 + e) * 32 + 16 * (t & 1) + c16, acc[j][t][e], a.slab != 0);
      }
    }
  } else {
    const int r = lane & 31, hh = lane >> 5;
    if (ones_last && (items_mine > 0 || a.slab != 0) && hh == 0 && co_t * 32 + r < cout)   // row 0 of the ones product = column sums
      sg_wg_out(a.dbias + (int64_t)blockIdx.x * a.bslab + co_t * 32 + r, acc[MAXT - 1][0], a.slab != 0);
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int tap = wave + 4 * j;
      if (tap < TAPS && (items_mine > 0 || a.slab != 0)) {
        float* dst = a.dwt + (int64_t)blockIdx.x * a.slab + ((((int64_t)tap * a.ciT + ci_t) * a.coT + co_t) << 10);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
          sg_wg_out(dst + row * 32 + r, acc[j][i], a.slab != 0);
        }
      }
    }
  }
}

template <int KD, int KH, int KW>
static int launch_wgrad3(WgradArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  // 16-wide levels (4 x 16 x 16): the lean kernel in tiles of 2 x 8 x 16 voxels
  const bool w16 = KD == 3 && KH == 3 && KW == 3 && s->w == 16 && (s->h % 8) == 0 && (s->d % 2) == 0 && !a.dy_mask &&
                   !sg_cfg().wgrad_no_lean && !sg_cfg().wgrad_no_w16;
  a.g = sg_make_geom(s, 256, /*prefer_w32=*/true, /*td=*/2, /*th=*/w16 ? 8 : 4);
  const sg_tile_geom& g = a.g;
  if (g.TW != (w16 ? 16 : 32) || g.TN != 1 || g.TD * g.TH * g.TW != 256) return SG_OK;
  if (g.HD != 2 && g.HD != 4) return SG_OK;          // ring slots are addressed with a mask
  if (g.TD != 2 || g.TH != (w16 ? 8 : 4)) return SG_OK;   // the unrolled K loops are written for 2 x 4 x 32 and 2 x 8 x 16 tiles
  if (g.HH > 127 || g.HW > 127) return SG_OK;
  if ((s->cin % 8) || (s->cout % 8)) return SG_OK;
  // staging offsets are relative to the tile's first sample (64-bit tile bases): TN samples must fit 31 bits
  if ((int64_t)g.TN * s->d * s->h * s->w * (int64_t)(s->cin > s->cout ? s->cin : s->cout) >= (1ll << 31)) return SG_OK;
  const int ncol = g.nTn * g.nTh * g.nTw;
  const int pairs = a.ciT * a.coT;
  int gx = (256 / pairs) / 8 * 8;
  if (gx < 8) gx = 8;
  // few columns (the 16-wide levels; small batches at the wide ones): rather fewer blocks (>= 64) than the tap-per-wave kernel,
  // which runs these shapes at 0.2-0.4 of this kernel's rate
  while (gx > 8 && ncol < 2 * gx && (gx - 8) * pairs >= (w16 ? 128 : 64)) gx -= 8;
  // Small batches (round 5): fewer than two columns per block is still this kernel's case -- a block whose second wave group (or
  // whose every group) has no column idles through the barriers and sends nothing (items_mine == 0) -- because the tap-per-wave
  // kernel it would fall back to takes 33-133 us per launch on the 4 x 16 x 16 level at batch 2 (this one: one column of two
  // tiles per active block).  SG_WGRAD3L_MIN_COLS (default 2) columns are needed at least; 16 = the round-4 rule.
  const int min_cols = sg_cfg().wgrad3l_min_cols > 0 ? sg_cfg().wgrad3l_min_cols : 2;
  if (ncol < (2 * gx < min_cols ? 2 * gx : min_cols) || g.nTd < 2) return SG_OK;      // something to slide over
  a.gy = a.g;
  a.ntiles = ncol * g.nTd;
  a.rs = 64;
  a.plane_rows = (g.HH * g.HW + 15) & ~15;
  if (a.plane_rows / 16 > 16) return SG_OK;           // <= 4 pieces per wave per plane
  a.xbytes = g.HD * a.plane_rows * 64;
  a.ybytes = 256 * 64;
  size_t lds = 2ull * (a.xbytes + a.ybytes);
  if (lds > 160 * 1024) return SG_OK;
  a.tap0 = 0; a.taps_blk = a.taps;
  a.nslab = 2 * gx;      // (both wave groups of a block keep sums of their own)
  // the lean variant: 3x3x3 without fused up-sampling, whole 32-wide rows, one sample of either tensor below 2 GiB
  const bool lean = KD == 3 && KH == 3 && KW == 3 && g.HH == 6 && g.HW == 34 && g.HD == 4 && a.plane_rows == 208 &&
                    s->w % 32 == 0 && !sg_cfg().wgrad_no_lean && (!g.ups || ((s->d | s->h | s->w) & 1) == 0);
  const bool lean16 = w16 && g.HH == 10 && g.HW == 18 && g.HD == 4 && (!g.ups || ((s->d | s->h | s->w) & 1) == 0);
  if (w16 && !lean16) return SG_OK;
  if (lean16 || lean) a.nslab = gx;                   // (the lean kernels add their two groups' sums before they leave the block)
  if (lean) lds += 12288;                             // (+ the lean kernels' image of ones, behind the staging buffers)
  const bool m16 = sg_cfg().wgrad3l_16 != 0;      // the K loop on v_mfma_f32_16x16x32_bf16
#define SG_LAUNCH_WGRAD3L(UPS_, DYM_, TW_, NAME_)                                                                     \
  do {                                                                                                               \
    if (m16) {                                                                                                       \
      auto kern = conv_wgrad3l_kernel<UPS_, DYM_, TW_, true>;                                                        \
      SG_ALLOW_160K_LDS(kern);                                                                                       \
      SG_KNAME(NAME_);                                                                                               \
      hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)pairs), dim3(512), lds, st, a);                          \
    } else {                                                                                                         \
      auto kern = conv_wgrad3l_kernel<UPS_, DYM_, TW_, false>;                                                       \
      SG_ALLOW_160K_LDS(kern);                                                                                       \
      SG_KNAME(NAME_);                                                                                               \
      hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)pairs), dim3(512), lds, st, a);                          \
    }                                                                                                                \
  } while (0)
  if (lean16) {        // (halo plane slots of 208 rows like the 32-wide kernel's)
    lds = 2ull * (4 * 208 * 64 + 256 * 64) + 12288;   // (+ the image of ones)
    if (g.ups) SG_LAUNCH_WGRAD3L(true, false, 16, "conv_wgrad3l<ups,w16>");
    else SG_LAUNCH_WGRAD3L(false, false, 16, "conv_wgrad3l<w16>");
  } else if (a.dy_mask) {     // half-resolution dy, gathered and masked while staged: the lean kernel only
    if (!lean || g.ups || ((s->d | s->h | s->w) & 1)) return SG_OK;
    SG_LAUNCH_WGRAD3L(false, true, 32, "conv_wgrad3l<dy gather>");
  } else if (lean && g.ups) {
    SG_LAUNCH_WGRAD3L(true, false, 32, "conv_wgrad3l<ups>");
  } else if (lean) {
    SG_LAUNCH_WGRAD3L(false, false, 32, "conv_wgrad3l");
#undef SG_LAUNCH_WGRAD3L
  } else {
    auto kern = conv_wgrad3_kernel<KD, KH, KW>;
    SG_ALLOW_160K_LDS(kern);
    SG_KNAME("conv_wgrad3<%d,%d,%d>", KD, KH, KW);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)pairs), dim3(512), lds, st, a);
  }
  SG_LAUNCH_CHECK();
  *used = true;
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------------
// wgrad for the low-resolution 1x3x3 levels (bf16, H = W = 4 or 8: the 512 -> 512 layers of the 1x4x4 and 2x8x8 levels).
// The generic kernel above spends 14 us per 256-voxel tile there: its staging is issued with 64-bit pointer arithmetic per
// piece, lands while nobody computes, and three of its four waves own two taps where the fourth owns three.  Here a tile is
// 128 voxels = P whole (n, d) planes, so every tile has the SAME staging plan: per-lane buffer offsets computed once, a
// scalar offset per tile, LDS-DMA with hardware zero fill for the halo rows.  Tiles go through a ring of three LDS
// buffers, requested two tiles ahead; the four waves split a tile's eight K steps (all nine taps each: 18 MFMAs per
// wave and tile) and add their sums to the workspace on their own at the end.
// ------------------------------------------------------------------------------------------------------
struct WgradPlanesArgs {
  const void* x;
  const void* dy;
  float* dwt;
  float* dbias;              // optional: column sums of dy, by the ci_t == 0 blocks
  int cin, cout, ciT, coT;
  int ntiles;                // (N * D) / P
  int64_t slab, bslab;       // elements between per-block slabs; 0: atomics into one buffer
};

template <int HW_>
__device__ __forceinline__ void conv_wgrad_planes_body(const WgradPlanesArgs& a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = HW_, W = HW_, P = 128 / (H * W), RP = (H + 2) * (W + 2);
  constexpr int XROWS = P * RP, NPX = (XROWS * 4 + 255) / 256, NPY = 2, PPT = NPX + NPY;      // DMA instructions per thread and tile
  constexpr int XB = NPX * 4096, YB = 128 * 64, BUF = XB + YB;      // (every thread moves NPX pieces: the image is padded to whole rounds)
  constexpr uint32_t DEAD = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ci_t = blockIdx.y / a.coT, co_t = blockIdx.y % a.coT;
  const int cin = a.cin, cout = a.cout;
  // staging plan (the same for every tile)
  uint32_t vx[NPX], vy[NPY];
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int it = tid + 256 * k;
    const int row = it >> 2, sub = it & 3;
    const int p = row / RP, rr = row - p * RP;
    const int hh = rr / (W + 2), hw = rr - hh * (W + 2);
    const bool in = row < XROWS && hh >= 1 && hh <= H && hw >= 1 && hw <= W;
    vx[k] = in ? (uint32_t)((((p * H + hh - 1) * W + hw - 1) * cin + ci_t * 32 + sub * 8) * 2) : DEAD;
  }
#pragma unroll
  for (int k = 0; k < NPY; ++k) {
    const int it = tid + 256 * k;
    vy[k] = (uint32_t)(((it >> 2) * cout + co_t * 32 + (it & 3) * 8) * 2);
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)((int64_t)a.ntiles * 128 * cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)((int64_t)a.ntiles * 128 * cout * 2), 0x00020000);
  auto stage = [&](int t, int b) __attribute__((always_inline)) {      // tile t into ring buffer b; t beyond the list: zeros, same instruction count
    const bool ok = t < a.ntiles;
    const uint32_t sx = ok ? (uint32_t)t * (uint32_t)(128 * cin * 2) : 0u, sy = ok ? (uint32_t)t * (uint32_t)(128 * cout * 2) : 0u;
    char* xb = smem + b * BUF;
#pragma unroll
    for (int k = 0; k < NPX; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xb + (tid + 256 * k) * 16 - lane * 16), 16, ok ? vx[k] : DEAD, sx, 0, 0);
#pragma unroll
    for (int k = 0; k < NPY; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_ptr_t)(xb + XB + (tid + 256 * k) * 16 - lane * 16), 16, ok ? vy[k] : DEAD, sy, 0, 0);
  };
  // transposing-read lane map (as in the kernels above): 16-lane group = 4 voxel rows x 16 channels
  const int i16 = lane & 15, q16 = lane >> 4;
  const int colb = (16 * (q16 & 1) + 4 * (i16 & 3)) * 2;
  const int kb = 8 * (q16 >> 1) + (i16 >> 2);
  // Waves 0..2 own one kernel row each (three taps, all eight K steps of a tile); wave 3 multiplies dy by ones (the bias
  // gradient) in the ci_t == 0 blocks.  (Splitting the K steps over four waves instead was measured first: every wave then
  // adds all nine tiles to the workspace, 4 x the atomics, and the launch took 81 us where this layout's MFMAs take 8.)
  int xr[2], yr[2];            // rows m0 / m0 + 4 of K step 0: byte offsets inside a buffer; K step ks adds a compile-time term
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = kb + 4 * j;                                    // < 16: inside plane 0
    xr[j] = ((m / W + (wave < 3 ? wave : 0)) * (W + 2) + m % W) * 64 + colb;      // halo row of tap (kh = wave, kw = 0)
    yr[j] = XB + m * 64 + colb;
  }
  const bool ones = wave == 3 && a.dbias != nullptr && ci_t == 0;
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const u32x4 one8 = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};      // bf16 1.0 x 8

  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
  const int t0 = blockIdx.x, G = gridDim.x;
  stage(t0, 0);
  stage(t0 + G, 1);
  int b = 0;
  for (int t = t0; t < a.ntiles; t += G) {
    const int b2 = b >= 1 ? b - 1 : 2;                         // (b + 2) % 3
    stage(t + 2 * G, b2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");     // tile t has landed (the two younger tiles may still fly)
    __syncthreads();
    const char* base = smem + b * BUF;
    if (wave < 3 || ones) {                                     // (uniform per wave)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        // 16 voxels further on: two rows of an 8 x 8 plane (a new plane every four steps) / one whole 4 x 4 plane
        const int xoff = H == 8 ? ((ks >> 2) * RP + (ks & 3) * 2 * (W + 2)) * 64 : ks * RP * 64;
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + yr[0] + ks * 1024));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + yr[1] + ks * 1024));
        u32x4 bf;
        bf[0] = __builtin_bit_cast(u32x2, b0)[0]; bf[1] = __builtin_bit_cast(u32x2, b0)[1];
        bf[2] = __builtin_bit_cast(u32x2, b1)[0]; bf[3] = __builtin_bit_cast(u32x2, b1)[1];
        if (wave < 3) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + xr[0] + xoff + kw * 64));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + xr[1] + xoff + kw * 64));
            u32x4 af;
            af[0] = __builtin_bit_cast(u32x2, a0)[0]; af[1] = __builtin_bit_cast(u32x2, a0)[1];
            af[2] = __builtin_bit_cast(u32x2, a1)[0]; af[3] = __builtin_bit_cast(u32x2, a1)[1];
            acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[kw], 0, 0, 0);
          }
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, one8), __builtin_bit_cast(bf16x8, bf), acc[0], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                             // buffer b is free for the tile three further on
    b = b == 2 ? 0 : b + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the zero-fill DMAs of the tail have landed before the block ends)
  const int r = lane & 31, hh = lane >> 5;
  const bool slab = a.slab != 0;
  if (wave < 3) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      float* dst = a.dwt + (int64_t)blockIdx.x * a.slab + ((((int64_t)(wave * 3 + kw) * a.ciT + ci_t) * a.coT + co_t) << 10);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
        sg_wg_out(dst + row * 32 + r, acc[kw][i], slab);
      }
    }
  } else if (ones && hh == 0) {                                  // row 0 of the ones product = the column sums of dy
    sg_wg_out(a.dbias + (int64_t)blockIdx.x * a.bslab + co_t * 32 + r, acc[0][0], slab);
  }
}

// (plain kernels around a templated body: as a kernel TEMPLATE the host pass of this file under -save-temps left the stubs as
// comdat declarations and aborted with "Broken module found")
__global__ __launch_bounds__(256, 2) void conv_wgrad_planes8_kernel(WgradPlanesArgs a) { conv_wgrad_planes_body<8>(a); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_planes4_kernel(WgradPlanesArgs a) { conv_wgrad_planes_body<4>(a); }

static bool wgrad_planes_eligible(const sg_conv_shape* s) {
  if (s->kd != 1 || s->kh != 3 || s->kw != 3 || s->upsample_in || s->h != s->w || (s->h != 4 && s->h != 8)) return false;
  if ((s->cin % 32) || (s->cout % 32)) return false;
  const int P = 128 / (s->h * s->w);
  const int64_t planes = (int64_t)s->n * s->d;
  if (planes % P) return false;
  const int64_t vox = planes * s->h * s->w;
  return vox * (s->cin > s->cout ? s->cin : s->cout) * 2 < (1ll << 31);
}

static int launch_wgrad_planes(WgradArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  if (!wgrad_planes_eligible(s)) return SG_OK;
  WgradPlanesArgs p;
  p.x = a.x; p.dy = a.dy; p.dwt = a.dwt; p.dbias = a.dbias; p.cin = s->cin; p.cout = s->cout; p.ciT = a.ciT; p.coT = a.coT;
  p.ntiles = (int)((int64_t)s->n * s->d * s->h * s->w / 128);
  p.slab = a.slab; p.bslab = a.bslab;
  const int pairs = a.ciT * a.coT;
  int gx = sg_cdiv(512, pairs);                // two blocks per CU over all pairs
  if (gx > 16) gx = 16;                        // (one slab per block: stays within wgrad_slab_count)
  if (gx > p.ntiles) gx = p.ntiles;
  a.nslab = gx;
  if (s->h == 8) {
    auto kern = conv_wgrad_planes8_kernel;
    SG_ALLOW_160K_LDS(kern);
    SG_KNAME("conv_wgrad_planes<8>");
    constexpr int XB = ((2 * 100 * 4 + 255) / 256) * 4096;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)pairs), dim3(256), 3 * (size_t)(XB + 8192), st, p);
  } else {
    auto kern = conv_wgrad_planes4_kernel;
    SG_ALLOW_160K_LDS(kern);
    SG_KNAME("conv_wgrad_planes<4>");
    constexpr int XB = ((8 * 36 * 4 + 255) / 256) * 4096;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)pairs), dim3(256), 3 * (size_t)(XB + 8192), st, p);
  }
  SG_LAUNCH_CHECK();
  *used = true;
  return SG_OK;
}

static size_t wgrad_tile_bytes(const sg_conv_shape* s) {
  const size_t need = (size_t)(s->kd * s->kh * s->kw) * sg_cdiv(s->cin, 32) * sg_cdiv(s->cout, 32) * 4096;
  const bool pointwise = s->kd * s->kh * s->kw == 1 && (s->cin <= 4 || s->cout <= 4);
  const size_t pw = pointwise ? (size_t)PW_WGRAD_MAX_BLOCKS * 5 * (size_t)(s->cin > s->cout ? s->cin : s->cout) * sizeof(float) : 0;
  return ((need > pw ? need : pw) + 255) & ~(size_t)255;
}

extern "C" int sg_conv3d_wgrad(const void* x, const void* dy, float* dw, float coef, void* workspace,
                               size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st) {
  return sg_conv3d_wgrad_bias(x, dy, dw, nullptr, coef, workspace, workspace_bytes, s, dt, st);
}

// pw_dx / pw_wmat: sg_conv3d_pw_bwd's extra output and operand (pointwise layers from <= 4 input channels only)
static int wgrad_bias_impl(const void* x, const void* dy, float* dw, float* dbias, float coef, void* workspace,
                           size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st, void* pw_dx,
                           const float* pw_wmat, const void* dy_mask = nullptr, float dy_mask_slope = 0.f, float dy_gain = 1.f,
                           unsigned flags = 0) {
  const int accumulate = (flags & SG_WGRAD_ACCUMULATE) ? 1 : 0;
  const bool clean_ws = (flags & SG_WGRAD_CLEAN_WORKSPACE) != 0;
  if (!conv_shape_ok_w(s) || !x || !dy || !dw || !workspace) return SG_EINVAL;
  if (!sg_aligned16(x) || !sg_aligned16(dy) || !sg_aligned16(workspace)) return SG_EALIGN;
  if (dy_mask && (dt != SG_BF16 || s->kd != 3 || s->kh != 3 || s->kw != 3 || s->upsample_in || (s->cout % 32) ||
                  sg_cfg().wgrad_v1 || sg_cfg().wgrad_no_v3 || !sg_is_pow2f(dy_gain)))
    return SG_EUNSUPPORTED;
  const size_t need = sg_conv3d_wgrad_workspace(s, dt);
  if (workspace_bytes < need) return SG_EWORKSPACE;
  hipStream_t hs = sg_st(st);
  sg_prof_scope prof(1, s, dt, hs);
  {  // pointwise conv with <= 4 channels on one side: column reduction
    const int taps1 = s->kd * s->kh * s->kw;
    const int cs = s->cin < s->cout ? s->cin : s->cout, cb = s->cin < s->cout ? s->cout : s->cin;
    const int E = dt == SG_BF16 ? 8 : 4;
    if (taps1 == 1 && cs <= 4 && !s->upsample_in && cb % E == 0 && cb / E <= 256 && 256 % (cb / E) == 0) {
      if (accumulate || clean_ws) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }   // (the tile finalize below is the one that accumulates / cleans)
      const int small_is_cin = s->cin <= s->cout ? 1 : 0;
      const void* sm = small_is_cin ? x : dy;
      const void* bg = small_is_cin ? dy : x;
      const int64_t nvox = (int64_t)s->n * s->d * s->h * s->w;
      const int rows = 256 / (cb / E);
      int64_t nb = (nvox + rows - 1) / rows;
      // (one round of four blocks per CU: measured at 32 x 128 x 128, 768 / 1024 / 1536 / 2048 blocks read 4.7 / 5.2 / 4.9 /
      // 3.9 TB/s -- every block ends in an LDS reduction and a row of partial sums, tools/archive/pww_probe.py)
      if (nb > PW_WGRAD_MAX_BLOCKS) nb = PW_WGRAD_MAX_BLOCKS;
      float* part = reinterpret_cast<float*>(workspace);
      const int ones = (dbias != nullptr && small_is_cin) ? 1 : 0;   // the big side is dy: its column sums are the bias gradient
      SG_KNAME("pw_wgrad_partial");
      if (pw_dx && !small_is_cin) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }
      if (dt == SG_BF16 && cs == 1)
        hipLaunchKernelGGL((pw_wgrad_partial_kernel<bf16_t, 1>), dim3((unsigned)nb), dim3(256), 0, hs, (const bf16_t*)sm,
                           (const bf16_t*)bg, part, nvox, cs, cb, ones, (bf16_t*)pw_dx, pw_wmat);
      else if (dt == SG_BF16)
        hipLaunchKernelGGL(pw_wgrad_partial_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, hs, (const bf16_t*)sm,
                           (const bf16_t*)bg, part, nvox, cs, cb, ones, (bf16_t*)pw_dx, pw_wmat);
      else
        hipLaunchKernelGGL(pw_wgrad_partial_kernel<float>, dim3((unsigned)nb), dim3(256), 0, hs, (const float*)sm,
                           (const float*)bg, part, nvox, cs, cb, ones, (float*)pw_dx, pw_wmat);
      hipLaunchKernelGGL(pw_wgrad_final_kernel, dim3((unsigned)(((cs + ones) * cb + 31) / 32)), dim3(256), 0, hs, part,
                         dw, coef, (int)nb, cs, cb, small_is_cin, ones ? dbias : nullptr);
      int e0 = (int)hipGetLastError();
      if (e0 == 0 && dbias && !ones)   // bias lives on the small side (to_rgb): plain column sum of dy
        e0 = sg_bias_act_bwd(dy, nullptr, nullptr, dbias, reinterpret_cast<char*>(workspace) + wgrad_tile_bytes(s), nvox,
                             s->cout, 0.f, dt, st);
      prof.done(e0);
      return e0;
    }
  }
  if (pw_dx) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }   // only the pointwise pass has the extra output
  if (sg_small_wgrad_eligible(s) && !sg_cfg().no_small) {   // 2-D top levels (<= 16 channels): VALU kernel, slab reduction
    if (accumulate || clean_ws) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }
    const int rc_s = sg_small_wgrad(x, dy, dw, dbias, coef, workspace, workspace_bytes, s, dt, hs);
    prof.done(rc_s);
    return rc_s;
  }
  const size_t tile_bytes = (size_t)(s->kd * s->kh * s->kw) * sg_cdiv(s->cin, 32) * sg_cdiv(s->cout, 32) * 4096;
  const bool det = sg_cfg().deterministic != 0;
  const size_t ns_max = det ? (size_t)wgrad_slab_count(s) : 1;
  hipError_t e = hipSuccess;
  // The kernels add the bias gradient to a staging row right behind the tile (inside the workspace: the fallback pass's
  // region starts no later and is only used when no kernel summed the bias), so that ONE memset clears both; the
  // finalize kernel copies it out.  (Two memsets per weight gradient were ~70 launches per step.)
  const size_t tile_al = (tile_bytes + 255) & ~(size_t)255;
  float* bias_staged = (!det && dbias) ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + tile_al) : nullptr;
  if (bias_staged && tile_al + (size_t)s->cout * sizeof(float) > ns_max * wgrad_tile_bytes(s) + sg_bias_act_bwd_workspace(s->cout)) {
    prof.done(SG_EWORKSPACE);
    return SG_EWORKSPACE;
  }
  if (!det && !clean_ws)      // (reproducible mode: every block of the grid stores its whole slab, nothing to clear)
    e = hipMemsetAsync(workspace, 0, bias_staged ? tile_al + (size_t)s->cout * sizeof(float) : tile_bytes, hs);
  if (e != hipSuccess) { prof.done((int)e); return (int)e; }
  WgradArgs a;
  a.dbias = bias_staged ? bias_staged : dbias;
  a.slab = det ? (int64_t)(wgrad_tile_bytes(s) / 4) : 0;
  a.bslab = det ? (int64_t)(wgrad_bias_slab_bytes(s) / 4) : 0;
  a.nslab = 1;
  float* bias_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ns_max * wgrad_tile_bytes(s) +
                                               sg_bias_act_bwd_workspace(s->cout));
  if (det && dbias) a.dbias = bias_slabs;
  bool db_done = false;
  a.dbg_flags = sg_cfg().dbg_flags;
  a.dbg = g_dbg_ts;
  a.x = x; a.dy = dy; a.dwt = reinterpret_cast<float*>(workspace);
  a.dy_mask = reinterpret_cast<const uint32_t*>(dy_mask); a.dy_mask_slope = dy_mask_slope; a.dy_gain = dy_gain;
  a.cin = s->cin; a.cout = s->cout;
  a.taps = s->kd * s->kh * s->kw; a.kh = s->kh; a.kw = s->kw;
  a.ciT = sg_cdiv(s->cin, 32); a.coT = sg_cdiv(s->cout, 32);
  int rc = SG_OK;
  bool used = false;
  if (dt == SG_BF16 && !sg_cfg().wgrad_v1 && !sg_cfg().wgrad_no_v3) {
    if (s->kd == 3 && s->kh == 3 && s->kw == 3) rc = launch_wgrad3<3, 3, 3>(a, s, hs, &used);
    db_done = used;   // the sliding-halo kernel accumulates the bias gradient in its spare tap slot
  }
  if (dy_mask && rc == SG_OK && !used) { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }   // only that kernel gathers dy
  if (rc == SG_OK && !used && dt == SG_BF16 && !sg_cfg().wgrad_v1 && !sg_cfg().no_gemm)
  {
    rc = launch_wgrad_planes(a, s, hs, &used);      // 1x3x3 layers of the 4^2 / 8^2 levels: whole-plane tiles through an LDS ring
    if (used) db_done = true;                       // (its fourth wave sums dy)
  }
  if (rc == SG_OK && !used && dt == SG_BF16 && !sg_cfg().wgrad_v1) {
    if (s->kd == 3 && s->kh == 3 && s->kw == 3) rc = launch_wgrad2<3, 3, 3>(a, s, hs, &used);
    else if (s->kd == 1 && s->kh == 3 && s->kw == 3) rc = launch_wgrad2<1, 3, 3>(a, s, hs, &used);
  }
  if (rc == SG_OK && used) { /* done by the ping-pong kernel */ }
  else if (rc != SG_OK) { /* fall through to the error return below */ }
  else if (dt == SG_BF16) rc = launch_wgrad<bf16_t, 256>(a, s, hs);
  else if (dt == SG_F32) rc = launch_wgrad<float, 128>(a, s, hs);
  else rc = SG_EINVAL;
  if (rc == SG_OK) {
    const int64_t total = (int64_t)a.taps * s->cin * s->cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, hs, a.dwt, dw, coef, a.taps, s->cin,
                       s->cout, a.ciT, a.coT, det ? a.nslab : 1, a.slab, (bias_staged && db_done) ? bias_staged : nullptr, dbias,
                       accumulate, (clean_ws && !det) ? 1 : 0);
    if (det && dbias && db_done)
      hipLaunchKernelGGL(wgrad_bias_slabs_kernel, dim3((unsigned)sg_cdiv(s->cout, 256)), dim3(256), 0, hs, bias_slabs, dbias, s->cout,
                         a.nslab, a.bslab);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) rc = (int)e2;
  }
  if (rc == SG_OK && dbias && !db_done)
    rc = sg_bias_act_bwd(dy, nullptr, nullptr, dbias,
                         reinterpret_cast<char*>(workspace) + ns_max * wgrad_tile_bytes(s) + (clean_ws ? wgrad_bias_slab_bytes(s) : 0),
                         (int64_t)s->n * s->d * s->h * s->w, s->cout, 0.f, dt, st);
  prof.done(rc);
  return rc;
}

extern "C" int sg_conv3d_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, float coef, void* workspace,
                                    size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st) {
  return wgrad_bias_impl(x, dy, dw, dbias, coef, workspace, workspace_bytes, s, dt, st, nullptr, nullptr);
}

// dy_half: [n, d/2, h/2, w/2, cout], the gradient of the POOLED output of downscale3d(leaky_relu(conv3d(x) + b)); the weight
// and bias gradients are taken against dy_gain * where(bit, slope, 1) * nearest-x2(dy_half), formed while the tile is staged
extern "C" int sg_conv3d_wgrad_bias_up_masked(const void* x, const void* dy_half, const void* mask_bits, float mask_slope, float dy_gain,
                                              float* dw, float* dbias, float coef, void* workspace, size_t workspace_bytes,
                                              const sg_conv_shape* s, sg_dtype dt, sg_stream_t st) {
  if (!mask_bits) return SG_EINVAL;
  if (!sg_aligned16(mask_bits)) return SG_EALIGN;
  return wgrad_bias_impl(x, dy_half, dw, dbias, coef, workspace, workspace_bytes, s, dt, st, nullptr, nullptr, mask_bits, mask_slope, dy_gain);
}

// sg_conv3d_wgrad_bias / sg_conv3d_wgrad_bias_up_masked (mask_bits != NULL) with options:
//   SG_WGRAD_ACCUMULATE       dw += coef * sum (the parameter's gradient already holds another contribution: a second use of the
//                             weights in the graph, the gradient penalty's second-order term) -- rounded like the finished
//                             gradient added afterwards; dbias (optional) is WRITTEN.
//   SG_WGRAD_CLEAN_WORKSPACE  the workspace's first sg_conv3d_wgrad_clean_bytes() bytes are zero on entry (the caller keeps the
//                             buffer between calls) and are zero again when the call has drained: the finalize pass clears
//                             what it reads, and no memset is launched.
// SG_EUNSUPPORTED (nothing touched) on the pointwise / small-channel paths: their finalize kernels only write, and their
// partial sums do not leave the workspace clean.
extern "C" int sg_conv3d_wgrad_bias_ex(const void* x, const void* dy, const void* mask_bits, float mask_slope, float dy_gain,
                                       float* dw, float* dbias, float coef, unsigned flags, void* workspace, size_t workspace_bytes,
                                       const sg_conv_shape* s, sg_dtype dt, sg_stream_t st) {
  if (mask_bits && !sg_aligned16(mask_bits)) return SG_EALIGN;
  if (flags & ~(unsigned)(SG_WGRAD_ACCUMULATE | SG_WGRAD_CLEAN_WORKSPACE)) return SG_EINVAL;
  return wgrad_bias_impl(x, dy, dw, dbias, coef, workspace, workspace_bytes, s, dt, st, nullptr, nullptr, mask_bits, mask_slope,
                         mask_bits ? dy_gain : 1.f, flags);
}

extern "C" size_t sg_conv3d_wgrad_clean_bytes(const sg_conv_shape* s, sg_dtype dt) {
  (void)dt;
  if (!conv_shape_ok_w(s)) return 0;
  const size_t tile_bytes = (size_t)(s->kd * s->kh * s->kw) * sg_cdiv(s->cin, 32) * sg_cdiv(s->cout, 32) * 4096;
  return ((tile_bytes + 255) & ~(size_t)255) + (size_t)s->cout * sizeof(float);
}

extern "C" int sg_conv3d_pw_bwd(const void* x, const void* dy, const float* w_mat, float* dw, float* dbias, void* dx,
                                float coef, void* workspace, size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt,
                                sg_stream_t st) {
  if (!dx || !w_mat) return SG_EINVAL;
  if (!s || s->kd * s->kh * s->kw != 1 || s->cin > 4 || s->cin > s->cout || s->upsample_in) return SG_EUNSUPPORTED;
  return wgrad_bias_impl(x, dy, dw, dbias, coef, workspace, workspace_bytes, s, dt, st, dx, w_mat);
}
