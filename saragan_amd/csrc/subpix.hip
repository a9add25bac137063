// conv3d(upscale3d(x)) in sub-pixel form, ONE launch over all eight parity classes (bf16, gfx950).
// Replaces the pair networks/ops.py:276-289 (nearest x2) + :147-150 (3x3x3 conv) at pgan/generator.py:49-57.
//
// Nearest x2 followed by a 3-tap SAME convolution per dimension: output voxel 2i+a sees
//     a = 0:  x[i-1] * w0 + x[i] * (w1 + w2)         a = 1:  x[i] * (w0 + w1) + x[i+1] * w2
// so each of the 8 parity classes (a, b, c) of the fine grid is a 2x2x2-tap convolution of the LOW-resolution input with
// summed weights: 8 classes x 8 taps = 64 tap products per low-resolution voxel instead of 8 x 27 = 216 (3.4x fewer
// MFMAs).  All classes read the same 3x3x3 neighbourhood of the low-resolution voxel, so the kernel is laid out on the
// low-resolution grid: a block owns a tile of 256 low-resolution voxels (8 MFMA column tiles of 32 voxels, one per wave)
// and one 32-wide tile of output channels, stages the tile's halo ONCE per 16-channel chunk and uses it for all eight
// classes (eight accumulator tiles per wave: 128 registers); the summed weights (64 KiB per chunk) stream through LDS in
// two half-slabs (the four classes with a = 0, then a = 1), double buffered against the arithmetic.  LDS fill per CU:
// (64 KiB weights + 26 KiB halo) per 4096 MFMA cycles = 22 B/clk.  The epilogue (bias, LeakyReLU, pixel-norm, sign
// words) scatters class (a, b, c) to voxel (2d+a, 2h+b, 2w+c); a wave writes both W parities of its voxels, i.e. whole
// 128-byte lines of the 32-channel output.
//
// GEMM orientation as everywhere in this library: D[cout][voxel] += W[cout][cin] * X[cin][voxel] (A = weights).
#include "common.h"
#include "prof.h"
#include <type_traits>

struct SubpixArgs {
  const bf16_t* x;           // [N, d, h, w, cin]   low resolution
  const char* wp;            // [class 8][chunk][tap 8][ntile][lane 64][16 B]   summed weights, fragment order
  bf16_t* y;                 // [N, 2d, 2h, 2w, cout]
  const float* bias;
  float* pn_scale;           // [N * 8dhw] or null
  uint32_t* sign_out;        // [N * 8dhw][ntile] or null
  float slope, eps;
  int act, pixel_norm;
  int N, d, h, w, cin, cout, nchunk, ntile;
  int TD, TH, TW;            // low-resolution tile (TD * TH * TW = 256)
  int nTd, nTh, nTw;
  int HH, HW, hv;            // halo extents (TH + 2, TW + 2) and halo voxels (TD + 2) * HH * HW
  int64_t class_stride;      // bytes between the class images of wp
};

namespace {
constexpr int kWHalf = 4 * 8 * 1024;      // bytes of a half-slab: 4 classes x 8 taps x 1 KiB
constexpr int kXMax = 4 * 6 * 34 * 32;    // largest halo image of one 16-channel chunk (tile 2 x 4 x 32)

// the fragment products of one half-slab (classes with D parity A) for one wave: 18 halo fragments, 32 MFMAs
template <int A, int AO>      // A: D parity of the four classes; AO: index of their first accumulator tile
__device__ __forceinline__ void subpix_half(f32x16 (&acc)[8], const char* xs, const char* ws, int xbase, int planeB, int rowB,
                                            int lane) {
#pragma unroll
  for (int tz = 0; tz < 2; ++tz) {        // D tap of the class: neighbour plane A + tz
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const u32x4 xf = *reinterpret_cast<const u32x4*>(xs + xbase + (A + tz) * planeB + dy * rowB + dx * 32);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int th = dy - b;
          if (th < 0 || th > 1) continue;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int tw = dx - c;
            if (tw < 0 || tw > 1) continue;
            const int tap = (tz * 2 + th) * 2 + tw, cl = b * 2 + c;
            const u32x4 wf = *reinterpret_cast<const u32x4*>(ws + ((cl * 8 + tap) << 10) + lane * 16);
            acc[AO + cl] = sg_mfma_chunk<bf16_t>(wf, xf, acc[AO + cl]);
          }
        }
      }
    }
  }
}
}  // namespace

__global__ __launch_bounds__(512) void upconv_subpixel_fwd_kernel(SubpixArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wbuf = smem;                       // 2 half-slab buffers
  char* const xbuf = smem + 2 * kWHalf;          // 2 halo buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nt = blockIdx.y;
  // tile origin
  int t = blockIdx.x;
  const int tw_i = t % a.nTw; t /= a.nTw;
  const int th_i = t % a.nTh; t /= a.nTh;
  const int td_i = t % a.nTd;
  const int n = t / a.nTd;
  const int d0 = td_i * a.TD, h0 = th_i * a.TH, w0 = tw_i * a.TW;
  const int xbytes = a.hv * 32;
  const int rowB = a.HW * 32, planeB = a.HH * rowB;
  // this lane's voxel of the wave's column tile
  const int q = wave * 32 + r;
  const int vw = q % a.TW, vh = (q / a.TW) % a.TH, vd = q / (a.TW * a.TH);
  const int xbase = (vd * a.HH + vh) * rowB + vw * 32 + hh * 16;

  // staging: each thread moves up to 4 halo pieces and 4 weight pieces (16 B each) per step
  const bf16_t* xn = a.x + (int64_t)n * a.d * a.h * a.w * a.cin;
  int xoff[4];            // element offset of my halo pieces in x (chunk 0), -1: outside the volume (zero)
  int xdst[4];            // LDS byte offset of the piece, -1: none
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = tid + k * 512;          // piece = (halo voxel, half)
    xdst[k] = -1; xoff[k] = -1;
    if (p < a.hv * 2) {
      const int hvx = p >> 1, half = p & 1;
      const int hw_ = hvx % a.HW, hh_ = (hvx / a.HW) % a.HH, hd_ = hvx / (a.HW * a.HH);
      const int dd = d0 + hd_ - 1, yy = h0 + hh_ - 1, ww = w0 + hw_ - 1;
      xdst[k] = hvx * 32 + half * 16;
      if (dd >= 0 && dd < a.d && yy >= 0 && yy < a.h && ww >= 0 && ww < a.w)
        xoff[k] = (((dd * a.h + yy) * a.w + ww) * a.cin) + half * 8;
    }
  }
  auto load_x = [&](u32x4 (&st)[4], int chunk) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {   // unconditional load from a clamped address, then a select: a predicated load would sit in its own exec region
      // with an s_waitcnt vmcnt(0) behind it (seen in small.hip's first build) and serialise the staging
      const u32x4 v = *reinterpret_cast<const u32x4*>(xn + (xoff[k] >= 0 ? xoff[k] : 0) + chunk * 16);
      st[k] = xoff[k] >= 0 ? v : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto store_x = [&](const u32x4 (&st)[4], char* dst) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (xdst[k] >= 0) *reinterpret_cast<u32x4*>(dst + xdst[k]) = st[k];
  };
  // half-slab (chunk, A): classes A*4 .. A*4+3, taps 0..7: piece p = (cl, tap, lane16)
  auto load_w = [&](u32x4 (&st)[4], int chunk, int A, int nt_ = -1) {
    const int ntl = nt_ < 0 ? nt : nt_;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = tid + k * 512;            // 0 .. 2047
      const int cl = p >> 9, rem = p & 511;   // 512 pieces (8 taps x 64 lanes) per class
      const int tap = rem >> 6, ln = rem & 63;
      const char* src = a.wp + (int64_t)(A * 4 + cl) * a.class_stride + ((((int64_t)chunk * 8 + tap) * a.ntile + ntl) << 10) + ln * 16;
      st[k] = *reinterpret_cast<const u32x4*>(src);
    }
  };
  auto store_w = [&](const u32x4 (&st)[4], char* dst) {
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(dst + (tid + k * 512) * 16) = st[k];
  };

  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  u32x4 sx[4], sw[4];
  load_x(sx, 0);
  load_w(sw, 0, 0);
  store_x(sx, xbuf);
  store_w(sw, wbuf);
  __syncthreads();
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    char* const xcur = xbuf + (chunk & 1) * kXMax;
    char* const xnext = xbuf + ((chunk + 1) & 1) * kXMax;
    // half 0 (A = 0) from wbuf[0]; meanwhile fetch half 1 of this chunk
    load_w(sw, chunk, 1);
    subpix_half<0, 0>(acc, xcur, wbuf, xbase, planeB, rowB, lane);
    store_w(sw, wbuf + kWHalf);
    __syncthreads();
    // half 1 from wbuf[1]; meanwhile fetch the next chunk's halo and its half 0
    const bool more = chunk + 1 < a.nchunk;
    if (more) { load_x(sx, chunk + 1); load_w(sw, chunk + 1, 0); }
    subpix_half<1, 4>(acc, xcur, wbuf + kWHalf, xbase, planeB, rowB, lane);
    if (more) { store_x(sx, xnext); store_w(sw, wbuf); }
    __syncthreads();
  }
  (void)xbytes;

  // epilogue: class (a_, b, c) of my voxel -> fine voxel (2d+a_, 2h+b, 2w+c)
  const int D2 = 2 * a.d, H2 = 2 * a.h, W2 = 2 * a.w;
  const float inv_c = 1.f / (float)a.cout;
  float bv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bv[i] = a.bias ? a.bias[nt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh] : 0.f;
#pragma unroll
  for (int cls = 0; cls < 8; ++cls) {
    const int ca = cls >> 2, cb = (cls >> 1) & 1, cc = cls & 1;
    const int64_t ov = (((int64_t)n * D2 + 2 * (d0 + vd) + ca) * H2 + 2 * (h0 + vh) + cb) * W2 + 2 * (w0 + vw) + cc;
    f32x16 v = acc[cls];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float e = v[i] + bv[i];
      if (a.act) e = sg_lrelu(e, a.slope);
      v[i] = e;
      ss += e * e;
    }
    if (a.pixel_norm) {      // (cout == 32: the lane pair (r, r + 32) holds the voxel's channels)
      ss += __shfl_xor(ss, 32);
      const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] *= sc;
      if (a.pn_scale && hh == 0) a.pn_scale[ov] = sc;
    }
    if (a.sign_out) {
      const uint32_t sw_ = sg_sign_word(v, hh);
      if (hh == 0) a.sign_out[ov * a.ntile + nt] = sw_;
    }
    sg_store_tile_row_bf16(a.y + ov * a.cout + nt * 32, v, hh, true);
  }
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N - 1>)
template <int N, int I = 0>
struct sg_static_for {
  template <class F>
  static __device__ __forceinline__ void run(F&& f) {
    if constexpr (I < N) {
      f(std::integral_constant<int, I>{});
      sg_static_for<N, I + 1>::run(f);
    }
  }
};

// Persistent form of the kernel above (same tile, same arithmetic, same epilogue).  The one-tile-per-block version waits
// twice per chunk for weights it asked for one half-chunk (~0.7 us) earlier -- L2 latency under load is 1-2 us -- and pays
// its prologue per tile: 28 us per tile where its LDS fill needs 14 and its MFMAs 11.  Here a block walks items (tile, chunk);
// the summed weights come in four 16-KiB groups per chunk -- (D parity a, D tap tz): 4 classes x 4 taps, 16 MFMAs per wave --
// through two LDS buffers from a register ring that is loaded ONE CHUNK ahead; the next item's halo is requested one piece
// per group (vmcnt retires in order: see the data-gradient kernel) into the other halo buffer.
__global__ __launch_bounds__(512) void upconv_subpixel_fwd3_kernel(SubpixArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kWG = 16 * 1024;
  constexpr uint32_t DEAD = 0x80000000u;
  char* const wbuf = smem;                       // 2 weight-group buffers
  char* const xbuf = smem + 2 * kWG;             // 2 halo buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nt = blockIdx.y;
  const int rowB = a.HW * 32, planeB = a.HH * rowB;
  const int q = wave * 32 + r;
  const int vw = q % a.TW, vh = (q / a.TW) % a.TH, vd = q / (a.TW * a.TH);
  // halo rows are 32 bytes (16 channels); the 16-byte slot of a row is half ^ (w position >> 3 & 1): conflict-free b128 reads
  // for 32 consecutive positions (the un-swizzled image of the kernel above: SQ_LDS_BANK_CONFLICT = 22 % of its LDS cycles)
  const int xrow = (vd * a.HH + vh) * rowB + vw * 32;
  int xsw[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) xsw[dx] = dx * 32 + ((hh ^ (((vw + dx) >> 3) & 1)) << 4);
  int xdst[4];                // LDS byte offset of my halo pieces (the same for every tile)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = tid + k * 512;
    const int hvx = p >> 1, half = p & 1;
    xdst[k] = hvx * 32 + ((half ^ (((hvx % a.HW) >> 3) & 1)) << 4);
  }
  const int64_t xsb = (int64_t)a.d * a.h * a.w * a.cin * 2;      // bytes of one sample of x (< 2 GiB, host-checked)
  // tile schedule: XCD x owns a contiguous eighth of the tile list (neighbouring tiles share halo rows in one L2)
  const int64_t ntiles = (int64_t)a.N * a.nTd * a.nTh * a.nTw;
  const int64_t per_x = gridDim.x >> 3, cpx = (ntiles + 7) >> 3;
  const int64_t c_begin = (blockIdx.x & 7) * cpx, c_end = c_begin + cpx < ntiles ? c_begin + cpx : ntiles;
  const int64_t t_first = c_begin + (blockIdx.x >> 3);
  const int64_t ntl = t_first < c_end ? (c_end - t_first + per_x - 1) / per_x : 0;
  if (ntl == 0) return;
  const int nchunk = a.nchunk;

  int d0 = 0, h0 = 0, w0 = 0, n0 = 0;
  __amdgpu_buffer_rsrc_t rx;
  uint32_t xv[4];             // byte offsets of my four halo pieces inside the sample (DEAD: outside the volume / no piece)
  auto enter_tile = [&](int64_t t) __attribute__((always_inline)) {
    w0 = (int)(t % a.nTw) * a.TW; t /= a.nTw;
    h0 = (int)(t % a.nTh) * a.TH; t /= a.nTh;
    d0 = (int)(t % a.nTd) * a.TD;
    n0 = (int)(t / a.nTd);
    rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + n0 * xsb, 0, (int)xsb, 0x00020000);
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));      // (keeps the divisions below out of the item loop's live ranges)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = tid_ + k * 512;
      const int hvx = p >> 1, half = p & 1;
      const int hw_ = hvx % a.HW, hh_ = (hvx / a.HW) % a.HH, hd_ = hvx / (a.HW * a.HH);
      const int dd = d0 + hd_ - 1, yy = h0 + hh_ - 1, ww = w0 + hw_ - 1;
      const bool in = p < a.hv * 2 && dd >= 0 && dd < a.d && yy >= 0 && yy < a.h && ww >= 0 && ww < a.w;
      xv[k] = in ? (uint32_t)(((((dd * a.h + yy) * a.w + ww) * a.cin) + half * 8) * 2) : DEAD;
    }
  };
  u32x4 sx[4];
  auto store_x = [&](char* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (tid + k * 512 < a.hv * 2) *reinterpret_cast<u32x4*>(dst + xdst[k]) = sx[k];
  };
  u32x4 wr[4][2];
  auto load_w = [&](u32x4 (&dst)[2], int chunk, int g) __attribute__((always_inline)) {
    const char* src = a.wp + 8 * a.class_stride + (((((int64_t)nt * nchunk + chunk) * 4 + g)) << 14) + tid * 16;
    dst[0] = *reinterpret_cast<const u32x4*>(src);
    dst[1] = *reinterpret_cast<const u32x4*>(src + 8192);
  };
  auto store_w = [&](const u32x4 (&srcr)[2], char* dst) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(dst + tid * 16) = srcr[0];
    *reinterpret_cast<u32x4*>(dst + 8192 + tid * 16) = srcr[1];
  };

  f32x16 acc[8];
  // prologue: the first item's halo and weight group 0 into LDS, groups 1..3 into the ring
  enter_tile(t_first);
#pragma unroll
  for (int k = 0; k < 4; ++k) sx[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, xv[k], 0u, 0);
  load_w(wr[0], 0, 0);
  load_w(wr[1], 0, 1);
  load_w(wr[2], 0, 2);
  load_w(wr[3], 0, 3);
  store_x(xbuf);
  store_w(wr[0], wbuf);
  __syncthreads();

  const int D2 = 2 * a.d, H2 = 2 * a.h, W2 = 2 * a.w;
  const float inv_c = 1.f / (float)a.cout;
  int64_t ti = 0;
  int chunk = 0, xb = 0;
  for (;;) {
    int nchunk_i = chunk + 1;
    int64_t nti = ti;
    if (nchunk_i == nchunk) { nchunk_i = 0; ++nti; }
    const bool more = nti < ntl;
    const int od0 = d0, oh0 = h0, ow0 = w0, on0 = n0;
    if (more && nchunk_i == 0) enter_tile(t_first + nti * per_x);
    if (chunk == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    }
    const char* xs = xbuf + xb * kXMax;
    sg_static_for<4>::run([&](auto GG) __attribute__((always_inline)) {
      constexpr int g = decltype(GG)::value, A = g >> 1, tz = g & 1;
      load_w(wr[g], nchunk_i, g);                        // the same group of the next item: one chunk ahead
      // the next item's halo: four pieces over the first three groups (none in the last: its slack covers their latency)
      if (g < 3) sx[g] = __builtin_amdgcn_raw_buffer_load_b128(rx, more ? xv[g] : DEAD, (uint32_t)nchunk_i * 32u, 0);
      if (g == 0) sx[3] = __builtin_amdgcn_raw_buffer_load_b128(rx, more ? xv[3] : DEAD, (uint32_t)nchunk_i * 32u, 0);
      const char* ws = wbuf + (g & 1) * kWG;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const u32x4 xf = *reinterpret_cast<const u32x4*>(xs + xrow + (A + tz) * planeB + dy * rowB + xsw[dx]);
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int th = dy - b;
            if (th < 0 || th > 1) continue;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int tw = dx - c;
              if (tw < 0 || tw > 1) continue;
              const u32x4 wf = *reinterpret_cast<const u32x4*>(ws + ((((b * 2 + c) * 4 + th * 2 + tw)) << 10) + lane * 16);
              acc[A * 4 + b * 2 + c] = sg_mfma_chunk<bf16_t>(wf, xf, acc[A * 4 + b * 2 + c]);
            }
          }
        }
      store_w(wr[(g + 1) & 3], wbuf + ((g + 1) & 1) * kWG);      // (g == 3: group 0 of the next item, requested at step 0)
      if (g == 3) store_x(xbuf + (xb ^ 1) * kXMax);
      __syncthreads();
    });
    if (chunk + 1 == nchunk) {      // epilogue: class (a_, b, c) of my voxel -> fine voxel (2d+a_, 2h+b, 2w+c)
      float bv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) bv[i] = a.bias ? a.bias[nt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh] : 0.f;
#pragma unroll
      for (int cls = 0; cls < 8; ++cls) {
        const int ca = cls >> 2, cb = (cls >> 1) & 1, cc = cls & 1;
        const int64_t ov = (((int64_t)on0 * D2 + 2 * (od0 + vd) + ca) * H2 + 2 * (oh0 + vh) + cb) * W2 + 2 * (ow0 + vw) + cc;
        f32x16 v = acc[cls];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float e = v[i] + bv[i];
          if (a.act) e = sg_lrelu(e, a.slope);
          v[i] = e;
          ss += e * e;
        }
        if (a.pixel_norm) {      // (cout == 32: the lane pair (r, r + 32) holds the voxel's channels)
          ss += __shfl_xor(ss, 32);
          const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] *= sc;
          if (a.pn_scale && hh == 0) a.pn_scale[ov] = sc;
        }
        if (a.sign_out) {
          const uint32_t sw_ = sg_sign_word(v, hh);
          if (hh == 0) a.sign_out[ov * a.ntile + nt] = sw_;
        }
        sg_store_tile_row_bf16(a.y + ov * a.cout + nt * 32, v, hh, true);
      }
    }
    if (!more) break;
    ti = nti;
    chunk = nchunk_i;
    xb ^= 1;
  }
}

// The same for 64 output channels with pixel-norm in the epilogue: a voxel's channels span two N tiles, which must meet in
// one wave.  Eight classes x two tiles would be 256 accumulator registers, so the block runs the four classes of one D
// parity at a time (accumulator tile = nt * 4 + class), through all chunks, writes them, then the other parity: the halo
// is staged twice (it is the low-resolution tensor: cheap), the weights once.  Step = (parity, chunk, N tile).
__global__ __launch_bounds__(512) void upconv_subpixel_fwd2_kernel(SubpixArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wbuf = smem;
  char* const xbuf = smem + 2 * kWHalf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  int t = blockIdx.x;
  const int tw_i = t % a.nTw; t /= a.nTw;
  const int th_i = t % a.nTh; t /= a.nTh;
  const int td_i = t % a.nTd;
  const int n = t / a.nTd;
  const int d0 = td_i * a.TD, h0 = th_i * a.TH, w0 = tw_i * a.TW;
  const int rowB = a.HW * 32, planeB = a.HH * rowB;
  const int q = wave * 32 + r;
  const int vw = q % a.TW, vh = (q / a.TW) % a.TH, vd = q / (a.TW * a.TH);
  const int xbase = (vd * a.HH + vh) * rowB + vw * 32 + hh * 16;
  const bf16_t* xn = a.x + (int64_t)n * a.d * a.h * a.w * a.cin;
  int xoff[4], xdst[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = tid + k * 512;
    xdst[k] = -1; xoff[k] = -1;
    if (p < a.hv * 2) {
      const int hvx = p >> 1, half = p & 1;
      const int hw_ = hvx % a.HW, hh_ = (hvx / a.HW) % a.HH, hd_ = hvx / (a.HW * a.HH);
      const int dd = d0 + hd_ - 1, yy = h0 + hh_ - 1, ww = w0 + hw_ - 1;
      xdst[k] = hvx * 32 + half * 16;
      if (dd >= 0 && dd < a.d && yy >= 0 && yy < a.h && ww >= 0 && ww < a.w)
        xoff[k] = (((dd * a.h + yy) * a.w + ww) * a.cin) + half * 8;
    }
  }
  auto load_x = [&](u32x4 (&st)[4], int chunk) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(xn + (xoff[k] >= 0 ? xoff[k] : 0) + chunk * 16);
      st[k] = xoff[k] >= 0 ? v : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto store_x = [&](const u32x4 (&st)[4], char* dst) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (xdst[k] >= 0) *reinterpret_cast<u32x4*>(dst + xdst[k]) = st[k];
  };
  auto load_w = [&](u32x4 (&st)[4], int chunk, int A, int ntl) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = tid + k * 512;
      const int cl = p >> 9, rem = p & 511;
      const int tap = rem >> 6, ln = rem & 63;
      st[k] = *reinterpret_cast<const u32x4*>(a.wp + (int64_t)(A * 4 + cl) * a.class_stride +
                                              ((((int64_t)chunk * 8 + tap) * a.ntile + ntl) << 10) + ln * 16);
    }
  };
  auto store_w = [&](const u32x4 (&st)[4], char* dst) {
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(dst + (tid + k * 512) * 16) = st[k];
  };
  const int D2 = 2 * a.d, H2 = 2 * a.h, W2 = 2 * a.w;
  const float inv_c = 1.f / (float)a.cout;
  u32x4 sx[4], sw[4];
  // the pipeline runs over both parities back to back: step (A, chunk, nt); buffers alternate with the running step count
  load_x(sx, 0);
  load_w(sw, 0, 0, 0);
  store_x(sx, xbuf);
  store_w(sw, wbuf);
  __syncthreads();
  int xb = 0;      // halo buffer in use
  auto parity = [&](auto AA) {
    constexpr int A = decltype(AA)::value;
    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      char* const xcur = xbuf + xb * kXMax;
      char* const xnext = xbuf + (xb ^ 1) * kXMax;
      // N tile 0 from wbuf[0]; meanwhile fetch N tile 1 of this (parity, chunk)
      load_w(sw, chunk, A, 1);
      subpix_half<A, 0>(acc, xcur, wbuf, xbase, planeB, rowB, lane);
      store_w(sw, wbuf + kWHalf);
      __syncthreads();
      // N tile 1 from wbuf[1]; meanwhile fetch the next step's halo chunk and its N tile 0
      const bool last = chunk + 1 == a.nchunk;
      const bool more = !last || A == 0;
      const int nchunk_ = last ? 0 : chunk + 1, nA = last ? 1 : A;
      if (more) { load_x(sx, nchunk_); load_w(sw, nchunk_, nA, 0); }
      subpix_half<A, 4>(acc, xcur, wbuf + kWHalf, xbase, planeB, rowB, lane);
      if (more) { store_x(sx, xnext); store_w(sw, wbuf); }
      __syncthreads();
      xb ^= 1;
    }
    // epilogue of the four classes (A, b, c): both N tiles of a voxel are in this lane pair
    float bv[2][16];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int i = 0; i < 16; ++i) bv[t2][i] = a.bias ? a.bias[t2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh] : 0.f;
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
      const int cb = cl >> 1, cc = cl & 1;
      const int64_t ov = (((int64_t)n * D2 + 2 * (d0 + vd) + A) * H2 + 2 * (h0 + vh) + cb) * W2 + 2 * (w0 + vw) + cc;
      float ss = 0.f;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float e = acc[t2 * 4 + cl][i] + bv[t2][i];
          if (a.act) e = sg_lrelu(e, a.slope);
          acc[t2 * 4 + cl][i] = e;
          ss += e * e;
        }
      if (a.pixel_norm) {
        ss += __shfl_xor(ss, 32);
        const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t2 * 4 + cl][i] *= sc;
        if (a.pn_scale && hh == 0) a.pn_scale[ov] = sc;
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        if (a.sign_out) {
          const uint32_t sw_ = sg_sign_word(acc[t2 * 4 + cl], hh);
          if (hh == 0) a.sign_out[ov * 2 + t2] = sw_;
        }
        sg_store_tile_row_bf16(a.y + ov * 64 + t2 * 32, acc[t2 * 4 + cl], hh, true);
      }
    }
  };
  parity(std::integral_constant<int, 0>{});
  parity(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------------------
// summed weights of the eight classes, in fragment order
// ------------------------------------------------------------------------------------------------------
struct SubpixPackArgs {
  const float* w;    // [3][3][3][cin][cout] (forward) -- flip: [3][3][3][cout][cin] mirrored (data gradient of a plain conv)
  char* wp;
  float coef;
  int cin, cout, nchunk, ntile;
  int64_t class_stride;
};

__global__ void upconv_subpixel_pack_kernel(SubpixPackArgs a) {
  // element i of the packed image: (class, chunk, tap, ntile, lane, e)
  const int64_t per_class = (int64_t)a.nchunk * 8 * a.ntile * 64 * 8;
  const int64_t total = per_class * 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cls = (int)(i / per_class);
    int64_t q = i % per_class;
    const int e = (int)(q % 8); q /= 8;
    const int lane = (int)(q % 64); q /= 64;
    const int nt = (int)(q % a.ntile); q /= a.ntile;
    const int tap = (int)(q % 8);
    const int chunk = (int)(q / 8);
    const int ci = chunk * 16 + (lane >> 5) * 8 + e, co = nt * 32 + (lane & 31);
    float v = 0.f;
    if (ci < a.cin && co < a.cout) {
      const int par[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
      const int tp[3] = {tap >> 2, (tap >> 1) & 1, tap & 1};
      int lo[3], hi[3];      // original taps summed into (parity, tap): p=0: {0} / {1,2};  p=1: {0,1} / {2}
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (par[k] == 0) { lo[k] = tp[k] == 0 ? 0 : 1; hi[k] = tp[k] == 0 ? 0 : 2; }
        else { lo[k] = tp[k] == 0 ? 0 : 2; hi[k] = tp[k] == 0 ? 1 : 2; }
      }
      for (int kd = lo[0]; kd <= hi[0]; ++kd)
        for (int kh = lo[1]; kh <= hi[1]; ++kh)
          for (int kw = lo[2]; kw <= hi[2]; ++kw)
            v += a.w[((((int64_t)kd * 3 + kh) * 3 + kw) * a.cin + ci) * a.cout + co];
      v *= a.coef;
    }
    reinterpret_cast<bf16_t*>(a.wp + (int64_t)cls * a.class_stride)[i % per_class] = (bf16_t)v;
    // second copy in the order the persistent forward kernel streams it: [ntile][chunk][group = (a, tz)][class (b, c)][tap (th, tw)]
    // -- 16 fragments = 16 KiB per group, contiguous
    const int grp = (cls >> 2) * 2 + (tap >> 2), f = (cls & 3) * 4 + (tap & 3);
    const int64_t gi = ((((int64_t)nt * a.nchunk + chunk) * 4 + grp) * 16 + f) * 512 + lane * 8 + e;
    reinterpret_cast<bf16_t*>(a.wp + 8 * a.class_stride)[gi] = (bf16_t)v;
  }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static bool subpix_tile(const sg_conv_shape* s, int* td, int* th, int* tw) {
  // 256 low-resolution voxels: TW = min(w, 32), then H, then D; the volume must divide into whole tiles
  int w_ = s->w < 32 ? s->w : 32;
  if (w_ != 32 && w_ != 16 && w_ != 8) return false;
  int h_ = 256 / w_;
  if (h_ > s->h) h_ = s->h;
  if (h_ > 8) h_ = 8;
  if (w_ == 32 && h_ > 4) h_ = 4;
  int d_ = 256 / (w_ * h_);
  if (d_ > s->d || d_ < 1 || d_ * h_ * w_ != 256) return false;
  if (s->w % w_ || s->h % h_ || s->d % d_) return false;
  if ((d_ + 2) * (h_ + 2) * (w_ + 2) * 32 > kXMax) return false;
  *td = d_; *th = h_; *tw = w_;
  return true;
}

// s: the LOW-resolution shape (n, d, h, w, cin, cout); kd = kh = kw = 3 of the original convolution
extern "C" size_t sg_upconv3d_subpixel_packed_bytes(const sg_conv_shape* s, sg_dtype dt) {
  if (!s || dt != SG_BF16 || s->cin < 1 || s->cout < 1) return 0;
  return (size_t)2 * 8 * sg_cdiv(s->cin, 16) * 8 * sg_cdiv(s->cout, 32) * 1024;      // class-major image + the same in group order
}

extern "C" int sg_upconv3d_subpixel_supported(const sg_conv_shape* s, sg_dtype dt) {
  int td, th, tw;
  return (s && dt == SG_BF16 && s->cin > 0 && s->cout > 0 && s->cin % 16 == 0 && s->cout % 32 == 0 && subpix_tile(s, &td, &th, &tw) &&
          (int64_t)s->d * s->h * s->w * s->cin < (1ll << 31)) ? 1 : 0;
}

extern "C" int sg_upconv3d_subpixel_pack(const float* w_dhwio, float coef, void* wp, const sg_conv_shape* s, sg_dtype dt,
                                         sg_stream_t st) {
  if (!s || !w_dhwio || !wp || dt != SG_BF16) return SG_EINVAL;
  SubpixPackArgs a;
  a.w = w_dhwio; a.wp = reinterpret_cast<char*>(wp); a.coef = coef; a.cin = s->cin; a.cout = s->cout;
  a.nchunk = sg_cdiv(s->cin, 16); a.ntile = sg_cdiv(s->cout, 32);
  a.class_stride = (int64_t)a.nchunk * 8 * a.ntile * 1024;
  const int64_t total = a.class_stride * 8 / 2;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(upconv_subpixel_pack_kernel, dim3(blocks), dim3(256), 0, sg_st(st), a);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_upconv3d_subpixel_fwd(const void* x, const void* wp, void* y, const sg_conv_shape* s, const sg_conv_epilogue* ep,
                                        sg_dtype dt, sg_stream_t st) {
  if (!s || !x || !wp || !y) return SG_EINVAL;
  if (ep && ep->struct_size != (uint32_t)sizeof(sg_conv_epilogue)) return SG_EINVAL;
  if (!sg_aligned16(x) || !sg_aligned16(wp) || !sg_aligned16(y)) return SG_EALIGN;
  if (dt != SG_BF16 || s->cin % 16 || s->cout % 32) return SG_EUNSUPPORTED;
  if (ep && (ep->mask_bits || ep->pool || ep->pn_bwd_y || ep->x_plane_channels || ep->out_scale)) return SG_EUNSUPPORTED;
  if (ep && ep->pixel_norm && s->cout != 32 && s->cout != 64) return SG_EUNSUPPORTED;      // a lane pair must hold a voxel's channels
  const bool two_tiles = ep && ep->pixel_norm && s->cout == 64;
  SubpixArgs a;
  if (!subpix_tile(s, &a.TD, &a.TH, &a.TW)) return SG_EUNSUPPORTED;
  if ((int64_t)s->d * s->h * s->w * s->cin >= (1ll << 31)) return SG_EUNSUPPORTED;     // 32-bit element offsets per sample
  a.x = reinterpret_cast<const bf16_t*>(x); a.wp = reinterpret_cast<const char*>(wp); a.y = reinterpret_cast<bf16_t*>(y);
  a.bias = ep ? ep->bias : nullptr;
  a.pn_scale = ep ? ep->pn_scale : nullptr;
  a.sign_out = ep ? reinterpret_cast<uint32_t*>(ep->sign_out) : nullptr;
  a.slope = ep ? ep->slope : 0.f; a.eps = ep ? ep->eps : 0.f;
  a.act = ep ? ep->act : 0; a.pixel_norm = ep ? ep->pixel_norm : 0;
  a.N = s->n; a.d = s->d; a.h = s->h; a.w = s->w; a.cin = s->cin; a.cout = s->cout;
  a.nchunk = s->cin / 16; a.ntile = s->cout / 32;
  a.nTd = s->d / a.TD; a.nTh = s->h / a.TH; a.nTw = s->w / a.TW;
  a.HH = a.TH + 2; a.HW = a.TW + 2; a.hv = (a.TD + 2) * a.HH * a.HW;
  a.class_stride = (int64_t)a.nchunk * 8 * a.ntile * 1024;
  const int64_t tiles = (int64_t)s->n * a.nTd * a.nTh * a.nTw;
  if (tiles >= (1ll << 31)) return SG_EUNSUPPORTED;
  sg_conv_shape full = *s;       // (profiler key: the fine grid, as the fused-gather kernels report it)
  full.d *= 2; full.h *= 2; full.w *= 2; full.kd = full.kh = full.kw = 3; full.upsample_in = 1;
  sg_prof_scope prof(0, &full, dt, sg_st(st));
  const size_t lds = 2 * (size_t)kWHalf + 2 * (size_t)kXMax;
  if (two_tiles) {
    auto kern = upconv_subpixel_fwd2_kernel;
    SG_ALLOW_160K_LDS(kern);
    SG_KNAME("upconv_subpixel_fwd<2 N tiles>");
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, sg_st(st), a);
  } else if ((sg_cfg().dbg_flags & 64) || (int64_t)s->d * s->h * s->w * s->cin * 2 >= (1ll << 31)) {
    // (diagnostic switch, or a sample of x beyond a buffer resource: the one-tile-per-block version)
    auto kern = upconv_subpixel_fwd_kernel;
    SG_ALLOW_160K_LDS(kern);
    SG_KNAME("upconv_subpixel_fwd<one tile per block>");
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)a.ntile), dim3(512), lds, sg_st(st), a);
  } else {
    auto kern = upconv_subpixel_fwd3_kernel;
    SG_ALLOW_160K_LDS(kern);
    SG_KNAME("upconv_subpixel_fwd");
    int64_t gx = 256 / a.ntile > 8 ? (256 / a.ntile) / 8 * 8 : 8;      // whole rounds of the 8 XCDs, at most one block per tile
    while (gx > 8 && gx > tiles) gx -= 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)a.ntile), dim3(512), 2 * (size_t)(16 * 1024) + 2 * (size_t)kXMax, sg_st(st), a);
  }
  hipError_t e = hipGetLastError();
  prof.done((int)e);
  return (int)e;
}

// ------------------------------------------------------------------------------------------------------
// data gradient of conv3d(upscale3d(x)) in sub-pixel form
// ------------------------------------------------------------------------------------------------------
// gx[u] = sum over fine voxels f = 2u + j, j in {-1, 0, 1, 2}^3, of W4[j]^T gy[f]: a stride-2 convolution of the fine
// gradient with 4 x 4 x 4 taps whose weights are the forward's summed sets transposed (per dimension j = -1: {w2},
// 0: {w1 + w2}, 1: {w0 + w1}, 2: {w0}) -- 64 tap products per low-resolution voxel where the pooled 27-tap gradient
// (sg_conv_epilogue.pool + sg_downscale_sum) spends 216, and the result is rounded once.
// A block walks tiles of 256 low-resolution voxels (wave = one 32-voxel column tile, all NT 32-channel tiles of gx) and
// the 16-channel chunks of gy: item = (tile, chunk).  The item's fine halo -- (2TD+2) x (2TH+2) x (2TW+2) voxels, 32 bytes
// each, ~124 KiB: single-buffered -- is written to LDS with the W parities de-interleaved (a tap then reads consecutive
// rows for consecutive lanes; 16-byte slot = half ^ (row >> 3 & 1): conflict-free) from registers that were loaded
// during the previous item; the weights (64 taps x NT KiB per chunk) stream through two 16-KiB buffers, loaded FOUR groups
// ahead into registers (the L2 latency under load is 3-4 groups of MFMAs).  LDS fill per CU: (124 + 64 NT) KiB per
// 4096 NT MFMA cycles = 31 / 23 B/clk at NT = 2 / 4: the kernel is bound by that, not by the MFMA.
struct SubpixDgradArgs {
  const bf16_t* gy;          // [N, 2d, 2h, 2w, co]
  const char* wp;            // [ci / 64][chunk co/16][tap 64][nt 2][lane 64][16 B]
  bf16_t* gx;                // [N, d, h, w, ci]
  int N, d, h, w, ci, co, nchunk;
  int nTd, nTh, nTw;
  int64_t ntiles;
};

namespace {
constexpr int kDgW = 16 * 1024;           // one weight group: 16 fragments
constexpr int kDgY = 6 * 10 * 2 * 33 * 32; // largest halo image (tile 2 x 4 x 32): 126720 bytes
}  // namespace

template <int NT, int TD, int TH, int TW>
__global__ __launch_bounds__(512) void upconv_subpixel_dgrad_kernel(SubpixDgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(TD * TH * TW == 256, "256 low-resolution voxels per tile");
  constexpr int NG = 64 * NT / 16;        // weight groups per chunk
  constexpr int TPG = 16 / NT;            // taps per group
  constexpr int FD = 2 * TD + 2, FH = 2 * TH + 2, PW = TW + 1;
  constexpr int NPIECES = FD * FH * 2 * PW * 2, NK = (NPIECES + 511) / 512;
  static_assert(NK <= 16 && NPIECES * 16 <= kDgY, "halo image");
  constexpr uint32_t DEAD = 0x80000000u;
  char* const wbuf = smem;                // 2 x kDgW
  char* const ybuf = smem + 2 * kDgW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int cpart = blockIdx.y;           // which NT * 32 channels of gx
  const int q = wave * 32 + r;
  const int vw = q % TW, vh = (q / TW) % TH, vd = q / (TW * TH);
  // fragment address of tap (td, th, tw) of my voxel: row = ((2vd + td) * FH + 2vh + th) * 2PW + (tw & 1) * PW + vw + (tw >> 1),
  // 16-byte slot = half ^ (pos >> 3 & 1), pos = vw + (tw >> 1): four per-lane bases + compile-time offsets
  int xb[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      xb[i][j] = (((2 * vd + 2 * i) * FH + 2 * vh) * 2 * PW + vw + j) * 32 + ((hh ^ (((vw + j) >> 3) & 1)) << 4);
  const int D2 = 2 * a.d, H2 = 2 * a.h, W2 = 2 * a.w;
  const int64_t ysb = (int64_t)D2 * H2 * W2 * a.co * 2;           // bytes of one sample of gy (< 2 GiB, host-checked)

  __amdgpu_buffer_rsrc_t ry;
  uint32_t yv[NK];
  int d0 = 0, h0 = 0, w0 = 0, n0 = 0;
  auto enter_tile = [&](int64_t t) __attribute__((always_inline)) {
    w0 = (int)(t % a.nTw) * TW; t /= a.nTw;
    h0 = (int)(t % a.nTh) * TH; t /= a.nTh;
    d0 = (int)(t % a.nTd) * TD;
    n0 = (int)(t / a.nTd);
  };
  // halo staging plan of the tile (d0, h0, w0, n0): piece p = (row, half), row = ((fd * FH + fh) * 2 + parity) * PW + pos,
  // fine w = 2 pos + parity; recomputed per tile (divisions by constants) instead of kept in 32 registers
  auto plan_tile = [&]() __attribute__((always_inline)) {
    ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.gy)) + n0 * ysb, 0, (int)ysb, 0x00020000);
    const int gd0 = 2 * d0 - 1, gh0 = 2 * h0 - 1, gw0 = 2 * w0 - 1;      // fine coordinates of halo voxel (0, 0, 0)
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));      // (keeps the decomposition below inside the call: hoisted out of the item loop it holds 48 registers)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int p = tid_ + k * 512;
      const int row = p >> 1, half = p & 1;
      const int pos = row % PW, t1 = row / PW;
      const int par = t1 & 1, line = t1 >> 1;
      const int fh = line % FH, fd = line / FH;
      const int gd = gd0 + fd, gh = gh0 + fh, gw = gw0 + 2 * pos + par;
      const bool in = p < NPIECES && (unsigned)gd < (unsigned)D2 && (unsigned)gh < (unsigned)H2 && (unsigned)gw < (unsigned)W2;
      yv[k] = in ? (uint32_t)((((gd * H2 + gh) * W2 + gw) * a.co + half * 8) * 2) : DEAD;
    }
  };
  u32x4 gr[NK];
  auto load_y = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NK; ++k) gr[k] = __builtin_amdgcn_raw_buffer_load_b128(ry, yv[k], (uint32_t)chunk * 32u, 0);
  };
  // The next item's halo is requested a few pieces per weight group, not all at the item's start: vmcnt retires in order,
  // so a weight group requested AFTER the halo can only be waited for once the whole halo has landed -- 124 KiB that the
  // CU takes in ~5 us where the weights are needed 3 groups (~2 us) later (seen in the first version: s_waitcnt at the end
  // of group 0 held every wave until the halo was in).  The last two groups request nothing: their slack covers the latency
  // of the final pieces before store_y.
  constexpr int YPG = (NK + (NG > 2 ? NG - 2 : 1) - 1) / (NG > 2 ? NG - 2 : 1);      // halo pieces per group
  auto load_y_part = [&](int chunk, bool ok, auto GG) __attribute__((always_inline)) {
    constexpr int g = decltype(GG)::value;
#pragma unroll
    for (int k = g * YPG; k < (g + 1) * YPG && k < NK; ++k)
      gr[k] = __builtin_amdgcn_raw_buffer_load_b128(ry, ok ? yv[k] : DEAD, (uint32_t)chunk * 32u, 0);
  };
  auto store_y = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int p = tid + k * 512;
      const int row = p >> 1, half = p & 1;
      const int pos = row % PW;
      if (p < NPIECES) *reinterpret_cast<u32x4*>(ybuf + row * 32 + ((half ^ ((pos >> 3) & 1)) << 4)) = gr[k];
    }
  };
  // weight groups: 1024 pieces of 16 B, two per thread
  u32x4 wr[4][2];
  auto load_w = [&](u32x4 (&dst)[2], int chunk, int g) __attribute__((always_inline)) {
    const char* src = a.wp + ((((int64_t)cpart * a.nchunk + chunk) * NG + g) << 14) + tid * 16;
    dst[0] = *reinterpret_cast<const u32x4*>(src);
    dst[1] = *reinterpret_cast<const u32x4*>(src + 8192);
  };
  auto store_w = [&](const u32x4 (&srcr)[2], char* dst) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(dst + tid * 16) = srcr[0];
    *reinterpret_cast<u32x4*>(dst + 8192 + tid * 16) = srcr[1];
  };

  // Tile schedule: XCD x (= blockIdx.x % 8: consecutive block ids go round the 8 XCDs) owns a CONTIGUOUS eighth of the tile
  // list and its blocks walk it side by side, so the halo rows neighbouring tiles share are fetched into one L2 within a
  // few microseconds.  (Dealt round-robin over all blocks, neighbours sat on different XCDs and every shared row came from
  // HBM again: FETCH_SIZE 2.6-4.7x the algorithmic bytes, profiles/r03_pmc_traffic.txt.)  gridDim.x is a multiple of 8.
  const int64_t per_x = gridDim.x >> 3;                        // blocks per XCD = the stride inside its chunk
  const int64_t cpx = (a.ntiles + 7) >> 3;
  const int64_t c_begin = (blockIdx.x & 7) * cpx, c_end = c_begin + cpx < a.ntiles ? c_begin + cpx : a.ntiles;
  const int64_t t_first = c_begin + (blockIdx.x >> 3);
  const int64_t ntl = t_first < c_end ? (c_end - t_first + per_x - 1) / per_x : 0;   // my tiles
  if (ntl == 0) return;
  const int nchunk = a.nchunk;
  f32x16 acc[NT];

  // prologue: the first item's halo and weight group 0 into LDS, groups 1..3 into the register ring
  enter_tile(t_first);
  plan_tile();
  load_y(0);
  load_w(wr[0], 0, 0);
  load_w(wr[1], 0, 1 % NG);
  load_w(wr[2], 0, 2 % NG);
  load_w(wr[3], 0, 3 % NG);
  store_y();
  store_w(wr[0], wbuf);
  __syncthreads();

  int64_t ti = 0;          // index of the current tile in my list
  int chunk = 0;
  for (;;) {
    // the item after this one: its halo is requested now and written at the end of this item
    int nchunk_i = chunk + 1;
    int64_t nti = ti;
    if (nchunk_i == nchunk) { nchunk_i = 0; ++nti; }
    const bool more = nti < ntl;
    const int od0 = d0, oh0 = h0, ow0 = w0, on0 = n0;     // this item's tile origin (the epilogue needs it after the plan moved on)
    if (more && nchunk_i == 0) { enter_tile(t_first + nti * per_x); plan_tile(); }
    if (chunk == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    }
    sg_static_for<NG>::run([&](auto GG) __attribute__((always_inline)) {
      constexpr int g = decltype(GG)::value;
      // group g + 4 into the ring slot group g came from (groups beyond this chunk belong to the next item's chunk)
      {
        const int g4 = g + 4;
        const int c4 = g4 < NG ? chunk : nchunk_i;
        load_w(wr[g & 3], c4, g4 % NG);
      }
      load_y_part(nchunk_i, more, GG);      // (unconditional: dead offsets when nothing follows)
      const char* ws = wbuf + (g & 1) * kDgW;
      // fragments one tap ahead of the MFMAs that use them (left to the scheduler, the unrolled group hoists all 24 reads:
      // 96 registers on top of the 64 of the halo in flight)
      constexpr int FD_ = 3;              // fragment ring: reads run FD_ - 1 taps ahead of the MFMAs (one tap = 2 MFMAs = 64 cycles: one tap
                                          // ahead did not cover the LDS latency under load, the group took 2.6 x its MFMA time)
      u32x4 xf[FD_], wf[FD_][NT];
      auto read_tap = [&](int slot, int tt) __attribute__((always_inline)) {
        const int tap = g * TPG + tt;
        const int td = tap >> 4, th = (tap >> 2) & 3, tw = tap & 3;
        const int off = (((td & 1) * FH + th) * 2 * PW + (tw & 1) * PW) * 32;
        xf[slot] = *reinterpret_cast<const u32x4*>(ybuf + xb[td >> 1][tw >> 1] + off);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[slot][nt] = *reinterpret_cast<const u32x4*>(ws + ((tt * NT + nt) << 10) + lane * 16);
      };
#pragma unroll
      for (int tt = 0; tt < FD_ - 1 && tt < TPG; ++tt) read_tap(tt, tt);
#pragma unroll
      for (int tt = 0; tt < TPG; ++tt) {
        if (tt + FD_ - 1 < TPG) read_tap((tt + FD_ - 1) % FD_, tt + FD_ - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = sg_mfma_chunk<bf16_t>(wf[tt % FD_][nt], xf[tt % FD_], acc[nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
      store_w(wr[(g + 1) & 3], wbuf + ((g + 1) & 1) * kDgW);
      __syncthreads();
    });
    if (chunk + 1 == nchunk) {      // the tile is complete: one rounding, 64 contiguous bytes per lane pair and channel tile
      const int64_t ov = (((int64_t)on0 * a.d + od0 + vd) * a.h + oh0 + vh) * a.w + ow0 + vw;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sg_store_tile_row_bf16(a.gx + ov * a.ci + (cpart * NT + nt) * 32, acc[nt], hh, true);
    }
    store_y();      // (unconditional, like its loads: after the last item it writes zeros nobody reads)
    __syncthreads();
    if (!more) break;
    ti = nti;
    chunk = nchunk_i;
  }
}

struct SubpixDgradPackArgs {
  const float* w;    // [3][3][3][ci][co]: the forward weight (DHWIO)
  char* wp;
  float coef;
  int ci, co, nchunk, ntile;
};

__global__ void upconv_subpixel_dgrad_pack_kernel(SubpixDgradPackArgs a) {
  // element i of the packed image: (ci / 64, chunk, tap, nt & 1, lane, e); A fragment: row = gx channel, K = gy channel
  const int64_t total = (int64_t)a.nchunk * 64 * a.ntile * 64 * 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t qq = i;
    const int e = (int)(qq % 8); qq /= 8;
    const int lane = (int)(qq % 64); qq /= 64;
    const int ntl = (int)(qq % 2); qq /= 2;
    const int tap = (int)(qq % 64); qq /= 64;
    const int chunk = (int)(qq % a.nchunk);
    const int nt = (int)(qq / a.nchunk) * 2 + ntl;
    const int ci = nt * 32 + (lane & 31), co = chunk * 16 + (lane >> 5) * 8 + e;
    float v = 0.f;
    if (ci < a.ci && co < a.co) {
      const int tj[3] = {tap >> 4, (tap >> 2) & 3, tap & 3};      // j + 1 per dimension
      int lo[3], hi[3];      // original taps summed into j: j = -1: {2}; 0: {1, 2}; 1: {0, 1}; 2: {0}
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lo[k] = tj[k] == 0 ? 2 : (tj[k] == 1 ? 1 : 0);
        hi[k] = tj[k] <= 1 ? 2 : (tj[k] == 2 ? 1 : 0);
      }
      for (int kd = lo[0]; kd <= hi[0]; ++kd)
        for (int kh = lo[1]; kh <= hi[1]; ++kh)
          for (int kw = lo[2]; kw <= hi[2]; ++kw)
            v += a.w[((((int64_t)kd * 3 + kh) * 3 + kw) * a.ci + ci) * a.co + co];
      v *= a.coef;
    }
    reinterpret_cast<bf16_t*>(a.wp)[i] = (bf16_t)v;
  }
}

// s: the LOW-resolution shape (n, d, h, w, cin = channels of x / gx, cout = channels of gy); kd = kh = kw = 3 of the forward
extern "C" int sg_upconv3d_subpixel_dgrad_supported(const sg_conv_shape* s, sg_dtype dt) {
  int td, th, tw;
  if (!s || dt != SG_BF16 || s->cin < 32 || s->cout < 16 || s->cin % 32 || s->cout % 16 || !subpix_tile(s, &td, &th, &tw)) return 0;
  const int nt = s->cin / 32;
  if (nt % 2 || !((tw == 32 && td == 2 && th == 4) || (tw == 16 && td == 2 && th == 8))) return 0;   // whole 64-channel parts; instantiated tiles
  if ((int64_t)s->d * s->h * s->w * 8 * s->cout * 2 >= (1ll << 31)) return 0;      // one sample of gy behind a buffer resource
  return 1;
}

extern "C" size_t sg_upconv3d_subpixel_dgrad_packed_bytes(const sg_conv_shape* s, sg_dtype dt) {
  if (!s || dt != SG_BF16 || s->cin < 1 || s->cout < 1) return 0;
  return (size_t)sg_cdiv(s->cout, 16) * 64 * sg_cdiv(s->cin, 32) * 1024;
}

extern "C" int sg_upconv3d_subpixel_dgrad_pack(const float* w_dhwio, float coef, void* wp, const sg_conv_shape* s, sg_dtype dt,
                                               sg_stream_t st) {
  if (!s || !w_dhwio || !wp || dt != SG_BF16) return SG_EINVAL;
  SubpixDgradPackArgs a;
  a.w = w_dhwio; a.wp = reinterpret_cast<char*>(wp); a.coef = coef; a.ci = s->cin; a.co = s->cout;
  a.nchunk = sg_cdiv(s->cout, 16); a.ntile = sg_cdiv(s->cin, 32);
  const int64_t total = (int64_t)a.nchunk * 64 * a.ntile * 512;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(upconv_subpixel_dgrad_pack_kernel, dim3(blocks), dim3(256), 0, sg_st(st), a);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_upconv3d_subpixel_dgrad(const void* gy, const void* wp, void* gx, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st) {
  if (!s || !gy || !wp || !gx) return SG_EINVAL;
  if (!sg_aligned16(gy) || !sg_aligned16(wp) || !sg_aligned16(gx)) return SG_EALIGN;
  if (!sg_upconv3d_subpixel_dgrad_supported(s, dt)) return SG_EUNSUPPORTED;
  SubpixDgradArgs a;
  int TD, TH, TW;
  subpix_tile(s, &TD, &TH, &TW);
  a.gy = reinterpret_cast<const bf16_t*>(gy); a.wp = reinterpret_cast<const char*>(wp); a.gx = reinterpret_cast<bf16_t*>(gx);
  a.N = s->n; a.d = s->d; a.h = s->h; a.w = s->w; a.ci = s->cin; a.co = s->cout; a.nchunk = s->cout / 16;
  a.nTd = s->d / TD; a.nTh = s->h / TH; a.nTw = s->w / TW;
  a.ntiles = (int64_t)s->n * a.nTd * a.nTh * a.nTw;
  sg_conv_shape full = *s;       // (profiler key: the data-gradient convolution on the fine grid, as the pooled path reports it)
  full.d *= 2; full.h *= 2; full.w *= 2; full.kd = full.kh = full.kw = 3; full.upsample_in = 0;
  full.cin = s->cout; full.cout = s->cin;
  sg_prof_scope prof(0, &full, dt, sg_st(st));
  int64_t per_part = 256 / (s->cin / 64) > 8 ? 256 / (s->cin / 64) : 8;
  while (per_part > 8 && per_part > a.ntiles) per_part -= 8;      // whole rounds of the 8 XCDs, at most one block per tile
  const unsigned gx_ = (unsigned)per_part;
  const size_t lds = 2 * (size_t)kDgW + (size_t)kDgY;
  const int nt = s->cin / 32;
  SG_KNAME("upconv_subpixel_dgrad");
#define SG_DG(NT_, TD_, TH_, TW_)                                                    \
  do {                                                                               \
    auto kern = upconv_subpixel_dgrad_kernel<NT_, TD_, TH_, TW_>;                    \
    SG_ALLOW_160K_LDS(kern);                                                         \
    hipLaunchKernelGGL(kern, dim3(gx_, (unsigned)(nt / 2)), dim3(512), lds, sg_st(st), a); \
  } while (0)
  if (TW == 32) SG_DG(2, 2, 4, 32);
  else if (TW == 16) SG_DG(2, 2, 8, 16);
  else { prof.done(SG_EUNSUPPORTED); return SG_EUNSUPPORTED; }
#undef SG_DG
  hipError_t e = hipGetLastError();
  prof.done((int)e);
  return (int)e;
}

// ------------------------------------------------------------------------------------------------------
// weight gradient of conv3d(upscale3d(x)) in sub-pixel form
// ------------------------------------------------------------------------------------------------------
// dWeff[class][tap][ci][co] = sum over low-resolution voxels j of x[j + n(class, tap)][ci] * gy[2j + class][co]: 64
// accumulator tiles of 32 x 32 per (ci tile, co tile) instead of 8 x 27 tap products per low-resolution voxel; the 27-tap
// gradient is their fold (every original tap belongs to 2 x 2 x 2 (class, tap) sets), done by the finalize kernel.
// Block = 8 waves, wave w owns parity class w and its 8 taps (+ a ones row for the bias gradient): per 16-voxel K step it
// reads its class plane of gy once (transposing read) and 8 neighbour fragments of x -- 18 reads for 8 (9) MFMAs.  A tile is
// 1 x 2 x 32 low-resolution voxels: the x halo (3 x 4 x 34 voxels x 64 B) and the 2 x 4 x 64 fine voxels of gy, the latter
// de-interleaved into the 8 class planes while it is written to LDS (so that a class's rows are contiguous: conflict-free
// transposing reads).  Both are double buffered and requested one tile ahead.
struct SubpixWgradArgs {
  const bf16_t* x;           // [N, d, h, w, cin] low resolution
  const bf16_t* gy;          // [N, 2d, 2h, 2w, cout]
  float* dwt;                // [slab][class 8][tap 8][ciT][coT][32][32] f32
  float* dbias;              // [slab][coT * 32] or null
  int N, d, h, w, cin, cout, ciT, coT;
  int nTh, nTw;              // tiles per plane: h / 2, w / 32
  int64_t ntiles;            // N * d * nTh * nTw
  int64_t slab, bslab;       // elements between per-block slabs (0: atomics into one buffer)
};

namespace {
constexpr int kWX = 3 * 4 * 34 * 64;      // 26112: x halo image of a tile, 32 channels
constexpr int kWY = 8 * 2 * 32 * 64;      // 32768: the tile's gy as 8 class planes of 2 x 32 voxels, 32 channels
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4;

__device__ __forceinline__ bf16x8 sg_tr_frag(const char* p) {      // two transposing reads: voxel rows q and q + 4 of the block
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 256));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
}  // namespace

__global__ __launch_bounds__(512) void upconv_subpixel_wgrad_kernel(SubpixWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int cls = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave = parity class
  const int ca = cls >> 2, cb = (cls >> 1) & 1, cc = cls & 1;
  const int ci_t = blockIdx.y / a.coT, co_t = blockIdx.y % a.coT;
  const bool ones = a.dbias != nullptr && ci_t == 0;
  // transposing-read lane map (wgrad.hip): group of 16 lanes = 4 voxel rows x 16 channels
  const int i16 = lane & 15, q16 = lane >> 4;
  const int colb = (16 * (q16 & 1) + 4 * (i16 & 3)) * 2;
  const int kb = 8 * (q16 >> 1) + (i16 >> 2);
  // staging plan: pieces of 16 B; x halo 1632 pieces, gy 2048 pieces
  const int D2 = 2 * a.d, H2 = 2 * a.h, W2 = 2 * a.w;
  auto tile_of = [&](int64_t t, int& n, int& dd, int& h0, int& w0) {
    w0 = (int)(t % a.nTw) * 32; t /= a.nTw;
    h0 = (int)(t % a.nTh) * 2; t /= a.nTh;
    dd = (int)(t % a.d);
    n = (int)(t / a.d);
  };
  u32x4 sx[4], sy[4];
  auto load_tile = [&](int64_t t) {
    int n, dd, h0, w0;
    tile_of(t, n, dd, h0, w0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = tid + k * 512;
      const int hv = p >> 2, sub = p & 3;
      const int hw_ = hv % 34, hh_ = (hv / 34) & 3, hd_ = hv / 136;
      const int zd = dd + hd_ - 1, zy = h0 + hh_ - 1, zx = w0 + hw_ - 1;
      const bool ok = p < 1632 && zd >= 0 && zd < a.d && zy >= 0 && zy < a.h && zx >= 0 && zx < a.w;
      const int64_t off = ((((int64_t)n * a.d + (ok ? zd : 0)) * a.h + (ok ? zy : 0)) * a.w + (ok ? zx : 0)) * a.cin + ci_t * 32 + sub * 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.x + off);
      sx[k] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = tid + k * 512;
      const int fv = p >> 2, sub = p & 3;
      const int fw = fv & 63, fh = (fv >> 6) & 3, fd = fv >> 8;
      const int64_t off = ((((int64_t)n * D2 + 2 * dd + fd) * H2 + 2 * h0 + fh) * W2 + 2 * w0 + fw) * a.cout + co_t * 32 + sub * 8;
      sy[k] = *reinterpret_cast<const u32x4*>(a.gy + off);
    }
  };
  auto store_tile = [&](char* xb, char* yb) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = tid + k * 512;
      if (p < 1632) *reinterpret_cast<u32x4*>(xb + p * 16) = sx[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = tid + k * 512;
      const int fv = p >> 2, sub = p & 3;
      const int fw = fv & 63, fh = (fv >> 6) & 3, fd = fv >> 8;
      const int pc = fd * 4 + (fh & 1) * 2 + (fw & 1);
      *reinterpret_cast<u32x4*>(yb + (((pc * 2 + (fh >> 1)) * 32 + (fw >> 1)) << 6) + sub * 16) = sy[k];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const bf16x8 one8 = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});

  int buf = 0;
  int64_t t = blockIdx.x;
  if (t < a.ntiles) {
    load_tile(t);
    store_tile(smem, smem + kWX);
  }
  __syncthreads();
  for (; t < a.ntiles; t += gridDim.x) {
    const int64_t tn = t + gridDim.x;
    if (tn < a.ntiles) load_tile(tn);
    const char* xb = smem + buf * (kWX + kWY);
    const char* yb = xb + kWX;
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
      for (int wh = 0; wh < 2; ++wh) {
        const bf16x8 bf = sg_tr_frag(yb + (((cls * 2 + th) * 32 + 16 * wh + kb) << 6) + colb);
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) {
          const int tz = tap >> 2, ty = (tap >> 1) & 1, tx = tap & 1;
          const bf16x8 af = sg_tr_frag(xb + ((((ca + tz) * 4 + th + cb + ty) * 34 + 16 * wh + kb + cc + tx) << 6) + colb);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[tap], 0, 0, 0);
        }
        if (ones) acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one8, bf, acc[8], 0, 0, 0);
      }
    if (tn < a.ntiles) store_tile(smem + (buf ^ 1) * (kWX + kWY), smem + (buf ^ 1) * (kWX + kWY) + kWX);
    __syncthreads();
    buf ^= 1;
  }
  // D[row = ci][col = co]: lane holds col r, rows (i & 3) + 8 * (i >> 2) + 4 * hh
  const int r = lane & 31, hh = lane >> 5;
  const bool slab = a.slab != 0;
  // bias gradient: row 0 of each wave's ones product = the column sums of its class plane; the eight are added in order
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);       // [8 waves][32] column sums
  if (ones && hh == 0) red[cls * 32 + r] = acc[8][0];
  __syncthreads();
  if (ones && tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) s += red[w8 * 32 + tid];
    float* dst = a.dbias + (int64_t)blockIdx.x * a.bslab + co_t * 32 + tid;
    if (slab) *dst = s;
    else unsafeAtomicAdd(dst, s);
  }
#pragma unroll
  for (int tap = 0; tap < 8; ++tap) {
    float* dst = a.dwt + (int64_t)blockIdx.x * a.slab + (((((int64_t)cls * 8 + tap) * a.ciT + ci_t) * a.coT + co_t) << 10);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (slab) dst[row * 32 + r] = acc[tap][i];
      else unsafeAtomicAdd(dst + row * 32 + r, acc[tap][i]);
    }
  }
}

// dw[kd][kh][kw][ci][co] = coef * sum of the (class, tap) tiles the original tap is summed into (x slabs, in order)
__global__ void upconv_subpixel_wgrad_finalize_kernel(const float* __restrict__ dwt, float* __restrict__ dw, const float* __restrict__ bsl,
                                                      float* __restrict__ db, float coef, int cin, int cout, int ciT, int coT, int nslab,
                                                      int64_t slab, int64_t bslab) {
  const int64_t total = (int64_t)27 * cin * cout;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) {
    const int co = (int)(i % cout);
    const int ci = (int)((i / cout) % cin);
    const int k = (int)(i / ((int64_t)cout * cin));
    const int kk[3] = {k / 9, (k / 3) % 3, k % 3};
    float s = 0.f;
    for (int b = 0; b < nslab; ++b) {
      const float* base = dwt + (int64_t)b * slab;
#pragma unroll
      for (int m = 0; m < 8; ++m) {      // per dimension: original tap 0 in (p0,t0),(p1,t0); 1 in (p0,t1),(p1,t0); 2 in (p0,t1),(p1,t1)
        int cls = 0, tap = 0;
#pragma unroll
        for (int dmn = 0; dmn < 3; ++dmn) {
          const int par = (m >> (2 - dmn)) & 1;
          const int tp = kk[dmn] == 0 ? 0 : (kk[dmn] == 2 ? 1 : (par == 0 ? 1 : 0));
          cls = cls * 2 + par;
          tap = tap * 2 + tp;
        }
        s += base[(((((int64_t)cls * 8 + tap) * ciT + (ci >> 5)) * coT + (co >> 5)) << 10) + (ci & 31) * 32 + (co & 31)];
      }
    }
    dw[i] = coef * s;
  }
  if (db != nullptr && bsl != nullptr && i < cout) {
    float s = 0.f;
    for (int b = 0; b < nslab; ++b) s += bsl[(int64_t)b * bslab + i];
    db[i] = s;
  }
}

static int subpix_wgrad_blocks(const sg_conv_shape* s) {
  const int pairs = sg_cdiv(s->cin, 32) * sg_cdiv(s->cout, 32);
  int gx = 256 / pairs;
  if (gx < 8) gx = 8;
  return gx;
}

extern "C" int sg_upconv3d_subpixel_wgrad_supported(const sg_conv_shape* s, sg_dtype dt) {
  return (s && dt == SG_BF16 && s->cin % 32 == 0 && s->cout % 32 == 0 && s->w % 32 == 0 && s->h % 2 == 0 && s->d >= 1 &&
          (int64_t)s->n * s->d * s->h * s->w * 8 * s->cout < (1ll << 40)) ? 1 : 0;
}

extern "C" size_t sg_upconv3d_subpixel_wgrad_workspace(const sg_conv_shape* s, sg_dtype dt) {
  if (!sg_upconv3d_subpixel_wgrad_supported(s, dt)) return 0;
  const size_t tile = (size_t)64 * sg_cdiv(s->cin, 32) * sg_cdiv(s->cout, 32) * 4096;
  const size_t bias = ((size_t)sg_cdiv(s->cout, 32) * 128 + 255) & ~(size_t)255;
  const size_t ns = sg_cfg().deterministic ? (size_t)subpix_wgrad_blocks(s) : 1;
  return ns * (tile + bias);
}

// s: the LOW-resolution shape; x [n,d,h,w,cin], gy [n,2d,2h,2w,cout]; dw [3][3][3][cin][cout] f32, dbias [cout] or NULL
extern "C" int sg_upconv3d_subpixel_wgrad(const void* x, const void* gy, float* dw, float* dbias, float coef, void* workspace,
                                          size_t workspace_bytes, const sg_conv_shape* s, sg_dtype dt, sg_stream_t st) {
  if (!s || !x || !gy || !dw || !workspace) return SG_EINVAL;
  if (!sg_upconv3d_subpixel_wgrad_supported(s, dt)) return SG_EUNSUPPORTED;
  if (!sg_aligned16(x) || !sg_aligned16(gy) || !sg_aligned16(workspace)) return SG_EALIGN;
  if (workspace_bytes < sg_upconv3d_subpixel_wgrad_workspace(s, dt)) return SG_EWORKSPACE;
  hipStream_t hs = sg_st(st);
  SubpixWgradArgs a;
  a.x = reinterpret_cast<const bf16_t*>(x); a.gy = reinterpret_cast<const bf16_t*>(gy);
  a.N = s->n; a.d = s->d; a.h = s->h; a.w = s->w; a.cin = s->cin; a.cout = s->cout;
  a.ciT = s->cin / 32; a.coT = s->cout / 32;
  a.nTh = s->h / 2; a.nTw = s->w / 32;
  a.ntiles = (int64_t)s->n * s->d * a.nTh * a.nTw;
  const bool det = sg_cfg().deterministic != 0;
  const int gx = subpix_wgrad_blocks(s);
  const size_t tile = (size_t)64 * a.ciT * a.coT * 4096;
  const size_t bias = ((size_t)a.coT * 128 + 255) & ~(size_t)255;
  const size_t ns = det ? (size_t)gx : 1;
  a.dwt = reinterpret_cast<float*>(workspace);
  float* bias_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ns * tile);
  a.dbias = dbias ? bias_ws : nullptr;
  a.slab = det ? (int64_t)(tile / 4) : 0;
  a.bslab = det ? (int64_t)(bias / 4) : 0;
  if (!det) {
    hipError_t e = hipMemsetAsync(workspace, 0, tile + bias, hs);
    if (e != hipSuccess) return (int)e;
  }
  sg_conv_shape full = *s;
  full.d *= 2; full.h *= 2; full.w *= 2; full.kd = full.kh = full.kw = 3; full.upsample_in = 1;
  sg_prof_scope prof(1, &full, dt, hs);
  auto kern = upconv_subpixel_wgrad_kernel;
  SG_ALLOW_160K_LDS(kern);
  SG_KNAME("upconv_subpixel_wgrad");
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)(a.ciT * a.coT)), dim3(512), 2 * (size_t)(kWX + kWY), hs, a);
  const int64_t total = (int64_t)27 * s->cin * s->cout;
  hipLaunchKernelGGL(upconv_subpixel_wgrad_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, hs, a.dwt, dw,
                     dbias ? bias_ws : nullptr, dbias, coef, s->cin, s->cout, a.ciT, a.coT, (int)ns, a.slab, a.bslab);
  hipError_t e2 = hipGetLastError();
  prof.done((int)e2);
  return (int)e2;
}
