// Small-channel 2-D convolutions (D == 1 volumes, 1 x 3 x 3 taps, 4 / 8 / 16 channels a side): the top levels of the
// SURFGAN_2D pgan at 512^2 and 1024^2 (SURFGAN_2D/networks/ops.py:99-102, pgan/generator.py, pgan/discriminator.py; 'xs':
// 4 and 8 filters).  These layers are HBM-bound by two orders of magnitude (4 -> 8 at 1024^2: 48 bytes and 576 FLOP per
// pixel), and a 32 x 32 MFMA tile would be 1/32 to 1/8 full: they run on the vector ALU.  A wave walks a 64-pixel-wide strip
// of rows with a rolling three-row window in registers: per output row it requests ONE new input row (three lane-contiguous
// buffer loads, one row ahead of its use, zeros outside the image from the buffer bounds check -- no branches), every input
// row comes from HBM once (the strip's neighbours find its edge columns in L1 / L2), the weights sit in scalar registers.
//   forward / data gradient:  y[p][co] = epilogue(sum_{tap,ci} W[tap][ci][co] * x[p + tap][ci])
//   weight gradient:          dw[tap][ci][co] = sum_p x[p + tap][ci] * dy[p][co],  db[co] = sum_p dy[p][co]
// The weight gradient keeps 9 * CIN * CS sums per thread (CS = a slice of the output channels, blockIdx.y picks the slice),
// reduces them over the wave by shuffles and over the block through LDS, writes ONE slab per block and a second kernel adds
// the slabs in a fixed order: no atomics, bit-reproducible.
#include "common.h"
#include "prof.h"
#include <type_traits>

struct SmallFwdArgs {
  const void* x;
  void* y;
  const float* w;            // [9][CIN][COUT] f32: coef * w (mirrored / transposed for the data gradient), see sg_small_tail
  const float* bias;
  float* pn_scale;
  const uint32_t* mask_bits;
  uint32_t* sign_out;
  float mask_slope, slope, eps;
  int act, pixel_norm;
  int N, H, W;
  int R;                     // rows per strip: a wave walks a 64-pixel-wide strip of R rows with a rolling 3-row window
  int strips, segs;          // strips per image, 64-pixel segments per row
  int64_t items;             // N * strips * segs
};

constexpr uint32_t SG_DEAD = 0x80000000u;     // byte offset beyond every buffer: a buffer load there returns zeros, no branch

// C channels of one pixel through a buffer resource (zeros when the offset is SG_DEAD)
template <typename T, int C>
__device__ __forceinline__ void sg_bload(float (&v)[C], __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < C / 4; ++i) {
      const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u * i, soff, 0));
      v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
    }
  } else if constexpr (C == 4) {
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    v[0] = __builtin_bit_cast(float, t[0] << 16); v[1] = __builtin_bit_cast(float, t[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, t[1] << 16); v[3] = __builtin_bit_cast(float, t[1] & 0xffff0000u);
  } else {
#pragma unroll
    for (int i = 0; i < C / 8; ++i) {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u * i, soff, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[8 * i + 2 * j] = __builtin_bit_cast(float, t[j] << 16);
        v[8 * i + 2 * j + 1] = __builtin_bit_cast(float, t[j] & 0xffff0000u);
      }
    }
  }
}

template <typename T, int C>
__device__ __forceinline__ void sg_store_px(T* __restrict__ p, const float (&v)[C]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < C / 4; ++i) reinterpret_cast<f32x4*>(p)[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
  } else {
#pragma unroll
    for (int i = 0; i < C / 4; ++i) {
      bf16x4 t = {(bf16_t)v[4 * i], (bf16_t)v[4 * i + 1], (bf16_t)v[4 * i + 2], (bf16_t)v[4 * i + 3]};
      reinterpret_cast<bf16x4*>(p)[i] = t;
    }
  }
}

typedef const __attribute__((address_space(4))) float* sg_const_f32;   // wave-uniform reads become scalar loads

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sg_rsrc(const void* base, int64_t off, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + off, 0, (int)bytes, 0x00020000);
}

// item (wave-uniform) -> sample, first row of the strip, first column of the segment
struct sg_small_item { int n, h0, w0; };
__device__ __forceinline__ sg_small_item sg_small_decode(int64_t item, int segs, int strips, int R) {
  sg_small_item it;
  const int seg = (int)(item % segs);
  const int64_t q = item / segs;
  it.w0 = seg * 64;
  it.h0 = (int)(q % strips) * R;
  it.n = (int)(q / strips);
  return it;
}

template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(SmallFwdArgs a) {
  constexpr int ES = (int)sizeof(T);
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  sg_const_f32 wk = (sg_const_f32)a.w;
  const int lane = threadIdx.x & 63;
  const int64_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int64_t simg = (int64_t)a.H * a.W;                           // pixels per sample
  const uint32_t rowb = (uint32_t)a.W * CIN * ES;                    // bytes per input row
  for (int64_t item = wave; item < a.items; item += nwaves) {
    const sg_small_item it = sg_small_decode(item, a.segs, a.strips, a.R);
    const int w = it.w0 + lane;
    const bool live = w < a.W;
    const __amdgpu_buffer_rsrc_t rx = sg_rsrc(a.x, (int64_t)it.n * simg * CIN * ES, simg * CIN * ES);
    const __amdgpu_buffer_rsrc_t rm = sg_rsrc(a.mask_bits ? (const void*)a.mask_bits : a.x, a.mask_bits ? (int64_t)it.n * simg * 4 : 0,
                                              a.mask_bits ? simg * 4 : 0);
    uint32_t colo[3];                                                // byte offset of columns w - 1, w, w + 1 within a row
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int ww = w + c - 1;
      colo[c] = (live && ww >= 0 && ww < a.W) ? (uint32_t)ww * CIN * ES : SG_DEAD;
    }
    const int hend = it.h0 + a.R < a.H ? it.h0 + a.R : a.H;
    float win[3][3][CIN];                                            // rolling window: slot (row mod 3), column, channel
    auto load_row = [&](int slot, int hh) {
      const bool rv = hh >= 0 && hh < a.H;                           // wave-uniform
#pragma unroll
      for (int c = 0; c < 3; ++c) sg_bload<T, CIN>(win[slot][c], rx, rv ? colo[c] : SG_DEAD, rv ? (uint32_t)hh * rowb : 0u);
    };
    // slots: row h lives in slot (h - h0 + 1) % 3
    load_row(0, it.h0 - 1);
    load_row(1, it.h0);
    auto row = [&](auto SL, int h) {                                 // SL = slot of row h - 1; h in slot SL + 1, h + 1 in SL + 2
      constexpr int s0 = decltype(SL)::value, s1 = (s0 + 1) % 3, s2 = (s0 + 2) % 3;
      load_row(s2, h + 1);                                           // (in flight during this row's arithmetic)
      const uint32_t mb = a.mask_bits ? __builtin_amdgcn_raw_buffer_load_b32(rm, live ? (uint32_t)w * 4u : SG_DEAD,
                                                                              (uint32_t)h * (uint32_t)a.W * 4u, 0) : 0u;
      float acc[COUT];
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] = a.bias ? a.bias[co] : 0.f;
      // the weights are re-read from the scalar cache every row: hoisted out of the row loop they would need 9 * CIN * COUT
      // scalar registers and be spilled to vector lanes (400-700 v_readlane per row in the first build)
      sg_const_f32 wrow = wk;
      asm volatile("" : "+s"(wrow));
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        constexpr int slots[3] = {s0, s1, s2};
        // rows h - 1 and h are complete; row h + 1 was requested above: its use comes last
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          sg_const_f32 wt = wrow + (kh * 3 + kw) * CIN * COUT;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[co] = fmaf(win[slots[kh]][kw][ci], wt[ci * COUT + co], acc[co]);
        }
      }
      if (a.act) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaxf(acc[co], acc[co] * a.slope);
      }
      const int64_t pix = ((int64_t)it.n * a.H + h) * a.W + w;
      if (a.pixel_norm) {
        float ss = 0.f;
#pragma unroll
        for (int co = 0; co < COUT; ++co) ss += acc[co] * acc[co];
        const float sc = rsqrtf(ss * (1.0f / COUT) + a.eps);
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] *= sc;
        if (a.pn_scale && live) a.pn_scale[pix] = sc;
      }
      if (a.sign_out) {
        uint32_t sw = 0u;
#pragma unroll
        for (int co = 0; co < COUT; ++co) sw |= (__builtin_bit_cast(uint32_t, acc[co]) >> 31) << co;
        if (live) a.sign_out[pix] = sw;
      }
      if (a.mask_bits) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] *= ((mb >> co) & 1u) ? a.mask_slope : 1.0f;
      }
      if (live) sg_store_px<T, COUT>(y + pix * COUT, acc);
    };
    for (int h = it.h0; h < hend; h += 3) {
      row(std::integral_constant<int, 0>{}, h);
      if (h + 1 < hend) row(std::integral_constant<int, 1>{}, h + 1);
      if (h + 2 < hend) row(std::integral_constant<int, 2>{}, h + 2);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
struct SmallWgradArgs {
  const void* x;
  const void* dy;
  float* slabs;              // [blocks per slice][9 * CIN * COUT + COUT]
  int N, H, W;
  int R, strips, segs;
  int64_t items;
};

// sum over the 64 lanes, identical in every lane that reads it back: DPP butterflies inside the 16-lane rows (no LDS,
// nothing to wait for), then the four row sums through scalar registers in a fixed order
__device__ __forceinline__ float sg_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}

template <typename T, int CINT, int COUT, int CS>      // CINT input channels in the tensor; a block takes CIN = min(CINT, 8) of them
__global__ __launch_bounds__(256, CINT == 4 ? 3 : 2) void conv_small_wgrad_kernel(SmallWgradArgs a) {   // 3 (2) blocks per CU: <= 168 (256) registers
  constexpr int ES = (int)sizeof(T);
  constexpr int CIN = CINT < 8 ? CINT : 8, SI = CINT / CIN;          // SI groups of input channels
  constexpr int NACC = 9 * CIN * CS, NTOT = NACC + CS;
  __shared__ float red[4][NTOT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int S = (COUT / CS) * SI;                                // slices: output channels x input-channel groups
  // The S blocks that take the slices of the same pixels must share an L2: consecutive block ids go round the 8 XCDs, so
  // a pixel group's slices are the ids with the same id % 8 (dealt as neighbours they sat on S different XCDs and every
  // one fetched x and its dy lines from HBM again: FETCH_SIZE 4-8x the algorithmic bytes).  gridDim.x = S * nxb, nxb % 8 == 0.
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int xb = (jb / S) * 8 + xcd, nxb = gridDim.x / S;
  const int c0 = ((jb % S) / SI) * CS;                               // this block's slice of the output channels ...
  const int ci0 = ((jb % S) % SI) * CIN;                             // ... and of the input channels
  const int64_t wave = __builtin_amdgcn_readfirstlane(xb * 4 + wv), nwaves = (int64_t)nxb * 4;
  const int64_t simg = (int64_t)a.H * a.W;
  const uint32_t rowb = (uint32_t)a.W * CINT * ES, rowg = (uint32_t)a.W * COUT * ES;
  float acc[NACC], accb[CS];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
#pragma unroll
  for (int i = 0; i < CS; ++i) accb[i] = 0.f;
  for (int64_t item = wave; item < a.items; item += nwaves) {
    const sg_small_item it = sg_small_decode(item, a.segs, a.strips, a.R);
    const int w = it.w0 + lane;
    const bool live = w < a.W;
    const __amdgpu_buffer_rsrc_t rx = sg_rsrc(a.x, (int64_t)it.n * simg * CINT * ES, simg * CINT * ES);
    const __amdgpu_buffer_rsrc_t rg = sg_rsrc(a.dy, (int64_t)it.n * simg * COUT * ES, simg * COUT * ES);
    uint32_t colo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int ww = w + c - 1;
      colo[c] = (live && ww >= 0 && ww < a.W) ? (uint32_t)(ww * CINT + ci0) * ES : SG_DEAD;
    }
    const uint32_t go = live ? (uint32_t)(w * COUT + c0) * ES : SG_DEAD;
    const int hend = it.h0 + a.R < a.H ? it.h0 + a.R : a.H;
    float win[3][3][CIN], g[CS], gn[CS];
    auto load_row = [&](int slot, int hh) {
      const bool rv = hh >= 0 && hh < a.H;
#pragma unroll
      for (int c = 0; c < 3; ++c) sg_bload<T, CIN>(win[slot][c], rx, rv ? colo[c] : SG_DEAD, rv ? (uint32_t)hh * rowb : 0u);
    };
    auto load_g = [&](float (&dst)[CS], int hh) {                    // dy of row hh (zeros beyond the strip: adds nothing)
      const bool rv = hh < hend;
      if constexpr (CS >= 4) {
        sg_bload<T, CS>(dst, rg, rv ? go : SG_DEAD, rv ? (uint32_t)hh * rowg : 0u);
      } else {
#pragma unroll
        for (int c = 0; c < CS; ++c) {
          if constexpr (ES == 4) {
            dst[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, rv ? go + 4u * c : SG_DEAD, rv ? (uint32_t)hh * rowg : 0u, 0));
          } else {
            dst[c] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rg, rv ? go + 2u * c : SG_DEAD, rv ? (uint32_t)hh * rowg : 0u, 0) << 16);
          }
        }
      }
    };
    load_row(0, it.h0 - 1);
    load_row(1, it.h0);
    load_g(g, it.h0);
    auto row = [&](auto SL, int h) {
      constexpr int s0 = decltype(SL)::value, s1 = (s0 + 1) % 3, s2 = (s0 + 2) % 3;
      load_row(s2, h + 1);
      load_g(gn, h + 1);                                             // next row's dy, in flight during this row's arithmetic
#pragma unroll
      for (int c = 0; c < CS; ++c) accb[c] += g[c];
      constexpr int slots[3] = {s0, s1, s2};
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int c = 0; c < CS; ++c)
              acc[((kh * 3 + kw) * CIN + ci) * CS + c] = fmaf(win[slots[kh]][kw][ci], g[c], acc[((kh * 3 + kw) * CIN + ci) * CS + c]);
#pragma unroll
      for (int c = 0; c < CS; ++c) g[c] = gn[c];
    };
    for (int h = it.h0; h < hend; h += 3) {      // unrolled by the three window slots: every slot index is a compile-time constant
      row(std::integral_constant<int, 0>{}, h);
      if (h + 1 < hend) row(std::integral_constant<int, 1>{}, h + 1);
      if (h + 2 < hend) row(std::integral_constant<int, 2>{}, h + 2);
    }
  }
  // wave reduction (fixed order), block reduction through LDS, one slab per block
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const float v = sg_wave_sum(acc[i]);
    if (lane == (i & 63)) red[wv][i] = v;
  }
#pragma unroll
  for (int i = 0; i < CS; ++i) {
    const float v = sg_wave_sum(accb[i]);
    if (lane == 0) red[wv][NACC + i] = v;
  }
  __syncthreads();
  float* slab = a.slabs + (int64_t)xb * (9 * CINT * COUT + COUT);
  for (int i = threadIdx.x; i < NTOT; i += 256) {
    const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    if (i < NACC) {
      const int c = i % CS, tc = i / CS;                             // tc = tap * CIN + ci (of this block's input channels)
      slab[((tc / CIN) * CINT + ci0 + tc % CIN) * COUT + c0 + c] = v;
    } else if (ci0 == 0) {                                           // (the bias sums do not depend on the input-channel group)
      slab[9 * CINT * COUT + c0 + (i - NACC)] = v;
    }
  }
}

// 16 outputs per block, 16 threads per output: thread j adds slabs j, j + 16, ... in order, then a fixed binary tree over
// the 16 partial sums (the same order every run: reproducible)
__global__ __launch_bounds__(256) void conv_small_wgrad_finalize_kernel(const float* __restrict__ slabs, int nslab, int nw, int cout,
                                                                        float coef, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float part[16][17];
  const int o = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int i = blockIdx.x * 16 + o;
  float s = 0.f;
  if (i < nw + cout)
    for (int b = j; b < nslab; b += 16) s += slabs[(int64_t)b * (nw + cout) + i];
  part[o][j] = s;
  __syncthreads();
  if (j == 0 && i < nw + cout) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = part[o][k];
#pragma unroll
    for (int st = 1; st < 16; st <<= 1)
#pragma unroll
      for (int k = 0; k < 16; k += 2 * st) t[k] += t[k + st];
    if (i < nw) dw[i] = t[0] * coef;
    else if (db) db[i - nw] = t[0];
  }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
// Which layers these kernels take.  The packed weight image of such a layer carries a plain f32 [9][cin][cout] copy of
// coef * w behind the MFMA fragment image (sg_small_tail_*), written by sg_conv3d_pack_weights.
bool sg_small_eligible(const sg_conv_shape* s) {
  auto ok = [](int c) { return c == 4 || c == 8 || c == 16; };
  return s->kd == 1 && s->kh == 3 && s->kw == 3 && s->d == 1 && !s->upsample_in && ok(s->cin) && ok(s->cout) &&
         s->cin * s->cout <= 128 && (s->w % 2) == 0 && (int64_t)s->n * s->h * s->w >= (1 << 16) &&
         (int64_t)s->h * s->w * 16 * 4 < (1ll << 31);      // (buffer resources are rebased per sample)
}
size_t sg_small_tail_bytes(const sg_conv_shape* s) { return sg_small_eligible(s) ? (size_t)9 * s->cin * s->cout * 4 : 0; }

__global__ void pack_small_kernel(const float* __restrict__ w, float* __restrict__ out, float coef, int cin, int cout, int flip,
                                  int bf16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * cin * cout) return;
  const int co = i % cout, ci = (i / cout) % cin, tap = i / (cin * cout);
  float v = flip ? w[((int64_t)(8 - tap) * cout + co) * cin + ci] : w[i];     // flip: source is [taps][cout][cin], mirrored
  v *= coef;
  if (bf16) v = (float)(bf16_t)v;                                               // as the fragment image holds it
  out[i] = v;
}

int sg_small_pack(const float* w, float coef, int flip, void* tail, const sg_conv_shape* s, sg_dtype dt, hipStream_t st) {
  const int n = 9 * s->cin * s->cout;
  hipLaunchKernelGGL(pack_small_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, reinterpret_cast<float*>(tail), coef, s->cin,
                     s->cout, flip, dt == SG_BF16 ? 1 : 0);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// Rows per strip: long strips amortise the two halo rows a strip re-reads; short ones keep every wave busy.  The longest
// strip (<= 64 rows) that still gives each of `waves` waves an item.
static int small_rows_per_strip(const sg_conv_shape* s, int waves) {
  const int64_t segs = sg_cdiv(s->w, 64);
  int r = 64;
  while (r > 4 && (int64_t)s->n * sg_cdiv(s->h, r) * segs < waves) r >>= 1;
  return r;
}

template <typename T, int CIN>
static int small_fwd_cout(const SmallFwdArgs& a, int cout, unsigned blocks, hipStream_t st) {
  switch (cout) {
    case 4: hipLaunchKernelGGL((conv_small_fwd_kernel<T, CIN, 4>), dim3(blocks), dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((conv_small_fwd_kernel<T, CIN, 8>), dim3(blocks), dim3(256), 0, st, a); break;
    default:
      if constexpr (CIN < 16) hipLaunchKernelGGL((conv_small_fwd_kernel<T, CIN, 16>), dim3(blocks), dim3(256), 0, st, a);
      else return SG_EUNSUPPORTED;
      break;
  }
  return SG_OK;
}

// epilogue fields as sg_conv3d_fwd reads them from sg_conv_epilogue; `tail` = the plain weights behind the fragment image
int sg_small_fwd(const void* x, const void* tail, void* y, const sg_conv_shape* s, const float* bias, int act, float slope,
                 int pixel_norm, float eps, float* pn_scale, const uint32_t* mask_bits, float mask_slope, uint32_t* sign_out,
                 sg_dtype dt, hipStream_t st) {
  SmallFwdArgs a;
  a.x = x; a.y = y; a.w = reinterpret_cast<const float*>(tail); a.bias = bias; a.pn_scale = pn_scale; a.mask_bits = mask_bits;
  a.sign_out = sign_out; a.mask_slope = mask_slope; a.slope = slope; a.eps = eps; a.act = act; a.pixel_norm = pixel_norm;
  a.N = s->n; a.H = s->h; a.W = s->w;
  a.segs = sg_cdiv(s->w, 64);
  a.R = small_rows_per_strip(s, 256 * 8 * 4);
  a.strips = sg_cdiv(s->h, a.R);
  a.items = (int64_t)s->n * a.strips * a.segs;
  int64_t nb = (a.items + 3) / 4;
  if (nb > 256 * 8) nb = 256 * 8;
  const unsigned blocks = (unsigned)nb;
  SG_KNAME("conv_small_fwd<%s>", dt == SG_BF16 ? "bf16" : "f32");
  if (dt == SG_BF16) {
    switch (s->cin) {
      case 4: small_fwd_cout<bf16_t, 4>(a, s->cout, blocks, st); break;
      case 8: small_fwd_cout<bf16_t, 8>(a, s->cout, blocks, st); break;
      default: small_fwd_cout<bf16_t, 16>(a, s->cout, blocks, st); break;
    }
  } else {
    switch (s->cin) {
      case 4: small_fwd_cout<float, 4>(a, s->cout, blocks, st); break;
      case 8: small_fwd_cout<float, 8>(a, s->cout, blocks, st); break;
      default: small_fwd_cout<float, 16>(a, s->cout, blocks, st); break;
    }
  }
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// weight gradient (16 -> 16 stays on the MFMA kernels: a thread's slice of the sums would be one output channel wide, and
// the forward kernel's 2304 weights no longer fit the scalar registers)
bool sg_small_wgrad_eligible(const sg_conv_shape* s) {     // (16 input channels run as two groups of 8: the three-row window of all 16
  return (sg_small_eligible(s) && s->cin <= 8) ||          //  alone would be 144 registers; cout 8 or 16 -- at 32 the 64 slices lose to the MFMA kernel: 306 against 245 us)
         (s->kd == 1 && s->kh == 3 && s->kw == 3 && s->d == 1 && !s->upsample_in && s->cin == 16 &&
          (s->cout == 8 || s->cout == 16) && (s->w % 2) == 0 && (int64_t)s->n * s->h * s->w >= (1 << 16) &&
          (int64_t)s->h * s->w * 32 * 4 < (1ll << 31));
}
// A thread keeps 72 sums (9 taps x CIN x CS): CS = 2 output channels at 4 input channels, 1 at 8.  (144 sums per thread were
// tried first: 256 registers, one or two waves per SIMD, the row-ahead loads no longer hidden -- 0.85 TB/s.)
static int small_wgrad_slices(const sg_conv_shape* s) { return s->cout / (s->cin == 4 ? 2 : 1) * (s->cin > 8 ? s->cin / 8 : 1); }
static int small_wgrad_blocks(const sg_conv_shape* s) {      // blocks PER SLICE
  int64_t cap = (256 * (s->cin == 4 ? 3 : 2)) / small_wgrad_slices(s);      // three (two) blocks per CU over all slices
  if (cap < 32) cap = 32;
  const int r = small_rows_per_strip(s, (int)cap * 4);
  const int64_t items = (int64_t)s->n * sg_cdiv(s->h, r) * sg_cdiv(s->w, 64);
  int64_t nb = (items + 3) / 4;
  if (nb > cap) nb = cap;
  return (int)((nb + 7) & ~(int64_t)7);      // whole rounds of the 8 XCDs (blocks without items store zero slabs)
}
size_t sg_small_wgrad_workspace(const sg_conv_shape* s) {
  return sg_small_wgrad_eligible(s) ? (size_t)small_wgrad_blocks(s) * (9 * s->cin * s->cout + s->cout) * 4 : 0;
}

template <typename T>
static int small_wgrad_launch(const SmallWgradArgs& a, const sg_conv_shape* s, unsigned nb, hipStream_t st) {
  const int key = s->cin * 100 + s->cout;
#define SG_SW(CI, CO, CS_) case CI * 100 + CO: hipLaunchKernelGGL((conv_small_wgrad_kernel<T, CI, CO, CS_>), dim3(nb * (CO / CS_) * (CI > 8 ? CI / 8 : 1)), dim3(256), 0, st, a); break;
  switch (key) {
    SG_SW(4, 4, 2) SG_SW(4, 8, 2) SG_SW(4, 16, 2)
    SG_SW(8, 4, 1) SG_SW(8, 8, 1) SG_SW(8, 16, 1)
    SG_SW(16, 8, 1) SG_SW(16, 16, 1)
    default: return SG_EUNSUPPORTED;
  }
#undef SG_SW
  return SG_OK;
}

int sg_small_wgrad(const void* x, const void* dy, float* dw, float* dbias, float coef, void* workspace, size_t workspace_bytes,
                   const sg_conv_shape* s, sg_dtype dt, hipStream_t st) {
  if (workspace_bytes < sg_small_wgrad_workspace(s) || !workspace) return SG_EWORKSPACE;
  SmallWgradArgs a;
  a.x = x; a.dy = dy; a.slabs = reinterpret_cast<float*>(workspace);
  a.N = s->n; a.H = s->h; a.W = s->w;
  a.segs = sg_cdiv(s->w, 64);
  const unsigned nb = (unsigned)small_wgrad_blocks(s);
  int64_t cap = (256 * (s->cin == 4 ? 3 : 2)) / small_wgrad_slices(s);
  if (cap < 32) cap = 32;
  a.R = small_rows_per_strip(s, (int)cap * 4);
  a.strips = sg_cdiv(s->h, a.R);
  a.items = (int64_t)s->n * a.strips * a.segs;
  SG_KNAME("conv_small_wgrad<%s>", dt == SG_BF16 ? "bf16" : "f32");
  int rc = dt == SG_BF16 ? small_wgrad_launch<bf16_t>(a, s, nb, st) : small_wgrad_launch<float>(a, s, nb, st);
  if (rc != SG_OK) return rc;
  SG_LAUNCH_CHECK();
  const int nw = 9 * s->cin * s->cout;
  hipLaunchKernelGGL(conv_small_wgrad_finalize_kernel, dim3((nw + s->cout + 15) / 16), dim3(256), 0, st, a.slabs, (int)nb, nw,
                     s->cout, coef, dw, dbias);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
