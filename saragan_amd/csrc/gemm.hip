// GEMM-tiled convolution for the LOW-RESOLUTION levels (whole-plane tiles, D*H*W <= 1024 voxels per sample, >= 128
// channels): pgan/generator.py:26-45 (generator_in ... block 3) and pgan/discriminator.py:48-68 at 1x4x4, 2x8x8, 4x16x16.
// There the spatial kernels of conv3d.hip run at 0.02-0.3 of the MFMA peak: a block owned 128 voxels x 32 output
// channels and streamed its whole weight slab per tile -- 256 MB through L2 for a 4.7 MB weight tensor (DESIGN_NOTES.md section 7).
//
// Here the batch is folded into M: a block owns 256 consecutive voxels (1 plane of 16x16, one 2x8x8 sample pair, sixteen
// 1x4x4 samples) x 128 output channels, eight waves as 4 (M) x 2 (N), each 64 x 64 (four accumulator tiles: every fragment
// read feeds two MFMAs).  K = taps x cin runs in steps of (16-channel chunk, group of <= 9 taps): the step's weight slab
// (<= 36 KiB, fragment order, straight from the packed image) and -- once per chunk -- the halo image (<= 31 KiB) are
// double buffered in LDS against the arithmetic.  Per step a CU moves ~45 KiB for 288 MFMAs: 20 B/clk, the L2 -> LDS rate.
// Small levels do not fill 256 CUs with 256 x 128 tiles (2x8x8 at batch 32: 64 tiles): K is then split over blocks, the
// f32 partial tiles go to a workspace and a second kernel adds them in order and applies the epilogue.
//
// Forward and data gradient (packed weights with transpose_flip) alike; optional fused nearest-x2 gather of the input
// (upsample_in: the generator's conv_1 of the 2x8x8 / 4x16x16 levels).  GEMM orientation as in conv3d.hip.
//
// Small batches (round 5: the reference's own batch rule gives 1-4 volumes per rank at 128^2 and above): a tile may hold samples
// beyond the batch (sixteen 1x4x4 samples per tile at batch 2: their halo rows stay zero, their outputs are not stored), and the
// 3x3x3 layers of the 4x16x16 level run here too while the batch is small (one plane per tile, K steps = (16-channel chunk, kd)):
// the spatial kernels took 36-124 us per launch there at batch 2, streaming the whole weight tensor through every 128-voxel tile.
// A tile covers whole planes, so its halo in H and W is always zero: the image in LDS is cleared once and only interior voxels
// are staged (<= 3 pieces of 16 B per thread and chunk).
#include "common.h"
#include "prof.h"

struct GemmConvArgs {
  const bf16_t* x;
  const char* wp;            // [chunk16][tap][ntile][lane][16 B]
  bf16_t* y;
  float* partial;            // ksplit > 1: [ksplit][M][cout] f32
  const float* bias;
  const uint32_t* mask_bits;
  uint32_t* sign_out;
  float slope, mask_slope;
  int act;
  int N, d, h, w, cin, cout, nchunk, ntile, taps, kh, kw, pd, ph, pw;
  int ups;                   // x is the half-resolution tensor
  int TN, TD;                // tile: TN samples x TD planes x h x w = 256 voxels
  int HD, HH, HW, hv;        // halo extents and halo voxels of the tile
  int nTd;                   // D tiles per sample
  int TG, ntg;               // taps per step, tap groups
  int ksplit, steps_per_split, steps;   // steps = nchunk * ntg
};

namespace {
constexpr int kGW = 9 * 4 * 1024;         // weight slab buffer: 9 taps x 4 N tiles x 1 KiB
constexpr int kGX = 3 * 18 * 18 * 32;     // halo buffer: one 16x16 plane with a 3x3x3 halo, 32 B per voxel
}

// LeakyReLU (optional), sign words, LeakyReLU-backward mask (optional), bf16 store of one 32-voxel x 32-channel tile
__device__ __forceinline__ void gemm_tile_out(const GemmConvArgs& a, f32x16 v, int64_t m, int nt, int hh) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float e = v[i] + (a.bias ? a.bias[nt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh] : 0.f);
    if (a.act) e = sg_lrelu(e, a.slope);
    v[i] = e;
  }
  if (a.sign_out) {
    const uint32_t sw = sg_sign_word(v, hh);
    if (hh == 0) a.sign_out[m * a.ntile + nt] = sw;
  }
  if (a.mask_bits) sg_apply_sign_word(v, a.mask_bits[m * a.ntile + nt], hh, a.mask_slope);
  sg_store_tile_row_bf16(a.y + m * a.cout + nt * 32, v, hh, true);
}

__global__ __launch_bounds__(512) void conv_gemm_kernel(GemmConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wbuf = smem;                 // 2 x kGW
  char* const xbuf = smem + 2 * kGW;       // 2 x kGX
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int mw = wave >> 1, nw = wave & 1;
  const int nb = blockIdx.y;               // block of 128 output channels
  const int ks = blockIdx.z;               // K split
  // tile origin: tile index -> (sample group, D tile)
  const int tile = blockIdx.x;
  const int td_i = tile % a.nTd, n0 = (tile / a.nTd) * a.TN, d0 = td_i * a.TD;
  const int rowB = a.HW * 32, planeB = a.HH * rowB, sampB = a.HD * planeB;
  const int hw_ = a.h * a.w;
  // this lane's voxel in each of the wave's two column tiles: LDS byte offset of its halo row (tap 0,0,0)
  int xbase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int q = (mw * 2 + mt) * 32 + r;                 // voxel of the tile, (tn, td, h, w) order
    const int vw = q % a.w, vh = (q / a.w) % a.h, vd = (q / hw_) % a.TD, vn = q / (hw_ * a.TD);
    xbase[mt] = vn * sampB + vd * planeB + vh * rowB + vw * 32 + hh * 16;
  }
  // staging plan: up to 3 interior pieces and 5 weight pieces (16 B) per thread and step.  Interior voxel (cn, cd, vh, vw) of the
  // halo image: its H / W border is zero for every tile (tiles are whole planes) and is cleared once, below
  const int Di = a.ups ? a.d >> 1 : a.d, Hi = a.ups ? a.h >> 1 : a.h, Wi = a.ups ? a.w >> 1 : a.w;
  int xoff[3];              // (element offsets: n * d * h * w * cin < 2^31 is checked by the plan)
  int xdst[3];
  const int ipieces = a.TN * a.HD * hw_ * 2;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int p = tid + k * 512;
    xdst[k] = -1; xoff[k] = -1;
    if (p < ipieces) {
      const int iv = p >> 1, half = p & 1;
      const int vw = iv % a.w, vh = (iv / a.w) % a.h, cd = (iv / hw_) % a.HD, cn = iv / (hw_ * a.HD);
      int dd = d0 + cd - a.pd, yy = vh, ww = vw;
      if (dd >= 0 && dd < a.d && n0 + cn < a.N) {
        xdst[k] = cn * sampB + cd * planeB + (vh + a.ph) * rowB + (vw + a.pw) * 32 + half * 16;
        if (a.ups) { dd >>= 1; yy >>= 1; ww >>= 1; }
        xoff[k] = ((((n0 + cn) * Di + dd) * Hi + yy) * Wi + ww) * a.cin + half * 8;
      }
    }
  }
  const int wpieces = a.TG * 4 * 64;         // pieces of a full slab (the last group of 27 = 3 x 9 is full as well)
  auto load_x = [&](u32x4 (&st)[3], int chunk) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      st[k] = *reinterpret_cast<const u32x4*>(a.x + (xoff[k] >= 0 ? xoff[k] : 0) + chunk * 16);
    }
  };
  auto store_x = [&](const u32x4 (&st)[3], char* dst) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (xdst[k] >= 0) *reinterpret_cast<u32x4*>(dst + xdst[k]) = st[k];
  };
  auto load_w = [&](u32x4 (&st)[5], int chunk, int tg) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int p = tid + k * 512;
      if (p >= wpieces) p = wpieces - 1;      // (clamped: the duplicate is stored to the same place)
      const int tl = p >> 8, rem = p & 255, ntl = rem >> 6, ln = rem & 63;
      int nti = nb * 4 + ntl;      // (cout = 64 (mod 128): the last block's upper two tiles do not exist -- read tile ntile - 1 again, store nothing)
      if (nti >= a.ntile) nti = a.ntile - 1;
      st[k] = *reinterpret_cast<const u32x4*>(a.wp + ((((int64_t)chunk * a.taps + tg * a.TG + tl) * a.ntile + nti) << 10) + ln * 16);
    }
  };
  auto store_w = [&](const u32x4 (&st)[5], char* dst) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int p = tid + k * 512;
      if (p >= wpieces) p = wpieces - 1;
      *reinterpret_cast<u32x4*>(dst + p * 16) = st[k];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  const int s0 = ks * a.steps_per_split;
  const int s1 = s0 + a.steps_per_split < a.steps ? s0 + a.steps_per_split : a.steps;
  // One step = (16-channel chunk, tap group of 9 taps = one kd).  These launches are short (3-32 steps per block) and start cold,
  // so a step's weights are requested TWO steps ahead: two register sets alternate (A: even steps, B: odd steps), each written to
  // the LDS buffer of its parity at the end of the step before its use.  (With one set, requested one step ahead, a 2x8x8 layer
  // took 60 us: every step waited ~2 us for L2.)  The halo image changes per CHUNK: requested during the chunk's last step
  // (one set), written to the other image buffer at that step's end.
  u32x4 sx[3], swA[5], swB[5];
  {   // clear both halo images (the border, the planes outside the volume and the samples beyond the batch stay zero)
    const int nclr = (2 * kGX) / 16;
    for (int p = tid; p < nclr; p += 512) *reinterpret_cast<u32x4*>(xbuf + p * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  int chunk = s0 / a.ntg, tg = s0 - chunk * a.ntg;      // (scalars: the step, its chunk and tap group; the next two steps' likewise)
  int chunk1 = tg + 1 < a.ntg ? chunk : chunk + 1, tg1 = tg + 1 < a.ntg ? tg + 1 : 0;
  load_x(sx, chunk);
  load_w(swA, chunk, tg);
  if (s0 + 1 < s1) load_w(swB, chunk1, tg1);
  __syncthreads();
  store_x(sx, xbuf);
  store_w(swA, wbuf);
  __syncthreads();
  int xpar = 0;             // which halo image holds the current chunk
  auto compute = [&](int buf, const char* xs) {
    const char* ws = wbuf + buf * kGW + (nw * 2) * 1024 + lane * 16;
#pragma unroll
    for (int tl = 0; tl < 9; ++tl) {
      const int toff = (tl / 3) * rowB + (tl % 3) * 32;
      const u32x4 x0 = *reinterpret_cast<const u32x4*>(xs + xbase[0] + toff);
      const u32x4 x1 = *reinterpret_cast<const u32x4*>(xs + xbase[1] + toff);
      const u32x4 w0 = *reinterpret_cast<const u32x4*>(ws + (tl << 12));
      const u32x4 w1 = *reinterpret_cast<const u32x4*>(ws + (tl << 12) + 1024);
      acc[0][0] = sg_mfma_chunk<bf16_t>(w0, x0, acc[0][0]);
      acc[0][1] = sg_mfma_chunk<bf16_t>(w1, x0, acc[0][1]);
      acc[1][0] = sg_mfma_chunk<bf16_t>(w0, x1, acc[1][0]);
      acc[1][1] = sg_mfma_chunk<bf16_t>(w1, x1, acc[1][1]);
      if (tl & 1) __builtin_amdgcn_sched_barrier(0);     // (caps how many taps' fragments the scheduler keeps in flight: registers)
    }
  };
  // one step on weight buffer `buf`: `swFree` takes the weights of the step after next, `swNext` (the next step's) goes to LDS
  auto step = [&](int s, int buf, u32x4 (&swFree)[5], u32x4 (&swNext)[5]) {
    const bool has1 = s + 1 < s1, has2 = s + 2 < s1;
    const bool newx = has1 && tg1 == 0;
    const int chunk2 = tg1 + 1 < a.ntg ? chunk1 : chunk1 + 1, tg2 = tg1 + 1 < a.ntg ? tg1 + 1 : 0;
    if (newx) load_x(sx, chunk1);
    if (has2) load_w(swFree, chunk2, tg2);
    compute(buf, xbuf + xpar * kGX + tg * planeB);
    if (has1) {
      if (newx) store_x(sx, xbuf + (xpar ^ 1) * kGX);
      store_w(swNext, wbuf + (buf ^ 1) * kGW);
    }
    __syncthreads();
    if (newx) xpar ^= 1;
    chunk = chunk1; tg = tg1; chunk1 = chunk2; tg1 = tg2;
  };
  for (int s = s0; s < s1; s += 2) {
    step(s, 0, swA, swB);
    if (s + 1 >= s1) break;
    step(s + 1, 1, swB, swA);
  }

  // output: voxel m of the flattened (n, d, h, w) volume.  (The lane's coordinates are derived again from an opaque copy
  // of the thread index: computed before the K loop they stayed live across it and were spilled around it.)
  int tid2 = tid;
  asm volatile("" : "+v"(tid2));
  const int r2 = tid2 & 31, mw2 = tid2 >> 7;
  const int64_t svox = (int64_t)a.d * hw_;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int q = (mw2 * 2 + mt) * 32 + r2;
    const int vsp = q % (hw_ * a.TD), vn = q / (hw_ * a.TD);
    const int64_t m = (int64_t)(n0 + vn) * svox + (int64_t)d0 * hw_ + vsp;
    if (n0 + vn >= a.N) continue;      // (a sample slot beyond the batch: nothing to store)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int ntg_i = nb * 4 + nw * 2 + nt;
      if (ntg_i >= a.ntile) continue;
      if (a.ksplit > 1) {
        float* dst = a.partial + ((int64_t)ks * a.N * svox + m) * a.cout + ntg_i * 32 + 4 * hh;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<f32x4*>(dst + 8 * qd) = f32x4{acc[mt][nt][4 * qd], acc[mt][nt][4 * qd + 1], acc[mt][nt][4 * qd + 2], acc[mt][nt][4 * qd + 3]};
      } else {
        gemm_tile_out(a, acc[mt][nt], m, ntg_i, hh);
      }
    }
  }
}

// K-split second stage: one thread per (voxel, 4 channels) adds the partial tiles in order; a wave-row of 8 threads holds a
// voxel's 32-channel tile, so the sign word is assembled with three xor shuffles
__global__ __launch_bounds__(256) void conv_gemm_reduce_kernel(GemmConvArgs a, int64_t nvox) {
  const int64_t total = nvox * (a.cout / 4);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < total;
  const int64_t ii = live ? i : total - 1;
  const int c4 = (int)(ii % (a.cout / 4)) * 4;
  const int64_t m = ii / (a.cout / 4);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.ksplit; ++s) v += *reinterpret_cast<const f32x4*>(a.partial + ((int64_t)s * nvox + m) * a.cout + c4);
  uint32_t bits = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = v[e] + (a.bias ? a.bias[c4 + e] : 0.f);
    if (a.act) t = sg_lrelu(t, a.slope);
    v[e] = t;
    bits |= (__float_as_uint(t) >> 31) << ((c4 & 31) + e);
  }
  if (a.sign_out) {       // 8 consecutive threads = one 32-channel tile of one voxel (cout % 32 == 0)
    bits |= (uint32_t)__shfl_xor((int)bits, 1);
    bits |= (uint32_t)__shfl_xor((int)bits, 2);
    bits |= (uint32_t)__shfl_xor((int)bits, 4);
    if (live && (c4 & 31) == 0) a.sign_out[m * a.ntile + (c4 >> 5)] = bits;
  }
  if (a.mask_bits) {
    const uint32_t mb = a.mask_bits[m * a.ntile + (c4 >> 5)] >> (c4 & 31);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= ((mb >> e) & 1u) ? a.mask_slope : 1.0f;
  }
  if (live) {
    u32x2 o;
    o[0] = sg_pack_bf16(v[0], v[1]); o[1] = sg_pack_bf16(v[2], v[3]);
    *reinterpret_cast<u32x2*>(a.y + m * a.cout + c4) = o;
  }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static bool gemm_plan(const sg_conv_shape* s, GemmConvArgs* a) {
  if (s->cin % 16 || s->cout % 64 || s->cin < 64) return false;      // (cout = 64 mod 128: half of the last block's waves idle)
  if ((int64_t)s->n * s->d * s->h * s->w * s->cin >= (1ll << 31)) return false;
  const int hw = s->h * s->w;
  if (s->w > 16 || s->h > 16 || hw > 256 || 256 % hw) return false;
  const bool k133 = s->kd == 1 && s->kh == 3 && s->kw == 3, k333 = s->kd == 3 && s->kh == 3 && s->kw == 3;
  // 1x3x3 kernels on volumes of <= 128 voxels: 1x4x4 and 2x8x8.  At 4x16x16 the 27-tap layers are bound by the L2 -> LDS rate in
  // this tiling as in the streamed kernel's, which stages through LDS-DMA and is ahead there at batch 32 (169 against 221 us for
  // 128 -> 512); with few samples (SG_GEMM_K333_MAXVOX voxels in the batch, default 8192 = batch 8) the streamed kernel has
  // too few tiles and the layer runs here, one plane per tile.
  if (k133) { if (s->d * hw > 128 || s->cin < 128) return false; }
  else if (k333) { if (hw != 256 || (int64_t)s->n * s->d * hw > sg_cfg().gemm_k333_maxvox) return false; }
  else return false;
  if (s->upsample_in && ((s->d | s->h | s->w) & 1)) return false;
  int td = 256 / hw;
  if (td > s->d) td = s->d;
  if (s->d % td) return false;
  const int tn = 256 / (td * hw);
  if (tn < 1 || tn * td * hw != 256) return false;
  a->TN = tn; a->TD = td;
  a->pd = s->kd / 2; a->ph = 1; a->pw = 1;
  a->HD = td + 2 * a->pd; a->HH = s->h + 2; a->HW = s->w + 2;
  a->hv = tn * a->HD * a->HH * a->HW;
  if (a->hv * 32 > kGX || tn * a->HD * hw * 2 > 3 * 512) return false;      // the image in LDS; interior pieces per chunk
  a->nTd = s->d / td;
  a->taps = s->kd * 9; a->kh = 3; a->kw = 3;
  a->TG = 9; a->ntg = s->kd;
  a->nchunk = s->cin / 16; a->ntile = s->cout / 32;
  a->steps = a->nchunk * a->ntg;
  const int64_t tiles = (int64_t)sg_cdiv(s->n, tn) * a->nTd * sg_cdiv(s->cout, 128);
  // K split: fill the 256 CUs; every split gets >= 1 step.  On the 27-tap layers (M = 1024 voxels per sample) a split costs a
  // pass of f32 partial tiles out and back (2 x 4 B x M x cout per split, ~3 TB/s) against the ~1 us per step it saves: stop
  // where doubling no longer pays.
  int ks = 1;
  if (k333 || sg_cfg().gemm_ks_model) {
    const double part_us = 2.0 * 4.0 * (double)s->n * s->d * hw * s->cout / 3.0e6;
    auto est = [&](int k) { return (double)a->steps / k + (k > 1 ? part_us * k : 0.0); };
    while (tiles * ks < 192 && ks * 2 <= a->steps && ks < 16 && est(2 * ks) < est(ks)) ks *= 2;
  } else {
    while (tiles * ks < 192 && ks * 2 <= a->steps && ks < 16) ks *= 2;
  }
  a->steps_per_split = sg_cdiv(a->steps, ks);
  a->ksplit = sg_cdiv(a->steps, a->steps_per_split);
  return true;
}

bool sg_gemm_conv_eligible(const sg_conv_shape* s, sg_dtype dt) {
  GemmConvArgs a;
  return dt == SG_BF16 && gemm_plan(s, &a);
}

size_t sg_gemm_conv_workspace(const sg_conv_shape* s, sg_dtype dt) {
  GemmConvArgs a;
  if (dt != SG_BF16 || !gemm_plan(s, &a) || a.ksplit <= 1) return 0;
  return (size_t)a.ksplit * s->n * s->d * s->h * s->w * (size_t)s->cout * 4;
}

// the epilogue fields of sg_conv3d_fwd that this kernel implements: bias, act, sign_out, mask_bits
int sg_gemm_conv_fwd(const void* x, const void* wp, void* y, const sg_conv_shape* s, const float* bias, int act, float slope,
                     const uint32_t* mask_bits, float mask_slope, uint32_t* sign_out, void* workspace, size_t workspace_bytes,
                     hipStream_t st, bool* used) {
  *used = false;
  GemmConvArgs a;
  if (!gemm_plan(s, &a)) return SG_OK;
  if (a.ksplit > 1 && (!workspace || workspace_bytes < sg_gemm_conv_workspace(s, SG_BF16) || !sg_aligned16(workspace))) return SG_OK;
  a.x = reinterpret_cast<const bf16_t*>(x); a.wp = reinterpret_cast<const char*>(wp); a.y = reinterpret_cast<bf16_t*>(y);
  a.partial = reinterpret_cast<float*>(workspace);
  a.bias = bias; a.mask_bits = mask_bits; a.sign_out = sign_out; a.slope = slope; a.mask_slope = mask_slope; a.act = act;
  a.N = s->n; a.d = s->d; a.h = s->h; a.w = s->w; a.cin = s->cin; a.cout = s->cout; a.ups = s->upsample_in ? 1 : 0;
  auto kern = conv_gemm_kernel;
  SG_ALLOW_160K_LDS(kern);
  const size_t lds = 2 * (size_t)kGW + 2 * (size_t)kGX;
  const unsigned tiles = (unsigned)(sg_cdiv(s->n, a.TN) * a.nTd);
  if (a.ksplit > 1) SG_KNAME("conv_gemm (K split)");      // (SG_KNAME formats once per call site)
  else SG_KNAME("conv_gemm");
  hipLaunchKernelGGL(kern, dim3(tiles, (unsigned)sg_cdiv(s->cout, 128), (unsigned)a.ksplit), dim3(512), lds, st, a);
  SG_LAUNCH_CHECK();
  if (a.ksplit > 1) {
    const int64_t nvox = (int64_t)s->n * s->d * s->h * s->w;
    const int64_t total = nvox * (s->cout / 4);
    hipLaunchKernelGGL(conv_gemm_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, nvox);
    SG_LAUNCH_CHECK();
  }
  *used = true;
  return SG_OK;
}
