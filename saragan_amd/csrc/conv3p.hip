// One-pass 3x3x3 convolution / data gradient for 64 input and 32 output channels (bf16) on gfx950: SLIDING ACCUMULATORS.
// Replaces the two-pass "K split" of conv3d.hip (two conv_fwd3s passes over 32 channels each with f32 partial sums in
// HBM) for tf.nn.conv3d as called by SURFGAN_3D/networks/ops.py:147-150 -- in the benchmarked network the data gradient
// of discriminator_block conv_2 (networks/pgan/discriminator.py:36-44), whose input is M * upscale3d(gy) / 8
// (ops.py:292-305), gathered from the pooled gradient while the halo is staged.
//
// Why another formulation: with 64 input channels the resident weights are 27 x 4 KiB = 108 KiB of the CU's 160 KiB, and
// the sliding-halo kernel's ring (4 halo planes per wave group) needs 2 x 104 KiB.  Here the roles are turned round: a
// wave group keeps ONE input plane (6 x 34 halo voxels x 128 B = 25.5 KiB) in LDS and THREE output planes in registers.
// Input plane p contributes through tap plane kd to output plane p + 1 - kd; after the MFMAs of plane p, output plane
// p - 1 is complete, is stored, and its accumulator starts plane p + 2.  Per plane and wave: 36 steps (9 in-plane taps x 4
// channel chunks) of 1 activation + 3 weight fragment reads and 3 MFMAs -- the MFMA count of one sliding-halo phase, 4 / 3
// LDS reads per MFMA.  No partial sums, every input row fetched once as a whole 128-byte line.
//
// LDS map (163 200 B): [plane buffer of group 0][plane buffer of group 1][weights: [tap][chunk] 1-KiB fragments][bias].
// A plane buffer holds four sub-images (one per 32-byte channel chunk, 6560 B apart: 204 rows x 32 B + 32 B so that the
// four sub-images start 8 banks apart and a row's eight 16-byte pieces are stored without a bank conflict); inside a
// sub-image slot s of row r lives at s ^ ((r >> 3) & 1): the fragment read of 32 consecutive rows is conflict free for
// every tap shift, and the chunk is an immediate offset of the ds_read (9 fragment addresses per lane instead of 36).
#include "common.h"
#include "prof.h"
#include "conv_args.h"

typedef __attribute__((address_space(3))) void* lds_ptr3p_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr3p_t;
__device__ __attribute__((aligned(1024))) uint32_t sg_zero_page_p[256] = {0};

namespace {

constexpr int P3_SUBB = 6560;                 // sub-image stride
constexpr int P3_PBS = 4 * P3_SUBB;           // plane buffer of one wave group
constexpr int P3_SUB16 = 2 * P3_SUBB;         // 16x16x32 variant: sub-image of one 32-channel chunk (204 rows x 64 B + 64 B: 16 banks apart)
constexpr int P3_WOFF = 2 * P3_PBS;           // resident weights
constexpr int P3_WBYTES = 27 * 4 * 1024;
constexpr int P3_LDS = P3_WOFF + P3_WBYTES + 128;
static_assert(P3_LDS <= 160 * 1024, "LDS budget");

// K loop of one input plane.  Step s = (kh, kw) * 4 + chunk: one activation fragment and the weight fragments of the three
// tap planes kd; accumulator j receives kd = (ROT + 1 - j) mod 3, ROT = running phase index mod 3 (the three code variants
// follow each other in straight-line code, so the accumulators never move between registers).
template <int ROT, int PF>
struct sg_unrolled_kp {
  static constexpr int NS = 36, RING = PF + 1;
  static_assert(PF * 4 <= 15, "lgkmcnt is a 4-bit counter");
  template <int FRAG>
  static __device__ __forceinline__ void wload(u32x4& w, int wl_lo, int wl_hi) {   // ds_read offsets are 16 bits: two bases
    if constexpr (FRAG < 64) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(wl_lo), "n"(FRAG << 10));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(wl_hi), "n"((FRAG - 64) << 10));
  }
  template <int ST>
  static __device__ __forceinline__ void load(u32x4 (&xfr)[RING], u32x4 (&wfr)[RING][3], const int (&xa)[9], int wl_lo, int wl_hi) {
    constexpr int SL = ST % RING, khw = ST >> 2, gi = ST & 3;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xfr[SL]) : "v"(xa[khw]), "n"(gi * P3_SUBB));
    wload<(0 * 9 + khw) * 4 + gi>(wfr[SL][0], wl_lo, wl_hi);     // image order [tap][chunk]
    wload<(1 * 9 + khw) * 4 + gi>(wfr[SL][1], wl_lo, wl_hi);
    wload<(2 * 9 + khw) * 4 + gi>(wfr[SL][2], wl_lo, wl_hi);
  }
  template <int ST>
  static __device__ __forceinline__ void step(f32x16 (&acc)[3], u32x4 (&xfr)[RING], u32x4 (&wfr)[RING][3], const int (&xa)[9],
                                              int wl_lo, int wl_hi) {
    if constexpr (ST < NS) {
      if constexpr (ST + PF < NS) load<ST + PF>(xfr, wfr, xa, wl_lo, wl_hi);
      constexpr int younger = (NS - 1 - ST < PF ? NS - 1 - ST : PF) * 4;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
      __builtin_amdgcn_sched_barrier(0);
      sg_mfma_bf16_acc(acc[(ROT + 1) % 3], wfr[ST % RING][0], xfr[ST % RING]);
      sg_mfma_bf16_acc(acc[ROT % 3], wfr[ST % RING][1], xfr[ST % RING]);
      sg_mfma_bf16_acc(acc[(ROT + 2) % 3], wfr[ST % RING][2], xfr[ST % RING]);
      __builtin_amdgcn_sched_barrier(0);
      step<ST + 1>(acc, xfr, wfr, xa, wl_lo, wl_hi);
    }
  }
  template <int ST>
  static __device__ __forceinline__ void prologue(u32x4 (&xfr)[RING], u32x4 (&wfr)[RING][3], const int (&xa)[9], int wl_lo, int wl_hi) {
    if constexpr (ST < PF && ST < NS) {
      load<ST>(xfr, wfr, xa, wl_lo, wl_hi);
      prologue<ST + 1>(xfr, wfr, xa, wl_lo, wl_hi);
    }
  }
  static __device__ __forceinline__ void run(f32x16 (&acc)[3], const int (&xa)[9], int wl_lo, int wl_hi) {
    u32x4 xfr[RING], wfr[RING][3];
    SG_KLOOP_BEGIN();
    prologue<0>(xfr, wfr, xa, wl_lo, wl_hi);
    step<0>(acc, xfr, wfr, xa, wl_lo, wl_hi);
    // 18 wait states after the last in-place MFMA before anything reads the accumulators (see sg_mfma_drain)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
    SG_KLOOP_END();
  }
};

#ifndef SG_V3P_PF
#define SG_V3P_PF 2     // fragment reads run this many steps (of 3 MFMAs) ahead of their use
#endif

// The same K loop on v_mfma_f32_16x16x32_bf16 (A = weights 16 cout x 32 cin, B = activations 32 cin x 16 voxels): the bare loop of
// this shape sustains 1.086x the FLOP/s of 32x32x16 at equal cycles per FLOP on this board (profiles/r04_mfma_ceiling.txt: the chip
// holds a higher clock).  A wave's 32 voxels x 32 channels are 2 x 2 tiles; step s = (kh, kw) * 2 + chunk (32 channels): two
// activation fragments (voxel halves) and six weight fragments (three tap planes kd x two channel halves), twelve MFMAs of 16
// cycles: the same LDS bytes per FLOP as the 32x32x16 loop.  Fragment image order in LDS: [tap][chunk][channel half].
template <int ROT>
struct sg_unrolled_kp16 {
  // half-steps hs = ((kh, kw) * 2 + chunk) * 2 + ch: the activation fragments of a step arrive with its first half, the weight
  // fragments per channel half (a ring of three: reads run two half-steps = 12 MFMAs ahead of their use)
  static constexpr int NH = 36, PFH = 2, RW = 3;
  template <int FRAG>
  static __device__ __forceinline__ void wload(u32x4& w, int wl_lo, int wl_hi) {
    if constexpr (FRAG < 64) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(wl_lo), "n"(FRAG << 10));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(wl_hi), "n"((FRAG - 64) << 10));
  }
  template <int HS>
  static __device__ __forceinline__ void load(u32x4 (&xfr)[2][2], u32x4 (&wfr)[RW][3], const int (&xa)[9], int wl_lo, int wl_hi) {
    constexpr int st = HS >> 1, ch = HS & 1, khw = st >> 1, gi = st & 1, SW = HS % RW;
    if constexpr (ch == 0) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xfr[st & 1][0]) : "v"(xa[khw]), "n"(gi * P3_SUB16));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xfr[st & 1][1]) : "v"(xa[khw]), "n"(gi * P3_SUB16 + 1024));
    }
    wload<((0 * 9 + khw) * 2 + gi) * 2 + ch>(wfr[SW][0], wl_lo, wl_hi);
    wload<((1 * 9 + khw) * 2 + gi) * 2 + ch>(wfr[SW][1], wl_lo, wl_hi);
    wload<((2 * 9 + khw) * 2 + gi) * 2 + ch>(wfr[SW][2], wl_lo, wl_hi);
  }
  static constexpr int nloads(int hs) { return hs >= NH ? 0 : ((hs & 1) ? 3 : 5); }
  static constexpr int younger(int hs) {
    int n = 0;
    for (int t = hs + 1; t <= hs + PFH; ++t) n += nloads(t);
    return n;
  }
  static __device__ __forceinline__ void mfma(f32x4& c, u32x4 a_, u32x4 b_) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a_), "v"(b_));
  }
  template <int HS>
  static __device__ __forceinline__ void step(f32x4 (&acc)[3][2][2], u32x4 (&xfr)[2][2], u32x4 (&wfr)[RW][3], const int (&xa)[9],
                                              int wl_lo, int wl_hi) {
    if constexpr (HS < NH) {
      if constexpr (HS + PFH < NH) load<HS + PFH>(xfr, wfr, xa, wl_lo, wl_hi);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger(HS)));
      __builtin_amdgcn_sched_barrier(0);
      constexpr int st = HS >> 1, ch = HS & 1, SW = HS % RW;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int j = (ROT + 1 - kd + 3) % 3;      // accumulator (output plane) that tap plane kd feeds
        mfma(acc[j][0][ch], wfr[SW][kd], xfr[st & 1][0]);
        mfma(acc[j][1][ch], wfr[SW][kd], xfr[st & 1][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      step<HS + 1>(acc, xfr, wfr, xa, wl_lo, wl_hi);
    }
  }
  static __device__ __forceinline__ void run(f32x4 (&acc)[3][2][2], const int (&xa)[9], int wl_lo, int wl_hi) {
    u32x4 xfr[2][2], wfr[RW][3];
    SG_KLOOP_BEGIN();
    load<0>(xfr, wfr, xa, wl_lo, wl_hi);
    load<1>(xfr, wfr, xa, wl_lo, wl_hi);
    step<0>(acc, xfr, wfr, xa, wl_lo, wl_hi);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0][0][0]), "+v"(acc[0][0][1]), "+v"(acc[0][1][0]), "+v"(acc[0][1][1]),
                 "+v"(acc[1][0][0]), "+v"(acc[1][0][1]), "+v"(acc[1][1][0]), "+v"(acc[1][1][1]),
                 "+v"(acc[2][0][0]), "+v"(acc[2][0][1]), "+v"(acc[2][1][0]), "+v"(acc[2][1][1]));
    SG_KLOOP_END();
  }
};

template <int EPI, bool UPS, bool INM>
__global__ __launch_bounds__(512) void conv_fwd3p_kernel(ConvFwdArgs a) {
  static_assert(!INM || UPS, "the input mask rides on the fused gather");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TAPS = 27, ES = 2, CIN = 64, COUT = 32;
  constexpr int ROWS = 204;                          // halo voxels of a plane: 6 x 34
  constexpr int MAXP = 7;                            // 16-byte pieces per lane and plane: 204 x 8 / 256 -> 6.4
  constexpr uint32_t DEAD = 0x80000000u;             // byte offset beyond every buffer: loads return 0, stores drop
  constexpr bool SIGN = (EPI & SG_EP_SIGN) != 0, MASK = (EPI & SG_EP_MASK) != 0, PN = (EPI & SG_EP_PN) != 0;
  static_assert((EPI & ~(SG_EP_SIGN | SG_EP_MASK | SG_EP_PN)) == 0 && !(MASK && (SIGN || PN)), "unsupported epilogue combination");
  const sg_tile_geom& g = a.g;                       // TH = 4, TW = 32 (host-checked); the kernel walks whole columns along D
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int r = lane & 31, hh = lane >> 5;
  char* wlds = smem + P3_WOFF;
  const char* wp = reinterpret_cast<const char*>(a.wp);
  const int H = g.H, W = g.W, D = g.D;
  // Everything global goes through buffer resources rebased per batch sample (one SAMPLE of a tensor stays below 2 GiB,
  // host-checked), a scalar per-plane offset and a 32-bit per-lane offset computed once per column; dead lanes carry DEAD.
  const int64_t svox = (int64_t)D * H * W;
  auto rsrc_of = [&](const void* base, int64_t sample_bytes, int n0) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + n0 * sample_bytes, 0,
                                             (int)sample_bytes, 0x00020000);
  };
  const int64_t xsb = (UPS ? svox >> 3 : svox) * CIN * ES, ysb = svox * COUT * ES, wsb = svox * 4, psb = svox * 4;

  // column schedule (as conv_fwd3s): a block walks PAIRS of H-adjacent tile columns along D, wave group g taking the column
  // with tile row 2*k + g one phase apart, so that the halo rows the pair shares are fetched twice within microseconds on one
  // CU (the second fetch an L2 hit); XCD group xg owns a contiguous chunk of the pair list.
  const int nTh2 = (g.nTh + 1) >> 1;
  const int npair = g.nTn * nTh2 * g.nTw;
  const int xg = blockIdx.x & 7, bslot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (npair + 7) >> 3;
  const int c_begin = xg * cpx, c_end = min(npair, c_begin + cpx);
  const int cfirst = c_begin + bslot;
  const int ncols_blk = cfirst < c_end ? (c_end - cfirst + per_x - 1) / per_x : 0;
  const int items_mine = ncols_blk * D;              // phases of this wave group: one per input plane

  // fragment addresses inside the plane buffer: voxel (h = wave + kh, w = r + kw), half hh; the chunk is an immediate offset
  int xa[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int row = (wave + kh) * 34 + r + kw;
      xa[kh * 3 + kw] = grp * P3_PBS + row * 32 + ((hh ^ ((row >> 3) & 1)) << 4);
    }
  // staging table: this lane's 16-byte pieces (piece p = 8 channels) of halo rows (wave + 4k) * 8 + lane / 8.  The halo goes
  // global -> registers -> LDS (the masked gather has arithmetic to do on the way).
  const int pc = lane & 7;
  uint32_t relb[MAXP];   // byte offset of my piece relative to the plane's first halo voxel (h0 - 1, w0 - 1); dead: huge
  uint32_t relm[INM ? MAXP : 1];   // INM: byte offset of my piece's 8 sign bits relative to that voxel's first sign word
  int crdp[MAXP];        // packed (hw, hh) for the boundary test; dead pieces fail every range
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int row = (wave + 4 * k) * 8 + (lane >> 3);
    const int hh_ = row / 34, hw = row - hh_ * 34;
    const bool live = row < ROWS;
    // UPS: tile origins are even, so a halo voxel's halved coordinate is a per-lane constant relative to the tile's
    // half-resolution origin: ((h0 - 1 + hh) >> 1) = h0 / 2 + ((hh - 1) >> 1), likewise along W
    const int rel = UPS ? ((((hh_ - 1) >> 1) * (W >> 1) + ((hw - 1) >> 1)) * CIN + pc * 8) * ES : ((hh_ * W + hw) * CIN + pc * 8) * ES;
    relb[k] = live ? (uint32_t)rel : 0xC0000000u;    // dead: stays >= DEAD after + tile offset
    crdp[k] = live ? (hw | (hh_ << 8)) : 0x7F7F;
    if constexpr (INM) relm[k] = live ? (uint32_t)((hh_ * W + hw) * 8 + pc) : 0xC0000000u;   // 64 channels = 8 sign bytes per fine voxel
  }
  // LDS position of my piece k: sub-image pc / 2, row, slot (pc & 1) ^ ((row >> 3) & 1) -- row >> 3 = wave + 4k, so the
  // swizzle bit is the wave's parity and piece k sits k KiB after piece 0
  const int wofs = grp * P3_PBS + (pc >> 1) * P3_SUBB + (wave * 8 + (lane >> 3)) * 32 + (((pc & 1) ^ (wave & 1)) << 4);
  const bool live6 = (wave + 24) * 8 + (lane >> 3) < ROWS;
  // my output voxel (h = wave, w = r of the tile): byte / word offsets relative to the tile origin of a plane
  const uint32_t yvo = (uint32_t)((wave * W + r) * COUT * ES), svo = (uint32_t)(wave * W + r) * 4u;
  const uint32_t plane_bytes = (uint32_t)((UPS ? (H >> 1) * (W >> 1) : H * W) * CIN * ES);
  const int plane_vox = H * W;

  struct Cur { int cj, di, n0, h0, w0; };
  auto enter_column = [&](Cur& c) {
    const int pr = cfirst + c.cj * per_x;
    const int c1 = (int)sg_div((uint32_t)pr, g.fnTw);
    c.w0 = (pr - c1 * g.nTw) * 32;
    const int c2 = c1 / nTh2;
    c.h0 = (2 * (c1 - c2 * nTh2) + grp) * 4;         // may lie beyond H for the last odd row: a dead column
    c.n0 = c2;
  };
  // ---- halo side (cursor P: the plane requested last)
  Cur P{0, 0, 0, 0, 0};
  int qP = 0;
  __amdgpu_buffer_rsrc_t rxP, rmP;
  uint32_t vk[MAXP], vkm[INM ? MAXP : 1];
  const int64_t msb = svox * 8;                      // (INM) sign words of one sample of the fine input: two per voxel
  auto enter_column_P = [&]() {
    enter_column(P);
    rxP = rsrc_of(a.x, xsb, P.n0);
    if constexpr (INM) rmP = rsrc_of(a.in_mask, msb, P.n0);
    const int tile_off = UPS ? ((P.h0 >> 1) * (W >> 1) + (P.w0 >> 1)) * CIN * ES
                             : ((P.h0 - 1) * W + (P.w0 - 1)) * CIN * ES;   // may be negative: only dead lanes go below 0
    const int lo_w = max(0, 1 - P.w0), hi_w = min(34, W + 1 - P.w0) - 1;
    const int lo_h = max(0, 1 - P.h0), hi_h = min(6, H + 1 - P.h0) - 1;    // hi_h < 0 for a dead column
    const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8));
    const uint32_t hi = (uint32_t)(hi_w | ((hi_h & 0x7F) << 8)) | 0x8080u;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const uint32_t c_ = (uint32_t)crdp[k];
      const uint32_t t1 = (c_ | 0x8080u) - lo, t2 = hi - c_;
      const bool in = (t1 & t2 & 0x8080u) == 0x8080u && hi_h >= 0;
      vk[k] = in ? relb[k] + (uint32_t)tile_off : DEAD;
      if constexpr (INM) vkm[k] = in ? relm[k] + (uint32_t)(((P.h0 - 1) * W + (P.w0 - 1)) * 8) : DEAD;
    }
  };
  u32x4 stg[MAXP];
  uint32_t mstg[INM ? MAXP : 1];                     // (INM) the 8 sign bits of each piece in flight
  auto load_plane = [&](int gp) __attribute__((always_inline)) {   // input plane gp of P's column
    const uint32_t soff = (uint32_t)(UPS ? gp >> 1 : gp) * plane_bytes;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) stg[k] = __builtin_amdgcn_raw_buffer_load_b128(rxP, vk[k], soff, 0);
    if constexpr (INM) {
      const uint32_t moff = (uint32_t)gp * (uint32_t)(plane_vox * 8);
#pragma unroll
      for (int k = 0; k < MAXP; ++k) mstg[k] = __builtin_amdgcn_raw_buffer_load_b8(rmP, vkm[k], moff, 0);
    }
  };
  const float gain_in = a.in_gain, slope_in = a.in_mask_slope;   // (INM) sg_mask_piece_bf16: the arithmetic of sg_upscale2x_masked
  auto store_plane = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      u32x4 v = stg[k];
      if constexpr (INM) v = sg_mask_piece_bf16(v, mstg[k], gain_in, slope_in);
      if (k < 6) *reinterpret_cast<u32x4*>(smem + wofs + k * 1024) = v;
      else if (live6) *reinterpret_cast<u32x4*>(smem + wofs + k * 1024) = v;
    }
  };
  auto advance_P = [&]() {                           // the plane after the one requested last
    ++qP;
    if (++P.di == D) {
      P.di = 0;
      ++P.cj;
      if (qP < items_mine) enter_column_P();
    }
  };
  // ---- output side (cursor E: the input plane whose MFMAs ran last)
  Cur E{0, 0, 0, 0, 0};
  int qE = 0;
  __amdgpu_buffer_rsrc_t ryE, rsE, rmE, rpE;
  int colvoxE = 0;
  bool row_okE = false;
  auto enter_column_E = [&]() {
    enter_column(E);
    row_okE = E.h0 + wave < H;
    colvoxE = E.h0 * W + E.w0;
    ryE = rsrc_of(a.y, ysb, E.n0);
    if (SIGN) rsE = rsrc_of(a.sign_out, wsb, E.n0);
    if (MASK) rmE = rsrc_of(a.mask_bits, wsb, E.n0);
    if (PN) rpE = rsrc_of(a.pn_scale, psb, E.n0);
  };
  // LeakyReLU sign words of the output planes stored in the NEXT off-phase (masked epilogue): plane E.di - 1 and, at the top
  // of a column, plane D - 1 as well.  Requested one phase ahead like the halo plane; issued and consumed unconditionally
  // (DEAD offset: no memory access) so that no s_waitcnt vmcnt(0) ends up in front of an MFMA.
  uint32_t mbn[2] = {0u, 0u};
  auto request_mask = [&]() {
    if constexpr (MASK) {
      const bool live = qE < items_mine && row_okE;
      const uint32_t tv0 = (uint32_t)((E.di - 1) * plane_vox + colvoxE), tv1 = (uint32_t)(E.di * plane_vox + colvoxE);
      mbn[0] = __builtin_amdgcn_raw_buffer_load_b32(rmE, (live && E.di >= 1) ? svo : DEAD, tv0 * 4u, 0);
      mbn[1] = __builtin_amdgcn_raw_buffer_load_b32(rmE, (live && E.di == D - 1) ? svo : DEAD, tv1 * 4u, 0);
    }
  };

  // resident weights (all 8 waves) and bias
  for (int f = wave8; f < TAPS * 4; f += 8) {
    const int tap = f >> 2, gi = f & 3;
    __builtin_amdgcn_global_load_lds((gbl_ptr3p_t)(wp + ((int64_t)(gi * TAPS + tap) << 10) + lane * 16),
                                     (lds_ptr3p_t)(wlds + ((size_t)f << 10)), 16, 0, 0);
  }
  float* bias_lds = reinterpret_cast<float*>(wlds + P3_WBYTES);
  if (tid < 32) bias_lds[tid] = a.bias != nullptr ? a.bias[tid] : 0.f;

  const bool no_stage = (a.dbg_flags & 1) != 0, no_epi = (a.dbg_flags & 2) != 0;
  if ((a.dbg_flags & 2048) && grp == 1) __builtin_amdgcn_s_setprio(1);      // diagnostic: static priority for the younger wave group
  if (items_mine > 0) {
    enter_column_P();
    enter_column_E();
    load_plane(0);
    if (grp == 0) {      // the very first plane of group 0: latency exposed once
      store_plane();
      if (items_mine > 1) {
        advance_P();
        load_plane(P.di);
      }
    }
  }
  __syncthreads();

  const int wl_lo = P3_WOFF + lane * 16, wl_hi = wl_lo + 64 * 1024;
  const float inv_c = 1.f / (float)COUT;
  const float slope = a.act ? a.slope : 1.f;         // max(x, 1 * x) = x: no branch for "no activation"

  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && lane == 0 && wave == 0 && dbgi < 120)
      a.dbg[grp * 128 + dbgi] = __builtin_amdgcn_s_memtime();
    ++dbgi;
  };
  f32x16 acc[3];
  auto init_acc = [&](f32x16& c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = bias_lds[(i & 3) + 8 * (i >> 2) + 4 * hh];   // bias rides in C
  };
  // one finished output plane: bias is in, activation / pixel-norm / sign words / mask, 2 x 16 bytes per lane
  auto epilogue = [&](f32x16& c, int o, bool ok, uint32_t mb) __attribute__((always_inline)) {
    const uint32_t tile_vox = (uint32_t)(o * plane_vox + colvoxE);   // within sample E.n0
    if (slope != 1.f) {   // uniform
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = sg_lrelu(c[i], slope);
    }
    if (PN) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) ss += c[i] * c[i];
      const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
      ss = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);   // own half + partner lane ^ 32's
      const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] *= sc;
      if (a.pn_scale != nullptr)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc), rpE, (ok && hh == 0) ? svo : DEAD, tile_vox * 4u, 0);
    }
    if (SIGN) {
      uint32_t b = 0u;
#pragma unroll
      for (int i = 0; i < 16; ++i) b |= (__float_as_uint(c[i]) >> 31) << ((i & 3) + 8 * (i >> 2));
      b <<= 4 * hh;
      const auto sw2 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
      __builtin_amdgcn_raw_buffer_store_b32(sw2[0] | sw2[1], rsE, (ok && hh == 0) ? svo : DEAD, tile_vox * 4u, 0);
    }
    if (MASK) sg_apply_sign_word(c, mb, hh, a.mask_slope);
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // 16 contiguous bytes per lane (see sg_store_tile_row_bf16)
      const uint32_t a0 = sg_pack_bf16(c[8 * j + 0], c[8 * j + 1]), a1 = sg_pack_bf16(c[8 * j + 2], c[8 * j + 3]);
      const uint32_t b0 = sg_pack_bf16(c[8 * j + 4], c[8 * j + 5]), b1 = sg_pack_bf16(c[8 * j + 6], c[8 * j + 7]);
      const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      u32x4 out;
      out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
      __builtin_amdgcn_raw_buffer_store_b128(out, ryE, ok ? yvo + (uint32_t)((16 * j + 8 * hh) * 2) : DEAD,
                                             tile_vox * (uint32_t)(COUT * ES), 0);
      SG_STORE16_GUARD(out);
    }
  };
  // ---- off-phase after the MFMAs of E's plane p: the plane requested one phase ago goes to LDS; output plane p - 1 (in cA)
  // is complete and stored -- at the top of a column plane D - 1 (in cB) as well, and all three accumulators start afresh;
  // then E advances, the next sign words and the plane after P's are requested.
  auto off_phase = [&](f32x16& cA, f32x16& cB, f32x16& cC, bool closes, bool stage) __attribute__((always_inline)) {
    uint32_t mb[2] = {mbn[0], mbn[1]};
    if (MASK) asm volatile("" : "+v"(mb[0]), "+v"(mb[1]));
    if (stage && !no_stage) store_plane();
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    if (closes) {
      const int p = E.di;
      const bool top = p == D - 1;
      if (!no_epi) {
        if (p >= 1) epilogue(cA, p - 1, row_okE, mb[0]);
        if (top) epilogue(cB, p, row_okE, mb[1]);
      }
      init_acc(cA);
      if (top) {
        init_acc(cB);
        init_acc(cC);
      }
      ++qE;
      if (++E.di == D) {
        E.di = 0;
        ++E.cj;
        if (E.cj < ncols_blk) enter_column_E();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (a.dbg_flags & 128) stamp();
    request_mask();
    if (stage) {
      advance_P();
      if (qP < items_mine && !no_stage) load_plane(P.di);
    }
    if (a.dbg_flags & 128) stamp();
  };
  // accumulator roles at running phase t (ROT = t % 3): plane p - 1 in acc[(ROT + 2) % 3], p in acc[ROT], p + 1 in acc[(ROT + 1) % 3]
#define SG_P3_OFF(ROT, closes, stage) off_phase(acc[(ROT + 2) % 3], acc[ROT], acc[(ROT + 1) % 3], closes, stage)
  // Each wave group runs its own straight loop (one phase apart, paced by the block barrier), three planes per trip.
  // stage(t): plane t + 1 exists (it was requested); the request inside is for plane t + 2 (advance_P guards the end).
  if (grp == 0) {
    init_acc(acc[0]); init_acc(acc[1]); init_acc(acc[2]);
    request_mask();
    for (int q = 0; q < items_mine; q += 3) {
      stamp();
      sg_unrolled_kp<0, SG_V3P_PF>::run(acc, xa, wl_lo, wl_hi);
      stamp();
      __syncthreads();
      stamp();
      SG_P3_OFF(0, true, q + 1 < items_mine);
      stamp();
      __syncthreads();
      if (q + 1 < items_mine) sg_unrolled_kp<1, SG_V3P_PF>::run(acc, xa, wl_lo, wl_hi);
      __syncthreads();
      if (q + 1 < items_mine) SG_P3_OFF(1, true, q + 2 < items_mine);
      __syncthreads();
      if (q + 2 < items_mine) sg_unrolled_kp<2, SG_V3P_PF>::run(acc, xa, wl_lo, wl_hi);
      __syncthreads();
      if (q + 2 < items_mine) SG_P3_OFF(2, true, q + 3 < items_mine);
      __syncthreads();
    }
  } else {
    init_acc(acc[0]); init_acc(acc[1]); init_acc(acc[2]);
    if (items_mine > 0) SG_P3_OFF(2, false, true);   // writes plane 0, requests plane 1 (no accumulator is touched)
    for (int q = 0; q < items_mine; q += 3) {
      __syncthreads();
      stamp();
      sg_unrolled_kp<0, SG_V3P_PF>::run(acc, xa, wl_lo, wl_hi);
      stamp();
      __syncthreads();
      stamp();
      SG_P3_OFF(0, true, q + 1 < items_mine);
      stamp();
      __syncthreads();
      if (q + 1 < items_mine) sg_unrolled_kp<1, SG_V3P_PF>::run(acc, xa, wl_lo, wl_hi);
      __syncthreads();
      if (q + 1 < items_mine) SG_P3_OFF(1, true, q + 2 < items_mine);
      __syncthreads();
      if (q + 2 < items_mine) sg_unrolled_kp<2, SG_V3P_PF>::run(acc, xa, wl_lo, wl_hi);
      __syncthreads();
      if (q + 2 < items_mine) SG_P3_OFF(2, true, q + 3 < items_mine);
    }
  }
#undef SG_P3_OFF
}

template <int EPI, bool UPS, bool INM>
__global__ __launch_bounds__(512) void conv_fwd3p16_kernel(ConvFwdArgs a) {
  static_assert(!INM || UPS, "the input mask rides on the fused gather");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TAPS = 27, ES = 2, CIN = 64, COUT = 32;
  constexpr int ROWS = 204;                          // halo voxels of a plane: 6 x 34
  constexpr int MAXP = 7;                            // 16-byte pieces per lane and plane: 204 x 8 / 256 -> 6.4
  constexpr uint32_t DEAD = 0x80000000u;             // byte offset beyond every buffer: loads return 0, stores drop
  constexpr bool SIGN = (EPI & SG_EP_SIGN) != 0, MASK = (EPI & SG_EP_MASK) != 0, PN = (EPI & SG_EP_PN) != 0;
  static_assert((EPI & ~(SG_EP_SIGN | SG_EP_MASK | SG_EP_PN)) == 0 && !(MASK && (SIGN || PN)), "unsupported epilogue combination");
  const sg_tile_geom& g = a.g;                       // TH = 4, TW = 32 (host-checked); the kernel walks whole columns along D
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int v16 = lane & 15, q4 = lane >> 4;      // MFMA 16x16x32: voxel column / K group (inputs), row group (outputs)
  char* wlds = smem + P3_WOFF;
  const char* wp = reinterpret_cast<const char*>(a.wp) + P3_WBYTES;      // the 16x16x32 fragment image follows the 32x32x16 one
  const int H = g.H, W = g.W, D = g.D;
  // Everything global goes through buffer resources rebased per batch sample (one SAMPLE of a tensor stays below 2 GiB,
  // host-checked), a scalar per-plane offset and a 32-bit per-lane offset computed once per column; dead lanes carry DEAD.
  const int64_t svox = (int64_t)D * H * W;
  auto rsrc_of = [&](const void* base, int64_t sample_bytes, int n0) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + n0 * sample_bytes, 0,
                                             (int)sample_bytes, 0x00020000);
  };
  const int64_t xsb = (UPS ? svox >> 3 : svox) * CIN * ES, ysb = svox * COUT * ES, wsb = svox * 4, psb = svox * 4;

  // column schedule (as conv_fwd3s): a block walks PAIRS of H-adjacent tile columns along D, wave group g taking the column
  // with tile row 2*k + g one phase apart, so that the halo rows the pair shares are fetched twice within microseconds on one
  // CU (the second fetch an L2 hit); XCD group xg owns a contiguous chunk of the pair list.
  const int nTh2 = (g.nTh + 1) >> 1;
  const int npair = g.nTn * nTh2 * g.nTw;
  const int xg = blockIdx.x & 7, bslot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int cpx = (npair + 7) >> 3;
  const int c_begin = xg * cpx, c_end = min(npair, c_begin + cpx);
  const int cfirst = c_begin + bslot;
  const int ncols_blk = cfirst < c_end ? (c_end - cfirst + per_x - 1) / per_x : 0;
  const int items_mine = ncols_blk * D;              // phases of this wave group: one per input plane

  // fragment addresses inside the plane buffer (B operand of v_mfma_f32_16x16x32_bf16: lane = voxel column l & 15, K group l >> 4):
  // voxel (h = wave + kh, w = v16 + kw [+ 16 for the second voxel half]), 16-byte slot q4 of the 64-byte chunk row, stored at
  // slot ^ (((row >> 2) & 1) << 1) -- conflict-free for every tap shift (DESIGN_NOTES section 8).  Second voxel half: + 16 rows
  // = + 1024 B (same swizzle bit); second 32-channel chunk: + P3_SUB16: both immediate offsets.
  int xa[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int row = (wave + kh) * 34 + v16 + kw;
      xa[kh * 3 + kw] = grp * P3_PBS + row * 64 + ((q4 ^ (((row >> 2) & 1) << 1)) << 4);
    }
  // staging table: this lane's 16-byte pieces (piece p = 8 channels) of halo rows (wave + 4k) * 8 + lane / 8.  The halo goes
  // global -> registers -> LDS (the masked gather has arithmetic to do on the way).
  const int pc = lane & 7;
  int crdp[MAXP];        // packed (hw, hh) of my piece's halo voxel for the boundary test; dead pieces fail every range
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int row = (wave + 4 * k) * 8 + (lane >> 3);
    const int hh_ = row / 34, hw = row - hh_ * 34;
    crdp[k] = row < ROWS ? (hw | (hh_ << 8)) : 0x7F7F;
  }
  // LDS position of my piece k: sub-image pc / 4 (32-channel chunk), row, slot (pc & 3) ^ (((row >> 2) & 1) << 1); row >> 2 =
  // 2 * (wave + 4k) + (lane >> 5), so the swizzle bit is lane bit 5 and piece k (32 rows further) sits 2 KiB after piece k - 1
  const int wofs = grp * P3_PBS + (pc >> 2) * P3_SUB16 + (wave * 8 + (lane >> 3)) * 64 + (((pc & 3) ^ (((lane >> 5) & 1) << 1)) << 4);
  const bool live6 = (wave + 24) * 8 + (lane >> 3) < ROWS;
  // my two output voxels (h = wave, w = v16 + 16 * vh of the tile): byte / word offsets relative to the tile origin of a plane;
  // after the half-wave exchanges of the store path lane q4 holds channels [0, 16, 8, 24][q4] .. + 7 of its voxel
  // (voxel half 1 = voxel half 0 + 16 voxels: + 1024 B of output, + 64 B of sign words)
  const uint32_t yvo0 = (uint32_t)((wave * W + v16) * COUT * ES + ((q4 & 1) * 16 + (q4 >> 1) * 8) * ES);
  const uint32_t svo0 = (uint32_t)(wave * W + v16) * 4u;
  const uint32_t plane_bytes = (uint32_t)((UPS ? (H >> 1) * (W >> 1) : H * W) * CIN * ES);
  const int plane_vox = H * W;

  struct Cur { int cj, di, n0, h0, w0; };
  auto enter_column = [&](Cur& c) {
    const int pr = cfirst + c.cj * per_x;
    const int c1 = (int)sg_div((uint32_t)pr, g.fnTw);
    c.w0 = (pr - c1 * g.nTw) * 32;
    const int c2 = c1 / nTh2;
    c.h0 = (2 * (c1 - c2 * nTh2) + grp) * 4;         // may lie beyond H for the last odd row: a dead column
    c.n0 = c2;
  };
  // ---- halo side (cursor P: the plane requested last)
  Cur P{0, 0, 0, 0, 0};
  int qP = 0;
  __amdgpu_buffer_rsrc_t rxP, rmP;
  uint32_t vk[MAXP], vkm[INM ? MAXP : 1];
  const int64_t msb = svox * 8;                      // (INM) sign words of one sample of the fine input: two per voxel
  auto enter_column_P = [&]() {
    enter_column(P);
    rxP = rsrc_of(a.x, xsb, P.n0);
    if constexpr (INM) rmP = rsrc_of(a.in_mask, msb, P.n0);
    const int tile_off = UPS ? ((P.h0 >> 1) * (W >> 1) + (P.w0 >> 1)) * CIN * ES
                             : ((P.h0 - 1) * W + (P.w0 - 1)) * CIN * ES;   // may be negative: only dead lanes go below 0
    const int lo_w = max(0, 1 - P.w0), hi_w = min(34, W + 1 - P.w0) - 1;
    const int lo_h = max(0, 1 - P.h0), hi_h = min(6, H + 1 - P.h0) - 1;    // hi_h < 0 for a dead column
    const uint32_t lo = (uint32_t)(lo_w | (lo_h << 8));
    const uint32_t hi = (uint32_t)(hi_w | ((hi_h & 0x7F) << 8)) | 0x8080u;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      // the piece's byte offsets are rebuilt from its packed coordinates at every column entry (once per D phases) instead of
      // living in 14 more VGPRs; the empty asm keeps the compiler from hoisting them back out of the phase loop
      uint32_t c_ = (uint32_t)crdp[k];
      asm volatile("" : "+v"(c_));
      const int hw = (int)(c_ & 0xFFu), hh_ = (int)(c_ >> 8);
      // UPS: tile origins are even, so a halo voxel's halved coordinate is a per-lane constant relative to the tile's
      // half-resolution origin: ((h0 - 1 + hh) >> 1) = h0 / 2 + ((hh - 1) >> 1), likewise along W
      const int rel = UPS ? ((((hh_ - 1) >> 1) * (W >> 1) + ((hw - 1) >> 1)) * CIN + pc * 8) * ES : ((hh_ * W + hw) * CIN + pc * 8) * ES;
      const uint32_t t1 = (c_ | 0x8080u) - lo, t2 = hi - c_;
      const bool in = (t1 & t2 & 0x8080u) == 0x8080u && hi_h >= 0;
      vk[k] = in ? (uint32_t)rel + (uint32_t)tile_off : DEAD;
      if constexpr (INM)      // 64 channels = 8 sign bytes per fine voxel
        vkm[k] = in ? (uint32_t)((hh_ * W + hw) * 8 + pc) + (uint32_t)(((P.h0 - 1) * W + (P.w0 - 1)) * 8) : DEAD;
    }
  };
  u32x4 stg[MAXP];
  uint32_t mstg[INM ? MAXP : 1];                     // (INM) the 8 sign bits of each piece in flight
  auto load_plane = [&](int gp) __attribute__((always_inline)) {   // input plane gp of P's column
    const uint32_t soff = (uint32_t)(UPS ? gp >> 1 : gp) * plane_bytes;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) stg[k] = __builtin_amdgcn_raw_buffer_load_b128(rxP, vk[k], soff, 0);
    if constexpr (INM) {
      const uint32_t moff = (uint32_t)gp * (uint32_t)(plane_vox * 8);
#pragma unroll
      for (int k = 0; k < MAXP; ++k) mstg[k] = __builtin_amdgcn_raw_buffer_load_b8(rmP, vkm[k], moff, 0);
    }
  };
  const float gain_in = a.in_gain, slope_in = a.in_mask_slope;   // (INM) sg_mask_piece_bf16: the arithmetic of sg_upscale2x_masked
  auto store_plane = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      u32x4 v = stg[k];
      if constexpr (INM) v = sg_mask_piece_bf16(v, mstg[k], gain_in, slope_in);
      if (k < 6) *reinterpret_cast<u32x4*>(smem + wofs + k * 2048) = v;
      else if (live6) *reinterpret_cast<u32x4*>(smem + wofs + k * 2048) = v;
    }
  };
  auto advance_P = [&]() {                           // the plane after the one requested last
    ++qP;
    if (++P.di == D) {
      P.di = 0;
      ++P.cj;
      if (qP < items_mine) enter_column_P();
    }
  };
  // ---- output side (cursor E: the input plane whose MFMAs ran last)
  Cur E{0, 0, 0, 0, 0};
  int qE = 0;
  __amdgpu_buffer_rsrc_t ryE, rsE, rmE, rpE;
  int colvoxE = 0;
  bool row_okE = false;
  auto enter_column_E = [&]() {
    enter_column(E);
    row_okE = E.h0 + wave < H;
    colvoxE = E.h0 * W + E.w0;
    ryE = rsrc_of(a.y, ysb, E.n0);
    if (SIGN) rsE = rsrc_of(a.sign_out, wsb, E.n0);
    if (MASK) rmE = rsrc_of(a.mask_bits, wsb, E.n0);
    if (PN) rpE = rsrc_of(a.pn_scale, psb, E.n0);
  };
  // LeakyReLU sign words of the output planes stored in the NEXT off-phase (masked epilogue), for my two voxels: plane E.di - 1
  // and, at the top of a column, plane D - 1 as well.  Requested one phase ahead; issued and consumed unconditionally.
  // (the top plane's words -- one phase in D -- are loaded where they are used: two registers less across the MFMA phase)
  uint32_t mbn[2] = {0u, 0u};
  auto request_mask = [&]() {
    if constexpr (MASK) {
      const bool live = qE < items_mine && row_okE && E.di >= 1;
      const uint32_t tv0 = (uint32_t)((E.di - 1) * plane_vox + colvoxE);
      mbn[0] = __builtin_amdgcn_raw_buffer_load_b32(rmE, live ? svo0 : DEAD, tv0 * 4u, 0);
      mbn[1] = __builtin_amdgcn_raw_buffer_load_b32(rmE, live ? svo0 + 64u : DEAD, tv0 * 4u, 0);
    }
  };

  // resident weights (all 8 waves): 108 fragments [tap][chunk][cout half] of 1 KiB in the lane order of the A operand
  for (int f = wave8; f < TAPS * 4; f += 8)
    __builtin_amdgcn_global_load_lds((gbl_ptr3p_t)(wp + ((int64_t)f << 10) + lane * 16), (lds_ptr3p_t)(wlds + ((size_t)f << 10)), 16, 0, 0);
  float* bias_lds = reinterpret_cast<float*>(wlds + P3_WBYTES);
  if (tid < 32) bias_lds[tid] = a.bias != nullptr ? a.bias[tid] : 0.f;

  const bool no_stage = (a.dbg_flags & 1) != 0, no_epi = (a.dbg_flags & 2) != 0;
  if (items_mine > 0) {
    enter_column_P();
    enter_column_E();
    load_plane(0);
    if (grp == 0) {      // the very first plane of group 0: latency exposed once
      store_plane();
      if (items_mine > 1) {
        advance_P();
        load_plane(P.di);
      }
    }
  }
  __syncthreads();

  const int wl_lo = P3_WOFF + lane * 16, wl_hi = wl_lo + 64 * 1024;
  const float inv_c = 1.f / (float)COUT;
  const float slope = a.act ? a.slope : 1.f;

  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && lane == 0 && wave == 0 && dbgi < 120)
      a.dbg[grp * 128 + dbgi] = __builtin_amdgcn_s_memtime();
    ++dbgi;
  };
  // one output plane of the wave's 32 voxels x 32 channels: [voxel half][channel half], element i of a tile = channel
  // 16 * ch + 4 * q4 + i of voxel v16 + 16 * vh
  f32x4 acc[3][2][2];
  auto init_acc = [&](f32x4 (&c)[2][2]) {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float b = bias_lds[16 * ch + 4 * q4 + i];      // bias rides in C
        c[0][ch][i] = b;
        c[1][ch][i] = b;
      }
    asm volatile("" : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[1][0]), "+v"(c[1][1]));      // four separate tuples from here on
  };
  auto epilogue = [&](f32x4 (&c)[2][2], int o, bool ok, const uint32_t (&mb)[2]) __attribute__((always_inline)) {
    const uint32_t tile_vox = (uint32_t)(o * plane_vox + colvoxE);   // within sample E.n0
#pragma unroll
    for (int vh = 0; vh < 2; ++vh) {
      if (slope != 1.f) {   // uniform
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int i = 0; i < 4; ++i) c[vh][ch][i] = sg_lrelu(c[vh][ch][i], slope);
      }
      if (PN) {
        float ss = 0.f;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int i = 0; i < 4; ++i) ss += c[vh][ch][i] * c[vh][ch][i];
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int i = 0; i < 4; ++i) c[vh][ch][i] *= sc;
        if (a.pn_scale != nullptr)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc), rpE, (ok && q4 == 0) ? svo0 + 64u * vh : DEAD, tile_vox * 4u, 0);
      }
      if (SIGN) {
        uint32_t b = 0u;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int i = 0; i < 4; ++i) b |= (__float_as_uint(c[vh][ch][i]) >> 31) << (16 * ch + i);
        b <<= 4 * q4;
        b |= (uint32_t)__shfl_xor((int)b, 16);
        b |= (uint32_t)__shfl_xor((int)b, 32);
        __builtin_amdgcn_raw_buffer_store_b32(b, rsE, (ok && q4 == 0) ? svo0 + 64u * vh : DEAD, tile_vox * 4u, 0);
      }
      if (MASK) {      // v *= bit ? slope : 1 as v += bit ? (slope - 1) * v : 0 (see sg_apply_sign_word)
        const uint32_t wsh = mb[vh] >> (4 * q4);
        const float sm1 = a.mask_slope - 1.f;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int t = ((int)(wsh << (31 - (16 * ch + i)))) >> 31;      // 1-bit signed field: 0 or -1
            c[vh][ch][i] += __uint_as_float((uint32_t)t & __float_as_uint(c[vh][ch][i] * sm1));
          }
      }
      // 16 contiguous bytes per lane: rows 1 / 3 of the channel-half-0 registers swap with rows 0 / 2 of the channel-half-1
      // registers (v_permlane16_swap), after which lane q4 holds channels [0, 16, 8, 24][q4] .. + 7
      const uint32_t a0 = sg_pack_bf16(c[vh][0][0], c[vh][0][1]), a1 = sg_pack_bf16(c[vh][0][2], c[vh][0][3]);
      const uint32_t b0 = sg_pack_bf16(c[vh][1][0], c[vh][1][1]), b1 = sg_pack_bf16(c[vh][1][2], c[vh][1][3]);
      const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
      u32x4 out;
      out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
      __builtin_amdgcn_raw_buffer_store_b128(out, ryE, ok ? yvo0 + 1024u * vh : DEAD, tile_vox * (uint32_t)(COUT * ES), 0);
      SG_STORE16_GUARD(out);
    }
  };
  auto off_phase = [&](f32x4 (&cA)[2][2], f32x4 (&cB)[2][2], f32x4 (&cC)[2][2], bool closes, bool stage) __attribute__((always_inline)) {
    uint32_t mb[2] = {mbn[0], mbn[1]};
    if (MASK) asm volatile("" : "+v"(mb[0]), "+v"(mb[1]));
    if (stage && !no_stage) store_plane();
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    if (closes) {
      const int p = E.di;
      const bool top = p == D - 1;
      if (!no_epi) {
        if (p >= 1) epilogue(cA, p - 1, row_okE, mb);
        if (top) {
          uint32_t mt[2] = {0u, 0u};
          if constexpr (MASK) {
            const uint32_t tv1 = (uint32_t)(p * plane_vox + colvoxE);
            mt[0] = __builtin_amdgcn_raw_buffer_load_b32(rmE, row_okE ? svo0 : DEAD, tv1 * 4u, 0);
            mt[1] = __builtin_amdgcn_raw_buffer_load_b32(rmE, row_okE ? svo0 + 64u : DEAD, tv1 * 4u, 0);
          }
          epilogue(cB, p, row_okE, mt);
        }
      }
      init_acc(cA);
      if (top) {
        init_acc(cB);
        init_acc(cC);
      }
      ++qE;
      if (++E.di == D) {
        E.di = 0;
        ++E.cj;
        if (E.cj < ncols_blk) enter_column_E();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (a.dbg_flags & 128) stamp();
    request_mask();
    if (stage) {
      advance_P();
      if (qP < items_mine && !no_stage) load_plane(P.di);
    }
    if (a.dbg_flags & 128) stamp();
  };
#define SG_P16_OFF(ROT, closes, stage) off_phase(acc[(ROT + 2) % 3], acc[ROT], acc[(ROT + 1) % 3], closes, stage)
  if (grp == 0) {
    init_acc(acc[0]); init_acc(acc[1]); init_acc(acc[2]);
    request_mask();
    for (int q = 0; q < items_mine; q += 3) {
      stamp();
      sg_unrolled_kp16<0>::run(acc, xa, wl_lo, wl_hi);
      stamp();
      __syncthreads();
      stamp();
      SG_P16_OFF(0, true, q + 1 < items_mine);
      stamp();
      __syncthreads();
      if (q + 1 < items_mine) sg_unrolled_kp16<1>::run(acc, xa, wl_lo, wl_hi);
      __syncthreads();
      if (q + 1 < items_mine) SG_P16_OFF(1, true, q + 2 < items_mine);
      __syncthreads();
      if (q + 2 < items_mine) sg_unrolled_kp16<2>::run(acc, xa, wl_lo, wl_hi);
      __syncthreads();
      if (q + 2 < items_mine) SG_P16_OFF(2, true, q + 3 < items_mine);
      __syncthreads();
    }
  } else {
    init_acc(acc[0]); init_acc(acc[1]); init_acc(acc[2]);
    if (items_mine > 0) SG_P16_OFF(2, false, true);
    for (int q = 0; q < items_mine; q += 3) {
      __syncthreads();
      stamp();
      sg_unrolled_kp16<0>::run(acc, xa, wl_lo, wl_hi);
      stamp();
      __syncthreads();
      stamp();
      SG_P16_OFF(0, true, q + 1 < items_mine);
      stamp();
      __syncthreads();
      if (q + 1 < items_mine) sg_unrolled_kp16<1>::run(acc, xa, wl_lo, wl_hi);
      __syncthreads();
      if (q + 1 < items_mine) SG_P16_OFF(1, true, q + 2 < items_mine);
      __syncthreads();
      if (q + 2 < items_mine) sg_unrolled_kp16<2>::run(acc, xa, wl_lo, wl_hi);
      __syncthreads();
      if (q + 2 < items_mine) SG_P16_OFF(2, true, q + 3 < items_mine);
    }
  }
#undef SG_P16_OFF
}

template <int EPI, bool UPS, bool INM>
int launch_fwd3p_inst(const ConvFwdArgs& a, unsigned gx, hipStream_t st) {
  if (sg_cfg().fwd3p_16) {
    auto kern16 = conv_fwd3p16_kernel<EPI, UPS, INM>;
    SG_ALLOW_160K_LDS(kern16);
    hipLaunchKernelGGL(kern16, dim3(gx), dim3(512), P3_LDS, st, a);
    return SG_OK;
  }
  auto kern = conv_fwd3p_kernel<EPI, UPS, INM>;
  SG_ALLOW_160K_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(gx), dim3(512), P3_LDS, st, a);
  return SG_OK;
}

}  // namespace

// bf16, 3 x 3 x 3, 64 -> 32 channels, whole 32-wide rows; optional fused nearest-x2 gather (with the input mask).  Sets
// *used = false (and launches nothing) for anything else: the caller falls back to the two-pass K split.
int sg_launch_fwd3p(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used) {
  *used = false;
  if (s->kd != 3 || s->kh != 3 || s->kw != 3 || s->cin != 64 || s->cout != 32 || a.xcs != 64 || a.xco != 0) return SG_OK;
  if (s->upsample_in && ((s->d | s->h | s->w) & 1)) return SG_OK;
  if (a.in_mask && !s->upsample_in) return SG_OK;
  if (s->d < 2 || (s->w % 32) != 0) return SG_OK;
  if (a.pool || a.pnb_y || a.addend) return SG_OK;
  if (a.mask_bits && (a.sign_out || a.pixel_norm || a.bias || a.act)) return SG_OK;
  if (s->upsample_in && !a.in_mask && a.mask_bits) return SG_OK;      // (the plain fused-gather variants carry no output mask)
  if (a.in_mask && (a.sign_out || a.pixel_norm)) return SG_OK;
  a.g = sg_make_geom(s, 256, /*prefer_w32=*/true, /*td=*/2, /*th=*/4);
  const sg_tile_geom& g = a.g;
  if (g.TN != 1 || g.TH != 4 || g.TW != 32 || g.HH != 6 || g.HW != 34) return SG_OK;
  {   // buffer addressing (rebased per sample): one sample of every tensor this kernel touches stays below 2 GiB
    const int64_t svox = (int64_t)s->d * s->h * s->w;
    if (svox * 64 * 2 >= (1ll << 31) || svox * 8 >= (1ll << 31)) return SG_OK;
  }
  const int npair = g.nTn * ((g.nTh + 1) / 2) * g.nTw;
  int gx = 256;
  if (npair < gx) gx = npair / 8 * 8;
  if (gx < 8) return SG_OK;
  const int epi = (a.sign_out ? SG_EP_SIGN : 0) | (a.mask_bits ? SG_EP_MASK : 0) | (a.pixel_norm ? SG_EP_PN : 0);
  int rc = SG_OK;
  if (s->upsample_in && a.in_mask) {
    switch (epi) {
      case 0: rc = launch_fwd3p_inst<0, true, true>(a, (unsigned)gx, st); break;
      case SG_EP_MASK: rc = launch_fwd3p_inst<SG_EP_MASK, true, true>(a, (unsigned)gx, st); break;
      default: return SG_OK;
    }
  } else if (s->upsample_in) {
    switch (epi) {
      case 0: rc = launch_fwd3p_inst<0, true, false>(a, (unsigned)gx, st); break;
      case SG_EP_SIGN: rc = launch_fwd3p_inst<SG_EP_SIGN, true, false>(a, (unsigned)gx, st); break;
      case SG_EP_PN: rc = launch_fwd3p_inst<SG_EP_PN, true, false>(a, (unsigned)gx, st); break;
      case SG_EP_PN | SG_EP_SIGN: rc = launch_fwd3p_inst<SG_EP_PN | SG_EP_SIGN, true, false>(a, (unsigned)gx, st); break;
      default: return SG_OK;
    }
  } else {
    switch (epi) {
      case 0: rc = launch_fwd3p_inst<0, false, false>(a, (unsigned)gx, st); break;
      case SG_EP_SIGN: rc = launch_fwd3p_inst<SG_EP_SIGN, false, false>(a, (unsigned)gx, st); break;
      case SG_EP_MASK: rc = launch_fwd3p_inst<SG_EP_MASK, false, false>(a, (unsigned)gx, st); break;
      case SG_EP_PN: rc = launch_fwd3p_inst<SG_EP_PN, false, false>(a, (unsigned)gx, st); break;
      case SG_EP_PN | SG_EP_SIGN: rc = launch_fwd3p_inst<SG_EP_PN | SG_EP_SIGN, false, false>(a, (unsigned)gx, st); break;
      default: return SG_OK;
    }
  }
  if (rc != SG_OK) return rc;
  if (sg_cfg().fwd3p_16) SG_KNAME("conv_fwd3p16<bf16,64->32>");
  else SG_KNAME("conv_fwd3p<bf16,64->32>");
  SG_LAUNCH_CHECK();
  *used = true;
  return SG_OK;
}
