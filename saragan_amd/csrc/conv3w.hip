// 3x3x3 convolution / data gradient with 32 input channels (bf16) on gfx950: WAVE-PRIVATE halo planes, sliding accumulators,
// v_mfma_f32_16x16x32_bf16, NO workgroup barrier in the main loop.  Replaces the sliding-halo kernel (conv_fwd3s, conv3d.hip)
// for tf.nn.conv3d as called by SURFGAN_3D/networks/ops.py:147-150 at the top level of the benchmarked networks
// (networks/pgan/discriminator.py:25-45 conv_1 / conv_2 and their data gradients, networks/pgan/generator.py:48-71 conv_2).
//
// Why another formulation.  The ping-pong kernels (two 4-wave groups, one in its MFMA phase while the other stages and stores,
// two block barriers per phase) keep the matrix pipe 0.75-0.79 busy: every phase boundary costs a barrier, a fragment-read
// prologue and the slower group's slack, and the chip's clock under that load is set by what the MFMA phases burn beside the
// MFMAs -- mostly LDS fragment reads (7 per 6 MFMAs of 32 cycles).  Here
//  * a wave owns TWO output rows x 32 voxels and keeps THREE output planes of them in registers (24 accumulator tiles of 16 x 16,
//    96 VGPRs): input plane p contributes through tap plane kd to output plane p + 1 - kd; after the MFMAs of plane p output
//    plane p - 1 is complete, is stored, and its registers start plane p + 2 (three straight-line code variants, so that
//    accumulators never move).  Per input plane and wave: 216 MFMAs of 16 cycles from 54 weight fragments (each feeds 4 MFMAs:
//    2 rows x 2 voxel halves) and 24 activation fragments (4 halo rows x 3 kw x 2 voxel halves, each feeds up to 12):
//    0.72 fragment reads per 32 MFMA cycles;
//  * the wave stages ITS OWN halo plane (4 rows x 34 voxels x 64 B = 8.5 KiB) global -> registers -> LDS, one plane ahead, and is
//    the only reader of that buffer: no barrier, no partner group to wait for.  The two waves of a SIMD are independent pipelines
//    (K loop, then write the next plane, request the one after, store the finished output plane); whenever one of them is outside
//    its K loop the other has the matrix pipe to itself, and a single wave issues these MFMAs back to back.
//    The price: halo rows shared by the waves of a block are fetched once per wave (L1 / L2 hits: 4 rows per 2 instead of 18 per 16).
//
// LDS map (125 056 B): [weights of this block's 32-channel output tile: 54 fragments [tap][channel half] of 1 KiB in the lane
// order of the MFMA's A operand][8 wave halo buffers of 8 704 B][bias: 32 floats].  Halo rows are 64 B (32 channels); 16-byte
// slot s of row r lives at s ^ (((r >> 2) & 1) << 1): a B-operand read (16 consecutive rows, slot = lane >> 4) is conflict free
// for every tap shift (the layout of conv_fwd3p16, conv3p.hip; derivation in DESIGN_NOTES section 8).
#include "common.h"
#include "prof.h"
#include "conv_args.h"

typedef __attribute__((address_space(3))) void* lds_ptr3w_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr3w_t;

namespace {

constexpr int W3_NFRAG = 54;
constexpr int W3_WBYTES = W3_NFRAG * 1024;
constexpr int W3_HROWS = 4 * 34;                  // halo rows of a wave: 4 H rows x 34 voxels
constexpr int W3_HALO = W3_HROWS * 64;            // 8 704 B
constexpr int W3_HOFF = W3_WBYTES;
constexpr int W3_BOFF = W3_HOFF + 8 * W3_HALO;
constexpr int W3_POFF = W3_BOFF + 128;             // progress words of the 8 waves
constexpr int W3_MOFF = W3_POFF + 64;              // (RGB / PWB) the pointwise layer's 32 matrix values
constexpr int W3_AOFF = W3_MOFF + 128;             // (PWB) per-wave running sums: [64 lanes][16] f32
constexpr int W3_LDS = W3_MOFF + 128;
constexpr int W3_LDS_PWB = W3_AOFF + 8 * 4096;
static_assert(W3_LDS_PWB <= 160 * 1024, "LDS budget");
constexpr int W3_NPIECE = 9;                      // 16-byte pieces per lane and plane: 136 rows x 4 / 64 = 8.5
static_assert(W3_LDS <= 160 * 1024, "LDS budget");
#ifndef W3_PFW
#define W3_PFW 3      // weight fragments are read this many steps (of 4 MFMAs = 64 cycles) ahead of their use
#endif

// K loop of one input plane.  Step s = kw * 18 + kh * 6 + kd * 2 + ch: ONE weight fragment (tap (kd, kh, kw), output-channel
// half ch) and four MFMAs: output rows wr = 0, 1 (halo rows kh + wr) x voxel halves vh; the accumulator plane that tap plane kd
// feeds is (ROT + 1 - kd) mod 3, ROT = running phase index mod 3.  Activation fragments arrive as a stream of halo rows
// j = kw * 4 + r (two reads each: the voxel halves) into four row slots, at least a kh block (six steps = 384 MFMA cycles) before
// their first use; weight fragments W3_PFW steps ahead into a ring of W3_PFW + 1.
// ISSUE SLOTS.  A 16-cycle MFMA holds the SIMD's vector issue for 8 cycles and every other instruction of the wave costs an
// issue slot of ~4: with more than one instruction between two MFMAs the pipe waits for the wave (first version: reads, wait
// and hazard pad in front of four back-to-back MFMAs: 18.0 cycles per MFMA alone on the SIMD).  So a step is
//   MFMA, weight read (s + PFW), MFMA, row read A, MFMA, row read B, MFMA, s_waitcnt for step s + 1
// -- at most one instruction per gap, the wait through the builtin (an inline-asm wait between two asm statements on the same
// registers makes hipcc pad an s_nop).  The s_waitcnt immediates are computed from the issue order at compile time.
template <int ROT>
struct sg_kloop3w {
  static constexpr int NS = 54, PFW = W3_PFW, RW = PFW + 1, NROW = 12;
  static constexpr int kw_of(int s) { return s / 18; }
  static constexpr int kh_of(int s) { return (s % 18) / 6; }
  static constexpr int kd_of(int s) { return (s % 6) / 2; }
  static constexpr int ch_of(int s) { return s & 1; }
  static constexpr int frag_of(int s) { return (kd_of(s) * 9 + kh_of(s) * 3 + kw_of(s)) * 2 + ch_of(s); }
  // step in whose gaps halo row j = (kw, r) is read (< 0: in the prologue): its slot r is free by then and its first use is
  // at least five steps away; one row per step
  static constexpr int act_step(int j) {
    const int kw = j / 4, r = j % 4;
    return r == 0 ? kw * 18 - 11 : r == 1 ? kw * 18 - 6 : r == 2 ? kw * 18 : kw * 18 + 6;
  }
  static constexpr int row_at(int t) {           // the row read in step t's gaps, or -1
    for (int j = 0; j < NROW; ++j) if (act_step(j) == t) return j;
    return -1;
  }
  static constexpr bool one_row_per_step() {
    for (int a = 0; a < NROW; ++a)
      for (int b = a + 1; b < NROW; ++b) if (act_step(a) >= 0 && act_step(a) == act_step(b)) return false;
    return true;
  }
  static_assert(one_row_per_step(), "two rows in one step");
  static constexpr int pro_rows() {
    int n = 0;
    for (int j = 0; j < NROW; ++j) if (act_step(j) < 0) ++n;
    return n;
  }
  static constexpr int group_reads(int t) { return ((t + PFW < NS) ? 1 : 0) + (row_at(t) >= 0 ? 2 : 0); }   // reads in step t's gaps
  static constexpr int total_through(int t) {    // reads issued up to and including step t's gaps (t = -1: the prologue)
    int n = 2 * pro_rows() + PFW;
    for (int u = 0; u <= t; ++u) n += group_reads(u);
    return n;
  }
  static constexpr int idx_w(int s) {            // program-order index of the read of step s's weight fragment
    if (s < PFW) return 2 * pro_rows() + s;      // prologue: activation rows first, then the weights
    return total_through(s - PFW - 1);           // first read of step s - PFW
  }
  static constexpr int idx_a(int j) {            // ... of the LAST read of halo row j
    const int t = act_step(j);
    if (t < 0) {
      int before = 0;
      for (int i = 0; i < j; ++i) if (act_step(i) < 0) before += 2;
      return before + 1;
    }
    return total_through(t - 1) + ((t + PFW < NS) ? 1 : 0) + 1;
  }
  static constexpr int younger(int s) {          // reads issued after the last one step s needs, when its wait executes (end of step s - 1)
    const int j0 = kw_of(s) * 4 + kh_of(s);
    int need = idx_w(s);
    if (idx_a(j0) > need) need = idx_a(j0);
    if (idx_a(j0 + 1) > need) need = idx_a(j0 + 1);
    return total_through(s - 1) - 1 - need;
  }
  static constexpr bool counts_ok() {
    for (int s = 0; s < NS; ++s) if (younger(s) < 0 || younger(s) > 15) return false;
    return true;
  }
  static_assert(counts_ok(), "a needed read is issued too late, or lgkmcnt (a 4-bit counter) would overflow");

  template <int S>
  static __device__ __forceinline__ void wload(u32x4 (&wfr)[RW], int wl) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wfr[S % RW]) : "v"(wl), "n"(frag_of(S) << 10));
  }
  template <int J, int VH>
  static __device__ __forceinline__ void aload(u32x4 (&xfr)[4][2], const int (&xa)[4][3]) {
    constexpr int kw = J / 4, r = J % 4;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xfr[r][VH]) : "v"(xa[r][kw]), "n"(VH * 1024));
  }
  static __device__ __forceinline__ void mfma(f32x4& c, const u32x4& a_, const u32x4& b_) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a_), "v"(b_));
  }
  template <int S>
  static __device__ __forceinline__ void step(f32x4 (&acc)[3][2][2][2], u32x4 (&wfr)[RW], u32x4 (&xfr)[4][2], const int (&xa)[4][3],
                                              int wl) {
    if constexpr (S < NS) {
      constexpr int kh = kh_of(S), kd = kd_of(S), ch = ch_of(S), pl = (ROT + 1 - kd + 3) % 3, J = row_at(S);
      mfma(acc[pl][0][0][ch], wfr[S % RW], xfr[kh][0]);
      if constexpr (S + PFW < NS) wload<S + PFW>(wfr, wl);
      mfma(acc[pl][0][1][ch], wfr[S % RW], xfr[kh][1]);
      if constexpr (J >= 0) aload<(J >= 0 ? J : 0), 0>(xfr, xa);
      mfma(acc[pl][1][0][ch], wfr[S % RW], xfr[kh + 1][0]);
      if constexpr (J >= 0) aload<(J >= 0 ? J : 0), 1>(xfr, xa);
      mfma(acc[pl][1][1][ch], wfr[S % RW], xfr[kh + 1][1]);
      if constexpr (S + 1 < NS) {
        SG_WAIT_LGKM(younger(S + 1));
        __builtin_amdgcn_sched_barrier(0);
      }
      step<S + 1>(acc, wfr, xfr, xa, wl);
    }
  }
  template <int J>
  static __device__ __forceinline__ void aprologue(u32x4 (&xfr)[4][2], const int (&xa)[4][3]) {
    if constexpr (J < NROW) {
      if constexpr (act_step(J) < 0) {
        aload<J, 0>(xfr, xa);
        aload<J, 1>(xfr, xa);
      }
      aprologue<J + 1>(xfr, xa);
    }
  }
  template <int S>
  static __device__ __forceinline__ void wprologue(u32x4 (&wfr)[RW], int wl) {
    if constexpr (S < PFW) {
      wload<S>(wfr, wl);
      wprologue<S + 1>(wfr, wl);
    }
  }
  static __device__ __forceinline__ void run(f32x4 (&acc)[3][2][2][2], const int (&xa)[4][3], int wl) {
    u32x4 wfr[RW], xfr[4][2];
    SG_KLOOP_BEGIN();
    aprologue<0>(xfr, xa);
    wprologue<0>(wfr, wl);
    SG_WAIT_LGKM(younger(0));
    __builtin_amdgcn_sched_barrier(0);
    step<0>(acc, wfr, xfr, xa, wl);
    // wait states after the last in-place MFMA before anything reads the accumulators (see sg_mfma_drain); tied to all of them
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3"
                 : "+v"(acc[0][0][0][0]), "+v"(acc[0][0][0][1]), "+v"(acc[0][0][1][0]), "+v"(acc[0][0][1][1]),
                   "+v"(acc[0][1][0][0]), "+v"(acc[0][1][0][1]), "+v"(acc[0][1][1][0]), "+v"(acc[0][1][1][1]),
                   "+v"(acc[1][0][0][0]), "+v"(acc[1][0][0][1]), "+v"(acc[1][0][1][0]), "+v"(acc[1][0][1][1]),
                   "+v"(acc[1][1][0][0]), "+v"(acc[1][1][0][1]), "+v"(acc[1][1][1][0]), "+v"(acc[1][1][1][1]),
                   "+v"(acc[2][0][0][0]), "+v"(acc[2][0][0][1]), "+v"(acc[2][0][1][0]), "+v"(acc[2][0][1][1]),
                   "+v"(acc[2][1][0][0]), "+v"(acc[2][1][0][1]), "+v"(acc[2][1][1][0]), "+v"(acc[2][1][1][1]));
    SG_KLOOP_END();
  }
};

struct Fwd3wArgs {
  ConvFwdArgs a;
  int nHb, nWb, nseg, seglen;        // block columns per sample (16 x 32 output voxels), D segments per column and their length
  int nitems;                        // columns x segments
};

template <int EPI>
__global__ __launch_bounds__(512) void conv_fwd3w_kernel(Fwd3wArgs fa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ConvFwdArgs& a = fa.a;
  constexpr int ES = 2, CIN = 32;
  constexpr uint32_t DEAD = 0x80000000u;             // byte offset beyond every buffer: loads return 0, stores drop
  constexpr bool SIGN = (EPI & SG_EP_SIGN) != 0, MASK = (EPI & SG_EP_MASK) != 0, PN = (EPI & SG_EP_PN) != 0,
                 POOL = (EPI & SG_EP_POOL) != 0, PNB = (EPI & SG_EP_PNB) != 0, RGB = (EPI & SG_EP_RGB) != 0, PWB = (EPI & SG_EP_PWB) != 0,
                 POOL3 = (EPI & SG_EP_POOL3) != 0;
  static_assert(!(PN && (POOL || MASK || PNB)) && !(PNB && (POOL || SIGN || !MASK)) && !(MASK && SIGN) &&
                !(RGB && (POOL || MASK || PNB)) && !(PWB && (!MASK || POOL || PNB || RGB)), "unsupported epilogue combination");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int v16 = lane & 15, q4 = lane >> 4;         // MFMA 16x16x32: voxel column / K group (inputs), row group (outputs)
  const int nt0 = blockIdx.y;
  const int D = a.g.D, H = a.g.H, W = a.g.W, cout = a.cout, ntile = a.ntile;
  const int64_t svox = (int64_t)D * H * W;
  auto rsrc_of = [&](const void* base, int64_t sample_bytes, int n0) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + n0 * sample_bytes, 0,
                                             (int)sample_bytes, 0x00020000);
  };
  const int64_t xsb = svox * CIN * ES, ysb = svox * cout * ES, wsb = svox * ntile * 4, psb = svox * 4;

  // resident weights of output tile nt0 (all 8 waves) and bias
  {
    const char* wp = reinterpret_cast<const char*>(a.wp) + (size_t)nt0 * W3_WBYTES;
    for (int f = w8; f < W3_NFRAG; f += 8)
      __builtin_amdgcn_global_load_lds((gbl_ptr3w_t)(wp + ((int64_t)f << 10) + lane * 16), (lds_ptr3w_t)(smem + ((size_t)f << 10)), 16, 0, 0);
  }
  float* bias_lds = reinterpret_cast<float*>(smem + W3_BOFF);
  if (tid < 32) bias_lds[tid] = a.bias != nullptr ? a.bias[nt0 * 32 + tid] : 0.f;
  if (tid < 8) reinterpret_cast<int*>(smem + W3_POFF)[tid] = 0;
  if constexpr ((EPI & (SG_EP_RGB | SG_EP_PWB)) != 0) {
    if (tid < 32) reinterpret_cast<float*>(smem + W3_MOFF)[tid] = ((EPI & SG_EP_RGB) ? a.rgb_w : a.pw_wmat)[tid];
  }
  __syncthreads();      // the only barrier: from here on a wave reads what it wrote itself (and the weights)

  // item schedule: XCD group xg owns a contiguous chunk of the item list (neighbouring columns share halo rows in its L2)
  const int nitems_all = fa.nitems;
  const int xg = blockIdx.x & 7, bslot = blockIdx.x >> 3;
  const int per_x = gridDim.x >> 3;
  const int ipx = (nitems_all + 7) >> 3;
  const int i_begin = xg * ipx, i_end = min(nitems_all, i_begin + ipx);
  const int ifirst = i_begin + bslot;
  const int nitems = ifirst < i_end ? (i_end - ifirst + per_x - 1) / per_x : 0;
  if (nitems == 0) {
    if constexpr (PWB) {      // every wave owns a row of partial sums: mine are zero
      if (lane < 32) {
        a.pw_part[((size_t)(blockIdx.x * 8 + w8) * 2 + 0) * 32 + lane] = 0.f;
        a.pw_part[((size_t)(blockIdx.x * 8 + w8) * 2 + 1) * 32 + lane] = 0.f;
      }
    }
    return;
  }
  if ((a.dbg_flags & 8192) && w8 >= 4) return;      // diagnostic: one wave per SIMD (half the output is not computed)
  const bool no_stage = (a.dbg_flags & 1) != 0, no_epi = (a.dbg_flags & 2) != 0;      // diagnostic ablations (0 in production)

  const int hbase = W3_HOFF + w8 * W3_HALO;
  // fragment addresses (B operand: lane = voxel column l & 15, K group l >> 4): halo row r (0..3), tap column kw; the second voxel
  // half is + 16 rows = + 1024 B (same swizzle bit)
  int xa[4][3];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int row = r * 34 + v16 + kw;
      xa[r][kw] = hbase + row * 64 + ((q4 ^ (((row >> 2) & 1) << 1)) << 4);
    }
  // staging: piece k of this lane = 16-byte slot (lane & 3) of halo row k * 16 + (lane >> 2); (row >> 2) & 1 = (lane >> 4) & 1, so
  // the swizzle bit is the lane's and piece k sits k KiB after piece 0
  const int wofs = hbase + (lane >> 2) * 64 + (((lane & 3) ^ (((lane >> 4) & 1) << 1)) << 4);
  const uint32_t plane_bytes = (uint32_t)(H * W * CIN * ES);
  const int plane_vox = H * W;

  // A cursor walks my (item, input plane) sequence.  Two run one behind the other: P, the plane requested last, and E, the
  // plane whose MFMAs ran last.
  struct Cur { int j, p, n0, h0, w0, o_lo, o_hi, p_hi; };
  auto enter_item = [&](Cur& c) {
    const int t = ifirst + c.j * per_x;
    const int col = t / fa.nseg, seg = t - col * fa.nseg;
    const int c1 = col / fa.nWb;
    c.w0 = (col - c1 * fa.nWb) * 32;
    const int c2 = c1 / fa.nHb;
    c.h0 = (c1 - c2 * fa.nHb) * 16 + 2 * w8;           // my two rows (may lie beyond H: dead rows)
    c.n0 = c2;
    c.o_lo = seg * fa.seglen;
    c.o_hi = min(D, c.o_lo + fa.seglen);
    c.p = max(c.o_lo - 1, 0);
    c.p_hi = min(c.o_hi, D - 1);
  };
  // ---- halo side
  Cur P{0, 0, 0, 0, 0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rxP;
  uint32_t vk[W3_NPIECE];
  auto enter_item_P = [&]() {
    enter_item(P);
    rxP = rsrc_of(a.x, xsb, P.n0);
    const int tile_off = ((P.h0 - 1) * W + (P.w0 - 1)) * CIN * ES;      // may be negative: only dead lanes go below 0
#pragma unroll
    for (int k = 0; k < W3_NPIECE; ++k) {
      int row = k * 16 + (lane >> 2);
      asm volatile("" : "+v"(row));                   // (rebuilt per item: keeps 9 more offsets out of the registers that live across the K loops)
      const int hr = row / 34, hw = row - hr * 34;
      const int gh = P.h0 - 1 + hr, gw = P.w0 - 1 + hw;
      const bool in = row < W3_HROWS && gh >= 0 && gh < H && gw >= 0 && gw < W;
      vk[k] = in ? (uint32_t)(((hr * W + hw) * CIN + (lane & 3) * 8) * ES + tile_off) : DEAD;
    }
  };
  u32x4 stg[W3_NPIECE];
  auto load_plane = [&](int gp) __attribute__((always_inline)) {
    const uint32_t soff = (uint32_t)gp * plane_bytes;
#pragma unroll
    for (int k = 0; k < W3_NPIECE; ++k) stg[k] = __builtin_amdgcn_raw_buffer_load_b128(rxP, vk[k], soff, 0);
  };
  auto store_plane = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < W3_NPIECE; ++k) {
      if (k < 8) *reinterpret_cast<u32x4*>(smem + wofs + k * 1024) = stg[k];
      else if (lane < 32) *reinterpret_cast<u32x4*>(smem + wofs + k * 1024) = stg[k];
    }
  };
  auto advance_P = [&]() {
    if (P.p == P.p_hi) {
      ++P.j;
      if (P.j < nitems) enter_item_P();
    } else ++P.p;
  };
  // ---- output side
  Cur E{0, 0, 0, 0, 0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t ryE, rsE, rmE, rpE, rbE, rqE, rgE, rxiE, rdxE;
  int colvoxE = 0;
  bool row_ok[2] = {false, false};
  auto enter_item_E = [&]() {
    enter_item(E);
    row_ok[0] = E.h0 < H;
    row_ok[1] = E.h0 + 1 < H;
    colvoxE = E.h0 * W + E.w0;
    ryE = rsrc_of(a.y, POOL3 ? ysb / 8 : POOL ? ysb / 4 : ysb, E.n0);
    if (SIGN) rsE = rsrc_of(a.sign_out, wsb, E.n0);
    if (MASK) rmE = rsrc_of(a.mask_bits, wsb, E.n0);
    if (PN) rpE = rsrc_of(a.pn_scale, psb, E.n0);
    if (PNB) {
      rbE = rsrc_of(a.pnb_y, ysb, E.n0);
      rqE = rsrc_of(a.pnb_scale, psb, E.n0);
    }
    if (RGB) rgE = rsrc_of(a.rgb_out, svox * ES, E.n0);
    if (PWB) {
      rxiE = rsrc_of(a.pw_x, svox * ES, E.n0);
      rdxE = rsrc_of(a.pw_dx, svox * ES, E.n0);
    }
  };
  auto advance_E = [&]() {
    if (E.p == E.p_hi) {
      ++E.j;
      if (E.j < nitems) enter_item_E();
    } else ++E.p;
  };
  // my four output voxels of a plane: (row wr, w = v16 + 16 * vh) of the wave's 2 x 32 strip, relative to (h0, w0).  After the
  // half-row exchange of the store path lane q4 holds channels [0, 16, 8, 24][q4] .. + 7 of its voxel.
  const uint32_t cb = (uint32_t)((q4 & 1) * 16 + (q4 >> 1) * 8);
  const uint32_t yv0 = (uint32_t)((v16 * cout + nt0 * 32) * ES) + cb * ES;      // + (wr * W + 16 * vh) * cout * ES
  const uint32_t sv0 = (uint32_t)(v16 * ntile + nt0) * 4u;                      // + (wr * W + 16 * vh) * ntile * 4
  const uint32_t pv0 = (uint32_t)v16 * 4u;                                      // per-voxel f32 (pixel-norm factor)
  auto vrel = [&](int wr, int vh) { return (uint32_t)(wr * W + 16 * vh); };
  // LeakyReLU sign words of the output plane stored in the NEXT off-phase (masked epilogue), requested one phase ahead; issued and
  // consumed unconditionally (DEAD offset: no memory access) so that no s_waitcnt vmcnt(0) ends up in front of an MFMA
  uint32_t mbn[2][2] = {{0u, 0u}, {0u, 0u}};
  auto request_mask = [&]() {
    if constexpr (MASK) {
      const bool live = E.j < nitems && E.p - 1 >= E.o_lo;
      const uint32_t tv = (uint32_t)((E.p - 1) * plane_vox + colvoxE);
#pragma unroll
      for (int wr = 0; wr < 2; ++wr)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh)
          mbn[wr][vh] = __builtin_amdgcn_raw_buffer_load_b32(rmE, (live && row_ok[wr]) ? sv0 + vrel(wr, vh) * (uint32_t)(ntile * 4) : DEAD,
                                                             tv * (uint32_t)(ntile * 4), 0);
    }
  };

  // diagnostic stamps (tools/ts_conv3w.py; a.dbg is NULL in production): shader clock per wave of block (8, 0) at the start of a K
  // loop, at its end, after the staging part of the off-phase and at the end of the off-phase; entry 127 = s_memrealtime pairs
  int myphase = 0;
  int dbgi = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && lane == 0 && dbgi < 120) {
      a.dbg[w8 * 256 + dbgi] = __builtin_amdgcn_s_memtime();
      a.dbg[w8 * 256 + 128 + dbgi] = __builtin_amdgcn_s_memrealtime();
    }
    ++dbgi;
  };
  const float inv_c = 1.f / (float)cout;
  const float slope = a.act ? a.slope : 1.f;         // max(x, 1 * x) = x: no branch for "no activation"
  const int wl = lane * 16;
  // one output plane of the wave: [row][voxel half][channel half], element i of a tile = channel 16 * ch + 4 * q4 + i
  f32x4 acc[3][2][2][2];
  // bias rides in C: my eight channels' values, kept in registers (re-read from LDS per plane, the reads and their wait sat in
  // the off-phase: ~150 cycles of latency per phase); the masked variants -- data gradients -- have none
  float bq[2][4];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch)
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[ch][i] = MASK ? 0.f : bias_lds[16 * ch + 4 * q4 + i];
  auto init_acc = [&](f32x4 (&c)[2][2][2]) {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float b = bq[ch][i];
        c[0][0][ch][i] = b; c[0][1][ch][i] = b; c[1][0][ch][i] = b; c[1][1][ch][i] = b;
      }
    asm volatile("" : "+v"(c[0][0][0]), "+v"(c[0][0][1]), "+v"(c[0][1][0]), "+v"(c[0][1][1]),
                 "+v"(c[1][0][0]), "+v"(c[1][0][1]), "+v"(c[1][1][0]), "+v"(c[1][1][1]));      // eight separate tuples from here on
  };
  // (RGB / PWB) the eight values of the pointwise layer's matrix for my channels; (PWB) running sums of x * g and g over my voxels
  // (kept in LDS, not in registers that would live across the K loops: the matrix is re-read per plane, the sums are a
  // wave-private read-add-write per plane)
  const float* pwm_lds = reinterpret_cast<const float*>(smem + W3_MOFF);
  f32x4* pwacc = reinterpret_cast<f32x4*>(smem + W3_AOFF + w8 * 4096 + lane * 64);
  if constexpr (PWB) {
#pragma unroll
    for (int k = 0; k < 4; ++k) pwacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float rgb_b = (RGB && a.rgb_bias != nullptr) ? a.rgb_bias[0] : 0.f;
  const bool want_pw_dx = PWB && a.pw_dx != nullptr;
  f32x4 hold[POOL ? 2 : 1][POOL ? 2 : 1];            // (POOL) W-pair sums of the even plane of a D pair: [row][channel half]
  // sum / OR over the four lanes (l, l ^ 16, l ^ 32, l ^ 48) that share a voxel: two swaps on the VALU (v_permlane16_swap /
  // v_permlane32_swap exchange the odd rows / upper half of one operand with the even rows / lower half of the other: with both
  // operands = x every lane ends up with its own and its partner's value).  Through ds_bpermute (__shfl_xor) each step was an
  // LDS round trip behind s_waitcnt lgkmcnt(0): 16 of them in a row made the sign-word epilogue 2.7k cycles.
  auto quad_sum = [](float v) {
    const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s1 = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
  };
  auto quad_or = [](uint32_t v) {
    const auto r1 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const uint32_t s1 = r1[0] | r1[1];
    const auto r2 = __builtin_amdgcn_permlane32_swap(s1, s1, false, false);
    return r2[0] | r2[1];
  };
  // one finished output plane o of E's item.  Written as stages over the wave's four 16-voxel tiles (row wr, voxel half vh) so
  // that a uniform condition is ONE branch per plane, not one per tile.
  auto epilogue = [&](f32x4 (&c)[2][2][2], int o, const uint32_t (&mb)[2][2]) __attribute__((always_inline)) {
    const uint32_t tile_vox = (uint32_t)(o * plane_vox + colvoxE);   // within sample E.n0
    const uint32_t ysoff = tile_vox * (uint32_t)(cout * ES), ssoff = tile_vox * (uint32_t)(ntile * 4), psoff4 = tile_vox * 4u;
    if (slope != 1.f) {   // uniform
#pragma unroll
      for (int wr = 0; wr < 2; ++wr)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[wr][vh][ch][i] = sg_lrelu(c[wr][vh][ch][i], slope);
    }
    if constexpr (PN) {
      const bool want = a.pn_scale != nullptr;
#pragma unroll
      for (int wr = 0; wr < 2; ++wr)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh) {
          f32x4(&t)[2] = c[wr][vh];
          float ss = 0.f;
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) ss += t[ch][i] * t[ch][i];
          ss = quad_sum(ss);
          const float sc = rsqrtf(ss * inv_c + a.eps);
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) t[ch][i] *= sc;
          if (want)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc), rpE, (row_ok[wr] && q4 == 0) ? pv0 : DEAD,
                                                  psoff4 + vrel(wr, vh) * 4u, 0);
        }
    }
    if constexpr (SIGN) {
      // the tile's word of every voxel ends up in all four of its lanes: lane group q4 stores the words of tile q4 = 2 wr + vh
      uint32_t word = 0u;
#pragma unroll
      for (int wr = 0; wr < 2; ++wr)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh) {
          f32x4(&t)[2] = c[wr][vh];
          uint32_t b = 0u;
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) b |= (__float_as_uint(t[ch][i]) >> 31) << (16 * ch + i);
          b = quad_or(b << (4 * q4));
          if (q4 == 2 * wr + vh) word = b;
        }
      const int mwr = q4 >> 1, mvh = q4 & 1;
      __builtin_amdgcn_raw_buffer_store_b32(word, rsE, row_ok[mwr] ? sv0 + (uint32_t)(mwr * W + 16 * mvh) * (uint32_t)(ntile * 4) : DEAD,
                                            ssoff, 0);
    }
    if constexpr (PNB) {
      // d/dx of y = x * s, s = rsqrt(mean_c(x^2) + eps):  s * (g - y * mean_c(g * y)); the accumulator holds g
      // (ops.py:308-310 pixel_norm, backward); y: my 2 x 4 channels of the stage's output, s: its per-voxel factor
      u32x2 y0[2][2], y1[2][2];
      float ps[2][2];
#pragma unroll
      for (int wr = 0; wr < 2; ++wr)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh) {
          const uint32_t yoff = row_ok[wr] ? (uint32_t)((v16 * cout + nt0 * 32 + 4 * q4) * ES) : DEAD;
          const uint32_t so = ysoff + vrel(wr, vh) * (uint32_t)(cout * ES);
          y0[wr][vh] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rbE, yoff, so, 0));
          y1[wr][vh] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rbE, yoff, so + 32u, 0));
          ps[wr][vh] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rqE, row_ok[wr] ? pv0 : DEAD, psoff4 + vrel(wr, vh) * 4u, 0));
        }
#pragma unroll
      for (int wr = 0; wr < 2; ++wr)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh) {
          f32x4(&t)[2] = c[wr][vh];
          float yv[2][4];
          yv[0][0] = __uint_as_float(y0[wr][vh][0] << 16); yv[0][1] = __uint_as_float(y0[wr][vh][0] & 0xFFFF0000u);
          yv[0][2] = __uint_as_float(y0[wr][vh][1] << 16); yv[0][3] = __uint_as_float(y0[wr][vh][1] & 0xFFFF0000u);
          yv[1][0] = __uint_as_float(y1[wr][vh][0] << 16); yv[1][1] = __uint_as_float(y1[wr][vh][0] & 0xFFFF0000u);
          yv[1][2] = __uint_as_float(y1[wr][vh][1] << 16); yv[1][3] = __uint_as_float(y1[wr][vh][1] & 0xFFFF0000u);
          float dot = 0.f;
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) dot = fmaf(t[ch][i], yv[ch][i], dot);
          const float mean = quad_sum(dot) * inv_c;
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) t[ch][i] = ps[wr][vh] * fmaf(-yv[ch][i], mean, t[ch][i]);
        }
    }
    if constexpr (MASK) {      // v *= bit ? slope : 1: sign-extended bit (v_bfe_i32), factor select (v_bfi_b32), multiply
      const uint32_t f1 = __float_as_uint(1.f), fs = __float_as_uint(a.mask_slope);
#pragma unroll
      for (int wr = 0; wr < 2; ++wr)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh) {
          const uint32_t wsh = mb[wr][vh] >> (4 * q4);
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              uint32_t t_, f_;
              asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t_) : "v"(wsh), "n"(16 * ch + i));
              asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(f_) : "v"(t_), "v"(fs), "v"(f1));      // (t & fs) | (~t & 1.0f)
              c[wr][vh][ch][i] *= __uint_as_float(f_);
            }
        }
    }
    if constexpr (!POOL) {
      float pwm[(RGB || PWB) ? 2 : 1][(RGB || PWB) ? 4 : 1];      // the eight matrix values of my channels
      float gws[PWB ? 2 : 1][PWB ? 4 : 1], gbs[PWB ? 2 : 1][PWB ? 4 : 1];      // (PWB) this plane's sums of x * g and g over my voxels
      if constexpr (RGB || PWB) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            pwm[ch][i] = pwm_lds[16 * ch + 4 * q4 + i];
            if constexpr (PWB) gws[ch][i] = gbs[ch][i] = 0.f;
          }
      }
      uint32_t ximg[PWB ? 2 : 1][PWB ? 2 : 1];      // (PWB) from_rgb's input at my four voxels
      if constexpr (PWB) {
#pragma unroll
        for (int wr = 0; wr < 2; ++wr)
#pragma unroll
          for (int vh = 0; vh < 2; ++vh)
            ximg[wr][vh] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rxiE, row_ok[wr] ? (uint32_t)(v16 * ES) : DEAD,
                                                                                    (tile_vox + vrel(wr, vh)) * (uint32_t)ES, 0);
      }
#pragma unroll
      for (int wr = 0; wr < 2; ++wr) {
        const uint32_t vo = row_ok[wr] ? yv0 : DEAD;
#pragma unroll
        for (int vh = 0; vh < 2; ++vh) {
          f32x4(&t)[2] = c[wr][vh];
          // 16 contiguous bytes per lane: rows 1 / 3 of the channel-half-0 registers swap with rows 0 / 2 of the channel-half-1
          // registers (v_permlane16_swap), after which lane q4 holds channels [0, 16, 8, 24][q4] .. + 7
          const uint32_t a0 = sg_pack_bf16(t[0][0], t[0][1]), a1 = sg_pack_bf16(t[0][2], t[0][3]);
          const uint32_t b0 = sg_pack_bf16(t[1][0], t[1][1]), b1 = sg_pack_bf16(t[1][2], t[1][3]);
          if constexpr (RGB || PWB) {
            // the pointwise layer next to this one works on the STORED values (rounded to bf16): to_rgb of the stage's output
            // (pgan/generator.py:96-97), or from_rgb's backward on the gradient of ITS output (pgan/discriminator.py:9-12:
            // image gradient = sum_c g_c m_c; filter gradient = sum_v x_v g_vc; bias gradient = sum_v g_vc) -- sg_conv3d_pw_bwd's sums
            const uint32_t pk[2][2] = {{a0, a1}, {b0, b1}};
            float dot = 0.f;
            const float xv = PWB ? __uint_as_float(ximg[wr][vh] << 16) : 0.f;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
              for (int h2 = 0; h2 < 2; ++h2) {
                const float lo = __uint_as_float(pk[ch][h2] << 16), hi = __uint_as_float(pk[ch][h2] & 0xFFFF0000u);
                dot = fmaf(lo, pwm[ch][2 * h2], dot);
                dot = fmaf(hi, pwm[ch][2 * h2 + 1], dot);
                if constexpr (PWB) {      // (a row beyond H: x loads as 0; its accumulators are not zero -- the row above feeds them)
                  gws[ch][2 * h2] = fmaf(xv, lo, gws[ch][2 * h2]);
                  gws[ch][2 * h2 + 1] = fmaf(xv, hi, gws[ch][2 * h2 + 1]);
                  gbs[ch][2 * h2] += row_ok[wr] ? lo : 0.f;
                  gbs[ch][2 * h2 + 1] += row_ok[wr] ? hi : 0.f;
                }
              }
            dot = quad_sum(dot) + rgb_b;
            const uint32_t img = sg_pack_bf16(dot, 0.f);
            if constexpr (RGB)
              __builtin_amdgcn_raw_buffer_store_b16((short)img, rgE, (row_ok[wr] && q4 == 0) ? (uint32_t)(v16 * ES) : DEAD,
                                                    (tile_vox + vrel(wr, vh)) * (uint32_t)ES, 0);
            else
              __builtin_amdgcn_raw_buffer_store_b16((short)img, rdxE, (row_ok[wr] && q4 == 0 && want_pw_dx) ? (uint32_t)(v16 * ES) : DEAD,
                                                    (tile_vox + vrel(wr, vh)) * (uint32_t)ES, 0);
          }
          if constexpr (PWB) continue;      // the gradient of from_rgb's output itself is needed by nobody else: not written
          const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          u32x4 out;
          out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
          __builtin_amdgcn_raw_buffer_store_b128(out, ryE, vo, ysoff + vrel(wr, vh) * (uint32_t)(cout * ES), 0);
          SG_STORE16_GUARD(out);
        }
      }
      if constexpr (PWB) {      // this plane's sums into my running sums
        f32x4 r0 = pwacc[0], r1 = pwacc[1], r2 = pwacc[2], r3 = pwacc[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          r0[i] += gws[0][i]; r1[i] += gws[1][i]; r2[i] += gbs[0][i]; r3[i] += gbs[1][i];
        }
        pwacc[0] = r0; pwacc[1] = r1; pwacc[2] = r2; pwacc[3] = r3;
      }
    } else {
      // fused downscale3d (pgan/discriminator.py:44 after conv_2 + bias + LeakyReLU).  POOL: the mean over the 2 x 1 x 2 (D x W)
      // block, the H pairs are left to sg_downscale_sum(1, 2, 1), output [n, D/2, H, W/2, cout]; POOL3: the wave owns both rows of
      // an H pair, so the whole 2 x 2 x 2 mean leaves the kernel, output [n, D/2, H/2, W/2, cout] -- no second pass over the pooled
      // tensor.  W neighbours are adjacent lanes (DPP quad_perm [1,0,3,2]); even lanes keep the pairs of voxel half 0, odd lanes
      // those of voxel half 1; the even plane of a D pair waits one phase in `hold`.
      f32x4 sall[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int wr = 0; wr < 2; ++wr) {
        f32x4 s[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float u0 = c[wr][0][ch][i] + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(c[wr][0][ch][i]), 0xB1, 0xF, 0xF, true));
            const float u1 = c[wr][1][ch][i] + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(c[wr][1][ch][i]), 0xB1, 0xF, 0xF, true));
            s[ch][i] = (v16 & 1) ? u1 : u0;
            sall[ch][i] += s[ch][i];
          }
        if constexpr (POOL3) {
          if (wr == 0) continue;      // (both rows first)
          s[0] = sall[0];
          s[1] = sall[1];
        }
        f32x4(&hd)[2] = hold[POOL3 ? 0 : wr];
        if ((o & 1) == 0) {   // uniform
          hd[0] = s[0];
          hd[1] = s[1];
        } else {
          float m[2][4];
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) m[ch][i] = (hd[ch][i] + s[ch][i]) * (POOL3 ? 0.125f : 0.25f);
          const uint32_t a0 = sg_pack_bf16(m[0][0], m[0][1]), a1 = sg_pack_bf16(m[0][2], m[0][3]);
          const uint32_t b0 = sg_pack_bf16(m[1][0], m[1][1]), b1 = sg_pack_bf16(m[1][2], m[1][3]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          u32x4 out;
          out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
          // pooled voxel (o / 2, h0 + wr [POOL3: h0 / 2], w0 / 2 + wq), wq = (v16 >> 1) + 8 * (v16 & 1)
          const uint32_t psoff = POOL3 ? (uint32_t)((((o >> 1) * (H >> 1) + (E.h0 >> 1)) * (W >> 1) + (E.w0 >> 1)) * cout * ES)
                                       : (uint32_t)((((o >> 1) * H + E.h0 + wr) * (W >> 1) + (E.w0 >> 1)) * cout * ES);
          const uint32_t pvo = (uint32_t)((((v16 >> 1) + 8 * (v16 & 1)) * cout + nt0 * 32) * ES) + cb * ES;
          __builtin_amdgcn_raw_buffer_store_b128(out, ryE, row_ok[POOL3 ? 0 : wr] ? pvo : DEAD, psoff, 0);
          SG_STORE16_GUARD(out);
        }
      }
    }
  };
  // ---- off-phase after the MFMAs of E's plane p: the next plane (requested one phase ago) goes to my halo buffer and the one after
  // is requested; output plane p - 1 (in cA) is complete and stored if this item owns it -- at the top of the volume plane p (cB) as
  // well; at the end of an item all three accumulators start afresh.
  auto off_phase = [&](f32x4 (&cA)[2][2][2], f32x4 (&cB)[2][2][2], f32x4 (&cC)[2][2][2]) __attribute__((always_inline)) {
    uint32_t mb[2][2] = {{mbn[0][0], mbn[0][1]}, {mbn[1][0], mbn[1][1]}};
    if (MASK) asm volatile("" : "+v"(mb[0][0]), "+v"(mb[0][1]), "+v"(mb[1][0]), "+v"(mb[1][1]));
    // (fair share, below: my phase count goes out and the partner's is requested here, and read at the end of the off-phase)
    ++myphase;
    int other = 0;
    asm volatile("ds_write_b32 %1, %2\n\tds_read_b32 %0, %3"
                 : "=v"(other) : "v"(W3_POFF + 4 * w8), "v"(myphase), "v"(W3_POFF + 4 * (w8 ^ 4)) : "memory");
    if (P.j < nitems) {
      if (!no_stage) store_plane();
      advance_P();
      if (P.j < nitems && !no_stage) load_plane(P.p);
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    const int p = E.p;
    const bool last = p == E.p_hi;
    const bool top = p == D - 1 && p < E.o_hi;
    if (p - 1 >= E.o_lo && !no_epi) epilogue(cA, p - 1, mb);
    if (top && !no_epi) {
      uint32_t mt[2][2] = {{0u, 0u}, {0u, 0u}};
      if constexpr (MASK) {      // (one phase per column: loaded where they are used)
        const uint32_t tv = (uint32_t)(p * plane_vox + colvoxE);
#pragma unroll
        for (int wr = 0; wr < 2; ++wr)
#pragma unroll
          for (int vh = 0; vh < 2; ++vh)
            mt[wr][vh] = __builtin_amdgcn_raw_buffer_load_b32(rmE, row_ok[wr] ? sv0 + vrel(wr, vh) * (uint32_t)(ntile * 4) : DEAD,
                                                              tv * (uint32_t)(ntile * 4), 0);
      }
      epilogue(cB, p, mt);
    }
    init_acc(cA);
    if (last) {
      init_acc(cB);
      init_acc(cC);
    }
    advance_E();
    __builtin_amdgcn_sched_barrier(0);
    request_mask();
    // Fair share of the matrix pipe.  The two waves of a SIMD (w8 and w8 ^ 4) are arbitrated by priority, then AGE: left alone the
    // older wave runs its K loops at the pipe's rate and the younger one gets the gaps (in-kernel stamps: 4.2k against 7-10k cycles
    // per K loop), finishes its columns long after the partner and then runs alone, the pipe idle during its off-phases.  Each wave
    // publishes its phase count; whoever is behind raises its priority for the next phase: the two stay within a phase of each other.
    // (LDS instructions by hand: through a volatile generic pointer these become flat accesses behind s_waitcnt vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(other)::"memory");
    if (!(a.dbg_flags & 4096)) {
      if (__builtin_amdgcn_readfirstlane(other) > myphase) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    }
    stamp();
  };

  enter_item_P();
  enter_item_E();
  load_plane(P.p);
  store_plane();        // the very first plane: latency exposed once per wave
  advance_P();
  if (P.j < nitems) load_plane(P.p);
  init_acc(acc[0]); init_acc(acc[1]); init_acc(acc[2]);
  request_mask();
  // accumulator roles at running phase t (ROT = t % 3): plane p - 1 in acc[(ROT + 2) % 3], p in acc[ROT], p + 1 in acc[(ROT + 1) % 3]
#define SG_W3_OFF(ROT) off_phase(acc[(ROT + 2) % 3], acc[ROT], acc[(ROT + 1) % 3])
  stamp();
  for (;;) {
    sg_kloop3w<0>::run(acc, xa, wl);
    stamp();
    SG_W3_OFF(0);
    if (E.j >= nitems) break;
    sg_kloop3w<1>::run(acc, xa, wl);
    stamp();
    SG_W3_OFF(1);
    if (E.j >= nitems) break;
    sg_kloop3w<2>::run(acc, xa, wl);
    stamp();
    SG_W3_OFF(2);
    if (E.j >= nitems) break;
  }
#undef SG_W3_OFF
  if constexpr (PWB) {
    // my partial sums over the 16 voxel columns of each lane row (lanes with equal q4), then one row of [2][32] per wave:
    // pw_wgrad_final_kernel (wgrad.hip) adds the rows in order -- reproducible, no atomics
    const f32x4 fin[4] = {pwacc[0], pwacc[1], pwacc[2], pwacc[3]};
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float sw = fin[ch][i], sb = fin[2 + ch][i];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          sw += __shfl_xor(sw, m);
          sb += __shfl_xor(sb, m);
        }
        if (v16 == 0) {
          float* row = a.pw_part + (size_t)(blockIdx.x * 8 + w8) * 64;
          row[16 * ch + 4 * q4 + i] = sw;
          row[32 + 16 * ch + 4 * q4 + i] = sb;
        }
      }
  }
}

// weight image for the v_mfma_f32_16x16x32_bf16 kernels: 1-KiB fragments [32-channel output tile][tap][32-channel input chunk]
// [output-channel half]; lane l of a fragment = A operand row (cout) 32 * nt + 16 * ch + (l & 15), K elements (cin)
// 32 * gi + 8 * (l >> 4) + e
struct Pack16Args {
  const float* w;
  bf16_t* out;
  float coef;
  int cin, cout, ngi, flip;
};
constexpr int SG_PACK16_BATCH = 32;
struct Pack16Batch {
  Pack16Args item[SG_PACK16_BATCH];
};
__global__ void pack_weights16_batch_kernel(Pack16Batch b) {
  const Pack16Args& a = b.item[blockIdx.y];
  const int ntile = a.cout >> 5;
  const int total = ntile * 27 * a.ngi * 2 * 64 * 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i & 7, lane = (i >> 3) & 63;
    int f = i >> 9;
    const int ch = f & 1;
    f >>= 1;
    const int gi = f % a.ngi;
    f /= a.ngi;
    const int tap = f % 27, nt = f / 27;
    const int co = 32 * nt + 16 * ch + (lane & 15), ci = 32 * gi + 8 * (lane >> 4) + e;
    const float v = !a.flip ? a.w[((int64_t)tap * a.cin + ci) * a.cout + co]
                            : a.w[((int64_t)(26 - tap) * a.cout + co) * a.cin + ci];   // as pack_weights_kernel (conv3d.hip)
    a.out[i] = (bf16_t)(v * a.coef);
  }
}

// the per-wave rows of the fused from_rgb backward, added in a fixed order (16 row groups x 8 independent sums each: the loads of
// one thread are 128 deep instead of 512 -- one block of 256 threads with a single running sum took 120 us for 512 KiB)
__global__ __launch_bounds__(1024) void pw_part_final_kernel(const float* __restrict__ part, int rows, float* __restrict__ dw,
                                                             float* __restrict__ dbias, float coef) {
  __shared__ float red[16][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = rg;
  for (; b + 7 * 16 < rows; b += 8 * 16) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += part[(size_t)(b + u * 16) * 64 + col];
  }
  for (int u = 0; b < rows; b += 16, ++u) s8[u & 7] += part[(size_t)b * 64 + col];
  red[rg][col] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  __syncthreads();
  if (rg == 0) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][col];
    if (col < 32) { if (dw) dw[col] = coef * t; }
    else if (dbias) dbias[col - 32] = t;
  }
}

template <int EPI>
int launch_fwd3w_inst(const Fwd3wArgs& fa, unsigned gx, hipStream_t st) {
  auto kern = conv_fwd3w_kernel<EPI>;
  SG_ALLOW_160K_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(gx, (unsigned)fa.a.ntile), dim3(512), (EPI & SG_EP_PWB) ? W3_LDS_PWB : W3_LDS, st, fa);
  return SG_OK;
}

}  // namespace

// The second weight image (16x16x32 fragments) of the layers the 16x16x32 kernels take: 3 x 3 x 3, bf16, whole 32-channel chunks
// and tiles; 64 -> 32 (conv_fwd3p16) and 32 -> 32k (conv_fwd3w).
size_t sg_pack16_bytes(const sg_conv_shape* s, sg_dtype dt) {
  if (dt != SG_BF16 || s->kd != 3 || s->kh != 3 || s->kw != 3 || (s->cout & 31) || s->cout < 32) return 0;
  if (!((s->cin == 64 && s->cout == 32) || s->cin == 32)) return 0;
  return (size_t)(s->cout >> 5) * 27 * (s->cin >> 5) * 2 * 1024;
}

int sg_pack16_batch(int n, const float* const* w, const float* coef, const int* flip, void* const* dst, const sg_conv_shape* shapes,
                    hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += SG_PACK16_BATCH) {
    const int m = n - i0 < SG_PACK16_BATCH ? n - i0 : SG_PACK16_BATCH;
    Pack16Batch b;
    for (int j = 0; j < SG_PACK16_BATCH; ++j) {
      const int i = i0 + (j < m ? j : 0);
      Pack16Args& p = b.item[j];
      p.w = w[i]; p.out = reinterpret_cast<bf16_t*>(dst[i]); p.coef = coef[i];
      p.cin = shapes[i].cin; p.cout = shapes[i].cout; p.ngi = shapes[i].cin >> 5; p.flip = flip[i] ? 1 : 0;
    }
    hipLaunchKernelGGL(pack_weights16_batch_kernel, dim3(54, (unsigned)m), dim3(256), 0, st, b);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

int sg_pw_wgrad_finalize(const float* part, int rows, float* dw, float* dbias, float coef, hipStream_t st) {
  if (!dw && !dbias) return SG_OK;
  hipLaunchKernelGGL(pw_part_final_kernel, dim3(1), dim3(1024), 0, st, part, rows, dw, dbias, coef);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// bf16, 3 x 3 x 3, 32 -> 32k channels, whole 32-wide rows.  Sets *used = false (and launches nothing) for anything else: the
// caller falls back to the sliding-halo kernel.
int sg_launch_fwd3w(ConvFwdArgs& a, const sg_conv_shape* s, hipStream_t st, bool* used, int* pw_rows) {
  *used = false;
  *pw_rows = 0;
  if (s->kd != 3 || s->kh != 3 || s->kw != 3 || s->cin != 32 || (s->cout & 31) || a.xcs != 32 || a.xco != 0) return SG_OK;
  if (s->upsample_in || a.in_mask || a.addend) return SG_OK;
  if (s->d < 2 || s->h < 8 || (s->w % 32) != 0) return SG_OK;
  if (a.pool && ((a.pool != 1 && a.pool != 3) || (s->d & 1) || (a.pool == 3 && ((s->h | s->w) & 1)) || a.pixel_norm || a.pnb_y || (a.mask_bits && (a.sign_out || a.bias || a.act)))) return SG_OK;
  if (a.pixel_norm && (a.mask_bits || a.ntile != 1 || a.pnb_y)) return SG_OK;
  if (a.mask_bits && a.sign_out) return SG_OK;
  if (a.pnb_y && (a.ntile != 1 || !a.mask_bits || a.sign_out || a.bias || a.act || !a.pnb_scale)) return SG_OK;
  if (a.rgb_out && (a.ntile != 1 || !a.rgb_w || a.pool || a.mask_bits || a.pnb_y || !a.pixel_norm || !a.sign_out)) return SG_OK;
  if (a.pw_x && (a.ntile != 1 || !a.pw_wmat || !a.pw_part || !a.mask_bits || a.pool || a.pnb_y || a.rgb_out || a.pixel_norm ||
                 a.sign_out || a.bias || a.act)) return SG_OK;
  {   // buffer addressing (rebased per sample): one sample of every tensor this kernel touches stays below 2 GiB
    const int64_t svox = (int64_t)s->d * s->h * s->w;
    if (svox * 64 >= (1ll << 31) || svox * s->cout * 2 >= (1ll << 31) || svox * a.ntile * 4 >= (1ll << 31)) return SG_OK;
  }
  Fwd3wArgs fa;
  fa.a = a;
  fa.a.g.D = s->d; fa.a.g.H = s->h; fa.a.g.W = s->w; fa.a.g.N = s->n;
  fa.a.wp = reinterpret_cast<const char*>(a.wp) + (size_t)a.nchunk * a.taps * a.ntile * 1024;      // the 16x16x32 fragment image follows the standard one
  fa.nHb = (s->h + 15) / 16;
  fa.nWb = s->w / 32;
  const int64_t ncol = (int64_t)s->n * fa.nHb * fa.nWb;
  if (ncol >= (1 << 24)) return SG_OK;
  int gx = (256 / a.ntile) / 8 * 8;
  if (gx < 8) gx = 8;
  // few columns (small batches): cut the columns along D so that every CU has work; a segment of L output planes runs L + 2
  // input planes, so L stays >= 4 (even: the D pairs of the pooled epilogue stay inside a segment)
  int nseg = 1;
  if (ncol < gx) {
    nseg = (int)((gx + ncol - 1) / ncol);
    int L = (s->d + nseg - 1) / nseg;
    if (L < 4) L = 4;
    L = (L + 1) & ~1;
    nseg = (s->d + L - 1) / L;
    fa.seglen = L;
  } else fa.seglen = s->d;
  fa.nseg = nseg;
  fa.nitems = (int)ncol * nseg;
  if (fa.nitems < gx) gx = (fa.nitems + 7) / 8 * 8;
  if (a.pw_x && (size_t)gx * 8 > (size_t)SG_PW_PART_ROWS) return SG_OK;
  const int epi = (a.sign_out ? SG_EP_SIGN : 0) | (a.mask_bits ? SG_EP_MASK : 0) | (a.pixel_norm ? SG_EP_PN : 0) |
                  (a.pool ? SG_EP_POOL : 0) | (a.pool == 3 ? SG_EP_POOL3 : 0) | (a.pnb_y ? SG_EP_PNB : 0) | (a.rgb_out ? SG_EP_RGB : 0) |
                  (a.pw_x ? SG_EP_PWB : 0);
  int rc = SG_OK;
  switch (epi) {
    case 0: rc = launch_fwd3w_inst<0>(fa, (unsigned)gx, st); break;
    case SG_EP_SIGN: rc = launch_fwd3w_inst<SG_EP_SIGN>(fa, (unsigned)gx, st); break;
    case SG_EP_MASK: rc = launch_fwd3w_inst<SG_EP_MASK>(fa, (unsigned)gx, st); break;
    case SG_EP_PN: rc = launch_fwd3w_inst<SG_EP_PN>(fa, (unsigned)gx, st); break;
    case SG_EP_PN | SG_EP_SIGN: rc = launch_fwd3w_inst<SG_EP_PN | SG_EP_SIGN>(fa, (unsigned)gx, st); break;
    case SG_EP_POOL: rc = launch_fwd3w_inst<SG_EP_POOL>(fa, (unsigned)gx, st); break;
    case SG_EP_SIGN | SG_EP_POOL: rc = launch_fwd3w_inst<SG_EP_SIGN | SG_EP_POOL>(fa, (unsigned)gx, st); break;
    case SG_EP_MASK | SG_EP_POOL: rc = launch_fwd3w_inst<SG_EP_MASK | SG_EP_POOL>(fa, (unsigned)gx, st); break;
    case SG_EP_POOL | SG_EP_POOL3: rc = launch_fwd3w_inst<SG_EP_POOL | SG_EP_POOL3>(fa, (unsigned)gx, st); break;
    case SG_EP_SIGN | SG_EP_POOL | SG_EP_POOL3: rc = launch_fwd3w_inst<SG_EP_SIGN | SG_EP_POOL | SG_EP_POOL3>(fa, (unsigned)gx, st); break;
    case SG_EP_MASK | SG_EP_POOL | SG_EP_POOL3: rc = launch_fwd3w_inst<SG_EP_MASK | SG_EP_POOL | SG_EP_POOL3>(fa, (unsigned)gx, st); break;
    case SG_EP_MASK | SG_EP_PNB: rc = launch_fwd3w_inst<SG_EP_MASK | SG_EP_PNB>(fa, (unsigned)gx, st); break;
    case SG_EP_PN | SG_EP_SIGN | SG_EP_RGB: rc = launch_fwd3w_inst<SG_EP_PN | SG_EP_SIGN | SG_EP_RGB>(fa, (unsigned)gx, st); break;
    case SG_EP_MASK | SG_EP_PWB: rc = launch_fwd3w_inst<SG_EP_MASK | SG_EP_PWB>(fa, (unsigned)gx, st); break;
    default: return SG_OK;
  }
  if (rc != SG_OK) return rc;
  SG_KNAME("conv_fwd3w<bf16,32->%d>", 32);
  SG_LAUNCH_CHECK();
  *pw_rows = a.pw_x ? gx * 8 : 0;
  *used = true;
  return SG_OK;
}
