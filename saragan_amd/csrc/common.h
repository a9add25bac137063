// Shared device/host helpers for the gfx950 kernels of libsaragan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/saragan_hip.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// gfx950 write-data hazard of 16-byte vector-memory stores (tools/probe/store_war_probe.hip, measured on MI355X): a
// VALU instruction that writes one of the store's data VGPRs in the issue slot right after the store -- or one slot
// later for global_store / literal-soffset buffer stores -- reaches memory instead of the stored value in lanes
// 12-15 of each 16-lane row (0.05-0.3 % of such stores) whenever other waves are issuing on the CU.  The compiler
// pads one slot for the global / literal forms and none for SGPR-soffset buffer stores, one short in both cases.
// Placed after a 16-byte store, this keeps the data registers allocated over two more wait states.
#define SG_STORE16_GUARD(v) asm volatile("s_nop 1" ::"v"(v))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define SG_LAUNCH_CHECK()                          \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline hipStream_t sg_st(sg_stream_t s) { return (hipStream_t)s; }

// Diagnostic switches (environment variables SG_*).  They are read ONCE, at the first launch that asks, into an
// immutable snapshot; sg_config_reload() (include/saragan_hip.h) builds a new snapshot for tools that flip a switch
// between launches.  Launch paths never call getenv.
struct sg_config {
  int fwd_lds, fwd_tg;                       // SG_FWD_LDS, SG_FWD_TG (0 = automatic)
  int fwd_v1, fwd_no_pw, fwd_no_dense;       // SG_FWD_V1, SG_FWD_NO_PW, SG_FWD_NO_DENSE
  int fwd_no_v3, fwd_no_v3s, fwd_no_v4, fwd_no_v5, fwd_no_ksplit;   // SG_FWD_NO_V3, SG_FWD_NO_V3S, SG_FWD_NO_V4, SG_FWD_NO_V5, SG_FWD_NO_KSPLIT
  int fwd3p_16;                              // SG_FWD3P_16 (default 1): the one-pass 64 -> 32 kernel on v_mfma_f32_16x16x32_bf16; 0: the 32x32x16 form
  int fwd3s_16;                              // SG_FWD3S_16 (default 1): 32 -> 32k layers on conv_fwd3w (16x16x32, wave-private planes); 0: the sliding-halo kernel
  int fwd_no_3p;                             // SG_FWD_NO_3P: 64 -> 32 layers through the two-pass K split (A/B, tests)
  int fwd3_gx, fwd3_no_lean;                 // SG_FWD3_GX (0 = automatic), SG_FWD3_NO_LEAN
  int fwd4_gx, fwd4_no_lean, fwd4_no_wres;   // SG_FWD4_GX (0 = automatic), SG_FWD4_NO_LEAN, SG_FWD4_NO_WRES
  int wgrad_v1, wgrad_no_v3, wgrad_no_lean;  // SG_WGRAD_V1, SG_WGRAD_NO_V3, SG_WGRAD_NO_LEAN
  int wgrad_no_w16;                          // SG_WGRAD_NO_W16: 16-wide levels back on conv_wgrad2 (diagnostic)
  int wgrad3l_min_cols;                      // SG_WGRAD3L_MIN_COLS: fewest tile columns the sliding-halo weight gradient takes (0: 2)
  int wgrad3l_16;                            // SG_WGRAD3L_16: the sliding-halo weight gradient on v_mfma_f32_16x16x32_bf16
  int wgrad_v1_blocks;                       // SG_WGRAD_V1_BLOCKS: block target of the generic weight-gradient kernel (0: default)
  int dbg_flags;                             // SG_DBG_FLAGS
  int no_small;                              // SG_NO_SMALL: the small-channel 2-D layers through the MFMA kernels (A/B, tests)
  int deterministic;                         // SG_DETERMINISTIC: no float atomics anywhere (weight-gradient slabs, ordered sums)
  int gemm_ks_model;                         // SG_GEMM_KS_MODEL: K split of the 1x3x3 levels from the partial-tile cost model too (experiment)
  int gemm_k333_maxvox;                      // SG_GEMM_K333_MAXVOX: batch voxels up to which the 3x3x3 layers of the 4x16x16 level take the GEMM tiling
  int no_gemm;                               // SG_NO_GEMM: the low-resolution levels through the spatial kernels (A/B, tests)
};
const sg_config& sg_cfg();

// gemm.hip: GEMM-tiled forward / data-gradient convolution of the low-resolution levels (<= 16 x 16 planes, >= 128 channels)
bool sg_gemm_conv_eligible(const sg_conv_shape* s, sg_dtype dt);
size_t sg_gemm_conv_workspace(const sg_conv_shape* s, sg_dtype dt);
int sg_gemm_conv_fwd(const void* x, const void* wp, void* y, const sg_conv_shape* s, const float* bias, int act, float slope,
                     const uint32_t* mask_bits, float mask_slope, uint32_t* sign_out, void* workspace, size_t workspace_bytes,
                     hipStream_t st, bool* used);

// small.hip: VALU kernels for the 4 / 8 / 16-channel 1x3x3 layers of the 2-D pgan's top levels
bool sg_small_eligible(const sg_conv_shape* s);
bool sg_small_wgrad_eligible(const sg_conv_shape* s);
size_t sg_small_tail_bytes(const sg_conv_shape* s);
size_t sg_small_wgrad_workspace(const sg_conv_shape* s);
int sg_small_pack(const float* w, float coef, int flip, void* tail, const sg_conv_shape* s, sg_dtype dt, hipStream_t st);
int sg_small_fwd(const void* x, const void* tail, void* y, const sg_conv_shape* s, const float* bias, int act, float slope,
                 int pixel_norm, float eps, float* pn_scale, const uint32_t* mask_bits, float mask_slope, uint32_t* sign_out,
                 sg_dtype dt, hipStream_t st);
int sg_small_wgrad(const void* x, const void* dy, float* dw, float* dbias, float coef, void* workspace, size_t workspace_bytes,
                   const sg_conv_shape* s, sg_dtype dt, hipStream_t st);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel symbol (idempotent; thread-safe).
#define SG_ALLOW_160K_LDS(kern)                                                                                  \
  do {                                                                                                           \
    static std::once_flag once__;                                                                                \
    static hipError_t err__ = hipSuccess;                                                                        \
    std::call_once(once__, [&] {                                                                                 \
      err__ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
    });                                                                                                          \
    if (err__ != hipSuccess) return (int)err__;                                                                  \
  } while (0)
static inline size_t sg_esize(sg_dtype dt) { return dt == SG_BF16 ? 2 : 4; }
static inline bool sg_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- exact unsigned division by a small runtime constant (x < 2^16, d < 2^16) -------------------
struct sg_fastdiv {
  uint32_t d, m;
};
static inline __host__ __device__ sg_fastdiv sg_make_fastdiv(uint32_t d) {
  sg_fastdiv f;
  f.d = d;
  f.m = d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d);
  return f;
}
__device__ __forceinline__ uint32_t sg_div(uint32_t x, sg_fastdiv f) {
  return f.d <= 1 ? x : __umulhi(x, f.m);
}

template <typename T>
struct sg_traits;
template <>
struct sg_traits<float> {
  static constexpr int CH = 8;  // elements per 32-byte K chunk
  __device__ static __forceinline__ float to_f(float v) { return v; }
  __device__ static __forceinline__ float from_f(float v) { return v; }
};
template <>
struct sg_traits<bf16_t> {
  static constexpr int CH = 16;
  __device__ static __forceinline__ float to_f(bf16_t v) { return (float)v; }
  __device__ static __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }
};

// One 32-byte K chunk = (A 16 B/lane) x (B 16 B/lane): bf16 -> one 32x32x16 MFMA, f32 -> four 32x32x2.
// Within a chunk, lane half h owns elements [h*CH/2, (h+1)*CH/2) of the chunk for both operands.
template <typename T>
__device__ __forceinline__ f32x16 sg_mfma_chunk(u32x4 a, u32x4 b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 sg_mfma_chunk<bf16_t>(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                 c, 0, 0, 0);
}
// In-place form for the unrolled ping-pong loops (bf16): the accumulator is tied to one register tuple.  With the
// builtin the compiler may write the product to another tuple (D != C) and, at the joins of the per-variant code paths,
// copies 16-32 accumulator registers per phase behind the MFMA result hazard (s_nop 11 + v_mov runs: ~250 cycles of a
// 4k-cycle phase).  Hazards are the caller's: no VALU may read `c` within 18 issue slots of the last call (the loops end
// with sg_mfma_drain), and `c` must not have been written by a VALU instruction in the two slots before the first.
__device__ __forceinline__ void sg_mfma_bf16_acc(f32x16& c, u32x4 a, u32x4 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// 18 wait states after the last sg_mfma_bf16_acc on these accumulators; tied to them so that no compiler-generated
// read (a register copy at a join, say) can be scheduled in front of it
template <int N>
__device__ __forceinline__ void sg_mfma_drain(f32x16 (&acc)[N]) {
  static_assert(N == 2, "extend the operand list");
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
}
// The unrolled K loops below issue their LDS fragment reads and (bf16) their MFMAs through inline asm and synchronise them
// with hand-counted `s_waitcnt lgkmcnt(N > 0)`: that count is only right while NOTHING else that lgkmcnt counts (LDS,
// SMEM -- which returns out of order --, s_memtime, messages) is outstanding when the prologue starts or is emitted by the
// compiler inside the loop.  SG_KLOOP_BEGIN drains the counter first; the assembler comments delimit the region for
// tests/test_build_resources.py, which disassembles the build and fails on any compiler-emitted LGKM instruction between
// them and on any instruction of the compiler's that touches an accumulator of the tied MFMAs before the drain.
// s_waitcnt lgkmcnt(N) as an instruction the compiler SEES (vmcnt / expcnt fields at their maxima: not waited for).  Between two
// inline-asm statements that touch the same registers -- an in-place MFMA and the next one on that accumulator -- hipcc pads one
// wait state (`s_nop 0`) unless an instruction of its own stands between them; an inline-asm s_waitcnt does not count, this one
// does: 0.54 s_nop per MFMA left the sliding-halo loops (an s_nop costs a 4-cycle issue slot in a loop that has 8 per MFMA).
#define SG_WAIT_LGKM(N) __builtin_amdgcn_s_waitcnt(0xC07F | ((N) << 8))
#define SG_KLOOP_BEGIN() asm volatile("; SG_KLOOP_BEGIN\n\ts_waitcnt lgkmcnt(0)")
#define SG_KLOOP_END() asm volatile("; SG_KLOOP_END")
template <>
__device__ __forceinline__ f32x16 sg_mfma_chunk<float>(u32x4 a, u32x4 b, f32x16 c) {
  f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], c, 0, 0, 0);
  return c;
}

// ---- LeakyReLU sign words.  A lane holds, of voxel r and N tile nt, the 16 channels (i&3) + 8*(i>>2) + 4*hh; its
// partner lane r+32 holds the other 16, so one xor-32 shuffle completes the voxel's 32-bit word.
// Written with shifts by inline constants only: literal masks (1u << k) would each occupy a register for the whole
// persistent loop.  The sign BIT is used (x < 0 up to the sign of zero, which no accumulator path here produces).
__device__ __forceinline__ uint32_t sg_sign_word(const f32x16& v, int hh) {
  uint32_t b = 0u;
#pragma unroll
  for (int i = 0; i < 16; ++i) b |= (__float_as_uint(v[i]) >> 31) << ((i & 3) + 8 * (i >> 2));
  b <<= 4 * hh;
  return b | (uint32_t)__shfl_xor((int)b, 32);
}

__device__ __forceinline__ void sg_apply_sign_word(f32x16& v, uint32_t word, int hh, float slope) {
  // v *= bit ? slope : 1, as v += bit ? (slope - 1) * v : 0: a packed multiply per pair, then one bit-field extract
  // (0 or -1), one AND and one add per element.  (The off-phase shares its SIMD's vector issue with the other wave's
  // MFMAs: ~6 VALU slots per MFMA, so the epilogue's instruction count is what has to fit under the MFMA phase.)
  const uint32_t wsh = word >> (4 * hh);
  const float sm1 = slope - 1.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = ((int)(wsh << (31 - ((i & 3) + 8 * (i >> 2))))) >> 31;   // 1-bit signed field: 0 or -1
    v[i] += __uint_as_float((uint32_t)t & __float_as_uint(v[i] * sm1));
  }
}

// ---- 16 contiguous bytes per lane for the bf16 store of a 32-channel tile row.  After the MFMA lane (r, hh) holds
// channels 8*qd + 4*hh + e; v_permlane32_swap exchanges the qd-odd part of the lower half-wave with the qd-even part
// of the upper one, after which lane hh = 0 holds channels 16j + 0..7 and hh = 1 channels 16j + 8..15: two
// dwordx4 stores per M tile instead of four dwordx2 (the off-phase is bound by vector-memory instructions).
// LeakyReLU as max(x, slope * x), slope <= 1 (slope 1: identity).  v_max directly: fmaxf() adds a canonicalising
// v_max(x, x) per element, a third of the epilogue's arithmetic.
__device__ __forceinline__ float sg_lrelu(float x, float slope) {
  const float t = x * slope;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(t));
  return r;
}

__device__ __forceinline__ uint32_t sg_pack_bf16(float lo, float hi) {   // one v_cvt_pk_bf16_f32 (round to nearest even, as from_f)
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}

static inline bool sg_is_pow2f(float g) {      // a positive normal power of two: the mantissa bits are zero
  uint32_t u;
  __builtin_memcpy(&u, &g, 4);
  return g > 0.f && (u & 0x007FFFFFu) == 0 && (u >> 23) != 0 && (u >> 23) != 255;
}

// One staged 16-byte piece (8 bf16 channels) of a masked nearest up-scale: bf16(x * gain [* slope where the sign bit of
// the fine voxel's channel is set]), bit e of m = channel e of the piece.  `gain` is a power of two (host-checked), so
// x * (gain * slope) is bit for bit (x * gain) * slope, the arithmetic of sg_upscale2x_masked (elementwise.hip): a gather
// fused into a consumer is bit-identical to the tensor that kernel would write.  Per element: unpack, sign-extended bit
// (v_bfe_i32), factor select (v_bfi_b32), multiply -- 4.5 VALU with the pack, where and + cmp + cndmask + two multiplies
// were 6.5 (the mask arithmetic sets the off-phase of the gathered K-split passes).
__device__ __forceinline__ u32x4 sg_mask_piece_bf16(u32x4 v, uint32_t m, float gain, float slope) {
  const uint32_t fg = __float_as_uint(gain), fs = __float_as_uint(gain * slope);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    // (inline asm: written as C the compiler canonicalises the pair back to v_and + v_cmp + v_cndmask)
    uint32_t t0, t1, f0, f1;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t0) : "v"(m), "n"(2 * e));
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t1) : "v"(m), "n"(2 * e + 1));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(f0) : "v"(t0), "v"(fs), "v"(fg));      // (t & fs) | (~t & fg)
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(f1) : "v"(t1), "v"(fs), "v"(fg));
    const float lo = __uint_as_float(v[e] << 16) * __uint_as_float(f0);
    const float hi = __uint_as_float(v[e] & 0xFFFF0000u) * __uint_as_float(f1);
    o[e] = sg_pack_bf16(lo, hi);
  }
  return o;
}

__device__ __forceinline__ void sg_store_tile_row_bf16(bf16_t* row32, const f32x16& v, int hh, bool ok) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint32_t a0 = sg_pack_bf16(v[8 * j + 0], v[8 * j + 1]), a1 = sg_pack_bf16(v[8 * j + 2], v[8 * j + 3]);
    const uint32_t b0 = sg_pack_bf16(v[8 * j + 4], v[8 * j + 5]), b1 = sg_pack_bf16(v[8 * j + 6], v[8 * j + 7]);
    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    u32x4 out;
    out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
    if (ok) *reinterpret_cast<u32x4*>(row32 + 16 * j + 8 * hh) = out;
    SG_STORE16_GUARD(out);
  }
}


// Tile geometry shared by the conv forward and weight-gradient kernels.  A block owns TN x TD x TH x TW
// output voxels and stages their (TD+2PD) x (TH+2PH) x (TW+2PW) input halo per sample in LDS.
struct sg_tile_geom {
  int32_t N, D, H, W;      // output extent
  int32_t TN, TD, TH, TW;  // tile extent
  int32_t nTn, nTd, nTh, nTw;
  int32_t PD, PH, PW;      // halo per side
  int32_t HD, HH, HW;      // halo-tile extent
  int32_t ups;             // 1: input tensor is half resolution (nearest x2 gather)
  sg_fastdiv fTW, fTH, fTD;        // tile-voxel decomposition
  sg_fastdiv fHW, fHH, fHD;        // halo-voxel decomposition
  sg_fastdiv fnTw, fnTh, fnTd;     // block -> tile decomposition
};

static inline int sg_cdiv(int a, int b) { return (a + b - 1) / b; }

static inline int sg_pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Picks the tile (powers of two clipped to the extent, TW <= 32, at most `bm` voxels) that stages the
// fewest halo voxels over the whole tensor; small volumes fold batch samples into the tile (TN > 1).
static inline sg_tile_geom sg_make_geom(const sg_conv_shape* s, int bm, bool prefer_w32 = false, int force_td = 0,
                                        int force_th = 0) {
  sg_tile_geom g;
  g.N = s->n; g.D = s->d; g.H = s->h; g.W = s->w;
  g.PD = s->kd / 2; g.PH = s->kh / 2; g.PW = s->kw / 2;
  double best = 1e300;
  int bd = 1, bh = 1, bw = 1;
  const int mw = sg_pow2ceil(s->w) < 32 ? sg_pow2ceil(s->w) : 32;
  for (int tw = mw; tw >= 1; tw >>= 1)
    for (int th = 1; th <= sg_pow2ceil(s->h) && tw * th <= bm; th <<= 1)
      for (int td = 1; td <= sg_pow2ceil(s->d) && tw * th * td <= bm; td <<= 1) {
        const int cw = tw < s->w ? tw : s->w, chh = th < s->h ? th : s->h, cd = td < s->d ? td : s->d;
        const double tiles = (double)sg_cdiv(s->w, cw) * sg_cdiv(s->h, chh) * sg_cdiv(s->d, cd);
        const double halo = (double)(cw + 2 * g.PW) * (chh + 2 * g.PH) * (cd + 2 * g.PD);
        // staged voxels + a per-tile cost that favours full tiles (fixed barrier/epilogue overhead)
        double score = tiles * (halo + 0.25 * bm) - 1e-3 * cw;
        // v2 kernels read 32 consecutive halo rows per MFMA operand: only tiles spanning min(W,32) are conflict free
        if (prefer_w32 && cw != (s->w < 32 ? s->w : 32)) score *= 4.0;
        if (score < best) { best = score; bd = cd; bh = chh; bw = cw; }
      }
  if (force_td > 0 && force_th > 0) {   // caller-imposed tile (the sliding-halo kernel walks TD = 2 planes at a time)
    bd = force_td < s->d ? force_td : s->d;
    bh = force_th < s->h ? force_th : s->h;
    bw = s->w < 32 ? s->w : 32;
  }
  g.TW = bw; g.TH = bh; g.TD = bd;
  int rem = bm / (g.TW * g.TH * g.TD); if (rem < 1) rem = 1;
  g.TN = s->n < rem ? s->n : rem;
  g.nTn = sg_cdiv(s->n, g.TN); g.nTd = sg_cdiv(s->d, g.TD);
  g.nTh = sg_cdiv(s->h, g.TH); g.nTw = sg_cdiv(s->w, g.TW);
  g.HD = g.TD + 2 * g.PD; g.HH = g.TH + 2 * g.PH; g.HW = g.TW + 2 * g.PW;
  g.ups = s->upsample_in ? 1 : 0;
  g.fTW = sg_make_fastdiv(g.TW); g.fTH = sg_make_fastdiv(g.TH); g.fTD = sg_make_fastdiv(g.TD);
  g.fHW = sg_make_fastdiv(g.HW); g.fHH = sg_make_fastdiv(g.HH); g.fHD = sg_make_fastdiv(g.HD);
  g.fnTw = sg_make_fastdiv(g.nTw); g.fnTh = sg_make_fastdiv(g.nTh); g.fnTd = sg_make_fastdiv(g.nTd);
  return g;
}

struct sg_tile_origin {
  int32_t n0, d0, h0, w0;
};

// tile index (< 2^16 per fastdiv contract is checked on the host) -> tile origin
__device__ __forceinline__ sg_tile_origin sg_tile_of(const sg_tile_geom& g, uint32_t t) {
  sg_tile_origin o;
  uint32_t q = sg_div(t, g.fnTw);
  o.w0 = (int)(t - q * g.nTw) * g.TW;
  uint32_t q2 = sg_div(q, g.fnTh);
  o.h0 = (int)(q - q2 * g.nTh) * g.TH;
  uint32_t q3 = sg_div(q2, g.fnTd);
  o.d0 = (int)(q2 - q3 * g.nTd) * g.TD;
  o.n0 = (int)q3 * g.TN;
  return o;
}

// Stages channels [c0, c0 + nb*16/sizeof(T)) of the tile's input halo into LDS rows of `rs` bytes
// (row = halo voxel, zero outside the volume / beyond cin).  16-byte pieces; `vec_ok` says that
// cin*sizeof(T) is a multiple of 16 so that global 16-byte loads are aligned.
template <typename T>
__device__ __forceinline__ void sg_stage_halo(char* lds, int rs, const T* __restrict__ x, const sg_tile_geom& g,
                                              const sg_tile_origin& o, int cin, int c0, int npieces,
                                              sg_fastdiv fnp, bool vec_ok, int tid, int nthreads) {
  constexpr int EPP = 16 / (int)sizeof(T);  // elements per 16-byte piece
  const int hv = g.TN * g.HD * g.HH * g.HW;
  const int total = hv * npieces;
  const int Di = g.ups ? (g.D >> 1) : g.D, Hi = g.ups ? (g.H >> 1) : g.H, Wi = g.ups ? (g.W >> 1) : g.W;
  for (int it = tid; it < total; it += nthreads) {
    uint32_t v = sg_div((uint32_t)it, fnp);
    int piece = it - (int)v * npieces;
    uint32_t q = sg_div(v, g.fHW);
    int hw = (int)(v - q * g.HW);
    uint32_t q2 = sg_div(q, g.fHH);
    int hh = (int)(q - q2 * g.HH);
    uint32_t q3 = sg_div(q2, g.fHD);
    int hd = (int)(q2 - q3 * g.HD);
    int n = o.n0 + (int)q3;
    int d = o.d0 + hd - g.PD, h = o.h0 + hh - g.PH, w = o.w0 + hw - g.PW;
    u32x4 val = {0u, 0u, 0u, 0u};
    const int c = c0 + piece * EPP;
    if (n < g.N && (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W &&
        c < cin) {
      if (g.ups) { d >>= 1; h >>= 1; w >>= 1; }
      const T* src = x + ((((int64_t)n * Di + d) * Hi + h) * Wi + w) * (int64_t)cin + c;
      if (vec_ok && c + EPP <= cin) {
        val = *reinterpret_cast<const u32x4*>(src);
      } else {
        T tmp[EPP];
#pragma unroll
        for (int e = 0; e < EPP; ++e) tmp[e] = (c + e < cin) ? src[e] : sg_traits<T>::from_f(0.f);
        val = *reinterpret_cast<u32x4*>(tmp);
      }
    }
    *reinterpret_cast<u32x4*>(lds + (size_t)v * rs + piece * 16) = val;
  }
}
