// HBM-bound NDHWC kernels of the pgan step (gfx950): bias + LeakyReLU, pixel-norm, nearest x2 up / 2x2x2 sum
// down, fade-in lerp, instance noise, the per-(n,w) gradient-penalty reduction, minibatch-stddev, casts.
// Reference ops: SURFGAN_3D/networks/ops.py:130-136,167-192,250-325; networks/loss.py:122-123,140.
// All of them move 16 bytes per lane per access (8 bf16 / 4 f32) when the channel count allows it and
// fall back to element-wise access otherwise.
#include "common.h"

namespace {

template <typename T>
struct Piece {
  static constexpr int E = 16 / (int)sizeof(T);
  float v[E];
  __device__ __forceinline__ void load(const T* p) {
    u32x4 raw = *reinterpret_cast<const u32x4*>(p);
    const T* t = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = sg_traits<T>::to_f(t[e]);
  }
  __device__ __forceinline__ void store(T* p) const {
    u32x4 raw;
    T* t = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int e = 0; e < E; ++e) t[e] = sg_traits<T>::from_f(v[e]);
    *reinterpret_cast<u32x4*>(p) = raw;
    SG_STORE16_GUARD(raw);
  }
  // non-temporal form for write-dominated streams (the masked up-scale writes 8x what it reads: 3.8-4.0 -> 4.3-6.4 TB/s,
  // tools/ew_roofline.py; read-dominated kernels gain a few per cent at most and the whole step nothing measurable)
  __device__ __forceinline__ void store_nt(T* p) const {
    u32x4 raw;
    T* t = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int e = 0; e < E; ++e) t[e] = sg_traits<T>::from_f(v[e]);
    __builtin_nontemporal_store(raw, reinterpret_cast<u32x4*>(p));
    SG_STORE16_GUARD(raw);
  }
};

// sign words (include/saragan_hip.h): the E <= 8 channels of one 16-byte piece starting at channel c0 (c0 % E == 0)
// lie in one word; bit e of the result belongs to channel c0 + e
__device__ __forceinline__ uint32_t piece_signs(const uint32_t* __restrict__ words, int64_t vox, int nw, int c0) {
  return words[vox * nw + (c0 >> 5)] >> (c0 & 31);
}

inline int grid_for(int64_t items, int per_block = 256, int cap = 2048) {
  int64_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// Grid of a streaming kernel sized in TRIPS per block (grid-stride loop, no per-block partial results; never fewer blocks
// than the 2048 a full chip holds).  Swept over this file's kernels at their in-step shapes (2 / 4 / 8 / 16 trips against
// the capped grids): sg_axpby gains 10 % with two trips (5.2-5.4 -> 5.9-6.0 TB/s); the others stay within noise of their
// capped grids, which they keep.  (The write-dominated pointwise forward is the kernel this matters for: conv3d.hip.)
inline int grid_trips(int64_t items, int per_block, int trips) {
  int64_t full = (items + per_block - 1) / per_block;
  if (full < 1) full = 1;
  int64_t b = (full + trips - 1) / trips;
  if (b < 2048) b = full < 2048 ? full : 2048;
  if (b > (1 << 24)) b = 1 << 24;
  return (int)b;
}

// ---------------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ void bias_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias, T* __restrict__ y,
                                    int64_t nvox, int c, int act, float slope) {
  constexpr int E = Piece<T>::E;
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC) {
    const int P = c / E;
    const int64_t total = nvox * P;
    for (int64_t i = tid0; i < total; i += stride) {
      const int c0 = (int)(i % P) * E;
      Piece<T> p;
      p.load(x + i * E);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float v = p.v[e] + (bias ? bias[c0 + e] : 0.f);
        p.v[e] = act ? fmaxf(v, v * slope) : v;
      }
      p.store(y + i * E);
    }
  } else {
    const int64_t total = nvox * c;
    for (int64_t i = tid0; i < total; i += stride) {
      float v = sg_traits<T>::to_f(x[i]) + (bias ? bias[(int)(i % c)] : 0.f);
      y[i] = sg_traits<T>::from_f(act ? fmaxf(v, v * slope) : v);
    }
  }
}

// dx = mask(y) * dy, per-block partial channel sums of dx -> part[block][c].  Thread t of a block owns
// channel piece (t % P) (P | 256) so partial sums stay in registers; rows are strided by 256/P.
template <typename T>
__global__ __launch_bounds__(256) void bias_act_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                               T* __restrict__ dx, float* __restrict__ part,
                                                               int64_t nvox, int c, float slope,
                                                               const uint32_t* __restrict__ words) {
  constexpr int E = Piece<T>::E;
  __shared__ float red[256 * E];
  // more than 256 pieces per row (dense layers, c = 8192): blockIdx.y selects a 256-piece channel slice
  const int Pall = c / E;
  const int P = Pall > 256 ? 256 : Pall;   // divides 256
  const int pofs = blockIdx.y * 256;
  const int rows = 256 / P;       // voxel rows per block step
  const int p = pofs + threadIdx.x % P, rr = threadIdx.x / P;
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * rows + rr; v < nvox; v += (int64_t)gridDim.x * rows) {
    const int64_t off = v * c + (int64_t)p * E;
    Piece<T> g;
    g.load(dy + off);
    if (y) {
      Piece<T> yy;
      yy.load(y + off);
#pragma unroll
      for (int e = 0; e < E; ++e) g.v[e] = yy.v[e] >= 0.f ? g.v[e] : g.v[e] * slope;
    } else if (words) {
      const uint32_t sw = piece_signs(words, v, (c + 31) >> 5, p * E);
#pragma unroll
      for (int e = 0; e < E; ++e) g.v[e] = ((sw >> e) & 1u) ? g.v[e] * slope : g.v[e];
    }
    if (dx) g.store(dx + off);
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] += g.v[e];
  }
  if (part) {
#pragma unroll
    for (int e = 0; e < E; ++e) red[threadIdx.x * E + e] = s[e];
    __syncthreads();
    if (threadIdx.x < P) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float t = 0.f;
        for (int k = 0; k < rows; ++k) t += red[(k * P + threadIdx.x) * E + e];
        part[(int64_t)blockIdx.x * c + (pofs + threadIdx.x) * E + e] = t;
      }
    }
  }
}

template <typename T>
__global__ void bias_act_bwd_scalar_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                           float* __restrict__ part, int64_t nvox, int c, float slope,
                                           const uint32_t* __restrict__ words) {
  // generic fallback: one thread per channel column segment; correct for any c, not tuned
  const int64_t rows_per_block = (nvox + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t v1 = v0 + rows_per_block < nvox ? v0 + rows_per_block : nvox;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    float s = 0.f;
    for (int64_t v = v0; v < v1; ++v) {
      float g = sg_traits<T>::to_f(dy[v * c + ch]);
      if (y) g = sg_traits<T>::to_f(y[v * c + ch]) >= 0.f ? g : g * slope;
      else if (words && ((words[v * ((c + 31) >> 5) + (ch >> 5)] >> (ch & 31)) & 1u)) g *= slope;
      if (dx) dx[v * c + ch] = sg_traits<T>::from_f(g);
      s += g;
    }
    if (part) part[(int64_t)blockIdx.x * c + ch] = s;
  }
}

// c <= 8 channels that do not form 16-byte pieces (to_rgb / from_rgb sides, c = 1..7): one voxel per thread per
// step, channel sums in registers, one LDS tree per block.
template <typename T>
__global__ __launch_bounds__(256) void bias_act_bwd_small_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                                 T* __restrict__ dx, float* __restrict__ part,
                                                                 int64_t nvox, int c, float slope,
                                                                 const uint32_t* __restrict__ words) {
  __shared__ float red[4][8];
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256) {
    const uint32_t sw = words ? words[v] : 0u;     // c <= 8 < 32: one word per voxel
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e < c) {
        float g = sg_traits<T>::to_f(dy[v * c + e]);
        if (y) g = sg_traits<T>::to_f(y[v * c + e]) >= 0.f ? g : g * slope;
        else if ((sw >> e) & 1u) g *= slope;
        if (dx) dx[v * c + e] = sg_traits<T>::from_f(g);
        s[e] += g;
      }
    }
  }
  if (part) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s[e] += __shfl_xor(s[e], o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][e] = s[e];
    }
    __syncthreads();
    if (threadIdx.x < c) part[(int64_t)blockIdx.x * c + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] +
                                                                      red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

template <typename T>
__global__ void sign_words_kernel(const T* __restrict__ t, uint32_t* __restrict__ words, int64_t nvox, int c) {
  const int nw = (c + 31) >> 5;
  const int64_t total = nvox * nw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / nw;
    const int c0 = (int)(i - v * nw) * 32;
    uint32_t b = 0u;
    for (int j = 0; j < 32 && c0 + j < c; ++j) b |= (sg_traits<T>::to_f(t[v * c + c0 + j]) < 0.f) ? (1u << j) : 0u;
    words[i] = b;
  }
}

__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               int nb, int c) {
  // one block per 32 channels: 32 row-groups x 32 channels (coalesced 128-byte rows, 4 independent loads in flight
  // per thread), LDS tree at the end.  The previous 8-row-group version took 24-39 us for <= 256 KiB.
  __shared__ float red[32][33];
  const int ch = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (ch < c) {
    int b = rg;
    for (; b + 96 < nb; b += 128) {
      s0 += part[(int64_t)b * c + ch];
      s1 += part[(int64_t)(b + 32) * c + ch];
      s2 += part[(int64_t)(b + 64) * c + ch];
      s3 += part[(int64_t)(b + 96) * c + ch];
    }
    for (; b < nb; b += 32) s0 += part[(int64_t)b * c + ch];
  }
  red[rg][threadIdx.x & 31] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && ch < c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
    out[ch] = t;
  }
}

// ---------------------------------------------------------------------------------------------------
// pixel norm: a team of TP lanes (power of two <= 64) owns one voxel; KP pieces per lane (1 up to 512 bf16 channels,
// else 4) and U voxels per loop trip: the loads of all U voxels are issued before the first reduction, which is what
// the kernel needs to keep enough bytes in flight (one voxel per trip reached 2.8 of 6.3 TB/s).
template <typename T, bool BWD, int KP, int U, int CS = 0, bool WG = false>
__global__ __launch_bounds__(256) void pixel_norm_kernel(const T* __restrict__ a, const T* __restrict__ yv,
                                                         const float* __restrict__ scale_in, T* __restrict__ out,
                                                         float* __restrict__ scale_out, int64_t nvox, int c, int tp,
                                                         float eps, const uint32_t* __restrict__ words = nullptr,
                                                         float slope = 0.f, float* __restrict__ part = nullptr,
                                                         const float* __restrict__ wsm = nullptr) {
  // WG (CS > 0, with `part`): the pointwise convolution's own weight and bias gradient from the same read of y and `a` --
  // per block, behind the c channel sums: CS rows of sum_v a[v][j] * y[v][ch], then the CS sums of a[v][j]
  static_assert(!WG || (CS > 0 && KP == 1 && BWD), "weight gradient of the pointwise convolution: fused-gradient form only");
  // BWD with `words`: the LeakyReLU backward of the layer (mask from its sign words) is applied to the result and
  // `part` receives per-block channel sums (the bias gradient): pixel_norm(act(z + b)) differentiated in one pass.
  // CS > 0 (KP == 1): the incoming gradient is not a tensor but the product of a CS-channel tensor `a` [nvox][CS] with
  // wsm [CS][c] -- the data gradient of a pointwise convolution to CS channels (to_rgb), formed in registers.
  constexpr int E = Piece<T>::E;
  float wreg[CS > 0 ? CS : 1][E];
  if constexpr (CS > 0) {
    const int p0 = threadIdx.x % tp;
#pragma unroll
    for (int j = 0; j < CS; ++j)
#pragma unroll
      for (int e = 0; e < E; ++e) wreg[j][e] = p0 * E + e < c ? wsm[j * c + p0 * E + e] : 0.f;
  }
  __shared__ float red[BWD ? 256 * E : 1];
  const int P = c / E;
  const int lane_t = threadIdx.x % tp;
  const int teams = blockDim.x / tp;
  const float inv_c = 1.f / (float)c;
  const int nw = (c + 31) >> 5;
  float cs[KP][E];
#pragma unroll
  for (int k = 0; k < KP; ++k)
#pragma unroll
    for (int e = 0; e < E; ++e) cs[k][e] = 0.f;
  float dwr[WG ? CS : 1][E], gsum[WG ? CS : 1];
#pragma unroll
  for (int j = 0; j < (WG ? CS : 1); ++j) {
    gsum[j] = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) dwr[j][e] = 0.f;
  }
  const int64_t vstride = (int64_t)gridDim.x * teams;
  for (int64_t v0 = (int64_t)blockIdx.x * teams + threadIdx.x / tp; v0 < nvox; v0 += vstride * U) {
    Piece<T> pa[U][KP], py[U][KP];
    float s[U], scv[U];
    uint32_t sw[U][KP];
    float gsv[U][CS > 0 ? CS : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * vstride;
      s[u] = 0.f;
      scv[u] = 0.f;
      if constexpr (CS > 0) {
#pragma unroll
        for (int j = 0; j < CS; ++j) gsv[u][j] = v < nvox ? sg_traits<T>::to_f(a[v * CS + j]) : 0.f;
      }
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int p = lane_t + k * tp;
        sw[u][k] = 0u;
        if (v < nvox && p < P) {
          if constexpr (CS == 0) pa[u][k].load(a + v * c + (int64_t)p * E);
          if (BWD) {
            py[u][k].load(yv + v * c + (int64_t)p * E);
            if (words) sw[u][k] = piece_signs(words, v, nw, p * E);
          }
        }
      }
      if (BWD && v < nvox) scv[u] = scale_in[v];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * vstride;
      if constexpr (CS > 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < CS; ++j) t = fmaf(gsv[u][j], wreg[j][e], t);
          pa[u][0].v[e] = t;
        }
        if constexpr (WG) {       // (dead voxels and lanes beyond the last piece loaded zeros / are masked below)
          if (v < nvox && lane_t < P) {
#pragma unroll
            for (int j = 0; j < CS; ++j) {
#pragma unroll
              for (int e = 0; e < E; ++e) dwr[j][e] = fmaf(gsv[u][j], py[u][0].v[e], dwr[j][e]);
              if (lane_t == 0) gsum[j] += gsv[u][j];
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int p = lane_t + k * tp;
        if (v < nvox && p < P) {
#pragma unroll
          for (int e = 0; e < E; ++e) s[u] += pa[u][k].v[e] * (BWD ? py[u][k].v[e] : pa[u][k].v[e]);
        }
      }
      for (int m = tp >> 1; m >= 1; m >>= 1) s[u] += __shfl_xor(s[u], m);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * vstride;
      if (v >= nvox) continue;
      if (BWD) {
        const float sc = scv[u];
        const float mean = s[u] * inv_c;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const int p = lane_t + k * tp;
          if (p < P) {
#pragma unroll
            for (int e = 0; e < E; ++e) pa[u][k].v[e] = sc * (pa[u][k].v[e] - py[u][k].v[e] * mean);
            if (words) {
#pragma unroll
              for (int e = 0; e < E; ++e) pa[u][k].v[e] = ((sw[u][k] >> e) & 1u) ? pa[u][k].v[e] * slope : pa[u][k].v[e];
            }
            if (part) {
#pragma unroll
              for (int e = 0; e < E; ++e) cs[k][e] += pa[u][k].v[e];
            }
            pa[u][k].store(out + v * c + (int64_t)p * E);
          }
        }
      } else {
        const float sc = rsqrtf(s[u] * inv_c + eps);
        if (scale_out && lane_t == 0) scale_out[v] = sc;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const int p = lane_t + k * tp;
          if (p < P) {
#pragma unroll
            for (int e = 0; e < E; ++e) pa[u][k].v[e] *= sc;
            pa[u][k].store(out + v * c + (int64_t)p * E);
          }
        }
      }
    }
  }
  if (BWD && part) {   // channel sums of this block: teams hold the same pieces, one LDS pass per piece slot
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      if (k * tp < P) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) red[threadIdx.x * E + e] = cs[k][e];
        __syncthreads();
        const int p = threadIdx.x + k * tp;
        if (threadIdx.x < tp && p < P) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            float t = 0.f;
            for (int j = 0; j < teams; ++j) t += red[(j * tp + threadIdx.x) * E + e];
            part[(int64_t)blockIdx.x * (WG ? c * (1 + CS) + CS : c) + p * E + e] = t;
          }
        }
      }
    }
    if constexpr (WG) {
      const int64_t row = (int64_t)blockIdx.x * (c * (1 + CS) + CS);
#pragma unroll
      for (int j = 0; j < CS; ++j) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) red[threadIdx.x * E + e] = dwr[j][e];
        __syncthreads();
        if (threadIdx.x < tp && threadIdx.x < P) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            float t = 0.f;
            for (int q = 0; q < teams; ++q) t += red[(q * tp + threadIdx.x) * E + e];
            part[row + c * (1 + j) + threadIdx.x * E + e] = t;
          }
        }
        __syncthreads();
        red[threadIdx.x] = lane_t == 0 ? gsum[j] : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
          float t = 0.f;
          for (int q = 0; q < 256; ++q) t += red[q];
          part[row + c * (1 + CS) + j] = t;
        }
      }
    }
  }
}

template <typename T, bool BWD>
__global__ void pixel_norm_scalar_kernel(const T* __restrict__ a, const T* __restrict__ yv,
                                         const float* __restrict__ scale_in, T* __restrict__ out,
                                         float* __restrict__ scale_out, int64_t nvox, int c, float eps) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int ch = 0; ch < c; ++ch) {
      float av = sg_traits<T>::to_f(a[v * c + ch]);
      s += BWD ? av * sg_traits<T>::to_f(yv[v * c + ch]) : av * av;
    }
    if (BWD) {
      const float sc = scale_in[v], mean = s / (float)c;
      for (int ch = 0; ch < c; ++ch)
        out[v * c + ch] = sg_traits<T>::from_f(
            sc * (sg_traits<T>::to_f(a[v * c + ch]) - sg_traits<T>::to_f(yv[v * c + ch]) * mean));
    } else {
      const float sc = rsqrtf(s / (float)c + eps);
      if (scale_out) scale_out[v] = sc;
      for (int ch = 0; ch < c; ++ch) out[v * c + ch] = sg_traits<T>::from_f(sg_traits<T>::to_f(a[v * c + ch]) * sc);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Nearest-neighbour up-sampling / sum-pooling by a factor of 1 or 2 PER DIMENSION (shift sd, sh, sw in {0, 1}): (1,1,1)
// is upscale3d / downscale3d, (0,1,1) the 2-D twins (SURFGAN_2D/networks/ops.py:176-231), (0,1,0) the H-only pooling
// that finishes a convolution's fused D x W pooling.
template <typename T, bool VEC>
__global__ void upscale2x_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int d, int h, int w, int c,
                                 int sd, int sh, int sw, float gain, const uint32_t* __restrict__ mask_bits,
                                 float mask_slope) {
  constexpr int E = VEC ? Piece<T>::E : 1;
  const int P = c / E;
  const int OD = d << sd, OH = h << sh, OW = w << sw;
  const int64_t total = (int64_t)n * OD * OH * OW * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    int64_t q = i / P;
    const int ow = (int)(q % OW); q /= OW;
    const int oh = (int)(q % OH); q /= OH;
    const int od = (int)(q % OD);
    const int nn = (int)(q / OD);
    const int64_t src = ((((int64_t)nn * d + (od >> sd)) * h + (oh >> sh)) * w + (ow >> sw)) * c + (int64_t)p * E;
    if (VEC) {
      Piece<T> pc;
      pc.load(x + src);
      if (gain != 1.f) {
#pragma unroll
        for (int e = 0; e < Piece<T>::E; ++e) pc.v[e] *= gain;
      }
      if (mask_bits) {
        const uint32_t sw_ = piece_signs(mask_bits, i / P, (c + 31) >> 5, p * E);
#pragma unroll
        for (int e = 0; e < Piece<T>::E; ++e) pc.v[e] = ((sw_ >> e) & 1u) ? pc.v[e] * mask_slope : pc.v[e];
      }
      pc.store(y + i * E);
    } else {
      float v = sg_traits<T>::to_f(x[src]) * gain;
      if (mask_bits && ((mask_bits[(i / c) * ((c + 31) >> 5) + (p >> 5)] >> (p & 31)) & 1u)) v *= mask_slope;
      y[i] = sg_traits<T>::from_f(v);
    }
  }
}

// Row-wise variant for the vector case: one block per output row (n, od, oh) -- scalar index arithmetic once per
// block instead of 64-bit divisions per 16-byte piece (the flat kernel reached 3 TB/s).  HP = 2 (H is doubled): a block
// writes the PAIR of output rows (oh, oh + 1) that share one source row from a single read of it.
template <typename T, int HP>
__global__ __launch_bounds__(256) void upscale2x_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int d, int h, int w,
                                                             int c, int sd, int sh, int sw, float gain,
                                                             const uint32_t* __restrict__ mask_bits, float mask_slope,
                                                             int64_t nrows, int pc, int64_t plane_stride) {
  // pc < c: y is written as c / pc separate NDHWC tensors of pc channels each, plane_stride elements apart
  // (sg_upscale_nn_planes); pc == c: the ordinary layout
  constexpr int E = Piece<T>::E;
  const int P = c / E, nw = (c + 31) >> 5, PP = pc / E;
  const int OD = d << sd, OH = h << sh, OW = w << sw;
  const int per_row = OW * P;
  const int64_t ngroups = nrows / HP;            // HP == 2: OH is even, rows pair up within a (n, od) slab
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t row = grp * HP;
    const int oh = (int)(row % OH);
    const int64_t q = row / OH;
    const int od = (int)(q % OD);
    const int64_t nn = q / OD;
    const T* xrow = x + (((nn * d + (od >> sd)) * h + (oh >> sh)) * (int64_t)w) * c;
    T* yrow = y + row * (int64_t)OW * pc;
    const uint32_t* mrow = mask_bits ? mask_bits + row * (int64_t)OW * nw : nullptr;
    // four pieces per thread and trip, every load issued before the first store (a 128-voxel row of 64 channels is
    // exactly one trip): one piece per trip left the kernel at 3.7 of 6.3 TB/s
    for (int i0 = threadIdx.x; i0 < per_row; i0 += 256 * 4) {
      Piece<T> pcs[4];
      uint32_t sg[HP][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + 256 * j;
#pragma unroll
        for (int r = 0; r < HP; ++r) sg[r][j] = 0u;
        if (i < per_row) {
          const int ow = i / P, p = i - ow * P;
          pcs[j].load(xrow + (int64_t)(ow >> sw) * c + p * E);
          if (mrow) {
#pragma unroll
            for (int r = 0; r < HP; ++r) sg[r][j] = mrow[((int64_t)r * OW + ow) * nw + ((p * E) >> 5)] >> ((p * E) & 31);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + 256 * j;
        if (i < per_row) {
          const int ow = i / P, p = i - ow * P, pl = p / PP;
#pragma unroll
          for (int r = 0; r < HP; ++r) {
            Piece<T> o;
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const float v = pcs[j].v[e] * gain;
              o.v[e] = ((sg[r][j] >> e) & 1u) ? v * mask_slope : v;
            }
            o.store_nt(yrow + pl * plane_stride + ((int64_t)r * OW + ow) * pc + (p - pl * PP) * E);
          }
        }
      }
    }
  }
}

// MASKED: each source voxel is first multiplied by where(sign bit, mask_slope, 1) from sign words shaped like x
// (the gradient of a masked nearest up-scale in one pass: y = gain * sum_block M * x).
template <typename T, bool VEC, bool MASKED = false>
__global__ void downscale2x_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int d, int h, int w, int c,
                                   int sd, int sh, int sw, float gain, const uint32_t* __restrict__ bits = nullptr,
                                   float mask_slope = 0.f) {
  const int nw = (c + 31) >> 5;
  constexpr int E = VEC ? Piece<T>::E : 1;
  const int P = c / E;
  const int od = d >> sd, oh = h >> sh, ow = w >> sw;
  const int64_t total = (int64_t)n * od * oh * ow * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    int64_t q = i / P;
    const int xw = (int)(q % ow); q /= ow;
    const int xh = (int)(q % oh); q /= oh;
    const int xd = (int)(q % od);
    const int nn = (int)(q / od);
    float s[Piece<T>::E];
#pragma unroll
    for (int e = 0; e < Piece<T>::E; ++e) s[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kd = k >> 2, kh = (k >> 1) & 1, kw = k & 1;
      if (kd > sd || kh > sh || kw > sw) continue;     // uniform: a factor-1 dimension has one tap
      const int64_t vox = (((int64_t)nn * d + ((xd << sd) + kd)) * h + ((xh << sh) + kh)) * w + ((xw << sw) + kw);
      const int64_t src = vox * c + (int64_t)p * E;
      uint32_t mb = 0u;
      if (MASKED) mb = bits[vox * nw + ((p * E) >> 5)] >> ((p * E) & 31);
      if (VEC) {
        Piece<T> pc;
        pc.load(x + src);
#pragma unroll
        for (int e = 0; e < Piece<T>::E; ++e) s[e] += (MASKED && ((mb >> e) & 1u)) ? pc.v[e] * mask_slope : pc.v[e];
      } else {
        const float v = sg_traits<T>::to_f(x[src]);
        s[0] += (MASKED && (mb & 1u)) ? v * mask_slope : v;
      }
    }
    if (VEC) {
      Piece<T> o;
#pragma unroll
      for (int e = 0; e < Piece<T>::E; ++e) o.v[e] = s[e] * gain;
      o.store(y + i * E);
    } else {
      y[i] = sg_traits<T>::from_f(s[0] * gain);
    }
  }
}

// Trilinear x2 up-sampling (half-pixel centres, i.e. torch / TF2 `align_corners=False`): per axis
//   out[2i] = 0.25 x[i-1] + 0.75 x[i],  out[2i+1] = 0.75 x[i] + 0.25 x[i+1]   (indices clamped at the borders)
// and its adjoint (the gradient), written as a gather so that no atomics are needed: input voxel i receives
//   0.25 g[2i-1] + w0 g[2i] + w1 g[2i+1] + 0.25 g[2i+2],  w0 = 0.75 (+0.25 at i = 0), w1 = 0.75 (+0.25 at i = n-1).
// The reference has no trilinear op (nearest only); BASELINE north_star names one, so it is
// offered next to upscale3d with torch's CPU interpolate as its oracle.  Trilinear x2 DOWN-sampling with half-pixel
// centres samples exactly between two voxels per axis, i.e. it IS the 2x2x2 mean: sg_downscale2x(gain 1/8).
__device__ __forceinline__ void tri_up_taps(int o, int n, int& i0, int& i1, float& w0, float& w1) {
  const int i = o >> 1;
  if (o & 1) { i0 = i; i1 = min(i + 1, n - 1); w0 = 0.75f; w1 = 0.25f; }
  else { i0 = max(i - 1, 0); i1 = i; w0 = 0.25f; w1 = 0.75f; }
}

template <typename T, bool VEC>
__global__ void trilinear_up2x_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int d, int h, int w, int c) {
  constexpr int E = VEC ? Piece<T>::E : 1;
  const int P = c / E;
  const int64_t total = (int64_t)n * (2 * d) * (2 * h) * (2 * w) * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    int64_t q = i / P;
    const int ow = (int)(q % (2 * w)); q /= 2 * w;
    const int oh = (int)(q % (2 * h)); q /= 2 * h;
    const int od = (int)(q % (2 * d));
    const int nn = (int)(q / (2 * d));
    int id[2], ih[2], iw[2];
    float wd[2], wh[2], ww[2];
    tri_up_taps(od, d, id[0], id[1], wd[0], wd[1]);
    tri_up_taps(oh, h, ih[0], ih[1], wh[0], wh[1]);
    tri_up_taps(ow, w, iw[0], iw[1], ww[0], ww[1]);
    float acc[Piece<T>::E];
#pragma unroll
    for (int e = 0; e < Piece<T>::E; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int a = k >> 2, b = (k >> 1) & 1, cc = k & 1;
      const float wt = wd[a] * wh[b] * ww[cc];
      const int64_t src = ((((int64_t)nn * d + id[a]) * h + ih[b]) * w + iw[cc]) * c + (int64_t)p * E;
      if (VEC) {
        Piece<T> pc;
        pc.load(x + src);
#pragma unroll
        for (int e = 0; e < Piece<T>::E; ++e) acc[e] += wt * pc.v[e];
      } else {
        acc[0] += wt * sg_traits<T>::to_f(x[src]);
      }
    }
    if (VEC) {
      Piece<T> o;
#pragma unroll
      for (int e = 0; e < Piece<T>::E; ++e) o.v[e] = acc[e];
      o.store(y + i * E);
    } else {
      y[i] = sg_traits<T>::from_f(acc[0]);
    }
  }
}

// adjoint: g [n,2d,2h,2w,c] -> dx [n,d,h,w,c]
template <typename T, bool VEC>
__global__ void trilinear_up2x_adj_kernel(const T* __restrict__ g, T* __restrict__ dx, int n, int d, int h, int w, int c) {
  constexpr int E = VEC ? Piece<T>::E : 1;
  const int P = c / E;
  const int64_t total = (int64_t)n * d * h * w * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    int64_t q = i / P;
    const int xw = (int)(q % w); q /= w;
    const int xh = (int)(q % h); q /= h;
    const int xd = (int)(q % d);
    const int nn = (int)(q / d);
    float wd[4], wh[4], ww[4];      // taps o = 2i-1 .. 2i+2 (weight 0 where o falls outside)
    auto taps = [](int idx, int len, float (&wt)[4]) {
      wt[0] = idx > 0 ? 0.25f : 0.f;
      wt[1] = idx == 0 ? 1.0f : 0.75f;
      wt[2] = idx == len - 1 ? 1.0f : 0.75f;
      wt[3] = idx < len - 1 ? 0.25f : 0.f;
    };
    taps(xd, d, wd); taps(xh, h, wh); taps(xw, w, ww);
    float acc[Piece<T>::E];
#pragma unroll
    for (int e = 0; e < Piece<T>::E; ++e) acc[e] = 0.f;
    for (int a = 0; a < 4; ++a) {
      if (wd[a] == 0.f) continue;
      const int od = 2 * xd - 1 + a;
      for (int b = 0; b < 4; ++b) {
        if (wh[b] == 0.f) continue;
        const int oh = 2 * xh - 1 + b;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float wt = wd[a] * wh[b] * ww[cc];
          if (ww[cc] == 0.f) continue;
          const int ow = 2 * xw - 1 + cc;
          const int64_t src = ((((int64_t)nn * 2 * d + od) * 2 * h + oh) * 2 * w + ow) * c + (int64_t)p * E;
          if (VEC) {
            Piece<T> pc;
            pc.load(g + src);
#pragma unroll
            for (int e = 0; e < Piece<T>::E; ++e) acc[e] += wt * pc.v[e];
          } else {
            acc[0] += wt * sg_traits<T>::to_f(g[src]);
          }
        }
      }
    }
    if (VEC) {
      Piece<T> o;
#pragma unroll
      for (int e = 0; e < Piece<T>::E; ++e) o.v[e] = acc[e];
      o.store(dx + i * E);
    } else {
      dx[i] = sg_traits<T>::from_f(acc[0]);
    }
  }
}

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, float wa,
                             float wb, int64_t numel, const float* __restrict__ w_dev = nullptr) {
  constexpr int E = Piece<T>::E;
  if (w_dev) {      // the coefficients of a captured step live on the device (sg_axpby_dev): same f32 values, same arithmetic
    wa = w_dev[0];
    wb = b ? w_dev[1] : 0.f;
  }
  const int64_t nv = numel / E;
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid0; i < nv; i += stride) {
    Piece<T> pa, pb;
    pa.load(a + i * E);
    if (b) {
      pb.load(b + i * E);
#pragma unroll
      for (int e = 0; e < E; ++e) pa.v[e] = wa * pa.v[e] + wb * pb.v[e];
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) pa.v[e] = wa * pa.v[e];
    }
    pa.store(out + i * E);
  }
  for (int64_t i = nv * E + tid0; i < numel; i += stride) {
    float v = wa * sg_traits<T>::to_f(a[i]) + (b ? wb * sg_traits<T>::to_f(b[i]) : 0.f);
    out[i] = sg_traits<T>::from_f(v);
  }
}

// out[s][i] = g_s * a[s][i] + (1 - g_s) * b[s][i] with one weight per batch sample: the gradient penalty's interpolates
// (networks/loss.py:133-134, :70-71).  f32 arithmetic from the f32 weight, ONE rounding to T.
template <typename T>
__global__ void lerp_rows_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ gamma, T* __restrict__ out,
                                 int64_t per_sample, int64_t numel) {
  constexpr int E = Piece<T>::E;
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  if (per_sample % E == 0) {
    const int64_t nv = numel / E, pv = per_sample / E;
    for (int64_t i = tid0; i < nv; i += stride) {
      const float g = gamma[i / pv], h = 1.f - g;
      Piece<T> pa, pb;
      pa.load(a + i * E);
      pb.load(b + i * E);
#pragma unroll
      for (int e = 0; e < E; ++e) pa.v[e] = g * pa.v[e] + h * pb.v[e];
      pa.store(out + i * E);
    }
  } else {
    for (int64_t i = tid0; i < numel; i += stride) {
      const float g = gamma[i / per_sample];
      out[i] = sg_traits<T>::from_f(g * sg_traits<T>::to_f(a[i]) + (1.f - g) * sg_traits<T>::to_f(b[i]));
    }
  }
}

// Philox4x32-10 (Salmon et al. 2011), counter = element index / 4, key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int rnd = 0; rnd < 10; ++rnd) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename T>
__global__ void add_noise_kernel(const T* __restrict__ x, T* __restrict__ out, float stddev, uint64_t seed,
                                 uint64_t offset, int64_t numel, const uint64_t* __restrict__ offset_dev = nullptr) {
  if (offset_dev != nullptr) offset = *offset_dev;     // (sg_add_noise_dev: the counter lives on the device, graph replays move on)
  const int64_t ngroups = (numel + 3) / 4;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t ctr = (uint64_t)gi + offset;
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    float z[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // Box-Muller on two uniform pairs
      const float u1 = ((float)r[2 * k] + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
      const float u2 = (float)r[2 * k + 1] * 2.3283064365386963e-10f;
      const float rad = sqrtf(-2.f * __logf(u1));
      float sn, cs;
      __sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * k] = rad * cs;
      z[2 * k + 1] = rad * sn;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = gi * 4 + k;
      if (i < numel) out[i] = sg_traits<T>::from_f(sg_traits<T>::to_f(x[i]) + stddev * z[k]);
    }
  }
}

// out[n*W + w] += sum over a slab of (d,h) rows and all channels of g^2; threads run along w (coalesced).
template <typename T>
__global__ void sumsq_keep_w_kernel(const T* __restrict__ g, float* __restrict__ out, int n, int dh, int w, int c,
                                    int rows_per_block) {
  const int nn = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(dh, r0 + rows_per_block);
  for (int ww = threadIdx.x; ww < w; ww += blockDim.x) {
    float s = 0.f;
    for (int rr = r0; rr < r1; ++rr) {
      const T* p = g + (((int64_t)nn * dh + rr) * w + ww) * c;
      for (int ch = 0; ch < c; ++ch) {
        const float v = sg_traits<T>::to_f(p[ch]);
        s += v * v;
      }
    }
    atomicAdd(out + (int64_t)nn * w + ww, s);
  }
}

// Reproducible form (SG_DETERMINISTIC=1): one block per sample, thread (ty, w) adds rows ty, ty + TY, ... in order, the TY
// partial sums are added in order through LDS; no atomics, no memset.
template <typename T>
__global__ __launch_bounds__(1024) void sumsq_keep_w_ordered_kernel(const T* __restrict__ g, float* __restrict__ out, int dh, int w,
                                                                    int c, int tw) {
  __shared__ float part[1024];
  const int nn = blockIdx.x;
  const int ty = threadIdx.x / tw, tx = threadIdx.x % tw, TY = 1024 / tw;
  for (int w0 = 0; w0 < w; w0 += tw) {
    const int ww = w0 + tx;
    float s = 0.f;
    if (ww < w)
      for (int rr = ty; rr < dh; rr += TY) {
        const T* p = g + (((int64_t)nn * dh + rr) * w + ww) * c;
        for (int ch = 0; ch < c; ++ch) {
          const float v = sg_traits<T>::to_f(p[ch]);
          s += v * v;
        }
      }
    part[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && ww < w) {
      float t = 0.f;
      for (int k = 0; k < TY; ++k) t += part[k * tw + tx];
      out[(int64_t)nn * w + ww] = t;
    }
    __syncthreads();
  }
}

// minibatch stddev, stage 1: stat[m] = mean over (vox, c) of sqrt(var over the group + 1e-8)
template <typename T>
__global__ __launch_bounds__(256) void mbstd_stat_kernel(const T* __restrict__ x, float* __restrict__ stat, int group,
                                                         int mgroups, int64_t per_sample) {
  __shared__ float red[256];
  const int m = blockIdx.y;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * blockDim.x) {
    float mean = 0.f;
    for (int gidx = 0; gidx < group; ++gidx) mean += sg_traits<T>::to_f(x[((int64_t)gidx * mgroups + m) * per_sample + i]);
    mean /= (float)group;
    float var = 0.f;
    for (int gidx = 0; gidx < group; ++gidx) {
      const float dlt = sg_traits<T>::to_f(x[((int64_t)gidx * mgroups + m) * per_sample + i]) - mean;
      var += dlt * dlt;
    }
    s += sqrtf(var / (float)group + 1e-8f);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(stat + m, red[0] / (float)per_sample);
}

template <typename T>
__global__ void mbstd_concat_kernel(const T* __restrict__ x, const float* __restrict__ stat, T* __restrict__ y, int n,
                                    int64_t vox, int c, int mgroups) {
  const int64_t total = (int64_t)n * vox * (c + 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % (c + 1));
    const int64_t v = i / (c + 1);
    const int nn = (int)(v / vox);
    y[i] = ch < c ? x[v * c + ch] : sg_traits<T>::from_f(stat[nn % mgroups]);
  }
}

// minibatch stddev backward, stage 1: S[m] = sum over the group's samples and voxels of dy[.., c] (the statistic's
// channel); stage 2: dx = dy[.., :c] + S[m] / (P * G) * (x - mean_g x) / sqrt(var_g x + 1e-8), P = vox * c.
template <typename T>
__global__ __launch_bounds__(256) void mbstd_bwd_sum_kernel(const T* __restrict__ dy, float* __restrict__ sums, int n, int64_t vox,
                                                            int c, int mgroups) {
  __shared__ float red[256];
  const int nn = blockIdx.y;
  float s = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += (int64_t)gridDim.x * blockDim.x)
    s += sg_traits<T>::to_f(dy[((int64_t)nn * vox + v) * (c + 1) + c]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(sums + nn % mgroups, red[0]);
}

template <typename T>
__global__ __launch_bounds__(256) void mbstd_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const float* __restrict__ sums, T* __restrict__ dx, int group,
                                                        int mgroups, int64_t vox, int c) {
  const int m = blockIdx.y;
  const int64_t per_sample = vox * c;
  const float k = sums[m] / ((float)per_sample * (float)group);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * blockDim.x) {
    float mean = 0.f;
    for (int gidx = 0; gidx < group; ++gidx) mean += sg_traits<T>::to_f(x[((int64_t)gidx * mgroups + m) * per_sample + i]);
    mean /= (float)group;
    float var = 0.f;
    for (int gidx = 0; gidx < group; ++gidx) {
      const float dlt = sg_traits<T>::to_f(x[((int64_t)gidx * mgroups + m) * per_sample + i]) - mean;
      var += dlt * dlt;
    }
    const float rs = k * rsqrtf(var / (float)group + 1e-8f);
    const int64_t v = i / c;
    const int ch = (int)(i - v * c);
    for (int gidx = 0; gidx < group; ++gidx) {
      const int64_t nn = (int64_t)gidx * mgroups + m;
      const float xv = sg_traits<T>::to_f(x[nn * per_sample + i]);
      const float g0 = sg_traits<T>::to_f(dy[(nn * vox + v) * (c + 1) + ch]);
      dx[nn * per_sample + i] = sg_traits<T>::from_f(g0 + rs * (xv - mean));
    }
  }
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t numel) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = sg_traits<D>::from_f(sg_traits<S>::to_f(src[i]));
}

int team_size(int P) {
  int tp = 1;
  while (tp < P && tp < 64) tp <<= 1;
  return tp;
}

}  // namespace

#define SG_DISPATCH(dt, CALL_BF16, CALL_F32) \
  do {                                       \
    if ((dt) == SG_BF16) { CALL_BF16; }      \
    else if ((dt) == SG_F32) { CALL_F32; }   \
    else return SG_EINVAL;                   \
  } while (0)

extern "C" int sg_bias_act_fwd(const void* x, const float* bias, void* y, int64_t nvox, int32_t c, int32_t act,
                               float slope, sg_dtype dt, sg_stream_t st) {
  if (!x || !y || nvox < 1 || c < 1) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const bool vec = (c % E == 0) && sg_aligned16(x) && sg_aligned16(y);
  const int64_t items = vec ? nvox * (c / E) : nvox * c;
  const int blocks = grid_for(items);
#define L(T, V) hipLaunchKernelGGL((bias_act_fwd_kernel<T, V>), dim3(blocks), dim3(256), 0, hs, (const T*)x, bias, (T*)y, nvox, c, act, slope)
  if (vec) SG_DISPATCH(dt, L(bf16_t, true), L(float, true));
  else SG_DISPATCH(dt, L(bf16_t, false), L(float, false));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

static const int kBwdBlocks = 1024;      // (swept 512 / 1024 / 2048 / 4096 at the in-step shapes: 1024 is the best compromise, tools/ew_roofline.py)
extern "C" size_t sg_bias_act_bwd_workspace(int32_t c) { return (size_t)kBwdBlocks * (size_t)(c > 0 ? c : 0) * sizeof(float); }

static int bias_act_bwd_launch(const void* dy, const void* y, const uint32_t* words, void* dx, float* dbias,
                               void* workspace, int64_t nvox, int32_t c, float slope, sg_dtype dt, sg_stream_t st) {
  if (!dy || nvox < 1 || c < 1 || (!dx && !dbias)) return SG_EINVAL;
  if (dbias && !workspace) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const int P = c / E;
  const bool vec = (c % E == 0) && ((P <= 256 && 256 % P == 0) || P % 256 == 0) && sg_aligned16(dy) &&
                   (!y || sg_aligned16(y)) && (!dx || sg_aligned16(dx));
  float* part = dbias ? reinterpret_cast<float*>(workspace) : nullptr;
  int blocks;
  if (vec) {
    const int rows = P >= 256 ? 1 : 256 / P;
    const int ny = P > 256 ? P / 256 : 1;
    blocks = grid_for(nvox, rows, kBwdBlocks);
#define L(T) hipLaunchKernelGGL((bias_act_bwd_vec_kernel<T>), dim3(blocks, ny), dim3(256), 0, hs, (const T*)dy, (const T*)y, (T*)dx, part, nvox, c, slope, words)
    SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  } else if (c <= 8) {
    blocks = grid_for(nvox, 256, kBwdBlocks);
#define L(T) hipLaunchKernelGGL((bias_act_bwd_small_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)dy, (const T*)y, (T*)dx, part, nvox, c, slope, words)
    SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  } else {
    blocks = grid_for(nvox, 64, kBwdBlocks);
#define L(T) hipLaunchKernelGGL((bias_act_bwd_scalar_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)dy, (const T*)y, (T*)dx, part, nvox, c, slope, words)
    SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  }
  SG_LAUNCH_CHECK();
  if (dbias) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((c + 31) / 32), dim3(1024), 0, hs, part, dbias, blocks, c);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

extern "C" int sg_bias_act_bwd(const void* dy, const void* y, void* dx, float* dbias, void* workspace, int64_t nvox,
                               int32_t c, float slope, sg_dtype dt, sg_stream_t st) {
  return bias_act_bwd_launch(dy, y, nullptr, dx, dbias, workspace, nvox, c, slope, dt, st);
}

extern "C" int sg_bias_act_bwd_bits(const void* dy, const void* y_sign_words, void* dx, float* dbias, void* workspace,
                                    int64_t nvox, int32_t c, float slope, sg_dtype dt, sg_stream_t st) {
  if (!y_sign_words) return SG_EINVAL;
  return bias_act_bwd_launch(dy, nullptr, reinterpret_cast<const uint32_t*>(y_sign_words), dx, dbias, workspace, nvox,
                             c, slope, dt, st);
}

extern "C" size_t sg_sign_words_bytes(int64_t nvox, int32_t c) {
  return (nvox > 0 && c > 0) ? (size_t)nvox * (size_t)((c + 31) / 32) * 4u : 0u;
}

extern "C" int sg_sign_words(const void* t, void* words, int64_t nvox, int32_t c, sg_dtype dt, sg_stream_t st) {
  if (!t || !words || nvox < 1 || c < 1) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int blocks = grid_for(nvox * ((c + 31) / 32));
#define L(T) hipLaunchKernelGGL((sign_words_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)t, (uint32_t*)words, nvox, c)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

template <bool BWD>
static int pixel_norm_launch(const void* a, const void* yv, const float* scale_in, void* out, float* scale_out,
                             int64_t nvox, int32_t c, float eps, sg_dtype dt, hipStream_t hs) {
  const int E = dt == SG_BF16 ? 8 : 4;
  const int P = c / E;
  const bool vec = (c % E == 0) && P <= 256 && sg_aligned16(a) && sg_aligned16(out) && (!yv || sg_aligned16(yv));
  if (vec) {
    const int tp = team_size(P);
    const int teams = 256 / tp;
    const int blocks = grid_for(nvox, teams, 4096);
#define L(T) do { if (P <= tp) hipLaunchKernelGGL((pixel_norm_kernel<T, BWD, 1, 2>), dim3(blocks), dim3(256), 0, hs, (const T*)a, (const T*)yv, scale_in, (T*)out, scale_out, nvox, c, tp, eps); \
                  else hipLaunchKernelGGL((pixel_norm_kernel<T, BWD, 4, 1>), dim3(blocks), dim3(256), 0, hs, (const T*)a, (const T*)yv, scale_in, (T*)out, scale_out, nvox, c, tp, eps); } while (0)
    SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  } else {
    const int blocks = grid_for(nvox);
#define L(T) hipLaunchKernelGGL((pixel_norm_scalar_kernel<T, BWD>), dim3(blocks), dim3(256), 0, hs, (const T*)a, (const T*)yv, scale_in, (T*)out, scale_out, nvox, c, eps)
    SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  }
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_pixel_norm_fwd(const void* x, void* y, float* scale, int64_t nvox, int32_t c, float eps,
                                 sg_dtype dt, sg_stream_t st) {
  if (!x || !y || nvox < 1 || c < 1) return SG_EINVAL;
  return pixel_norm_launch<false>(x, nullptr, nullptr, y, scale, nvox, c, eps, dt, sg_st(st));
}

extern "C" int sg_pixel_norm_bwd(const void* dy, const void* y, const float* scale, void* dx, int64_t nvox,
                                 int32_t c, sg_dtype dt, sg_stream_t st) {
  if (!dy || !y || !scale || !dx || nvox < 1 || c < 1) return SG_EINVAL;
  return pixel_norm_launch<true>(dy, y, scale, dx, nullptr, nvox, c, 0.f, dt, sg_st(st));
}

extern "C" int sg_pixel_norm_act_bwd(const void* dy, const void* y, const float* scale, const void* y_sign_words,
                                     float slope, void* dz, float* dbias, void* workspace, int64_t nvox, int32_t c,
                                     sg_dtype dt, sg_stream_t st) {
  if (!dy || !y || !scale || !y_sign_words || !dz || nvox < 1 || c < 1) return SG_EINVAL;
  if (dbias && !workspace) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const int P = c / E;
  const bool vec = (c % E == 0) && P <= 256 && sg_aligned16(dy) && sg_aligned16(dz) && sg_aligned16(y);
  if (!vec) {   // generic channel counts: the two passes, the second in place
    int rc = pixel_norm_launch<true>(dy, y, scale, dz, nullptr, nvox, c, 0.f, dt, hs);
    if (rc != SG_OK) return rc;
    return sg_bias_act_bwd_bits(dz, y_sign_words, dz, dbias, workspace, nvox, c, slope, dt, st);
  }
  const int tp = team_size(P);
  const int teams = 256 / tp;
  const int blocks = grid_for(nvox, teams, kBwdBlocks);
  float* part = dbias ? reinterpret_cast<float*>(workspace) : nullptr;
#define L(T) do { if (P <= tp) hipLaunchKernelGGL((pixel_norm_kernel<T, true, 1, 2>), dim3(blocks), dim3(256), 0, hs, (const T*)dy, (const T*)y, scale, (T*)dz, (float*)nullptr, nvox, c, tp, 0.f, (const uint32_t*)y_sign_words, slope, part); \
                  else hipLaunchKernelGGL((pixel_norm_kernel<T, true, 4, 1>), dim3(blocks), dim3(256), 0, hs, (const T*)dy, (const T*)y, scale, (T*)dz, (float*)nullptr, nvox, c, tp, 0.f, (const uint32_t*)y_sign_words, slope, part); } while (0)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  if (dbias) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((c + 31) / 32), dim3(1024), 0, hs, part, dbias, blocks, c);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

// column sums of `part` ([nb][c0 + c1 + c2]) into three destinations (the middle one scaled): the bias gradient of the stage, the
// pointwise convolution's weight gradient and its bias gradient
__global__ __launch_bounds__(1024) void colsum_finalize3_kernel(const float* __restrict__ part, int nb, float* __restrict__ out0, int c0,
                                                                float* __restrict__ out1, int c1, float coef1, float* __restrict__ out2, int c2) {
  __shared__ float red[32][33];
  const int c = c0 + c1 + c2;
  const int ch = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (ch < c) {
    int b = rg;
    for (; b + 96 < nb; b += 128) {
      s0 += part[(int64_t)b * c + ch];
      s1 += part[(int64_t)(b + 32) * c + ch];
      s2 += part[(int64_t)(b + 64) * c + ch];
      s3 += part[(int64_t)(b + 96) * c + ch];
    }
    for (; b < nb; b += 32) s0 += part[(int64_t)b * c + ch];
  }
  red[rg][threadIdx.x & 31] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && ch < c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
    if (ch < c0) { if (out0) out0[ch] = t; }
    else if (ch < c0 + c1) { if (out1) out1[ch - c0] = coef1 * t; }
    else if (out2) out2[ch - c0 - c1] = t;
  }
}

extern "C" size_t sg_pixel_norm_act_bwd_pw_wg_workspace(int32_t c, int32_t cs) {
  return (c > 0 && cs > 0) ? (size_t)kBwdBlocks * ((size_t)c * (1 + cs) + cs) * sizeof(float) : 0;
}

// sg_pixel_norm_act_bwd_pw that ALSO returns the pointwise convolution's own gradients from the same pass over y and g_small:
// dw_small [c][cs] (the layout of a [1,1,1,c,cs] filter) = coef_small * sum_v y[v][ch] * g_small[v][j], db_small [cs] = sum_v g_small[v][j].
// cs == 1 (the image layers of the 3-D networks); otherwise SG_EUNSUPPORTED.  dbias / dw_small / db_small: each optional.
extern "C" int sg_pixel_norm_act_bwd_pw_wg(const void* g_small, int32_t cs, const float* w_small, const void* y, const float* scale,
                                           const void* y_sign_words, float slope, void* dz, float* dbias, float* dw_small,
                                           float* db_small, float coef_small, void* workspace, size_t workspace_bytes, int64_t nvox,
                                           int32_t c, sg_dtype dt, sg_stream_t st) {
  if (!g_small || !w_small || !y || !scale || !y_sign_words || !dz || !workspace || nvox < 1 || c < 1 || cs < 1 || cs > 4) return SG_EINVAL;
  if (cs != 1) return SG_EUNSUPPORTED;
  if (workspace_bytes < sg_pixel_norm_act_bwd_pw_wg_workspace(c, cs)) return SG_EWORKSPACE;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const int P = c / E;
  if ((c % E) != 0 || P > 64 || !sg_aligned16(dz) || !sg_aligned16(y)) return SG_EUNSUPPORTED;
  const int tp = team_size(P);
  if (P > tp) return SG_EUNSUPPORTED;
  const int teams = 256 / tp;
  const int blocks = grid_for(nvox, teams, kBwdBlocks);
  float* part = reinterpret_cast<float*>(workspace);
#define L(T) hipLaunchKernelGGL((pixel_norm_kernel<T, true, 1, 2, 1, true>), dim3(blocks), dim3(256), 0, hs, (const T*)g_small, (const T*)y, scale, (T*)dz, (float*)nullptr, nvox, c, tp, 0.f, (const uint32_t*)y_sign_words, slope, part, w_small)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  const int cols = c * (1 + cs) + cs;
  hipLaunchKernelGGL(colsum_finalize3_kernel, dim3((cols + 31) / 32), dim3(1024), 0, hs, part, blocks, dbias, c, dw_small, c * cs, coef_small,
                     db_small, cs);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_pixel_norm_act_bwd_pw(const void* g_small, int32_t cs, const float* w_small, const void* y, const float* scale,
                                        const void* y_sign_words, float slope, void* dz, float* dbias, void* workspace,
                                        int64_t nvox, int32_t c, sg_dtype dt, sg_stream_t st) {
  if (!g_small || !w_small || !y || !scale || !y_sign_words || !dz || nvox < 1 || c < 1 || cs < 1 || cs > 4) return SG_EINVAL;
  if (dbias && !workspace) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const int P = c / E;
  if ((c % E) != 0 || P > 64 || !sg_aligned16(dz) || !sg_aligned16(y)) return SG_EUNSUPPORTED;
  const int tp = team_size(P);
  if (P > tp) return SG_EUNSUPPORTED;
  const int teams = 256 / tp;
  const int blocks = grid_for(nvox, teams, kBwdBlocks);
  float* part = dbias ? reinterpret_cast<float*>(workspace) : nullptr;
#define LC(T, CSV) hipLaunchKernelGGL((pixel_norm_kernel<T, true, 1, 2, CSV>), dim3(blocks), dim3(256), 0, hs, (const T*)g_small, (const T*)y, scale, (T*)dz, (float*)nullptr, nvox, c, tp, 0.f, (const uint32_t*)y_sign_words, slope, part, w_small)
#define L(T) do { switch (cs) { case 1: LC(T, 1); break; case 2: LC(T, 2); break; case 3: LC(T, 3); break; default: LC(T, 4); break; } } while (0)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
#undef LC
  SG_LAUNCH_CHECK();
  if (dbias) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((c + 31) / 32), dim3(1024), 0, hs, part, dbias, blocks, c);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

static int sg_factor_shift(int32_t f) { return f == 1 ? 0 : (f == 2 ? 1 : -1); }

static int upscale_nn_impl(const void* x, void* y, const void* mask_bits, float mask_slope, int32_t n, int32_t d,
                           int32_t h, int32_t w, int32_t c, int32_t fd, int32_t fh, int32_t fw, float gain,
                           int32_t plane_channels, sg_dtype dt, sg_stream_t st) {
  const int sd = sg_factor_shift(fd), sh = sg_factor_shift(fh), sw = sg_factor_shift(fw);
  if (!x || !y || n < 1 || d < 1 || h < 1 || w < 1 || c < 1 || sd < 0 || sh < 0 || sw < 0) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const bool vec = (c % E == 0) && sg_aligned16(x) && sg_aligned16(y);
  const bool planes = plane_channels != c;
  if (planes && (plane_channels < E || plane_channels % E != 0 || c % plane_channels != 0)) return SG_EINVAL;
  if (vec && (w << sw) * (c / E) >= 128) {   // enough pieces per output row to fill a block
    const int64_t nrows = (int64_t)n * (d << sd) * (h << sh);
    const int rb = (int)(nrows < 16384 ? nrows : 16384);
    const int64_t plane_stride = nrows * (w << sw) * plane_channels;
#define LR(T) do { if (sh == 1) hipLaunchKernelGGL((upscale2x_rows_kernel<T, 2>), dim3(rb), dim3(256), 0, hs, (const T*)x, (T*)y, d, h, w, c, sd, sh, sw, gain, (const uint32_t*)mask_bits, mask_slope, nrows, (int)plane_channels, plane_stride); \
                   else hipLaunchKernelGGL((upscale2x_rows_kernel<T, 1>), dim3(rb), dim3(256), 0, hs, (const T*)x, (T*)y, d, h, w, c, sd, sh, sw, gain, (const uint32_t*)mask_bits, mask_slope, nrows, (int)plane_channels, plane_stride); } while (0)
    SG_DISPATCH(dt, LR(bf16_t), LR(float));
#undef LR
    SG_LAUNCH_CHECK();
    return SG_OK;
  }
  if (planes) return SG_EUNSUPPORTED;   // only the row-wise kernel writes separate planes
  const int64_t items = (int64_t)n * (d << sd) * (h << sh) * (w << sw) * (vec ? c / E : c);
  const int blocks = grid_for(items, 256, 4096);
#define L(T, V) hipLaunchKernelGGL((upscale2x_kernel<T, V>), dim3(blocks), dim3(256), 0, hs, (const T*)x, (T*)y, n, d, h, w, c, sd, sh, sw, gain, (const uint32_t*)mask_bits, mask_slope)
  if (vec) SG_DISPATCH(dt, L(bf16_t, true), L(float, true));
  else SG_DISPATCH(dt, L(bf16_t, false), L(float, false));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_upscale_nn(const void* x, void* y, const void* mask_bits, float mask_slope, int32_t n, int32_t d,
                             int32_t h, int32_t w, int32_t c, int32_t fd, int32_t fh, int32_t fw, float gain,
                             sg_dtype dt, sg_stream_t st) {
  return upscale_nn_impl(x, y, mask_bits, mask_slope, n, d, h, w, c, fd, fh, fw, gain, c, dt, st);
}

extern "C" int sg_upscale_nn_planes(const void* x, void* y, const void* mask_bits, float mask_slope, int32_t n, int32_t d,
                                    int32_t h, int32_t w, int32_t c, int32_t fd, int32_t fh, int32_t fw, float gain,
                                    int32_t plane_channels, sg_dtype dt, sg_stream_t st) {
  return upscale_nn_impl(x, y, mask_bits, mask_slope, n, d, h, w, c, fd, fh, fw, gain, plane_channels, dt, st);
}

extern "C" int sg_upscale2x(const void* x, void* y, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                            float gain, sg_dtype dt, sg_stream_t st) {
  return sg_upscale_nn(x, y, nullptr, 0.f, n, d, h, w, c, 2, 2, 2, gain, dt, st);
}

extern "C" int sg_upscale2x_masked(const void* x, void* y, const void* mask_bits, float mask_slope, int32_t n, int32_t d,
                                   int32_t h, int32_t w, int32_t c, float gain, sg_dtype dt, sg_stream_t st) {
  return sg_upscale_nn(x, y, mask_bits, mask_slope, n, d, h, w, c, 2, 2, 2, gain, dt, st);
}

extern "C" int sg_downscale_sum(const void* x, void* y, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                                int32_t fd, int32_t fh, int32_t fw, float gain, sg_dtype dt, sg_stream_t st) {
  const int sd = sg_factor_shift(fd), sh = sg_factor_shift(fh), sw = sg_factor_shift(fw);
  if (!x || !y || n < 1 || d < fd || h < fh || w < fw || c < 1 || sd < 0 || sh < 0 || sw < 0) return SG_EINVAL;
  if ((sd && (d & 1)) || (sh && (h & 1)) || (sw && (w & 1))) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const bool vec = (c % E == 0) && sg_aligned16(x) && sg_aligned16(y);
  const int64_t items = (int64_t)n * (d >> sd) * (h >> sh) * (w >> sw) * (vec ? c / E : c);
  const int blocks = grid_for(items, 256, 4096);
#define L(T, V) hipLaunchKernelGGL((downscale2x_kernel<T, V>), dim3(blocks), dim3(256), 0, hs, (const T*)x, (T*)y, n, d, h, w, c, sd, sh, sw, gain)
  if (vec) SG_DISPATCH(dt, L(bf16_t, true), L(float, true));
  else SG_DISPATCH(dt, L(bf16_t, false), L(float, false));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_downscale_sum_masked(const void* x, const void* mask_bits, float mask_slope, void* y, int32_t n, int32_t d,
                                       int32_t h, int32_t w, int32_t c, int32_t fd, int32_t fh, int32_t fw, float gain,
                                       sg_dtype dt, sg_stream_t st) {
  const int sd = sg_factor_shift(fd), sh = sg_factor_shift(fh), sw = sg_factor_shift(fw);
  if (!x || !y || !mask_bits || n < 1 || d < fd || h < fh || w < fw || c < 1 || sd < 0 || sh < 0 || sw < 0) return SG_EINVAL;
  if ((sd && (d & 1)) || (sh && (h & 1)) || (sw && (w & 1))) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const bool vec = (c % E == 0) && sg_aligned16(x) && sg_aligned16(y);
  const int64_t items = (int64_t)n * (d >> sd) * (h >> sh) * (w >> sw) * (vec ? c / E : c);
  const int blocks = grid_for(items, 256, 4096);
#define L(T, V) hipLaunchKernelGGL((downscale2x_kernel<T, V, true>), dim3(blocks), dim3(256), 0, hs, (const T*)x, (T*)y, n, d, h, w, c, sd, sh, sw, gain, (const uint32_t*)mask_bits, mask_slope)
  if (vec) SG_DISPATCH(dt, L(bf16_t, true), L(float, true));
  else SG_DISPATCH(dt, L(bf16_t, false), L(float, false));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_downscale2x(const void* x, void* y, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                              float gain, sg_dtype dt, sg_stream_t st) {
  return sg_downscale_sum(x, y, n, d, h, w, c, 2, 2, 2, gain, dt, st);
}

extern "C" int sg_trilinear_up2x(const void* x, void* y, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                                 int32_t adjoint, sg_dtype dt, sg_stream_t st) {
  if (!x || !y || n < 1 || d < 1 || h < 1 || w < 1 || c < 1) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const bool vec = (c % E == 0) && sg_aligned16(x) && sg_aligned16(y);
  const int64_t items = (int64_t)n * d * h * w * (adjoint ? 1 : 8) * (vec ? c / E : c);
  const int blocks = grid_for(items, 256, 8192);
#define LU(T, V) hipLaunchKernelGGL((trilinear_up2x_kernel<T, V>), dim3(blocks), dim3(256), 0, hs, (const T*)x, (T*)y, n, d, h, w, c)
#define LA(T, V) hipLaunchKernelGGL((trilinear_up2x_adj_kernel<T, V>), dim3(blocks), dim3(256), 0, hs, (const T*)x, (T*)y, n, d, h, w, c)
  if (!adjoint) {
    if (vec) SG_DISPATCH(dt, LU(bf16_t, true), LU(float, true));
    else SG_DISPATCH(dt, LU(bf16_t, false), LU(float, false));
  } else {
    if (vec) SG_DISPATCH(dt, LA(bf16_t, true), LA(float, true));
    else SG_DISPATCH(dt, LA(bf16_t, false), LA(float, false));
  }
#undef LU
#undef LA
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_axpby(const void* a, const void* b, void* out, float wa, float wb, int64_t numel, sg_dtype dt,
                        sg_stream_t st) {
  if (!a || !out || numel < 1) return SG_EINVAL;
  if (!sg_aligned16(a) || !sg_aligned16(out) || (b && !sg_aligned16(b))) return SG_EALIGN;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const int blocks = grid_trips(numel / E + 1, 256, 2);
#define L(T) hipLaunchKernelGGL((axpby_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)a, (const T*)b, (T*)out, wa, wb, numel)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_axpby_dev(const void* a, const void* b, void* out, const float* w, int64_t numel, sg_dtype dt, sg_stream_t st) {
  if (!a || !out || !w || numel < 1) return SG_EINVAL;
  if (!sg_aligned16(a) || !sg_aligned16(out) || (b && !sg_aligned16(b))) return SG_EALIGN;
  hipStream_t hs = sg_st(st);
  const int E = dt == SG_BF16 ? 8 : 4;
  const int blocks = grid_trips(numel / E + 1, 256, 2);
#define L(T) hipLaunchKernelGGL((axpby_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)a, (const T*)b, (T*)out, 0.f, 0.f, numel, w)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_lerp_rows(const void* a, const void* b, const float* gamma, void* out, int32_t n, int64_t per_sample, sg_dtype dt,
                            sg_stream_t st) {
  if (!a || !b || !gamma || !out || n < 1 || per_sample < 1) return SG_EINVAL;
  if (!sg_aligned16(a) || !sg_aligned16(b) || !sg_aligned16(out)) return SG_EALIGN;
  hipStream_t hs = sg_st(st);
  const int64_t numel = (int64_t)n * per_sample;
  const int E = dt == SG_BF16 ? 8 : 4;
  const int blocks = grid_trips(numel / E + 1, 256, 2);
#define L(T) hipLaunchKernelGGL((lerp_rows_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)a, (const T*)b, gamma, (T*)out, per_sample, numel)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_add_noise(const void* x, void* out, float stddev, uint64_t seed, uint64_t offset, int64_t numel,
                            sg_dtype dt, sg_stream_t st) {
  if (!x || !out || numel < 1) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int blocks = grid_for((numel + 3) / 4);
#define L(T) hipLaunchKernelGGL((add_noise_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)x, (T*)out, stddev, seed, offset, numel)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

__global__ void counter_add_kernel(uint64_t* ctr, uint64_t inc) { *ctr += inc; }

extern "C" int sg_add_noise_dev(const void* x, void* out, float stddev, uint64_t seed, uint64_t* offset_dev, uint64_t bump,
                                int64_t numel, sg_dtype dt, sg_stream_t st) {
  if (!x || !out || !offset_dev || numel < 1) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int blocks = grid_for((numel + 3) / 4);
#define L(T) hipLaunchKernelGGL((add_noise_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)x, (T*)out, stddev, seed, (uint64_t)0, numel, (const uint64_t*)offset_dev)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  if (bump) {      // stream order: every block of the launch above has read the counter before this runs
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, hs, offset_dev, bump);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

extern "C" int sg_sumsq_ndhwc_keep_w(const void* g, float* out, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                                     sg_dtype dt, sg_stream_t st) {
  if (!g || !out || n < 1 || d < 1 || h < 1 || w < 1 || c < 1) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int dh = d * h;
  if (sg_cfg().deterministic) {
    int tw = 1;
    while (tw < w && tw < 256) tw <<= 1;
#define LO(T) hipLaunchKernelGGL((sumsq_keep_w_ordered_kernel<T>), dim3(n), dim3(1024), 0, hs, (const T*)g, out, dh, w, c, tw)
    SG_DISPATCH(dt, LO(bf16_t), LO(float));
#undef LO
    SG_LAUNCH_CHECK();
    return SG_OK;
  }
  hipError_t e = hipMemsetAsync(out, 0, (size_t)n * w * sizeof(float), hs);
  if (e != hipSuccess) return (int)e;
  int rows = (dh + 63) / 64;
  if (rows < 1) rows = 1;
  const int bx = (dh + rows - 1) / rows;
  const int threads = w >= 256 ? 256 : (w >= 128 ? 128 : 64);
#define L(T) hipLaunchKernelGGL((sumsq_keep_w_kernel<T>), dim3(bx, n), dim3(threads), 0, hs, (const T*)g, out, n, dh, w, c, rows)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_minibatch_stddev_fwd(const void* x, void* y, float* workspace, int32_t n, int64_t vox_per_sample,
                                       int32_t c, int32_t group_size, sg_dtype dt, sg_stream_t st) {
  if (!x || !y || !workspace || n < 1 || vox_per_sample < 1 || c < 1 || group_size < 1) return SG_EINVAL;
  const int group = group_size < n ? group_size : n;
  if (n % group != 0) return SG_EINVAL;  // tf.reshape at networks/ops.py:317 fails likewise
  const int mgroups = n / group;
  hipStream_t hs = sg_st(st);
  hipError_t e = hipMemsetAsync(workspace, 0, (size_t)mgroups * sizeof(float), hs);
  if (e != hipSuccess) return (int)e;
  const int64_t per_sample = vox_per_sample * c;
  const int bx = grid_for(per_sample, 256, 256);
#define L(T) hipLaunchKernelGGL((mbstd_stat_kernel<T>), dim3(bx, mgroups), dim3(256), 0, hs, (const T*)x, workspace, group, mgroups, per_sample)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  const int blocks = grid_for((int64_t)n * vox_per_sample * (c + 1));
#define L(T) hipLaunchKernelGGL((mbstd_concat_kernel<T>), dim3(blocks), dim3(256), 0, hs, (const T*)x, workspace, (T*)y, n, vox_per_sample, c, mgroups)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_minibatch_stddev_bwd(const void* dy, const void* x, void* dx, float* workspace, int32_t n,
                                       int64_t vox_per_sample, int32_t c, int32_t group_size, sg_dtype dt, sg_stream_t st) {
  if (!dy || !x || !dx || !workspace || n < 1 || vox_per_sample < 1 || c < 1 || group_size < 1) return SG_EINVAL;
  const int group = group_size < n ? group_size : n;
  if (n % group != 0) return SG_EINVAL;
  const int mgroups = n / group;
  hipStream_t hs = sg_st(st);
  hipError_t e = hipMemsetAsync(workspace, 0, (size_t)mgroups * sizeof(float), hs);
  if (e != hipSuccess) return (int)e;
  const int bs = grid_for(vox_per_sample, 256, 64);
#define L(T) hipLaunchKernelGGL((mbstd_bwd_sum_kernel<T>), dim3(bs, n), dim3(256), 0, hs, (const T*)dy, workspace, n, vox_per_sample, c, mgroups)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  const int bx = grid_for(vox_per_sample * c, 256, 1024);
#define L(T) hipLaunchKernelGGL((mbstd_bwd_kernel<T>), dim3(bx, mgroups), dim3(256), 0, hs, (const T*)dy, (const T*)x, workspace, (T*)dx, group, mgroups, vox_per_sample, c)
  SG_DISPATCH(dt, L(bf16_t), L(float));
#undef L
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_cast(const void* src, sg_dtype dt_src, void* dst, sg_dtype dt_dst, int64_t numel, sg_stream_t st) {
  if (!src || !dst || numel < 1) return SG_EINVAL;
  hipStream_t hs = sg_st(st);
  const int blocks = grid_for(numel);
  if (dt_src == SG_F32 && dt_dst == SG_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, hs, (const float*)src, (bf16_t*)dst, numel);
  else if (dt_src == SG_BF16 && dt_dst == SG_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, hs, (const bf16_t*)src, (float*)dst, numel);
  else if (dt_src == SG_F32 && dt_dst == SG_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, hs, (const float*)src, (float*)dst, numel);
  else if (dt_src == SG_BF16 && dt_dst == SG_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, hs, (const bf16_t*)src, (bf16_t*)dst, numel);
  else
    return SG_EINVAL;
  SG_LAUNCH_CHECK();
  return SG_OK;
}
