// Validation metrics of the training loop on the GPU:
//   SURFGAN_3D/metrics/swd.py:13-123      Laplacian pyramid, neighbourhood descriptors, sliced Wasserstein distance
//   SURFGAN_3D/metrics/skim_metrics.py:8-45   mean squared error, NRMSE, PSNR, SSIM (scikit-image definitions)
// All of it is bandwidth-bound f32 / f64 work on NCDHW volumes: one thread per output element, reads coalesced along
// the innermost extent, reductions as two deterministic stages (per-block partial sums in f64, then one block).
// The one sort (projections of every descriptor on every random direction, sorted per direction) is a bitonic network:
// 4096-element chunks in LDS, the strides above a chunk as global compare-exchange passes.
#include "common.h"

namespace {

constexpr int MAX_TAPS = 15;
struct FilterArgs {
  const void* x;
  void* y;
  const void* add;
  int64_t outer, inner;
  int32_t n, n_out, ntaps, mode, border;
  double alpha;
  double taps[MAX_TAPS];
};

// index into a line of n samples with scipy.ndimage's border rules: 0 'mirror' (d c b | a b c d | c b a),
// 1 'reflect' (d c b a | a b c d | d c b a)
__device__ __forceinline__ int sg_border(int i, int n, int border) {
  if (n == 1) return 0;
  if (border == 0) {
    const int period = 2 * (n - 1);
    i = i < 0 ? -i : i;
    i %= period;
    return i >= n ? period - i : i;
  }
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i >= n ? period - 1 - i : i;
}

template <typename T>
__global__ __launch_bounds__(256) void filter_axis_kernel(FilterArgs a) {
  const int64_t total = a.outer * a.n_out * a.inner;
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* add = reinterpret_cast<const T*>(a.add);
  T* y = reinterpret_cast<T*>(a.y);
  const int r = a.ntaps >> 1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t in_ = e % a.inner;
    const int64_t q = e / a.inner;
    const int o = (int)(q % a.n_out);
    const int64_t ou = q / a.n_out;
    const T* line = x + ou * a.n * a.inner + in_;
    double acc = 0.0;
    if (a.mode == 2) {            // zero insertion: z[2k] = x[k], z[odd] = 0, extent 2n
      for (int t = 0; t < a.ntaps; ++t) {
        const int zi = sg_border(o + t - r, 2 * a.n, a.border);
        if ((zi & 1) == 0) acc += a.taps[t] * (double)line[(int64_t)(zi >> 1) * a.inner];
      }
    } else {
      const int c = a.mode == 1 ? 2 * o : o;
      for (int t = 0; t < a.ntaps; ++t)
        acc += a.taps[t] * (double)line[(int64_t)sg_border(c + t - r, a.n, a.border) * a.inner];
    }
    acc *= a.alpha;
    if (add != nullptr) acc += (double)add[e];
    y[e] = (T)acc;
  }
}

// out[nh, c, i, j, k] = x[img, c, d0 + i - rd, h0 + k - rw, w0 + j - rh]   (swd.py:20-26: the reference's `x` offsets,
// 2*rh+1 of them on the FOURTH axis, are added to the W coordinate, its `y` offsets on the fifth to the H coordinate)
__global__ __launch_bounds__(256) void swd_gather_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         const int32_t* __restrict__ d0, const int32_t* __restrict__ h0,
                                                         const int32_t* __restrict__ w0, int64_t total, int c, int d, int h,
                                                         int w, int per_image, int rd, int rh, int rw) {
  const int ed = 2 * rd + 1, eh = 2 * rh + 1, ew = 2 * rw + 1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int k = (int)(e % ew);
    int64_t q = e / ew;
    const int j = (int)(q % eh);
    q /= eh;
    const int i = (int)(q % ed);
    q /= ed;
    const int ch = (int)(q % c);
    const int64_t nh = q / c;
    const int64_t img = nh / per_image;
    const int dd = d0[nh] + i - rd, hh = h0[nh] + k - rw, ww = w0[nh] + j - rh;
    out[e] = x[(((img * c + ch) * d + dd) * h + hh) * (int64_t)w + ww];
  }
}

__device__ __forceinline__ double sg_block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
  return s;   // valid in thread 0
}

// per-channel sum and sum of squares of desc [n, c, inner] -> part[block][2c] (f64)
__global__ __launch_bounds__(256) void desc_stats_kernel(const float* __restrict__ desc, double* __restrict__ part,
                                                         int64_t n, int c, int64_t inner) {
  __shared__ double sh[8];
  for (int ch = 0; ch < c; ++ch) {
    double s = 0.0, ss = 0.0;
    const int64_t cnt = n * inner;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < cnt; e += (int64_t)gridDim.x * 256) {
      const int64_t i = e / inner, r = e - i * inner;
      const double v = desc[(i * c + ch) * inner + r];
      s += v;
      ss += v * v;
    }
    const double S = sg_block_sum(s, sh);
    const double SS = sg_block_sum(ss, sh);
    if (threadIdx.x == 0) {
      part[((int64_t)blockIdx.x * c + ch) * 2 + 0] = S;
      part[((int64_t)blockIdx.x * c + ch) * 2 + 1] = SS;
    }
  }
}

// stats[ch] = (mean, 1 / population std) from the per-block partial sums, in block order
__global__ __launch_bounds__(64) void desc_stats_final_kernel(const double* __restrict__ part, double* __restrict__ stats,
                                                              int blocks, int c, double count) {
  const int ch = blockIdx.x;
  if (threadIdx.x != 0) return;
  double s = 0.0, ss = 0.0;
  for (int b = 0; b < blocks; ++b) {
    s += part[((int64_t)b * c + ch) * 2 + 0];
    ss += part[((int64_t)b * c + ch) * 2 + 1];
  }
  const double mean = s / count;
  const double var = ss / count - mean * mean;
  stats[2 * ch + 0] = mean;
  stats[2 * ch + 1] = 1.0 / sqrt(var > 0.0 ? var : 0.0);
}

__global__ __launch_bounds__(256) void desc_normalize_kernel(float* __restrict__ desc, const double* __restrict__ stats,
                                                             int64_t total, int c, int64_t inner) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int ch = (int)((e / inner) % c);
    desc[e] = (float)(((double)desc[e] - stats[2 * ch]) * stats[2 * ch + 1]);
  }
}

// pt[dir][row] = sum_f a[row][f] * dirs[f][dir]: 64 rows x 64 directions per block, 4 x 4 per thread, K in steps of 16
// through LDS; rows n .. npad-1 are +inf (they sort to the end of the row and are never compared).
__global__ __launch_bounds__(256) void swd_project_kernel(const float* __restrict__ a, const float* __restrict__ dirs,
                                                          float* __restrict__ pt, int n, int f, int ndirs, int npad) {
  __shared__ float As[16][65];   // [k][row]
  __shared__ float Bs[16][64];   // [k][dir]
  const int row0 = blockIdx.x * 64, dir0 = blockIdx.y * 64;
  const int tid = threadIdx.x, tr = tid & 15, tc = tid >> 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < f; k0 += 16) {
    for (int i = tid; i < 64 * 16; i += 256) {
      const int rr = i >> 4, kk = i & 15;
      const int row = row0 + rr, k = k0 + kk;
      As[kk][rr] = (row < n && k < f) ? a[(int64_t)row * f + k] : 0.f;
      const int kk2 = i >> 6, cc = i & 63;
      const int k2 = k0 + kk2, dir = dir0 + cc;
      Bs[kk2][cc] = (k2 < f && dir < ndirs) ? dirs[(int64_t)k2 * ndirs + dir] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[kk][tr * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[kk][tc * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  const float inf = __builtin_inff();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int dir = dir0 + tc * 4 + j;
    if (dir >= ndirs) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + tr * 4 + i;
      if (row < npad) pt[(int64_t)dir * npad + row] = row < n ? acc[i][j] : inf;
    }
  }
}

// bitonic network over rows of npad (a power of two) floats, ascending.  A block owns one chunk (<= 4096 elements of
// one row) in LDS.  FULL: every stage with k <= chunk.  Otherwise: the strides j < chunk of stage k.
constexpr int SORT_CHUNK = 4096;
template <bool FULL>
__global__ __launch_bounds__(1024) void bitonic_local_kernel(float* __restrict__ data, int chunk, int npad, int k_stage) {
  __shared__ float sh[SORT_CHUNK];
  const int64_t base = (int64_t)blockIdx.x * chunk;          // rows are npad long and npad % chunk == 0
  float* p = data + base;
  for (int i = threadIdx.x; i < chunk; i += 1024) sh[i] = p[i];
  __syncthreads();
  const int pairs = chunk >> 1;
  const uint32_t gbase = (uint32_t)(base & (int64_t)(npad - 1));   // position inside the row decides the direction
  for (int k = FULL ? 2 : k_stage; k <= (FULL ? chunk : k_stage); k <<= 1) {
    for (int j = (k >> 1) < chunk ? (k >> 1) : (chunk >> 1); j > 0; j >>= 1) {
      for (int q = threadIdx.x; q < pairs; q += 1024) {
        const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
        const int l = i | j;
        const bool up = ((gbase + (uint32_t)i) & (uint32_t)k) == 0;
        const float x0 = sh[i], x1 = sh[l];
        if ((x0 > x1) == up) { sh[i] = x1; sh[l] = x0; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < chunk; i += 1024) p[i] = sh[i];
}

__global__ __launch_bounds__(256) void bitonic_global_kernel(float* __restrict__ data, int64_t pairs, int npad, int k, int j) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < pairs; q += (int64_t)gridDim.x * 256) {
    const int64_t i = ((q & ~(int64_t)(j - 1)) << 1) | (q & (j - 1));
    const int64_t l = i | j;
    const bool up = ((i & (npad - 1)) & k) == 0;
    const float x0 = data[i], x1 = data[l];
    if ((x0 > x1) == up) { data[i] = x1; data[l] = x0; }
  }
}

// out[1 + row] = sum_{i < n} |pa[row][i] - pb[row][i]|
__global__ __launch_bounds__(256) void swd_rowdist_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                          double* __restrict__ out, int n, int npad) {
  __shared__ double sh[8];
  const float* ra = pa + (int64_t)blockIdx.x * npad;
  const float* rb = pb + (int64_t)blockIdx.x * npad;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)fabsf(ra[i] - rb[i]);
  const double S = sg_block_sum(s, sh);
  if (threadIdx.x == 0) out[1 + blockIdx.x] = S;
}

// out[0] = scale * sum(part[0 .. count))   (one block, fixed order)
__global__ __launch_bounds__(256) void sum_final_kernel(const double* __restrict__ part, double* __restrict__ out, int count,
                                                        double scale) {
  __shared__ double sh[8];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) s += part[i];
  const double S = sg_block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = S * scale;
}

__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                             double* __restrict__ part, int64_t numel) {
  __shared__ double sh[8];
  double s = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += (int64_t)gridDim.x * 256) {
    const double d = a[e] - b[e];
    s += d * d;
  }
  const double S = sg_block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = S;
}

__global__ __launch_bounds__(256) void minmax_partial_kernel(const double* __restrict__ a, double* __restrict__ part,
                                                             int64_t numel) {
  __shared__ double lo_s[4], hi_s[4];
  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += (int64_t)gridDim.x * 256) {
    const double v = a[e];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double l2 = __shfl_down(lo, o), h2 = __shfl_down(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { lo_s[threadIdx.x >> 6] = lo; hi_s[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) { lo = lo_s[i] < lo ? lo_s[i] : lo; hi = hi_s[i] > hi ? hi_s[i] : hi; }
    part[2 * blockIdx.x] = lo;
    part[2 * blockIdx.x + 1] = hi;
  }
}

__global__ __launch_bounds__(64) void minmax_final_kernel(const double* __restrict__ part, double* __restrict__ out, int blocks) {
  if (threadIdx.x != 0) return;
  double lo = part[0], hi = part[1];
  for (int b = 1; b < blocks; ++b) {
    lo = part[2 * b] < lo ? part[2 * b] : lo;
    hi = part[2 * b + 1] > hi ? part[2 * b + 1] : hi;
  }
  out[0] = lo;
  out[1] = hi;
}

__global__ __launch_bounds__(256) void ssim_products_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                            double* __restrict__ xx, double* __restrict__ yy,
                                                            double* __restrict__ xy, int64_t numel) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += (int64_t)gridDim.x * 256) {
    const double a = x[e], b = y[e];
    xx[e] = a * a;
    yy[e] = b * b;
    xy[e] = a * b;
  }
}

// sum of the SSIM map over the voxels at least `crop` away from every border of [s0, s1, s2, c] (channels last)
__global__ __launch_bounds__(256) void ssim_partial_kernel(const double* __restrict__ ux, const double* __restrict__ uy,
                                                           const double* __restrict__ uxx, const double* __restrict__ uyy,
                                                           const double* __restrict__ uxy, double* __restrict__ part,
                                                           int s0, int s1, int s2, int c, int crop0, int crop, double cov_norm,
                                                           double C1, double C2) {
  __shared__ double sh[8];
  const int64_t numel = (int64_t)s0 * s1 * s2 * c;
  double s = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += (int64_t)gridDim.x * 256) {
    int64_t q = e / c;
    const int i2 = (int)(q % s2);
    q /= s2;
    const int i1 = (int)(q % s1);
    const int i0 = (int)(q / s1);
    if (i0 < crop0 || i0 >= s0 - crop0 || i1 < crop || i1 >= s1 - crop || i2 < crop || i2 >= s2 - crop) continue;
    const double mx = ux[e], my = uy[e];
    const double vx = cov_norm * (uxx[e] - mx * mx), vy = cov_norm * (uyy[e] - my * my), vxy = cov_norm * (uxy[e] - mx * my);
    s += ((2.0 * mx * my + C1) * (2.0 * vxy + C2)) / ((mx * mx + my * my + C1) * (vx + vy + C2));
  }
  const double S = sg_block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = S;
}

inline int grid_for(int64_t total, int per_block = 256, int cap = 4096) {
  int64_t g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" {

int sg_filter_axis(const void* x, void* y, const void* add, int64_t outer, int32_t n, int64_t inner, const double* taps,
                   int32_t ntaps, int32_t mode, int32_t border, double alpha, int32_t f64, sg_stream_t st) {
  if (x == nullptr || y == nullptr || taps == nullptr || x == y) return SG_EINVAL;
  if (outer < 1 || n < 1 || inner < 1 || ntaps < 1 || ntaps > MAX_TAPS || (ntaps & 1) == 0) return SG_EINVAL;
  if (mode < 0 || mode > 2 || border < 0 || border > 1 || n > (1 << 28)) return SG_EINVAL;
  FilterArgs a;
  a.x = x; a.y = y; a.add = add; a.outer = outer; a.inner = inner; a.n = n;
  a.n_out = mode == 1 ? (n + 1) / 2 : (mode == 2 ? 2 * n : n);
  a.ntaps = ntaps; a.mode = mode; a.border = border; a.alpha = alpha;
  for (int t = 0; t < MAX_TAPS; ++t) a.taps[t] = t < ntaps ? taps[t] : 0.0;
  const int64_t total = outer * a.n_out * inner;
  if (f64) hipLaunchKernelGGL(filter_axis_kernel<double>, dim3(grid_for(total, 256, 16384)), dim3(256), 0, sg_st(st), a);
  else hipLaunchKernelGGL(filter_axis_kernel<float>, dim3(grid_for(total, 256, 16384)), dim3(256), 0, sg_st(st), a);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

int sg_swd_gather(const float* x, float* out, const int32_t* d0, const int32_t* h0, const int32_t* w0, int32_t n_img,
                  int32_t c, int32_t d, int32_t h, int32_t w, int32_t per_image, int32_t rd, int32_t rh, int32_t rw,
                  sg_stream_t st) {
  if (x == nullptr || out == nullptr || d0 == nullptr || h0 == nullptr || w0 == nullptr) return SG_EINVAL;
  if (n_img < 1 || c < 1 || per_image < 1 || rd < 0 || rh < 0 || rw < 0) return SG_EINVAL;
  // a centre is drawn from [r, extent - r): the neighbourhood must fit (swd.py:22-24; the W centre carries the 2*rh+1
  // offsets and the H centre the 2*rw+1 ones)
  if (d < 2 * rd + 1 || w < 2 * rh + 1 || h < 2 * rw + 1) return SG_EINVAL;
  const int64_t total = (int64_t)n_img * per_image * c * (2 * rd + 1) * (2 * rh + 1) * (2 * rw + 1);
  hipLaunchKernelGGL(swd_gather_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, sg_st(st), x, out, d0, h0, w0, total, c,
                     d, h, w, per_image, rd, rh, rw);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

size_t sg_desc_normalize_workspace(int32_t c) { return ((size_t)256 * c * 2 + (size_t)2 * c) * sizeof(double); }

int sg_desc_normalize(float* desc, int64_t n, int32_t c, int64_t inner, void* workspace, size_t workspace_bytes, sg_stream_t st) {
  if (desc == nullptr || workspace == nullptr || n < 1 || c < 1 || inner < 1) return SG_EINVAL;
  if (workspace_bytes < sg_desc_normalize_workspace(c)) return SG_EWORKSPACE;
  double* part = reinterpret_cast<double*>(workspace);
  double* stats = part + (size_t)256 * c * 2;
  const int blocks = grid_for(n * inner, 256, 256);
  hipLaunchKernelGGL(desc_stats_kernel, dim3(blocks), dim3(256), 0, sg_st(st), desc, part, n, c, inner);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(desc_stats_final_kernel, dim3(c), dim3(64), 0, sg_st(st), part, stats, blocks, c, (double)(n * inner));
  SG_LAUNCH_CHECK();
  const int64_t total = n * c * inner;
  hipLaunchKernelGGL(desc_normalize_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, sg_st(st), desc, stats, total, c, inner);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

int32_t sg_swd_padded_rows(int32_t n) {
  int32_t p = 64;
  while (p < n && p < (1 << 30)) p <<= 1;
  return p;
}

int sg_swd_project(const float* a, const float* dirs, float* pt, int32_t n, int32_t f, int32_t ndirs, int32_t npad,
                   sg_stream_t st) {
  if (a == nullptr || dirs == nullptr || pt == nullptr || n < 1 || f < 1 || ndirs < 1) return SG_EINVAL;
  if (npad != sg_swd_padded_rows(n)) return SG_EINVAL;
  hipLaunchKernelGGL(swd_project_kernel, dim3(npad / 64, (ndirs + 63) / 64), dim3(256), 0, sg_st(st), a, dirs, pt, n, f, ndirs, npad);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

int sg_sort_rows(float* data, int32_t rows, int32_t npad, sg_stream_t st) {
  if (data == nullptr || rows < 1 || npad < 2 || (npad & (npad - 1)) != 0) return SG_EINVAL;
  const int chunk = npad < SORT_CHUNK ? npad : SORT_CHUNK;
  const int64_t nchunks = (int64_t)rows * (npad / chunk);
  if (nchunks > 0x7FFFFFFF) return SG_EINVAL;
  hipLaunchKernelGGL(bitonic_local_kernel<true>, dim3((unsigned)nchunks), dim3(1024), 0, sg_st(st), data, chunk, npad, 0);
  SG_LAUNCH_CHECK();
  const int64_t pairs = (int64_t)rows * npad / 2;
  for (int k = 2 * chunk; k <= npad && k > 0; k <<= 1) {
    for (int j = k >> 1; j >= chunk; j >>= 1) {
      hipLaunchKernelGGL(bitonic_global_kernel, dim3(grid_for(pairs, 256, 65536)), dim3(256), 0, sg_st(st), data, pairs, npad, k, j);
      SG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bitonic_local_kernel<false>, dim3((unsigned)nchunks), dim3(1024), 0, sg_st(st), data, chunk, npad, k);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

int sg_swd_distance(const float* pa, const float* pb, double* out, int32_t rows, int32_t n, int32_t npad, sg_stream_t st) {
  if (pa == nullptr || pb == nullptr || out == nullptr || rows < 1 || n < 1 || npad < n) return SG_EINVAL;
  hipLaunchKernelGGL(swd_rowdist_kernel, dim3(rows), dim3(256), 0, sg_st(st), pa, pb, out, n, npad);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, sg_st(st), out + 1, out, rows, 1.0 / ((double)rows * (double)n));
  SG_LAUNCH_CHECK();
  return SG_OK;
}

size_t sg_metric_workspace(void) { return (size_t)1024 * 2 * sizeof(double); }

int sg_sqdiff_mean(const double* a, const double* b, double* out, int64_t numel, void* workspace, size_t workspace_bytes,
                   sg_stream_t st) {
  if (a == nullptr || b == nullptr || out == nullptr || workspace == nullptr || numel < 1) return SG_EINVAL;
  if (workspace_bytes < sg_metric_workspace()) return SG_EWORKSPACE;
  double* part = reinterpret_cast<double*>(workspace);
  const int blocks = grid_for(numel, 256 * 8, 1024);
  hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(blocks), dim3(256), 0, sg_st(st), a, b, part, numel);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, sg_st(st), part, out, blocks, 1.0 / (double)numel);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

int sg_minmax(const double* a, double* out, int64_t numel, void* workspace, size_t workspace_bytes, sg_stream_t st) {
  if (a == nullptr || out == nullptr || workspace == nullptr || numel < 1) return SG_EINVAL;
  if (workspace_bytes < sg_metric_workspace()) return SG_EWORKSPACE;
  double* part = reinterpret_cast<double*>(workspace);
  const int blocks = grid_for(numel, 256 * 8, 1024);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(blocks), dim3(256), 0, sg_st(st), a, part, numel);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, sg_st(st), part, out, blocks);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

int sg_ssim_products(const double* x, const double* y, double* xx, double* yy, double* xy, int64_t numel, sg_stream_t st) {
  if (x == nullptr || y == nullptr || xx == nullptr || yy == nullptr || xy == nullptr || numel < 1) return SG_EINVAL;
  hipLaunchKernelGGL(ssim_products_kernel, dim3(grid_for(numel, 256, 16384)), dim3(256), 0, sg_st(st), x, y, xx, yy, xy, numel);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

int sg_ssim_mean(const double* ux, const double* uy, const double* uxx, const double* uyy, const double* uxy, double* out,
                 int32_t s0, int32_t s1, int32_t s2, int32_t c, int32_t crop0, int32_t crop, double cov_norm, double c1,
                 double c2, void* workspace, size_t workspace_bytes, sg_stream_t st) {
  if (ux == nullptr || uy == nullptr || uxx == nullptr || uyy == nullptr || uxy == nullptr || out == nullptr ||
      workspace == nullptr)
    return SG_EINVAL;
  if (s0 < 1 || s1 < 1 || s2 < 1 || c < 1 || crop0 < 0 || crop < 0) return SG_EINVAL;
  if (s0 <= 2 * crop0 || s1 <= 2 * crop || s2 <= 2 * crop) return SG_EINVAL;   // the cropped map would be empty
  if (workspace_bytes < sg_metric_workspace()) return SG_EWORKSPACE;
  double* part = reinterpret_cast<double*>(workspace);
  const int64_t numel = (int64_t)s0 * s1 * s2 * c;
  const int blocks = grid_for(numel, 256 * 4, 1024);
  hipLaunchKernelGGL(ssim_partial_kernel, dim3(blocks), dim3(256), 0, sg_st(st), ux, uy, uxx, uyy, uxy, part, s0, s1, s2, c, crop0,
                     crop, cov_norm, c1, c2);
  SG_LAUNCH_CHECK();
  const double count = (double)(s0 - 2 * crop0) * (double)(s1 - 2 * crop) * (double)(s2 - 2 * crop) * (double)c;
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, sg_st(st), part, out, blocks, 1.0 / count);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

}  // extern "C"
