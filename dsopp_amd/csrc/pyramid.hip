// Image pyramid kernels (hot loop G of SURVEY.md §3.3 / §8 a20) and the dsopp_hip_pyramid_* entry points.
//
// Replaces PixelDataFrame's constructor chain (src/features/src/pixel_data_frame.cpp:12-31):
//   level 0  = LUT[u8] * vmax / (vignette + 1)           photometrically_corrected_image.cpp:9-29
//   level l  = 0.25 * sum of the 2x2 block of level l-1   downscale_image.hpp:16-33 (from the scalar plane)
//   texel    = (I, 0.5*(I[x+1]-I[x-1]), 0.5*(I[y+1]-I[y-1])), one-sided (x1.0) at the borders
//                                                          calculate_pixelinfo.cpp:340-374
// One pure streaming kernel per level: every output texel needs its 4-neighbourhood of the level's scalar plane, which
// each thread regenerates from the parent plane (u8 + LUT at level 0, 2x2 means above) — reads are coalesced rows that
// hit L2, writes are one full 16/32-byte texel per lane.  HBM-bound by construction: ~(1 + 4*sizeof(S)) bytes per pixel.
#include "pyramid.hpp"
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

namespace dsopp_hip {

std::string &lastError() {
  static thread_local std::string e;
  return e;
}

namespace {

constexpr int kTileX = 64, kTileY = 4;

template <typename S>
__device__ __forceinline__ S level0Value(const uint8_t *__restrict__ img, const uint8_t *__restrict__ vig, const double *__restrict__ lut,
                                         double vmax, int W, int x, int y) {
  const size_t i = static_cast<size_t>(y) * W + x;
  S v = lut ? static_cast<S>(lut[img[i]]) : static_cast<S>(img[i]);
  if (vig) v *= static_cast<S>(vmax) / (static_cast<S>(vig[i]) + S(1));
  return v;
}

/** level 0: u8 (+LUT, +vignette) -> scalar plane + texels */
template <typename S>
__global__ void __launch_bounds__(kTileX *kTileY) pyramidLevel0Kernel(const uint8_t *__restrict__ img, const uint8_t *__restrict__ vig,
                                                                      const double *__restrict__ lut, double vmax, int W, int H,
                                                                      S *__restrict__ plane, Texel<S> *__restrict__ tex) {
  const int x = blockIdx.x * kTileX + threadIdx.x;
  const int y = blockIdx.y * kTileY + threadIdx.y;
  if (x >= W || y >= H) return;
  const S c = level0Value<S>(img, vig, lut, vmax, W, x, y);
  const int xm = x > 0 ? x - 1 : x, xp = x < W - 1 ? x + 1 : x;
  const int ym = y > 0 ? y - 1 : y, yp = y < H - 1 ? y + 1 : y;
  const S l = (xm == x) ? c : level0Value<S>(img, vig, lut, vmax, W, xm, y);
  const S r = (xp == x) ? c : level0Value<S>(img, vig, lut, vmax, W, xp, y);
  const S u = (ym == y) ? c : level0Value<S>(img, vig, lut, vmax, W, x, ym);
  const S d = (yp == y) ? c : level0Value<S>(img, vig, lut, vmax, W, x, yp);
  const S sx = (x == 0 || x == W - 1) ? S(1) : S(0.5);
  const S sy = (y == 0 || y == H - 1) ? S(1) : S(0.5);
  const size_t i = static_cast<size_t>(y) * W + x;
  plane[i] = c;
  Texel<S> t;
  t.I = c;
  t.mask = tex[i].mask;  // mask lane is owned by set_mask (initialised to 1)
  t.Ix = sx * (r - l);
  t.Iy = sy * (d - u);
  tex[i] = t;
}

template <typename S>
__device__ __forceinline__ S boxValue(const S *__restrict__ parent, int Wp, int x, int y) {
  const size_t i00 = static_cast<size_t>(2 * y) * Wp + 2 * x;
  // same association order as the reference expression: ((p00 + p11) + p01) + p10
  return S(0.25) * (parent[i00] + parent[i00 + Wp + 1] + parent[i00 + 1] + parent[i00 + Wp]);
}

/** level l >= 1: parent scalar plane -> scalar plane + texels */
template <typename S>
__global__ void __launch_bounds__(kTileX *kTileY) pyramidLevelKernel(const S *__restrict__ parent, int Wp, int W, int H, S *__restrict__ plane,
                                                                     Texel<S> *__restrict__ tex) {
  const int x = blockIdx.x * kTileX + threadIdx.x;
  const int y = blockIdx.y * kTileY + threadIdx.y;
  if (x >= W || y >= H) return;
  const S c = boxValue<S>(parent, Wp, x, y);
  const S l = (x == 0) ? c : boxValue<S>(parent, Wp, x - 1, y);
  const S r = (x == W - 1) ? c : boxValue<S>(parent, Wp, x + 1, y);
  const S u = (y == 0) ? c : boxValue<S>(parent, Wp, x, y - 1);
  const S d = (y == H - 1) ? c : boxValue<S>(parent, Wp, x, y + 1);
  const S sx = (x == 0 || x == W - 1) ? S(1) : S(0.5);
  const S sy = (y == 0 || y == H - 1) ? S(1) : S(0.5);
  const size_t i = static_cast<size_t>(y) * W + x;
  plane[i] = c;
  Texel<S> t;
  t.I = c;
  t.mask = tex[i].mask;
  t.Ix = sx * (r - l);
  t.Iy = sy * (d - u);
  tex[i] = t;
}

// ---- all levels in ONE launch (no vignette).  Level 0 as before, one thread per texel.  The upper levels by "region" workgroups:
// a workgroup owns kUpperTile x kUpperTile texels of the TOP level and everything below them — it samples the 8-bit image once
// into the level-1 values of its region (+ the halo the gradients of the levels above need), forms levels 2 .. TOP from those in
// LDS by the same 2 x 2 means in the same association order as the level-by-level chain (scaling by 0.25 is exact: the texels are
// bit-identical), and writes the texels of every level inside its region.  No level waits for its parent's launch: the chain of 5
// dependent kernels (38 us at 1280 x 1024, the four upper ones pure launch latency) becomes one kernel as long as level 0.
constexpr int kUpperTile = 4;

struct AllLevelsArgs {
  const uint8_t *img;
  const double *lut;
  int levels;
  int width[DSOPP_HIP_MAX_LEVELS], height[DSOPP_HIP_MAX_LEVELS];
  int tiles0_x;                    // level-0 workgroups per row (kTileX x kTileY texels each)
  int n_upper, upper_tiles_x;      // region workgroups come first in the grid
  void *tex[DSOPP_HIP_MAX_LEVELS];
  void *plane[DSOPP_HIP_MAX_LEVELS];
};

template <typename S>
__device__ __forceinline__ S sample0(const AllLevelsArgs &a, int x, int y) {
  const uint8_t v = a.img[static_cast<size_t>(y) * a.width[0] + x];
  return a.lut ? static_cast<S>(a.lut[v]) : static_cast<S>(v);
}

template <typename S>
__device__ __forceinline__ void storeTexel(const AllLevelsArgs &a, int lvl, int x, int y, S c, S l, S r, S u, S d) {
  const int W = a.width[lvl], H = a.height[lvl];
  const S sx = (x == 0 || x == W - 1) ? S(1) : S(0.5);
  const S sy = (y == 0 || y == H - 1) ? S(1) : S(0.5);
  const size_t i = static_cast<size_t>(y) * W + x;
  Texel<S> *tex = static_cast<Texel<S> *>(a.tex[lvl]);
  static_cast<S *>(a.plane[lvl])[i] = c;  // (kept current: a later chain build or set_level reads the parents' planes)
  Texel<S> t;
  t.I = c;
  t.mask = tex[i].mask;  // mask lane is owned by set_mask (initialised to 1)
  t.Ix = sx * (r - l);
  t.Iy = sy * (d - u);
  tex[i] = t;
}

/** region workgroup, TOP = index of the coarsest level (1 .. 4).  Level l of the region: n_l = (kUpperTile + 2) << (TOP - l) values per
 *  side, origin o_l = (tile * kUpperTile - 1) << (TOP - l) in level-l pixels, so that value (rx, ry) of level l is the mean of the
 *  values (2 rx .. 2 rx + 1, 2 ry .. 2 ry + 1) of level l - 1. */
template <typename S, int TOP>
__device__ __forceinline__ void upperLevelsRegion(const AllLevelsArgs &a, int tile, S *lds) {
  constexpr int kThreads = kTileX * kTileY;
  const int tid = threadIdx.y * kTileX + threadIdx.x;
  const int tx = tile % a.upper_tiles_x, ty = tile / a.upper_tiles_x;
  S *v[TOP + 1];
  {
    S *p = lds;
#pragma unroll
    for (int l = 1; l <= TOP; ++l) {
      v[l] = p;
      const int n = (kUpperTile + 2) << (TOP - l);
      p += n * n;
    }
  }
  // ---- level 1 from the image: every value once, independent loads
  {
    constexpr int n1 = (kUpperTile + 2) << (TOP - 1);
    const int ox = (tx * kUpperTile - 1) * (1 << (TOP - 1)), oy = (ty * kUpperTile - 1) * (1 << (TOP - 1));
    const int W1 = a.width[1], H1 = a.height[1];
    for (int idx = tid; idx < n1 * n1; idx += kThreads) {
      const int rx = idx % n1, ry = idx / n1;
      const int X = ox + rx, Y = oy + ry;
      S val = S(0);
      if (X >= 0 && Y >= 0 && X < W1 && Y < H1) {
        // boxValue's order: ((p00 + p11) + p01) + p10 with p01 = (2X + 1, 2Y), p10 = (2X, 2Y + 1)
        const S p00 = sample0<S>(a, 2 * X, 2 * Y), p11 = sample0<S>(a, 2 * X + 1, 2 * Y + 1);
        const S p01 = sample0<S>(a, 2 * X + 1, 2 * Y), p10 = sample0<S>(a, 2 * X, 2 * Y + 1);
        val = S(0.25) * (p00 + p11 + p01 + p10);
      }
      v[1][idx] = val;
    }
  }
  __syncthreads();
#pragma unroll
  for (int l = 1; l <= TOP; ++l) {
    const int n = (kUpperTile + 2) << (TOP - l);
    if (l > 1) {
      const int np = 2 * n;
      for (int idx = tid; idx < n * n; idx += kThreads) {
        const int rx = idx % n, ry = idx / n;
        const S *q = v[l - 1] + (2 * ry) * np + 2 * rx;
        v[l][idx] = S(0.25) * (q[0] + q[np + 1] + q[1] + q[np]);
      }
      __syncthreads();
    }
    // texels of level l inside the region (the halo belongs to the neighbours)
    const int h = 1 << (TOP - l), m = kUpperTile << (TOP - l);
    const int ox = (tx * kUpperTile - 1) * h, oy = (ty * kUpperTile - 1) * h;
    const int W = a.width[l], H = a.height[l];
    for (int idx = tid; idx < m * m; idx += kThreads) {
      const int rx = h + idx % m, ry = h + idx / m;
      const int X = ox + rx, Y = oy + ry;
      if (X >= W || Y >= H) continue;
      const S *q = v[l] + ry * n + rx;
      const S c = q[0];
      const S lft = (X == 0) ? c : q[-1], rgt = (X == W - 1) ? c : q[1];
      const S up = (Y == 0) ? c : q[-n], dn = (Y == H - 1) ? c : q[n];
      storeTexel<S>(a, l, X, Y, c, lft, rgt, up, dn);
    }
  }
}

template <typename S>
__global__ void __launch_bounds__(kTileX *kTileY) pyramidAllLevelsKernel(AllLevelsArgs a) {
  // LDS of a region workgroup at TOP = 4: (48^2 + 24^2 + 12^2 + 6^2) values
  __shared__ S lds[((kUpperTile + 2) * 8) * ((kUpperTile + 2) * 8) + ((kUpperTile + 2) * 4) * ((kUpperTile + 2) * 4) +
                   ((kUpperTile + 2) * 2) * ((kUpperTile + 2) * 2) + (kUpperTile + 2) * (kUpperTile + 2)];
  const int b = blockIdx.x;
  if (b < a.n_upper) {  // the region workgroups first: they run longer, level 0 fills the chip behind them
    switch (a.levels) {
      case 5: upperLevelsRegion<S, 4>(a, b, lds); break;
      case 4: upperLevelsRegion<S, 3>(a, b, lds); break;
      case 3: upperLevelsRegion<S, 2>(a, b, lds); break;
      default: upperLevelsRegion<S, 1>(a, b, lds); break;
    }
    return;
  }
  const int t0 = b - a.n_upper;
  const int W = a.width[0], H = a.height[0];
  const int x = (t0 % a.tiles0_x) * kTileX + threadIdx.x, y = (t0 / a.tiles0_x) * kTileY + threadIdx.y;
  if (x >= W || y >= H) return;
  const S c = sample0<S>(a, x, y);
  const S l = (x == 0) ? c : sample0<S>(a, x - 1, y), r = (x == W - 1) ? c : sample0<S>(a, x + 1, y);
  const S u = (y == 0) ? c : sample0<S>(a, x, y - 1), d = (y == H - 1) ? c : sample0<S>(a, x, y + 1);
  storeTexel<S>(a, 0, x, y, c, l, r, u, d);
}

template <typename S>
__global__ void fillMaskKernel(Texel<S> *tex, const uint8_t *mask, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) tex[i].mask = mask ? (mask[i] ? S(1) : S(0)) : S(1);
}

template <typename S>
__global__ void setLevelKernel(Texel<S> *tex, S *plane, const double *pixelinfo, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Texel<S> t;
  t.I = static_cast<S>(pixelinfo[3 * i]);
  t.mask = tex[i].mask;
  t.Ix = static_cast<S>(pixelinfo[3 * i + 1]);
  t.Iy = static_cast<S>(pixelinfo[3 * i + 2]);
  tex[i] = t;
  plane[i] = t.I;
}

template <typename S>
__global__ void getLevelKernel(const Texel<S> *tex, double *pixelinfo, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pixelinfo[3 * i] = static_cast<double>(tex[i].I);
  pixelinfo[3 * i + 1] = static_cast<double>(tex[i].Ix);
  pixelinfo[3 * i + 2] = static_cast<double>(tex[i].Iy);
}

/** texels -> tiled intensity plane (pyramid.hpp): one thread per pixel of the padded tile grid.
 *  f64: 8-byte words, 4 x 2 pixels per 64-byte tile; f32: 4-byte words, 4 x 4 pixels per 64-byte tile.  CameraMask bit in the lowest mantissa bit. */
__global__ void buildIntensityPlaneKernel(const Texel<double> *__restrict__ tex, int W, int H, int tiles_x, int tiles_y,
                                          unsigned long long *__restrict__ out) {
  const int x = blockIdx.x * kTileX + threadIdx.x;
  const int y = blockIdx.y * kTileY + threadIdx.y;
  if (x >= 4 * tiles_x || y >= 2 * tiles_y) return;
  unsigned long long bits = 0;
  if (x < W && y < H) {
    const Texel<double> t = tex[static_cast<size_t>(y) * W + x];
    bits = (static_cast<unsigned long long>(__double_as_longlong(t.I)) & ~1ull) | (t.mask != 0.0 ? 1ull : 0ull);
  }
  out[(static_cast<size_t>(y >> 1) * tiles_x + (x >> 2)) * 8 + ((y & 1) << 2) + (x & 3)] = bits;
}

__global__ void buildIntensityPlaneKernel(const Texel<float> *__restrict__ tex, int W, int H, int tiles_x, int tiles_y, unsigned *__restrict__ out) {
  const int x = blockIdx.x * kTileX + threadIdx.x;
  const int y = blockIdx.y * kTileY + threadIdx.y;
  if (x >= 4 * tiles_x || y >= 4 * tiles_y) return;
  unsigned bits = 0;
  if (x < W && y < H) {
    const Texel<float> t = tex[static_cast<size_t>(y) * W + x];
    bits = (__float_as_uint(t.I) & ~1u) | (t.mask != 0.0f ? 1u : 0u);
  }
  out[(static_cast<size_t>(y >> 2) * tiles_x + (x >> 2)) * 16 + ((y & 3) << 2) + (x & 3)] = bits;
}

template <typename S>
void buildTyped(dsopp_hip_pyramid *p, const uint8_t *img_dev, const uint8_t *vig_dev, const double *lut_dev, double vmax) {
  hipStream_t st = p->sr.stream;
  dim3 block(kTileX, kTileY);
  static const bool no_fused = std::getenv("DSOPP_HIP_PYRAMID_CHAIN") != nullptr;  // tuning aid: always the level-by-level chain
  if (!vig_dev && !no_fused) {
    // no vignette: every level straight from the 8-bit image in one launch (with a vignette every level-0 value costs a division,
    // which the nested means would repeat 4^l times: the chain below stays)
    AllLevelsArgs a;
    std::memset(&a, 0, sizeof(a));
    a.img = img_dev;
    a.lut = lut_dev;
    a.levels = p->levels;
    for (int l = 0; l < p->levels; ++l) {
      a.width[l] = p->w(l);
      a.height[l] = p->h(l);
      a.tex[l] = p->texels[l];
      a.plane[l] = p->planes[l];
    }
    a.tiles0_x = (p->w(0) + kTileX - 1) / kTileX;
    const int tiles0 = a.tiles0_x * ((p->h(0) + kTileY - 1) / kTileY);
    if (p->levels > 1) {
      // regions are counted on level 1 (the finest upper level): every level-1 texel must lie in some region
      const int side1 = kUpperTile << (p->levels - 2);
      a.upper_tiles_x = (p->w(1) + side1 - 1) / side1;
      a.n_upper = a.upper_tiles_x * ((p->h(1) + side1 - 1) / side1);
    }
    pyramidAllLevelsKernel<S><<<static_cast<unsigned>(a.n_upper + tiles0), block, 0, st>>>(a);
    HIP_CHECK(hipGetLastError());
    return;
  }
  {
    dim3 grid((p->w(0) + kTileX - 1) / kTileX, (p->h(0) + kTileY - 1) / kTileY);
    pyramidLevel0Kernel<S><<<grid, block, 0, st>>>(img_dev, vig_dev, lut_dev, vmax, p->w(0), p->h(0), static_cast<S *>(p->planes[0]),
                                                   static_cast<Texel<S> *>(p->texels[0]));
  }
  for (int l = 1; l < p->levels; ++l) {
    dim3 grid((p->w(l) + kTileX - 1) / kTileX, (p->h(l) + kTileY - 1) / kTileY);
    pyramidLevelKernel<S><<<grid, block, 0, st>>>(static_cast<const S *>(p->planes[l - 1]), p->w(l - 1), p->w(l), p->h(l),
                                                  static_cast<S *>(p->planes[l]), static_cast<Texel<S> *>(p->texels[l]));
  }
  HIP_CHECK(hipGetLastError());
}

template <typename S>
void fillMask(dsopp_hip_pyramid *p, int level, const uint8_t *mask_dev) {
  const size_t n = static_cast<size_t>(p->w(level)) * p->h(level);
  fillMaskKernel<S><<<static_cast<unsigned>((n + 255) / 256), 256, 0, p->sr.stream>>>(static_cast<Texel<S> *>(p->texels[level]), mask_dev, n);
  HIP_CHECK(hipGetLastError());
}

void checkLevel(dsopp_hip_pyramid *p, int level) {
  if (!p) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null pyramid");
  if (level < 0 || level >= p->levels) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "level %d out of range [0,%d)", level, p->levels);
}

}  // namespace
}  // namespace dsopp_hip

using namespace dsopp_hip;

const void *dsopp_hip_pyramid::intensityPlane(int level, hipStream_t consumer) const {
  if (level < 0 || level >= levels) return nullptr;
  std::lock_guard<std::mutex> lock(iplane_mutex);
  const int tx = itilesX(level), ty = itilesY(level);
  if (!iplane_valid[level]) {
    if (!iplane[level]) HIP_CHECK(hipMalloc(&iplane[level], static_cast<size_t>(tx) * ty * 64));  // one 64-byte tile per (tx, ty)
    if (!iplane_ready[level]) HIP_CHECK(hipEventCreateWithFlags(&iplane_ready[level], hipEventDisableTiming));
    waitReady(consumer);  // the texels' last build
    dim3 block(kTileX, kTileY), grid((4 * tx + kTileX - 1) / kTileX, ((dtype == DSOPP_HIP_F64 ? 2 : 4) * ty + kTileY - 1) / kTileY);
    if (dtype == DSOPP_HIP_F64)
      buildIntensityPlaneKernel<<<grid, block, 0, consumer>>>(static_cast<const Texel<double> *>(texels[level]), w(level), h(level), tx, ty,
                                                              static_cast<unsigned long long *>(iplane[level]));
    else
      buildIntensityPlaneKernel<<<grid, block, 0, consumer>>>(static_cast<const Texel<float> *>(texels[level]), w(level), h(level), tx, ty,
                                                              static_cast<unsigned *>(iplane[level]));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(iplane_ready[level], consumer));
    iplane_stream[level] = consumer;
    iplane_valid[level] = true;
  } else if (consumer != iplane_stream[level]) {
    HIP_CHECK(hipStreamWaitEvent(consumer, iplane_ready[level], 0));
  }
  return iplane[level];
}

extern "C" {

const char *dsopp_hip_last_error(void) { return lastError().c_str(); }
const char *dsopp_hip_version(void) { return "dsopp-hip 0.1 (gfx950)"; }

int dsopp_hip_device_count(int *count) {
  return guarded([&] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
  });
}

int dsopp_hip_pyramid_create(int device, void *stream, int width, int height, int levels, int dtype, dsopp_hip_pyramid **out) {
  return guarded([&] {
    if (!out || width <= 0 || height <= 0 || levels <= 0) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad pyramid dimensions");
    if (dtype != DSOPP_HIP_F64 && dtype != DSOPP_HIP_F32) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad dtype %d", dtype);
    levels = levels > DSOPP_HIP_MAX_LEVELS ? DSOPP_HIP_MAX_LEVELS : levels;  // pixel_data_frame.cpp:14
    auto *p = new dsopp_hip_pyramid();
    try {
      p->sr.init(device, stream);
      p->width = width;
      p->height = height;
      p->levels = levels;
      p->dtype = dtype;
      for (int l = 0; l < levels; ++l) {
        const size_t n = static_cast<size_t>(p->w(l)) * p->h(l);
        if (n == 0) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "level %d is empty", l);
        HIP_CHECK(hipMalloc(&p->texels[l], n * 4 * p->elemSize()));
        HIP_CHECK(hipMalloc(&p->planes[l], n * p->elemSize()));
        if (dtype == DSOPP_HIP_F64)
          fillMask<double>(p, l, nullptr);
        else
          fillMask<float>(p, l, nullptr);
      }
      HIP_CHECK(hipMalloc(&p->staging_u8, static_cast<size_t>(width) * height));
      HIP_CHECK(hipMalloc(&p->staging_vig, static_cast<size_t>(width) * height));
      HIP_CHECK(hipMalloc(&p->lut_dev, 256 * sizeof(double)));
      p->sr.sync();
    } catch (...) {
      dsopp_hip_pyramid_destroy(p);
      throw;
    }
    *out = p;
  });
}

void dsopp_hip_pyramid_destroy(dsopp_hip_pyramid *p) {
  if (!p) return;
  (void)hipSetDevice(p->sr.device);
  if (p->sr.stream) (void)hipStreamSynchronize(p->sr.stream);
  for (int l = 0; l < DSOPP_HIP_MAX_LEVELS; ++l) {
    if (p->texels[l]) (void)hipFree(p->texels[l]);
    if (p->planes[l]) (void)hipFree(p->planes[l]);
    if (p->iplane[l]) (void)hipFree(p->iplane[l]);
    if (p->iplane_ready[l]) (void)hipEventDestroy(p->iplane_ready[l]);
  }
  if (p->staging_u8) (void)hipFree(p->staging_u8);
  if (p->staging_vig) (void)hipFree(p->staging_vig);
  if (p->h_image) (void)hipHostFree(p->h_image);
  if (p->lut_dev) (void)hipFree(p->lut_dev);
  if (p->ready) (void)hipEventDestroy(p->ready);
  p->sr.destroy();
  delete p;
}

int dsopp_hip_pyramid_build_device(dsopp_hip_pyramid *p, const void *image_dev, const double *lut256, const void *vignetting_dev,
                                   double vignetting_max) {
  return guarded([&] {
    if (!p || !image_dev) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    p->sr.use();
    const double *lut_dev = nullptr;
    if (lut256) {
      HIP_CHECK(hipMemcpyAsync(p->lut_dev, lut256, 256 * sizeof(double), hipMemcpyHostToDevice, p->sr.stream));
      lut_dev = p->lut_dev;
    }
    if (p->dtype == DSOPP_HIP_F64)
      buildTyped<double>(p, static_cast<const uint8_t *>(image_dev), static_cast<const uint8_t *>(vignetting_dev), lut_dev, vignetting_max);
    else
      buildTyped<float>(p, static_cast<const uint8_t *>(image_dev), static_cast<const uint8_t *>(vignetting_dev), lut_dev, vignetting_max);
    p->markReady();
  });
}

namespace {
/** the caller's image into this pyramid's pinned buffer with non-temporal stores: the destination is read next by the DMA engine, not by
 *  the host, so the copy neither reads the old lines of the buffer first (write-allocate) nor leaves 1.3 MB of them in the caches —
 *  memcpy() takes 67 us for a 1280 x 1024 image on the tracker's per-frame path (DSOPP_HIP_IMAGE_COPY=memcpy: A/B) */
void copyToPinned(uint8_t *dst, const uint8_t *src, size_t n) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  static const bool plain = std::getenv("DSOPP_HIP_IMAGE_COPY") != nullptr && std::string(std::getenv("DSOPP_HIP_IMAGE_COPY")) == "memcpy";
  if (!plain && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
      const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i)), b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 16));
      const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 32)), d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 48));
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), a);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 16), b);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 32), c);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 48), d);
    }
    if (i < n) std::memcpy(dst + i, src + i, n - i);
    _mm_sfence();  // the streamed lines are in memory before the copy engine is told to read them
    return;
  }
#endif
  std::memcpy(dst, src, n);
}
}  // namespace

int dsopp_hip_pyramid_build(dsopp_hip_pyramid *p, const uint8_t *image_host, const double *lut256, const uint8_t *vignetting_host) {
  return guarded([&] {
    if (!p || !image_host) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    p->sr.use();
    const size_t n = static_cast<size_t>(p->width) * p->height;
    // The caller's image is pageable memory: copied from there the upload is a staged, blocking transfer (100 us for 1280 x 1024, the
    // largest part of the tracker's per-frame pyramid time, rocprofv3 --hip-trace).  It is copied into this pyramid's own pinned buffer
    // instead (a memcpy of 1.3 MB) and leaves from there as a DMA the call does not wait for: consumers order themselves behind the build
    // with waitReady(), as they do behind build_device.  (The buffer's previous upload has long completed: a pyramid is rebuilt once per
    // frame, and the synchronisation below covers the paths that keep it.)
    if (!p->h_image) {
      HIP_CHECK(hipStreamSynchronize(p->sr.stream));
      HIP_CHECK(hipHostMalloc(&p->h_image, n, hipHostMallocDefault));
    } else {
      HIP_CHECK(hipStreamSynchronize(p->sr.stream));  // (free when the stream is idle — the usual case; guards a rebuild while the last upload is in flight)
    }
    // In pieces: the DMA of a piece (1.3 MB over the host link: ~50 us, as long as the memcpy itself) runs while the host copies the next
    // one — the consumer's wait for the pyramid shrinks by three quarters of the transfer (DSOPP_HIP_IMAGE_PIECES=1: one piece, A/B)
    static const int pieces_env = std::getenv("DSOPP_HIP_IMAGE_PIECES") ? std::atoi(std::getenv("DSOPP_HIP_IMAGE_PIECES")) : 4;
    const size_t pieces = n >= (size_t(1) << 19) ? static_cast<size_t>(std::max(1, std::min(pieces_env, 16))) : 1;
    const size_t piece = ((n + pieces - 1) / pieces + 4095) & ~static_cast<size_t>(4095);
    for (size_t off = 0; off < n; off += piece) {
      const size_t len = std::min(piece, n - off);
      copyToPinned(static_cast<uint8_t *>(p->h_image) + off, image_host + off, len);
      HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t *>(p->staging_u8) + off, static_cast<uint8_t *>(p->h_image) + off, len, hipMemcpyHostToDevice, p->sr.stream));
    }
    double vmax = 0;
    if (vignetting_host) {
      // cv::minMaxLoc(vignetting, nullptr, &max) — photometrically_corrected_image.cpp:11-13
      for (size_t i = 0; i < n; ++i) vmax = vignetting_host[i] > vmax ? vignetting_host[i] : vmax;
      HIP_CHECK(hipMemcpyAsync(p->staging_vig, vignetting_host, n, hipMemcpyHostToDevice, p->sr.stream));
    }
    int rc = dsopp_hip_pyramid_build_device(p, p->staging_u8, lut256, vignetting_host ? p->staging_vig : nullptr, vmax);
    if (rc != DSOPP_HIP_OK) throw Error(rc, lastError());
    // the LUT and the vignette are read straight from the caller's (pageable) arrays: those paths wait; the plain image path does not
    if (lut256 || vignetting_host) p->sr.sync();
  });
}

int dsopp_hip_pyramid_set_level(dsopp_hip_pyramid *p, int level, const double *pixelinfo_host) {
  return guarded([&] {
    checkLevel(p, level);
    if (!pixelinfo_host) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null pixelinfo");
    p->sr.use();
    const size_t n = static_cast<size_t>(p->w(level)) * p->h(level);
    double *tmp = nullptr;
    HIP_CHECK(hipMalloc(&tmp, n * 3 * sizeof(double)));
    HIP_CHECK(hipMemcpyAsync(tmp, pixelinfo_host, n * 3 * sizeof(double), hipMemcpyHostToDevice, p->sr.stream));
    const unsigned grid = static_cast<unsigned>((n + 255) / 256);
    if (p->dtype == DSOPP_HIP_F64)
      setLevelKernel<double><<<grid, 256, 0, p->sr.stream>>>(static_cast<Texel<double> *>(p->texels[level]), static_cast<double *>(p->planes[level]), tmp, n);
    else
      setLevelKernel<float><<<grid, 256, 0, p->sr.stream>>>(static_cast<Texel<float> *>(p->texels[level]), static_cast<float *>(p->planes[level]), tmp, n);
    HIP_CHECK(hipGetLastError());
    p->markReady();
    p->sr.sync();
    (void)hipFree(tmp);
  });
}

int dsopp_hip_pyramid_set_mask(dsopp_hip_pyramid *p, int level, const uint8_t *mask_host) {
  return guarded([&] {
    checkLevel(p, level);
    p->sr.use();
    const size_t n = static_cast<size_t>(p->w(level)) * p->h(level);
    const uint8_t *mask_dev = nullptr;
    if (mask_host) {
      HIP_CHECK(hipMemcpyAsync(p->staging_u8, mask_host, n, hipMemcpyHostToDevice, p->sr.stream));
      mask_dev = static_cast<const uint8_t *>(p->staging_u8);
    }
    if (p->dtype == DSOPP_HIP_F64)
      fillMask<double>(p, level, mask_dev);
    else
      fillMask<float>(p, level, mask_dev);
    p->markReady();
    p->sr.sync();
  });
}

int dsopp_hip_pyramid_get_level(dsopp_hip_pyramid *p, int level, double *pixelinfo_host) {
  return guarded([&] {
    checkLevel(p, level);
    if (!pixelinfo_host) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null output");
    p->sr.use();
    const size_t n = static_cast<size_t>(p->w(level)) * p->h(level);
    double *tmp = nullptr;
    HIP_CHECK(hipMalloc(&tmp, n * 3 * sizeof(double)));
    const unsigned grid = static_cast<unsigned>((n + 255) / 256);
    if (p->dtype == DSOPP_HIP_F64)
      getLevelKernel<double><<<grid, 256, 0, p->sr.stream>>>(static_cast<const Texel<double> *>(p->texels[level]), tmp, n);
    else
      getLevelKernel<float><<<grid, 256, 0, p->sr.stream>>>(static_cast<const Texel<float> *>(p->texels[level]), tmp, n);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(pixelinfo_host, tmp, n * 3 * sizeof(double), hipMemcpyDeviceToHost, p->sr.stream));
    p->sr.sync();
    (void)hipFree(tmp);
  });
}

int dsopp_hip_pyramid_level_size(dsopp_hip_pyramid *p, int level, int *width, int *height) {
  return guarded([&] {
    checkLevel(p, level);
    if (width) *width = p->w(level);
    if (height) *height = p->h(level);
  });
}

}  // extern "C"
