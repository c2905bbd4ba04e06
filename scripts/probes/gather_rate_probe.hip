// probe: what the texture path (TA / TCP) charges for the sweep's bilinear footprint — 128 bytes per pixel as two runs of 64 bytes —
// fetched (a) by the pixel's own lane as 8 scattered 16-byte loads, (b) transposed: 4 adjacent lanes take the 4 pieces of a run
// (b128 loads into registers), (c) as (b) but through global_load_lds_dwordx4, (d) own lane, 4 x 8 bytes (the intensity plane's pattern).
// Prints ns per pixel and lane-accesses per clock and compute unit.   hipcc --offload-arch=gfx950 -O3 gather_rate_probe.hip -o bin/gather_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double Vec2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *LdsPtr;
typedef const __attribute__((address_space(1))) void *GlbPtr;
constexpr int W = 640, H = 480, kImages = 12, kIters = 32;
__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// position of "pixel" p: items of 8 pixels within a 5 x 5 neighbourhood, items scattered over the image (clump = 1: all items in one 48 x 48 window)
__device__ __forceinline__ size_t footprint(unsigned pixel_id, int clump) {
  const unsigned item = pixel_id >> 3, k = pixel_id & 7;
  const unsigned h = hash(item);
  int x = clump ? 200 + (h % 48) : 8 + (h % (W - 16)), y = clump ? 200 + ((h >> 12) % 48) : 8 + ((h >> 12) % (H - 16));
  const int ox = static_cast<int>((0x21420312u >> (4 * k)) & 0xFu) - 2, oy = static_cast<int>((0x01222334u >> (4 * k)) & 0xFu) - 2;
  const unsigned img = (h >> 24) % kImages;
  return (static_cast<size_t>(img) * W * H + static_cast<size_t>(y + oy) * W + (x + ox)) * 32;
}
template <int MODE>
__global__ void __launch_bounds__(128) gather(const char *img, double *out, int clump) {
  __shared__ __attribute__((aligned(16))) char stage[2][8 * 1040];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0;
  for (int it = 0; it < kIters; ++it) {
    const unsigned pix = ((blockIdx.x * kIters + it) * 128u + threadIdx.x);
    const size_t e = footprint(pix, clump);
    if (MODE == 0) {
      const Vec2 *p = reinterpret_cast<const Vec2 *>(img + e), *q = reinterpret_cast<const Vec2 *>(img + e + W * 32);
      const Vec2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], b0 = q[0], b1 = q[1], b2 = q[2], b3 = q[3];
      acc += a0.x + a1.y + a2.x + a3.y + b0.x + b1.y + b2.x + b3.y;
    } else if (MODE == 3) {
      const double *p = reinterpret_cast<const double *>(img + e), *q = reinterpret_cast<const double *>(img + e + W * 32);
      acc += p[0] + p[4] + q[0] + q[4];
    } else {
      // transposed: instruction n fetches pixel n of every item; lane sub of the item takes row sub >> 2, piece sub & 3
      const int sub = lane & 7;
      const size_t piece = static_cast<size_t>(sub >> 2) * W * 32 + (sub & 3) * 16;
      const unsigned elo = static_cast<unsigned>(e), ehi = static_cast<unsigned>(e >> 32);
#define STEP(n)                                                                                                    \
      {                                                                                                            \
        const size_t en = (static_cast<size_t>(__builtin_amdgcn_ds_swizzle(ehi, 0x18 | ((n) << 5))) << 32) |      \
                          static_cast<unsigned>(__builtin_amdgcn_ds_swizzle(elo, 0x18 | ((n) << 5)));             \
        if (MODE == 1) {                                                                                           \
          const Vec2 v = *reinterpret_cast<const Vec2 *>(img + en + piece);                                        \
          acc += v.x + v.y;                                                                                        \
        } else {                                                                                                   \
          __builtin_amdgcn_global_load_lds((GlbPtr)(img + en + piece), (LdsPtr)(stage[wave] + (n) * 1040), 16, 0, 0); \
        }                                                                                                          \
      }
      STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
#undef STEP
      if (MODE == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const Vec2 *own = reinterpret_cast<const Vec2 *>(stage[wave] + (lane & 7) * 1040 + (lane >> 3) * 128);
        acc += own[0].x + own[1].y + own[2].x + own[3].y + own[4].x + own[5].y + own[6].x + own[7].y;
      }
    }
  }
  out[blockIdx.x * 128 + threadIdx.x] = acc;
}
int main() {
  const size_t bytes = static_cast<size_t>(kImages) * W * H * 32 + (W + 8) * 32 * 4;
  char *img;
  double *out;
  hipMalloc(&img, bytes);
  hipMemset(img, 0, bytes);
  const int blocks = 4096;
  hipMalloc(&out, blocks * 128 * 8);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char *names[4] = {"own lane, 8 x 16 B", "transposed, b128 to registers", "transposed, LDS-DMA + LDS reads", "own lane, 4 x 8 B"};
  for (int clump = 0; clump < 2; ++clump)
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        if (mode == 0) gather<0><<<blocks, 128>>>(img, out, clump);
        if (mode == 1) gather<1><<<blocks, 128>>>(img, out, clump);
        if (mode == 2) gather<2><<<blocks, 128>>>(img, out, clump);
        if (mode == 3) gather<3><<<blocks, 128>>>(img, out, clump);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      const double pixels = static_cast<double>(blocks) * 128 * kIters, acc_per_px = mode == 3 ? 4 : 8;
      printf("%s %-34s %8.1f us  %.3f ns/pixel  %.2f lane-accesses / clk / CU (2.2 GHz, 256 CUs)  %.2f TB/s of footprint bytes\n", clump ? "clump " : "random", names[mode], best * 1e3,
             best * 1e6 / pixels, pixels * acc_per_px / (best * 1e-3 * 2.2e9 * 256), pixels * (mode == 3 ? 32 : 128) / (best * 1e-3) / 1e12);
    }
  return 0;
}
