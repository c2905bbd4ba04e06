// probe: (1) global_load_lds_dwordx4 — every lane moves 16 bytes from its own global address into LDS at base + lane * 16 (gfx950);
//        (2) ds_swizzle broadcast of lane n of every group of 8 lanes.   hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o bin/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *LdsPtr;
typedef const __attribute__((address_space(1))) void *GlbPtr;
__global__ void probe(const double *src, double *out, int *sw) {
  __shared__ __attribute__((aligned(16))) double stage[2 * 130];
  const int lane = threadIdx.x;
  // lane reads src[2 * perm(lane)], perm = reversed lanes: a scattered per-lane address
  const double *g = src + 2 * (63 - lane);
  __builtin_amdgcn_global_load_lds((GlbPtr)g, (LdsPtr)stage, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((GlbPtr)(g + 128), (LdsPtr)(stage + 130), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[2 * lane] = stage[2 * lane];
  out[2 * lane + 1] = stage[2 * lane + 1];
  out[128 + 2 * lane] = stage[130 + 2 * lane];
  out[128 + 2 * lane + 1] = stage[130 + 2 * lane + 1];
#define SWZ(n) sw[n * 64 + lane] = __builtin_amdgcn_ds_swizzle(lane * 10, 0x18 | (n << 5))
  SWZ(0); SWZ(1); SWZ(2); SWZ(3); SWZ(4); SWZ(5); SWZ(6); SWZ(7);
}
int main() {
  std::vector<double> h(256);
  for (int i = 0; i < 256; ++i) h[i] = i;
  double *d, *o;
  int *s;
  hipMalloc(&d, 256 * 8);
  hipMalloc(&o, 256 * 8);
  hipMalloc(&s, 512 * 4);
  hipMemcpy(d, h.data(), 256 * 8, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o, s);
  std::vector<double> r(256);
  std::vector<int> w(512);
  hipMemcpy(r.data(), o, 256 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(w.data(), s, 512 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    if (r[2 * l] != 2 * (63 - l) || r[2 * l + 1] != 2 * (63 - l) + 1) ++bad;
    if (r[128 + 2 * l] != 128 + 2 * (63 - l) || r[128 + 2 * l + 1] != 128 + 2 * (63 - l) + 1) ++bad;
  }
  int badsw = 0;
  for (int n = 0; n < 8; ++n)
    for (int l = 0; l < 64; ++l)
      if (w[n * 64 + l] != ((l & ~7) | n) * 10) ++badsw;
  printf("lds dma mismatches %d (lane 0 got %g %g, lane 5 got %g %g); swizzle mismatches %d (n=3: lane 0 -> %d, lane 13 -> %d, lane 40 -> %d)\n", bad, r[0], r[1], r[10], r[11], badsw,
         w[3 * 64], w[3 * 64 + 13], w[3 * 64 + 40]);
  return bad || badsw;
}
