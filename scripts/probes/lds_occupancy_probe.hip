// probe: how many 512-thread workgroups share a compute unit as a function of their static LDS size (gfx950: 160 KB per CU) — what the
// occupancy API says, and what a launch of 512 workgroups that each spin for 20 us takes (one round = all resident at once).
//   hipcc --offload-arch=gfx950 -O3 lds_occupancy_probe.hip -o bin/lds_occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ void __launch_bounds__(512, 4) k(double *out, int ticks) {
  __shared__ double buf[BYTES / 8];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  out[threadIdx.x] = buf[(threadIdx.x * 7) % (BYTES / 8)];
}
template <int BYTES>
void report(double *out) {
  int n = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES>, 512, 0);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best[3] = {1e9f, 1e9f, 1e9f};
  const int grids[3] = {256, 512, 768};
  for (int g = 0; g < 3; ++g)
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(a);
      k<BYTES><<<grids[g], 512>>>(out, 2000);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (ms < best[g]) best[g] = ms;
    }
  printf("static LDS %6d B: API says %d workgroups per CU (%s); 20 us spin: grid 256 -> %.1f us, 512 -> %.1f us, 768 -> %.1f us\n", BYTES, n, hipGetErrorString(e),
         best[0] * 1e3, best[1] * 1e3, best[2] * 1e3);
}
int main() {
  double *out; (void)hipMalloc(&out, 4096);
  report<40960>(out); report<57344>(out); report<65536>(out); report<69632>(out); report<73728>(out); report<77824>(out); report<78784>(out); report<80896>(out); report<81920>(out); report<83968>(out);
  return 0;
}
