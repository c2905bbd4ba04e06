// Probe: is the instruction cache cold at kernel launch? Times two passes over a ~24 KB straight-line block in one launch.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__device__ __forceinline__ float chain(float x) {
  if constexpr (N == 0) return x;
  else {
    x = __builtin_fmaf(x, 1.0f + N * 1e-6f, 0.5f + N * 1e-3f);  // distinct 32-bit literals => 12 bytes per instruction
    return chain<N - 1>(x);
  }
}
__global__ void probe(float *out, long long *stamps, int passes) {
  float x = out[0];
  for (int p = 0; p < passes; ++p) {
    long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(x) : : "memory");
    x = chain<512>(x);
    x = chain<512>(x);
    x = chain<512>(x);
    x = chain<512>(x);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(x) : : "memory");
    if (threadIdx.x == 0) stamps[p] = t1 - t0;
  }
  out[1] = x;
}
int main() {
  float *d; long long *s;
  hipMalloc(&d, 64); hipMalloc(&s, 64);
  hipMemset(d, 0, 64);
  for (int launch = 0; launch < 3; ++launch) {
    probe<<<1, 64>>>(d, s, 3);
    hipDeviceSynchronize();
    long long h[3]; hipMemcpy(h, s, sizeof(h), hipMemcpyDeviceToHost);
    printf("launch %d: pass ticks (10 ns): %lld %lld %lld\n", launch, h[0], h[1], h[2]);
  }
  return 0;
}
