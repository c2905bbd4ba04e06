// Probe: effective shader clock seen by ONE latency-bound wave (s_memtime = shader cycles, wall_clock64 = 100 MHz constant), idle
// chip vs a chip kept busy by a streaming kernel on another stream.  A single-workgroup kernel (the dense solve of the LM loop) runs
// at whatever clock the power management grants a nearly idle chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chain(double *io, long long *out, int reps) {
  double x = io[threadIdx.x], y = io[64 + threadIdx.x];
  long long c0, c1;
  const long long w0 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "+v"(x)::"memory");
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 64; ++i) x = fma(x, 1.0000001, y);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "+v"(x)::"memory");
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
  }
  io[threadIdx.x] = x;
}
__global__ void burn(float4 *a, const float4 *b, size_t n, int passes) {
  for (int p = 0; p < passes; ++p)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      float4 v = b[i];
      v.x += 1.f;
      a[i] = v;
    }
}
int main() {
  double *d; long long *s;
  hipMalloc(&d, 4096); hipMalloc(&s, 256);
  double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + i * 1e-3;
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  float4 *a, *b; size_t n = (size_t)1 << 26; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(b, 0, n * 16);
  auto run = [&](const char *what, int reps) {
    chain<<<1, 64, 0, s1>>>(d, s, reps);
    hipStreamSynchronize(s1);
    long long o[2]; hipMemcpy(o, s, sizeof(o), hipMemcpyDeviceToHost);
    printf("%-44s %8lld fma  %9lld shader cycles  %8.2f us  -> %.3f GHz, %.2f cycles / dependent f64 fma, %.2f ns each\n", what, 64LL * reps, o[0], o[1] / 100.0,
           o[0] / (o[1] * 10.0), (double)o[0] / (64.0 * reps), o[1] * 10.0 / (64.0 * reps));
  };
  run("idle chip, short (2 us)", 4);
  run("idle chip, 20 us", 40);
  run("idle chip, 200 us", 400);
  run("idle chip, 2 ms", 4000);
  for (int k = 0; k < 3; ++k) run("idle chip, 20 us again", 40);
  burn<<<2048, 256, 0, s2>>>(a, b, n, 40);
  for (int k = 0; k < 4; ++k) run("while a streaming kernel runs (other stream)", 40);
  hipStreamSynchronize(s2);
  for (int k = 0; k < 4; ++k) run("right after the streaming kernel", 40);
  return 0;
}
