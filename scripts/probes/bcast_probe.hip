// Probe: cost of broadcasting a lane's f64 to the wave inside a dependent chain (the pivot loops of the dense solve), single wave.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double readLane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#define T0 long long c0, c1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "+v"(x)::"memory");
#define T1(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "+v"(x)::"memory"); if (threadIdx.x == 0) out[i] = c1 - c0;
__global__ void probe(double *io, long long *out) {
  __shared__ double buf[64];
  double x = io[threadIdx.x], y = io[64 + threadIdx.x], z = y * 0.5;
  for (int rep = 0; rep < 2; ++rep) {
    { T0
#pragma unroll
      for (int i = 0; i < 64; ++i) x = fma(x, 1.0000001, y);
      T1(0) }
    { T0  // dependent: fma -> readlane -> fma
#pragma unroll
      for (int i = 0; i < 64; ++i) { const double s = readLane(x, i); x = fma(y, s, x); }
      T1(1) }
    { T0  // readlane of a value that does NOT depend on the chain (issue cost), fma chain dependent
#pragma unroll
      for (int i = 0; i < 64; ++i) { const double s = readLane(y, i); x = fma(z, s, x); }
      T1(2) }
    { T0  // dependent: mul -> readlane -> fma (the old back-substitution step)
#pragma unroll
      for (int i = 0; i < 64; ++i) { const double s = readLane(x * z, i); x = fma(y, -s, x); }
      T1(3) }
    { T0  // dependent broadcast through ds_bpermute
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const int lo = __builtin_amdgcn_ds_bpermute(4 * i, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(4 * i, __double2hiint(x));
        x = fma(y, __hiloint2double(hi, lo), x);
      }
      T1(4) }
    { T0  // dependent broadcast through LDS: one lane writes, all read the same address
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        if (threadIdx.x == i) buf[0] = x;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const double s = buf[0];
        x = fma(y, s, x);
      }
      T1(5) }
    { T0  // 8 independent chains of dependent (readlane -> fma): does the wave overlap them?
      double a0 = x, a1 = y, a2 = z, a3 = x + y, a4 = x - y, a5 = y + z, a6 = z - x, a7 = x * 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a0 = fma(y, readLane(a0, i), a0); a1 = fma(y, readLane(a1, i), a1); a2 = fma(y, readLane(a2, i), a2); a3 = fma(y, readLane(a3, i), a3);
        a4 = fma(y, readLane(a4, i), a4); a5 = fma(y, readLane(a5, i), a5); a6 = fma(y, readLane(a6, i), a6); a7 = fma(y, readLane(a7, i), a7);
      }
      x += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
      T1(6) }
    { T0  // v_rcp_f64 + 2 Newton steps, dependent, x 16
#pragma unroll
      for (int i = 0; i < 16; ++i) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); r = fma(fma(-x, r, 1.0), r, r); x = r + 1.5; }
      T1(7) }
    { T0  // 64 x (v_cmp + 2 v_cndmask) dependent
#pragma unroll
      for (int i = 0; i < 64; ++i) x = x > y ? x * 1.0000001 : z;
      T1(8) }
  }
  io[threadIdx.x] = x + y;
}
int main() {
  double *d; long long *s;
  hipMalloc(&d, 4096); hipMalloc(&s, 256);
  double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + i * 1e-3;
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, s);
  hipDeviceSynchronize();
  long long o[16]; hipMemcpy(o, s, sizeof(o), hipMemcpyDeviceToHost);
  const char *names[] = {"64 dep fma", "64 dep (readlane pair -> fma)", "64 (indep readlane pair, dep fma)", "64 dep (mul -> readlane -> fma)", "64 dep (bpermute pair -> fma)",
                         "64 dep (LDS write -> bcast read -> fma)", "8 chains x 8 dep (readlane -> fma)", "16 dep rcp + 2 Newton + add", "64 dep cmp/select/mul"};
  for (int i = 0; i < 9; ++i) printf("%-42s %6lld cycles\n", names[i], o[i]);
  return 0;
}
