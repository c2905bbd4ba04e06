// Probe: single-wave f64 latencies / issue rates on gfx950 (shader clock cycles via s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#define TIME_BLOCK(idx, BODY)                                                              \
  {                                                                                        \
    long long t0, t1;                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(x), "+v"(y)::"memory"); \
    BODY;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(x), "+v"(y)::"memory"); \
    if (threadIdx.x == 0) out[idx] = t1 - t0;                                              \
  }
__global__ void probe(double *io, long long *out) {
  double x = io[threadIdx.x], y = io[64 + threadIdx.x];
  double a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;
  for (int rep = 0; rep < 2; ++rep) {
    // 0: 64 dependent FMAs
    TIME_BLOCK(0, {
_Pragma("unroll")
      for (int i = 0; i < 64; ++i) x = fma(x, 1.0000001, y);
    });
    // 1: 64 independent FMAs (8 chains x 8)
    TIME_BLOCK(1, {
_Pragma("unroll")
      for (int i = 0; i < 8; ++i) {
        a0 = fma(a0, 1.0000001, y); a1 = fma(a1, 1.0000001, y); a2 = fma(a2, 1.0000001, y); a3 = fma(a3, 1.0000001, y);
        a4 = fma(a4, 1.0000001, y); a5 = fma(a5, 1.0000001, y); a6 = fma(a6, 1.0000001, y); a7 = fma(a7, 1.0000001, y);
      }
      x += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    });
    // 2: 16 dependent v_rcp_f64
    TIME_BLOCK(2, {
_Pragma("unroll")
      for (int i = 0; i < 16; ++i) x = __builtin_amdgcn_rcp(x) + 0.0;
    });
    // 3: 16 dependent (cvt f32, rsq f32, cvt f64)
    TIME_BLOCK(3, {
_Pragma("unroll")
      for (int i = 0; i < 16; ++i) x = (double)__frsqrt_rn((float)x);
    });
    // 4: 16 dependent f64 sqrt (library)
    TIME_BLOCK(4, {
_Pragma("unroll")
      for (int i = 0; i < 16; ++i) x = sqrt(x + 2.0);
    });
    // 5: 16 dependent f64 divisions
    TIME_BLOCK(5, {
_Pragma("unroll")
      for (int i = 0; i < 16; ++i) x = y / (x + 2.0);
    });
    // 6: 64 dependent f64 mul
    TIME_BLOCK(6, {
_Pragma("unroll")
      for (int i = 0; i < 64; ++i) x = x * 1.0000001;
    });
    // 7: 64 dependent f32 FMAs
    {
      float xf = (float)x, yf = (float)y;
      long long t0, t1;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(xf)::"memory");
_Pragma("unroll")
      for (int i = 0; i < 64; ++i) xf = fmaf(xf, 1.0000001f, yf);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(xf)::"memory");
      if (threadIdx.x == 0) out[7] = t1 - t0;
      x += xf;
    }
    // 8: LDS write -> read round trip x 16 (dependent)
    {
      __shared__ double buf[128];
      long long t0, t1;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(x)::"memory");
_Pragma("unroll")
      for (int i = 0; i < 16; ++i) {
        buf[threadIdx.x] = x;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        x = buf[(threadIdx.x + 1) & 63] + 1.0;
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(x)::"memory");
      if (threadIdx.x == 0) out[8] = t1 - t0;
    }
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
  const char *names[] = {"64 dep f64 fma", "64 indep f64 fma (+8 adds)", "16 dep rcp_f64(+add)", "16 dep cvt+rsq_f32+cvt", "16 dep sqrt f64", "16 dep div f64", "64 dep f64 mul", "64 dep f32 fma", "16 dep LDS write->read"};
  for (int i = 0; i < 9; ++i) printf("%-30s %6lld cycles\n", names[i], o[i]);
  return 0;
}
