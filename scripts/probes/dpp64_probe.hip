// What v_mov_b64_dpp / v_fmac_f64_dpp with row_newbcast:k do on gfx950: lane k of every row of 16 lanes is the source for all 16 lanes of
// that row (the only DPP control the 64-bit VALU operations accept).  Prints the first lane of every row and checks all 64.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__device__ __forceinline__ double bcast(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
template <int K>
__device__ __forceinline__ void fmacBcast(double &d, double src0_bcast, double src1) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(src0_bcast), "v"(src1), "n"(K));
}

__global__ void probe(double *out) {
  const int lane = threadIdx.x;
  const double v = 100.0 + lane;
  out[lane] = bcast<3>(v);                 // expect 100 + 16 * (lane / 16) + 3
  double d = 1000.0 * lane;
  fmacBcast<5>(d, v, 2.0);                 // expect 1000 lane + (100 + 16 (lane / 16) + 5) * 2
  out[64 + lane] = d;
  double e = 0.5;
  fmacBcast<15>(e, v, -1.0 * lane);        // src1 per lane: 0.5 - (115 + 16 (lane / 16)) * lane
  out[128 + lane] = e;
}

int main() {
  double *d, h[192];
  hipMalloc(&d, sizeof(h));
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const double row = 16.0 * (l / 16);
    bad += h[l] != 100 + row + 3;
    bad += h[64 + l] != 1000.0 * l + (100 + row + 5) * 2;
    bad += h[128 + l] != 0.5 - (115 + row) * l;
  }
  printf("v_mov_b64_dpp row_newbcast:3 -> lanes 0,16,32,48: %g %g %g %g\n", h[0], h[16], h[32], h[48]);
  printf("v_fmac_f64_dpp row_newbcast:5 -> lanes 1,17,33,49: %g %g %g %g\n", h[65], h[81], h[97], h[113]);
  printf("%s (%d mismatches)\n", bad ? "UNEXPECTED" : "as expected: lane k of each row of 16 is broadcast to that row", bad);
  return bad != 0;
}
