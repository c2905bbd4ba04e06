// Probe (round 5): what does it cost when G workgroups of a landmark-major linearisation each add their own 16-landmark Schur
// tiles and their six per-pair blocks into the ONE combined system (1596 + 56 doubles at 7 frames) with f64 atomics — against the
// round-4 split, where 35 Schur workgroups of 64 landmarks and 49 pair workgroups do it?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_fanin_probe.hip -o bin/atomic_fanin_probe && timeout 60 bin/atomic_fanin_probe
// Per workgroup (768 threads = 12 waves): waves 0..9 own one 16 x 16 tile of the upper triangle of a 64 x 64 matrix (4 adds per lane,
// entries mapped into the block-packed lower triangle as pba_solve_kernels.hpp: combIndex), waves 0, 2, .. 10 then add three 8 x 8
// blocks + two 8-vectors of "their" frame pair.  Variants: atomics / plain stores into private slots (the floor: same instruction
// stream, no contention) / one extra returning atomic on a per-pair counter (the ticket form).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kF = 7, kK = 8 * kF;
__host__ __device__ constexpr int blockIndex(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }
__device__ __forceinline__ int combIndex(int row, int col) { return blockIndex(row >> 3, col >> 3) * 64 + ((row & 7) << 3) + (col & 7); }
constexpr int kComb = kF * (kF + 1) / 2 * 64 + kK;

__device__ __forceinline__ void addTo(double *p, double v, int mode) {
  if (mode == 1)
    *p = v;
  else if (mode == 3)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // executed in THIS XCD's L2: only valid on a per-XCD copy
  else
    atomicAdd(p, v);
}

// 0 agent-scope atomics into ONE system, 1 private plain stores, 2 as 0 + a returning ticket per pair,
// 3 workgroup-scope atomics into the copy of the workgroup's XCD (8 copies), 4 agent-scope atomics into the XCD's copy
template <int MODE>
__global__ void __launch_bounds__(768) fanin(double *comb, double *priv, unsigned *tickets, int spin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x % kF;
  unsigned xcc = 0;
  if (MODE >= 3) xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 7u;  // HW_REG_XCC_ID, bits 3:0
  double *dst = MODE == 1 ? priv + static_cast<size_t>(blockIdx.x) * 2048 : comb + (MODE >= 3 ? xcc * 2048 : 0);
  // a little arithmetic in front, so that the workgroups do not arrive in lock step
  double v = 1.0 + lane * 1e-3;
  for (int i = 0; i < spin + (blockIdx.x & 7) * 8; ++i) v = v * 1.0000001 + 1e-9;
  if (spin < 0) v = 1.0;
  if (wave < 10) {
    int ti = 0, rem = wave;
    while (rem >= 4 - ti) {
      rem -= 4 - ti;
      ++ti;
    }
    const int tj = ti + rem, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = 16 * ti + lk + 4 * reg, col = 16 * tj + li;
      if (row < kK && col < kK && col >= row) {
        addTo(&dst[combIndex(col, row)], v, MODE);
      }
      if (row < kK && col == kK) {
        addTo(&dst[kComb - kK + row], v, MODE);
      }
    }
  }
  if ((wave & 1) == 0) {
    int t = wave >> 1;
    if (t >= r) ++t;
    const int hi = r > t ? r : t, lo = r > t ? t : r;
    addTo(&dst[blockIndex(r, r) * 64 + lane], v, MODE);
    addTo(&dst[blockIndex(t, t) * 64 + lane], v, MODE);
    addTo(&dst[blockIndex(hi, lo) * 64 + lane], v, MODE);
    if (lane < 8) addTo(&dst[kComb - kK + 8 * r + lane], v, MODE), addTo(&dst[kComb - kK + 8 * t + lane], v, MODE);
    if (MODE == 2 && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = atomicAdd(&tickets[r * kF + t], 1u);
      if (old == 0xFFFFFFFFu) comb[0] = 0;  // (keeps the returned value live)
    }
  }
}

// the round-4 shape: 35 workgroups x 10 tiles of 64 landmarks, 49 pair workgroups
__global__ void __launch_bounds__(512) round4(double *comb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double v = 1.0 + lane * 1e-3;
  if (blockIdx.x < 35) {
    for (int tile = wave; tile < 10; tile += 8) {
      int ti = 0, rem = tile;
      while (rem >= 4 - ti) {
        rem -= 4 - ti;
        ++ti;
      }
      const int tj = ti + rem, li = lane & 15, lk = lane >> 4;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = 16 * ti + lk + 4 * reg, col = 16 * tj + li;
        if (row < kK && col < kK && col >= row) atomicAdd(&comb[combIndex(col, row)], v);
        if (row < kK && col == kK) atomicAdd(&comb[kComb - kK + row], v);
      }
    }
  } else if (wave == 0) {
    const int p = blockIdx.x - 35, r = p / kF, t = p % kF;
    if (r == t) return;
    const int hi = r > t ? r : t, lo = r > t ? t : r;
    atomicAdd(&comb[blockIndex(r, r) * 64 + lane], v);
    atomicAdd(&comb[blockIndex(t, t) * 64 + lane], v);
    atomicAdd(&comb[blockIndex(hi, lo) * 64 + lane], v);
    if (lane < 8) atomicAdd(&comb[kComb - kK + 8 * r + lane], v), atomicAdd(&comb[kComb - kK + 8 * t + lane], v);
  }
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  double *comb, *priv;
  unsigned *tickets;
  hipMalloc(&comb, 8 * 2048 * sizeof(double));
  hipMalloc(&priv, 1024 * 2048 * sizeof(double));
  hipMalloc(&tickets, 64 * sizeof(unsigned));
  hipMemset(comb, 0, 8 * 2048 * sizeof(double));
  hipMemset(tickets, 0, 64 * sizeof(unsigned));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int n = 200;
  auto timeIt = [&](auto &&launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a, 0);
      for (int i = 0; i < n; ++i) launch();
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
    }
    return best * 1e3f / n;
  };
  std::printf("round-4 shape (35 x 64-landmark Schur workgroups + 49 pair workgroups): %.2f us per launch\n",
              timeIt([&] { round4<<<35 + 49, 512>>>(comb); }));
  for (int spin : {0, 400}) {
    for (int g : {32, 63, 126, 252, 504}) {
      const float t_atomic = timeIt([&] { fanin<0><<<g, 768>>>(comb, priv, tickets, spin); });
      const float t_plain = timeIt([&] { fanin<1><<<g, 768>>>(comb, priv, tickets, spin); });
      const float t_ticket = timeIt([&] { fanin<2><<<g, 768>>>(comb, priv, tickets, spin); });
      const float t_xcd_wg = timeIt([&] { fanin<3><<<g, 768>>>(comb, priv, tickets, spin); });
      const float t_xcd_agent = timeIt([&] { fanin<4><<<g, 768>>>(comb, priv, tickets, spin); });
      std::printf("spin %3d, %3d workgroups: atomics %.2f us, private plain stores %.2f us, atomics + ticket %.2f us, per-XCD copies: workgroup-scope "
                  "atomics %.2f us, agent-scope atomics %.2f us per launch\n", spin, g, t_atomic, t_plain, t_ticket, t_xcd_wg, t_xcd_agent);
    }
  }
  // are workgroup-scope atomics of different workgroups of one XCD atomic with respect to each other, and is the result visible to the
  // host after the kernel?  every add is 1.0 here: entry sums over the 8 copies must equal the number of adds
  {
    hipMemset(comb, 0, 8 * 2048 * sizeof(double));
    const int g = 252, reps = 50;
    for (int i = 0; i < reps; ++i) fanin<3><<<g, 768>>>(comb, priv, tickets, -1);
    std::vector<double> h(8 * 2048);
    hipMemcpy(h.data(), comb, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    double total = 0, per_copy[8] = {0};
    for (int c = 0; c < 8; ++c)
      for (int e = 0; e < 2048; ++e) total += h[c * 2048 + e], per_copy[c] += h[c * 2048 + e];
    // adds per workgroup: tiles: entries with row < K, col < K, col >= row (1596 + 0) + pad column 56; pairs: 6 x (192 + 16)
    const double expect = double(g) * reps * (1596.0 + 56.0 + 6 * 208.0);
    std::printf("workgroup-scope atomics on per-XCD copies: total %.0f, expected %.0f (%s); per copy:", total, expect, total == expect ? "exact" : "MISMATCH");
    for (int c = 0; c < 8; ++c) std::printf(" %.0f", per_copy[c]);
    std::printf("\n");
  }
  return 0;
}
