// Probe: what does a grid-wide barrier cost on MI355X next to a dependent kernel launch (~3.8 us)?
// cooperative launch of B workgroups x 256 threads, N x cooperative_groups::grid.sync().
//   hipcc --offload-arch=gfx950 -O3 grid_sync_probe.hip -o grid_sync_probe && timeout 60 ./grid_sync_probe
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
namespace cg = cooperative_groups;

__global__ void syncLoop(int n, double *out) {
  cg::grid_group grid = cg::this_grid();
  double acc = 0;
  for (int i = 0; i < n; ++i) {
    acc += threadIdx.x * 1e-9;
    grid.sync();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}
__global__ void tiny(double *out) { out[threadIdx.x & 1] += 1; }

int main() {
  double *d;
  hipMalloc(&d, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int blocks : {8, 31, 64, 256}) {
    int n = 200;
    void *args[] = {&n, &d};
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a, 0);
      hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<void *>(syncLoop), dim3(blocks), dim3(256), args, 0, 0);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      if (rep) std::printf("blocks %3d: %s, %.2f us per grid.sync\n", blocks, hipGetErrorString(e), ms * 1e3 / n);
    }
  }
  hipEventRecord(a, 0);
  for (int i = 0; i < 200; ++i) tiny<<<31, 256>>>(d);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::printf("200 dependent launches of a trivial kernel: %.2f us each\n", ms * 1e3 / 200);
  return 0;
}
