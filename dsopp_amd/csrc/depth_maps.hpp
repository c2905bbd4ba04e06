// Reference depth maps of the newest keyframe (createReferenceDepthMaps, src/tracker/tracker/src/create_depth_maps.cpp:18-147),
// resident in HBM: the step between the bundle adjustment (which owns the landmarks) and the coarse tracker (which
// aligns new frames against these maps).  Shared by pba.hip (producer) and align.hip (consumer).
#pragma once
#include <vector>

#include "common.hpp"

struct dsopp_hip_window;
struct dsopp_hip_depth_maps {
  dsopp_hip::StreamRef sr;  // borrowed from the window that produced the maps (reset to the device's default stream when that window goes first)
  dsopp_hip_window *owner = nullptr;  // the producing window while it lives: it detaches its maps when it is destroyed
  int levels = 0;
  std::vector<int> width, height;
  // per level two row-major H x W planes: weighted idepth sum and weight (energy::problem::DepthMap::map, (x, y)-indexed there)
  std::vector<dsopp_hip::DeviceBuffer<double>> idepth_sum, weight;
  // Reference points of each level as the LocalFrame depth-map constructor extracts them (PBA_INT/local_frame.hpp:367-392),
  // cached by the first dsopp_hip_aligner_push_reference_depth_maps of that level: the maps belong to one keyframe and
  // every frame tracked against it scans the same maps.  Intensities are sampled from `pyramid` (the keyframe's own).
  struct LevelPoints {
    int n = -1;  // -1: not extracted yet
    const void *pyramid = nullptr;
    dsopp_hip::DeviceBuffer<double> u, v, idepth, intensity;
    // recorded by the extracting stream behind the compaction: a consumer on ANOTHER stream (the optical-flow measure runs on the maps' own)
    // orders itself behind it
    hipEvent_t ready = nullptr;
    hipStream_t ordered = nullptr;  // the consumer stream that has already been ordered behind `ready` (one wait per extraction, not per frame)
    LevelPoints() = default;
    LevelPoints(const LevelPoints &) = delete;
    LevelPoints &operator=(const LevelPoints &) = delete;
    LevelPoints(LevelPoints &&o) noexcept
        : n(o.n), pyramid(o.pyramid), u(std::move(o.u)), v(std::move(o.v)), idepth(std::move(o.idepth)), intensity(std::move(o.intensity)), ready(o.ready), ordered(o.ordered) {
      o.ready = nullptr;
      o.n = -1;
    }
    ~LevelPoints() {
      if (ready) (void)hipEventDestroy(ready);
    }
    void markReady(hipStream_t producer) {
      if (!ready) HIP_CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(ready, producer));
      ordered = producer;
    }
    void orderBehind(hipStream_t consumer) {
      if (!ready || ordered == consumer) return;
      HIP_CHECK(hipStreamWaitEvent(consumer, ready, 0));
      ordered = consumer;
    }
  };
  mutable std::vector<LevelPoints> points;
  mutable dsopp_hip::DeviceBuffer<double> flow_scratch;  // per-workgroup partials of the optical-flow measure
  // its result, in pinned host memory the closing workgroup writes itself (the tracker asks for the flow on every frame: a 16-byte copy into
  // the caller's pageable array was a staged transfer — a copy kernel, 20 us inside hipMemcpyAsync and a second wait)
  mutable double *h_flow = nullptr;
  dsopp_hip_depth_maps() = default;
  dsopp_hip_depth_maps(const dsopp_hip_depth_maps &) = delete;
  dsopp_hip_depth_maps &operator=(const dsopp_hip_depth_maps &) = delete;
  ~dsopp_hip_depth_maps() {
    if (h_flow) (void)hipHostFree(h_flow);
  }
};
