// Reference depth maps of the newest keyframe (createReferenceDepthMaps, src/tracker/tracker/src/create_depth_maps.cpp:18-147),
// resident in HBM: the step between the bundle adjustment (which owns the landmarks) and the coarse tracker (which
// aligns new frames against these maps).  Shared by pba.hip (producer) and align.hip (consumer).
#pragma once
#include <vector>

#include "common.hpp"

struct dsopp_hip_depth_maps {
  dsopp_hip::StreamRef sr;  // borrowed from the window that produced the maps
  int levels = 0;
  std::vector<int> width, height;
  // per level two row-major H x W planes: weighted idepth sum and weight (energy::problem::DepthMap::map, (x, y)-indexed there)
  std::vector<dsopp_hip::DeviceBuffer<double>> idepth_sum, weight;
};
