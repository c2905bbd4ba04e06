// The immature landmarks of one keyframe, resident on the device across frames (track::landmarks::ImmatureTrackingLandmark as
// struct-of-arrays, src/track/landmarks/include/track/landmarks/immature_tracking_landmark.hpp:93-106).  Shared by
// depth_estimation.hip (owner: the per-frame depth estimator) and pba.hip (the landmark activator reads the estimator
// state and writes the refined inverse depth back).
#pragma once
#include "common.hpp"

struct dsopp_hip_immature_set {
  dsopp_hip::StreamRef sr;
  int n = 0;
  dsopp_hip::DeviceBuffer<double> d_in;      // projection 2n | direction 3n | patch 8n | gradient 2n
  dsopp_hip::DeviceBuffer<double> d_io;      // idepth_min | idepth_max | uniqueness | search_pixel_interval
  dsopp_hip::DeviceBuffer<uint8_t> d_flags;  // status | traced
  void *h_stage = nullptr;                   // pinned read-back staging
  size_t h_stage_bytes = 0;
  void *h_tables = nullptr;                  // pinned descriptor tables of a batched estimate led by this set
  dsopp_hip::DeviceBuffer<char> d_tables;
  hipEvent_t tables_copied = nullptr;        // the previous batch's table upload has left the pinned buffer
};
