// Device-side camera-model predicates shared by the BA sweeps and the aligner (templated on the evaluation scalar S).
#pragma once
#include <hip/hip_runtime.h>

namespace dsopp_hip {

template <typename S>
__device__ __forceinline__ bool insideROI(S u, S v, S width, S height) {
  // CameraModelBase::insideCameraROI — camera_model_base.hpp:52-60 (border 4)
  return (u >= S(4)) && (v >= S(4)) && (u <= width - S(5)) && (v <= height - S(5));
}
template <typename S>
__device__ __forceinline__ bool validIdepth(S idepth) {
  // CameraModelBase::validIdepth — camera_model_base.hpp:67-74
  return idepth > S(-1e-4) && idepth < S(1.0 / 0.001 + 1e1);
}


}  // namespace dsopp_hip
