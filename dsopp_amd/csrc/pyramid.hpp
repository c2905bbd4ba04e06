// Device-resident image pyramid: per level one "texel image" in HBM.
//
// HBM layout (DESIGN.md §Data layout): a level is a row-major H x W array of 4-scalar texels
//     { I, mask, dI/dx, dI/dy }            (scalar = double: 32 B / texel, float: 16 B / texel)
// i.e. the reference's PixelInfo<1>::data_ triplet (src/features/include/features/camera/pixel_map.hpp:79-132) padded
// to a power-of-two size with the CameraMask byte of the same pixel folded into the spare lane (1 = valid, 0 = masked;
// src/sensors/camera_calibration/include/sensors/camera_calibration/mask/camera_mask.hpp:48-89).  The bilinear footprint of a sample is
// then two 2-texel segments (2 x 64 B in double) and the mask lookup at round(x), round(y) — always one of the four
// footprint texels — costs no extra memory request.  {I, mask} sit in the first half so the residual-only sweep
// needs 16 B per texel.
#pragma once
#include "common.hpp"

namespace dsopp_hip {

template <typename S>
struct alignas(sizeof(S) * 4) Texel {
  S I, mask, Ix, Iy;
};

struct LevelView {
  const void *texels;  // Texel<S>*
  int width, height;
};

}  // namespace dsopp_hip

struct dsopp_hip_pyramid {
  dsopp_hip::StreamRef sr;
  int width = 0, height = 0, levels = 0, dtype = DSOPP_HIP_F64;
  void *texels[DSOPP_HIP_MAX_LEVELS] = {nullptr};  // Texel<S>[h_l * w_l]
  void *planes[DSOPP_HIP_MAX_LEVELS] = {nullptr};  // S[h_l * w_l] scalar plane (downscale source)
  void *staging_u8 = nullptr;                       // level-0 u8 image / vignette / mask staging
  void *staging_vig = nullptr;
  double *lut_dev = nullptr;                        // 256 doubles
  // Recorded on the pyramid's stream behind every write of the texels (build / build_device / set_level / set_mask).  A
  // consumer that reads the texels on another stream orders itself behind it with waitReady(): build_device only ENQUEUES
  // work, so without this a solve on the aligner's or the window's own stream could sample a half-built image.
  hipEvent_t ready = nullptr;
  void markReady() {
    if (!ready) HIP_CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ready, sr.stream));
  }
  /** everything enqueued on `consumer` after this call sees the texels of the last build (no host synchronisation) */
  void waitReady(hipStream_t consumer) const {
    if (ready && consumer != sr.stream) HIP_CHECK(hipStreamWaitEvent(consumer, ready, 0));
  }
  int w(int l) const { return width >> l; }
  int h(int l) const { return height >> l; }
  size_t elemSize() const { return dtype == DSOPP_HIP_F64 ? sizeof(double) : sizeof(float); }
  dsopp_hip::LevelView view(int l) const { return dsopp_hip::LevelView{texels[l], w(l), h(l)}; }
};
