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

#include <mutex>

// Pointer members of descriptors that live in device memory: in the device pass they are typed as address_space(1) pointers, so
// every access through them is a global_load / global_store instead of a FLAT access (the compiler cannot know where a pointer
// it read from memory points to); in the host pass they are ordinary pointers of the same size.
#if defined(__HIP_DEVICE_COMPILE__)
#define DSOPP_HBM __attribute__((address_space(1)))
#else
#define DSOPP_HBM
#endif
// Data a kernel only reads and that an earlier kernel wrote (pair constants, control blocks): typed as constant-address-space
// so that uniform loads stay scalar loads even after the kernel's own stores (which "may alias" any generic or global pointer
// and otherwise force the compiler to re-load such values with vector loads inside loops).
#if defined(__HIP_DEVICE_COMPILE__)
#define DSOPP_CONSTANT __attribute__((address_space(4)))
#else
#define DSOPP_CONSTANT
#endif
typedef double DSOPP_HBM hbm_f64;
typedef unsigned char DSOPP_HBM hbm_u8;
typedef int DSOPP_HBM hbm_i32;
typedef void DSOPP_HBM hbm_void;
/** host side: a device allocation's address as the descriptor member type (identity outside the device pass) */
template <typename T>
inline T DSOPP_HBM *hbm(T *p) {
  return (T DSOPP_HBM *)p;
}

namespace dsopp_hip {

template <typename S>
struct alignas(sizeof(S) * 4) Texel {
  S I, mask, Ix, Iy;
};

/** Pointers read out of a descriptor in memory are generic to the compiler, which then emits FLAT loads / stores (both
 *  counters, LDS aperture check).  Everything the sweeps touch lives in HBM: glb() re-types such a pointer as
 *  address_space(1) so that the accesses become global_load / global_store. */
template <typename T>
using GlobalPtr = T __attribute__((address_space(1))) *;
template <typename T>
__device__ __forceinline__ GlobalPtr<T> glb(T *p) {
  return (GlobalPtr<T>)p;
}

/** one texel {I, mask, Ix, Iy} as a single aligned vector load from HBM */
template <typename S>
__device__ __forceinline__ Texel<S> loadTexel(const Texel<S> DSOPP_HBM *p) {
  typedef S Vec4 __attribute__((ext_vector_type(4)));
  const Vec4 q = *(GlobalPtr<const Vec4>)p;
  Texel<S> t;
  t.I = q.x;
  t.mask = q.y;
  t.Ix = q.z;
  t.Iy = q.w;
  return t;
}
#if defined(__HIP_DEVICE_COMPILE__)
template <typename S>
__device__ __forceinline__ Texel<S> loadTexel(const Texel<S> *p) {  // a generic pointer that is known to address HBM
  return loadTexel((const Texel<S> DSOPP_HBM *)p);
}
#endif

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
  void *h_image = nullptr;                          // pinned copy of the caller's 8-bit image (dsopp_hip_pyramid_build): the upload is a true DMA, the call does not wait for it
  double *lut_dev = nullptr;                        // 256 doubles
  // Recorded on the pyramid's stream behind every write of the texels (build / build_device / set_level / set_mask).  A
  // consumer that reads the texels on another stream orders itself behind it with waitReady(): build_device only ENQUEUES
  // work, so without this a solve on the aligner's or the window's own stream could sample a half-built image.
  hipEvent_t ready = nullptr;
  void markReady() {
    if (!ready) HIP_CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ready, sr.stream));
    std::lock_guard<std::mutex> lock(iplane_mutex);
    for (bool &v : iplane_valid) v = false;  // the texels changed: the intensity planes derived from them are stale
    ++generation;  // (a window that borrowed this pyramid compares it before its next sweep and rebuilds the plane it samples)
  }
  // Intensity planes (built on demand by the first consumer, pyramid.hip: intensityPlane): one word per pixel — the intensity
  // with the CameraMask bit in the lowest mantissa bit — tiled per 64-byte segment: 8-byte words, 4 x 2 pixels (f64 pyramids);
  // 4-byte words, 4 x 4 pixels (f32 pyramids, the reference's -DUSE_FLOAT build; round 6).  What the residual-only sweeps read
  // instead of the 32- / 16-byte texels: a bilinear footprint then lies in 1.9 (f32: 1.6) segments on average instead of 3, and
  // a pattern's 8 footprints share them.
  mutable void *iplane[DSOPP_HIP_MAX_LEVELS] = {nullptr};
  mutable bool iplane_valid[DSOPP_HIP_MAX_LEVELS] = {false};
  unsigned generation = 0;  // number of rewrites of the texels (markReady)
  mutable hipEvent_t iplane_ready[DSOPP_HIP_MAX_LEVELS] = {nullptr};
  mutable hipStream_t iplane_stream[DSOPP_HIP_MAX_LEVELS] = {nullptr};
  mutable std::mutex iplane_mutex;
  int itilesX(int l) const { return (w(l) + 3) / 4; }
  int itilesY(int l) const { return dtype == DSOPP_HIP_F64 ? (h(l) + 1) / 2 : (h(l) + 3) / 4; }
  /** the level's intensity plane, valid for everything enqueued on `consumer` after the call */
  const void *intensityPlane(int level, hipStream_t consumer) const;
  /** everything enqueued on `consumer` after this call sees the texels of the last build (no host synchronisation) */
  void waitReady(hipStream_t consumer) const {
    if (ready && consumer != sr.stream) HIP_CHECK(hipStreamWaitEvent(consumer, ready, 0));
  }
  int w(int l) const { return width >> l; }
  int h(int l) const { return height >> l; }
  size_t elemSize() const { return dtype == DSOPP_HIP_F64 ? sizeof(double) : sizeof(float); }
  dsopp_hip::LevelView view(int l) const { return dsopp_hip::LevelView{texels[l], w(l), h(l)}; }
};
