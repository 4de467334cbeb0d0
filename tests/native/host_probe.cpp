// Test shim (NOT part of the product): C hooks over pieces of the drop-in host library that only tests call -- the Options stream
// operator and Calibration helpers (reference src/photobundle.h:81-82, src/calibration.h:40-70), the host descriptor channels
// (host/imgproc.h) and the pyramid helpers (host/photobundle_pyramid.h).  Links libphotobundle.so; bound through ctypes by
// tests/test_host_api_cpu.py, tests/test_pyramid_cpu.py and tests/test_gpu_producers.py.
#include "../../photobundle_amd/host/photobundle.h"
#include "../../photobundle_amd/host/photobundle_pyramid.h"
#include "../../photobundle_amd/host/imgproc.h"

#include <algorithm>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

// test hook (tests/test_host_api_cpu.py): the small API pieces of the reference headers that nothing else in the library calls
extern "C" int pb_api_probe(char* text, int cap, double* out8) {
  PhotometricBundleAdjustment::Options o;
  o.patchRadius = 3;
  o.descriptorType = PhotometricBundleAdjustment::Options::DescriptorType::BitPlanes;
  std::ostringstream ss;
  ss << o;
  const std::string t = ss.str();
  if ((int)t.size() + 1 > cap) return -1;
  std::memcpy(text, t.c_str(), t.size() + 1);
  Mat33 K = Mat33::Identity();
  K(0, 0) = 718.856; K(1, 1) = 718.856; K(0, 2) = 607.1928; K(1, 2) = 185.2157;
  Calibration c(K, 0.5372);
  const double uvd[3] = {700.0, 100.0, 12.5};
  const Vec3 X = c.triangulate(uvd);
  out8[0] = X[0]; out8[1] = X[1]; out8[2] = X[2];
  const double Xp[3] = {X[0], X[1], X[2]};
  double uv[2];
  c.project(Xp, uv);
  out8[3] = uv[0]; out8[4] = uv[1];
  c.scale(2.0);
  out8[5] = c.fx(); out8[6] = c.cx(); out8[7] = c.b();
  c.scale(0.5);      // ignored (s <= 1)
  return (c.fx() == out8[5]) ? 0 : -2;
}

extern "C" int pb_descriptor_channels(const uint8_t* img, int rows, int cols, int kind, float* out) {
  std::vector<Image_<float>> ch;
  const size_t n = (size_t)rows * cols;
  if (kind == 2) {
    imgproc::computeBitPlanes(img, rows, cols, ch);
  } else {
    ch.resize(3);
    for (auto& c : ch) c.resize(rows, cols);
    for (size_t i = 0; i < n; ++i) ch[0].d[i] = (float)img[i];
    imgproc::imgradient(img, rows, cols, ch[1].data(), ch[2].data());
  }
  for (size_t k = 0; k < ch.size(); ++k) std::copy(ch[k].d.begin(), ch[k].d.end(), out + k * n);
  return (int)ch.size();
}

// C hooks (tests bind them through ctypes)
extern "C" void pb_pyr_down_u8(const uint8_t* src, int rows, int cols, uint8_t* dst) {
  std::vector<uint8_t> d;
  pyrDownU8(src, rows, cols, d);
  std::copy(d.begin(), d.end(), dst);
}
extern "C" void pb_resize_bilinear_f32(const float* src, int rows, int cols, int drows, int dcols, float* dst) {
  std::vector<float> d;
  resizeBilinearF32(src, rows, cols, drows, dcols, d);
  std::copy(d.begin(), d.end(), dst);
}
