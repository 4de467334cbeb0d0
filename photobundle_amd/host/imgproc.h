// imgproc.h -- channel producers of the multi-channel descriptors (reference src/imgproc.{h,cc}) on the host.
//
// The reference builds its descriptor channels with Eigen / SSE / OpenCV; this is the same arithmetic as plain loops:
//   imgradient        imgproc.cc:27-95, imgproc.h:48-58   0.5 * central difference into float, zero one-pixel border
//   censusTransform   imgproc.cc:126-197                  bit b set when the b-th 3x3 neighbour (row-major, centre
//                                                         skipped) >= centre, zero border
//   computeBitPlanes  imgproc.cc:199-245, imgproc.h:44-46 census of the 3x3 (sigma 1) smoothed frame, its eight bit
//                                                         planes as float, each smoothed 5x5 (sigma 1.5)
// The two cv::GaussianBlur calls are OpenCV's (not installed here, version unpinned in the reference): the 8-bit one is
// restated in OpenCV's 8-bit fixed point (kernel cvRound(k * 256), result (sum + 2^15) >> 16), the float one in the
// symmetric form k0 c + k1 (l1 + r1) + k2 (l2 + r2), both with BORDER_REFLECT_101.
#ifndef PHOTOBUNDLE_AMD_IMGPROC_H
#define PHOTOBUNDLE_AMD_IMGPROC_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "types.h"

namespace imgproc {

template <class TSrc>
inline void imgradient(const TSrc* src, int rows, int cols, float* Ix, float* Iy) {
  std::fill(Ix, Ix + (size_t)rows * cols, 0.0f);
  std::fill(Iy, Iy + (size_t)rows * cols, 0.0f);
  for (int y = 1; y < rows - 1; ++y) {
    const TSrc* s = src + (size_t)y * cols;
    float* ix = Ix + (size_t)y * cols;
    float* iy = Iy + (size_t)y * cols;
    for (int x = 1; x < cols - 1; ++x) {
      ix[x] = 0.5f * ((float)s[x + 1] - (float)s[x - 1]);
      iy[x] = 0.5f * ((float)s[x + cols] - (float)s[x - cols]);
    }
  }
}

inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; }
  return i;
}

// cv::getGaussianKernel(n, sigma > 0, CV_32F)
inline void gaussianKernel(int n, double sigma, float* k) {
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0.0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    k[i] = (float)std::exp(scale2x * x * x);
    sum += k[i];
  }
  sum = 1.0 / sum;
  for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

inline void gaussianBlur3x3(const uint8_t* src, int rows, int cols, double sigma, uint8_t* dst) {
  float kf[3];
  gaussianKernel(3, sigma, kf);
  int k[3];
  for (int i = 0; i < 3; ++i) k[i] = (int)std::nearbyint((double)kf[i] * 256.0);
  std::vector<int> tmp((size_t)rows * cols);
  for (int y = 0; y < rows; ++y) {
    const uint8_t* s = src + (size_t)y * cols;
    int* t = tmp.data() + (size_t)y * cols;
    for (int x = 0; x < cols; ++x) t[x] = k[0] * s[reflect101(x - 1, cols)] + k[1] * s[x] + k[2] * s[reflect101(x + 1, cols)];
  }
  for (int y = 0; y < rows; ++y) {
    const int* a = tmp.data() + (size_t)reflect101(y - 1, rows) * cols;
    const int* b = tmp.data() + (size_t)y * cols;
    const int* c = tmp.data() + (size_t)reflect101(y + 1, rows) * cols;
    for (int x = 0; x < cols; ++x) {
      const int r = (k[0] * a[x] + k[1] * b[x] + k[2] * c[x] + (1 << 15)) >> 16;
      dst[(size_t)y * cols + x] = (uint8_t)std::min(255, std::max(0, r));
    }
  }
}

inline void gaussianBlur5x5(const float* src, int rows, int cols, double sigma, float* dst) {
  float k[5];
  gaussianKernel(5, sigma, k);
  std::vector<float> tmp((size_t)rows * cols);
  for (int y = 0; y < rows; ++y) {
    const float* s = src + (size_t)y * cols;
    float* t = tmp.data() + (size_t)y * cols;
    for (int x = 0; x < cols; ++x) {
      float v = s[x] * k[2];
      v += (s[reflect101(x - 1, cols)] + s[reflect101(x + 1, cols)]) * k[1];
      v += (s[reflect101(x - 2, cols)] + s[reflect101(x + 2, cols)]) * k[0];
      t[x] = v;
    }
  }
  for (int y = 0; y < rows; ++y) {
    const float* r0 = tmp.data() + (size_t)y * cols;
    const float* m1 = tmp.data() + (size_t)reflect101(y - 1, rows) * cols;
    const float* p1 = tmp.data() + (size_t)reflect101(y + 1, rows) * cols;
    const float* m2 = tmp.data() + (size_t)reflect101(y - 2, rows) * cols;
    const float* p2 = tmp.data() + (size_t)reflect101(y + 2, rows) * cols;
    for (int x = 0; x < cols; ++x) {
      float v = r0[x] * k[2];
      v += (m1[x] + p1[x]) * k[1];
      v += (m2[x] + p2[x]) * k[0];
      dst[(size_t)y * cols + x] = v;
    }
  }
}

inline void censusTransform(const uint8_t* src, int rows, int cols, uint8_t* dst) {
  std::memset(dst, 0, (size_t)rows * cols);
  for (int y = 1; y < rows - 1; ++y) {
    const uint8_t* s = src + (size_t)y * cols;
    uint8_t* d = dst + (size_t)y * cols;
    for (int x = 1; x < cols - 1; ++x) {
      const uint8_t c = s[x];
      d[x] = (uint8_t)(((s[x - cols - 1] >= c) << 0) | ((s[x - cols] >= c) << 1) | ((s[x - cols + 1] >= c) << 2) | ((s[x - 1] >= c) << 3) |
                       ((s[x + 1] >= c) << 4) | ((s[x + cols - 1] >= c) << 5) | ((s[x + cols] >= c) << 6) | ((s[x + cols + 1] >= c) << 7));
    }
  }
}

// eight channels of rows*cols floats
inline void computeBitPlanes(const uint8_t* image, int rows, int cols, std::vector<Image_<float>>& dst, float sigma_ct = 1.0f,
                             float sigma_bp = 1.5f) {
  const size_t n = (size_t)rows * cols;
  std::vector<uint8_t> smooth, census(n);
  const uint8_t* src = image;
  if (sigma_ct > 0.0f) { smooth.resize(n); gaussianBlur3x3(image, rows, cols, sigma_ct, smooth.data()); src = smooth.data(); }
  censusTransform(src, rows, cols, census.data());
  dst.resize(8);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < 8; ++b) {
    dst[b].resize(rows, cols);
    std::vector<float> plane(n);
    for (size_t i = 0; i < n; ++i) plane[i] = (float)((census[i] & (1 << b)) >> b);
    if (sigma_bp > 0.0f) gaussianBlur5x5(plane.data(), rows, cols, sigma_bp, dst[b].data());
    else std::copy(plane.begin(), plane.end(), dst[b].data());
  }
}

}  // namespace imgproc

#endif
