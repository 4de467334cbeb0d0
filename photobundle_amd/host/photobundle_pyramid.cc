// photobundle_pyramid.cc -- see photobundle_pyramid.h
#include "photobundle_pyramid.h"

#include <ctime>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../include/pba.h"

namespace {
inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * n - 2 - i;
  return i;
}
}  // namespace

// rows are independent in both passes: a few host threads (integer / per-pixel arithmetic, results unchanged)
constexpr int kPyrThreads = 4;

void pyrDownU8(const uint8_t* src, int rows, int cols, std::vector<uint8_t>& dst) {
  const int drows = (rows + 1) / 2, dcols = (cols + 1) / 2;
  dst.assign((size_t)drows * dcols, 0);
  static const int w[5] = {1, 4, 6, 4, 1};
  std::vector<int> hrow((size_t)dcols);
  std::vector<int> hbuf((size_t)rows * dcols);
#pragma omp parallel for schedule(static) num_threads(kPyrThreads)
  for (int y = 0; y < rows; ++y) {           // horizontal pass at the even columns, integer sums
    const uint8_t* s = src + (size_t)y * cols;
    for (int x = 0; x < dcols; ++x) {
      int acc = 0;
      for (int k = 0; k < 5; ++k) acc += w[k] * (int)s[reflect101(2 * x + k - 2, cols)];
      hbuf[(size_t)y * dcols + x] = acc;
    }
  }
#pragma omp parallel for schedule(static) num_threads(kPyrThreads)
  for (int y = 0; y < drows; ++y) {
    for (int x = 0; x < dcols; ++x) {
      int acc = 0;
      for (int k = 0; k < 5; ++k) acc += w[k] * hbuf[(size_t)reflect101(2 * y + k - 2, rows) * dcols + x];
      dst[(size_t)y * dcols + x] = (uint8_t)((acc + 128) >> 8);
    }
  }
}

void resizeBilinearF32(const float* src, int rows, int cols, int drows, int dcols, std::vector<float>& dst) {
  dst.assign((size_t)drows * dcols, 0.0f);
  const double scale_x = (double)cols / dcols, scale_y = (double)rows / drows;
  std::vector<int> sx(dcols);
  std::vector<float> ax(dcols);
  for (int dx = 0; dx < dcols; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int ix = (int)std::floor(fx);
    fx -= ix;
    if (ix < 0) { fx = 0.f; ix = 0; }
    if (ix >= cols - 1) { fx = 0.f; ix = cols - 1; }
    sx[dx] = ix; ax[dx] = fx;
  }
#pragma omp parallel for schedule(static) num_threads(kPyrThreads)
  for (int dy = 0; dy < drows; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int iy = (int)std::floor(fy);
    fy -= iy;
    if (iy < 0) { fy = 0.f; iy = 0; }
    if (iy >= rows - 1) { fy = 0.f; iy = rows - 1; }
    const float* r0 = src + (size_t)iy * cols;
    const float* r1 = src + (size_t)std::min(iy + 1, rows - 1) * cols;
    for (int dx = 0; dx < dcols; ++dx) {
      const int x0 = sx[dx], x1 = std::min(x0 + 1, cols - 1);
      const float a1 = ax[dx], a0 = 1.f - a1;
      const float h0 = r0[x0] * a0 + r0[x1] * a1;
      const float h1 = r1[x0] * a0 + r1[x1] * a1;
      dst[(size_t)dy * dcols + dx] = h0 * (1.f - fy) + h1 * fy;
    }
  }
}

PhotometricBundleAdjustmentPyr::PhotometricBundleAdjustmentPyr(int num_levels, const Calibration& calib,
                                                               const ImageSize& im_size, const Options& options)
    : _rows(im_size.rows), _cols(im_size.cols) {
  if (num_levels <= 0) throw std::runtime_error("num_levels must be > 0");
  Calibration calib_pyr(calib);
  ImageSize size_pyr(im_size);
  for (int i = 0; i < num_levels; ++i) {
    _pyr.emplace_back(new PhotometricBundleAdjustment(calib_pyr, size_pyr, options));
    _sizes.push_back(size_pyr);
    calib_pyr = calib_pyr.pyrDown();
    size_pyr = size_pyr.pyrDown();
  }
  _im_pyr.resize(num_levels);
  _z_pyr.resize(num_levels);
}

PhotometricBundleAdjustmentPyr::~PhotometricBundleAdjustmentPyr() {}

void PhotometricBundleAdjustmentPyr::addFrame(const uint8_t* image, const float* depth, const Mat44& T, Result* result) {
  const int n = (int)_pyr.size();
  timespec ts0, ts1;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  _im_pyr[0].assign(image, image + (size_t)_rows * _cols);
  _z_pyr[0].assign(depth, depth + (size_t)_rows * _cols);
  // Intensity descriptors: the image pyramid is built on the device, level to level, straight into every level's ring slot
  // (pba_set_frame_pyr_down; the host front-end of each level gets its u8 image back, 1/4 of the previous one).
  // PBA_HOST_PYRAMID (test hook) / multi-channel descriptors: cv::pyrDown semantics on the host, every level uploads.
  static const bool host_pyramid = std::getenv("PBA_HOST_PYRAMID") != nullptr;
  const bool device_pyramid = !host_pyramid && n > 1 && _pyr[0]->_options_ptr->descriptorType == Options::DescriptorType::Intensity;
  if (device_pyramid) {
    const int window = _pyr[0]->_options_ptr->slidingWindowSize;
    const int slot = (int)(_pyr[0]->_frame_id % window);
    if (pba_set_frame_u8(_pyr[0]->_engine, slot, image) != PBA_OK) throw std::runtime_error(std::string("pba_set_frame_u8: ") + pba_last_error(_pyr[0]->_engine));
    _pyr[0]->_frame_resident = true;
  }
  for (int i = 1; i < n; ++i) {
    if (device_pyramid) {
      const int window = _pyr[i]->_options_ptr->slidingWindowSize;
      _im_pyr[i].resize((size_t)_sizes[i].rows * _sizes[i].cols);
      if (pba_set_frame_pyr_down(_pyr[i]->_engine, (int)(_pyr[i]->_frame_id % window), _pyr[i - 1]->_engine,
                                 (int)(_pyr[i - 1]->_frame_id % window), _im_pyr[i].data()) != PBA_OK)
        throw std::runtime_error(std::string("pba_set_frame_pyr_down: ") + pba_last_error(_pyr[i]->_engine));
      _pyr[i]->_frame_resident = true;
    } else {
      pyrDownU8(_im_pyr[i - 1].data(), _sizes[i - 1].rows, _sizes[i - 1].cols, _im_pyr[i]);
    }
    resizeBilinearF32(_z_pyr[i - 1].data(), _sizes[i - 1].rows, _sizes[i - 1].cols, _sizes[i].rows, _sizes[i].cols, _z_pyr[i]);
  }
  clock_gettime(CLOCK_MONOTONIC, &ts1);
  std::fprintf(stderr, "pyramid build %.2f ms (%d levels)\n", 1e3 * (double)(ts1.tv_sec - ts0.tv_sec) + 1e-6 * (double)(ts1.tv_nsec - ts0.tv_nsec), n);
  Mat44 T_init(T);
  Result last;
  for (int i = n - 1; i >= 0; --i) {
    std::fprintf(stderr, "Pyramid level %d\n", i);
    Result tmp;
    _pyr[i]->addFrame(_im_pyr[i].data(), _z_pyr[i].data(), T_init, &tmp);
    const size_t m = tmp.poses.size();
    if (m >= 2) {
      // an optimisation ran at this level: hand its refined frame-to-frame pose to the next finer level
      // (T_w_i = T_w_(i-1) * inv(T_i)  =>  T_i = inv(T_w_i) * T_w_(i-1), reference src/trajectory.cc:7-16)
      T_init = tmp.poses[m - 1].inverse() * tmp.poses[m - 2];
    }
    if (i == 0) last = tmp;
  }
  if (result) *result = last;
}
