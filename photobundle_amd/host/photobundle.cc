// photobundle.cc -- PhotometricBundleAdjustment on the MI355X engine.
//
// addFrame() restates the reference's CPU front-end (reference src/photobundle.cc:482-615: trajectory update,
// ZNCC visibility test, saliency non-maximum selection, descriptor extraction); optimize() assembles the same
// problem the reference hands to Ceres (reference :764-876) and solves it through the C-ABI of include/pba.h.
#include "photobundle.h"

#include <ostream>
#include <sstream>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <omp.h>
#include <map>
#include <stdexcept>

#include "../../include/pba.h"
#include "imgproc.h"
#include "utils.h"

namespace {

typedef PhotometricBundleAdjustment::Options::DescriptorType DescriptorType;

DescriptorType DescriptorTypeFromString(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return std::tolower(c); });
  if (s == "intensity") return DescriptorType::Intensity;
  if (s == "intensityandgradient") return DescriptorType::IntensityAndGradient;
  if (s == "bitplanes") return DescriptorType::BitPlanes;
  std::fprintf(stderr, "unknown descriptorType %s, using Intensity\n", s.c_str());
  return DescriptorType::Intensity;
}

inline int PatchSizeFromRadius(int r) { return (2 * r + 1) * (2 * r + 1); }

// reference photobundle.cc:262-294: floor-based bilinear lookup with fill value at the far border
template <class Image>
inline float interp2(const Image& I, float xf, float yf, float fillval = 0.0f) {
  const int max_cols = I.cols() - 1, max_rows = I.rows() - 1;
  const int xi = (int)std::floor(xf), yi = (int)std::floor(yf);
  xf -= xi; yf -= yi;
  if (xi >= 0 && xi < max_cols && yi >= 0 && yi < max_rows) {
    const float wx = 1.0 - xf;
    return (1.0 - yf) * (I(yi, xi) * wx + I(yi, xi + 1) * xf) + yf * (I(yi + 1, xi) * wx + I(yi + 1, xi + 1) * xf);
  }
  if (xi == max_cols && yi < max_rows) return (xf > 0) ? fillval : (1.0 - yf) * I(yi, xi) + yf * I(yi + 1, xi);
  if (yi == max_rows && xi < max_cols) return (yf > 0) ? fillval : (1.0 - xf) * I(yi, xi) + xf * I(yi, xi + 1);
  if (xi == max_cols && yi == max_rows) return (xf > 0 || yf > 0) ? fillval : I(yi, xi);
  return fillval;
}

// reference photobundle.cc:315-361 ZnccPatch_<2, float>
struct ZnccPatch {
  static constexpr int R = 2, N = 25;
  float data[N];
  float norm = 0.f;
  template <class Image>
  void set(const Image& I, double u, double v) {
    const float x = (float)u, y = (float)v;
    int k = 0;
    for (int r = -R; r <= R; ++r) for (int c = -R; c <= R; ++c) data[k++] = interp2(I, c + x, r + y, 0.0f);
    float sum = 0.f;
    for (int i = 0; i < N; ++i) sum += data[i];
    const float mean = sum / (float)N;
    float n2 = 0.f;
    for (int i = 0; i < N; ++i) { data[i] -= mean; n2 += data[i] * data[i]; }
    norm = std::sqrt(n2);
  }
  float score(const ZnccPatch& o) const {
    const float d = norm * o.norm;
    if (!(d > 1e-6)) return -1.0f;
    float dot = 0.f;
    for (int i = 0; i < N; ++i) dot += data[i] * o.data[i];
    return dot / d;
  }
};

// view of the caller's u8 frame with the Eigen-map call syntax
struct U8View {
  const uint8_t* p; int rows_, cols_;
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  float operator()(int r, int c) const { return (float)p[(size_t)r * cols_ + c]; }
};

// reference photobundle.cc:617-644
std::vector<double> MakePatchWeights(int radius, bool do_gaussian) {
  const int n = PatchSizeFromRadius(radius);
  std::vector<double> ret(n, 1.0);
  if (!do_gaussian) return ret;
  double sum = 0.0;
  for (int r = -radius, i = 0; r <= radius; ++r)
    for (int c = -radius; c <= radius; ++c, ++i) { ret[i] = std::exp(-0.5 * ((double)(r * r) + (double)(c * c))); sum += ret[i]; }
  for (double& w : ret) w /= sum;
  return ret;
}

// reference photobundle.cc:646-667 with the Ceres rotation.h conventions (matrix -> quaternion -> angle-axis)
void PoseToParams(const Mat44& T, double* p) {
  double q[4];
  const double trace = T(0, 0) + T(1, 1) + T(2, 2);
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (T(2, 1) - T(1, 2)) * t; q[2] = (T(0, 2) - T(2, 0)) * t; q[3] = (T(1, 0) - T(0, 1)) * t;
  } else {
    int i = 0;
    if (T(1, 1) > T(0, 0)) i = 1;
    if (T(2, 2) > T(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(T(i, i) - T(j, j) - T(k, k) + 1.0);
    q[i + 1] = 0.5 * t; t = 0.5 / t;
    q[0] = (T(k, j) - T(j, k)) * t; q[j + 1] = (T(j, i) + T(i, j)) * t; q[k + 1] = (T(k, i) + T(i, k)) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double k = 2.0;
  if (s2 > 0.0) {
    const double s = std::sqrt(s2);
    const double two_theta = 2.0 * ((q[0] < 0.0) ? std::atan2(-s, -q[0]) : std::atan2(s, q[0]));
    k = two_theta / s;
  }
  p[0] = q[1] * k; p[1] = q[2] * k; p[2] = q[3] * k;
  p[3] = T(0, 3); p[4] = T(1, 3); p[5] = T(2, 3);
}

Mat44 ParamsToPose(const double* p) {
  Mat44 T = Mat44::Identity();
  const double theta2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  if (theta2 > 2.220446049250313e-16) {
    const double theta = std::sqrt(theta2);
    const double wx = p[0] / theta, wy = p[1] / theta, wz = p[2] / theta;
    const double c = std::cos(theta), s = std::sin(theta), oc = 1.0 - c;
    T(0, 0) = c + wx * wx * oc;       T(1, 0) = wz * s + wx * wy * oc;  T(2, 0) = -wy * s + wx * wz * oc;
    T(0, 1) = wx * wy * oc - wz * s;  T(1, 1) = c + wy * wy * oc;       T(2, 1) = wx * s + wy * wz * oc;
    T(0, 2) = wy * s + wx * wz * oc;  T(1, 2) = -wx * s + wy * wz * oc; T(2, 2) = c + wz * wz * oc;
  } else {
    T(0, 0) = 1; T(1, 0) = p[2]; T(2, 0) = -p[1];
    T(0, 1) = -p[2]; T(1, 1) = 1; T(2, 1) = p[0];
    T(0, 2) = p[1]; T(1, 2) = -p[0]; T(2, 2) = 1;
  }
  T(0, 3) = p[3]; T(1, 3) = p[4]; T(2, 3) = p[5];
  return T;
}

void check(pba_engine* e, int rc, const char* what) {
  if (rc != PBA_OK)
    throw std::runtime_error(std::string(what) + ": " + pba_status_string(rc) + " (" + (e ? pba_last_error(e) : "") + ")");
}

}  // namespace

PhotometricBundleAdjustment::Options::Options(const utils::ConfigFile& cf)
    : maxNumPoints(cf.get<int>("maxNumPoints", 4096)),
      nonMaxSuppRadius(cf.get<int>("nonMaxSuppRadius", 1)),
      maskBlockRadius(cf.get<int>("maskBlockRadius", 1)),
      maxFrameDistance(cf.get<int>("maxFrameDistance", 1)),
      minScore(cf.get<double>("minScore", 0.75)),
      minValidDepth(cf.get<double>("minValidDepth", 0.01)),
      maxValidDepth(cf.get<double>("maxValidDepth", 1000.0)),
      slidingWindowSize(cf.get<int>("slidingWindowSize", 5)),
      patchRadius(cf.get<int>("patchRadius", 2)),
      doGaussianWeighting((bool)cf.get<int>("doGaussianWeighting", 0)),
      robustThreshold(cf.get<double>("robustThreshold", 0.05)),
      descriptorType(DescriptorTypeFromString(cf.get<std::string>("descriptorType", "Intensity"))),
      numThreads(cf.get<int>("numThreads", -1)),
      verbose((bool)cf.get<int>("verbose", 1)),
      device(cf.get<int>("device", 0)) {}

std::ostream& operator<<(std::ostream& os, const PhotometricBundleAdjustment::Options& o) {
  using DT = PhotometricBundleAdjustment::Options::DescriptorType;
  const char* dt = o.descriptorType == DT::BitPlanes ? "BitPlanes" : (o.descriptorType == DT::IntensityAndGradient ? "IntensityAndGradient" : "Intensity");
  os << "maxNumPoints = " << o.maxNumPoints << "\nnonMaxSuppRadius = " << o.nonMaxSuppRadius << "\nmaskBlockRadius = " << o.maskBlockRadius
     << "\nmaxFrameDistance = " << o.maxFrameDistance << "\nminScore = " << o.minScore << "\nminValidDepth = " << o.minValidDepth
     << "\nmaxValidDepth = " << o.maxValidDepth << "\nslidingWindowSize = " << o.slidingWindowSize << "\npatchRadius = " << o.patchRadius
     << "\ndoGaussianWeighting = " << (o.doGaussianWeighting ? 1 : 0) << "\nrobustThreshold = " << o.robustThreshold
     << "\ndescriptorType = " << dt << "\nnumThreads = " << o.numThreads << "\nverbose = " << (o.verbose ? 1 : 0) << "\ndevice = " << o.device << "\n";
  return os;
}

bool PhotometricBundleAdjustment::Result::Writer::add(const Result&) {
  std::fprintf(stderr, "Result::Writer needs cereal (dead code in the reference's default build too)\n");
  ++_counter;
  return false;
}
PhotometricBundleAdjustment::Result PhotometricBundleAdjustment::Result::FromFile(std::string) {
  throw std::runtime_error("Result::FromFile needs cereal (dead code in the reference's default build too)");
}

// reference photobundle.cc:151-257: the descriptor channels of one frame (DescriptorFrame::Create :225-248) and the
// saliency map summed over every channel's gradient magnitude (:213-221).  The per-channel gradient images the
// residuals sample (:172-175) are built by the engine on the device.
struct PhotometricBundleAdjustment::DescriptorFrame {
  uint32_t id;
  std::vector<Image_<float>> channels;
  Image_<float>& I;              // channel 0
  static std::vector<Image_<float>> MakeChannels(const uint8_t* img, int rows, int cols, DescriptorType type, int nt) {
    std::vector<Image_<float>> ch;
    const long n = (long)rows * cols;
    if (type == DescriptorType::BitPlanes) {
      imgproc::computeBitPlanes(img, rows, cols, ch);
      return ch;
    }
    ch.resize(type == DescriptorType::IntensityAndGradient ? 3 : 1);
    ch[0].resize(rows, cols);
#pragma omp parallel for schedule(static) num_threads(nt)
    for (long i = 0; i < n; ++i) ch[0].d[i] = (float)img[i];
    if (type == DescriptorType::IntensityAndGradient) {
      ch[1].resize(rows, cols);
      ch[2].resize(rows, cols);
      imgproc::imgradient(img, rows, cols, ch[1].data(), ch[2].data());     // :236-237, from the u8 frame
    }
    return ch;
  }
  DescriptorFrame(uint32_t frame_id, const uint8_t* img, int rows, int cols, DescriptorType type, int nt)
      : id(frame_id), channels(MakeChannels(img, rows, cols, type, nt)), I(channels[0]) {}
  // channel images produced elsewhere (the engine's device-side producer): [C][rows*cols]
  DescriptorFrame(uint32_t frame_id, std::vector<Image_<float>>&& ch) : id(frame_id), channels(std::move(ch)), I(channels[0]) {}
  // device front-end: the channel images never reach the host, only their number is known here
  DescriptorFrame(uint32_t frame_id, int n_channels) : id(frame_id), channels((size_t)n_channels), I(channels[0]) {}
  size_t numChannels() const { return channels.size(); }
  void computeSaliencyMap(Image_<float>& smap, int nt) const {
    const int rows = I.rows(), cols = I.cols();
    std::fill(smap.d.begin(), smap.d.end(), 0.0f);
    for (size_t k = 0; k < channels.size(); ++k) {
      const Image_<float>& C = channels[k];
#pragma omp parallel for schedule(static) num_threads(nt)
      for (int y = 1; y < rows - 1; ++y)
        for (int x = 1; x < cols - 1; ++x) {
          const float ix = 0.5f * (C(y, x + 1) - C(y, x - 1));
          const float iy = 0.5f * (C(y + 1, x) - C(y - 1, x));
          const float mag = std::fabs(ix) + std::fabs(iy);
          smap(y, x) = (k == 0) ? mag : smap(y, x) + mag;
        }
    }
  }
};

// reference photobundle.cc:366-446
struct PhotometricBundleAdjustment::ScenePoint {
  Vec3 X, X_original;
  std::vector<uint32_t> f;
  ZnccPatch patch;
  std::vector<double> descriptor;
  double saliency = 0.0;
  bool was_refined = false;
  int x0 = 0, y0 = 0;
  ScenePoint(const Vec3& X_, uint32_t f_id) : X(X_), X_original(X_) { f.reserve(8); f.push_back(f_id); }
  uint32_t refFrameId() const { return f.front(); }
  uint32_t lastFrameId() const { return f.back(); }
  size_t numFrames() const { return f.size(); }
};

PhotometricBundleAdjustment::PhotometricBundleAdjustment(const Calibration& calib, const ImageSize& image_size)
    : PhotometricBundleAdjustment(calib, image_size, Options()) {}

PhotometricBundleAdjustment::PhotometricBundleAdjustment(const Calibration& calib, const ImageSize& image_size,
                                                         const Options& options)
    : _calib(calib), _image_size(image_size), _options_ptr(new Options(options)) {
  // Multi-channel descriptor types: the reference's debug builds stop at `assert(p0.size() == w.size())`
  // (photobundle.cc:684: C P descriptor entries against P patch weights); its release builds run, with the weights
  // restarting per channel (:714-721), and that is what is built here.
  _mask.resize(_image_size.rows, _image_size.cols);
  _saliency_map.resize(_image_size.rows, _image_size.cols);
  _K_inv = calib.K().inverse();
  pba_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.rows = image_size.rows; cfg.cols = image_size.cols;
  cfg.max_frames = options.slidingWindowSize; cfg.radius = options.patchRadius;
  cfg.fx = calib.fx(); cfg.fy = calib.fy(); cfg.cx = calib.cx(); cfg.cy = calib.cy();
  cfg.huber = options.robustThreshold; cfg.device = options.device; cfg.flags = 0;
  cfg.channels = options.descriptorType == DescriptorType::BitPlanes ? 8 : (options.descriptorType == DescriptorType::IntensityAndGradient ? 3 : 1);
  check(nullptr, pba_create(&cfg, &_engine), "pba_create");
}

PhotometricBundleAdjustment::~PhotometricBundleAdjustment() { pba_destroy(_engine); }

namespace {
double wall_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}
}  // namespace

void PhotometricBundleAdjustment::addFrame(const uint8_t* I_ptr, const float* Z_ptr, const Mat44& T, Result* result) {
  const double t_enter = wall_ms();
  _trajectory.push_back(T, _frame_id);
  const Mat44 T_w = _trajectory.back();
  const Mat44 T_c = T_w.inverse();
  const int rows = _image_size.rows, cols = _image_size.cols;
  const U8View I{I_ptr, rows, cols};

  // host threads of the front-end loops: Options::numThreads if given, else at most 8 (the loops are short)
  const int nt = _options_ptr->numThreads > 0 ? _options_ptr->numThreads : std::max(1, std::min(8, omp_get_max_threads()));
  double t_ph[6] = {0, 0, 0, 0, 0, 0};
  double t_last = wall_ms();
  auto lap = [&](int k) { const double t = wall_ms(); t_ph[k] += t - t_last; t_last = t; };
  // the engine keeps its own device plane(s) of this frame in the ring slot id % window
  const int window = _options_ptr->slidingWindowSize;
  const int slot = (int)(_frame_id % window);
  const Options::DescriptorType dtype = _options_ptr->descriptorType;
  static const bool host_channels = std::getenv("PBA_HOST_CHANNELS") != nullptr;    // test hook: the host-side channel producers
  // r4: visibility, saliency, candidate scan and descriptor patches run on the device, on the frame that sits in the engine anyway
  // (pba_frontend_*); PBA_HOST_FRONTEND (test hook, implied by PBA_HOST_CHANNELS) keeps the host restatement below -- byte-identical
  static const bool host_frontend = std::getenv("PBA_HOST_FRONTEND") != nullptr || host_channels;
  const int n_desc_channels = dtype == Options::DescriptorType::BitPlanes ? 8 : (dtype == Options::DescriptorType::IntensityAndGradient ? 3 : 1);
  UniquePointer<DescriptorFrame> frame;
  if (!host_frontend) {
    if (dtype != Options::DescriptorType::Intensity) {
      const int32_t kind = dtype == Options::DescriptorType::BitPlanes ? PBA_DESCRIPTOR_BITPLANES : PBA_DESCRIPTOR_INTENSITY_AND_GRADIENT;
      check(_engine, pba_set_frame_descriptor_u8(_engine, slot, I_ptr, kind, 1.0f, 1.5f), "pba_set_frame_descriptor_u8");
    } else if (_frame_resident) {
      _frame_resident = false;                 // the pyramid class produced this level's frame on the device
    } else {
      check(_engine, pba_set_frame_u8(_engine, slot, I_ptr), "pba_set_frame_u8");
    }
    frame.reset(new DescriptorFrame(_frame_id, n_desc_channels));
    lap(5);
  } else if (dtype != Options::DescriptorType::Intensity && !host_channels) {
    // (host front-end) Multi-channel descriptors: the engine builds the channel images on the device from the u8 frame (0.47 MB up instead of
    // 5.6 / 15 MB) and the front-end reads them back (saliency, descriptor patches) instead of running DescriptorFrame::Create
    // on the CPU as well -- bit-identical images (tests/test_gpu_producers.py), ~190 ms less per KITTI frame for BitPlanes.
    const int32_t kind = dtype == Options::DescriptorType::BitPlanes ? PBA_DESCRIPTOR_BITPLANES : PBA_DESCRIPTOR_INTENSITY_AND_GRADIENT;
    const int C = dtype == Options::DescriptorType::BitPlanes ? 8 : 3;
    check(_engine, pba_set_frame_descriptor_u8(_engine, slot, I_ptr, kind, 1.0f, 1.5f), "pba_set_frame_descriptor_u8");
    std::vector<float> flat((size_t)C * rows * cols);
    check(_engine, pba_get_frame_channels_f32(_engine, slot, flat.data()), "pba_get_frame_channels_f32");
    std::vector<Image_<float>> ch(C);
    for (int k = 0; k < C; ++k) {
      ch[k].resize(rows, cols);
      std::copy(flat.begin() + (size_t)k * rows * cols, flat.begin() + (size_t)(k + 1) * rows * cols, ch[k].d.begin());
    }
    frame.reset(new DescriptorFrame(_frame_id, std::move(ch)));
    lap(5);
  } else {
    frame.reset(new DescriptorFrame(_frame_id, I_ptr, rows, cols, dtype, nt));
    lap(5);
    if (_frame_resident) {
      _frame_resident = false;                 // the pyramid class produced this level's frame on the device
    } else if (frame->numChannels() == 1) {
      check(_engine, pba_set_frame_u8(_engine, slot, I_ptr), "pba_set_frame_u8");
    } else {
      const int nc = (int)frame->numChannels();
      std::vector<float> flat((size_t)nc * rows * cols);
      for (int k = 0; k < nc; ++k)
        std::copy(frame->channels[k].d.begin(), frame->channels[k].d.end(), flat.begin() + (size_t)k * rows * cols);
      check(_engine, pba_set_frame_channels_f32(_engine, slot, nc, flat.data()), "pba_set_frame_channels_f32");
    }
  }

  const int num_channels = (int)frame->numChannels();
  const int B = std::max(_options_ptr->maskBlockRadius, std::max(2, _options_ptr->patchRadius));
  const int max_rows = rows - B - 1, max_cols = cols - B - 1;
  const int radius = _options_ptr->patchRadius, patch_length = PatchSizeFromRadius(radius);
  const int mask_radius = _options_ptr->maskBlockRadius;

  lap(0);
  // ---- visibility list update (reference :505-542) ------------------------------------------------------------
  int num_updated = 0, max_num_to_update = 0;
  double t_vis_par = 0.0; int n_vis = 0;
  if (!host_frontend) {
    // device: the host only projects the tracked points (double arithmetic of :519-523) and ships the ones inside the border
    // with their stored patches; the ZNCC test and the mask live on the device (pba_frontend_visibility)
    const int n_sp = (int)_scene_points.size();
    std::vector<int> idx;
    std::vector<double> uvs;
    std::vector<int32_t> rcs;
    std::vector<float> pats;
    for (int k = 0; k < n_sp; ++k) {
      const ScenePoint& pt = *_scene_points[k];
      const int f_dist = (int)_frame_id - (int)pt.lastFrameId();
      if (f_dist > _options_ptr->maxFrameDistance) continue;
      ++max_num_to_update;
      const Vec2 uv = _calib.project(TransformPoint(T_c, pt.X));
      const int r = (int)std::round(uv[1]), c = (int)std::round(uv[0]);
      if (r >= B && r < max_rows && c >= B && c <= max_cols) {
        idx.push_back(k);
        uvs.push_back(uv[0]); uvs.push_back(uv[1]);
        rcs.push_back(r); rcs.push_back(c);
        static_assert(sizeof(ZnccPatch) == 26 * sizeof(float), "patch record = 25 values + norm");
        const float* pd = reinterpret_cast<const float*>(&pt.patch);
        pats.insert(pats.end(), pd, pd + 26);
      }
    }
    std::vector<uint8_t> hit(idx.size());
    check(_engine, pba_frontend_visibility(_engine, (int32_t)idx.size(), uvs.data(), rcs.data(), pats.data(), _options_ptr->minScore, mask_radius,
                                           hit.data()), "pba_frontend_visibility");
    for (size_t q = 0; q < idx.size(); ++q)
      if (hit[q]) { ++num_updated; _scene_points[idx[q]]->f.push_back(_frame_id); }
    n_vis = n_sp;
  } else {
  std::fill(_mask.d.begin(), _mask.d.end(), (uint16_t)1);
  {
    // the ZNCC test of every tracked point is independent: evaluated on all host threads, applied in list order
    const int n_sp = (int)_scene_points.size();
    std::vector<int> hit_r(n_sp, -1), hit_c(n_sp, 0);
    std::vector<char> tried(n_sp, 0);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (int k = 0; k < n_sp; ++k) {
      const ScenePoint& pt = *_scene_points[k];
      const int f_dist = (int)_frame_id - (int)pt.lastFrameId();
      if (f_dist > _options_ptr->maxFrameDistance) continue;
      tried[k] = 1;
      const Vec2 uv = _calib.project(TransformPoint(T_c, pt.X));
      const int r = (int)std::round(uv[1]), c = (int)std::round(uv[0]);
      if (r >= B && r < max_rows && c >= B && c <= max_cols) {
        ZnccPatch other;
        other.set(I, uv[0], uv[1]);
        if (pt.patch.score(other) > _options_ptr->minScore) { hit_r[k] = r; hit_c[k] = c; }
      }
    }
    t_vis_par = wall_ms() - t_last;
    n_vis = n_sp;
    for (int k = 0; k < n_sp; ++k) {
      max_num_to_update += tried[k];
      if (hit_r[k] < 0) continue;
      ++num_updated;
      _scene_points[k]->f.push_back(_frame_id);
      for (int r_i = -mask_radius; r_i <= mask_radius; ++r_i)
        for (int c_i = -mask_radius; c_i <= mask_radius; ++c_i) _mask(hit_r[k] + r_i, hit_c[k] + c_i) = 0;
    }
  }

  }

  lap(1);
  // ---- new scene points (reference :545-585): valid depth AND strict local maximum of the saliency under the mask --
  ScenePointPointerList new_points;
  if (host_frontend) frame->computeSaliencyMap(_saliency_map, nt);
  lap(2);
  const int nms = _options_ptr->nonMaxSuppRadius;
  auto is_local_max = [&](int row, int col) {
    if (nms > 0) {
      const float v = _saliency_map(row, col);
      if (!_mask(row, col) || v < 0.0f) return false;
      for (int r = -nms; r <= nms; ++r)
        for (int c = -nms; c <= nms; ++c)
          if (!(!r && !c) && _saliency_map(r + row, c + col) >= v) return false;
    }
    return true;
  };
  // The reference builds a ScenePoint for every candidate and cuts to the maxNumPoints most salient ones afterwards
  // (:545-585, :587-595).  Same selection here on flat (saliency, x, y) records: nth_element sees the same sequence of
  // comparison outcomes as on the pointer list, so the same candidates survive in the same order -- but only those
  // get their scene point, ZNCC patch and descriptor built.
  struct Candidate { float saliency; int x, y; };
  std::vector<Candidate> cands;
  if (!host_frontend) {
    // device: saliency, depth test, mask, strict local maximum; the list comes back in the row-major order of the scan below
    static_assert(sizeof(Candidate) == sizeof(pba_candidate), "candidate record");
    int32_t n_c = 0;
    check(_engine, pba_frontend_candidates(_engine, slot, Z_ptr, _options_ptr->minValidDepth, _options_ptr->maxValidDepth, nms, B, &n_c),
          "pba_frontend_candidates");
    cands.resize((size_t)n_c);
    check(_engine, pba_frontend_get_candidates(_engine, reinterpret_cast<pba_candidate*>(cands.data()), n_c), "pba_frontend_get_candidates");
  } else {
    std::vector<std::vector<Candidate>> per_row(max_rows > B ? max_rows - B : 0);
#pragma omp parallel for schedule(dynamic, 8) num_threads(nt)
    for (int y = B; y < max_rows; ++y) {
      std::vector<Candidate>& row = per_row[y - B];
      for (int x = B; x < max_cols; ++x) {
        const float z = Z_ptr[(size_t)y * cols + x];
        if (z >= _options_ptr->minValidDepth && z <= _options_ptr->maxValidDepth && is_local_max(y, x))
          row.push_back(Candidate{_saliency_map(y, x), x, y});
      }
    }
    size_t total = 0;
    for (const auto& row : per_row) total += row.size();
    cands.reserve(total);
    for (const auto& row : per_row) cands.insert(cands.end(), row.begin(), row.end());   // row-major, like the scan
  }
  lap(3);
  if (cands.size() > (size_t)_options_ptr->maxNumPoints) {
    auto nth = cands.begin() + _options_ptr->maxNumPoints;
    std::nth_element(cands.begin(), nth, cands.end(), [](const Candidate& a, const Candidate& b) { return a.saliency > b.saliency; });
    cands.erase(nth, cands.end());
  }
  new_points.resize(cands.size());
#pragma omp parallel for schedule(static) num_threads(nt)
  for (int k = 0; k < (int)cands.size(); ++k) {
    const int x = cands[k].x, y = cands[k].y;
    const float z = Z_ptr[(size_t)y * cols + x];
    const Vec3 ray = _K_inv * MakeVec3((double)x, (double)y, 1.0);
    const Vec3 X = TransformPoint(T_w, MakeVec3((double)z * ray[0], (double)z * ray[1], (double)z * ray[2]));
    ScenePointPointer p(new ScenePoint(X, _frame_id));
    p->patch.set(I, (double)x, (double)y);
    p->descriptor.resize((size_t)num_channels * patch_length);
    p->saliency = cands[k].saliency;
    p->x0 = x; p->y0 = y;
    new_points[k] = std::move(p);
  }
  std::fprintf(stderr, "updated %d [%0.2f%%] max %d new %d\n", num_updated,
               _scene_points.empty() ? 0.0 : 100.0 * num_updated / _scene_points.size(), max_num_to_update, (int)new_points.size());

  // ---- descriptors (reference :466-479, :597-603): integer-pixel patch, indices clamped ------------------------
  if (!host_frontend) {
    const int n_new = (int)new_points.size();
    std::vector<int32_t> xy((size_t)2 * n_new);
    for (int k = 0; k < n_new; ++k) { xy[2 * k] = new_points[k]->x0; xy[2 * k + 1] = new_points[k]->y0; }
    std::vector<float> dv((size_t)n_new * num_channels * patch_length);
    check(_engine, pba_frontend_descriptors(_engine, slot, n_new, xy.data(), dv.data()), "pba_frontend_descriptors");
    const size_t per = (size_t)num_channels * patch_length;
    for (int k = 0; k < n_new; ++k)
      for (size_t i = 0; i < per; ++i) new_points[k]->descriptor[i] = (double)dv[(size_t)k * per + i];
  } else {
    const int mc = cols - radius - 1, mr = rows - radius - 1;
    for (int k = 0; k < num_channels; ++k) {
      const Image_<float>& channel = frame->channels[k];
      for (auto& p : new_points) {
        int i = k * patch_length;
        for (int r = -radius; r <= radius; ++r) {
          const int r_i = std::max(radius, std::min(p->y0 + r, mr));
          for (int c = -radius; c <= radius; ++c, ++i) {
            const int c_i = std::max(radius, std::min(p->x0 + c, mc));
            p->descriptor[i] = (double)channel(r_i, c_i);
          }
        }
      }
    }
  }
  for (auto& p : new_points) _scene_points.push_back(std::move(p));

  if ((int)_frame_buffer.size() == window) _frame_buffer.erase(_frame_buffer.begin());
  _frame_buffer.push_back(std::move(frame));
  lap(4);
  const double t_front = wall_ms();
  if ((int)_frame_buffer.size() == window) optimize(result);
  if (_options_ptr->verbose)
    std::fprintf(stderr, "addFrame %.2f ms (front-end %.2f ms, optimize %.2f ms)  [frame+upload %.2f, visibility %.2f, saliency %.2f, "
                 "candidates %.2f, top-N+descriptors %.2f]  (float plane %.2f, visibility parallel part %.2f of %d points, %d tried)\n", wall_ms() - t_enter, t_front - t_enter, wall_ms() - t_front,
                 t_ph[0] + t_ph[5], t_ph[1], t_ph[2], t_ph[3], t_ph[4], t_ph[5], t_vis_par, n_vis, max_num_to_update);
  ++_frame_id;
}

void PhotometricBundleAdjustment::optimize(Result* result) {
  double t_o[6] = {0, 0, 0, 0, 0, 0};
  double t_lo = wall_ms();
  auto lap_o = [&](int k) { const double t = wall_ms(); t_o[k] += t - t_lo; t_lo = t; };
  const uint32_t frame_id_start = _frame_buffer.front()->id, frame_id_end = _frame_buffer.back()->id;
  const int window = _options_ptr->slidingWindowSize;
  const std::vector<double> patch_weights = MakePatchWeights(_options_ptr->patchRadius, _options_ptr->doGaussianWeighting);
  const int P = (int)patch_weights.size() * (int)_frame_buffer.front()->numChannels();   // descriptor entries per point

  // cameras: INVERTED world poses as angle-axis + t (reference :774-778), stored by ring slot id % window
  std::vector<double> cams(6 * (size_t)window, 0.0);
  for (uint32_t id = frame_id_start; id <= frame_id_end; ++id)
    PoseToParams(_trajectory.atId((int)id).inverse(), &cams[6 * (id % window)]);

  // points with >= 3 observations whose reference frame is inside the window; one block per visible frame (:786-806)
  std::vector<ScenePoint*> selected;
  std::vector<double> xyz, desc;
  std::vector<int32_t> obs_point, obs_slot;
  selected.reserve(_scene_points.size());
  xyz.reserve(3 * _scene_points.size());
  desc.reserve((size_t)P * _scene_points.size());
  obs_point.reserve((size_t)window * _scene_points.size());
  obs_slot.reserve((size_t)window * _scene_points.size());
  for (auto& pt : _scene_points) {
    if (pt->numFrames() >= 3 && pt->refFrameId() >= frame_id_start) {
      int32_t slots[PBA_MAX_FRAMES];
      int n_slots = 0;
      for (uint32_t id : pt->f)
        if (id >= frame_id_start && id <= frame_id_end && n_slots < PBA_MAX_FRAMES) slots[n_slots++] = (int32_t)(id % window);
      if (n_slots == 0) continue;
      std::sort(slots, slots + n_slots);
      pt->was_refined = true;
      const int32_t idx = (int32_t)selected.size();
      selected.push_back(pt.get());
      for (int k = 0; k < 3; ++k) xyz.push_back(pt->X[k]);
      desc.insert(desc.end(), pt->descriptor.begin(), pt->descriptor.end());
      for (int k = 0; k < n_slots; ++k) { obs_point.push_back(idx); obs_slot.push_back(slots[k]); }
    }
  }
  std::fprintf(stderr, "Using %d points (%d residual blocks) [id start %d]\n", (int)selected.size(), (int)obs_point.size(), (int)frame_id_start);

  // Test hook (PBA_DUMP_WINDOWS=<dir>): the problem exactly as it is handed to the engine, so that a test can solve the
  // SAME window independently on the CPU (tests/test_gpu_configs0.py).  Not part of the reference.
  if (const char* dump_dir = std::getenv("PBA_DUMP_WINDOWS")) {
    char fn[1024];
    // (PBA_DUMP_TAG_SIZE: the image size joins the name -- the levels of a pyramid are objects of this class with the same frame ids)
    if (std::getenv("PBA_DUMP_TAG_SIZE")) std::snprintf(fn, sizeof(fn), "%s/window_%06u_%dx%d.bin", dump_dir, (unsigned)frame_id_end, _image_size.cols, _image_size.rows);
    else std::snprintf(fn, sizeof(fn), "%s/window_%06u.bin", dump_dir, (unsigned)frame_id_end);
    if (std::FILE* f = std::fopen(fn, "wb")) {
      const Options& op = *_options_ptr;
      const int32_t hdr[12] = {window, (int32_t)selected.size(), (int32_t)obs_point.size(), P, op.patchRadius,
                               (int32_t)(frame_id_start % window), (int32_t)frame_id_start, (int32_t)frame_id_end,
                               op.maxNumPoints, (int32_t)op.descriptorType, op.doGaussianWeighting ? 1 : 0, (int32_t)patch_weights.size()};
      const double dopt[4] = {op.minScore, op.robustThreshold, op.minValidDepth, op.maxValidDepth};
      std::fwrite(hdr, sizeof(hdr), 1, f);
      std::fwrite(dopt, sizeof(dopt), 1, f);
      std::fwrite(cams.data(), sizeof(double), cams.size(), f);
      std::fwrite(xyz.data(), sizeof(double), xyz.size(), f);
      std::fwrite(desc.data(), sizeof(double), desc.size(), f);
      std::fwrite(obs_point.data(), sizeof(int32_t), obs_point.size(), f);
      std::fwrite(obs_slot.data(), sizeof(int32_t), obs_slot.size(), f);
      std::fwrite(patch_weights.data(), sizeof(double), patch_weights.size(), f);
      std::fclose(f);
    }
  }

  lap_o(0);
  pba_solver_summary summary;
  std::memset(&summary, 0, sizeof(summary));
  std::vector<pba_iteration_summary> its(512);
  if (!selected.empty()) {
    check(_engine, pba_set_problem(_engine, (int32_t)selected.size(), xyz.data(), desc.data(), (int32_t)obs_point.size(),
                                  obs_point.data(), obs_slot.data(), patch_weights.data()), "pba_set_problem");
    lap_o(1);
    // "set the first camera constant" only if it is part of the problem (reference :809-815, HasParameterBlock).  When it
    // is not, the reference only warns: no camera is constant and the first one is not a parameter block of the program.
    // The engine gets the first slot as its constant slot in BOTH cases: a constant camera without residual blocks is the
    // same problem as a camera that is absent from it (it has no columns in the reduced system, and the engine's
    // column-is-live flags keep it out of |x|), and the number of free cameras stays <= window - 1 -- with -1 here a
    // 16-frame window had 16 free cameras = 136 pair blocks, more than the Schur tile holds (PBA_ERR_INVALID).
    const int32_t first_slot = (int32_t)(frame_id_start % window);
    const bool first_in_bundle = std::find(obs_slot.begin(), obs_slot.end(), first_slot) != obs_slot.end();
    if (!first_in_bundle) std::fprintf(stderr, "first camera is not in bundle\n");
    check(_engine, pba_set_cameras(_engine, cams.data(), window, first_slot), "pba_set_cameras");
    lap_o(2);
    pba_solver_options so;
    pba_default_solver_options(&so);     // GetSolverOptions (:738-761)
    so.verbose = _options_ptr->verbose ? 1 : 0;
    check(_engine, pba_solve(_engine, &so, &summary, its.data(), (int32_t)its.size()), "pba_solve");
    lap_o(3);
    if (_options_ptr->verbose)
      std::printf("pba_solve: %s  initial %.6e  final %.6e  iterations %d (successful %d)  %.3f s\n", summary.message,
                  summary.initial_cost, summary.final_cost, summary.num_iterations, summary.num_successful_steps,
                  summary.total_time_in_seconds);
    check(_engine, pba_get_state(_engine, cams.data(), xyz.data()), "pba_get_state");
    for (size_t i = 0; i < selected.size(); ++i) for (int k = 0; k < 3; ++k) selected[i]->X[k] = xyz[3 * i + k];
    // put back the refined camera poses (:841-844)
    for (uint32_t id = frame_id_start; id <= frame_id_end; ++id)
      _trajectory.atId((int)id) = ParamsToPose(&cams[6 * (id % window)]).inverse();
  } else {
    std::fprintf(stderr, "first camera is not in bundle\n");   // empty problem: ceres::Solve would return at once
  }
  (void)P;

  lap_o(4);
  auto points_to_remove = removePointsAtFrame(frame_id_start);
  std::printf("removing %zu old points\n", points_to_remove.size());
  lap_o(5);
  if (_options_ptr->verbose)
    std::fprintf(stderr, "optimize phases ms: assemble %.2f, set_problem %.2f, set_cameras %.2f, solve %.2f, read-back %.2f, evict %.2f\n",
                 t_o[0], t_o[1], t_o[2], t_o[3], t_o[4], t_o[5]);

  if (result) {
    result->poses = _trajectory.poses();
    const size_t npts = points_to_remove.size();
    result->refinedPoints.resize(npts);
    result->originalPoints.resize(npts);
    for (size_t i = 0; i < npts; ++i) { result->refinedPoints[i] = points_to_remove[i]->X; result->originalPoints[i] = points_to_remove[i]->X_original; }
    result->initialCost = summary.initial_cost;
    result->finalCost = summary.final_cost;
    result->fixedCost = summary.fixed_cost;
    result->numSuccessfulStep = summary.num_successful_steps;
    result->totalTime = summary.total_time_in_seconds;
    result->numResiduals = summary.num_residuals;
    result->message = summary.message;
    result->iterationSummary.clear();
    for (int i = 0; i < summary.num_iterations; ++i) {
      const pba_iteration_summary& s = its[i];
      ceres::IterationSummary o;
      o.iteration = s.iteration; o.step_is_valid = s.step_is_valid; o.step_is_nonmonotonic = s.step_is_nonmonotonic;
      o.step_is_successful = s.step_is_successful; o.cost = s.cost; o.cost_change = s.cost_change;
      o.gradient_max_norm = s.gradient_max_norm; o.gradient_norm = s.gradient_norm; o.step_norm = s.step_norm;
      o.relative_decrease = s.relative_decrease; o.trust_region_radius = s.trust_region_radius; o.eta = s.eta;
      o.step_size = s.step_size; o.linear_solver_iterations = s.linear_solver_iterations;
      o.iteration_time_in_seconds = s.iteration_time_in_seconds; o.step_solver_time_in_seconds = s.step_solver_time_in_seconds;
      o.cumulative_time_in_seconds = s.cumulative_time_in_seconds;
      result->iterationSummary.push_back(o);
    }
  }
}

const PhotometricBundleAdjustment::DescriptorFrame* PhotometricBundleAdjustment::getFrameAtId(uint32_t id) const {
  for (const auto& f : _frame_buffer) if (f->id == id) return f.get();
  throw std::runtime_error("could not find frame id!");
}

// reference :888-905: everything whose reference frame is <= id leaves the system
PhotometricBundleAdjustment::ScenePointPointerList PhotometricBundleAdjustment::removePointsAtFrame(uint32_t id) {
  ScenePointPointerList keep, remove;
  for (auto& p : _scene_points) (p->refFrameId() <= id ? remove : keep).push_back(std::move(p));
  _scene_points.swap(keep);
  return remove;
}
