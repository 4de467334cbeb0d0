// photobundle.h -- the reference's public class, MI355X engine underneath.
//
// Same class name, nested types, field names, defaults and call signatures as reference src/photobundle.h:19-196, so
// that apps/run_kitti.cc-style callers compile against it unchanged:
//
//     PhotometricBundleAdjustment photoba(calib, image_size, {cf});
//     photoba.addFrame(I, Z, T_init[f_i], &result);
//
// What changed underneath: the Ceres problem / ceres::Solve block of optimize() (reference src/photobundle.cc:784-829)
// is replaced by the C-ABI engine of include/pba.h (hand-written HIP for gfx950); Eigen / boost / ceres types in the
// header are replaced by the stand-ins of types.h and the ceres::IterationSummary struct below.
#ifndef PHOTOBUNDLE_AMD_PHOTOBUNDLE_H
#define PHOTOBUNDLE_AMD_PHOTOBUNDLE_H

#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

#include "calibration.h"
#include "trajectory.h"
#include "types.h"

namespace utils { class ConfigFile; }

namespace ceres {
// The fields the reference reads / serialises (reference src/ceres_cereal.h:13-30), same names.
struct IterationSummary {
  int iteration = 0;
  bool step_is_valid = false;
  bool step_is_nonmonotonic = false;
  bool step_is_successful = false;
  double cost = 0.0;
  double cost_change = 0.0;
  double gradient_max_norm = 0.0;
  double gradient_norm = 0.0;
  double step_norm = 0.0;
  double relative_decrease = 0.0;
  double trust_region_radius = 0.0;
  double eta = 0.0;
  double step_size = 0.0;
  int line_search_function_evaluations = 0;
  int line_search_gradient_evaluations = 0;
  int line_search_iterations = 0;
  int linear_solver_iterations = 0;
  double iteration_time_in_seconds = 0.0;
  double step_solver_time_in_seconds = 0.0;
  double cumulative_time_in_seconds = 0.0;
};
}  // namespace ceres

struct pba_engine;

class PhotometricBundleAdjustment {
 public:
  struct Options {
    int maxNumPoints = 4096;          // maximum number of points to initialise from a new frame
    int slidingWindowSize = 5;        // number of frames in the sliding window
    int patchRadius = 2;              // radius of the image patch
    int maskBlockRadius = 1;          // area blocked around re-observed points when selecting new ones
    int maxFrameDistance = 1;         // maximum age of a scene point
    int numThreads = -1;              // kept for source compatibility; the solve runs on the GPU
    bool doGaussianWeighting = false;
    bool verbose = true;
    double minScore = 0.75;           // ZNCC threshold of the visibility test
    double robustThreshold = 0.05;    // HuberLoss threshold (if > 0)
    double minValidDepth = 0.01;
    double maxValidDepth = 1000.0;
    int nonMaxSuppRadius = 1;
    enum class DescriptorType { Intensity, IntensityAndGradient, BitPlanes };
    DescriptorType descriptorType = DescriptorType::Intensity;   // (uninitialised in the reference, photobundle.h:77-79)
    int device = 0;                   // NEW: HIP device ordinal

    Options() {}
    Options(const utils::ConfigFile& cf);
  };

  struct Result {
    EigenAlignedContainer_<Mat44> poses;           // refined world poses (whole trajectory so far)
    EigenAlignedContainer_<Vec3> refinedPoints;    // points that left the window in this call
    EigenAlignedContainer_<Vec3> originalPoints;
    double initialCost = -1.0;
    double finalCost = -1.0;
    double fixedCost = -1.0;
    int numSuccessfulStep = 0;
    int numResiduals = 0;
    double totalTime = -1.0;
    std::string message;
    std::vector<ceres::IterationSummary> iterationSummary;

    // Dead code in the reference's default build (WITH_CEREAL is never defined, photobundle.cc:51-86): kept as stubs.
    struct Writer {
      Writer(std::string prefix = "./") : _counter(0), _prefix(prefix) {}
      bool add(const Result&);
     private:
      int _counter;
      std::string _prefix;
    };
    static Result FromFile(std::string);
  };

 public:
  PhotometricBundleAdjustment(const Calibration&, const ImageSize&, const Options& = Options());
  ~PhotometricBundleAdjustment();
  PhotometricBundleAdjustment(const PhotometricBundleAdjustment&) = delete;
  PhotometricBundleAdjustment& operator=(const PhotometricBundleAdjustment&) = delete;

  // image: dense row-major rows x cols u8; depth_map: dense row-major float; T: frame-to-frame pose initialisation;
  // result (optional) is overwritten whenever an optimisation ran (reference photobundle.h:154-160).
  void addFrame(const uint8_t* image, const float* depth_map, const Mat44& T, Result* = nullptr);

 protected:
  void optimize(Result*);

 private:
  struct ScenePoint;
  struct DescriptorFrame;
  typedef UniquePointer<ScenePoint> ScenePointPointer;
  typedef std::vector<ScenePointPointer> ScenePointPointerList;

  ScenePointPointerList removePointsAtFrame(uint32_t id);
  const DescriptorFrame* getFrameAtId(uint32_t id) const;

  uint32_t _frame_id = 0;
  Calibration _calib;
  ImageSize _image_size;
  Options _options;
  Trajectory _trajectory;
  std::vector<UniquePointer<DescriptorFrame>> _frame_buffer;   // ring of the last slidingWindowSize frames
  ScenePointPointerList _scene_points;
  Image_<uint16_t> _mask;
  Image_<float> _saliency_map;
  Mat33 _K_inv;
  pba_engine* _engine = nullptr;
};

#endif
