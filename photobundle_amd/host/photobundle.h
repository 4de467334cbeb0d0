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
class PhotometricBundleAdjustmentPyr;

// Same public surface as the reference class (nested Options / Result, addFrame, protected optimize); the private
// part is this implementation's own.
class PhotometricBundleAdjustment {
 public:
  struct Options;
  struct Result;

  PhotometricBundleAdjustment(const Calibration& calibration, const ImageSize& image_size, const Options& options);
  PhotometricBundleAdjustment(const Calibration& calibration, const ImageSize& image_size);
  ~PhotometricBundleAdjustment();
  PhotometricBundleAdjustment(const PhotometricBundleAdjustment&) = delete;
  PhotometricBundleAdjustment& operator=(const PhotometricBundleAdjustment&) = delete;

  // image: dense row-major rows x cols u8; depth_map: dense row-major float (<= 0 / out of [minValidDepth,
  // maxValidDepth] = invalid); T: frame-to-frame pose initialisation.  `result` (optional) is overwritten whenever
  // an optimisation ran, i.e. once the sliding window is full.
  void addFrame(const uint8_t* image, const float* depth_map, const Mat44& T, Result* result = nullptr);

  // ---- solver / front-end settings: field names, meanings and ConfigFile keys of the reference ----
  struct Options {
    // front-end
    int maxNumPoints = 4096;          // new scene points kept per frame (largest saliency first)
    int nonMaxSuppRadius = 1;         // saliency non-maximum suppression radius
    int maskBlockRadius = 1;          // pixels blocked around a re-observed point
    int maxFrameDistance = 1;         // a point not seen for more frames than this is no longer tracked
    double minScore = 0.75;           // ZNCC acceptance threshold, in [-1, 1]
    double minValidDepth = 0.01;
    double maxValidDepth = 1000.0;
    // problem shape
    int slidingWindowSize = 5;
    int patchRadius = 2;
    bool doGaussianWeighting = false;
    double robustThreshold = 0.05;    // Huber threshold; <= 0 disables the loss
    enum class DescriptorType { Intensity, IntensityAndGradient, BitPlanes };
    DescriptorType descriptorType = DescriptorType::Intensity;   // (left uninitialised by the reference)
    // execution
    int numThreads = -1;              // accepted for source compatibility; the solve runs on the GPU
    bool verbose = true;
    int device = 0;                   // HIP device ordinal (new)

    Options() {}
    Options(const utils::ConfigFile& cf);

   private:
    // declared by the reference (src/photobundle.h:81-82) and never defined there: prints the settings as ConfigFile lines
    // (key = value, the keys Options(const ConfigFile&) reads), so that the output of one run configures the next
    friend std::ostream& operator<<(std::ostream&, const Options&);
  };

  // ---- what an optimisation reports ----
  struct Result {
    EigenAlignedContainer_<Mat44> poses;           // refined world poses of the whole trajectory so far
    EigenAlignedContainer_<Vec3> refinedPoints;    // points that left the window in this call ...
    EigenAlignedContainer_<Vec3> originalPoints;   // ... and what they were initialised to
    double initialCost = -1.0;
    double finalCost = -1.0;
    double fixedCost = -1.0;
    int numSuccessfulStep = 0;
    int numResiduals = 0;
    double totalTime = -1.0;                        // seconds
    std::string message;
    std::vector<ceres::IterationSummary> iterationSummary;

    // cereal-based serialisation is dead code in the reference's default build; kept as stubs for the signatures
    struct Writer {
      explicit Writer(std::string prefix = "./") : _counter(0), _prefix(prefix) {}
      bool add(const Result&);
     private:
      int _counter;
      std::string _prefix;
    };
    static Result FromFile(std::string);
  };

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
  UniquePointer<Options> _options_ptr;
  Trajectory _trajectory;
  std::vector<UniquePointer<DescriptorFrame>> _frame_buffer;   // the last slidingWindowSize frames, oldest first
  ScenePointPointerList _scene_points;
  Image_<uint16_t> _mask;
  Image_<float> _saliency_map;
  Mat33 _K_inv;
  pba_engine* _engine = nullptr;
  // set by the pyramid class when it has already put the next frame into the engine's ring slot on the device
  // (pba_set_frame_pyr_down): addFrame() then skips its own upload
  bool _frame_resident = false;
  friend class PhotometricBundleAdjustmentPyr;
};

#endif
