// run_kitti.cc -- the reference's driver loop (reference apps/run_kitti.cc:17-59) on the MI355X engine.
//
// OpenCV / the stereo matcher are outside the hot path and absent from this image, so frames come from a directory
// of precomputed inputs instead of Dataset::Create:
//     <data>/image_%06d.pgm   binary PGM (P5, 8 bit)
//     <data>/depth_%06d.bin   rows*cols float32 depth map (what disparityToDepth would hand over; <= 0 invalid)
//     <data>/calib.txt        fx fy cx cy baseline
// The config file is the reference's (config/kitti_stereo.cfg keys) plus `DataDirectory`; `Trajectory` is the KITTI
// pose text file of frame-to-frame initial poses (reference data/kitti_init_poor/*.txt).
#include <csignal>
#include <cstdio>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "../host/photobundle.h"
#include "../host/photobundle_pyramid.h"
#include "../host/pose_utils.h"
#include "../host/utils.h"

static volatile bool gStop = false;
static void sigHandler(int) { gStop = true; }

static bool readPgm(const std::string& fn, std::vector<uint8_t>& img, int& rows, int& cols) {
  std::ifstream ifs(fn, std::ios::binary);
  if (!ifs.is_open()) return false;
  std::string magic;
  int maxval = 0;
  ifs >> magic;
  if (magic != "P5") return false;
  auto skip = [&]() { while (ifs.peek() == '#' || std::isspace(ifs.peek())) { if (ifs.peek() == '#') { std::string l; std::getline(ifs, l); } else ifs.get(); } };
  skip(); ifs >> cols; skip(); ifs >> rows; skip(); ifs >> maxval;
  ifs.get();
  if (maxval != 255) return false;
  img.resize((size_t)rows * cols);
  ifs.read(reinterpret_cast<char*>(img.data()), img.size());
  return (bool)ifs;
}

static void dumpResult(const std::string& fn, int frame, const PhotometricBundleAdjustment::Result& r) {
  std::FILE* f = std::fopen(fn.c_str(), "a");
  if (!f) throw std::runtime_error("cannot open " + fn);
  std::fprintf(f, "result frame %d poses %zu points %zu iterations %zu\n", frame, r.poses.size(), r.refinedPoints.size(), r.iterationSummary.size());
  std::fprintf(f, "cost %.17g %.17g %.17g steps %d residuals %d\n", r.initialCost, r.finalCost, r.fixedCost, r.numSuccessfulStep, r.numResiduals);
  std::fprintf(f, "message %s\n", r.message.c_str());
  for (const auto& it : r.iterationSummary)
    std::fprintf(f, "it %d %d %d %.17g %.17g %.17g %.17g %.17g %.17g\n", it.iteration, (int)it.step_is_valid, (int)it.step_is_successful, it.cost,
                 it.cost_change, it.gradient_max_norm, it.step_norm, it.relative_decrease, it.trust_region_radius);
  // every pose of the trajectory so far, round-trip precision (photobundle.cc:858: Result::poses covers all frames)
  for (size_t i = 0; i < r.poses.size(); ++i) {
    std::fprintf(f, "pose %zu", i);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 4; ++b) std::fprintf(f, " %.17g", r.poses[i](a, b));
    std::fprintf(f, "\n");
  }
  for (size_t i = 0; i < r.refinedPoints.size(); ++i)
    std::fprintf(f, "pt %.17g %.17g %.17g %.17g %.17g %.17g\n", r.refinedPoints[i][0], r.refinedPoints[i][1], r.refinedPoints[i][2],
                 r.originalPoints[i][0], r.originalPoints[i][1], r.originalPoints[i][2]);
  std::fclose(f);
}

int main(int argc, char** argv) {
  signal(SIGINT, sigHandler);
  // -r (not in the reference's driver): text dump of every Result the class hands back (reference photobundle.cc:857-875)
  // -p (not in the reference's driver): poses with round-trip precision instead of the reference's 6 significant digits
  std::string config = "../config/kitti_stereo.cfg", output = "refined_poses.txt", results;
  bool full_precision = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if ((a == "-c" || a == "--config") && i + 1 < argc) config = argv[++i];
    else if ((a == "-o" || a == "--output") && i + 1 < argc) output = argv[++i];
    else if ((a == "-r" || a == "--results") && i + 1 < argc) results = argv[++i];
    else if (a == "-p" || a == "--full-precision") full_precision = true;
    else { std::fprintf(stderr, "usage: %s [-c config] [-o output] [-r result-dump] [-p]\n", argv[0]); return 1; }
  }
  try {
    utils::ConfigFile cf(config);
    const std::string data = cf.get<std::string>("DataDirectory");
    double c5[5];
    {
      std::ifstream ifs(data + "/calib.txt");
      if (!(ifs >> c5[0] >> c5[1] >> c5[2] >> c5[3] >> c5[4])) throw std::runtime_error("bad calib.txt");
    }
    Calibration calib;
    calib.setParameters(c5);
    calib.baseline() = c5[4];
    const auto T_init = loadPosesKittiFormat(cf.get<std::string>("trajectory"));

    std::vector<uint8_t> img;
    std::vector<float> depth;
    int rows = 0, cols = 0;
    char name[64];
    std::snprintf(name, sizeof(name), "/image_%06d.pgm", 0);
    if (!readPgm(data + name, img, rows, cols)) throw std::runtime_error("cannot read the first frame");

    PhotometricBundleAdjustment::Result result;
    const int num_levels = cf.get<int>("numLevels", 1);   // > 1: coarse-to-fine (photobundle_pyramid path)
    std::unique_ptr<PhotometricBundleAdjustment> photoba;
    std::unique_ptr<PhotometricBundleAdjustmentPyr> photoba_pyr;
    if (num_levels > 1) photoba_pyr.reset(new PhotometricBundleAdjustmentPyr(num_levels, calib, ImageSize(rows, cols), {cf}));
    else photoba.reset(new PhotometricBundleAdjustment(calib, ImageSize(rows, cols), {cf}));
    for (int f_i = 0; f_i < (int)T_init.size() && !gStop; ++f_i) {
      std::snprintf(name, sizeof(name), "/image_%06d.pgm", f_i);
      int r2, c2;
      if (!readPgm(data + name, img, r2, c2)) break;
      if (r2 != rows || c2 != cols) throw std::runtime_error("frame size changed");
      std::snprintf(name, sizeof(name), "/depth_%06d.bin", f_i);
      depth.resize((size_t)rows * cols);
      std::ifstream dfs(data + name, std::ios::binary);
      if (!dfs.read(reinterpret_cast<char*>(depth.data()), depth.size() * sizeof(float))) throw std::runtime_error("bad depth file");
      std::printf("Frame %05d\n", f_i);
      result.initialCost = -1.0;    // the class only touches `result` when an optimisation ran (photobundle.cc:857)
      if (photoba_pyr) photoba_pyr->addFrame(img.data(), depth.data(), T_init[f_i], &result);
      else photoba->addFrame(img.data(), depth.data(), T_init[f_i], &result);
      if (!results.empty() && result.initialCost >= 0.0) dumpResult(results, f_i, result);
    }
    std::fprintf(stderr, "Writing refined poses to %s\n", output.c_str());
    if (full_precision) writePosesKittiFormatFullPrecision(output, result.poses);
    else writePosesKittiFormat(output, result.poses);   // reference format (src/pose_utils.cc:43-59)
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "error: %s\n", ex.what());
    return 1;
  }
  return 0;
}
