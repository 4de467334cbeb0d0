// pose_utils.h -- KITTI pose text format: 12 numbers per line = row-major 3x4 (reference src/pose_utils.cc:9-59).
#ifndef PHOTOBUNDLE_AMD_POSE_UTILS_H
#define PHOTOBUNDLE_AMD_POSE_UTILS_H

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

#include "types.h"

inline EigenAlignedContainer_<Mat44> loadPosesKittiFormat(const std::string& filename) {
  std::ifstream ifs(filename);
  if (!ifs.is_open()) throw std::runtime_error("could not open " + filename);
  EigenAlignedContainer_<Mat44> ret;
  std::string line;
  while (std::getline(ifs, line)) {
    std::istringstream iss(line);
    Mat44 T = Mat44::Identity();
    bool ok = true;
    for (int r = 0; r < 3 && ok; ++r) for (int c = 0; c < 4; ++c) if (!(iss >> T(r, c))) { ok = false; break; }
    if (ok) ret.push_back(T);
  }
  return ret;
}

// Byte-compatible with the reference writer (src/pose_utils.cc:43-59): `ofs << T(r, c) << " "` = default ostream
// formatting of a double (%g, 6 significant digits), a blank after EVERY number including the last, then '\n'.
inline bool writePosesKittiFormat(const std::string& filename, const EigenAlignedContainer_<Mat44>& poses) {
  FILE* fp = fopen(filename.c_str(), "w");
  if (!fp) return false;
  for (const auto& T : poses) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) fprintf(fp, "%g ", T(r, c));
    fputc('\n', fp);
  }
  fclose(fp);
  return true;
}

// Same layout with round-trip precision (%.17g, no trailing blank): not the reference's format; run_kitti -p.
inline bool writePosesKittiFormatFullPrecision(const std::string& filename, const EigenAlignedContainer_<Mat44>& poses) {
  FILE* fp = fopen(filename.c_str(), "w");
  if (!fp) return false;
  for (const auto& T : poses) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) fprintf(fp, "%0.17g%s", T(r, c), (r == 2 && c == 3) ? "\n" : " ");
  }
  fclose(fp);
  return true;
}

// T_w list -> frame-to-frame local poses: T_i = inv(T_w_i) * T_w_(i-1)   (reference pose_utils.cc:62-74)
inline EigenAlignedContainer_<Mat44> convertPoseToLocal(const EigenAlignedContainer_<Mat44>& poses) {
  EigenAlignedContainer_<Mat44> ret(poses.size());
  if (poses.empty()) return ret;
  ret[0] = poses[0].inverse();
  for (size_t i = 1; i < poses.size(); ++i) ret[i] = poses[i].inverse() * poses[i - 1];
  return ret;
}

#endif
