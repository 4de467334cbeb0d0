// trajectory.h -- world-pose list keyed by frame id (reference src/trajectory.h:7-80, src/trajectory.cc:7-59).
#ifndef PHOTOBUNDLE_AMD_TRAJECTORY_H
#define PHOTOBUNDLE_AMD_TRAJECTORY_H

#include <stdexcept>
#include <vector>

#include "types.h"

class Trajectory {
 public:
  typedef int Id_t;

  // `pose` is the relative (frame-to-frame) estimate; the stored world pose is T_w_i = T_w_(i-1) * inv(T_i)
  void push_back(const Mat44& pose, const Id_t id) {
    for (const auto& p : _data) if (p.id == id) throw std::runtime_error("duplicate id in trajectory\n");
    const Mat44 T_inv = pose.inverse();
    if (!_data.empty()) _data.push_back({back() * T_inv, id});
    else _data.push_back({T_inv, id});
  }
  const Mat44& operator[](size_t i) const { return _data[i].pose; }
  Mat44& operator[](size_t i) { return _data[i].pose; }
  const Mat44& atId(const Id_t id) const {
    for (const auto& p : _data) if (p.id == id) return p.pose;
    throw std::runtime_error("could not find pose with id");
  }
  Mat44& atId(const Id_t id) {
    for (auto& p : _data) if (p.id == id) return p.pose;
    throw std::runtime_error("could not find pose with id");
  }
  const Mat44& back() const { return _data.back().pose; }
  EigenAlignedContainer_<Mat44> poses() const {
    EigenAlignedContainer_<Mat44> ret(_data.size());
    for (size_t i = 0; i < ret.size(); ++i) ret[i] = _data[i].pose;
    return ret;
  }
  EigenAlignedContainer_<Vec3> cameraPositions() const {
    EigenAlignedContainer_<Vec3> ret(_data.size());
    for (size_t i = 0; i < ret.size(); ++i) ret[i] = MakeVec3(_data[i].pose(0, 3), _data[i].pose(1, 3), _data[i].pose(2, 3));
    return ret;
  }
  size_t size() const { return _data.size(); }

 private:
  struct PoseWithId { Mat44 pose; Id_t id; };
  std::vector<PoseWithId> _data;
};

#endif
