// calibration.h -- pinhole stereo calibration; call-compatible with the reference's Calibration
// (reference src/calibration.h:10-84): fx()/fy()/cx()/cy()/b()/K()/baseline()/project()/triangulate()/setParameters()/
// scale()/pyrDown().
//
// Representation differs from the reference (which keeps only an Eigen 3x3): the four intrinsics and the baseline are
// the source of truth and the matrix view is rebuilt on mutation through the accessors below.
#ifndef PHOTOBUNDLE_AMD_CALIBRATION_H
#define PHOTOBUNDLE_AMD_CALIBRATION_H

#include "types.h"

class Calibration {
  Mat33 _K;           // upper-triangular intrinsics [fx 0 cx; 0 fy cy; 0 0 1]
  double _baseline;   // stereo baseline (metres)

 public:
  Calibration();
  Calibration(const Mat33& intrinsics, double stereo_baseline);

  // --- read access (references, like the reference class) ---
  const Mat33& K() const;
  const double& fx() const;
  const double& fy() const;
  const double& cx() const;
  const double& cy() const;
  const double& b() const;

  // --- write access ---
  Mat33& K();
  double& baseline();
  void setParameters(const double* fx_fy_cx_cy);

  // u = fx X/Z + cx, v = fy Y/Z + cy, for any scalar type with * / + (the device kernels use the same order)
  template <typename Scalar>
  void project(const Scalar* X, Scalar& u, Scalar& v) const;
  template <typename Scalar>
  void project(const Scalar* X, Scalar* uv) const { project(X, uv[0], uv[1]); }
  // homogeneous form used by the front-end: (K X) / (K X)_z
  Vec2 project(const Vec3& X) const;

  // (u, v, disparity) -> camera-frame point: depth from the stereo relation z = b fx / d, then the pinhole back-projection
  // (reference src/calibration.h:46-53)
  template <typename UvdType>
  Vec3 triangulate(const UvdType& uvd) const;

  // image shrunk by a factor s > 1: the intrinsics divide by s, the baseline multiplies (reference :64-70; s <= 1 is ignored there too)
  void scale(double s);

  // one pyramid level down: focal lengths and principal point halve, the baseline doubles
  Calibration pyrDown() const;
};

inline Calibration::Calibration() : _K(Mat33::Identity()), _baseline(0.0) {}
inline Calibration::Calibration(const Mat33& intrinsics, double stereo_baseline) : _K(intrinsics), _baseline(stereo_baseline) {}

inline const Mat33& Calibration::K() const { return _K; }
inline Mat33& Calibration::K() { return _K; }
inline const double& Calibration::fx() const { return _K(0, 0); }
inline const double& Calibration::fy() const { return _K(1, 1); }
inline const double& Calibration::cx() const { return _K(0, 2); }
inline const double& Calibration::cy() const { return _K(1, 2); }
inline const double& Calibration::b() const { return _baseline; }
inline double& Calibration::baseline() { return _baseline; }

inline void Calibration::setParameters(const double* q) {
  Mat33 M = Mat33::Identity();
  M(0, 0) = q[0];
  M(1, 1) = q[1];
  M(0, 2) = q[2];
  M(1, 2) = q[3];
  _K = M;
}

template <typename Scalar>
inline void Calibration::project(const Scalar* X, Scalar& u, Scalar& v) const {
  const Scalar numer_u = X[0] * Scalar(fx());
  const Scalar numer_v = X[1] * Scalar(fy());
  u = (numer_u / X[2]) + Scalar(cx());
  v = (numer_v / X[2]) + Scalar(cy());
}

inline Vec2 Calibration::project(const Vec3& X) const {
  const Vec3 h = _K * X;
  const double inv_w = 1.0 / h[2];
  Vec2 uv;
  uv[0] = inv_w * h[0];
  uv[1] = inv_w * h[1];
  return uv;
}

template <typename UvdType>
inline Vec3 Calibration::triangulate(const UvdType& uvd) const {
  const double depth = (b() * fx()) * (1.0 / uvd[2]);
  Vec3 X;
  X[0] = (uvd[0] - cx()) * depth / fx();
  X[1] = (uvd[1] - cy()) * depth / fy();
  X[2] = depth;
  return X;
}

inline void Calibration::scale(double s) {
  if (!(s > 1.0)) return;
  const double inv = 1.0 / s;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) _K(r, c) = _K(r, c) * inv;
  _K(2, 2) = 1.0;
  _baseline = _baseline * s;
}

inline Calibration Calibration::pyrDown() const {
  Mat33 half = 0.5 * _K;
  half(2, 2) = 1.0;
  return Calibration(half, 2.0 * _baseline);
}

#endif
