// calibration.h -- pinhole stereo calibration with the reference's interface (reference src/calibration.h:10-84).
#ifndef PHOTOBUNDLE_AMD_CALIBRATION_H
#define PHOTOBUNDLE_AMD_CALIBRATION_H

#include "types.h"

class Calibration {
 public:
  Calibration() : _K(Mat33::Identity()), _baseline(0.0) {}
  Calibration(const Mat33& K, double b) : _K(K), _baseline(b) {}

  const double& b() const { return _baseline; }
  const double& fx() const { return _K(0, 0); }
  const double& fy() const { return _K(1, 1); }
  const double& cx() const { return _K(0, 2); }
  const double& cy() const { return _K(1, 2); }
  const Mat33& K() const { return _K; }
  Mat33& K() { return _K; }
  double& baseline() { return _baseline; }

  // reference calibration.h:33-38
  template <typename T>
  void project(const T* X, T& u, T& v) const {
    u = ((X[0] * T(fx())) / X[2]) + T(cx());
    v = ((X[1] * T(fy())) / X[2]) + T(cy());
  }
  // reference calibration.h:43 + eigen.h normHomog: (1 / p[2]) * (K X).head<2>()
  Vec2 project(const Vec3& X) const {
    const Vec3 p = _K * X;
    const double s = 1.0 / p[2];
    Vec2 uv; uv[0] = s * p[0]; uv[1] = s * p[1];
    return uv;
  }
  void setParameters(const double* p) {
    _K = Mat33::Identity();
    _K(0, 0) = p[0]; _K(1, 1) = p[1]; _K(0, 2) = p[2]; _K(1, 2) = p[3];
  }
  // reference calibration.h:72-78: K * 0.5 (K(2,2) = 1), baseline * 2
  Calibration pyrDown() const {
    Mat33 K = 0.5 * _K;
    K(2, 2) = 1.0;
    return Calibration(K, _baseline * 2);
  }

 private:
  Mat33 _K;
  double _baseline;
};

#endif
