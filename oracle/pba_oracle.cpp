/*
 * pba_oracle.cpp -- CPU ORACLE (test infrastructure, see pba_oracle.h).
 *
 * Restates, function by function, the reference hot path and the Ceres 1.x
 * machinery it calls.  Citations are `path:line` relative to /root/reference
 * (or name the Ceres component when the code lives in the absent dependency).
 *
 * Built with -ffp-contract=off: the reference is compiled with -msse4.1 only
 * (CMakeLists.txt:20-23), i.e. without FMA, so no product/sum is ever fused.
 */
#include "pba_oracle.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// ceres::Jet<double, 9> restatement (ceres/jet.h): value + 9 partials, same operation order.
// ---------------------------------------------------------------------------------------------
constexpr int kN = 9;

struct Dual {
  double a;
  double v[kN];
  Dual() : a(0.0) { for (int i = 0; i < kN; ++i) v[i] = 0.0; }
  explicit Dual(double s) : a(s) { for (int i = 0; i < kN; ++i) v[i] = 0.0; }
  Dual(double s, int k) : a(s) { for (int i = 0; i < kN; ++i) v[i] = 0.0; v[k] = 1.0; }
};

inline Dual operator+(const Dual& f, const Dual& g) { Dual h; h.a = f.a + g.a; for (int i = 0; i < kN; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
inline Dual operator-(const Dual& f, const Dual& g) { Dual h; h.a = f.a - g.a; for (int i = 0; i < kN; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
inline Dual operator-(const Dual& f) { Dual h; h.a = -f.a; for (int i = 0; i < kN; ++i) h.v[i] = -f.v[i]; return h; }
inline Dual operator*(const Dual& f, const Dual& g) { Dual h; h.a = f.a * g.a; for (int i = 0; i < kN; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
inline Dual operator/(const Dual& f, const Dual& g) {
  // jet.h: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; v = (f.v - f_a_by_g_a * g.v) * g_a_inverse
  Dual h;
  const double gi = 1.0 / g.a;
  const double q = f.a * gi;
  h.a = q;
  for (int i = 0; i < kN; ++i) h.v[i] = (f.v[i] - q * g.v[i]) * gi;
  return h;
}
inline Dual operator+(const Dual& f, double s) { Dual h = f; h.a = f.a + s; return h; }
inline Dual operator+(double s, const Dual& f) { Dual h = f; h.a = f.a + s; return h; }
inline Dual operator-(const Dual& f, double s) { Dual h = f; h.a = f.a - s; return h; }
inline Dual operator-(double s, const Dual& f) { Dual h; h.a = s - f.a; for (int i = 0; i < kN; ++i) h.v[i] = -f.v[i]; return h; }
inline Dual operator*(const Dual& f, double s) { Dual h; h.a = f.a * s; for (int i = 0; i < kN; ++i) h.v[i] = f.v[i] * s; return h; }
inline Dual operator*(double s, const Dual& f) { return f * s; }
inline Dual operator/(const Dual& f, double s) { Dual h; const double si = 1.0 / s; h.a = f.a * si; for (int i = 0; i < kN; ++i) h.v[i] = f.v[i] * si; return h; }
inline Dual operator/(double s, const Dual& g) { Dual h; const double m = -s / (g.a * g.a); h.a = s / g.a; for (int i = 0; i < kN; ++i) h.v[i] = g.v[i] * m; return h; }
inline bool operator>(const Dual& f, const Dual& g) { return f.a > g.a; }

inline double Sqrt(double x) { return std::sqrt(x); }
inline double Sin(double x) { return std::sin(x); }
inline double Cos(double x) { return std::cos(x); }
inline Dual Sqrt(const Dual& f) { Dual h; const double t = std::sqrt(f.a); const double k = 1.0 / (2.0 * t); h.a = t; for (int i = 0; i < kN; ++i) h.v[i] = f.v[i] * k; return h; }
inline Dual Sin(const Dual& f) { Dual h; const double s = std::sin(f.a), c = std::cos(f.a); h.a = s; for (int i = 0; i < kN; ++i) h.v[i] = c * f.v[i]; return h; }
inline Dual Cos(const Dual& f) { Dual h; const double s = std::sin(f.a), c = std::cos(f.a); h.a = c; for (int i = 0; i < kN; ++i) h.v[i] = -s * f.v[i]; return h; }

inline double GetScalar(double x) { return x; }          // jet_extras.h:47
inline double GetScalar(const Dual& x) { return x.a; }   // jet_extras.h:63

// ---------------------------------------------------------------------------------------------
// Ceres rotation.h restatements.
// ---------------------------------------------------------------------------------------------
template <class T>
inline void AngleAxisRotatePointT(const T aa[3], const T pt[3], T result[3]) {
  // ceres::AngleAxisRotatePoint (call site photobundle.cc:700)
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = Sqrt(theta2);
    const T costheta = Cos(theta);
    const T sintheta = Sin(theta);
    const T theta_inverse = T(1.0) / theta;
    const T w[3] = {aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse};
    const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - costheta);
    result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
  } else {
    const T w_cross_pt[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
    result[0] = pt[0] + w_cross_pt[0];
    result[1] = pt[1] + w_cross_pt[1];
    result[2] = pt[2] + w_cross_pt[2];
  }
}

// column-major 3x3 accessor: R(i,j) = m[i + 3*j]  (ColumnMajorAdapter3x3)
inline double& CM(double* m, int i, int j) { return m[i + 3 * j]; }
inline double CMc(const double* m, int i, int j) { return m[i + 3 * j]; }

void AngleAxisToRotationMatrix(const double aa[3], double* R) {
  // ceres::AngleAxisToRotationMatrix (call site photobundle.cc:661)
  static const double kOne = 1.0;
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double costheta = std::cos(theta), sintheta = std::sin(theta);
    CM(R, 0, 0) = costheta + wx * wx * (kOne - costheta);
    CM(R, 1, 0) = wz * sintheta + wx * wy * (kOne - costheta);
    CM(R, 2, 0) = -wy * sintheta + wx * wz * (kOne - costheta);
    CM(R, 0, 1) = wx * wy * (kOne - costheta) - wz * sintheta;
    CM(R, 1, 1) = costheta + wy * wy * (kOne - costheta);
    CM(R, 2, 1) = wx * sintheta + wy * wz * (kOne - costheta);
    CM(R, 0, 2) = wy * sintheta + wx * wz * (kOne - costheta);
    CM(R, 1, 2) = -wx * sintheta + wy * wz * (kOne - costheta);
    CM(R, 2, 2) = costheta + wz * wz * (kOne - costheta);
  } else {
    CM(R, 0, 0) = kOne;   CM(R, 1, 0) = aa[2];  CM(R, 2, 0) = -aa[1];
    CM(R, 0, 1) = -aa[2]; CM(R, 1, 1) = kOne;   CM(R, 2, 1) = aa[0];
    CM(R, 0, 2) = aa[1];  CM(R, 1, 2) = -aa[0]; CM(R, 2, 2) = kOne;
  }
}

void RotationMatrixToAngleAxis(const double* R, double aa[3]) {
  // ceres::RotationMatrixToAngleAxis = RotationMatrixToQuaternion + QuaternionToAngleAxis (photobundle.cc:650)
  double q[4];
  const double trace = CMc(R, 0, 0) + CMc(R, 1, 1) + CMc(R, 2, 2);
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (CMc(R, 2, 1) - CMc(R, 1, 2)) * t;
    q[2] = (CMc(R, 0, 2) - CMc(R, 2, 0)) * t;
    q[3] = (CMc(R, 1, 0) - CMc(R, 0, 1)) * t;
  } else {
    int i = 0;
    if (CMc(R, 1, 1) > CMc(R, 0, 0)) i = 1;
    if (CMc(R, 2, 2) > CMc(R, i, i)) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    double t = std::sqrt(CMc(R, i, i) - CMc(R, j, j) - CMc(R, k, k) + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (CMc(R, k, j) - CMc(R, j, k)) * t;
    q[j + 1] = (CMc(R, j, i) + CMc(R, i, j)) * t;
    q[k + 1] = (CMc(R, k, i) + CMc(R, i, k)) * t;
  }
  const double q1 = q[1], q2 = q[2], q3 = q[3];
  const double sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3;
  if (sin_squared_theta > 0.0) {
    const double sin_theta = std::sqrt(sin_squared_theta);
    const double cos_theta = q[0];
    const double two_theta = 2.0 * ((cos_theta < 0.0) ? std::atan2(-sin_theta, -cos_theta) : std::atan2(sin_theta, cos_theta));
    const double k = two_theta / sin_theta;
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  } else {
    const double k = 2.0;
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  }
}

// ---------------------------------------------------------------------------------------------
// Sampler (sample_eigen.h).
// ---------------------------------------------------------------------------------------------
inline int TruncToInt(float x) {
  // static_cast<int>(float) is UB out of range; on the reference's x86 target cvttss2si yields INT_MIN
  // for NaN / out-of-range, which LinearInitAxis then treats as "ix < 0".
  return (x >= -2147483648.0f && x < 2147483648.0f) ? static_cast<int>(x) : INT_MIN;
}

inline void LinearInitAxis(float x, int size, int* x1, int* x2, float* dx) {
  // sample_eigen.h:34-52
  const int ix = TruncToInt(x);
  if (ix < 0) { *x1 = 0; *x2 = 0; *dx = 1.0f; }
  else if (ix > size - 2) { *x1 = size - 1; *x2 = size - 1; *dx = 1.0f; }
  else { *x1 = ix; *x2 = ix + 1; *dx = static_cast<float>(*x2) - x; }
}

inline float Blend(float dy, float dx, float a11, float a12, float a21, float a22) {
  // sample_eigen.h:82-83: float*float products are float; (1.0 - dx) promotes to double; (1 - dy) is float;
  // the sum is double and is rounded to float on the store into PixelType sample[].
  const double top = static_cast<double>(dx * a11) + (1.0 - static_cast<double>(dx)) * static_cast<double>(a12);
  const double bot = static_cast<double>(dx * a21) + (1.0 - static_cast<double>(dx)) * static_cast<double>(a22);
  const float omdy = 1 - dy;
  return static_cast<float>(static_cast<double>(dy) * top + static_cast<double>(omdy) * bot);
}

inline void SampleLinear(const float* I, const float* Gx, const float* Gy, int rows, int cols,
                         float y, float x, float* sample) {
  // sample_eigen.h:56-102
  int x1, y1, x2, y2; float dx, dy;
  LinearInitAxis(y, rows, &y1, &y2, &dy);
  LinearInitAxis(x, cols, &x1, &x2, &dx);
  const size_t i11 = (size_t)y1 * cols + x1, i12 = (size_t)y1 * cols + x2;
  const size_t i21 = (size_t)y2 * cols + x1, i22 = (size_t)y2 * cols + x2;
  sample[0] = Blend(dy, dx, I[i11], I[i12], I[i21], I[i22]);
  sample[1] = Blend(dy, dx, Gx[i11], Gx[i12], Gx[i21], Gx[i22]);
  sample[2] = Blend(dy, dx, Gy[i11], Gy[i12], Gy[i21], Gy[i22]);
}

// SampleWithDerivative (sample_eigen.h:108-126) + Chain::Rule (jet_extras.h:87-111)
inline double SampleWithDerivative(const float* I, const float* Gx, const float* Gy, int rows, int cols,
                                   const double& x, const double& y) {
  float s[3];
  SampleLinear(I, Gx, Gy, rows, cols, static_cast<float>(y), static_cast<float>(x), s);
  return static_cast<double>(s[0]);   // Chain<float,2,double>::Rule returns f
}
inline Dual SampleWithDerivative(const float* I, const float* Gx, const float* Gy, int rows, int cols,
                                 const Dual& x, const Dual& y) {
  float s[3];
  SampleLinear(I, Gx, Gy, rows, cols, static_cast<float>(y.a), static_cast<float>(x.a), s);
  Dual f; f.a = static_cast<double>(s[0]);
  const double gx = static_cast<double>(s[1]), gy = static_cast<double>(s[2]);
  for (int i = 0; i < kN; ++i) f.v[i] = gx * x.v[i] + gy * y.v[i];
  return f;
}

inline int NumChannels(const oracle_problem* p) { return p->n_channels > 0 ? p->n_channels : 1; }
// residuals of one block: channels x patch pixels (photobundle.cc:692 AutoDiffCostFunction(..., p0.size()))
inline int BlockRows(const oracle_problem* p) { return NumChannels(p) * (2 * p->radius + 1) * (2 * p->radius + 1); }

// ---------------------------------------------------------------------------------------------
// DescriptorError::operator() (photobundle.cc:696-727): channel-major residuals, the patch weights restart with
// every channel (the index j of :714 is declared in the row loop's initialiser).
// ---------------------------------------------------------------------------------------------
template <class T>
inline void DescriptorError(const oracle_problem* p, int slot, const double* p0, const T* camera, const T* point,
                            T* residuals) {
  T xw[3];
  AngleAxisRotatePointT(camera, point, xw);
  xw[0] = xw[0] + camera[3];
  xw[1] = xw[1] + camera[4];
  xw[2] = xw[2] + camera[5];
  // Calibration::project (calibration.h:34-38)
  const T u_w = ((xw[0] * T(p->fx)) / xw[2]) + T(p->cx);
  const T v_w = ((xw[1] * T(p->fy)) / xw[2]) + T(p->cy);

  const size_t npix = (size_t)p->rows * p->cols;
  const int R = p->radius, C = NumChannels(p);
  int i = 0;
  for (int k = 0; k < C; ++k) {
    const float* I = p->planes + ((size_t)slot * C + k) * 3 * npix;   // getChannel(k), getChannelGradient(k)
    const float* Gx = I + npix;
    const float* Gy = Gx + npix;
    for (int y = -R, j = 0; y <= R; ++y) {
      const T v = v_w + T((double)y);
      for (int x = -R; x <= R; ++x, ++i, ++j) {
        const T u = u_w + T((double)x);
        const T i0 = T(p0[i]);
        const T i1 = SampleWithDerivative(I, Gx, Gy, p->rows, p->cols, u, v);
        residuals[i] = p->weights[j] * (i0 - i1);
      }
    }
  }
}

// Analytic 2x9 Jacobian A = d(u,v)/d(camera[6], point[3]) of the SAME arithmetic (incl. the small-angle branch).
void ProjectionJacobian(const oracle_problem* p, const double* cam, const double* pt, double* u, double* v,
                        double A[2][9]) {
  const double* w = cam;
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double Rm[9];  // row-major d(xw)/d(pt)
  double dW[3][3];  // d(xw)/d(omega): dW[i][k]
  double xw[3];
  AngleAxisRotatePointT<double>(cam, pt, xw);
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    double Rc[9];
    AngleAxisToRotationMatrix(cam, Rc);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rm[3 * i + j] = CMc(Rc, i, j);
    // Gallego & Yezzi: d(R p)/d w = -R [p]x (w w^T + (R^T - I)[w]x) / theta^2
    double px[9] = {0, -pt[2], pt[1], pt[2], 0, -pt[0], -pt[1], pt[0], 0};
    double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double B[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      double acc = w[i] * w[j];
      for (int k = 0; k < 3; ++k) acc += (Rm[3 * k + i] - (i == k ? 1.0 : 0.0)) * wx[3 * k + j];
      B[3 * i + j] = acc / theta2;
    }
    double Rp[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double acc = 0; for (int k = 0; k < 3; ++k) acc += Rm[3 * i + k] * px[3 * k + j]; Rp[3 * i + j] = acc; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double acc = 0; for (int k = 0; k < 3; ++k) acc += Rp[3 * i + k] * B[3 * k + j]; dW[i][j] = -acc; }
  } else {
    // result = pt + w x pt  ->  d/dpt = I + [w]x,  d/dw = -[pt]x
    const double Rl[9] = {1, -w[2], w[1], w[2], 1, -w[0], -w[1], w[0], 1};
    std::memcpy(Rm, Rl, sizeof(Rl));
    const double npx[9] = {0, pt[2], -pt[1], -pt[2], 0, pt[0], pt[1], -pt[0], 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dW[i][j] = npx[3 * i + j];
  }
  xw[0] += cam[3]; xw[1] += cam[4]; xw[2] += cam[5];
  *u = ((xw[0] * p->fx) / xw[2]) + p->cx;
  *v = ((xw[1] * p->fy) / xw[2]) + p->cy;
  const double iz = 1.0 / xw[2];
  const double Ju[3] = {p->fx * iz, 0.0, -p->fx * xw[0] * iz * iz};
  const double Jv[3] = {0.0, p->fy * iz, -p->fy * xw[1] * iz * iz};
  for (int k = 0; k < 3; ++k) {
    A[0][k] = Ju[0] * dW[0][k] + Ju[1] * dW[1][k] + Ju[2] * dW[2][k];
    A[1][k] = Jv[0] * dW[0][k] + Jv[1] * dW[1][k] + Jv[2] * dW[2][k];
    A[0][3 + k] = Ju[k];
    A[1][3 + k] = Jv[k];
    A[0][6 + k] = Ju[0] * Rm[k] + Ju[1] * Rm[3 + k] + Ju[2] * Rm[6 + k];
    A[1][6 + k] = Jv[0] * Rm[k] + Jv[1] * Rm[3 + k] + Jv[2] * Rm[6 + k];
  }
}

// One block; raw residuals and (optionally) raw Jacobians, row-major P x 6 and P x 3.
void EvalBlock(const oracle_problem* p, int obs, bool autodiff, double* r, double* Jc, double* Jp) {
  const int P = BlockRows(p);
  const int pt = p->obs_point[obs], slot = p->obs_slot[obs];
  const double* cam = p->cams + 6 * slot;
  const double* X = p->xyz + 3 * (size_t)pt;
  const double* p0 = p->desc + (size_t)pt * P;
  if (!Jc && !Jp) {
    DescriptorError<double>(p, slot, p0, cam, X, r);
    return;
  }
  if (autodiff) {
    // AutoDiffCostFunction<DescriptorError, DYNAMIC, 6, 3> (photobundle.cc:692): Jet<double, 9>
    Dual c[6], x[3];
    for (int k = 0; k < 6; ++k) c[k] = Dual(cam[k], k);
    for (int k = 0; k < 3; ++k) x[k] = Dual(X[k], 6 + k);
    Dual stack_res[128];
    std::vector<Dual> heap_res;
    Dual* res = stack_res;
    if (P > 128) { heap_res.resize(P); res = heap_res.data(); }
    DescriptorError<Dual>(p, slot, p0, c, x, res);
    for (int i = 0; i < P; ++i) {
      r[i] = res[i].a;
      if (Jc) for (int k = 0; k < 6; ++k) Jc[6 * i + k] = res[i].v[k];
      if (Jp) for (int k = 0; k < 3; ++k) Jp[3 * i + k] = res[i].v[6 + k];
    }
  } else {
    double u, v, A[2][9];
    ProjectionJacobian(p, cam, X, &u, &v, A);
    const size_t npix = (size_t)p->rows * p->cols;
    const int R = p->radius, C = NumChannels(p);
    int i = 0;
    for (int ch = 0; ch < C; ++ch) {
      const float* I = p->planes + ((size_t)slot * C + ch) * 3 * npix;
      const float* Gx = I + npix;
      const float* Gy = Gx + npix;
      int j = 0;
      for (int y = -R; y <= R; ++y) {
        const double vv = v + (double)y;
        for (int x = -R; x <= R; ++x, ++i, ++j) {
          const double uu = u + (double)x;
          float s[3];
          SampleLinear(I, Gx, Gy, p->rows, p->cols, static_cast<float>(vv), static_cast<float>(uu), s);
          const double w = p->weights[j];
          r[i] = w * (p0[i] - (double)s[0]);
          const double gx = (double)s[1], gy = (double)s[2];
          if (Jc) for (int k = 0; k < 6; ++k) Jc[6 * i + k] = -w * (gx * A[0][k] + gy * A[1][k]);
          if (Jp) for (int k = 0; k < 3; ++k) Jp[3 * i + k] = -w * (gx * A[0][6 + k] + gy * A[1][6 + k]);
        }
      }
    }
  }
}

// ceres::HuberLoss::Evaluate (loss_function.cc)
template <class Real>
inline void HuberEvaluate(double a_, Real s, Real rho[3]) {
  const Real a = a_;
  const Real b = a * a;
  if (s > b) {
    const Real r = std::sqrt(s);
    rho[0] = (Real)2.0 * a * r - b;
    rho[1] = std::max((Real)std::numeric_limits<double>::min(), a / r);
    rho[2] = -rho[1] / ((Real)2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// Program state: materialised block-sparse Jacobian like Ceres' BlockSparseMatrix for SPARSE_SCHUR.
// ---------------------------------------------------------------------------------------------
// Points per parallel work item of the reductions over points (gradient, column norms, Schur elimination): the
// partial sums are combined in chunk order, which makes every result independent of the thread count.
constexpr int kChunkPoints = 1024;

template <class Real>
struct ProgramT {
  const oracle_problem* p;
  int P, n_c, n_p, n_obs;
  std::vector<int> cam_col;      // slot -> first column among free camera params, -1 if constant
  int n_cam_params;              // 6 * free cameras
  int n_params;                  // n_cam_params + 3 n_p
  std::vector<int> pt_begin;     // CSR over obs (grouped by point)
  std::vector<char> cam_in_program;   // slot has at least one residual block (a camera nobody observes is never handed
                                      // to AddResidualBlock, photobundle.cc:791-804: Ceres does not know it exists)
  bool autodiff;
  int threads;
  // materialised (corrected, and after ScaleColumns: scaled) residuals + Jacobian
  std::vector<Real> r, Jc, Jp;
  int64_t jac_passes = 0, cost_passes = 0;

  void init(const oracle_problem* prob, bool ad, int nthreads) {
    p = prob; autodiff = ad; threads = std::max(1, nthreads);
    P = BlockRows(p);
    n_c = p->n_frames; n_p = p->n_points; n_obs = p->n_obs;
    cam_col.assign(n_c, -1);
    int col = 0;
    for (int c = 0; c < n_c; ++c) if (c != p->fixed_slot) { cam_col[c] = col; col += 6; }
    n_cam_params = col;
    n_params = n_cam_params + 3 * n_p;
    pt_begin.assign(n_p + 1, 0);
    for (int o = 0; o < n_obs; ++o) pt_begin[p->obs_point[o] + 1]++;
    for (int i = 0; i < n_p; ++i) pt_begin[i + 1] += pt_begin[i];
    cam_in_program.assign(n_c, 0);
    for (int o = 0; o < n_obs; ++o) cam_in_program[p->obs_slot[o]] = 1;
  }
  // |x| over the parameter blocks of the Ceres program: the free cameras WITH residual blocks and every point.  (The
  // columns of an unobserved free camera stay in this restatement's x -- zero Jacobian, zero step -- but not in |x|.)
  Real programNorm(const std::vector<double>& x) const {
    Real s = 0.0;
    for (int c = 0; c < n_c; ++c)
      if (cam_col[c] >= 0 && cam_in_program[c]) for (int k = 0; k < 6; ++k) s += x[cam_col[c] + k] * x[cam_col[c] + k];
    for (int i = n_cam_params; i < n_params; ++i) s += (Real)x[i] * (Real)x[i];
    return std::sqrt(s);
  }

  // x layout: [free cameras in slot order | points]
  void pack(const double* cams, const double* xyz, std::vector<double>& x) const {
    x.resize(n_params);
    for (int c = 0; c < n_c; ++c) if (cam_col[c] >= 0) std::memcpy(&x[cam_col[c]], cams + 6 * c, 6 * sizeof(double));
    std::memcpy(x.data() + n_cam_params, xyz, sizeof(double) * 3 * n_p);
  }
  void unpack(const std::vector<double>& x, double* cams, double* xyz) const {
    for (int c = 0; c < n_c; ++c) if (cam_col[c] >= 0) std::memcpy(cams + 6 * c, &x[cam_col[c]], 6 * sizeof(double));
    std::memcpy(xyz, x.data() + n_cam_params, sizeof(double) * 3 * n_p);
  }

  // Evaluator::Evaluate.  With want_jac: fills r/Jc/Jp (loss-corrected, unscaled) and gradient = J^T r.
  // The residual block itself (projection, fp32 sampler, dual-number / analytic Jacobian rows) is always evaluated in
  // double: that IS the function the reference minimises.  Every sum over rows / blocks runs in Real.
  bool evaluate(const std::vector<double>& x, Real* cost, bool want_jac, std::vector<Real>* gradient,
                std::vector<double>* block_sqnorm = nullptr) {
    std::vector<double> cams(p->cams, p->cams + 6 * n_c), xyz(3 * (size_t)n_p);
    unpack(x, cams.data(), xyz.data());
    oracle_problem q = *p; q.cams = cams.data(); q.xyz = xyz.data();
    if (want_jac) { r.resize((size_t)n_obs * P); Jc.resize((size_t)n_obs * P * 6); Jp.resize((size_t)n_obs * P * 3); ++jac_passes; }
    else ++cost_passes;
    std::vector<Real> block_cost(n_obs);
    if (block_sqnorm) block_sqnorm->resize(n_obs);
    bool ok = true;
    constexpr bool kSame = std::is_same<Real, double>::value;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int o = 0; o < n_obs; ++o) {
      std::vector<double> rl, jcl, jpl;
      double* rb; double* jc = nullptr; double* jp = nullptr;
      if (want_jac && kSame) {
        rb = reinterpret_cast<double*>(&r[(size_t)o * P]); jc = reinterpret_cast<double*>(&Jc[(size_t)o * P * 6]); jp = reinterpret_cast<double*>(&Jp[(size_t)o * P * 3]);
      } else {
        rl.resize(P); rb = rl.data();
        if (want_jac) { jcl.resize((size_t)P * 6); jpl.resize((size_t)P * 3); jc = jcl.data(); jp = jpl.data(); }
      }
      EvalBlock(&q, o, autodiff, rb, jc, jp);
      Real s = 0.0;
      for (int i = 0; i < P; ++i) s += (Real)rb[i] * (Real)rb[i];
      if (block_sqnorm) (*block_sqnorm)[o] = (double)s;
      if (!std::isfinite((double)s)) {
#pragma omp critical
        ok = false;
      }
      Real k = 1.0;
      if (p->huber > 0.0) {
        // ResidualBlock::Evaluate + Corrector (corrector.cc): rho'' <= 0 for Huber => plain sqrt(rho') scaling
        Real rho[3];
        HuberEvaluate<Real>(p->huber, s, rho);
        block_cost[o] = (Real)0.5 * rho[0];
        k = std::sqrt(rho[1]);
      } else {
        block_cost[o] = (Real)0.5 * s;
      }
      if (want_jac) {
        if (kSame) {
          if (p->huber > 0.0) {
            for (int i = 0; i < P * 6; ++i) jc[i] *= (double)k;
            for (int i = 0; i < P * 3; ++i) jp[i] *= (double)k;
            for (int i = 0; i < P; ++i) rb[i] *= (double)k;
          }
        } else {
          Real* ro = &r[(size_t)o * P]; Real* jco = &Jc[(size_t)o * P * 6]; Real* jpo = &Jp[(size_t)o * P * 3];
          for (int i = 0; i < P * 6; ++i) jco[i] = (Real)jc[i] * k;
          for (int i = 0; i < P * 3; ++i) jpo[i] = (Real)jp[i] * k;
          for (int i = 0; i < P; ++i) ro[i] = (Real)rb[i] * k;
        }
      }
    }
    Real c = 0.0;
    for (int o = 0; o < n_obs; ++o) c += block_cost[o];
    *cost = c;
    if (want_jac && gradient) {
      // Ceres accumulates J^T r with per-thread scratch (ProgramEvaluator); here: fixed chunks of kChunkPoints points, the
      // camera part reduced in chunk order, so the result does not depend on the thread count
      gradient->assign(n_params, 0.0);
      const int n_chunks = (n_p + kChunkPoints - 1) / kChunkPoints;
      std::vector<Real> part((size_t)n_chunks * n_cam_params, 0.0);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
      for (int ch = 0; ch < n_chunks; ++ch) {
        Real* gc = part.data() + (size_t)ch * n_cam_params;
        const int o0 = pt_begin[ch * kChunkPoints], o1 = pt_begin[std::min(n_p, (ch + 1) * kChunkPoints)];
        for (int o = o0; o < o1; ++o) {
          const int cc = cam_col[p->obs_slot[o]];
          Real* gp = gradient->data() + n_cam_params + 3 * (size_t)p->obs_point[o];
          const Real* rb = &r[(size_t)o * P];
          const Real* jc = &Jc[(size_t)o * P * 6];
          const Real* jp = &Jp[(size_t)o * P * 3];
          for (int i = 0; i < P; ++i) {
            if (cc >= 0) for (int k = 0; k < 6; ++k) gc[cc + k] += jc[6 * i + k] * rb[i];
            for (int k = 0; k < 3; ++k) gp[k] += jp[3 * i + k] * rb[i];
          }
        }
      }
      for (int ch = 0; ch < n_chunks; ++ch)
        for (int k = 0; k < n_cam_params; ++k) (*gradient)[k] = (ch == 0) ? part[k] : (*gradient)[k] + part[(size_t)ch * n_cam_params + k];
    }
    return ok;
  }

  // BlockSparseMatrix::SquaredColumnNorm
  void squared_column_norm(std::vector<Real>& out) const {
    out.assign(n_params, 0.0);
    const int n_chunks = (n_p + kChunkPoints - 1) / kChunkPoints;
    std::vector<Real> part((size_t)n_chunks * n_cam_params, 0.0);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int ch = 0; ch < n_chunks; ++ch) {
      Real* oc = part.data() + (size_t)ch * n_cam_params;
      const int o0 = pt_begin[ch * kChunkPoints], o1 = pt_begin[std::min(n_p, (ch + 1) * kChunkPoints)];
      for (int o = o0; o < o1; ++o) {
        const int cc = cam_col[p->obs_slot[o]];
        Real* np = out.data() + n_cam_params + 3 * (size_t)p->obs_point[o];
        const Real* jc = &Jc[(size_t)o * P * 6];
        const Real* jp = &Jp[(size_t)o * P * 3];
        for (int i = 0; i < P; ++i) {
          if (cc >= 0) for (int k = 0; k < 6; ++k) oc[cc + k] += jc[6 * i + k] * jc[6 * i + k];
          for (int k = 0; k < 3; ++k) np[k] += jp[3 * i + k] * jp[3 * i + k];
        }
      }
    }
    for (int ch = 0; ch < n_chunks; ++ch)
      for (int k = 0; k < n_cam_params; ++k) out[k] = (ch == 0) ? part[k] : out[k] + part[(size_t)ch * n_cam_params + k];
  }
  // BlockSparseMatrix::ScaleColumns
  void scale_columns(const std::vector<Real>& scale) {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int o = 0; o < n_obs; ++o) {
      const int cc = cam_col[p->obs_slot[o]];
      const Real* sp = scale.data() + n_cam_params + 3 * (size_t)p->obs_point[o];
      Real* jc = &Jc[(size_t)o * P * 6];
      Real* jp = &Jp[(size_t)o * P * 3];
      for (int i = 0; i < P; ++i) {
        if (cc >= 0) for (int k = 0; k < 6; ++k) jc[6 * i + k] *= scale[cc + k];
        for (int k = 0; k < 3; ++k) jp[3 * i + k] *= sp[k];
      }
    }
  }
  // y = J x
  void right_multiply(const std::vector<Real>& x, std::vector<Real>& y) const {
    y.assign((size_t)n_obs * P, 0.0);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int o = 0; o < n_obs; ++o) {
      const int cc = cam_col[p->obs_slot[o]];
      const Real* xp = x.data() + n_cam_params + 3 * (size_t)p->obs_point[o];
      const Real* jc = &Jc[(size_t)o * P * 6];
      const Real* jp = &Jp[(size_t)o * P * 3];
      Real* yb = &y[(size_t)o * P];
      for (int i = 0; i < P; ++i) {
        Real acc = 0.0;
        if (cc >= 0) for (int k = 0; k < 6; ++k) acc += jc[6 * i + k] * x[cc + k];
        for (int k = 0; k < 3; ++k) acc += jp[3 * i + k] * xp[k];
        yb[i] = acc;
      }
    }
  }
};

// 3x3 SPD inverse through Cholesky (Ceres InvertPSDMatrix -> Eigen LLT solve against identity).
template <class Real>
bool InvertPSD3(const Real* m /*row-major sym*/, Real* inv) {
  Real L[9] = {0};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j <= i; ++j) {
      Real s = m[3 * i + j];
      for (int k = 0; k < j; ++k) s -= L[3 * i + k] * L[3 * j + k];
      if (i == j) { if (!(s > 0.0)) return false; L[3 * i + i] = std::sqrt(s); }
      else L[3 * i + j] = s / L[3 * j + j];
    }
  }
  for (int c = 0; c < 3; ++c) {
    Real y[3], x[3];
    for (int i = 0; i < 3; ++i) { Real s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[3 * i + k] * y[k]; y[i] = s / L[3 * i + i]; }
    for (int i = 2; i >= 0; --i) { Real s = y[i]; for (int k = i + 1; k < 3; ++k) s -= L[3 * k + i] * x[k]; x[i] = s / L[3 * i + i]; }
    for (int i = 0; i < 3; ++i) inv[3 * i + c] = x[i];
  }
  return true;
}

// Dense Cholesky solve (the reduced camera system; SPARSE_SCHUR factorises exactly).
template <class Real>
bool CholeskySolve(std::vector<Real>& A, int n, std::vector<Real>& b) {
  for (int j = 0; j < n; ++j) {
    Real s = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(s > 0.0) || !std::isfinite((double)s)) return false;
    const Real d = std::sqrt(s);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      Real t = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = t / d;
    }
  }
  for (int i = 0; i < n; ++i) { Real s = b[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { Real s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
  return true;
}

// SchurComplementSolver::Solve: min |J y - r|^2 + |D y|^2, points eliminated (SchurEliminator::Eliminate /
// ::BackSubstitute), reduced system by Cholesky.  Returns false on LINEAR_SOLVER_FAILURE.
template <class Real>
bool SchurSolve(const ProgramT<Real>& g, const std::vector<Real>& D, std::vector<Real>& y) {
  const oracle_problem* p = g.p;
  const int n = g.n_cam_params, P = g.P;
  std::vector<Real> S((size_t)n * n, 0.0), rhs(n, 0.0);
  std::vector<Real> inv_ete((size_t)g.n_p * 9), gvec((size_t)g.n_p * 3);
  bool ok = true;
  // SchurEliminator::Eliminate runs its chunks on num_linear_solver_threads threads (photobundle.cc:754) with per-thread
  // buffers; here every chunk of kChunkPoints points accumulates into its own S / rhs and the chunks are added in order
  const int n_chunks = (g.n_p + kChunkPoints - 1) / kChunkPoints;
  std::vector<Real> S_part((size_t)n_chunks * n * n, 0.0), rhs_part((size_t)n_chunks * n, 0.0);
  std::vector<char> chunk_ok(n_chunks, 1);
  for (int i = 0; i < n; ++i) S_part[(size_t)i * n + i] = D[i] * D[i];     // chunk 0 starts from the LM diagonal
#pragma omp parallel for num_threads(g.threads) schedule(dynamic, 1)
  for (int ch = 0; ch < n_chunks; ++ch) {
  Real* S = S_part.data() + (size_t)ch * n * n;
  Real* rhs = rhs_part.data() + (size_t)ch * n;
  std::vector<Real> buf((size_t)g.n_c * 18 + 64);
  for (int pt = ch * kChunkPoints; pt < std::min(g.n_p, (ch + 1) * kChunkPoints); ++pt) {
    const int b = g.pt_begin[pt], e = g.pt_begin[pt + 1];
    if (b == e) continue;
    const Real* Dp = D.data() + n + 3 * (size_t)pt;
    Real ete[9] = {Dp[0] * Dp[0], 0, 0, 0, Dp[1] * Dp[1], 0, 0, 0, Dp[2] * Dp[2]};
    Real ge[3] = {0, 0, 0};
    if ((int)buf.size() < (e - b) * 18) buf.resize((size_t)(e - b) * 18);
    for (int o = b; o < e; ++o) {
      const int cc = g.cam_col[p->obs_slot[o]];
      const Real* rb = &g.r[(size_t)o * P];
      const Real* jc = &g.Jc[(size_t)o * P * 6];
      const Real* jp = &g.Jp[(size_t)o * P * 3];
      Real* etf = &buf[(size_t)(o - b) * 18];  // 3 x 6
      for (int k = 0; k < 18; ++k) etf[k] = 0.0;
      for (int i = 0; i < P; ++i) {
        for (int a = 0; a < 3; ++a) {
          for (int c = 0; c < 3; ++c) ete[3 * a + c] += jp[3 * i + a] * jp[3 * i + c];
          ge[a] += jp[3 * i + a] * rb[i];
        }
        if (cc >= 0) {
          for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) S[(size_t)(cc + a) * n + cc + c] += jc[6 * i + a] * jc[6 * i + c];
          for (int a = 0; a < 3; ++a) for (int c = 0; c < 6; ++c) etf[6 * a + c] += jp[3 * i + a] * jc[6 * i + c];
        }
      }
    }
    Real* inv = &inv_ete[(size_t)pt * 9];
    if (!InvertPSD3(ete, inv)) { chunk_ok[ch] = 0; break; }
    for (int a = 0; a < 3; ++a) gvec[(size_t)pt * 3 + a] = ge[a];
    Real ig[3];
    for (int a = 0; a < 3; ++a) ig[a] = inv[3 * a] * ge[0] + inv[3 * a + 1] * ge[1] + inv[3 * a + 2] * ge[2];
    for (int o = b; o < e; ++o) {
      const int cc = g.cam_col[p->obs_slot[o]];
      if (cc < 0) continue;
      const Real* rb = &g.r[(size_t)o * P];
      const Real* jc = &g.Jc[(size_t)o * P * 6];
      const Real* jp = &g.Jp[(size_t)o * P * 3];
      for (int i = 0; i < P; ++i) {
        const Real sj = rb[i] - (jp[3 * i] * ig[0] + jp[3 * i + 1] * ig[1] + jp[3 * i + 2] * ig[2]);
        for (int a = 0; a < 6; ++a) rhs[cc + a] += jc[6 * i + a] * sj;
      }
      const Real* etf1 = &buf[(size_t)(o - b) * 18];
      Real t[18];  // (E^T F_1)^T inv  : 6 x 3
      for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c)
        t[3 * a + c] = etf1[a] * inv[c] + etf1[6 + a] * inv[3 + c] + etf1[12 + a] * inv[6 + c];
      for (int o2 = b; o2 < e; ++o2) {
        const int c2 = g.cam_col[p->obs_slot[o2]];
        if (c2 < 0) continue;
        const Real* etf2 = &buf[(size_t)(o2 - b) * 18];
        for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c)
          S[(size_t)(cc + a) * n + c2 + c] -= t[3 * a] * etf2[c] + t[3 * a + 1] * etf2[6 + c] + t[3 * a + 2] * etf2[12 + c];
      }
    }
  }
  }
  for (int ch = 0; ch < n_chunks; ++ch) {
    if (!chunk_ok[ch]) ok = false;
    const Real* Sc = S_part.data() + (size_t)ch * n * n;
    const Real* rc = rhs_part.data() + (size_t)ch * n;
    for (size_t k = 0; k < (size_t)n * n; ++k) S[k] = (ch == 0) ? Sc[k] : S[k] + Sc[k];
    for (int k = 0; k < n; ++k) rhs[k] = (ch == 0) ? rc[k] : rhs[k] + rc[k];
  }
  if (n_chunks == 0) for (int i = 0; i < n; ++i) S[(size_t)i * n + i] = D[i] * D[i];
  if (!ok) return false;
  std::vector<Real> yc = rhs;
  if (n > 0 && !CholeskySolve(S, n, yc)) return false;
  y.assign(g.n_params, 0.0);
  for (int i = 0; i < n; ++i) y[i] = yc[i];
  // BackSubstitute (points are independent)
#pragma omp parallel for num_threads(g.threads) schedule(static)
  for (int pt = 0; pt < g.n_p; ++pt) {
    const int b = g.pt_begin[pt], e = g.pt_begin[pt + 1];
    if (b == e) continue;
    Real acc[3] = {0, 0, 0};
    for (int o = b; o < e; ++o) {
      const int cc = g.cam_col[p->obs_slot[o]];
      const Real* rb = &g.r[(size_t)o * P];
      const Real* jc = &g.Jc[(size_t)o * P * 6];
      const Real* jp = &g.Jp[(size_t)o * P * 3];
      for (int i = 0; i < P; ++i) {
        Real sj = rb[i];
        if (cc >= 0) for (int a = 0; a < 6; ++a) sj -= jc[6 * i + a] * yc[cc + a];
        for (int a = 0; a < 3; ++a) acc[a] += jp[3 * i + a] * sj;
      }
    }
    const Real* inv = &inv_ete[(size_t)pt * 9];
    for (int a = 0; a < 3; ++a) y[n + 3 * (size_t)pt + a] = inv[3 * a] * acc[0] + inv[3 * a + 1] * acc[1] + inv[3 * a + 2] * acc[2];
  }
  for (Real v : y) if (!std::isfinite((double)v)) return false;
  return true;
}

template <class Real>
Real Norm(const std::vector<Real>& v) { Real s = 0; for (Real x : v) s += x * x; return std::sqrt(s); }
template <class Real>
Real MaxAbs(const std::vector<Real>& v) { Real s = 0; for (Real x : v) s = std::max(s, std::fabs(x)); return s; }


using Program = ProgramT<double>;

double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// ceres::Solve (photobundle.cc:829) -> TrustRegionMinimizer::Minimize, in the accumulation type Real
template <class Real>
int SolveT(oracle_problem* p, const oracle_options* opt, oracle_summary* sum, oracle_iteration* its, int max_its_out) {
  // ceres::Solve (photobundle.cc:829) -> TrustRegionMinimizer::Minimize (Ceres >= 1.12 control flow)
  const double t_start = Now();
  ProgramT<Real> g; g.init(p, opt->use_autodiff != 0, opt->num_threads);
  std::memset(sum, 0, sizeof(*sum));
  sum->num_residual_blocks = g.n_obs;
  sum->num_residuals = g.n_obs * g.P;
  sum->fixed_cost = 0.0;  // every block has a free point (SURVEY 8c "Constant camera 0")
  sum->termination_type = 1;
  std::snprintf(sum->message, sizeof(sum->message), "Maximum number of iterations reached.");

  std::vector<double> x, candidate_x;
  std::vector<Real> gradient, scale, diagonal, D, step, delta, model_residuals;
  g.pack(p->cams, p->xyz, x);
  Real x_cost = 0.0, candidate_cost = 0.0, x_norm = g.programNorm(x);
  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int num_consecutive_invalid_steps = 0;
  int n_it = 0;
  Real minimum_cost;
  std::vector<double> best_x = x;

  oracle_iteration it;
  std::memset(&it, 0, sizeof(it));
  auto push_iteration = [&](const oracle_iteration& s) {
    if (n_it < max_its_out && its) its[n_it] = s;
    ++n_it;
  };
  auto eval_grad_jac = [&]() -> bool {
    // TrustRegionMinimizer::EvaluateGradientAndJacobian
    if (!g.evaluate(x, &x_cost, true, &gradient)) return false;
    it.cost = (double)x_cost + sum->fixed_cost;
    if (opt->jacobi_scaling) {
      if (it.iteration == 0) {
        g.squared_column_norm(scale);
        for (Real& s : scale) s = (Real)1.0 / ((Real)1.0 + std::sqrt(s));
      }
      g.scale_columns(scale);
    } else if (scale.empty()) scale.assign(g.n_params, 1.0);
    it.gradient_max_norm = (double)MaxAbs(gradient);   // Euclidean Plus: |x - (x - g)|_inf
    it.gradient_norm = (double)Norm(gradient);
    return true;
  };

  // IterationZero
  double t_iter = Now();
  it.iteration = 0; it.eta = 1e-1;
  if (!eval_grad_jac()) {
    sum->termination_type = 2;
    std::snprintf(sum->message, sizeof(sum->message), "Initial residual and Jacobian evaluation failed.");
    sum->total_time_in_seconds = Now() - t_start;
    return 2;
  }
  sum->initial_cost = (double)x_cost + sum->fixed_cost;
  minimum_cost = x_cost;
  it.step_is_valid = 1; it.step_is_successful = 1;

  bool done = false;
  auto finalize = [&]() -> bool {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful) {
      ++sum->num_successful_steps;
      if (x_cost < minimum_cost || it.iteration == 0) { minimum_cost = x_cost; best_x = x; it.step_is_nonmonotonic = 0; }
      else it.step_is_nonmonotonic = 1;
    } else ++sum->num_unsuccessful_steps;
    it.trust_region_radius = radius;
    const double now = Now();
    it.iteration_time_in_seconds = now - t_iter;
    it.cumulative_time_in_seconds = now - t_start;
    push_iteration(it);
    if (it.iteration >= opt->max_num_iterations) {
      sum->termination_type = 1; std::snprintf(sum->message, sizeof(sum->message), "Maximum number of iterations reached. Number of iterations: %d.", it.iteration); return false; }
    if (it.step_is_successful && it.gradient_max_norm <= opt->gradient_tolerance) {
      sum->termination_type = 0; std::snprintf(sum->message, sizeof(sum->message), "Gradient tolerance reached. Gradient max norm: %e <= %e", it.gradient_max_norm, opt->gradient_tolerance); return false; }
    if (radius <= opt->min_trust_region_radius) {
      sum->termination_type = 0; std::snprintf(sum->message, sizeof(sum->message), "Minimum trust region radius reached. Trust region radius: %e <= %e", radius, opt->min_trust_region_radius); return false; }
    return true;
  };
  auto step_rejected = [&]() { radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; };

  while (!done && finalize()) {
    t_iter = Now();
    const int iteration = it.iteration + 1;
    const double prev_gmax = it.gradient_max_norm, prev_gnorm = it.gradient_norm;
    std::memset(&it, 0, sizeof(it));
    it.iteration = iteration; it.eta = 1e-1;
    it.gradient_max_norm = prev_gmax; it.gradient_norm = prev_gnorm;

    // ComputeTrustRegionStep -> LevenbergMarquardtStrategy::ComputeStep
    const double t_solve = Now();
    if (!reuse_diagonal) {
      g.squared_column_norm(diagonal);
      for (Real& d : diagonal) d = std::min(std::max(d, (Real)opt->min_lm_diagonal), (Real)opt->max_lm_diagonal);
    }
    D.resize(g.n_params);
    for (int i = 0; i < g.n_params; ++i) D[i] = std::sqrt(diagonal[i] / (Real)radius);
    bool solved = SchurSolve(g, D, step);
    reuse_diagonal = true;
    bool step_is_valid = false;
    Real model_cost_change = 0.0;
    if (solved) {
      for (Real& s : step) s *= (Real)-1.0;
      // model_cost_change = -(J step)^T (r + J step / 2)
      g.right_multiply(step, model_residuals);
      Real acc = 0.0;
      for (size_t i = 0; i < model_residuals.size(); ++i) acc += model_residuals[i] * (g.r[i] + model_residuals[i] / (Real)2.0);
      model_cost_change = -acc;
      step_is_valid = model_cost_change > 0.0;
    }
    it.step_solver_time_in_seconds = Now() - t_solve;
    it.model_cost_change = (double)model_cost_change;
    it.linear_solver_iterations = 1;
    if (!step_is_valid) {
      // HandleInvalidStep
      ++num_consecutive_invalid_steps;
      if (num_consecutive_invalid_steps >= opt->max_num_consecutive_invalid_steps) {
        sum->termination_type = 2;
        std::snprintf(sum->message, sizeof(sum->message), "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d", opt->max_num_consecutive_invalid_steps);
        it.cost = (double)x_cost + sum->fixed_cost; it.trust_region_radius = radius;
        push_iteration(it);
        done = true; break;
      }
      step_rejected();  // LevenbergMarquardtStrategy::StepIsInvalid == StepRejected(0)
      it.cost = (double)x_cost + sum->fixed_cost;
      it.cost_change = 0.0; it.step_norm = 0.0; it.relative_decrease = 0.0;
      it.step_is_valid = 0; it.step_is_successful = 0;
      continue;
    }
    it.step_is_valid = 1;
    num_consecutive_invalid_steps = 0;
    delta.resize(g.n_params);
    for (int i = 0; i < g.n_params; ++i) delta[i] = step[i] * scale[i];

    // ComputeCandidatePointAndEvaluateCost
    candidate_x.resize(g.n_params);
    for (int i = 0; i < g.n_params; ++i) candidate_x[i] = (double)((Real)x[i] + delta[i]);
    if (!g.evaluate(candidate_x, &candidate_cost, false, nullptr)) candidate_cost = std::numeric_limits<double>::max();
    it.candidate_cost = (double)candidate_cost;

    // ParameterToleranceReached
    Real step_norm;
    { Real s = 0; for (int i = 0; i < g.n_params; ++i) { const Real d = x[i] - candidate_x[i]; s += d * d; } step_norm = std::sqrt(s); it.step_norm = (double)step_norm; }
    const Real step_size_tolerance = (Real)opt->parameter_tolerance * (x_norm + (Real)opt->parameter_tolerance);
    if (step_norm <= step_size_tolerance) {
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Parameter tolerance reached. Relative step_norm: %e <= %e.", (double)(step_norm / (x_norm + (Real)opt->parameter_tolerance)), opt->parameter_tolerance);
      it.cost = (double)x_cost; it.trust_region_radius = radius;
      done = true; break;
    }
    // FunctionToleranceReached
    const Real cost_change = x_cost - candidate_cost;
    it.cost_change = (double)cost_change;
    const Real absolute_function_tolerance = (Real)opt->function_tolerance * x_cost;
    if (std::fabs(cost_change) <= absolute_function_tolerance) {
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Function tolerance reached. |cost_change|/cost: %e <= %e", (double)(std::fabs(cost_change) / x_cost), opt->function_tolerance);
      it.cost = (double)x_cost; it.trust_region_radius = radius;
      done = true; break;
    }
    // IsStepSuccessful (monotonic TrustRegionStepEvaluator::StepQuality)
    const Real relative_decrease = (x_cost - candidate_cost) / model_cost_change;
    it.relative_decrease = (double)relative_decrease;
    if (relative_decrease > (Real)opt->min_relative_decrease) {
      // HandleSuccessfulStep
      x = candidate_x; x_norm = g.programNorm(x);
      if (!eval_grad_jac()) {
        sum->termination_type = 2; std::snprintf(sum->message, sizeof(sum->message), "Residual and Jacobian evaluation failed.");
        done = true; break;
      }
      it.step_is_successful = 1;
      // LevenbergMarquardtStrategy::StepAccepted
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = std::min(opt->max_trust_region_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
    } else {
      // HandleUnsuccessfulStep
      it.step_is_successful = 0;
      step_rejected();
      it.cost = (double)candidate_cost + sum->fixed_cost;
    }
  }
  if (done && sum->termination_type == 0 && n_it <= it.iteration) {
    // convergence detected inside the loop body: Ceres returns without pushing this iteration's summary
  }
  g.unpack(best_x, p->cams, p->xyz);
  sum->final_cost = (double)minimum_cost + sum->fixed_cost;
  sum->num_iterations = std::min(n_it, max_its_out);
  sum->num_jacobian_passes = g.jac_passes;
  sum->num_cost_passes = g.cost_passes;
  sum->total_time_in_seconds = Now() - t_start;
  return sum->termination_type;
}


// =============================================================================================
// C interface
// =============================================================================================
extern "C" {

void oracle_default_options(oracle_options* o) {
  o->max_num_iterations = 500;             // photobundle.cc:751
  int t = 4;
#ifdef _OPENMP
  t = std::min(omp_get_max_threads(), 4);  // photobundle.cc:824
#endif
  o->num_threads = t;
  o->function_tolerance = 1e-6;            // photobundle.cc:756-758
  o->gradient_tolerance = 1e-6;
  o->parameter_tolerance = 1e-6;
  o->initial_trust_region_radius = 1e4;    // Ceres Solver::Options defaults
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->use_autodiff = 1;
  o->extended_precision = 0;
}

void oracle_imgradient_f32(const float* src, int rows, int cols, float* Ix, float* Iy) {
  // imgproc.cc:27-44 (row), :74-95 (image): first/last row and col are zero, interior 0.5 * central difference
  std::memset(Ix, 0, sizeof(float) * (size_t)rows * cols);
  std::memset(Iy, 0, sizeof(float) * (size_t)rows * cols);
  for (int y = 1; y < rows - 1; ++y) {
    const float* s = src + (size_t)y * cols;
    float* ix = Ix + (size_t)y * cols;
    float* iy = Iy + (size_t)y * cols;
    for (int x = 1; x < cols - 1; ++x) {
      ix[x] = 0.5f * (s[x + 1] - s[x - 1]);
      iy[x] = 0.5f * (s[x + cols] - s[x - cols]);
    }
  }
}

void oracle_planes_from_u8(const uint8_t* img, int rows, int cols, float* I, float* Gx, float* Gy) {
  // photobundle.cc:231 (image.cast<float>()) then DescriptorFrame ctor :172-175 -> imgradient on the float channel
  for (size_t i = 0; i < (size_t)rows * cols; ++i) I[i] = static_cast<float>(img[i]);
  oracle_imgradient_f32(I, rows, cols, Gx, Gy);
}

// ---- multi-channel descriptors (DescriptorFrame::Create, photobundle.cc:225-248) -----------------------------------
namespace {
inline int Reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; }
  return i;
}
// cv::getGaussianKernel(n, sigma, CV_32F) for sigma > 0: exp(-x^2 / (2 sigma^2)) in double, stored as float, normalised
// by the (double) sum of the stored floats.
void GaussianKernelF32(int n, double sigma, float* k) {
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
}  // namespace

void oracle_census(const uint8_t* src, int rows, int cols, uint8_t* dst) {
  // imgproc.cc:126-197: bit b is set when the b-th neighbour (row-major 3x3 order without the centre) >= centre;
  // first / last row and column are zero
  std::memset(dst, 0, (size_t)rows * cols);
  for (int y = 1; y < rows - 1; ++y)
    for (int x = 1; x < cols - 1; ++x) {
      const uint8_t* s = src + (size_t)y * cols + x;
      const uint8_t c = *s;
      dst[(size_t)y * cols + x] = (uint8_t)(((s[-cols - 1] >= c) ? 0x01 : 0) | ((s[-cols] >= c) ? 0x02 : 0) | ((s[-cols + 1] >= c) ? 0x04 : 0) |
                                            ((s[-1] >= c) ? 0x08 : 0) | ((s[1] >= c) ? 0x10 : 0) | ((s[cols - 1] >= c) ? 0x20 : 0) |
                                            ((s[cols] >= c) ? 0x40 : 0) | ((s[cols + 1] >= c) ? 0x80 : 0));
    }
}

void oracle_gaussian_blur_u8_3x3(const uint8_t* src, int rows, int cols, double sigma, uint8_t* dst) {
  // cv::GaussianBlur(8U, Size(3,3), sigma) as OpenCV 2.4 / 3.x run it: the float kernel in 8-bit fixed point
  // (coefficients cvRound(k * 256)), integer row pass, integer column pass, (sum + 2^15) >> 16, BORDER_REFLECT_101.
  // (OpenCV is an absent dependency: restated from its documented behaviour, not pinned.)
  float kf[3];
  GaussianKernelF32(3, sigma, kf);
  int k[3];
  for (int i = 0; i < 3; ++i) k[i] = (int)std::nearbyint((double)kf[i] * 256.0);
  std::vector<int> tmp((size_t)rows * cols);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const uint8_t* s = src + (size_t)y * cols;
      tmp[(size_t)y * cols + x] = k[0] * s[Reflect101(x - 1, cols)] + k[1] * s[x] + k[2] * s[Reflect101(x + 1, cols)];
    }
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const int v = k[0] * tmp[(size_t)Reflect101(y - 1, rows) * cols + x] + k[1] * tmp[(size_t)y * cols + x] +
                    k[2] * tmp[(size_t)Reflect101(y + 1, rows) * cols + x];
      const int r = (v + (1 << 15)) >> 16;
      dst[(size_t)y * cols + x] = (uint8_t)std::min(255, std::max(0, r));
    }
}

void oracle_gaussian_blur_f32_5x5(const float* src, int rows, int cols, double sigma, float* dst) {
  // cv::GaussianBlur(32F, Size(5,5), sigma): separable, symmetric form k0 c + k1 (l1 + r1) + k2 (l2 + r2) in float,
  // BORDER_REFLECT_101 (restated, not pinned: OpenCV's vector and scalar paths associate the sum differently)
  float k[5];
  GaussianKernelF32(5, sigma, k);
  std::vector<float> tmp((size_t)rows * cols);
  for (int y = 0; y < rows; ++y) {
    const float* s = src + (size_t)y * cols;
    for (int x = 0; x < cols; ++x) {
      float v = s[x] * k[2];
      v += (s[Reflect101(x - 1, cols)] + s[Reflect101(x + 1, cols)]) * k[1];
      v += (s[Reflect101(x - 2, cols)] + s[Reflect101(x + 2, cols)]) * k[0];
      tmp[(size_t)y * cols + x] = v;
    }
  }
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float v = tmp[(size_t)y * cols + x] * k[2];
      v += (tmp[(size_t)Reflect101(y - 1, rows) * cols + x] + tmp[(size_t)Reflect101(y + 1, rows) * cols + x]) * k[1];
      v += (tmp[(size_t)Reflect101(y - 2, rows) * cols + x] + tmp[(size_t)Reflect101(y + 2, rows) * cols + x]) * k[0];
      dst[(size_t)y * cols + x] = v;
    }
}

int oracle_num_channels(int descriptor_type) { return descriptor_type == 1 ? 3 : (descriptor_type == 2 ? 8 : 1); }

void oracle_descriptor_channels(const uint8_t* img, int rows, int cols, int descriptor_type, float* channels) {
  // DescriptorFrame::Create (photobundle.cc:225-248); descriptor_type: 0 Intensity, 1 IntensityAndGradient, 2 BitPlanes
  const size_t npix = (size_t)rows * cols;
  if (descriptor_type == 2) {
    // computeBitPlanes (imgproc.cc:219-245) with its defaults sigma_ct = 1, sigma_bp = 1.5 (imgproc.h:44-46)
    std::vector<uint8_t> smooth(npix), census(npix);
    oracle_gaussian_blur_u8_3x3(img, rows, cols, 1.0, smooth.data());
    oracle_census(smooth.data(), rows, cols, census.data());
    std::vector<float> plane(npix);
    for (int b = 0; b < 8; ++b) {
      for (size_t i = 0; i < npix; ++i) plane[i] = (float)((census[i] & (1 << b)) >> b);
      oracle_gaussian_blur_f32_5x5(plane.data(), rows, cols, 1.5, channels + (size_t)b * npix);
    }
    return;
  }
  for (size_t i = 0; i < npix; ++i) channels[i] = (float)img[i];
  if (descriptor_type == 1) {
    // imgradient(u8 image) -> channels 1, 2 (:236-237); destination float => scale 0.5 (imgproc.h:54-58)
    float* gx = channels + npix;
    float* gy = channels + 2 * npix;
    std::memset(gx, 0, sizeof(float) * npix);
    std::memset(gy, 0, sizeof(float) * npix);
    for (int y = 1; y < rows - 1; ++y)
      for (int x = 1; x < cols - 1; ++x) {
        const uint8_t* s = img + (size_t)y * cols + x;
        gx[(size_t)y * cols + x] = 0.5f * ((float)s[1] - (float)s[-1]);
        gy[(size_t)y * cols + x] = 0.5f * ((float)s[cols] - (float)s[-cols]);
      }
  }
}

void oracle_channel_planes(const float* channels, int n_channels, int rows, int cols, float* planes) {
  // DescriptorFrame ctor (photobundle.cc:172-175): every channel gets its own gradient images
  const size_t npix = (size_t)rows * cols;
  for (int k = 0; k < n_channels; ++k) {
    float* I = planes + (size_t)k * 3 * npix;
    std::memcpy(I, channels + (size_t)k * npix, sizeof(float) * npix);
    oracle_imgradient_f32(I, rows, cols, I + npix, I + 2 * npix);
  }
}

void oracle_sample_linear(const float* I, const float* Gx, const float* Gy, int rows, int cols, float y, float x,
                          float out[3]) {
  SampleLinear(I, Gx, Gy, rows, cols, y, x, out);
}

void oracle_angle_axis_rotate_point(const double aa[3], const double pt[3], double out[3]) {
  AngleAxisRotatePointT<double>(aa, pt, out);
}
void oracle_angle_axis_to_rotation_matrix(const double aa[3], double R[9]) { AngleAxisToRotationMatrix(aa, R); }
void oracle_rotation_matrix_to_angle_axis(const double R[9], double aa[3]) { RotationMatrixToAngleAxis(R, aa); }

void oracle_make_patch_weights(int radius, int do_gaussian, double* w) {
  // photobundle.cc:617-644 with s_x = s_y = a = 1
  const int n = (2 * radius + 1) * (2 * radius + 1);
  if (!do_gaussian) { for (int i = 0; i < n; ++i) w[i] = 1.0; return; }
  double sum = 0.0;
  for (int r = -radius, i = 0; r <= radius; ++r) {
    const double d_r = (r * r) / 1.0;
    for (int c = -radius; c <= radius; ++c, ++i) {
      const double d_c = (c * c) / 1.0;
      const double v = 1.0 * std::exp(-0.5 * (d_r + d_c));
      w[i] = v; sum += v;
    }
  }
  for (int i = 0; i < n; ++i) w[i] /= sum;
}

void oracle_extract_patch(const float* I, int rows, int cols, int u, int v, int radius, double* dst) {
  // photobundle.cc:466-479
  const int max_cols = cols - radius - 1, max_rows = rows - radius - 1;
  for (int r = -radius, i = 0; r <= radius; ++r) {
    const int r_i = std::max(radius, std::min(v + r, max_rows));
    for (int c = -radius; c <= radius; ++c, ++i) {
      const int c_i = std::max(radius, std::min(u + c, max_cols));
      dst[i] = static_cast<double>(I[(size_t)r_i * cols + c_i]);
    }
  }
}

void oracle_eval_block(const oracle_problem* p, int obs, int use_autodiff, double* residuals, double* jac_cam,
                       double* jac_pt) {
  EvalBlock(p, obs, use_autodiff != 0, residuals, jac_cam, jac_pt);
}

void oracle_linearize(const oracle_problem* p, int use_autodiff, int num_threads, double* cost, double* block_sqnorm,
                      double* grad_cams, double* grad_pts, double* U, double* V, double* W) {
  Program g; g.init(p, use_autodiff != 0, num_threads);
  std::vector<double> x; g.pack(p->cams, p->xyz, x);
  std::vector<double> grad, sq;
  const bool want = grad_cams || grad_pts || U || V || W;
  g.evaluate(x, cost, want, want ? &grad : nullptr, block_sqnorm ? &sq : nullptr);
  if (block_sqnorm) std::memcpy(block_sqnorm, sq.data(), sizeof(double) * g.n_obs);
  if (!want) return;
  if (grad_cams) {
    std::memset(grad_cams, 0, sizeof(double) * 6 * g.n_c);
    for (int c = 0; c < g.n_c; ++c) if (g.cam_col[c] >= 0) std::memcpy(grad_cams + 6 * c, &grad[g.cam_col[c]], 6 * sizeof(double));
  }
  if (grad_pts) std::memcpy(grad_pts, grad.data() + g.n_cam_params, sizeof(double) * 3 * g.n_p);
  if (U) std::memset(U, 0, sizeof(double) * 36 * g.n_c);
  if (V) std::memset(V, 0, sizeof(double) * 9 * (size_t)g.n_p);
  const int P = g.P;
  for (int o = 0; o < g.n_obs; ++o) {
    const int slot = p->obs_slot[o], pt = p->obs_point[o];
    const bool free_cam = g.cam_col[slot] >= 0;
    const double* jc = &g.Jc[(size_t)o * P * 6];
    const double* jp = &g.Jp[(size_t)o * P * 3];
    double w[18] = {0};
    for (int i = 0; i < P; ++i) {
      if (U && free_cam) for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) U[36 * slot + 6 * a + c] += jc[6 * i + a] * jc[6 * i + c];
      if (V) for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) V[9 * (size_t)pt + 3 * a + c] += jp[3 * i + a] * jp[3 * i + c];
      if (W && free_cam) for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c) w[3 * a + c] += jc[6 * i + a] * jp[3 * i + c];
    }
    if (W) std::memcpy(W + 18 * (size_t)o, w, sizeof(w));
  }
}

void oracle_block_products(const oracle_problem* p, int use_autodiff, int num_threads, double* out) {
  // test hook: what one residual block contributes to J^T J and J^T r (ResidualBlock::Evaluate + Corrector applied)
  const int P = BlockRows(p);
#pragma omp parallel for num_threads(std::max(1, num_threads)) schedule(static)
  for (int o = 0; o < p->n_obs; ++o) {
    std::vector<double> r(P), jc((size_t)P * 6), jp((size_t)P * 3);
    EvalBlock(p, o, use_autodiff != 0, r.data(), jc.data(), jp.data());
    double s = 0.0;
    for (int i = 0; i < P; ++i) s += r[i] * r[i];
    if (p->huber > 0.0) {
      double rho[3];
      HuberEvaluate<double>(p->huber, s, rho);
      const double k = std::sqrt(rho[1]);
      for (double& v : jc) v *= k;
      for (double& v : jp) v *= k;
      for (double& v : r) v *= k;
    }
    double* q = out + 72 * (size_t)o;
    for (int k = 0; k < 72; ++k) q[k] = 0.0;
    for (int i = 0; i < P; ++i) {
      for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) q[6 * a + c] += jc[6 * i + a] * jc[6 * i + c];
      for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c) q[36 + 3 * a + c] += jc[6 * i + a] * jp[3 * i + c];
      for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) q[54 + 3 * a + c] += jp[3 * i + a] * jp[3 * i + c];
      for (int a = 0; a < 6; ++a) q[63 + a] += jc[6 * i + a] * r[i];
      for (int a = 0; a < 3; ++a) q[69 + a] += jp[3 * i + a] * r[i];
    }
  }
}

int oracle_solve(oracle_problem* p, const oracle_options* opt, oracle_summary* sum, oracle_iteration* its, int max_its_out) {
  // opt->extended_precision: the REFEREE -- the same program with every accumulation (block norms, cost, gradient,
  // column norms, Schur elimination, Cholesky, back-substitution, model cost change, step norms) in x87 extended
  // precision (64-bit significand: rounding noise 2^-11 of the double program's); residual blocks and their Jacobian
  // rows are still evaluated in double with the fp32 sampler (they define the function).  Used by the tests to tell
  // rounding noise from drift (tests/test_gpu_fullsize.py).
  if (opt->extended_precision) return SolveT<long double>(p, opt, sum, its, max_its_out);
  return SolveT<double>(p, opt, sum, its, max_its_out);
}

}  // extern "C"
