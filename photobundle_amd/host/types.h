// types.h -- Eigen-free stand-ins for the public types of the reference API (reference src/types.h:9-77,
// src/eigen.h).  Eigen is not installed in this image; these PODs give the same spelling at the call sites of
// PhotometricBundleAdjustment (Mat44 T; T(r, c); T.data(); Mat44::Identity(); T.inverse(); A * B).
// Storage is column-major like Eigen's default, so `data()` is layout-compatible with a real Eigen::Matrix4d.
#ifndef PHOTOBUNDLE_AMD_TYPES_H
#define PHOTOBUNDLE_AMD_TYPES_H

#include <cmath>
#include <cstddef>
#include <iosfwd>
#include <memory>
#include <stdexcept>
#include <vector>

template <int R, int C>
struct MatRC {
  double m[R * C];
  MatRC() { for (int i = 0; i < R * C; ++i) m[i] = 0.0; }
  static MatRC Zero() { return MatRC(); }
  static MatRC Identity() { MatRC a; for (int i = 0; i < (R < C ? R : C); ++i) a(i, i) = 1.0; return a; }
  double& operator()(int r, int c) { return m[r + R * c]; }
  const double& operator()(int r, int c) const { return m[r + R * c]; }
  double& operator[](int i) { return m[i]; }
  const double& operator[](int i) const { return m[i]; }
  double* data() { return m; }
  const double* data() const { return m; }
  static constexpr int rows() { return R; }
  static constexpr int cols() { return C; }
  MatRC<C, R> transpose() const { MatRC<C, R> t; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) t(c, r) = (*this)(r, c); return t; }
  MatRC inverse() const;   // square only (3x3, 4x4), Gauss-Jordan with partial pivoting
};

template <int R, int K, int C>
inline MatRC<R, C> operator*(const MatRC<R, K>& a, const MatRC<K, C>& b) {
  MatRC<R, C> o;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) { double s = 0.0; for (int k = 0; k < K; ++k) s += a(r, k) * b(k, c); o(r, c) = s; }
  return o;
}
template <int R, int C>
inline MatRC<R, C> operator*(double s, const MatRC<R, C>& a) { MatRC<R, C> o; for (int i = 0; i < R * C; ++i) o.m[i] = s * a.m[i]; return o; }
template <int R, int C>
inline MatRC<R, C> operator+(const MatRC<R, C>& a, const MatRC<R, C>& b) { MatRC<R, C> o; for (int i = 0; i < R * C; ++i) o.m[i] = a.m[i] + b.m[i]; return o; }

template <int R, int C>
inline MatRC<R, C> MatRC<R, C>::inverse() const {
  static_assert(R == C, "inverse of a square matrix");
  double a[R][2 * R];
  for (int r = 0; r < R; ++r) for (int c = 0; c < R; ++c) { a[r][c] = (*this)(r, c); a[r][R + c] = (r == c) ? 1.0 : 0.0; }
  for (int i = 0; i < R; ++i) {
    int piv = i;
    for (int r = i + 1; r < R; ++r) if (std::fabs(a[r][i]) > std::fabs(a[piv][i])) piv = r;
    if (a[piv][i] == 0.0) throw std::runtime_error("singular matrix");
    if (piv != i) for (int c = 0; c < 2 * R; ++c) std::swap(a[i][c], a[piv][c]);
    const double d = 1.0 / a[i][i];
    for (int c = 0; c < 2 * R; ++c) a[i][c] *= d;
    for (int r = 0; r < R; ++r) if (r != i) { const double f = a[r][i]; if (f != 0.0) for (int c = 0; c < 2 * R; ++c) a[r][c] -= f * a[i][c]; }
  }
  MatRC o;
  for (int r = 0; r < R; ++r) for (int c = 0; c < R; ++c) o(r, c) = a[r][R + c];
  return o;
}

typedef MatRC<3, 3> Mat33;
typedef MatRC<4, 4> Mat44;
typedef MatRC<3, 4> Mat34;
typedef MatRC<2, 1> Vec2;
typedef MatRC<3, 1> Vec3;
typedef MatRC<4, 1> Vec4;

inline Vec3 MakeVec3(double x, double y, double z) { Vec3 v; v[0] = x; v[1] = y; v[2] = z; return v; }

// rigid transform of a 3-vector: (T * [X;1]).head<3>()   (Eigen::Isometry3d * Vec3 in the reference)
inline Vec3 TransformPoint(const Mat44& T, const Vec3& X) {
  Vec3 o;
  for (int r = 0; r < 3; ++r) o[r] = T(r, 0) * X[0] + T(r, 1) * X[1] + T(r, 2) * X[2] + T(r, 3);
  return o;
}

template <class T> using EigenAlignedContainer_ = std::vector<T>;
template <class T> using UniquePointer = std::unique_ptr<T>;
template <class T> using SharedPointer = std::shared_ptr<T>;

// row-major dense image (reference Image_<T>, types.h:19)
template <class T>
struct Image_ {
  int rows_ = 0, cols_ = 0;
  std::vector<T> d;
  Image_() {}
  Image_(int r, int c) { resize(r, c); }
  void resize(int r, int c) { rows_ = r; cols_ = c; d.assign((size_t)r * c, T()); }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  T& operator()(int r, int c) { return d[(size_t)r * cols_ + c]; }
  const T& operator()(int r, int c) const { return d[(size_t)r * cols_ + c]; }
  T* data() { return d.data(); }
  const T* data() const { return d.data(); }
};

struct ImageSize {
  int rows = 0, cols = 0;
  ImageSize(int r = 0, int c = 0) : rows(r), cols(c) {}
  int numel() const { return rows * cols; }
  int area() const { return numel(); }
  bool empty() const { return 0 == numel(); }
  ImageSize pyrDown() const { return ImageSize((rows + 1) / 2, (cols + 1) / 2); }   // reference types.h:70-73
};

#endif
