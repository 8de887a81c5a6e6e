// stand-in for the handful of Eigen types the shim touches (fixed-size vectors, VectorXd, Quaterniond)
#pragma once
#include <cstddef>
#include <vector>
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
namespace Eigen {
template <class T> struct aligned_allocator : std::allocator<T> {
  aligned_allocator() = default;
  template <class U> aligned_allocator(const aligned_allocator<U>&) {}
  template <class U> struct rebind { typedef aligned_allocator<U> other; };
};
enum { ColMajor = 0, RowMajor = 1 };
template <class S, int R, int C, int O = ColMajor>
struct Matrix {
  S v[R * C > 0 ? R * C : 1] = {};
  Matrix() = default;
  Matrix(S a, S b) { v[0] = a; v[1] = b; }
  Matrix(S a, S b, S c) { v[0] = a; v[1] = b; v[2] = c; }
  Matrix(S a, S b, S c, S d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
  S& operator[](int i) { return v[i]; }
  const S& operator[](int i) const { return v[i]; }
  S& operator()(int i, int j) { return v[O == RowMajor ? i * C + j : j * R + i]; }   // column-major like Eigen's default
  const S& operator()(int i, int j) const { return v[O == RowMajor ? i * C + j : j * R + i]; }
  S* data() { return v; }
  const S* data() const { return v; }
};
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
struct VectorXd {
  std::vector<double> v;
  void resize(size_t n) { v.resize(n); }
  size_t size() const { return v.size(); }
  double& operator[](size_t i) { return v[i]; }
  const double& operator[](size_t i) const { return v[i]; }
};
struct Quaterniond {
  double qw = 1, qx = 0, qy = 0, qz = 0;
  Quaterniond() = default;
  Quaterniond(double w, double x, double y, double z) : qw(w), qx(x), qy(y), qz(z) {}
  double w() const { return qw; }
  double x() const { return qx; }
  double y() const { return qy; }
  double z() const { return qz; }
};
}  // namespace Eigen
