// NOT Eigen.  Declaration-only test double (see boost_double.h).
#pragma once
namespace Eigen {
template <typename S, int R, int C> struct Matrix { S &operator()(int, int); S operator()(int, int) const; S &operator[](int); S operator[](int) const; };
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, 2, 3> Matrix23d;
template <typename S> struct Quaternion {};
typedef Quaternion<double> Quaterniond;
}  // namespace Eigen
