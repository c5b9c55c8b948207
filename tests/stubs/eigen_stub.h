// eigen_stub.h -- DECLARATIONS-ONLY stand-in for the Eigen names the reference's headers and integration/*.cpp mention (see cv_stub.h: syntax check only).
#ifndef CMS_TEST_EIGEN_STUB_H
#define CMS_TEST_EIGEN_STUB_H
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
namespace Eigen {
template <class T, int R, int C> struct Matrix {
  Matrix(); Matrix(T, T); Matrix(T, T, T); Matrix(T, T, T, T);
  T& operator()(int, int); const T& operator()(int, int) const; T& operator()(int); const T& operator()(int) const; T& operator[](int); const T& operator[](int) const;
  Matrix operator+(const Matrix&) const; Matrix operator-(const Matrix&) const; Matrix operator*(T) const; template <int C2> Matrix<T, R, C2> operator*(const Matrix<T, C, C2>&) const;
  Matrix<T, C, R> transpose() const; Matrix inverse() const; T norm() const; T squaredNorm() const; T dot(const Matrix&) const; Matrix cross(const Matrix&) const; void setZero(); void setIdentity(); void normalize(); Matrix normalized() const;
  static Matrix Zero(); static Matrix Identity(); T* data(); const T* data() const; T x() const; T y() const; T z() const;
};
typedef Matrix<double, 2, 1> Vector2d; typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<double, 4, 1> Vector4d; typedef Matrix<double, 6, 1> Vector6d; typedef Matrix<double, 7, 1> Vector7d;
typedef Matrix<double, 2, 2> Matrix2d; typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d; typedef Matrix<float, 3, 3> Matrix3f; typedef Matrix<float, 3, 1> Vector3f;
template <class T> struct Quaternion { Quaternion(); Quaternion(T w, T x, T y, T z); explicit Quaternion(const Matrix<T, 3, 3>&); T x() const; T y() const; T z() const; T w() const; T& x(); T& y(); T& z(); T& w();
  Matrix<T, 3, 3> toRotationMatrix() const; Matrix<T, 4, 1> coeffs() const; void normalize(); Quaternion conjugate() const; Quaternion inverse() const; Quaternion operator*(const Quaternion&) const; Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1>&) const; };
typedef Quaternion<double> Quaterniond; typedef Quaternion<float> Quaternionf;
template <class T> struct aligned_allocator { typedef T value_type; };
}  // namespace Eigen
#endif
