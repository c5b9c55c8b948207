// g2o_stub.h -- DECLARATIONS-ONLY stand-in for the g2o names the reference's headers and integration/*.cpp mention (see cv_stub.h: syntax check only).
#ifndef CMS_TEST_G2O_STUB_H
#define CMS_TEST_G2O_STUB_H
#include "eigen_stub.h"
#include <iostream>
namespace g2o {
using Eigen::Vector2d; using Eigen::Vector3d; using Eigen::Matrix3d; using Eigen::Quaterniond;
typedef Eigen::Matrix<double, 6, 1> Vector6d; typedef Eigen::Matrix<double, 7, 1> Vector7d; typedef Eigen::Matrix<double, 6, 6> Matrix6d;
struct SE3Quat { SE3Quat(); SE3Quat(const Matrix3d&, const Vector3d&); SE3Quat(const Quaterniond&, const Vector3d&); const Vector3d& translation() const; const Quaterniond& rotation() const;
  SE3Quat operator*(const SE3Quat&) const; SE3Quat inverse() const; Vector3d map(const Vector3d&) const; Eigen::Matrix<double, 4, 4> to_homogeneous_matrix() const; static SE3Quat exp(const Vector6d&); Vector6d log() const; };
struct Sim3 { Sim3(); Sim3(const Matrix3d&, const Vector3d&, double); const Vector3d& translation() const; const Quaterniond& rotation() const; double scale() const; Sim3 inverse() const; Sim3 operator*(const Sim3&) const; Vector3d map(const Vector3d&) const; };
template <int D, class T> struct BaseVertex { typedef T EstimateType; virtual ~BaseVertex(); const T& estimate() const; void setEstimate(const T&); void setId(int); int id() const; void setFixed(bool); void setMarginalized(bool);
  virtual bool read(std::istream&) = 0; virtual bool write(std::ostream&) const = 0; virtual void setToOriginImpl() = 0; virtual void oplusImpl(const double*) = 0; T _estimate; };
template <int D, class E, class V> struct BaseUnaryEdge { virtual ~BaseUnaryEdge(); virtual bool read(std::istream&) = 0; virtual bool write(std::ostream&) const = 0; virtual void computeError() = 0; virtual void linearizeOplus();
  E _measurement; Eigen::Matrix<double, D, 1> _error; Eigen::Matrix<double, D, 6> _jacobianOplusXi; void* _vertices[1]; double chi2() const; void setLevel(int); int level() const; void setMeasurement(const E&); void setInformation(const Eigen::Matrix<double, D, D>&); void setVertex(int, void*); void setRobustKernel(void*); };
template <int D, class E, class V1, class V2> struct BaseBinaryEdge { virtual ~BaseBinaryEdge(); virtual bool read(std::istream&) = 0; virtual bool write(std::ostream&) const = 0; virtual void computeError() = 0; virtual void linearizeOplus();
  E _measurement; Eigen::Matrix<double, D, 1> _error; Eigen::Matrix<double, D, 3> _jacobianOplusXi; Eigen::Matrix<double, D, 6> _jacobianOplusXj; void* _vertices[2]; double chi2() const; void setLevel(int); int level() const; void setMeasurement(const E&); void setInformation(const Eigen::Matrix<double, D, D>&); void setVertex(int, void*); void setRobustKernel(void*); };
struct VertexSE3Expmap : BaseVertex<6, SE3Quat> { VertexSE3Expmap(); bool read(std::istream&); bool write(std::ostream&) const; void setToOriginImpl(); void oplusImpl(const double*); };
struct VertexSBAPointXYZ : BaseVertex<3, Vector3d> { VertexSBAPointXYZ(); bool read(std::istream&); bool write(std::ostream&) const; void setToOriginImpl(); void oplusImpl(const double*); };
struct VertexSim3Expmap : BaseVertex<7, Sim3> { VertexSim3Expmap(); bool read(std::istream&); bool write(std::ostream&) const; void setToOriginImpl(); void oplusImpl(const double*); };
}  // namespace g2o
#endif
