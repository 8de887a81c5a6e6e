// stand-in for okvis_kinematics/include/okvis/kinematics/Transformation.hpp (r(), q(), constructor from r and q)
#pragma once
#include "mock_eigen.hpp"
namespace okvis {
namespace kinematics {
class Transformation {
 public:
  Transformation() = default;
  Transformation(const Eigen::Vector3d& r, const Eigen::Quaterniond& q) : r_(r), q_(q) {}
  const Eigen::Vector3d& r() const { return r_; }
  const Eigen::Quaterniond& q() const { return q_; }
 private:
  Eigen::Vector3d r_;
  Eigen::Quaterniond q_;
};
}  // namespace kinematics
}  // namespace okvis
