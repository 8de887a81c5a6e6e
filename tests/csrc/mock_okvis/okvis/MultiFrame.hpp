// stand-in for okvis_cv/include/okvis/MultiFrame.hpp:60-250 + cameras/CameraBase.hpp:70-312 (the accessors the estimator uses)
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "mock_eigen.hpp"
#include <okvis/Time.hpp>
#include <okvis/kinematics/Transformation.hpp>
namespace okvis {
namespace cameras {
class CameraBase {
 public:
  CameraBase(int w, int h, std::string dist, std::vector<double> intr) : w_(w), h_(h), dist_(std::move(dist)), intr_(std::move(intr)) {}
  uint32_t imageWidth() const { return (uint32_t)w_; }
  uint32_t imageHeight() const { return (uint32_t)h_; }
  void getIntrinsics(Eigen::VectorXd& intrinsics) const { intrinsics.resize(intr_.size()); for (size_t i = 0; i < intr_.size(); ++i) intrinsics[i] = intr_[i]; }
  const std::string distortionType() const { return dist_; }
 private:
  int w_, h_; std::string dist_; std::vector<double> intr_;
};
}  // namespace cameras
class MultiFrame {
 public:
  struct Keypoint { double u, v, size; };
  uint64_t id() const { return id_; }
  const okvis::Time& timestamp() const { return stamp_; }
  size_t numFrames() const { return T_SC_.size(); }
  std::shared_ptr<const okvis::kinematics::Transformation> T_SC(size_t i) const { return T_SC_[i]; }
  std::shared_ptr<const cameras::CameraBase> geometry(size_t i) const { return geo_[i]; }
  bool getKeypoint(size_t cam, size_t k, Eigen::Vector2d& kp) const { kp = Eigen::Vector2d(kps_[cam][k].u, kps_[cam][k].v); return true; }
  bool getKeypointSize(size_t cam, size_t k, double& size) const { size = kps_[cam][k].size; return true; }
  size_t numKeypoints(size_t cam) const { return kps_[cam].size(); }
  size_t numKeypoints() const { size_t n = 0; for (auto& k : kps_) n += k.size(); return n; }
  // test-side construction
  uint64_t id_ = 0; okvis::Time stamp_;
  std::vector<std::shared_ptr<const okvis::kinematics::Transformation>> T_SC_;
  std::vector<std::shared_ptr<const cameras::CameraBase>> geo_;
  std::vector<std::vector<Keypoint>> kps_;
};
typedef std::shared_ptr<MultiFrame> MultiFramePtr;
}  // namespace okvis
