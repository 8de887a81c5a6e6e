// integration/okvis/ceres/ParameterBlock.hpp -- okvis::ceres::ParameterBlock
// (okvis_ceres/include/okvis/ceres/ParameterBlock.hpp:56-150) without `ceres/ceres.h`.  On this backend the window's
// parameter blocks live in libsvin_ba.so; objects of this hierarchy are VALUES: the ones okvis_frontend builds on the
// stack (ProbabilisticStereoTriangulator.hpp:173-175, VioKeyframeWindowMatchingAlgorithm.cpp:453) and the snapshots
// Map::parameterBlockPtr / id2parameterBlockMap hand out.  Same members as the reference, including the manifold pointer
// (typed with the interface of CeresTypes.hpp).
#ifndef INTEGRATION_OKVIS_CERES_PARAMETERBLOCK_HPP_
#define INTEGRATION_OKVIS_CERES_PARAMETERBLOCK_HPP_

#include <cstddef>
#include <cstdint>
#include <string>

#include <okvis/ceres/CeresTypes.hpp>

namespace okvis {
namespace ceres {

class ParameterBlock {
 public:
  ParameterBlock() : id_(0), fixed_(false), manifoldPtr_(nullptr) {}
  virtual ~ParameterBlock() {}

  void setId(uint64_t id) { id_ = id; }
  uint64_t id() const { return id_; }
  void setFixed(bool fixed) { fixed_ = fixed; }
  bool fixed() const { return fixed_; }

  virtual void setParameters(const double* parameters) = 0;
  virtual double* parameters() = 0;
  virtual const double* parameters() const = 0;
  virtual size_t dimension() const = 0;
  virtual size_t minimalDimension() const = 0;

  // x0_plus_Delta = x0 [+] Delta_Chi and its companions, in the block's own minimal representation
  virtual void plus(const double* x0, const double* Delta_Chi, double* x0_plus_Delta) const = 0;
  virtual void plusJacobian(const double* x0, double* jacobian) const = 0;
  virtual void minus(const double* x0_plus_Delta, const double* x0, double* Delta_Chi) const = 0;
  virtual void liftJacobian(const double* x0, double* jacobian) const = 0;

  virtual void setManifoldPtr(const ::ceres::Manifold* manifoldPtr) { manifoldPtr_ = manifoldPtr; }
  virtual const ::ceres::Manifold* manifoldPtr() const { return manifoldPtr_; }

  virtual std::string typeInfo() const = 0;

 protected:
  uint64_t id_;
  bool fixed_;
  const ::ceres::Manifold* manifoldPtr_;
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_PARAMETERBLOCK_HPP_
