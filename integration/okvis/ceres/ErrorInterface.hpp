// integration/okvis/ceres/ErrorInterface.hpp -- okvis::ceres::ErrorInterface
// (okvis_ceres/include/okvis/ceres/ErrorInterface.hpp:54-100): the pure interface the CPU-side error-term classes of
// this shim (PoseError, ReprojectionError, HomogeneousPointError) implement.  It never needed Ceres; it is re-declared
// here so that the shim set is self-contained.
#ifndef INTEGRATION_OKVIS_CERES_ERRORINTERFACE_HPP_
#define INTEGRATION_OKVIS_CERES_ERRORINTERFACE_HPP_

#include <cstddef>
#include <stdexcept>
#include <string>

#include <okvis/assert_macros.hpp>

namespace okvis {
namespace ceres {

class ErrorInterface {
 public:
  OKVIS_DEFINE_EXCEPTION(Exception, std::runtime_error)
  ErrorInterface() {}
  virtual ~ErrorInterface() {}
  virtual size_t residualDim() const = 0;
  virtual size_t parameterBlocks() const = 0;
  virtual size_t parameterBlockDim(size_t parameterBlockId) const = 0;
  /// residuals, ambient Jacobians (row-major, per parameter block) and the Jacobians in the minimal representation
  virtual bool EvaluateWithMinimalJacobians(double const* const* parameters, double* residuals, double** jacobians,
                                            double** jacobiansMinimal) const = 0;
  virtual std::string typeInfo() const = 0;
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_ERRORINTERFACE_HPP_
