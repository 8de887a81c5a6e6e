// integration/okvis/ceres/ParameterBlockSized.hpp -- okvis::ceres::ParameterBlockSized<Dim, MinDim, T>
// (okvis_ceres/include/okvis/ceres/ParameterBlockSized.hpp:59-134): fixed-size storage + the estimate accessors.
#ifndef INTEGRATION_OKVIS_CERES_PARAMETERBLOCKSIZED_HPP_
#define INTEGRATION_OKVIS_CERES_PARAMETERBLOCKSIZED_HPP_

#include <cstring>
#include <iosfwd>
#include <stdexcept>

#include <okvis/assert_macros.hpp>
#include <okvis/ceres/ParameterBlock.hpp>

namespace okvis {
namespace ceres {

template <int Dim, int MinDim, class T>
class ParameterBlockSized : public okvis::ceres::ParameterBlock {
 public:
  OKVIS_DEFINE_EXCEPTION(Exception, std::runtime_error)
  static const int Dimension = Dim;
  static const int MinimalDimension = MinDim;
  typedef T parameter_t;

  ParameterBlockSized() { std::memset(parameters_, 0, sizeof(parameters_)); }
  virtual ~ParameterBlockSized() {}

  virtual void setEstimate(const parameter_t& estimate) = 0;
  virtual parameter_t estimate() const = 0;

  virtual void setParameters(const double* parameters) {
    if (!parameters) OKVIS_THROW(Exception, "Null pointer");
    std::memcpy(parameters_, parameters, sizeof(parameters_));
  }
  virtual double* parameters() { return parameters_; }
  virtual const double* parameters() const { return parameters_; }
  virtual size_t dimension() const { return Dimension; }
  virtual size_t minimalDimension() const { return MinimalDimension; }

  virtual bool read(std::istream&) { return false; }          // not implemented upstream either (:122-126)
  virtual bool write(std::ostream&) const { return false; }

 protected:
  double parameters_[Dimension];
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_PARAMETERBLOCKSIZED_HPP_
