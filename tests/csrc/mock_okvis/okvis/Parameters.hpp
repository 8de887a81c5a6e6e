// stand-in for okvis_common/include/okvis/Parameters.hpp:76-191 (the three parameter structs the estimator takes)
#pragma once
#include "mock_eigen.hpp"
#include <okvis/kinematics/Transformation.hpp>
namespace okvis {
struct ExtrinsicsEstimationParameters {
  double sigma_absolute_translation = 0, sigma_absolute_orientation = 0, sigma_c_relative_translation = 0, sigma_c_relative_orientation = 0;
};
struct ImuParameters {
  okvis::kinematics::Transformation T_BS;
  double a_max, g_max, sigma_g_c, sigma_bg, sigma_a_c, sigma_ba, sigma_gw_c, sigma_aw_c, tau, g;
  Eigen::Vector3d a0;
  int rate;
};
struct SonarParameters { okvis::kinematics::Transformation T_SSo; };
}  // namespace okvis
