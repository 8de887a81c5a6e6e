// stand-in for okvis_common/include/okvis/Variables.hpp (SpeedAndBias = 9-vector)
#pragma once
#include "mock_eigen.hpp"
namespace okvis { typedef Eigen::Matrix<double, 9, 1> SpeedAndBias; }
