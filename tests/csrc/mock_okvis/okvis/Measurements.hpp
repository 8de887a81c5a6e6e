// stand-in for okvis_common/include/okvis/Measurements.hpp:60-204 (Measurement<T>, IMU / sonar / depth readings, deques)
#pragma once
#include <deque>
#include "mock_eigen.hpp"
#include <okvis/Time.hpp>
namespace okvis {
template <class T> struct Measurement { okvis::Time timeStamp; T measurement; };
struct ImuSensorReadings { Eigen::Vector3d gyroscopes, accelerometers; };
struct SonarReading { double range, heading; };
struct DepthReading { double depth; };
typedef Measurement<ImuSensorReadings> ImuMeasurement;
typedef std::deque<ImuMeasurement> ImuMeasurementDeque;
typedef Measurement<SonarReading> SonarMeasurement;
typedef std::deque<SonarMeasurement> SonarMeasurementDeque;
typedef Measurement<DepthReading> DepthMeasurement;
typedef std::deque<DepthMeasurement> DepthMeasurementDeque;
}  // namespace okvis
