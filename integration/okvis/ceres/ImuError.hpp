// integration/okvis/ceres/ImuError.hpp -- the part of okvis::ceres::ImuError callers outside the window solve use:
// the two static propagation() overloads (okvis_ceres/include/okvis/ceres/ImuError.hpp:122-142, src/ImuError.cpp:266-476,
// :479-697), called per IMU sample by ThreadedKFVio::imuConsumerLoop (ThreadedKFVio.cpp:808-819), per frame by :599 and
// by Frontend.cpp:258.  They run on the CPU (svin_host_imu_propagation: sequential, allocation-free); the IMU *factor*
// of the window lives on the GPU and is created by okvis::Estimator::addStates.
#ifndef INTEGRATION_OKVIS_CERES_IMUERROR_HPP_
#define INTEGRATION_OKVIS_CERES_IMUERROR_HPP_

#include <svin_ba.h>

#include <vector>

#include <okvis/Measurements.hpp>
#include <okvis/Parameters.hpp>
#include <okvis/Time.hpp>
#include <okvis/Variables.hpp>
#include <okvis/kinematics/Transformation.hpp>

namespace okvis {
namespace ceres {

class ImuError {
 public:
  typedef Eigen::Matrix<double, 15, 15> covariance_t;   ///< row-major storage is irrelevant for the symmetric covariance
  typedef Eigen::Matrix<double, 15, 15> jacobian_t;

  static int propagation(const okvis::ImuMeasurementDeque& imuMeasurements, const okvis::ImuParameters& imuParams,
                         okvis::kinematics::Transformation& T_WS, okvis::SpeedAndBias& speedAndBiases,   // NOLINT
                         const okvis::Time& t_start, const okvis::Time& t_end, covariance_t* covariance = 0, jacobian_t* jacobian = 0) {
    double integrals[7];
    return run(imuMeasurements, imuParams, T_WS, speedAndBiases, t_start, t_end, covariance, jacobian, integrals);
  }
  static int propagation(const okvis::ImuMeasurementDeque& imuMeasurements, const okvis::ImuParameters& imuParams,
                         okvis::kinematics::Transformation& T_WS, okvis::SpeedAndBias& speedAndBiases,   // NOLINT
                         const okvis::Time& t_start, const okvis::Time& t_end, covariance_t* covariance, jacobian_t* jacobian,
                         Eigen::Vector3d& acc_doubleinteg, Eigen::Vector3d& acc_integ, double& Del_t) {   // NOLINT
    double integrals[7];
    const int n = run(imuMeasurements, imuParams, T_WS, speedAndBiases, t_start, t_end, covariance, jacobian, integrals);
    if (n >= 0) {
      for (int k = 0; k < 3; ++k) { acc_doubleinteg[k] = integrals[k]; acc_integ[k] = integrals[3 + k]; }
      Del_t = integrals[6];
    }
    return n;
  }

 private:
  static int run(const okvis::ImuMeasurementDeque& imuMeasurements, const okvis::ImuParameters& p, okvis::kinematics::Transformation& T_WS,
                 okvis::SpeedAndBias& sb, const okvis::Time& t0, const okvis::Time& t1, covariance_t* cov, jacobian_t* jac, double* integrals) {
    // ImuMeasurementDeque is a std::deque: the samples are gathered once into the C layout (no heap beyond this vector)
    std::vector<svin_imu_sample> imu(imuMeasurements.size());
    size_t n = 0;
    for (const auto& m : imuMeasurements) {
      svin_imu_sample& s = imu[n++];
      s.sec = m.timeStamp.sec; s.nsec = m.timeStamp.nsec;
      for (int k = 0; k < 3; ++k) { s.gyr[k] = m.measurement.gyroscopes[k]; s.acc[k] = m.measurement.accelerometers[k]; }
    }
    svin_imu_params q;
    q.a_max = p.a_max; q.g_max = p.g_max; q.sigma_g_c = p.sigma_g_c; q.sigma_a_c = p.sigma_a_c; q.sigma_bg = p.sigma_bg;
    q.sigma_ba = p.sigma_ba; q.sigma_gw_c = p.sigma_gw_c; q.sigma_aw_c = p.sigma_aw_c; q.tau = p.tau; q.g = p.g;
    for (int k = 0; k < 3; ++k) q.a0[k] = p.a0[k];
    const Eigen::Vector3d r = T_WS.r();
    const Eigen::Quaterniond qq = T_WS.q();
    double T[7] = {r[0], r[1], r[2], qq.x(), qq.y(), qq.z(), qq.w()}, s9[9];
    for (int k = 0; k < 9; ++k) s9[k] = sb[k];
    // the 15x15 outputs are written row by row: Eigen's default (column-major) storage holds the transpose, so the
    // Jacobian goes through a row-major scratch and is copied element-wise
    double jacRows[225];
    const int used = svin_host_imu_propagation(imu.empty() ? nullptr : imu.data(), (int)imu.size(), &q, T, s9, t0.sec, t0.nsec, t1.sec,
                                               t1.nsec, cov ? cov->data() : nullptr, jac ? jacRows : nullptr, integrals);
    if (used < 0) return used;
    T_WS = okvis::kinematics::Transformation(Eigen::Vector3d(T[0], T[1], T[2]), Eigen::Quaterniond(T[6], T[3], T[4], T[5]));
    for (int k = 0; k < 9; ++k) sb[k] = s9[k];
    if (jac)
      for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) (*jac)(i, j) = jacRows[i * 15 + j];
    return used;
  }
};

}  // namespace ceres
}  // namespace okvis

#endif  // INTEGRATION_OKVIS_CERES_IMUERROR_HPP_
