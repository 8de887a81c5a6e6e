// ORACLE -- test infrastructure only.
// Pinhole projection + distortion models, restating (paths under
// /root/reference/okvis_ros/okvis/okvis_cv/include/okvis/cameras/implementation/):
//   PinholeCamera.hpp:143-212 (project with 2x3 point Jacobian), :332-348 (projectHomogeneous)
//   RadialTangentialDistortion.hpp:90-133, EquidistantDistortion.hpp:91-186,
//   RadialTangentialDistortion8.hpp:113-175, NoDistortion.hpp
// The distortion Jacobians are written from the closed-form derivatives of the same
// distortion functions (mathematically identical to the reference's generated code).
#pragma once
#include "orc_math.hpp"

namespace orc {

enum DistortionModel { DIST_NONE = 0, DIST_RADTAN = 1, DIST_EQUIDISTANT = 2, DIST_RADTAN8 = 3 };

struct Camera {
  int model = DIST_NONE;
  double fu = 1, fv = 1, cu = 0, cv = 0;
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // radtan: k1 k2 p1 p2; equi: k1..k4; radtan8: k1 k2 p1 p2 k3 k4 k5 k6
  int width = 0, height = 0;
};

// returns false when the distortion model rejects the point (RadTan8: rho > 9).
inline bool distort(const Camera& c, const double* u, double* d, double* J /*2x2 row-major or null*/) {
  const double u0 = u[0], u1 = u[1];
  switch (c.model) {
    case DIST_NONE: {
      d[0] = u0; d[1] = u1;
      if (J) { J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1; }
      return true;
    }
    case DIST_RADTAN: {  // RadialTangentialDistortion.hpp:90-110
      const double k1 = c.k[0], k2 = c.k[1], p1 = c.k[2], p2 = c.k[3];
      const double mx = u0 * u0, my = u1 * u1, mxy = u0 * u1, rho = mx + my;
      const double rad = k1 * rho + k2 * rho * rho;
      d[0] = u0 + u0 * rad + 2.0 * p1 * mxy + p2 * (rho + 2.0 * mx);
      d[1] = u1 + u1 * rad + 2.0 * p2 * mxy + p1 * (rho + 2.0 * my);
      if (J) {
        J[0] = 1 + rad + k1 * 2.0 * mx + k2 * rho * 4 * mx + 2.0 * p1 * u1 + 6 * p2 * u0;
        J[2] = k1 * 2.0 * u0 * u1 + k2 * 4 * rho * u0 * u1 + p1 * 2.0 * u0 + 2.0 * p2 * u1;
        J[1] = J[2];
        J[3] = 1 + rad + k1 * 2.0 * my + k2 * rho * 4 * my + 6 * p1 * u1 + 2.0 * p2 * u0;
      }
      return true;
    }
    case DIST_EQUIDISTANT: {  // EquidistantDistortion.hpp:91-186
      const double k1 = c.k[0], k2 = c.k[1], k3 = c.k[2], k4 = c.k[3];
      const double r = std::sqrt(u0 * u0 + u1 * u1);
      const double th = std::atan(r);
      const double th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
      const double poly = 1 + k1 * th2 + k2 * th4 + k3 * th6 + k4 * th8;
      const double thd = th * poly;
      const double s = (r > 1e-8) ? thd / r : 1.0;
      d[0] = s * u0; d[1] = s * u1;
      if (J) {
        if (r > 1e-8) {
          // d = s(r) u ;  dd/du = s I + (ds/dr)/r * u u^T
          const double dthd = 1 + 3 * k1 * th2 + 5 * k2 * th4 + 7 * k3 * th6 + 9 * k4 * th8;  // d(thd)/d(th)
          const double dth = 1.0 / (1.0 + r * r);                                             // d(th)/dr
          const double ds_over_r = (dthd * dth * r - thd) / (r * r * r);                      // (ds/dr)/r
          J[0] = s + ds_over_r * u0 * u0;
          J[1] = ds_over_r * u0 * u1;
          J[2] = J[1];
          J[3] = s + ds_over_r * u1 * u1;
        } else {
          J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1;
        }
      }
      return true;
    }
    case DIST_RADTAN8: {  // RadialTangentialDistortion8.hpp:113-175
      const double k1 = c.k[0], k2 = c.k[1], p1 = c.k[2], p2 = c.k[3];
      const double k3 = c.k[4], k4 = c.k[5], k5 = c.k[6], k6 = c.k[7];
      const double mx = u0 * u0, my = u1 * u1, mxy = u0 * u1, rho = mx + my;
      if (rho > 9.0) return false;
      const double num = 1.0 + ((k3 * rho + k2) * rho + k1) * rho;
      const double den = 1.0 + ((k6 * rho + k5) * rho + k4) * rho;
      const double rad = num / den;
      d[0] = u0 * rad + 2.0 * p1 * mxy + p2 * (rho + 2.0 * mx);
      d[1] = u1 * rad + 2.0 * p2 * mxy + p1 * (rho + 2.0 * my);
      if (J) {
        const double dnum = k1 + rho * (2 * k2 + 3 * k3 * rho);
        const double dden = k4 + rho * (2 * k5 + 3 * k6 * rho);
        const double drad = (dnum * den - num * dden) / (den * den);  // d(rad)/d(rho)
        J[0] = rad + 2 * drad * mx + 2.0 * p1 * u1 + 6.0 * p2 * u0;
        J[1] = 2 * drad * mxy + 2.0 * p1 * u0 + 2.0 * p2 * u1;
        J[2] = J[1];
        J[3] = rad + 2 * drad * my + 6.0 * p1 * u1 + 2.0 * p2 * u0;
      }
      return true;
    }
  }
  return false;
}

enum ProjStatus { PROJ_OK = 0, PROJ_OUTSIDE = 1, PROJ_MASKED = 2, PROJ_BEHIND = 3, PROJ_INVALID = 4 };

// PinholeCamera.hpp:143-212.  J is 2x3 row-major (may be null).
inline int project(const Camera& c, const double* p, double* kp, double* J) {
  if (std::fabs(p[2]) < 1.0e-12) {
    // reference returns Invalid and leaves the outputs unwritten; we define them as zero.
    kp[0] = kp[1] = 0;
    if (J) std::memset(J, 0, 6 * sizeof(double));
    return PROJ_INVALID;
  }
  const double rz = 1.0 / p[2], rz2 = rz * rz;
  const double u[2] = {p[0] * rz, p[1] * rz};
  double d[2] = {0, 0}, Jd[4] = {0, 0, 0, 0};
  const bool ok = distort(c, u, d, J ? Jd : nullptr);
  if (J) {
    J[0] = c.fu * Jd[0] * rz;
    J[1] = c.fu * Jd[1] * rz;
    J[2] = -c.fu * (p[0] * Jd[0] + p[1] * Jd[1]) * rz2;
    J[3] = c.fv * Jd[2] * rz;
    J[4] = c.fv * Jd[3] * rz;
    J[5] = -c.fv * (p[0] * Jd[2] + p[1] * Jd[3]) * rz2;
  }
  kp[0] = c.fu * d[0] + c.cu;
  kp[1] = c.fv * d[1] + c.cv;
  if (!ok) return PROJ_INVALID;
  if (c.width > 0 && !(kp[0] >= 0 && kp[0] < c.width && kp[1] >= 0 && kp[1] < c.height)) return PROJ_OUTSIDE;
  return p[2] > 0.0 ? PROJ_OK : PROJ_BEHIND;
}

// PinholeCamera.hpp:332-348.  Jh is 2x4 row-major (last column zero); the sign flip for
// w<0 is applied to the point only, exactly as the reference does.
inline int projectHomogeneous(const Camera& c, const double* hp, double* kp, double* Jh) {
  double head[3] = {hp[0], hp[1], hp[2]};
  if (hp[3] < 0) { head[0] = -head[0]; head[1] = -head[1]; head[2] = -head[2]; }
  double J3[6];
  const int st = project(c, head, kp, Jh ? J3 : nullptr);
  if (Jh) {
    Jh[0] = J3[0]; Jh[1] = J3[1]; Jh[2] = J3[2]; Jh[3] = 0;
    Jh[4] = J3[3]; Jh[5] = J3[4]; Jh[6] = J3[5]; Jh[7] = 0;
  }
  return st;
}

}  // namespace orc
