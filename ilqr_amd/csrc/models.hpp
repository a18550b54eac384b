// models.hpp -- device twins of the reference's Model plugins (include/model.h:6-21).
//
// A device model is a POD with compile-time NX/NU and three inlined members
//     dynamics(x,u,dx)   cost(x,u)   final_cost(x)
// i.e. exactly the three virtuals of the reference interface, resolved statically so the
// rollout and finite-difference kernels can inline them (a GPU cannot call host virtuals).
#pragma once
#include "common.hpp"

namespace ilqr {

// include/acrobot.h  (n=4, m=1).  I1=I2=l1=l2=m1=m2=1, lc1=lc2=.5, g=9.81 (:19-25).
struct AcrobotModel {
  static constexpr int NX = 4;
  static constexpr int NU = 1;
  double goal[4];  // acrobot.h:21  (3.1415, 0, 0, 0)
  double u_min[1], u_max[1];

  __device__ __forceinline__ void dynamics(const double* x, const double* u, double* dx) const {
    const double I1 = 1, I2 = 1, l1 = 1, l2 = 1, m1 = 1, m2 = 1, g = 9.81;
    const double lc1 = 0.5 * l1, lc2 = 0.5 * l2;
    const double q0 = x[0], q1 = x[1], qd0 = x[2], qd1 = x[3];
    // H(q), acrobot.h:43-51
    const double c2 = cos(q1);
    const double H00 = I1 + I2 + m2 * l1 * l1 + 2 * m2 * l1 * lc2 * c2;
    const double H01 = I2 + m2 * l1 * lc2 * c2;
    const double H10 = H01;
    const double H11 = I2;
    // C(q,qd), acrobot.h:53-61
    const double s2 = sin(q1);
    const double C00 = -2 * m2 * l1 * lc2 * s2 * qd1;
    const double C01 = -m2 * l2 * lc2 * s2 * qd1;
    const double C10 = m2 * l1 * lc2 * s2 * qd0;
    // G(q), acrobot.h:63-70
    const double s1 = sin(q0);
    const double s12 = sin(q0 + q1);
    const double G0 = m1 * g * lc1 * s1 + m2 * g * (l1 * s1 + lc2 * s12);
    const double G1 = m2 * g * lc2 * s12;
    // rhs = (0,u) - C*qd - G, acrobot.h:80
    const double r0 = (0.0 - (C00 * qd0 + C01 * qd1)) - G0;
    const double r1 = (u[0] - (C10 * qd0)) - G1;
    // H^-1 as Eigen's fixed 2x2 inverse (LU/InverseImpl.h:76-96): invdet then 4 products
    const double invdet = 1.0 / (H00 * H11 - H10 * H01);
    dx[0] = qd0;
    dx[1] = qd1;
    dx[2] = (H11 * invdet) * r0 + (-H01 * invdet) * r1;
    dx[3] = (-H10 * invdet) * r0 + (H00 * invdet) * r1;
  }
  // acrobot.h:83-92: Ks = Kd = 0, Kr = 0.1 -> the state terms are exact zeros for finite x
  __device__ __forceinline__ double cost(const double* x, const double* u) const {
    (void)x;
    const double Kr = 0.1;
    return Kr * Kr * (u[0] * u[0]);
  }
  // acrobot.h:94-100: Ks = Kd = 20
  __device__ __forceinline__ double final_cost(const double* x) const {
    const double q0 = goal[0] - x[0], q1 = goal[1] - x[1];
    const double qd0 = goal[2] - x[2], qd1 = goal[3] - x[3];
    const double Ks = 20.0, Kd = 20.0;
    return Ks * Ks * (q0 * q0 + q1 * q1) + Kd * Kd * (qd0 * qd0 + qd1 * qd1);
  }
};

// include/double_integrator.h  (n=4, m=2), mass = 1, Hx = diag(1,1,.2,.2), Hu = I.
struct DoubleIntegratorModel {
  static constexpr int NX = 4;
  static constexpr int NU = 2;
  double goal[4];
  double u_min[2], u_max[2];

  __device__ __forceinline__ void dynamics(const double* x, const double* u, double* dx) const {
    const double mass = 1.0;  // double_integrator.h:29-37
    dx[0] = x[2];
    dx[1] = x[3];
    dx[2] = u[0] / mass;
    dx[3] = u[1] / mass;
  }
  __device__ __forceinline__ double quad(const double* x, double scale) const {
    const double hx[4] = {1, 1, 0.2, 0.2};
    double d[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      d[i] = goal[i] - x[i];
      r[i] = scale * (hx[i] * d[i]);
    }
    return (r[0] * d[0] + r[2] * d[2]) + (r[1] * d[1] + r[3] * d[3]);
  }
  // double_integrator.h:39-43
  __device__ __forceinline__ double cost(const double* x, const double* u) const {
    return quad(x, 1.0) + (u[0] * u[0] + u[1] * u[1]);
  }
  // double_integrator.h:45-48
  __device__ __forceinline__ double final_cost(const double* x) const { return quad(x, 10.0); }
};

// include/model.h:12-15  x1 = x + dynamics(x,u)*dt
template <class M>
__device__ __forceinline__ void integrate_dynamics(const M& m, const double* x, const double* u, double dt,
                                                   double* x1) {
  double dx[M::NX];
  m.dynamics(x, u, dx);
#pragma unroll
  for (int i = 0; i < M::NX; i++) x1[i] = x[i] + dx[i] * dt;
}

}  // namespace ilqr
