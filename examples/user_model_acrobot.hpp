// examples/user_model_acrobot.hpp -- a worked example of a USER device model (ILQR_MODEL_USER).
//
// The reference's Model is an open plugin interface (include/model.h:6-21): a user derives from it and hands the object
// to iLQR.  A GPU cannot call host virtuals, so the device-side counterpart of "deriving from Model" is a header like this
// one, compiled into a build of the library WITHOUT touching the library's sources:
//
//     python -c "from ilqr_amd import _build; _build.build_user('examples/user_model_acrobot.hpp', 'libilqr_amd_mine.so')"
//     (= hipcc <the library's flags> -DILQR_USER_MODEL_HEADER='"examples/user_model_acrobot.hpp"' -o libilqr_amd_mine.so ilqr_amd/csrc/capi.hip)
//
// and used through the same C ABI with ilqr_desc.model = ILQR_MODEL_USER (Python: BatchILQR("user", ..., lib=..., nx=4, nu=1,
// user_params=[...])).  The header is included inside namespace ilqr (ilqr_amd/csrc/models.hpp states the contract).
//
// The model here is a two-link acrobot whose goal state and terminal weights are RUN-TIME parameters
// (user_params = goal[4], Ks, Kd); with (3.1415, 0, 0, 0, 20, 20) it is the shipped acrobot, which is what
// tests/test_gpu_user_model.py uses it for: every kernel instantiated for this type must leave the same bits as for the
// built-in one.
template <class real_>
struct UserModelT {
  using real = real_;
  static constexpr int NX = 4;
  static constexpr int NU = 1;
  real u_min[1], u_max[1];  // filled by ilqr_create from ilqr_desc.u_min / u_max
  real goal[4];
  real Ks, Kd;

  void set_params(const double* p, int n) {  // host side: ilqr_desc.user_params
    const double dflt[6] = {3.1415, 0, 0, 0, 20, 20};
    for (int i = 0; i < 4; i++) goal[i] = (real)(i < n ? p[i] : dflt[i]);
    Ks = (real)(4 < n ? p[4] : dflt[4]);
    Kd = (real)(5 < n ? p[5] : dflt[5]);
  }

  // equations of motion of the two-link arm, unit masses / lengths / inertias, torque on the second joint
  __device__ __forceinline__ void dynamics(const real* x, const real* u, real* dx) const {
#pragma clang fp contract(on)  // every inlined copy must round identically (commit / getter re-integrate from checkpoints)
    const real g = real(9.81), half = real(0.5);
    const real qd0 = x[2], qd1 = x[3];
    real s1, c1, s2, c2;
    sincos_shared(x[0], s1, c1);
    sincos_shared(x[1], s2, c2);
    const real s12 = s1 * c2 + c1 * s2;
    // inertia matrix, Coriolis terms, gravity
    const real H00 = 1 + 1 + 1 * 1 * 1 + 2 * 1 * 1 * half * c2;
    const real H01 = 1 + 1 * 1 * half * c2;
    const real H11 = 1;
    const real C00 = -2 * 1 * 1 * half * s2 * qd1;
    const real C01 = -1 * 1 * half * s2 * qd1;
    const real C10 = 1 * 1 * half * s2 * qd0;
    const real G0 = 1 * g * half * s1 + 1 * g * (1 * s1 + half * s12);
    const real G1 = 1 * g * half * s12;
    const real r0 = (real(0.0) - (C00 * qd0 + C01 * qd1)) - G0;
    const real r1 = (u[0] - (C10 * qd0)) - G1;
    const real invdet = recip(H00 * H11 - H01 * H01);  // (1 / det by v_rcp + Newton, boxqp.hpp -- as the shipped acrobot does)
    dx[0] = qd0;
    dx[1] = qd1;
    dx[2] = (H11 * invdet) * r0 + (-H01 * invdet) * r1;
    dx[3] = (-H01 * invdet) * r0 + (H00 * invdet) * r1;
  }
  __device__ __forceinline__ real cost(const real* x, const real* u) const {
#pragma clang fp contract(on)
    (void)x;
    const real Kr = real(0.1);
    return Kr * Kr * (u[0] * u[0]);
  }
  __device__ __forceinline__ real final_cost(const real* x) const {
#pragma clang fp contract(on)
    const real q0 = goal[0] - x[0], q1 = goal[1] - x[1];
    const real qd0 = goal[2] - x[2], qd1 = goal[3] - x[3];
    return Ks * Ks * (q0 * q0 + q1 * q1) + Kd * Kd * (qd0 * qd0 + qd1 * qd1);
  }
  // (no analytic_record: this model offers finite differences only, like every Model of the reference)
};
