// A user's device twin that is neither small nor linear-quadratic (ILQR_MODEL_USER, NX = 16, NU = 4): a chain of eight damped pendulums
// coupled by torsion springs, every second one actuated -- trigonometric dynamics, a non-quadratic cost (1 - cos of the angle error), dense
// coupling between neighbours.  It is written the way a user writes a Model (include/model.h:6-21): plain loops over its own parameters,
// no hand-written derivatives, so it runs where ANY twin of n > 8 runs: the generic kernels (generic.hpp) -- thread-per-rollout forward
// passes, wavefront-per-knot finite differences that evaluate every perturbed point through dynamics() / cost() / final_cost() exactly as
// src/derivatives.cpp does (2 (n + m) = 40 Euler maps and 880 cost evaluations per knot), the matrix-core backward pass k_backward_w3.
// Nothing of the LQ twin's declared structure (k_derivatives_lq, the record-free exact route) applies to it: this is the model
// bench.py --extra-configs uses to measure the generic path on a model that cannot take those shortcuts.
// The CPU twin the parity tests compare it with: oracle/orc_models.inc, chain_* (term for term the same expressions).
//   x = (theta_0 .. theta_7, omega_0 .. omega_7)
//   user_params[8]: g/l, damping, coupling, w_theta, w_omega, w_u, final scale, target angle
template <class real_>
struct UserModelT {
  using real = real_;
  static constexpr int NL = 8;             // links
  static constexpr int NX = 2 * NL, NU = NL / 2;
  real u_min[NU], u_max[NU];
  real gl, damp, kc, wq, ww, wu, wf, target;

  void set_params(const double* p, int n) {
    const double d[8] = {9.81, 0.1, 2.0, 10.0, 1.0, 0.1, 50.0, 0.0};
    gl = (real)(n >= 8 ? p[0] : d[0]);
    damp = (real)(n >= 8 ? p[1] : d[1]);
    kc = (real)(n >= 8 ? p[2] : d[2]);
    wq = (real)(n >= 8 ? p[3] : d[3]);
    ww = (real)(n >= 8 ? p[4] : d[4]);
    wu = (real)(n >= 8 ? p[5] : d[5]);
    wf = (real)(n >= 8 ? p[6] : d[6]);
    target = (real)(n >= 8 ? p[7] : d[7]);
  }
  __device__ void dynamics(const real* x, const real* u, real* dx) const {
    for (int i = 0; i < NL; i++) dx[i] = x[NL + i];
    for (int i = 0; i < NL; i++) {
      const real th = x[i];
      real lap = 0;
      if (i > 0) lap += x[i - 1] - th;
      if (i < NL - 1) lap += x[i + 1] - th;
      real s, c;
      sincos_shared(th, s, c);
      real acc = -gl * s;
      acc -= damp * x[NL + i];
      acc += kc * lap;
      if ((i & 1) == 0) acc += u[i >> 1];
      dx[NL + i] = acc;
    }
  }
  __device__ real state_cost(const real* x) const {
    real sum = 0;
    for (int i = 0; i < NL; i++) {
      real s, c;
      sincos_shared(x[i] - target, s, c);
      sum += wq * (real(1.0) - c);
      sum += (real(0.5) * ww) * (x[NL + i] * x[NL + i]);
    }
    return sum;
  }
  __device__ real cost(const real* x, const real* u) const {
    real sum = state_cost(x);
    for (int j = 0; j < NU; j++) sum += (real(0.5) * wu) * (u[j] * u[j]);
    return sum;
  }
  __device__ real final_cost(const real* x) const { return wf * state_cost(x); }
};
