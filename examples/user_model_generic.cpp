// A user's Model with dimensions of its own (n = 6, m = 2) solved three ways through the facade, same problems:
//   1. its device twin (examples/user_model_linear6.hpp, compiled into this build of the library: ILQR_MODEL_USER, generic kernels),
//   2. no twin at all: the host virtuals evaluated by the facade (ILQR_MODEL_HOST: rollouts and finite differences on the host,
//      backward pass / box-QP / accept logic on the device),
//   3. the shipped LQ twin.
// The three agree to the finite differences' rounding (1e-6 on costs after three iterations).
//   g++ -std=c++14 -I include examples/user_model_generic.cpp ilqr_amd/lib/libilqr_amd_user_linear6.so -L/opt/rocm/lib -lamdhip64
#include <cmath>
#include <cstdio>
#include <vector>

#include "ilqr_amd.hpp"

using namespace ilqr_amd;

class MyLinear6 : public LinearQuadratic {
 public:
  MyLinear6(int route, const std::vector<double>& A, const std::vector<double>& B, const std::vector<double>& Q, const std::vector<double>& R,
            const std::vector<double>& Qf, double lim)
      : LinearQuadratic(6, 2, A, B, Q, R, Qf, -lim, lim), route_(route) {
    for (const auto* v : {&A, &B, &Q, &R, &Qf}) params_.insert(params_.end(), v->begin(), v->end());
  }
  int device_model_id() const override { return route_ == 0 ? ILQR_MODEL_USER : route_ == 1 ? ILQR_MODEL_HOST : ILQR_MODEL_LQ; }
  void fill_device_desc(ilqr_desc& d) const override {
    if (route_ == 0) {
      d.user_params = params_.data();
      d.n_user_params = (int)params_.size();
    } else {
      LinearQuadratic::fill_device_desc(d);
    }
  }

 private:
  int route_;
  std::vector<double> params_;
};

int main() {
  try {
    if (!ilqr_has_user_model()) {
      std::fprintf(stderr, "error: this build of the library carries no user model\n");
      return 3;
    }
    const int n = 6, m = 2, B = 24, T = 50;
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() {  // xorshift in (-1, 1)
      s ^= s << 13;
      s ^= s >> 7;
      s ^= s << 17;
      return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
    };
    std::vector<double> A(n * n), Bm(n * m), Q(n * n, 0.0), R(m * m, 0.0), Qf(n * n, 0.0);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) A[i * n + j] = (i == j ? -1.0 : 0.0) + 0.15 * rnd();
    for (double& v : Bm) v = 0.5 * rnd();
    for (int i = 0; i < n; i++) {
      Q[i * n + i] = 1.0;
      Qf[i * n + i] = 3.0;
      for (int j = 0; j < i; j++) Q[i * n + j] = Q[j * n + i] = 0.1 * rnd();
    }
    R[0] = R[3] = 0.2;
    R[1] = R[2] = 0.02;
    std::vector<double> x0((size_t)B * n), u0((size_t)B * T * m, 0.0);
    for (double& v : x0) v = rnd();
    std::vector<double> cost[3];
    for (int route = 0; route < 3; route++) {
      BatchILQR solver(std::make_shared<MyLinear6>(route, A, Bm, Q, R, Qf, 0.4), B, T, 0.02);
      solver.init_traj(x0, u0);
      solver.iterate(3);
      cost[route] = solver.cost();
    }
    double worst = 0;
    for (int b = 0; b < B; b++)
      for (int r = 1; r < 3; r++) worst = std::fmax(worst, std::fabs(cost[0][b] - cost[r][b]) / std::fabs(cost[0][b]));
    std::printf("user twin / host virtuals / shipped LQ twin: cost[0] %.12g %.12g %.12g, worst relative difference %.3g\n", cost[0][0], cost[1][0],
                cost[2][0], worst);
    std::printf("agree %d\n", worst < 1e-6 ? 1 : 0);
    // ILQR_FLAG_REFERENCE_FIXES (opt-in: the clamped rollout of src/ilqr_core.cpp:327-329, "the right way"): the device twins clamp in their rollout
    // kernels, the host-evaluated route in the facade's host rollouts -- the same problems again, tight limits so that the clamp matters
    std::vector<double> fixed[3], us_host;
    for (int route = 0; route < 3; route++) {
      BatchILQR solver(std::make_shared<MyLinear6>(route, A, Bm, Q, R, Qf, 0.05), B, T, 0.02, 0, ILQR_FLAG_REFERENCE_FIXES);
      solver.init_traj(x0, u0);
      solver.iterate(3);
      fixed[route] = solver.cost();
      if (route == 1) us_host = solver.controls();
    }
    double worst_fixed = 0, umax = 0;
    for (int b = 0; b < B; b++)
      for (int r = 1; r < 3; r++) worst_fixed = std::fmax(worst_fixed, std::fabs(fixed[0][b] - fixed[r][b]) / std::fabs(fixed[0][b]));
    for (double v : us_host) umax = std::fmax(umax, std::fabs(v));
    std::printf("with ILQR_FLAG_REFERENCE_FIXES: cost[0] %.12g %.12g %.12g, worst relative difference %.3g, max |u| of the host-evaluated route %.6g (limit 0.05)\n",
                fixed[0][0], fixed[1][0], fixed[2][0], worst_fixed, umax);
    const bool fixes_ok = worst_fixed < 1e-6 && umax <= 0.05;
    std::printf("fixes agree %d\n", fixes_ok ? 1 : 0);
    return (worst < 1e-6 && fixes_ok) ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
