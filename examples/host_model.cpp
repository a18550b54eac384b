// A Model that exists only as host code -- a user's subclass of the reference's plugin interface
// (include/model.h:6-21), compiled unchanged -- solved through the facade: its rollouts and finite
// differences call the virtuals below on the host, the backward pass / box-QPs / accept logic run
// on the GPU (ILQR_MODEL_HOST handle).  For comparison the same problem is solved with the shipped
// device twin of the same model; both take the same iterations.
//
//   hipcc/g++ -std=c++14 -I include examples/host_model.cpp -L ilqr_amd/lib -lilqr_amd ...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "ilqr_amd.hpp"

using namespace ilqr_amd;

// the double integrator of include/double_integrator.h, written by "the user": no device_model_id()
class MyIntegrator : public Model {
 public:
  explicit MyIntegrator(const VectorXd& xd) : goal(xd) {
    x_dims = 4;
    u_dims = 2;
    u_min = VectorXd(2);
    u_max = VectorXd(2);
    for (int j = 0; j < 2; j++) {
      u_min(j) = -0.5;
      u_max(j) = 0.5;
    }
  }
  virtual VectorXd dynamics(const VectorXd& x, const VectorXd& u) override {
    VectorXd dx(4);
    dx(0) = x(2);
    dx(1) = x(3);
    dx(2) = u(0);
    dx(3) = u(1);
    return dx;
  }
  virtual double cost(const VectorXd& x, const VectorXd& u) override { return quad(x, 1.0) + (u(0) * u(0) + u(1) * u(1)); }
  virtual double final_cost(const VectorXd& x) override { return quad(x, 10.0); }

 private:
  double quad(const VectorXd& x, double scale) const {  // (scale*Hx*d).dot(d), Hx = diag(1,1,.2,.2)
    const double hx[4] = {1, 1, 0.2, 0.2};
    double d[4], r[4];
    for (int i = 0; i < 4; i++) {
      d[i] = goal(i) - x(i);
      r[i] = scale * (hx[i] * d[i]);
    }
    return (r[0] * d[0] + r[2] * d[2]) + (r[1] * d[1] + r[3] * d[3]);
  }
  VectorXd goal;
};

int main(int argc, char** argv) {
  const int iters = (argc > 1) ? std::atoi(argv[1]) : 0;  // 0 = solve to termination
  try {
    const int T = 99;
    VectorXd x0(4), goal(4);
    const double x0v[4] = {-1, 0, 0, -0.2}, gv[4] = {1, 0.5, 0, 0};
    for (int i = 0; i < 4; i++) {
      x0(i) = x0v[i];
      goal(i) = gv[i];
    }
    VecOfVecXd u0(T, VectorXd(2));
    for (int t = 0; t < T; t++) u0[t](0) = u0[t](1) = 0;

    double cost[2], c0[2];
    int its[2], st[2];
    VecOfVecXd xs[2], us[2];
    for (int which = 0; which < 2; which++) {
      Model* m = which == 0 ? static_cast<Model*>(new MyIntegrator(goal)) : static_cast<Model*>(new DoubleIntegrator(goal));
      iLQR solver(m, 0.02);  // takes ownership like the reference (ilqr.h:31)
      solver.verbose = false;
      solver.write_csv = false;
      c0[which] = solver.init_traj(x0, u0);
      if (iters > 0)
        for (int i = 0; i < iters && solver.status() == ILQR_RUNNING; i++) solver.step();
      else
        solver.generate_trajectory();
      cost[which] = solver.cost();
      its[which] = solver.iterations();
      st[which] = solver.status();
      xs[which] = solver.states();
      us[which] = solver.controls();
    }
    if (argc > 2) {  // a batch of host-evaluated problems, model evaluated by 1 and by N host threads (-fopenmp)
      const int threads = std::atoi(argv[2]), B = 6, its = iters > 0 ? iters : 4;
      std::vector<double> bx0((size_t)B * 4), bu0((size_t)B * T * 2, 0.0), res[2];
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 4; i++) bx0[(size_t)b * 4 + i] = x0v[i] + 0.1 * b * (i + 1);
      for (int which = 0; which < 2; which++) {
        BatchILQR batch(std::make_shared<MyIntegrator>(goal), B, T, 0.02);
        batch.set_host_threads(which == 0 ? 1 : threads);
        batch.init_traj(bx0, bu0);
        batch.iterate(its);
        res[which] = batch.states();
        const std::vector<double> c = batch.cost();
        res[which].insert(res[which].end(), c.begin(), c.end());
      }
      double d = 0;
      for (size_t e = 0; e < res[0].size(); e++) d = std::max(d, std::fabs(res[0][e] - res[1][e]));
      std::printf("host_threads  %d max_abs_diff %.3e\n", threads, d);
    }
    double dx = 0, du = 0;
    for (int t = 0; t <= T; t++)
      for (int i = 0; i < 4; i++) dx = std::max(dx, std::fabs(xs[0][t](i) - xs[1][t](i)));
    for (int t = 0; t < T; t++)
      for (int j = 0; j < 2; j++) du = std::max(du, std::fabs(us[0][t](j) - us[1][t](j)));
    std::printf("host_model   initial %.12g cost %.12g iterations %d status %d\n", c0[0], cost[0], its[0], st[0]);
    std::printf("device_twin  initial %.12g cost %.12g iterations %d status %d\n", c0[1], cost[1], its[1], st[1]);
    std::printf("max_abs_diff xs %.3e us %.3e\n", dx, du);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
