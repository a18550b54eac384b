// ref_driver.cpp -- extern "C" shim over the REAL reference code (TEST INFRASTRUCTURE ONLY).
//
// Compiled by oracle/Makefile together with /root/reference/src/boxqp.cpp, straight from the
// reference sources where they lie (nothing is copied into this repo), into
// oracle/_ref/libref_ilqr.so.  It exposes the pieces of the hot path that build from the
// reference's own files with the vendored Eigen and nothing else:
//     src/boxqp.cpp + include/boxqp.h      boxQP, quadclamp_line_search, quadCost, clamp_to_limits
//     include/finite_diff.h                finite_diff_gradient / _jacobian / _hessian
//     include/model.h, acrobot.h, double_integrator.h   the Model plugin + the two shipped models
// class iLQR (include/ilqr.h, src/ilqr_core.cpp, src/derivatives.cpp) is NOT built: ilqr.h:12
// includes gtest/gtest_prod.h, which this image does not have, and writing a stand-in header
// is not allowed -- see DESIGN.md.  The std::bind / lambda bindings below are the ones
// src/derivatives.cpp:20-21,40-42,90-92,109 makes, re-made here so that the reference's own
// finite-difference operators are exercised on the reference's own models.
//
// Only the oracle tests (CPU tests: tests/test_oracle_vs_ref.py, scripts/make_golden.py) load this library.  The BUILT file is git-ignored
// and travels to the GPU box with the snapshot like every other built .so (it is not gpurun-ignored), where nothing loads it; the
// reference's sources never leave /root/reference.
#include <functional>
#include <memory>

#include "boxqp.h"
#include "finite_diff.h"
#include "acrobot.h"
#include "double_integrator.h"

using namespace std::placeholders;

namespace {
std::unique_ptr<Model> make_model(int id, const double* goal) {
  if (id == 0) return std::unique_ptr<Model>(new Acrobot());
  VectorXd g(4);
  for (int i = 0; i < 4; i++) g(i) = goal ? goal[i] : 0.0;
  return std::unique_ptr<Model>(new DoubleIntegrator(g));
}
VectorXd vec(const double* p, int n) { return Eigen::Map<const VectorXd>(p, n); }
MatrixXd mat(const double* p, int r, int c) { return Eigen::Map<const MatrixXd>(p, r, c); }
}  // namespace

extern "C" {

int ref_boxqp(int n, const double* Q, const double* c, const double* x0, const double* lo,
              const double* hi, double* x_opt, int* v_free, double* R_free, int* nfree) {
  boxQPResult res = boxQP(mat(Q, n, n), vec(c, n), vec(x0, n), vec(lo, n), vec(hi, n));
  for (int i = 0; i < n; i++) {
    x_opt[i] = res.x_opt(i);
    v_free[i] = res.v_free(i);
  }
  const int nf = (int)res.R_free.rows();
  *nfree = nf;
  for (int j = 0; j < res.R_free.cols(); j++)
    for (int i = 0; i < nf; i++) R_free[i + nf * j] = res.R_free(i, j);
  return res.result;
}

int ref_line_search(int n, const double* x0, const double* dir, const double* Q, const double* c,
                    const double* lo, const double* hi, double* x_opt, double* v_opt,
                    int* n_steps) {
  lineSearchResult r = quadclamp_line_search(vec(x0, n), vec(dir, n), mat(Q, n, n), vec(c, n),
                                             vec(lo, n), vec(hi, n));
  *n_steps = r.n_steps;
  if (!(r.failed && r.n_steps == 0)) {  // x_opt/v_opt are unset on the early "wrong direction" exit
    for (int i = 0; i < n; i++) x_opt[i] = r.x_opt(i);
    *v_opt = r.v_opt;
  }
  return r.failed ? 1 : 0;
}

double ref_quad_cost(int n, const double* Q, const double* c, const double* x) {
  return quadCost(mat(Q, n, n), vec(c, n), vec(x, n));
}

void ref_clamp(int n, const double* x, const double* lo, const double* hi, double* out) {
  VectorXd r = clamp_to_limits(vec(x, n), vec(lo, n), vec(hi, n));
  for (int i = 0; i < n; i++) out[i] = r(i);
}

void ref_subvec_w_ind(int n, const double* v, const int* ind, double* out, int* nout) {
  Eigen::VectorXi idx(n);
  for (int i = 0; i < n; i++) idx(i) = ind[i];
  VectorXd r = subvec_w_ind(vec(v, n), idx);
  *nout = (int)r.size();
  for (int i = 0; i < r.size(); i++) out[i] = r(i);
}

// Model plugin (include/model.h)
void ref_model_dims(int id, int* nx, int* nu, double* u_min, double* u_max) {
  auto m = make_model(id, nullptr);
  *nx = m->x_dims;
  *nu = m->u_dims;
  for (int i = 0; i < m->u_dims; i++) {
    u_min[i] = m->u_min(i);
    u_max[i] = m->u_max(i);
  }
}
void ref_model_eval(int id, const double* goal, const double* x, const double* u, double dt,
                    double* dx, double* x1, double* cost, double* final_cost) {
  auto m = make_model(id, goal);
  VectorXd xv = vec(x, m->x_dims), uv = vec(u, m->u_dims);
  VectorXd d = m->dynamics(xv, uv);
  VectorXd n1 = m->integrate_dynamics(xv, uv, dt);
  for (int i = 0; i < m->x_dims; i++) {
    dx[i] = d(i);
    x1[i] = n1(i);
  }
  *cost = m->cost(xv, uv);
  *final_cost = m->final_cost(xv);
}

// Derivatives of one knot point exactly as src/derivatives.cpp binds them.
// is_final != 0 reproduces the t == T branches.
void ref_fd_knot(int id, const double* goal, const double* x, const double* u_in, double dt,
                 int is_final, double* fx, double* fu, double* cx, double* cu, double* cxx,
                 double* cuu) {
  auto mp = make_model(id, goal);
  Model* model = mp.get();
  const int n = model->x_dims, mu = model->u_dims;
  VectorXd xt = vec(x, n);
  VectorXd ut = is_final ? VectorXd(VectorXd::Zero(mu)) : vec(u_in, mu);

  if (!is_final) {  // derivatives.cpp:19-25
    std::function<VectorXd(VectorXd)> dyn_x = std::bind(&Model::integrate_dynamics, model, _1, ut, dt);
    std::function<VectorXd(VectorXd)> dyn_u = std::bind(&Model::integrate_dynamics, model, xt, _1, dt);
    MatrixXd Fx = finite_diff_jacobian(dyn_x, xt, n);
    MatrixXd Fu = finite_diff_jacobian(dyn_u, ut, n);
    Eigen::Map<MatrixXd>(fx, n, n) = Fx;
    Eigen::Map<MatrixXd>(fu, n, mu) = Fu;
  }
  // derivatives.cpp:40-52
  std::function<double(VectorXd)> cost_x = std::bind(&Model::cost, model, _1, ut);
  std::function<double(VectorXd)> cost_u = std::bind(&Model::cost, model, xt, _1);
  std::function<double(VectorXd)> cost_f = std::bind(&Model::final_cost, model, _1);
  if (!is_final) {
    Eigen::Map<VectorXd>(cx, n) = finite_diff_gradient(cost_x, xt);
    Eigen::Map<VectorXd>(cu, mu) = finite_diff_gradient(cost_u, ut);
  } else {
    Eigen::Map<VectorXd>(cx, n) = finite_diff_gradient(cost_f, xt);
    Eigen::Map<VectorXd>(cu, mu).setZero();
  }
  // derivatives.cpp:89-94, 109-110
  MatrixXd Cxx(n, n), Cuu(mu, mu);
  finite_diff_hessian(is_final ? cost_f : cost_x, xt, Cxx);
  finite_diff_hessian(cost_u, ut, Cuu);
  Eigen::Map<MatrixXd>(cxx, n, n) = Cxx;
  Eigen::Map<MatrixXd>(cuu, mu, mu) = Cuu;
}

// Scalar / generic operators for the known answers of test/test_finite_diff.cpp.
double ref_fd_scalar_negquad(double x) {
  std::function<double(double)> f = [](double v) { return -pow(v, 2); };
  return finite_diff_gradient(f, x);
}
void ref_fd_grad_quadvec(const double* x, double* out) {  // x^2 + 5 y^2
  std::function<double(VectorXd)> f = [](VectorXd v) { return v(0) * v(0) + 5 * v(1) * v(1); };
  VectorXd g = finite_diff_gradient(f, vec(x, 2));
  out[0] = g(0);
  out[1] = g(1);
}
void ref_fd_jac_addones(const double* x, double* out) {  // v + 1
  std::function<VectorXd(VectorXd)> f = [](VectorXd v) { return VectorXd(v + VectorXd::Ones(v.size())); };
  MatrixXd J = finite_diff_jacobian(f, vec(x, 2), 2);
  Eigen::Map<MatrixXd>(out, 2, 2) = J;
}

}  // extern "C"
