// ilqr_amd.hpp -- C++ host facade over the C ABI (include/ilqr_amd.h).
//
// Mirrors the reference's plugin and solver interfaces so that code written against
// kazuotani14/iLQR keeps compiling against the MI355X engine:
//     class Model   <->  include/model.h:6-21      (same virtuals, same public fields)
//     class iLQR    <->  include/ilqr.h:28-107     (same ctor/ownership, generate_trajectory x3,
//                                                    init_traj, output_to_csv; + solve(), accessors)
//     class BatchILQR    new: B independent problems sharing one model
// Header-only; link with -lilqr_amd.  Vector/matrix types are Eigen's when <Eigen/Core> is on the
// include path (the reference's users have it), otherwise the minimal dense types below.
//
// A Model subclass runs inside the HIP rollout / finite-difference kernels only through a device
// twin: the subclass says which one by overriding device_model_id(), as the shipped
// Acrobot / DoubleIntegrator / LinearQuadratic do.  A Model that exists only as host virtuals
// (device_model_id() == ILQR_MODEL_HOST, the default: a user's subclass compiled unchanged) is
// evaluated where it lives -- its rollouts and finite differences call those virtuals on the host --
// while the backward pass, its box-QPs and the accept / lambda / termination logic run on the
// device (nx <= 32, nu <= 16).  Nothing here works without a GPU: ilqr_create fails loudly.
#ifndef ILQR_AMD_HPP_
#define ILQR_AMD_HPP_

#include <algorithm>
#include <cmath>
#include <initializer_list>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "ilqr_amd.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(ILQR_AMD_NO_EIGEN)
#include <Eigen/Core>
#include <Eigen/StdVector>
#define ILQR_AMD_HAVE_EIGEN 1
#endif
#endif

namespace ilqr_amd {

#ifdef ILQR_AMD_HAVE_EIGEN
using VectorXd = Eigen::VectorXd;
using MatrixXd = Eigen::MatrixXd;
typedef std::vector<VectorXd, Eigen::aligned_allocator<VectorXd>> VecOfVecXd;  // include/common.h:29
typedef std::vector<MatrixXd, Eigen::aligned_allocator<MatrixXd>> VecOfMatXd;  // include/common.h:30
#else
// Minimal stand-ins with the subset of the Eigen API the reference's interface uses.
class VectorXd {
 public:
  VectorXd() {}
  explicit VectorXd(int n) : d_(n, 0.0) {}
  VectorXd(std::initializer_list<double> v) : d_(v) {}
  int size() const { return (int)d_.size(); }
  void resize(int n) { d_.resize(n); }
  void setZero() { std::fill(d_.begin(), d_.end(), 0.0); }
  double& operator()(int i) { return d_[i]; }
  double operator()(int i) const { return d_[i]; }
  double& operator[](int i) { return d_[i]; }
  double operator[](int i) const { return d_[i]; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
  static VectorXd Zero(int n) { return VectorXd(n); }

 private:
  std::vector<double> d_;
};
class MatrixXd {  // column-major like Eigen's default
 public:
  MatrixXd() : r_(0), c_(0) {}
  MatrixXd(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
  int rows() const { return r_; }
  int cols() const { return c_; }
  void resize(int r, int c) { r_ = r; c_ = c; d_.assign((size_t)r * c, 0.0); }
  double& operator()(int i, int j) { return d_[i + (size_t)r_ * j]; }
  double operator()(int i, int j) const { return d_[i + (size_t)r_ * j]; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }

 private:
  int r_, c_;
  std::vector<double> d_;
};
typedef std::vector<VectorXd> VecOfVecXd;
typedef std::vector<MatrixXd> VecOfMatXd;
#endif

// ---------------------------------------------------------------------------------------------
// include/model.h:6-21
// ---------------------------------------------------------------------------------------------
class Model {
 public:
  virtual ~Model() {}
  virtual VectorXd dynamics(const VectorXd& x, const VectorXd& u) = 0;
  virtual double cost(const VectorXd& x, const VectorXd& u) = 0;
  virtual double final_cost(const VectorXd& x) = 0;

  VectorXd integrate_dynamics(const VectorXd& x, const VectorXd& u, double dt) {  // model.h:12-15
    VectorXd dx = dynamics(x, u);
    VectorXd x1(x.size());
    for (int i = 0; i < (int)x.size(); i++) x1(i) = x(i) + dx(i) * dt;
    return x1;
  }

  VectorXd u_min, u_max;
  int x_dims = 0;
  int u_dims = 0;

  // Which device twin evaluates this model inside the HIP kernels (enum ilqr_model_id).
  virtual int device_model_id() const { return ILQR_MODEL_HOST; }
  // Model-specific parameter block handed to the device twin (the goal for DoubleIntegrator).
  virtual const double* device_goal() const { return nullptr; }
  // Everything else a device twin needs in the handle descriptor (the LQ matrices).
  virtual void fill_device_desc(ilqr_desc& d) const { d.goal = device_goal(); }
};

// include/acrobot.h (host evaluation; the kernels use ilqr::AcrobotModel)
class Acrobot : public Model {
 public:
  Acrobot() {
    x_dims = 4;
    u_dims = 1;
    u_min = VectorXd(1);
    u_max = VectorXd(1);
    u_min(0) = -5;  // acrobot.h:37
    u_max(0) = 5;
    goal_[0] = 3.1415;  // acrobot.h:21
    goal_[1] = goal_[2] = goal_[3] = 0;
  }
  VectorXd dynamics(const VectorXd& x, const VectorXd& u) override {  // acrobot.h:72-81
    const double g = 9.81;
    const double c2 = std::cos(x(1)), s2 = std::sin(x(1)), s1 = std::sin(x(0)), s12 = std::sin(x(0) + x(1));
    const double H00 = 3 + c2, H01 = 1 + 0.5 * c2, H11 = 1;
    const double C00 = -s2 * x(3), C01 = -0.5 * s2 * x(3), C10 = 0.5 * s2 * x(2);
    const double G0 = g * 0.5 * s1 + g * (s1 + 0.5 * s12), G1 = g * 0.5 * s12;
    const double r0 = (0.0 - (C00 * x(2) + C01 * x(3))) - G0;
    const double r1 = (u(0) - C10 * x(2)) - G1;
    const double invdet = 1.0 / (H00 * H11 - H01 * H01);
    VectorXd dx(4);
    dx(0) = x(2);
    dx(1) = x(3);
    dx(2) = (H11 * invdet) * r0 + (-H01 * invdet) * r1;
    dx(3) = (-H01 * invdet) * r0 + (H00 * invdet) * r1;
    return dx;
  }
  double cost(const VectorXd&, const VectorXd& u) override { return 0.1 * 0.1 * (u(0) * u(0)); }  // :83-92
  double final_cost(const VectorXd& x) override {  // :94-100
    double q = 0, qd = 0;
    for (int i = 0; i < 2; i++) q += (goal_[i] - x(i)) * (goal_[i] - x(i));
    for (int i = 2; i < 4; i++) qd += (goal_[i] - x(i)) * (goal_[i] - x(i));
    return 400.0 * q + 400.0 * qd;
  }
  int device_model_id() const override { return ILQR_MODEL_ACROBOT; }

 private:
  double goal_[4];
};

// include/double_integrator.h
class DoubleIntegrator : public Model {
 public:
  explicit DoubleIntegrator(const VectorXd& xd) {
    x_dims = 4;
    u_dims = 2;
    u_min = VectorXd(2);
    u_max = VectorXd(2);
    for (int j = 0; j < 2; j++) {
      u_min(j) = -0.5;  // double_integrator.h:25-26
      u_max(j) = 0.5;
    }
    for (int i = 0; i < 4; i++) goal_[i] = xd(i);
  }
  VectorXd dynamics(const VectorXd& x, const VectorXd& u) override {  // :29-37
    VectorXd dx(4);
    dx(0) = x(2);
    dx(1) = x(3);
    dx(2) = u(0);
    dx(3) = u(1);
    return dx;
  }
  double cost(const VectorXd& x, const VectorXd& u) override { return quad(x, 1.0) + u(0) * u(0) + u(1) * u(1); }
  double final_cost(const VectorXd& x) override { return quad(x, 10.0); }
  int device_model_id() const override { return ILQR_MODEL_DOUBLE_INTEGRATOR; }
  const double* device_goal() const override { return goal_; }

 private:
  double quad(const VectorXd& x, double s) const {
    const double hx[4] = {1, 1, 0.2, 0.2};
    double c = 0;
    for (int i = 0; i < 4; i++) c += s * hx[i] * (goal_[i] - x(i)) * (goal_[i] - x(i));
    return c;
  }
  double goal_[4];
};

// The synthetic LQ model of BASELINE.json configs[4] (no counterpart in the reference):
// xdot = A x + B u, cost 0.5 (x'Qx + u'Ru), final cost 0.5 x'Qf x; matrices row-major.
class LinearQuadratic : public Model {
 public:
  LinearQuadratic(int n, int m, const std::vector<double>& A, const std::vector<double>& B, const std::vector<double>& Q,
                  const std::vector<double>& R, const std::vector<double>& Qf, double u_lo, double u_hi)
      : A_(A), B_(B), Q_(Q), R_(R), Qf_(Qf) {
    if (A.size() != (size_t)n * n || B.size() != (size_t)n * m || Q.size() != (size_t)n * n || R.size() != (size_t)m * m ||
        Qf.size() != (size_t)n * n)
      throw std::invalid_argument("LinearQuadratic: A [n][n], B [n][m], Q [n][n], R [m][m], Qf [n][n]");
    x_dims = n;
    u_dims = m;
    u_min = VectorXd(m);
    u_max = VectorXd(m);
    for (int j = 0; j < m; j++) {
      u_min(j) = u_lo;
      u_max(j) = u_hi;
    }
  }
  VectorXd dynamics(const VectorXd& x, const VectorXd& u) override {
    VectorXd dx(x_dims);
    for (int i = 0; i < x_dims; i++) {
      double a = 0;
      for (int j = 0; j < x_dims; j++) a += A_[(size_t)i * x_dims + j] * x(j);
      for (int j = 0; j < u_dims; j++) a += B_[(size_t)i * u_dims + j] * u(j);
      dx(i) = a;
    }
    return dx;
  }
  double cost(const VectorXd& x, const VectorXd& u) override { return 0.5 * (quad(x_dims, Q_, x) + quad(u_dims, R_, u)); }
  double final_cost(const VectorXd& x) override { return 0.5 * quad(x_dims, Qf_, x); }
  int device_model_id() const override { return ILQR_MODEL_LQ; }
  void fill_device_desc(ilqr_desc& d) const override {
    d.lq_A = A_.data();
    d.lq_B = B_.data();
    d.lq_Q = Q_.data();
    d.lq_R = R_.data();
    d.lq_Qf = Qf_.data();
  }

 private:
  static double quad(int n, const std::vector<double>& M, const VectorXd& v) {
    double s = 0;
    for (int i = 0; i < n; i++) {
      double r = 0;
      for (int j = 0; j < n; j++) r += M[(size_t)i * n + j] * v(j);
      s += v(i) * r;
    }
    return s;
  }
  std::vector<double> A_, B_, Q_, R_, Qf_;
};

inline void check(int rc, const char* what) {
  if (rc != ILQR_OK) throw std::runtime_error(std::string(what) + ": " + ilqr_last_error());
}

// ---------------------------------------------------------------------------------------------
// B independent problems, one shared model.  Arrays in the canonical layouts of ilqr_amd.h.
// ---------------------------------------------------------------------------------------------
class BatchILQR {
 public:
  BatchILQR(std::shared_ptr<Model> model, int B, int T, double dt, int device = 0, int flags = 0)
      : model_(model), B_(B), T_(T), n_(model->x_dims), m_(model->u_dims), dt_(dt), h_(nullptr), flags_(flags) {
    // A Model without a device twin exists only as host virtuals, which no kernel can call: its
    // rollouts and finite differences are evaluated HERE by calling those virtuals (that is the
    // plugin, not a substitute for a kernel), and everything that does not need the model --
    // the backward pass with its box-QPs, the lambda retries, the accept / lambda schedule /
    // termination logic -- still runs on the device (ILQR_MODEL_HOST handle; nx <= 32, nu <= 16).
    // There is still no path without a GPU.
    host_ = (model->device_model_id() == ILQR_MODEL_HOST);
    std::vector<double> lo(m_), hi(m_);
    for (int j = 0; j < m_; j++) {
      lo[j] = model->u_min(j);
      hi[j] = model->u_max(j);
    }
    ilqr_desc d = {};
    d.abi_version = ILQR_AMD_ABI_VERSION;
    d.model = model->device_model_id();
    d.nx = n_;
    d.nu = m_;
    d.T = T;
    d.B = B;
    d.dt = dt;
    d.device = device;
    d.flags = flags;
    d.u_min = lo.data();
    d.u_max = hi.data();
    model->fill_device_desc(d);
    check(ilqr_create(&d, &h_), "ilqr_create");
  }
  ~BatchILQR() { ilqr_destroy(h_); }
  BatchILQR(const BatchILQR&) = delete;
  BatchILQR& operator=(const BatchILQR&) = delete;

  ilqr_batch* handle() { return h_; }
  // Host-evaluated models (no device twin): how many host threads evaluate the model's virtuals -- rollouts and
  // finite differences of different trajectories side by side (the reference's own, commented-out intent was
  // `omp parallel for` over t: src/derivatives.cpp:18,32,84).  Default 1: Model's methods are non-const
  // (include/model.h:8-10) and a user's subclass may keep state; a stateless model (both shipped ones are) can
  // be called from many threads.  Takes effect when the including translation unit is compiled with -fopenmp.
  void set_host_threads(int n) { host_threads_ = n < 1 ? 1 : n; }
  int batch() const { return B_; }
  int horizon() const { return T_; }
  bool host_evaluated() const { return host_; }  // no device twin: rollouts and finite differences call the Model's virtuals on the host

  std::vector<double> init_traj(const std::vector<double>& x0, const std::vector<double>& u0) {
    require(x0.size() == (size_t)B_ * n_ && u0.size() == (size_t)B_ * T_ * m_, "init_traj: x0 [B][nx], u0 [B][T][nu]");
    std::vector<double> cost(B_);
    if (host_) return host_init_traj(x0, u0);
    check(ilqr_init_traj(h_, x0.data(), u0.data(), cost.data()), "ilqr_init_traj");
    return cost;
  }
  void generate_trajectory() {
    if (host_) {
      int running = 0;
      for (;;) {
        check(ilqr_count_running(h_, &running), "ilqr_count_running");
        if (!running) break;
        host_iteration();
      }
      return;
    }
    check(ilqr_generate_trajectory(h_), "ilqr_generate_trajectory");
  }
  void generate_trajectory(const std::vector<double>& x0) {
    if (host_) {
      host_warm_start(x0);
      generate_trajectory();
      return;
    }
    check(ilqr_warm_start(h_, x0.data()), "ilqr_warm_start");
  }
  void generate_trajectory(const std::vector<double>& x0, const std::vector<double>& u0) {
    init_traj(x0, u0);
    generate_trajectory();
  }
  void solve(const std::vector<double>& x0, const std::vector<double>& u0) { generate_trajectory(x0, u0); }
  void iterate(int n) {
    if (host_) {
      for (int i = 0; i < n; i++) host_iteration();
      return;
    }
    check(ilqr_iterate(h_, n), "ilqr_iterate");
  }

  std::vector<double> states() {
    std::vector<double> xs((size_t)B_ * (T_ + 1) * n_);
    check(ilqr_get_trajectory(h_, xs.data(), nullptr), "ilqr_get_trajectory");
    return xs;
  }
  std::vector<double> controls() {
    std::vector<double> us((size_t)B_ * T_ * m_);
    check(ilqr_get_trajectory(h_, nullptr, us.data()), "ilqr_get_trajectory");
    return us;
  }
  std::vector<double> gains_k() {
    std::vector<double> k((size_t)B_ * T_ * m_);
    check(ilqr_get_gains(h_, k.data(), nullptr), "ilqr_get_gains");
    return k;
  }
  std::vector<double> gains_K() {  // [B][T][nu*nx], each K_t column-major nu x nx
    std::vector<double> K((size_t)B_ * T_ * m_ * n_);
    check(ilqr_get_gains(h_, nullptr, K.data()), "ilqr_get_gains");
    return K;
  }
  std::vector<double> cost() {
    std::vector<double> c(B_);
    check(ilqr_get_cost(h_, c.data()), "ilqr_get_cost");
    return c;
  }
  // Every result in one call (ABI 5): the device-to-host copies are enqueued back to back, nothing is waited for until synchronize().
  // The buffers belong to the caller (sizes as the getters above; nullptr = skip); page-lock buffers that are reused across solves once
  // with ilqr_host_register so that the copies are DMA transfers.
  void results_async(double* xs, double* us, double* k, double* K, double* cost) {
    check(ilqr_get_results_async(h_, xs, us, k, K, cost), "ilqr_get_results_async");
  }
  // ... or keep them on the GPU: canonical layouts into caller-owned device memory (the next warm start, ilqr_core.cpp:65-76, of an MPC loop)
  void copy_trajectory_to_device(void* xs_device, void* us_device) { check(ilqr_copy_trajectory_to_device(h_, xs_device, us_device), "ilqr_copy_trajectory_to_device"); }
  void copy_gains_to_device(void* k_device, void* K_device) { check(ilqr_copy_gains_to_device(h_, k_device, K_device), "ilqr_copy_gains_to_device"); }
  void synchronize() { check(ilqr_synchronize(h_), "ilqr_synchronize"); }
  std::vector<int> status() {
    std::vector<int> s(B_);
    check(ilqr_get_status(h_, s.data(), nullptr, nullptr), "ilqr_get_status");
    return s;
  }
  std::vector<int> iterations() {
    std::vector<int> s(B_);
    check(ilqr_get_status(h_, nullptr, s.data(), nullptr), "ilqr_get_status");
    return s;
  }

 private:
  static void require(bool ok, const char* msg) {
    if (!ok) throw std::invalid_argument(msg);  // the reference asserts (ilqr_core.cpp:66,80-82)
  }
  // ---- host-evaluated models -----------------------------------------------------------------
  static constexpr double kEps = 1e-3;  // include/finite_diff.h:9
  VectorXd vec(const double* p, int n) const {
    VectorXd v(n);
    for (int i = 0; i < n; i++) v(i) = p[i];
    return v;
  }
  // iLQR::forward_pass (src/ilqr_core.cpp:305-337) for trajectory b: controls u_in [T][m], optional
  // feedback K [T][m*n] (column-major m x n) around xnom [T+1][n]; returns the cost
  double host_forward(int b, const double* u_in, const double* K, const double* xnom, double* xs_out, double* us_out) {
    VectorXd x = vec(&hx0_[(size_t)b * n_], n_);
    double total = 0;
    for (int t = 0; t < T_; t++) {
      VectorXd u = vec(u_in + (size_t)t * m_, m_);
      if (K) {  // :315-316
        for (int a = 0; a < m_; a++) {
          double acc = 0;
          for (int j = 0; j < n_; j++) acc += K[(size_t)t * m_ * n_ + a + (size_t)m_ * j] * (x(j) - xnom[(size_t)t * n_ + j]);
          u(a) += acc;
        }
      }
      if (flags_ & ILQR_FLAG_REFERENCE_FIXES)  // opt-in: "the right way" of :327-329 -- the clamped control is stored and integrated
        for (int a = 0; a < m_; a++) u(a) = std::min(std::max(u(a), model_->u_min(a)), model_->u_max(a));
      for (int i = 0; i < n_; i++) xs_out[(size_t)t * n_ + i] = x(i);
      for (int a = 0; a < m_; a++) us_out[(size_t)t * m_ + a] = u(a);  // :323 (no clamping unless the flag asks for it)
      total += model_->cost(x, u);
      x = model_->integrate_dynamics(x, u, dt_);
    }
    for (int i = 0; i < n_; i++) xs_out[(size_t)T_ * n_ + i] = x(i);
    return total + model_->final_cost(x);
  }
  // central differences in the reference's own sequence of perturbed points
  template <class F>
  void fd_gradient(const VectorXd& x, F f, double* out) {  // finite_diff.h:22-33
    for (int i = 0; i < (int)x.size(); i++) {
      VectorXd p = x, m = x;
      p(i) += kEps;
      m(i) -= kEps;
      out[i] = (f(p) - f(m)) / (2 * kEps);
    }
  }
  template <class F>
  void fd_hessian(const VectorXd& x, F f, double* out) {  // finite_diff.h:67-86, column-major N x N
    const int N = (int)x.size();
    for (int i = 0; i < N; i++)
      for (int j = i; j < N; j++) {
        VectorXd pp = x, pm = x, mp = x, mm = x;
        pp(i) += kEps;
        pp(j) += kEps;
        pm(i) += kEps;
        pm(j) -= kEps;
        mp(i) -= kEps;
        mp(j) += kEps;
        mm(i) -= kEps;
        mm(j) -= kEps;
        const double v = (f(pp) - f(mp) - f(pm) + f(mm)) / (4 * kEps * kEps);
        out[i + (size_t)N * j] = v;
        out[j + (size_t)N * i] = v;
      }
  }
  // src/derivatives.cpp for trajectory b, all knots, into the canonical arrays of ilqr_set_derivatives
  void host_derivatives(int b) {
    const int n = n_, m = m_, T1 = T_ + 1;
    for (int t = 0; t <= T_; t++) {
      const VectorXd x = vec(&hxs_[((size_t)b * T1 + t) * n], n);
      VectorXd u(m);
      for (int a = 0; a < m; a++) u(a) = (t < T_) ? hus_[((size_t)b * T_ + t) * m + a] : 0.0;
      double* fx = &d_fx_[((size_t)b * T1 + t) * n * n];
      double* fu = &d_fu_[((size_t)b * T1 + t) * n * m];
      double* cx = &d_cx_[((size_t)b * T1 + t) * n];
      double* cu = &d_cu_[((size_t)b * T1 + t) * m];
      double* cxx = &d_cxx_[((size_t)b * T1 + t) * n * n];
      double* cxu = &d_cxu_[((size_t)b * T1 + t) * n * m];
      double* cuu = &d_cuu_[((size_t)b * T1 + t) * m * m];
      if (t < T_) {
        for (int i = 0; i < n; i++) {  // derivatives.cpp:19-25
          VectorXd p = x, q = x;
          p(i) += kEps;
          q(i) -= kEps;
          const VectorXd fp = model_->integrate_dynamics(p, u, dt_), fm = model_->integrate_dynamics(q, u, dt_);
          for (int r = 0; r < n; r++) fx[r + (size_t)n * i] = (fp(r) - fm(r)) / (2 * kEps);
        }
        for (int i = 0; i < m; i++) {
          VectorXd p = u, q = u;
          p(i) += kEps;
          q(i) -= kEps;
          const VectorXd fp = model_->integrate_dynamics(x, p, dt_), fm = model_->integrate_dynamics(x, q, dt_);
          for (int r = 0; r < n; r++) fu[r + (size_t)n * i] = (fp(r) - fm(r)) / (2 * kEps);
        }
        fd_gradient(x, [&](const VectorXd& xx) { return model_->cost(xx, u); }, cx);  // :44-47
        fd_gradient(u, [&](const VectorXd& uu) { return model_->cost(x, uu); }, cu);
        fd_hessian(x, [&](const VectorXd& xx) { return model_->cost(xx, u); }, cxx);  // :76-96
      } else {  // fx[T], fu[T] stay zero; :49-51, :92
        for (int e = 0; e < n * n; e++) fx[e] = 0;
        for (int e = 0; e < n * m; e++) fu[e] = 0;
        fd_gradient(x, [&](const VectorXd& xx) { return model_->final_cost(xx); }, cx);
        for (int a = 0; a < m; a++) cu[a] = 0;
        fd_hessian(x, [&](const VectorXd& xx) { return model_->final_cost(xx); }, cxx);
      }
      fd_hessian(u, [&](const VectorXd& uu) { return model_->cost(x, uu); }, cuu);  // :98-112 (u = 0 at T)
      for (int i = 0; i < n; i++)  // :114-144
        for (int j = 0; j < m; j++) {
          VectorXd px = x, mx = x, pu = u, mu = u;
          px(i) += kEps;
          mx(i) -= kEps;
          pu(j) += kEps;
          mu(j) -= kEps;
          double v;
          if (t < T_)
            v = (model_->cost(px, pu) - model_->cost(mx, pu) - model_->cost(px, mu) + model_->cost(mx, mu)) / (4 * (kEps * kEps));
          else  // the reference's own "this is wrong" formula; the value is never consumed
            v = (model_->final_cost(px) - model_->final_cost(mx) - model_->final_cost(px) + model_->final_cost(mx)) / (4 * (kEps * kEps));
          cxu[i + (size_t)n * j] = v;
        }
    }
  }
  void host_alloc() {
    const size_t B = B_, T1 = T_ + 1, n = n_, m = m_;
    hxs_.assign(B * T1 * n, 0.0);
    hus_.assign(B * T_ * m, 0.0);
    d_fx_.assign(B * T1 * n * n, 0.0);
    d_fu_.assign(B * T1 * n * m, 0.0);
    d_cx_.assign(B * T1 * n, 0.0);
    d_cu_.assign(B * T1 * m, 0.0);
    d_cxx_.assign(B * T1 * n * n, 0.0);
    d_cxu_.assign(B * T1 * n * m, 0.0);
    d_cuu_.assign(B * T1 * m * m, 0.0);
    need_derivs_.assign(B, 1);
  }
  std::vector<double> host_init_traj(const std::vector<double>& x0, const std::vector<double>& u0) {  // ilqr_core.cpp:11-56
    hx0_ = x0;
    host_alloc();
    std::vector<double> cost(B_);
#ifdef _OPENMP
#pragma omp parallel for num_threads(host_threads_) schedule(dynamic) if (host_threads_ > 1)
#endif
    for (int b = 0; b < B_; b++)
      cost[b] = host_forward(b, &u0[(size_t)b * T_ * m_], nullptr, nullptr, &hxs_[(size_t)b * (T_ + 1) * n_], &hus_[(size_t)b * T_ * m_]);
    check(ilqr_reset_state(h_, 0), "ilqr_reset_state");
    check(ilqr_set_trajectory(h_, hx0_.data(), hxs_.data(), hus_.data(), cost.data()), "ilqr_set_trajectory");
    return cost;
  }
  void host_warm_start(const std::vector<double>& x0) {  // ilqr_core.cpp:65-76
    require(!hus_.empty(), "warm start needs a previous solve (assert us.size()>0, ilqr_core.cpp:66)");
    require(x0.size() == (size_t)B_ * n_, "warm start: x0 [B][nx]");
    hx0_ = x0;
    const std::vector<double> K = gains_K();
    std::vector<double> nxs(hxs_.size()), nus(hus_.size()), cost(B_);
    for (int b = 0; b < B_; b++)
      cost[b] = host_forward(b, &hus_[(size_t)b * T_ * m_], &K[(size_t)b * T_ * m_ * n_], &hxs_[(size_t)b * (T_ + 1) * n_],
                             &nxs[(size_t)b * (T_ + 1) * n_], &nus[(size_t)b * T_ * m_]);
    hxs_.swap(nxs);
    hus_.swap(nus);
    need_derivs_.assign(B_, 1);
    check(ilqr_reset_state(h_, 1), "ilqr_reset_state");
    check(ilqr_set_trajectory(h_, hx0_.data(), hxs_.data(), hus_.data(), cost.data()), "ilqr_set_trajectory");
  }
  // one body of the outer loop (src/ilqr_core.cpp:103-288) for every running trajectory
  void host_iteration() {
    require(!hus_.empty(), "iterate before init_traj");
    std::vector<int> st(B_);
    check(ilqr_get_status(h_, st.data(), nullptr, nullptr), "ilqr_get_status");
    bool any = false;
    for (int b = 0; b < B_; b++) any = any || (st[b] == ILQR_RUNNING && need_derivs_[b]);
#ifdef _OPENMP
#pragma omp parallel for num_threads(host_threads_) schedule(dynamic) if (host_threads_ > 1)
#endif
    for (int b = 0; b < B_; b++)
      if (st[b] == ILQR_RUNNING && need_derivs_[b]) {  // STEP 1, :115-120 (flgChange)
        host_derivatives(b);
        need_derivs_[b] = 0;
      }
    if (any)
      check(ilqr_set_derivatives(h_, d_fx_.data(), d_fu_.data(), d_cx_.data(), d_cu_.data(), d_cxx_.data(), d_cxu_.data(), d_cuu_.data()),
            "ilqr_set_derivatives");
    check(ilqr_backward_step(h_), "ilqr_backward_step");  // STEP 2 on the device
    check(ilqr_get_status(h_, st.data(), nullptr, nullptr), "ilqr_get_status");
    const std::vector<double> k = gains_k(), K = gains_K();
    static const double alphas[11] = {1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316, 0.0158, 0.0079, 0.0040, 0.0020, 0.0010};
    std::vector<double> cost_c((size_t)B_ * 11, 0.0);
    struct Scratch {
      std::vector<double> u_try, xs_try, us_try;
    };
    auto roll = [&](int b, double alpha, Scratch& w) {  // :188-190
      w.u_try.resize((size_t)T_ * m_);
      w.xs_try.resize((size_t)(T_ + 1) * n_);
      w.us_try.resize((size_t)T_ * m_);
      for (size_t e = 0; e < w.u_try.size(); e++) w.u_try[e] = hus_[(size_t)b * T_ * m_ + e] + k[(size_t)b * T_ * m_ + e] * alpha;
      return host_forward(b, w.u_try.data(), &K[(size_t)b * T_ * m_ * n_], &hxs_[(size_t)b * (T_ + 1) * n_], w.xs_try.data(), w.us_try.data());
    };
#ifdef _OPENMP
#pragma omp parallel for num_threads(host_threads_) schedule(dynamic) if (host_threads_ > 1)
#endif
    for (int b = 0; b < B_; b++)
      if (st[b] == ILQR_RUNNING) {
        Scratch w;
        for (int a = 0; a < 11; a++) cost_c[(size_t)b * 11 + a] = roll(b, alphas[a], w);  // STEP 3 rollouts
      }
    std::vector<int> acc(B_);
    check(ilqr_accept_candidates(h_, cost_c.data(), acc.data()), "ilqr_accept_candidates");  // STEP 3/4 decisions on the device
    bool moved = false;
    for (int b = 0; b < B_; b++) moved = moved || acc[b] >= 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(host_threads_) schedule(dynamic) if (host_threads_ > 1)
#endif
    for (int b = 0; b < B_; b++)
      if (acc[b] >= 0) {  // :210-213: xs, us keep the accepted rollout
        Scratch w;
        roll(b, alphas[acc[b]], w);
        std::copy(w.xs_try.begin(), w.xs_try.end(), hxs_.begin() + (size_t)b * (T_ + 1) * n_);
        std::copy(w.us_try.begin(), w.us_try.end(), hus_.begin() + (size_t)b * T_ * m_);
        need_derivs_[b] = 1;
      }
    if (moved) check(ilqr_set_trajectory(h_, nullptr, hxs_.data(), hus_.data(), nullptr), "ilqr_set_trajectory");
  }

  std::shared_ptr<Model> model_;
  int B_, T_, n_, m_;
  double dt_;
  ilqr_batch* h_;
  int flags_ = 0;
  bool host_ = false;
  int host_threads_ = 1;
  std::vector<double> hx0_, hxs_, hus_, d_fx_, d_fu_, d_cx_, d_cu_, d_cxx_, d_cxu_, d_cuu_;
  std::vector<char> need_derivs_;
};

// ---------------------------------------------------------------------------------------------
// One batch over several devices of a node, ONE process: shard i of the batch is a BatchILQR of its own on
// devices[i], holding the contiguous block [i ceil(B/n), ...) of the problems.  Trajectories never interact, so there is
// no data-path exchange; the one collective of the path is the gather of the per-trajectory costs (ilqr_group_*: RCCL
// ncclAllGather over xGMI between distinct devices, plain copies between shards that share one).  `devices` may name
// a device several times: N logical shards on one GPU give, bit for bit, what N GPUs give (and what one handle of B does).
// A caller of the reference who solves many problems loops over src/run_ilqr.cpp:27-59; this is that loop, sharded.
// ---------------------------------------------------------------------------------------------
class ShardedBatchILQR {
 public:
  ShardedBatchILQR(std::shared_ptr<Model> model, int B, int T, double dt, const std::vector<int>& devices, int flags = 0)
      : B_(B), T_(T), n_(model->x_dims), m_(model->u_dims), group_(nullptr) {
    if (devices.empty() || B < 1) throw std::invalid_argument("ShardedBatchILQR: at least one device and one trajectory");
    const int n = (int)devices.size(), per = (B + n - 1) / n;
    for (int i = 0; i < n && i * per < B; i++) {
      offset_.push_back(i * per);
      shards_.emplace_back(new BatchILQR(model, std::min(per, B - i * per), T, dt, devices[i], flags));
    }
    offset_.push_back(B);
    std::vector<ilqr_batch*> hs;
    for (auto& sh : shards_) hs.push_back(sh->handle());
    check(ilqr_group_create(hs.data(), (int)hs.size(), 0, &group_), "ilqr_group_create");
  }
  ~ShardedBatchILQR() { ilqr_group_destroy(group_); }
  ShardedBatchILQR(const ShardedBatchILQR&) = delete;
  ShardedBatchILQR& operator=(const ShardedBatchILQR&) = delete;

  int shards() const { return (int)shards_.size(); }
  int batch() const { return B_; }
  BatchILQR& shard(int i) { return *shards_[i]; }
  // how the gather travels: ranks of the RCCL communicator (0: the shards share a device and are copied)
  int rccl_ranks() const {
    int r = 0;
    return ilqr_group_uses_rccl(group_, &r) ? r : 0;
  }

  std::vector<double> init_traj(const std::vector<double>& x0, const std::vector<double>& u0) {
    if (x0.size() != (size_t)B_ * n_ || u0.size() != (size_t)B_ * T_ * m_) throw std::invalid_argument("init_traj: x0 [B][nx], u0 [B][T][nu]");
    std::vector<double> cost;
    for (int i = 0; i < shards(); i++) {
      const size_t lo = offset_[i], hi = offset_[i + 1];
      const std::vector<double> c = shards_[i]->init_traj(std::vector<double>(x0.begin() + lo * n_, x0.begin() + hi * n_),
                                                          std::vector<double>(u0.begin() + lo * T_ * m_, u0.begin() + hi * T_ * m_));
      cost.insert(cost.end(), c.begin(), c.end());
    }
    return cost;
  }
  // ilqr_iterate only enqueues on the shard's own stream: the shards' kernels run side by side, the first getter waits
  void iterate(int n_iters) {
    for (auto& sh : shards_) sh->iterate(n_iters);
  }
  // whole solves decide on the host when to stop (and re-pack running trajectories between chunks): one host thread per shard.
  // A host-evaluated Model (no device twin) is shared by all shards and its methods are non-const (include/model.h:8-10: a subclass may
  // keep state), so those shards are solved one after the other -- the same rule that makes BatchILQR::set_host_threads default to 1.
  void generate_trajectory() {
    if (!shards_.empty() && shards_[0]->host_evaluated()) {
      for (auto& sh : shards_) sh->generate_trajectory();
      return;
    }
    each_in_its_own_thread([](BatchILQR& sh) { sh.generate_trajectory(); });
  }
  void generate_trajectory(const std::vector<double>& x0, const std::vector<double>& u0) {
    init_traj(x0, u0);
    generate_trajectory();
  }
  void solve(const std::vector<double>& x0, const std::vector<double>& u0) { generate_trajectory(x0, u0); }

  // the path's one exchange: [B] costs in global order
  std::vector<double> cost() {
    std::vector<double> c(B_);
    check(ilqr_group_gather_costs(group_, c.data()), "ilqr_group_gather_costs");
    return c;
  }
  std::vector<double> states() { return concat<double>([](BatchILQR& sh) { return sh.states(); }); }
  std::vector<double> controls() { return concat<double>([](BatchILQR& sh) { return sh.controls(); }); }
  std::vector<double> gains_k() { return concat<double>([](BatchILQR& sh) { return sh.gains_k(); }); }
  std::vector<double> gains_K() { return concat<double>([](BatchILQR& sh) { return sh.gains_K(); }); }
  std::vector<int> status() { return concat<int>([](BatchILQR& sh) { return sh.status(); }); }
  std::vector<int> iterations() { return concat<int>([](BatchILQR& sh) { return sh.iterations(); }); }

 private:
  template <class V, class F>
  std::vector<V> concat(F get) {
    std::vector<V> all;
    for (auto& sh : shards_) {
      const std::vector<V> part = get(*sh);
      all.insert(all.end(), part.begin(), part.end());
    }
    return all;
  }
  template <class F>
  void each_in_its_own_thread(F f) {
    std::vector<std::thread> th;
    std::vector<std::string> err(shards_.size());
    for (size_t i = 0; i < shards_.size(); i++)
      th.emplace_back([&, i] {
        try {
          f(*shards_[i]);
        } catch (const std::exception& e) {
          err[i] = e.what();
        }
      });
    for (auto& t : th) t.join();
    for (auto& e : err)
      if (!e.empty()) throw std::runtime_error(e);
  }
  int B_, T_, n_, m_;
  std::vector<std::unique_ptr<BatchILQR>> shards_;
  std::vector<int> offset_;
  ilqr_group* group_;
};

// ---------------------------------------------------------------------------------------------
// include/ilqr.h:28-107 -- the single-trajectory solver, B = 1 over the batch engine
// ---------------------------------------------------------------------------------------------
class iLQR {
 public:
  iLQR(Model* p_dyn, double timeDelta) : dt(timeDelta) { model.reset(p_dyn); }  // takes ownership, ilqr.h:31
  iLQR() = default;

  std::shared_ptr<Model> model;  // public in the reference too (ilqr.h:47)
  bool verbose = true;           // SHOWPROGRESS / "Saved iLQR result" prints (ilqr_core.cpp:1)
  bool write_csv = true;         // output_to_csv("ilqr_result.csv") at the end of every solve (:300)

  void generate_trajectory() {  // ilqr_core.cpp:79-302
    if (!engine_) throw std::logic_error("generate_trajectory(): no trajectory initialised (asserts of ilqr_core.cpp:80-82)");
    int iter = 0;
    for (;; iter++) {
      int running = 0;
      check(ilqr_count_running(engine_->handle(), &running), "ilqr_count_running");
      if (!running) break;
      const double cost_before = engine_->cost()[0];
      engine_->iterate(1);
      if (verbose) print_progress(iter, cost_before);
    }
    if (write_csv) output_to_csv("ilqr_result.csv");
  }
  void generate_trajectory(const VectorXd& x_0) {  // warm start, ilqr_core.cpp:65-76
    if (!engine_) throw std::logic_error("warm start needs a previous solve (assert us.size()>0, ilqr_core.cpp:66)");
    std::vector<double> x0(x_0.data(), x_0.data() + x_0.size());
    engine_->generate_trajectory(x0);
    if (write_csv) output_to_csv("ilqr_result.csv");
  }
  void generate_trajectory(const VectorXd& x_0, const VecOfVecXd& u0) {  // fresh start, :59-62
    init_traj(x_0, u0);
    generate_trajectory();
  }
  void solve(const VectorXd& x_0, const VecOfVecXd& u0) { generate_trajectory(x_0, u0); }  // BASELINE.json's name
  // one body of the outer loop (src/ilqr_core.cpp:103-288); generate_trajectory() loops over it
  void step() {
    if (!engine_) throw std::logic_error("step(): no trajectory initialised");
    engine_->iterate(1);
  }

  double init_traj(const VectorXd& x_0, const VecOfVecXd& u_0) {  // ilqr_core.cpp:11-56
    T = (int)u_0.size();
    const int n = model->x_dims, m = model->u_dims;
    if ((int)x_0.size() != n) throw std::invalid_argument("init_traj: x_0 has the wrong size");
    engine_.reset(new BatchILQR(model, 1, T, dt));
    std::vector<double> x0(x_0.data(), x_0.data() + n), u0((size_t)T * m);
    for (int t = 0; t < T; t++)
      for (int j = 0; j < m; j++) u0[(size_t)t * m + j] = u_0[t](j);
    const double c = engine_->init_traj(x0, u0)[0];
    if (verbose) std::printf("Initial cost: %g\n", c);
    return c;
  }

  // ilqr_core.cpp:414-431, byte-compatible (including the one u column too many in the header
  // and the unterminated last row) so plot_results.py-style consumers keep working
  void output_to_csv(const std::string filename) {
    const int n = model->x_dims, m = model->u_dims;
    const std::vector<double> xs = engine_->states(), us = engine_->controls();
    FILE* XU = std::fopen(filename.c_str(), "w");
    if (!XU) throw std::runtime_error("output_to_csv: cannot open " + filename);
    for (int i = 1; i <= n; i++) std::fprintf(XU, "x%d, ", i);
    for (int j = 0; j < m; j++) std::fprintf(XU, "u%d, ", j);
    std::fprintf(XU, "u%d\n", m);
    for (int t = 0; t < T; t++) {
      for (int i = 0; i < n; i++) std::fprintf(XU, "%f, ", xs[(size_t)t * n + i]);
      for (int j = 0; j < m - 1; j++) std::fprintf(XU, "%f, ", us[(size_t)t * m + j]);
      std::fprintf(XU, "%f\n", us[(size_t)t * m + m - 1]);
    }
    for (int i = 0; i < n; i++) std::fprintf(XU, "%f, ", xs[(size_t)T * n + i]);
    std::fclose(XU);
    if (verbose) std::printf("Saved iLQR result to %s\n", filename.c_str());
  }

  // accessors the reference keeps private (tests reach them through FRIEND_TEST, ilqr.h:103-106)
  VecOfVecXd states() const { return unpack(engine_->states(), T + 1, model->x_dims); }
  VecOfVecXd controls() const { return unpack(engine_->controls(), T, model->u_dims); }
  VecOfVecXd gains_k() const { return unpack(engine_->gains_k(), T, model->u_dims); }
  VecOfMatXd gains_K() const {
    const int n = model->x_dims, m = model->u_dims;
    const std::vector<double> K = engine_->gains_K();
    VecOfMatXd out(T);
    for (int t = 0; t < T; t++) {
      out[t] = MatrixXd(m, n);
      for (int j = 0; j < n; j++)
        for (int a = 0; a < m; a++) out[t](a, j) = K[(size_t)t * m * n + a + (size_t)m * j];
    }
    return out;
  }
  double cost() const { return engine_->cost()[0]; }
  int iterations() const { return engine_->iterations()[0]; }
  int status() const { return engine_->status()[0]; }  // enum ilqr_traj_status

 private:
  double dt = 0;
  int T = 0;
  std::unique_ptr<BatchILQR> engine_;

  static VecOfVecXd unpack(const std::vector<double>& a, int S, int E) {
    VecOfVecXd out(S);
    for (int s = 0; s < S; s++) {
      out[s] = VectorXd(E);
      for (int e = 0; e < E; e++) out[s](e) = a[(size_t)s * E + e];
    }
    return out;
  }
  // the table of ilqr_core.cpp:238-246 / :270-273
  void print_progress(int iter, double cost_before) {
    if (iter == 0) std::printf("iteration\tcost\t\treduction\texpect\t\tgrad\t\tlog10(lambda)\n");
    int st = 0, it = 0, al = 0;
    double lam = 0, gn = 0, c = 0, dV[2] = {0, 0};
    ilqr_batch* h = engine_->handle();
    check(ilqr_get_status(h, &st, &it, &al), "ilqr_get_status");
    check(ilqr_get_lambda(h, &lam, nullptr), "ilqr_get_lambda");
    check(ilqr_get_gnorm(h, &gn), "ilqr_get_gnorm");
    check(ilqr_get_cost(h, &c), "ilqr_get_cost");
    check(ilqr_get_dV(h, dV), "ilqr_get_dV");
    static const double alphas[11] = {1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316, 0.0158, 0.0079, 0.0040, 0.0020, 0.0010};
    if (st == ILQR_CONVERGED_GRAD) {
      std::printf("\nSUCCESS: gradient norm < tolGrad\n\n");
      return;
    }
    if (al >= 0) {
      const double a = alphas[al];
      std::printf("%-12d\t%-12.3g\t%-12.3g\t%-12.3g\t%-12.3g\t%-12.1f\n", iter, c, cost_before - c, -a * (dV[0] + a * dV[1]), gn, std::log10(lam));
      if (st == ILQR_CONVERGED_COST) std::printf("\nSUCCESS: cost change < tolFun\n");
    } else {
      std::printf("%-12d\t%-12s\t%-12.3g\t%-12.3g\t%-12.3g\t%-12.1f\n", iter, "NO STEP", 0.0, 0.0, gn, std::log10(lam));
      if (st == ILQR_LAMBDA_MAX) std::printf("\nEXIT: lambda > lambdaMax\n");
    }
  }
};

}  // namespace ilqr_amd

#endif  // ILQR_AMD_HPP_
