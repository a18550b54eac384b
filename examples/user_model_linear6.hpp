// A user's device twin with dimensions of its own (ILQR_MODEL_USER, any NX <= 32, NU <= 16): a six-state, two-control
// linear-quadratic model -- xdot = A x + B u, cost 0.5 (x'Qx + u'Ru), final cost 0.5 x'Qf x -- written the way a user writes a
// Model (include/model.h:6-21): plain loops, its matrices as plain data.  Compiled into a build of the library with
// -DILQR_USER_MODEL_HEADER (ilqr_amd._build.build_user); not 4 states, so it runs in the generic kernels (generic.hpp):
// thread-per-rollout forward passes, wavefront-per-knot finite differences that evaluate every perturbed point through
// dynamics() / cost() / final_cost() exactly as src/derivatives.cpp does, the matrix-core backward pass.
//   user_params: A [6][6], B [6][2], Q [6][6], R [2][2], Qf [6][6], row-major (124 doubles), optionally followed by wb: the weight of
//   a soft penalty wb sum_j (u_j / u_max_j)^2 on the controls -- a cost that reads the model's own limits, as a Model subclass may
//   (include/model.h:17: u_min / u_max are public members set by the subclass)
template <class real_>
struct UserModelT {
  using real = real_;
  static constexpr int NX = 6, NU = 2;
  real u_min[NU], u_max[NU];
  real A[NX][NX], B[NX][NU], Q[NX][NX], R[NU][NU], Qf[NX][NX];
  real wb;

  void set_params(const double* p, int n) {
    const int need = 3 * NX * NX + NX * NU + NU * NU;
    for (int i = 0; i < NX; i++)
      for (int j = 0; j < NX; j++) {
        A[i][j] = (n >= need) ? (real)p[i * NX + j] : real(0);
        Q[i][j] = (n >= need) ? (real)p[NX * NX + NX * NU + i * NX + j] : real(i == j);
        Qf[i][j] = (n >= need) ? (real)p[2 * NX * NX + NX * NU + NU * NU + i * NX + j] : real(i == j);
      }
    for (int i = 0; i < NX; i++)
      for (int j = 0; j < NU; j++) B[i][j] = (n >= need) ? (real)p[NX * NX + i * NU + j] : real(0);
    for (int i = 0; i < NU; i++)
      for (int j = 0; j < NU; j++) R[i][j] = (n >= need) ? (real)p[2 * NX * NX + NX * NU + i * NU + j] : real(i == j);
    wb = (n > need) ? (real)p[need] : real(0);
  }
  __device__ void dynamics(const real* x, const real* u, real* dx) const {
    for (int i = 0; i < NX; i++) {
      real acc = 0;
      for (int j = 0; j < NX; j++) acc += A[i][j] * x[j];
      for (int j = 0; j < NU; j++) acc += B[i][j] * u[j];
      dx[i] = acc;
    }
  }
  template <int N>
  static __device__ real quad(const real (*M)[N], const real* v) {
    real s = 0;
    for (int i = 0; i < N; i++) {
      real r = 0;
      for (int j = 0; j < N; j++) r += M[i][j] * v[j];
      s += v[i] * r;
    }
    return s;
  }
  __device__ real cost(const real* x, const real* u) const {
    real c = real(0.5) * (quad<NX>(Q, x) + quad<NU>(R, u));
    if (wb != real(0))
      for (int j = 0; j < NU; j++) c += wb * (u[j] / u_max[j]) * (u[j] / u_max[j]);
    return c;
  }
  __device__ real final_cost(const real* x) const { return real(0.5) * quad<NX>(Qf, x); }
  // optional: exact derivatives (ILQR_FLAG_ANALYTIC_DERIVATIVES).  One thread writes the whole record: fx | fu | cx | cxx | cxu | cu | cuu,
  // matrices column-major, with the conventions of src/derivatives.cpp at the last knot (fx = fu = 0, cx / cxx from final_cost, cu = 0,
  // cuu from cost(x_T, 0), cxu = 0).
  __device__ void analytic_record(const real* x, const real* u, real dt, bool last, real* rec) const {
    real* fx = rec;
    real* fu = fx + NX * NX;
    real* cx = fu + NX * NU;
    real* cxx = cx + NX;
    real* cxu = cxx + NX * NX;
    real* cu = cxu + NX * NU;
    real* cuu = cu + NU;
    const real(*W)[NX] = last ? Qf : Q;
    for (int c = 0; c < NX; c++)
      for (int r = 0; r < NX; r++) {
        fx[r + NX * c] = last ? real(0) : real(r == c) + dt * A[r][c];
        cxx[r + NX * c] = real(0.5) * (W[r][c] + W[c][r]);
      }
    for (int c = 0; c < NU; c++)
      for (int r = 0; r < NX; r++) {
        fu[r + NX * c] = last ? real(0) : dt * B[r][c];
        cxu[r + NX * c] = real(0);
      }
    for (int i = 0; i < NX; i++) {
      real acc = 0;
      for (int j = 0; j < NX; j++) acc += real(0.5) * (W[i][j] + W[j][i]) * x[j];
      cx[i] = acc;
    }
    for (int c = 0; c < NU; c++)
      for (int r = 0; r < NU; r++) cuu[r + NU * c] = real(0.5) * (R[r][c] + R[c][r]) + ((r == c && wb != real(0)) ? real(2) * wb / (u_max[r] * u_max[r]) : real(0));
    for (int i = 0; i < NU; i++) {
      real acc = 0;
      if (!last) {
        for (int j = 0; j < NU; j++) acc += real(0.5) * (R[i][j] + R[j][i]) * u[j];
        if (wb != real(0)) acc += real(2) * wb * u[i] / (u_max[i] * u_max[i]);
      }
      cu[i] = acc;
    }
  }
};
