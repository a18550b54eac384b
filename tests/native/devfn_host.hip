// Host-side instantiation of the product's device functions (boxqp.hpp is __host__ __device__)
// so their logic can be unit-tested on a machine without a GPU.  TEST INFRASTRUCTURE.
#include "../../ilqr_amd/csrc/boxqp.hpp"
using namespace ilqr;

template <int M>
static int run(const double* Q, const double* c, const double* x0, const double* lo, const double* hi, double* x,
               int* vfree, double* R, int* nfR) {
  BoxQPResult<M> r;
  box_qp<M>(Q, c, x0, lo, hi, r);
  for (int i = 0; i < M; i++) {
    x[i] = r.x[i];
    vfree[i] = r.v_free[i];
  }
  for (int e = 0; e < M * M; e++) R[e] = r.R[e];
  *nfR = r.nfR;
  return r.result;
}

extern "C" int devfn_box_qp(int m, const double* Q, const double* c, const double* x0, const double* lo,
                            const double* hi, double* x, int* vfree, double* R, int* nfR) {
  switch (m) {
    case 1: return run<1>(Q, c, x0, lo, hi, x, vfree, R, nfR);
    case 2: return run<2>(Q, c, x0, lo, hi, x, vfree, R, nfR);
    case 3: return run<3>(Q, c, x0, lo, hi, x, vfree, R, nfR);
    case 4: return run<4>(Q, c, x0, lo, hi, x, vfree, R, nfR);
    default: return -100;
  }
}

extern "C" int devfn_box_qp_scalar(double Q, double c, double x0, double lo, double hi, double* x, int* fr, double* minv) {
  return box_qp_scalar(Q, c, x0, lo, hi, *x, *fr, *minv);
}
extern "C" int devfn_box_qp_scalar_fast(double Q, double c, double x0, double lo, double hi, double* x, int* fr, double* minv) {
  return box_qp_scalar_fast(Q, c, x0, lo, hi, *x, *fr, *minv);
}

// ---- the same device functions instantiated for float (the product's fp32 mode) ----
template <int M>
static int run_f(const float* Q, const float* c, const float* x0, const float* lo, const float* hi, float* x, int* vfree) {
  BoxQPResult<M, float> r;
  box_qp<M>(Q, c, x0, lo, hi, r);
  for (int i = 0; i < M; i++) {
    x[i] = r.x[i];
    vfree[i] = r.v_free[i];
  }
  return r.result;
}
extern "C" int devfn_box_qp_f32(int m, const float* Q, const float* c, const float* x0, const float* lo, const float* hi, float* x,
                                int* vfree) {
  switch (m) {
    case 1: return run_f<1>(Q, c, x0, lo, hi, x, vfree);
    case 2: return run_f<2>(Q, c, x0, lo, hi, x, vfree);
    case 3: return run_f<3>(Q, c, x0, lo, hi, x, vfree);
    case 4: return run_f<4>(Q, c, x0, lo, hi, x, vfree);
    default: return -100;
  }
}
extern "C" int devfn_box_qp_scalar_f32(float Q, float c, float x0, float lo, float hi, float* x, int* fr, float* minv) {
  return box_qp_scalar(Q, c, x0, lo, hi, *x, *fr, *minv);
}
extern "C" int devfn_box_qp_scalar_fast_f32(float Q, float c, float x0, float lo, float hi, float* x, int* fr, float* minv) {
  return box_qp_scalar_fast(Q, c, x0, lo, hi, *x, *fr, *minv);
}

// sqrt(gn2) < minGrad without the square root (boxqp.hpp): the product's test next to the literal one
extern "C" int devfn_grad_norm_below_min(double gn2) { return grad_norm_below_min(gn2) ? 1 : 0; }
extern "C" int devfn_grad_norm_below_min_f32(float gn2) { return grad_norm_below_min(gn2) ? 1 : 0; }

// m = 2 scalarised solver (boxqp.hpp: box_qp2), double and float
extern "C" int devfn_box_qp2(const double* Q, const double* c, const double* x0, const double* lo, const double* hi, double* x, int* vfree,
                             double* minv3, int* nfR, int detect_indefinite) {
  BoxQP2Result<double> r;
  box_qp2(Q, c, x0, lo, hi, r, detect_indefinite != 0);
  x[0] = r.x[0]; x[1] = r.x[1];
  vfree[0] = r.free0; vfree[1] = r.free1;
  minv3[0] = r.m00; minv3[1] = r.m01; minv3[2] = r.m11;
  *nfR = r.nfR;
  return r.result;
}
// the loop box_qp2 falls back to (and whose exits its straight-line part reproduces)
extern "C" int devfn_box_qp2_loop(const double* Q, const double* c, const double* x0, const double* lo, const double* hi, double* x, int* vfree,
                                  double* minv3, int* nfR, int detect_indefinite) {
  BoxQP2Result<double> r;
  box_qp2_loop(Q, c, x0, lo, hi, r, detect_indefinite != 0);
  x[0] = r.x[0]; x[1] = r.x[1];
  vfree[0] = r.free0; vfree[1] = r.free1;
  minv3[0] = r.m00; minv3[1] = r.m01; minv3[2] = r.m11;
  *nfR = r.nfR;
  return r.result;
}
extern "C" int devfn_box_qp2_f32(const float* Q, const float* c, const float* x0, const float* lo, const float* hi, float* x, int* vfree) {
  BoxQP2Result<float> r;
  box_qp2(Q, c, x0, lo, hi, r);
  x[0] = r.x[0]; x[1] = r.x[1];
  vfree[0] = r.free0; vfree[1] = r.free1;
  return r.result;
}

// qp1_finish (the reference's result code) against qp1_finish_ok (what the kernels use: the two predicates): returns 0 when they agree
extern "C" int devfn_qp1_finish_flavours_agree(double Q, double c, double x0, double lo, double hi, int detect_indefinite) {
  QP1StateT<double> q;
  qp1_begin(Q, c, x0, lo, hi, q, detect_indefinite != 0);
  qp1_backtrack_seq(q);
  double xa, xb, ma, mb;
  int fa, fb;
  bool goes_on;
  const int code = qp1_finish(q, xa, fa, ma);
  const bool ok = qp1_finish_ok(q, xb, fb, mb, goes_on);
  if (xa != xb || fa != fb || ma != mb) return 1;
  if (goes_on != (code == kQpGoesOn)) return 2;
  if (!goes_on && ok != (code >= 1)) return 3;
  return 0;
}
