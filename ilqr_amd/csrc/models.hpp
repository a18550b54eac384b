// models.hpp -- device twins of the reference's Model plugins (include/model.h:6-21).
//
// A device model is a POD with compile-time NX/NU and three inlined members
//     dynamics(x,u,dx)   cost(x,u)   final_cost(x)
// i.e. exactly the three virtuals of the reference interface, resolved statically so the
// rollout and finite-difference kernels can inline them (a GPU cannot call host virtuals).
#pragma once
#include <type_traits>
#include <utility>

#include "boxqp.hpp"
#include "common.hpp"

namespace ilqr {

// sin and cos of one argument with ONE shared range reduction: ~35 fp64 instructions instead
// of the ~2 x 100 of two library calls.  The rollout and finite-difference kernels are bound by
// fp64 VALU issue, and four libm trig calls per dynamics evaluation were 60 % of it.
//   reduction: j = rint(x 2/pi), r = x - j pi/2 in three FMAs (pi/2 split in three doubles)
//   kernels  : the classic minimax polynomials on [-pi/4, pi/4] (fdlibm k_sin / k_cos)
// Absolute error about 1.2e-16 for |x| <= 1e5 (measured against long double on 2e7 samples).
// Deliberately branch-free: a library fallback for huge arguments put ~20 taken branches into
// every rollout step.  NaN/Inf propagate to NaN as in libm; a finite |x| beyond ~1e9 (a rollout
// that has already diverged -- the line search rejects it on cost) loses accuracy gracefully.
// double -> int32 as the hardware does it (round toward zero, out-of-range values saturate, NaN gives 0)
__device__ __forceinline__ int cvt_i32_saturating(double v) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
// A double constant held in a vector register.  A VOP3 instruction takes one scalar (or literal) operand, so the first FMA of a
// Horner chain -- two constants -- needs one of them in a VGPR, and under the scalar-register pressure of the persistent kernels
// the compiler re-materialised those and the reduction's in every rollout step and finite-difference point (18 s_mov_b32 +
// 2 v_mov_b64 of the 185 instructions of a rollout step; an s_mov costs a wavefront that has its SIMD to itself the same
// 5 cycles as a v_fma_f64, scripts/ubench/operands.hip).  Opaque to the optimiser: the value stays in its register.  Same bits.
// (All sixteen coefficients this way was measured slower: the compiler then copies each one into the two-address v_fmac.)
__device__ __forceinline__ double vreg_const(double c) {
  asm("" : "+v"(c));
  return c;
}
// The six constants sincos_shared keeps in vector registers.  vreg_const inside a loop body is a copy per trip (the asm's operand is
// read-write: the hoisted constant is copied into a fresh register every time, six v_mov_b64 of a rollout step's ~135 instructions);
// a loop that evaluates the model every trip takes them ONCE, before the loop (TrigConsts k = trig_consts(), WithTrigConsts below).
struct TrigConsts {
  double two_over_pi, pio2_1, pio2_2, pio2_3, s5, c5;
};
__device__ __forceinline__ TrigConsts trig_consts() {
  TrigConsts k;
  k.two_over_pi = vreg_const(6.36619772367581382433e-01);  // 2/pi
  k.pio2_1 = vreg_const(1.57079632679489655800e+00);       // pi/2, leading 53 bits
  k.pio2_2 = vreg_const(6.12323399573676603587e-17);       // next 53 bits
  k.pio2_3 = vreg_const(-1.49738490485916983294e-33);      // and the rest
  k.s5 = vreg_const(-2.50507602534068634195e-08);
  k.c5 = vreg_const(2.08757232129817482790e-09);
  return k;
}
__device__ __forceinline__ void sincos_shared(double x, double& s_out, double& c_out, const TrigConsts& k) {
  // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
  // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
  const double j = __builtin_rint(x * k.two_over_pi);
  double r = __builtin_fma(-j, k.pio2_1, x);
  r = __builtin_fma(-j, k.pio2_2, r);
  r = __builtin_fma(-j, k.pio2_3, r);
  // quadrant q = j mod 4: (sin, cos) = (sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr).  The sine kernel is odd, every operation in it
  // symmetric under negation: its sign goes in with r (quadrants 1 and 2), the cosine kernel's is applied to its result
  // (quadrants 2 and 3), both as XORs of the sign bit; then one swap.  14 instead of 18 instructions; the same bits as
  // selecting among sr, -sr, cr, -cr.  One v_cvt_i32_f64 (the 64-bit conversion this replaced was six instructions: ldexp,
  // floor, two cvt, ...; twenty sincos per finite-difference knot).  |j| >= 2^31 -- |x| beyond 3e9, a rollout long since
  // rejected on cost -- saturates: that is the INSTRUCTION's behaviour; a C++ cast out of range is undefined (poison to the
  // optimiser), hence the asm.
  const unsigned q = (unsigned)cvt_i32_saturating(j);
  r = __hiloint2double(__double2hiint(r) ^ (int)(((q + 1u) << 30) & 0x80000000u), __double2loint(r));
  const double z = r * r;
  // sin(r) = r + r^3 (S1 + z (S2 + ... ))
  const double ps = 8.33333333332248946124e-03 +
                    z * (-1.98412698298579493134e-04 +
                         z * (2.75573137070700676789e-06 + z * (k.s5 + z * 1.58969099521155010221e-10)));
  const double sr = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
  // cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ... )), summed so that the 1 - z/2 rounding is compensated
  const double pc = z * (4.16666666666666019037e-02 +
                         z * (-1.38888888888741095749e-03 +
                              z * (2.48015872894767294178e-05 +
                                   z * (-2.75573143513906633035e-07 +
                                        z * (k.c5 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  double cr = w + (((1.0 - w) - hz) + z * pc);
  cr = __hiloint2double(__double2hiint(cr) ^ (int)((q << 30) & 0x80000000u), __double2loint(cr));
  s_out = (q & 1u) ? cr : sr;
  c_out = (q & 1u) ? sr : cr;
}

__device__ __forceinline__ void sincos_shared(double x, double& s_out, double& c_out) { sincos_shared(x, s_out, c_out, trig_consts()); }
struct NoTrigConsts {};  // (the float kernels' constants are literals of VOP2 / VOP3 instructions)
__device__ __forceinline__ void sincos_shared(float x, float& s_out, float& c_out, const NoTrigConsts&);

// fp32 flavour of the above.  The range reduction runs in DOUBLE (conversion, product, rint, one FMA against a 53-bit pi/2:
// the same six instructions as a three-term float Cody-Waite reduction, whose exactness ends at |x| ~ 3.2e3): r = x - j pi/2
// is good to 1e-16 |j|, i.e. float-exact for every |x| a float rollout can reach before the line search rejects it on cost.
// Then the classic single-precision minimax kernels on [-pi/4, pi/4]: ~1 ulp.
__device__ __forceinline__ void sincos_shared(float x, float& s_out, float& c_out) {
  // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
  // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
  const double xd = (double)x;
  const double j = __builtin_rint(xd * 6.36619772367581382433e-01);  // 2/pi
  const unsigned q = (unsigned)cvt_i32_saturating(j);  // (|j| >= 2^31 saturates: see the double version, and for the signs)
  const float r = __uint_as_float(__float_as_uint((float)__builtin_fma(-j, 1.57079632679489655800e+00, xd)) ^ (((q + 1u) << 30) & 0x80000000u));
  const float z = r * r;
  const float sr = r + (z * r) * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
  const float cr = __uint_as_float(__float_as_uint((1.0f - 0.5f * z) + (z * z) * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f))) ^
                                  ((q << 30) & 0x80000000u));
  s_out = (q & 1u) ? cr : sr;
  c_out = (q & 1u) ? sr : cr;
}

__device__ __forceinline__ void sincos_shared(float x, float& s_out, float& c_out, const NoTrigConsts&) { sincos_shared(x, s_out, c_out); }
template <class real> struct TrigConstsOf { using type = NoTrigConsts; static __device__ __forceinline__ type get() { return type(); } };
template <> struct TrigConstsOf<double> { using type = TrigConsts; static __device__ __forceinline__ type get() { return trig_consts(); } };

// include/acrobot.h  (n=4, m=1).  I1=I2=l1=l2=m1=m2=1, lc1=lc2=.5, g=9.81 (:19-25).
template <class real_>
struct AcrobotModelT {
  using real = real_;
  static constexpr int NX = 4;
  static constexpr int NU = 1;
  real goal[4];  // acrobot.h:21  (3.1415, 0, 0, 0)
  real u_min[1], u_max[1];

  using trig_consts_t = typename TrigConstsOf<real>::type;
  __device__ __forceinline__ void dynamics(const real* x, const real* u, real* dx) const { dynamics_k(x, u, dx, TrigConstsOf<real>::get()); }
  // (the same function with sincos_shared's register constants handed in: WithTrigConsts)
  __device__ __forceinline__ void dynamics_k(const real* x, const real* u, real* dx, const trig_consts_t& tk) const {
    // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
    // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
    const real I1 = 1, I2 = 1, l1 = 1, l2 = 1, m1 = 1, m2 = 1, g = real(9.81);
    const real lc1 = real(0.5) * l1, lc2 = real(0.5) * l2;
    const real q0 = x[0], q1 = x[1], qd0 = x[2], qd1 = x[3];
    // the four trig values of acrobot.h:44,55,65,66 from two shared-reduction evaluations;
    // sin(q0+q1) by the angle-sum identity (about 2e-16 absolute)
    real s1, c1, s2, c2;
    sincos_shared(q0, s1, c1, tk);
    sincos_shared(q1, s2, c2, tk);
    const real s12 = s1 * c2 + c1 * s2;
    // H(q), acrobot.h:43-51
    const real H00 = I1 + I2 + m2 * l1 * l1 + 2 * m2 * l1 * lc2 * c2;
    const real H01 = I2 + m2 * l1 * lc2 * c2;
    const real H10 = H01;
    const real H11 = I2;
    // C(q,qd), acrobot.h:53-61
    const real C00 = -2 * m2 * l1 * lc2 * s2 * qd1;
    const real C01 = -m2 * l2 * lc2 * s2 * qd1;
    const real C10 = m2 * l1 * lc2 * s2 * qd0;
    // G(q), acrobot.h:63-70
    const real G0 = m1 * g * lc1 * s1 + m2 * g * (l1 * s1 + lc2 * s12);
    const real G1 = m2 * g * lc2 * s12;
    // rhs = (0,u) - C*qd - G, acrobot.h:80
    const real r0 = (real(0.0) - (C00 * qd0 + C01 * qd1)) - G0;
    const real r1 = (u[0] - (C10 * qd0)) - G1;
    // H^-1 as Eigen's fixed 2x2 inverse (LU/InverseImpl.h:76-96): invdet then 4 products.  invdet by v_rcp + Newton (<= 1 ulp;
    // det = (3 + c2) - (1 + c2/2)^2 lies in [0.75, 1.75]): 5 instead of the 12 instructions of an IEEE division, in every
    // rollout step and twenty times per finite-difference knot.
    const real invdet = recip(H00 * H11 - H10 * H01);
    dx[0] = qd0;
    dx[1] = qd1;
    dx[2] = (H11 * invdet) * r0 + (-H01 * invdet) * r1;
    dx[3] = (-H10 * invdet) * r0 + (H00 * invdet) * r1;
  }
  // Exact derivatives (opt-in, ILQR_FLAG_ANALYTIC_DERIVATIVES; the reference only has finite
  // differences): the record of knot (x, u) in Rec<4,1> order, column-major blocks.
  // qdd = H^-1 r with r = (0,u) - C qd - G, so d qdd / dz = H^-1 (dr/dz - (dH/dz) qdd); the Euler map
  // has fx = I + dt df/dx, fu = dt df/du.  t = T follows the conventions of derivatives.cpp
  // (fx = fu = 0, cx / cxx from final_cost, cu = 0, cuu from cost(x_T, .), cxu = 0).
  __device__ __forceinline__ void analytic_record(const real* x, const real* u, real dt, bool last, real* rec) const {
    using R = Rec<4, 1>;
#pragma unroll
    for (int e = 0; e < R::SIZE; e++) rec[e] = real(0);
    rec[R::CUU] = 2 * real(0.1) * real(0.1);  // d2/du2 of Kr^2 u^2
    if (last) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        rec[R::CX + i] = -real(2.0) * real(400.0) * (goal[i] - x[i]);  // Ks^2 = Kd^2 = 400
        rec[R::CXX + i + 4 * i] = real(2.0) * real(400.0);
      }
      return;
    }
    const real g = real(9.81), b = real(0.5);  // b = m2 l1 lc2
    const real qd0 = x[2], qd1 = x[3];
    real s1, c1, s2, c2;
    sincos_shared(x[0], s1, c1);
    sincos_shared(x[1], s2, c2);
    const real s12 = s1 * c2 + c1 * s2, c12 = c1 * c2 - s1 * s2;
    const real H00 = real(3.0) + 2 * b * c2, H01 = real(1.0) + b * c2, H11 = real(1.0);
    const real Cq0 = -2 * b * s2 * qd1 * qd0 - b * s2 * qd1 * qd1, Cq1 = b * s2 * qd0 * qd0;
    const real G0 = g * (real(1.5) * s1 + real(0.5) * s12), G1 = real(0.5) * g * s12;
    const real r0 = -Cq0 - G0, r1 = u[0] - Cq1 - G1;
    const real invdet = real(1.0) / (H00 * H11 - H01 * H01);
    const real a0 = invdet * (H11 * r0 - H01 * r1), a1 = invdet * (-H01 * r0 + H00 * r1);  // qdd
    real dr[5][2];  // columns q0, q1, qd0, qd1, u of dr/dz - (dH/dz) qdd
    dr[0][0] = -g * (real(1.5) * c1 + real(0.5) * c12);
    dr[0][1] = -real(0.5) * g * c12;
    {
      const real dC0 = -2 * b * c2 * qd1 * qd0 - b * c2 * qd1 * qd1, dC1 = b * c2 * qd0 * qd0;
      const real dG = real(0.5) * g * c12;
      const real dH00 = -2 * b * s2, dH01 = -b * s2;
      dr[1][0] = -dC0 - dG - (dH00 * a0 + dH01 * a1);
      dr[1][1] = -dC1 - dG - (dH01 * a0);
    }
    dr[2][0] = 2 * b * s2 * qd1;
    dr[2][1] = -2 * b * s2 * qd0;
    dr[3][0] = 2 * b * s2 * qd0 + 2 * b * s2 * qd1;
    dr[3][1] = real(0.0);
    dr[4][0] = real(0.0);
    dr[4][1] = real(1.0);
    real dq[5][2];
#pragma unroll
    for (int z = 0; z < 5; z++) {
      dq[z][0] = invdet * (H11 * dr[z][0] - H01 * dr[z][1]);
      dq[z][1] = invdet * (-H01 * dr[z][0] + H00 * dr[z][1]);
    }
    // fx = I + dt * [[0,0,1,0],[0,0,0,1],[d qdd0/dx],[d qdd1/dx]]   (element (r, c) at r + 4 c)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      rec[R::FX + c + 4 * c] = real(1.0);
      rec[R::FX + 2 + 4 * c] += dt * dq[c][0];
      rec[R::FX + 3 + 4 * c] += dt * dq[c][1];
    }
    rec[R::FX + 0 + 4 * 2] += dt;
    rec[R::FX + 1 + 4 * 3] += dt;
    rec[R::FU + 2] = dt * dq[4][0];
    rec[R::FU + 3] = dt * dq[4][1];
    rec[R::CU] = 2 * real(0.1) * real(0.1) * u[0];
  }

  // acrobot.h:83-92: Ks = Kd = 0, Kr = 0.1 -> the state terms are exact zeros for finite x
  __device__ __forceinline__ real cost(const real* x, const real* u) const {
    // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
    // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
    (void)x;
    const real Kr = real(0.1);
    return Kr * Kr * (u[0] * u[0]);
  }
  // acrobot.h:94-100: Ks = Kd = 20
  __device__ __forceinline__ real final_cost(const real* x) const {
    // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
    // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
    const real q0 = goal[0] - x[0], q1 = goal[1] - x[1];
    const real qd0 = goal[2] - x[2], qd1 = goal[3] - x[3];
    const real Ks = real(20.0), Kd = real(20.0);
    return Ks * Ks * (q0 * q0 + q1 * q1) + Kd * Kd * (qd0 * qd0 + qd1 * qd1);
  }
};

// include/double_integrator.h  (n=4, m=2), mass = 1, Hx = diag(1,1,.2,.2), Hu = I.
template <class real_>
struct DoubleIntegratorModelT {
  using real = real_;
  static constexpr int NX = 4;
  static constexpr int NU = 2;
  real goal[4];
  real u_min[2], u_max[2];

  __device__ __forceinline__ void dynamics(const real* x, const real* u, real* dx) const {
    // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
    // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
    const real mass = real(1.0);  // double_integrator.h:29-37
    dx[0] = x[2];
    dx[1] = x[3];
    dx[2] = u[0] / mass;
    dx[3] = u[1] / mass;
  }
  __device__ __forceinline__ real quad(const real* x, real scale) const {
    // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
    // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
    const real hx[4] = {1, 1, real(0.2), real(0.2)};
    real d[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      d[i] = goal[i] - x[i];
      r[i] = scale * (hx[i] * d[i]);
    }
    return (r[0] * d[0] + r[2] * d[2]) + (r[1] * d[1] + r[3] * d[3]);
  }
  // exact derivatives (opt-in, see AcrobotModel::analytic_record): linear dynamics, quadratic costs
  __device__ __forceinline__ void analytic_record(const real* x, const real* u, real dt, bool last, real* rec) const {
    using R = Rec<4, 2>;
    const real hx[4] = {1, 1, real(0.2), real(0.2)};
#pragma unroll
    for (int e = 0; e < R::SIZE; e++) rec[e] = real(0);
    const real scale = last ? real(10.0) : real(1.0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      rec[R::CX + i] = -real(2.0) * scale * hx[i] * (goal[i] - x[i]);
      rec[R::CXX + i + 4 * i] = real(2.0) * scale * hx[i];
    }
    rec[R::CUU + 0] = real(2.0);  // cuu at every t (at t = T from cost(x_T, .))
    rec[R::CUU + 3] = real(2.0);
    if (last) return;
#pragma unroll
    for (int c = 0; c < 4; c++) rec[R::FX + c + 4 * c] = real(1.0);
    rec[R::FX + 0 + 4 * 2] = dt;
    rec[R::FX + 1 + 4 * 3] = dt;
    rec[R::FU + 2 + 4 * 0] = dt;  // mass = 1
    rec[R::FU + 3 + 4 * 1] = dt;
    rec[R::CU + 0] = real(2.0) * u[0];
    rec[R::CU + 1] = real(2.0) * u[1];
  }
  // double_integrator.h:39-43
  __device__ __forceinline__ real cost(const real* x, const real* u) const {
    // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
    // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
    return quad(x, real(1.0)) + (u[0] * u[0] + u[1] * u[1]);
  }
  // double_integrator.h:45-48
  __device__ __forceinline__ real final_cost(const real* x) const { return quad(x, real(10.0)); }
};

// A model whose dynamics() takes sincos_shared's register constants from the object instead of materialising them per call: built
// once before a loop that evaluates the model every trip (rollout_tile).  Same expressions, same constants: the same bits.  Models
// without dynamics_k (the double integrator, user twins) pass through unchanged.
template <class M, class = void>
struct WithTrigConsts : M {
  __device__ __forceinline__ explicit WithTrigConsts(const M& m) : M(m) {}
};
template <class M>
struct WithTrigConsts<M, std::void_t<typename M::trig_consts_t>> : M {
  using real = typename M::real;
  typename M::trig_consts_t tk;
  __device__ __forceinline__ explicit WithTrigConsts(const M& m) : M(m), tk(TrigConstsOf<real>::get()) {}
  __device__ __forceinline__ void dynamics(const real* x, const real* u, real* dx) const { M::dynamics_k(x, u, dx, tk); }
};

// include/model.h:12-15  x1 = x + dynamics(x,u)*dt
template <class M>
__device__ __forceinline__ void integrate_dynamics(const M& m, const typename M::real* x, const typename M::real* u, typename M::real dt,
                                                   typename M::real* x1) {
  // contraction by the source, not by the optimiser: every inlined copy of this function (rollout kernels, the
  // persistent kernel, the commit / getter re-integration, the finite differences) must round identically
#pragma clang fp contract(on)
  typename M::real dx[M::NX];
  m.dynamics(x, u, dx);
#pragma unroll
  for (int i = 0; i < M::NX; i++) x1[i] = x[i] + dx[i] * dt;
}

using AcrobotModel = AcrobotModelT<double>;
using DoubleIntegratorModel = DoubleIntegratorModelT<double>;

// does a device model ship exact derivatives (analytic_record)?  Optional: without it ILQR_FLAG_ANALYTIC_DERIVATIVES
// is refused for that model and the finite-difference sweep is the only one.
template <class M, class = void>
struct has_analytic_record : std::false_type {};
template <class M>
struct has_analytic_record<M, std::void_t<decltype(std::declval<const M&>().analytic_record((const typename M::real*)nullptr, (const typename M::real*)nullptr,
                                                                                            typename M::real(0), false, (typename M::real*)nullptr))>> : std::true_type {};

}  // namespace ilqr

// ------------------------------------------------------------------------------------------
// A user's device twin WITHOUT editing the library (the reference's Model is an open plugin interface,
// include/model.h:6-21): build with  -DILQR_USER_MODEL_HEADER='"my_model.hpp"'  (ilqr_amd._build.build_user) and create
// handles with ILQR_MODEL_USER.  The header is included here, inside namespace ilqr, and defines
//
//     template <class real_> struct UserModelT {
//       using real = real_;
//       static constexpr int NX = 4;             // state dimension, <= 32
//       static constexpr int NU = 1;             // control dimension, <= 16
//       real u_min[NU], u_max[NU];               // filled by ilqr_create from ilqr_desc.u_min / u_max
//       ... its own parameters (plain data) ...
//       void set_params(const double* p, int n);                                  // host: ilqr_desc.user_params
//       __device__ void dynamics(const real* x, const real* u, real* dx) const;   // Model::dynamics
//       __device__ real cost(const real* x, const real* u) const;                 // Model::cost
//       __device__ real final_cost(const real* x) const;                          // Model::final_cost
//       // optional: __device__ void analytic_record(const real* x, const real* u, real dt, bool last, real* rec) const;
//     };
//
// It may use what this file offers (sincos_shared, Rec<>).  NX = 4 with NU = 1 or 2 runs in the tiled lane-quad kernels (the
// persistent routes of the shipped acrobot / double integrator; both arithmetic flavours are instantiated: fp32 handles take
// their finite differences in UserModelT<double>) -- examples/user_model_acrobot.hpp.  Any other NX <= 32, NU <= 16 runs in
// the generic kernels (generic.hpp: thread-per-rollout k_rollout_g, wavefront-per-knot finite differences k_derivatives_g,
// the matrix-core backward pass k_backward_w3; fp64; an analytic_record, if the model has one, is called by one lane per knot under
// ILQR_FLAG_ANALYTIC_DERIVATIVES) -- examples/user_model_linear6.hpp.  A SMALL twin (even NX <= 8, NU <= 4) is compiled into both and
// runs, unless ILQR_ROUTE_WAVE_PER_TRAJECTORY asks for the generic kernels, in the tiled thread kernels: one thread per knot
// (k_derivatives), per trajectory (k_backward_t) and per rollout (k_rollout) with the whole 6 x 6 algebra in registers -- a 16 x 16
// matrix-core tile is (6 / 16)^2 full at n = 6 --, in fp64 or fp32.
// ------------------------------------------------------------------------------------------
#ifdef ILQR_USER_MODEL_HEADER
namespace ilqr {
#include ILQR_USER_MODEL_HEADER
static_assert(UserModelT<double>::NX >= 1 && UserModelT<double>::NX <= MAXN && UserModelT<double>::NU >= 1 && UserModelT<double>::NU <= MAXM,
              "user device models: 1 <= NX <= 32, 1 <= NU <= 16");
// which kernels the build's user model runs in
constexpr bool kUserQuad = UserModelT<double>::NX == 4 && (UserModelT<double>::NU == 1 || UserModelT<double>::NU == 2);  // lane-quad kernels, persistent routes
constexpr bool kUserSmall = !kUserQuad && UserModelT<double>::NX % 2 == 0 && UserModelT<double>::NX <= 8 && UserModelT<double>::NU <= 4;  // tiled thread kernels
constexpr bool kUserTiled = kUserQuad || kUserSmall;   // has tiled kernels (trajectory-interleaved layout)
constexpr bool kUserGeneric = !kUserQuad;              // has generic kernels (trajectory-contiguous layout, wavefront per trajectory)
}  // namespace ilqr
#define ILQR_HAVE_USER_MODEL 1
#endif

namespace ilqr {

}  // namespace ilqr
