// backward_thread.hpp -- backward_pass + box-QP + lambda retry (src/ilqr_core.cpp:350-401, 136-159), one THREAD per
// trajectory with the generic box_qp<M>: the cross-check of the quad kernel (ILQR_FLAG_BACKWARD_THREAD_PER_TRAJ).
#pragma once
#include "derivatives.hpp"

namespace ilqr {

// ------------------------------------------------------------------------------------------
// backward pass, one thread per trajectory
// ------------------------------------------------------------------------------------------
// mode 0: exactly one backward_pass() at the current lambda for every trajectory (stage call)
// mode 1: STEP 2 of the outer loop for running trajectories: retry with increased lambda while
//         the pass diverges (ilqr_core.cpp:136-150), then the gradient-norm test (:153-159).
template <class M>
__global__ __launch_bounds__(64) void k_backward_t(BatchViewT<typename M::real> v, M model, SolverParams sp, int mode) {
  using real = typename M::real;  // what is stored per knot
  using creal = double;           // what the recursion computes in (backward_quad.hpp: the mixed mode of fp32 handles)
  constexpr int NX = M::NX, NU = M::NU;
  using R = Rec<NX, NU>;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= v.B) return;
  if (mode == 1 && v.status[b] != 0) return;
  const int tile = b / TW, l = b % TW;
  const int T = v.T;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  const real* Dt = v.D + didx(tile, 0, 0, l, T + 1, R::SIZE);
  auto rec = [&](int t, int e) { return Dt[((size_t)t * (R::SIZE / 2) + (e >> 1)) * (2 * TW) + (e & 1)]; };

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0;  // (per-trajectory accumulators: double in both modes)
  while (true) {
    creal Vx[NX], Vxx[NX * NX], kprev[NU];
#pragma unroll
    for (int i = 0; i < NX; i++) Vx[i] = rec(T, R::CX + i);  // :353
#pragma unroll
    for (int e = 0; e < NX * NX; e++) Vxx[e] = rec(T, R::CXX + e);  // :354
#pragma unroll
    for (int j = 0; j < NU; j++) kprev[j] = v.kff[tidx(tile, T - 1, j, l, T, NU)];  // k[min(i+1,T-1)] at i=T-1
    dV0 = dV1 = 0;  // :356
    diverge = 0;

    for (int i = T - 1; i >= 0; i--) {
      creal fx[NX * NX], fu[NX * NU], cx[NX], cu[NU], cxx[NX * NX], cxu[NX * NU], cuu[NU * NU], us[NU];
#pragma unroll
      for (int e = 0; e < NX * NX; e++) fx[e] = rec(i, R::FX + e);
#pragma unroll
      for (int e = 0; e < NX * NU; e++) fu[e] = rec(i, R::FU + e);
#pragma unroll
      for (int e = 0; e < NX; e++) cx[e] = rec(i, R::CX + e);
#pragma unroll
      for (int e = 0; e < NU; e++) cu[e] = rec(i, R::CU + e);
#pragma unroll
      for (int e = 0; e < NX * NX; e++) cxx[e] = rec(i, R::CXX + e);
#pragma unroll
      for (int e = 0; e < NX * NU; e++) cxu[e] = rec(i, R::CXU + e);
#pragma unroll
      for (int e = 0; e < NU * NU; e++) cuu[e] = rec(i, R::CUU + e);
#pragma unroll
      for (int j = 0; j < NU; j++) us[j] = v.us[tidx(tile, i, j, l, T, NU)];

      creal Qx[NX], Qu[NU], Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU], QuuF[NU * NU];
      creal A1[NX * NX], A2[NU * NX];
      // :359-360
#pragma unroll
      for (int a = 0; a < NX; a++) {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < NX; q++) acc += fx[q + NX * a] * Vx[q];
        Qx[a] = cx[a] + acc;
      }
#pragma unroll
      for (int a = 0; a < NU; a++) {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < NX; q++) acc += fu[q + NX * a] * Vx[q];
        Qu[a] = cu[a] + acc;
      }
      // :361  Qxx = cxx + (fx'Vxx) fx
#pragma unroll
      for (int a = 0; a < NX; a++)
#pragma unroll
        for (int c = 0; c < NX; c++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += fx[q + NX * a] * Vxx[q + NX * c];
          A1[a + NX * c] = acc;
        }
#pragma unroll
      for (int a = 0; a < NX; a++)
#pragma unroll
        for (int c = 0; c < NX; c++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += A1[a + NX * q] * fx[q + NX * c];
          Qxx[a + NX * c] = cxx[a + NX * c] + acc;
        }
      // :362/:366  Qux = cxu' + (fu'Vxx) fx
#pragma unroll
      for (int a = 0; a < NU; a++)
#pragma unroll
        for (int c = 0; c < NX; c++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += fu[q + NX * a] * Vxx[q + NX * c];
          A2[a + NU * c] = acc;
        }
#pragma unroll
      for (int a = 0; a < NU; a++)
#pragma unroll
        for (int c = 0; c < NX; c++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += A2[a + NU * q] * fx[q + NX * c];
          Qux[a + NU * c] = cxu[c + NX * a] + acc;
        }
      // :363/:367  Quu = cuu + (fu'Vxx) fu ; QuuF = cuu + lambda I + (fu'Vxx) fu
#pragma unroll
      for (int a = 0; a < NU; a++)
#pragma unroll
        for (int c = 0; c < NU; c++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += A2[a + NU * q] * fu[q + NX * c];
          Quu[a + NU * c] = cuu[a + NU * c] + acc;
          QuuF[a + NU * c] = (cuu[a + NU * c] + ((a == c) ? (creal)lambda : creal(0))) + acc;
        }
      // opt-in (sp.fixes & 4): lambda regularises Vxx' ([Tassa 2012] eq. 10) instead of Quu:
      // Quu_reg = Quu + lambda fu'fu, Qux_reg = Qux + lambda fu'fx; the value update keeps Quu, Qux
      creal Quxr[NU * NX];
#pragma unroll
      for (int e = 0; e < NU * NX; e++) Quxr[e] = Qux[e];
      if (sp.fixes & 4) {
        const creal lam = (creal)lambda;
#pragma unroll
        for (int a = 0; a < NU; a++) {
#pragma unroll
          for (int c = 0; c < NU; c++) {
            creal acc = 0;
#pragma unroll
            for (int q = 0; q < NX; q++) acc += fu[q + NX * a] * fu[q + NX * c];
            QuuF[a + NU * c] = Quu[a + NU * c] + lam * acc;
          }
#pragma unroll
          for (int c = 0; c < NX; c++) {
            creal acc = 0;
#pragma unroll
            for (int q = 0; q < NX; q++) acc += fu[q + NX * a] * fx[q + NX * c];
            Quxr[a + NU * c] = Qux[a + NU * c] + lam * acc;
          }
        }
      }

      // :369
      creal lo[NU], hi[NU];
#pragma unroll
      for (int j = 0; j < NU; j++) {
        lo[j] = model.u_min[j] - us[j];
        hi[j] = model.u_max[j] - us[j];
      }
      BoxQPResult<NU, creal> qp;
      box_qp<NU>(QuuF, Qu, kprev, lo, hi, qp, (sp.fixes & 2) != 0);
      if (qp.result < 1) {  // :371
        diverge = i;
        break;
      }

      // :373-385
      creal K[NU * NX];
#pragma unroll
      for (int e = 0; e < NU * NX; e++) K[e] = 0;
      {
        int rank[NU], nf = 0;
#pragma unroll
        for (int j = 0; j < NU; j++) {
          rank[j] = nf;
          nf += qp.v_free[j] ? 1 : 0;
        }
        if (nf > 0) {
          creal Minv[NU * NU];
          rinv_rinvT<NU>(qp.nfR, qp.R, Minv);
          const int nuse = (nf < qp.nfR) ? nf : qp.nfR;
#pragma unroll
          for (int c = 0; c < NX; c++) {
            creal qf[NU];  // rows_w_ind(Qux_reg, v_free)(:, c)
#pragma unroll
            for (int a = 0; a < NU; a++) {
              creal val = 0;
#pragma unroll
              for (int j = 0; j < NU; j++)
                if (qp.v_free[j] && rank[j] == a) val = Quxr[j + NU * c];
              qf[a] = val;
            }
#pragma unroll
            for (int j = 0; j < NU; j++) {
              if (qp.v_free[j] && rank[j] < nuse) {
                creal acc = 0;
#pragma unroll
                for (int a = 0; a < NU; a++)
                  if (a < nuse) {
                    creal mrow = 0;  // Minv[rank[j]][a]
#pragma unroll
                    for (int r = 0; r < NU; r++)
                      if (r == rank[j]) mrow = Minv[r + NU * a];
                    acc += -mrow * qf[a];
                  }
                K[j + NU * c] = acc;
              }
            }
          }
        }
      }

      // :388-389
      {
        creal d0 = 0;
#pragma unroll
        for (int j = 0; j < NU; j++) d0 += qp.x[j] * Qu[j];
        dV0 += (double)d0;
        creal d1 = 0;
#pragma unroll
        for (int c = 0; c < NU; c++) {
          creal r = 0;
#pragma unroll
          for (int a = 0; a < NU; a++) r += (creal(0.5) * qp.x[a]) * Quu[a + NU * c];
          d1 += r * qp.x[c];
        }
        dV1 += (double)d1;
      }
      // :391-393
      {
        creal T1[NX * NU];  // K' Quu  (NX x NU)
#pragma unroll
        for (int a = 0; a < NX; a++)
#pragma unroll
          for (int c = 0; c < NU; c++) {
            creal acc = 0;
#pragma unroll
            for (int q = 0; q < NU; q++) acc += K[q + NU * a] * Quu[q + NU * c];
            T1[a + NX * c] = acc;
          }
        creal Vxn[NX], Vn[NX * NX];
#pragma unroll
        for (int a = 0; a < NX; a++) {
          creal t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
          for (int c = 0; c < NU; c++) {
            t1 += T1[a + NX * c] * qp.x[c];
            t2 += K[c + NU * a] * Qu[c];
            t3 += Qux[c + NU * a] * qp.x[c];
          }
          Vxn[a] = ((Qx[a] + t1) + t2) + t3;
        }
#pragma unroll
        for (int a = 0; a < NX; a++)
#pragma unroll
          for (int c = 0; c < NX; c++) {
            creal t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
            for (int q = 0; q < NU; q++) {
              t1 += T1[a + NX * q] * K[q + NU * c];
              t2 += K[q + NU * a] * Qux[q + NU * c];
              t3 += Qux[q + NU * a] * K[q + NU * c];
            }
            Vn[a + NX * c] = ((Qxx[a + NX * c] + t1) + t2) + t3;
          }
#pragma unroll
        for (int a = 0; a < NX; a++) {
          Vx[a] = Vxn[a];
#pragma unroll
          for (int c = 0; c < NX; c++) Vxx[a + NX * c] = creal(0.5) * (Vn[a + NX * c] + Vn[c + NX * a]);
        }
      }
      // :396-397
#pragma unroll
      for (int j = 0; j < NU; j++) {
        v.kff[tidx(tile, i, j, l, T, NU)] = (real)qp.x[j];
        kprev[j] = (creal)(real)qp.x[j];  // the stored gain, as the reference reads k[i + 1] back (:369)
      }
#pragma unroll
      for (int e = 0; e < NU * NX; e++) v.Kfb[tidx(tile, i, e, l, T, NU * NX)] = (real)K[e];
    }  // for i

    if (mode == 0) {
      done = (diverge == 0);
      break;
    }
    if (diverge != 0) {  // :142-148
      dlambda = fmax(dlambda * sp.lambda_factor, sp.lambda_factor);
      lambda = fmax(lambda * dlambda, sp.lambda_min);
      if (lambda > sp.lambda_max) break;
      continue;
    }
    done = true;
    break;
  }

  v.dV[b] = dV0;
  v.dV[v.Bp + b] = dV1;
  v.diverge[b] = diverge;
  v.backpass_done[b] = done ? 1 : 0;
  if (mode == 1) {
    v.lambda[b] = lambda;
    v.dlambda[b] = dlambda;
  }
  // :153 / :405-412  gnorm = mean_t max_j |k_j| / (|u_j| + 1), ascending t like std::accumulate
  double acc = 0;
  for (int t = 0; t < T; t++) {
    real mx = 0;
#pragma unroll
    for (int j = 0; j < NU; j++) {
      const real val = abs_of(v.kff[tidx(tile, t, j, l, T, NU)]) / (abs_of(v.us[tidx(tile, t, j, l, T, NU)]) + 1);
      mx = (j == 0 || val > mx) ? val : mx;
    }
    acc += (double)mx;
  }
  const double gnorm = acc / T;
  v.gnorm[b] = gnorm;
  if (mode == 1 && !sp.fixed_work && gnorm < sp.tol_grad && lambda < 1e-5) {  // :154-159
    v.status[b] = 1;
    v.iters[b] += 1;  // this iteration was started
  }
}

}  // namespace ilqr
