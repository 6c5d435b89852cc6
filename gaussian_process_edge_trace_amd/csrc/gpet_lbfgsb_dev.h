// L-BFGS-B state machine of the converged fits (gpet_lbfgsb.hip has the description): device functions shared by the
// round-based driver (k_lb_advance, one thread per problem and round) and the persistent objective kernel of
// gpet_kernels.hip (k_lml16_fit: a problem's evaluations and its state machine in one workgroup).  The functions are
// NOT inlined: both callers then run the same machine code, so a problem takes the same path either way.
#pragma once
#include "gpet_kernels.h"

// LB_FN: how the state machine's functions are compiled into the including unit.  Default: not inlined (the round-based driver
// k_lb_advance, gpet_lbfgsb.hip).  gpet_kernels.hip defines it as inline for k_lml16_fit (round 6): inside that kernel a call costs
// the spill and reload of the objective's live registers around it and leaves every field of the problem in LDS; inlined, the
// optimiser keeps the scalars of the step in registers.  (The two modes already differ in the objective's last bits: the tests
// compare them to tolerance, tests/test_gpu_trace.py::test_final_fit_one_workgroup_per_problem_equals_rounds.)
#ifndef LB_FN
#define LB_FN __attribute__((noinline))
#endif

namespace gpet {

#define LB_M 10
#define LB_EPS 2.220446049250313e-16
#define LB_FACTR_EPS (1e7 * LB_EPS)
#define LB_PGTOL 1e-5
#define LB_MAXLS 20
#define LB_MAXITER 15000
#define LB_FTOL 1e-3
#define LB_GTOL 0.9
#define LB_XTOL 0.1

enum { LB_TASK_FIRST = 0, LB_TASK_LS = 1, LB_TASK_DONE = 2 };

struct LbProb {
  double x[3], g[3], f;
  double xe[3];  // the point whose objective value is pending
  double xold[3], gold[3], fold;
  double d[3], z[3];
  double S[LB_M][3], Y[LB_M][3];
  double theta;
  double stp, stpmx, gd, gdold;
  double finit, ginit, gtest, width, width1, stx, fx, gx, sty, fy, gy, stmin, stmax;
  int ncorr, brackt, stage, iter, nfev, ifun, task, why, edge, slot;
};

// bounds of theta = log(constant, length_scale, noise_level)  (gpet.py:246-248)
static __device__ __forceinline__ void lb_bounds(double* lo, double* hi) {
  lo[0] = log(0.01);
  hi[0] = log(1e3);
  lo[1] = log(0.1);
  hi[1] = log(100.0);
  lo[2] = log(1e-18);
  hi[2] = log(1.0);
}

static __device__ LB_FN double lb_projgr(const double* x, const double* g, const double* l, const double* u) {
  double s = 0.0;
  for (int i = 0; i < 3; ++i) {
    double gi = g[i];
    if (gi < 0.0) gi = fmax(x[i] - u[i], gi);
    else gi = fmin(x[i] - l[i], gi);
    s = fmax(s, fabs(gi));
  }
  return s;
}

// B = theta I updated by the stored pairs, oldest first
static __device__ LB_FN void lb_dense_B(const LbProb& p, double B[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) B[i][j] = (i == j) ? p.theta : 0.0;
  for (int k = 0; k < p.ncorr; ++k) {
    const double* s = p.S[k];
    const double* y = p.Y[k];
    double Bs[3];
    for (int i = 0; i < 3; ++i) Bs[i] = B[i][0] * s[0] + B[i][1] * s[1] + B[i][2] * s[2];
    const double sBs = s[0] * Bs[0] + s[1] * Bs[1] + s[2] * Bs[2];
    const double ys = y[0] * s[0] + y[1] * s[1] + y[2] * s[2];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) B[i][j] = B[i][j] - Bs[i] * Bs[j] / sBs + y[i] * y[j] / ys;
  }
}

// generalised Cauchy point (algorithm CP of Byrd et al.): first local minimiser of the quadratic model along the
// projected steepest-descent path.  free[i] = variable i is not at a bound at xcp.
static __device__ LB_FN void lb_cauchy(const double* x, const double* g, const double* l, const double* u, const double B[3][3],
                          double sbgnrm, double* xcp, int* free_) {
  for (int i = 0; i < 3; ++i) {
    xcp[i] = x[i];
    free_[i] = 0;
  }
  if (sbgnrm <= 0.0) return;
  double t[3], d[3];
  int fixed[3], hit[3] = {0, 0, 0};
  for (int i = 0; i < 3; ++i) {
    const double neg = -g[i];
    const double tl = x[i] - l[i], tu = u[i] - x[i];
    int iw = 0;
    if (tl <= 0.0) {
      if (neg <= 0.0) iw = 1;
    } else if (tu <= 0.0) {
      if (neg >= 0.0) iw = 2;
    } else if (fabs(neg) <= 0.0) {
      iw = -3;
    }
    fixed[i] = iw > 0;
    t[i] = INFINITY;
    if (iw != 0) {
      d[i] = 0.0;
    } else {
      d[i] = neg;
      if (neg < 0.0) t[i] = tl / (-neg);
      else if (neg > 0.0) t[i] = tu / neg;
    }
  }
  // breakpoints in increasing order (3 variables: insertion sort)
  int order[3], nb = 0;
  for (int i = 0; i < 3; ++i)
    if (d[i] != 0.0 && isfinite(t[i])) {
      int k = nb++;
      while (k > 0 && t[order[k - 1]] > t[i]) {
        order[k] = order[k - 1];
        --k;
      }
      order[k] = i;
    }
  double z[3] = {0.0, 0.0, 0.0};
  double f1 = -(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double f2 = 0.0;
  for (int i = 0; i < 3; ++i) f2 += d[i] * (B[i][0] * d[0] + B[i][1] * d[1] + B[i][2] * d[2]);
  const double f2_org = f2;
  double dtm = f2 != 0.0 ? -f1 / f2 : INFINITY;
  double tsum = 0.0;
  bool done = false;
  for (int k = 0; k < nb; ++k) {
    const int ib = order[k];
    const double dt = t[ib] - tsum;
    if (dtm < dt) break;
    tsum += dt;
    for (int i = 0; i < 3; ++i) z[i] += dt * d[i];
    if (d[ib] > 0.0) {
      z[ib] = u[ib] - x[ib];
      xcp[ib] = u[ib];
    } else {
      z[ib] = l[ib] - x[ib];
      xcp[ib] = l[ib];
    }
    hit[ib] = 1;
    d[ib] = 0.0;
    if (d[0] == 0.0 && d[1] == 0.0 && d[2] == 0.0) {  // every variable is fixed
      dtm = 0.0;
      done = true;
      break;
    }
    f1 = 0.0;
    f2 = 0.0;
    for (int i = 0; i < 3; ++i) {
      const double Bz = B[i][0] * z[0] + B[i][1] * z[1] + B[i][2] * z[2];
      const double Bd = B[i][0] * d[0] + B[i][1] * d[1] + B[i][2] * d[2];
      f1 += (g[i] + Bz) * d[i];
      f2 += d[i] * Bd;
    }
    f2 = fmax(LB_EPS * f2_org, f2);
    dtm = -f1 / f2;
  }
  if (!done) {
    if (dtm <= 0.0) dtm = 0.0;
    tsum += dtm;
    for (int i = 0; i < 3; ++i)
      if (d[i] != 0.0) xcp[i] = x[i] + tsum * d[i];
  }
  for (int i = 0; i < 3; ++i) free_[i] = !(fixed[i] || hit[i]);
}

// x = A^-1 b for an m x m system, m <= 3, LU with partial pivoting (what numpy.linalg.solve / LAPACK gesv does)
static __device__ LB_FN void lb_solve(int m, double A[3][3], double* b, double* x) {
  int perm[3] = {0, 1, 2};
  for (int c = 0; c < m; ++c) {
    int pv = c;
    for (int r = c + 1; r < m; ++r)
      if (fabs(A[perm[r]][c]) > fabs(A[perm[pv]][c])) pv = r;
    const int tmp = perm[c];
    perm[c] = perm[pv];
    perm[pv] = tmp;
    for (int r = c + 1; r < m; ++r) {
      const double fct = A[perm[r]][c] / A[perm[c]][c];
      A[perm[r]][c] = fct;
      for (int k = c + 1; k < m; ++k) A[perm[r]][k] -= fct * A[perm[c]][k];
    }
  }
  double y[3];
  for (int r = 0; r < m; ++r) {
    double v = b[perm[r]];
    for (int k = 0; k < r; ++k) v -= A[perm[r]][k] * y[k];
    y[r] = v;
  }
  for (int r = m - 1; r >= 0; --r) {
    double v = y[r];
    for (int k = r + 1; k < m; ++k) v -= A[perm[r]][k] * x[k];
    x[r] = v / A[perm[r]][r];
  }
}

// subspace minimisation over the free variables at the Cauchy point, with the projection / backtracking of L-BFGS-B 3.0
static __device__ LB_FN void lb_subsm(const double* x, const double* g, const double* xcp, const int* free_, const double* l,
                         const double* u, const double B[3][3], double* xn) {
  int idx[3], m = 0;
  for (int i = 0; i < 3; ++i) {
    xn[i] = xcp[i];
    if (free_[i]) idx[m++] = i;
  }
  if (m == 0) return;
  double A[3][3], r[3], dF[3];
  for (int a = 0; a < m; ++a) {
    const int i = idx[a];
    double Bz = 0.0;
    for (int j = 0; j < 3; ++j) Bz += B[i][j] * (xcp[j] - x[j]);
    r[a] = -(g[i] + Bz);
    for (int c = 0; c < m; ++c) A[a][c] = B[i][idx[c]];
  }
  lb_solve(m, A, r, dF);
  int iword = 0;
  for (int a = 0; a < m; ++a) {
    const int k = idx[a];
    const double xk = fmax(l[k], xcp[k] + dF[a]);
    xn[k] = fmin(u[k], xk);
    if (xn[k] == l[k] || xn[k] == u[k]) iword = 1;
  }
  if (!iword) return;
  double dd_p = 0.0;
  for (int i = 0; i < 3; ++i) dd_p += (xn[i] - x[i]) * g[i];
  if (dd_p > 0.0) {
    for (int i = 0; i < 3; ++i) xn[i] = xcp[i];
    double alpha = 1.0, temp1 = 1.0;
    int ibd = -1;
    for (int a = 0; a < m; ++a) {
      const int k = idx[a];
      const double dk = dF[a];
      if (dk < 0.0) {
        const double temp2 = l[k] - xn[k];
        if (temp2 >= 0.0) temp1 = 0.0;
        else if (dk * alpha < temp2) temp1 = temp2 / dk;
      } else if (dk > 0.0) {
        const double temp2 = u[k] - xn[k];
        if (temp2 <= 0.0) temp1 = 0.0;
        else if (dk * alpha > temp2) temp1 = temp2 / dk;
      }
      if (temp1 < alpha) {
        alpha = temp1;
        ibd = a;
      }
    }
    if (alpha < 1.0 && ibd >= 0) {
      const double dk = dF[ibd];
      const int k = idx[ibd];
      if (dk > 0.0) {
        xn[k] = u[k];
        dF[ibd] = 0.0;
      } else if (dk < 0.0) {
        xn[k] = l[k];
        dF[ibd] = 0.0;
      }
    }
    for (int a = 0; a < m; ++a) xn[idx[a]] = xn[idx[a]] + alpha * dF[a];
  }
}

// MINPACK-2 dcstep: safeguarded cubic / quadratic step of the More-Thuente line search
static __device__ LB_FN void lb_dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
                          double fp, double dp, int& brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (dx / fabs(dx));
  double stpf;
  if (fp > fx) {
    const double th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(th), fmax(fabs(dx), fabs(dp)));
    double gam = s * sqrt(fmax(0.0, (th / s) * (th / s) - (dx / s) * (dp / s)));
    if (stp < stx) gam = -gam;
    const double p = (gam - dx) + th, q = ((gam - dx) + gam) + dp, r = p / q;
    const double stpc = stx + r * (stp - stx);
    const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    stpf = (fabs(stpc - stx) < fabs(stpq - stx)) ? stpc : stpc + (stpq - stpc) / 2.0;
    brackt = 1;
  } else if (sgnd < 0.0) {
    const double th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(th), fmax(fabs(dx), fabs(dp)));
    double gam = s * sqrt(fmax(0.0, (th / s) * (th / s) - (dx / s) * (dp / s)));
    if (stp > stx) gam = -gam;
    const double p = (gam - dp) + th, q = ((gam - dp) + gam) + dx, r = p / q;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    stpf = (fabs(stpc - stp) > fabs(stpq - stp)) ? stpc : stpq;
    brackt = 1;
  } else if (fabs(dp) < fabs(dx)) {
    const double th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(th), fmax(fabs(dx), fabs(dp)));
    double gam = s * sqrt(fmax(0.0, (th / s) * (th / s) - (dx / s) * (dp / s)));
    if (stp > stx) gam = -gam;
    const double p = (gam - dp) + th, q = (gam + (dx - dp)) + gam, r = p / q;
    double stpc;
    if (r < 0.0 && gam != 0.0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = stpmax;
    else stpc = stpmin;
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      stpf = (fabs(stpc - stp) < fabs(stpq - stp)) ? stpc : stpq;
      if (stp > stx) stpf = fmin(stp + 0.66 * (sty - stp), stpf);
      else stpf = fmax(stp + 0.66 * (sty - stp), stpf);
    } else {
      stpf = (fabs(stpc - stp) > fabs(stpq - stp)) ? stpc : stpq;
      stpf = fmin(stpmax, stpf);
      stpf = fmax(stpmin, stpf);
    }
  } else {
    if (brackt) {
      const double th = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double s = fmax(fabs(th), fmax(fabs(dy), fabs(dp)));
      double gam = s * sqrt(fmax(0.0, (th / s) * (th / s) - (dy / s) * (dp / s)));
      if (stp > sty) gam = -gam;
      const double p = (gam - dp) + th, q = ((gam - dp) + gam) + dy, r = p / q;
      stpf = stp + r * (sty - stp);
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stp = stpf;
}

// MINPACK-2 dcsrch, one return per objective evaluation.  Returns 0: evaluate at the new p.stp; 1: converged; 2: warning
static __device__ LB_FN int lb_ls_step(LbProb& p, double f, double g) {
  const double stpmin = 0.0, stpmax = p.stpmx;
  double stp = p.stp;
  const double ftest = p.finit + stp * p.gtest;
  if (p.stage == 1 && f <= ftest && g >= 0.0) p.stage = 2;
  int task = 0;
  if (p.brackt && (stp <= p.stmin || stp >= p.stmax)) task = 2;
  if (p.brackt && p.stmax - p.stmin <= LB_XTOL * p.stmax) task = 2;
  if (stp == stpmax && f <= ftest && g <= p.gtest) task = 2;
  if (stp == stpmin && (f > ftest || g >= p.gtest)) task = 2;
  if (f <= ftest && fabs(g) <= LB_GTOL * (-p.ginit)) task = 1;
  if (task != 0) return task;
  if (p.stage == 1 && f <= p.fx && f > ftest) {
    const double fm = f - stp * p.gtest;
    double fxm = p.fx - p.stx * p.gtest, fym = p.fy - p.sty * p.gtest;
    const double gm = g - p.gtest;
    double gxm = p.gx - p.gtest, gym = p.gy - p.gtest;
    lb_dcstep(p.stx, fxm, gxm, p.sty, fym, gym, stp, fm, gm, p.brackt, p.stmin, p.stmax);
    p.fx = fxm + p.stx * p.gtest;
    p.fy = fym + p.sty * p.gtest;
    p.gx = gxm + p.gtest;
    p.gy = gym + p.gtest;
  } else {
    lb_dcstep(p.stx, p.fx, p.gx, p.sty, p.fy, p.gy, stp, f, g, p.brackt, p.stmin, p.stmax);
  }
  if (p.brackt) {
    if (fabs(p.sty - p.stx) >= 0.66 * p.width1) stp = p.stx + 0.5 * (p.sty - p.stx);
    p.width1 = p.width;
    p.width = fabs(p.sty - p.stx);
  }
  if (p.brackt) {
    p.stmin = fmin(p.stx, p.sty);
    p.stmax = fmax(p.stx, p.sty);
  } else {
    p.stmin = stp + 1.1 * (stp - p.stx);
    p.stmax = stp + 4.0 * (stp - p.stx);
  }
  stp = fmax(stp, stpmin);
  stp = fmin(stp, stpmax);
  if ((p.brackt && (stp <= p.stmin || stp >= p.stmax)) || (p.brackt && p.stmax - p.stmin <= LB_XTOL * p.stmax)) stp = p.stx;
  p.stp = stp;
  return 0;
}

static __device__ LB_FN void lb_emit_trial(LbProb& p) {
  p.ifun += 1;
  p.nfev += 1;
  for (int i = 0; i < 3; ++i) p.xe[i] = (p.stp == 1.0) ? p.z[i] : p.stp * p.d[i] + p.xold[i];
  p.task = LB_TASK_LS;
}

// search direction of a new iteration + the first trial point of its line search (or termination)
static __device__ LB_FN void lb_begin_iteration(LbProb& p, const double* l, const double* u) {
  for (;;) {
    double B[3][3];
    lb_dense_B(p, B);
    const double sbgnrm = lb_projgr(p.x, p.g, l, u);
    double xcp[3];
    int free_[3];
    lb_cauchy(p.x, p.g, l, u, B, sbgnrm, xcp, free_);
    if ((free_[0] || free_[1] || free_[2]) && p.ncorr > 0) lb_subsm(p.x, p.g, xcp, free_, l, u, B, p.z);
    else
      for (int i = 0; i < 3; ++i) p.z[i] = xcp[i];
    for (int i = 0; i < 3; ++i) p.d[i] = p.z[i] - p.x[i];
    if (p.iter == 0) {
      p.stpmx = 1.0;
    } else {
      double stpmx = 1e10;
      for (int i = 0; i < 3; ++i) {
        const double a1 = p.d[i];
        if (a1 < 0.0) {
          const double a2 = l[i] - p.x[i];
          if (a2 >= 0.0) stpmx = 0.0;
          else if (a1 * stpmx < a2) stpmx = a2 / a1;
        } else if (a1 > 0.0) {
          const double a2 = u[i] - p.x[i];
          if (a2 <= 0.0) stpmx = 0.0;
          else if (a1 * stpmx > a2) stpmx = a2 / a1;
        }
      }
      p.stpmx = stpmx;
    }
    p.stp = 1.0;  // (every variable has both bounds: "boxed")
    for (int i = 0; i < 3; ++i) {
      p.xold[i] = p.x[i];
      p.gold[i] = p.g[i];
    }
    p.fold = p.f;
    p.gd = p.g[0] * p.d[0] + p.g[1] * p.d[1] + p.g[2] * p.d[2];
    p.gdold = p.gd;
    p.ifun = 0;
    if (p.gd >= 0.0) {  // ascent direction in the projection: refresh the memory and retry, or give up
      if (p.ncorr == 0) {
        p.task = LB_TASK_DONE;
        p.why = 4;
        return;
      }
      p.ncorr = 0;
      p.theta = 1.0;
      continue;
    }
    // dcsrch 'START'
    p.brackt = 0;
    p.stage = 1;
    p.finit = p.f;
    p.ginit = p.gd;
    p.gtest = LB_FTOL * p.gd;
    p.width = p.stpmx;
    p.width1 = 2.0 * p.width;
    p.stx = 0.0;
    p.fx = p.f;
    p.gx = p.gd;
    p.sty = 0.0;
    p.fy = p.f;
    p.gy = p.gd;
    p.stmin = 0.0;
    p.stmax = p.stp + 4.0 * p.stp;
    lb_emit_trial(p);
    return;
  }
}

// consume the objective value at p.xe and move to the next trial point
static __device__ LB_FN void lb_advance(LbProb& p, double f, const double* g, const double* l, const double* u) {
  if (p.task == LB_TASK_FIRST) {
    p.f = f;
    for (int i = 0; i < 3; ++i) p.g[i] = g[i];
    if (lb_projgr(p.x, p.g, l, u) <= LB_PGTOL) {
      p.task = LB_TASK_DONE;
      p.why = 1;
      return;
    }
    lb_begin_iteration(p, l, u);
    return;
  }
  // inside a line search
  p.f = f;
  for (int i = 0; i < 3; ++i) {
    p.g[i] = g[i];
    p.x[i] = p.xe[i];
  }
  p.gd = p.g[0] * p.d[0] + p.g[1] * p.d[1] + p.g[2] * p.d[2];
  const double stp_used = p.stp;
  const int ls = lb_ls_step(p, p.f, p.gd);
  if (ls == 0) {
    if (p.ifun >= LB_MAXLS) {  // line search failed: back to the previous iterate, refresh the memory or give up
      for (int i = 0; i < 3; ++i) {
        p.x[i] = p.xold[i];
        p.g[i] = p.gold[i];
      }
      p.f = p.fold;
      if (p.ncorr == 0) {
        p.task = LB_TASK_DONE;
        p.why = 5;
        return;
      }
      p.ncorr = 0;
      p.theta = 1.0;
      lb_begin_iteration(p, l, u);
      return;
    }
    lb_emit_trial(p);
    return;
  }
  // the line search accepted p.x: end of the iteration
  p.stp = stp_used;
  p.iter += 1;
  if (lb_projgr(p.x, p.g, l, u) <= LB_PGTOL) {
    p.task = LB_TASK_DONE;
    p.why = 2;
    return;
  }
  const double ddum0 = fmax(fabs(p.fold), fmax(fabs(p.f), 1.0));
  if ((p.fold - p.f) <= LB_FACTR_EPS * ddum0 || p.iter >= LB_MAXITER) {
    p.task = LB_TASK_DONE;
    p.why = 3;
    return;
  }
  double r[3], rr = 0.0;
  for (int i = 0; i < 3; ++i) {
    r[i] = p.g[i] - p.gold[i];
    rr += r[i] * r[i];
  }
  double dr, ddum;
  if (p.stp == 1.0) {
    dr = p.gd - p.gdold;
    ddum = -p.gdold;
  } else {
    dr = (p.gd - p.gdold) * p.stp;
    for (int i = 0; i < 3; ++i) p.d[i] *= p.stp;
    ddum = -p.gdold * p.stp;
  }
  if (dr > LB_EPS * ddum) {  // otherwise the update is skipped
    if (p.ncorr == LB_M) {
      for (int k = 1; k < LB_M; ++k)
        for (int i = 0; i < 3; ++i) {
          p.S[k - 1][i] = p.S[k][i];
          p.Y[k - 1][i] = p.Y[k][i];
        }
      p.ncorr = LB_M - 1;
    }
    for (int i = 0; i < 3; ++i) {
      p.S[p.ncorr][i] = p.d[i];
      p.Y[p.ncorr][i] = r[i];
    }
    p.ncorr += 1;
    p.theta = rr / dr;
  }
  lb_begin_iteration(p, l, u);
}


}  // namespace gpet
