/* CPU oracle (plain C) for the batched Euler-Bernoulli beam FE solve.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or called from the product
 * library (openpystruct_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the shared object built from this file.
 *
 * PARITY UNPINNED: the reference's arithmetic for this path is inside the third-party,
 * un-pinned `openseespy` wheel (/root/reference/environment.yml:13-14), absent from
 * /root/reference and from this image; the reference holds no tests or golden vectors.
 * This file restates the OpenSees semantics selected by the reference's call sites:
 *
 *   setup_model                 OpenPyStruct_BeamOpt_training_SingleCore.py:89-124
 *   system('BandSPD')           ...SingleCore.py:120   -> LAPACK dpbsv = dpbtrf + dpbtrs
 *   constraints('Plain')        ...SingleCore.py:122   -> constrained DOFs get no equation
 *   analyze(1)                  ...SingleCore.py:182   -> 0 on success, non-zero if not SPD
 *   eleResponse(e,'forces')[1|2]...SingleCore.py:189-190
 *   nodeDisp(n, 2|3)            ...SingleCore.py:224-232
 *
 * Bending sub-problem only (2 DOF/node): on a straight horizontal beam the axial DOFs
 * decouple (oracle/beam_oracle.py::solve_model_3dof keeps them and is compared with this
 * file by tests/test_oracle.py).
 *
 * Build:  make -C oracle      (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KD 3 /* half bandwidth of the 2-DOF/node chain after omitting constrained DOFs */

/* dpbtf2-style unblocked upper Cholesky of an SPD band matrix, ab[(KD+1) x n] column
 * major with ab[KD + i - j + j*(KD+1)] = A(i,j), i <= j.  Returns 0, or k+1 if the
 * k-th leading minor is not positive definite (LAPACK `info`). */
static int band_cholesky(int n, double *ab) {
  const int ld = KD + 1;
  for (int j = 0; j < n; ++j) {
    double ajj = ab[KD + j * ld];
    if (!(ajj > 0.0)) return j + 1;
    ajj = sqrt(ajj);
    ab[KD + j * ld] = ajj;
    int kn = (KD < n - j - 1) ? KD : n - j - 1;
    /* row j of U right of the diagonal: U(j, j+c), c = 1..kn, stored at ab[KD-c + (j+c)*ld] */
    for (int c = 1; c <= kn; ++c) ab[KD - c + (j + c) * ld] /= ajj;
    /* trailing update A(j+r, j+c) -= U(j,j+r) U(j,j+c), r <= c */
    for (int c = 1; c <= kn; ++c) {
      double ujc = ab[KD - c + (j + c) * ld];
      for (int r = 1; r <= c; ++r) {
        double ujr = ab[KD - r + (j + r) * ld];
        ab[KD - (c - r) + (j + c) * ld] -= ujr * ujc;
      }
    }
  }
  return 0;
}

/* dpbtrs: solve U^T U x = b in place */
static void band_solve(int n, const double *ab, double *b) {
  const int ld = KD + 1;
  for (int j = 0; j < n; ++j) { /* U^T y = b */
    double s = b[j];
    int k0 = (j - KD > 0) ? j - KD : 0;
    for (int k = k0; k < j; ++k) s -= ab[KD - (j - k) + j * ld] * b[k];
    b[j] = s / ab[KD + j * ld];
  }
  for (int j = n - 1; j >= 0; --j) { /* U x = y */
    double s = b[j];
    int k1 = (j + KD < n - 1) ? j + KD : n - 1;
    for (int k = j + 1; k <= k1; ++k) s -= ab[KD - (k - j) + k * ld] * b[k];
    b[j] = s / ab[KD + j * ld];
  }
}

/* One beam.  Scratch: eq[2N] ints, ab[(KD+1)*2N], rhs[2N]. */
static int solve_one(int Ne, const double *x, const double *E, int E_per_elem, const double *I,
                     const uint8_t *fix, const double *Fy, const double *wy, int w_per_elem,
                     double *v, double *th, double *V, double *M, int *eq, double *ab, double *rhs) {
  const int N = Ne + 1;
  int n = 0;
  for (int a = 0; a < N; ++a) { /* PlainHandler: number only the free DOFs, node order */
    eq[2 * a] = (fix[a] & 1) ? -1 : n++;
    eq[2 * a + 1] = (fix[a] & 2) ? -1 : n++;
  }
  memset(ab, 0, sizeof(double) * (size_t)(KD + 1) * (size_t)(n > 0 ? n : 1));
  for (int a = 0; a < N; ++a)
    if (eq[2 * a] >= 0) rhs[eq[2 * a]] = Fy[a];
  for (int a = 0; a < N; ++a)
    if (eq[2 * a + 1] >= 0) rhs[eq[2 * a + 1]] = 0.0;
  for (int e = 0; e < Ne; ++e) {
    const double L = x[e + 1] - x[e];
    const double EI = (E_per_elem ? E[e] : E[0]) * I[e];
    const double w = w_per_elem ? wy[e] : wy[0];
    const double k = EI / (L * L * L);
    const double ke[4][4] = {{12 * k, 6 * k * L, -12 * k, 6 * k * L},
                             {6 * k * L, 4 * k * L * L, -6 * k * L, 2 * k * L * L},
                             {-12 * k, -6 * k * L, 12 * k, -6 * k * L},
                             {6 * k * L, 2 * k * L * L, -6 * k * L, 4 * k * L * L}};
    const double fe[4] = {w * L / 2, w * L * L / 12, w * L / 2, -w * L * L / 12};
    for (int r = 0; r < 4; ++r) {
      int ir = eq[2 * e + r];
      if (ir < 0) continue;
      rhs[ir] += fe[r];
      for (int c = 0; c < 4; ++c) {
        int ic = eq[2 * e + c];
        if (ic < 0 || ic < ir) continue;
        ab[KD - (ic - ir) + ic * (KD + 1)] += ke[r][c];
      }
    }
  }
  int info = band_cholesky(n, ab);
  if (info == 0) band_solve(n, ab, rhs);
  for (int a = 0; a < N; ++a) {
    v[a] = (eq[2 * a] >= 0) ? (info ? NAN : rhs[eq[2 * a]]) : 0.0;
    th[a] = (eq[2 * a + 1] >= 0) ? (info ? NAN : rhs[eq[2 * a + 1]]) : 0.0;
  }
  for (int e = 0; e < Ne; ++e) { /* ElasticBeam2d::getResistingForce, bending part */
    const double L = x[e + 1] - x[e];
    const double EI = (E_per_elem ? E[e] : E[0]) * I[e];
    const double w = w_per_elem ? wy[e] : wy[0];
    const double chord = (v[e + 1] - v[e]) / L;
    const double p1 = th[e] - chord, p2 = th[e + 1] - chord;
    const double q1 = 4 * EI / L * p1 + 2 * EI / L * p2 - w * L * L / 12;
    const double q2 = 2 * EI / L * p1 + 4 * EI / L * p2 + w * L * L / 12;
    V[e] = (q1 + q2) / L - w * L / 2; /* forces[1] */
    M[e] = q1;                        /* forces[2] */
  }
  return info;
}

/* Same argument meaning as include/openpystruct_amd.h::ops_beam_solve_batched_f64, with
 * HOST pointers and no stream.  Strides are in elements; a batch stride of 0 means the
 * array is shared by all beams (x, fix) or a scalar (E, wy). */
int oracle_beam_solve_batched_f64(int B, int Ne, const double *x, long x_bstride, const double *E,
                                  long E_bstride, const double *I, long I_bstride, const uint8_t *fix,
                                  long fix_bstride, const double *Fy, long Fy_bstride, const double *wy,
                                  long wy_bstride, double *v, double *theta, double *V, double *M,
                                  int32_t *status, int n_threads) {
  const int N = Ne + 1;
  int failed = 0;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : failed)
  {
    int *eq = (int *)malloc(sizeof(int) * 2 * (size_t)N);
    double *ab = (double *)malloc(sizeof(double) * (KD + 1) * 2 * (size_t)N);
    double *rhs = (double *)malloc(sizeof(double) * 2 * (size_t)N);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      int info = solve_one(Ne, x + (size_t)b * x_bstride, E + (size_t)b * E_bstride, E_bstride != 0,
                           I + (size_t)b * I_bstride, fix + (size_t)b * fix_bstride,
                           Fy + (size_t)b * Fy_bstride, wy + (size_t)b * wy_bstride, wy_bstride != 0,
                           v + (size_t)b * N, theta + (size_t)b * N, V + (size_t)b * Ne,
                           M + (size_t)b * Ne, eq, ab, rhs);
      if (status) status[b] = info;
      failed += (info != 0);
    }
    free(eq);
    free(ab);
    free(rhs);
  }
  return failed;
}
