// Per-lane arithmetic of the batched Euler-Bernoulli beam solve.
//
// One beam is solved by P co-operating lanes of a 64-wide wavefront.  Lane j owns the
// M consecutive elements [j*M, (j+1)*M) ("segment"), i.e. local nodes 0..M where local
// node 0 (global node j*M) is the lane's LEFT boundary and local node M is the next
// lane's left boundary.  The solve is a substructured block-LDL^T (block Cholesky)
// of the block-tridiagonal stiffness matrix (2x2 blocks: u_y, theta_z per node):
//
//   1. seg_condense : frontal elimination of the segment's interior nodes 1..M-1; what is
//                     left is a 4x4 "super element" on (left, right) boundary nodes
//   2. make_row     : the P boundary nodes form a P-row block-tridiagonal interface system
//   3. pcr_step     : parallel cyclic reduction over the P lanes, log2(P) steps
//   4. seg_backsub  : back substitution of the interior + element end-force recovery
//
// What it restates: the element, load, constraint and recovery semantics OpenSees applies
// to the model built by the reference's `setup_model`
// (/root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py:89-124) and solved by
// `ops.analyze(1)` (:182) with `system('BandSPD')` (:120, SPD band Cholesky, here
// re-ordered as a substructured block Cholesky) and `constraints('Plain')` (:122,
// constrained DOFs carry no equation: here their row/column is replaced by the identity
// with a zero right-hand side, which leaves the free-DOF solution unchanged).
//
// The same header is compiled by hipcc into the kernel (beam_solve.hip) and by g++ into
// the lane-level emulator under tests/ that checks this arithmetic against the oracle on
// machines without a GPU.  It contains no I/O and no cross-lane traffic.
#pragma once

#if defined(__HIPCC__)
#define BEAM_HD __host__ __device__ __forceinline__
#else
#define BEAM_HD inline
#endif

namespace opsamd {

struct Sym2 { double a, b, c; };      // [[a, b], [b, c]]
struct Mat2 { double a, b, c, d; };   // [[a, b], [c, d]]
struct Vec2 { double x, y; };

BEAM_HD double fast_rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  // v_rcp_f64 + two Newton steps: full double precision for normal-range operands
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
#else
  return 1.0 / d;
#endif
}

// inverse of an SPD 2x2 block; *bad is set when a Cholesky pivot is not positive
// (the condition under which LAPACK dpbsv / `analyze` report failure)
BEAM_HD Sym2 inv_spd(const Sym2& s, int& bad) {
  const double bb = s.b * s.b;
  // Kahan-style determinant: the rounding error of b*b is recovered by the second fma
  const double det = __builtin_fma(s.a, s.c, -bb) + __builtin_fma(-s.b, s.b, bb);
  bad |= !(s.a > 0.0) | !(det > 0.0);
  const double r = fast_rcp(det);
  return Sym2{s.c * r, -s.b * r, s.a * r};
}

BEAM_HD Mat2 mul(const Mat2& m, const Sym2& s) {   // m * s
  return Mat2{__builtin_fma(m.a, s.a, m.b * s.b), __builtin_fma(m.a, s.b, m.b * s.c),
              __builtin_fma(m.c, s.a, m.d * s.b), __builtin_fma(m.c, s.b, m.d * s.c)};
}
BEAM_HD Mat2 mulT(const Mat2& m, const Sym2& s) {  // m^T * s
  return Mat2{__builtin_fma(m.a, s.a, m.c * s.b), __builtin_fma(m.a, s.b, m.c * s.c),
              __builtin_fma(m.b, s.a, m.d * s.b), __builtin_fma(m.b, s.b, m.d * s.c)};
}
BEAM_HD Mat2 neg_mul(const Mat2& p, const Mat2& q) {  // -(p * q)
  return Mat2{-__builtin_fma(p.a, q.a, p.b * q.c), -__builtin_fma(p.a, q.b, p.b * q.d),
              -__builtin_fma(p.c, q.a, p.d * q.c), -__builtin_fma(p.c, q.b, p.d * q.d)};
}
// s - p * q^T, symmetric part only (the product is symmetric by construction)
BEAM_HD Sym2 sub_mulT(const Sym2& s, const Mat2& p, const Mat2& q) {
  return Sym2{__builtin_fma(-p.a, q.a, __builtin_fma(-p.b, q.b, s.a)),
              __builtin_fma(-p.a, q.c, __builtin_fma(-p.b, q.d, s.b)),
              __builtin_fma(-p.c, q.c, __builtin_fma(-p.d, q.d, s.c))};
}
// s - p * q, symmetric part only
BEAM_HD Sym2 sub_mul(const Sym2& s, const Mat2& p, const Mat2& q) {
  return Sym2{__builtin_fma(-p.a, q.a, __builtin_fma(-p.b, q.c, s.a)),
              __builtin_fma(-p.a, q.b, __builtin_fma(-p.b, q.d, s.b)),
              __builtin_fma(-p.c, q.b, __builtin_fma(-p.d, q.d, s.c))};
}
BEAM_HD Vec2 sub_mul(const Vec2& v, const Mat2& p, const Vec2& u) {  // v - p u
  return Vec2{__builtin_fma(-p.a, u.x, __builtin_fma(-p.b, u.y, v.x)),
              __builtin_fma(-p.c, u.x, __builtin_fma(-p.d, u.y, v.y))};
}
BEAM_HD Vec2 sub_mulT(const Vec2& v, const Mat2& p, const Vec2& u) {  // v - p^T u
  return Vec2{__builtin_fma(-p.a, u.x, __builtin_fma(-p.c, u.y, v.x)),
              __builtin_fma(-p.b, u.x, __builtin_fma(-p.d, u.y, v.y))};
}
BEAM_HD Vec2 mul(const Sym2& s, const Vec2& u) {
  return Vec2{__builtin_fma(s.a, u.x, s.b * u.y), __builtin_fma(s.b, u.x, s.c * u.y)};
}

// ---------------------------------------------------------------------------------------
// Element terms (OpenSees ElasticBeam2d, bending part; SingleCore.py:107, :117).
//   c2 = 2E/L, c6 = 6E/L^2, c12 = 12E/L^3 (unit-inertia stiffness tile entries, staged in
//   LDS by the kernel), Ie = element inertia, pw = w L / 2, mw = w L^2 / 12.
//   fa/ft, fb/fbt: 1.0 when the DOF (u_y / theta_z of node a / b) is free, else 0.0.
// ---------------------------------------------------------------------------------------
struct ElemK {
  Sym2 k11, k22;
  Mat2 k12;      // couples node a (rows) to node b (columns); k21 = k12^T
  Vec2 fa, fb;   // consistent UDL loads on the free DOFs
};

BEAM_HD ElemK elem_terms(double c2, double c6, double c12, double Ie, double pw, double mw,
                         double av, double at, double bv, double bt) {
  const double k2 = c2 * Ie, k4 = k2 + k2, k6 = c6 * Ie, k12 = c12 * Ie;
  ElemK e;
  e.k11 = Sym2{av * k12, (av * at) * k6, at * k4};
  e.k22 = Sym2{bv * k12, -(bv * bt) * k6, bt * k4};
  e.k12 = Mat2{-(av * bv) * k12, (av * bt) * k6, -(at * bv) * k6, (at * bt) * k2};
  e.fa = Vec2{av * pw, at * mw};
  e.fb = Vec2{bv * pw, -(bt * mw)};
  return e;
}

BEAM_HD double free_flag(unsigned long long bits, int dof) {
  return ((bits >> dof) & 1ull) ? 0.0 : 1.0;
}

// State a lane keeps between condensation and back substitution.
template <int M>
struct SegState {
  // interior nodes 1..M-1 (index 0 unused)
  Sym2 Ginv[M];   // inverse pivot block
  Mat2 SLi[M];    // coupling (left boundary rows, node i columns) at elimination time
  Vec2 g[M];      // right-hand side at elimination time
  // condensed super element on (left, right) boundary
  Sym2 SLL, Scc;
  Mat2 SLc;
  Vec2 gL, gc;
};

// Acc supplies the lane's inputs by LOCAL index:
//   c2(i), c6(i), c12(i), Ie(i), pw(i), mw(i)   element i in [0, M)
//   Fy(i)                                        nodal load at local node i in [0, M)
//   fixbits()   bit 2i = u_y of local node i fixed, bit 2i+1 = theta_z fixed, i in [0, M]
template <int M, class Acc>
BEAM_HD void seg_condense(SegState<M>& s, const Acc& acc, int& bad) {
  const unsigned long long fb = acc.fixbits();
  {
    const double av = free_flag(fb, 0), at = free_flag(fb, 1);
    const double bv = free_flag(fb, 2), bt = free_flag(fb, 3);
    const ElemK e = elem_terms(acc.c2(0), acc.c6(0), acc.c12(0), acc.Ie(0), acc.pw(0), acc.mw(0), av, at, bv, bt);
    s.SLL = e.k11;
    s.SLc = e.k12;
    s.Scc = e.k22;
    s.gL = Vec2{__builtin_fma(av, acc.Fy(0), e.fa.x), e.fa.y};
    s.gc = e.fb;
  }
#pragma unroll
  for (int i = 1; i < M; ++i) {
    const double av = free_flag(fb, 2 * i), at = free_flag(fb, 2 * i + 1);
    const double bv = free_flag(fb, 2 * i + 2), bt = free_flag(fb, 2 * i + 3);
    const ElemK e = elem_terms(acc.c2(i), acc.c6(i), acc.c12(i), acc.Ie(i), acc.pw(i), acc.mw(i), av, at, bv, bt);
    // node i is complete: left element (already in Scc/gc) + right element + identity for fixed DOFs
    const Sym2 Sii{s.Scc.a + e.k11.a + (1.0 - av), s.Scc.b + e.k11.b, s.Scc.c + e.k11.c + (1.0 - at)};
    const Vec2 gi{s.gc.x + __builtin_fma(av, acc.Fy(i), e.fa.x), s.gc.y + e.fa.y};
    const Sym2 G = inv_spd(Sii, bad);
    s.Ginv[i] = G;
    s.SLi[i] = s.SLc;
    s.g[i] = gi;
    const Mat2 Pm = mul(s.SLc, G);      // S_Li * Sii^-1
    const Mat2 Qm = mulT(e.k12, G);     // k21 * Sii^-1
    s.SLL = sub_mulT(s.SLL, Pm, s.SLc);
    s.gL = sub_mul(s.gL, Pm, gi);
    s.SLc = neg_mul(Pm, e.k12);
    s.Scc = sub_mul(e.k22, Qm, e.k12);
    s.gc = sub_mul(e.fb, Qm, gi);
  }
}

// One row of the interface system: K[j,j-s] = Alow, K[j,j] = D, K[j,j+s] = Cup.
struct IfaceRow {
  Mat2 Alow, Cup;
  Sym2 D;
  Vec2 f;
};

// prevC, prevg, prevB: Scc, gc, SLc of lane j-1 (all zero for lane 0); fixbits of own local node 0.
template <int M>
BEAM_HD IfaceRow make_row(const SegState<M>& s, const Sym2& prevC, const Vec2& prevg, const Mat2& prevB,
                          unsigned long long fixbits) {
  const double av = free_flag(fixbits, 0), at = free_flag(fixbits, 1);
  IfaceRow r;
  r.D = Sym2{s.SLL.a + prevC.a + (1.0 - av), s.SLL.b + prevC.b, s.SLL.c + prevC.c + (1.0 - at)};
  r.f = Vec2{s.gL.x + prevg.x, s.gL.y + prevg.y};
  r.Alow = Mat2{prevB.a, prevB.c, prevB.b, prevB.d};  // B_{j-1}^T
  r.Cup = s.SLc;
  return r;
}

// One PCR step.  G = own D^-1 is not needed here; Gm/Am/fm come from row j-s and
// Gp/Cp/fp from row j+s (zeros when that row does not exist).
BEAM_HD void pcr_step(IfaceRow& r, const Sym2& Gm, const Mat2& Am, const Vec2& fm, const Sym2& Gp,
                      const Mat2& Cp, const Vec2& fp) {
  const Mat2 al = mul(r.Alow, Gm);  // K[j,j-s] D_{j-s}^-1
  const Mat2 ga = mul(r.Cup, Gp);   // K[j,j+s] D_{j+s}^-1
  // K[j-s,j] = Alow^T and K[j+s,j] = Cup^T by symmetry
  r.D = sub_mulT(sub_mulT(r.D, al, r.Alow), ga, r.Cup);
  r.f = sub_mul(sub_mul(r.f, al, fm), ga, fp);
  r.Alow = neg_mul(al, Am);
  r.Cup = neg_mul(ga, Cp);
}

// Out receives results by LOCAL index: node(i, v, theta) for i in [0, M), elem(i, V, Mz) for i in [0, M).
template <int M, class Acc, class Out>
BEAM_HD void seg_backsub(const SegState<M>& s, const Acc& acc, const Vec2& uL, const Vec2& uR, Out& out) {
  const unsigned long long fb = acc.fixbits();
  Vec2 un = uR;  // displacement of local node i+1
#pragma unroll
  for (int i = M - 1; i >= 0; --i) {
    const double c2 = acc.c2(i), c6 = acc.c6(i), c12 = acc.c12(i), Ie = acc.Ie(i);
    Vec2 ui;
    if (i > 0) {
      const double av = free_flag(fb, 2 * i), at = free_flag(fb, 2 * i + 1);
      const double bv = free_flag(fb, 2 * i + 2), bt = free_flag(fb, 2 * i + 3);
      const double k2 = c2 * Ie, k6 = c6 * Ie, k12 = c12 * Ie;
      const Mat2 K12{-(av * bv) * k12, (av * bt) * k6, -(at * bv) * k6, (at * bt) * k2};
      Vec2 t = sub_mulT(s.g[i], s.SLi[i], uL);
      t = sub_mul(t, K12, un);
      ui = mul(s.Ginv[i], t);
    } else {
      ui = uL;
    }
    // ElasticBeam2d::getResistingForce, bending part (eleResponse 'forces' [1], [2])
    {
      const double rl = acc.rL(i);
      const double k2 = c2 * Ie, k4 = k2 + k2;
      const double chord = (un.x - ui.x) * rl;
      const double p1 = ui.y - chord, p2 = un.y - chord;
      const double mw = acc.mw(i);
      const double q1 = __builtin_fma(k4, p1, __builtin_fma(k2, p2, -mw));
      const double q2 = __builtin_fma(k2, p1, __builtin_fma(k4, p2, mw));
      out.elem(i, __builtin_fma(q1 + q2, rl, -acc.pw(i)), q1);
    }
    out.node(i, ui.x, ui.y);
    un = ui;
  }
}

}  // namespace opsamd
