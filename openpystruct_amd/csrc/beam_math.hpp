// Per-lane arithmetic of the batched Euler-Bernoulli beam solve.
//
// One beam is solved by P co-operating lanes of a 64-wide wavefront.  Lane j owns the
// M consecutive elements [j*M, (j+1)*M) ("segment"), i.e. local nodes 0..M where local
// node 0 (global node j*M) is the lane's LEFT boundary and local node M is the next
// lane's left boundary.  The solve is a substructured block-LDL^T (block Cholesky)
// of the block-tridiagonal stiffness matrix (2x2 blocks: u_y, theta_z per node):
//
//   A. seg_condense : frontal elimination of the segment's interior nodes 1..M-1; what is
//                     left is a 4x4 "super element" on (left, right) boundary nodes.  Only
//                     the inverse pivot blocks (3 doubles per node) are kept.
//   B. make_row / cr_eliminate / cr_back : the P boundary nodes form a P-row block-tridiagonal interface
//                     system, solved by cyclic reduction over the P lanes
//   C. seg_solve    : with both boundary displacements known the interior is a one-sided
//                     block-Thomas solve: right-hand-side sweep (re-using the stored pivot
//                     inverses), back substitution, element end-force recovery
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
// Element terms (OpenSees ElasticBeam2d, bending part; SingleCore.py:107, :117), UNMASKED:
//   kA = 12EI/L^3, kB = 6EI/L^2, kC = 4EI/L, kD = 2EI/L from the unit-inertia tile entries
//   c2 = 2E/L, c6 = 6E/L^2, c12 = 12E/L^3 (staged in LDS by the kernel) times Ie;
//   k11 = [[kA, kB],[kB, kC]], k12 = [[-kA, kB],[-kB, kD]], k22 = [[kA, -kB],[-kB, kC]];
//   consistent UDL loads (pw, mw) on node a and (pw, -mw) on node b, pw = wL/2, mw = wL^2/12.
// Constraints are applied per NODE when the node is eliminated (mask_node / mask_cols /
// mask_rows): fixed DOFs get an identity row/column and a zero right-hand side.
// ---------------------------------------------------------------------------------------
struct ElemK { double kA, kB, kC, kD; };

BEAM_HD ElemK elem_k(double c2, double c6, double c12, double Ie) {
  const double kD = c2 * Ie;
  return ElemK{c12 * Ie, c6 * Ie, kD + kD, kD};
}

// Per-node constraint flags.  RZ = false is the fast path for waves in which no node has a fixed
// rotation (the reference only ever fixes translations: `ops.fix(n, 0|1, 1, 0)`, SingleCore.py:100-102):
// the rotation flag is the constant 1 and all its multiplications fold away at compile time.
template <bool RZ>
struct Flags {
  double v, dv;   // v: 1.0 free / 0.0 fixed; dv = 1 - v (identity on the fixed diagonal)
  double t, dt;
};
template <bool RZ>
BEAM_HD Flags<RZ> node_flags(unsigned long long bits, int node) {
  const bool fv = (bits >> (2 * node)) & 1ull, ft = RZ && ((bits >> (2 * node + 1)) & 1ull);
  return Flags<RZ>{fv ? 0.0 : 1.0, fv ? 1.0 : 0.0, ft ? 0.0 : 1.0, ft ? 1.0 : 0.0};
}
// the same flags from two separate bit fields (bit i of v / t: u_y / theta_z of local node i fixed; at most 32 nodes)
struct FixPair { unsigned v, t; };
template <bool RZ>
BEAM_HD Flags<RZ> node_flags(const FixPair& bits, int node) {
  const bool fv = (bits.v >> node) & 1u, ft = RZ && ((bits.t >> node) & 1u);
  return Flags<RZ>{fv ? 0.0 : 1.0, fv ? 1.0 : 0.0, ft ? 0.0 : 1.0, ft ? 1.0 : 0.0};
}
template <bool RZ>
BEAM_HD Sym2 mask_node(const Sym2& s, const Flags<RZ>& c) {  // identity on fixed DOFs
  if (RZ) return Sym2{__builtin_fma(c.v, s.a, c.dv), (c.v * c.t) * s.b, __builtin_fma(c.t, s.c, c.dt)};
  return Sym2{__builtin_fma(c.v, s.a, c.dv), c.v * s.b, s.c};
}
template <bool RZ>
BEAM_HD Vec2 mask_vec(const Vec2& g, const Flags<RZ>& c) {
  if (RZ) return Vec2{c.v * g.x, c.t * g.y};
  return Vec2{c.v * g.x, g.y};
}
template <bool RZ>
BEAM_HD Mat2 mask_cols(const Mat2& m, const Flags<RZ>& c) {
  if (RZ) return Mat2{m.a * c.v, m.b * c.t, m.c * c.v, m.d * c.t};
  return Mat2{m.a * c.v, m.b, m.c * c.v, m.d};
}
template <bool RZ>
BEAM_HD Mat2 mask_rows(const Mat2& m, const Flags<RZ>& c) {
  if (RZ) return Mat2{m.a * c.v, m.b * c.v, m.c * c.t, m.d * c.t};
  return Mat2{m.a * c.v, m.b * c.v, m.c, m.d};
}

// Inverse of the constrained pivot block, PROJECTED onto the free DOFs: fixed DOFs get a unit pivot for the
// factorisation (identity row/column, so the pivot test passes) and a ZERO row/column in the returned inverse.
template <bool RZ>
BEAM_HD Sym2 proj_inv(const Sym2& s, const Flags<RZ>& c, int& bad) {
  const Sym2 g = inv_spd(mask_node(s, c), bad);   // off-diagonal already zero when a DOF is fixed
  if (RZ) return Sym2{c.v * g.a, g.b, c.t * g.c};
  return Sym2{c.v * g.a, g.b, g.c};
}

// "Compute this HERE": the left-boundary accumulators S_LL, g_L and the pivot test feed nothing inside the
// condensation loop, and the compiler, left alone, sinks all M of their updates behind the loop -- keeping every
// iteration's operands alive (measured at M = 17: ~10 doubles per iteration, 500 live registers, AGPR copies and
// scratch).  An empty asm that takes the value as a read-write register operand pins it to its iteration.
#if defined(__HIP_DEVICE_COMPILE__)
#define BEAM_PIN_F64(x) __asm__ volatile("" : "+v"(x))
#define BEAM_PIN_I32(x) __asm__ volatile("" : "+v"(x))
#else
#define BEAM_PIN_F64(x) ((void)0)
#define BEAM_PIN_I32(x) ((void)0)
#endif

// State a lane keeps across the phases.
template <int M>
struct SegState {
  Sym2 Ginv[M];   // inverse pivot blocks of interior nodes 1..M-1 (index 0 unused)
  // condensed super element on (left, right) boundary nodes, not yet masked for either
  Sym2 SLL, Scc;
  Mat2 SLc;
  Vec2 gL, gc;
};

// Acc supplies the lane's inputs by LOCAL index:
//   c2(i), c6(i), c12(i), rL(i), Ie(i), pw(i), mw(i)   element i in [0, M)
//   Fy(i)                                                nodal load at local node i in [0, M)
//   fixbits()   bit 2i = u_y of local node i fixed, bit 2i+1 = theta_z fixed, i in [0, M]
//   fence()     compiler-only memory fence (device) / no-op (host)
template <int M, bool RZ, class Acc>
BEAM_HD void seg_condense(SegState<M>& s, const Acc& acc, int& bad) {
  const unsigned long long fb = acc.fixbits();
  {
    const ElemK k = elem_k(acc.c2(0), acc.c6(0), acc.c12(0), acc.Ie(0));
    s.SLL = Sym2{k.kA, k.kB, k.kC};
    s.SLc = Mat2{-k.kA, k.kB, -k.kB, k.kD};
    s.Scc = Sym2{k.kA, -k.kB, k.kC};
    s.gL = Vec2{acc.pw(0) + acc.Fy(0), acc.mw(0)};
    s.gc = Vec2{acc.pw(0), -acc.mw(0)};
  }
#pragma unroll
  for (int i = 1; i < M; ++i) {
    acc.fence();  // keep element i's loads behind element i-1's: bounds the live registers
    const Flags<RZ> c = node_flags<RZ>(fb, i);
    const ElemK k = elem_k(acc.c2(i), acc.c6(i), acc.c12(i), acc.Ie(i));
    const double pw = acc.pw(i), mw = acc.mw(i);
    // node i is complete: left element (in Scc/gc) + right element.  Its constraints enter ONLY through
    // the projected inverse G (zero rows/columns on fixed DOFs): S_Li G, K_{i+1,i} G and G h then ignore the
    // fixed DOFs by themselves, so couplings and right-hand sides need no masking.
    const Sym2 G = proj_inv(Sym2{s.Scc.a + k.kA, s.Scc.b + k.kB, s.Scc.c + k.kC}, c, bad);
    const Vec2 gi{s.gc.x + pw + acc.Fy(i), s.gc.y + mw};
    const Mat2 Kr{-k.kA, k.kB, -k.kB, k.kD};                             // node i <-> node i+1
    s.Ginv[i] = G;
    const Mat2 Pm = mul(s.SLc, G);   // S_Li * G
    const Mat2 Qm = mulT(Kr, G);     // K_{i+1,i} * G
    s.SLL = sub_mulT(s.SLL, Pm, s.SLc);
    s.gL = sub_mul(s.gL, Pm, gi);
    s.SLc = neg_mul(Pm, Kr);
    s.Scc = sub_mul(Sym2{k.kA, -k.kB, k.kC}, Qm, Kr);
    s.gc = sub_mul(Vec2{pw, -mw}, Qm, gi);
  }
}

// One row of the interface system: K[j,j-s] = Alow, K[j,j] = D, K[j,j+s] = Cup.
struct IfaceRow {
  Mat2 Alow, Cup;
  Sym2 D;
  Vec2 f;
};

// The lane's own coupling to the next boundary node, constrained on both sides; lane j+1
// receives it (transposed) as its Alow.
template <int M, bool RZ, class Bits>
BEAM_HD Mat2 masked_cup(const SegState<M>& s, const Bits& fixbits) {
  return mask_cols(mask_rows(s.SLc, node_flags<RZ>(fixbits, 0)), node_flags<RZ>(fixbits, M));
}

// prevC, prevg: Scc, gc of lane j-1; prevCup: masked_cup of lane j-1 (all zero for lane 0).
template <int M, bool RZ, class Bits>
BEAM_HD IfaceRow make_row(const SegState<M>& s, const Mat2& ownCup, const Sym2& prevC, const Vec2& prevg,
                          const Mat2& prevCup, const Bits& fixbits) {
  const Flags<RZ> c = node_flags<RZ>(fixbits, 0);
  IfaceRow r;
  r.D = mask_node(Sym2{s.SLL.a + prevC.a, s.SLL.b + prevC.b, s.SLL.c + prevC.c}, c);
  r.f = mask_vec(Vec2{s.gL.x + prevg.x, s.gL.y + prevg.y}, c);
  r.Alow = Mat2{prevCup.a, prevCup.c, prevCup.b, prevCup.d};  // transpose
  r.Cup = ownCup;
  return r;
}

// ---------------------------------------------------------------------------------------
// Interface solve: CYCLIC REDUCTION over the P rows of a beam (levels s = 1, 2, 4, ..., P/2).
//
// At level s the rows j = s (mod 2s) are eliminated: they FREEZE (their equation now couples them to rows
// j -+ s only), and the rows j = 0 (mod 2s) absorb them (cr_eliminate: the Schur-complement update).  After the
// last level row 0 stands alone; then, level by level downwards, every frozen row is solved from ITS OWN
// frozen equation with its two neighbours known (cr_back).  That is the block Cholesky factorisation of the
// interface matrix in nested-dissection order followed by forward/back substitution -- like the band solver the
// reference calls (LAPACK dpbsv behind system('BandSPD'), SingleCore.py:120) it is backward stable, and every
// boundary displacement satisfies an equilibrium equation with its neighbours to rounding level.
//
// r01 used PARALLEL cyclic reduction (every row reduced to a 2x2 system of its own).  Same elimination
// arithmetic, but the P solutions carry INDEPENDENT rounding errors of relative size eps * kappa; end forces are
// stiffness x displacement differences, and wherever a whole segment is stiff the condensed segment stiffness
// multiplied those independent errors directly: measured against a 50-digit solution (tests/golden/force_truth.npz)
// up to 1e3 x (16 lanes x 7 elements, adversarial inertias), 1e5 x (M = 4) and 1e7 x (M = 2) the band solver's
// own force error.  With cyclic reduction every tiling is within 8 x of it (tests/test_force_truth.py).
// On a SIMD machine both variants execute the same instructions per level (idle rows are masked, not skipped);
// the back substitution adds 8 exchanged doubles + 12 fma per level, about a fifth of a level's elimination cost.
// ---------------------------------------------------------------------------------------
BEAM_HD bool cr_active(int j, int s) { return (j & (2 * s - 1)) == 0; }   // row j absorbs rows j -+ s at level s
BEAM_HD bool cr_frozen(int j, int s) { return (j & (2 * s - 1)) == s; }   // row j is eliminated at level s

// Level-s update of an ACTIVE row.  Gm/Am/fm come from row j-s, Gp/Cp/fp from row j+s (zeros, or anything
// finite, when that row does not exist: the own coupling towards it is exactly zero).
// LAST: the couplings are not needed after the final level.
// One side of the update: absorb the neighbour row at distance s on the `Aside` coupling (G, Afar, fn: that row's inverse
// pivot, ITS coupling further out in the same direction, its right-hand side).  Aside becomes the coupling at distance 2s.
template <bool LAST>
BEAM_HD void cr_absorb(Sym2& D, Vec2& f, Mat2& Aside, const Sym2& G, const Mat2& Afar, const Vec2& fn) {
  const Mat2 al = mul(Aside, G);    // K[j,j-+s] D_{j-+s}^-1
  D = sub_mulT(D, al, Aside);       // K[j-+s,j] = Aside^T by symmetry
  f = sub_mul(f, al, fn);
  if (!LAST) Aside = neg_mul(al, Afar);
}
template <bool LAST>
BEAM_HD void cr_eliminate(IfaceRow& r, const Sym2& Gm, const Mat2& Am, const Vec2& fm, const Sym2& Gp,
                          const Mat2& Cp, const Vec2& fp) {
  cr_absorb<LAST>(r.D, r.f, r.Alow, Gm, Am, fm);
  cr_absorb<LAST>(r.D, r.f, r.Cup, Gp, Cp, fp);
}

// A frozen row solved from its own equation: u_j = D^-1 (f - Alow u_{j-s} - Cup u_{j+s}); G = D^-1.
BEAM_HD Vec2 cr_back(const IfaceRow& r, const Sym2& G, const Vec2& um, const Vec2& up) {
  return mul(G, sub_mul(sub_mul(r.f, r.Alow, um), r.Cup, up));
}

// Phase C.  Out receives results by LOCAL index: node(i, v, theta), elem(i, V, Mz), i in [0, M).
template <int M, bool RZ, class Acc, class Out>
BEAM_HD void seg_solve(const SegState<M>& s, const Acc& acc, const Vec2& uL, const Vec2& uR, Out& out) {
  (void)acc.fixbits();
  // right-hand-side sweep with the left boundary displacement prescribed (no masks: see seg_condense)
  Vec2 h[M];
  {
    const ElemK k0 = elem_k(acc.c2(0), acc.c6(0), acc.c12(0), acc.Ie(0));
    Vec2 carry = sub_mulT(Vec2{acc.pw(0), -acc.mw(0)}, Mat2{-k0.kA, k0.kB, -k0.kB, k0.kD}, uL);
#pragma unroll
    for (int i = 1; i < M; ++i) {
      acc.fence();
      const double pw = acc.pw(i), mw = acc.mw(i);
      h[i] = Vec2{carry.x + pw + acc.Fy(i), carry.y + mw};
      if (i + 1 < M) {
        const ElemK k = elem_k(acc.c2(i), acc.c6(i), acc.c12(i), acc.Ie(i));
        const Vec2 y = mul(s.Ginv[i], h[i]);
        carry = sub_mulT(Vec2{pw, -mw}, Mat2{-k.kA, k.kB, -k.kB, k.kD}, y);
      }
    }
  }
  // back substitution + ElasticBeam2d::getResistingForce (eleResponse 'forces' [1], [2])
  Vec2 un = uR;  // displacement of local node i+1
#pragma unroll
  for (int i = M - 1; i >= 0; --i) {
    acc.fence();
    const ElemK k = elem_k(acc.c2(i), acc.c6(i), acc.c12(i), acc.Ie(i));
    Vec2 ui;
    if (i > 0) {
      ui = mul(s.Ginv[i], sub_mul(h[i], Mat2{-k.kA, k.kB, -k.kB, k.kD}, un));
    } else {
      ui = uL;
    }
    {
      const double rl = acc.rL(i), mw = acc.mw(i);
      const double chord = (un.x - ui.x) * rl;
      const double p1 = ui.y - chord, p2 = un.y - chord;
      const double q1 = __builtin_fma(k.kC, p1, __builtin_fma(k.kD, p2, -mw));
      const double q2 = __builtin_fma(k.kD, p1, __builtin_fma(k.kC, p2, mw));
      out.elem(i, __builtin_fma(q1 + q2, rl, -acc.pw(i)), q1);
    }
    out.node(i, ui.x, ui.y);
    un = ui;
  }
}

// ---------------------------------------------------------------------------------------
// Phases A and C once more for the "fat wave" tilings (beam_fat.hip: few lanes per beam, many elements per lane,
// ONE wave per SIMD).  Same arithmetic, statement for statement, as seg_condense / seg_solve; what differs is how
// the inputs arrive: with a single resident wave nothing hides the LDS latency of an element's inputs, so element
// i + 1 is requested BEFORE element i is worked on (`acc.elem(i)` returns all inputs of one element at once; the
// fence keeps the request in front of the arithmetic and stops the compiler from hoisting every request of the
// unrolled loop to the top).
// Acc: elem(i) -> ElemIn for local element i in [0, M); fixbits(); fence().
// ---------------------------------------------------------------------------------------
struct ElemIn { double c2, c6, c12, rl, pw, mw, Ie, Fy; };


template <int M, bool RZ, class Acc>
BEAM_HD void seg_condense_pf(SegState<M>& s, const Acc& acc, int& bad) {
  const auto fb = acc.fixbits();
  ElemIn nx = acc.elem(0);
  {
    const ElemIn e = nx;
    if (M > 1) nx = acc.elem(1);
    acc.fence();
    const ElemK k = elem_k(e.c2, e.c6, e.c12, e.Ie);
    s.SLL = Sym2{k.kA, k.kB, k.kC};
    s.SLc = Mat2{-k.kA, k.kB, -k.kB, k.kD};
    s.Scc = Sym2{k.kA, -k.kB, k.kC};
    s.gL = Vec2{e.pw + e.Fy, e.mw};
    s.gc = Vec2{e.pw, -e.mw};
  }
#pragma unroll
  for (int i = 1; i < M; ++i) {
    const ElemIn e = nx;
    if (i + 1 < M) nx = acc.elem(i + 1);
    acc.fence();
    const Flags<RZ> c = node_flags<RZ>(fb, i);
    const ElemK k = elem_k(e.c2, e.c6, e.c12, e.Ie);
    const double pw = e.pw, mw = e.mw;
    const Sym2 G = proj_inv(Sym2{s.Scc.a + k.kA, s.Scc.b + k.kB, s.Scc.c + k.kC}, c, bad);
    const Vec2 gi{s.gc.x + pw + e.Fy, s.gc.y + mw};
    const Mat2 Kr{-k.kA, k.kB, -k.kB, k.kD};
    s.Ginv[i] = G;
    const Mat2 Pm = mul(s.SLc, G);
    const Mat2 Qm = mulT(Kr, G);
    s.SLL = sub_mulT(s.SLL, Pm, s.SLc);
    s.gL = sub_mul(s.gL, Pm, gi);
    s.SLc = neg_mul(Pm, Kr);
    s.Scc = sub_mul(Sym2{k.kA, -k.kB, k.kC}, Qm, Kr);
    s.gc = sub_mul(Vec2{pw, -mw}, Qm, gi);
    BEAM_PIN_F64(s.SLL.a); BEAM_PIN_F64(s.SLL.b); BEAM_PIN_F64(s.SLL.c); BEAM_PIN_F64(s.gL.x); BEAM_PIN_F64(s.gL.y);
    BEAM_PIN_I32(bad);
  }
}

// Hs: where the right-hand-side sweep parks h_i for the back substitution: put(i, h) / get(i) -- registers, or
// (fat tilings, to stay inside 256 VGPRs) the LDS slots that the back substitution overwrites with its results.
// PF: how many elements ahead the inputs are requested (an iteration of these two loops is 14 / 24 FP64 instructions,
// shorter than one LDS round trip).
template <int M, bool RZ, int PF, class Acc, class Out, class Hs>
BEAM_HD void seg_solve_pf(const SegState<M>& s, const Acc& acc, const Vec2& uL, const Vec2& uR, Out& out, Hs& hs) {
  static_assert(PF >= 1 && PF <= 4, "prefetch distance");
  {
    ElemIn q[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) q[d] = acc.elem(d < M ? d : M - 1);
    const ElemIn e0 = q[0];
    if (PF < M) q[0] = acc.elem(PF);
    acc.fence();
    const ElemK k0 = elem_k(e0.c2, e0.c6, e0.c12, e0.Ie);
    Vec2 carry = sub_mulT(Vec2{e0.pw, -e0.mw}, Mat2{-k0.kA, k0.kB, -k0.kB, k0.kD}, uL);
#pragma unroll
    for (int i = 1; i < M; ++i) {
      const ElemIn e = q[i % PF];
      if (i + PF < M) q[i % PF] = acc.elem(i + PF);
      acc.fence();
      const double pw = e.pw, mw = e.mw;
      const Vec2 hi{carry.x + pw + e.Fy, carry.y + mw};
      hs.put(i, hi);
      if (i + 1 < M) {
        const ElemK k = elem_k(e.c2, e.c6, e.c12, e.Ie);
        const Vec2 y = mul(s.Ginv[i], hi);
        carry = sub_mulT(Vec2{pw, -mw}, Mat2{-k.kA, k.kB, -k.kB, k.kD}, y);
      }
    }
  }
  Vec2 un = uR;
  ElemIn q[PF];
  Vec2 hq[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    const int i = M - 1 - d;
    q[(M - 1 - d + PF) % PF] = acc.elem(i >= 0 ? i : 0);
    hq[(M - 1 - d + PF) % PF] = hs.get(i >= 1 ? i : (M > 1 ? 1 : 0));
  }
#pragma unroll
  for (int i = M - 1; i >= 0; --i) {
    const ElemIn e = q[i % PF];
    const Vec2 hi = hq[i % PF];
    if (i - PF >= 0) q[i % PF] = acc.elem(i - PF);
    if (i - PF >= 1) hq[i % PF] = hs.get(i - PF);
    acc.fence();
    const ElemK k = elem_k(e.c2, e.c6, e.c12, e.Ie);
    Vec2 ui;
    if (i > 0) {
      ui = mul(s.Ginv[i], sub_mul(hi, Mat2{-k.kA, k.kB, -k.kB, k.kD}, un));
    } else {
      ui = uL;
    }
    {
      const double rl = e.rl, mw = e.mw;
      const double chord = (un.x - ui.x) * rl;
      const double p1 = ui.y - chord, p2 = un.y - chord;
      const double q1 = __builtin_fma(k.kC, p1, __builtin_fma(k.kD, p2, -mw));
      const double q2 = __builtin_fma(k.kD, p1, __builtin_fma(k.kC, p2, mw));
      out.elem(i, __builtin_fma(q1 + q2, rl, -e.pw), q1);
    }
    out.node(i, ui.x, ui.y);
    un = ui;
  }
}

// ---------------------------------------------------------------------------------------
// Interface solve for FEW rows (fat tilings: P = 6): every lane publishes its part of the interface system once,
// every lane of the beam reads all P parts back (LDS broadcast reads) and runs the block-Thomas elimination over the P
// rows by itself.  Same SPD block-tridiagonal system as make_row builds, same update formulas as cr_absorb, but
// natural order and ONE exchange round instead of 2 log2 P + 2; the P lanes of a beam execute the identical
// instruction sequence on identical data, so their boundary displacements agree bit for bit (what the force
// recovery needs, see the cyclic-reduction note above).
//   row k: K[k,k-1] = cup_{k-1}^T, K[k,k] = Dl_k + Dr_{k-1}, K[k,k+1] = cup_k, rhs fl_k + fr_{k-1}
// ---------------------------------------------------------------------------------------
struct alignas(16) IfacePiece {
  Sym2 Dl; Vec2 fl;   // the lane's own left-boundary block and load, constrained (identity on fixed DOFs included)
  Sym2 Dr; Vec2 fr;   // what it adds to the NEXT lane's left boundary node (its own right boundary), constrained
  Mat2 cup;           // coupling left boundary -> right boundary, constrained on both sides
};
template <int M, bool RZ, class Bits>
BEAM_HD IfacePiece make_piece(const SegState<M>& s, const Bits& fixbits) {
  const Flags<RZ> c0 = node_flags<RZ>(fixbits, 0), cM = node_flags<RZ>(fixbits, M);
  IfacePiece p;
  p.Dl = mask_node(s.SLL, c0);
  p.fl = mask_vec(s.gL, c0);
  p.Dr = RZ ? Sym2{cM.v * s.Scc.a, (cM.v * cM.t) * s.Scc.b, cM.t * s.Scc.c} : Sym2{cM.v * s.Scc.a, cM.v * s.Scc.b, s.Scc.c};
  p.fr = mask_vec(s.gc, cM);
  p.cup = mask_cols(mask_rows(s.SLc, c0), cM);
  return p;
}
// Rd: piece(k) -> IfacePiece of lane k of the reader's beam; cup(k) -> its coupling alone; fence().
// The last lane's Dr / fr / cup belong to the clamped node behind the beam (u = 0) and are never read.
template <int P, class Rd, class Bad>
BEAM_HD void iface_thomas(const Rd& rd, Vec2 (&u)[P], Bad& bad) {
  Sym2 G[P];
  Vec2 fp[P];
  IfacePiece nx = rd.piece(0);
  Sym2 Dr{0.0, 0.0, 0.0};
  Vec2 fr{0.0, 0.0};
  Mat2 cp{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < P; ++k) {
    const IfacePiece e = nx;
    if (k + 1 < P) nx = rd.piece(k + 1);
    rd.fence();
    Sym2 D{e.Dl.a + Dr.a, e.Dl.b + Dr.b, e.Dl.c + Dr.c};
    Vec2 f{e.fl.x + fr.x, e.fl.y + fr.y};
    if (k > 0) {
      const Mat2 A{cp.a, cp.c, cp.b, cp.d};        // K[k,k-1] = cup_{k-1}^T
      const Mat2 W = mul(A, G[k - 1]);
      D = sub_mulT(D, W, A);
      f = sub_mul(f, W, fp[k - 1]);
    }
    G[k] = inv_spd(D, bad);
    fp[k] = f;
    Dr = e.Dr; fr = e.fr; cp = e.cup;
  }
  u[P - 1] = mul(G[P - 1], fp[P - 1]);
  Mat2 cn = P > 1 ? rd.cup(P - 2) : Mat2{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = P - 2; k >= 0; --k) {
    const Mat2 c = cn;
    if (k > 0) cn = rd.cup(k - 1);
    rd.fence();
    u[k] = mul(G[k], sub_mul(fp[k], c, u[k + 1]));
  }
}

}  // namespace opsamd
