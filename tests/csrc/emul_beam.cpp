// Lane-level CPU emulation of openpystruct_amd/csrc/beam_solve.hip -- TEST CODE ONLY.
//
// Runs exactly the per-lane arithmetic of the kernel (beam_math.hpp, shared verbatim) with
// the cross-lane traffic (interface hand-over, parallel cyclic reduction, right-boundary
// fetch) replaced by array reads, so that the algorithm can be checked against the oracle
// on a machine without a GPU.  It is not a product path: nothing under openpystruct_amd/
// loads or links it.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../openpystruct_amd/csrc/beam_math.hpp"

using namespace opsamd;

namespace {
struct HostAcc {
  const double *t2, *t6, *t12, *trl, *tpw, *tmw, *sI, *sF;
  unsigned long long bits;
  double c2(int i) const { return t2[i]; }
  double c6(int i) const { return t6[i]; }
  double c12(int i) const { return t12[i]; }
  double rL(int i) const { return trl[i]; }
  double pw(int i) const { return tpw[i]; }
  double mw(int i) const { return tmw[i]; }
  double Ie(int i) const { return sI[i]; }
  double Fy(int i) const { return sF[i]; }
  unsigned long long fixbits() const { return bits; }
  void fence() const {}
};
// the fat-wave tilings' accessors (beam_fat.hip): one element's inputs per call, flags from two bit fields, h_i parked
struct HostAccPf {
  const double *t2, *t6, *t12, *trl, *tpw, *tmw, *sI, *sF;
  FixPair bits;
  ElemIn elem(int i) const { return ElemIn{t2[i], t6[i], t12[i], trl[i], tpw[i], tmw[i], sI[i], sF[i]}; }
  FixPair fixbits() const { return bits; }
  void fence() const {}
};
struct HostIface {
  const IfacePiece* grp;
  IfacePiece piece(int k) const { return grp[k]; }
  Mat2 cup(int k) const { return grp[k].cup; }
  void fence() const {}
};
struct HostH {
  std::vector<Vec2> h;
  void put(int i, const Vec2& v) { h[i] = v; }
  Vec2 get(int i) const { return h[i]; }
};
struct HostOut {
  double *v, *th, *V, *Mz;
  void elem(int i, double a, double b) { V[i] = a; Mz[i] = b; }
  void node(int i, double a, double b) { v[i] = a; th[i] = b; }
};

template <int P, int M, bool RZ, bool FAT = false>
int solve_one_rz(int Ne, const double* x, const double* E, bool E_pe, const double* I, const uint8_t* fix,
              const double* Fy, const double* wy, bool w_pe, double* v, double* th, double* V, double* Mz) {
  constexpr int PM = P * M;
  const int N = Ne + 1;
  // same padding scheme as the kernel: element Ne has no stiffness, later elements are unit elements
  // (I = 1), free, hanging on the implicit clamp behind the last lane (u_R = 0)
  std::vector<double> tab(6 * PM, 0.0), sI(PM, 1.0), sF(PM, 0.0), ov(PM), ot(PM), oV(PM), oM(PM);
  std::vector<uint8_t> sfix(PM + 8, 0);   // padding nodes are free: the chain of unit elements is clamped behind the last lane
  for (int e = Ne + 1; e < PM; ++e) { tab[0 * PM + e] = 2.0; tab[1 * PM + e] = 6.0; tab[2 * PM + e] = 12.0; tab[3 * PM + e] = 1.0; }
  for (int e = 0; e < Ne; ++e) {
    const double L = x[e + 1] - x[e], rl = fast_rcp(L);
    const double Ee = E_pe ? E[e] : E[0], w = w_pe ? wy[e] : wy[0];
    const double c2 = 2.0 * Ee * rl, c6 = 3.0 * c2 * rl, c12 = 2.0 * c6 * rl, pw = 0.5 * w * L;
    tab[0 * PM + e] = c2; tab[1 * PM + e] = c6; tab[2 * PM + e] = c12;
    tab[3 * PM + e] = rl; tab[4 * PM + e] = pw; tab[5 * PM + e] = pw * L * (1.0 / 6.0);
    sI[e] = I[e];
  }
  for (int n = 0; n < N; ++n) { sF[n] = Fy[n]; sfix[n] = fix[n] & 3; }
  std::vector<SegState<M>> st(P);
  std::vector<HostAcc> acc(P);
  std::vector<HostAccPf> accp(P);
  int bad = 0;
  for (int j = 0; j < P; ++j) {
    const int e0 = j * M;
    acc[j] = HostAcc{&tab[0 * PM + e0], &tab[1 * PM + e0], &tab[2 * PM + e0], &tab[3 * PM + e0],
                     &tab[4 * PM + e0], &tab[5 * PM + e0], &sI[e0], &sF[e0], 0};
    accp[j] = HostAccPf{acc[j].t2, acc[j].t6, acc[j].t12, acc[j].trl, acc[j].tpw, acc[j].tmw, acc[j].sI, acc[j].sF, FixPair{0u, 0u}};
    for (int i = 0; i <= M; ++i) {
      acc[j].bits |= (unsigned long long)sfix[e0 + i] << (2 * i);
      accp[j].bits.v |= (unsigned)(sfix[e0 + i] & 1) << i;
      accp[j].bits.t |= (unsigned)((sfix[e0 + i] >> 1) & 1) << i;
    }
    if (FAT) seg_condense_pf<M, RZ>(st[j], accp[j], bad);
    else seg_condense<M, RZ>(st[j], acc[j], bad);
  }
  std::vector<Vec2> u(P, Vec2{0, 0});
  if (FAT) {   // beam_fat.hip: publish the pieces, block-Thomas over the P rows in natural order
    std::vector<IfacePiece> pieces(P);
    for (int j = 0; j < P; ++j) pieces[j] = make_piece<M, RZ>(st[j], accp[j].bits);
    Vec2 uu[P];
    iface_thomas<P>(HostIface{pieces.data()}, uu, bad);
    for (int j = 0; j < P; ++j) u[j] = uu[j];
  } else {
  std::vector<IfaceRow> row(P), nxt(P);
  std::vector<Mat2> cup(P);
  const Sym2 z3{0, 0, 0}; const Mat2 z4{0, 0, 0, 0}; const Vec2 z2{0, 0};
  for (int j = 0; j < P; ++j) cup[j] = masked_cup<M, RZ>(st[j], acc[j].bits);
  for (int j = 0; j < P; ++j)
    row[j] = make_row<M, RZ>(st[j], cup[j], j ? st[j - 1].Scc : z3, j ? st[j - 1].gc : z2, j ? cup[j - 1] : z4, acc[j].bits);
  // cyclic reduction (beam_math.hpp): forward levels, then the frozen rows back from the top level
  for (int s = 1; s < P; s *= 2) {
    std::vector<Sym2> G(P);
    for (int j = 0; j < P; ++j) G[j] = inv_spd(row[j].D, bad);
    nxt = row;
    for (int j = 0; j < P; ++j) {
      if (!cr_active(j, s)) continue;
      const bool okm = j >= s, okp = j + s < P;
      if (2 * s < P)
        cr_eliminate<false>(nxt[j], okm ? G[j - s] : z3, okm ? row[j - s].Alow : z4, okm ? row[j - s].f : z2,
                            okp ? G[j + s] : z3, okp ? row[j + s].Cup : z4, okp ? row[j + s].f : z2);
      else
        cr_eliminate<true>(nxt[j], okm ? G[j - s] : z3, z4, okm ? row[j - s].f : z2, okp ? G[j + s] : z3, z4,
                           okp ? row[j + s].f : z2);
    }
    row = nxt;
  }
  std::vector<Sym2> Gf(P);
  for (int j = 0; j < P; ++j) Gf[j] = inv_spd(row[j].D, bad);
  u[0] = mul(Gf[0], row[0].f);
  int top = 1;
  while (2 * top < P) top *= 2;   // cr_top_level(P)
  for (int s = top; s >= 1; s /= 2)
    for (int j = 0; j < P; ++j)
      if (cr_frozen(j, s)) u[j] = cr_back(row[j], Gf[j], j >= s ? u[j - s] : z2, j + s < P ? u[j + s] : z2);
  }
  for (int j = 0; j < P; ++j) {
    const int e0 = j * M;
    HostOut out{&ov[e0], &ot[e0], &oV[e0], &oM[e0]};
    if (FAT) {
      HostH hs{std::vector<Vec2>(M)};
      seg_solve_pf<M, RZ, 3>(st[j], accp[j], u[j], j + 1 < P ? u[j + 1] : Vec2{0, 0}, out, hs);
    } else {
      seg_solve<M, RZ>(st[j], acc[j], u[j], j + 1 < P ? u[j + 1] : Vec2{0, 0}, out);
    }
  }
  for (int n = 0; n < N; ++n) { v[n] = bad ? NAN : ov[n]; th[n] = bad ? NAN : ot[n]; }
  for (int e = 0; e < Ne; ++e) { V[e] = bad ? NAN : oV[e]; Mz[e] = bad ? NAN : oM[e]; }
  return bad;
}

template <int P, int M, bool FAT = false>
int solve_one(int Ne, const double* x, const double* E, bool E_pe, const double* I, const uint8_t* fix,
              const double* Fy, const double* wy, bool w_pe, double* v, double* th, double* V, double* Mz) {
  bool rz = false;   // the kernel takes the general path when any lane of the wave sees a fixed rotation
  for (int n = 0; n <= Ne; ++n) rz |= (fix[n] & 2) != 0;
  return rz ? solve_one_rz<P, M, true, FAT>(Ne, x, E, E_pe, I, fix, Fy, wy, w_pe, v, th, V, Mz)
            : solve_one_rz<P, M, false, FAT>(Ne, x, E, E_pe, I, fix, Fy, wy, w_pe, v, th, V, Mz);
}
}  // namespace

extern "C" int emul_beam_solve_batched_f64(int P, int M, int B, int Ne, const double* x, long x_bs, const double* E,
                                           long E_bs, const double* I, long I_bs, const uint8_t* fix, long fix_bs,
                                           const double* Fy, long Fy_bs, const double* wy, long wy_bs, double* v,
                                           double* theta, double* V, double* Mz, int32_t* status) {
  const int N = Ne + 1;
  if (P * M < N) return -1;
  for (int b = 0; b < B; ++b) {
    int r = -2;
#define CASE(p_, m_)                                                                                         \
  if (P == p_ && M == m_)                                                                                    \
    r = solve_one<p_, m_>(Ne, x + b * x_bs, E + b * E_bs, E_bs != 0, I + b * I_bs, fix + b * fix_bs,          \
                          Fy + b * Fy_bs, wy + b * wy_bs, wy_bs != 0, v + (long)b * N, theta + (long)b * N,   \
                          V + (long)b * Ne, Mz + (long)b * Ne);
    CASE(8, 13) CASE(16, 7) CASE(32, 4) CASE(64, 2) CASE(64, 4) CASE(64, 8) CASE(64, 16)
#undef CASE
    if (P == 6 && M == 17)   // fat-wave tiling: the prefetching phases, two-field flags, parked h_i
      r = solve_one<6, 17, true>(Ne, x + b * x_bs, E + b * E_bs, E_bs != 0, I + b * I_bs, fix + b * fix_bs, Fy + b * Fy_bs,
                                 wy + b * wy_bs, wy_bs != 0, v + (long)b * N, theta + (long)b * N, V + (long)b * Ne, Mz + (long)b * Ne);
    if (r == -2) return -2;
    if (status) status[b] = r;
  }
  return 0;
}
