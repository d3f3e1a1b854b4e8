// Analytic forces of the AMBER-style potential of tw_energy.hip, and Langevin dynamics on them: one workgroup per
// conformation (one wave up to 64 atoms, sixteen above), fp64 arithmetic, everything of a conformation (coordinates, velocities,
// forces, Born radii) in LDS.
//
// What this replaces: the hybrid moves of sample_with_model (utils/evaluation_utils.py:439-466 `openmm_step`, called from
// :558-565, :594-602, :623-626) advance a state by `num_openmm_steps` steps of the OpenMM integrator that
// simulation/md.py:213-231 builds (LangevinMiddleIntegrator, or LangevinIntegrator for the oldest datasets; 310 K,
// friction 0.3 / ps, 0.5 fs, md.py:75-93).  OpenMM is a third-party dependency that is not vendored in the reference; the
// formulas below restate its published Reference-platform algorithms (ReferenceBondForce / AngleBondIxn /
// ProperDihedralBond / LJCoulombIxn / ReferenceObc::computeBornEnergyForces, ReferenceLangevinMiddleDynamics,
// ReferenceStochasticDynamics).  Pinned: the forces against the reference's own OpenMM known-answer file
// (tests/golden/energy_kat_2olx.npz, 40 x 65 x 3 components) and against finite differences of oracle/energy_oracle.c.
// The integrators draw their own Gaussian noise (counter-based, per conformation / step / atom): trajectories are
// statistically, not bitwise, those of OpenMM.
#include "tw_common.h"

namespace tw {

#define TW_ONE_4PI_EPS0 138.935456

__device__ __forceinline__ double md_wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void pair_index(int p, int& i, int& j) {
  i = (int)((sqrt(8.0 * p + 1.0) + 1.0) * 0.5);
  while (i * (i - 1) / 2 > p) --i;
  while ((i + 1) * i / 2 <= p) ++i;
  j = p - i * (i - 1) / 2;  // j < i
}

// F[3 a + c] += v (LDS, fp64 atomics: ds_add_f64)
__device__ __forceinline__ void facc(double* F, int a, double fx, double fy, double fz) {
  atomicAdd(F + 3 * a, fx);
  atomicAdd(F + 3 * a + 1, fy);
  atomicAdd(F + 3 * a + 2, fz);
}

// Forces of one conformation.  x [3V], F [3V] (zeroed here), born / dEdB / chain [V] and excl [V*V] in LDS; the wave's
// 64 lanes share the terms.  Returns this lane's share of the potential energy (sum over lanes = E).
__device__ double amber_forces_wave(const tw_forcefield& ff, const double* x, double* F, double* born, double* dEdB,
                                    double* chain, const unsigned* excl, int lane) {
  const int V = ff.n_atoms;
  for (int i = lane; i < 3 * V; i += 64) F[i] = 0.0;
  for (int i = lane; i < V; i += 64) dEdB[i] = 0.0;
  __syncthreads();
  double e = 0.0;
  // HarmonicBondForce
  for (int b = lane; b < ff.n_bonds; b += 64) {
    const int i = ff.bond_idx[2 * b], j = ff.bond_idx[2 * b + 1];
    const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    const double d = r - ff.bond_par[2 * b], k = ff.bond_par[2 * b + 1];
    e += 0.5 * k * d * d;
    const double g = -k * d / r;  // force on i = g * (xi - xj)
    facc(F, i, g * dx, g * dy, g * dz);
    facc(F, j, -g * dx, -g * dy, -g * dz);
  }
  // HarmonicAngleForce: theta between v0 = xi - xj and v1 = xk - xj; grad_i theta = (v0 x p) / (|v0|^2 |p|), p = v0 x v1
  for (int a = lane; a < ff.n_angles; a += 64) {
    const int i = ff.angle_idx[3 * a], j = ff.angle_idx[3 * a + 1], k = ff.angle_idx[3 * a + 2];
    double v0[3], v1[3];
    for (int c = 0; c < 3; ++c) { v0[c] = x[3 * i + c] - x[3 * j + c]; v1[c] = x[3 * k + c] - x[3 * j + c]; }
    const double d00 = v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2];
    const double d11 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    const double d01 = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
    double cs = d01 / sqrt(d00 * d11);
    cs = fmin(1.0, fmax(-1.0, cs));
    const double d = acos(cs) - ff.angle_par[2 * a], kk = ff.angle_par[2 * a + 1];
    e += 0.5 * kk * d * d;
    const double dE = kk * d;
    const double p[3] = {v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]};
    double rp = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    if (rp < 1e-12) rp = 1e-12;
    const double ta = -dE / (d00 * rp), tc = dE / (d11 * rp);
    const double fi[3] = {ta * (v0[1] * p[2] - v0[2] * p[1]), ta * (v0[2] * p[0] - v0[0] * p[2]), ta * (v0[0] * p[1] - v0[1] * p[0])};
    const double fk[3] = {tc * (v1[1] * p[2] - v1[2] * p[1]), tc * (v1[2] * p[0] - v1[0] * p[2]), tc * (v1[0] * p[1] - v1[1] * p[0])};
    facc(F, i, fi[0], fi[1], fi[2]);
    facc(F, k, fk[0], fk[1], fk[2]);
    facc(F, j, -fi[0] - fk[0], -fi[1] - fk[1], -fi[2] - fk[2]);
  }
  // PeriodicTorsionForce (the dihedral and its sign exactly as tw_energy.hip; gradient as OpenMM's ReferenceProperDihedralBond)
  for (int t = lane; t < ff.n_torsions; t += 64) {
    const int a = ff.torsion_idx[4 * t], b = ff.torsion_idx[4 * t + 1], c = ff.torsion_idx[4 * t + 2], d = ff.torsion_idx[4 * t + 3];
    double r0[3], r1[3], r2[3];
    for (int q = 0; q < 3; ++q) {
      r0[q] = x[3 * a + q] - x[3 * b + q];
      r1[q] = x[3 * c + q] - x[3 * b + q];
      r2[q] = x[3 * c + q] - x[3 * d + q];
    }
    const double c0[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    const double c1[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
    const double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
    const double dt = c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2];
    double cs = dt / sqrt(n0 * n1);
    cs = fmin(1.0, fmax(-1.0, cs));
    double phi = acos(cs);
    if (r0[0] * c1[0] + r0[1] * c1[1] + r0[2] * c1[2] < 0) phi = -phi;
    const double per = ff.torsion_par[3 * t], phase = ff.torsion_par[3 * t + 1], kk = ff.torsion_par[3 * t + 2];
    e += kk * (1.0 + cos(per * phi - phase));
    const double dE = -kk * per * sin(per * phi - phase);
    const double nbc2 = r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2], nbc = sqrt(nbc2);
    const double f0 = (-dE * nbc) / fmax(n0, 1e-24), f3 = (dE * nbc) / fmax(n1, 1e-24);
    const double f1 = (r0[0] * r1[0] + r0[1] * r1[1] + r0[2] * r1[2]) / nbc2;
    const double f2 = (r2[0] * r1[0] + r2[1] * r1[1] + r2[2] * r1[2]) / nbc2;
    double fa[3], fd[3], fb[3], fc[3];
    for (int q = 0; q < 3; ++q) {
      fa[q] = f0 * c0[q];
      fd[q] = f3 * c1[q];
      const double s = f1 * fa[q] - f2 * fd[q];
      fb[q] = fa[q] - s;
      fc[q] = fd[q] + s;
    }
    facc(F, a, fa[0], fa[1], fa[2]);
    facc(F, b, -fb[0], -fb[1], -fb[2]);
    facc(F, c, -fc[0], -fc[1], -fc[2]);
    facc(F, d, fd[0], fd[1], fd[2]);
  }
  // NonbondedForce exceptions
  for (int ex = lane; ex < ff.n_exceptions; ex += 64) {
    const double qq = ff.exc_par[3 * ex], sig = ff.exc_par[3 * ex + 1], eps = ff.exc_par[3 * ex + 2];
    if (qq == 0.0 && eps == 0.0) continue;
    const int i = ff.exc_idx[2 * ex], j = ff.exc_idx[2 * ex + 1];
    const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2);
    const double sr2 = (sig / r) * (sig / r), sr6 = sr2 * sr2 * sr2;
    e += TW_ONE_4PI_EPS0 * qq / r + 4.0 * eps * (sr6 * sr6 - sr6);
    // -dE/dr / r
    const double g = (TW_ONE_4PI_EPS0 * qq / r + 4.0 * eps * (12.0 * sr6 * sr6 - 6.0 * sr6)) / r2;
    facc(F, i, g * dx, g * dy, g * dz);
    facc(F, j, -g * dx, -g * dy, -g * dz);
  }
  // NonbondedForce pairs
  const bool use_cut = ff.cutoff > 0.0;
  const double rc = ff.cutoff;
  const double krf = use_cut ? (1.0 / (rc * rc * rc)) * (ff.rf_dielectric - 1.0) / (2.0 * ff.rf_dielectric + 1.0) : 0.0;
  const double crf = use_cut ? (1.0 / rc) * (3.0 * ff.rf_dielectric) / (2.0 * ff.rf_dielectric + 1.0) : 0.0;
  const int npairs = V * (V - 1) / 2;
  for (int p = lane; p < npairs; p += 64) {
    int i, j;
    pair_index(p, i, j);
    if (excl_test(excl, i * V + j)) continue;
    const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2);
    if (use_cut && r >= rc) continue;
    const double* pi = ff.atom_par + 5 * i;
    const double* pj = ff.atom_par + 5 * j;
    const double sig = 0.5 * (pi[1] + pj[1]), eps = sqrt(pi[2] * pj[2]);
    const double sr2 = (sig * sig) / r2, sr6 = sr2 * sr2 * sr2;
    const double qq = TW_ONE_4PI_EPS0 * pi[0] * pj[0];
    e += 4.0 * eps * (sr6 * sr6 - sr6) + qq * (use_cut ? (1.0 / r + krf * r2 - crf) : 1.0 / r);
    const double g = 4.0 * eps * (12.0 * sr6 * sr6 - 6.0 * sr6) / r2 + qq * (1.0 / (r2 * r) - (use_cut ? 2.0 * krf : 0.0));
    facc(F, i, g * dx, g * dy, g * dz);
    facc(F, j, -g * dx, -g * dy, -g * dz);
  }
  if (ff.has_gbsa) {
    const double offset = 0.009, probe = 0.14;
    const double alpha = ff.has_gbsa == 2 ? 0.8 : 1.0, beta = ff.has_gbsa == 2 ? 0.0 : 0.8, gamma = ff.has_gbsa == 2 ? 2.909125 : 4.85;
    // Born radii and dB_i / dI_i (I_i = the pair sum below)
    for (int i = lane; i < V; i += 64) {
      const double rad_i = ff.atom_par[5 * i + 3], off_i = rad_i - offset;
      double sum = 0.0;
      for (int j = 0; j < V; ++j) {
        if (j == i) continue;
        const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        if (use_cut && r > rc) continue;
        const double sr_j = (ff.atom_par[5 * j + 3] - offset) * ff.atom_par[5 * j + 4];
        const double r_sr = r + sr_j;
        if (off_i < r_sr) {
          const double rinv = 1.0 / r, ad = fabs(r - sr_j);
          const double l = 1.0 / (off_i > ad ? off_i : ad), u = 1.0 / r_sr, l2 = l * l, u2 = u * u;
          double term = l - u + 0.25 * r * (u2 - l2) + 0.5 * rinv * log(u / l) + 0.25 * sr_j * sr_j * rinv * (l2 - u2);
          if (off_i < (sr_j - r)) term += 2.0 * (1.0 / off_i - l);
          sum += term;
        }
      }
      const double s = 0.5 * off_i * sum, s2 = s * s;
      const double th = tanh(alpha * s - beta * s2 + gamma * s * s2);
      const double B = 1.0 / (1.0 / off_i - th / rad_i);
      born[i] = B;
      chain[i] = B * B * (1.0 - th * th) * (alpha - 2.0 * beta * s + 3.0 * gamma * s2) * 0.5 * off_i / rad_i;  // dB / dI
    }
    __syncthreads();
    const double pre = -TW_ONE_4PI_EPS0 * (1.0 / ff.solute_dielectric - 1.0 / ff.solvent_dielectric);
    const double pi4a = 4.0 * 3.14159265358979323846 * ff.surface_area_energy;
    for (int i = lane; i < V; i += 64) {
      const double rad = ff.atom_par[5 * i + 3], B = born[i], q = ff.atom_par[5 * i];
      double dB = 0.0;
      if (B > 0.0) {
        const double rr = rad + probe, ratio = rad / B, r3 = ratio * ratio * ratio;
        const double ace = pi4a * rr * rr * r3 * r3;
        e += ace;
        dB += -6.0 * ace / B;
      }
      e += 0.5 * pre * q * q / B;
      dB += -0.5 * pre * q * q / (B * B);
      atomicAdd(dEdB + i, dB);
    }
    for (int p = lane; p < npairs; p += 64) {
      int i, j;
      pair_index(p, i, j);
      const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      const double r2 = dx * dx + dy * dy + dz * dz;
      if (use_cut && sqrt(r2) > rc) continue;
      const double a2 = born[i] * born[j], D = r2 / (4.0 * a2), ex = exp(-D);
      const double den2 = r2 + a2 * ex, den = sqrt(den2);
      const double qq = pre * ff.atom_par[5 * i] * ff.atom_par[5 * j];
      e += qq / den - (use_cut ? qq / rc : 0.0);
      const double inv3 = qq / (den2 * den);
      const double g = inv3 * (1.0 - 0.25 * ex);            // -dE/dr / r
      facc(F, i, g * dx, g * dy, g * dz);
      facc(F, j, -g * dx, -g * dy, -g * dz);
      const double dEda = -0.5 * inv3 * ex * (1.0 + D);     // dE / d(B_i B_j)
      atomicAdd(dEdB + i, dEda * born[j]);
      atomicAdd(dEdB + j, dEda * born[i]);
    }
    __syncthreads();
    // chain rule through the Born radii: E depends on x through I_i(r_ij)
    for (int p = lane; p < V * V; p += 64) {
      const int i = p / V, j = p - i * V;
      if (i == j) continue;
      const double rad_i = ff.atom_par[5 * i + 3], off_i = rad_i - offset;
      const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      const double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2);
      if (use_cut && r > rc) continue;
      const double s = (ff.atom_par[5 * j + 3] - offset) * ff.atom_par[5 * j + 4];
      if (!(off_i < r + s)) continue;
      const double ad = fabs(r - s);
      const bool moving = ad >= off_i;                       // l = 1 / |r - s| follows r; otherwise l = 1 / off_i is constant
      const double l = 1.0 / (moving ? ad : off_i), u = 1.0 / (r + s), l2 = l * l, u2 = u * u;
      const double dl = moving ? -l2 * (r >= s ? 1.0 : -1.0) : 0.0, du = -u2;
      double dT = dl - du + 0.25 * (u2 - l2) + 0.5 * r * (u * du - l * dl) - 0.5 * log(u / l) / r2 + 0.5 / r * (du / u - dl / l) -
                  0.25 * s * s / r2 * (l2 - u2) + 0.5 * s * s / r * (l * dl - u * du);
      if (off_i < (s - r)) dT += -2.0 * dl;
      const double g = -dEdB[i] * chain[i] * dT / r;         // force on i along (xi - xj)
      facc(F, i, g * dx, g * dy, g * dz);
      facc(F, j, -g * dx, -g * dy, -g * dz);
    }
  }
  __syncthreads();
  return e;
}

// The same forces with MD_W waves per conformation (r06: molecules above 64 atoms - the reference's 691-atom test protein took 12.5 ms
// per evaluation on one wave, i.e. per Langevin step).  No cross-wave atomics, so that a trajectory stays a function of its seed:
//   A  nonbonded pairs: a wave per atom i, lanes over ALL partners j (every pair is evaluated from both ends - twice the arithmetic of
//      the triangular loop, sixteen times the lanes), butterfly over the wave, F[i] = the row sum (a plain store: one owner per atom);
//      the Born radii the same way
//   B  bonded terms and 1-4 exceptions: wave 0 alone, LDS atomics as in the single-wave kernel (one wave: a fixed order)
//   C  GB pair terms: a wave per atom again - F[i] += row sum, dE/dB_i = self terms + row sum
//   D  the chain rule through the Born radii: atom a collects g_aj + g_ja from its row
// The energy: every pair counted at its j < i end; per-thread partial sums, wave butterflies, the waves' sums added in wave order.
#define MD_W 16
__device__ double amber_forces_block(const tw_forcefield& ff, const double* x, double* F, double* born, double* dEdB, double* chain,
                                     const unsigned* excl, double* part, int tid) {
  const int V = ff.n_atoms, lane = tid & 63, wave = tid >> 6;
  double e = 0.0;
  const bool use_cut = ff.cutoff > 0.0;
  const double rc = ff.cutoff;
  const double krf = use_cut ? (1.0 / (rc * rc * rc)) * (ff.rf_dielectric - 1.0) / (2.0 * ff.rf_dielectric + 1.0) : 0.0;
  const double crf = use_cut ? (1.0 / rc) * (3.0 * ff.rf_dielectric) / (2.0 * ff.rf_dielectric + 1.0) : 0.0;
  const double offset = 0.009, probe = 0.14;
  const double alpha = ff.has_gbsa == 2 ? 0.8 : 1.0, beta = ff.has_gbsa == 2 ? 0.0 : 0.8, gamma = ff.has_gbsa == 2 ? 2.909125 : 4.85;
  // ---- A: nonbonded pairs and Born radii, a wave per atom
  for (int i = wave; i < V; i += MD_W) {
    const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    const double* pi = ff.atom_par + 5 * i;
    const double rad_i = pi[3], off_i = rad_i - offset;
    double fx = 0.0, fy = 0.0, fz = 0.0, bsum = 0.0;
    for (int j = lane; j < V; j += 64) {
      if (j == i) continue;
      const double dx = xi - x[3 * j], dy = yi - x[3 * j + 1], dz = zi - x[3 * j + 2];
      const double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2);
      const double* pj = ff.atom_par + 5 * j;
      if (!excl_test(excl, i * V + j) && !(use_cut && r >= rc)) {
        const double sig = 0.5 * (pi[1] + pj[1]), eps = sqrt(pi[2] * pj[2]);
        const double sr2 = (sig * sig) / r2, sr6 = sr2 * sr2 * sr2;
        const double qq = TW_ONE_4PI_EPS0 * pi[0] * pj[0];
        if (j < i) e += 4.0 * eps * (sr6 * sr6 - sr6) + qq * (use_cut ? (1.0 / r + krf * r2 - crf) : 1.0 / r);
        const double g = 4.0 * eps * (12.0 * sr6 * sr6 - 6.0 * sr6) / r2 + qq * (1.0 / (r2 * r) - (use_cut ? 2.0 * krf : 0.0));
        fx += g * dx;
        fy += g * dy;
        fz += g * dz;
      }
      if (ff.has_gbsa && !(use_cut && r > rc)) {
        const double sr_j = (pj[3] - offset) * pj[4];
        const double r_sr = r + sr_j;
        if (off_i < r_sr) {
          const double rinv = 1.0 / r, ad = fabs(r - sr_j);
          const double l = 1.0 / (off_i > ad ? off_i : ad), u = 1.0 / r_sr, l2 = l * l, u2 = u * u;
          double term = l - u + 0.25 * r * (u2 - l2) + 0.5 * rinv * log(u / l) + 0.25 * sr_j * sr_j * rinv * (l2 - u2);
          if (off_i < (sr_j - r)) term += 2.0 * (1.0 / off_i - l);
          bsum += term;
        }
      }
    }
    fx = md_wsum(fx);
    fy = md_wsum(fy);
    fz = md_wsum(fz);
    bsum = md_wsum(bsum);
    if (lane == 0) {
      F[3 * i] = fx;
      F[3 * i + 1] = fy;
      F[3 * i + 2] = fz;
      if (ff.has_gbsa) {
        const double sb = 0.5 * off_i * bsum, s2 = sb * sb;
        const double th = tanh(alpha * sb - beta * s2 + gamma * sb * s2);
        const double B = 1.0 / (1.0 / off_i - th / rad_i);
        born[i] = B;
        chain[i] = B * B * (1.0 - th * th) * (alpha - 2.0 * beta * sb + 3.0 * gamma * s2) * 0.5 * off_i / rad_i;  // dB / dI
      }
    }
  }
  __syncthreads();
  // ---- B: bonded terms and exceptions on wave 0 (the single-wave kernel's loops)
  if (wave == 0) {
    for (int b = lane; b < ff.n_bonds; b += 64) {
      const int i = ff.bond_idx[2 * b], j = ff.bond_idx[2 * b + 1];
      const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      const double d = r - ff.bond_par[2 * b], k = ff.bond_par[2 * b + 1];
      e += 0.5 * k * d * d;
      const double g = -k * d / r;
      facc(F, i, g * dx, g * dy, g * dz);
      facc(F, j, -g * dx, -g * dy, -g * dz);
    }
    for (int a = lane; a < ff.n_angles; a += 64) {
      const int i = ff.angle_idx[3 * a], j = ff.angle_idx[3 * a + 1], k = ff.angle_idx[3 * a + 2];
      double v0[3], v1[3];
      for (int c = 0; c < 3; ++c) { v0[c] = x[3 * i + c] - x[3 * j + c]; v1[c] = x[3 * k + c] - x[3 * j + c]; }
      const double d00 = v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2];
      const double d11 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
      const double d01 = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
      double cs = d01 / sqrt(d00 * d11);
      cs = fmin(1.0, fmax(-1.0, cs));
      const double d = acos(cs) - ff.angle_par[2 * a], kk = ff.angle_par[2 * a + 1];
      e += 0.5 * kk * d * d;
      const double dE = kk * d;
      const double p[3] = {v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]};
      double rp = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      if (rp < 1e-12) rp = 1e-12;
      const double ta = -dE / (d00 * rp), tc = dE / (d11 * rp);
      const double fi[3] = {ta * (v0[1] * p[2] - v0[2] * p[1]), ta * (v0[2] * p[0] - v0[0] * p[2]), ta * (v0[0] * p[1] - v0[1] * p[0])};
      const double fk[3] = {tc * (v1[1] * p[2] - v1[2] * p[1]), tc * (v1[2] * p[0] - v1[0] * p[2]), tc * (v1[0] * p[1] - v1[1] * p[0])};
      facc(F, i, fi[0], fi[1], fi[2]);
      facc(F, k, fk[0], fk[1], fk[2]);
      facc(F, j, -fi[0] - fk[0], -fi[1] - fk[1], -fi[2] - fk[2]);
    }
    for (int t = lane; t < ff.n_torsions; t += 64) {
      const int a = ff.torsion_idx[4 * t], b = ff.torsion_idx[4 * t + 1], c = ff.torsion_idx[4 * t + 2], d = ff.torsion_idx[4 * t + 3];
      double r0[3], r1[3], r2[3];
      for (int q = 0; q < 3; ++q) {
        r0[q] = x[3 * a + q] - x[3 * b + q];
        r1[q] = x[3 * c + q] - x[3 * b + q];
        r2[q] = x[3 * c + q] - x[3 * d + q];
      }
      const double c0[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
      const double c1[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
      const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
      const double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
      const double dt = c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2];
      double cs = dt / sqrt(n0 * n1);
      cs = fmin(1.0, fmax(-1.0, cs));
      double phi = acos(cs);
      if (r0[0] * c1[0] + r0[1] * c1[1] + r0[2] * c1[2] < 0) phi = -phi;
      const double per = ff.torsion_par[3 * t], phase = ff.torsion_par[3 * t + 1], kk = ff.torsion_par[3 * t + 2];
      e += kk * (1.0 + cos(per * phi - phase));
      const double dE = -kk * per * sin(per * phi - phase);
      const double nbc2 = r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2], nbc = sqrt(nbc2);
      const double f0 = (-dE * nbc) / fmax(n0, 1e-24), f3 = (dE * nbc) / fmax(n1, 1e-24);
      const double f1 = (r0[0] * r1[0] + r0[1] * r1[1] + r0[2] * r1[2]) / nbc2;
      const double f2 = (r2[0] * r1[0] + r2[1] * r1[1] + r2[2] * r1[2]) / nbc2;
      double fa[3], fd[3], fb[3], fc[3];
      for (int q = 0; q < 3; ++q) {
        fa[q] = f0 * c0[q];
        fd[q] = f3 * c1[q];
        const double sq = f1 * fa[q] - f2 * fd[q];
        fb[q] = fa[q] - sq;
        fc[q] = fd[q] + sq;
      }
      facc(F, a, fa[0], fa[1], fa[2]);
      facc(F, b, -fb[0], -fb[1], -fb[2]);
      facc(F, c, -fc[0], -fc[1], -fc[2]);
      facc(F, d, fd[0], fd[1], fd[2]);
    }
    for (int ex = lane; ex < ff.n_exceptions; ex += 64) {
      const double qq = ff.exc_par[3 * ex], sig = ff.exc_par[3 * ex + 1], eps = ff.exc_par[3 * ex + 2];
      if (qq == 0.0 && eps == 0.0) continue;
      const int i = ff.exc_idx[2 * ex], j = ff.exc_idx[2 * ex + 1];
      const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      const double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2);
      const double sr2 = (sig / r) * (sig / r), sr6 = sr2 * sr2 * sr2;
      e += TW_ONE_4PI_EPS0 * qq / r + 4.0 * eps * (sr6 * sr6 - sr6);
      const double g = (TW_ONE_4PI_EPS0 * qq / r + 4.0 * eps * (12.0 * sr6 * sr6 - 6.0 * sr6)) / r2;
      facc(F, i, g * dx, g * dy, g * dz);
      facc(F, j, -g * dx, -g * dy, -g * dz);
    }
  }
  __syncthreads();
  if (ff.has_gbsa) {
    const double pre = -TW_ONE_4PI_EPS0 * (1.0 / ff.solute_dielectric - 1.0 / ff.solvent_dielectric);
    const double pi4a = 4.0 * 3.14159265358979323846 * ff.surface_area_energy;
    // ---- C: GB self and pair terms, a wave per atom
    for (int i = wave; i < V; i += MD_W) {
      const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
      const double Bi = born[i], qi = ff.atom_par[5 * i];
      double fx = 0.0, fy = 0.0, fz = 0.0, dB = 0.0;
      for (int j = lane; j < V; j += 64) {
        if (j == i) continue;
        const double dx = xi - x[3 * j], dy = yi - x[3 * j + 1], dz = zi - x[3 * j + 2];
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (use_cut && sqrt(r2) > rc) continue;
        const double a2 = Bi * born[j], D = r2 / (4.0 * a2), ex = exp(-D);
        const double den2 = r2 + a2 * ex, den = sqrt(den2);
        const double qq = pre * qi * ff.atom_par[5 * j];
        if (j < i) e += qq / den - (use_cut ? qq / rc : 0.0);
        const double inv3 = qq / (den2 * den);
        const double g = inv3 * (1.0 - 0.25 * ex);            // -dE/dr / r
        fx += g * dx;
        fy += g * dy;
        fz += g * dz;
        dB += -0.5 * inv3 * ex * (1.0 + D) * born[j];         // dE / d(B_i B_j) x B_j
      }
      fx = md_wsum(fx);
      fy = md_wsum(fy);
      fz = md_wsum(fz);
      dB = md_wsum(dB);
      if (lane == 0) {
        const double rad = ff.atom_par[5 * i + 3];
        if (Bi > 0.0) {
          const double rr = rad + probe, ratio = rad / Bi, r3 = ratio * ratio * ratio;
          const double ace = pi4a * rr * rr * r3 * r3;
          e += ace;
          dB += -6.0 * ace / Bi;
        }
        e += 0.5 * pre * qi * qi / Bi;
        dB += -0.5 * pre * qi * qi / (Bi * Bi);
        dEdB[i] = dB;
        F[3 * i] += fx;
        F[3 * i + 1] += fy;
        F[3 * i + 2] += fz;
      }
    }
    __syncthreads();
    // ---- D: chain rule through the Born radii; the ordered pair (i, j) pushes i by g_ij (xi - xj) and j by the opposite, so atom a
    //         collects (g_aj + g_ja) (xa - xj) over its row
    for (int a = wave; a < V; a += MD_W) {
      const double xa = x[3 * a], ya = x[3 * a + 1], za = x[3 * a + 2];
      const double off_a = ff.atom_par[5 * a + 3] - offset, s_a = off_a * ff.atom_par[5 * a + 4];
      const double w_a = dEdB[a] * chain[a];
      double fx = 0.0, fy = 0.0, fz = 0.0;
      // dT(off of the atom whose radius is integrated, s of the partner, r)
      auto dterm = [&](double off_i, double sp, double r, double r2) -> double {
        if (!(off_i < r + sp)) return 0.0;
        const double ad = fabs(r - sp);
        const bool moving = ad >= off_i;
        const double l = 1.0 / (moving ? ad : off_i), u = 1.0 / (r + sp), l2 = l * l, u2 = u * u;
        const double dl = moving ? -l2 * (r >= sp ? 1.0 : -1.0) : 0.0, du = -u2;
        double dT = dl - du + 0.25 * (u2 - l2) + 0.5 * r * (u * du - l * dl) - 0.5 * log(u / l) / r2 + 0.5 / r * (du / u - dl / l) -
                    0.25 * sp * sp / r2 * (l2 - u2) + 0.5 * sp * sp / r * (l * dl - u * du);
        if (off_i < (sp - r)) dT += -2.0 * dl;
        return dT;
      };
      for (int j = lane; j < V; j += 64) {
        if (j == a) continue;
        const double dx = xa - x[3 * j], dy = ya - x[3 * j + 1], dz = za - x[3 * j + 2];
        const double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2);
        if (use_cut && r > rc) continue;
        const double off_j = ff.atom_par[5 * j + 3] - offset, s_j = off_j * ff.atom_par[5 * j + 4];
        const double g = -(w_a * dterm(off_a, s_j, r, r2) + dEdB[j] * chain[j] * dterm(off_j, s_a, r, r2)) / r;
        fx += g * dx;
        fy += g * dy;
        fz += g * dz;
      }
      fx = md_wsum(fx);
      fy = md_wsum(fy);
      fz = md_wsum(fz);
      if (lane == 0) {
        F[3 * a] += fx;
        F[3 * a + 1] += fy;
        F[3 * a + 2] += fz;
      }
    }
  }
  // ---- the energy: waves' sums in wave order
  e = md_wsum(e);
  if (lane == 0) part[wave] = e;
  __syncthreads();
  double total = 0.0;
  for (int w = 0; w < MD_W; ++w) total += part[w];
  return total;
}

struct MdLds {
  double *x, *F, *born, *dEdB, *chain, *v, *part;
  unsigned* excl;
};
__device__ __forceinline__ MdLds md_carve(double* smd, int V, bool with_v) {
  MdLds m;
  m.x = smd;
  m.F = m.x + 3 * V;
  m.born = m.F + 3 * V;
  m.dEdB = m.born + V;
  m.chain = m.dEdB + V;
  m.v = m.chain + V;
  m.part = m.v + (with_v ? 3 * V : 0);
  m.excl = (unsigned*)(m.part + MD_W);
  return m;
}
static size_t md_lds_bytes(int V, bool with_v) {
  size_t b = ((size_t)(9 + (with_v ? 3 : 0)) * V + MD_W) * sizeof(double) + excl_bytes(V);
  return (b + 15) / 16 * 16;
}
// W = 1: one wave per conformation (up to 64 atoms); W = MD_W: amber_forces_block
template <int W>
__global__ void __launch_bounds__(64 * W) amber_forces_kernel(const tw_forcefield ff, const float* __restrict__ coords,
                                                               double* __restrict__ out_energy, double* __restrict__ out_forces) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  constexpr int NTH = 64 * W;
  const int V = ff.n_atoms, lane = threadIdx.x;
  const int64_t n = blockIdx.x;
  MdLds m = md_carve(smd, V, false);
  for (int i = lane; i < 3 * V; i += NTH) m.x[i] = (double)coords[n * 3 * V + i];
  excl_fill(ff.exc_idx, ff.n_exceptions, V, m.excl, lane, NTH);
  double e;
  if constexpr (W == 1) e = md_wsum(amber_forces_wave(ff, m.x, m.F, m.born, m.dEdB, m.chain, m.excl, lane));
  else e = amber_forces_block(ff, m.x, m.F, m.born, m.dEdB, m.chain, m.excl, m.part, lane);
  __syncthreads();
  if (out_energy && lane == 0) out_energy[n] = e;
  for (int i = lane; i < 3 * V; i += NTH) out_forces[n * 3 * V + i] = m.F[i];
}

// counter-based standard normal: splitmix64 of (seed, conformation, step, atom-component) -> two uniforms -> Box-Muller
__device__ __forceinline__ unsigned long long md_mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double md_normal(unsigned long long seed, long long n, int step, int idx) {
  const unsigned long long k = md_mix(seed ^ md_mix((unsigned long long)n * 0x100000001B3ull + (unsigned long long)step) ^
                                      md_mix(0xD6E8FEB86659FD93ull * (unsigned long long)(idx + 1)));
  const unsigned long long k2 = md_mix(k);
  const double u1 = ((double)(k >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0, 1)
  const double u2 = (double)(k2 >> 11) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// scheme 0: LangevinMiddleIntegrator (leapfrog "LF-middle": v += dt F / m; x += dt/2 v; v <- a v + sqrt(1 - a^2) sqrt(kT/m) N;
//           x += dt/2 v;  a = exp(-friction dt));  velocities live at the half step, as in OpenMM
// scheme 1: LangevinIntegrator of OpenMM <= 7.x:  v <- a v + (1 - a) / friction F / m + sqrt(kT (1 - a^2) / m) N;  x += dt v
// friction == 0 in either scheme is plain leapfrog (used by the energy-conservation test).
template <int W>
__global__ void __launch_bounds__(64 * W) langevin_kernel(const tw_forcefield ff, const float* __restrict__ masses, float* __restrict__ coords,
                                                           float* __restrict__ velocs, int n_steps, double dt, double friction, double kbT,
                                                           int scheme, unsigned long long seed, long long step0, double* __restrict__ out_energy) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  constexpr int NTH = 64 * W;
  const int V = ff.n_atoms, lane = threadIdx.x;
  const int64_t n = blockIdx.x;
  MdLds m = md_carve(smd, V, true);
  for (int i = lane; i < 3 * V; i += NTH) {
    m.x[i] = (double)coords[n * 3 * V + i];
    m.v[i] = (double)velocs[n * 3 * V + i];
  }
  excl_fill(ff.exc_idx, ff.n_exceptions, V, m.excl, lane, NTH);
  const double a = friction > 0.0 ? exp(-friction * dt) : 1.0;
  const double fscale = friction > 0.0 ? (1.0 - a) / friction : dt;
  double e = 0.0;
  for (int s = 0; s < n_steps; ++s) {
    if constexpr (W == 1) e = amber_forces_wave(ff, m.x, m.F, m.born, m.dEdB, m.chain, m.excl, lane);
    else e = amber_forces_block(ff, m.x, m.F, m.born, m.dEdB, m.chain, m.excl, m.part, lane);
    __syncthreads();
    for (int i = lane; i < 3 * V; i += NTH) {
      const double mass = (double)masses[i / 3];
      const double noise = friction > 0.0 ? md_normal(seed, n, (int)(step0 + s), i) : 0.0;
      double v = m.v[i], xx = m.x[i];
      if (scheme == 0) {
        v += dt * m.F[i] / mass;
        xx += 0.5 * dt * v;
        v = a * v + sqrt((1.0 - a * a) * kbT / mass) * noise;
        xx += 0.5 * dt * v;
      } else {
        v = a * v + fscale * m.F[i] / mass + sqrt(kbT * (1.0 - a * a) / mass) * noise;
        xx += dt * v;
      }
      m.v[i] = v;
      m.x[i] = xx;
    }
    __syncthreads();
  }
  for (int i = lane; i < 3 * V; i += NTH) {
    coords[n * 3 * V + i] = (float)m.x[i];
    velocs[n * 3 * V + i] = (float)m.v[i];
  }
  if (out_energy) {  // potential energy at the positions the LAST force evaluation saw (before the last update)
    if constexpr (W == 1) e = md_wsum(e);
    if (lane == 0) out_energy[n] = e;
  }
}

int amber_energy_forces(const tw_forcefield* ff, const float* coords, double* out_energy, double* out_forces, int64_t n, hipStream_t s) {
  if (n == 0) return TW_OK;
  const size_t shm = md_lds_bytes(ff->n_atoms, false);
  TW_REQUIRE(shm <= (size_t)160 * 1024, "force kernel: %d atoms need %zu bytes of LDS (one conformation per wave; limit 160 KiB)", ff->n_atoms, shm);
  static LdsLimit lim1, limw;
  int rc;
  if (ff->n_atoms <= 64) {   // small molecules: one wave per conformation
    if (shm > (size_t)64 * 1024 && (rc = lim1.ensure((const void*)amber_forces_kernel<1>, 160 * 1024))) return rc;
    hipLaunchKernelGGL(amber_forces_kernel<1>, dim3((unsigned)n), dim3(64), shm, s, *ff, coords, out_energy, out_forces);
  } else {
    if (shm > (size_t)64 * 1024 && (rc = limw.ensure((const void*)amber_forces_kernel<MD_W>, 160 * 1024))) return rc;
    hipLaunchKernelGGL(amber_forces_kernel<MD_W>, dim3((unsigned)n), dim3(64 * MD_W), shm, s, *ff, coords, out_energy, out_forces);
  }
  TW_LAUNCH_CHECK();
  return TW_OK;
}

int langevin_steps(const tw_forcefield* ff, const float* masses, float* coords, float* velocs, int n_steps, double dt,
                   double friction, double kbT, int scheme, unsigned long long seed, long long step0, double* out_energy,
                   int64_t n, hipStream_t s) {
  if (n == 0 || n_steps <= 0) return TW_OK;
  const size_t shm = md_lds_bytes(ff->n_atoms, true);
  TW_REQUIRE(shm <= (size_t)160 * 1024, "Langevin kernel: %d atoms need %zu bytes of LDS (one conformation per wave; limit 160 KiB)", ff->n_atoms, shm);
  static LdsLimit lim1, limw;
  int rc;
  if (ff->n_atoms <= 64) {
    if (shm > (size_t)64 * 1024 && (rc = lim1.ensure((const void*)langevin_kernel<1>, 160 * 1024))) return rc;
    hipLaunchKernelGGL(langevin_kernel<1>, dim3((unsigned)n), dim3(64), shm, s, *ff, masses, coords, velocs, n_steps, dt, friction,
                       kbT, scheme, seed, step0, out_energy);
  } else {
    if (shm > (size_t)64 * 1024 && (rc = limw.ensure((const void*)langevin_kernel<MD_W>, 160 * 1024))) return rc;
    hipLaunchKernelGGL(langevin_kernel<MD_W>, dim3((unsigned)n), dim3(64 * MD_W), shm, s, *ff, masses, coords, velocs, n_steps, dt,
                       friction, kbT, scheme, seed, step0, out_energy);
  }
  TW_LAUNCH_CHECK();
  return TW_OK;
}

}  // namespace tw
