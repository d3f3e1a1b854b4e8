/* CPU oracle for the AMBER-style potential energy.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference obtains E_pot from OpenMM 7.7 (timewarp-environment.yml:22) through bgflow
 * (utils/openmm/openmm_bridge.py:11,17,206-221,281-294; the System is built in simulation/md.py:128-187).  Neither
 * package, nor OpenMM's force-field XML files, exist in the reference tree or in this image, so this file restates the
 * published algorithms of OpenMM's forces:
 *   HarmonicBondForce      E = 1/2 k (r - r0)^2
 *   HarmonicAngleForce     E = 1/2 k (theta - theta0)^2
 *   PeriodicTorsionForce   E = k (1 + cos(n phi - phase))
 *   NonbondedForce         CutoffNonPeriodic: 4 eps [(s/r)^12 - (s/r)^6] + K q q (1/r + k_rf r^2 - c_rf)
 *                          inside the cutoff; exceptions (1-4) without cutoff / reaction field
 *   GBSAOBCForce           OBC-II Born radii (alpha 1, beta 0.8, gamma 4.85, offset 0.009 nm),
 *                          GB pair + self energy, ACE surface term 4 pi sigma (r+0.14)^2 (r/B)^6
 *
 * PINNED (has_gbsa = 1, the amber99sbildn + amber99_obc preset of alanine dipeptide and T1-peptides) against the
 * reference's own OpenMM known-answer data, simulation/testdata/implicit-2olx-traj-cpu-arrays.npz -- the file
 * simulation/tests/test_md.py:35-83 checks OpenMM with: 40 frames of the 65-atom peptide NNQQ, E_pot and forces.
 * With the parameter tables of timewarp_amd/forcefield.py this code reproduces the 40 energies to 2e-3 kJ/mol (of
 * -1690) and the 7800 force components to 0.008 kJ/mol/nm rms (of 933), the float32 noise of the file
 * (tests/test_energy_kat.py; analysis in tools/pin_energy/fit_2olx.py).  Two asparagine side-chain torsion series
 * of those tables are fitted to the file (their ILDN values could not be recalled offline); every formula in this
 * file and every other parameter is pinned independently of them (same test file).
 * NOT pinned: has_gbsa = 2 (GBSA-OBC I coefficients of implicit/obc1.xml, amber14 preset) -- no known-answer data.
 *
 * Build: make -C oracle   ->  oracle/_build/libenergy_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define ONE_4PI_EPS0 138.935456

typedef struct {
  int32_t n_atoms, n_bonds, n_angles, n_torsions, n_exceptions, has_gbsa;
  double cutoff, rf_dielectric, solute_dielectric, solvent_dielectric, surface_area_energy;
  const int32_t* bond_idx;    const double* bond_par;
  const int32_t* angle_idx;   const double* angle_par;
  const int32_t* torsion_idx; const double* torsion_par;
  const int32_t* exc_idx;     const double* exc_par;
  const double* atom_par;
} oracle_ff;

static double dist(const double* x, int i, int j) {
  double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
  return sqrt(dx * dx + dy * dy + dz * dz);
}

/* energy of one conformation; terms[5] = bond, angle, torsion, nonbonded, gbsa */
static double energy_one(const oracle_ff* ff, const double* x, double* terms) {
  const int V = ff->n_atoms;
  double eb = 0, ea = 0, et = 0, en = 0, eg = 0;
  for (int b = 0; b < ff->n_bonds; ++b) {
    double d = dist(x, ff->bond_idx[2 * b], ff->bond_idx[2 * b + 1]) - ff->bond_par[2 * b];
    eb += 0.5 * ff->bond_par[2 * b + 1] * d * d;
  }
  for (int a = 0; a < ff->n_angles; ++a) {
    int i = ff->angle_idx[3 * a], j = ff->angle_idx[3 * a + 1], k = ff->angle_idx[3 * a + 2];
    double v0[3], v1[3];
    for (int c = 0; c < 3; ++c) { v0[c] = x[3 * i + c] - x[3 * j + c]; v1[c] = x[3 * k + c] - x[3 * j + c]; }
    double d00 = v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2];
    double d11 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    double d01 = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
    double cs = d01 / sqrt(d00 * d11);
    if (cs > 1.0) cs = 1.0;
    if (cs < -1.0) cs = -1.0;
    double d = acos(cs) - ff->angle_par[2 * a];
    ea += 0.5 * ff->angle_par[2 * a + 1] * d * d;
  }
  for (int t = 0; t < ff->n_torsions; ++t) {
    int a = ff->torsion_idx[4 * t], b = ff->torsion_idx[4 * t + 1], c = ff->torsion_idx[4 * t + 2], d = ff->torsion_idx[4 * t + 3];
    double r0[3], r1[3], r2[3];
    for (int q = 0; q < 3; ++q) {
      r0[q] = x[3 * a + q] - x[3 * b + q];
      r1[q] = x[3 * c + q] - x[3 * b + q];
      r2[q] = x[3 * c + q] - x[3 * d + q];
    }
    double c0[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    double c1[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
    double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
    double cs = (c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2]) / sqrt(n0 * n1);
    if (cs > 1.0) cs = 1.0;
    if (cs < -1.0) cs = -1.0;
    double phi = acos(cs);
    if (r0[0] * c1[0] + r0[1] * c1[1] + r0[2] * c1[2] < 0) phi = -phi;
    et += ff->torsion_par[3 * t + 2] * (1.0 + cos(ff->torsion_par[3 * t] * phi - ff->torsion_par[3 * t + 1]));
  }
  unsigned char* excl = (unsigned char*)calloc((size_t)V * V, 1);
  for (int e = 0; e < ff->n_exceptions; ++e) {
    int i = ff->exc_idx[2 * e], j = ff->exc_idx[2 * e + 1];
    excl[i * V + j] = excl[j * V + i] = 1;
    double qq = ff->exc_par[3 * e], sig = ff->exc_par[3 * e + 1], eps = ff->exc_par[3 * e + 2];
    if (qq == 0.0 && eps == 0.0) continue;
    double r = dist(x, i, j);
    double sr2 = (sig / r) * (sig / r), sr6 = sr2 * sr2 * sr2;
    en += ONE_4PI_EPS0 * qq / r + 4.0 * eps * (sr6 * sr6 - sr6);
  }
  const int use_cut = ff->cutoff > 0.0;
  const double rc = ff->cutoff;
  const double krf = use_cut ? (1.0 / (rc * rc * rc)) * (ff->rf_dielectric - 1.0) / (2.0 * ff->rf_dielectric + 1.0) : 0.0;
  const double crf = use_cut ? (1.0 / rc) * (3.0 * ff->rf_dielectric) / (2.0 * ff->rf_dielectric + 1.0) : 0.0;
  for (int i = 0; i < V; ++i)
    for (int j = 0; j < i; ++j) {
      if (excl[i * V + j]) continue;
      double r = dist(x, i, j);
      if (use_cut && r >= rc) continue;
      const double* pi = ff->atom_par + 5 * i;
      const double* pj = ff->atom_par + 5 * j;
      double sig = 0.5 * (pi[1] + pj[1]), eps = sqrt(pi[2] * pj[2]);
      double sr2 = (sig * sig) / (r * r), sr6 = sr2 * sr2 * sr2;
      en += 4.0 * eps * (sr6 * sr6 - sr6);
      en += ONE_4PI_EPS0 * pi[0] * pj[0] * (use_cut ? (1.0 / r + krf * r * r - crf) : 1.0 / r);
    }
  free(excl);
  if (ff->has_gbsa) {
    const double offset = 0.009, probe = 0.14;
    /* has_gbsa 1: OBC-II (GBSAOBCForce), 2: OBC-I (the tanh coefficients of OpenMM's implicit/obc1.xml) */
    const double alpha = ff->has_gbsa == 2 ? 0.8 : 1.0, beta = ff->has_gbsa == 2 ? 0.0 : 0.8,
                 gamma = ff->has_gbsa == 2 ? 2.909125 : 4.85;
    double* born = (double*)malloc(sizeof(double) * V);
    for (int i = 0; i < V; ++i) {
      double rad_i = ff->atom_par[5 * i + 3], off_i = rad_i - offset, sum = 0.0;
      for (int j = 0; j < V; ++j) {
        if (j == i) continue;
        double r = dist(x, i, j);
        if (use_cut && r > rc) continue;
        double off_j = ff->atom_par[5 * j + 3] - offset, sr_j = off_j * ff->atom_par[5 * j + 4], r_sr = r + sr_j;
        if (off_i < r_sr) {
          double rinv = 1.0 / r, ad = fabs(r - sr_j);
          double l = 1.0 / (off_i > ad ? off_i : ad), u = 1.0 / r_sr;
          double l2 = l * l, u2 = u * u, ratio = log(u / l);
          double term = l - u + 0.25 * r * (u2 - l2) + 0.5 * rinv * ratio + 0.25 * sr_j * sr_j * rinv * (l2 - u2);
          if (off_i < (sr_j - r)) term += 2.0 * (1.0 / off_i - l);
          sum += term;
        }
      }
      sum *= 0.5 * off_i;
      double s2 = sum * sum, s3 = sum * s2;
      born[i] = 1.0 / (1.0 / off_i - tanh(alpha * sum - beta * s2 + gamma * s3) / rad_i);
    }
    const double pre = -ONE_4PI_EPS0 * (1.0 / ff->solute_dielectric - 1.0 / ff->solvent_dielectric);
    const double pi4a = 4.0 * 3.14159265358979323846 * ff->surface_area_energy;
    for (int i = 0; i < V; ++i) {
      double rad = ff->atom_par[5 * i + 3], q = ff->atom_par[5 * i];
      if (born[i] > 0.0) {
        double rr = rad + probe, ratio = rad / born[i], r3 = ratio * ratio * ratio;
        eg += pi4a * rr * rr * r3 * r3;
      }
      eg += 0.5 * pre * q * q / born[i];
      for (int j = 0; j < i; ++j) {
        double r = dist(x, i, j), r2 = r * r;
        if (use_cut && r > rc) continue;
        double a2 = born[i] * born[j];
        double den = sqrt(r2 + a2 * exp(-r2 / (4.0 * a2)));
        double qq = pre * q * ff->atom_par[5 * j];
        double e = qq / den;
        if (use_cut) e -= qq / rc;
        eg += e;
      }
    }
    free(born);
  }
  if (terms) { terms[0] = eb; terms[1] = ea; terms[2] = et; terms[3] = en; terms[4] = eg; }
  return eb + ea + et + en + eg;
}

/* coords [n_rows, n_atoms, 3] float32 (as the bridge receives them) -> out [n_rows] (kJ/mol) */
int oracle_amber_energy(const oracle_ff* ff, const float* coords, double* out, double* terms, int64_t n_rows) {
  const int V = ff->n_atoms;
  double* x = (double*)malloc(sizeof(double) * 3 * V);
  for (int64_t n = 0; n < n_rows; ++n) {
    for (int i = 0; i < 3 * V; ++i) x[i] = (double)coords[n * 3 * V + i];
    out[n] = energy_one(ff, x, terms ? terms + 5 * n : 0);
  }
  free(x);
  return 0;
}

/* the same on float64 coordinates (finite-difference forces in tests/test_energy_kat.py) */
int oracle_amber_energy_f64(const oracle_ff* ff, const double* coords, double* out, double* terms, int64_t n_rows) {
  const int V = ff->n_atoms;
  for (int64_t n = 0; n < n_rows; ++n) out[n] = energy_one(ff, coords + n * 3 * V, terms ? terms + 5 * n : 0);
  return 0;
}
