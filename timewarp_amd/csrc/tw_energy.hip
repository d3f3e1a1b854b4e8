// AMBER-style potential energy on the GPU: one workgroup per conformation - four waves up to 64 atoms (r05: at 22 atoms a single
// wave spent most of its 27 us in the 21 serial iterations of the Born-radius loop with 22 of 64 lanes active; the kernel sits on
// the MH iteration's critical path since the flow's launches fill the chip), sixteen above (r06: the reference's 691-atom test
// protein took 6.2 ms on one wave - a quarter of an MH iteration at 16 proposals) - fp64 arithmetic.
//
// Replaces the OpenMM call chain behind OpenmmPotentialEnergyTorch.forward
// (utils/openmm/openmm_bridge.py:281-294 -> bgflow -> Context.getState(getEnergy=True)) for Systems
// built by simulation/md.py:128-187.  OpenMM 7.7 itself is a third-party dependency that is not
// vendored in the reference; the functional forms below restate its published Reference-platform
// algorithms (HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce, NonbondedForce with
// CutoffNonPeriodic + reaction field, GBSAOBCForce = OBC-II + ACE surface term).  The same maths is
// restated in plain C in oracle/energy_oracle.c, which is what the parity tests compare against.
#include "tw_common.h"

namespace tw {

#define TW_ONE_4PI_EPS0 138.935456  // OpenMM SimTKOpenMMRealType.h (kJ nm / (mol e^2))

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// W waves per conformation.  W > 1: every loop strides over the whole workgroup.  The Born-radius sums: W = 4 (small molecules)
// computes them pair-parallel into a V x V table that one lane per atom then adds up in the single-wave kernel's order (same radii
// bit for bit); W = 16 (no room for the table) gives every atom a wave - lanes over the partners, partial sums in partner order per
// lane, a butterfly over the wave: the same sums up to the order of the fp64 additions.
template <int W>
__global__ void __launch_bounds__(64 * W) amber_energy_kernel(const tw_forcefield ff, const float* __restrict__ coords,
                                                               double* __restrict__ out, double* __restrict__ terms) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  constexpr int NTH = 64 * W;
  const int V = ff.n_atoms;
  double* x = smd;             // [V*3]
  double* born = x + 3 * V;    // [V]
  double* part = born + V;     // [W * 5] partial sums of the waves
  double* tmat = part + 5 * W; // [V*V] (W > 1 only)
  unsigned* excl = (unsigned*)(tmat + (W == 4 ? V * V : 0));  // [V*V] bits
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x;   // (thread of the workgroup)
  for (int i = lane; i < 3 * V; i += NTH) x[i] = (double)coords[n * 3 * V + i];
  excl_fill(ff.exc_idx, ff.n_exceptions, V, excl, lane, NTH);

  double e_bond = 0, e_angle = 0, e_tors = 0, e_nb = 0, e_gb = 0;

  // HarmonicBondForce: 1/2 k (r - r0)^2
  for (int b = lane; b < ff.n_bonds; b += NTH) {
    const int i = ff.bond_idx[2 * b], j = ff.bond_idx[2 * b + 1];
    const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    const double d = r - ff.bond_par[2 * b];
    e_bond += 0.5 * ff.bond_par[2 * b + 1] * d * d;
  }
  // HarmonicAngleForce: 1/2 k (theta - theta0)^2
  for (int a = lane; a < ff.n_angles; a += NTH) {
    const int i = ff.angle_idx[3 * a], j = ff.angle_idx[3 * a + 1], k = ff.angle_idx[3 * a + 2];
    double v0[3], v1[3];
    for (int c = 0; c < 3; ++c) { v0[c] = x[3 * i + c] - x[3 * j + c]; v1[c] = x[3 * k + c] - x[3 * j + c]; }
    const double d00 = v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2];
    const double d11 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    const double d01 = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
    double cs = d01 / sqrt(d00 * d11);
    cs = fmin(1.0, fmax(-1.0, cs));
    const double d = acos(cs) - ff.angle_par[2 * a];
    e_angle += 0.5 * ff.angle_par[2 * a + 1] * d * d;
  }
  // PeriodicTorsionForce: k (1 + cos(n phi - phase))
  for (int t = lane; t < ff.n_torsions; t += NTH) {
    const int a = ff.torsion_idx[4 * t], b = ff.torsion_idx[4 * t + 1], c = ff.torsion_idx[4 * t + 2],
              d = ff.torsion_idx[4 * t + 3];
    double r0[3], r1[3], r2[3];
    for (int q = 0; q < 3; ++q) {
      r0[q] = x[3 * a + q] - x[3 * b + q];
      r1[q] = x[3 * c + q] - x[3 * b + q];
      r2[q] = x[3 * c + q] - x[3 * d + q];
    }
    double c0[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    double c1[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
    const double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
    const double dt = c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2];
    double cs = dt / sqrt(n0 * n1);
    cs = fmin(1.0, fmax(-1.0, cs));
    double phi = acos(cs);
    // sign convention of OpenMM's reference kernel: sign of r0 . (r1 x r2)
    const double sgn = r0[0] * c1[0] + r0[1] * c1[1] + r0[2] * c1[2];
    if (sgn < 0) phi = -phi;
    e_tors += ff.torsion_par[3 * t + 2] * (1.0 + cos(ff.torsion_par[3 * t] * phi - ff.torsion_par[3 * t + 1]));
  }
  // NonbondedForce exceptions (1-4 pairs; 1-2/1-3 carry zeros): no cutoff, no reaction field
  for (int e = lane; e < ff.n_exceptions; e += NTH) {
    const double qq = ff.exc_par[3 * e], sig = ff.exc_par[3 * e + 1], eps = ff.exc_par[3 * e + 2];
    if (qq == 0.0 && eps == 0.0) continue;
    const int i = ff.exc_idx[2 * e], j = ff.exc_idx[2 * e + 1];
    const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    const double sr2 = (sig / r) * (sig / r), sr6 = sr2 * sr2 * sr2;
    e_nb += TW_ONE_4PI_EPS0 * qq / r + 4.0 * eps * (sr6 * sr6 - sr6);
  }
  // NonbondedForce, all non-excluded pairs inside the cutoff
  const bool use_cut = ff.cutoff > 0.0;
  const double rc = ff.cutoff;
  const double krf = use_cut ? (1.0 / (rc * rc * rc)) * (ff.rf_dielectric - 1.0) / (2.0 * ff.rf_dielectric + 1.0) : 0.0;
  const double crf = use_cut ? (1.0 / rc) * (3.0 * ff.rf_dielectric) / (2.0 * ff.rf_dielectric + 1.0) : 0.0;
  const int npairs = V * (V - 1) / 2;
  // every pair j < i once: up to four waves a linear pair index strided over the workgroup and decoded; sixteen waves: a wave per
  // row i, lanes over j (no decode - an fp64 square root and two loops per pair - and 238 k pairs at 691 atoms)
  auto for_pairs = [&](auto&& body) {
    if constexpr (W > 4) {
      for (int i = lane >> 6; i < V; i += W)
        for (int j = lane & 63; j < i; j += 64) body(i, j);
    } else {
      for (int p = lane; p < npairs; p += NTH) {
        int i = (int)((sqrt(8.0 * p + 1.0) + 1.0) * 0.5);
        while (i * (i - 1) / 2 > p) --i;
        while ((i + 1) * i / 2 <= p) ++i;
        body(i, p - i * (i - 1) / 2);  // j < i
      }
    }
  };
  for_pairs([&](int i, int j) {
    if (excl_test(excl, i * V + j)) return;
    const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double r = sqrt(r2);
    if (use_cut && r >= rc) return;
    const double* pi = ff.atom_par + 5 * i;
    const double* pj = ff.atom_par + 5 * j;
    const double sig = 0.5 * (pi[1] + pj[1]);
    const double eps = sqrt(pi[2] * pj[2]);
    const double sr2 = (sig * sig) / r2, sr6 = sr2 * sr2 * sr2;
    e_nb += 4.0 * eps * (sr6 * sr6 - sr6);
    e_nb += TW_ONE_4PI_EPS0 * pi[0] * pj[0] * (use_cut ? (1.0 / r + krf * r2 - crf) : 1.0 / r);
  });
  // GBSA-OBC (has_gbsa 1: OBC-II alpha=1 beta=0.8 gamma=4.85 = GBSAOBCForce / amber99_obc.xml; 2: OBC-I alpha=0.8
  // beta=0 gamma=2.909125 = implicit/obc1.xml; dielectric offset 0.009 nm, probe 0.14 nm)
  if (ff.has_gbsa) {
    const double offset = 0.009, probe = 0.14;
    const double alpha = ff.has_gbsa == 2 ? 0.8 : 1.0, beta = ff.has_gbsa == 2 ? 0.0 : 0.8,
                 gamma = ff.has_gbsa == 2 ? 2.909125 : 4.85;
    // one (i, j) term of atom i's pair integral (OpenMM's ReferenceObc::computeBornRadii)
    auto born_term = [&](int i, int j, double off_i) -> double {
      const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      if (use_cut && r > rc) return 0.0;
      const double off_j = ff.atom_par[5 * j + 3] - offset;
      const double sr_j = off_j * ff.atom_par[5 * j + 4];
      const double r_sr = r + sr_j;
      if (!(off_i < r_sr)) return 0.0;
      const double rinv = 1.0 / r;
      const double ad = fabs(r - sr_j);
      const double l = 1.0 / (off_i > ad ? off_i : ad);
      const double u = 1.0 / r_sr;
      const double l2 = l * l, u2 = u * u;
      const double ratio = log(u / l);
      double term = l - u + 0.25 * r * (u2 - l2) + 0.5 * rinv * ratio + 0.25 * sr_j * sr_j * rinv * (l2 - u2);
      if (off_i < (sr_j - r)) term += 2.0 * (1.0 / off_i - l);
      return term;
    };
    if constexpr (W == 4) {
      for (int p = lane; p < V * V; p += NTH) {
        const int i = p / V, j = p - i * V;
        tmat[p] = i == j ? 0.0 : born_term(i, j, ff.atom_par[5 * i + 3] - offset);
      }
      __syncthreads();
    }
    if constexpr (W > 4) {
      __syncthreads();   // (x is complete)
      for (int i = lane >> 6; i < V; i += W) {
        const double rad_i = ff.atom_par[5 * i + 3];
        const double off_i = rad_i - offset;
        double sum = 0.0;
        for (int j = lane & 63; j < V; j += 64)
          if (j != i) sum += born_term(i, j, off_i);
        sum = wsum(sum);
        sum *= 0.5 * off_i;
        const double s2 = sum * sum, s3 = sum * s2;
        const double th = tanh(alpha * sum - beta * s2 + gamma * s3);
        if ((lane & 63) == 0) born[i] = 1.0 / (1.0 / off_i - th / rad_i);
      }
    } else
    for (int i = lane; i < V; i += NTH) {
      const double rad_i = ff.atom_par[5 * i + 3];
      const double off_i = rad_i - offset;
      double sum = 0.0;
      for (int j = 0; j < V; ++j) {
        if (j == i) continue;
        if constexpr (W == 4) sum += tmat[i * V + j];
        else sum += born_term(i, j, off_i);
      }
      sum *= 0.5 * off_i;
      const double s2 = sum * sum, s3 = sum * s2;
      const double th = tanh(alpha * sum - beta * s2 + gamma * s3);
      born[i] = 1.0 / (1.0 / off_i - th / rad_i);
    }
    __syncthreads();
    const double pre = -TW_ONE_4PI_EPS0 * (1.0 / ff.solute_dielectric - 1.0 / ff.solvent_dielectric);
    // ACE non-polar term: 4 pi * surface_area_energy * (r+probe)^2 (r/B)^6
    const double pi4a = 4.0 * 3.14159265358979323846 * ff.surface_area_energy;
    for (int i = lane; i < V; i += NTH) {
      const double rad = ff.atom_par[5 * i + 3];
      if (born[i] > 0.0) {
        const double rr = rad + probe;
        const double ratio = rad / born[i];
        const double r3 = ratio * ratio * ratio;
        e_gb += pi4a * rr * rr * r3 * r3;
      }
      // self term
      const double q = ff.atom_par[5 * i];
      e_gb += 0.5 * pre * q * q / born[i];
    }
    for_pairs([&](int i, int j) {
      const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      const double r2 = dx * dx + dy * dy + dz * dz;
      if (use_cut && sqrt(r2) > rc) return;
      const double a2 = born[i] * born[j];
      const double dij = r2 / (4.0 * a2);
      const double den = sqrt(r2 + a2 * exp(-dij));
      const double qq = pre * ff.atom_par[5 * i] * ff.atom_par[5 * j];
      double e = qq / den;
      if (use_cut) e -= qq / rc;
      e_gb += e;
    });
  }
  e_bond = wsum(e_bond); e_angle = wsum(e_angle); e_tors = wsum(e_tors); e_nb = wsum(e_nb); e_gb = wsum(e_gb);
  if constexpr (W > 1) {   // the waves' sums, added in wave order
    if ((lane & 63) == 0) {
      double* q = part + 5 * (lane >> 6);
      q[0] = e_bond; q[1] = e_angle; q[2] = e_tors; q[3] = e_nb; q[4] = e_gb;
    }
    __syncthreads();
    if (lane == 0) {
      e_bond = e_angle = e_tors = e_nb = e_gb = 0.0;
      for (int w = 0; w < W; ++w) {
        e_bond += part[5 * w]; e_angle += part[5 * w + 1]; e_tors += part[5 * w + 2]; e_nb += part[5 * w + 3]; e_gb += part[5 * w + 4];
      }
    }
  }
  if (lane == 0) {
    out[n] = e_bond + e_angle + e_tors + e_nb + e_gb;
    if (terms) {
      terms[5 * n] = e_bond; terms[5 * n + 1] = e_angle; terms[5 * n + 2] = e_tors; terms[5 * n + 3] = e_nb;
      terms[5 * n + 4] = e_gb;
    }
  }
}

int amber_energy(const tw_forcefield* ff, const float* coords, double* out, double* terms, int64_t n, hipStream_t s) {
  if (n == 0) return TW_OK;
  const int V = ff->n_atoms;
  const bool four = V <= 64;   // small molecules: latency-bound, four waves per conformation (+ the V x V table: <= 32 KiB)
  const int W = four ? 4 : 16;
  size_t shm = (size_t)(4 * V + 5 * W + (four ? V * V : 0)) * sizeof(double) + excl_bytes(V);
  shm = (shm + 15) / 16 * 16;
  TW_REQUIRE(shm <= (size_t)160 * 1024, "energy kernel: %d atoms need %zu bytes of LDS (one conformation per workgroup; limit 160 KiB)", V, shm);
  int rc;
  if (four) {
    hipLaunchKernelGGL(amber_energy_kernel<4>, dim3((unsigned)n), dim3(256), shm, s, *ff, coords, out, terms);
  } else {
    static LdsLimit lim;
    if (shm > (size_t)64 * 1024 && (rc = lim.ensure((const void*)amber_energy_kernel<16>, 160 * 1024))) return rc;
    hipLaunchKernelGGL(amber_energy_kernel<16>, dim3((unsigned)n), dim3(1024), shm, s, *ff, coords, out, terms);
  }
  TW_LAUNCH_CHECK();
  return TW_OK;
}

}  // namespace tw
