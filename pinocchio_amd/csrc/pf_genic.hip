// pf_genic.hip -- the feeder of the path on the device: PINOCCHIO's Gaussian
// initial-condition generator writes delta(k) straight into HBM, in the library's internal layout, so the
// 8.6 GB host->device copy of kdensity at 1024^3 disappears and a run is reproducible from (seed, cosmology).
//
//   GenIC_large                src/GenIC.c:73-460: one ranlxd1 stream per (kx,ky) column seeded from the seed plane;
//                              phase and Rayleigh amplitude per kz; Hermitian partners on the kz = 0 plane take the
//                              mirrored column's seed; Nyquist planes, DC and modes outside the Nyquist sphere stay 0;
//                              final factor N^3.
//   seed plane                 src/GenIC.c:840-990: the get_map(P)-th draw of MT19937(RandomSeed) at the spiral
//                              coordinates P of the column (decomposition independent) -- built on the host.
//   PowerSpec_EH / transf_EH   src/cosmo.c:1443-1497: k-independent constants evaluated once on the host with the
//                              reference's expressions, the k-dependent part per mode on the device.
//   normalize_PowerSpectrum    src/cosmo.c:1058-1075: sigma8^2 / top-hat variance at 8/h Mpc (host, Gauss-Legendre).
//   gsl_rng_ranlxd1            GSL 2.7.1 rng/ranlxd.c (Luescher's RANLUX, 48-bit doubles, luxury 202), including its
//                              seeding of seeds >= 2^31 through a negative int; gsl_rng_mt19937: rng/mt.c.
//
// One thread per column: the ranlxd1 stream of a column is inherently serial (2 draws per kz), the columns are
// independent.  Each rank generates the columns of its own ky slab (KY layout), no exchange.
#include <hip/hip_runtime.h>
#include <math.h>

#include <vector>

#include "../../include/pinfmax.h"
#include "pf_internal.h"
#include "pf_collapse_core.h"  // pf_spline_coeffs: GSL's natural cubic spline (host)
#include <string.h>

#define PFG_PI 3.14159265358979323846

// ------------------------------------------------------------------ host: MT19937 + spiral seeds ----
static void mt19937_draws(unsigned int seed, size_t count, std::vector<unsigned int> &out) {
  const int N = 624, M = 397;
  unsigned int mt[624];
  mt[0] = seed;
  for (int i = 1; i < N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned int)i;
  int mti = N;
  out.resize(count);
  for (size_t c = 0; c < count; c++) {
    if (mti >= N) {
      for (int kk = 0; kk < N; kk++) {
        const unsigned int y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % N] & 0x7fffffffu);
        mt[kk] = mt[(kk + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      mti = 0;
    }
    unsigned int y = mt[mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    out[c] = y;
  }
}

static long long spiral_map(long long x, long long y) {  // get_map, src/GenIC.c:840-855
  const long long mx = x < 0 ? -x : x, my = y < 0 ? -y : y;
  const long long l = 2 * (mx > my ? mx : my);
  const int c = (y > x) + (x > 0) * (x == y);
  const long long d = c ? l * 3 + x + y : l - x - y;
  return (l - 1) * (l - 1) + d;
}

// seed[jj*n + ii]
static void seed_plane(int n, unsigned int random_seed, std::vector<unsigned int> &seed) {
  long long maxm = 0;
  std::vector<long long> m((size_t)n * n);
  for (int jj = 0; jj < n; jj++)
    for (int ii = 0; ii < n; ii++) {
      const long long v = spiral_map(ii >= n / 2 ? ii - n : ii, jj >= n / 2 ? jj - n : jj);
      m[(size_t)jj * n + ii] = v;
      if (v > maxm) maxm = v;
    }
  std::vector<unsigned int> draws;
  mt19937_draws(random_seed, (size_t)maxm, draws);
  seed.resize((size_t)n * n);
  for (size_t i = 0; i < seed.size(); i++) seed[i] = draws[(size_t)m[i] - 1];
}

// ------------------------------------------------------------------ E&H constants (host) ----
struct EHConst {
  double Omega0, OB, Omegac, Oh2, Teta2_over_Oh2, s, alc, bec, beb, bno, ksi, alb, ns;
};

static void eh_constants(const pf_genic_params *p, EHConst *e) {  // the k-independent lines of transf_EH
  const double Teta_27 = 1.0104;
  const double OB = (p->OmegaBaryon > 1.e-6 ? p->OmegaBaryon : 1.e-6);
  const double Omegac = p->Omega0 - OB;
  const double Oh2 = p->Omega0 * p->Hubble100 * p->Hubble100;
  const double Ob2 = OB * p->Hubble100 * p->Hubble100;
  const double b1 = 0.313 * pow(Oh2, -0.419) * (1 + 0.607 * pow(Oh2, 0.674));
  const double b2 = 0.238 * pow(Oh2, 0.223);
  const double zd = 1291. * pow(Oh2, 0.251) * (1. + b1 * pow(Ob2, b2)) / (1. + 0.659 * pow(Oh2, 0.828));
  const double Rd = 31.5 * Ob2 / (pow(Teta_27, 4.0) * 0.001 * zd);
  const double zeq = 2.5e4 * Oh2 / pow(Teta_27, 4.0);
  const double Req = 31.5 * Ob2 / (pow(Teta_27, 4.0) * 0.001 * zeq);
  const double keq = 7.46e-2 * Oh2 / Teta_27 / Teta_27;
  const double s = 1.633 * log((sqrt(1. + Rd) + sqrt(Rd + Req)) / (1 + sqrt(Req))) / (keq * sqrt(Req));
  const double alc = pow(pow(46.9 * Oh2, 0.670) * (1. + pow(32.1 * Oh2, -0.532)), -OB / p->Omega0) *
                     pow(pow(12.0 * Oh2, 0.424) * (1. + pow(45.0 * Oh2, -0.582)), -pow(OB / p->Omega0, 3.0));
  const double bec = 1. / (1. + (0.944 / (1. + pow(458. * Oh2, -0.708))) * (pow(Omegac / p->Omega0, pow(0.395 * Oh2, -0.0266)) - 1.));
  const double beb = 0.5 + OB / p->Omega0 + (3. - 2. * OB / p->Omega0) * sqrt(pow(17.2 * Oh2, 2.0) + 1.);
  const double bno = 8.41 * pow(Oh2, 0.435);
  const double ksi = 1.6 * pow(Ob2, 0.52) * pow(Oh2, 0.73) * (1. + pow(10.4 * Oh2, -0.95));
  const double y = (1. + zeq) / (1 + zd);
  const double alb = 2.07 * keq * s * pow(1.0 + Rd, -0.75) * (y * (-6. * sqrt(1. + y) + (2. + 3. * y) * log((sqrt(1. + y) + 1.) / (sqrt(1. + y) - 1.))));
  e->Omega0 = p->Omega0; e->OB = OB; e->Omegac = Omegac; e->Oh2 = Oh2; e->Teta2_over_Oh2 = Teta_27 * Teta_27 / Oh2;
  e->s = s; e->alc = alc; e->bec = bec; e->beb = beb; e->bno = bno; e->ksi = ksi; e->alb = alb; e->ns = p->PrimordialIndex;
}

__host__ __device__ inline double pfg_T0(double q, double a, double b) {  // src/cosmo.c:1489-1497
  const double ll = log(exp(1.) + 1.8 * b * q);
  const double C = 14.2 / a + 386. / (1. + 69.9 * pow(q, 1.08));
  return ll / (ll + C * q * q);
}
// PowerSpec_EH(k) = k^ns T(k)^2, k-dependent part of transf_EH (src/cosmo.c:1471-1486), un-normalised
__host__ __device__ inline double pfg_powerspec_EH(double fk, const EHConst &e) {
  const double ks = fk * e.s;
  const double q = fk * e.Teta2_over_Oh2;
  const double f = 1. / (1 + pow(ks / 5.4, 4.0));
  const double Tc = f * pfg_T0(q, 1., e.bec) + (1. - f) * pfg_T0(q, e.alc, e.bec);
  const double kst = ks / pow(1. + pow(e.bno / ks, 3.0), 0.3333);
  const double Tb = (pfg_T0(q, 1., 1.) / (1. + pow(ks / 5.2, 2.0)) + e.alb / (1. + pow(e.beb / ks, 3.0)) * exp(-pow(fk / e.ksi, 1.4))) * sin(kst) / kst;
  const double Tr = (e.OB * Tb + e.Omegac * Tc) / e.Omega0;
  return pow(fk, e.ns) * pow(Tr, 2.);
}

// ------------------------------------------------------------------ ranlxd1 (device) ----
struct Ranlxd {
  double xdbl[12];
  double carry;
  unsigned int ir, jr, ir_old;
};
#define PFG_ONE_BIT (1.0 / 281474976710656.0)
#define PFG_LUX 202

__device__ inline void pfg_ranlxd_set(Ranlxd &st, unsigned int s32) {
  int xbit[31];
  int i = (int)s32;  // GSL keeps the seed in an int: seeds >= 2^31 turn negative and so do their remainders below
  if (s32 == 0) i = 1;
  for (int k = 0; k < 31; ++k) { xbit[k] = i % 2; i /= 2; }
  int ibit = 0, jbit = 18;
  for (int k = 0; k < 12; ++k) {
    double x = 0;
    for (int l = 1; l <= 48; ++l) {
      const double y = (double)((xbit[ibit] + 1) % 2);
      x += x + y;
      xbit[ibit] = (xbit[ibit] + xbit[jbit]) % 2;
      ibit = (ibit + 1) % 31;
      jbit = (jbit + 1) % 31;
    }
    st.xdbl[k] = PFG_ONE_BIT * x;
  }
  st.carry = 0;
  st.ir = 11; st.jr = 7; st.ir_old = 0;
}

__device__ inline void pfg_ranlxd_increment(Ranlxd &st) {
  int k;
  double y1, y2;
  double carry = st.carry;
  unsigned int ir = st.ir, jr = st.jr;
  // GSL unrolls the middle part by hand; the recurrence is the same subtract-with-borrow step throughout
  for (k = 0; k < PFG_LUX; ++k) {
    y1 = st.xdbl[jr] - st.xdbl[ir];
    y2 = y1 - carry;
    if (y2 < 0) { carry = PFG_ONE_BIT; y2 += 1; } else carry = 0;
    st.xdbl[ir] = y2;
    ir = (ir + 1) % 12;
    jr = (jr + 1) % 12;
  }
  st.ir = ir; st.ir_old = ir; st.jr = jr; st.carry = carry;
}

__device__ inline double pfg_ranlxd_uniform(Ranlxd &st) {
  const unsigned int ir = st.ir;
  st.ir = (ir + 1) % 12;
  if (st.ir == st.ir_old) pfg_ranlxd_increment(st);
  return st.xdbl[st.ir];
}

// ------------------------------------------------------------------ GenIC kernel ----
struct GenicArgs {
  void *dk;                  // KY layout [n][nyl][nzp] complex F
  const unsigned int *seed;  // [jj*n + ii], whole plane
  int n, nzp, nyl, y0;
  double box, fac, pknorm, n3;
  EHConst eh;
  bool fixed, paired;        // params.FixedIC, params.PairedIC
  int pkn;                   // > 0: tabulated spectrum, knots log10 k -> log10(k^3 P) with their natural-spline c
  const double *pkx, *pky, *pkc;
  int form;                  // 1 Eisenstein & Hu, 2 tabulated, 3 Efstathiou, 4 power law (WhichSpectrum, src/cosmo.c:953-986)
  double wdm_alpha, hubble, unit_ratio;  // warm-dark-matter cut-off (src/cosmo.c:987-1005): alpha [Mpc/h], Hubble100, 3.085678e24 / UnitLength_in_cm; alpha 0: none
};
#define PFG_SHAPE_EFST 0.21  // src/cosmo.c:44
// PowerSpec_Efstathiou, PowerSpec_PowerLaw (src/cosmo.c:1437-1445) and the cut-off Tf^2 (:998-1004), as the reference writes them
__host__ __device__ inline double pfg_powerspec_efstathiou(double k, double ns) {
  return pow(k, ns) / pow(1 + pow(6.4 / PFG_SHAPE_EFST * k + pow(3.0 / PFG_SHAPE_EFST * k, 1.5) + pow(1.7 / PFG_SHAPE_EFST, 2.0) * k * k, 1.13), 2 / 1.13);
}
__host__ __device__ inline double pfg_wdm_cutoff(double k, double alpha, double hubble, double unit_ratio) {
  const double Tf = pow(1 + pow(alpha * k / hubble * unit_ratio, 2), -5.0);
  return Tf * Tf;
}
static double pfg_wdm_alpha(const pf_genic_params *p) {
  return p->WDM_PartMass_in_kev > 0.
             ? 0.05 * pow((p->Omega0 - p->OmegaBaryon) / 0.4, 0.15) * pow(p->Hubble100 / 0.65, 1.3) * pow(1.0 / p->WDM_PartMass_in_kev, 1.15)
             : 0.0;
}
// PowerSpec_Tabulated (src/cosmo.c:1432-1435): my_spline_eval of SPLINE[SP_PK] (linear beyond the knots), then 10^. / k^3
__device__ __forceinline__ double pfg_powerspec_tab(double k, const GenicArgs &a) {
  const double x = log10(k);
  const double *xa = a.pkx, *ya = a.pky, *ca = a.pkc;
  const int last = a.pkn - 1;
  double s;
  if (x < xa[0]) s = ya[0] + (x - xa[0]) * (ya[1] - ya[0]) / (xa[1] - xa[0]);
  else if (x > xa[last]) s = ya[last] + (x - xa[last]) * (ya[last] - ya[last - 1]) / (xa[last] - xa[last - 1]);
  else {
    int lo = 0, hi = last;
    while (hi > lo + 1) { const int m = (hi + lo) >> 1; if (xa[m] > x) hi = m; else lo = m; }
    const double dx = xa[lo + 1] - xa[lo], dy = ya[lo + 1] - ya[lo], delx = x - xa[lo];
    const double b = (dy / dx) - dx * (ca[lo + 1] + 2.0 * ca[lo]) / 3.0, d = (ca[lo + 1] - ca[lo]) / (3.0 * dx);
    s = ya[lo] + delx * (b + delx * (ca[lo] + delx * d));
  }
  return pow(10., s) / k / k / k;
}

template <typename F>
__global__ void __launch_bounds__(64) k_genic(const GenicArgs a) {
  const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // col = ii*nyl + jl
  if (col >= (long long)a.n * a.nyl) return;
  const int n = a.n, Nmesh_2 = n / 2;
  const int ii = (int)(col / a.nyl), jl = (int)(col % a.nyl), jj = jl + a.y0;
  F *row = reinterpret_cast<F *>(a.dk) + 2 * col * a.nzp;
  for (int k = 0; k <= Nmesh_2; k++) { row[2 * k] = 0; row[2 * k + 1] = 0; }
  if (ii == Nmesh_2 || jj == Nmesh_2) return;
  const double Box = a.box;
  const double kx = (ii < Nmesh_2) ? ii * 2 * PFG_PI / Box : -(n - ii) * 2 * PFG_PI / Box;
  const double ky = (jj < Nmesh_2) ? jj * 2 * PFG_PI / Box : -(n - jj) * 2 * PFG_PI / Box;
  const double kmag2_ij = kx * kx + ky * ky;
  Ranlxd gen;
  pfg_ranlxd_set(gen, a.seed[(size_t)jj * n + ii]);
  for (int kk = 0; kk < Nmesh_2; kk++) {
    double phase = pfg_ranlxd_uniform(gen) * 2 * PFG_PI;
    double ampl;
    do ampl = pfg_ranlxd_uniform(gen); while (ampl == 0);
    if (ii == 0 && jj == 0 && kk == 0) continue;
    const double kz = kk * 2 * PFG_PI / Box;
    const double kmag = sqrt(kmag2_ij + kz * kz);
    if (kmag * Box / (2 * PFG_PI) > 1. * n / 2) continue;  // NYQUIST = 1.
    double power = a.form == 2 ? pfg_powerspec_tab(kmag, a) : a.form == 3 ? pfg_powerspec_efstathiou(kmag, a.eh.ns)
                   : a.form == 4 ? pow(kmag, a.eh.ns) : pfg_powerspec_EH(kmag, a.eh);
    if (a.wdm_alpha > 0.) power *= pfg_wdm_cutoff(kmag, a.wdm_alpha, a.hubble, a.unit_ratio);
    double p_of_k = a.pknorm * power;
    double sign = 1.0;
    if (kk == 0) {  // Hermitian partners on the kz = 0 plane (src/GenIC.c:289-368)
      if (ii == 0 && jj == Nmesh_2) continue;
      if (ii > Nmesh_2 || (ii == 0 && jj > Nmesh_2)) {
        int jjj = n - jj;
        if (jjj == n) jjj = 0;
        const int iii = (ii > Nmesh_2) ? n - ii : ii;
        sign = -1.0;
        Ranlxd k0;
        pfg_ranlxd_set(k0, a.seed[(size_t)jjj * n + iii]);
        phase = pfg_ranlxd_uniform(k0) * 2 * PFG_PI;
        do ampl = pfg_ranlxd_uniform(k0); while (ampl == 0);
      }
    }
    if (a.paired) phase += PFG_PI;      // src/GenIC.c:371
    if (!a.fixed) p_of_k *= -log(ampl);  // src/GenIC.c:375
    const double delta = a.fac * sqrt(p_of_k);
    row[2 * kk] = (F)(delta * cos(phase) * a.n3);
    row[2 * kk + 1] = (F)(sign * delta * sin(phase) * a.n3);
  }
}

// ------------------------------------------------------------------ host API ----
extern "C" int pf_pk_norm(const pf_genic_params *p, double sigma8, double *pknorm) {
  if (!p || !pknorm) return 1;
  EHConst e;
  eh_constants(p, &e);
  // sigma^2(R) = int dlnk P(k) W^2(kR) k^3 / (2 pi^2), top-hat W, R = 8/h Mpc, from ln k = -10 to ln(500/R)
  const double R = 8.0 / p->Hubble100, lo = -10.0, hi = log(500.0 / R);
  const int panels = 4096;  // 4-point Gauss-Legendre per panel
  const double gx[4] = {-0.8611363115940526, -0.3399810435848563, 0.3399810435848563, 0.8611363115940526};
  const double gw[4] = {0.3478548451374538, 0.6521451548625461, 0.6521451548625461, 0.3478548451374538};
  double sum = 0.0;
  const double h = (hi - lo) / panels;
  if (p->pk_n > 0) return 1;  // (a tabulated spectrum comes with its own normalisation: the caller passes PkNorm)
  const double wdm_alpha = pfg_wdm_alpha(p), unit_ratio = 3.085678e24 / (p->UnitLength_in_cm > 0. ? p->UnitLength_in_cm : 3.085678e24);
  for (int i = 0; i < panels; i++) {
    const double c = lo + (i + 0.5) * h;
    for (int g = 0; g < 4; g++) {
      const double k = exp(c + 0.5 * h * gx[g]);
      const double kr = k * R, kr2 = kr * kr;
      const double w = (kr < 1.e-5) ? 1.0 : 3. * (sin(kr) / kr2 / kr - cos(kr) / kr2);
      double power = p->spectrum == 3 ? pfg_powerspec_efstathiou(k, p->PrimordialIndex) : p->spectrum == 4 ? pow(k, p->PrimordialIndex) : pfg_powerspec_EH(k, e);
      if (wdm_alpha > 0.) power *= pfg_wdm_cutoff(k, wdm_alpha, p->Hubble100, unit_ratio);
      sum += gw[g] * 0.5 * h * power * w * w * k * k * k / (2. * PFG_PI * PFG_PI);
    }
  }
  *pknorm = sigma8 * sigma8 / sum;
  return 0;
}

int pf_genic_launch(int fb, void *dk, int n, int nzp, int nyl, int y0, const pf_genic_params *p, hipStream_t st,
                    unsigned int **seed_dev_out) {
  std::vector<unsigned int> seed;
  seed_plane(n, p->RandomSeed, seed);
  unsigned int *dseed = nullptr;
  double *dpk = nullptr;
  auto fail = [&]() { if (dseed) hipFree(dseed); if (dpk) hipFree(dpk); return 1; };  // nothing allocated here outlives an error
  if (hipMalloc(&dseed, seed.size() * sizeof(unsigned int)) != hipSuccess) return 1;
  if (hipMemcpyAsync(dseed, seed.data(), seed.size() * sizeof(unsigned int), hipMemcpyHostToDevice, st) != hipSuccess) return fail();
  if (hipStreamSynchronize(st) != hipSuccess) return fail();  // `seed` is a local vector
  GenicArgs a;
  a.dk = dk; a.seed = dseed; a.n = n; a.nzp = nzp; a.nyl = nyl; a.y0 = y0;
  a.box = p->BoxSize_true_Mpc; a.fac = pow(1. / a.box, 1.5); a.pknorm = p->PkNorm; a.n3 = pow((double)n, 3.0);
  a.fixed = p->FixedIC != 0; a.paired = p->PairedIC != 0;
  a.pkn = 0; a.pkx = a.pky = a.pkc = nullptr;
  if (p->pk_n > 0) {  // SPLINE[SP_PK]: GSL's natural cubic spline through the table, coefficients on the host
    if (p->pk_n < 3 || !p->pk_logk || !p->pk_logk3p) return fail();
    std::vector<double> tab(3 * (size_t)p->pk_n);
    memcpy(tab.data(), p->pk_logk, sizeof(double) * p->pk_n);
    memcpy(tab.data() + p->pk_n, p->pk_logk3p, sizeof(double) * p->pk_n);
    if (pf_spline_coeffs(p->pk_logk, p->pk_logk3p, p->pk_n, tab.data() + 2 * (size_t)p->pk_n)) return fail();
    if (hipMalloc(&dpk, tab.size() * sizeof(double)) != hipSuccess) return fail();
    if (hipMemcpyAsync(dpk, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return fail();
    if (hipStreamSynchronize(st) != hipSuccess) return fail();  // `tab` is a local vector
    a.pkn = p->pk_n; a.pkx = dpk; a.pky = dpk + p->pk_n; a.pkc = dpk + 2 * (size_t)p->pk_n;
  }
  eh_constants(p, &a.eh);
  a.form = p->pk_n > 0 ? 2 : (p->spectrum == 3 || p->spectrum == 4) ? p->spectrum : 1;
  a.wdm_alpha = pfg_wdm_alpha(p); a.hubble = p->Hubble100;
  a.unit_ratio = 3.085678e24 / (p->UnitLength_in_cm > 0. ? p->UnitLength_in_cm : 3.085678e24);
  const long long ncol = (long long)n * nyl;
  const unsigned blocks = (unsigned)((ncol + 63) / 64);
  if (fb == 8) hipLaunchKernelGGL(k_genic<double>, dim3(blocks), dim3(64), 0, st, a);
  else hipLaunchKernelGGL(k_genic<float>, dim3(blocks), dim3(64), 0, st, a);
  const int rc = hipGetLastError() == hipSuccess ? 0 : 1;
  if (dpk) {  // the kernel has the table for its lifetime only
    const bool done = hipStreamSynchronize(st) == hipSuccess;
    hipFree(dpk); dpk = nullptr;
    if (!done) return fail();
  }
  if (rc) return fail();
  *seed_dev_out = dseed;  // the caller's from here on
  return 0;
}
