/*
 * pf_genic.c -- CPU ORACLE, part 2 (test infrastructure, see pf_oracle.h): the
 * feeder of the hot path restated, so that the oracle can be pinned end to end
 * against the reference's committed HMF_Validation run (seed 486604, 128^3).
 *
 *   GenIC_large              src/GenIC.c:73-460   (single task, non-transposed layout, large_plane seeds)
 *   PowerSpec_EH, transf_EH  src/cosmo.c:1443-1497
 *   gsl_rng_ranlxd1          GSL 2.7.1 rng/ranlxd.c (Luescher's RANLUX, double precision, luxury 202);
 *                            not vendored in the reference; restated here and pinned by GSL's own
 *                            published test value (rng/test.c: seed 1, 10000th gsl_rng_get = 1998227290 = 0.465248546261094020 * 2^32)
 *   seeds                    the caller passes the per-(kx,ky) seed table; it is built in
 *                            tests/ic_oracle.py from MT19937 along the spiral (src/GenIC.c:840-990)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.14159265358979323846

/* pf_oracle.c: GSL's natural cubic spline and my_spline_eval on explicit arrays (for the tabulated spectrum) */
int orc_cspline_coeffs(const double *xa, const double *ya, int size, double *sc);
double orc_my_spline_eval(const double *sx, const double *sy, const double *sc, int size, double x);

/* ------------------------------------------------------------ ranlxd1 ---- */
typedef struct {
  double xdbl[12];
  double carry;
  unsigned int ir, jr, ir_old, pr;
} ranlxd_state;

static const int nxt[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 0};
static const double one_bit = 1.0 / 281474976710656.0; /* 2^-48 */

#define RANLUX_STEP(x1, x2, i1, i2, i3) \
  x1 = xdbl[i1] - xdbl[i2];             \
  if (x2 < 0) {                         \
    x1 -= one_bit;                      \
    x2 += 1;                            \
  }                                     \
  xdbl[i3] = x2

static void ranlxd_increment(ranlxd_state *state) {
  int k, kmax;
  double y1, y2, y3;
  double *xdbl = state->xdbl;
  double carry = state->carry;
  unsigned int ir = state->ir, jr = state->jr;
  for (k = 0; ir > 0; ++k) {
    y1 = xdbl[jr] - xdbl[ir];
    y2 = y1 - carry;
    if (y2 < 0) { carry = one_bit; y2 += 1; } else carry = 0;
    xdbl[ir] = y2;
    ir = nxt[ir];
    jr = nxt[jr];
  }
  kmax = (int)state->pr - 12;
  for (; k <= kmax; k += 12) {
    y1 = xdbl[7] - xdbl[0];
    y1 -= carry;
    RANLUX_STEP(y2, y1, 8, 1, 0);
    RANLUX_STEP(y3, y2, 9, 2, 1);
    RANLUX_STEP(y1, y3, 10, 3, 2);
    RANLUX_STEP(y2, y1, 11, 4, 3);
    RANLUX_STEP(y3, y2, 0, 5, 4);
    RANLUX_STEP(y1, y3, 1, 6, 5);
    RANLUX_STEP(y2, y1, 2, 7, 6);
    RANLUX_STEP(y3, y2, 3, 8, 7);
    RANLUX_STEP(y1, y3, 4, 9, 8);
    RANLUX_STEP(y2, y1, 5, 10, 9);
    RANLUX_STEP(y3, y2, 6, 11, 10);
    if (y3 < 0) { carry = one_bit; y3 += 1; } else carry = 0;
    xdbl[11] = y3;
  }
  kmax = (int)state->pr;
  for (; k < kmax; ++k) {
    y1 = xdbl[jr] - xdbl[ir];
    y2 = y1 - carry;
    if (y2 < 0) { carry = one_bit; y2 += 1; } else carry = 0;
    xdbl[ir] = y2;
    ir = nxt[ir];
    jr = nxt[jr];
  }
  state->ir = ir;
  state->ir_old = ir;
  state->jr = jr;
  state->carry = carry;
}

static double ranlxd_uniform(ranlxd_state *state) {
  int ir = state->ir;
  state->ir = nxt[ir];
  if (state->ir == state->ir_old) ranlxd_increment(state);
  return state->xdbl[state->ir];
}

static void ranlxd_set(ranlxd_state *state, unsigned long s, unsigned int luxury) {
  int ibit, jbit, i, k, l, xbit[31];
  double x, y;
  long seed;
  if (s == 0) s = 1;
  seed = (long)s;
  /* GSL: i = seed & 0xFFFFFFFFUL with int i -- seeds >= 2^31 wrap to negative ints and the 31-bit expansion below
     sees negative remainders.  Kept as is: the reference's logged sigmas are reproduced only this way (pinned by
     tests/test_hmf_validation_kat.py). */
  i = (int)(unsigned int)(seed & 0xFFFFFFFFUL);
  for (k = 0; k < 31; ++k) { xbit[k] = i % 2; i /= 2; }
  ibit = 0;
  jbit = 18;
  for (k = 0; k < 12; ++k) {
    x = 0;
    for (l = 1; l <= 48; ++l) {
      y = (double)((xbit[ibit] + 1) % 2);
      x += x + y;
      xbit[ibit] = (xbit[ibit] + xbit[jbit]) % 2;
      ibit = (ibit + 1) % 31;
      jbit = (jbit + 1) % 31;
    }
    state->xdbl[k] = one_bit * x;
  }
  state->carry = 0;
  state->ir = 11;
  state->jr = 7;
  state->ir_old = 0;
  state->pr = luxury;
}

/* n-th draw of gsl_rng_get(ranlxd1) after gsl_rng_set(seed): the form GSL's rng/test.c pins */
unsigned long orc_ranlxd1_nth(unsigned long seed, int n) {
  ranlxd_state st;
  unsigned long v = 0;
  ranlxd_set(&st, seed, 202);
  for (int i = 0; i < n; i++) v = (unsigned long)(ranlxd_uniform(&st) * 4294967296.0);
  return v;
}
void orc_ranlxd1_uniforms(unsigned long seed, int n, double *out) {
  ranlxd_state st;
  ranlxd_set(&st, seed, 202);
  for (int i = 0; i < n; i++) out[i] = ranlxd_uniform(&st);
}

/* ------------------------------------------------- Eisenstein & Hu P(k) -- */
typedef struct { double Omega0, OmegaBaryon, Hubble100, PrimordialIndex; } orc_cosmo;

static double T0(double q, double a, double b) { /* cosmo.c:1489-1497 */
  double ll = log(exp(1.) + 1.8 * b * q);
  double C = 14.2 / a + 386. / (1. + 69.9 * pow(q, 1.08));
  return ll / (ll + C * q * q);
}

double orc_transf_EH(double fk, const orc_cosmo *p) { /* cosmo.c:1452-1487 */
  static double Teta_27 = 1.0104;
  double q, Omegac, Oh2, b1, b2, zd, Rd, zeq, Req, keq, s, ks, alc, bec, f, Tc, beb, bno, kst, ksi, Tb, Tr, y, alb, Ob2, OB;
  OB = (p->OmegaBaryon > 1.e-6 ? p->OmegaBaryon : 1.e-6);
  Omegac = p->Omega0 - OB;
  Oh2 = p->Omega0 * p->Hubble100 * p->Hubble100;
  Ob2 = OB * p->Hubble100 * p->Hubble100;
  b1 = 0.313 * pow(Oh2, -0.419) * (1 + 0.607 * pow(Oh2, 0.674));
  b2 = 0.238 * pow(Oh2, 0.223);
  zd = 1291. * pow(Oh2, 0.251) * (1. + b1 * pow(Ob2, b2)) / (1. + 0.659 * pow(Oh2, 0.828));
  Rd = 31.5 * Ob2 / (pow(Teta_27, 4.0) * 0.001 * zd);
  zeq = 2.5e4 * Oh2 / pow(Teta_27, 4.0);
  Req = 31.5 * Ob2 / (pow(Teta_27, 4.0) * 0.001 * zeq);
  keq = 7.46e-2 * Oh2 / Teta_27 / Teta_27;
  s = 1.633 * log((sqrt(1. + Rd) + sqrt(Rd + Req)) / (1 + sqrt(Req))) / (keq * sqrt(Req));
  ks = fk * s;
  q = fk * Teta_27 * Teta_27 / Oh2;
  alc = pow(pow(46.9 * Oh2, 0.670) * (1. + pow(32.1 * Oh2, -0.532)), -OB / p->Omega0) *
        pow(pow(12.0 * Oh2, 0.424) * (1. + pow(45.0 * Oh2, -0.582)), -pow(OB / p->Omega0, 3.0));
  bec = 1. / (1. + (0.944 / (1. + pow(458. * Oh2, -0.708))) * (pow(Omegac / p->Omega0, pow(0.395 * Oh2, -0.0266)) - 1.));
  f = 1. / (1 + pow(ks / 5.4, 4.0));
  Tc = f * T0(q, 1., bec) + (1. - f) * T0(q, alc, bec);
  beb = 0.5 + OB / p->Omega0 + (3. - 2. * OB / p->Omega0) * sqrt(pow(17.2 * Oh2, 2.0) + 1.);
  bno = 8.41 * pow(Oh2, 0.435);
  kst = ks / pow(1. + pow(bno / ks, 3.0), 0.3333);
  ksi = 1.6 * pow(Ob2, 0.52) * pow(Oh2, 0.73) * (1. + pow(10.4 * Oh2, -0.95));
  y = (1. + zeq) / (1 + zd);
  alb = 2.07 * keq * s * pow(1.0 + Rd, -0.75) * (y * (-6. * sqrt(1. + y) + (2. + 3. * y) * log((sqrt(1. + y) + 1.) / (sqrt(1. + y) - 1.))));
  Tb = (T0(q, 1., 1.) / (1. + pow(ks / 5.2, 2.0)) + alb / (1. + pow(beb / ks, 3.0)) * exp(-pow(fk / ksi, 1.4))) * sin(kst) / kst;
  Tr = (OB * Tb + Omegac * Tc) / p->Omega0;
  return Tr;
}

/* PowerSpec_EH (cosmo.c:1447-1450), un-normalised; k in 1/Mpc */
double orc_powerspec_EH(double k, const orc_cosmo *p) { return pow(k, p->PrimordialIndex) * pow(orc_transf_EH(k, p), 2.); }

/* ------------------------------------------------------------- GenIC ---- */
/* GenIC_large for ONE task owning the whole grid, non-transposed layout.
   seed[jj*n + ii] = seed of the (ii, jj) column (spiral order, built by the caller).
   kdensity out: [n][n][n/2+1][2], already multiplied by n^3 (src/GenIC.c:430-445). */
int orc_genic_ic(int n, double box, const unsigned int *seed, double pknorm, const orc_cosmo *cosmo, int FixedIC, int PairedIC, double *kdensity);
int orc_genic_pk(int n, double box, const unsigned int *seed, double pknorm, const orc_cosmo *cosmo, int pk_n, const double *pk_logk,
                 const double *pk_logk3p, int FixedIC, int PairedIC, double *kdensity);
int orc_genic(int n, double box, const unsigned int *seed, double pknorm, const orc_cosmo *cosmo, double *kdensity) {
  return orc_genic_ic(n, box, seed, pknorm, cosmo, 0, 0, kdensity);
}
/* the same with the two run-time options of src/GenIC.c:370-376: PairedIC adds pi to every phase, FixedIC leaves the Rayleigh
   factor -log(ampl) out ("non-random modules of the Fourier modes") */
int orc_genic_ic(int n, double box, const unsigned int *seed, double pknorm, const orc_cosmo *cosmo, int FixedIC, int PairedIC, double *kdensity) {
  return orc_genic_pk(n, box, seed, pknorm, cosmo, 0, NULL, NULL, FixedIC, PairedIC, kdensity);
}
/* ... and with a tabulated spectrum (WhichSpectrum 2 and 5: FileWithInputSpectrum, CAMBTable): PowerSpec_Tabulated
   (src/cosmo.c:1432-1435) = 10^my_spline_eval(SPLINE[SP_PK], log10 k) / k^3 with the knots log10 k [1/Mpc], log10(k^3 P)
   of read_Pk_from_file / read_Pk_table_from_CAMB (:1099-1170, 1290-1330); pk_n = 0: Eisenstein & Hu */
/* PowerSpectrum() without its PkNorm (src/cosmo.c:953-1007): WhichSpectrum 1 Eisenstein & Hu, 2 / 5 tabulated, 3 the Efstathiou fit
   (PowerSpec_Efstathiou :1437-1440, SHAPE_EFST = 0.21 :44), 4 a power law (:1442-1445); then, for params.WDM_PartMass_in_kev > 0, the
   warm-dark-matter cut-off Tf^2 of Bode, Ostriker & Turok (the form just after their A7, :998-1004; unit_length_cm = UnitLength_in_cm) */
#define ORC_SHAPE_EFST ((double)0.21)
double orc_power_spectrum_form(double k, int which, const orc_cosmo *p, int pk_n, const double *pk_logk, const double *pk_logk3p, const double *pk_c,
                               double wdm_mass_kev, double unit_length_cm) {
  double power, alpha, Tf;
  switch (which) {
    case 1: power = orc_powerspec_EH(k, p); break;
    case 2: case 5: power = pow(10., orc_my_spline_eval(pk_logk, pk_logk3p, pk_c, pk_n, log10(k))) / k / k / k; break;
    case 3: power = pow(k, p->PrimordialIndex) / pow(1 + pow(6.4 / ORC_SHAPE_EFST * k + pow(3.0 / ORC_SHAPE_EFST * k, 1.5) + pow(1.7 / ORC_SHAPE_EFST, 2.0) * k * k, 1.13), 2 / 1.13); break;
    case 4: power = pow(k, p->PrimordialIndex); break;
    default: power = 0.0; break;
  }
  if (wdm_mass_kev > 0.) {
    alpha = 0.05 * pow((p->Omega0 - p->OmegaBaryon) / 0.4, 0.15) * pow(p->Hubble100 / 0.65, 1.3) * pow(1.0 / wdm_mass_kev, 1.15);
    Tf = pow(1 + pow(alpha * k / p->Hubble100 * (3.085678e24 / unit_length_cm), 2), -5.0);
    power *= Tf * Tf;
  }
  return power;
}
int orc_genic_form(int n, double box, const unsigned int *seed, double pknorm, const orc_cosmo *cosmo, int which, double wdm_mass_kev,
                   double unit_length_cm, int pk_n, const double *pk_logk, const double *pk_logk3p, int FixedIC, int PairedIC, double *kdensity);
int orc_genic_pk(int n, double box, const unsigned int *seed, double pknorm, const orc_cosmo *cosmo, int pk_n, const double *pk_logk,
                 const double *pk_logk3p, int FixedIC, int PairedIC, double *kdensity) {
  return orc_genic_form(n, box, seed, pknorm, cosmo, pk_n > 0 ? 2 : 1, 0.0, 3.085678e24, pk_n, pk_logk, pk_logk3p, FixedIC, PairedIC, kdensity);
}
int orc_genic_form(int n, double box, const unsigned int *seed, double pknorm, const orc_cosmo *cosmo, int which, double wdm_mass_kev,
                   double unit_length_cm, int pk_n, const double *pk_logk, const double *pk_logk3p, int FixedIC, int PairedIC, double *kdensity) {
  double *pk_c = NULL;
  if (pk_n > 0) {
    pk_c = (double *)malloc(sizeof(double) * pk_n);
    if (!pk_c || orc_cspline_coeffs(pk_logk, pk_logk3p, pk_n, pk_c)) { free(pk_c); return 1; }
  }
  const int Nmesh = n, Nsample = n, Nmesh_2 = n / 2, Nmesh_odd = n % 2;
  const int nzh = n / 2 + 1;
  const double Box = box;
  double fac = pow(1. / Box, 1.5);
  ranlxd_state random_generator, k0_generator;
  memset(kdensity, 0, sizeof(double) * 2 * (size_t)n * n * nzh);
  ranlxd_set(&random_generator, 1, 202);

  for (int i = 0; i < n; i++) {
    int ii = i;
    if (ii == Nmesh_2) continue;
    double kvec[3];
    if (ii < Nmesh_2) kvec[0] = ii * 2 * ORC_PI / Box; else kvec[0] = -(Nmesh - ii) * 2 * ORC_PI / Box;
    double kmag2_i = kvec[0] * kvec[0];
    int iii = ii;
    for (int j = 0; j < n; j++) {
      int jj = j;
      if (jj == Nmesh_2) continue;
      int jjj = jj;
      if (jj < Nmesh_2) kvec[1] = jj * 2 * ORC_PI / Box; else kvec[1] = -(Nmesh - jj) * 2 * ORC_PI / Box;
      double kmag2_ij = kmag2_i + kvec[1] * kvec[1];
      ranlxd_set(&random_generator, seed[(size_t)jj * Nmesh + ii], 202);
      double phase;
      for (int k = 0; (k < nzh) && (k < Nmesh_2); k++) {
        int kk = k;
        phase = ranlxd_uniform(&random_generator) * 2 * ORC_PI;
        double ampl;
        do ampl = ranlxd_uniform(&random_generator); while (ampl == 0);
        if (ii == 0 && jj == 0 && kk == 0) continue;
        if (kk == Nmesh_2) continue;
        if (kk < Nmesh_2) kvec[2] = kk * 2 * ORC_PI / Box; else kvec[2] = -(Nmesh - kk) * 2 * ORC_PI / Box;
        double kmag2_local = kmag2_ij + kvec[2] * kvec[2];
        double kmag = sqrt(kmag2_local);
        if (kmag * Box / (2 * ORC_PI) > 1. * Nsample / 2) continue; /* NYQUIST = 1. */
        double p_of_k = pknorm * orc_power_spectrum_form(kmag, which, cosmo, pk_n, pk_logk, pk_logk3p, pk_c, wdm_mass_kev, unit_length_cm);
        double sign = 1.0;
        int addr_j = j;
        iii = ii; jjj = jj;
        if (kk == 0) {
          if ((ii == 0) && (jj == Nmesh_2 || jj == Nmesh_2 + Nmesh_odd)) continue;
          if ((ii == Nmesh_2) || (ii == Nmesh_2 + Nmesh_odd)) continue;
          if ((ii > Nmesh_2) || (ii == 0 && jj > Nmesh / 2)) {
            jjj = Nmesh - jj;
            if (jjj == Nmesh) jjj = 0;
            if (Nmesh_odd && jj == Nmesh_2 + 1) { jjj = Nmesh_2 + 1; addr_j = Nmesh_2; }
            if (ii > Nmesh_2) iii = Nmesh - ii;
            sign = -1.0;
            ranlxd_set(&k0_generator, seed[(size_t)jjj * Nmesh + iii], 202);
            phase = ranlxd_uniform(&k0_generator) * 2 * ORC_PI;
            do ampl = ranlxd_uniform(&k0_generator); while (ampl == 0);
          }
        }
        if (PairedIC) phase += ORC_PI;
        if (!FixedIC) p_of_k *= -log(ampl);
        double delta = fac * sqrt(p_of_k);
        size_t addr = 2 * (((size_t)i * n + addr_j) * nzh + k);
        kdensity[addr] = delta * cos(phase);
        kdensity[addr + 1] = sign * delta * sin(phase);
      }
    }
  }
  fac = pow((double)Nmesh, 3.0);
  for (size_t i = 0; i < 2 * (size_t)n * n * nzh; i++) kdensity[i] *= fac;
  free(pk_c);
  return 0;
}
