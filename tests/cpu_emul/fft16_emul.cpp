// Host emulation of the sixteen-points-per-thread plan (pinocchio_amd/csrc/pf_fft16.h): the 128 threads of one column of a
// workgroup run one after the other through load / stage 0 / exchange 1 / stage 1 / exchange 2 / stage 2 / store with the
// header's own index functions and butterflies; `lds` is an array of slots.  A unit test of that header's index algebra
// (tests/test_fft_core.py), not a CPU path of the library.  alg 0: the double algebra; alg 1: the (re, im)-pair fp32 algebra in
// its host form.
#include <cmath>
#include <vector>

#include "../../pinocchio_amd/csrc/pf_fft16.h"

template <typename A, int DIR, typename MK, typename GET>
static void run(const double *in, double *out, MK mk, GET get) {
  typedef typename A::C C;
  typedef typename A::TW TW;
  const int N = PF16_N;
  std::vector<TW> tw(N);
  for (int j = 0; j < N; j++) tw[j] = mk(std::cos(2.0 * M_PI * j / N), std::sin(2.0 * M_PI * j / N));
  std::vector<C> lds(2048), reg(128 * 16);
  auto R = [&](int tl, int m) -> C & { return reg[tl * 16 + m]; };
  for (int tl = 0; tl < 128; tl++) {
    C v[16];
    for (int m = 0; m < 16; m++) { const int e = pf16_line_index(tl, m); v[m] = mk(in[2 * e], in[2 * e + 1]); }
    pfx_bfly16<A, DIR>(v);
    for (int m = 0; m < 16; m++) R(tl, m) = v[m];
  }
  for (int tl = 0; tl < 128; tl++) for (int t = 0; t < 16; t++) lds[pf16_x1_write(tl, t)] = R(tl, t);
  for (int tl = 0; tl < 128; tl++) for (int r = 0; r < 16; r++) R(tl, r) = lds[pf16_x1_read(tl, r)];
  for (int tl = 0; tl < 128; tl++) {
    C v[16];
    const int w = tl >> 3;
    v[0] = R(tl, 0);
    for (int r = 1; r < 16; r++) v[r] = A::template cmul_s<DIR>(R(tl, r), tw[pf16_tw1(w) * r]);
    pfx_bfly16<A, DIR>(v);
    for (int m = 0; m < 16; m++) R(tl, m) = v[m];
  }
  for (int b = 0; b < 2; b++) {
    for (int tl = 0; tl < 128; tl++) for (int s8 = 0; s8 < 8; s8++) lds[pf16_x2_write(tl, s8, 0)] = R(tl, 8 * b + s8);
    for (int tl = 0; tl < 128; tl++) for (int r = 0; r < 8; r++) R(tl, 8 * b + r) = lds[pf16_x2_read(tl, r, 0)];
  }
  for (int tl = 0; tl < 128; tl++)
    for (int b = 0; b < 2; b++) {
      TW wp[7];
      pfx_powers7<A>(tw[pf16_tw2(tl, b)], wp);
      C u[8];
      u[0] = R(tl, 8 * b);
      for (int r = 1; r < 8; r++) u[r] = A::template cmul<DIR>(R(tl, 8 * b + r), wp[r - 1]);
      pfx_bfly8<A, DIR>(u);
      for (int r = 0; r < 8; r++) { const int k = pf16_out_index(tl, 8 * b + r); get(u[r], out[2 * k], out[2 * k + 1]); }
    }
}

extern "C" int emul_fft16(int alg, int dir, const double *in, double *out) {
  auto mkd = [](double re, double im) { return pf_mk<double>(re, im); };
  auto getd = [](pfc<double> c, double &re, double &im) { re = c.x; im = c.y; };
  auto mkf = [](double re, double im) { return PfCxPk::mk((float)re, (float)im); };
  auto getf = [](pf_f2 c, double &re, double &im) { re = c.x; im = c.y; };
  if (alg == 0) { if (dir > 0) run<PfCxStd, +1>(in, out, mkd, getd); else run<PfCxStd, -1>(in, out, mkd, getd); }
  else { if (dir > 0) run<PfCxPk, +1>(in, out, mkf, getf); else run<PfCxPk, -1>(in, out, mkf, getf); }
  return 0;
}
