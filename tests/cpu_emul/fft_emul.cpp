// Host-side unit-test driver for pinocchio_amd/csrc/pf_fft_core.h: a serial loop
// over the thread index tl stands in for the wavefront, a plain array for LDS.
// TEST ONLY -- this is not a CPU path of the library (which has none).
#include <cmath>
#include <vector>
#include "../../pinocchio_amd/csrc/pf_fft_core.h"

typedef pfc<double> C;

template <int N, int S, int DIR>
static void run_stages(std::vector<C> &reg, std::vector<C> &lds, const C *tw, int tws_dummy) {
  constexpr int NT = N / 8;
  if constexpr (S < pf_nstages(N)) {
    for (int tl = 0; tl < NT; tl++) {
      C v[8];
      for (int m = 0; m < 8; m++) v[m] = reg[tl * 8 + m];
      pf_stage<double, N, S, DIR, 1>(v, tl, tw);
      for (int m = 0; m < 8; m++) lds[pf_stage_pos<N, S>(tl, m)] = v[m];
    }
    for (int tl = 0; tl < NT; tl++)
      for (int m = 0; m < 8; m++) reg[tl * 8 + m] = lds[tl + m * NT];
    run_stages<N, S + 1, DIR>(reg, lds, tw, 0);
  }
}

// the paired plan (pf_fft_core.h, p16): first stage by thread pairs -- every thread's own part first, then the combination with
// what the partner holds (on the device: through the lanes of the wave) --, radix-8 stages after it
template <int N, int S, int DIR>
static void run_stages_p16(std::vector<C> &reg, std::vector<C> &lds, const C *tw) {
  constexpr int NT = N / 8;
  if constexpr (S < pf_nstages(N, true)) {
    if constexpr (S == 0) {
      for (int tl = 0; tl < NT; tl++) {
        C v[8];
        for (int m = 0; m < 8; m++) v[m] = reg[tl * 8 + m];
        pf_pair16_local<DIR>(v, tl);
        for (int m = 0; m < 8; m++) reg[tl * 8 + m] = v[m];
      }
      std::vector<C> part(reg);
      for (int tl = 0; tl < NT; tl++) {
        C v[8], o[8];
        for (int m = 0; m < 8; m++) { v[m] = part[tl * 8 + m]; o[m] = part[(tl ^ 1) * 8 + m]; }
        pf_pair16_combine(v, o, tl);
        for (int m = 0; m < 8; m++) lds[pf_stage_pos<N, 0, true>(tl, m)] = v[m];
      }
    } else {
      for (int tl = 0; tl < NT; tl++) {
        C v[8], w[1];
        for (int m = 0; m < 8; m++) v[m] = reg[tl * 8 + m];
        pf_stage_twiddles<double, N, S, DIR, 1, 0, true>(tl, tw, w);
        pf_stage_apply<double, N, S, DIR, 0, true>(v, w);
        for (int m = 0; m < 8; m++) lds[pf_stage_pos<N, S, true>(tl, m)] = v[m];
      }
    }
    for (int tl = 0; tl < NT; tl++)
      for (int m = 0; m < 8; m++) reg[tl * 8 + m] = lds[tl + m * NT];
    run_stages_p16<N, S + 1, DIR>(reg, lds, tw);
  }
}
template <int N, int DIR>
static void fft_line_p16(const C *in, C *out) {
  constexpr int NT = N / 8;
  std::vector<C> tw(N), reg(N), lds(N);
  for (int j = 0; j < N; j++) tw[j] = pf_mk<double>(cos(2 * M_PI * j / N), sin(2 * M_PI * j / N));
  for (int tl = 0; tl < NT; tl++)
    for (int m = 0; m < 8; m++) reg[tl * 8 + m] = in[pf_line_index<N, true>(tl, m)];
  run_stages_p16<N, 0, DIR>(reg, lds, tw.data());
  for (int tl = 0; tl < NT; tl++)
    for (int m = 0; m < 8; m++) out[tl + m * NT] = reg[tl * 8 + m];
}
extern "C" int emul_fft_p16(int n, int dir, const double *in, double *out) {
  if (n == 128) { if (dir > 0) fft_line_p16<128, +1>((const C *)in, (C *)out); else fft_line_p16<128, -1>((const C *)in, (C *)out); }
  else if (n == 1024) { if (dir > 0) fft_line_p16<1024, +1>((const C *)in, (C *)out); else fft_line_p16<1024, -1>((const C *)in, (C *)out); }
  else return 1;
  return 0;
}

template <int N, int DIR>
static void fft_line(const C *in, C *out) {
  constexpr int NT = N / 8;
  std::vector<C> tw(N), reg(N), lds(N);
  for (int j = 0; j < N; j++) tw[j] = pf_mk<double>(cos(2 * M_PI * j / N), sin(2 * M_PI * j / N));
  for (int tl = 0; tl < NT; tl++)
    for (int m = 0; m < 8; m++) reg[tl * 8 + m] = in[tl + m * NT];
  run_stages<N, 0, DIR>(reg, lds, tw.data(), 0);
  for (int tl = 0; tl < NT; tl++)
    for (int m = 0; m < 8; m++) out[tl + m * NT] = reg[tl * 8 + m];
}

#define DISPATCH(FN, n, ...)                    \
  switch (n) {                                  \
    case 8: FN<8> (__VA_ARGS__); break;         \
    case 16: FN<16>(__VA_ARGS__); break;        \
    case 32: FN<32>(__VA_ARGS__); break;        \
    case 64: FN<64>(__VA_ARGS__); break;        \
    case 128: FN<128>(__VA_ARGS__); break;      \
    case 256: FN<256>(__VA_ARGS__); break;      \
    case 512: FN<512>(__VA_ARGS__); break;      \
    case 1024: FN<1024>(__VA_ARGS__); break;    \
    case 2048: FN<2048>(__VA_ARGS__); break;    \
    default: return 1;                          \
  }

template <int N> static void fwd(const C *i, C *o) { fft_line<N, -1>(i, o); }
template <int N> static void inv(const C *i, C *o) { fft_line<N, +1>(i, o); }

extern "C" int emul_fft(int n, int dir, const double *in, double *out) {
  if (dir > 0) { DISPATCH(inv, n, (const C *)in, (C *)out) } else { DISPATCH(fwd, n, (const C *)in, (C *)out) }
  return 0;
}

// length-n real line <-> n/2+1 complex, via the half-size complex FFT
extern "C" int emul_c2r(int n, const double *spec, double *real_out) {
  const int M = n / 2;
  const C *X = (const C *)spec;
  std::vector<C> z(M), o(M);
  for (int k = 0; k < M; k++) {
    C wk = pf_mk<double>(cos(2 * M_PI * k / n), sin(2 * M_PI * k / n));
    z[k] = pf_c2r_pre<double>(X[k], X[M - k], wk, k == 0);
  }
  if (emul_fft(M, +1, (const double *)z.data(), (double *)o.data())) return 1;
  for (int i = 0; i < M; i++) { real_out[2 * i] = o[i].x; real_out[2 * i + 1] = o[i].y; }
  return 0;
}

extern "C" int emul_r2c(int n, const double *real_in, double *spec) {
  const int M = n / 2;
  std::vector<C> z(M), Z(M);
  for (int i = 0; i < M; i++) z[i] = pf_mk<double>(real_in[2 * i], real_in[2 * i + 1]);
  if (emul_fft(M, -1, (const double *)z.data(), (double *)Z.data())) return 1;
  C *X = (C *)spec;
  for (int k = 0; k < M; k++) {
    C wk = pf_mk<double>(cos(2 * M_PI * k / n), sin(2 * M_PI * k / n));
    X[k] = pf_r2c_post<double>(Z[k], Z[(M - k) & (M - 1)], wk);
  }
  X[M] = pf_mk<double>(Z[0].x - Z[0].y, 0.0);
  return 0;
}
