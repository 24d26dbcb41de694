// tools/fft3_emulate.cpp — host replay of the 32 x 32 x 16 register-pass FFT of waa_conv3.hip (tests/test_fft3_emulation.py).
// Compiles web-audio-api-rs_amd/csrc/waa_fft3.hpp for the HOST (plain C++ forms of the packed-f32 primitives, same
// operations in the same order) and walks the kernel's choreography — 512 "threads", the shared E1 / E2 / T2 buffers,
// barriers = loop boundaries — for one forward and one inverse transform.
//   fft3_emulate <in.bin> <spec.bin> <inv.bin>: reads 16384 complex f32, writes the forward spectrum in the kernel's
//   POSITION order (position k3 * 1024 + k1 * 32 + k2 holds bin k1 + 32 k2 + 1024 k3) and the unscaled inverse of that
//   spectrum (all 16384 outputs for the even slots a separate full pass is run; the kernel itself only forms 8192..16383).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../web-audio-api-rs_amd/csrc/waa_fft3.hpp"

using namespace waa::fft3;

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::vector<c2v> z(N), tw(N), spec(N), out(N);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(z.data(), sizeof(c2v), N, f) != (size_t)N) return 3;
  fclose(f);
  for (int j = 0; j < N; j++) {
    const double a = -2.0 * 3.14159265358979323846 * (double)j / (double)N;
    tw[j] = c2v{(float)std::cos(a), (float)std::sin(a)};
  }
  std::vector<c2v> lds(LDS_SLOTS);
  std::vector<std::vector<c2v>> regs(NT, std::vector<c2v>(32));
  auto X = [&](int t) -> c2v(&)[32] { return *reinterpret_cast<c2v(*)[32]>(regs[t].data()); };
  // ---- forward ----
  for (int t = 0; t < NT; t++) {
    c2v tws[32];
    load_tw1_slots(tw.data(), t, tws);
    for (int n1 = 0; n1 < 32; n1++) X(t)[n1] = z[n1 * 512 + t];
    if (getenv("F3_HALF_TW")) {  // (the filter-stage kernel's form of pass 1)
      c2v twh[16], w16;
      load_tw1_half(tw.data(), t, twh, w16);
      fwd_pass1_compute_half(X(t), twh, w16);
    } else {
      fwd_pass1_compute(X(t), tws);
    }
    fwd_pass1_write(X(t), lds.data(), t);
  }
  for (int t = 0; t < NT; t++) fwd_pass2_compute(X(t), lds.data(), t);
  for (int t = 0; t < NT; t++) fwd_pass2_write(X(t), lds.data(), t);
  for (int t = 0; t < NT; t++)
    for (int set = 0; set < 2; set++) {
      const int r = t + set * NT;
      c2v y[16], tw3[16];
      load_tw3(tw.data(), t, tw3);
      fwd_pass3(y, tw3, lds.data(), r);
      for (int s = 0; s < 16; s++) spec[K16(s) * 1024 + r] = y[s];
    }
  f = fopen(argv[2], "wb");
  fwrite(spec.data(), sizeof(c2v), N, f);
  fclose(f);
  // ---- inverse ----
  for (int t = 0; t < NT; t++)
    for (int set = 0; set < 2; set++) {
      const int r = t + set * NT;
      c2v y[16];
      for (int k3 = 0; k3 < 16; k3++) y[k3] = spec[k3 * 1024 + r];
      c2v tw3[16];
      load_tw3(tw.data(), t, tw3);
      inv_pass1_compute(y, tw3);
      inv_pass1_write(y, lds.data(), r);
    }
  for (int t = 0; t < NT; t++) inv_pass2_compute(X(t), lds.data(), t);
  for (int t = 0; t < NT; t++) inv_pass2_write(X(t), lds.data(), t);
  for (int t = 0; t < NT; t++) {
    c2v twn[32];
    load_tw1_natural(tw.data(), t, twn);
    inv_pass3(X(t), twn, lds.data(), t);
    for (int s = 1; s < 32; s += 2) out[K32(s) * 512 + t] = X(t)[s];  // second half of the window only (as the kernel)
  }
  // the first half, through the unpruned transform, so that the test sees the whole inverse
  for (int t = 0; t < NT; t++) {
    c2v twn[32], x[32];
    load_tw1_natural(tw.data(), t, twn);
    for (int k1 = 0; k1 < 32; k1++) x[k1] = lds[e1(k1, 0) + t];
    for (int k1 = 1; k1 < 32; k1++) x[k1] = cmulc(x[k1], twn[k1]);
    dft32<true>(x);
    for (int s = 0; s < 32; s += 2) out[K32(s) * 512 + t] = x[s];
  }
  f = fopen(argv[3], "wb");
  fwrite(out.data(), sizeof(c2v), N, f);
  fclose(f);
  return 0;
}
