// tools/osfft_emulate.cpp — host replay of waa_osfft.hip (tests/test_osfft_emulation.py): the header the kernel is built from
// (web-audio-api-rs_amd/csrc/waa_osfft.hpp) compiled for the HOST — the packed-f32 primitives fall back to plain C++ with the
// same operations in the same order — and the kernel's choreography walked group by group: 16 "lanes", the exchange buffer,
// the run heads (the two processed quanta in front of a run rendered without being stored), LINK_SKIP / LINK_FRESH.
//   osfft_emulate <in.bin> <out.bin>
//   in.bin : int32 R, nch, n_quanta, curve_n, seg_len; int32 prev[n_quanta]; float curve[curve_n]; float x[nch][n_quanta * 128]
//   out.bin: float y[nch][n_quanta * 128]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../web-audio-api-rs_amd/csrc/waa_osfft_tables.hpp"

using namespace waa::osfft;
constexpr int32_t LINK_SKIP = -2, LINK_FRESH = -1;
constexpr int RQ = 128;

template <int R>
static void run(int nch, int nq, int seg_len, const std::vector<int32_t>& prev, const std::vector<float>& curve,
                const std::vector<float>& x, std::vector<float>& y) {
  const std::vector<float> tabf = tables(R), twf = tw256();
  const c2v* tab = reinterpret_cast<const c2v*>(tabf.data());
  const c2v* tw = reinterpret_cast<const c2v*>(twf.data());
  const int n_seg = (nq + seg_len - 1) / seg_len;
  const int cn = (int)curve.size();
  for (int seg = 0; seg < n_seg; seg++) {
    const int q_lo = seg * seg_len, q_hi = q_lo + seg_len < nq ? q_lo + seg_len : nq;
    int p1 = -1, p2 = -1;
    for (int qq = q_lo - 1; qq >= 0; qq--)
      if (prev[(size_t)qq] != LINK_SKIP) {
        p1 = qq;
        break;
      }
    if (p1 >= 0 && prev[(size_t)p1] >= 0) p2 = prev[(size_t)p1];
    std::vector<Lane<R>> L(16);
    std::vector<c2v> ex(XSLOTS);
    for (int t = 0; t < 16; t++) {
      load_tw(tw, t, L[t].tws);
      lane_reset(L[t]);
    }
    auto exchange = [&]() {
      for (int t = 0; t < 16; t++) xwrite(L[t].a, ex.data(), t);
      for (int t = 0; t < 16; t++) xread(L[t].a, ex.data(), t);
    };
    for (int it = 0; it < seg_len + 2; it++) {
      const int q = it == 0 ? p2 : it == 1 ? p1 : q_lo + it - 2;
      const bool in_run = it >= 2 && q < q_hi;
      const bool valid = q >= 0 && (it < 2 || in_run);
      const int32_t link = valid ? prev[(size_t)q] : LINK_SKIP;
      const bool proc = valid && link != LINK_SKIP;
      if (!proc) {
        if (in_run)
          for (int c = 0; c < nch; c++)
            for (int i = 0; i < RQ; i++) y[((size_t)c * nq + q) * RQ + i] = 0.f;
        continue;
      }
      if (link == LINK_FRESH)
        for (int t = 0; t < 16; t++) lane_reset(L[t]);
      for (int t = 0; t < 16; t++) {
        c2v xin[8];
        for (int j = 0; j < 8; j++) {
          xin[j].x = x[((size_t)0 * nq + q) * RQ + 16 * j + t];
          xin[j].y = nch == 2 ? x[((size_t)1 * nq + q) * RQ + 16 * j + t] : 0.f;
        }
        ph_in(L[t], xin);
      }
      exchange();
      for (int t = 0; t < 16; t++) ph_spec(L[t]);
      const float c_first = cn ? curve[0] : 0.f, c_last = cn ? curve[(size_t)cn - 1] : 0.f;
      for (int r = 0; r < R; r++) {
        for (int t = 0; t < 16; t++) ph_up(L[t], tab + r * TAB_SLOTS, t);
        exchange();
        for (int t = 0; t < 16; t++) {
          c2v uu[8];
          ph_up_out(L[t], r, proc, uu);
          for (int j = 0; j < 8; j++) uu[j] = shape2<false>(curve.data(), cn, c_first, c_last, uu[j]);
          ph_dn(L[t], uu);
        }
        exchange();
        for (int t = 0; t < 16; t++) ph_dn_acc(L[t], r, tab + (R + r) * TAB_SLOTS, t);
      }
      for (int t = 0; t < 16; t++) ph_out(L[t]);
      exchange();
      for (int t = 0; t < 16; t++) {
        c2v o[8];
        ph_out_end(L[t], proc, o);
        if (in_run)
          for (int j = 0; j < 8; j++) {
            y[((size_t)0 * nq + q) * RQ + 16 * j + t] = o[j].x;
            if (nch == 2) y[((size_t)1 * nq + q) * RQ + 16 * j + t] = o[j].y;
          }
      }
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[5];
  if (fread(hdr, 4, 5, f) != 5) return 3;
  const int R = hdr[0], nch = hdr[1], nq = hdr[2], cn = hdr[3], seg = hdr[4];
  std::vector<int32_t> prev((size_t)nq);
  std::vector<float> curve((size_t)cn), x((size_t)nch * nq * RQ), y((size_t)nch * nq * RQ, -777.f);
  if (fread(prev.data(), 4, prev.size(), f) != prev.size() || fread(curve.data(), 4, curve.size(), f) != curve.size() ||
      fread(x.data(), 4, x.size(), f) != x.size())
    return 3;
  fclose(f);
  if (R == 2)
    run<2>(nch, nq, seg, prev, curve, x, y);
  else
    run<4>(nch, nq, seg, prev, curve, x, y);
  f = fopen(argv[2], "wb");
  fwrite(y.data(), 4, y.size(), f);
  fclose(f);
  return 0;
}
