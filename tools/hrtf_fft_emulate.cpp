// tools/hrtf_fft_emulate.cpp — host replay of waa_hrtf_fft.hip (tests/test_hrtf_fft_emulation.py): the header the kernel is built
// from compiled for the HOST, the choreography of one group walked lane by lane — runs of quanta, the four processed quanta in
// front of a run rendered without being stored, LINK_SKIP / LINK_FRESH.
//   hrtf_fft_emulate <in.bin> <out.bin>
//   in.bin : int32 taps, n_quanta, seg_len; int32 prev[n_quanta]; float pair[taps][2]; float x[n_quanta * 128] (the mono mix)
//   out.bin: float y[2][n_quanta * 128]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../web-audio-api-rs_amd/csrc/waa_hrtf_fft_tables.hpp"

using namespace waa::hrtffft;
constexpr int32_t LINK_SKIP = -2, LINK_FRESH = -1;
constexpr int RQ = 128;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 3;
  const int taps = hdr[0], nq = hdr[1], seg_len = hdr[2];
  std::vector<int32_t> prev((size_t)nq);
  std::vector<float> pair((size_t)taps * 2), x((size_t)nq * RQ), y((size_t)2 * nq * RQ, -777.f);
  if (fread(prev.data(), 4, prev.size(), f) != prev.size() || fread(pair.data(), 4, pair.size(), f) != pair.size() ||
      fread(x.data(), 4, x.size(), f) != x.size())
    return 3;
  fclose(f);
  const std::vector<float> tabf = make_tables(pair.data(), taps), twf = tw256();
  const c2v* tab = reinterpret_cast<const c2v*>(tabf.data());
  const c2v* tw = reinterpret_cast<const c2v*>(twf.data());
  const int n_seg = (nq + seg_len - 1) / seg_len;
  for (int seg = 0; seg < n_seg; seg++) {
    const int q_lo = seg * seg_len, q_hi = q_lo + seg_len < nq ? q_lo + seg_len : nq;
    int ph[HEADS];  // ph[HEADS - 1] = the nearest processed quantum in front of the run, ph[0] the farthest
    for (int k = 0; k < HEADS; k++) ph[k] = -1;
    {
      int p = -1;
      for (int qq = q_lo - 1; qq >= 0; qq--)
        if (prev[(size_t)qq] != LINK_SKIP) {
          p = qq;
          break;
        }
      for (int k = HEADS - 1; k >= 0 && p >= 0; k--) {
        ph[k] = p;
        p = prev[(size_t)p] >= 0 ? prev[(size_t)p] : -1;
      }
    }
    std::vector<HLane> L(16);
    std::vector<c2v> ex(XSLOTS);
    for (int t = 0; t < 16; t++) {
      load_tw(tw, t, L[t].tws);
      lane_reset_if(L[t], true);
    }
    auto exchange = [&](bool second) {
      for (int t = 0; t < 16; t++) xwrite(second ? L[t].Y : L[t].a, ex.data(), t);
      for (int t = 0; t < 16; t++) xread(second ? L[t].Y : L[t].a, ex.data(), t);
    };
    for (int it = 0; it < seg_len + HEADS; it++) {
      const int q = it < HEADS ? ph[it] : q_lo + it - HEADS;
      const bool in_run = it >= HEADS && q < q_hi;
      const bool valid = q >= 0 && (it < HEADS || in_run);
      const int32_t link = valid ? prev[(size_t)q] : LINK_SKIP;
      const bool proc = valid && link != LINK_SKIP;
      // (the kernel runs every step for every group, masked; so does the replay)
      for (int t = 0; t < 16; t++) {
        lane_reset_if(L[t], proc && link == LINK_FRESH);
        float xin[8];
        for (int j = 0; j < 8; j++) xin[j] = proc ? x[(size_t)q * RQ + 16 * j + t] : 0.f;
        ph_in(L[t], xin);
      }
      exchange(false);
      for (int t = 0; t < 16; t++) ph_spec(L[t], tab, t, proc);
      exchange(true);
      for (int t = 0; t < 16; t++) {
        c2v o[8];
        ph_out(L[t], proc, o);
        if (in_run)
          for (int j = 0; j < 8; j++) {
            y[((size_t)0 * nq + q) * RQ + 16 * j + t] = proc ? o[j].x : 0.f;
            y[((size_t)1 * nq + q) * RQ + 16 * j + t] = proc ? o[j].y : 0.f;
          }
      }
    }
  }
  f = fopen(argv[2], "wb");
  fwrite(y.data(), 4, y.size(), f);
  fclose(f);
  return 0;
}
