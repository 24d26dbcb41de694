// waa_plan_conv.cpp — ConvolverNode: block size, spectra of the impulse response, the forward / product / inverse steps
// (split out of waa_plan.cpp in round 4).
#include <array>
#include <set>

#include "waa_host.hpp"
#include "waa_plan_parts.hpp"

namespace waa {
namespace host {

// the partition size the FFT path picks for a node's impulse response (0: none / all-zero / direct FIR)
int conv_block_size(const waa_batch* b, const Node& n) {
  if (!n.has_ir) return 0;
  uint64_t len = 0;
  for (int c = 0; c < n.ir_nch; c++) {
    uint64_t l = n.ir_len;
    while (l > 0 && std::fabs(n.ir[c][l - 1]) < 0.000001f) l--;
    len = std::max(len, l);
  }
  if (len == 0) return 0;
  if (len <= (uint64_t)DIRECT_MAX_TAPS && !b->dynamic && !measure_switch("WAA_NO_DIRECT_FIR")) return 0;
  for (int cand : {128, 512, 2048, 8192})
    if ((len + cand - 1) / cand <= 24) return cand;
  return 8192;
}

int plan_convolver(waa_batch* b, uint32_t id) {
  Node& n = b->nodes[id];
  SignalRef in_sig{};
  uint64_t in_valid = b->lp;
  if (b->dynamic && n.hist.base) {
    in_sig = n.hist;  // dynamic plans: the mixed input was published by the DK_CONV_IN item (waa_dyn.hip)
  } else if (n.pre_biquad >= 0) {
    // the Biquad in front is rendered by the forward transform: the transform reads the BIQUAD's input
    const Node& q = b->nodes[(uint32_t)n.pre_biquad];
    const Node& sn = b->nodes[b->edges[q.in_edges[0]].from];
    in_sig = sn.is_view ? sn.view_sig : sn.sig;
    in_valid = sn.is_view ? sn.view_valid : b->lp;
    if (!in_sig.base) return fail(WAA_ERR_INVALID_STATE, "internal: the input of folded biquad node %d is not planned yet", n.pre_biquad);
  } else {
    int e = node_input_signal(b, id, &in_sig, nullptr, &in_valid);
    if (e) return e;
  }
  // one FFTConvolver per IR channel, at least two (convolver.rs:291-306); each trims its own trailing
  // |h| < 1e-6 samples (fft-convolver init) — only the longest trimmed length matters here
  const int ir_nch = n.ir_nch;
  uint64_t len = 0;
  for (int c = 0; c < ir_nch; c++) {
    uint64_t l = n.ir_len;
    while (l > 0 && std::fabs(n.ir[c][l - 1]) < 0.000001f) l--;
    len = std::max(len, l);
  }
  Step st;
  st.kind = 2;
  ConvDesc& cv = st.conv;
  std::memset(&cv, 0, sizeof cv);
  if (len == 0) {
    // all-zero impulse response: FFTConvolver::process outputs zeros
    Step z;
    z.kind = 3;
    z.zero_ptr = n.sig.base;
    z.zero_bytes = (size_t)b->n_inst * n.out_nch * b->lp * sizeof(float);
    b->steps.push_back(z);
    plan_note(b, "convolver node %u: all-zero impulse response -> zero fill", id);
    return 0;
  }
  // Short impulse responses: the direct FIR is exact where the reference's FFT convolver leaves roundoff noise (its
  // delta-IR tests ask for 1e-7).  In a dynamic-count plan that difference is audible further down: silence is DATA
  // dependent there (a DelayNode reports silence when it read nothing but zeros, delay.rs:660-668; filter tails end
  // when their state leaves the normal range), and exact zeros behind a convolver that has seen input turn "still
  // ringing with noise, stereo" into "silent, mono" for every count-sensitive node behind it.  Dynamic plans therefore
  // take the FFT form for every length, like the reference (fuzz seeds 1658, 1340 of the 1500-seed runs).
  const bool direct_fir = len <= (uint64_t)DIRECT_MAX_TAPS && !b->dynamic && !measure_switch("WAA_NO_DIRECT_FIR");
  int B = 8192;
  for (int cand : {128, 512, 2048, 8192})
    if ((len + cand - 1) / cand <= 24) {
      B = cand;
      break;
    }
  cv.block = B;
  cv.n = 2 * B;
  cv.fft3 = cv.n == 16384 && !measure_switch("WAA_CONV_FFT_R4");  // (the round-2 radix-4-in-LDS kernels: same-box A/B only)
  cv.parts = (int)((len + B - 1) / B);
  cv.nb = (int)((b->lp + B - 1) / B);
  cv.cin = n.in_nch;
  cv.cout = n.out_nch;
  cv.in = in_sig;
  cv.out = n.sig;
  cv.frames = b->lp;
  cv.in_valid = in_valid;
  cv.n_inst = b->n_inst;
  cv.n_pairs = (b->n_inst + 1) / 2;
  cv.ir_nch = ir_nch;
  cv.ir_len = len;
  cv.kb0 = 0;
  cv.kb1 = direct_fir ? (int)((b->lp + 1023) / 1024) : cv.nb;  // (the direct kernel works in 1024-frame pieces)
  // routing (convolver.rs:384-466)
  auto term = [&](int in_ch, int ir_ch, int out_ch) { cv.terms[cv.n_terms++] = ConvTerm{in_ch, ir_ch, out_ch, 0}; };
  if (n.in_nch == 1 && ir_nch == 1) {
    term(0, 0, 0);
  } else if (n.in_nch == 1 && ir_nch == 2) {
    term(0, 0, 0);
    term(0, 1, 1);
  } else if (n.in_nch == 2 && ir_nch == 1) {
    term(0, 0, 0);
    term(1, 0, 1);
  } else if (n.in_nch == 2 && ir_nch == 2) {
    term(0, 0, 0);
    term(1, 1, 1);
  } else if (n.in_nch == 2 && ir_nch == 4) {
    term(0, 0, 0);
    term(0, 1, 1);
    term(1, 2, 0);
    term(1, 3, 1);
  } else {
    term(0, 0, 0);
    term(0, 1, 1);
    term(0, 2, 0);
    term(0, 3, 1);
  }
  // device resources
  std::vector<float> irflat((size_t)ir_nch * len);
  for (int c = 0; c < ir_nch; c++) {
    uint64_t l = n.ir_len;
    while (l > 0 && std::fabs(n.ir[c][l - 1]) < 0.000001f) l--;  // samples past a channel's own trim are dropped
    for (uint64_t i = 0; i < len; i++) irflat[(size_t)c * len + i] = i < l ? n.ir[c][i] : 0.f;
  }
  float* d_ir = nullptr;
  int e = dev_upload(b, &d_ir, irflat);
  if (e) return e;
  cv.ir = d_ir;
  if (direct_fir) {
    st.kind = 4;
    st.slot_mac = slot_for(b, "conv_direct_kernel");
    b->steps.push_back(st);
    plan_note(b, "convolver node %u: direct FIR taps=%llu cin=%d cout=%d terms=%d", id, (unsigned long long)len, cv.cin,
              cv.cout, cv.n_terms);
    return 0;
  }
  std::vector<Cplx> tw(cv.n);
  for (int t = 0; t < cv.n; t++) {
    const double a = -2.0 * 3.14159265358979323846 * (double)t / (double)cv.n;
    tw[t] = Cplx{(float)std::cos(a), (float)std::sin(a)};
  }
  Cplx* d_tw = nullptr;
  if ((e = dev_upload(b, &d_tw, tw))) return e;
  cv.tw = d_tw;
  Cplx *dH = nullptr, *dX = nullptr, *dY = nullptr;
  if ((e = dev_alloc(b, &dH, (size_t)ir_nch * cv.parts * cv.n))) return e;
  if ((e = dev_alloc(b, &dX, (size_t)cv.n_pairs * cv.cin * cv.nb * cv.n))) return e;
  if ((e = dev_alloc(b, &dY, (size_t)cv.n_pairs * cv.cout * cv.nb * cv.n))) return e;
  cv.H = dH;
  cv.X = dX;
  cv.Y = dY;
  if (n.pre_biquad >= 0) {
    if (!cv.fft3) return fail(WAA_ERR_INVALID_STATE, "internal: biquad node %d folded into a convolver without the three-pass transforms", n.pre_biquad);
    std::vector<OpDesc> qops;
    int q_out = 0;
    if ((e = emit_node_ops(b, (uint32_t)n.pre_biquad, cv.cin, true, qops, &q_out))) return e;
    if (qops.size() != 1 || qops[0].kind != OP_BIQUAD || qops[0].i0 != 0)
      return fail(WAA_ERR_INVALID_STATE, "internal: folded biquad node %d is not a constant-coefficient filter", n.pre_biquad);
    cv.pre_coefs = reinterpret_cast<const double*>(qops[0].ptr0);
    cv.pre_coef_stride = qops[0].u0;
    cv.pre_state = reinterpret_cast<double*>(qops[0].ptr1);
  }
  if (!b->dry) {
    launch_conv_ir_spectra(cv, b->stream);  // control-side work of ConvolverNode::set_buffer, once
    HIP_TRY(hipGetLastError());
  }
  plan_note(b, "convolver node %u: fft B=%d N=%d P=%d blocks=%d pairs=%u cin=%d cout=%d terms=%d ir_len=%llu%s", id, cv.block,
            cv.n, cv.parts, cv.nb, cv.n_pairs, cv.cin, cv.cout, cv.n_terms, (unsigned long long)len,
            cv.pre_coefs ? " (+ the Biquad in front, in the forward transform)" : "");
  st.slot_fwd = slot_for(b, "conv_fft_kernel<fwd>");
  st.slot_mac = slot_for(b, "conv_mac_kernel");
  st.slot_inv = slot_for(b, "conv_fft_kernel<inv>");
  b->steps.push_back(st);
  return 0;
}

}  // namespace host
}  // namespace waa
