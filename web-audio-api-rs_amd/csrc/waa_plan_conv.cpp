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

// The Biquad in front as part of the impulse response.  Both nodes are linear and time-invariant when the filter's four
// AudioParams are constants, and every context of the batch convolves with the same response — so if they also share the
// filter's coefficients, the cascade  source -> Biquad -> Convolver  IS one convolution with  h' = biquad(h):  the response the
// reference hands to FFTConvolver::init (normalised, each channel's trailing |h| < 1e-6 samples dropped, convolver.rs:259-317)
// run through the reference's own Direct Form I recurrence in f64 (biquad_filter.rs:877), followed by the filter's ringing.
// The ringing is cut where what is left of it cannot matter: tail L1 norm <= 1e-12 of the response's L1 norm (an output error of
// at most 1e-12 x the largest output the response can produce) — and the fold is only taken when that point lies inside the
// simulated extension (two blocks) with the geometric remainder bounded the same way, i.e. never for a filter with a long
// memory (poles within ~1e-3 of the unit circle): those keep conv_fft3_fwd_bq_kernel, the exact-order filter inside the
// forward transform.  What changes numerically: the filter acts in the f32 frequency domain (like the convolver itself)
// instead of as an f64 recurrence in front of it; the every-instance T1 / C4 tests hold the 1e-6 RMS bar against the oracle's
// sample-by-sample cascade.  Not covered: non-finite source samples — the reference's Biquad turns a
// NaN sample into a zero (biquad_filter.rs:881-883) before the convolver sees it, the folded form convolves it.
bool conv_fold_biquad_into_ir(const waa_batch* b, Node& conv, const Node& q) {
  conv.ir_lti.clear();
  conv.ir_lti_len = 0;
  // (a RUNTIME switch of the product library, not a measurement switch — ADVICE round 5: a caller whose sources may hold NaN / Inf
  // samples and who needs the reference's recovery from them sets WAA_NO_CONV_BIQUAD_IR_FOLD=1 and gets the exact-order filter
  // stage of conv_fft3_fwd_bq_kernel back; DESIGN.md section 5.9, INTEGRATION.md "runtime switches")
  if (getenv("WAA_NO_CONV_BIQUAD_IR_FOLD") || !conv.has_ir || q.params.size() < 4) return false;  // (read per plan: tests toggle it)
  float pv[4];
  for (int k = 0; k < 4; k++) {
    const ParamStore& ps = q.params[k];
    if (ps.mode() != 0 || !ps.blocks.empty() || !ps.timelines.empty() || ps.dev_tl || ps.cst.empty()) return false;
    for (uint32_t i = 1; i < b->n_inst; i++)
      if (ps.cst[i] != ps.cst[0]) return false;  // (per-context coefficients: per-context responses — the kernel form keeps those)
    pv[k] = ps.fix(ps.cst[0]);
  }
  const Coefs c = biquad_coefs(q.desc.i[0], (double)b->sr, (double)computed_freq(pv[WAA_PARAM_BIQUAD_FREQUENCY], pv[WAA_PARAM_BIQUAD_DETUNE]),
                               (double)pv[WAA_PARAM_BIQUAD_GAIN], (double)pv[WAA_PARAM_BIQUAD_Q]);
  for (double v : {c.b0, c.b1, c.b2, c.a1, c.a2})
    if (!std::isfinite(v)) return false;
  // spectral radius of z^2 + a1 z + a2
  double rho;
  {
    const double disc = c.a1 * c.a1 - 4. * c.a2;
    rho = disc < 0. ? std::sqrt(std::fabs(c.a2)) : std::max(std::fabs(-c.a1 + std::sqrt(disc)), std::fabs(-c.a1 - std::sqrt(disc))) * 0.5;
  }
  if (!(rho < 1.)) return false;
  const int nch = conv.ir_nch;
  uint64_t lmax = 0;
  std::vector<uint64_t> trim(nch);
  for (int ch = 0; ch < nch; ch++) {
    uint64_t l = conv.ir_len;
    while (l > 0 && std::fabs(conv.ir[ch][l - 1]) < 0.000001f) l--;
    trim[ch] = l;
    lmax = std::max(lmax, l);
  }
  if (lmax == 0) return false;
  const uint64_t EXT = 2 * 8192, T = lmax + EXT;
  std::vector<std::vector<double>> y(nch, std::vector<double>(T));
  uint64_t len2 = 0;
  for (int ch = 0; ch < nch; ch++) {
    double x1 = 0., x2 = 0., y1 = 0., y2 = 0., l1 = 0.;
    for (uint64_t i = 0; i < T; i++) {
      const double x = i < trim[ch] ? (double)conv.ir[ch][i] : 0.;
      const double v = (c.b0 * x + c.b1 * x1 + c.b2 * x2) - c.a1 * y1 - c.a2 * y2;  // biquad_filter.rs:877
      x2 = x1;
      x1 = x;
      y2 = y1;
      y1 = v;
      y[ch][i] = v;
      l1 += std::fabs(v);
    }
    if (!std::isfinite(l1)) return false;
    if (l1 == 0.) continue;  // (a silent channel stays silent)
    // what lies beyond the simulated extension: |state| / (1 - rho) bounds its L1 norm up to a small constant
    if ((std::fabs(y1) + std::fabs(y2)) / (1. - rho) > 1e-13 * l1) return false;
    double tail = 0.;
    uint64_t n = T;
    while (n > 0 && tail + std::fabs(y[ch][n - 1]) <= 1e-12 * l1) tail += std::fabs(y[ch][--n]);
    if (n + 1024 > T) return false;  // not converged inside the extension
    len2 = std::max(len2, n);
  }
  if (len2 == 0) return false;
  // the block size must stay the one the three-pass transforms serve, and the partitions within the product kernel's range
  if ((len2 + 2047) / 2048 <= 24 || (len2 + 8191) / 8192 > 24) return false;
  conv.ir_lti.assign(nch, std::vector<float>(len2, 0.f));
  for (int ch = 0; ch < nch; ch++)
    for (uint64_t i = 0; i < len2; i++) conv.ir_lti[ch][i] = (float)y[ch][i];
  conv.ir_lti_len = len2;
  return true;
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
  const bool lti = n.pre_biquad >= 0 && n.ir_lti_len > 0;  // the Biquad in front lives in the impulse response
  const std::vector<std::vector<float>>& ir_use = lti ? n.ir_lti : n.ir;
  const uint64_t ir_use_len = lti ? n.ir_lti_len : n.ir_len;
  uint64_t len = 0;
  if (lti) {
    len = n.ir_lti_len;  // (already cut where the folded response ends)
  } else {
    for (int c = 0; c < ir_nch; c++) {
      uint64_t l = n.ir_len;
      while (l > 0 && std::fabs(n.ir[c][l - 1]) < 0.000001f) l--;
      len = std::max(len, l);
    }
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
    uint64_t l = ir_use_len;
    if (!lti)
      while (l > 0 && std::fabs(ir_use[c][l - 1]) < 0.000001f) l--;  // samples past a channel's own trim are dropped
    for (uint64_t i = 0; i < len; i++) irflat[(size_t)c * len + i] = i < l ? ir_use[c][i] : 0.f;
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
  if (n.pre_biquad >= 0 && !lti) {
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
            cv.pre_coefs ? " (+ the Biquad in front, in the forward transform)" : lti ? " (+ the Biquad in front, in the impulse response)" : "");
  st.slot_fwd = slot_for(b, "conv_fft_kernel<fwd>");
  st.slot_mac = slot_for(b, "conv_mac_kernel");
  st.slot_inv = slot_for(b, "conv_fft_kernel<inv>");
  b->steps.push_back(st);
  return 0;
}

}  // namespace host
}  // namespace waa
