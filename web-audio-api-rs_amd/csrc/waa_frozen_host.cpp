// waa_frozen_host.cpp — host side of the WaveShaper's 2x / 4x oversampling and of the HRTF panning model: the written
// definitions of the two third-party algorithms (rubato 0.16 FftFixedInOut, hrtf 0.8.1) turned
// into the tables the kernels of waa_frozen.hip consume, and the planning of their steps.
#include <array>
#include <mutex>

#include <thread>

#include "waa_host.hpp"
#include "waa_hrtf_fft_tables.hpp"
#include "waa_osfft_tables.hpp"

namespace waa {
namespace host {

int node_input_signal(waa_batch* b, uint32_t id, SignalRef* out_sig, const SignalRef* target = nullptr, uint64_t* valid = nullptr);

// ===============================================================================================================
// WaveShaper oversampling.  One rubato FftResampler stage (synchro.rs), fft_size_in = fi, fft_size_out = fo:
//   x[0..fi) zero-padded to 2 fi -> real FFT -> bins [0, new_len) times the filter spectrum F, the rest zero
//   (new_len = fi + 1 when up-sampling, fo when down-sampling) -> inverse real FFT of 2 fo points (unnormalised; F
//   carries the 1 / (2 fi)) -> the first fo samples plus the overlap kept from the previous call are the output, the
//   last fo samples the new overlap.
// F is the spectrum of a windowed sinc of fi taps: cutoff 0.4^(16 / fi) (times fo / fi when down-sampling), window
// BlackmanHarris squared, normalised to unit sum — evaluated in f32 like the crate does (that IS the filter), then
// everything downstream in f64.  The stage is linear and time-invariant per block:
//   out_buf[m] = sum_j x[j] * phi[(m * fm / fo - j * fm / fi) mod 2 fm],   fm = max(fi, fo),
//   phi[p] = sum_{k < new_len} c_k Re(F[k] exp(2 pi i k p / (2 fm))),  c_0 = 1, c_k = 2,
// which is what the device multiplies with: A[k][m] (k < fi) = the response of output frame m to input frame k of the
// SAME block, A[fi + k][m] = the response of frame m to input frame k of the PREVIOUS processed block (its overlap).
using osfft::rubato_filter_taps;  // (waa_osfft_tables.hpp: shared with the transform form)
static std::vector<float> resampler_matrix(int fi, int fo) {
  const std::vector<float> g = rubato_filter_taps(fi, fo);
  const int new_len = fi < fo ? fi + 1 : fo;
  const int fm = std::max(fi, fo);
  const double two_pi = 6.283185307179586476925286766559;
  std::vector<double> fre((size_t)new_len), fim((size_t)new_len);
  for (int k = 0; k < new_len; k++) {
    double re = 0., im = 0.;
    for (int n = 0; n < fi; n++) {
      const double a = -two_pi * (double)((int64_t)k * n % (2 * fi)) / (double)(2 * fi);
      re += (double)g[(size_t)n] * std::cos(a);
      im += (double)g[(size_t)n] * std::sin(a);
    }
    fre[(size_t)k] = re;
    fim[(size_t)k] = im;
  }
  // (the inverse real transform ignores the imaginary part of bin 0; F[0] is real anyway)
  std::vector<double> phi((size_t)2 * fm);
  for (int p = 0; p < 2 * fm; p++) {
    double acc = fre[0];
    for (int k = 1; k < new_len; k++) {
      const double a = two_pi * (double)((int64_t)k * p % (2 * fm)) / (double)(2 * fm);
      acc += 2. * (fre[(size_t)k] * std::cos(a) - fim[(size_t)k] * std::sin(a));
    }
    phi[(size_t)p] = acc;
  }
  const int so = fm / fo, si = fm / fi;
  std::vector<float> A((size_t)2 * fi * fo);
  for (int k = 0; k < fi; k++)
    for (int m = 0; m < 2 * fo; m++) {
      int p = (m * so - k * si) % (2 * fm);
      if (p < 0) p += 2 * fm;
      const float v = (float)phi[(size_t)p];
      if (m < fo)
        A[(size_t)k * fo + m] = v;
      else
        A[(size_t)(fi + k) * fo + (m - fo)] = v;
    }
  return A;
}

// The matrix as three bf16 planes, A = hi + mid + lo exactly (round to nearest even at each step), in the tile order
// of qgemm_bf16x6_kernel: [k / 16][plane][M][16].  A is [K][M], k-major.
static std::vector<uint16_t> split_matrix_bf16x3(const std::vector<float>& A, int K, int M) {
  auto rne = [](float x) -> uint32_t {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  };
  auto as_float = [](uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  std::vector<uint16_t> out((size_t)3 * K * M);
  for (int k = 0; k < K; k++)
    for (int m = 0; m < M; m++) {
      const float x = A[(size_t)k * M + m];
      const uint32_t hi = rne(x);
      const float r = x - as_float(hi);
      const uint32_t mid = rne(r);
      const float l = r - as_float(mid);
      uint32_t lo;
      std::memcpy(&lo, &l, 4);
      const uint32_t plane[3] = {hi, mid, lo};
      for (int p = 0; p < 3; p++) out[(((size_t)(k / 16) * 3 + p) * M + m) * 16 + (size_t)(k % 16)] = (uint16_t)(plane[p] >> 16);
    }
  return out;
}

// host-known codes (count | CODE_SILENT per quantum) of a source node: waa_plan.cpp
int source_code_rows(waa_batch* b, uint32_t id, uint64_t cs, std::vector<uint8_t>* host);

// codes of the node's mixed input and the prev table; `src_id` >= 0: static plan, the input is that source
static int plan_link(waa_batch* b, uint32_t id, int kind, int src_id, uint32_t tail_frames, int can_propagate, const uint8_t** in_code,
                     int32_t** prev_out, uint64_t* prev_stride_out) {
  PlanTrace trace("plan_link");
  *prev_stride_out = b->n_quanta;
  Node& n = b->nodes[id];
  const uint64_t cs = b->code_stride ? b->code_stride : (((uint64_t)b->n_quanta + 15) & ~(uint64_t)15);
  const uint8_t* d_in = n.in_code;
  std::vector<uint8_t> host_codes;
  if (src_id >= 0) {
    int e;
    {
      PlanTrace t2("plan_link: source_code_rows");
      e = source_code_rows(b, (uint32_t)src_id, cs, &host_codes);
    }
    if (e) return e;
    // the node's MIXED input: a silent (mono) quantum of the source is mixed to the node's computed count like any other
    // (quantum.rs:532-569) — with channelCountMode explicit that is the node's channelCount, and the quantum stays silent
    const uint8_t silent_code = (uint8_t)((uint32_t)computed_in_nch(n, 1) | CODE_SILENT);
    for (uint8_t& c : host_codes)
      if (c & CODE_SILENT) c = silent_code;
    uint8_t* up = nullptr;
    if ((e = dev_upload(b, &up, host_codes))) return e;
    d_in = up;
  }
  if (!d_in) return fail(WAA_ERR_INVALID_STATE, "internal: node %u has no input codes", id);
  int32_t* d_prev = nullptr;
  if (src_id >= 0 && !b->dynamic && !measure_switch("WAA_LINK_KERNEL")) {
    // static plan: the codes are host-known, so is the replay (the automaton of link_kernel, once per plan instead of
    // one single-thread-per-instance launch per render: 0.7 ms of a 10-15 ms render)
    // every context with the same codes (the usual batch: one schedule): ONE row of links, instance stride 0 — the table of a
    // 1024-context, 10 s batch was 15 MB built, copied and uploaded per plan (most of the 15 ms an HRTF plan took, round 5)
    bool one_row = true;
    for (uint32_t i = 1; i < b->n_inst && one_row; i++)
      one_row = std::memcmp(host_codes.data() + (size_t)i * cs, host_codes.data(), b->n_quanta) == 0;
    const uint32_t rows = one_row ? 1u : b->n_inst;
    if (one_row) *prev_stride_out = 0;
    std::vector<int32_t> hp((size_t)rows * b->n_quanta);
    for (uint32_t i = 0; i < rows; i++) {
      const uint8_t* row = host_codes.data() + (size_t)i * cs;
      if (i > 0 && std::memcmp(row, row - cs, b->n_quanta) == 0) {  // the same codes as the previous instance: the same links
        std::copy(hp.begin() + (size_t)(i - 1) * b->n_quanta, hp.begin() + (size_t)i * b->n_quanta, hp.begin() + (size_t)i * b->n_quanta);
        continue;
      }
      int32_t last = LINK_FRESH;
      int cur_ch = 1;             // kind 0: channels_x2 / channels_x4 start at 1 (waveshaper.rs:526-527)
      uint64_t tail_counter = 0;  // kind 1: only ever grows (panner.rs:697-711)
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        const uint32_t c = row[q];
        const bool silent = (c & CODE_SILENT) != 0;
        int32_t link;
        if (kind == 0) {  // WaveShaperRenderer::process, X2 / X4 (waveshaper.rs:395-400, 409-425)
          if (silent && can_propagate) {
            link = LINK_SKIP;
          } else {
            const int nch = (int)(c & 63u);  // (the mixed count, also in silent quanta)
            if (nch != cur_ch) {
              cur_ch = nch;
              last = LINK_FRESH;
            }
            link = last;
            last = (int32_t)q;
          }
        } else {  // PannerRenderer::process, HRTF (panner.rs:697-711)
          bool skip = false;
          if (silent) {
            if (!((uint64_t)tail_frames > tail_counter))
              skip = true;
            else
              tail_counter += RQ;
          }
          if (skip) {
            link = LINK_SKIP;
          } else {
            link = last;
            last = (int32_t)q;
          }
        }
        hp[(size_t)i * b->n_quanta + q] = link;
      }
    }
    int e = dev_upload(b, &d_prev, hp);
    if (e) return e;
    *in_code = d_in;
    *prev_out = d_prev;
    return 0;
  }
  int e = dev_alloc(b, &d_prev, (size_t)b->n_inst * b->n_quanta);
  if (e) return e;
  uint8_t* d_out = nullptr;
  if (b->dynamic) {
    if ((e = dev_alloc(b, &d_out, (size_t)b->n_inst * cs))) return e;
    n.code = d_out;
  }
  Step st;
  st.kind = 15;
  LinkDesc& d = st.link;
  std::memset(&d, 0, sizeof d);
  d.in_code = d_in;
  d.out_code = d_out;
  d.prev = d_prev;
  d.code_stride = cs;
  d.prev_stride = b->n_quanta;
  d.n_inst = b->n_inst;
  d.n_quanta = b->n_quanta;
  d.kind = kind;
  d.can_propagate_silence = can_propagate;
  d.tail_frames = tail_frames;
  st.profile_slot = slot_for(b, "link_kernel");
  b->steps.push_back(st);
  *in_code = d_in;
  *prev_out = d_prev;
  return 0;
}

int plan_oversampler(waa_batch* b, uint32_t id, int src_id) {
  Node& n = b->nodes[id];
  const int R = n.desc.i[0] == WAA_OVERSAMPLE_X2 ? 2 : 4;
  SignalRef in_sig{};
  if (b->dynamic && n.hist.base) {
    in_sig = n.hist;  // dynamic plans: the mixed input was published by the group in front (waa_dyn.hip)
  } else {
    uint64_t valid = 0;  // (a source read in place: whole quanta only, and quanta past its end are skipped, never read)
    int e = node_input_signal(b, id, &in_sig, nullptr, &valid);
    if (e) return e;
  }
  const size_t cn = n.curve.size();
  const float mid = cn == 0 ? 0.f : (cn % 2 ? n.curve[cn / 2] : (n.curve[cn / 2 - 1] + n.curve[cn / 2]) / 2.f);
  const int can_propagate = (cn == 0 || std::fabs(mid) < 1e-9f) ? 1 : 0;  // waveshaper.rs:498-509
  const uint8_t* in_code = nullptr;
  int32_t* prev = nullptr;
  uint64_t prev_stride = 0;
  int e = plan_link(b, id, 0, src_id, 0, can_propagate, &in_code, &prev, &prev_stride);
  if (e) return e;
  if (!n.d_curve && (e = dev_upload(b, &n.d_curve, n.curve))) return e;
  const int up_len = RQ * R;
  const int nch = n.in_nch;
  const bool matrix_form = measure_switch("WAA_OS_MATRIX") != nullptr;  // (A/B: the round-2 dense products on the matrix cores)
  if (!matrix_form) {
    // Transform form (waa_osfft.hip): one launch per channel PAIR (a pair is one complex transform, L + iR; an odd last channel goes
    // alone), the stages as 256-point transforms, nothing at the high rate in HBM.  Every pair follows the same link table: the
    // node's state is reset for all channels when the input's count changes (waveshaper.rs:409-425).
    float *d_tab = nullptr, *d_tw = nullptr;
    float* d_trash = nullptr;
    if ((e = dev_upload(b, &d_tab, osfft::tables(R))) || (e = dev_upload(b, &d_tw, osfft::tw256())) || (e = dev_alloc(b, &d_trash, 64))) return e;
    uint32_t seg_len = 0, n_seg = 0;
    for (int c0 = 0; c0 < nch; c0 += 2) {
      Step os;
      os.kind = 20;
      OsFftDesc& f = os.osfft;
      std::memset(&f, 0, sizeof f);
      f.src = in_sig.base + (uint64_t)c0 * in_sig.ch_stride;
      f.src_inst = in_sig.inst_stride;
      f.src_ch = in_sig.ch_stride;
      f.dst = n.sig.base + (uint64_t)c0 * n.sig.ch_stride;
      f.dst_inst = n.sig.inst_stride;
      f.dst_ch = n.sig.ch_stride;
      f.prev = prev;
      f.prev_stride = prev_stride;
      f.curve = n.d_curve;
      f.curve_n = (int32_t)cn;
      f.R = R;
      f.tables = d_tab;
      f.tw256 = d_tw;
      f.trash = d_trash;
      f.nch = std::min(2, nch - c0);
      f.n_inst = b->n_inst;
      f.n_quanta = b->n_quanta;
      // runs: ~16 k groups of 16 lanes in the launch (4 waves per SIMD's worth), never shorter than 8 quanta (two more are
      // rendered in front of every run for its overlaps)
      const uint32_t want = std::max<uint32_t>(1, (16384 + b->n_inst - 1) / b->n_inst);
      uint32_t seg = std::max<uint32_t>(8, (b->n_quanta + want - 1) / want);
      if (const char* sv = measure_switch("WAA_OSFFT_SEG")) seg = std::max(1, atoi(sv));  // (tests: short runs exercise the run heads)
      f.seg_len = seg;
      f.n_seg = (b->n_quanta + seg - 1) / seg;
      seg_len = f.seg_len;
      n_seg = f.n_seg;
      os.profile_slot = slot_for(b, "osfft_kernel");
      os.loop_reads.push_back(in_sig.base);
      os.loop_writes.push_back(n.sig.base);
      b->steps.push_back(os);
    }
    plan_note(b, "waveshaper node %u: %dx oversampling as %d 256-point transforms per quantum in one launch (runs of %u quanta, %u per instance), %d channel(s)%s",
              id, R, 2 + 2 * R, seg_len, n_seg, nch, can_propagate ? ", silent input skips the block" : "");
    return 0;
  }
  float *d_up = nullptr, *d_dn = nullptr, *sbuf = nullptr;
  uint16_t *d_up16 = nullptr, *d_dn16 = nullptr;
  const std::vector<float> m_up = resampler_matrix(RQ, up_len), m_dn = resampler_matrix(up_len, RQ);
  if ((e = dev_upload(b, &d_up, m_up)) || (e = dev_upload(b, &d_dn, m_dn)) ||
      (e = dev_upload(b, &d_up16, split_matrix_bf16x3(m_up, 2 * RQ, up_len))) ||
      (e = dev_upload(b, &d_dn16, split_matrix_bf16x3(m_dn, 2 * up_len, RQ))) ||
      (e = dev_alloc(b, &sbuf, (size_t)b->n_inst * nch * b->n_quanta * up_len)))
    return e;
  Step up;
  up.kind = 16;
  QGemmDesc& g = up.qgemm;
  std::memset(&g, 0, sizeof g);
  g.A = d_up;
  g.A16 = d_up16;
  g.M = up_len;
  g.Kh = RQ;
  g.src = in_sig.base;
  g.src_inst = in_sig.inst_stride;
  g.src_ch = in_sig.ch_stride;
  g.src_q = RQ;
  g.dst = sbuf;
  g.dst_ch = (uint64_t)b->n_quanta * up_len;
  g.dst_inst = g.dst_ch * nch;
  g.dst_q = up_len;
  g.prev = prev;
  g.prev_stride = prev_stride;
  g.curve = n.d_curve;
  g.curve_n = (int32_t)cn;
  g.nch = nch;
  g.n_inst = b->n_inst;
  g.n_quanta = b->n_quanta;
  up.profile_slot = slot_for(b, "qgemm_kernel<up+curve>");
  up.loop_reads.push_back(in_sig.base);
  up.loop_writes.push_back(sbuf);
  b->steps.push_back(up);
  Step dn;
  dn.kind = 16;
  QGemmDesc& h = dn.qgemm;
  std::memset(&h, 0, sizeof h);
  h.A = d_dn;
  h.A16 = d_dn16;
  h.M = RQ;
  h.Kh = up_len;
  h.src = sbuf;
  h.src_inst = g.dst_inst;
  h.src_ch = g.dst_ch;
  h.src_q = up_len;
  h.dst = n.sig.base;
  h.dst_inst = n.sig.inst_stride;
  h.dst_ch = n.sig.ch_stride;
  h.dst_q = RQ;
  h.prev = prev;
  h.prev_stride = prev_stride;
  h.nch = nch;
  h.n_inst = b->n_inst;
  h.n_quanta = b->n_quanta;
  dn.profile_slot = slot_for(b, "qgemm_kernel<down>");
  dn.loop_reads.push_back(sbuf);
  dn.loop_writes.push_back(n.sig.base);
  b->steps.push_back(dn);
  plan_note(b, "waveshaper node %u: %dx oversampling as two matrix products over render quanta (%d x %d and %d x %d), %d channel(s)%s", id,
            R, up_len, 2 * RQ, RQ, 2 * up_len, nch, can_propagate ? ", silent input skips the block" : "");
  return 0;
}

// ===============================================================================================================
// HRTF panning: the HRIR sphere (crate hrtf 0.8.1 file format and algorithm, restated)
struct Sphere {
  uint32_t sr = 0;
  int taps = 0;
  std::vector<uint32_t> faces;  // [nf][3]
  std::vector<float> pos;       // [nv][3]
  std::vector<float> left, right;  // [nv][taps]
  int nv() const { return (int)(pos.size() / 3); }
  int nf() const { return (int)(faces.size() / 3); }
};
static std::mutex g_sphere_lock;
static std::shared_ptr<Sphere> g_sphere_file;
static std::map<uint32_t, std::shared_ptr<Sphere>> g_sphere_cache;  // per sample rate, like panner.rs:39-60

static double resample_kernel_value(double x, double fc) {
  if (std::fabs(x) >= 128.) return 0.;
  const double pi = 3.14159265358979323846;
  const double u = (x + 128.) / 256.;
  const double bh = 0.35875 - 0.48829 * std::cos(2. * pi * u) + 0.14128 * std::cos(4. * pi * u) - 0.01168 * std::cos(6. * pi * u);
  const double a = pi * x * fc;
  return bh * bh * (x == 0. ? 1. : std::sin(a) / a) * fc;
}
// HRIRs at another rate — OUR definition of the crate's one-chunk rubato SincFixedIn pass (sinc_len 256, f_cutoff 0.95,
// BlackmanHarris2): output n is the band-limited signal at input position t_n = (n + 1) / ratio - 128, while
// t_n < taps - 257 - 1 / ratio; kernel evaluated directly instead of through the oversampled cubic table.
// The kernel weights depend on the output index and the tap offset only — one table for all 2 x 187 impulse responses
// (evaluating the window per product made the first plan of a process take 0.7 s).
struct HrirResampler {
  int len = 0, n_out = 0;
  std::vector<int> m0, m1;
  std::vector<double> w;  // [n_out][257]
  HrirResampler(int len_, double ratio) : len(len_) {
    while ((double)(n_out + 1) / ratio - 128. < (double)len - 257. - 1. / ratio) n_out++;
    const double fc = 0.95 * (ratio < 1. ? ratio : 1.);
    m0.resize((size_t)n_out);
    m1.resize((size_t)n_out);
    w.assign((size_t)n_out * 257, 0.);
    // (140 000 kernel values, four trigonometric calls each: 8 ms on one core — rows are independent)
    const int nt = std::max(1, std::min(8, std::min((int)std::thread::hardware_concurrency(), n_out / 16)));
    auto rows = [&](int t0) {
      for (int n = t0; n < n_out; n += nt) {
        const double t = (double)(n + 1) / ratio - 128.;
        m0[(size_t)n] = std::max((int)std::ceil(t - 128.), 0);
        m1[(size_t)n] = std::min((int)std::floor(t + 128.), len - 1);
        for (int m = m0[(size_t)n]; m <= m1[(size_t)n]; m++) w[(size_t)n * 257 + (size_t)(m - m0[(size_t)n])] = resample_kernel_value(t - (double)m, fc);
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(rows, t);
    rows(0);
    for (auto& th : pool) th.join();
  }
  void run_into(const float* in, float* out) const {
    for (int n = 0; n < n_out; n++) {
      double acc = 0.;
      const double* wn = &w[(size_t)n * 257];
      for (int m = m0[(size_t)n]; m <= m1[(size_t)n]; m++) acc += (double)in[m] * wn[m - m0[(size_t)n]];
      out[n] = (float)acc;
    }
  }
  void run(const float* in, std::vector<float>* out) const {
    const size_t at = out->size();
    out->resize(at + (size_t)n_out);
    run_into(in, out->data() + at);
  }
};
static std::shared_ptr<Sphere> sphere_for_rate(uint32_t sample_rate) {
  if (sample_rate < 27000) sample_rate = 27000;  // panner.rs:46-49
  std::lock_guard<std::mutex> lock(g_sphere_lock);
  if (!g_sphere_file) return nullptr;
  if (g_sphere_file->sr == sample_rate) return g_sphere_file;
  auto it = g_sphere_cache.find(sample_rate);
  if (it != g_sphere_cache.end()) return it->second;
  const Sphere& f = *g_sphere_file;
  auto s = std::make_shared<Sphere>();
  s->sr = sample_rate;
  s->faces = f.faces;
  s->pos = f.pos;
  const double ratio = (double)sample_rate / (double)f.sr;
  const HrirResampler rs(f.taps, ratio);
  // 2 x 187 impulse responses x ~550 outputs x 257 taps: 20 ms on one core, and the first plan of a process at a rate other
  // than the file's pays it — the responses are independent, a few threads share them (each output value is computed by
  // exactly the same sequence of operations as before)
  const int nv = f.nv();
  s->left.assign((size_t)nv * rs.n_out, 0.f);
  s->right.assign((size_t)nv * rs.n_out, 0.f);
  {
    const int nt = std::max(1, std::min(8, std::min((int)std::thread::hardware_concurrency(), nv / 8)));
    std::vector<std::thread> pool;
    auto work = [&](int t) {
      for (int v = t; v < nv; v += nt) {
        rs.run_into(f.left.data() + (size_t)v * f.taps, s->left.data() + (size_t)v * rs.n_out);
        rs.run_into(f.right.data() + (size_t)v * f.taps, s->right.data() + (size_t)v * rs.n_out);
      }
    };
    for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  s->taps = nv ? rs.n_out : 0;
  g_sphere_cache[sample_rate] = s;
  return s;
}
// HrirSphere::sample_bilinear: the face the ray origin -> dir pierces, barycentric weights of the piercing point.  Of
// all faces whose plane lies in front, the one in which the point sits deepest (largest smallest weight) is the face
// that contains it; on an edge either neighbour interpolates to the same HRIR.  f32, like the crate.
static void sphere_locate(const Sphere& s, const float dir[3], int vtx[3], float wgt[3]) {
  float best = -1e30f;
  vtx[0] = vtx[1] = vtx[2] = 0;
  wgt[0] = 1.f;
  wgt[1] = wgt[2] = 0.f;
  auto dot = [](const float* a, const float* c) { return a[0] * c[0] + a[1] * c[1] + a[2] * c[2]; };
  for (int f = 0; f < s.nf(); f++) {
    const float* a = &s.pos[3 * (size_t)s.faces[3 * (size_t)f]];
    const float* bb = &s.pos[3 * (size_t)s.faces[3 * (size_t)f + 1]];
    const float* c = &s.pos[3 * (size_t)s.faces[3 * (size_t)f + 2]];
    const float ba[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]}, ca[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const float nrm[3] = {ba[1] * ca[2] - ba[2] * ca[1], ba[2] * ca[0] - ba[0] * ca[2], ba[0] * ca[1] - ba[1] * ca[0]};
    const float denom = dot(dir, nrm), num = dot(a, nrm);
    if (denom == 0.f) continue;
    const float t = num / denom;
    if (!(t > 0.f)) continue;
    const float pnt[3] = {dir[0] * t, dir[1] * t, dir[2] * t};
    const float v2[3] = {pnt[0] - a[0], pnt[1] - a[1], pnt[2] - a[2]};
    const float d00 = dot(ba, ba), d01 = dot(ba, ca), d11 = dot(ca, ca), d20 = dot(v2, ba), d21 = dot(v2, ca);
    const float den = d00 * d11 - d01 * d01;
    const float v = (d11 * d20 - d01 * d21) / den;
    const float w = (d00 * d21 - d01 * d20) / den;
    const float u = 1.0f - v - w;
    const float m = std::min(u, std::min(v, w));
    if (m > best) {
      best = m;
      for (int k = 0; k < 3; k++) vtx[k] = (int)s.faces[3 * (size_t)f + k];
      wgt[0] = u;
      wgt[1] = v;
      wgt[2] = w;
    }
  }
}

int plan_hrtf(waa_batch* b, uint32_t id, int src_id) {
  PlanTrace trace("plan_hrtf");
  Node& n = b->nodes[id];
  std::shared_ptr<Sphere> sp = sphere_for_rate((uint32_t)b->sr);
  if (!sp) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - HRTF panning needs the HRIR sphere (waa_hrtf_load_sphere)");
  if (sp->taps > HRTF_MAX_TAPS || sp->taps < 1)
    return fail(WAA_ERR_OUT_OF_SCOPE, "HRTF panning at %g Hz needs %d-tap impulse responses; the device kernel holds at most %d", (double)b->sr,
                sp->taps, HRTF_MAX_TAPS);
  SignalRef in_sig{};
  if (b->dynamic && n.hist.base) {
    in_sig = n.hist;
  } else {
    uint64_t valid = 0;  // (a source read in place: silent quanta are recognised by their code, never read)
    int e = node_input_signal(b, id, &in_sig, nullptr, &valid);
    if (e) return e;
  }
  const uint8_t* in_code = nullptr;
  int32_t* prev = nullptr;
  uint64_t prev_stride = 0;
  int e = plan_link(b, id, 1, src_id, (uint32_t)sp->taps, 0, &in_code, &prev, &prev_stride);
  if (e) return e;
  // geometry per (instance, quantum) on the host, always k-rate: the first value of every param (panner.rs:781-799)
  PlanTrace t3("plan_hrtf: geometry + tables");
  bool shared = true, varies = false;
  for (int k = 0; k < 15; k++) {
    const ParamStore& ps = n.params[k];
    for (uint32_t i = 1; i < b->n_inst && shared; i++) shared = ps.cst[i] == ps.cst[0];
    for (auto& blk : ps.blocks) shared &= blk.inst == WAA_ALL_INSTANCES;
    varies |= !ps.blocks.empty();
    if (ps.dev_tl) return fail(WAA_ERR_INVALID_STATE, "internal: device-side automation on an HRTF panner param");
  }
  const uint32_t rows = shared ? 1u : b->n_inst, per_row = varies ? b->n_quanta : 1u;
  std::vector<HrtfQ> table((size_t)rows * per_row);
  std::map<std::tuple<float, float, float>, std::pair<std::array<int, 3>, std::array<float, 3>>> located;
  for (uint32_t i = 0; i < rows; i++) {
    std::vector<std::vector<float>> pv(15);
    for (int k = 0; k < 15; k++) pv[k] = param_per_quantum(b, n.params[k], i, nullptr);
    for (uint32_t q = 0; q < per_row; q++) {
      auto at = [&](int p) { return pv[p][pv[p].size() == 1 ? 0 : q]; };
      V3 spos{at(0), at(1), at(2)}, so{at(3), at(4), at(5)}, lp{at(6), at(7), at(8)}, lf{at(9), at(10), at(11)}, lu{at(12), at(13), at(14)};
      float az, el;
      azimuth_elevation(spos, lp, lf, lu, &az, &el);
      const float az_rad = az * PI_F / 180.f, el_rad = el * PI_F / 180.f;
      float ps[3] = {sinf(az_rad) * cosf(el_rad), sinf(el_rad), cosf(az_rad) * cosf(el_rad)};
      if (std::fabs(ps[0]) <= 1e-6f && std::fabs(ps[1]) <= 1e-6f && std::fabs(ps[2]) <= 1e-6f) {
        ps[0] = ps[1] = 0.f;
        ps[2] = 1.f;
      }
      const float dir[3] = {ps[0], ps[2], ps[1]};  // Vec3 { x: p[0], z: p[1], y: p[2] } (panner.rs:246-250)
      HrtfQ& r = table[(size_t)i * per_row + q];
      auto key = std::make_tuple(dir[0], dir[1], dir[2]);
      auto it = located.find(key);
      if (it == located.end()) {
        int vtx[3];
        float wgt[3];
        sphere_locate(*sp, dir, vtx, wgt);
        it = located.emplace(key, std::make_pair(std::array<int, 3>{vtx[0], vtx[1], vtx[2]}, std::array<float, 3>{wgt[0], wgt[1], wgt[2]})).first;
      }
      for (int k = 0; k < 3; k++) {
        r.v[k] = it->second.first[(size_t)k];
        r.w[k] = it->second.second[(size_t)k];
      }
      r.gain = cone_gain(n.desc, spos, so, lp) * dist_gain(n.desc, spos, lp);
      r.pad = 0;
    }
  }
  std::vector<float> hr((size_t)sp->nv() * 2 * sp->taps);
  for (int v = 0; v < sp->nv(); v++) {
    std::copy(sp->left.begin() + (size_t)v * sp->taps, sp->left.begin() + (size_t)(v + 1) * sp->taps, hr.begin() + (size_t)v * 2 * sp->taps);
    std::copy(sp->right.begin() + (size_t)v * sp->taps, sp->right.begin() + (size_t)(v + 1) * sp->taps,
              hr.begin() + (size_t)v * 2 * sp->taps + sp->taps);
  }
  float* d_hr = nullptr;
  HrtfQ* d_table = nullptr;
  if ((e = dev_upload(b, &d_hr, hr)) || (e = dev_upload(b, &d_table, table))) return e;
  // one direction for the whole batch: the HRIR pair once, in the device kernel's arithmetic (a * u + b * v + c * w, f32)
  float* d_hstatic = nullptr;
  std::vector<float> hs_host;
  if (per_row == 1) {  // (one direction per instance, or one for the batch)
    const int O = (sp->taps + 3) & ~3;
    std::vector<float>& hs = hs_host;
    hs.assign((size_t)rows * 2 * O, 0.f);
    for (uint32_t i = 0; i < rows; i++) {
      const HrtfQ& r = table[i];
      for (int ear = 0; ear < 2; ear++)
        for (int t = 0; t < sp->taps; t++) {
          const size_t o = (size_t)ear * sp->taps + t;
          const float a = hr[(size_t)r.v[0] * 2 * sp->taps + o], bb = hr[(size_t)r.v[1] * 2 * sp->taps + o], c = hr[(size_t)r.v[2] * 2 * sp->taps + o];
          volatile float p0 = a * r.w[0], p1 = bb * r.w[1], p2 = c * r.w[2];  // (no contraction: the device code is built with -ffp-contract=off)
          volatile float s01 = p0 + p1;
          hs[((size_t)i * O + t) * 2 + ear] = s01 + p2;
        }
    }
    if ((e = dev_upload(b, &d_hstatic, hs))) return e;
  }
  Step st;
  st.kind = 17;
  HrtfDesc& d = st.hrtf;
  std::memset(&d, 0, sizeof d);
  d.in = in_sig;
  d.out = n.sig;
  d.in_code = in_code;
  d.code_stride = b->code_stride ? b->code_stride : (((uint64_t)b->n_quanta + 15) & ~(uint64_t)15);
  d.prev = prev;
  d.prev_stride = prev_stride;
  d.hrir = d_hr;
  d.hstatic = d_hstatic;
  d.table = d_table;
  d.rows = rows;
  d.per_row = per_row;
  d.taps = sp->taps;
  d.n_inst = b->n_inst;
  d.n_quanta = b->n_quanta;
  bool fft_form = false;
  // Where the direct form puts out EXACT zeros (frames more than `taps` behind the last non-zero input frame: the end of the node's
  // tail) the transforms leave roundoff of 1e-10, and in a dynamic plan a node behind the panner may decide on exactly that (a
  // DelayNode that read nothing but zeros, a BiquadFilterNode whose state is no longer normal: biquad_filter.rs:775-790) — fuzz seeds
  // 502 and 630 of the frozen-state generator, DESIGN.md section 5.  When anything but GainNodes and the destination hears the panner
  // of a dynamic plan, the kernel runs its exact-zeros form (HrtfDesc::jmax: 0 wherever no non-zero input frame lies within the
  // response's reach, like the direct sum).
  auto plain_listeners_only = [&]() {
    std::vector<uint32_t> todo{id};
    std::vector<char> seen(b->nodes.size(), 0);
    while (!todo.empty()) {
      const uint32_t u = todo.back();
      todo.pop_back();
      for (const waa_edge_desc& ed : b->edges) {
        if (ed.from != u) continue;
        if (ed.to_input >= WAA_PARAM_INPUT(0)) return false;  // (into an AudioParam)
        const int k = b->nodes[ed.to].desc.kind;
        if (k != WAA_NODE_DESTINATION && k != WAA_NODE_GAIN) return false;
        if (!seen[ed.to]) seen[ed.to] = 1, todo.push_back(ed.to);
      }
    }
    return true;
  };
  if (per_row == 1 && sp->taps <= 128 * hrtffft::PARTS && !b->dry) {
    // PannerNode and AudioListener at rest, the same for every context: the HRIR pair's partition spectra (both ears in one
    // complex table), the transform form of waa_hrtf_fft.hip.  Runs as for the oversampled WaveShaper: ~16 k groups per launch.
    const int O = (sp->taps + 3) & ~3;
    float *d_tab = nullptr, *d_tw = nullptr, *d_trash = nullptr;
    if (rows == 1) {
      // (the pair plan_hrtf built above for hrtf8_kernel, in the device kernel's f32 arithmetic: hs = [tap][ear])
      if ((e = dev_upload(b, &d_tab, hrtffft::make_tables(hs_host.data(), sp->taps)))) return e;
    } else {
      // one direction per context: the tables of all rows on the device, from the pairs already uploaded (same sums, same order)
      std::vector<double> cs_sn(512);
      for (int j = 0; j < 256; j++) {
        cs_sn[(size_t)j] = std::cos(-6.283185307179586476925286766559 * (double)j / 256.);
        cs_sn[(size_t)256 + j] = std::sin(-6.283185307179586476925286766559 * (double)j / 256.);
      }
      double* d_cs = nullptr;
      if ((e = dev_upload(b, &d_cs, cs_sn)) || (e = dev_alloc(b, &d_tab, (size_t)rows * hrtffft::PARTS * osfft::TAB_SLOTS * 2))) return e;
      launch_hrtf_fft_tables(d_hstatic, (uint32_t)(2 * O), sp->taps, rows, d_cs, d_tab, b->stream);
      HIP_TRY(hipGetLastError());
    }
    if ((e = dev_upload(b, &d_tw, osfft::tw256())) || (e = dev_alloc(b, &d_trash, 64))) return e;
    if (b->dynamic && !plain_listeners_only()) {
      // a node behind the panner may decide on exact zeros: the kernel's exact-zeros form needs the last non-zero tap per ear and row
      std::vector<int32_t> jm((size_t)rows * 2, 0);
      for (uint32_t i = 0; i < rows; i++)
        for (int ear = 0; ear < 2; ear++)
          for (int t = sp->taps - 1; t >= 0; t--)
            if (hs_host[((size_t)i * O + t) * 2 + ear] != 0.f) {
              jm[(size_t)i * 2 + ear] = t;
              break;
            }
      int32_t* d_jm = nullptr;
      if ((e = dev_upload(b, &d_jm, jm))) return e;
      d.jmax = d_jm;
    }
    d.fft_tables = d_tab;
    d.tw256 = d_tw;
    d.trash = d_trash;
    // runs of ~16 k groups per launch; with one table per context the runs of a context come in sixteens (one workgroup = one table)
    uint32_t seg = std::max<uint32_t>(8, (uint32_t)(((uint64_t)b->n_inst * b->n_quanta + 16383) / 16384));
    if (rows > 1) {
      const uint32_t sixteens = std::max<uint32_t>(1, (uint32_t)((16384 + 8 * (uint64_t)b->n_inst) / (16 * (uint64_t)b->n_inst)));
      seg = std::max<uint32_t>(8, (b->n_quanta + 16 * sixteens - 1) / (16 * sixteens));
    }
    seg = std::min(seg, b->n_quanta);
    d.seg_len = seg;
    d.n_seg = (b->n_quanta + seg - 1) / seg;
    fft_form = true;
  }
  st.profile_slot = slot_for(b, "hrtf_kernel");
  st.loop_reads.push_back(in_sig.base);
  st.loop_writes.push_back(n.sig.base);
  b->steps.push_back(st);
  if (fft_form)
    plan_note(b, "panner node %u: HRTF, %d-tap impulse responses at %u Hz, %s: %d partitions of 128 taps as 256-point "
                 "transforms (2 per quantum, runs of %u quanta, %u per instance)", id, sp->taps, sp->sr,
              rows == 1 ? (d.jmax ? "one direction for the whole batch (exact zeros behind the response's reach)" : "one direction for the whole batch")
                        : (d.jmax ? "one direction per context (exact zeros behind the response's reach)" : "one direction per context"),
              hrtffft::PARTS, d.seg_len, d.n_seg);
  else
    plan_note(b, "panner node %u: HRTF, %d-tap impulse responses at %u Hz, geometry table %u row(s) x %u, direct FIR per render quantum", id,
              sp->taps, sp->sr, rows, per_row);
  return 0;
}

}  // namespace host
}  // namespace waa

// ---- C ABI (include/waa_hip.h) ---------------------------------------------------------------------------------
using namespace waa::host;

extern "C" waa_status waa_hrtf_load_sphere(const void* data, uint64_t size) {
  const unsigned char* d = static_cast<const unsigned char*>(data);
  if (!d || size < 20 || std::memcmp(d, "HRIR", 4) != 0) return fail(WAA_ERR_INVALID_ARGUMENT, "HRIR sphere: bad magic");
  uint32_t hdr[4];
  std::memcpy(hdr, d + 4, 16);
  const uint64_t len = hdr[1], nv = hdr[2], ni = hdr[3];
  // (a caller-supplied header: every field is bounded before it sizes anything — no vertex / face counts of zero, which
  // sphere_locate would index, no length beyond what the FIR kernels hold, and the three u32 fields cannot wrap the u64 size)
  if (len == 0 || len > (uint64_t)waa::HRTF_MAX_TAPS || nv == 0 || nv > (1u << 20) || ni == 0 || ni % 3 != 0 || ni > (1u << 24) ||
      hdr[0] == 0)
    return fail(WAA_ERR_INVALID_ARGUMENT, "HRIR sphere: header out of range (rate %u, %llu taps, %llu vertices, %llu indices)", hdr[0],
                (unsigned long long)len, (unsigned long long)nv, (unsigned long long)ni);
  if (size != 20 + 4 * ni + nv * (12 + 8 * len)) return fail(WAA_ERR_INVALID_ARGUMENT, "HRIR sphere: inconsistent sizes");
  auto s = std::make_shared<Sphere>();
  s->sr = hdr[0];
  s->taps = (int)len;
  s->faces.resize(ni);
  std::memcpy(s->faces.data(), d + 20, 4 * ni);
  for (uint32_t f : s->faces)
    if (f >= nv) return fail(WAA_ERR_INVALID_ARGUMENT, "HRIR sphere: face index out of range");
  s->pos.resize(3 * nv);
  s->left.resize(nv * len);
  s->right.resize(nv * len);
  const unsigned char* p = d + 20 + 4 * ni;
  for (uint64_t v = 0; v < nv; v++) {
    std::memcpy(&s->pos[3 * v], p, 12);
    std::memcpy(&s->left[v * len], p + 12, 4 * len);
    std::memcpy(&s->right[v * len], p + 12 + 4 * len, 4 * len);
    p += 12 + 8 * len;
  }
  std::lock_guard<std::mutex> lock(g_sphere_lock);
  g_sphere_file = s;
  g_sphere_cache.clear();
  return WAA_OK;
}

extern "C" uint32_t waa_hrtf_hrir_length(float sample_rate) {
  auto s = sphere_for_rate((uint32_t)sample_rate);
  return s ? (uint32_t)s->taps : 0u;
}

extern "C" void waa_hrtf_sample(float sample_rate, const float* dir, float* left, float* right) {
  auto s = sphere_for_rate((uint32_t)sample_rate);
  if (!s) return;
  int vtx[3];
  float wgt[3];
  sphere_locate(*s, dir, vtx, wgt);
  for (int i = 0; i < s->taps; i++) {
    left[i] = s->left[(size_t)vtx[0] * s->taps + i] * wgt[0] + s->left[(size_t)vtx[1] * s->taps + i] * wgt[1] +
              s->left[(size_t)vtx[2] * s->taps + i] * wgt[2];
    right[i] = s->right[(size_t)vtx[0] * s->taps + i] * wgt[0] + s->right[(size_t)vtx[1] * s->taps + i] * wgt[1] +
               s->right[(size_t)vtx[2] * s->taps + i] * wgt[2];
  }
}

namespace waa {
namespace host {
bool hrtf_sphere_loaded() {
  std::lock_guard<std::mutex> lock(g_sphere_lock);
  return (bool)g_sphere_file;
}
}  // namespace host
}  // namespace waa
