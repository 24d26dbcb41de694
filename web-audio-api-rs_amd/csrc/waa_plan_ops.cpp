// waa_plan_ops.cpp — a node as chain ops (emit_node_ops) and the per-kind channel configuration defaults (split out of
// waa_plan.cpp in round 4).
#include <array>
#include <set>

#include "waa_host.hpp"
#include "waa_plan_parts.hpp"

namespace waa {
namespace host {

// Emit the fused ops of node `id` given the running channel count.
int emit_node_ops(waa_batch* b, uint32_t id, int cur_nch, bool head, std::vector<OpDesc>& ops, int* out_nch) {
  Node& n = b->nodes[id];
  const uint32_t kind = n.desc.kind;
  if (kind == WAA_NODE_BUFFER_SOURCE || kind == WAA_NODE_CONSTANT_SOURCE) {
    *out_nch = n.out_nch;
    return 0;
  }
  // input mixing to the node's computed channel count (quantum.rs:532-569); the chain head's inputs are
  // mixed by the input stage already
  if (!head && cur_nch != n.in_nch) {
    OpDesc m{};
    m.kind = OP_MIX;
    m.nch_in = cur_nch;
    m.nch_out = n.in_nch;
    m.i0 = n.interp;
    ops.push_back(m);
  }
  const int nch = n.in_nch;
  *out_nch = n.out_nch;
  switch (kind) {
    case WAA_NODE_GAIN: {
      OpDesc o{};
      o.kind = OP_GAIN;
      o.nch_in = o.nch_out = nch;
      int e = node_param(b, id, 0, &o.p0);
      if (e) return e;
      ops.push_back(o);
      break;
    }
    case WAA_NODE_BIQUAD: {
      OpDesc o{};
      o.kind = OP_BIQUAD;
      o.nch_in = o.nch_out = nch;
      bool varies = false, a_rate = false;
      for (size_t k = 0; k < n.params.size(); k++) {
        if (param_mode(n, k) == 2) a_rate = true;
        if (param_mode(n, k) == 1) varies = true;
      }
      if (a_rate) {
        // a-rate params: coefficients per frame (biquad_filter.rs:837-855), computed on the device in f64 from
        // the per-frame param values into a table the chain kernel streams
        Step cs;
        cs.kind = 5;
        BiquadCoefDesc& cdsc = cs.coef;
        std::memset(&cdsc, 0, sizeof cdsc);
        int e;
        if ((e = node_param(b, id, WAA_PARAM_BIQUAD_FREQUENCY, &cdsc.frequency)) ||
            (e = node_param(b, id, WAA_PARAM_BIQUAD_DETUNE, &cdsc.detune)) ||
            (e = node_param(b, id, WAA_PARAM_BIQUAD_Q, &cdsc.q)) ||
            (e = node_param(b, id, WAA_PARAM_BIQUAD_GAIN, &cdsc.gain)))
          return e;
        cdsc.n_frames = (uint64_t)b->n_quanta * RQ;
        cdsc.frames_padded = b->lp;
        // one table for all instances when the four params do not depend on the instance (the usual automation:
        // the same timeline scheduled on every context): 40 B per frame instead of 40 B per frame-instance
        bool shared = true;
        for (size_t k = 0; k < 4; k++) {
          const bool modulated = k < n.pin_edges.size() && !n.pin_edges[k].empty();
          const ParamStore& ps = n.params[k];
          if (modulated || ps.dev_tl) shared = false;
          for (uint32_t i = 1; i < b->n_inst && shared; i++) shared = ps.cst[i] == ps.cst[0];
          for (auto& blk : ps.blocks) shared &= blk.inst == WAA_ALL_INSTANCES;
        }
        cdsc.rows = shared ? 1u : b->n_inst;
        cdsc.type = n.desc.i[0];
        cdsc.sample_rate = b->sr;
        double* dco = nullptr;
        if ((e = dev_alloc(b, &dco, (size_t)cdsc.rows * cdsc.frames_padded * 5))) return e;
        cdsc.coefs = dco;
        cs.profile_slot = slot_for(b, "biquad_coef_kernel");
        b->steps.push_back(cs);
        double* dst = nullptr;
        if ((e = dev_alloc(b, &dst, (size_t)b->n_inst * STATE_STRIDE))) return e;
        b->state_bufs.push_back({dst, (size_t)b->n_inst * STATE_STRIDE * sizeof(double)});
        o.i0 = 2;
        o.i1 = (int32_t)b->steps.size() - 1;  // the coefficient step: emit_segments may switch it to the lane-major layout
        {
          Step hs;  // placeholder for the digest of a shared table (a no-op unless emit_segments fills it in)
          hs.kind = 12;
          std::memset(&hs.hp, 0, sizeof hs.hp);
          hs.profile_slot = slot_for(b, "biquad_hp_kernel");
          b->steps.push_back(hs);
        }
        o.ptr0 = dco;
        o.ptr1 = dst;
        o.u0 = shared ? 0 : cdsc.frames_padded * 5;
        ops.push_back(o);
        break;
      }
      const uint64_t per = varies ? (uint64_t)b->n_quanta * 5 : 5;
      std::vector<double> co((size_t)b->n_inst * per);
      std::vector<float> pf, pd, pq, pg;  // the previous instance's values: the same values give the same coefficient row
      for (uint32_t i = 0; i < b->n_inst; i++) {
        auto f = param_per_quantum(b, n.params[WAA_PARAM_BIQUAD_FREQUENCY], i, nullptr);
        auto d = param_per_quantum(b, n.params[WAA_PARAM_BIQUAD_DETUNE], i, nullptr);
        auto q = param_per_quantum(b, n.params[WAA_PARAM_BIQUAD_Q], i, nullptr);
        auto g = param_per_quantum(b, n.params[WAA_PARAM_BIQUAD_GAIN], i, nullptr);
        if (i > 0 && f == pf && d == pd && q == pq && g == pg) {
          // (one k-rate sweep for all 1024 contexts: 3.8 M coefficient sets — sin, cos, pow each — were 0.7 s of the plan)
          std::copy(co.begin() + (size_t)(i - 1) * per, co.begin() + (size_t)i * per, co.begin() + (size_t)i * per);
          continue;
        }
        pf = f;
        pd = d;
        pq = q;
        pg = g;
        const uint32_t cnt = varies ? b->n_quanta : 1;
        for (uint32_t k = 0; k < cnt; k++) {
          auto at = [&](const std::vector<float>& v) { return v[v.size() == 1 ? 0 : k]; };
          Coefs c = biquad_coefs(n.desc.i[0], (double)b->sr, (double)computed_freq(at(f), at(d)), (double)at(g), (double)at(q));
          double* dst = &co[(size_t)i * per + (size_t)k * 5];
          dst[0] = c.b0;
          dst[1] = c.b1;
          dst[2] = c.b2;
          dst[3] = c.a1;
          dst[4] = c.a2;
        }
      }
      double* dco = nullptr;
      int e = dev_upload(b, &dco, co);
      if (e) return e;
      double* dst = nullptr;
      e = dev_alloc(b, &dst, (size_t)b->n_inst * STATE_STRIDE);
      if (e) return e;
      b->state_bufs.push_back({dst, (size_t)b->n_inst * STATE_STRIDE * sizeof(double)});
      o.i0 = varies ? 1 : 0;
      o.ptr0 = dco;
      o.ptr1 = dst;
      o.u0 = per;
      ops.push_back(o);
      break;
    }
    case WAA_NODE_IIR_FILTER: {
      // iir_filter.rs:323-405.  N = len - 1 state variables, padded with zero coefficients to a kernel size.
      OpDesc o{};
      o.kind = OP_IIR;
      o.nch_in = o.nch_out = nch;
      const int len = (int)n.iir_b.size();
      const int ns = iir_padded_states(len - 1);
      if (ns < 0) return fail(WAA_ERR_DEVICE, "internal: IIR order");
      std::vector<double> co(2 * (size_t)(ns + 1), 0.);
      for (int k = 0; k < len; k++) {
        co[k] = n.iir_b[k];
        co[ns + 1 + k] = n.iir_a[k];
      }
      // zero-input state transition M: s_i' = -a_{i+1} s_0 + s_{i+1}; powers M^(32 * 2^k), k = 0..5, for the
      // lane scan of the kernel (double-double on the host, rounded once).  `growth` = largest entry of any power
      // the scan can form (intermediate squarings and all A^j, j <= 64): the scan's rounding error relative to
      // the state is about ns * growth * 2^-53, so ill-conditioned direct forms (clustered poles, high order)
      // and unstable filters go to the exact lane-per-stream kernel instead.
      // (double-double arithmetic, ~106 bits: repeated squaring of a matrix with large transient entries loses
      // growth^2 * eps per step, which long double cannot absorb for the filters that are still worth scanning)
      std::vector<DD> m((size_t)ns * ns), t((size_t)ns * ns);
      for (int i = 0; i < ns; i++) {
        m[(size_t)i * ns] = DD{-co[ns + 1 + i + 1], 0.};
        if (i + 1 < ns) m[(size_t)i * ns + i + 1] = dd_add(m[(size_t)i * ns + i + 1], DD{1., 0.});
      }
      double growth = 0.;
      auto note = [&](const std::vector<DD>& a) {
        for (const DD& v : a) growth = std::isfinite(v.hi) ? std::max(growth, std::fabs(v.hi)) : INFINITY;
      };
      auto mul = [&](const std::vector<DD>& x, const std::vector<DD>& y, std::vector<DD>& out) {
        for (int r = 0; r < ns; r++)
          for (int c = 0; c < ns; c++) {
            DD acc{0., 0.};
            for (int k = 0; k < ns; k++) acc = dd_add(acc, dd_mul(x[(size_t)r * ns + k], y[(size_t)k * ns + c]));
            out[(size_t)r * ns + c] = acc;
          }
      };
      for (int k = 0; k < 5; k++) {  // M^32
        mul(m, m, t);
        m.swap(t);
        note(m);
      }
      const std::vector<DD> A = m;
      std::vector<double> pw(6 * (size_t)ns * ns);
      for (int lvl = 0; lvl < 6; lvl++) {
        for (size_t k = 0; k < (size_t)ns * ns; k++) pw[lvl * (size_t)ns * ns + k] = m[k].hi + m[k].lo;
        mul(m, m, t);
        m.swap(t);
        note(m);
      }
      m = A;
      for (int j = 2; j <= 64 && std::isfinite(growth); j++) {  // every A^j a lane can see
        mul(m, A, t);
        m.swap(t);
        note(m);
      }
      const char* genv = measure_switch("WAA_IIR_GROWTH");  // experiments only
      const double growth_limit = genv ? atof(genv) : 1e4;
      const bool exact = !(growth <= growth_limit) || getenv("WAA_IIR_EXACT") != nullptr;  // env: debugging aid
      if (exact)
        for (auto& v : pw) v = 0.;  // unused
      double *dco = nullptr, *dpw = nullptr, *dst = nullptr;
      int e;
      if ((e = dev_upload(b, &dco, co)) || (e = dev_upload(b, &dpw, pw))) return e;
      const size_t n_state = (size_t)b->n_inst * nch * ns;
      if ((e = dev_alloc(b, &dst, n_state))) return e;
      b->state_bufs.push_back({dst, n_state * sizeof(double)});
      o.i0 = exact ? -ns : ns;
      o.ptr0 = dco;
      o.ptr1 = dst;
      o.ptr2 = dpw;
      ops.push_back(o);
      break;
    }
    case WAA_NODE_WAVESHAPER: {
      if (n.has_curve) {
        OpDesc o{};
        o.kind = OP_WAVESHAPER;
        o.nch_in = o.nch_out = nch;
        if (!n.d_curve) {
          int e = dev_upload(b, &n.d_curve, n.curve);
          if (e) return e;
        }
        o.ptr0 = n.d_curve;
        o.i0 = (int)n.curve.size();
        ops.push_back(o);
      }
      break;
    }
    case WAA_NODE_STEREO_PANNER: {
      OpDesc o{};
      o.kind = OP_STEREO_PAN;
      o.nch_in = nch;
      o.nch_out = 2;
      const ParamStore& p = n.params[0];
      int e = node_param(b, id, 0, &o.p0);
      if (e) return e;
      if (param_mode(n, 0) != 2) {
        // gains on the host with the same libm sinf the reference's f32::sin resolves to (stereo_panner.rs:74-79)
        const uint32_t cnt = p.mode() == 1 ? b->n_quanta : 1;
        std::vector<float> gl((size_t)b->n_inst * cnt), gr((size_t)b->n_inst * cnt);
        for (uint32_t i = 0; i < b->n_inst; i++) {
          auto pv = param_per_quantum(b, p, i, nullptr);
          for (uint32_t k = 0; k < cnt; k++) {
            float pan = pv[pv.size() == 1 ? 0 : k];
            float x = nch == 1 ? (pan + 1.f) * 0.5f : (pan <= 0.f ? pan + 1.f : pan);
            gl[(size_t)i * cnt + k] = sinf((1.f - x) * PI_F / 2.f);
            gr[(size_t)i * cnt + k] = sinf(x * PI_F / 2.f);
          }
        }
        if ((e = upload_values(b, gl, p.mode(), &o.p1))) return e;
        if ((e = upload_values(b, gr, p.mode(), &o.p2))) return e;
      }
      ops.push_back(o);
      break;
    }
    case WAA_NODE_PANNER: {
      OpDesc o{};
      o.kind = OP_PANNER;
      o.nch_in = nch;
      o.nch_out = 2;
      int mode = 0;
      for (auto& p : n.params) mode = std::max(mode, p.mode());
      bool listener_a_rate = false;
      for (int k = 6; k < 15; k++) listener_a_rate |= n.params[k].mode() == 2;
      if (listener_a_rate) {
        // audio-rate AudioListener automation (panner.rs:830-897, the `else` of `single_valued`): per-frame geometry on
        // the device (waa_panner.hip).  Quanta in which all nine listener params happen to be single-valued keep the
        // once-per-quantum rule (first value of every param), flagged per quantum from the value blocks.
        Step gs;
        gs.kind = 13;
        PannerGeomDesc& g = gs.geom;
        std::memset(&g, 0, sizeof g);
        bool shared = true;
        for (int k = 0; k < 15; k++) {
          const ParamStore& ps = n.params[k];
          if (ps.dev_tl) shared = false;
          for (uint32_t i = 1; i < b->n_inst && shared; i++) shared = ps.cst[i] == ps.cst[0];
          for (auto& blk : ps.blocks) shared &= blk.inst == WAA_ALL_INSTANCES;
          int e = upload_param(b, ps, &g.p[k]);
          if (e) return e;
          if (k >= 6 && ps.dev_tl) g.dev_len[k - 6] = ps.dev_lens;  // slice lengths come from the device replay
        }
        g.rows = shared ? 1u : b->n_inst;
        g.n_frames = (uint64_t)b->n_quanta * RQ;
        std::vector<uint8_t> single((size_t)g.rows * b->n_quanta, 1);
        for (uint32_t r = 0; r < g.rows; r++)
          for (int k = 6; k < 15; k++) {
            // length of the slice param k delivers in quantum q: the LAST block that covers (instance, q) decides
            std::vector<uint8_t> len128(b->n_quanta, 0);
            for (auto& blk : n.params[k].blocks) {
              if (!(blk.inst == WAA_ALL_INSTANCES || blk.inst == r)) continue;
              for (uint32_t j = 0; j < blk.nq; j++)
                if (blk.q0 + j < b->n_quanta) len128[blk.q0 + j] = blk.vpq == 1 ? 0 : 1;
            }
            for (uint32_t q = 0; q < b->n_quanta; q++)
              if (len128[q]) single[(size_t)r * b->n_quanta + q] = 0;
          }
        uint8_t* d_single = nullptr;
        int e = dev_upload(b, &d_single, single);
        if (e) return e;
        g.single = d_single;
        g.single_stride = b->n_quanta;
        float* tabs[7];
        for (auto& t : tabs)
          if ((e = dev_alloc(b, &t, (size_t)g.rows * g.n_frames))) return e;
        g.az = tabs[0];
        g.gl_mono = tabs[1];
        g.gr_mono = tabs[2];
        g.gl_stereo = tabs[3];
        g.gr_stereo = tabs[4];
        g.dg = tabs[5];
        g.cg = tabs[6];
        g.distance_model = n.desc.i[1];
        g.ref_distance = n.desc.d[0];
        g.max_distance = n.desc.d[1];
        g.rolloff = n.desc.d[2];
        g.cone_inner = (float)n.desc.d[3];
        g.cone_outer = (float)n.desc.d[4];
        g.cone_outer_gain = (float)n.desc.d[5];
        gs.profile_slot = slot_for(b, "panner_geom_kernel");
        b->steps.push_back(gs);
        auto ref = [&](float* base) {
          ParamRef r{};
          r.base = base;
          r.stride = g.rows == 1 ? 0 : g.n_frames;
          r.mode = 2;
          return r;
        };
        o.p0 = ref(g.az);
        o.p1 = ref(nch == 1 ? g.gl_mono : g.gl_stereo);
        o.p2 = ref(nch == 1 ? g.gr_mono : g.gr_stereo);
        o.p3 = ref(g.dg);
        o.p4 = ref(g.cg);
        plan_note(b, "panner node %u: audio-rate AudioListener automation -> per-frame geometry on the device (%u table row(s))", id,
                  g.rows);
        ops.push_back(o);
        break;
      }
      // listener single-valued => the reference evaluates the geometry once per quantum from the first value of
      // every param (panner.rs:833-846)
      const uint32_t cnt = mode == 0 ? 1 : b->n_quanta;
      const int vmode = mode == 0 ? 0 : 1;
      std::vector<float> az((size_t)b->n_inst * cnt), gl(az.size()), gr(az.size()), dg(az.size()), cg(az.size());
      for (uint32_t i = 0; i < b->n_inst; i++) {
        std::vector<std::vector<float>> pv(15);
        for (int k = 0; k < 15; k++) pv[k] = param_per_quantum(b, n.params[k], i, nullptr);
        for (uint32_t k = 0; k < cnt; k++) {
          auto at = [&](int p) { return pv[p][pv[p].size() == 1 ? 0 : k]; };
          V3 sp{at(0), at(1), at(2)}, so{at(3), at(4), at(5)}, lp{at(6), at(7), at(8)}, lf{at(9), at(10), at(11)},
              lu{at(12), at(13), at(14)};
          float a, el;
          azimuth_elevation(sp, lp, lf, lu, &a, &el);
          // panner.rs:996-1004
          a = a < -180.f ? -180.f : a > 180.f ? 180.f : a;
          if (a < -90.f)
            a = -180.f - a;
          else if (a > 90.f)
            a = 180.f - a;
          float x = nch == 1 ? (a + 90.f) / 180.f : (a <= 0.f ? (a + 90.f) / 90.f : a / 90.f);
          const size_t ix = (size_t)i * cnt + k;
          az[ix] = a;
          gl[ix] = cosf(x * PI_F / 2.f);
          gr[ix] = sinf(x * PI_F / 2.f);
          dg[ix] = dist_gain(n.desc, sp, lp);
          cg[ix] = cone_gain(n.desc, sp, so, lp);
        }
      }
      int e;
      if ((e = upload_values(b, az, vmode, &o.p0)) || (e = upload_values(b, gl, vmode, &o.p1)) ||
          (e = upload_values(b, gr, vmode, &o.p2)) || (e = upload_values(b, dg, vmode, &o.p3)) ||
          (e = upload_values(b, cg, vmode, &o.p4)))
        return e;
      ops.push_back(o);
      break;
    }
    case WAA_NODE_CONVOLVER:
      // no buffer set: passthrough (convolver.rs:368-374)
      break;
    case WAA_NODE_ANALYSER:
    case WAA_NODE_DESTINATION:
    default:
      break;
  }
  return 0;
}

void default_channel_config(Node& n, uint32_t n_out) {
  int cc = 2, mode = WAA_COUNT_MODE_MAX, interp = WAA_INTERP_SPEAKERS;
  switch (n.desc.kind) {
    case WAA_NODE_DESTINATION:
      cc = (int)n_out;
      mode = WAA_COUNT_MODE_EXPLICIT;
      break;
    case WAA_NODE_CONVOLVER:
    case WAA_NODE_STEREO_PANNER:
    case WAA_NODE_PANNER:
      mode = WAA_COUNT_MODE_CLAMPED_MAX;
      break;
    default: break;
  }
  if (n.desc.channel_count != 0) {
    cc = (int)n.desc.channel_count;
    mode = (int)n.desc.channel_count_mode;
    interp = (int)n.desc.channel_interpretation;
  }
  n.cc = cc;
  n.mode = mode;
  n.interp = interp;
}

}  // namespace host
}  // namespace waa
