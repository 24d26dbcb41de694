// waa_plan.cpp — the planner: processing order with the reference's cycle breaker, liveness, static channel
// counts, materialisation points, fusion of single-consumer paths into chain launches, node-major steps
// (convolver, delay, oscillator, IIR), feedback loops (block-scheduled or quantum-serial), AudioParam chains.
#include <array>
#include <set>

#include <memory>

#include "waa_host.hpp"
#include "waa_plan_parts.hpp"

namespace waa {
namespace host {

int prepare_source_input(waa_batch* b, uint32_t id, InputRef* in);
int validate_plan(waa_batch* b);

// Upload a param as a device ParamRef (mode 0 / 1 / 2), values clamped like the reference.
// Per-instance automation of an a-rate param: upload the event queues, plan one timeline_kernel step (waa_timeline.hip)
// that replays them for all quanta into a per-frame table, and hand that table out as the param's values.
int device_timeline_param(waa_batch* b, const ParamStore& p, ParamRef* ref) {
  if (p.dev_ready) {
    *ref = p.dev_ref;
    return 0;
  }
  std::vector<TlHeader> hdr(b->n_inst);
  std::vector<TlEvent> events;
  std::vector<float> curves;
  for (uint32_t i = 0; i < b->n_inst; i++) {
    TlHeader& h = hdr[i];
    std::memset(&h, 0, sizeof h);
    if (i < p.timelines.size() && p.timelines[i]) {
      p.timelines[i]->export_queue(&h, &events, &curves);
    } else {  // no automation on this instance: the constant
      h.intrinsic = p.cst[i];
      h.a_rate = 1;
      h.ev_off = (int32_t)events.size();
    }
    h.minv = p.minv;
    h.maxv = p.maxv;
    h.defv = p.defv;
  }
  Step st;
  st.kind = 14;
  TimelineDesc& d = st.tl;
  std::memset(&d, 0, sizeof d);
  TlHeader* d_hdr = nullptr;
  TlEvent *d_ev = nullptr, *d_work = nullptr;
  float *d_curves = nullptr, *d_out = nullptr;
  uint8_t* d_lens = nullptr;
  int e;
  if ((e = dev_upload(b, &d_hdr, hdr)) || (e = dev_upload(b, &d_ev, events)) || (e = dev_alloc(b, &d_work, events.size())) ||
      (e = dev_upload(b, &d_curves, curves)) || (e = dev_alloc(b, &d_out, (size_t)b->n_inst * b->n_quanta * RQ)) ||
      (e = dev_alloc(b, &d_lens, (size_t)b->n_inst * b->n_quanta)))
    return e;
  d.hdr = d_hdr;
  d.events = d_ev;
  d.work = d_work;
  d.curves = d_curves;
  d.out = d_out;
  d.lens = d_lens;
  d.out_stride = (uint64_t)b->n_quanta * RQ;
  d.rows = b->n_inst;
  d.n_quanta = b->n_quanta;
  d.sample_rate = (double)b->sr;
  st.profile_slot = slot_for(b, "timeline_kernel");
  b->steps.push_back(st);
  plan_note(b, "automation: %u per-instance timeline(s), %zu event(s) in total, replayed on the device", b->n_inst, events.size());
  p.dev_ref = ParamRef{d_out, d.out_stride, 2, 0};
  p.dev_lens = d_lens;
  p.dev_ready = true;
  *ref = p.dev_ref;
  return 0;
}

int upload_param(waa_batch* b, const ParamStore& p, ParamRef* ref) {
  if (p.dev_tl) return device_timeline_param(b, p, ref);
  const int mode = p.mode();
  std::vector<float> host;
  if (mode == 0) {
    host.resize(b->n_inst);
    for (uint32_t i = 0; i < b->n_inst; i++) host[i] = p.fix(p.cst[i]);
    ref->stride = 0;
  } else {
    const uint64_t per = mode == 1 ? b->n_quanta : (uint64_t)b->n_quanta * RQ;
    // the same values for every instance (constants equal, every block for all instances): one row, stride 0
    bool same = true;
    for (uint32_t i = 1; i < b->n_inst && same; i++) same = p.cst[i] == p.cst[0];
    for (auto& blk : p.blocks) same &= blk.inst == WAA_ALL_INSTANCES;
    const uint32_t rows = same ? 1u : b->n_inst;
    host.resize((size_t)rows * per);
    for (uint32_t i = 0; i < rows; i++) {
      float c = p.fix(p.cst[i]);
      std::fill(host.begin() + (size_t)i * per, host.begin() + (size_t)(i + 1) * per, c);
    }
    for (auto& blk : p.blocks) {
      uint32_t lo = blk.inst == WAA_ALL_INSTANCES ? 0 : blk.inst, hi = blk.inst == WAA_ALL_INSTANCES ? rows : blk.inst + 1;
      for (uint32_t i = lo; i < hi; i++)
        for (uint32_t k = 0; k < blk.nq; k++) {
          uint64_t q = blk.q0 + k;
          if (q >= b->n_quanta) continue;
          if (mode == 1) {
            host[(size_t)i * per + q] = p.fix(blk.v[k]);
          } else {
            for (int s = 0; s < RQ; s++)
              host[(size_t)i * per + q * RQ + s] = p.fix(blk.v[(size_t)k * blk.vpq + (blk.vpq == 1 ? 0 : s)]);
          }
        }
    }
    ref->stride = same ? 0 : per;
  }
  float* d = nullptr;
  int e = dev_upload(b, &d, host);
  if (e) return e;
  ref->base = d;
  ref->mode = mode;
  ref->pad = 0;
  return 0;
}
// Value mode of param k of node `id` as the kernels will see it: a param with an audio-rate input is per-frame.
int param_mode(const Node& n, size_t k) {
  if (k < n.pin_edges.size() && !n.pin_edges[k].empty()) return 2;
  return n.params[k].mode();
}
int push_chain_step(waa_batch* b, const std::vector<InputRef>& inputs, int in_nch, int in_interp,
                    const std::vector<OpDesc>& ops, const SignalRef& out);
int temp_signal(waa_batch* b, int nch, SignalRef* out);
int reduce_fan_in(waa_batch* b, std::vector<InputRef>& ins, int in_nch, int interp);
// ParamRef of param k of node `id`.  Without an audio-rate input: the uploaded constants / value blocks.  With one
// (param.rs:686-795): a chain that sums the connected outputs mixed to ONE channel (count 1, explicit, discrete,
// param.rs:309-311), adds the intrinsic value and clamps, written to a one-channel signal the consumer reads
// as per-frame values.  Planned once, right before the first consumer (its producers are materialised and
// precede the owner in processing order).
int build_edge_input(waa_batch* b, uint32_t head, int ie, InputRef* out);
int node_param(waa_batch* b, uint32_t id, size_t k, ParamRef* ref) {
  Node& n = b->nodes[id];
  if (k >= n.pin_edges.size() || n.pin_edges[k].empty()) return upload_param(b, n.params[k], ref);
  if (n.pin_ready[k]) {
    *ref = n.pin_ref[k];
    return 0;
  }
  std::vector<InputRef> ins;
  for (int ie : n.pin_edges[k]) {
    // (a GainNode between a modulator and the param — the LFO depth — rides on the edge, see the materialisation pass)
    InputRef in{};
    int e = build_edge_input(b, id, ie, &in);
    if (e) return e;
    ins.push_back(in);
  }
  int e = reduce_fan_in(b, ins, 1, WAA_INTERP_DISCRETE);
  if (e) return e;
  OpDesc o{};
  o.kind = OP_PARAM_ADD;
  o.nch_in = o.nch_out = 1;
  if ((e = upload_param(b, n.params[k], &o.p0))) return e;
  auto bits = [](float f) {
    int32_t i;
    std::memcpy(&i, &f, 4);
    return i;
  };
  o.i0 = bits(n.params[k].minv);
  o.i1 = bits(n.params[k].maxv);
  o.i2 = bits(n.params[k].defv);
  SignalRef sig;
  if ((e = temp_signal(b, 1, &sig))) return e;
  if ((e = push_chain_step(b, ins, 1, WAA_INTERP_DISCRETE, {o}, sig))) return e;
  ParamRef r{};
  r.base = sig.base;
  r.stride = sig.inst_stride;
  r.mode = 2;
  n.pin_ref[k] = r;
  n.pin_ready[k] = 1;
  *ref = r;
  return 0;
}

// One input edge of a summing node as an InputRef: a materialised signal, a source fetched by the kernel itself, a
// DelayNode read from its delay line (IN_DELAYED), each optionally through a GainNode folded into the edge.
int build_edge_input(waa_batch* b, uint32_t head, int ie, InputRef* out) {
  uint32_t pid = b->edges[ie].from;
  InputRef in{};
  if (!b->nodes[pid].materialized && b->nodes[pid].desc.kind == WAA_NODE_GAIN) {
    // a GainNode folded into the edge (see the materialisation pass)
    int e = node_param(b, pid, 0, &in.gain);
    if (e) return e;
    in.has_gain = 1;
    plan_note(b, "gain node %u folded into an input edge of node %u", pid, head);
    pid = b->edges[b->nodes[pid].in_edges[0]].from;
  }
  Node& pn = b->nodes[pid];
  in.nch = pn.out_nch;
  if (pn.materialized) {
    in.kind = IN_SIGNAL;
    in.sig = pn.sig;
  } else if (pn.delay_folded) {
    if (!pn.hist.base) return fail(WAA_ERR_INVALID_STATE, "internal: delay line of node %u not planned", pid);
    in.kind = IN_DELAYED;
    in.sig = pn.hist;
    in.nch = pn.in_nch;
    int e = node_param(b, pid, WAA_PARAM_DELAY_DELAY_TIME, &in.offset);
    if (e) return e;
    if (in.offset.mode == 0) {
      // one delayTime for the whole batch: an immediate (mode 3, the value in `stride`) instead of a load every wave has
      // to wait for before it can form the addresses of its samples
      const ParamStore& ps = pn.params[WAA_PARAM_DELAY_DELAY_TIME];
      bool same = true;
      for (uint32_t i = 1; i < b->n_inst && same; i++) same = ps.cst[i] == ps.cst[0];
      if (same && !ps.dev_tl) {
        const float v0 = ps.fix(ps.cst[0]);  // (the clamped value upload_param wrote)
        uint32_t bits;
        std::memcpy(&bits, &v0, 4);
        in.offset.mode = 3;
        in.offset.stride = bits;
      }
    }
    in.delay_lo = 1.f;
    in.delay_hi = 0.f;
    {
      const ParamStore& ps = pn.params[WAA_PARAM_DELAY_DELAY_TIME];
      if ((in.offset.mode == 0 || in.offset.mode == 3) && !ps.dev_tl && ps.blocks.empty()) {
        float lo = 1e30f, hi = 0.f;
        for (uint32_t i = 0; i < b->n_inst; i++) {
          const float fr = ps.fix(ps.cst[i]) * b->sr;
          lo = std::min(lo, fr);
          hi = std::max(hi, fr);
        }
        in.delay_lo = lo;
        in.delay_hi = hi;
      }
    }
    in.sample_rate = (double)b->sr;
    in.num_quanta = (int32_t)std::ceil(pn.desc.d[0] * (double)b->sr / (double)RQ);
    in.valid = pn.hist_valid;
    // the delay line of a loop-breaking DelayNode is written by a LATER launch of the same block (the reads go to
    // earlier blocks: the loop is block-scheduled with blocks shorter than the delay)
    in.feedback = pid < b->cut.size() && b->cut[pid] ? 1 : 0;
  } else if (pn.is_view && !measure_switch("WAA_NO_VIEW_SIGNAL")) {
    // a BufferSource that renders its AudioBuffer unchanged: a plain signal with an end (one layout for all instances:
    // no per-instance source record in front of the samples); silence past the buffer
    in.kind = IN_SIGNAL;
    in.sig = pn.view_sig;
    in.valid = pn.view_valid;
  } else if (pn.desc.kind == WAA_NODE_BUFFER_SOURCE || pn.desc.kind == WAA_NODE_CONSTANT_SOURCE) {
    in.kind = pn.desc.kind == WAA_NODE_BUFFER_SOURCE ? IN_SOURCE : IN_CONSTANT;
    int e = prepare_source_input(b, pid, &in);
    if (e) return e;
  } else {
    return fail(WAA_ERR_INVALID_STATE, "internal: unmaterialised fan-in input");
  }
  *out = in;
  return 0;
}


// Upload host-computed per-instance (mode 0) or per-(instance, quantum) (mode 1) values.
int upload_values(waa_batch* b, const std::vector<float>& host, int mode, ParamRef* ref) {
  float* d = nullptr;
  int e = dev_upload(b, &d, host);
  if (e) return e;
  ref->base = d;
  ref->mode = mode;
  ref->stride = mode == 0 ? 0 : b->n_quanta;
  ref->pad = 0;
  return 0;
}

int computed_in_nch(const Node& n, int maxc) {
  switch (n.mode) {
    case WAA_COUNT_MODE_MAX: return maxc;
    case WAA_COUNT_MODE_EXPLICIT: return n.cc;
    default: return std::min(maxc, n.cc);
  }
}

// ---- the ORDER of a node's inputs (graph.rs:524-535 + AudioRenderQuantum::add, quantum.rs:425-470) ----------------------------
// The reference sums a node's inputs one after the other into a bus that starts as one silent channel: for every input the bus is
// mixed to f(max(bus, input)) channels FIRST (f = the node's count mode), then the input is mixed to that count and added.  An input
// therefore reaches the final count through every width the bus takes behind it, and the speakers table is not transitive:
// mono -> stereo -> 5.1 leaves the mono signal in L and R, mono -> 5.1 puts it in C; quad -> 5.1 -> 8 leaves the surrounds in
// channels 4 and 5, quad -> 8 (discrete above six channels) in 2 and 3.  mix_matrix is mix_regs (waa_mix.hpp) as a matrix
// [to][from]; an input whose path differs from its direct mix is pre-mixed along the path (premix_ordered_inputs).
static std::vector<double> mix_matrix(int from, int to, int interp) {
  std::vector<double> m((size_t)to * from, 0.);
  auto at = [&](int r, int c) -> double& { return m[(size_t)r * from + c]; };
  const double s = 0.70710678118654752440;
  bool done = true;
  if (from == to || interp == WAA_INTERP_DISCRETE || from > 6 || to > 6) {
    done = false;
  } else if (from == 1 && to == 2) {
    at(0, 0) = at(1, 0) = 1;
  } else if (from == 2 && to == 1) {
    at(0, 0) = at(0, 1) = 0.5;
  } else if (from == 1 && to == 4) {
    at(0, 0) = at(1, 0) = 1;
  } else if (from == 2 && to == 4) {
    at(0, 0) = at(1, 1) = 1;
  } else if (from == 4 && to == 1) {
    for (int c = 0; c < 4; c++) at(0, c) = 0.25;
  } else if (from == 4 && to == 2) {
    at(0, 0) = at(0, 2) = at(1, 1) = at(1, 3) = 0.5;
  } else if (from == 1 && to == 6) {
    at(2, 0) = 1;
  } else if (from == 2 && to == 6) {
    at(0, 0) = at(1, 1) = 1;
  } else if (from == 4 && to == 5) {
    at(0, 0) = at(1, 1) = at(3, 2) = at(4, 3) = 1;
  } else if (from == 4 && to == 6) {
    at(0, 0) = at(1, 1) = at(4, 2) = at(5, 3) = 1;
  } else if (from == 6 && to == 1) {
    at(0, 0) = at(0, 1) = s;
    at(0, 2) = 1;
    at(0, 4) = at(0, 5) = 0.5;
  } else if (from == 6 && to == 2) {
    at(0, 0) = at(1, 1) = 1;
    at(0, 2) = at(1, 2) = at(0, 4) = at(1, 5) = s;
  } else if (from == 6 && to == 4) {
    at(0, 0) = at(1, 1) = at(2, 4) = at(3, 5) = 1;
    at(0, 2) = at(1, 2) = s;
  } else {
    done = false;
  }
  if (!done)  // pad with silence / truncate
    for (int c = 0; c < std::min(from, to); c++) at(c, c) = 1;
  return m;
}
// widths the bus takes: w[k] = the count input k is mixed to when it is added (static widths)
static std::vector<int> bus_widths(const Node& n, const std::vector<int>& in_w) {
  std::vector<int> w(in_w.size());
  int bus = 1;
  for (size_t k = 0; k < in_w.size(); k++) {
    bus = computed_in_nch(n, std::max(bus, in_w[k]));
    w[k] = bus;
  }
  return w;
}
// does input k reach the final count by another matrix than its direct mix?
static bool path_differs(const Node& n, const std::vector<int>& in_w, const std::vector<int>& w, size_t k) {  // (all inputs active)
  const int last = w.back();
  std::vector<double> pm = mix_matrix(in_w[k], w[k], n.interp);
  int cur = w[k];
  for (size_t j = k + 1; j < w.size(); j++) {
    if (w[j] == cur) continue;
    const std::vector<double> step = mix_matrix(cur, w[j], n.interp);
    std::vector<double> nm((size_t)w[j] * in_w[k], 0.);
    for (int r = 0; r < w[j]; r++)
      for (int c = 0; c < in_w[k]; c++) {
        double acc = 0;
        for (int t = 0; t < cur; t++) acc += step[(size_t)r * cur + t] * pm[(size_t)t * in_w[k] + c];
        nm[(size_t)r * in_w[k] + c] = acc;
      }
    pm.swap(nm);
    cur = w[j];
  }
  return pm != mix_matrix(in_w[k], last, n.interp);
}
// Does the reference's sum differ from what the static plan renders (every input mixed along the path premix_ordered_inputs derives from
// the STATIC widths) in a quantum in which exactly the inputs of `active` (bit k = input k) are active?  A silent input is one channel
// of zeros: it adds nothing and does not widen the bus, so the inputs behind it see another sequence of widths.
bool inputs_mix_in_order_differs(const waa_batch* b, const Node& n, uint64_t active) {
  if (n.mode == WAA_COUNT_MODE_EXPLICIT || n.in_edges.size() < 2 || n.in_edges.size() > 64) return false;  // (explicit: every input is mixed to `count` directly)
  std::vector<int> in_w, dyn_w;
  for (size_t k = 0; k < n.in_edges.size(); k++) {
    in_w.push_back(b->nodes[b->edges[n.in_edges[k]].from].out_nch);
    dyn_w.push_back((active >> k) & 1 ? in_w.back() : 1);
  }
  const std::vector<int> ws = bus_widths(n, in_w), wd = bus_widths(n, dyn_w);
  if (wd.back() != ws.back()) return false;  // (narrower than the static count altogether: the finding above)
  for (size_t k = 0; k < in_w.size(); k++) {
    if (!((active >> k) & 1)) continue;
    // the matrix the static plan applies to input k  vs  the one the reference applies in this quantum
    auto path = [&](const std::vector<int>& w) {
      std::vector<double> pm = mix_matrix(in_w[k], w[k], n.interp);
      int cur = w[k];
      for (size_t j = k + 1; j < w.size(); j++) {
        if (w[j] == cur) continue;
        const std::vector<double> step = mix_matrix(cur, w[j], n.interp);
        std::vector<double> nm((size_t)w[j] * in_w[k], 0.);
        for (int r = 0; r < w[j]; r++)
          for (int c = 0; c < in_w[k]; c++) {
            double acc = 0;
            for (int t = 0; t < cur; t++) acc += step[(size_t)r * cur + t] * pm[(size_t)t * in_w[k] + c];
            nm[(size_t)r * in_w[k] + c] = acc;
          }
        pm.swap(nm);
        cur = w[j];
      }
      return pm;
    };
    if (path(ws) != path(wd)) return true;
  }
  return false;
}
// `ins` = the node's inputs in edge order (before any fan-in reduction): every input whose path differs is replaced by a
// temporary signal of the final count that holds it mixed along its path
int premix_ordered_inputs(waa_batch* b, uint32_t id, std::vector<InputRef>& ins) {
  const Node& n = b->nodes[id];
  if (n.mode == WAA_COUNT_MODE_EXPLICIT || ins.size() < 2) return 0;
  std::vector<int> in_w;
  for (auto& in : ins) in_w.push_back(in.nch);
  const std::vector<int> w = bus_widths(n, in_w);
  if (w.back() != n.in_nch) return 0;  // (static counts that are not the plain maximum: a fold upstream — nothing to order)
  for (size_t k = 0; k < ins.size(); k++) {
    if (!path_differs(n, in_w, w, k)) continue;
    std::vector<OpDesc> ops;
    int cur = w[k];
    for (size_t j = k + 1; j < w.size(); j++) {
      if (w[j] == cur) continue;
      OpDesc o{};
      o.kind = OP_MIX;
      o.nch_in = cur;
      o.nch_out = w[j];
      o.i0 = n.interp;
      ops.push_back(o);
      cur = w[j];
    }
    SignalRef tmp;
    if (int e = temp_signal(b, cur, &tmp)) return e;
    plan_note(b, "node %u: input %zu (%d channels) is mixed along the widths the reference's input bus takes behind it (%d -> ... -> %d): not its direct mix",
              id, k, in_w[k], w[k], cur);
    if (int e = push_chain_step(b, {ins[k]}, w[k], n.interp, ops, tmp)) return e;
    InputRef in{};
    in.kind = IN_SIGNAL;
    in.nch = cur;
    in.sig = tmp;
    ins[k] = in;
  }
  return 0;
}

// graph.rs:323-487 order_nodes / visit.  A DelayNode is two graph nodes in the reference (delay.rs:283-366:
// writer, then reader, edge writer->reader); vertex `id` is the writer (or a plain node), `id | VTX_READER` the
// reader.  Cycles are broken at the first DelayNode writer on the detected loop (its writer->reader edge is
// cleared and the ordering restarts); nodes of a loop without one are dropped from the ordering (muted).
struct OrderCtx {
  const waa_batch* b;
  std::vector<uint8_t> cut;
  std::vector<uint32_t> marked, temp, ordered, in_cycle;
  uint32_t breaker = 0;
};
bool is_delay(const waa_batch* b, uint32_t id);
void vertex_targets(const waa_batch* b, uint32_t v, const std::vector<uint8_t>& cut, std::vector<uint32_t>& out);
bool order_visit(OrderCtx& c, uint32_t v) {
  auto pos = std::find(c.temp.begin(), c.temp.end(), v);
  if (pos != c.temp.end()) {
    for (auto it = pos; it != c.temp.end(); ++it)
      if (!(*it & VTX_READER) && is_delay(c.b, *it)) {
        c.breaker = *it;
        return true;
      }
    c.in_cycle.insert(c.in_cycle.end(), pos, c.temp.end());
    return false;
  }
  if (std::find(c.marked.begin(), c.marked.end(), v) != c.marked.end()) return false;
  c.marked.push_back(v);
  c.temp.push_back(v);
  std::vector<uint32_t> targets;
  vertex_targets(c.b, v, c.cut, targets);
  for (uint32_t t : targets)
    if (order_visit(c, t)) return true;
  c.ordered.push_back(v);
  c.temp.erase(std::remove(c.temp.begin(), c.temp.end(), v), c.temp.end());
  return false;
}

int plan_convolver(waa_batch* b, uint32_t id);
int reduce_fan_in(waa_batch* b, std::vector<InputRef>& ins, int in_nch, int interp);

void plan_note(waa_batch* b, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  b->plan_log.emplace_back(buf);
}
const char* input_kind_name(int k) {
  switch (k) {
    case IN_SIGNAL: return "signal";
    case IN_SOURCE: return "source";
    case IN_CONSTANT: return "constant";
    case IN_DELAYED: return "delayed";
    default: return "silent";
  }
}
const char* op_name(int k) {
  switch (k) {
    case OP_GAIN: return "GAIN";
    case OP_BIQUAD: return "BIQUAD";
    case OP_WAVESHAPER: return "WAVESHAPER";
    case OP_STEREO_PAN: return "STEREO_PAN";
    case OP_PANNER: return "PANNER";
    case OP_MIX: return "MIX";
    case OP_IIR: return "IIR";
    case OP_PARAM_ADD: return "PARAM_ADD";
    default: return "?";
  }
}

int slot_for(waa_batch* b, const char* name) {
  for (size_t i = 0; i < b->prof.size(); i++)
    if (b->prof[i].name == name) return (int)i;
  b->prof.push_back(ProfileEntry{name});
  return (int)b->prof.size() - 1;
}

// ---------------------------------------------------------------------------------------
// planner
// ---------------------------------------------------------------------------------------
int emit_node_ops(waa_batch* b, uint32_t id, int cur_nch, bool head, std::vector<OpDesc>& ops, int* out_nch);

// ---- signals wider than six channels (MAX_CHANNELS = 32, src/lib.rs:21) ----------------------------------------------
// AudioRenderQuantum::mix (quantum.rs:285-306): as soon as either side of a mix has more than six channels the interpretation is
// "discrete" whatever the node says — channel c of the result is channel c of the input or silence.  Everything the chain kernel
// does to a signal is then independent per channel (sums, gains, curves; the panners and the speakers table only ever see <= 6),
// so a wide chain is rendered as SLICES of at most six channels: slice s is the same chain on channels [6 s, 6 s + 6) of every
// input, op and output, with every count n replaced by clamp(n - 6 s, 0, 6).  Counts <= 6 live in slice 0 unchanged (with the
// node's own interpretation); in the other slices they are "no channel at all", modelled as one channel of zeros that no op
// touches (a discrete up-mix of it pads with those zeros).
constexpr int SLICE_CH = 6;
static int slice_count(int n, int c0) { return std::max(0, std::min(n - c0, SLICE_CH)); }
static SignalRef slice_signal(const SignalRef& sg, int c0) {
  SignalRef r = sg;
  r.base = sg.base + (uint64_t)c0 * sg.ch_stride;
  r.nch = slice_count(sg.nch, c0);
  return r;
}
int push_chain_step_narrow(waa_batch* b, const std::vector<InputRef>& inputs, int in_nch, int in_interp, const std::vector<OpDesc>& ops,
                           const SignalRef& out);
int push_chain_step(waa_batch* b, const std::vector<InputRef>& inputs, int in_nch, int in_interp, const std::vector<OpDesc>& ops,
                    const SignalRef& out) {
  int cmax = std::max(in_nch, out.nch);
  for (auto& in : inputs) cmax = std::max(cmax, in.nch);
  for (auto& o : ops) cmax = std::max({cmax, o.nch_in, o.nch_out});
  if (cmax <= SLICE_CH) return push_chain_step_narrow(b, inputs, in_nch, in_interp, ops, out);
  plan_note(b, "chain on up to %d channels: rendered as %d slices of at most %d channels (discrete mixing above six channels is per channel)", cmax,
            (cmax + SLICE_CH - 1) / SLICE_CH, SLICE_CH);
  auto wide_mix = [](int from, int to, int interp) { return interp == WAA_INTERP_DISCRETE || from > 6 || to > 6; };
  for (int c0 = 0; c0 < out.nch; c0 += SLICE_CH) {
    std::vector<InputRef> ins;
    for (const InputRef& in0 : inputs) {
      InputRef in = in0;
      // (an input narrower than the receiver is padded discretely when the receiver is wide: nothing of it in the later slices;
      // when both are narrow the speakers rules apply — in slice 0, where all their channels are)
      in.nch = slice_count(in0.nch, c0);
      if (in0.nch > 6 && in_nch <= 6) in.nch = std::min(in.nch, slice_count(in_nch, c0));  // (a wide input into a narrow receiver: truncated)
      if (in.nch == 0) continue;
      if (c0 > 0) switch (in0.kind) {
          case IN_SIGNAL:
          case IN_DELAYED: in.sig = slice_signal(in0.sig, c0); break;
          case IN_SOURCE: {
            auto it = b->src_tables.find(in0.src);
            if (it == b->src_tables.end()) return fail(WAA_ERR_DEVICE, "internal: a source table without its host copy");
            std::vector<SrcInst> t = it->second;
            for (auto& si : t)
              if (si.base) si.base += (uint64_t)c0 * si.ch_stride;
            SrcInst* dt = nullptr;
            if (int e = dev_upload(b, &dt, t)) return e;
            in.src = dt;
            break;
          }
          default: return fail(WAA_ERR_DEVICE, "internal: input kind %d with more than %d channels", in0.kind, SLICE_CH);
        }
      ins.push_back(in);
    }
    // the chain's counts in this slice; 0 = "nothing here" (see above)
    int cur = slice_count(in_nch, c0);
    int s_in_nch = cur, s_in_interp = in_interp;
    if (in_nch > 6) s_in_interp = WAA_INTERP_DISCRETE;
    bool nothing = cur == 0;  // the data so far are zeros that stand for no channel
    if (nothing) {
      // (inputs narrower than a wide in_nch were dropped above; a narrow in_nch leaves every input of this slice without channels)
      ins.clear();
      s_in_nch = 1;
    }
    if (ins.empty()) {
      InputRef z{};
      z.kind = IN_SILENT;
      z.nch = 1;
      ins.push_back(z);
    }
    std::vector<OpDesc> sops;
    for (const OpDesc& o0 : ops) {
      const int ni = slice_count(o0.nch_in, c0), no = slice_count(o0.nch_out, c0);
      if (ni == 0 && no == 0) continue;  // the op works on channels that are not in this slice
      if (no == 0)  // the slice's channels end here; anything behind would have to start from silence again
        return fail(WAA_ERR_OUT_OF_SCOPE, "a signal of more than six channels that narrows and widens again within one fused chain is out of scope");
      OpDesc o = o0;
      if (o0.kind == OP_MIX) {
        if (ni == 0 && !nothing) return fail(WAA_ERR_DEVICE, "internal: slice bookkeeping");
        o.nch_in = ni == 0 ? 1 : ni;  // (from nothing: from the one channel of zeros)
        o.nch_out = no;
        if (wide_mix(o0.nch_in, o0.nch_out, o0.i0)) o.i0 = WAA_INTERP_DISCRETE;
        nothing = false;
      } else if (o0.nch_in > 6 || o0.nch_out > 6) {
        // an op on a wide signal: per channel or not at all
        if (o0.nch_in != o0.nch_out || (o0.kind != OP_GAIN && o0.kind != OP_WAVESHAPER))
          return fail(WAA_ERR_OUT_OF_SCOPE, "a %s op on a signal of more than six channels is out of scope", op_name(o0.kind));
        o.nch_in = o.nch_out = ni;
      }  // (else: an op on a narrow signal, all of it in slice 0 — the later slices skipped it above)
      sops.push_back(o);
      cur = no;
    }
    const SignalRef so = slice_signal(out, c0);
    if (nothing) {  // the chain never reaches this slice's channels: they are silence
      if (cur == 0 && so.nch > 0) cur = 1;
    }
    if (int e = push_chain_step_narrow(b, ins, s_in_nch, s_in_interp, sops, so)) return e;
  }
  return 0;
}
int push_chain_step_narrow(waa_batch* b, const std::vector<InputRef>& inputs, int in_nch, int in_interp,
                           const std::vector<OpDesc>& ops, const SignalRef& out) {
  if (ops.size() > (size_t)MAX_OPS) return fail(WAA_ERR_OUT_OF_SCOPE, "more than %d fused ops in one chain", MAX_OPS);
  Step st;
  ChainDesc& cd = st.chain;
  std::memset(&cd, 0, sizeof cd);
  int cmax = std::max(in_nch, out.nch);
  cd.n_inputs = (int)inputs.size();
  for (size_t k = 0; k < inputs.size(); k++) {
    cd.in[k] = inputs[k];
    cmax = std::max(cmax, inputs[k].nch);
  }
  cd.in_nch = in_nch;
  cd.in_interp = in_interp;
  for (auto& o : ops) cmax = std::max({cmax, o.nch_in, o.nch_out});
  for (auto& o : ops)
    if (o.kind == OP_IIR) return fail(WAA_ERR_DEVICE, "internal: IIR op reached the interpreter kernel");
  for (auto& o : ops)
    if (o.kind == OP_BIQUAD && cmax > 2)
      return fail(WAA_ERR_OUT_OF_SCOPE,
                  "the serial chain interpreter renders a BiquadFilter on at most 2 channels (needs %d; only reached with the WAA_NO_*_STREAM debugging switches)",
                  cmax);
  cd.n_ops = (int)ops.size();
  for (size_t k = 0; k < ops.size(); k++) cd.ops[k] = ops[k];
  cd.out = out;
  cd.n_inst = b->n_inst;
  cd.n_tiles = b->n_tiles;
  cd.tile0 = 0;
  cd.tile1 = b->n_tiles;
  cd.n_quanta = b->n_quanta;
  st.cmax = cmax;
  st.profile_slot = slot_for(b, cmax <= 1 ? "chain_kernel<1>" : cmax <= 2 ? "chain_kernel<2>" : "chain_kernel<4+>");
  {
    int curve_op = -1;
    bool has_biquad = false;
    for (auto& o : ops) has_biquad |= o.kind == OP_BIQUAD;
    if (!has_biquad && resample_shape(cd, &curve_op)) st.profile_slot = slot_for(b, "resample_kernel");
  }
  b->steps.push_back(st);
  {
    bool serial = false;
    std::string desc;
    for (auto& o : ops) {
      serial |= o.kind == OP_BIQUAD;
      char t[64];
      if (o.kind == OP_MIX)
        snprintf(t, sizeof t, "MIX(%d->%d)", o.nch_in, o.nch_out);
      else if (o.kind == OP_BIQUAD)
        snprintf(t, sizeof t, "BIQUAD(%s)", o.i0 == 0 ? "const" : o.i0 == 1 ? "k-rate" : "a-rate");
      else
        snprintf(t, sizeof t, "%s", op_name(o.kind));
      desc += desc.empty() ? t : std::string(",") + t;
    }
    std::string ins;
    for (auto& in : inputs) {
      char t[48];
      snprintf(t, sizeof t, "%s%s:%dch", in.has_gain ? "gain*" : "", input_kind_name(in.kind), in.nch);
      ins += ins.empty() ? t : std::string("+") + t;
    }
    plan_note(b, "chain %s C=%d in=[%s]->%dch ops=[%s] out=%dch", serial ? "serial" : "parallel", cmax, ins.c_str(), in_nch,
              desc.c_str(), out.nch);
  }
  return 0;
}

int temp_signal(waa_batch* b, int nch, SignalRef* out) {
  float* p = nullptr;
  int e = dev_alloc(b, &p, (size_t)b->n_inst * nch * b->lp);
  if (e) return e;
  *out = SignalRef{p, (uint64_t)nch * b->lp, b->lp, nch, 0};
  return 0;
}

// Turn one fused chain into kernel launches.  A Biquad with constant coefficients (plus up to two constant
// gains right behind it) goes to the streaming kernel (one wave per instance-channel, ~60 % of HBM peak); the
// ops around it run on the tile-parallel element-wise kernel.  Splitting costs one extra pass through HBM
// per cut but keeps every segment on a kernel that is 3-6x closer to the roofline than the serial interpreter,
// which remains the path for per-quantum / per-frame coefficient biquads.
int emit_segments(waa_batch* b, std::vector<InputRef> inputs, int in_nch, int in_interp, const std::vector<OpDesc>& ops,
                  const SignalRef& out) {
  bool any_stream = false;
  // debugging aids: force k-rate / a-rate biquads onto the serial interpreter
  const int max_mode = measure_switch("WAA_NO_KRATE_STREAM") ? 0 : (measure_switch("WAA_NO_ARATE_STREAM") ? 1 : 2);
  // (any channel count: the streaming kernels run one wavefront per (instance, channel) with per-channel state, like the
  // reference's per-channel state vector, biquad_filter.rs:797-812 — 4- and 6-channel signals included)
  auto streams = [&](const OpDesc& o) { return o.kind == OP_IIR || (o.kind == OP_BIQUAD && o.i0 <= max_mode && o.nch_in <= MAX_CH); };
  for (auto& o : ops) any_stream |= streams(o);
  if (!any_stream) return push_chain_step(b, inputs, in_nch, in_interp, ops, out);
  std::vector<OpDesc> pending;
  int cur_nch = in_nch;
  size_t i = 0;
  while (i < ops.size()) {
    const OpDesc& o = ops[i];
    if (!streams(o)) {
      pending.push_back(o);
      cur_nch = o.nch_out;
      i++;
      continue;
    }
    // the streaming kernel wants ONE plain input (signal or source) of exactly the biquad's channel count
    const bool iir_exact = o.kind == OP_IIR && o.i0 < 0;  // the lane-per-stream kernel reads a materialised signal
    const bool plain = pending.empty() && inputs.size() == 1 && !inputs[0].has_gain &&
                       ((inputs[0].kind == IN_SOURCE && !iir_exact) || (inputs[0].kind == IN_SIGNAL && inputs[0].valid == 0)) &&
                       inputs[0].nch == cur_nch;
    if (!plain) {
      SignalRef tmp;
      int e = temp_signal(b, cur_nch, &tmp);
      if (e) return e;
      if ((e = push_chain_step(b, inputs, in_nch, in_interp, pending, tmp))) return e;
      pending.clear();
      InputRef in{};
      in.kind = IN_SIGNAL;
      in.nch = cur_nch;
      in.sig = tmp;
      inputs.assign(1, in);
      in_nch = cur_nch;
    }
    size_t j = i + 1;
    if (o.kind == OP_BIQUAD)  // the biquad kernel applies up to two constant gains on the way out
      while (j < ops.size() && j - i <= 2 && ops[j].kind == OP_GAIN && ops[j].p0.mode == 0 && ops[j].nch_in == cur_nch) j++;
    // a mono biquad whose result only goes through the speakers up-mix 1 -> 2 into `out`: the kernel writes both channels
    const bool dup = o.kind == OP_BIQUAD && o.i0 == 0 && cur_nch == 1 && out.nch == 2 && j + 1 == ops.size() && ops[j].kind == OP_MIX &&
                     ops[j].nch_in == 1 && ops[j].nch_out == 2 && ops[j].i0 == WAA_INTERP_SPEAKERS && !measure_switch("WAA_NO_STREAM_DUP");
    SignalRef seg_out = out;
    if (!dup && (j < ops.size() || out.nch != cur_nch)) {
      int e = temp_signal(b, cur_nch, &seg_out);
      if (e) return e;
    }
    if (o.kind == OP_IIR) {
      Step st;
      st.kind = 6;
      IirStreamDesc& q = st.iir;
      std::memset(&q, 0, sizeof q);
      q.in = inputs[0];
      q.coef = reinterpret_cast<const double*>(o.ptr0);
      q.pow = reinterpret_cast<const double*>(o.ptr2);
      q.state = reinterpret_cast<double*>(o.ptr1);
      q.ns = std::abs(o.i0);
      // exact kernels: one lane per stream pays off for low orders or very many streams, one DPP row per
      // stream otherwise (issue cycles per frame: ~12 ns + 32 per 64 streams vs ~64 per 4 streams, on 1024 SIMDs)
      q.exact = 0u;
      if (iir_exact) {
        const double streams = (double)b->n_inst * cur_nch;
        const double lane_cost = std::ceil(streams / 64. / 1024.) * (12. * q.ns + 32.);
        const double row_cost = std::ceil(streams / 4. / 1024.) * 64.;
        q.exact = (row_cost < lane_cost && !measure_switch("WAA_IIR_LANE")) || measure_switch("WAA_IIR_ROW") ? 2u : 1u;
      }
      q.nch = cur_nch;
      q.out = seg_out;
      q.n_inst = b->n_inst;
      q.n_tiles = b->n_tiles;
      q.tile0 = 0;
      q.tile1 = b->n_tiles;
      q.n_quanta = b->n_quanta;
      char name[32];
      snprintf(name, sizeof name, "%s<%d>", q.exact == 2 ? "iir_row_kernel" : q.exact == 1 ? "iir_lane_kernel" : "iir_stream_kernel", q.ns);
      st.profile_slot = slot_for(b, name);
      b->steps.push_back(st);
      plan_note(b, "%s states=%d in=%s:%dch out=%s", q.exact == 2 ? "iir_exact(row)" : q.exact == 1 ? "iir_exact(lane)" : "iir_stream", q.ns,
                input_kind_name(inputs[0].kind), cur_nch, seg_out.base == out.base ? "final" : "temp");
      InputRef in{};
      in.kind = IN_SIGNAL;
      in.nch = cur_nch;
      in.sig = seg_out;
      inputs.assign(1, in);
      in_nch = cur_nch;
      i = j;
      if (i == ops.size() && seg_out.base == out.base) return 0;
      continue;
    }
    if (o.kind == OP_BIQUAD && o.i0 == 2 && !dup && b->steps[(size_t)o.i1].coef.rows == 1 && !measure_switch("WAA_ARATE_STREAM") &&
        (inputs[0].kind == IN_SIGNAL || inputs[0].kind == IN_SOURCE) &&
        (uint64_t)b->n_inst * seg_out.inst_stride < (1ull << 32)) {  // (its output rows are 32-bit element offsets)
      // per-frame coefficients, ONE table for all instances: a lane per stream, tiles in parallel (waa_biquad_lanes.hip)
      Step& cstep = b->steps[(size_t)o.i1];
      Step& dstep = b->steps[(size_t)o.i1 + 1];  // (reserved right behind the coefficient step by emit_node_ops)
      cstep.coef.lane_major = 0;
      const size_t n_streams = (size_t)b->n_inst * cur_nch;
      double *ht = nullptr, *z = nullptr, *sin = nullptr;
      int e;
      if ((e = dev_alloc(b, &ht, (size_t)b->n_tiles * BIQUAD_HT_WORDS)) || (e = dev_alloc(b, &z, (size_t)b->n_tiles * n_streams * 2)) ||
          (e = dev_alloc(b, &sin, (size_t)b->n_tiles * n_streams * 2)))
        return e;
      BiquadLanesDesc L;
      std::memset(&L, 0, sizeof L);
      L.in = inputs[0];
      L.coefs = cstep.coef.coefs;
      L.ht = ht;
      L.z = z;
      L.sin = sin;
      L.state = reinterpret_cast<double*>(o.ptr1);
      L.n_gain = (int)(j - i - 1);
      for (size_t k = i + 1; k < j; k++) L.gain[k - i - 1] = ops[k].p0;
      L.nch = cur_nch;
      L.out = seg_out;
      L.n_inst = b->n_inst;
      L.n_tiles = b->n_tiles;
      L.n_quanta = b->n_quanta;
      L.tile0 = 0;
      L.tile1 = b->n_tiles;
      L.fast_tiles = inputs[0].kind == IN_SOURCE ? inputs[0].fast_tiles
                                                 : (inputs[0].valid ? (uint32_t)std::min<uint64_t>(inputs[0].valid / TILE, b->n_tiles) : b->n_tiles);
      dstep.kind = 18;
      dstep.lanes = L;
      dstep.profile_slot = slot_for(b, "biquad_tile_digest_kernel");
      Step ls;
      ls.kind = 19;
      ls.lanes = L;
      ls.profile_slot = slot_for(b, "biquad_lanes_kernel");
      b->steps.push_back(ls);
      plan_note(b, "biquad_lanes(a-rate, shared table: one lane per stream, tiles in parallel) in=%s:%dch gains=%d out=%s",
                input_kind_name(inputs[0].kind), cur_nch, L.n_gain, seg_out.base == out.base ? "final" : "temp");
      InputRef in{};
      in.kind = IN_SIGNAL;
      in.nch = cur_nch;
      in.sig = seg_out;
      inputs.assign(1, in);
      in_nch = cur_nch;
      i = j;
      if (i == ops.size() && seg_out.base == out.base) return 0;
      continue;
    }
    Step st;
    st.kind = 1;
    BiquadStreamDesc& q = st.bq;
    std::memset(&q, 0, sizeof q);
    q.in = inputs[0];
    q.coefs = reinterpret_cast<const double*>(o.ptr0);
    q.coef_stride = o.u0;
    q.vary = o.i0;
    if (o.i0 == 2) {
      Step& cstep = b->steps[(size_t)o.i1];
      cstep.coef.lane_major = 1;  // the table is read lane by lane (waa_biquad_stream.hip)
      if (cstep.coef.rows == 1 && !measure_switch("WAA_NO_ARATE_DIGEST")) {
        // one table for all instances: digest it once (zero-state end state as a dot product, BiquadHpDesc); the
        // digest step was reserved right behind the coefficient step by emit_node_ops
        Step& hstep = b->steps[(size_t)o.i1 + 1];
        double* dhp = nullptr;
        int e = dev_alloc(b, &dhp, (size_t)b->n_tiles * HP_WORDS * 64);
        if (e) return e;
        hstep.hp.coefs = cstep.coef.coefs;
        hstep.hp.hp = dhp;
        hstep.hp.n_tiles = b->n_tiles;
        q.hp = dhp;
        q.vary = 3;
      }
    }
    q.state = reinterpret_cast<double*>(o.ptr1);
    q.n_gain = (int)(j - i - 1);
    for (size_t k = i + 1; k < j; k++) q.gain[k - i - 1] = ops[k].p0;
    q.nch = cur_nch;
    q.out = seg_out;
    q.dup_out = dup ? 1u : 0u;
    q.n_inst = b->n_inst;
    q.n_tiles = b->n_tiles;
    q.tile0 = 0;
    q.tile1 = b->n_tiles;
    q.n_quanta = b->n_quanta;
    st.profile_slot = slot_for(b, "biquad_stream_kernel");
    // constant coefficients: parallel in time as well (one unit per tile and stream, chained scan over the tiles of a stream)
    // (opt-in, WAA_BIQUAD_SCAN=1: measured on C2 the chained scan renders the batch in 2.45 ms against 1.4-1.6 ms for one
    // wavefront per stream — 78 % of its wave-cycles are waits on the unit's dependent round trips (dequeue, source record,
    // predecessor state, coefficient powers); kept as the cross-check of the hand-off protocol and for very long renders of
    // few streams, where the per-stream kernel cannot fill the chip)
    if (q.vary == 0 && !b->dry && measure_switch("WAA_BIQUAD_SCAN")) {
      const size_t n_streams = (size_t)b->n_inst * cur_nch;
      if (!b->scan_counter) {
        int e = dev_alloc(b, &b->scan_counter, 8 * 16 + 16);
        if (e) return e;
        b->state_bufs.push_back({b->scan_counter, (8 * 16 + 16) * sizeof(uint32_t)});
      }
      double *payload = nullptr, *pw = nullptr;
      int e;
      if ((e = dev_alloc(b, &payload, n_streams * b->n_tiles * 8)) || (e = dev_alloc(b, &pw, (size_t)b->n_inst * BIQUAD_SCAN_PW)))
        return e;
      b->ones_bufs.push_back({payload, n_streams * b->n_tiles * 8 * sizeof(double)});
      launch_biquad_scan_powers(q.coefs, q.coef_stride, pw, b->n_inst, b->stream);
      HIP_TRY(hipGetLastError());
      st.scan.counter = b->scan_counter;
      st.scan.error = b->scan_counter + 8 * 16;
      st.scan.payload = payload;
      st.scan.pw = pw;
      st.profile_slot = slot_for(b, "biquad_scan_kernel");
    }
    b->steps.push_back(st);
    if (dup) {
      plan_note(b, "biquad_stream in=%s:1ch gains=%d out=final (both channels: the speakers up-mix 1 -> 2 behind it)",
                input_kind_name(inputs[0].kind), q.n_gain);
      return 0;
    }
    plan_note(b, "biquad_stream%s in=%s:%dch gains=%d out=%s",
              q.vary == 3 ? "(a-rate, shared table)" : q.vary == 2 ? "(a-rate, per-instance table)" : q.vary ? "(k-rate)" : "",
              input_kind_name(inputs[0].kind),
              cur_nch, q.n_gain, seg_out.base == out.base ? "final" : "temp");
    InputRef in{};
    in.kind = IN_SIGNAL;
    in.nch = cur_nch;
    in.sig = seg_out;
    inputs.assign(1, in);
    in_nch = cur_nch;
    i = j;
    if (i == ops.size() && seg_out.base == out.base) return 0;  // the streaming kernel wrote the final output
  }
  // trailing element-wise ops (or a pure channel-count change into `out`)
  return push_chain_step(b, inputs, in_nch, in_interp, pending, out);
}

bool is_delay(const waa_batch* b, uint32_t id) { return b->nodes[id].desc.kind == WAA_NODE_DELAY; }
// outgoing edges of a vertex of the expanded graph, in insertion order
void vertex_targets(const waa_batch* b, uint32_t v, const std::vector<uint8_t>& cut, std::vector<uint32_t>& out) {
  const uint32_t id = v & ~VTX_READER;
  if (!(v & VTX_READER) && is_delay(b, id)) {
    if (!cut[id]) out.push_back(id | VTX_READER);
    return;
  }
  for (auto& e : b->edges) {
    if (e.from != id) continue;
    // inputs go to the writer half; delayTime belongs to the reader half (delay.rs:316-318)
    out.push_back(is_delay(b, e.to) && (e.to_input & 0x80000000u) ? (e.to | VTX_READER) : e.to);
  }
}

// Scheduled automation -> value blocks: every timeline is evaluated for all quanta of the render, in order
// (AudioParamProcessor::process calls compute_intrinsic_values(current_time, 1 / sample_rate, 128) once per
// quantum, param.rs:686-699), and stored like caller-provided blocks: runs of single-valued quanta as k-rate
// blocks, runs of 128-valued quanta as a-rate blocks.
int materialise_automation(waa_batch* b) {
  const double sample_rate = (double)b->sr, dt = 1. / sample_rate;
  for (Node& n : b->nodes)
    for (size_t pk = 0; pk < n.params.size(); pk++) {
      ParamStore& p = n.params[pk];
      if (p.timelines.empty()) continue;
      // identical timelines (every event scheduled for all instances, same initial value): evaluate once
      bool shared = p.timelines_shared && p.timelines[0];
      for (uint32_t i = 1; i < b->n_inst && shared; i++) shared = p.cst[i] == p.cst[0] && p.timelines[i];
      // Different event lists per instance on an a-rate param the DEVICE consumes: replay them there
      // (waa_timeline.hip) instead of evaluating n_inst x n_quanta x 128 values here and uploading them.  Params the
      // host needs per quantum (source playbackRate / detune: k-rate, the playhead replay) stay on this path.
      {
        const uint32_t kind = n.desc.kind;
        // (PannerNode: only the AudioListener's params of an equal-power panner go through the per-frame geometry
        // kernel; its own position / orientation and everything of an HRTF panner is geometry the host evaluates)
        const bool consumable = kind == WAA_NODE_GAIN || kind == WAA_NODE_BIQUAD || kind == WAA_NODE_DELAY ||
                                kind == WAA_NODE_STEREO_PANNER || kind == WAA_NODE_CONSTANT_SOURCE ||
                                kind == WAA_NODE_OSCILLATOR ||
                                (kind == WAA_NODE_PANNER && pk >= 6 && n.desc.i[0] != WAA_PANNING_HRTF);
        const bool want = measure_switch("WAA_DEVICE_AUTOMATION") ? true : !shared;  // (switch: also replay shared timelines there)
        bool arrivals = false;  // (events that arrive at a suspend point: the host's replay applies them in front of their quantum)
        for (const auto& tl : p.timelines) arrivals |= tl && tl->has_arrivals();
        if (consumable && !p.k_rate && want && !arrivals && !b->dry && !measure_switch("WAA_HOST_AUTOMATION")) {
          p.dev_tl = true;
          continue;  // (the timelines are consumed when the param is uploaded)
        }
      }
      for (uint32_t inst = 0; inst < (shared ? 1u : b->n_inst); inst++) {
        Timeline* tl = p.timelines[inst].get();
        if (!tl) continue;
        // (appended after caller-provided blocks: the scheduled automation wins where both exist)
        float buf[RQ];
        ParamBlock run;
        run.inst = shared ? WAA_ALL_INSTANCES : inst;
        run.nq = 0;
        auto flush = [&] {
          if (run.nq) p.blocks.push_back(run);
          run.nq = 0;
          run.v.clear();
        };
        for (uint32_t q = 0; q < b->n_quanta; q++) {
          const double block_time = (double)((uint64_t)q * RQ) / sample_rate;
          if (int ea = tl->apply_arrivals(q)) return ea;  // (control messages of a suspend point in front of this quantum)
          const uint32_t len = tl->compute(block_time, dt, RQ, buf);
          if (run.nq && run.vpq != len) flush();
          if (!run.nq) {
            run.q0 = q;
            run.vpq = len;
          }
          run.v.insert(run.v.end(), buf, buf + len);
          run.nq++;
        }
        flush();
      }
      p.timelines.clear();  // consumed (a batch renders one timeline once, like an OfflineAudioContext)
    }
  return 0;
}

// Host-known codes (count | CODE_SILENT per quantum, [n_inst][cs]) of a source node: active quanta carry the source's
// channel count (the buffer source's active quanta come from the scheduling replay).
int source_code_rows(waa_batch* b, uint32_t id, uint64_t cs, std::vector<uint8_t>* out) {
  Node& n = b->nodes[id];
  std::vector<uint8_t>& host = *out;
  host.assign((size_t)b->n_inst * cs, (uint8_t)(1u | CODE_SILENT));
  const double sample_rate = (double)b->sr, dt = 1. / sample_rate;
  std::map<SchedKey, std::vector<uint8_t>> cache;
  for (uint32_t i = 0; i < b->n_inst; i++) {
    uint8_t* row = host.data() + (size_t)i * cs;
    const SourceSched& ss = n.sched[i];
    if (n.desc.kind == WAA_NODE_BUFFER_SOURCE) {
      const DeviceBuffer& bf = n.bufs[i];
      const ParamStore& p_rate = n.params[WAA_PARAM_SOURCE_PLAYBACK_RATE];
      const ParamStore& p_det = n.params[WAA_PARAM_SOURCE_DETUNE];
      const bool automated = !p_rate.blocks.empty() || !p_det.blocks.empty();
      std::vector<float> rate_q = param_per_quantum(b, p_rate, i, nullptr);
      std::vector<float> det_q = param_per_quantum(b, p_det, i, nullptr);
      const SchedKey key(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end,
                         bf.valid ? bf.frames : 0, bf.valid ? bf.sr : 0.f, rate_q[0], det_q[0]);
      // (rows differ by the buffer's own channel count too: one cache per count would do; mixed counts are rare -> no cache)
      auto itc = automated || bf.nch_true ? cache.end() : cache.find(key);
      if (itc == cache.end()) {
        SchedOut own;
        std::shared_ptr<SchedOut> shared;
        if (automated)
          schedule_source(b, ss, bf.frames, bf.sr, bf.valid, rate_q, det_q, &own);
        else
          shared = schedule_source_cached(b, id, key, ss, bf.frames, bf.sr, bf.valid, rate_q, det_q);
        const SchedOut& so = automated ? own : *shared;
        std::vector<uint8_t> r(b->n_quanta);
        for (uint32_t q = 0; q < b->n_quanta; q++)
          r[q] = so.qrec[q].mode == Q_SILENT ? (uint8_t)(1u | CODE_SILENT) : (uint8_t)(bf.valid ? bf.count() : (uint32_t)n.out_nch);
        itc = cache.emplace(automated || bf.nch_true ? SchedKey(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end,
                                                 (uint64_t)i, -1.f, 0.f, 0.f)
                                      : key,
                            std::move(r)).first;
      }
      std::copy(itc->second.begin(), itc->second.end(), row);
    } else {
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        const double ct = (double)((uint64_t)q * RQ) / sample_rate, nbt = ct + dt * (double)RQ;
        bool silent;
        if (n.desc.kind == WAA_NODE_CONSTANT_SOURCE)
          silent = ss.start >= nbt;                    // constant_source.rs:203-216: silent until it starts, never again
        else
          silent = ss.stop <= ct || ss.start >= nbt;   // oscillator.rs:382-404
        row[q] = silent ? (uint8_t)(1u | CODE_SILENT) : (uint8_t)1u;
      }
    }
  }
  return 0;
}

static int build_plan_impl(waa_batch* b);
static int build_plan_rest(waa_batch* b, std::vector<uint32_t>& items, std::vector<uint8_t>& muted);
int build_plan(waa_batch* b) {
  b->sched_cache.clear();
  const int e = build_plan_impl(b);
  b->sched_cache.clear();  // (the replays of one plan: a 10 s slow-track table is 7.7 MB)
  return e;
}

static int build_plan_impl(waa_batch* b) {
  const uint32_t N = (uint32_t)b->nodes.size();
  b->short_ring_loops.clear();
  PlanTrace trace_total("build_plan (host + device calls)");
  std::unique_ptr<PlanTrace> ph(new PlanTrace("phase: automation + order + loops + counts"));
  if (int e = materialise_automation(b)) return e;
  for (uint32_t i = 0; i < N; i++)  // the reference takes the coefficients in the constructor
    if (b->nodes[i].desc.kind == WAA_NODE_IIR_FILTER && b->nodes[i].iir_b.empty())
      return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - IIRFilterNode %u has no coefficients", i);
  // processing order = reversed DFS post-order over outgoing edges in insertion order, cycle breakers applied
  // (graph.rs:323-487); `items` has two entries per DelayNode (writer, reader), none for muted nodes
  std::vector<uint32_t> items;
  std::vector<uint8_t> muted(N, 0);
  compute_order(b, &b->cut, &muted, &items);
  // one entry per node, at the position where its OUTPUT is produced (the reader half of a DelayNode)
  b->order.clear();
  for (uint32_t v : items)
    if (!is_delay(b, v & ~VTX_READER) || (v & VTX_READER)) b->order.push_back(v & ~VTX_READER);
  for (uint32_t i = 0; i < N; i++)
    if (muted[i]) plan_note(b, "node %u is part of a cycle without a DelayNode: muted (graph.rs:362-368)", i);
  return build_plan_rest(b, items, muted);
}

// graph.rs:323-487 for the batch's current nodes and edges: which DelayNodes lose their writer->reader edge (`cut`), which nodes
// sit in a cycle without one (`muted`), and the render order (two entries per DelayNode: writer, reader; none for muted nodes)
void compute_order(const waa_batch* b, std::vector<uint8_t>* cut_out, std::vector<uint8_t>* muted_out, std::vector<uint32_t>* items_out) {
  const uint32_t N = (uint32_t)b->nodes.size();
  std::vector<uint32_t>& items = *items_out;
  std::vector<uint8_t>& muted = *muted_out;
  items.clear();
  muted.assign(N, 0);
  {
    OrderCtx c;
    c.b = b;
    c.cut.assign(N, 0);
    for (;;) {
      c.marked.clear();
      c.temp.clear();
      c.ordered.clear();
      c.in_cycle.clear();
      bool applied = false;
      for (uint32_t i = 0; i < N && !applied; i++) {
        applied = order_visit(c, i);
        if (!applied && is_delay(b, i)) applied = order_visit(c, i | VTX_READER);
        // the AudioListener is the reference's graph node 1, right behind the destination, with an edge to every
        // PannerNode in creation order (concrete_base.rs:511-534): visited as a DFS root it pulls the panner branches
        // to the front of the traversal, i.e. to the back of the reversed post-order — which decides the f32 summing
        // order wherever three or more signals meet
        if (i == 0)
          for (uint32_t pn = 0; pn < N && !applied; pn++)
            if (b->nodes[pn].desc.kind == WAA_NODE_PANNER) applied = order_visit(c, pn);
      }
      if (!applied) break;
      c.cut[c.breaker] = 1;
    }
    for (uint32_t v : c.in_cycle) muted[v & ~VTX_READER] = 1;
    for (auto it = c.ordered.rbegin(); it != c.ordered.rend(); ++it)
      if (!muted[*it & ~VTX_READER]) items.push_back(*it);
    *cut_out = c.cut;
  }
}

static int build_plan_rest(waa_batch* b, std::vector<uint32_t>& items, std::vector<uint8_t>& muted) {
  const uint32_t N = (uint32_t)b->nodes.size();
  std::unique_ptr<PlanTrace> ph(new PlanTrace("phase: loops + counts"));
  // feedback loops: strongly connected components of the graph with the writer->reader edges in place
  // (Tarjan); their members are rendered quantum by quantum by the loop kernel
  std::vector<int> scc_of(N, -1);
  int n_scc = 0;
  {
    const uint32_t V = 2 * N;
    auto vidx = [&](uint32_t v) { return (v & VTX_READER) ? N + (v & ~VTX_READER) : v; };
    std::vector<int> index(V, -1), low(V, 0), comp(V, -1);
    std::vector<uint8_t> on(V, 0);
    std::vector<uint32_t> stk;
    int counter = 0, n_comp = 0;
    const std::vector<uint8_t> no_cut(N, 0);
    std::vector<int> comp_size;
    std::function<void(uint32_t)> strong = [&](uint32_t v) {
      const uint32_t vi = vidx(v);
      index[vi] = low[vi] = counter++;
      stk.push_back(v);
      on[vi] = 1;
      std::vector<uint32_t> ts;
      vertex_targets(b, v, no_cut, ts);
      for (uint32_t t : ts) {
        if (muted[t & ~VTX_READER]) continue;
        const uint32_t ti = vidx(t);
        if (index[ti] < 0) {
          strong(t);
          low[vi] = std::min(low[vi], low[ti]);
        } else if (on[ti]) {
          low[vi] = std::min(low[vi], index[ti]);
        }
      }
      if (low[vi] == index[vi]) {
        int size = 0;
        for (;;) {
          const uint32_t w = stk.back();
          stk.pop_back();
          on[vidx(w)] = 0;
          comp[vidx(w)] = n_comp;
          size++;
          if (w == v) break;
        }
        comp_size.push_back(size);
        n_comp++;
      }
    };
    for (uint32_t i = 0; i < N; i++) {
      if (muted[i]) continue;
      if (index[i] < 0) strong(i);
      if (is_delay(b, i) && index[N + i] < 0) strong(i | VTX_READER);
    }
    std::map<int, int> renum;
    for (uint32_t i = 0; i < N; i++) {
      if (muted[i] || comp[i] < 0 || comp_size[comp[i]] < 2) continue;
      auto it = renum.find(comp[i]);
      if (it == renum.end()) it = renum.emplace(comp[i], n_scc++).first;
      scc_of[i] = it->second;
    }
  }
  std::vector<uint32_t> pos(N, 0xffffffffu);
  for (uint32_t i = 0; i < b->order.size(); i++) pos[b->order[i]] = i;
  // incoming edges in summing order: by processing position of the producer, then edge insertion order
  for (auto& n : b->nodes) {
    n.in_edges.clear();
    n.pin_edges.assign(n.params.size(), {});
    n.pin_ref.assign(n.params.size(), ParamRef{});
    n.pin_ready.assign(n.params.size(), 0);
    n.n_consumers = 0;
    n.live = n.materialized = false;
  }
  for (uint32_t e = 0; e < b->edges.size(); e++) {
    const waa_edge_desc& ed = b->edges[e];
    Node& to = b->nodes[ed.to];
    if (muted[ed.from] || muted[ed.to]) continue;  // a muted node renders nothing and contributes nothing
    if (ed.to_input & 0x80000000u) {
      const uint32_t pid = ed.to_input & 0x7fffffffu;
      if (pid >= to.params.size()) return fail(WAA_ERR_INVALID_ARGUMENT, "IndexSizeError - node %u has no param %u", ed.to, pid);
      const uint32_t k = to.desc.kind;
      const bool source_rate = k == WAA_NODE_BUFFER_SOURCE && (pid == WAA_PARAM_SOURCE_PLAYBACK_RATE || pid == WAA_PARAM_SOURCE_DETUNE);
      if (source_rate && !b->prepass)  // (with a device these edges are resolved before the plan is built, waa_abi.cpp)
        return fail(WAA_ERR_OUT_OF_SCOPE,
                    "playbackRate / detune of source node %u are modulated from the graph: the modulator is rendered at plan time, "
                    "which needs the device (plan-only batch)", ed.to);
      // PannerNode position / orientation with a single-valued AudioListener: the reference takes the FIRST value of every
      // param per quantum (panner.rs:833-846; HRTF: :781-829) — the same plan-time resolution (round 4).  Edges that are still
      // here outside the prepass were not resolved: an audio-rate listener next to them, or a plan-only batch.
      const bool panner_geom = k == WAA_NODE_PANNER && pid <= WAA_PARAM_PANNER_ORIENTATION_Z;
      if (panner_geom && !b->prepass)
        return fail(WAA_ERR_OUT_OF_SCOPE,
                    "position / orientation of panner node %u are modulated from the graph: resolved at plan time on the device for a "
                    "single-valued AudioListener only (audio-rate listener automation next to it, or a plan-only batch)", ed.to);
      if (!(k == WAA_NODE_GAIN || k == WAA_NODE_BIQUAD || k == WAA_NODE_DELAY || k == WAA_NODE_STEREO_PANNER ||
            k == WAA_NODE_CONSTANT_SOURCE || k == WAA_NODE_OSCILLATOR || source_rate || panner_geom))
        return fail(WAA_ERR_OUT_OF_SCOPE, "audio-rate modulation of a host-evaluated param (node %u) is out of scope", ed.to);
      to.pin_edges[pid].push_back((int)e);
    } else {
      to.in_edges.push_back((int)e);
    }
    b->nodes[ed.from].n_consumers++;
  }
  auto by_position = [&](int x, int y) { return pos[b->edges[x].from] < pos[b->edges[y].from]; };
  for (auto& n : b->nodes) {
    std::stable_sort(n.in_edges.begin(), n.in_edges.end(), by_position);
    for (auto& pe : n.pin_edges) std::stable_sort(pe.begin(), pe.end(), by_position);
  }
  // liveness: everything that reaches the destination or an analyser
  {
    std::vector<uint32_t> stack;
    for (uint32_t i = 0; i < N; i++)
      if (!b->prepass && (b->nodes[i].desc.kind == WAA_NODE_DESTINATION || b->nodes[i].desc.kind == WAA_NODE_ANALYSER)) stack.push_back(i);
    // (prepass: only what feeds the graph-modulated playbackRate / detune params)
    if (b->prepass)
      for (auto& pp : b->prepass_params) {
        // only what feeds the modulated PARAM: a PannerNode's audio input is not part of the prepass
        b->nodes[pp.first].live = true;
        for (int e : b->nodes[pp.first].pin_edges[pp.second]) stack.push_back(b->edges[e].from);
      }
    while (!stack.empty()) {
      uint32_t id = stack.back();
      stack.pop_back();
      if (b->nodes[id].live) continue;
      b->nodes[id].live = true;
      for (int e : b->nodes[id].in_edges) stack.push_back(b->edges[e].from);
      for (auto& pe : b->nodes[id].pin_edges)
        for (int e : pe) stack.push_back(b->edges[e].from);
    }
  }
  // (prepass) a modulated source that itself feeds the modulating subgraph of a modulated source: its own schedule is
  // not known before ITS modulation has been resolved — a second prepass level nobody has asked for yet; refused loudly
  // (it used to be skipped by plan_single and the outer param chain read an empty signal)
  if (b->prepass)
    for (auto& ed : b->edges) {
      if (ed.from >= N || ed.to >= N || !b->nodes[ed.to].live) continue;
      for (auto& pp : b->prepass_params)
        if (pp.first == ed.from)
          return fail(WAA_ERR_OUT_OF_SCOPE, "source node %u has a graph-modulated playbackRate / detune and feeds the modulation of source node %u: nested modulation of source rates is out of scope", ed.from, ed.to);
    }
  // static channel counts (the reference's counts are dynamic: a silent quantum is mono; every case the
  // static count differs from the dynamic one carries zeros — see DESIGN.md "Silence and channel counts")
  // (inside a feedback loop a producer can come later in the order: iterate to the fixed point; counts only grow)
  for (auto& n : b->nodes) n.in_nch = n.out_nch = 1;
  bool mixed_buffer_counts = false;
  for (int pass = 0; pass < 16; pass++) {
  bool changed = false;
  for (uint32_t id : b->order) {
    Node& n = b->nodes[id];
    if (!n.live) continue;
    const int old_in = n.in_nch, old_out = n.out_nch;
    int maxc = 1;
    for (int e : n.in_edges) maxc = std::max(maxc, b->nodes[b->edges[e].from].out_nch);
    n.in_nch = computed_in_nch(n, maxc);
    switch (n.desc.kind) {
      case WAA_NODE_BUFFER_SOURCE: {
        // instances may play AudioBuffers of different channel counts: the signal is laid out for the widest one, the
        // per-quantum codes of the dynamic-count plan carry every instance's own count (below: widen_narrow_buffers)
        uint32_t nch = 0;
        for (auto& bf : n.bufs)
          if (bf.valid) {
            if (nch && bf.count() != nch) mixed_buffer_counts = true;
            nch = std::max(nch, bf.count());
          }
        n.out_nch = nch ? (int)nch : 1;
        break;
      }
      case WAA_NODE_CONSTANT_SOURCE:
      case WAA_NODE_OSCILLATOR: n.out_nch = 1; break;
      case WAA_NODE_STEREO_PANNER:
      case WAA_NODE_PANNER: n.out_nch = 2; break;
      case WAA_NODE_CONVOLVER:
        if (!n.has_ir)
          n.out_nch = n.in_nch;
        else
          n.out_nch = (n.in_nch == 1 && n.ir_nch == 1) ? 1 : 2;
        break;
      default: n.out_nch = n.in_nch; break;
    }
    if (n.in_nch > WAA_MAX_CHANNELS || n.out_nch > WAA_MAX_CHANNELS)
      return fail(WAA_ERR_OUT_OF_SCOPE, "at most %d channels per signal (node %u needs %d)", WAA_MAX_CHANNELS, id, std::max(n.in_nch, n.out_nch));
    changed |= n.in_nch != old_in || n.out_nch != old_out;
  }
  if (!changed || n_scc == 0) break;
  }
  if (mixed_buffer_counts) {
    // buffer.rs / audio_buffer_source.rs:560-600: the source's output has the channel count of ITS buffer; with several
    // counts in one batch no static count fits every instance -> the dynamic-count plan, whose codes are per instance
    if (measure_switch("WAA_STATIC_CHANNEL_COUNTS"))
      return fail(WAA_ERR_OUT_OF_SCOPE, "instances of one batch must use AudioBuffers with the same channel count (WAA_STATIC_CHANNEL_COUNTS)");
    if (b->prepass)
      return fail(WAA_ERR_OUT_OF_SCOPE, "the graph that modulates a host-evaluated param plays AudioBuffers of different channel counts across instances: out of scope");
    for (auto& n : b->nodes)
      if (n.live && n.desc.kind == WAA_NODE_BUFFER_SOURCE)
        if (int e = widen_narrow_buffers(b, n, (uint32_t)n.out_nch)) return e;
    b->force_dynamic = true;
  }
  ph.reset();
  ph.reset(new PlanTrace("phase: silence / count replay"));
  // Static vs dynamic channel counts.  The reference counts a silent input as mono, so the channel count of a
  // signal changes mid-render when a narrow and a wide producer are not active over the same quanta, when a source
  // ends, when a Gain is (at times) zero.  Count-sensitive nodes then differ from this plan, which renders the
  // static (maximal) count throughout: filters keep per-channel state, the panners use another law for mono input,
  // the convolver routes by count, a DelayNode re-mixes its whole line to the count of the current input, layouts
  // above stereo and the discrete interpretation do not commute with an earlier speakers up-mix.  The planner
  // replays the reference's silence / count propagation per quantum from what the host knows (source windows,
  // delay times, zero gains; data-dependent filter tails are bounded from both sides) and reports the
  // first affected node in the plan; WAA_STRICT_CHANNEL_COUNTS turns the note into status 4.
  bool count_change_found = false;
  {
    const uint32_t nq = b->n_quanta;
    const double qsec = (double)RQ / (double)b->sr;
    std::vector<std::vector<uint8_t>> act(N, std::vector<uint8_t>(nq)), cnt(N, std::vector<uint8_t>(nq));
    std::vector<std::vector<uint8_t>> in_act(N, std::vector<uint8_t>(nq)), in_cnt(N, std::vector<uint8_t>(nq));
    std::map<std::string, bool> seen;  // instances with identical host-known inputs are simulated once
    bool reported = false;
    std::map<std::pair<uint32_t, SchedKey>, int64_t> end_cache;  // (source node, schedule) -> quantum of its end
    for (uint32_t inst = 0; inst < b->n_inst && !reported; inst++) {
      // per-instance inputs of the simulation
      std::vector<double> lo(N, 1e300), hi(N, -1.), shift(N, 0.), shift_hi(N, 0.);
      std::vector<std::vector<uint8_t>> zero_gain(N);
      std::string sig;
      auto add_sig = [&](double v) { sig.append(reinterpret_cast<const char*>(&v), sizeof v); };
      for (uint32_t id : b->order) {
        Node& n = b->nodes[id];
        if (!n.live) continue;
        const uint32_t kind = n.desc.kind;
        if (kind == WAA_NODE_BUFFER_SOURCE || kind == WAA_NODE_CONSTANT_SOURCE || kind == WAA_NODE_OSCILLATOR) {
          const SourceSched& ss = n.sched[inst];
          lo[id] = ss.start == DBL_MAX ? 1e300 : std::floor(ss.start / qsec);
          hi[id] = ss.stop != DBL_MAX ? std::ceil(ss.stop / qsec) : 1e300;
          // a ConstantSourceNode is silent until it starts and NEVER again: past its stop time it keeps rendering zeros
          // into a non-silent mono quantum (constant_source.rs:203-258) — a narrower-than-static input for whatever
          // count-sensitive node it feeds together with a wider source that has ended (fuzz seed 1579), yet nothing
          // but zeros for the nodes whose silence is data dependent (a DelayNode behind it falls silent: seed 3071).
          // Both readings are simulated (const_forever below).
          if (kind == WAA_NODE_BUFFER_SOURCE && n.bufs[inst].valid) {
            // the quantum in which the source really ends: the scheduling replay with this instance's playbackRate
            // and detune per quantum (an automated rate moves the end; an estimate from the slowest rate claimed the
            // source active — and the signal stereo — for longer than the reference renders it)
            const DeviceBuffer& bf = n.bufs[inst];
            const ParamStore& p_rate = n.params[WAA_PARAM_SOURCE_PLAYBACK_RATE];
            const ParamStore& p_det = n.params[WAA_PARAM_SOURCE_DETUNE];
            const bool automated = !p_rate.blocks.empty() || !p_det.blocks.empty();
            // (constant params: the key needs their one value; the per-quantum vectors only for a replay)
            const SchedKey key(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end, bf.frames, bf.sr,
                               p_rate.fix(p_rate.cst[inst]), p_det.fix(p_det.cst[inst]));
            int64_t endq;
            auto it = automated ? end_cache.end() : end_cache.find(std::make_pair(id, key));
            if (it != end_cache.end()) {
              endq = it->second;
            } else {
              std::vector<float> rate_q = param_per_quantum(b, p_rate, inst, nullptr);
              std::vector<float> det_q = param_per_quantum(b, p_det, inst, nullptr);
              if (automated) {
                SchedOut so;
                schedule_source(b, ss, bf.frames, bf.sr, true, rate_q, det_q, &so);
                endq = so.ended_quantum;
              } else {
                endq = schedule_source_cached(b, id, key, ss, bf.frames, bf.sr, true, rate_q, det_q)->ended_quantum;
                end_cache[std::make_pair(id, key)] = endq;
              }
            }
            if (endq >= 0) hi[id] = std::min(hi[id], (double)endq + 1.);
          }
          if (kind == WAA_NODE_BUFFER_SOURCE && !n.bufs[inst].valid) lo[id] = 1e300;
          add_sig(lo[id]);
          add_sig(hi[id]);
        } else if (kind == WAA_NODE_DELAY) {
          {
            // a delay that is not a whole number of quanta: whether the first delayed samples land in this quantum or
            // the next depends on where inside its quantum the input started — both are simulated.  A delayTime that
            // varies (k-rate or a-rate automation) is simulated at its shortest and at its longest value; one that is
            // modulated by the graph is only known to lie in [0, maxDelayTime].
            const ParamStore& pd = n.params[WAA_PARAM_DELAY_DELAY_TIME];
            double dmin = (double)pd.fix(pd.cst[inst]), dmax = dmin;
            bool covered_all = false;
            for (auto& blk : pd.blocks) {
              if (!(blk.inst == WAA_ALL_INSTANCES || blk.inst == inst)) continue;
              for (float dv : blk.v) {
                dmin = std::min(dmin, (double)pd.fix(dv));
                dmax = std::max(dmax, (double)pd.fix(dv));
              }
              covered_all |= blk.q0 == 0 && blk.nq >= nq;
            }
            (void)covered_all;
            const bool modulated = (WAA_PARAM_DELAY_DELAY_TIME < n.pin_edges.size() && !n.pin_edges[WAA_PARAM_DELAY_DELAY_TIME].empty()) ||
                                   pd.dev_tl;  // (replayed on the device: only known to lie in [0, maxDelayTime])
            if (modulated) {
              dmin = 0.;
              dmax = n.desc.d[0];
            }
            shift[id] = std::floor(dmin / qsec);
            shift_hi[id] = std::ceil(dmax / qsec);
          }
          if (id < b->cut.size() && b->cut[id]) {  // inside a loop: >= one quantum
            shift[id] = std::max(shift[id], 1.);
            shift_hi[id] = std::max(shift_hi[id], 1.);
          }
          add_sig(shift[id]);
          add_sig(shift_hi[id]);
        } else if (kind == WAA_NODE_GAIN && param_mode(n, 0) != 2) {
          const auto gv = param_per_quantum(b, n.params[0], inst, nullptr);
          bool any = false;
          if (gv.size() == 1) {
            // one value for the whole render (the usual case): no per-quantum table unless that value is a zero —
            // filling and scanning n_quanta entries per instance and GainNode was 7.7 ms of a 1024-context plan
            any = std::fabs(gv[0]) <= 1e-6f;
            if (any) zero_gain[id].assign(nq, 1);
          } else {
            zero_gain[id].assign(nq, 0);
            for (uint32_t q = 0; q < nq; q++) {
              zero_gain[id][q] = std::fabs(gv[q]) <= 1e-6f;
              any |= zero_gain[id][q] != 0;
            }
          }
          if (!any) zero_gain[id].clear();
          for (uint8_t z : zero_gain[id]) sig.push_back((char)z);
          sig.push_back('|');
        }
      }
      if (seen.count(sig)) continue;
      seen[sig] = true;
      // Tails are data dependent (a filter renders until its state has decayed).  Both bounds are simulated per node
      // with memory: it falls silent with its input (tail 0), or never again once it was active (tail infinite); a
      // finding under any combination is reported.
      std::vector<uint32_t> mem_nodes;
      for (uint32_t id : b->order) {
        const Node& n = b->nodes[id];
        const uint32_t k = n.desc.kind;
        if (n.live && (k == WAA_NODE_BIQUAD || k == WAA_NODE_IIR_FILTER || k == WAA_NODE_DELAY || (k == WAA_NODE_CONVOLVER && n.has_ir) ||
                       (k == WAA_NODE_PANNER && n.desc.i[0] == WAA_PANNING_HRTF)))
          mem_nodes.push_back(id);
      }
      // every combination of short / long tails for up to 8 nodes with memory, the two uniform bounds beyond that
      const uint32_t n_modes = mem_nodes.size() <= 8 ? (1u << mem_nodes.size()) : 2u;
      std::vector<uint8_t> long_tail(N, 0);
      bool stopped_constant = false;
      for (uint32_t id : b->order)
        stopped_constant |= b->nodes[id].live && b->nodes[id].desc.kind == WAA_NODE_CONSTANT_SOURCE && hi[id] < 1e299;
      for (int const_forever = 0; const_forever < (stopped_constant ? 2 : 1) && !reported; const_forever++)
      for (uint32_t mode_index = 0; mode_index < 2 * n_modes && !reported; mode_index++) {
      const uint32_t tail_mode = mode_index >> 1;
      const std::vector<double>& dshift = (mode_index & 1) ? shift_hi : shift;
      if ((mode_index & 1) && shift_hi == shift) continue;
      for (size_t k = 0; k < mem_nodes.size(); k++)
        long_tail[mem_nodes[k]] = mem_nodes.size() <= 8 ? ((tail_mode >> k) & 1u) : (uint8_t)tail_mode;
      for (uint32_t id = 0; id < N; id++) {
        std::fill(act[id].begin(), act[id].end(), 0);
        std::fill(cnt[id].begin(), cnt[id].end(), 1);
      }
      const int passes = n_scc > 0 ? 4 : 1;  // activity travels once around a feedback loop per pass
      for (int pass = 0; pass < passes; pass++)
        for (uint32_t id : b->order) {
          Node& n = b->nodes[id];
          if (!n.live) continue;
          const uint32_t kind = n.desc.kind;
          if (kind == WAA_NODE_BUFFER_SOURCE || kind == WAA_NODE_CONSTANT_SOURCE || kind == WAA_NODE_OSCILLATOR) {
            for (uint32_t q = 0; q < nq; q++) {
              act[id][q] = (double)q >= lo[id] && ((double)q < hi[id] || (const_forever && kind == WAA_NODE_CONSTANT_SOURCE));
              cnt[id][q] = act[id][q] ? (uint8_t)n.out_nch : 1;
            }
            continue;
          }
          for (uint32_t q = 0; q < nq; q++) {
            uint8_t a = 0, c = 1;
            for (int e : n.in_edges) {
              const uint32_t p = b->edges[e].from;
              a |= act[p][q];
              c = std::max(c, cnt[p][q]);
            }
            in_act[id][q] = a;
            in_cnt[id][q] = (uint8_t)computed_in_nch(n, c);
          }
          for (uint32_t q = 0; q < nq; q++) {
            uint8_t a = in_act[id][q], c = in_cnt[id][q];
            if (kind == WAA_NODE_DELAY) {
              // the data is `shift` quanta old; the channel count is the line's, which follows the writer's CURRENT
              // input (delay.rs:469-489) — of the previous quantum when the reader renders first (inside a loop)
              const int64_t qs = (int64_t)q - (int64_t)dshift[id];
              a = qs >= 0 ? in_act[id][qs] : 0;
              const bool in_loop = id < b->cut.size() && b->cut[id];
              const int64_t qc = in_loop ? (int64_t)q - 1 : (int64_t)q;
              c = qc >= 0 ? in_cnt[id][qc] : 1;
            } else if (kind == WAA_NODE_GAIN && !zero_gain[id].empty() && zero_gain[id][q]) {
              a = 0;
            } else if (kind == WAA_NODE_WAVESHAPER && n.has_curve && !a) {
              // a curve that does not map 0 to 0 turns silence into a signal (waveshaper.rs:498-509): active, mono
              const size_t cn = n.curve.size();
              const float mid = cn == 0 ? 0.f : (cn % 2 ? n.curve[cn / 2] : (n.curve[cn / 2 - 1] + n.curve[cn / 2]) / 2.f);
              if (!(std::fabs(mid) < 1e-9f)) a = 1;
            }
            if (a) {
              if (kind == WAA_NODE_STEREO_PANNER || kind == WAA_NODE_PANNER) c = 2;
              if (kind == WAA_NODE_CONVOLVER && n.has_ir) c = (c == 1 && n.ir_nch == 1) ? 1 : 2;
            }
            const bool has_memory = kind == WAA_NODE_BIQUAD || kind == WAA_NODE_IIR_FILTER || kind == WAA_NODE_DELAY ||
                                    (kind == WAA_NODE_CONVOLVER && n.has_ir) ||
                                    (kind == WAA_NODE_PANNER && n.desc.i[0] == WAA_PANNING_HRTF);  // (HRIR tail)
            if (has_memory && long_tail[id] && !a && q > 0 && act[id][q - 1]) {  // still ringing, with the old layout
              a = 1;
              c = cnt[id][q - 1];
            }
            act[id][q] = a;
            cnt[id][q] = a ? c : 1;
          }
        }
      // A GainNode whose `gain` has an audio-rate input that is SILENT in some quantum: the param is then one value for that quantum
      // (param.rs:737-760) and gain.rs:163-179's fast paths apply — a value of 0 emits the silent block (ONE channel), where the
      // modulated form of the kernels multiplies by a per-frame table of zeros and keeps the static layout.  Only the zero matters (a
      // value of 1 is the same samples either way); rendering it differently from the reference is not on offer: status 4 (suspend
      // fuzz seed 90491, round 6: an LFO stopped at a suspend point in front of a k-rate gain that visits 0).
      for (uint32_t id : b->order) {
        const Node& n = b->nodes[id];
        if (!n.live || n.desc.kind != WAA_NODE_GAIN || n.pin_edges.empty() || n.pin_edges[0].empty() || in_act[id].empty()) continue;
        const auto gv = param_per_quantum(b, n.params[0], inst, nullptr);
        for (uint32_t q = 0; q < nq; q++) {
          if (!(std::fabs(gv[gv.size() == 1 ? 0 : q]) <= 1e-6f) || !in_act[id][q]) continue;
          bool mod_active = false;
          for (int e : n.pin_edges[0]) mod_active |= act[b->edges[e].from][q] != 0;
          if (!mod_active)
            return fail(WAA_ERR_OUT_OF_SCOPE,
                        "gain node %u: its gain is 0 in quantum %u (instance %u) while the audio-rate input of the param is silent — the reference emits "
                        "a silent (one-channel) quantum there (gain.rs:163-171), the modulated gain kernels do not: out of scope",
                        id, q, inst);
        }
      }
      // findings
      for (uint32_t id : b->order) {
        Node& n = b->nodes[id];
        if (!n.live || n.in_edges.empty()) continue;
        {
          bool wide_producer = false;
          for (int e : n.in_edges) wide_producer |= b->nodes[b->edges[e].from].out_nch > 2;
          if (n.in_nch <= 1 && !wide_producer) continue;
        }
        const uint32_t kind = n.desc.kind;
        const bool line = kind == WAA_NODE_DELAY || (kind == WAA_NODE_CONVOLVER && n.has_ir);
        bool sensitive = kind == WAA_NODE_BIQUAD || kind == WAA_NODE_IIR_FILTER || kind == WAA_NODE_STEREO_PANNER ||
                         kind == WAA_NODE_PANNER || line || n.in_nch > 2 || n.interp == WAA_INTERP_DISCRETE;
        // (a producer wider than stereo into a mono / stereo node: its DOWN-mix does not commute with the up-mix the static plan made
        // upstream either — a 5.1 signal that is in fact mono goes to L and R as it is, its static 5.1 form through the 5.1 -> stereo
        // matrix; round 6, wide fuzz seed 293: a WaveShaper with curve(0) != 0 on a silent 5.1 input in front of a stereo destination)
        for (int e : n.in_edges) sensitive |= b->nodes[b->edges[e].from].out_nch > 2;
        if (!sensitive) continue;
        // a DelayNode up-mixes its line by copying when the wide signal arrives (= the static plan) but collapses it
        // when the input narrows or falls silent while the line still holds wide material
        const bool narrowing_only = kind == WAA_NODE_DELAY && n.in_nch <= 2 && n.interp != WAA_INTERP_DISCRETE;
        int64_t last_wide = -1;
        const int64_t memory_q = kind == WAA_NODE_DELAY ? (int64_t)std::ceil(n.desc.d[0] / qsec) + 1
                                 : kind == WAA_NODE_CONVOLVER ? (int64_t)(n.ir_len / RQ) + 2 : 0;
        const char* what = nullptr;
        uint32_t at = 0;
        for (uint32_t q = 0; q < nq && !what; q++) {
          const bool wide_now = in_act[id][q] && in_cnt[id][q] >= n.in_nch;
          if (in_act[id][q] && !wide_now && !(narrowing_only && last_wide < 0)) {
            if (!narrowing_only || (int64_t)q - last_wide <= memory_q) {
              what = "is narrower than its static channel count";
              at = q;
            }
          }
          if (line && !in_act[id][q] && last_wide >= 0 && (int64_t)q - last_wide <= memory_q && (int64_t)q - last_wide >= 1) {
            what = "falls silent (= mono) while the node still holds multi-channel material";
            at = q;
          }
          if (wide_now) last_wide = q;
        }
        // mixing rules that do not commute with the speakers up-mix the static plan applied upstream: an input that
        // is active but narrower than its static width, into a discrete or wider-than-stereo mix
        if (!what)
          for (int e : n.in_edges) {
            const uint32_t p = b->edges[e].from;
            const int pw = b->nodes[p].out_nch;
            if (!(n.interp == WAA_INTERP_DISCRETE || n.in_nch > 2 || pw > 2)) continue;
            for (uint32_t q = 0; q < nq && !what; q++)
              if (act[p][q] && cnt[p][q] < pw) {
                what = "mixes a signal that is narrower than its static width with a rule that does not commute with the "
                       "speakers up-mix made upstream (discrete interpretation or more than two channels);";
                at = q;
              }
          }
        // the order of the inputs matters to the mix (inputs_mix_in_order) and not all of them are active over the same quanta: the
        // widths the bus takes — and with them an earlier input's path — change with the activity of the later ones
        if (!what && n.mode != WAA_COUNT_MODE_EXPLICIT && n.in_edges.size() >= 2 && n.in_edges.size() <= 64) {
          std::map<uint64_t, bool> seen;  // (few distinct activity patterns per node)
          for (uint32_t q = 0; q < nq && !what; q++) {
            uint64_t mask = 0;
            for (size_t k = 0; k < n.in_edges.size(); k++)
              if (act[b->edges[n.in_edges[k]].from][q]) mask |= (uint64_t)1 << k;
            if (!mask) continue;
            auto it = seen.find(mask);
            if (it == seen.end()) it = seen.emplace(mask, inputs_mix_in_order_differs(b, n, mask)).first;
            if (it->second) {
              what = "sums signals of different widths whose up-mixes depend on the order and on which of them are active (the reference's input bus grows input by input);";
              at = q;
            }
          }
        }
        if (what) {
          const bool keep_static = measure_switch("WAA_STATIC_CHANNEL_COUNTS") != nullptr;  // A/B aid: the round-1 behaviour
          if (keep_static)
            plan_note(b,
                      "note: the input of node %u %s at quantum %u (instance %u): the reference's dynamic channel count changes "
                      "mid-render, the device renders %d channel(s) throughout (DESIGN.md section 5)",
                      id, what, at, inst, n.in_nch);
          else
            plan_note(b,
                      "the input of node %u %s at quantum %u (instance %u): the reference's channel count changes mid-render -> "
                      "exact per-quantum channel counts (dyn_kernel, DESIGN.md section 3.4)",
                      id, what, at, inst);
          reported = true;
          count_change_found = !keep_static;
          if (keep_static && getenv("WAA_STRICT_CHANNEL_COUNTS"))
            return fail(WAA_ERR_OUT_OF_SCOPE, "node %u: a dynamic channel-count change is not rendered exactly on the device path", id);
          break;
        }
      }
      }  // tail_mode
    }
  }
  ph.reset();
  ph.reset(new PlanTrace("phase: folds + materialisation + views"));
  // Nodes whose state freezes while they do not process (WaveShaper 2x / 4x, HRTF panner; waa_frozen.hip) follow the
  // per-quantum silence / count codes of their input EXACTLY.  Directly behind one source those codes are host-known
  // (the scheduling replay); behind anything else they come out of the dynamic-count rendering.
  std::vector<int> frozen_src(N, -1);
  for (uint32_t id = 0; id < N && !count_change_found && !b->force_dynamic; id++) {
    Node& n = b->nodes[id];
    if (!n.live || !is_frozen_node(n)) continue;
    bool simple = n.in_edges.size() == 1 && scc_of[id] < 0 && !measure_switch("WAA_FROZEN_DYNAMIC");
    if (simple) {
      const uint32_t p = b->edges[n.in_edges[0]].from;
      const uint32_t pk = b->nodes[p].desc.kind;
      simple = (pk == WAA_NODE_BUFFER_SOURCE || pk == WAA_NODE_CONSTANT_SOURCE || pk == WAA_NODE_OSCILLATOR) &&
               b->nodes[p].out_nch == n.in_nch && n.in_nch <= (n.desc.kind == WAA_NODE_WAVESHAPER ? 6 : 2);
      if (simple) frozen_src[id] = (int)p;
    }
    if (!simple && !measure_switch("WAA_STATIC_CHANNEL_COUNTS")) {
      plan_note(b, "node %u keeps frozen state over silent quanta and is not fed by a single source -> exact per-quantum codes (dyn_kernel)", id);
      count_change_found = true;
    }
  }
  // A DelayNode outside a feedback loop whose delayTime is one value per quantum and whose consumers are all input
  // stages of chain kernels is not rendered by a pass of its own: the consumers gather from the delay line (IN_DELAYED).
  // (Static plans only; an echo — source -> Delay -> Gain -> bus — then costs one pass instead of three.)
  // Inside a feedback loop the same holds when the loop is block-scheduled (every loop-breaking delay longer than a block:
  // the gather then only reaches into earlier blocks), and a GainNode of such a loop can ride on an input edge like
  // outside: the classic echo loop Delay <-> Gain is ONE launch per block (line = source + gain * delayed(line)).
  std::vector<uint32_t> scc_block((size_t)n_scc, 0);
  if (!count_change_found && !b->force_dynamic && !measure_switch("WAA_NO_LOOP_FOLD"))
    for (int sc = 0; sc < n_scc; sc++) {
      std::vector<uint32_t> loop_items;
      for (uint32_t v : items)
        if (scc_of[v & ~VTX_READER] == sc) loop_items.push_back(v);
      scc_block[(size_t)sc] = loop_block_tiles(b, loop_items);
    }
  auto block_loop = [&](uint32_t id) { return scc_of[id] >= 0 && scc_block[(size_t)scc_of[id]] > 0; };
  std::vector<uint8_t> folded_delay(N, 0);
  for (uint32_t id = 0; id < N; id++) {
    Node& n = b->nodes[id];
    n.delay_folded = false;
    if (!n.live || n.desc.kind != WAA_NODE_DELAY || count_change_found || b->force_dynamic || measure_switch("WAA_NO_DELAY_FOLD")) continue;
    if ((scc_of[id] >= 0 || (id < b->cut.size() && b->cut[id])) && !block_loop(id)) continue;
    const ParamStore& dt = n.params[WAA_PARAM_DELAY_DELAY_TIME];
    bool ok = dt.mode() != 2 && n.in_edges.size() >= 1;
    if ((size_t)WAA_PARAM_DELAY_DELAY_TIME < n.pin_edges.size() && !n.pin_edges[WAA_PARAM_DELAY_DELAY_TIME].empty()) ok = false;
    int consumers = 0;
    for (auto& e : b->edges) {
      if (e.from != id || !b->nodes[e.to].live) continue;
      consumers++;
      const Node& c = b->nodes[e.to];
      const uint32_t ck = c.desc.kind;
      const bool other_loop = scc_of[e.to] >= 0 && !(scc_of[e.to] == scc_of[id] && block_loop(id));
      if ((e.to_input & 0x80000000u) || other_loop || (ck == WAA_NODE_CONVOLVER && c.has_ir) || is_frozen_node(c) || c.in_nch > 2 ||
          n.out_nch > 2)
        ok = false;
    }
    if (ok && consumers > 0) folded_delay[id] = 1;
  }
  // materialisation points
  std::vector<uint8_t> mat_hard(N, 0), fan_in_only(N, 0);
  for (uint32_t id = 0; id < N; id++) {
    Node& n = b->nodes[id];
    if (!n.live) continue;
    if (folded_delay[id]) {
      n.delay_folded = true;
      n.materialized = false;
      continue;
    }
    bool mat = false, fan = false;
    const uint32_t kind = n.desc.kind;
    if (kind == WAA_NODE_DESTINATION || kind == WAA_NODE_ANALYSER || kind == WAA_NODE_CONVOLVER || kind == WAA_NODE_DELAY) mat = true;
    if (is_frozen_node(n)) mat = true;  // rendered node-major (waa_frozen.hip)
    // (a GainNode of a block-scheduled loop may ride on an input edge of its consumer, see above)
    const bool relaxed = kind == WAA_NODE_GAIN && block_loop(id);
    if (scc_of[id] >= 0 && !relaxed) mat = true;  // loop members publish their own signal
    if (kind == WAA_NODE_OSCILLATOR) mat = true;  // rendered by its own (lane-per-instance) kernel
    int live_consumers = 0;
    for (auto& e : b->edges)
      if (e.from == id && b->nodes[e.to].live) {
        live_consumers++;
        const Node& c = b->nodes[e.to];
        if ((c.desc.kind == WAA_NODE_CONVOLVER && c.has_ir) || is_frozen_node(c)) mat = true;
        if (c.desc.kind == WAA_NODE_DELAY) {
          // a DelayNode mixes its inputs like a summing chain head (node_input_signal): materialised unless foldable
          if (relaxed)
            fan = true;
          else
            mat = true;
        }
        if (e.to_input & 0x80000000u) {
          // feeds an AudioParam: read back as a per-frame value signal — except a plain GainNode (the depth of an LFO),
          // which rides on the input edge of the param's summing chain like on any other summing input
          if (kind == WAA_NODE_GAIN && scc_of[id] < 0 && !count_change_found && !b->force_dynamic && !measure_switch("WAA_NO_EDGE_FOLD"))
            fan = true;
          else
            mat = true;
        }
        if (scc_of[e.to] >= 0 && !(relaxed && scc_of[e.to] == scc_of[id])) mat = true;  // feeds a feedback loop
        int live_in = 0;
        for (int ie : c.in_edges)
          if (b->nodes[b->edges[ie].from].live) live_in++;
        if (live_in > 1) fan = true;
      }
    if (live_consumers != 1) mat = true;
    mat_hard[id] = mat;
    fan_in_only[id] = !mat && fan;
    n.materialized = mat || fan;
  }
  // A producer that is materialised ONLY because its single consumer sums several inputs can instead be folded
  // into that consumer's input stage: a source is fetched by the summing kernel itself, and a GainNode on a
  // materialised signal (or on a source) becomes a per-edge gain — the mixer pattern source->Gain->bus costs no
  // pass through HBM of its own.
  auto is_source_kind = [&](uint32_t k) { return k == WAA_NODE_BUFFER_SOURCE || k == WAA_NODE_CONSTANT_SOURCE; };
  if (!measure_switch("WAA_NO_EDGE_FOLD"))
    for (uint32_t id = 0; id < N; id++) {
      Node& n = b->nodes[id];
      if (!n.live || !fan_in_only[id]) continue;
      if (is_source_kind(n.desc.kind)) {
        n.materialized = false;
        continue;
      }
      if (n.desc.kind != WAA_NODE_GAIN || n.in_edges.size() != 1 || n.in_nch != n.out_nch) continue;
      bool modulated = false;
      for (auto& pe : n.pin_edges) modulated |= !pe.empty();
      if (modulated) continue;
      const uint32_t x = b->edges[n.in_edges[0]].from;
      const Node& xn = b->nodes[x];
      if (!xn.live || xn.out_nch != n.in_nch) continue;
      // x is either materialised for its own reasons, or a source whose only consumer is this gain
      if (mat_hard[x] || folded_delay[x] || (is_source_kind(xn.desc.kind) && !fan_in_only[x])) n.materialized = false;
    }
  // A BufferSource whose output IS its AudioBuffer (fast track from frame 0: start 0, no offset / duration / stop /
  // loop, playbackRate 1, detune 0, buffer at the context's rate, one layout for all instances) and whose single
  // consumer is a node-major step that accepts a bounded view: read in place, no copy through HBM.
  for (uint32_t id = 0; id < N && !measure_switch("WAA_NO_SOURCE_VIEW"); id++) {
    Node& n = b->nodes[id];
    n.is_view = false;
    if (!n.live || n.desc.kind != WAA_NODE_BUFFER_SOURCE || !n.materialized || count_change_found || b->force_dynamic) continue;
    int consumer = -1, n_live = 0;
    for (auto& e : b->edges)
      if (e.from == id && b->nodes[e.to].live) {
        n_live++;
        consumer = (e.to_input & 0x80000000u) ? -1 : (int)e.to;
      }
    // ... or whose consumers are all summing input stages (chain heads, DelayNode mixes: they fetch a source themselves)
    // and folded single-input DelayNodes (whose delay line the source's buffer then IS): no copy either
    bool shared_ok = n_live >= 2;
    for (auto& e : b->edges) {
      if (e.from != id || !b->nodes[e.to].live || !shared_ok) continue;
      const Node& c = b->nodes[e.to];
      const uint32_t ck = c.desc.kind;
      if ((e.to_input & 0x80000000u) || (scc_of[e.to] >= 0 && !block_loop(e.to)) || (ck == WAA_NODE_CONVOLVER && c.has_ir) ||
          is_frozen_node(c) || ck == WAA_NODE_IIR_FILTER)
        shared_ok = false;
    }
    if (!shared_ok && (n_live != 1 || consumer < 0)) continue;
    const Node& c = b->nodes[(uint32_t)(shared_ok ? 0 : consumer)];
    const bool conv = !shared_ok && c.desc.kind == WAA_NODE_CONVOLVER && c.has_ir;
    bool frozen_ok = shared_ok;
    if (!shared_ok && is_frozen_node(c) && frozen_src[(uint32_t)consumer] == (int)id) {
      if (c.desc.kind == WAA_NODE_PANNER) {
        frozen_ok = true;
      } else {  // (a shaper that processes silent quanta would read them from the view: copy instead)
        const size_t cn = c.curve.size();
        const float mid = cn == 0 ? 0.f : (cn % 2 ? c.curve[cn / 2] : (c.curve[cn / 2 - 1] + c.curve[cn / 2]) / 2.f);
        frozen_ok = cn == 0 || std::fabs(mid) < 1e-9f;
      }
    }
    if (!(conv || frozen_ok)) continue;
    if (!shared_ok && (c.in_edges.size() != 1 || c.in_nch != n.out_nch || scc_of[(uint32_t)consumer] >= 0)) continue;
    const ParamStore& pr = n.params[WAA_PARAM_SOURCE_PLAYBACK_RATE];
    const ParamStore& pd = n.params[WAA_PARAM_SOURCE_DETUNE];
    bool ok = pr.blocks.empty() && pd.blocks.empty() && pr.timelines.empty() && pd.timelines.empty() && !pr.dev_tl && !pd.dev_tl;
    const DeviceBuffer& b0 = n.bufs[0];
    ok = ok && b0.valid && b0.sr == b->sr && (uintptr_t)b0.base % 16 == 0 && b0.ch_stride % 4 == 0 && b0.frames % RQ == 0 && b0.frames > 0;
    const int64_t inst_stride = b->n_inst > 1 && n.bufs[1].valid ? n.bufs[1].base - b0.base : (int64_t)b0.ch_stride * b0.nch;
    ok = ok && inst_stride >= 0 && inst_stride % 4 == 0;  // (0: one AudioBuffer shared by every instance, set_buffer(ALL))
    for (uint32_t i = 0; i < b->n_inst && ok; i++) {
      const DeviceBuffer& bf = n.bufs[i];
      const SourceSched& ss = n.sched[i];
      ok = bf.valid && bf.base == b0.base + (int64_t)i * inst_stride && bf.ch_stride == b0.ch_stride && bf.frames == b0.frames &&
           bf.nch == b0.nch && bf.sr == b0.sr && pr.cst[i] == 1.f && pd.cst[i] == 0.f && ss.start == 0. && ss.stop == DBL_MAX &&
           ss.offset == 0. && ss.duration == DBL_MAX && !ss.looping;
    }
    // ... and every quantum the reference renders as ACTIVE lies inside the buffer.  The quantum right behind a buffer that ends on
    // a quantum boundary can be one: the source's clock reaches the duration by additions of dt, and where it still compares below it
    // the renderer produces one more quantum — of zeros, but not the silent block — before it ends (audio_buffer_source.rs:560-600).
    // Read in place that quantum is the next context's first one (frozen-state fuzz seed 38723, round 6: a 2304-frame buffer in
    // front of an HRTF panner; 0.2 of full scale for two of three contexts in the panner's tail).
    for (uint32_t i = 0; i < b->n_inst && ok; i++) {
      const DeviceBuffer& bf = n.bufs[i];
      const SourceSched& ss = n.sched[i];
      const SchedKey key(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end, bf.frames, bf.sr, pr.fix(pr.cst[i]),
                         pd.fix(pd.cst[i]));
      const auto so = schedule_source_cached(b, id, key, ss, bf.frames, bf.sr, true, param_per_quantum(b, pr, i, nullptr), param_per_quantum(b, pd, i, nullptr));
      for (size_t q = 0; q < so->qrec.size() && ok; q++)
        if (so->qrec[q].mode != Q_SILENT)
          ok = so->qrec[q].mode == Q_FAST && so->qrec[q].start == (int64_t)q * RQ && (uint64_t)(q + 1) * RQ <= bf.frames;
    }
    if (!ok) continue;
    n.is_view = true;
    n.materialized = false;
    n.view_sig = SignalRef{b0.base, (uint64_t)inst_stride, b0.ch_stride, (int32_t)b0.nch, 0};
    n.view_valid = b0.frames;
    if (shared_ok)
      plan_note(b, "source node %u renders its AudioBuffer unchanged: its %d consumers read it in place (%llu frames per channel)", id, n_live,
                (unsigned long long)b0.frames);
    else
      plan_note(b, "source node %u renders its AudioBuffer unchanged: node %d reads it in place (%llu frames per channel)", id, consumer,
                (unsigned long long)b0.frames);
  }
  // A BiquadFilterNode with constant coefficients directly in front of a long ConvolverNode (source -> Biquad -> Convolver,
  // the north-star graph): the forward transform's input stage filters its blocks itself (conv_fft3_fwd_bq_kernel) — the
  // filtered signal never crosses HBM and the Biquad costs no launch.  Its own input must be a plain signal: a
  // BufferSource that renders its AudioBuffer unchanged (read in place) or a signal some other node materialises anyway.
  for (auto& n : b->nodes) n.fold_conv = n.pre_biquad = -1;
  if (!count_change_found && !b->force_dynamic && !measure_switch("WAA_NO_CONV_BIQUAD_FOLD") && !measure_switch("WAA_CONV_FFT_R4"))
    for (uint32_t cid = 0; cid < N; cid++) {
      Node& c = b->nodes[cid];
      if (!c.live || c.desc.kind != WAA_NODE_CONVOLVER || !c.has_ir || scc_of[cid] >= 0 || c.in_edges.size() != 1) continue;
      if (conv_block_size(b, c) != 8192) continue;
      const uint32_t qid = b->edges[c.in_edges[0]].from;
      Node& q = b->nodes[qid];
      if (!q.live || q.desc.kind != WAA_NODE_BIQUAD || scc_of[qid] >= 0 || q.in_nch != q.out_nch || q.out_nch != c.in_nch || q.in_nch > 2 ||
          q.in_edges.size() != 1)
        continue;
      int q_consumers = 0;
      for (auto& e : b->edges) q_consumers += e.from == qid && b->nodes[e.to].live;
      bool plain = q_consumers == 1;
      for (auto& pe : q.pin_edges) plain = plain && pe.empty();
      for (auto& ps : q.params) plain = plain && ps.mode() == 0 && ps.timelines.empty() && !ps.dev_tl;
      if (!plain) continue;
      const uint32_t sid = b->edges[q.in_edges[0]].from;
      Node& sn = b->nodes[sid];
      if (!sn.live || sn.out_nch != q.in_nch) continue;
      if (sn.desc.kind == WAA_NODE_BUFFER_SOURCE && !sn.materialized && !sn.is_view && !measure_switch("WAA_NO_SOURCE_VIEW")) {
        int s_consumers = 0;
        for (auto& e : b->edges) s_consumers += e.from == sid && b->nodes[e.to].live;
        const ParamStore& pr = sn.params[WAA_PARAM_SOURCE_PLAYBACK_RATE];
        const ParamStore& pd = sn.params[WAA_PARAM_SOURCE_DETUNE];
        bool ok = s_consumers == 1 && pr.blocks.empty() && pd.blocks.empty() && pr.timelines.empty() && pd.timelines.empty() &&
                  !pr.dev_tl && !pd.dev_tl;
        for (auto& pe : sn.pin_edges) ok = ok && pe.empty();
        const DeviceBuffer& b0 = sn.bufs[0];
        ok = ok && b0.valid && b0.sr == b->sr && (uintptr_t)b0.base % 16 == 0 && b0.ch_stride % 4 == 0 && b0.frames % RQ == 0 && b0.frames > 0;
        const int64_t inst_stride = b->n_inst > 1 && sn.bufs[1].valid ? sn.bufs[1].base - b0.base : (int64_t)b0.ch_stride * b0.nch;
        ok = ok && inst_stride >= 0 && inst_stride % 4 == 0;  // (0: one AudioBuffer shared by every instance, set_buffer(ALL))
        for (uint32_t i = 0; i < b->n_inst && ok; i++) {
          const DeviceBuffer& bf = sn.bufs[i];
          const SourceSched& ss = sn.sched[i];
          ok = bf.valid && bf.base == b0.base + (int64_t)i * inst_stride && bf.ch_stride == b0.ch_stride && bf.frames == b0.frames &&
               bf.nch == b0.nch && bf.sr == b0.sr && pr.cst[i] == 1.f && pd.cst[i] == 0.f && ss.start == 0. && ss.stop == DBL_MAX &&
               ss.offset == 0. && ss.duration == DBL_MAX && !ss.looping;
        }
        if (!ok) continue;
        sn.is_view = true;
        sn.view_sig = SignalRef{b0.base, (uint64_t)inst_stride, b0.ch_stride, (int32_t)b0.nch, 0};
        sn.view_valid = b0.frames;
      } else if (!(mat_hard[sid] && !sn.is_view && !sn.delay_folded)) {
        continue;
      }
      q.fold_conv = (int)cid;
      q.materialized = false;
      c.pre_biquad = (int)qid;
      if (conv_fold_biquad_into_ir(b, c, q))
        plan_note(b, "biquad node %u has the same constant coefficients on every context and only feeds convolver node %u: both are LTI, its "
                     "transfer function is folded into the impulse response (%llu -> %llu taps)%s",
                  qid, cid, (unsigned long long)c.ir_len, (unsigned long long)c.ir_lti_len, sn.is_view ? " (its source is read in place)" : "");
      else
        plan_note(b, "biquad node %u has constant coefficients and only feeds convolver node %u: filtered by the forward transform's input stage%s",
                  qid, cid, sn.is_view ? " (its source is read in place)" : "");
    }
  auto alloc_signal = [&](Node& n) -> int {
    float* p = nullptr;
    int e = dev_alloc(b, &p, (size_t)b->n_inst * n.out_nch * b->lp);
    if (e) return e;
    n.sig = SignalRef{p, (uint64_t)n.out_nch * b->lp, b->lp, n.out_nch, 0};
    return 0;
  };
  ph.reset();
  ph.reset(new PlanTrace("phase: units + steps"));
  // planning units: single nodes and whole feedback loops, producers first (the condensed graph is acyclic),
  // otherwise in processing order
  std::vector<Unit> units;
  {
    std::vector<uint8_t> node_done(N, 0), scc_done(n_scc, 0);
    std::function<void(uint32_t)> visit_unit = [&](uint32_t id) {
      const int sc = scc_of[id];
      if (sc >= 0 ? scc_done[sc] : node_done[id]) return;
      std::vector<uint32_t> members;
      if (sc >= 0) {
        scc_done[sc] = 1;
        for (uint32_t m : b->order)
          if (scc_of[m] == sc) members.push_back(m);
      } else {
        node_done[id] = 1;
        members.push_back(id);
      }
      for (uint32_t m : members) {
        for (int e : b->nodes[m].in_edges)
          if (sc < 0 || scc_of[b->edges[e].from] != sc) visit_unit(b->edges[e].from);
        for (auto& pe : b->nodes[m].pin_edges)
          for (int e : pe)
            if (sc < 0 || scc_of[b->edges[e].from] != sc) visit_unit(b->edges[e].from);
      }
      units.push_back(Unit{sc, id});
    };
    for (uint32_t id : b->order) visit_unit(id);
  }
  b->steps.clear();
  std::function<int(uint32_t)> plan_single = [&](uint32_t id) -> int {
    Node& term = b->nodes[id];
    if (b->prepass) {
      // a source whose playbackRate / detune the graph modulates: plan the params' summing chains (their producers are
      // planned by now), not the source
      bool mine = false;
      for (size_t k = 0; k < b->prepass_params.size(); k++)
        if (b->prepass_params[k].first == id) {
          mine = true;
          if (int e = node_param(b, id, b->prepass_params[k].second, &b->prepass_refs[k])) return e;
        }
      if (mine) return 0;
    }
    if (term.live && term.delay_folded) {
      plan_note(b, "delay node %u: %dch, read by its consumers from the delay line (no pass of its own)", id, term.in_nch);
      return node_input_signal(b, id, &term.hist, nullptr, &term.hist_valid);  // the delay line = the node's mixed input
    }
    if (!term.live || !term.materialized) return 0;
    if (term.desc.kind == WAA_NODE_CONVOLVER && term.has_ir) {
      if (scc_of[id] >= 0 && !block_loop(id))
        return fail(WAA_ERR_OUT_OF_SCOPE, "a ConvolverNode inside a feedback loop is out of scope when the loop's delay is shorter than one partition of its impulse response (node %u)", id);
      int e = alloc_signal(term);
      if (e) return e;
      if ((e = plan_convolver(b, id))) return e;
      return 0;
    }
    if (term.desc.kind == WAA_NODE_OSCILLATOR) {
      int e = alloc_signal(term);
      if (e) return e;
      return plan_oscillator(b, id);
    }
    if (is_frozen_node(term)) {
      if (scc_of[id] >= 0)
        return fail(WAA_ERR_OUT_OF_SCOPE, "an oversampled WaveShaperNode / HRTF PannerNode inside a feedback loop is out of scope (node %u)", id);
      int e = alloc_signal(term);
      if (e) return e;
      return term.desc.kind == WAA_NODE_PANNER ? plan_hrtf(b, id, frozen_src[id]) : plan_oversampler(b, id, frozen_src[id]);
    }
    if (term.desc.kind == WAA_NODE_DELAY) {  // outside a loop: writer and reader halves back to back
      int e = alloc_signal(term);
      if (e) return e;
      if ((e = plan_delay_writer(b, id))) return e;
      if ((e = plan_delay_reader(b, id))) return e;
      return 0;
    }
    // identity node on a materialised signal of the same layout (destination / analyser / passthrough right
    // behind a materialised producer): alias instead of copying 8 B per frame-channel through HBM
    if (term.in_edges.size() == 1) {
      Node& p = b->nodes[b->edges[term.in_edges[0]].from];
      const uint32_t k = term.desc.kind;
      const bool identity = k == WAA_NODE_DESTINATION || k == WAA_NODE_ANALYSER ||
                            (k == WAA_NODE_CONVOLVER && !term.has_ir) || (k == WAA_NODE_WAVESHAPER && !term.has_curve);
      if (identity && scc_of[id] < 0 && p.materialized && p.out_nch == term.in_nch && term.in_nch == term.out_nch) {
        term.sig = p.sig;
        plan_note(b, "alias node %u -> output of node %u", id, b->edges[term.in_edges[0]].from);
        return 0;
      }
    }
    if (!term.sig.base) {  // (members of a feedback loop are allocated up front)
      int e = alloc_signal(term);
      if (e) return e;
    }
    // walk back through fused single-input predecessors
    std::vector<uint32_t> path;  // terminal first
    uint32_t cur = id;
    Step st;
    ChainDesc& cd = st.chain;
    std::memset(&cd, 0, sizeof cd);
    for (;;) {
      path.push_back(cur);
      Node& n = b->nodes[cur];
      const uint32_t kind = n.desc.kind;
      if (kind == WAA_NODE_BUFFER_SOURCE || kind == WAA_NODE_CONSTANT_SOURCE) break;  // chain input = this source
      if (n.in_edges.size() != 1) break;                                               // silent or fan-in head
      uint32_t p = b->edges[n.in_edges[0]].from;
      if (b->nodes[p].materialized || b->nodes[p].delay_folded) break;
      cur = p;
    }
    // inputs of the head node
    const uint32_t head = path.back();
    Node& hn = b->nodes[head];
    int cmax = 1;
    if (hn.desc.kind == WAA_NODE_BUFFER_SOURCE || hn.desc.kind == WAA_NODE_CONSTANT_SOURCE) {
      cd.n_inputs = 1;
      cd.in_nch = hn.out_nch;
      cd.in_interp = 0;
      InputRef& in = cd.in[0];
      in.nch = hn.out_nch;
      if (hn.desc.kind == WAA_NODE_BUFFER_SOURCE) {
        in.kind = IN_SOURCE;
        // filled by prepare_source below
      } else {
        in.kind = IN_CONSTANT;
      }
    } else if (hn.in_edges.empty()) {
      cd.n_inputs = 1;
      cd.in[0].kind = IN_SILENT;
      cd.in[0].nch = 1;
      cd.in_nch = hn.in_nch;
      cd.in_interp = hn.interp;
    } else {
      cd.in_nch = hn.in_nch;
      cd.in_interp = hn.interp;
      std::vector<InputRef> ins;
      for (int ie : hn.in_edges) {
        InputRef in{};
        int e = build_edge_input(b, head, ie, &in);
        if (e) return e;
        ins.push_back(in);
      }
      int e = premix_ordered_inputs(b, head, ins);
      if (e) return e;
      e = reduce_fan_in(b, ins, hn.in_nch, hn.interp);
      if (e) return e;
      cd.n_inputs = (int)ins.size();
      for (int k = 0; k < cd.n_inputs; k++) {
        cd.in[k] = ins[k];
        cmax = std::max(cmax, ins[k].nch);
      }
    }
    cmax = std::max(cmax, cd.in_nch);
    // ops: head first
    std::vector<OpDesc> ops;
    int cur_nch = cd.in_nch;
    for (size_t k = path.size(); k-- > 0;) {
      uint32_t nid = path[k];
      const bool is_head = (k == path.size() - 1);
      int out_nch = cur_nch;
      int e = emit_node_ops(b, nid, cur_nch, is_head, ops, &out_nch);
      if (e) return e;
      cur_nch = out_nch;
    }
    // source inputs: schedules / buffer tables / constant ranges
    if (hn.desc.kind == WAA_NODE_BUFFER_SOURCE || hn.desc.kind == WAA_NODE_CONSTANT_SOURCE) {
      int e = prepare_source_input(b, head, &cd.in[0]);
      if (e) return e;
    }
    // An oscillator that feeds nothing but this chain, and the chain nothing but constant gains and the speakers up-mix
    // 1 -> 2 (Oscillator -> Gain -> stereo destination: the plain tone generator): the oscillator's own launch writes the
    // result, no second pass over it.
    if (cd.n_inputs == 1 && cd.in[0].kind == IN_SIGNAL && !cd.in[0].has_gain && cd.in[0].nch == 1 && cd.in_nch == 1 &&
        scc_of[id] < 0 && !measure_switch("WAA_NO_OSC_POST")) {
      int prod = -1;
      for (uint32_t k = 0; k < N; k++)
        if (b->nodes[k].live && b->nodes[k].desc.kind == WAA_NODE_OSCILLATOR && b->nodes[k].sig.base == cd.in[0].sig.base &&
            b->nodes[k].osc_step >= 0)
          prod = (int)k;
      // every live reader of the oscillator's signal — also through nodes that ALIAS it (an AnalyserNode / pass-through right
      // behind it shares the buffer, and the analyser's FFT reads it): with the fold the oscillator no longer writes that
      // buffer at all (fuzz seed 502310: Oscillator -> Analyser -> two Gains; one Gain's chain was folded, the other read
      // memory nobody had written — found with WAA_POISON_ALLOC)
      int consumers = 0;
      if (prod >= 0) {
        const float* osc_sig = b->nodes[(size_t)prod].sig.base;
        for (uint32_t k = 0; k < N; k++) {
          const Node& an = b->nodes[k];
          if (!an.live || an.sig.base != osc_sig) continue;
          if (k != (uint32_t)prod && (an.desc.kind == WAA_NODE_ANALYSER || an.desc.kind == WAA_NODE_DESTINATION))
            consumers += 2;  // (the analyser's FFT / the caller read that buffer)
          for (auto& e2 : b->edges)
            if (e2.from == k && b->nodes[e2.to].live && b->nodes[e2.to].sig.base != osc_sig) consumers++;
        }
      }
      size_t n_gain = 0;
      bool ok = prod >= 0 && consumers == 1 && b->steps[(size_t)b->nodes[(size_t)prod].osc_step].group < 0;
      while (ok && n_gain < ops.size() && ops[n_gain].kind == OP_GAIN) {
        ok = ops[n_gain].p0.mode == 0 && ops[n_gain].nch_in == 1 && n_gain < 2;
        n_gain++;
      }
      const bool dup = ok && n_gain + 1 == ops.size() && ops[n_gain].kind == OP_MIX && ops[n_gain].nch_in == 1 &&
                       ops[n_gain].nch_out == 2 && ops[n_gain].i0 == WAA_INTERP_SPEAKERS && term.sig.nch == 2;
      if (ok && (dup || (n_gain == ops.size() && term.sig.nch == 1 && n_gain > 0))) {
        OscDesc& od = b->steps[(size_t)b->nodes[(size_t)prod].osc_step].osc;
        od.out = term.sig;
        od.n_post = (int32_t)n_gain;
        for (size_t k = 0; k < n_gain; k++) od.post_gain[k] = ops[k].p0;
        od.post_dup = dup ? 1 : 0;
        plan_note(b, "oscillator node %d renders %zu gain(s)%s of node %u's chain itself", prod, n_gain,
                  dup ? " and the up-mix 1 -> 2" : "", id);
        return 0;
      }
    }
    std::vector<InputRef> inputs(cd.in, cd.in + cd.n_inputs);
    int e = emit_segments(b, inputs, cd.in_nch, cd.in_interp, ops, term.sig);
    if (e) return e;
    return 0;
  };

  // ---------------------------------------------------------------------------------------------------------
  // Dynamic channel counts (waa_dyn.hip): sources, oscillators and FFT convolvers stay node-major launches; every
  // other live node becomes an item of a quantum-serial dyn_kernel launch that carries per-quantum codes
  // (count | silent) with every signal.  A convolver splits the items into groups (its input is produced by the
  // group in front of it, its output consumed by the group behind it).
  if (b->prepass && (count_change_found || b->force_dynamic))
    return fail(WAA_ERR_OUT_OF_SCOPE, "the graph that modulates a source's playbackRate / detune needs exact per-quantum channel counts: out of scope");
  if (count_change_found || b->force_dynamic) {
    DynPlanCtx dc{items, units, scc_of, alloc_signal, plan_single, count_change_found, mixed_buffer_counts};
    return plan_dynamic_groups(b, dc);
  }
  // chains, in processing order of their terminal node
  for (const Unit& unit : units) {
    const uint32_t id = unit.id;
    if (unit.scc >= 0) {
      std::vector<uint32_t> loop_items;
      bool any_live = false;
      for (uint32_t v : items)
        if (scc_of[v & ~VTX_READER] == unit.scc) {
          loop_items.push_back(v);
          any_live |= b->nodes[v & ~VTX_READER].live;
        }
      if (!any_live) continue;
      for (uint32_t v : loop_items) {
        Node& m = b->nodes[v & ~VTX_READER];
        if (!m.sig.base) {
          int e = alloc_signal(m);
          if (e) return e;
        }
      }
      const uint32_t bt = loop_block_tiles(b, loop_items);
      if (bt == 0) {  // short or modulated loop delay: quantum-serial loop kernel
        int e = plan_loop(b, loop_items);
        if (e == WAA_ERR_OUT_OF_SCOPE && !b->force_dynamic && !measure_switch("WAA_STATIC_CHANNEL_COUNTS")) {
          // the static loop kernel covers Gain / Biquad / WaveShaper / k-rate StereoPanner / Delay members only; the
          // dynamic-count kernel renders every node kind quantum by quantum (waa_dyn.hip): plan the graph again with it
          b->force_dynamic = true;
          b->steps.clear();
          b->group_tiles.clear();
          b->qgroup_quanta.clear();
          b->state_bufs.clear();
          b->plan_log.clear();
          for (auto& nd : b->nodes) {
            nd.sig = SignalRef{};
            nd.hist = SignalRef{};
            nd.hist_is_temp = false;
          }
          return build_plan_impl(b);
        }
        if (e) return e;
        continue;
      }
      // Block-scheduled loop: every delay that breaks the loop is longer than `bt` tiles, so a block of bt tiles
      // only reads loop history from earlier blocks: the members are planned as ordinary node-major steps (same
      // kernels as outside a loop) in the reference's processing order and launched block by block.
      const size_t first_step = b->steps.size();
      for (uint32_t v : loop_items) {
        const uint32_t mid = v & ~VTX_READER;
        int e = 0;
        if (is_delay(b, mid) && b->nodes[mid].delay_folded)
          e = (v & VTX_READER) ? plan_folded_delay_line(b, mid) : plan_delay_writer(b, mid);
        else if (is_delay(b, mid))
          e = (v & VTX_READER) ? plan_delay_reader(b, mid) : plan_delay_writer(b, mid);
        else
          e = plan_single(mid);
        if (e) return e;
      }
      const int group = (int)b->group_tiles.size();
      b->group_tiles.push_back(bt);
      for (size_t k = first_step; k < b->steps.size(); k++) {
        Step& st = b->steps[k];
        st.group = group;
        // steps that only depend on data from outside the loop run once, over the full range, before the blocks
        st.prologue = st.kind == 5 || st.kind == 12 || st.kind == 13 || st.kind == 14 || st.kind == 3 || (st.kind == 0 && st.chain.n_ops == 1 && st.chain.ops[0].kind == OP_PARAM_ADD);
        if (st.prologue && st.kind == 0) {
          // ... unless the param is modulated from INSIDE the loop: its summing chain then reads what a launch of this
          // group writes and belongs to the blocks, in its place in the order
          const StepIo io = step_io(st);
          for (size_t j = first_step; j < b->steps.size() && st.prologue; j++) {
            if (j == k) continue;
            const StepIo w = step_io(b->steps[j]);
            for (const void* r : io.reads)
              if (std::find(w.writes.begin(), w.writes.end(), r) != w.writes.end()) st.prologue = false;
          }
        }
        st.prologue |= st.kind == 18;
        if (st.kind == 15 || st.kind == 16 || st.kind == 17 || st.kind == 20)
          return fail(WAA_ERR_OUT_OF_SCOPE, "this node kind cannot be rendered inside a feedback loop");
      }
      plan_note(b, "feedback loop: block-scheduled, %u tile(s) = %u frames per block, %zu step(s) per block", bt, bt * TILE,
                b->steps.size() - first_step);
      {
        // one element-wise launch per block whose only loop-carried input is its own delay line (Delay <-> Gain): the
        // LDS-ring kernel renders the whole loop in one launch (waa_echo.hip) when every instance's delay fits its window
        size_t n_body = 0, body = 0;
        std::vector<size_t> bodies;
        for (size_t k = first_step; k < b->steps.size(); k++)
          if (!b->steps[k].prologue) {
            n_body++;
            body = k;
            bodies.push_back(k);
          }
        float range[2] = {1e30f, 0.f};
        auto delay_range = [&]() {
          for (uint32_t v : loop_items) {
            const uint32_t did = v & ~VTX_READER;
            if (!(v & VTX_READER) || !b->cut[did]) continue;
            Node& dn = b->nodes[did];
            for (uint32_t i = 0; i < b->n_inst; i++)
              for (float dvv : param_per_quantum(b, dn.params[WAA_PARAM_DELAY_DELAY_TIME], i, nullptr)) {
                const float fr = std::max(dvv, (float)RQ / (float)b->sr) * (float)b->sr;
                range[0] = std::min(range[0], fr);
                range[1] = std::max(range[1], fr);
              }
          }
        };
        if (n_body == 1 && b->steps[body].kind == 0 && !measure_switch("WAA_NO_ECHO_RING")) {
          delay_range();
          ChainDesc cd = b->steps[body].chain;
          cd.tile0 = 0;
          cd.tile1 = b->n_tiles;
          int chunk = 0;
          const int fb = echo_ring_applicable(cd, range, &chunk);
          if (fb >= 0) {
            b->steps[body].echo_fb = fb;
            b->steps[body].echo_chunk = chunk;
            b->steps[body].echo_ring = echo_ring_frames(range[1], chunk);
            b->steps[body].profile_slot = slot_for(b, "echo_ring_kernel");
            plan_note(b, "  ... rendered by the LDS-ring kernel in ONE launch: delay %.0f .. %.0f frames, chunks of %d frames, the line's last %d frames stay in LDS",
                      (double)range[0], (double)range[1], chunk, b->steps[body].echo_ring);
          }
        }
        // delayed read -> streaming biquad (constant coefficients) -> sum into the line: the filtered echo, the ring kernel's BQ form
        if (n_body == 3 && b->steps[bodies[0]].kind == 0 && b->steps[bodies[1]].kind == 1 && b->steps[bodies[2]].kind == 0 &&
            !b->steps[bodies[1]].scan.payload && !measure_switch("WAA_NO_ECHO_RING") && !measure_switch("WAA_NO_ECHO_BQ")) {
          delay_range();
          Step &rd = b->steps[bodies[0]], &fl = b->steps[bodies[1]], &sm = b->steps[bodies[2]];
          int chunk = 0;
          EchoBq q{};
          ChainDesc cd = sm.chain;
          cd.tile0 = 0;
          cd.tile1 = b->n_tiles;
          int fb = echo_bq_applicable(rd.chain, fl.bq, cd, range, &chunk, &q);
          // the delayed samples are not stored: nothing but the filter may read them
          for (size_t k = 0; k < b->steps.size() && fb >= 0; k++) {
            if (k == bodies[1]) continue;
            const StepIo io = step_io(b->steps[k]);
            if (std::find(io.reads.begin(), io.reads.end(), (const void*)rd.chain.out.base) != io.reads.end()) fb = -1;
          }
          for (const Node& an : b->nodes)
            if (an.live && an.sig.base == rd.chain.out.base && (an.desc.kind == WAA_NODE_ANALYSER || an.desc.kind == WAA_NODE_DESTINATION)) fb = -1;
          if (fb >= 0) {
            rd.echo_fused = true;
            fl.echo_fused = true;
            sm.echo_fb = fb;
            sm.echo_chunk = chunk;
            sm.echo_ring = echo_ring_frames(range[1], chunk);
            sm.echo_bq = q;
            sm.profile_slot = slot_for(b, "echo_ring_kernel");
            plan_note(b, "  ... rendered by the LDS-ring kernel in ONE launch with the Biquad between the delayed read and the sum: delay %.0f .. %.0f frames, chunks of %d frames",
                      (double)range[0], (double)range[1], chunk);
          }
        }
        // a loop SHORTER than a tile was planned this way only for the ring kernel (loop_block_tiles): launches per block cannot
        // render it — if it did not qualify, plan the graph again with the quantum-serial loop kernel for such loops
        bool short_ring = false, qualified = false;
        for (uint32_t v : loop_items) short_ring |= b->short_ring_loops.count(v & ~VTX_READER) != 0;
        for (size_t k : bodies) qualified |= b->steps[k].echo_fb >= 0;
        if (short_ring && !qualified) {
          b->no_short_ring = true;
          b->steps.clear();
          b->group_tiles.clear();
          b->qgroup_quanta.clear();
          b->state_bufs.clear();
          b->plan_log.clear();
          for (auto& nd : b->nodes) {
            nd.sig = SignalRef{};
            nd.hist = SignalRef{};
            nd.hist_is_temp = false;
          }
          return build_plan_impl(b);
        }
        if (short_ring) plan_note(b, "  (a feedback delay shorter than a tile: the ring kernel walks it in chunks shorter than the delay)");
      }
      continue;
    }
    int e = plan_single(id);
    if (e) return e;
  }
  fuse_echo_tails(b);
  ring_feed_forward_echoes(b);
  fuse_fm_operators(b);
  fuse_lfo_params(b);
  // (self-test of the check below: a reversed launch list must not get past it, tests/test_plan.py)
  ph.reset();
  ph.reset(new PlanTrace("phase: validate"));
  if (measure_switch("WAA_DEBUG_REVERSE_PLAN")) std::reverse(b->steps.begin(), b->steps.end());
  if (int e = validate_plan(b)) return e;
  b->planned = true;
  return 0;
}

}  // namespace host
}  // namespace waa
