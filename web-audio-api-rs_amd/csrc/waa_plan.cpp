// waa_plan.cpp — the planner: processing order with the reference's cycle breaker, liveness, static channel
// counts, materialisation points, fusion of single-consumer paths into chain launches, node-major steps
// (convolver, delay, oscillator, IIR), feedback loops (block-scheduled or quantum-serial), AudioParam chains.
#include <array>
#include <set>

#include "waa_host.hpp"

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
  } else if (pn.is_view && !getenv("WAA_NO_VIEW_SIGNAL")) {
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

// graph.rs:323-487 order_nodes / visit.  A DelayNode is two graph nodes in the reference (delay.rs:283-366:
// writer, then reader, edge writer->reader); vertex `id` is the writer (or a plain node), `id | VTX_READER` the
// reader.  Cycles are broken at the first DelayNode writer on the detected loop (its writer->reader edge is
// cleared and the ordering restarts); nodes of a loop without one are dropped from the ordering (muted).
constexpr uint32_t VTX_READER = 0x80000000u;
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

// One interpreter-kernel step: input(s) -> [mix to in_nch] -> ops -> out
int push_chain_step(waa_batch* b, const std::vector<InputRef>& inputs, int in_nch, int in_interp,
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
  const int max_mode = getenv("WAA_NO_KRATE_STREAM") ? 0 : (getenv("WAA_NO_ARATE_STREAM") ? 1 : 2);
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
                     ops[j].nch_in == 1 && ops[j].nch_out == 2 && ops[j].i0 == WAA_INTERP_SPEAKERS && !getenv("WAA_NO_STREAM_DUP");
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
        q.exact = (row_cost < lane_cost && !getenv("WAA_IIR_LANE")) || getenv("WAA_IIR_ROW") ? 2u : 1u;
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
    if (o.kind == OP_BIQUAD && o.i0 == 2 && !dup && !b->dry && b->steps[(size_t)o.i1].coef.rows == 1 && !getenv("WAA_ARATE_STREAM") &&
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
      if (cstep.coef.rows == 1 && !getenv("WAA_NO_ARATE_DIGEST")) {
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
    if (q.vary == 0 && !b->dry && getenv("WAA_BIQUAD_SCAN")) {
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

// what a launch reads and writes (plan validation, and the prologue decision of block-scheduled loops)
struct StepIo {
  std::vector<const void*> reads, writes;
  bool feedback_reader = false;
};
namespace {
StepIo step_io(const Step& st);
}
void fuse_echo_tails(waa_batch* b);
void ring_feed_forward_echoes(waa_batch* b);
int plan_loop(waa_batch* b, const std::vector<uint32_t>& loop_items);
int plan_delay_writer(waa_batch* b, uint32_t id);
int node_input_signal(waa_batch* b, uint32_t id, SignalRef* out_sig, const SignalRef* target = nullptr, uint64_t* valid = nullptr);
int conv_block_size(const waa_batch* b, const Node& n);
int emit_node_ops(waa_batch* b, uint32_t id, int cur_nch, bool head, std::vector<OpDesc>& ops, int* out_nch);
int plan_oscillator(waa_batch* b, uint32_t id);
int plan_delay_reader(waa_batch* b, uint32_t id);
int plan_folded_delay_line(waa_batch* b, uint32_t id);
uint32_t loop_block_tiles(waa_batch* b, const std::vector<uint32_t>& loop_items);

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
        const bool want = getenv("WAA_DEVICE_AUTOMATION") ? true : !shared;  // (switch: also replay shared timelines there)
        if (consumable && !p.k_rate && want && !b->dry && !getenv("WAA_HOST_AUTOMATION")) {
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
      auto itc = automated ? cache.end() : cache.find(key);
      if (itc == cache.end()) {
        SchedOut so;
        schedule_source(b, ss, bf.frames, bf.sr, bf.valid, rate_q, det_q, &so);
        std::vector<uint8_t> r(b->n_quanta);
        for (uint32_t q = 0; q < b->n_quanta; q++)
          r[q] = so.qrec[q].mode == Q_SILENT ? (uint8_t)(1u | CODE_SILENT) : (uint8_t)n.out_nch;
        itc = cache.emplace(automated ? SchedKey(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end,
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

int build_plan(waa_batch* b) {
  const uint32_t N = (uint32_t)b->nodes.size();
  if (int e = materialise_automation(b)) return e;
  for (uint32_t i = 0; i < N; i++)  // the reference takes the coefficients in the constructor
    if (b->nodes[i].desc.kind == WAA_NODE_IIR_FILTER && b->nodes[i].iir_b.empty())
      return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - IIRFilterNode %u has no coefficients", i);
  // processing order = reversed DFS post-order over outgoing edges in insertion order, cycle breakers applied
  // (graph.rs:323-487); `items` has two entries per DelayNode (writer, reader), none for muted nodes
  std::vector<uint32_t> items;
  std::vector<uint8_t> muted(N, 0);
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
    b->cut = c.cut;
  }
  // one entry per node, at the position where its OUTPUT is produced (the reader half of a DelayNode)
  b->order.clear();
  for (uint32_t v : items)
    if (!is_delay(b, v & ~VTX_READER) || (v & VTX_READER)) b->order.push_back(v & ~VTX_READER);
  for (uint32_t i = 0; i < N; i++)
    if (muted[i]) plan_note(b, "node %u is part of a cycle without a DelayNode: muted (graph.rs:362-368)", i);
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
      if (!(k == WAA_NODE_GAIN || k == WAA_NODE_BIQUAD || k == WAA_NODE_DELAY || k == WAA_NODE_STEREO_PANNER ||
            k == WAA_NODE_CONSTANT_SOURCE || k == WAA_NODE_OSCILLATOR || source_rate))
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
      for (auto& pp : b->prepass_params) stack.push_back(pp.first);
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
        uint32_t nch = 0;
        for (auto& bf : n.bufs)
          if (bf.valid) {
            if (nch && bf.nch != nch)
              return fail(WAA_ERR_OUT_OF_SCOPE, "instances of one batch must use AudioBuffers with the same channel count");
            nch = bf.nch;
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
    if (n.in_nch > 6 || n.out_nch > 6)
      return fail(WAA_ERR_OUT_OF_SCOPE, "the device path renders at most 6 channels per signal (node %u needs %d)", id,
                  std::max(n.in_nch, n.out_nch));
    changed |= n.in_nch != old_in || n.out_nch != old_out;
  }
  if (!changed || n_scc == 0) break;
  }
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
            std::vector<float> rate_q = param_per_quantum(b, p_rate, inst, nullptr);
            std::vector<float> det_q = param_per_quantum(b, p_det, inst, nullptr);
            const bool automated = !p_rate.blocks.empty() || !p_det.blocks.empty();
            const SchedKey key(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end, bf.frames, bf.sr,
                               rate_q[0], det_q[0]);
            int64_t endq;
            auto it = automated ? end_cache.end() : end_cache.find(std::make_pair(id, key));
            if (it != end_cache.end()) {
              endq = it->second;
            } else {
              SchedOut so;
              schedule_source(b, ss, bf.frames, bf.sr, true, rate_q, det_q, &so);
              endq = so.ended_quantum;
              if (!automated) end_cache[std::make_pair(id, key)] = endq;
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
      // findings
      for (uint32_t id : b->order) {
        Node& n = b->nodes[id];
        if (!n.live || n.in_nch <= 1 || n.in_edges.empty()) continue;
        const uint32_t kind = n.desc.kind;
        const bool line = kind == WAA_NODE_DELAY || (kind == WAA_NODE_CONVOLVER && n.has_ir);
        const bool sensitive = kind == WAA_NODE_BIQUAD || kind == WAA_NODE_IIR_FILTER || kind == WAA_NODE_STEREO_PANNER ||
                               kind == WAA_NODE_PANNER || line || n.in_nch > 2 || n.interp == WAA_INTERP_DISCRETE;
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
        if (what) {
          const bool keep_static = getenv("WAA_STATIC_CHANNEL_COUNTS") != nullptr;  // A/B aid: the round-1 behaviour
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
  // Nodes whose state freezes while they do not process (WaveShaper 2x / 4x, HRTF panner; waa_frozen.hip) follow the
  // per-quantum silence / count codes of their input EXACTLY.  Directly behind one source those codes are host-known
  // (the scheduling replay); behind anything else they come out of the dynamic-count rendering.
  std::vector<int> frozen_src(N, -1);
  for (uint32_t id = 0; id < N && !count_change_found && !b->force_dynamic; id++) {
    Node& n = b->nodes[id];
    if (!n.live || !is_frozen_node(n)) continue;
    bool simple = n.in_edges.size() == 1 && scc_of[id] < 0 && !getenv("WAA_FROZEN_DYNAMIC");
    if (simple) {
      const uint32_t p = b->edges[n.in_edges[0]].from;
      const uint32_t pk = b->nodes[p].desc.kind;
      simple = (pk == WAA_NODE_BUFFER_SOURCE || pk == WAA_NODE_CONSTANT_SOURCE || pk == WAA_NODE_OSCILLATOR) &&
               b->nodes[p].out_nch == n.in_nch && n.in_nch <= 2;
      if (simple) frozen_src[id] = (int)p;
    }
    if (!simple && !getenv("WAA_STATIC_CHANNEL_COUNTS")) {
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
  if (!count_change_found && !b->force_dynamic && !getenv("WAA_NO_LOOP_FOLD"))
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
    if (!n.live || n.desc.kind != WAA_NODE_DELAY || count_change_found || b->force_dynamic || getenv("WAA_NO_DELAY_FOLD")) continue;
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
          if (kind == WAA_NODE_GAIN && scc_of[id] < 0 && !count_change_found && !b->force_dynamic && !getenv("WAA_NO_EDGE_FOLD"))
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
  if (!getenv("WAA_NO_EDGE_FOLD"))
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
  for (uint32_t id = 0; id < N && !getenv("WAA_NO_SOURCE_VIEW"); id++) {
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
  if (!count_change_found && !b->force_dynamic && !getenv("WAA_NO_CONV_BIQUAD_FOLD") && !getenv("WAA_CONV_FFT_R4"))
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
      if (sn.desc.kind == WAA_NODE_BUFFER_SOURCE && !sn.materialized && !sn.is_view && !getenv("WAA_NO_SOURCE_VIEW")) {
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
  // planning units: single nodes and whole feedback loops, producers first (the condensed graph is acyclic),
  // otherwise in processing order
  struct Unit {
    int scc;
    uint32_t id;
  };
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
      int e = reduce_fan_in(b, ins, hn.in_nch, hn.interp);
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
        scc_of[id] < 0 && !getenv("WAA_NO_OSC_POST")) {
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
    if (!count_change_found)
      plan_note(b, "a feedback loop needs quantum-serial rendering with node kinds the loop kernel does not cover -> dyn_kernel");
    b->dynamic = true;
    b->code_stride = ((uint64_t)b->n_quanta + 15) & ~(uint64_t)15;
    const uint64_t cs = b->code_stride;
    for (uint32_t id = 0; id < N; id++) {
      const Node& n = b->nodes[id];
      // layouts up to 5.1 are rendered by dyn_kernel<6> (round 3) for Gain / Biquad / IIR / WaveShaper / the panners / DelayNodes
      // (whose line is then re-mixed in place when the count changes) / analysers (whose kernel follows the per-quantum codes)
      // / the destination; convolver and frozen-node inputs are stereo by their channel config: those stay mono / stereo
      const uint32_t k = n.desc.kind;
      const bool narrow_only = (k == WAA_NODE_CONVOLVER && n.has_ir) || is_frozen_node(n);
      if (n.live && narrow_only && (n.in_nch > 2 || n.out_nch > 2))
        return fail(WAA_ERR_OUT_OF_SCOPE,
                    "node %u: the reference's channel count changes mid-render and a signal is wider than stereo (%d channels): "
                    "exact dynamic counts above stereo are not rendered for this node kind",
                    id, std::max(n.in_nch, n.out_nch));
    }
    auto is_src = [&](uint32_t k) { return k == WAA_NODE_BUFFER_SOURCE || k == WAA_NODE_CONSTANT_SOURCE || k == WAA_NODE_OSCILLATOR; };
    auto alloc_codes = [&](uint8_t** out) -> int { return dev_alloc(b, out, (size_t)b->n_inst * cs); };
    for (auto& n : b->nodes) n.materialized = n.live;  // every signal is published
    // position of every vertex in the processing order
    std::map<uint32_t, size_t> vpos;
    for (size_t k = 0; k < items.size(); k++) vpos[items[k]] = k;
    std::vector<uint32_t> pending;        // vertices of the current group
    std::set<uint32_t> pending_nodes;     // their node ids
    std::vector<uint8_t> planned_node(N, 0);
    auto in_pending = [&](uint32_t node) { return pending_nodes.count(node) != 0; };
    // ---- host-known codes of a source: active quanta carry the source's channel count
    auto source_codes = [&](uint32_t id) -> int {
      std::vector<uint8_t> host;
      int e = source_code_rows(b, id, cs, &host);
      if (e) return e;
      uint8_t* dcode = nullptr;
      if ((e = dev_upload(b, &dcode, host))) return e;
      b->nodes[id].code = dcode;
      return 0;
    };
    // ---- one dyn_kernel launch for the pending vertices
    auto flush = [&]() -> int {
      if (pending.empty()) return 0;
      std::sort(pending.begin(), pending.end(), [&](uint32_t x, uint32_t y) { return vpos[x] < vpos[y]; });
      if (pending.size() > (size_t)DYN_MAX_ITEMS)
        return fail(WAA_ERR_OUT_OF_SCOPE, "more than %d nodes in one dynamic-count group", DYN_MAX_ITEMS);
      std::map<uint32_t, int> out_item, writer_item;  // node id -> item producing its output / its delay line
      for (size_t k = 0; k < pending.size(); k++) {
        const uint32_t v = pending[k], id = v & ~VTX_READER;
        if (is_delay(b, id) && !(v & VTX_READER))
          writer_item[id] = (int)k;
        else
          out_item[id] = (int)k;
      }
      std::vector<DynItem> host(pending.size());
      Step st;
      st.kind = 10;
      std::string desc;
      for (size_t k = 0; k < pending.size(); k++) {
        const uint32_t v = pending[k], id = v & ~VTX_READER;
        Node& n = b->nodes[id];
        DynItem& li = host[k];
        std::memset(&li, 0, sizeof li);
        const bool reader = is_delay(b, id) && (v & VTX_READER);
        li.cc = n.cc;
        li.mode = n.mode;
        li.interp = n.interp;
        li.code_stride = cs;
        li.writer_item = -1;
        if (!reader) {
          if (n.in_edges.size() > (size_t)DYN_MAX_IN)
            return fail(WAA_ERR_OUT_OF_SCOPE, "more than %d inputs on node %u of a dynamic-count graph", DYN_MAX_IN, id);
          li.n_in = (int)n.in_edges.size();
          for (int j = 0; j < li.n_in; j++) {
            const uint32_t pid = b->edges[n.in_edges[j]].from;
            Node& pn = b->nodes[pid];
            DynInput& in = li.in[j];
            auto it = out_item.find(pid);
            if (it != out_item.end()) {
              // same quantum through LDS; a producer that renders LATER in the quantum is only legal for ... nothing:
              // the cycle breaker guarantees producers first, except through a delay reader (which reads the line)
              if (it->second >= (int)k) return fail(WAA_ERR_INVALID_STATE, "internal: dynamic group member order (node %u)", id);
              in.item = it->second;
            } else {
              if (!pn.sig.base || !planned_node[pid])
                return fail(WAA_ERR_INVALID_STATE, "internal: input %u of node %u is not planned yet", pid, id);
              in.item = -1;
              in.nch = pn.out_nch;
              in.sig = pn.sig;
              in.code = pn.code;
              in.remap = pn.remap;
              in.code_stride = cs;
              st.loop_reads.push_back(pn.sig.base);
            }
          }
        }
        char t[64];
        const uint32_t kind = n.desc.kind;
        if (is_delay(b, id)) {
          if (!reader) {
            li.kind = DI_DELAY_W;
            li.num_quanta = (int32_t)std::ceil(n.desc.d[0] * (double)b->sr / (double)RQ);  // (ring capacity - 1, as for the reader)
            int e = temp_signal(b, n.in_nch, &li.out);  // the delay line in absolute time, native layout
            if (e) return e;
            li.nch_pub = n.in_nch;
            if ((e = dev_alloc(b, &li.aux32, (size_t)b->n_inst * cs))) return e;
            snprintf(t, sizeof t, "delayW%u", id);
          } else {
            li.kind = DI_DELAY_R;
            li.out = n.sig;
            li.nch_pub = n.out_nch;
            li.writer_item = writer_item.at(id);
            li.in_cycle = li.writer_item > (int)k ? 1 : 0;
            li.num_quanta = (int32_t)std::ceil(n.desc.d[0] * (double)b->sr / (double)RQ);
            int e = node_param(b, id, WAA_PARAM_DELAY_DELAY_TIME, &li.op.p0);
            if (e) return e;
            if ((e = alloc_codes(&n.code))) return e;
            li.out_code = n.code;
            snprintf(t, sizeof t, "delayR%u%s", id, li.in_cycle ? "(clamped)" : "");
          }
        } else {
          li.kind = DI_NODE;
          li.out = n.sig;
          li.nch_pub = n.out_nch;
          int e = alloc_codes(&n.code);
          if (e) return e;
          li.out_code = n.code;
          std::vector<OpDesc> ops;
          int out_nch = 0;
          if ((kind == WAA_NODE_STEREO_PANNER || kind == WAA_NODE_PANNER) && !is_frozen_node(n)) {
            // both laws: the gains for a mono input (alt) and for a stereo input (op)
            const int keep = n.in_nch;
            n.in_nch = 1;
            e = emit_node_ops(b, id, 1, true, ops, &out_nch);
            if (!e) {
              li.alt1 = ops[0].p1;
              li.alt2 = ops[0].p2;
              ops.clear();
              n.in_nch = 2;
              e = emit_node_ops(b, id, 2, true, ops, &out_nch);
            }
            n.in_nch = keep;
            if (e) return e;
          } else if ((kind == WAA_NODE_CONVOLVER && n.has_ir) || is_frozen_node(n)) {
            // (the mixed input of the node; its node-major steps follow the launch)
          } else {
            if ((e = emit_node_ops(b, id, n.in_nch, true, ops, &out_nch))) return e;
          }
          if (ops.size() > 1) return fail(WAA_ERR_DEVICE, "internal: node %u emitted %zu ops", id, ops.size());
          li.dk = DK_PASS;
          if (!ops.empty()) {
            li.op = ops[0];
            switch (ops[0].kind) {
              case OP_GAIN: li.dk = DK_GAIN; break;
              case OP_BIQUAD: li.dk = DK_BIQUAD; break;
              case OP_IIR: li.dk = DK_IIR; break;
              case OP_WAVESHAPER: li.dk = DK_WAVESHAPER; break;
              case OP_STEREO_PAN: li.dk = DK_STEREO_PAN; break;
              case OP_PANNER: li.dk = DK_PANNER; break;
              default: return fail(WAA_ERR_DEVICE, "internal: op %d in a dynamic-count group", ops[0].kind);
            }
          }
          if (kind == WAA_NODE_WAVESHAPER && !is_frozen_node(n)) {
            li.dk = DK_WAVESHAPER;  // (without a curve: ptr0 == null, output = input)
            const size_t cn = n.curve.size();
            const float mid = cn == 0 ? 0.f : (cn % 2 ? n.curve[cn / 2] : (n.curve[cn / 2 - 1] + n.curve[cn / 2]) / 2.f);
            li.flags = (!n.has_curve || cn == 0 || std::fabs(mid) < 1e-9f) ? 1 : 0;
          }
          if (kind == WAA_NODE_CONVOLVER && !n.has_ir) li.flags |= 2;
          if (kind == WAA_NODE_ANALYSER) li.publish_upmix = 1;  // the analyser FFT reads a static stereo signal
          if (is_frozen_node(n)) {
            // the item publishes the node's INPUT (n.hist) and its codes; n.sig is written by the node-major steps
            li.dk = DK_CONV_IN;
            int e2 = temp_signal(b, n.in_nch, &n.hist);
            if (e2) return e2;
            li.out = n.hist;
            li.nch_pub = n.in_nch;
            if ((e2 = alloc_codes(&n.in_code))) return e2;
            li.out_code = n.in_code;
            li.publish_upmix = 1;
          }
          if (kind == WAA_NODE_CONVOLVER && n.has_ir) {
            li.dk = DK_CONV_IN;
            // the item publishes the convolver's INPUT (n.hist); its output signal n.sig is written by the FFT steps
            int e2 = temp_signal(b, n.in_nch, &n.hist);
            if (e2) return e2;
            li.out = n.hist;
            li.nch_pub = n.in_nch;
            if ((e2 = alloc_codes(&n.in_code))) return e2;
            li.out_code = n.in_code;
            if (n.ir_nch == 1 && n.in_nch == 2) {
              li.compact_ch1 = 1;
              if ((e2 = dev_alloc(b, &n.remap, (size_t)b->n_inst * cs))) return e2;
              li.aux32 = n.remap;
              // channel 1 is written in compacted time: quanta it never reaches must read as zeros
              b->state_bufs.push_back({n.hist.base, (size_t)b->n_inst * n.hist.inst_stride * sizeof(float)});
            } else {
              li.publish_upmix = 1;
            }
          }
          snprintf(t, sizeof t, "%s%u", li.dk == DK_PASS ? "pass" : li.dk == DK_CONV_IN ? "convIn" : op_name(li.op.kind), id);
        }
        st.loop_writes.push_back(li.out.base);
        desc += desc.empty() ? t : std::string(",") + t;
      }
      DynItem* dev = nullptr;
      int e = dev_upload(b, &dev, host);
      if (e) return e;
      DynDesc& d = st.dyn;
      std::memset(&d, 0, sizeof d);
      d.items = dev;
      d.n_items = (int32_t)host.size();
      d.n_inst = b->n_inst;
      d.n_quanta = b->n_quanta;
      d.sample_rate = (double)b->sr;
      d.quantum_duration = (double)RQ * (1. / (double)b->sr);  // delay.rs:546-548
      d.cmax = 1;
      for (uint32_t v : pending) {
        const Node& pn = b->nodes[v & ~VTX_READER];
        d.cmax = std::max(d.cmax, std::max(pn.in_nch, pn.out_nch));
        for (int e : pn.in_edges) d.cmax = std::max(d.cmax, b->nodes[b->edges[e].from].out_nch);
      }
      if (dyn_lds_bytes(d.n_items, d.cmax) > 160 * 1024)
        return fail(WAA_ERR_OUT_OF_SCOPE, "dynamic-count group of %d nodes with %d-channel signals does not fit the kernel's local memory",
                    d.n_items, d.cmax);
      st.profile_slot = slot_for(b, "dyn_kernel");
      b->steps.push_back(st);
      plan_note(b, "dynamic-count group: %d item(s) per quantum [%s]", d.n_items, desc.c_str());
      for (uint32_t v : pending) planned_node[v & ~VTX_READER] = 1;
      pending.clear();
      pending_nodes.clear();
      return 0;
    };
    for (const Unit& unit : units) {
      std::vector<uint32_t> verts;
      if (unit.scc >= 0) {
        for (uint32_t v : items)
          if (scc_of[v & ~VTX_READER] == unit.scc) verts.push_back(v);
      } else {
        if (is_delay(b, unit.id)) verts.push_back(unit.id);
        verts.push_back(is_delay(b, unit.id) ? (unit.id | VTX_READER) : unit.id);
      }
      bool any_live = false;
      for (uint32_t v : verts) any_live |= b->nodes[v & ~VTX_READER].live;
      if (!any_live) continue;
      if (unit.scc >= 0)
        for (uint32_t v : verts) {
          const Node& m = b->nodes[v & ~VTX_READER];
          if (m.desc.kind == WAA_NODE_CONVOLVER && m.has_ir)
            return fail(WAA_ERR_OUT_OF_SCOPE, "a ConvolverNode inside a feedback loop is out of scope (node %u)", v & ~VTX_READER);
          if (is_frozen_node(m))
            return fail(WAA_ERR_OUT_OF_SCOPE, "an oversampled WaveShaperNode / HRTF PannerNode inside a feedback loop is out of scope (node %u)",
                        v & ~VTX_READER);
        }
      // AudioParam inputs are summed by a node-major chain in front of the group: their producers must be complete
      bool param_dep = false;
      for (uint32_t v : verts)
        for (auto& pe : b->nodes[v & ~VTX_READER].pin_edges)
          for (int e : pe) {
            const uint32_t from = b->edges[e].from;
            if (unit.scc >= 0 && scc_of[from] == unit.scc)
              return fail(WAA_ERR_OUT_OF_SCOPE, "an AudioParam of node %u is modulated from inside its own feedback loop",
                          v & ~VTX_READER);
            param_dep |= in_pending(from);
          }
      if (param_dep)
        if (int e = flush()) return e;
      const uint32_t id = unit.id;
      Node& n = b->nodes[id];
      if (unit.scc < 0 && is_src(n.desc.kind)) {
        int e = alloc_signal(n);
        if (e) return e;
        if (n.desc.kind == WAA_NODE_OSCILLATOR)
          e = plan_oscillator(b, id);
        else
          e = plan_single(id);
        if (e) return e;
        if ((e = source_codes(id))) return e;
        planned_node[id] = 1;
        continue;
      }
      for (uint32_t v : verts) {
        Node& m = b->nodes[v & ~VTX_READER];
        if (!m.sig.base) {
          int e = alloc_signal(m);
          if (e) return e;
        }
        if (std::find(pending.begin(), pending.end(), v) == pending.end()) pending.push_back(v);
        pending_nodes.insert(v & ~VTX_READER);
      }
      if (unit.scc < 0 && is_frozen_node(n)) {
        // the group ends with the node's mixed input and its codes; then the link table (which also writes the node's
        // output codes) and the node-major steps
        if (int e = flush()) return e;
        if (!n.in_code) return fail(WAA_ERR_INVALID_STATE, "internal: input codes of node %u", id);
        int e = n.desc.kind == WAA_NODE_PANNER ? plan_hrtf(b, id, -1) : plan_oversampler(b, id, -1);
        if (e) return e;
      }
      if (unit.scc < 0 && n.desc.kind == WAA_NODE_CONVOLVER && n.has_ir) {
        // the group ends with the convolver's mixed input; then the node-major FFT steps and the code kernel
        if (int e = flush()) return e;
        uint8_t* in_code = n.in_code;  // published by the DK_CONV_IN item of the group just flushed
        if (!in_code) return fail(WAA_ERR_INVALID_STATE, "internal: convolver input codes");
        int e = plan_convolver(b, id);
        if (e) return e;
        if ((e = alloc_codes(&n.code))) return e;
        Step cst;
        cst.kind = 11;
        ConvCodeDesc& cd = cst.ccode;
        std::memset(&cd, 0, sizeof cd);
        cd.in_code = in_code;
        cd.out_code = n.code;
        cd.code_stride = cs;
        cd.impulse_length = n.ir_len;
        cd.ir_nch = n.ir_nch;
        cd.n_inst = b->n_inst;
        cd.n_quanta = b->n_quanta;
        // (a mono impulse response on a stereo input keeps channel 1 in compacted time: only channel 0 is cleared in place)
        cd.cout = (n.ir_nch == 1 && n.in_nch == 2) ? 1 : n.out_nch;
        cd.out = n.sig;
        if ((e = dev_alloc(b, &cd.clean, (size_t)b->n_inst * cs))) return e;
        cst.loop_writes.push_back(n.sig.base);
        cst.profile_slot = slot_for(b, "conv_code_kernel");
        b->steps.push_back(cst);
      }
    }
    if (int e = flush()) return e;
    if (getenv("WAA_DEBUG_REVERSE_PLAN")) std::reverse(b->steps.begin(), b->steps.end());
    if (int e = validate_plan(b)) return e;
    b->planned = true;
    return 0;
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
        if (e == WAA_ERR_OUT_OF_SCOPE && !b->force_dynamic && !getenv("WAA_STATIC_CHANNEL_COUNTS")) {
          // the static loop kernel covers Gain / Biquad / WaveShaper / k-rate StereoPanner / Delay members only; the
          // dynamic-count kernel renders every node kind quantum by quantum (waa_dyn.hip): plan the graph again with it
          b->force_dynamic = true;
          b->steps.clear();
          b->group_tiles.clear();
          b->state_bufs.clear();
          b->plan_log.clear();
          for (auto& nd : b->nodes) {
            nd.sig = SignalRef{};
            nd.hist = SignalRef{};
            nd.hist_is_temp = false;
          }
          return build_plan(b);
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
        for (size_t k = first_step; k < b->steps.size(); k++)
          if (!b->steps[k].prologue) {
            n_body++;
            body = k;
          }
        if (n_body == 1 && b->steps[body].kind == 0 && !getenv("WAA_NO_ECHO_RING")) {
          float range[2] = {1e30f, 0.f};
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
          ChainDesc cd = b->steps[body].chain;
          cd.tile0 = 0;
          cd.tile1 = b->n_tiles;
          int chunk = 0;
          const int fb = echo_ring_applicable(cd, range, &chunk);
          if (fb >= 0) {
            b->steps[body].echo_fb = fb;
            b->steps[body].echo_chunk = chunk;
            b->steps[body].profile_slot = slot_for(b, "echo_ring_kernel");
            plan_note(b, "  ... rendered by the LDS-ring kernel in ONE launch: delay %.0f .. %.0f frames, chunks of %d frames, the line's last %d frames stay in LDS",
                      (double)range[0], (double)range[1], chunk * 256, 16384);
          }
        }
      }
      continue;
    }
    int e = plan_single(id);
    if (e) return e;
  }
  fuse_echo_tails(b);
  ring_feed_forward_echoes(b);
  // (self-test of the check below: a reversed launch list must not get past it, tests/test_plan.py)
  if (getenv("WAA_DEBUG_REVERSE_PLAN")) std::reverse(b->steps.begin(), b->steps.end());
  if (int e = validate_plan(b)) return e;
  b->planned = true;
  return 0;
}

// ---- plan validation -----------------------------------------------------------------------------------------
// The plan is a linear list of launches over shared device buffers; nothing but their order makes a consumer see
// its producer's data.  This check walks the list once and refuses a plan in which a launch reads a buffer that
// some launch of the plan writes, but none has written yet — an ordering bug of the planner would otherwise
// render stale or zero data silently.  The only legal read-before-write is a DelayNode reader inside a feedback
// loop (it reads the PREVIOUS quanta of a line that is filled later in the same pass).
namespace {
void io_param(const ParamRef& p, StepIo& io) {
  if (p.base && p.mode == 2) io.reads.push_back(p.base);  // per-frame values: possibly produced by a param chain
}
void io_input(const InputRef& in, StepIo& io) {
  if (in.kind == IN_SIGNAL || (in.kind == IN_DELAYED && !in.feedback)) io.reads.push_back(in.sig.base);
  if (in.kind == IN_CONSTANT) io_param(in.offset, io);
  if (in.has_gain) io_param(in.gain, io);
}
StepIo step_io(const Step& st) {
  StepIo io;
  switch (st.kind) {
    case 0: {
      const ChainDesc& c = st.chain;
      for (int k = 0; k < c.n_inputs; k++) io_input(c.in[k], io);
      for (int o = 0; o < c.n_ops; o++) {
        const OpDesc& op = c.ops[o];
        io_param(op.p0, io);
        io_param(op.p1, io);
        io_param(op.p2, io);
        io_param(op.p3, io);
        io_param(op.p4, io);
        if (op.kind == OP_BIQUAD && op.i0 == 2) io.reads.push_back(op.ptr0);  // per-frame coefficient table
      }
      io.writes.push_back(c.out.base);
      break;
    }
    case 1:
      io_input(st.bq.in, io);
      if (st.bq.vary >= 2) io.reads.push_back(st.bq.coefs);
      if (st.bq.vary == 3) io.reads.push_back(st.bq.hp);
      io.writes.push_back(st.bq.out.base);
      break;
    case 2:
    case 4:
      io.reads.push_back(st.conv.in.base);
      io.writes.push_back(st.conv.out.base);
      break;
    case 3:
      io.writes.push_back(st.zero_ptr);
      break;
    case 5:
      io_param(st.coef.frequency, io);
      io_param(st.coef.detune, io);
      io_param(st.coef.q, io);
      io_param(st.coef.gain, io);
      io.writes.push_back(st.coef.coefs);
      break;
    case 14:
      io.writes.push_back(st.tl.out);
      break;
    case 13:
      for (int k = 0; k < 15; k++) io_param(st.geom.p[k], io);
      io.writes.push_back(st.geom.az);
      io.writes.push_back(st.geom.gl_mono);
      io.writes.push_back(st.geom.gr_mono);
      io.writes.push_back(st.geom.gl_stereo);
      io.writes.push_back(st.geom.gr_stereo);
      io.writes.push_back(st.geom.dg);
      io.writes.push_back(st.geom.cg);
      break;
    case 12:
      if (st.hp.coefs) {
        io.reads.push_back(st.hp.coefs);
        io.writes.push_back(st.hp.hp);
      }
      break;
    case 18:
      io.reads.push_back(st.lanes.coefs);
      io.writes.push_back(st.lanes.ht);
      break;
    case 19:
      io_input(st.lanes.in, io);
      io.reads.push_back(st.lanes.coefs);
      io.reads.push_back(st.lanes.ht);
      io.writes.push_back(st.lanes.out.base);
      break;
    case 6:
      io_input(st.iir.in, io);
      io.writes.push_back(st.iir.out.base);
      break;
    case 7:
      io.reads.push_back(st.delay.in.base);
      io_param(st.delay.delay, io);
      io.writes.push_back(st.delay.out.base);
      io.feedback_reader = st.delay.in_cycle != 0;
      break;
    case 8:
    case 10:
    case 16:
    case 17:
    case 20:
      io.reads = st.loop_reads;
      io.writes = st.loop_writes;
      break;
    case 9:
      io_param(st.osc.frequency, io);
      io_param(st.osc.detune, io);
      io.writes.push_back(st.osc.out.base);
      break;
    default:
      break;
  }
  return io;
}
}  // namespace

// An echo loop rendered by the LDS-ring kernel (Step::echo_fb): when the line it writes has exactly ONE reader in the whole plan
// and that reader is a plain sum of the delayed line and of signals the loop step reads too (the destination's  dry + wet),
// the ring kernel renders that sum as well and the line is never stored (waa_echo.hip, "the tail").  Decided on the finished
// launch list, buffer by buffer, with the same read / write sets the validation below uses; any launch kind those sets do
// not describe keeps the plan as it is.
void fuse_echo_tails(waa_batch* b) {
  if (getenv("WAA_NO_ECHO_TAIL")) return;
  for (const Step& st : b->steps)
    if (st.kind == 11 || st.kind == 15 || st.kind > 20) return;
  for (size_t l = 0; l < b->steps.size(); l++) {
    Step& ls = b->steps[l];
    if (ls.kind != 0 || ls.echo_fb < 0) continue;
    const void* line = ls.chain.out.base;
    size_t reader = 0;
    int n_readers = 0;
    bool other_writer = false;
    for (size_t k = 0; k < b->steps.size(); k++) {
      if (k == l) continue;
      const Step& sk = b->steps[k];
      const StepIo io = step_io(sk);
      // (a delayed read marked `feedback` is left out of the read sets: the validation's legal read-before-write)
      auto delayed_from = [&](const InputRef& in) { return in.kind == IN_DELAYED && in.sig.base == line; };
      bool reads = std::find(io.reads.begin(), io.reads.end(), line) != io.reads.end();
      if (sk.kind == 0)
        for (int q = 0; q < sk.chain.n_inputs; q++) reads |= delayed_from(sk.chain.in[q]);
      reads |= (sk.kind == 1 && delayed_from(sk.bq.in)) || (sk.kind == 6 && delayed_from(sk.iir.in)) ||
               (sk.kind == 19 && delayed_from(sk.lanes.in));
      if (reads) {
        n_readers++;
        reader = k;
      }
      other_writer |= std::find(io.writes.begin(), io.writes.end(), line) != io.writes.end();
    }
    if (n_readers != 1 || other_writer || reader < l) {
      plan_note(b, "echo loop: the delay line has %d reader(s) outside the loop: stored, read by them from memory", n_readers);
      continue;
    }
    // Readers that are not launches: an AnalyserNode (pulled by analyser_kernel after the render) or the destination
    // (downloaded) that ALIASES the line through an identity node of the loop never shows up in the read sets above.
    // The line must then be stored for them (the same class as the oscillator post-op fold, fuzz seed 502310).
    int alias_reader = -1;
    for (size_t k = 0; k < b->nodes.size(); k++) {
      const Node& an = b->nodes[k];
      if (an.live && an.sig.base == line &&
          (an.desc.kind == WAA_NODE_ANALYSER || an.desc.kind == WAA_NODE_DESTINATION))
        alias_reader = (int)k;
    }
    if (alias_reader >= 0) {
      plan_note(b, "echo loop: node %d (analyser / destination) aliases the loop's delay line and is read outside the launch list: the line is stored", alias_reader);
      continue;
    }
    Step& ts = b->steps[reader];
    EchoTail t{};
    const char* why = "it is not an element-wise launch";
    if (ts.kind != 0 || ts.group >= 0 || !echo_tail_applicable(ls.chain, ls.echo_fb, ts.chain, &t, &why)) {
      plan_note(b, "echo loop: launch %zu, the only reader of the delay line, is not a plain sum of the delayed line and of the loop's inputs (%s): the line is stored", reader, why);
      continue;
    }
    t.store_line = 0;
    ls.echo_tail = t;
    ls.echo_tail_step = (int)reader;
    ts.echo_fused = true;
    plan_note(b, "echo loop: launch %zu (the only reader of the loop's delay line: %d input(s) -> %d channel(s)) is rendered by the LDS-ring kernel too; the line is not stored",
              reader, t.n_inputs, t.in_nch);
  }
}

// The feed-forward echo  out = X + g * delayed(X)  as a chain launch reads X twice (the second time mostly out of L2) from
// short-lived wavefronts, one per 256 frames: 3.8 TB/s on its compulsory bytes.  The ring kernel walks every instance's
// stream with two chunks in flight and takes the delayed samples from LDS: 5 TB/s — when there is at least one instance
// per CU to walk (WAA_ECHO_FF_MIN_INST, default 256: below that the tile-parallel launch fills the device better).
void ring_feed_forward_echoes(waa_batch* b) {
  if (getenv("WAA_NO_ECHO_RING") || getenv("WAA_NO_ECHO_FF")) return;
  const char* mi = getenv("WAA_ECHO_FF_MIN_INST");
  if (b->n_inst < (uint32_t)(mi ? atoi(mi) : 256)) return;
  for (size_t k = 0; k < b->steps.size(); k++) {
    Step& st = b->steps[k];
    if (st.kind != 0 || st.group >= 0 || st.echo_fused || st.chain.n_ops != 0) continue;
    bool any = false;
    for (int q = 0; q < st.chain.n_inputs; q++) any |= st.chain.in[q].kind == IN_DELAYED;
    if (!any) continue;
    const char* why = "";
    ChainDesc line{};
    EchoTail t{};
    const int chunk = echo_feed_forward(st.chain, &line, &t, &why);
    if (!chunk) {
      plan_note(b, "launch %zu sums a delayed signal but keeps the tile-parallel kernel: %s", k, why);
      continue;
    }
    float delayed_lo = 0.f, delayed_hi = 0.f;
    for (int q = 0; q < st.chain.n_inputs; q++)
      if (st.chain.in[q].kind == IN_DELAYED) {
        delayed_lo = st.chain.in[q].delay_lo;
        delayed_hi = st.chain.in[q].delay_hi;
      }
    st.echo_ff = true;
    st.echo_line = line;
    st.echo_tail = t;
    st.echo_chunk = chunk;
    st.profile_slot = slot_for(b, "echo_ring_kernel");
    plan_note(b, "launch %zu (delayed signal + %d more input(s), no ops) is rendered by the LDS-ring kernel with nothing fed back: delay %.0f .. %.0f frames, chunks of %d frames",
              k, t.n_inputs - 1, (double)delayed_lo, (double)delayed_hi, chunk * 256);
  }
}

int validate_plan(waa_batch* b) {
  std::vector<StepIo> ios;
  std::set<const void*> produced, written;
  for (const Step& st : b->steps) {
    ios.push_back(step_io(st));
    for (const void* w : ios.back().writes)
      if (w) produced.insert(w);
  }
  for (size_t k = 0; k < b->steps.size(); k++) {
    const StepIo& io = ios[k];
    if (b->steps[k].kind == 8 || b->steps[k].kind == 10) {  // the items of a quantum-serial launch hand over inside the kernel
      for (const void* w : io.writes) written.insert(w);
    }
    for (const void* r : io.reads) {
      if (!r || !produced.count(r) || written.count(r)) continue;
      if (io.feedback_reader && r == b->steps[k].delay.in.base) continue;
      return fail(WAA_ERR_INVALID_STATE, "internal: launch %zu of the plan (kind %d) reads a buffer that a later launch produces", k,
                  b->steps[k].kind);
    }
    for (const void* w : io.writes)
      if (w) written.insert(w);
  }
  return 0;
}

// Resolve a source node into an InputRef: schedules, per-instance buffer table, constant ranges.
int prepare_source_input(waa_batch* b, uint32_t id, InputRef* in) {
  Node& n = b->nodes[id];
  if (n.desc.kind == WAA_NODE_CONSTANT_SOURCE) {
    int e = node_param(b, id, 0, &in->offset);
    if (e) return e;
    // active frame range per instance (constant_source.rs:203-258), found by replaying the quantum loop
    std::vector<int64_t> act((size_t)b->n_inst * 2);
    const double dt = 1. / (double)b->sr;
    for (uint32_t i = 0; i < b->n_inst; i++) {
      const double start = n.sched[i].start, stop = n.sched[i].stop;
      int64_t a0 = -1, a1 = -1;
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        const double ct = (double)((uint64_t)q * RQ) / (double)b->sr;
        const double nbt = ct + dt * (double)RQ;
        if (start >= nbt) continue;
        if (start <= ct && stop >= nbt) {
          if (a0 < 0) a0 = (int64_t)q * RQ;
          a1 = (int64_t)(q + 1) * RQ;
        } else {
          double t = ct;
          for (int s = 0; s < RQ; s++) {
            if (!(t < start || t >= stop)) {
              if (a0 < 0) a0 = (int64_t)q * RQ + s;
              a1 = (int64_t)q * RQ + s + 1;
            }
            t += dt;
          }
        }
        if (stop <= nbt) break;
      }
      act[(size_t)i * 2] = a0 < 0 ? 0 : a0;
      act[(size_t)i * 2 + 1] = a0 < 0 ? 0 : a1;
    }
    int64_t* d = nullptr;
    e = dev_upload(b, &d, act);
    if (e) return e;
    in->active = d;
    plan_note(b, "constant source node %u: active frames [%lld, %lld) for instance 0", id, (long long)act[0], (long long)act[1]);
    return 0;
  }
  // AudioBufferSourceNode
  std::vector<SrcInst> insts(b->n_inst);
  std::vector<SrcSchedule> scheds;
  std::vector<std::pair<int64_t, uint32_t>> linear;  // per schedule: (linear_start, fast_prefix)
  std::vector<uint32_t> linear_all;                   // per schedule: the whole render is that linear run
  std::map<SchedKey, uint32_t> dedup;
  const ParamStore& p_rate = n.params[WAA_PARAM_SOURCE_PLAYBACK_RATE];
  const ParamStore& p_det = n.params[WAA_PARAM_SOURCE_DETUNE];
  const bool automated = !p_rate.blocks.empty() || !p_det.blocks.empty();
  for (uint32_t i = 0; i < b->n_inst; i++) {
    const DeviceBuffer& bf = n.bufs[i];
    SrcInst& si = insts[i];
    si.base = bf.base;
    si.ch_stride = bf.ch_stride;
    si.frames = bf.frames;
    si.aligned = (bf.valid && ((uintptr_t)bf.base % 16 == 0) && (bf.ch_stride % 4 == 0)) ? 1 : 0;
    std::vector<float> rate_q = param_per_quantum(b, p_rate, i, nullptr);
    std::vector<float> det_q = param_per_quantum(b, p_det, i, nullptr);
    const SourceSched& ss = n.sched[i];
    const SchedKey key(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end,
                       bf.valid ? bf.frames : 0, bf.valid ? bf.sr : 0.f, rate_q[0], det_q[0]);
    if (!automated) {
      auto it = dedup.find(key);
      if (it != dedup.end()) {
        si.sched = it->second;
        continue;
      }
    }
    SchedOut so;
    schedule_source(b, n.sched[i], bf.frames, bf.sr, bf.valid, rate_q, det_q, &so);
    {
      uint32_t nf = 0, nl = 0, ns = 0, nt = 0;
      for (auto& r : so.qrec) {
        nf += r.mode == Q_FAST;
        nl += r.mode == Q_FAST_LOOP;
        ns += r.mode == Q_SLOW;
      }
      for (auto t : so.tile_fast) nt += t;
      plan_note(b, "source node %u schedule %zu: quanta fast=%u fast_loop=%u slow=%u silent=%u fast_tiles=%u/%u", id,
                scheds.size(), nf, nl, ns, (uint32_t)so.qrec.size() - nf - nl - ns, nt, b->n_tiles);
    }
    {
      // leading tiles that are fast and form one linear run of the buffer
      int64_t start0 = 0;
      uint32_t prefix = 0;
      if (!so.tile_fast.empty() && so.tile_fast[0]) {
        start0 = so.qrec[0].start;
        while (prefix < b->n_tiles && so.tile_fast[prefix] &&
               so.qrec[(size_t)prefix * QUANTA_PER_TILE].start == start0 + (int64_t)prefix * TILE)
          prefix++;
      }
      linear.push_back({start0, prefix});
      // ... and the render's last, partial tile continues that run as far as the render goes (quanta behind the render's end do
      // not exist): the whole render is one linear run — consumers treat the source like a signal of n_quanta * 128 frames
      bool all = prefix == b->n_tiles;
      if (prefix + 1 == b->n_tiles && (size_t)prefix * QUANTA_PER_TILE < (size_t)b->n_quanta) {
        all = true;
        for (size_t q = (size_t)prefix * QUANTA_PER_TILE; q < (size_t)b->n_quanta && q < so.qrec.size(); q++)
          all = all && so.qrec[q].mode == Q_FAST && so.qrec[q].start == start0 + (int64_t)q * RQ;
      }
      linear_all.push_back(all ? 1u : 0u);
      plan_note(b, "source node %u schedule %zu: tiles [0, %u) are one linear run from buffer frame %lld%s", id, scheds.size(), prefix,
                (long long)start0, all && prefix < b->n_tiles ? " (and so is the rest of the render)" : "");
    }
    SrcSchedule ds{};
    QRec* dq = nullptr;
    int e = dev_upload(b, &dq, so.qrec);
    if (e) return e;
    ds.qrec = dq;
    if (so.any_slow) {
      SlowRec* dsr = nullptr;
      e = dev_upload(b, &dsr, so.slow);
      if (e) return e;
      ds.slow = dsr;
    }
    uint8_t* dtf = nullptr;
    e = dev_upload(b, &dtf, so.tile_fast);
    if (e) return e;
    ds.tile_fast = dtf;
    si.sched = (uint32_t)scheds.size();
    scheds.push_back(ds);
    if (!automated) dedup[key] = si.sched;
  }
  for (auto& si : insts) {
    si.sc = scheds[si.sched];
    si.linear_start = linear[si.sched].first;
    si.fast_prefix = si.aligned && !getenv("WAA_NO_LINEAR_PREFIX") ? linear[si.sched].second : 0;  // (switch: A/B aid)
    si.linear_all = si.fast_prefix ? linear_all[si.sched] : 0;
  }
  SrcInst* d_insts = nullptr;
  int e = dev_upload(b, &d_insts, insts);
  if (e) return e;
  SrcSchedule* d_scheds = nullptr;
  e = dev_upload(b, &d_scheds, scheds);
  if (e) return e;
  in->src = d_insts;
  in->sched = d_scheds;
  in->fast_tiles = b->n_tiles;
  for (auto& si : insts) in->fast_tiles = std::min(in->fast_tiles, !si.base ? 0u : (si.linear_all ? b->n_tiles : si.fast_prefix));
  plan_note(b, "source node %u: %zu distinct schedule(s) for %u instance(s)", id, scheds.size(), b->n_inst);
  return 0;
}

// Fan-in above MAX_INPUTS: sum the first MAX_INPUTS inputs (mixed to the receiver's channel count) into a
// temporary signal and continue; the left-to-right order of the f32 additions (graph.rs:524-535) is kept.
int reduce_fan_in(waa_batch* b, std::vector<InputRef>& ins, int in_nch, int interp) {
  while (ins.size() > (size_t)MAX_INPUTS) {
    float* ptr = nullptr;
    int e = dev_alloc(b, &ptr, (size_t)b->n_inst * in_nch * b->lp);
    if (e) return e;
    Step st;
    ChainDesc& cd = st.chain;
    std::memset(&cd, 0, sizeof cd);
    cd.n_inputs = MAX_INPUTS;
    for (int k = 0; k < MAX_INPUTS; k++) cd.in[k] = ins[k];
    cd.in_nch = in_nch;
    cd.in_interp = interp;
    cd.out = SignalRef{ptr, (uint64_t)in_nch * b->lp, b->lp, in_nch, 0};
    cd.n_inst = b->n_inst;
    cd.n_tiles = b->n_tiles;
    cd.tile0 = 0;
    cd.tile1 = b->n_tiles;
  cd.tile0 = 0;
  cd.tile1 = b->n_tiles;
    cd.n_quanta = b->n_quanta;
    int cmax = in_nch;
    for (int k = 0; k < MAX_INPUTS; k++) cmax = std::max(cmax, ins[k].nch);
    st.cmax = cmax;
    st.profile_slot = slot_for(b, cmax <= 1 ? "chain_kernel<1>" : "chain_kernel<2>");
    b->steps.push_back(st);
    InputRef partial{};
    partial.kind = IN_SIGNAL;
    partial.nch = in_nch;
    partial.sig = cd.out;
    ins.erase(ins.begin(), ins.begin() + MAX_INPUTS);
    ins.insert(ins.begin(), partial);
    plan_note(b, "fan-in partial sum of %d inputs -> %dch", MAX_INPUTS, in_nch);
  }
  return 0;
}

// ConvolverNode with an impulse response (convolver.rs:259-317, 343-490): input mix chain (if needed)
// + forward FFT / spectral MAC / inverse FFT steps.
// Input of a node-major step (convolver, delay): the single producer's signal if its channel count already
// matches, else a mixing chain into a temporary.
int node_input_signal(waa_batch* b, uint32_t id, SignalRef* out_sig, const SignalRef* target, uint64_t* valid) {
  Node& n = b->nodes[id];
  if (valid) *valid = b->lp;
  if (!target && n.in_edges.size() == 1) {
    Node& p = b->nodes[b->edges[n.in_edges[0]].from];
    if (p.is_view && valid) {  // a source read in place (see build_plan)
      *out_sig = p.view_sig;
      *valid = p.view_valid;
      return 0;
    }
    if (p.materialized && p.out_nch == n.in_nch) {
      *out_sig = p.sig;
      return 0;
    }
  }
  SignalRef in_sig;
  int e = 0;
  if (target)
    in_sig = *target;  // mix into a signal somebody already reads from
  else
    e = temp_signal(b, n.in_nch, &in_sig);
  if (e) return e;
  std::vector<InputRef> ins;
  if (n.in_edges.empty()) {
    InputRef in{};
    in.kind = IN_SILENT;
    in.nch = 1;
    ins.push_back(in);
  } else {
    for (int ie : n.in_edges) {
      InputRef in{};
      if ((e = build_edge_input(b, id, ie, &in))) return e;
      ins.push_back(in);
    }
    if ((e = reduce_fan_in(b, ins, n.in_nch, n.interp))) return e;
  }
  if ((e = push_chain_step(b, ins, n.in_nch, n.interp, {}, in_sig))) return e;
  *out_sig = in_sig;
  return 0;
}

// OscillatorNode (oscillator.rs:323-660): one kernel, one lane per instance (the phase accumulator is serial)
// Is every frame of the quantum [block_time, next_block_time) inside [start_time, stop_time) — also for the reference's
// clock, which reaches frame k by k additions of dt (rounding: far below the one-frame margin asked of stop_time)?
// Then OscillatorRenderer::process renders frames 0 .. 127 and the replay below need not walk them.
static inline bool osc_quantum_fully_active(double block_time, double next_block_time, double start_time, double stop_time, double dt) {
  return start_time <= block_time && stop_time >= next_block_time + dt;
}

int plan_oscillator(waa_batch* b, uint32_t id) {
  // WAA_OSC_PLAN_CHECK=1 (tests): every quantum is walked frame by frame as before and the shortcut's answer is checked
  const bool check_replay = getenv("WAA_OSC_PLAN_CHECK") != nullptr;
  Node& n = b->nodes[id];
  Step st;
  st.kind = 9;
  OscDesc& d = st.osc;
  std::memset(&d, 0, sizeof d);
  int e;
  if ((e = node_param(b, id, WAA_PARAM_OSCILLATOR_FREQUENCY, &d.frequency)) ||
      (e = node_param(b, id, WAA_PARAM_OSCILLATOR_DETUNE, &d.detune)))
    return e;
  std::vector<double> start(b->n_inst), stop(b->n_inst);
  for (uint32_t i = 0; i < b->n_inst; i++) {
    start[i] = n.sched[i].start;
    stop[i] = n.sched[i].stop;
  }
  double *d_start = nullptr, *d_stop = nullptr;
  if ((e = dev_upload(b, &d_start, start)) || (e = dev_upload(b, &d_stop, stop))) return e;
  d.start = d_start;
  d.stop = d_stop;
  d.type = n.osc_wave.empty() ? n.desc.i[0] : WAA_OSC_CUSTOM;
  if (d.type == WAA_OSC_CUSTOM && n.osc_wave.empty())
    return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - custom oscillator %u has no PeriodicWave", id);
  std::vector<float> table;
  if (d.type == WAA_OSC_CUSTOM) {
    table = n.osc_wave;
  } else {  // oscillator.rs:16-28 (same libm sinf as the reference's f32::sin)
    table.resize(2048);
    const float pi = 3.14159265358979323846f;
    for (int x = 0; x < 2048; x++) table[x] = std::sin(((float)x) * 2.0f * pi * (1.f / 2048.f));
  }
  float* d_table = nullptr;
  if ((e = dev_upload(b, &d_table, table))) return e;
  d.table = d_table;
  d.table_len = (int32_t)table.size();
  d.out = n.sig;
  d.frames = b->lp;
  d.n_inst = b->n_inst;
  d.n_quanta = b->n_quanta;
  d.sample_rate = (double)b->sr;
  const bool parallel = d.frequency.mode != 2 && d.detune.mode != 2 && !getenv("WAA_OSC_EXACT");
  if (parallel) {
    // host-known frequency: replay the per-quantum decisions of OscillatorRenderer::process (oscillator.rs:336-452)
    // and record the phase at the first active frame of every quantum
    std::vector<OscQuantum> tq((size_t)b->n_inst * b->n_quanta);
    const double sample_rate = (double)b->sr, dt = 1. / sample_rate, nyquist = sample_rate / 2.;
    auto frac = [](long double x) {
      long double r = x - floorl(x);
      return (double)(r >= 1.L ? r - 1.L : r);
    };
    // instances with the same start / stop times and one frequency / detune value for the whole render replay alike: the
    // row of the first such instance is copied (1024 contexts of one patch: one replay instead of 1024)
    std::map<std::array<double, 4>, uint32_t> replayed;
    for (uint32_t i = 0; i < b->n_inst; i++) {
      const auto fq = param_per_quantum(b, n.params[WAA_PARAM_OSCILLATOR_FREQUENCY], i, nullptr);
      const auto dq = param_per_quantum(b, n.params[WAA_PARAM_OSCILLATOR_DETUNE], i, nullptr);
      double start_time = start[i];
      const double stop_time = stop[i];
      if (fq.size() == 1 && dq.size() == 1) {
        const std::array<double, 4> key = {start_time, stop_time, (double)fq[0], (double)dq[0]};
        auto it = replayed.find(key);
        if (it != replayed.end()) {
          std::copy(tq.begin() + (size_t)it->second * b->n_quanta, tq.begin() + (size_t)(it->second + 1) * b->n_quanta,
                    tq.begin() + (size_t)i * b->n_quanta);
          continue;
        }
        replayed.emplace(key, i);
      }
      long double phase = 0.L;
      bool started = false;
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        OscQuantum& oq = tq[(size_t)i * b->n_quanta + q];
        oq = OscQuantum{0., 0., 0, 0, 0};
        const double block_time = (double)((uint64_t)q * RQ) / sample_rate;
        const double next_block_time = block_time + dt * (double)RQ;
        if (stop_time <= block_time || start_time >= next_block_time) continue;
        if (!started && start_time < block_time) start_time = block_time;
        const float f = fq[fq.size() == 1 ? 0 : q], det = dq[dq.size() == 1 ? 0 : q];
        const double computed_freq = (double)f * std::exp2((double)det / 1200.);
        const double incr = computed_freq / sample_rate;
        oq.incr = incr;
        oq.outside_nyquist = std::fabs(computed_freq) >= nyquist ? 1 : 0;
        // the reference advances current_time by repeated addition: replay it to find the active frame range
        int first = -1, end = RQ;
        if (osc_quantum_fully_active(block_time, next_block_time, start_time, stop_time, dt) && !check_replay) {
          // (the usual quantum: 128 additions and comparisons per instance and quantum were 2.4 s of a 1024-context plan)
          first = 0;
          started = true;  // (start_time == block_time here when the node starts in this quantum: no sub-sample offset)
        } else {
          const bool expect_full = osc_quantum_fully_active(block_time, next_block_time, start_time, stop_time, dt);
          const bool was_started = started;
          double current_time = block_time;
          for (int k = 0; k < RQ; k++) {
            const bool active = !(current_time < start_time || current_time >= stop_time);
            if (active && first < 0) {
              first = k;
              if (!started) {
                if (current_time > start_time) {
                  if (expect_full) return fail(WAA_ERR_INVALID_STATE, "internal: oscillator replay shortcut (sub-sample start)");
                  phase = frac((long double)incr * (long double)((current_time - start_time) / dt));
                }
                started = true;
              }
            }
            if (!active && first >= 0) {
              end = k;
              break;
            }
            current_time += dt;
          }
          (void)was_started;
          if (expect_full && !(first == 0 && end == RQ)) return fail(WAA_ERR_INVALID_STATE, "internal: oscillator replay shortcut (range)");
        }
        if (first < 0) continue;
        oq.first = (int16_t)first;
        oq.end = (int16_t)end;
        oq.phase = (double)phase;
        phase = frac(phase + (long double)(end - first) * (long double)incr);
      }
    }
    OscQuantum* d_tq = nullptr;
    if ((e = dev_upload(b, &d_tq, tq))) return e;
    d.table_q = d_tq;
  }
  const bool scan = !parallel && !getenv("WAA_OSC_EXACT");
  if (scan) {
    // a-rate / graph-modulated frequency: the device forms the phase as a prefix sum of per-frame increments; the
    // host replays only the reference's clock (current_time += dt per frame, oscillator.rs:505-552) to find the
    // active frame range and the sub-sample start offset of every instance
    std::vector<int64_t> act((size_t)b->n_inst * 2, 0);
    std::vector<double> ratio(b->n_inst, 0.);
    const double sample_rate = (double)b->sr, dt = 1. / sample_rate;
    for (uint32_t i = 0; i < b->n_inst; i++) {
      double start_time = start[i];
      const double stop_time = stop[i];
      int64_t first = -1, end = -1;
      bool started = false;
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        const double block_time = (double)((uint64_t)q * RQ) / sample_rate;
        const double next_block_time = block_time + dt * (double)RQ;
        if (stop_time <= block_time || start_time >= next_block_time) continue;
        if (!started && start_time < block_time) start_time = block_time;
        if (osc_quantum_fully_active(block_time, next_block_time, start_time, stop_time, dt) && !check_replay) {
          if (first < 0) {
            first = (int64_t)q * RQ;
            started = true;
          }
          end = (int64_t)(q + 1) * RQ;
          continue;
        }
        const bool expect_full = osc_quantum_fully_active(block_time, next_block_time, start_time, stop_time, dt);
        const int64_t end_before = end;
        double current_time = block_time;
        for (int k = 0; k < RQ; k++) {
          const bool active = !(current_time < start_time || current_time >= stop_time);
          if (active) {
            if (first < 0) {
              first = (int64_t)q * RQ + k;
              if (current_time > start_time) {
                if (expect_full) return fail(WAA_ERR_INVALID_STATE, "internal: oscillator replay shortcut (sub-sample start)");
                ratio[i] = (current_time - start_time) / dt;
              }
              started = true;
            }
            end = (int64_t)q * RQ + k + 1;
          }
          current_time += dt;
        }
        (void)end_before;
        if (expect_full && !(end == (int64_t)(q + 1) * RQ && first <= (int64_t)q * RQ))
          return fail(WAA_ERR_INVALID_STATE, "internal: oscillator replay shortcut (range)");
      }
      act[(size_t)i * 2] = first < 0 ? 0 : first;
      act[(size_t)i * 2 + 1] = first < 0 ? 0 : end;
    }
    int64_t* d_act = nullptr;
    double* d_ratio = nullptr;
    if ((e = dev_upload(b, &d_act, act)) || (e = dev_upload(b, &d_ratio, ratio))) return e;
    d.active = d_act;
    d.start_ratio = d_ratio;
    double* d_seg = nullptr;
    if ((e = dev_alloc(b, &d_seg, (size_t)b->n_inst * OSC_SEGMENTS))) return e;
    d.seg_phase = d_seg;
  }
  st.profile_slot = slot_for(b, parallel ? "osc_par_kernel" : scan ? "osc_scan_kernel" : "osc_kernel");
  n.osc_step = (parallel || scan) ? (int)b->steps.size() : -1;  // (the serial cross-check kernel takes no post ops)
  b->steps.push_back(st);
  static const char* names[] = {"sine", "square", "sawtooth", "triangle", "custom"};
  plan_note(b, "oscillator node %u: %s (%s) frequency=%s detune=%s", id, names[d.type],
            parallel ? "time-parallel, closed-form phase" : scan ? "prefix-sum phase" : "lane per instance, serial phase",
            d.frequency.mode == 0 ? "const" : d.frequency.mode == 1 ? "k-rate" : "a-rate",
            d.detune.mode == 0 ? "const" : d.detune.mode == 1 ? "k-rate" : "a-rate");
  return 0;
}

// DelayNode (delay.rs:428-745).  Writer half: the node's mixed input becomes the delay line `hist` (an alias of the
// producer's signal when nothing has to be mixed).  Reader half: one gather kernel from the delay line.  Outside a
// loop the two are planned back to back; inside a block-scheduled loop each at its own place in the order.
int plan_delay_writer(waa_batch* b, uint32_t id) {
  Node& n = b->nodes[id];
  if (n.hist.base) {  // the reader half was planned first (inside a loop) and chose the delay line
    if (!n.hist_is_temp) return 0;
    SignalRef same;
    return node_input_signal(b, id, &same, &n.hist);
  }
  return node_input_signal(b, id, &n.hist);
}
// A folded DelayNode inside a block-scheduled loop (reader half): no launch, only the choice of the delay line — the
// producer's signal when there is exactly one materialised producer of the right layout, else a temporary the writer
// half fills (as plan_delay_reader does for the node-major form).
int plan_folded_delay_line(waa_batch* b, uint32_t id) {
  Node& n = b->nodes[id];
  if (!n.hist.base) {
    bool direct = false;
    if (n.in_edges.size() == 1) {
      Node& p = b->nodes[b->edges[n.in_edges[0]].from];
      direct = p.materialized && p.out_nch == n.in_nch && p.sig.base;
      if (direct) n.hist = p.sig;
    }
    if (!direct) {
      int e = temp_signal(b, n.in_nch, &n.hist);
      if (e) return e;
      n.hist_is_temp = true;
    }
  }
  n.hist_valid = b->lp;
  plan_note(b, "delay node %u: %dch, read by its consumers from the delay line (no pass of its own, inside a block-scheduled loop)", id,
            n.in_nch);
  return 0;
}
int plan_delay_reader(waa_batch* b, uint32_t id) {
  Node& n = b->nodes[id];
  Step st;
  st.kind = 7;
  DelayDesc& d = st.delay;
  std::memset(&d, 0, sizeof d);
  const bool in_cycle = id < b->cut.size() && b->cut[id];
  if (in_cycle && !n.hist.base) {
    // the reader renders before its writer: the delay line is not planned yet.  It is the producer's signal when
    // there is exactly one materialised producer of the right layout, else a temporary the writer half fills.
    bool direct = false;
    if (n.in_edges.size() == 1) {
      Node& p = b->nodes[b->edges[n.in_edges[0]].from];
      direct = p.materialized && p.out_nch == n.in_nch && p.sig.base;
      if (direct) n.hist = p.sig;
    }
    if (!direct) {
      int e = temp_signal(b, n.in_nch, &n.hist);
      if (e) return e;
      n.hist_is_temp = true;
    }
  }
  if (!n.hist.base) return fail(WAA_ERR_INVALID_STATE, "internal: delay line of node %u not planned", id);
  d.in = n.hist;
  d.out = n.sig;
  int e = node_param(b, id, WAA_PARAM_DELAY_DELAY_TIME, &d.delay);
  if (e) return e;
  d.sample_rate = (double)b->sr;
  d.frames = b->lp;
  d.num_quanta = (int32_t)std::ceil(n.desc.d[0] * (double)b->sr / (double)RQ);
  d.nch = n.in_nch;
  d.n_inst = b->n_inst;
  d.n_quanta = b->n_quanta;
  d.tile0 = 0;
  d.tile1 = b->n_tiles;
  d.in_cycle = in_cycle ? 1 : 0;
  const double dt = 1. / (double)b->sr;
  d.quantum_duration = (double)RQ * dt;  // delay.rs:546-548
  st.profile_slot = slot_for(b, "delay_kernel");
  b->steps.push_back(st);
  plan_note(b, "delay node %u: %dch delayTime=%s ring=%d quanta%s", id, d.nch,
            d.delay.mode == 0 ? "const" : d.delay.mode == 1 ? "k-rate" : "a-rate", d.num_quanta + 1,
            in_cycle ? " (in a loop: clamped to one quantum)" : "");
  return 0;
}

// Block size (in 2048-frame tiles) for a block-scheduled feedback loop, 0 if the loop needs the quantum-serial
// kernel.  Every DelayNode whose writer->reader edge the cycle breaker removed must have a host-known delay
// (constant or k-rate blocks, not modulated from the graph) strictly longer than the block: then no frame of a
// block depends on loop history of the same block.
uint32_t loop_block_tiles(waa_batch* b, const std::vector<uint32_t>& loop_items) {
  if (getenv("WAA_LOOP_KERNEL")) return 0;  // debugging aid: force the quantum-serial kernel
  const double dt = 1. / (double)b->sr;
  const double quantum_duration = (double)RQ * dt;
  double dmin = 1e300;
  uint32_t conv_tiles = 1;  // partition size of the largest convolver in the loop, in tiles (partitions <= 2048 frames divide a tile)
  for (uint32_t v : loop_items) {
    const uint32_t id = v & ~VTX_READER;
    Node& n = b->nodes[id];
    if (!(v & VTX_READER)) {
      // a ConvolverNode renders whole partitions: the block must hold a whole number of them (below)
      if (n.desc.kind == WAA_NODE_CONVOLVER && n.has_ir) conv_tiles = std::max(conv_tiles, (uint32_t)conv_block_size(b, n) / (uint32_t)TILE);
      continue;
    }
    if (!b->cut[id]) continue;  // keeps its writer->reader edge: reads the current block like any other node
    if (param_mode(n, WAA_PARAM_DELAY_DELAY_TIME) == 2) return 0;
    for (uint32_t i = 0; i < b->n_inst; i++)
      for (float dv : param_per_quantum(b, n.params[WAA_PARAM_DELAY_DELAY_TIME], i, nullptr))
        dmin = std::min(dmin, std::max((double)dv, quantum_duration) * (double)b->sr);
  }
  if (!(dmin < 1e300)) return 0;
  const double tiles = std::ceil(dmin / (double)TILE) - 1.;  // block < delay, strictly
  if (tiles < 1.) return 0;
  const uint32_t bt = (uint32_t)std::min(tiles, 64.);
  return bt / conv_tiles * conv_tiles;  // (0: the delay is shorter than the convolver's partition -> quantum-serial -> refused there)
}

// A feedback loop (strongly connected group around at least one DelayNode): one loop_kernel launch renders all
// members quantum by quantum in the reference's processing order.  `loop_items` = that order, two entries per
// DelayNode (writer / reader halves).
int plan_loop(waa_batch* b, const std::vector<uint32_t>& loop_items) {
  if (loop_items.size() > (size_t)LOOP_MAX_ITEMS)
    return fail(WAA_ERR_OUT_OF_SCOPE, "feedback loop with more than %d members", LOOP_MAX_ITEMS);
  std::map<uint32_t, int> out_item;  // node id -> item that produces its output
  std::map<uint32_t, int> writer_item;
  for (size_t k = 0; k < loop_items.size(); k++) {
    const uint32_t v = loop_items[k], id = v & ~VTX_READER;
    if (is_delay(b, id)) {
      if (v & VTX_READER)
        out_item[id] = (int)k;
      else
        writer_item[id] = (int)k;
    } else {
      out_item[id] = (int)k;
    }
  }
  std::vector<LoopItem> host(loop_items.size());
  std::string desc;
  for (size_t k = 0; k < loop_items.size(); k++) {
    const uint32_t v = loop_items[k], id = v & ~VTX_READER;
    Node& n = b->nodes[id];
    LoopItem& li = host[k];
    std::memset(&li, 0, sizeof li);
    if (n.in_nch > 2 || n.out_nch > 2)
      return fail(WAA_ERR_OUT_OF_SCOPE, "feedback loops render at most 2 channels per signal (node %u)", id);
    if (is_frozen_node(n))
      return fail(WAA_ERR_OUT_OF_SCOPE, "an oversampled WaveShaperNode / HRTF PannerNode inside a feedback loop is out of scope (node %u)", id);
    for (auto& pe : n.pin_edges)
      for (int e : pe)
        if (out_item.count(b->edges[e].from))
          return fail(WAA_ERR_OUT_OF_SCOPE, "an AudioParam of node %u is modulated from inside its own feedback loop", id);
    const bool reader = is_delay(b, id) && (v & VTX_READER);
    li.nch_in = n.in_nch;
    li.nch_out = n.out_nch;
    li.interp = n.interp;
    if (!reader) {
      // inputs of the node (of the writer half for a DelayNode), in summing order
      if (n.in_edges.size() > (size_t)MAX_INPUTS)
        return fail(WAA_ERR_OUT_OF_SCOPE, "more than %d inputs on node %u inside a feedback loop", MAX_INPUTS, id);
      li.n_in = (int)n.in_edges.size();
      for (int j = 0; j < li.n_in; j++) {
        const uint32_t pid = b->edges[n.in_edges[j]].from;
        Node& pn = b->nodes[pid];
        if (pn.out_nch > 2)  // (the loop kernel loads and mixes mono / stereo inputs only)
          return fail(WAA_ERR_OUT_OF_SCOPE, "feedback loops render at most 2 channels per signal (input %u of node %u)", pid, id);
        li.in_nch[j] = pn.out_nch;
        auto it = out_item.find(pid);
        if (it != out_item.end()) {
          if (it->second >= (int)k) return fail(WAA_ERR_INVALID_STATE, "internal: loop member order");
          li.in_item[j] = it->second;
        } else {
          if (!pn.materialized || !pn.sig.base) return fail(WAA_ERR_INVALID_STATE, "internal: loop input not planned");
          li.in_item[j] = -1;
          li.in_sig[j] = pn.sig;
        }
      }
    }
    char t[96];
    if (is_delay(b, id)) {
      if (!reader) {
        li.kind = LI_DELAY_W;
        int e = temp_signal(b, n.in_nch, &li.out);  // the delay line, in absolute time
        if (e) return e;
        snprintf(t, sizeof t, "delayW%u", id);
      } else {
        li.kind = LI_DELAY_R;
        li.out = n.sig;
        li.writer_item = writer_item.at(id);
        li.in_cycle = li.writer_item > (int)k ? 1 : 0;  // delay.rs:535-541: the writer has not rendered yet
        li.num_quanta = (int32_t)std::ceil(n.desc.d[0] * (double)b->sr / (double)RQ);
        int e = node_param(b, id, WAA_PARAM_DELAY_DELAY_TIME, &li.op.p0);
        if (e) return e;
        snprintf(t, sizeof t, "delayR%u%s", id, li.in_cycle ? "(clamped)" : "");
      }
    } else {
      li.kind = LI_NODE;
      li.out = n.sig;
      std::vector<OpDesc> ops;
      int out_nch = 0;
      int e = emit_node_ops(b, id, n.in_nch, true, ops, &out_nch);
      if (e) return e;
      if (ops.size() > 1) return fail(WAA_ERR_OUT_OF_SCOPE, "node %u cannot be rendered inside a feedback loop", id);
      if (ops.empty()) {  // only true pass-through nodes may render nothing
        const uint32_t k = n.desc.kind;
        const bool pass = k == WAA_NODE_ANALYSER || (k == WAA_NODE_WAVESHAPER && !n.has_curve) ||
                          (k == WAA_NODE_CONVOLVER && !n.has_ir);
        if (!pass)
          return fail(WAA_ERR_OUT_OF_SCOPE, "node %u (kind %u) cannot be rendered inside a feedback loop on the device path", id, k);
      }
      if (!ops.empty()) {
        const OpDesc& o = ops[0];
        const bool ok = o.kind == OP_GAIN || o.kind == OP_BIQUAD || o.kind == OP_WAVESHAPER ||
                        (o.kind == OP_STEREO_PAN && o.p0.mode != 2);
        if (!ok)
          return fail(WAA_ERR_OUT_OF_SCOPE, "node %u (%s) cannot be rendered inside a feedback loop on the device path", id,
                      op_name(o.kind));
        li.op = o;
      }
      snprintf(t, sizeof t, "%s%u", ops.empty() ? "pass" : op_name(ops[0].kind), id);
    }
    desc += desc.empty() ? t : std::string(",") + t;
  }
  // fix up the reader items' writer outputs are read through host[writer_item].out on the device: same array
  LoopItem* dev = nullptr;
  int e = dev_upload(b, &dev, host);
  if (e) return e;
  Step st;
  st.kind = 8;
  LoopDesc& d = st.loop;
  std::memset(&d, 0, sizeof d);
  d.items = dev;
  d.n_items = (int32_t)host.size();
  d.n_inst = b->n_inst;
  d.n_quanta = b->n_quanta;
  d.sample_rate = (double)b->sr;
  const double dt = 1. / (double)b->sr;
  d.quantum_duration = (double)RQ * dt;  // delay.rs:546-548
  st.profile_slot = slot_for(b, "loop_kernel");
  for (const LoopItem& li : host) {
    for (int j = 0; j < li.n_in; j++)
      if (li.in_item[j] < 0) st.loop_reads.push_back(li.in_sig[j].base);
    st.loop_writes.push_back(li.out.base);
  }
  b->steps.push_back(st);
  plan_note(b, "feedback loop: %d item(s) per quantum [%s]", d.n_items, desc.c_str());
  return 0;
}

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
  if (len <= (uint64_t)DIRECT_MAX_TAPS && !b->dynamic && !getenv("WAA_NO_DIRECT_FIR")) return 0;
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
  const bool direct_fir = len <= (uint64_t)DIRECT_MAX_TAPS && !b->dynamic && !getenv("WAA_NO_DIRECT_FIR");
  int B = 8192;
  for (int cand : {128, 512, 2048, 8192})
    if ((len + cand - 1) / cand <= 24) {
      B = cand;
      break;
    }
  cv.block = B;
  cv.n = 2 * B;
  cv.fft3 = cv.n == 16384 && !getenv("WAA_CONV_FFT_R4");  // (the round-2 radix-4-in-LDS kernels: same-box A/B only)
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
      const char* genv = getenv("WAA_IIR_GROWTH");  // experiments only
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
