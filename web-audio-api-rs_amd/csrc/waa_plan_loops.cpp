// waa_plan_loops.cpp — DelayNodes and feedback loops: writer / reader halves, delay lines read in place, block size of a
// block-scheduled loop, the quantum-serial loop kernel's plan, the LDS-ring forms of echo loops and feed-forward echoes
// (split out of waa_plan.cpp in round 4).
#include <array>
#include <set>

#include "waa_host.hpp"
#include "waa_plan_parts.hpp"

namespace waa {
namespace host {

// An echo loop rendered by the LDS-ring kernel (Step::echo_fb): when the line it writes has exactly ONE reader in the whole plan
// and that reader is a plain sum of the delayed line and of signals the loop step reads too (the destination's  dry + wet),
// the ring kernel renders that sum as well and the line is never stored (waa_echo.hip, "the tail").  Decided on the finished
// launch list, buffer by buffer, with the same read / write sets the validation below uses; any launch kind those sets do
// not describe keeps the plan as it is.
// The filtered echo (Step::echo_bq): two signals could leave the launch — the delay line (read inside the launch by the fused
// delayed read) and the filter's output y (read by the loop's sum).  Each is stored only for readers outside the launch; when y
// has exactly one, a later plain sum of y and of signals the loop reads anyway, that sum is rendered by the launch as well.
static void fuse_filtered_echo_tail(waa_batch* b, size_t l) {
  Step& ls = b->steps[l];
  const void* line = ls.chain.out.base;
  const void* y = ls.echo_bq.y.base;
  int line_readers = 0, y_readers = 0;
  size_t reader = 0;
  bool other_writer = false;
  for (size_t k = 0; k < b->steps.size(); k++) {
    if (k == l) continue;
    const Step& sk = b->steps[k];
    if (sk.echo_fused && sk.group == ls.group) continue;  // (the delayed read and the filter: inside the launch)
    const StepIo io = step_io(sk);
    auto delayed_from = [&](const InputRef& in) { return in.kind == IN_DELAYED && in.sig.base == line; };
    bool rl = std::find(io.reads.begin(), io.reads.end(), line) != io.reads.end();
    if (sk.kind == 0)
      for (int q = 0; q < sk.chain.n_inputs; q++) rl |= delayed_from(sk.chain.in[q]);
    rl |= (sk.kind == 1 && delayed_from(sk.bq.in)) || (sk.kind == 6 && delayed_from(sk.iir.in)) || (sk.kind == 19 && delayed_from(sk.lanes.in));
    line_readers += rl ? 1 : 0;
    if (std::find(io.reads.begin(), io.reads.end(), y) != io.reads.end()) {
      y_readers++;
      reader = k;
    }
    other_writer |= std::find(io.writes.begin(), io.writes.end(), y) != io.writes.end() ||
                    std::find(io.writes.begin(), io.writes.end(), line) != io.writes.end();
  }
  bool line_alias = false, y_alias = false;  // (analyser / destination nodes that alias a signal are read outside the launch list)
  for (const Node& an : b->nodes)
    if (an.live && (an.desc.kind == WAA_NODE_ANALYSER || an.desc.kind == WAA_NODE_DESTINATION)) {
      line_alias |= an.sig.base == line;
      y_alias |= an.sig.base == y;
    }
  if (other_writer) return;  // (everything stays stored)
  ls.echo_bq.store_line = line_readers > 0 || line_alias ? 1 : 0;
  ls.echo_bq.store_y = y_readers > 0 || y_alias ? 1 : 0;
  if (y_readers == 1 && !y_alias && reader > l) {
    Step& ts = b->steps[reader];
    EchoTail t{};
    const char* why = "it is not an element-wise launch";
    if (ts.kind == 0 && ts.group < 0 && echo_tail_applicable(ls.chain, ls.echo_fb, ts.chain, &t, &why, &ls.echo_bq)) {
      t.store_line = ls.echo_bq.store_line;
      ls.echo_tail = t;
      ls.echo_tail_step = (int)reader;
      ts.echo_fused = true;
      ls.echo_bq.store_y = 0;
      plan_note(b, "filtered echo loop: launch %zu (the only reader of the filter's output: %d input(s) -> %d channel(s)) is rendered by the LDS-ring kernel too; the filter's output is not stored",
                reader, t.n_inputs, t.in_nch);
    } else {
      plan_note(b, "filtered echo loop: launch %zu, the only reader of the filter's output, is not a plain sum of it and of the loop's inputs (%s): the output is stored", reader, why);
    }
  }
  plan_note(b, "filtered echo loop: the delay line is %s, the filter's output is %s", ls.echo_bq.store_line ? "stored (read outside the launch)" : "not stored",
            ls.echo_bq.store_y ? "stored" : "not stored");
}

void fuse_echo_tails(waa_batch* b) {
  if (measure_switch("WAA_NO_ECHO_TAIL")) return;
  for (const Step& st : b->steps)
    if (st.kind == 11 || st.kind == 15 || st.kind > 20) return;
  for (size_t l = 0; l < b->steps.size(); l++) {
    Step& ls = b->steps[l];
    if (ls.kind != 0 || ls.echo_fb < 0) continue;
    if (ls.echo_bq.coefs) {
      fuse_filtered_echo_tail(b, l);
      continue;
    }
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
  if (measure_switch("WAA_NO_ECHO_RING") || measure_switch("WAA_NO_ECHO_FF")) return;
  const char* mi = measure_switch("WAA_ECHO_FF_MIN_INST");
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
    st.echo_ring = echo_ring_frames(delayed_hi, chunk);
    st.profile_slot = slot_for(b, "echo_ring_kernel");
    plan_note(b, "launch %zu (delayed signal + %d more input(s), no ops) is rendered by the LDS-ring kernel with nothing fed back: delay %.0f .. %.0f frames, chunks of %d frames",
              k, t.n_inputs - 1, (double)delayed_lo, (double)delayed_hi, chunk);
  }
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
  if (measure_switch("WAA_LOOP_KERNEL")) return 0;  // debugging aid: force the quantum-serial kernel
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
  if (tiles < 1.) {
    // Shorter than a tile: launches per block cannot render it — but the LDS-ring kernel (waa_echo.hip) walks in chunks down to
    // 256 frames.  A loop of the shapes it renders (ONE DelayNode with one delayTime per instance, GainNodes, at most one
    // constant-coefficient Biquad, nothing modulated) is planned like a block-scheduled loop; build_plan checks that the
    // ring kernel took it and plans again without this branch otherwise (no_short_ring).
    if (b->no_short_ring || measure_switch("WAA_NO_ECHO_RING") || measure_switch("WAA_NO_SHORT_RING") || dmin < 136.) return 0;
    int n_delay = 0, n_biquad = 0;
    uint32_t delay_id = 0;
    for (uint32_t v : loop_items) {
      if (v & VTX_READER) continue;
      const uint32_t id = v & ~VTX_READER;
      const Node& n = b->nodes[id];
      for (auto& pe : n.pin_edges)
        if (!pe.empty()) return 0;
      switch (n.desc.kind) {
        case WAA_NODE_DELAY:
          if (!b->cut[id] || param_mode(n, WAA_PARAM_DELAY_DELAY_TIME) == 1) return 0;  // (one value per quantum: not one per instance)
          n_delay++;
          delay_id = id;
          break;
        case WAA_NODE_GAIN:
          if (param_mode(n, WAA_PARAM_GAIN_GAIN) == 2) return 0;
          break;
        case WAA_NODE_BIQUAD:
          for (size_t k = 0; k < n.params.size(); k++)
            if (param_mode(n, k) != 0) return 0;
          n_biquad++;
          break;
        default: return 0;
      }
    }
    if (n_delay != 1 || n_biquad > 1) return 0;
    b->short_ring_loops.insert(delay_id);
    return 1;
  }
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

}  // namespace host
}  // namespace waa
