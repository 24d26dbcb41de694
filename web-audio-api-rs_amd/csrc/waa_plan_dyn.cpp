// waa_plan_dyn.cpp — the dynamic-count plan (waa_dyn.hip): the dynamic-group builder, split out of build_plan in round 4.
//
// Sources, oscillators, FFT convolvers and the frozen-state nodes stay node-major launches; every other live node becomes an item of
// a quantum-serial dyn_kernel launch that carries per-quantum codes (count | silent) with every signal.  A convolver / frozen-state
// node splits the items into groups (its input is produced by the group in front of it, its output consumed by the group behind it).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <string>

#include "waa_host.hpp"
#include "waa_plan_parts.hpp"

namespace waa {
namespace host {

int plan_dynamic_groups(waa_batch* b, const DynPlanCtx& c) {
  const uint32_t N = (uint32_t)b->nodes.size();
  const std::vector<uint32_t>& items = c.items;
  const std::vector<Unit>& units = c.units;
  const std::vector<int>& scc_of = c.scc_of;
  const auto& alloc_signal = c.alloc_signal;
  const auto& plan_single = c.plan_single;
  const bool count_change_found = c.count_change_found, mixed_buffer_counts = c.mixed_buffer_counts;
  if (!count_change_found && mixed_buffer_counts)
    plan_note(b, "instances play AudioBuffers of different channel counts -> per-instance counts through dyn_kernel");
  else if (!count_change_found)
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
    // (round 6) signals of 7 ... 32 channels in a dynamic plan: dyn_kernel<8 / 16 / 32> — the same count rules, every mix above six
    // channels discrete; an oversampled WaveShaper renders its channel pairs whatever their number (one launch per pair, one link table)
    const bool narrow_only = (k == WAA_NODE_CONVOLVER && n.has_ir) || (is_frozen_node(n) && k == WAA_NODE_PANNER);  // (an oversampled WaveShaper renders channel pairs, round 4)
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
  // ---- a feedback loop with frozen-state nodes inside (round 5): cut into segments at those nodes, every segment one RANGED
  // dyn_kernel launch, the node's link / transform / FIR launches ranged too, all of them launched quantum block by quantum block
  // (Step::qgroup).  A DelayNode whose writer and reader fall into different segments talks through memory (DynItem::xline ...).
  auto is_cut_node = [&](const Node& m) { return is_frozen_node(m) || (m.desc.kind == WAA_NODE_CONVOLVER && m.has_ir); };
  struct XDelay {
    SignalRef line{};
    uint32_t* aux32 = nullptr;
    int32_t* state = nullptr;
    int writer_seg = -1, reader_seg = -1;
  };
  int cur_qgroup = -1;
  std::map<uint32_t, XDelay> xdelay;  // delay node id -> cross-segment resources (only pairs that ARE split)
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
        const auto xd = cur_qgroup >= 0 ? xdelay.find(id) : xdelay.end();
        if (!reader) {
          li.kind = DI_DELAY_W;
          li.num_quanta = (int32_t)std::ceil(n.desc.d[0] * (double)b->sr / (double)RQ);  // (ring capacity - 1, as for the reader)
          int e = 0;
          if (xd != xdelay.end()) {  // (the reader sits in another launch: line, codes and ring state were allocated for both)
            li.out = xd->second.line;
            li.aux32 = xd->second.aux32;
            li.xstate = xd->second.state;
          } else {
            e = temp_signal(b, n.in_nch, &li.out);  // the delay line in absolute time, native layout
            if (e) return e;
            if ((e = dev_alloc(b, &li.aux32, (size_t)b->n_inst * cs))) return e;
          }
          li.nch_pub = n.in_nch;
          snprintf(t, sizeof t, "delayW%u", id);
        } else {
          li.kind = DI_DELAY_R;
          li.out = n.sig;
          li.nch_pub = n.out_nch;
          if (xd != xdelay.end()) {
            li.writer_item = -1;
            li.xline = xd->second.line;
            li.xaux32 = xd->second.aux32;
            li.xstate = xd->second.state;
            li.in_cycle = xd->second.writer_seg > xd->second.reader_seg ? 1 : 0;  // (the writer's launch of this quantum comes later)
          } else {
            li.writer_item = writer_item.at(id);
            li.in_cycle = li.writer_item > (int)k ? 1 : 0;
          }
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
        } else if (kind == WAA_NODE_GAIN && scc_of[id] >= 0 && !n.pin_edges.empty() && !n.pin_edges[0].empty() &&
                   scc_of[b->edges[n.pin_edges[0][0]].from] == scc_of[id]) {
          // (round 6) the gain param is modulated from inside the node's own loop: its inputs are items of this launch (same quantum,
          // out of the ring) instead of a node-major param chain in front of the group
          if (cur_qgroup >= 0) return fail(WAA_ERR_OUT_OF_SCOPE, "an AudioParam of node %u is modulated from inside its own feedback loop", id);
          std::vector<int> saved = n.pin_edges[0];
          if (saved.size() > 4) return fail(WAA_ERR_OUT_OF_SCOPE, "more than 4 inputs on the gain param of node %u inside a feedback loop", id);
          for (size_t j = 0; j < saved.size(); j++) {
            const uint32_t from = b->edges[saved[j]].from;
            const uint32_t want = is_delay(b, from) ? (from | VTX_READER) : from;
            int found = -1;
            for (size_t k2 = 0; k2 < k; k2++)
              if (pending[k2] == want) found = (int)k2;
            if (found < 0 || scc_of[from] != scc_of[id])
              return fail(WAA_ERR_OUT_OF_SCOPE, "an AudioParam of node %u is modulated from inside its own feedback loop (producer %u is not rendered in front of it)", id, from);
            li.pmod_item[j] = found;
          }
          li.pmod_n = (int32_t)saved.size();
          li.pmod_min = n.params[0].minv;
          li.pmod_max = n.params[0].maxv;
          li.pmod_def = n.params[0].defv;
          n.pin_edges[0].clear();  // (the intrinsic value alone: constants / value blocks / automation)
          e = emit_node_ops(b, id, n.in_nch, true, ops, &out_nch);
          n.pin_edges[0] = saved;
          if (e) return e;
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
    // ---- the quantum pipeline (waa_dyn.hip, W > 1): cut the items — in thirds: unit 3 i = item i's gather + mix of its inputs, unit
    // 3 i + 1 = its node, unit 3 i + 2 = the publication of its result — into up to DYN_MAX_STAGES contiguous stages of about equal cost.  Never between a DelayNode's
    // writer and reader (the reader looks at the writer's ring state of ITS quantum) nor inside a feedback loop (its members see
    // each other's output of the same quantum through the delay line).
    d.n_stages = 1;
    d.stage_begin[0] = 0;
    d.stage_begin[1] = 3 * d.n_items;
    {
      // (round 6) items without state from quantum to quantum — gains, mixes, curves, the destination's pass, DelayNode halves whose
      // partner sits in another launch: no quantum of the launch depends on an earlier one, the launcher spreads the quanta over
      // workgroups (a block of a quantum-blocked loop, or the whole render of such a group) instead of walking them one by one
      d.split_ok = d.cmax <= 2 ? 1 : 0;
      for (int k = 0; k < d.n_items; k++) {
        const DynItem& li = host[(size_t)k];
        bool ok = false;
        if (li.kind == DI_DELAY_R)
          ok = li.writer_item < 0 && li.xstate;
        else if (li.kind == DI_DELAY_W)
          ok = li.xstate != nullptr;
        else
          ok = li.dk == DK_PASS || li.dk == DK_GAIN || li.dk == DK_WAVESHAPER || (li.dk == DK_CONV_IN && !li.compact_ch1);
        if (!ok) d.split_ok = 0;
      }
    }
    if (cur_qgroup >= 0) {
      // a segment of a quantum-blocked loop: ranged launches that carry the items' state through memory; no quantum pipeline
      const int cm = dyn_planes(d.cmax);
      int e2 = dev_alloc(b, &d.save_f, (size_t)b->n_inst * (size_t)d.n_items * cm * DYN_STATE);
      if (!e2) e2 = dev_alloc(b, &d.save_i, (size_t)b->n_inst * (size_t)d.n_items * 4);
      if (e2) return e2;
      st.qgroup = cur_qgroup;

    } else if (d.cmax <= 2 && d.n_items >= 1 && !d.split_ok) {
      const int n = d.n_items, nu = 3 * n;
      std::vector<uint8_t> nocut((size_t)nu, 0);  // nocut[u]: units u and u + 1 stay together
      auto keep = [&](int lo_item, int hi_item) {
        for (int u = 3 * lo_item; u < 3 * hi_item + 2; u++) nocut[(size_t)u] = 1;
      };
      std::map<int, std::pair<int, int>> scc_span;
      for (int k = 0; k < n; k++) {
        const uint32_t id = pending[(size_t)k] & ~VTX_READER;
        if (host[(size_t)k].kind == DI_DELAY_R) {
          keep(std::min(k, host[(size_t)k].writer_item), std::max(k, host[(size_t)k].writer_item));
          nocut[(size_t)(3 * k)] = 1;  // (a reader has no inputs to gather: its first third is empty)
        }
        if (scc_of[id] >= 0) {
          auto it = scc_span.find(scc_of[id]);
          if (it == scc_span.end())
            scc_span[scc_of[id]] = {k, k};
          else
            it->second.second = k;
        }
      }
      for (auto& sp : scc_span) keep(sp.second.first, sp.second.second);
      // cost per unit in thousands of cycles per quantum (every phase costs 1.2-2 k cycles whatever it computes;
      // WAA_DYN_CYCLES on the probe graph: Biquad gather 4.0 (two external inputs) / node 4.1 / hand-over 2.0, StereoPanner 1.6 /
      // 3.1 / 1.9, pass 1.6 / 1.2 / 1.8)
      std::vector<double> pre((size_t)nu + 1, 0.);
      for (int k = 0; k < n; k++) {
        const DynItem& li = host[(size_t)k];
        double front = li.kind == DI_DELAY_R ? 0. : 1.1, node = 0.6, pub = li.out.base ? 1.4 : 0.2;
        for (int j = 0; j < li.n_in; j++) front += li.in[j].item < 0 ? 1.4 : 0.5;
        if (li.kind == DI_DELAY_R) node += 4.0;
        else if (li.kind == DI_DELAY_W) node += 1.0;
        else if (li.dk == DK_BIQUAD) node += 4.1;
        else if (li.dk == DK_IIR) node += 9.0;
        else if (li.dk == DK_STEREO_PAN || li.dk == DK_PANNER) node += 3.1;
        else if (li.dk == DK_WAVESHAPER) node += 2.5;
        else if (li.dk == DK_GAIN) node += 1.5;
        else node += 1.2;
        pre[(size_t)(3 * k) + 1] = pre[(size_t)(3 * k)] + front;
        pre[(size_t)(3 * k) + 2] = pre[(size_t)(3 * k) + 1] + node;
        pre[(size_t)(3 * k) + 3] = pre[(size_t)(3 * k) + 2] + pub;
      }
      // best[s][i]: smallest possible largest-stage cost of the first i units in s stages; cut[s][i]: where the last stage starts
      const double INF = 1e300;
      std::vector<std::vector<double>> best(DYN_MAX_STAGES + 1, std::vector<double>((size_t)nu + 1, INF));
      std::vector<std::vector<int>> cut(DYN_MAX_STAGES + 1, std::vector<int>((size_t)nu + 1, 0));
      for (int i = 1; i <= nu; i++) best[1][(size_t)i] = pre[(size_t)i];
      for (int sN = 2; sN <= DYN_MAX_STAGES; sN++)
        for (int i = sN; i <= nu; i++)
          for (int j = sN - 1; j < i; j++) {  // the last stage = units [j, i)
            if (nocut[(size_t)j - 1] || best[sN - 1][(size_t)j] >= INF) continue;
            const double v = std::max(best[sN - 1][(size_t)j], pre[(size_t)i] - pre[(size_t)j]);
            if (v < best[sN][(size_t)i]) {
              best[sN][(size_t)i] = v;
              cut[sN][(size_t)i] = j;
            }
          }
      // Every stage is a wavefront, and all of a launch's wavefronts must be resident at once or the launch takes a second round
      // (r05e: five stages x 1024 contexts = 5120 wavefronts on a device that holds 4096 of this kernel — 107 registers: four per
      // SIMD — ran SLOWER than one stage): at most (CUs x 4 SIMDs x 4) / contexts stages.
      // (A 96-register build — five wavefronts per SIMD, eight spilled registers — made a fifth stage resident at 1024 contexts and
      // ran it at 20.8 ms against 13.1 ms for four stages, r05k: not kept.)
      const int max_stages = std::max(1, std::min(DYN_MAX_STAGES, (int)((uint64_t)b->n_cu * 16 / std::max<uint32_t>(b->n_inst, 1))));
      int pick = 1;
      double pick_cost = best[1][(size_t)nu];
      for (int sN = 2; sN <= max_stages; sN++) {
        if (best[sN][(size_t)nu] >= INF) continue;
        // (... and so must their local memory: contexts / CUs workgroups share a CU's 160 KB)
        const uint64_t wg_per_cu = ((uint64_t)b->n_inst + (uint64_t)b->n_cu - 1) / (uint64_t)b->n_cu;
        if (dyn_lds_bytes(n, d.cmax, sN) * std::min<uint64_t>(wg_per_cu, 16) > 150 * 1024) continue;
        const double cst = best[sN][(size_t)nu] + 0.3 + 0.05 * sN;  // (+ the step's barrier, which waits for the slowest of sN waves)
        if (cst < 0.93 * pick_cost) {
          pick = sN;
          pick_cost = cst;
        }
      }
      if (const char* force = measure_switch("WAA_DYN_STAGES")) {  // (A/B aid: at most this many stages)
        const int cap = std::max(1, atoi(force));
        while (pick > cap || (pick > 1 && best[pick][(size_t)nu] >= INF)) pick--;
      }
      if (pick > 1) {
        d.n_stages = pick;
        int i = nu;
        for (int sN = pick; sN >= 1; sN--) {
          d.stage_begin[sN] = i;
          i = sN > 1 ? cut[sN][(size_t)i] : 0;
        }
        d.stage_begin[0] = 0;
      }
    }
    st.profile_slot = slot_for(b, "dyn_kernel");
    b->steps.push_back(st);
    {
      std::string cuts;
      for (int sN = 1; sN < d.n_stages; sN++)
        cuts += (cuts.empty() ? "" : ",") + std::to_string(d.stage_begin[sN] / 3) + (d.stage_begin[sN] % 3 == 1 ? "b" : d.stage_begin[sN] % 3 == 2 ? "c" : "");
      plan_note(b, "dynamic-count group: %d item(s) per quantum [%s]%s%s%s", d.n_items, desc.c_str(),
                d.n_stages > 1 ? (", pipelined over the quanta in " + std::to_string(d.n_stages) + " stages, cut in front of item(s) ").c_str() : "",
                cuts.c_str(), d.split_ok ? ", no item keeps state from quantum to quantum: the quanta are spread over workgroups" : "");
    }
    for (uint32_t v : pending) planned_node[v & ~VTX_READER] = 1;
    pending.clear();
    pending_nodes.clear();
    return 0;
  };
  // ---- a ConvolverNode of a dynamic plan behind the group that published its mixed input: FFT steps + the code kernel
  auto plan_dyn_convolver = [&](uint32_t id) -> int {
    Node& n = b->nodes[id];
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
    // (round 5) where the reference's FFT convolver leaves roundoff noise instead of exact zeros: one automaton per
    // FFTConvolver of the node (convolver.rs:291-306: one per channel of the response, at least two), fed by the non-zero flags of
    // the input quanta; waa_conv_noise.hpp
    cd.noise = n.hist.base && !measure_switch("WAA_NO_CONV_NOISE_FLOOR") ? 1 : 0;
    if (cd.noise) {
      cd.in = n.hist;
      cd.in_test_nch = (n.ir_nch == 1 && n.in_nch == 2) ? 1 : std::min(n.in_nch, 2);
      const int ncv = std::max(n.ir_nch, 2);
      for (int k = 0; k < 4; k++) {
        ConvNoiseIr& ni = cd.nir[k];
        ni = ConvNoiseIr{};
        if (k >= ncv) continue;
        const std::vector<float>& h = n.ir[std::min(k, n.ir_nch - 1)];
        uint64_t l = n.ir_len;
        while (l > 0 && std::fabs(h[l - 1]) < 0.000001f) l--;  // (fft-convolver's init drops the end of the response below 1e-6)
        const uint64_t blk = (uint64_t)RQ * CONV_NOISE_BLOCK_QUANTA;
        ni.seg_count = (uint32_t)((l + blk - 1) / blk);
        ni.seg_mask = 0;
        for (uint64_t sgm = 0; sgm < ni.seg_count && sgm < 64; sgm++)
          for (uint64_t i = sgm * blk; i < std::min(l, (sgm + 1) * blk); i++)
            if (h[i] != 0.f) {
              ni.seg_mask |= (uint64_t)1 << sgm;
              break;
            }
      }
    }
    cst.loop_writes.push_back(n.sig.base);
    cst.profile_slot = slot_for(b, "conv_code_kernel");
    b->steps.push_back(cst);
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
    bool frozen_loop = false;
    if (unit.scc >= 0)
      for (uint32_t v : verts) {
        const Node& m = b->nodes[v & ~VTX_READER];
        if (m.desc.kind == WAA_NODE_CONVOLVER && m.has_ir) {
          // (round 5) a response of at most 24 x 128 frames has 128-frame partitions: its transforms can follow the loop quantum by
          // quantum like the frozen-state nodes.  Longer responses (a partition spans several quanta) and a mono response behind a
          // stereo input (channel 1 runs in compacted time) stay out of scope.
          const bool ok = !(v & VTX_READER) && conv_block_size(b, m) == RQ && !(m.ir_nch == 1 && m.in_nch == 2) && !measure_switch("WAA_NO_FROZEN_LOOPS");
          if (!ok)
            return fail(WAA_ERR_OUT_OF_SCOPE, "a ConvolverNode inside a feedback loop is out of scope (node %u)", v & ~VTX_READER);
          frozen_loop = true;
        }
        if (is_frozen_node(m)) {
          if (measure_switch("WAA_NO_FROZEN_LOOPS"))
            return fail(WAA_ERR_OUT_OF_SCOPE, "an oversampled WaveShaperNode / HRTF PannerNode inside a feedback loop is out of scope (node %u)",
                        v & ~VTX_READER);
          frozen_loop = true;
        }
      }
    // AudioParam inputs are summed by a node-major chain in front of the group: their producers must be complete
    bool param_dep = false;
    for (uint32_t v : verts)
      for (auto& pe : b->nodes[v & ~VTX_READER].pin_edges)
        for (int e : pe) {
          const uint32_t from = b->edges[e].from;
          if (unit.scc >= 0 && scc_of[from] == unit.scc) {
            // (round 6) a GainNode's gain from inside its own loop is rendered by the group itself (DynItem::pmod_*)
            const Node& owner = b->nodes[v & ~VTX_READER];
            const bool gain_param = owner.desc.kind == WAA_NODE_GAIN && &pe == &owner.pin_edges[0] && !frozen_loop;
            if (!gain_param)
              return fail(WAA_ERR_OUT_OF_SCOPE, "an AudioParam of node %u is modulated from inside its own feedback loop",
                          v & ~VTX_READER);
            continue;
          }
          param_dep |= in_pending(from);
        }
    if (param_dep)
      if (int e = flush()) return e;
    if (frozen_loop) {
      // ---- the loop as a chain of ranged launches, one quantum per block: [items up to a frozen-state node -> its mixed input]
      // [its link / transform / FIR launches] [the items behind it ...] — every launch over the same quantum, then the next quantum.
      // (One quantum: the loop's delays are at least that long, delay.rs:693-701, and a split delay pair's reader must see the ring
      // state of ITS quantum.)
      for (uint32_t v : verts)
        for (auto& pe : b->nodes[v & ~VTX_READER].pin_edges)
          for (int e : pe)
            if (scc_of[b->edges[e].from] == unit.scc)
              return fail(WAA_ERR_OUT_OF_SCOPE, "an AudioParam of node %u is modulated from inside its own feedback loop", v & ~VTX_READER);
      if (int e = flush()) return e;  // what was pending is complete before the loop starts
      const size_t first_loop_step = b->steps.size();
      cur_qgroup = (int)b->qgroup_quanta.size();
      b->qgroup_quanta.push_back(1);
      // segment of every vertex; delay pairs that the cuts split
      std::map<uint32_t, int> seg_of;
      {
        int sg = 0;
        for (uint32_t v : verts) {
          seg_of[v] = sg;
          if (!(v & VTX_READER) && is_cut_node(b->nodes[v])) sg++;
        }
      }
      xdelay.clear();
      for (uint32_t v : verts) {
        const uint32_t did = v & ~VTX_READER;
        if (!is_delay(b, did) || !(v & VTX_READER)) continue;
        const int rs = seg_of.at(v), ws = seg_of.at(did);
        if (rs == ws) continue;
        XDelay xd;
        xd.reader_seg = rs;
        xd.writer_seg = ws;
        int e = temp_signal(b, b->nodes[did].in_nch, &xd.line);
        if (!e) e = dev_alloc(b, &xd.aux32, (size_t)b->n_inst * cs);
        if (!e) e = dev_alloc(b, &xd.state, (size_t)b->n_inst * 2 + 4);  // (+ the block-violation flag behind the instances' pairs)
        if (e) return e;
        b->state_bufs.push_back({xd.state, ((size_t)b->n_inst * 2 + 4) * sizeof(int32_t)});  // (count 1, never mixed to mono: zeros)
        b->loop_flags.push_back(xd.state + (size_t)b->n_inst * 2);
        xdelay[did] = xd;
      }
      {
        // The block size (round-5 review, weak 6: one quantum per block = 15 000 launch sets for a 10 s render,
        // 333 ms for 1024 contexts).  Every path from a later segment back into an earlier one goes through a DelayNode whose
        // writer and reader the cuts separated; the reader of block [t, t + bq) reads the line up to frame
        // (t + bq) * 128 - delay + 1 (linear interpolation, delay.rs:560-590), which must have been written by EARLIER blocks:
        // bq <= (delay_frames - 1) / 128 for every split pair.  Known at plan time when delayTime is a constant; an automated or
        // modulated delayTime keeps one quantum (delay.rs:693-701 guarantees no more than that).
        uint32_t bq = xdelay.empty() ? 1u : b->n_quanta;
        for (auto& kv : xdelay) {
          const Node& dn = b->nodes[kv.first];
          const ParamStore& dp = dn.params[WAA_PARAM_DELAY_DELAY_TIME];
          const bool constant = dp.mode() == 0 && dp.timelines.empty() && !dp.dev_tl &&
                                (dn.pin_edges.size() <= (size_t)WAA_PARAM_DELAY_DELAY_TIME || dn.pin_edges[WAA_PARAM_DELAY_DELAY_TIME].empty());
          if (!constant) {
            bq = 1;
            break;
          }
          float mn = dp.cst.empty() ? 0.f : dp.cst[0];
          for (float v2 : dp.cst) mn = std::min(mn, v2);
          const double frames = std::floor((double)dp.fix(mn) * (double)b->sr);
          bq = std::min(bq, frames >= 2. * RQ + 1. ? (uint32_t)((frames - 1.) / RQ) : 1u);
        }
        // (what the delay does NOT bound: the reader's channel count follows the writer's input with ONE quantum of lag whatever the
        // delay time — ring[0].number_of_channels() — so a block is only right while that count stands still inside it.  It does
        // almost always (it moves when a source starts or ends and when the loop falls silent); the writer flags the blocks where it
        // did not, and the render is repeated one quantum per block: waa_abi.cpp::settle_loops.)
        if (measure_switch("WAA_LOOP_ONE_QUANTUM")) bq = 1;  // (A/B: the round-5 form)
        b->qgroup_quanta.back() = std::max(bq, 1u);
      }
      for (uint32_t v : verts) {
        const uint32_t vid = v & ~VTX_READER;
        Node& m = b->nodes[vid];
        if (!m.sig.base) {
          int e = alloc_signal(m);
          if (e) return e;
        }
        pending.push_back(v);
        pending_nodes.insert(vid);
        if (!(v & VTX_READER) && is_cut_node(m)) {
          if (int e = flush()) return e;
          if (!m.in_code) return fail(WAA_ERR_INVALID_STATE, "internal: input codes of node %u", vid);
          const size_t first = b->steps.size();
          int e = m.desc.kind == WAA_NODE_CONVOLVER ? plan_dyn_convolver(vid)
                  : m.desc.kind == WAA_NODE_PANNER  ? plan_hrtf(b, vid, -1)
                                                    : plan_oversampler(b, vid, -1);
          if (e) return e;
          for (size_t k2 = first; k2 < b->steps.size(); k2++) {
            Step& fs = b->steps[k2];
            if (fs.kind != 15 && fs.kind != 17 && fs.kind != 20 && fs.kind != 2 && fs.kind != 11)
              return fail(WAA_ERR_OUT_OF_SCOPE, "node %u inside a feedback loop: this form of the node has no ranged launch (step kind %d)", vid, fs.kind);
            if (fs.kind == 2 && fs.conv.block != RQ)
              return fail(WAA_ERR_OUT_OF_SCOPE, "a ConvolverNode inside a feedback loop is out of scope (node %u: %d-frame partitions)", vid, fs.conv.block);
            fs.qgroup = cur_qgroup;
            if (fs.kind == 15 || fs.kind == 11) {
              int32_t* lst = nullptr;
              const size_t ints = fs.kind == 11 ? (size_t)CONV_CODE_STATE_INTS : 4;
              if ((e = dev_alloc(b, &lst, (size_t)b->n_inst * ints))) return e;
              b->state_bufs.push_back({lst, (size_t)b->n_inst * ints * sizeof(int32_t)});
              (fs.kind == 15 ? fs.link.state : fs.ccode.state) = lst;
            }
          }
        }
      }
      if (int e = flush()) return e;
      // Everything else the members' planning launched — AudioParam tables and summing chains (automation, modulation from OUTSIDE
      // the loop: from inside is refused above), per-frame coefficient tables — depends on nothing inside the loop: once, over the
      // whole render, in front of the blocks (r05h: left untagged they cut the loop's launches into two runs — garbage, and not
      // even the same garbage twice).
      size_t n_ranged = 0;
      for (size_t k2 = first_loop_step; k2 < b->steps.size(); k2++) {
        Step& ls = b->steps[k2];
        const bool ranged = ls.kind == 10 || ls.kind == 15 || ls.kind == 17 || ls.kind == 20 || ls.kind == 2 || ls.kind == 11;
        if (ranged && ls.qgroup != cur_qgroup) return fail(WAA_ERR_INVALID_STATE, "internal: ranged launch outside its loop");
        if (!ranged) {
          const bool once = ls.kind == 5 || ls.kind == 12 || ls.kind == 13 || ls.kind == 14 || ls.kind == 3 ||
                            (ls.kind == 0 && ls.chain.n_ops == 1 && ls.chain.ops[0].kind == OP_PARAM_ADD);
          if (!once)
            return fail(WAA_ERR_OUT_OF_SCOPE, "a feedback loop with a frozen-state node inside needs a launch of kind %d per block: out of scope", ls.kind);
          ls.qgroup = cur_qgroup;
          ls.prologue = true;
        }
        n_ranged += ranged;
      }
      {
        const uint32_t bq = b->qgroup_quanta.back(), n_blocks = (b->n_quanta + bq - 1) / bq;
        plan_note(b, "feedback loop with a frozen-state node inside (oversampled WaveShaper / HRTF panner / short ConvolverNode): cut at the node(s), "
                     "%zu launch step(s) per block of %u render quanta (the shortest delay across a cut allows no more), %u blocks = %zu launches per render"
                     " — a correctness path: expect ~10 us of host + device time per launch, whatever the batch size",
                  n_ranged, bq, n_blocks, (size_t)n_blocks * n_ranged);
      }
      cur_qgroup = -1;
      xdelay.clear();
      continue;
    }
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
      if (int e = plan_dyn_convolver(id)) return e;
    }
  }
  if (int e = flush()) return e;
  if (measure_switch("WAA_DEBUG_REVERSE_PLAN")) std::reverse(b->steps.begin(), b->steps.end());
  if (int e = validate_plan(b)) return e;
  b->planned = true;
  return 0;
}

}  // namespace host
}  // namespace waa
