// waa_plan_sources.cpp — inputs of chains: AudioBufferSource / ConstantSource schedules and buffer tables, fan-in reduction,
// a node's mixed input as a signal, the OscillatorNode's plan (split out of waa_plan.cpp in round 4).
#include <array>
#include <set>

#include "waa_host.hpp"
#include "waa_plan_parts.hpp"

namespace waa {
namespace host {

// Instances of one batch may play AudioBuffers of different channel counts (audio_buffer_source.rs:560-600: the output of a
// quantum has the count of the instance's buffer).  Every kernel that reads a source walks `out_nch` channels of every
// instance's buffer, so a buffer narrower than the widest one is replaced by a copy with silent extra channels; the
// per-quantum codes of the dynamic-count plan (source_code_rows) carry the buffer's OWN count, which is what the consumers
// mix by — the extra channels are never looked at.  Copies are per distinct buffer, made once (they are payload: a re-plan
// finds them in place).
int widen_narrow_buffers(waa_batch* b, Node& n, uint32_t nch) {
  std::map<const float*, DeviceBuffer> done;
  for (auto& bf : n.bufs) {
    if (!bf.valid || bf.nch >= nch) continue;
    auto it = done.find(bf.base);
    if (it == done.end()) {
      const uint64_t stride = std::max<uint64_t>(bf.ch_stride, 4);
      float* d = nullptr;
      int e = dev_alloc(b, &d, (size_t)nch * stride, true);
      if (e) return e;
      const size_t all = (size_t)nch * stride * sizeof(float), plane = (size_t)bf.frames * sizeof(float);
      if (b->dry) {
        std::memset(d, 0, all);
        for (uint32_t c = 0; c < bf.nch && plane; c++) std::memcpy(d + (size_t)c * stride, bf.base + (size_t)c * bf.ch_stride, plane);
      } else {
        HIP_TRY(hipMemsetAsync(d, 0, all, b->stream));
        for (uint32_t c = 0; c < bf.nch && plane; c++)
          HIP_TRY(hipMemcpyAsync(d + (size_t)c * stride, bf.base + (size_t)c * bf.ch_stride, plane, hipMemcpyDeviceToDevice, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
      }
      DeviceBuffer w = bf;
      w.base = d;
      w.ch_stride = stride;
      w.nch_true = bf.count();
      w.nch = nch;
      it = done.emplace(bf.base, w).first;
    }
    bf = it->second;
  }
  return 0;
}

// Resolve a source node into an InputRef: schedules, per-instance buffer table, constant ranges.
int prepare_source_input(waa_batch* b, uint32_t id, InputRef* in) {
  PlanTrace trace_all("prepare_source_input");
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
  SchedKey last_key(0., 0., 0., 0., 0, 0., 0., 0, 0.f, 0.f, 0.f);
  uint32_t last_sched = 0;
  bool last_key_valid = false;
  for (uint32_t i = 0; i < b->n_inst; i++) {
    const DeviceBuffer& bf = n.bufs[i];
    SrcInst& si = insts[i];
    si.base = bf.base;
    si.ch_stride = bf.ch_stride;
    si.frames = bf.frames;
    si.aligned = (bf.valid && ((uintptr_t)bf.base % 16 == 0) && (bf.ch_stride % 4 == 0)) ? 1 : 0;
    const SourceSched& ss = n.sched[i];
    // (constant params: the key needs their one value — no per-quantum vectors for the 2047 contexts that share a schedule)
    const SchedKey key(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end,
                       bf.valid ? bf.frames : 0, bf.valid ? bf.sr : 0.f, p_rate.fix(p_rate.cst[i]), p_det.fix(p_det.cst[i]));
    if (!automated) {
      if (i > 0 && last_key_valid && key == last_key) {  // (the usual batch: every context like its neighbour — no tree walk)
        si.sched = last_sched;
        continue;
      }
      auto it = dedup.find(key);
      if (it != dedup.end()) {
        si.sched = last_sched = it->second;
        last_key = key;
        last_key_valid = true;
        continue;
      }
    }
    std::vector<float> rate_q = param_per_quantum(b, p_rate, i, nullptr);
    std::vector<float> det_q = param_per_quantum(b, p_det, i, nullptr);
    SchedOut own;
    std::shared_ptr<SchedOut> shared;
    {
      PlanTrace trace("source input: schedule_source");
      if (automated)
        schedule_source(b, n.sched[i], bf.frames, bf.sr, bf.valid, rate_q, det_q, &own);
      else
        shared = schedule_source_cached(b, id, key, n.sched[i], bf.frames, bf.sr, bf.valid, rate_q, det_q);
    }
    const SchedOut& so = automated ? own : *shared;
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
        // ... and the AudioBuffer reaches the END of the last render quantum: the consumers of a linear_all source load whole
        // quanta without looking at the buffer's length, and a buffer that ends inside the last quantum (a render length that is
        // no multiple of 128, the buffer as long as the render) made them read up to 127 frames behind it — the next plane's
        // samples, or, for the last plane, whatever lies behind the allocation: harmless to a causal consumer, but a CONVOLVER
        // behind it transforms the whole block, and 1e22 of stale memory in the block's unused tail is 1e15 of roundoff in its
        // valid part (suspend fuzz seed 16381 run behind other graphs, round 6; the generic loader zero-fills).
        all = all && start0 + (int64_t)b->n_quanta * RQ <= (int64_t)bf.frames;
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
    if (!automated) {
      dedup[key] = si.sched;
      last_key = key;
      last_sched = si.sched;
      last_key_valid = true;
    }
  }
  for (auto& si : insts) {
    si.sc = scheds[si.sched];
    si.linear_start = linear[si.sched].first;
    si.fast_prefix = si.aligned && !measure_switch("WAA_NO_LINEAR_PREFIX") ? linear[si.sched].second : 0;  // (switch: A/B aid)
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
  b->src_tables[d_insts] = insts;
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
    const SignalRef part{ptr, (uint64_t)in_nch * b->lp, b->lp, in_nch, 0};
    if ((e = push_chain_step(b, std::vector<InputRef>(ins.begin(), ins.begin() + MAX_INPUTS), in_nch, interp, {}, part))) return e;
    InputRef partial{};
    partial.kind = IN_SIGNAL;
    partial.nch = in_nch;
    partial.sig = part;
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
    if ((e = premix_ordered_inputs(b, id, ins))) return e;
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
  const bool check_replay = measure_switch("WAA_OSC_PLAN_CHECK") != nullptr;
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
    // one row of n_quanta records per DISTINCT replay (tq grows row by row), a row index per instance: 1024 contexts of one patch
    // used to get 1024 copies of the same 90 KB row (92 MB built, copied and uploaded per oscillator: most of a 45 ms plan)
    std::vector<OscQuantum> tq;
    std::vector<uint32_t> row_of(b->n_inst, 0);
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
          row_of[i] = it->second;
          continue;
        }
        replayed.emplace(key, (uint32_t)(tq.size() / b->n_quanta));
      }
      const size_t row = tq.size() / b->n_quanta;
      row_of[i] = (uint32_t)row;
      tq.resize(tq.size() + b->n_quanta);
      long double phase = 0.L;
      bool started = false;
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        OscQuantum& oq = tq[row * b->n_quanta + q];
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
    uint32_t* d_row = nullptr;
    if ((e = dev_upload(b, &d_tq, tq)) || (e = dev_upload(b, &d_row, row_of))) return e;
    d.table_q = d_tq;
    d.tq_row = d_row;
  }
  const bool scan = !parallel && !getenv("WAA_OSC_EXACT");
  if (scan) {
    // a-rate / graph-modulated frequency: the device forms the phase as a prefix sum of per-frame increments; the
    // host replays only the reference's clock (current_time += dt per frame, oscillator.rs:505-552) to find the
    // active frame range and the sub-sample start offset of every instance
    std::vector<int64_t> act((size_t)b->n_inst * 2, 0);
    std::vector<double> ratio(b->n_inst, 0.);
    const double sample_rate = (double)b->sr, dt = 1. / sample_rate;
    std::map<std::pair<double, double>, uint32_t> clock_of;  // (start, stop) -> the first instance replayed with them
    for (uint32_t i = 0; i < b->n_inst; i++) {
      double start_time = start[i];
      const double stop_time = stop[i];
      {
        auto it = clock_of.find({start_time, stop_time});
        if (it != clock_of.end()) {  // the clock replay depends on nothing else
          act[(size_t)i * 2] = act[(size_t)it->second * 2];
          act[(size_t)i * 2 + 1] = act[(size_t)it->second * 2 + 1];
          ratio[i] = ratio[it->second];
          continue;
        }
        clock_of.emplace(std::make_pair(start_time, stop_time), i);
      }
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

// Two-operator FM (oscillator -> [Gain] -> carrier.frequency): three launches — the modulator's table-driven kernel, the
// AudioParam's summing chain (PARAM_ADD), the carrier's prefix-sum kernel, which reads the per-frame frequency table in both of
// its passes — moved 13.8 GB to produce 3.9 GB of output.  When the frequency param has ONE input, a materialised oscillator
// with a host-known frequency that nothing else reads, the carrier evaluates modulator, edge gain and
// AudioParamProcessor::mix_to_output per frame itself (OscDesc::fm_*: the same arithmetic as the two launches it stands for,
// bit for bit) and those launches are not run.  Decided on the finished launch list like fuse_echo_tails.
void fuse_fm_operators(waa_batch* b) {
  if (measure_switch("WAA_NO_FM_FOLD")) return;
  for (size_t c = 0; c < b->steps.size(); c++) {
    Step& cs = b->steps[c];
    if (cs.kind != 9 || !cs.osc.active || cs.osc.frequency.mode != 2 || cs.osc.fm_q || cs.group >= 0) continue;
    const void* table = cs.osc.frequency.base;
    int pi = -1, mi = -1;
    for (size_t k = 0; k < c; k++) {
      const Step& st = b->steps[k];
      if (st.kind == 0 && st.group < 0 && !st.echo_fused && (const void*)st.chain.out.base == table) pi = (int)k;
    }
    if (pi < 0) continue;
    Step& ps = b->steps[(size_t)pi];
    const ChainDesc& ch = ps.chain;
    if (ch.n_ops != 1 || ch.ops[0].kind != OP_PARAM_ADD || ch.n_inputs != 1 || ch.in_nch != 1 || ch.in[0].kind != IN_SIGNAL ||
        ch.in[0].nch != 1 || ch.out.inst_stride != cs.osc.frequency.stride || ch.ops[0].p0.mode > 1 ||
        (ch.in[0].has_gain && ch.in[0].gain.mode > 1) || (ch.in[0].valid != 0 && ch.in[0].valid < (uint64_t)b->n_quanta * RQ))
      continue;
    for (size_t k = 0; k < (size_t)pi; k++) {
      const Step& st = b->steps[k];
      if (st.kind == 9 && st.group < 0 && !st.echo_fused && st.osc.table_q && st.osc.out.base == ch.in[0].sig.base) mi = (int)k;
    }
    if (mi < 0) continue;
    Step& ms = b->steps[(size_t)mi];
    if (ms.osc.n_post != 0 || ms.osc.post_dup || ms.osc.out.nch != 1 || ms.osc.out.inst_stride != ch.in[0].sig.inst_stride) continue;
    // nothing else may read the param's table or the modulator's signal (launches, analysers, the destination)
    bool shared = false;
    for (size_t k = 0; k < b->steps.size() && !shared; k++) {
      if ((int)k == pi || k == c) continue;
      const StepIo io = step_io(b->steps[k]);
      for (const void* r : io.reads) shared |= r == table || ((int)k != pi && r == (const void*)ms.osc.out.base);
    }
    {
      const StepIo io = step_io(cs);  // (the carrier itself: its detune could be modulated from the same oscillator)
      for (const void* r : io.reads) shared |= r == (const void*)ms.osc.out.base;
    }
    for (const Node& an : b->nodes)
      if (an.live && (an.desc.kind == WAA_NODE_ANALYSER || an.desc.kind == WAA_NODE_DESTINATION) &&
          (an.sig.base == ms.osc.out.base || (const void*)an.sig.base == table))
        shared = true;
    if (shared) continue;
    OscDesc& d = cs.osc;
    d.fm_q = ms.osc.table_q;
    d.fm_row = ms.osc.tq_row;
    d.fm_table = ms.osc.table;
    d.fm_table_len = ms.osc.table_len;
    d.fm_type = ms.osc.type;
    d.fm_has_gain = ch.in[0].has_gain ? 1 : 0;
    d.fm_gain = ch.in[0].gain;
    std::memcpy(&d.fm_min, &ch.ops[0].i0, 4);
    std::memcpy(&d.fm_max, &ch.ops[0].i1, 4);
    std::memcpy(&d.fm_default, &ch.ops[0].i2, 4);
    d.frequency = ch.ops[0].p0;  // the intrinsic value (constant or one per quantum)
    ms.echo_fused = true;
    ps.echo_fused = true;
    plan_note(b, "FM: launches %d (the modulating oscillator) and %d (the frequency param's sum) are folded into the carrier's prefix-sum kernel (launch %zu): modulator, %sparam sum and clamp per frame in registers",
              mi, pi, c, d.fm_has_gain ? "edge gain, " : "");
  }
}

// An LFO (an oscillator with a host-known frequency, nothing else reading it) on ONE AudioParam — tremolo, a filter sweep, a
// wobbling delay: the param's summing chain (edge gain + PARAM_ADD) was a launch of its own that read the oscillator's signal
// and wrote the per-frame table.  The time-parallel oscillator kernel applies both in its store (OscDesc::post_gain / pa_*: the
// chain kernel's arithmetic, bit for bit) and writes the table directly; the chain launch is not run.  After fuse_fm_operators
// (an oscillator's frequency param takes the modulator all the way into the carrier).
void fuse_lfo_params(waa_batch* b) {
  if (measure_switch("WAA_NO_LFO_FOLD")) return;
  for (size_t pi = 0; pi < b->steps.size(); pi++) {
    Step& ps = b->steps[pi];
    if (ps.kind != 0 || ps.group >= 0 || ps.echo_fused) continue;
    const ChainDesc& ch = ps.chain;
    if (ch.n_ops != 1 || ch.ops[0].kind != OP_PARAM_ADD || ch.n_inputs != 1 || ch.in_nch != 1 || ch.in[0].kind != IN_SIGNAL ||
        ch.in[0].nch != 1 || ch.out.nch != 1 || ch.ops[0].p0.mode > 1 || (ch.in[0].has_gain && ch.in[0].gain.mode != 0) ||
        (ch.in[0].valid != 0 && ch.in[0].valid < (uint64_t)b->n_quanta * RQ))
      continue;
    int mi = -1;
    for (size_t k = 0; k < pi; k++) {
      const Step& st = b->steps[k];
      if (st.kind == 9 && st.group < 0 && !st.echo_fused && st.osc.table_q && !st.osc.pa_on && st.osc.out.base == ch.in[0].sig.base) mi = (int)k;
    }
    if (mi < 0) continue;
    Step& ms = b->steps[(size_t)mi];
    if (ms.osc.post_dup || ms.osc.out.nch != 1 || ms.osc.n_post + (ch.in[0].has_gain ? 1 : 0) > 2 ||
        ms.osc.out.inst_stride != ch.in[0].sig.inst_stride || ch.out.inst_stride != ms.osc.out.inst_stride)
      continue;
    bool shared = false;
    for (size_t k = 0; k < b->steps.size() && !shared; k++) {
      if (k == pi) continue;
      const StepIo io = step_io(b->steps[k]);
      for (const void* r : io.reads) shared |= r == (const void*)ms.osc.out.base;
    }
    for (const Node& an : b->nodes)
      if (an.live && (an.desc.kind == WAA_NODE_ANALYSER || an.desc.kind == WAA_NODE_DESTINATION) && an.sig.base == ms.osc.out.base) shared = true;
    if (shared) continue;
    OscDesc& d = ms.osc;
    if (ch.in[0].has_gain) d.post_gain[d.n_post++] = ch.in[0].gain;
    d.pa_on = 1;
    d.pa_intrinsic = ch.ops[0].p0;
    std::memcpy(&d.pa_min, &ch.ops[0].i0, 4);
    std::memcpy(&d.pa_max, &ch.ops[0].i1, 4);
    std::memcpy(&d.pa_default, &ch.ops[0].i2, 4);
    d.out = ch.out;  // the oscillator writes the param's per-frame table itself
    ps.echo_fused = true;
    plan_note(b, "LFO: launch %zu (the param's sum: %sintrinsic value, clamp) is folded into the store of the oscillator that drives it (launch %d)", pi,
              ch.in[0].has_gain ? "depth gain, " : "", mi);
  }
}

}  // namespace host
}  // namespace waa
