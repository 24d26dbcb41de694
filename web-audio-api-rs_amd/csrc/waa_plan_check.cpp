// waa_plan_check.cpp — what every launch of a plan reads and writes (step_io) and the read-before-write validation of the
// finished launch list (split out of waa_plan.cpp in round 4).
#include <array>
#include <set>

#include "waa_host.hpp"
#include "waa_plan_parts.hpp"

namespace waa {
namespace host {

// ---- plan validation -----------------------------------------------------------------------------------------
// The plan is a linear list of launches over shared device buffers; nothing but their order makes a consumer see
// its producer's data.  This check walks the list once and refuses a plan in which a launch reads a buffer that
// some launch of the plan writes, but none has written yet — an ordering bug of the planner would otherwise
// render stale or zero data silently.  The only legal read-before-write is a DelayNode reader inside a feedback
// loop (it reads the PREVIOUS quanta of a line that is filled later in the same pass).
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

}  // namespace host
}  // namespace waa
