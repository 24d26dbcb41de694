// waa_plan_parts.hpp — what the parts of the planner share (waa_plan.cpp: order, liveness, chains, dynamic groups;
// waa_plan_sources.cpp, waa_plan_loops.cpp, waa_plan_conv.cpp, waa_plan_ops.cpp, waa_plan_check.cpp).  Host only.
#pragma once
#include <functional>
#include <vector>

#include "waa_host.hpp"

namespace waa {
namespace host {

// what a launch reads and writes (plan validation, and the prologue decision of block-scheduled loops)
struct StepIo {
  std::vector<const void*> reads, writes;
  bool feedback_reader = false;
};
struct OrderCtx;
// planning units: single nodes and whole feedback loops (scc >= 0), producers first
struct Unit {
  int scc;
  uint32_t id;
};
// what build_plan hands to the dynamic-group builder (waa_plan_dyn.cpp)
struct DynPlanCtx {
  const std::vector<uint32_t>& items;   // vertices in processing order (two per DelayNode)
  const std::vector<Unit>& units;
  const std::vector<int>& scc_of;
  std::function<int(Node&)> alloc_signal;
  std::function<int(uint32_t)> plan_single;
  bool count_change_found, mixed_buffer_counts;
};
int plan_dynamic_groups(waa_batch* b, const DynPlanCtx& c);
// vertices of the expanded graph the cycle breaker works on: a DelayNode is two (writer `id`, reader `id | VTX_READER`)
constexpr uint32_t VTX_READER = 0x80000000u;

int device_timeline_param(waa_batch* b, const ParamStore& p, ParamRef* ref);
int upload_param(waa_batch* b, const ParamStore& p, ParamRef* ref);
int param_mode(const Node& n, size_t k);
int node_param(waa_batch* b, uint32_t id, size_t k, ParamRef* ref);
int build_edge_input(waa_batch* b, uint32_t head, int ie, InputRef* out);
int upload_values(waa_batch* b, const std::vector<float>& host, int mode, ParamRef* ref);
int computed_in_nch(const Node& n, int maxc);
bool order_visit(OrderCtx& c, uint32_t v);
void plan_note(waa_batch* b, const char* fmt, ...);
const char* input_kind_name(int k);
const char* op_name(int k);
int slot_for(waa_batch* b, const char* name);
int push_chain_step(waa_batch* b, const std::vector<InputRef>& inputs, int in_nch, int in_interp, const std::vector<OpDesc>& ops, const SignalRef& out);
int temp_signal(waa_batch* b, int nch, SignalRef* out);
int emit_segments(waa_batch* b, std::vector<InputRef> inputs, int in_nch, int in_interp, const std::vector<OpDesc>& ops, const SignalRef& out);
bool is_delay(const waa_batch* b, uint32_t id);
void vertex_targets(const waa_batch* b, uint32_t v, const std::vector<uint8_t>& cut, std::vector<uint32_t>& out);
void vertex_targets(const waa_batch* b, uint32_t v, const std::vector<uint8_t>& cut, std::vector<uint32_t>& out);
int materialise_automation(waa_batch* b);
int source_code_rows(waa_batch* b, uint32_t id, uint64_t cs, std::vector<uint8_t>* out);
int build_plan(waa_batch* b);
void io_param(const ParamRef& p, StepIo& io);
void io_input(const InputRef& in, StepIo& io);
StepIo step_io(const Step& st);
int validate_plan(waa_batch* b);
void fuse_echo_tails(waa_batch* b);
void ring_feed_forward_echoes(waa_batch* b);
void fuse_fm_operators(waa_batch* b);
void fuse_lfo_params(waa_batch* b);
int plan_delay_writer(waa_batch* b, uint32_t id);
int plan_folded_delay_line(waa_batch* b, uint32_t id);
int plan_delay_reader(waa_batch* b, uint32_t id);
uint32_t loop_block_tiles(waa_batch* b, const std::vector<uint32_t>& loop_items);
int plan_loop(waa_batch* b, const std::vector<uint32_t>& loop_items);
int prepare_source_input(waa_batch* b, uint32_t id, InputRef* in);
int widen_narrow_buffers(waa_batch* b, Node& n, uint32_t nch);
int reduce_fan_in(waa_batch* b, std::vector<InputRef>& ins, int in_nch, int interp);
int premix_ordered_inputs(waa_batch* b, uint32_t id, std::vector<InputRef>& ins);
int node_input_signal(waa_batch* b, uint32_t id, SignalRef* out_sig, const SignalRef* target = nullptr, uint64_t* valid = nullptr);
int plan_oscillator(waa_batch* b, uint32_t id);
int conv_block_size(const waa_batch* b, const Node& n);
// source -> Biquad(the same constant coefficients on every context) -> long Convolver: Biquad and Convolver are both LTI, so
// (x * h_biquad) * h_ir = x * (h_biquad * h_ir): fills conv.ir_lti with the filtered impulse response when the filter's memory
// dies out inside the partitions the response occupies anyway (true: folded; the Biquad then costs nothing per render)
bool conv_fold_biquad_into_ir(const waa_batch* b, Node& conv, const Node& q);
int plan_convolver(waa_batch* b, uint32_t id);
int emit_node_ops(waa_batch* b, uint32_t id, int cur_nch, bool head, std::vector<OpDesc>& ops, int* out_nch);
void default_channel_config(Node& n, uint32_t n_out);

}  // namespace host
}  // namespace waa
