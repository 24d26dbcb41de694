// waa_conv_noise.hpp — WHERE the reference's ConvolverNode leaves FFT roundoff noise instead of exact zeros (round 5).
//
// ConvolverNode renders through the crate fft-convolver 0.3 (convolver.rs:284-306 init, :384-466 process; the test suite's CPU
// restatement of it is what this is checked against): 128-frame calls inside blocks of 1024 frames, every call an inverse transform of
//     pre (segments 1 … n-1 of the impulse response x the spectra of the previous n-1 blocks)  +  spectrum of the block so far x segment 0
// plus the overlap the LAST call of the previous block left.  A spectrum is exactly zero iff its block holds nothing but zeros, and
// the inverse transform of anything else carries roundoff (about 1e-7 of what the block and the overlap hold) on nearly every
// frame — so "does this quantum of the reference's output hold a non-zero sample" is a function of WHICH input quanta held a
// non-zero sample and which 1024-frame segments of the impulse response do, nothing else.  Nodes whose silence is data dependent (a
// DelayNode is silent when it read nothing but zeros, delay.rs:660-668) follow that noise; the device's overlap-save transforms are
// cleaner (exact zeros as soon as the window is), so dynamic plans floor / clear the convolver's output where this automaton says
// the reference's is noise / exact zeros (waa_dyn.hip: conv_code_kernel, conv_floor_kernel).  It matches the RESTATED crate: the
// real one (rustfft's butterflies) is not in /root/reference, and which of its samples are exact zeros is not pinned by anything.
//
// Plain C++ (no HIP): tools/conv_noise_check.cpp runs it on the CPU against that restatement (tests/test_conv_noise.py).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define WAA_CN_FN __host__ __device__ inline
#else
#define WAA_CN_FN inline
#endif

namespace waa {

constexpr int CONV_NOISE_BLOCK_QUANTA = 8;  // FFTConvolver::init(RENDER_QUANTUM_SIZE * 8, ..), convolver.rs:294

// one FFTConvolver (one per channel of the impulse response, at least two per node)
struct ConvNoiseIr {
  uint32_t seg_count;  // 1024-frame segments of the response after fft-convolver trimmed |h| < 1e-6 off its end (0: outputs zeros)
  uint32_t pad;
  uint64_t seg_mask;   // bit i: segment i (i < 64) holds a non-zero tap; segments >= 64 count as non-zero
};
struct ConvNoiseState {
  uint64_t hist;   // bit k: the block k + 1 blocks ago held a non-zero input sample
  uint32_t age;    // completed blocks since the last one that did (saturating; seg_count > 64 only)
  uint32_t flags;  // bits 0-3: quanta of the current block so far, 4: a non-zero sample in them, 5: the overlap is non-zero
};
WAA_CN_FN void conv_noise_reset(ConvNoiseState& s) {
  s.hist = 0;
  s.age = 0xffffffffu;
  s.flags = 0;
}
// one 128-frame call whose input holds (in_nonzero) / does not hold a non-zero sample: does the output?
WAA_CN_FN bool conv_noise_step(const ConvNoiseIr& ir, ConvNoiseState& s, bool in_nonzero) {
  if (ir.seg_count == 0) return false;
  uint32_t fill = s.flags & 15u;
  bool cur = (s.flags & 16u) != 0 || in_nonzero;
  const bool ovl = (s.flags & 32u) != 0;
  const bool cur_out = cur && (ir.seg_mask & 1u);
  bool pre;
  if (ir.seg_count <= 64) {
    const uint64_t live = ir.seg_count == 64 ? ~(uint64_t)0 : (((uint64_t)1 << ir.seg_count) - 1);  // segments 0 … n-1
    pre = (s.hist & ((ir.seg_mask & live) >> 1)) != 0;                                               // segment i x the block i blocks ago
  } else {
    pre = s.age != 0xffffffffu && s.age + 1 < ir.seg_count;
  }
  const bool noisy = cur_out || pre || ovl;
  if (++fill == (uint32_t)CONV_NOISE_BLOCK_QUANTA) {  // the block is full: its last inverse transform is the next block's overlap
    s.hist = (s.hist << 1) | (cur ? 1u : 0u);
    s.age = cur ? 0u : (s.age == 0xffffffffu ? s.age : s.age + 1u);
    s.flags = (cur_out || pre) ? 32u : 0u;
  } else {
    s.flags = fill | (cur ? 16u : 0u) | (ovl ? 32u : 0u);
  }
  return noisy;
}

// ---- one ConvolverNode: the tail counter and the routing of ConvolverRenderer::process (convolver.rs:343-392, :384-466) around
// its FFTConvolvers — one per channel of the response, at least two
struct ConvNoiseNode {
  uint64_t tail;         // frames of silent input processed since the last active quantum (tail_count)
  ConvNoiseState cv[4];
};
WAA_CN_FN void conv_noise_node_reset(ConvNoiseNode& s) {
  s.tail = 0;
  for (int k = 0; k < 4; k++) conv_noise_reset(s.cv[k]);
}
constexpr uint32_t CONV_NOISE_CUT = 0x80u;  // the tail has elapsed: silent output, the convolvers are not called (their blocks stand still)
// one render quantum: the input is coded silent (= one channel of zeros) or carries in_count (1, 2) channels, bit c of nz = channel
// c of it holds a non-zero sample.  Returns CONV_NOISE_CUT, or (output channels << 4) | bit c: output channel c is noise.
WAA_CN_FN uint32_t conv_noise_node_step(const ConvNoiseIr* ir, int ir_nch, uint64_t impulse_length, ConvNoiseNode& s, bool in_silent,
                                        int in_count, uint32_t nz) {
  if (in_silent) {
    if (s.tail >= impulse_length) return CONV_NOISE_CUT;
    s.tail += 128;
    in_count = 1;
    nz = 0;
  } else {
    s.tail = 0;
  }
  const bool l = (nz & 1u) != 0, r = in_count == 2 ? (nz & 2u) != 0 : l;
  if (ir_nch == 4) {  // true stereo: output c = convolver c of the left + convolver 2 + c of the right channel (a mono input: of itself)
    const bool o0 = conv_noise_step(ir[0], s.cv[0], l), o1 = conv_noise_step(ir[1], s.cv[1], l);
    const bool o2 = conv_noise_step(ir[2], s.cv[2], r), o3 = conv_noise_step(ir[3], s.cv[3], r);
    return (2u << 4) | ((o0 || o2) ? 1u : 0u) | ((o1 || o3) ? 2u : 0u);
  }
  if (in_count == 1 && ir_nch == 1) return (1u << 4) | (conv_noise_step(ir[0], s.cv[0], l) ? 1u : 0u);  // the second convolver is not called
  const bool o0 = conv_noise_step(ir[0], s.cv[0], l);
  const bool o1 = conv_noise_step(ir[1], s.cv[1], r);  // (mono input: both convolvers hear it)
  return (2u << 4) | (o0 ? 1u : 0u) | (o1 ? 2u : 0u);
}

// the floor a sample of a "noise" quantum is raised to: far below anything audible or testable (1e-20), far enough above the
// denormal range (1e-38) that the gains and filters behind it keep it normal as long as they keep the reference's noise normal
constexpr float CONV_NOISE_FLOOR = 1e-20f;

}  // namespace waa
