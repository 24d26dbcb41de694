"""Host-side mirror of the reference's control API for the offline render path.

The names follow the reference crate (web-audio-api 1.6.0):

* ``OfflineAudioContext(number_of_channels, length, sample_rate)``  src/context/offline.rs:78
* ``create_buffer_source / create_biquad_filter / create_gain / create_convolver /
  create_stereo_panner / create_panner / create_analyser / create_wave_shaper /
  create_constant_source``                                         src/context/base.rs:23-367
* ``AudioNode.connect``                                            src/node/audio_node.rs:247-289
* ``AudioScheduledSourceNode.start/start_at/stop/stop_at``         src/node/scheduled_source.rs:6-44
* ``AudioParam.value / set_value``                                 src/param.rs:268-662
* ``start_rendering_sync``                                         src/context/offline.rs:157-185

The one extension is ``n_instances``: a context object stands for a *batch* of N
identically shaped OfflineAudioContexts rendered together by one ``waa_batch``
(include/waa_hip.h).  Per-instance payloads are addressed with ``instance=k``.

This module only talks to a C-ABI library through ctypes.  Which library is decided by
the caller (``bind(lib, prefix)``): the product binds ``libwaa_hip.so`` with prefix
``waa_`` (see __init__.py); the tests may bind the CPU oracle with prefix ``orc_``.
Nothing in this package imports or loads the oracle.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np

RENDER_QUANTUM_SIZE = 128
PLAN_ONLY = -2  # WAA_DEVICE_PLAN_ONLY: configure + plan without a device (never renders)
ALL = 0xFFFFFFFF
ENDED_NEVER, ENDED_AT_UNLOAD = -1, -2
F64_MAX = 1.7976931348623157e308

# node kinds / enums (include/waa_hip.h)
NODE_DESTINATION, NODE_BUFFER_SOURCE, NODE_BIQUAD, NODE_GAIN, NODE_CONVOLVER = 0, 1, 2, 3, 4
NODE_STEREO_PANNER, NODE_PANNER, NODE_ANALYSER, NODE_WAVESHAPER, NODE_CONSTANT_SOURCE = 5, 6, 7, 8, 9
NODE_IIR_FILTER = 10
NODE_DELAY = 11
NODE_OSCILLATOR = 12
OSCILLATOR_TYPE = {"sine": 0, "square": 1, "sawtooth": 2, "triangle": 3, "custom": 4}
MAX_IIR_COEFFS = 20
PARAM_INPUT = 0x80000000  # WAA_PARAM_INPUT(param): edge into an AudioParam of the target node
(EVENT_SET_VALUE, EVENT_SET_VALUE_AT_TIME, EVENT_LINEAR_RAMP, EVENT_EXPONENTIAL_RAMP, EVENT_CANCEL_SCHEDULED_VALUES,
 EVENT_SET_TARGET, EVENT_CANCEL_AND_HOLD, EVENT_SET_VALUE_CURVE) = range(8)  # WAA_EVENT_* (src/param.rs:151-160)
COUNT_MODE = {"max": 0, "clamped-max": 1, "explicit": 2}
INTERPRETATION = {"speakers": 0, "discrete": 1}
BIQUAD_TYPE = {"lowpass": 0, "highpass": 1, "bandpass": 2, "notch": 3, "allpass": 4, "peaking": 5,
               "lowshelf": 6, "highshelf": 7}
PANNING_MODEL = {"equalpower": 0, "HRTF": 1}
DISTANCE_MODEL = {"linear": 0, "inverse": 1, "exponential": 2}
OVERSAMPLE = {"none": 0, "2x": 1, "4x": 2}


class WaaError(RuntimeError):
    """Raised where the reference panics (the message keeps the W3C error name)."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class NodeDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("channel_count", C.c_uint32), ("channel_count_mode", C.c_uint32),
                ("channel_interpretation", C.c_uint32), ("i", C.c_int32 * 4), ("d", C.c_double * 8)]


class EdgeDesc(C.Structure):
    _fields_ = [("from_", C.c_uint32), ("from_output", C.c_uint32), ("to", C.c_uint32), ("to_input", C.c_uint32)]


class GraphDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("nodes", C.POINTER(NodeDesc)), ("n_edges", C.c_uint32),
                ("edges", C.POINTER(EdgeDesc))]



_FP = C.POINTER(C.c_float)
_DP = C.POINTER(C.c_double)
_FPP = C.POINTER(_FP)
_VP = C.c_void_p

# waa_render_sharded (include/waa_hip.h): the callbacks run on the library's sub-batch threads (ctypes takes the GIL for them)
SHARD_FN = C.CFUNCTYPE(C.c_int32, _VP, C.c_uint32, C.c_uint32, C.c_int32, _VP)


class ShardedJob(C.Structure):
    _fields_ = [("graph", C.POINTER(GraphDesc)), ("n_instances", C.c_uint32), ("n_channels_out", C.c_uint32),
                ("length_frames", C.c_uint64), ("sample_rate", C.c_float), ("n_devices", C.c_uint32),
                ("devices", C.POINTER(C.c_int32)), ("sub_batches", C.c_uint32), ("source_node", C.c_uint32),
                ("host_in", _VP), ("in_channels", C.c_uint32), ("in_pcm16", C.c_int32), ("in_frames", C.c_uint64),
                ("in_sample_rate", C.c_float), ("out_pcm16", C.c_int32), ("host_out", _VP), ("setup", SHARD_FN),
                ("pull", SHARD_FN), ("user", _VP), ("reuse_batches", C.c_uint32)]

class ArenaStats(C.Structure):
    """waa_arena_stats (include/waa_hip.h)"""
    _fields_ = [(n, C.c_uint64) for n in ("reserved_bytes", "in_use_bytes", "peak_bytes", "largest_free_bytes", "served", "misses",
                                          "miss_bytes")]


class ArenaGrades(C.Structure):
    """waa_arena_grades (include/waa_hip.h)"""
    _fields_ = [("unit_bytes", C.c_uint64), ("n_units", C.c_uint32), ("n_candidates", C.c_uint32), ("best_ms", C.c_float),
                ("worst_kept_ms", C.c_float), ("worst_candidate_ms", C.c_float), ("grading_ms", C.c_float)]


def arena_grades(binding, device=-1) -> dict:
    """waa_device_arena_grades as a dict (+ "unit_ms": the kept units' grades in address order); all zero without a graded arena"""
    g = ArenaGrades()
    binding.check(binding.device_arena_grades(int(device), C.cast(C.pointer(g), _VP), None, 0))
    out = {n: (int if n in ("unit_bytes", "n_units", "n_candidates") else float)(getattr(g, n)) for n, _ in ArenaGrades._fields_}
    ms = (C.c_float * max(int(g.n_units), 1))()
    binding.check(binding.device_arena_grades(int(device), C.cast(C.pointer(g), _VP), ms, int(g.n_units)))
    out["unit_ms"] = [float(ms[k]) for k in range(int(g.n_units))]
    return out


def arena_stats(binding, device=-1) -> dict:
    """waa_device_arena_stats as a dict (all zero when no arena is reserved on the device)"""
    st = ArenaStats()
    binding.check(binding.device_arena_stats(int(device), C.cast(C.pointer(st), _VP)))
    return {n: int(getattr(st, n)) for n, _ in ArenaStats._fields_}


# name -> (restype, argtypes); every symbol include/waa_hip.h declares
ABI = {
    "batch_create": (C.c_int32, [C.POINTER(GraphDesc), C.c_uint32, C.c_uint32, C.c_uint64, C.c_float, C.c_int32,
                                 C.POINTER(_VP)]),
    "batch_destroy": (None, [_VP]),
    "last_error": (C.c_char_p, []),
    "device_count": (C.c_int32, []),
    "device_arena_reserve": (C.c_int32, [C.c_int32, C.c_uint64]),
    "device_arena_stats": (C.c_int32, [C.c_int32, _VP]),
    "batch_rearm": (C.c_int32, [_VP]),
    "render_range": (C.c_int32, [_VP, C.c_uint64, C.c_uint32]),
    "connect": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "disconnect": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "device_arena_reserve_graded": (C.c_int32, [C.c_int32, C.c_uint64, C.c_uint64]),
    "device_arena_grades": (C.c_int32, [C.c_int32, _VP, _FP, C.c_uint32]),
    "source_set_buffer": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, _FPP, C.c_uint32, C.c_uint64, C.c_float]),
    "source_set_buffer_batch": (C.c_int32, [_VP, C.c_uint32, _FP, C.c_uint32, C.c_uint64, C.c_float]),
    "source_set_buffer_pcm16": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_int16), C.c_uint32, C.c_uint64, C.c_float]),
    "source_set_buffer_pcm16_batch": (C.c_int32, [_VP, C.c_uint32, C.POINTER(C.c_int16), C.c_uint32, C.c_uint64, C.c_float]),
    "convolver_set_buffer_pcm16": (C.c_int32, [_VP, C.c_uint32, C.POINTER(C.c_int16), C.c_uint32, C.c_uint64, C.c_float]),
    "source_adopt_device": (C.c_int32, [_VP, C.c_uint32, _VP, C.c_uint32, C.c_uint64, C.c_float]),
    "source_start": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double]),
    "source_stop": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.c_double]),
    "source_set_loop": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.c_int32, C.c_double, C.c_double]),
    "source_ended": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_int64)]),
    "convolver_set_buffer": (C.c_int32, [_VP, C.c_uint32, _FPP, C.c_uint32, C.c_uint64, C.c_float]),
    "waveshaper_set_curve": (C.c_int32, [_VP, C.c_uint32, _FP, C.c_uint32]),
    "oscillator_set_periodic_wave": (C.c_int32, [_VP, C.c_uint32, _FP, _FP, C.c_uint32, C.c_int32]),
    "oscillator_set_wavetable": (C.c_int32, [_VP, C.c_uint32, _FP, C.c_uint32]),
    "iir_set_coefficients": (C.c_int32, [_VP, C.c_uint32, _DP, C.c_uint32, _DP, C.c_uint32]),
    "iir_frequency_response": (C.c_int32, [_DP, C.c_uint32, _DP, C.c_uint32, C.c_float, _FP, _FP, _FP, C.c_uint32]),
    "param_schedule_event": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_float, C.c_double, C.c_double,
                                         _FP, C.c_uint32]),
    "timeline_create": (_VP, [C.c_float, C.c_float, C.c_float, C.c_int32]),
    "timeline_destroy": (None, [_VP]),
    "timeline_event": (C.c_int32, [_VP, C.c_int32, C.c_float, C.c_double, C.c_double, _FP, C.c_uint32]),
    "timeline_compute": (C.c_uint32, [_VP, C.c_double, C.c_double, C.c_uint32, _FP]),
    "timeline_value": (C.c_float, [_VP]),
    "timeline_render_device": (C.c_int32, [_VP, C.c_uint32, C.c_float, _FP, C.POINTER(C.c_uint8)]),
    "set_param_const": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float]),
    "set_param_block": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, _FP]),
    "render": (C.c_int32, [_VP]),
    "sync": (C.c_int32, [_VP]),
    "download": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, _FP, C.c_uint64]),
    "download_all": (C.c_int32, [_VP, _FP]),
    "download_all_pcm16": (C.c_int32, [_VP, C.POINTER(C.c_int16)]),
    "render_sharded": (C.c_int32, [_VP, C.POINTER(C.c_double)]),
    "sharded_in_flight": (C.c_int32, [C.c_uint32]),
    "shard_range": (C.c_int32, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "output_device": (C.c_int32, [_VP, C.POINTER(_VP), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "analyser_get_float_frequency_data": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, _FP, C.c_uint32]),
    "analyser_get_byte_frequency_data": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32]),
    "analyser_get_float_time_domain_data": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, _FP, C.c_uint32]),
    "analyser_get_byte_time_domain_data": (C.c_int32, [_VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32]),
    "analyser_get_float_frequency_data_batch": (C.c_int32, [_VP, C.c_uint32, _FP, C.c_uint32]),
    "analyser_get_byte_frequency_data_batch": (C.c_int32, [_VP, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32]),
    "analyser_get_float_time_domain_data_batch": (C.c_int32, [_VP, C.c_uint32, _FP, C.c_uint32]),
    "analyser_get_byte_time_domain_data_batch": (C.c_int32, [_VP, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32]),
    "hrtf_load_sphere": (C.c_int32, [C.c_char_p, C.c_uint64]),
    "hrtf_hrir_length": (C.c_uint32, [C.c_float]),
    "hrtf_sample": (None, [C.c_float, _FP, _FP, _FP]),
    "buffer_resample": (C.c_uint64, [_FP, C.c_uint64, C.c_float, C.c_float, _FP, C.c_uint64]),
    "biquad_frequency_response": (C.c_int32, [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _FP,
                                              _FP, _FP, C.c_uint32]),
    "plan_describe": (C.c_int32, [_VP, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "profile_enable": (C.c_int32, [_VP, C.c_int32]),
    "profile_count": (C.c_int32, [_VP]),
    "profile_get": (C.c_int32, [_VP, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "profile_reset": (C.c_int32, [_VP]),
}


class Binding:
    """ctypes view of one C-ABI library (prefix ``waa_`` for the product)."""

    def __init__(self, lib: C.CDLL, prefix: str):
        self.lib = lib
        self.prefix = prefix
        for name, (restype, argtypes) in ABI.items():
            fn = getattr(lib, prefix + name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)

    def check(self, status: int):
        if status != 0:
            msg = self.last_error()
            raise WaaError(status, msg.decode() if msg else f"status {status}")


def bind(lib: C.CDLL, prefix: str = "waa_") -> Binding:
    return Binding(lib, prefix)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_FP)


def _chan_ptrs(arr2d: np.ndarray):
    ptrs = (_FP * arr2d.shape[0])()
    for c in range(arr2d.shape[0]):
        ptrs[c] = arr2d[c].ctypes.data_as(_FP)
    return ptrs


class AudioBuffer:
    """src/buffer.rs:69 — channels x frames of f32 plus a sample rate."""

    def __init__(self, data, sample_rate: float):
        data = _f32(data)
        if data.ndim == 1:
            data = data[None, :]
        self.data = data
        self.sample_rate = float(np.float32(sample_rate))

    @property
    def number_of_channels(self) -> int:
        return self.data.shape[0]

    @property
    def length(self) -> int:
        return self.data.shape[1]

    @property
    def duration(self) -> float:
        return self.length / self.sample_rate

    def get_channel_data(self, c: int) -> np.ndarray:
        return self.data[c]


class AudioParam:
    """Value source for one AudioParam of one node (src/param.rs:268).

    Constants, explicit per-quantum value blocks and automation events all cross the C ABI; the timeline itself is
    evaluated by the library on the host (automation is control-side work)."""

    def __init__(self, node: "AudioNode", pid: int, default: float):
        self._node, self._pid = node, pid
        self._const = {ALL: float(default)}
        self._blocks = []
        self._events = []

    def _ctx(self):
        """the context whose control clock this param follows (an AudioListener's params: the listener's context)"""
        return self._node.context if self._node is not None else getattr(self, "_listener_ctx", None)

    def _owner_ids(self, ctx):
        """the node ids the param is addressed through (the listener's nine params: every PannerNode, like AudioListener::_apply)"""
        if self._node is not None:
            return [self._node.id]
        return [nd.id for nd in ctx._nodes if nd.kind == NODE_PANNER]

    @property
    def value(self) -> float:
        return self._const[ALL]

    @value.setter
    def value(self, v: float):
        self.set_value(v)

    def set_value(self, v: float, instance: int = ALL):
        ctx = self._ctx()
        if ctx is not None and ctx._now_q > 0:  # inside a suspend_sync callback: a SetValue event the render thread handles at that quantum
            pid, v = self._pid, float(v)
            for nid in self._owner_ids(ctx):
                ctx._log(lambda b, h, nid=nid: b.check(b.set_param_const(h, nid, pid, instance, v)))
            return self
        self._const[instance] = float(v)
        if instance == ALL:
            self._const = {ALL: float(v)}
        return self

    def set_block(self, quantum0: int, values, instance: int = ALL):
        """values: [n_quanta] (k-rate, len-1 slices) or [n_quanta, 128] (a-rate slices)."""
        v = _f32(values)
        if v.ndim == 1:
            v = v[:, None]
        assert v.shape[1] in (1, RENDER_QUANTUM_SIZE)
        self._blocks.append((int(quantum0), v, instance))
        return self

    # -- automation methods (src/param.rs:428-596).  The events cross the ABI (waa_param_schedule_event) and are
    #    evaluated by the library's restatement of AudioParamProcessor; nothing is computed in this mirror.
    #    `instance`: the context of the batch the call is made on (ALL = the same call on every context).
    def _event(self, kind: int, value: float, time: float, aux: float = 0.0, curve=None, instance: int = ALL):
        ev = (kind, float(value), float(time), float(aux), None if curve is None else _f32(curve), instance)
        ctx = self._ctx()
        if ctx is not None and ctx._now_q > 0:  # inside a suspend_sync callback
            pid = self._pid
            for nid in self._owner_ids(ctx):
                ctx._log(lambda b, h, nid=nid: b.check(b.param_schedule_event(h, nid, pid, ev[5], ev[0], ev[1], ev[2], ev[3],
                                                                              None if ev[4] is None else _fp(ev[4]), 0 if ev[4] is None else ev[4].size)))
            return self
        self._events.append(ev)
        return self

    def set_value_at_time(self, value: float, start_time: float, instance: int = ALL):
        return self._event(EVENT_SET_VALUE_AT_TIME, value, start_time, instance=instance)

    def linear_ramp_to_value_at_time(self, value: float, end_time: float, instance: int = ALL):
        return self._event(EVENT_LINEAR_RAMP, value, end_time, instance=instance)

    def exponential_ramp_to_value_at_time(self, value: float, end_time: float, instance: int = ALL):
        return self._event(EVENT_EXPONENTIAL_RAMP, value, end_time, instance=instance)

    def set_target_at_time(self, value: float, start_time: float, time_constant: float, instance: int = ALL):
        return self._event(EVENT_SET_TARGET, value, start_time, time_constant, instance=instance)

    def cancel_scheduled_values(self, cancel_time: float, instance: int = ALL):
        return self._event(EVENT_CANCEL_SCHEDULED_VALUES, 0.0, cancel_time, instance=instance)

    def cancel_and_hold_at_time(self, cancel_time: float, instance: int = ALL):
        return self._event(EVENT_CANCEL_AND_HOLD, 0.0, cancel_time, instance=instance)

    def set_value_curve_at_time(self, values, start_time: float, duration: float, instance: int = ALL):
        return self._event(EVENT_SET_VALUE_CURVE, 0.0, start_time, duration, values, instance=instance)

    def _apply(self, ctx: "OfflineAudioContext", node_id=None, pid=None):
        b, h = ctx._b, ctx._handle
        node_id = self._node.id if node_id is None else node_id
        pid = self._pid if pid is None else pid
        for inst, v in sorted(self._const.items(), key=lambda kv: kv[0] != ALL):
            b.check(b.set_param_const(h, node_id, pid, inst, v))
        for q0, v, inst in self._blocks:
            b.check(b.set_param_block(h, node_id, pid, inst, q0, v.shape[0], v.shape[1], _fp(v)))
        for kind, value, time, aux, curve, inst in self._events:  # in call order, like the reference's message queue
            b.check(b.param_schedule_event(h, node_id, pid, inst, kind, value, time, aux,
                                           None if curve is None else _fp(curve), 0 if curve is None else curve.size))


class AudioNode:
    kind = -1
    default_channel_config = (2, "max", "speakers")

    def __init__(self, ctx: "OfflineAudioContext", channel_count=None, channel_count_mode=None,
                 channel_interpretation=None):
        self.context = ctx
        cc, mode, interp = self.default_channel_config
        self.channel_count = cc if channel_count is None else channel_count
        self.channel_count_mode = mode if channel_count_mode is None else channel_count_mode
        self.channel_interpretation = interp if channel_interpretation is None else channel_interpretation
        self._explicit_config = not (channel_count is None and channel_count_mode is None and
                                     channel_interpretation is None)
        self.params: List[AudioParam] = []
        self.id = ctx._register(self)

    # AudioNode channel config setters (audio_node.rs:296-340)
    def set_channel_count(self, v: int):
        self.channel_count = int(v)
        self._explicit_config = True

    def set_channel_count_mode(self, v: str):
        self.channel_count_mode = v
        self._explicit_config = True

    def set_channel_interpretation(self, v: str):
        self.channel_interpretation = v
        self._explicit_config = True

    # AudioNode::connect (audio_node.rs:247): returns the destination node for chaining
    def connect(self, dest, output: int = 0, input: int = 0):
        """AudioNode::connect; `dest` may be an AudioParam (audio-rate modulation, src/param.rs:300-320)."""
        if isinstance(dest, AudioParam):
            if dest._node.context is not self.context:
                raise WaaError(1, "InvalidAccessError - Attempting to connect nodes from different contexts")
            self.context._connect((self.id, output, dest._node.id, PARAM_INPUT | dest._pid))
            return dest
        if dest.context is not self.context:
            raise WaaError(1, "InvalidAccessError - Attempting to connect nodes from different contexts")
        self.context._connect((self.id, output, dest.id, input))
        return dest

    # AudioNode::disconnect and its four narrower forms (audio_node.rs:291-420); `dest` may be an AudioParam
    def disconnect(self, dest=None, output: Optional[int] = None, input: Optional[int] = None):
        if dest is not None:
            dctx = dest._node.context if isinstance(dest, AudioParam) else dest.context
            if dctx is not self.context:
                raise WaaError(1, "InvalidAccessError - Attempting to disconnect nodes from different contexts")
        if output is not None and output != 0:
            raise WaaError(1, f"IndexSizeError - output port {output} is out of bounds")
        to = None if dest is None else (dest._node.id if isinstance(dest, AudioParam) else dest.id)
        ti = None if dest is None else (PARAM_INPUT | dest._pid if isinstance(dest, AudioParam) else input)
        self.context._disconnect(self.id, output, to, ti)

    def disconnect_dest(self, dest):
        self.disconnect(dest)

    def disconnect_output(self, output: int):
        self.disconnect(None, output)

    def disconnect_dest_from_output(self, dest, output: int):
        self.disconnect(dest, output)

    def disconnect_dest_from_output_to_input(self, dest, output: int, input: int):
        if input != 0:
            raise WaaError(1, f"IndexSizeError - input port {input} is out of bounds")
        self.disconnect(dest, output, input)

    def _desc(self) -> NodeDesc:
        d = NodeDesc()
        d.kind = self.kind
        if self._explicit_config:
            d.channel_count = self.channel_count
            d.channel_count_mode = COUNT_MODE[self.channel_count_mode]
            d.channel_interpretation = INTERPRETATION[self.channel_interpretation]
        self._fill_desc(d)
        return d

    def _fill_desc(self, d: NodeDesc):
        pass

    def _apply(self, ctx: "OfflineAudioContext"):
        for p in self.params:
            p._apply(ctx)


class AudioDestinationNode(AudioNode):
    kind = NODE_DESTINATION

    def __init__(self, ctx):
        # destination.rs:103-107: count = context channels, Explicit, Speakers
        self.default_channel_config = (ctx.number_of_channels, "explicit", "speakers")
        super().__init__(ctx)


class _ScheduledSource(AudioNode):
    def __init__(self, ctx, **kw):
        super().__init__(ctx, **kw)
        self._starts = {}
        self._stops = {}
        self._late = set()

    def start(self):
        return self.start_at(0.0)

    def start_at(self, when: float, instance: int = ALL):
        return self.start_at_with_offset_and_duration(when, 0.0, F64_MAX, instance)

    def start_at_with_offset(self, when: float, offset: float, instance: int = ALL):
        return self.start_at_with_offset_and_duration(when, offset, F64_MAX, instance)

    def start_at_with_offset_and_duration(self, when: float, offset: float, duration: float, instance: int = ALL):
        if instance in self._starts or ALL in self._starts:
            raise WaaError(3, "InvalidStateError - Cannot call `start` twice")
        self._starts[instance] = (float(when), float(offset), float(duration))
        ctx = self.context
        if ctx._now_q > 0:  # inside a suspend_sync callback: the start message is handled in front of that quantum
            nid, args = self.id, self._starts[instance]
            self._late.add(("start", instance))
            ctx._log(lambda b, h: b.check(b.source_start(h, nid, instance, *args)))
        return self

    def stop(self):
        return self.stop_at(0.0)

    def stop_at(self, when: float, instance: int = ALL):
        if instance not in self._starts and ALL not in self._starts:
            raise WaaError(3, "InvalidStateError - cannot stop before start")
        self._stops[instance] = float(when)
        ctx = self.context
        if ctx._now_q > 0:
            nid, w = self.id, float(when)
            self._late.add(("stop", instance))
            ctx._log(lambda b, h: b.check(b.source_stop(h, nid, instance, w)))
        return self

    def ended_quantum(self, instance: int = 0) -> int:
        """When the reference fires `onended` (scheduled_source.rs:44) for this source of `instance`: the index of
        the render quantum after which the event is dispatched, ENDED_AT_UNLOAD (after the last quantum, from
        before_drop) or ENDED_NEVER.  Call after the render."""
        ctx = self.context
        q = C.c_int64()
        ctx._b.check(ctx._b.source_ended(ctx._handle, self.id, instance, C.byref(q)))
        return q.value

    def _apply(self, ctx):
        super()._apply(ctx)
        b, h = ctx._b, ctx._handle
        for inst, (w, o, d) in self._starts.items():
            if ("start", inst) not in self._late:  # (calls made at a suspend point are replayed from the control log)
                b.check(b.source_start(h, self.id, inst, w, o, d))
        for inst, w in self._stops.items():
            if ("stop", inst) not in self._late:
                b.check(b.source_stop(h, self.id, inst, w))


class AudioBufferSourceNode(_ScheduledSource):
    kind = NODE_BUFFER_SOURCE

    def __init__(self, ctx, **kw):
        super().__init__(ctx, **kw)
        self.playback_rate = AudioParam(self, 0, 1.0)
        self.detune = AudioParam(self, 1, 0.0)
        self.params = [self.playback_rate, self.detune]
        self._buffers = {}
        self._batch = None
        self._pcm = None
        self._pcm_one = {}
        self._device = None
        self._loop = {}

    def set_buffer(self, buffer: AudioBuffer, instance: int = ALL):
        self._buffers[instance] = buffer
        return self

    def set_buffer_batch(self, data, sample_rate: float):
        """data: [n_instances, channels, frames], distinct per instance."""
        self._batch = (_f32(data), float(np.float32(sample_rate)))
        return self

    def set_buffer_pcm16_batch(self, pcm, sample_rate: float):
        """decode_audio_data_sync for decoded 16-bit PCM: pcm = [n_instances, frames, channels] int16 (interleaved); the
        sample conversion and the resampling to the context's rate run in the library (on the device)."""
        self._pcm = (np.ascontiguousarray(pcm, dtype=np.int16), float(np.float32(sample_rate)))
        return self

    def set_buffer_pcm16(self, pcm, sample_rate: float, instance: int = ALL):
        """pcm = [frames, channels] int16 for one instance (or all)."""
        self._pcm_one[instance] = (np.ascontiguousarray(pcm, dtype=np.int16), float(np.float32(sample_rate)))
        return self

    def adopt_device_buffer(self, device_ptr: int, n_channels: int, frames: int, sample_rate: float):
        """device_ptr: address of a resident [n_instances, channels, frames] f32 device array."""
        self._device = (int(device_ptr), int(n_channels), int(frames), float(np.float32(sample_rate)))
        return self

    def set_loop(self, value: bool, instance: int = ALL):
        self._loop.setdefault(instance, [False, 0.0, 0.0])[0] = bool(value)
        return self

    def set_loop_start(self, value: float, instance: int = ALL):
        self._loop.setdefault(instance, [False, 0.0, 0.0])[1] = float(value)
        return self

    def set_loop_end(self, value: float, instance: int = ALL):
        self._loop.setdefault(instance, [False, 0.0, 0.0])[2] = float(value)
        return self

    def _apply(self, ctx):
        b, h = ctx._b, ctx._handle
        if self._device is not None:
            p, nch, frames, sr = self._device
            b.check(b.source_adopt_device(h, self.id, p, nch, frames, sr))
        if self._batch is not None:
            data, sr = self._batch
            assert data.ndim == 3 and data.shape[0] == ctx.n_instances
            b.check(b.source_set_buffer_batch(h, self.id, _fp(data), data.shape[1], data.shape[2], sr))
        _I16 = C.POINTER(C.c_int16)
        if self._pcm is not None:
            pcm, sr = self._pcm
            assert pcm.ndim == 3 and pcm.shape[0] == ctx.n_instances
            b.check(b.source_set_buffer_pcm16_batch(h, self.id, pcm.ctypes.data_as(_I16), pcm.shape[2], pcm.shape[1], sr))
        for inst, (pcm, sr) in sorted(self._pcm_one.items(), key=lambda kv: kv[0] != ALL):
            b.check(b.source_set_buffer_pcm16(h, self.id, inst, pcm.ctypes.data_as(_I16), pcm.shape[1], pcm.shape[0], sr))
        for inst, buf in sorted(self._buffers.items(), key=lambda kv: kv[0] != ALL):
            b.check(b.source_set_buffer(h, self.id, inst, _chan_ptrs(buf.data), buf.number_of_channels, buf.length,
                                        buf.sample_rate))
        for inst, (lp, ls, le) in sorted(self._loop.items(), key=lambda kv: kv[0] != ALL):
            b.check(b.source_set_loop(h, self.id, inst, int(lp), ls, le))
        super()._apply(ctx)


class ConstantSourceNode(_ScheduledSource):
    kind = NODE_CONSTANT_SOURCE

    def __init__(self, ctx, offset: float = 1.0, **kw):
        super().__init__(ctx, **kw)
        self.offset = AudioParam(self, 0, offset)
        self.params = [self.offset]


class PeriodicWave:
    """src/periodic_wave.rs:35-190: PeriodicWaveOptions{real, imag, disable_normalization}; the 8192-point table is
    generated by the library when the wave is set on an oscillator."""

    def __init__(self, ctx=None, real=None, imag=None, disable_normalization: bool = False):
        if real is None and imag is None:  # defaults to a sine wave (periodic_wave.rs:140-143)
            real, imag = [0.0, 0.0], [0.0, 1.0]
        r = None if real is None else _f32(real).reshape(-1)
        i = None if imag is None else _f32(imag).reshape(-1)
        if r is not None and i is not None and r.size != i.size:
            raise WaaError(1, "IndexSizeError - `real` and `imag` length should be equal")
        n = (r if r is not None else i).size
        if n < 2:
            raise WaaError(1, "IndexSizeError - `real` and `imag` length should at least 2")
        self.real = r if r is not None else np.zeros(n, np.float32)
        self.imag = i if i is not None else np.zeros(n, np.float32)
        self.disable_normalization = bool(disable_normalization)
        self.table = None

    @classmethod
    def from_wavetable(cls, table):
        """A PeriodicWave that already IS its 8192-point table — what the reference's render side holds (periodic_wave.rs:72-74,
        oscillator.rs:487-493) and what the Rust shim forwards (waa_oscillator_set_wavetable)."""
        t = _f32(table).reshape(-1)
        if t.size != 8192:
            raise WaaError(1, f"IndexSizeError - a PeriodicWave table has 8192 points (got {t.size})")
        w = cls()
        w.table = t
        return w


class OscillatorNode(_ScheduledSource):
    """src/node/oscillator.rs:64-321 (OscillatorOptions{type = sine, frequency = 440, detune = 0, periodic_wave})"""

    kind = NODE_OSCILLATOR

    def __init__(self, ctx, type_="sine", frequency=440.0, detune=0.0, periodic_wave=None, **kw):
        super().__init__(ctx, **kw)
        if type_ == "custom" and periodic_wave is None:
            raise WaaError(3, "InvalidStateError: Custom type cannot be set manually")
        self.type_ = type_
        self.frequency = AudioParam(self, 0, frequency)
        self.detune = AudioParam(self, 1, detune)
        self.params = [self.frequency, self.detune]
        self.periodic_wave = None
        if periodic_wave is not None:
            self.set_periodic_wave(periodic_wave)

    def set_type(self, type_: str):
        if type_ == "custom":  # oscillator.rs:305-309
            raise WaaError(3, "InvalidStateError: Custom type cannot be set manually")
        if self.type_ == "custom":  # ignored once a periodic wave is set (oscillator.rs:311-314)
            return
        self.type_ = type_

    def set_periodic_wave(self, wave: PeriodicWave):
        self.type_ = "custom"
        self.periodic_wave = wave

    def _fill_desc(self, d):
        d.i[0] = OSCILLATOR_TYPE[self.type_]

    def _apply(self, ctx):
        super()._apply(ctx)
        if self.periodic_wave is not None:
            w = self.periodic_wave
            b, h = ctx._b, ctx._handle
            if w.table is not None:
                b.check(b.oscillator_set_wavetable(h, self.id, _fp(w.table), w.table.size))
            else:
                b.check(b.oscillator_set_periodic_wave(h, self.id, _fp(w.real), _fp(w.imag), w.real.size,
                                                       int(w.disable_normalization)))


class BiquadFilterNode(AudioNode):
    kind = NODE_BIQUAD

    def __init__(self, ctx, type_="lowpass", frequency=350.0, detune=0.0, q=1.0, gain=0.0, **kw):
        super().__init__(ctx, **kw)
        self.type_ = type_
        self.frequency = AudioParam(self, 0, frequency)
        self.detune = AudioParam(self, 1, detune)
        self.q = AudioParam(self, 2, q)
        self.gain = AudioParam(self, 3, gain)
        self.params = [self.frequency, self.detune, self.q, self.gain]

    def set_type(self, type_: str):
        self.type_ = type_
        return self

    def _fill_desc(self, d):
        d.i[0] = BIQUAD_TYPE[self.type_]

    def get_frequency_response(self, frequency_hz) -> tuple:
        hz = _f32(frequency_hz)
        mag = np.empty_like(hz)
        phase = np.empty_like(hz)
        b = self.context._b
        b.check(b.biquad_frequency_response(BIQUAD_TYPE[self.type_], self.context.sample_rate, self.frequency.value,
                                            self.detune.value, self.q.value, self.gain.value, _fp(hz), _fp(mag),
                                            _fp(phase), hz.size))
        return mag, phase


class GainNode(AudioNode):
    kind = NODE_GAIN

    def __init__(self, ctx, gain: float = 1.0, **kw):
        super().__init__(ctx, **kw)
        self.gain = AudioParam(self, 0, gain)
        self.params = [self.gain]


class ConvolverNode(AudioNode):
    kind = NODE_CONVOLVER
    default_channel_config = (2, "clamped-max", "speakers")

    def __init__(self, ctx, buffer: Optional[AudioBuffer] = None, disable_normalization: bool = False, **kw):
        super().__init__(ctx, **kw)
        self.normalize = not disable_normalization
        self.buffer = None
        self._normalize_at_set = self.normalize
        if buffer is not None:
            self.set_buffer(buffer)

    def set_normalize(self, value: bool):
        self.normalize = bool(value)

    def set_buffer(self, buffer: AudioBuffer):
        if buffer.sample_rate != self.context.sample_rate:
            raise WaaError(2, "NotSupportedError - sample rate of the convolution buffer must match the audio context")
        if buffer.number_of_channels not in (1, 2, 4):
            raise WaaError(2, "NotSupportedError - the convolution buffer must consist of 1, 2 or 4 channels")
        self.buffer = buffer
        self._normalize_at_set = self.normalize
        return self

    def set_buffer_pcm16(self, pcm, sample_rate: float):
        """decode_audio_data_sync + set_buffer: pcm = [frames, channels] int16 at `sample_rate`; decoded and resampled to
        the context's rate by the library (on the device)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        if pcm.shape[1] not in (1, 2, 4):
            raise WaaError(2, "NotSupportedError - the convolution buffer must consist of 1, 2 or 4 channels")
        self._pcm = (pcm, float(np.float32(sample_rate)))
        self._normalize_at_set = self.normalize
        return self

    def _fill_desc(self, d):
        d.i[0] = 0 if self._normalize_at_set else 1

    def _apply(self, ctx):
        if getattr(self, "_pcm", None) is not None:
            pcm, sr = self._pcm
            ctx._b.check(ctx._b.convolver_set_buffer_pcm16(ctx._handle, self.id, pcm.ctypes.data_as(C.POINTER(C.c_int16)),
                                                           pcm.shape[1], pcm.shape[0], sr))
        if self.buffer is not None:
            b, h = ctx._b, ctx._handle
            buf = self.buffer
            b.check(b.convolver_set_buffer(h, self.id, _chan_ptrs(buf.data), buf.number_of_channels, buf.length,
                                           buf.sample_rate))


class StereoPannerNode(AudioNode):
    kind = NODE_STEREO_PANNER
    default_channel_config = (2, "clamped-max", "speakers")

    def __init__(self, ctx, pan: float = 0.0, **kw):
        super().__init__(ctx, **kw)
        self.pan = AudioParam(self, 0, pan)
        self.params = [self.pan]


class AudioListener:
    """src/spatial.rs:127-144 — shared by every PannerNode of the context."""

    def __init__(self):
        # nine a-rate AudioParams (spatial.rs:18-24 defaults); they cross the ABI through every PannerNode (params 6..14)
        names = ["position_x", "position_y", "position_z", "forward_x", "forward_y", "forward_z", "up_x", "up_y", "up_z"]
        self.params = [AudioParam(None, 6 + i, v) for i, v in enumerate([0.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0, 1.0, 0.0])]
        for name, prm in zip(names, self.params):
            setattr(self, name, prm)

    @property
    def values(self):
        return [p.value for p in self.params]

    def set_position(self, x, y, z):
        for p, v in zip(self.params[0:3], (x, y, z)):
            p.set_value(v)

    def set_orientation(self, fx, fy, fz, ux, uy, uz):
        for p, v in zip(self.params[3:9], (fx, fy, fz, ux, uy, uz)):
            p.set_value(v)


class PannerNode(AudioNode):
    kind = NODE_PANNER
    default_channel_config = (2, "clamped-max", "speakers")

    def __init__(self, ctx, panning_model="equalpower", distance_model="inverse", position=(0.0, 0.0, 0.0),
                 orientation=(1.0, 0.0, 0.0), ref_distance=1.0, max_distance=10000.0, rolloff_factor=1.0,
                 cone_inner_angle=360.0, cone_outer_angle=360.0, cone_outer_gain=0.0, **kw):
        super().__init__(ctx, **kw)
        self.panning_model, self.distance_model = panning_model, distance_model
        self.ref_distance, self.max_distance, self.rolloff_factor = ref_distance, max_distance, rolloff_factor
        self.cone_inner_angle, self.cone_outer_angle, self.cone_outer_gain = (cone_inner_angle, cone_outer_angle,
                                                                              cone_outer_gain)
        self.position_x = AudioParam(self, 0, position[0])
        self.position_y = AudioParam(self, 1, position[1])
        self.position_z = AudioParam(self, 2, position[2])
        self.orientation_x = AudioParam(self, 3, orientation[0])
        self.orientation_y = AudioParam(self, 4, orientation[1])
        self.orientation_z = AudioParam(self, 5, orientation[2])
        self.params = [self.position_x, self.position_y, self.position_z, self.orientation_x, self.orientation_y,
                       self.orientation_z]

    def set_position(self, x, y, z):
        self.position_x.set_value(x)
        self.position_y.set_value(y)
        self.position_z.set_value(z)

    def set_orientation(self, x, y, z):
        self.orientation_x.set_value(x)
        self.orientation_y.set_value(y)
        self.orientation_z.set_value(z)

    def _fill_desc(self, d):
        d.i[0] = PANNING_MODEL[self.panning_model]
        d.i[1] = DISTANCE_MODEL[self.distance_model]
        d.d[0], d.d[1], d.d[2] = self.ref_distance, self.max_distance, self.rolloff_factor
        d.d[3], d.d[4], d.d[5] = self.cone_inner_angle, self.cone_outer_angle, self.cone_outer_gain

    def _apply(self, ctx):
        super()._apply(ctx)
        for i, prm in enumerate(ctx.listener().params):  # the AudioListener's params, addressed through this panner
            prm._apply(ctx, node_id=self.id, pid=6 + i)


class AnalyserNode(AudioNode):
    kind = NODE_ANALYSER

    def __init__(self, ctx, fft_size=2048, smoothing_time_constant=0.8, min_decibels=-100.0, max_decibels=-30.0,
                 **kw):
        super().__init__(ctx, **kw)
        self.fft_size, self.smoothing_time_constant = fft_size, smoothing_time_constant
        self.min_decibels, self.max_decibels = min_decibels, max_decibels

    @property
    def frequency_bin_count(self) -> int:
        return self.fft_size // 2

    def _fill_desc(self, d):
        d.i[0] = self.fft_size
        d.d[0], d.d[1], d.d[2] = self.smoothing_time_constant, self.min_decibels, self.max_decibels

    def _pull(self, fn_name: str, n: int, instance: int, dtype):
        ctx = self.context
        if ctx._now_q > 0 and ctx._handle is None:  # inside a suspend_sync callback: what has been rendered up to here
            with ctx._prefix_render(ctx._now_q):
                return self._pull(fn_name, n, instance, dtype)
        if ctx._handle is None:
            raise WaaError(3, "InvalidStateError - analyser data is only available after start_rendering_sync")
        out = np.zeros(n, dtype=dtype)
        ptr = out.ctypes.data_as(_FP if dtype == np.float32 else C.POINTER(C.c_uint8))
        ctx._b.check(getattr(ctx._b, fn_name)(ctx._handle, self.id, instance, ptr, n))
        return out

    def _pull_all(self, fn_name: str, n: int, dtype, out=None):
        """[n_instances][n]: every context's pull in one call (one launch, one transfer on the device)."""
        ctx = self.context
        if ctx._now_q > 0 and ctx._handle is None:
            with ctx._prefix_render(ctx._now_q):
                return self._pull_all(fn_name, n, dtype, out)
        if ctx._handle is None:
            raise WaaError(3, "InvalidStateError - analyser data is only available after start_rendering_sync")
        if out is None:
            out = np.zeros((ctx.n_instances, n), dtype=dtype)
        if not isinstance(out, np.ndarray) or out.shape != (ctx.n_instances, n) or out.dtype != dtype or not out.flags.c_contiguous:
            # (the library writes n values per context through the raw pointer: a wrong buffer is a silent out-of-bounds write)
            raise ValueError(f"out: expected a C-contiguous {np.dtype(dtype).name} array of shape {(ctx.n_instances, n)}")
        ptr = out.ctypes.data_as(_FP if dtype == np.float32 else C.POINTER(C.c_uint8))
        ctx._b.check(getattr(ctx._b, fn_name)(ctx._handle, self.id, ptr, n))
        return out

    def get_float_frequency_data_all(self, n: Optional[int] = None, out=None) -> np.ndarray:
        return self._pull_all("analyser_get_float_frequency_data_batch", n or self.frequency_bin_count, np.float32, out)

    def get_byte_frequency_data_all(self, n: Optional[int] = None, out=None) -> np.ndarray:
        return self._pull_all("analyser_get_byte_frequency_data_batch", n or self.frequency_bin_count, np.uint8, out)

    def get_float_time_domain_data_all(self, n: Optional[int] = None, out=None) -> np.ndarray:
        return self._pull_all("analyser_get_float_time_domain_data_batch", n or self.fft_size, np.float32, out)

    def get_byte_time_domain_data_all(self, n: Optional[int] = None, out=None) -> np.ndarray:
        return self._pull_all("analyser_get_byte_time_domain_data_batch", n or self.fft_size, np.uint8, out)

    def get_float_frequency_data(self, n: Optional[int] = None, instance: int = 0) -> np.ndarray:
        return self._pull("analyser_get_float_frequency_data", n or self.frequency_bin_count, instance, np.float32)

    def get_byte_frequency_data(self, n: Optional[int] = None, instance: int = 0) -> np.ndarray:
        return self._pull("analyser_get_byte_frequency_data", n or self.frequency_bin_count, instance, np.uint8)

    def get_float_time_domain_data(self, n: Optional[int] = None, instance: int = 0) -> np.ndarray:
        return self._pull("analyser_get_float_time_domain_data", n or self.fft_size, instance, np.float32)

    def get_byte_time_domain_data(self, n: Optional[int] = None, instance: int = 0) -> np.ndarray:
        return self._pull("analyser_get_byte_time_domain_data", n or self.fft_size, instance, np.uint8)


class WaveShaperNode(AudioNode):
    kind = NODE_WAVESHAPER

    def __init__(self, ctx, curve=None, oversample="none", **kw):
        super().__init__(ctx, **kw)
        self.oversample = oversample
        self.curve = None
        if curve is not None:
            self.set_curve(curve)

    def set_curve(self, curve):
        if self.curve is not None:
            raise WaaError(3, "InvalidStateError - cannot assign curve twice")
        self.curve = _f32(curve)
        return self

    def set_oversample(self, oversample: str):
        self.oversample = oversample

    def _fill_desc(self, d):
        d.i[0] = OVERSAMPLE[self.oversample]

    def _apply(self, ctx):
        if self.curve is not None:
            b, h = ctx._b, ctx._handle
            b.check(b.waveshaper_set_curve(h, self.id, _fp(self.curve), self.curve.size))


def _dp(a: np.ndarray):
    return a.ctypes.data_as(_DP)


class IIRFilterNode(AudioNode):
    """src/node/iir_filter.rs:140-263.  Coefficients are fixed at construction (IIRFilterOptions)."""

    kind = NODE_IIR_FILTER

    def __init__(self, ctx, feedforward, feedback, **kw):
        ff = np.ascontiguousarray(feedforward, dtype=np.float64).reshape(-1)
        fb = np.ascontiguousarray(feedback, dtype=np.float64).reshape(-1)
        # iir_filter.rs:17-46
        if not 1 <= ff.size <= MAX_IIR_COEFFS:
            raise WaaError(2, "NotSupportedError - IIR Filter feedforward coefficients should have length >= 0 and <= 20")
        if not np.any(ff != 0.0):
            raise WaaError(3, "InvalidStateError - IIR Filter feedforward coefficients cannot be all zeros")
        if not 1 <= fb.size <= MAX_IIR_COEFFS:
            raise WaaError(2, "NotSupportedError - IIR Filter feedback coefficients should have length >= 0 and <= 20")
        if fb[0] == 0.0:
            raise WaaError(3, "InvalidStateError - IIR Filter feedback first coefficient cannot be zero")
        super().__init__(ctx, **kw)
        self.feedforward, self.feedback = ff, fb

    def _apply(self, ctx):
        b, h = ctx._b, ctx._handle
        b.check(b.iir_set_coefficients(h, self.id, _dp(self.feedforward), self.feedforward.size, _dp(self.feedback),
                                       self.feedback.size))

    def get_frequency_response(self, frequency_hz) -> tuple:
        hz = _f32(frequency_hz)
        mag = np.empty_like(hz)
        phase = np.empty_like(hz)
        b = self.context._b
        b.check(b.iir_frequency_response(_dp(self.feedforward), self.feedforward.size, _dp(self.feedback),
                                         self.feedback.size, self.context.sample_rate, _fp(hz), _fp(mag), _fp(phase),
                                         hz.size))
        return mag, phase


class DelayNode(AudioNode):
    """src/node/delay.rs:127-376 (DelayOptions{max_delay_time = 1, delay_time = 0}).  Inside a graph cycle the
    node acts as the reference's cycle breaker (delay clamped to one render quantum)."""

    kind = NODE_DELAY

    def __init__(self, ctx, max_delay_time: float = 1.0, delay_time: float = 0.0, **kw):
        if not (0.0 < max_delay_time < 180.0):  # delay.rs:290-293
            raise WaaError(2, "NotSupportedError - maxDelayTime MUST be greater than zero and less than three minutes")
        super().__init__(ctx, **kw)
        self.max_delay_time = float(max_delay_time)
        self.delay_time = AudioParam(self, 0, delay_time)
        self.params = [self.delay_time]

    def _fill_desc(self, d):
        d.d[0] = self.max_delay_time


class RenderedBatch:
    """What start_rendering_sync returns: one AudioBuffer per instance (array [inst, ch, frames])."""

    def __init__(self, data: np.ndarray, sample_rate: float):
        self.data = data
        self.sample_rate = sample_rate

    def buffer(self, instance: int = 0) -> AudioBuffer:
        return AudioBuffer(self.data[instance], self.sample_rate)

    def get_channel_data(self, c: int, instance: int = 0) -> np.ndarray:
        return self.data[instance, c]

    @property
    def length(self):
        return self.data.shape[2]

    @property
    def number_of_channels(self):
        return self.data.shape[1]


class OfflineAudioContext:
    """A batch of ``n_instances`` identically shaped OfflineAudioContexts (offline.rs:68)."""

    def __init__(self, number_of_channels: int, length: int, sample_rate: float, n_instances: int = 1,
                 binding: Optional[Binding] = None, device: int = -1):
        if binding is None:
            from . import default_binding  # loads libwaa_hip.so; raises if it is not built
            binding = default_binding()
        self._b = binding
        self.number_of_channels = int(number_of_channels)
        self.length = int(length)
        self.sample_rate = float(np.float32(sample_rate))
        self.n_instances = int(n_instances)
        self.device = device
        self._nodes: List[AudioNode] = []
        self._edges = []
        self._handle = None
        self._rendered = False
        self._foreign = False
        # OfflineAudioContext::suspend_sync (offline.rs:359-397): callbacks by render quantum; while one runs, _now_q is its quantum
        # and every control call (connect / disconnect, AudioParam methods, start / stop) goes to _ctl[q] instead of the node's
        # static description — replayed between the ranges of waa_render_range, where the reference's render thread handles them
        self._suspends = {}
        self._ctl = {}
        self._now_q = 0
        self._replayed = 0  # the control log has been replayed up to this quantum on the current batch
        self._live = []   # the connections that exist "now" (disconnect's InvalidAccessError, disconnect() of everything)
        self._state = "suspended"  # AudioContextState of a context that has not started rendering (offline.rs:451)
        self._listener = AudioListener()
        for prm in self._listener.params:
            prm._listener_ctx = self
        self._destination = AudioDestinationNode(self)

    def state(self) -> str:
        return self._state

    def _log(self, action):
        self._ctl.setdefault(self._now_q, []).append(action)

    def _connect(self, edge):
        if edge in self._live:
            return  # (connecting twice is a no-op: the reference keeps a set of connections)
        self._live.append(edge)
        if self._now_q > 0:
            self._log(lambda b, h: b.check(b.connect(h, *edge)))
        else:
            self._edges.append(edge)

    def _disconnect(self, frm, output, to, to_input):
        hit = [e for e in self._live if e[0] == frm and (output is None or e[1] == output) and (to is None or e[2] == to) and
               (to_input is None or e[3] == to_input)]
        if to is not None and not hit:
            raise WaaError(1, "InvalidAccessError - attempting to disconnect unconnected nodes")
        for e in hit:
            self._live.remove(e)
            if self._now_q > 0:
                self._log(lambda b, h, e=e: b.check(b.disconnect(h, *e)))
            else:
                self._edges.remove(e)

    def suspend_sync(self, suspend_time: float, callback):
        """OfflineAudioContext::suspend_sync (offline.rs:359-397): run `callback(context)` when the render reaches `suspend_time`
        (quantised UP to a render quantum, calculate_suspend_frame offline.rs:241-251); the render resumes when it returns."""
        if self._rendered:
            raise WaaError(3, "InvalidStateError - cannot suspend when rendering has already started")
        if not suspend_time >= 0.0:
            raise WaaError(3, "InvalidStateError: suspendTime cannot be negative")
        if not suspend_time < self.length / float(self.sample_rate):
            raise WaaError(3, "InvalidStateError: suspendTime cannot be greater than or equal to the total render duration")
        q = int(np.ceil(suspend_time * float(self.sample_rate) / RENDER_QUANTUM_SIZE))
        if q in self._suspends:
            raise WaaError(3, "InvalidStateError - cannot suspend multiple times at the same render quantum")
        self._suspends[q] = callback

    def _run_suspend_callbacks(self):
        """The callbacks in render order, each with the control clock at its quantum (thread.rs:281-287)."""
        for q in sorted(self._suspends):
            cb = self._suspends.pop(q)
            self._now_q = q if q > 0 else 0  # (a suspend at time 0 runs in front of the first quantum: plain graph edits)
            self._state = "suspended"
            try:
                cb(self)
            finally:
                self._state = "running"
                self._now_q = 0

    def _register(self, node: AudioNode) -> int:
        if self._handle is not None:
            raise WaaError(3, "InvalidStateError - graph is frozen once rendering has started")
        self._nodes.append(node)
        return len(self._nodes) - 1

    # BaseAudioContext
    def destination(self) -> AudioDestinationNode:
        return self._destination

    def listener(self) -> AudioListener:
        return self._listener

    def create_buffer(self, number_of_channels: int, length: int, sample_rate: float) -> AudioBuffer:
        return AudioBuffer(np.zeros((number_of_channels, length), np.float32), sample_rate)

    def create_buffer_source(self, **kw):
        return AudioBufferSourceNode(self, **kw)

    def create_constant_source(self, **kw):
        return ConstantSourceNode(self, **kw)

    def create_biquad_filter(self, **kw):
        return BiquadFilterNode(self, **kw)

    def create_gain(self, **kw):
        return GainNode(self, **kw)

    def create_convolver(self, **kw):
        return ConvolverNode(self, **kw)

    def create_stereo_panner(self, **kw):
        return StereoPannerNode(self, **kw)

    def create_panner(self, **kw):
        return PannerNode(self, **kw)

    def create_analyser(self, **kw):
        return AnalyserNode(self, **kw)

    def create_wave_shaper(self, **kw):
        return WaveShaperNode(self, **kw)

    def create_oscillator(self, **kw):
        return OscillatorNode(self, **kw)

    def create_periodic_wave(self, **kw):
        return PeriodicWave(self, **kw)

    def create_delay(self, max_delay_time: float = 1.0, **kw):
        return DelayNode(self, max_delay_time=max_delay_time, **kw)

    def create_iir_filter(self, feedforward, feedback, **kw):
        return IIRFilterNode(self, feedforward, feedback, **kw)

    # -- render -------------------------------------------------------------------------
    def graph_desc(self) -> "GraphDesc":
        """The waa_graph_desc of this context's graph (the arrays it points at stay alive with the returned object)."""
        n = len(self._nodes)
        nodes = (NodeDesc * n)(*[nd._desc() for nd in self._nodes])
        m = len(self._edges)
        edges = (EdgeDesc * max(m, 1))()
        for k, (f, fo, t, ti) in enumerate(self._edges):
            edges[k].from_, edges[k].from_output, edges[k].to, edges[k].to_input = f, fo, t, ti
        g = GraphDesc(n, nodes, m, edges)
        g._keep = (nodes, edges)
        if any(getattr(nd, "panning_model", None) == "HRTF" for nd in self._nodes):
            ensure_hrtf_database(self._b)
        return g

    def _prefix_render(self, q):
        """What a suspend callback sees when it READS rendered audio (an analyser pull at quantum q): the engine renders node-major,
        so the quanta in front of the suspend point are rendered by a batch of their own — the graph as it is so far, `q` quanta
        long, the control log of the earlier suspend points replayed (a render is causal: its first q quanta do not depend on what
        happens later).  The batch lives for the duration of the `with` block."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            now, replayed, rendered = self._now_q, self._replayed, self._rendered
            self._now_q = 0
            try:
                self._build(length=q * RENDER_QUANTUM_SIZE, run_callbacks=False)
                self._replayed = 0
                self._render_ranges(q)
                yield
            finally:
                if self._handle is not None:
                    self._b.batch_destroy(self._handle)
                self._handle = None
                self._now_q, self._replayed, self._rendered = now, replayed, rendered
        return scope()

    def _build(self, length=None, run_callbacks=True):
        if run_callbacks:
            self._run_suspend_callbacks()  # (every node a callback creates must be in the graph description)
        g = self.graph_desc()
        h = _VP()
        self._b.check(self._b.batch_create(C.byref(g), self.n_instances, self.number_of_channels, self.length if length is None else length,
                                           self.sample_rate, self.device, C.byref(h)))
        self._handle = h
        for nd in self._nodes:
            nd._apply(self)

    def _render_ranges(self, n_quanta, final=True):
        """waa_render_range between the suspend points, the control log of each point replayed in front of its quantum; the last
        range renders (final=False: everything but that last range — what a plan description needs).  Without suspend points:
        waa_render."""
        points = [q for q in sorted(self._ctl) if 0 < q < n_quanta]
        if self._replayed:
            points = []
        if not points:
            if final:
                self._b.check(self._b.render(self._handle) if not self._replayed else
                              self._b.render_range(self._handle, self._replayed, n_quanta - self._replayed))
            return
        prev = 0
        for q in points:
            self._b.check(self._b.render_range(self._handle, prev, q - prev))
            for action in self._ctl[q]:
                action(self._b, self._handle)
            prev = q
        self._replayed = prev
        if final:
            self._b.check(self._b.render_range(self._handle, prev, n_quanta - prev))

    def _adopt(self, handle):
        """Configure a batch the LIBRARY created for this graph (waa_render_sharded's setup callback): node payloads, params,
        schedules.  The batch stays the library's: _release() forgets it without destroying it."""
        self._handle = _VP(handle) if not isinstance(handle, _VP) else handle
        self._foreign = True
        try:
            for nd in self._nodes:
                nd._apply(self)
        except BaseException:
            self._handle = None  # a node refused its payload: the batch is still the library's to destroy, never ours
            raise

    def _release(self):
        self._handle = None
        self._foreign = False  # (the adopted batch is forgotten: whatever this context creates next is its own)

    def prepare(self):
        """Create the batch and upload every payload (outside any timed region)."""
        if self._handle is None:
            self._build()
        return self

    def render_async(self):
        """Launch the render on the batch's stream without waiting (bench inner loop)."""
        self.prepare()
        self._state = "running"
        if self._rendered or not self._ctl:
            self._b.check(self._b.render(self._handle))  # (a second render of the same batch: the bench loop)
        else:
            self._render_ranges((self.length + RENDER_QUANTUM_SIZE - 1) // RENDER_QUANTUM_SIZE)
        self._rendered = True

    def sync(self):
        self._b.check(self._b.sync(self._handle))

    def start_rendering_sync(self) -> RenderedBatch:
        if self._rendered:
            raise WaaError(3, "InvalidStateError - Cannot call `startRendering` twice")
        self.render_async()
        out = np.empty((self.n_instances, self.number_of_channels, self.length), np.float32)
        self._b.check(self._b.download_all(self._handle, _fp(out)))
        self._state = "closed"  # offline.rs:176
        return RenderedBatch(out, self.sample_rate)

    def render_instances(self, instances) -> np.ndarray:
        """start_rendering_sync for the whole batch, but only the AudioBuffers of `instances` are downloaded
        (waa_download per channel): full-size batches whose result would not fit a test's host memory budget."""
        if self._rendered:
            raise WaaError(3, "InvalidStateError - Cannot call `startRendering` twice")
        self.render_async()
        out = np.empty((len(instances), self.number_of_channels, self.length), np.float32)
        for k, inst in enumerate(instances):
            for c in range(self.number_of_channels):
                self._b.check(self._b.download(self._handle, int(inst), c, _fp(out[k, c]), self.length))
        return out

    def plan_describe(self) -> str:
        """The launch plan derived from the graph (works on a plan-only context: device=PLAN_ONLY)."""
        self.prepare()
        if self._ctl and not self._rendered:
            self._render_ranges((self.length + RENDER_QUANTUM_SIZE - 1) // RENDER_QUANTUM_SIZE, final=False)
        need = C.c_size_t()
        self._b.check(self._b.plan_describe(self._handle, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value + 1)
        self._b.check(self._b.plan_describe(self._handle, buf, need.value + 1, None))
        return buf.value.decode()

    def output_device(self):
        p, s_i, s_c = _VP(), C.c_uint64(), C.c_uint64()
        self._b.check(self._b.output_device(self._handle, C.byref(p), C.byref(s_i), C.byref(s_c)))
        return p.value, s_i.value, s_c.value

    def profile(self, enable=True):
        self.prepare()
        self._b.check(self._b.profile_enable(self._handle, int(enable)))

    def profile_entries(self):
        out = []
        for i in range(self._b.profile_count(self._handle)):
            name, launches, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            self._b.check(self._b.profile_get(self._handle, i, C.byref(name), C.byref(launches), C.byref(ms)))
            out.append((name.value.decode(), launches.value, ms.value))
        return out

    def profile_reset(self):
        self._b.check(self._b.profile_reset(self._handle))

    def close(self):
        if self._handle is not None:
            if not getattr(self, "_foreign", False):  # an adopted batch belongs to the library (waa_render_sharded destroys it)
                self._b.batch_destroy(self._handle)
            self._handle = None
            self._foreign = False  # (a context that is prepared on its own afterwards owns THAT batch)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resample(binding: Binding, data, source_sample_rate: float, target_sample_rate: float) -> np.ndarray:
    """AudioBuffer::resample (src/buffer.rs:311-363) for a [channels, frames] array."""
    data = _f32(data)
    if data.ndim == 1:
        data = data[None, :]
    n = binding.buffer_resample(_fp(data[0]), data.shape[1], source_sample_rate, target_sample_rate, None, 0)
    out = np.empty((data.shape[0], n), np.float32)
    for c in range(data.shape[0]):
        binding.buffer_resample(_fp(data[c]), data.shape[1], source_sample_rate, target_sample_rate, _fp(out[c]), n)
    return out


# ---- HRTF database (src/node/panner.rs:39-68) --------------------------------------------------------------
# The reference embeds resources/IRC_1003_C.bin in the crate (`include_bytes!`); a library behind the C ABI gets the
# same bytes through waa_hrtf_load_sphere, once per process.  `set_hrtf_database` names the file (or the bytes) this
# mirror hands over the first time a context with an HRTF PannerNode is built.
_HRTF_DATABASE = None
_HRTF_LOADED = set()


def set_hrtf_database(path_or_bytes):
    global _HRTF_DATABASE
    if isinstance(path_or_bytes, (bytes, bytearray)):
        _HRTF_DATABASE = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            _HRTF_DATABASE = f.read()
    _HRTF_LOADED.clear()


def ensure_hrtf_database(binding: Binding):
    if id(binding.lib) in _HRTF_LOADED:
        return
    if _HRTF_DATABASE is None:
        raise WaaError(3, "InvalidStateError - HRTF panning needs the HRIR sphere: call set_hrtf_database(path) first")
    binding.check(binding.hrtf_load_sphere(_HRTF_DATABASE, len(_HRTF_DATABASE)))
    _HRTF_LOADED.add(id(binding.lib))


def hrtf_sample(binding: Binding, sample_rate: float, direction) -> np.ndarray:
    """hrtf::HrirSphere::sample_bilinear (test hook): [2, taps] left / right HRIR for a direction (sphere coordinates)."""
    ensure_hrtf_database(binding)
    n = int(binding.hrtf_hrir_length(sample_rate))
    out = np.zeros((2, n), np.float32)
    d = _f32(direction)
    binding.hrtf_sample(sample_rate, _fp(d), _fp(out[0]), _fp(out[1]))
    return out
