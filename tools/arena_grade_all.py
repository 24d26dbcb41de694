"""tools/arena_grade_all.py [GiB] — grade (nearly) all of the device's memory with waa_device_arena_reserve_graded and print the
candidates' grades in creation order (WAA_ARENA_TRACE) and the sorted kept ones.  (GPU box)"""
import os
import sys
os.environ["WAA_ARENA_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import web_audio_api_rs_amd as waa  # noqa: E402

hip = waa.default_binding()
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 272
hip.check(hip.device_arena_reserve_graded(0, gib << 30, gib << 30))
g = waa.arena_grades(hip, 0)
print({k: v for k, v in g.items() if k != "unit_ms"})
fast = [v for v in g["unit_ms"] if v < 1.42]
print(f"{len(fast)} of {g['n_units']} units below 1.42 ms:", " ".join("%.3f" % v for v in fast))
hip.check(hip.device_arena_reserve(0, 0))
