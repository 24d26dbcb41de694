/* Sanitizer harness for the AudioParam timelines (test infrastructure).  Replays an event script
 *   T <a_rate> <min> <max> | E <type> <value> <time> <aux> <n_curve> [curve...] | C <block_time> <count>
 * against the oracle's C timeline or the product's C++ one and prints every status and block head, so the two
 * outputs can be diffed and both run under ASan/UBSan:
 *   gcc -O1 -g -fsanitize=address,undefined -D'PFX(x)=orc_##x' -Iinclude tools/timeline_harness.c oracle/waa_oracle.c -lm
 *   g++ -x c++ ... -D'PFX(x)=waa_##x' tools/timeline_harness.c web-audio-api-rs_amd/csrc/waa_automation.cpp <stub of waa::host::fail>
 * (tests/test_automation.py::test_random_schedules_agree_between_the_two_restatements is the in-suite version.) */
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef struct orc_timeline orc_timeline;
#ifdef __cplusplus
extern "C" {
#endif
orc_timeline* PFX(timeline_create)(float, float, float, int32_t);
int PFX(timeline_event)(orc_timeline*, int32_t, float, double, double, const float*, uint32_t);
uint32_t PFX(timeline_compute)(orc_timeline*, double, double, uint32_t, float*);
#ifdef __cplusplus
}
#endif
int main(int argc, char** argv) {
  (void)argc;
  FILE* f = fopen(argv[1], "r");
  char tag[4]; orc_timeline* t = NULL;
  while (fscanf(f, "%3s", tag) == 1) {
    if (tag[0] == 'T') { int a; float lo, hi; fscanf(f, "%d %f %f", &a, &lo, &hi); t = PFX(timeline_create)(0.25f, lo, hi, a); }
    else if (tag[0] == 'E') { int k, n; double v, tm, aux; fscanf(f, "%d %lf %lf %lf %d", &k, &v, &tm, &aux, &n);
      float* c = n ? (float*)malloc(n * sizeof(float)) : NULL; for (int i = 0; i < n; i++) fscanf(f, "%f", &c[i]);
      int r = PFX(timeline_event)(t, k, (float)v, tm, aux, c, n); printf("E %d %g %g -> %d\n", k, v, tm, r); fflush(stdout); free(c); }
    else { double bt; int n; fscanf(f, "%lf %d", &bt, &n); float* out = (float*)malloc(n * sizeof(float));
      uint32_t m = PFX(timeline_compute)(t, bt, 1.0, n, out); printf("C %g -> %u: %g\n", bt, m, out[0]); fflush(stdout); free(out); }
  }
  return 0;
}
