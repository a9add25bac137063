// Opt-in per-launch timing of the conv kernels (sg_prof_* in include/saragan_hip.h).
#pragma once
#include "common.h"

void sg_prof_begin(int kind, const sg_conv_shape* s, sg_dtype dt, hipStream_t st, int* slot);
void sg_prof_end(int slot, hipStream_t st);
void sg_prof_name(int slot, const char* name);   // the kernel the dispatcher chose (static string)
bool sg_prof_on();
extern thread_local const char* sg_tls_kernel;   // name of the kernel the last launcher on this thread dispatched

#include <stdio.h>
template <typename T> inline const char* sg_tname() { return sizeof(T) == 2 ? "bf16" : "f32"; }
// Formats the kernel's name once per template instantiation and notes it for the profiler.
#define SG_KNAME(...)                                                               \
  do {                                                                              \
    static char buf__[64];                                                          \
    static std::once_flag o__;                                                      \
    std::call_once(o__, [&] { snprintf(buf__, sizeof(buf__), __VA_ARGS__); });      \
    sg_tls_kernel = buf__;                                                          \
  } while (0)

struct sg_prof_scope {
  int slot;
  hipStream_t st;
  sg_prof_scope(int kind, const sg_conv_shape* s, sg_dtype dt, hipStream_t st_) : slot(-1), st(st_) {
    sg_tls_kernel = "";      // a launch that dispatches nothing must not inherit the previous call's kernel name
    if (sg_prof_on()) sg_prof_begin(kind, s, dt, st, &slot);
  }
  // an early return (a declined request) ends the record as failed: no event pair is left recorded and open
  ~sg_prof_scope() {
    if (slot >= 0) sg_prof_end(-1 - slot, st);
  }
  sg_prof_scope(const sg_prof_scope&) = delete;
  sg_prof_scope& operator=(const sg_prof_scope&) = delete;
  void name(const char* n) {
    if (slot >= 0) sg_prof_name(slot, n);
  }
  void done(int rc) {
    if (slot >= 0) sg_prof_name(slot, sg_tls_kernel);
    if (slot >= 0) sg_prof_end(rc == 0 ? slot : -1 - slot, st);
    slot = -1;
  }
};
