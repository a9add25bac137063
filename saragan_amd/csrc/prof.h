// Opt-in per-launch timing of the conv kernels (sg_prof_* in include/saragan_hip.h).
#pragma once
#include "common.h"

void sg_prof_begin(int kind, const sg_conv_shape* s, sg_dtype dt, hipStream_t st, int* slot);
void sg_prof_end(int slot, hipStream_t st);
bool sg_prof_on();

struct sg_prof_scope {
  int slot;
  hipStream_t st;
  sg_prof_scope(int kind, const sg_conv_shape* s, sg_dtype dt, hipStream_t st_) : slot(-1), st(st_) {
    if (sg_prof_on()) sg_prof_begin(kind, s, dt, st, &slot);
  }
  void done(int rc) {
    if (slot >= 0) sg_prof_end(rc == 0 ? slot : -1 - slot, st);
    slot = -1;
  }
};
