// hipEvent bracketing of conv launches; off by default.  The only global state of the library.
#include "prof.h"
#include <mutex>
#include <vector>
#include <string.h>
#include <stdlib.h>
#include <atomic>

namespace {
struct Rec { int kind; sg_conv_shape shape; int dtype; hipEvent_t e0, e1; bool ok; const char* name; };
std::mutex g_mu;
bool g_on = false;
bool g_filter = false;
int g_fkind = 0;
sg_conv_shape g_fshape;
std::vector<Rec> g_recs;
const char* kVersion = "saragan_hip 0.4 (gfx950)";
}  // namespace

bool sg_prof_on() { return g_on; }
thread_local const char* sg_tls_kernel = "";

namespace {
std::atomic<const sg_config*> g_cfg{nullptr};
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
const sg_config* read_config() {
  sg_config* c = new sg_config;   // snapshots are never freed: a launch may still hold the previous one
  c->fwd_lds = env_int("SG_FWD_LDS", 80 * 1024);
  c->fwd_tg = env_int("SG_FWD_TG", 0);
  c->fwd_v1 = env_int("SG_FWD_V1", 0);
  c->fwd_no_pw = env_int("SG_FWD_NO_PW", 0);
  c->fwd_no_dense = env_int("SG_FWD_NO_DENSE", 0);
  c->fwd_no_v3 = env_int("SG_FWD_NO_V3", 0);
  c->fwd_no_v3s = env_int("SG_FWD_NO_V3S", 0);
  c->wgrad_no_lean = env_int("SG_WGRAD_NO_LEAN", 0);
  c->wgrad_no_w16 = env_int("SG_WGRAD_NO_W16", 0);
  c->wgrad3l_16 = env_int("SG_WGRAD3L_16", 0);
  c->wgrad3l_min_cols = env_int("SG_WGRAD3L_MIN_COLS", 0);
  c->wgrad_v1_blocks = env_int("SG_WGRAD_V1_BLOCKS", 0);
  c->fwd_no_v4 = env_int("SG_FWD_NO_V4", 0);
  c->fwd_no_v5 = env_int("SG_FWD_NO_V5", 0);
  c->fwd_no_ksplit = env_int("SG_FWD_NO_KSPLIT", 0);
  c->fwd_no_3p = env_int("SG_FWD_NO_3P", 0);
  c->fwd3p_16 = env_int("SG_FWD3P_16", 1);
  c->fwd3s_16 = env_int("SG_FWD3S_16", 1);
  c->fwd3_gx = env_int("SG_FWD3_GX", 0);
  c->fwd3_no_lean = env_int("SG_FWD3_NO_LEAN", 0);
  c->fwd4_gx = env_int("SG_FWD4_GX", 0);
  c->fwd4_no_lean = env_int("SG_FWD4_NO_LEAN", 0);
  c->fwd4_no_wres = env_int("SG_FWD4_NO_WRES", 0);
  c->wgrad_v1 = env_int("SG_WGRAD_V1", 0);
  c->wgrad_no_v3 = env_int("SG_WGRAD_NO_V3", 0);
  c->dbg_flags = env_int("SG_DBG_FLAGS", 0);
  c->no_small = env_int("SG_NO_SMALL", 0);
  c->deterministic = env_int("SG_DETERMINISTIC", 0);
  c->no_gemm = env_int("SG_NO_GEMM", 0);
  c->gemm_ks_model = env_int("SG_GEMM_KS_MODEL", 0);
  c->gemm_k333_maxvox = env_int("SG_GEMM_K333_MAXVOX", 8192);
  return c;
}
}  // namespace

const sg_config& sg_cfg() {
  const sg_config* c = g_cfg.load(std::memory_order_acquire);
  if (c == nullptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    c = g_cfg.load(std::memory_order_acquire);
    if (c == nullptr) {
      c = read_config();
      g_cfg.store(c, std::memory_order_release);
    }
  }
  return *c;
}

extern "C" int sg_config_reload(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_cfg.store(read_config(), std::memory_order_release);
  return SG_OK;
}

void sg_prof_begin(int kind, const sg_conv_shape* s, sg_dtype dt, hipStream_t st, int* slot) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_filter && (kind != g_fkind || memcmp(s, &g_fshape, sizeof(sg_conv_shape)) != 0)) { *slot = -1; return; }
  Rec r; r.kind = kind; r.shape = *s; r.dtype = (int)dt; r.ok = false; r.name = "";
  if (hipEventCreate(&r.e0) != hipSuccess) { *slot = -1; return; }
  if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); *slot = -1; return; }
  (void)hipEventRecord(r.e0, st);
  g_recs.push_back(r);
  *slot = (int)g_recs.size() - 1;
}

void sg_prof_end(int slot, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  bool ok = slot >= 0;
  int idx = ok ? slot : -1 - slot;
  if (idx < 0 || idx >= (int)g_recs.size()) return;
  (void)hipEventRecord(g_recs[idx].e1, st);
  g_recs[idx].ok = ok;
}

void sg_prof_name(int slot, const char* name) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot >= 0 && slot < (int)g_recs.size()) g_recs[slot].name = name;
}

extern "C" const char* sg_version(void) { return kVersion; }

extern "C" const char* sg_error_string(int code) {
  switch (code) {
    case SG_OK: return "ok";
    case SG_EINVAL: return "invalid argument (shape, null pointer or unsupported combination)";
    case SG_EWORKSPACE: return "workspace too small";
    case SG_EALIGN: return "pointer not 16-byte aligned";
    case SG_EUNSUPPORTED: return "no kernel covers this request";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

extern "C" int sg_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  if (!g_on) {
    for (auto& r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_recs.clear();
  }
  return SG_OK;
}

extern "C" int sg_prof_enabled(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_on ? 1 : 0;
}

extern "C" int sg_prof_set_filter(int kind, const sg_conv_shape* s) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_filter = s != nullptr;
  if (s) { g_fkind = kind; g_fshape = *s; }
  return SG_OK;
}

extern "C" int sg_prof_collect(sg_prof_entry* out, int32_t max_entries, int32_t* n_entries) {
  if (!out || !n_entries || max_entries < 0) return SG_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (auto& r : g_recs) {
    float ms = 0.f;
    bool ok = r.ok && hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess;
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    if (!ok) continue;
    int j = 0;
    for (; j < n; ++j)
      if (out[j].kind == r.kind && out[j].dtype == r.dtype && memcmp(&out[j].shape, &r.shape, sizeof(sg_conv_shape)) == 0 &&
          strncmp(out[j].kernel, r.name ? r.name : "", sizeof(out[j].kernel) - 1) == 0) break;   // pooled / masked / plain variants apart
    if (j == n) {
      if (n >= max_entries) continue;
      out[n].kind = r.kind; out[n].shape = r.shape; out[n].dtype = r.dtype; out[n].launches = 0; out[n].total_ms = 0;
      const sg_conv_shape& s = r.shape;
      out[n].flops_per_launch = 2.0 * s.n * s.d * s.h * s.w * (double)s.cin * s.cout * s.kd * s.kh * s.kw;
      strncpy(out[n].kernel, r.name ? r.name : "", sizeof(out[n].kernel) - 1);
      out[n].kernel[sizeof(out[n].kernel) - 1] = 0;
      ++n;
    }
    out[j].launches += 1;
    out[j].total_ms += ms;
  }
  g_recs.clear();
  *n_entries = n;
  return SG_OK;
}
