// Error reporting and library identification of libsmrf_hip.
#include <atomic>
#include <cstring>

#include "smrf_common.h"

namespace {
thread_local char g_err[512] = "";
}

namespace {
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
SmrfSwitches read_switches() {
  SmrfSwitches s;
  s.fused = env_int("SMRF_FUSED", 1);
  s.chain = env_int("SMRF_CHAIN", 1);
  s.nan_ride = env_int("SMRF_NAN_RIDE", 1);
  s.nt = env_int("SMRF_NT", -1);
  s.ring_seg = env_int("SMRF_RING_SEG", 0);
  s.ring_dual = env_int("SMRF_RING_DUAL", -1);
  s.ring_rounds = env_int("SMRF_RING_ROUNDS", 1);
  s.ring_slope = env_int("SMRF_RING_SLOPE", -1);
  s.seg_rule = env_int("SMRF_SEG_RULE", 0);
  s.xcd_remap = env_int("SMRF_XCD_REMAP", 1);
  s.fused_rounds = env_int("SMRF_FUSED_ROUNDS", 1);
  s.chain_rounds = env_int("SMRF_CHAIN_ROUNDS", 3);
  s.ring_debug = env_int("SMRF_RING_DEBUG", 0);
  return s;
}
// Read ONCE, at library load: a process that sets SMRF_* after importing neilpy_amd changes nothing until it calls
// smrf_switches_reload() (README "Developer switches").  The switches live in immutable snapshots behind an atomic
// pointer: a reload publishes a new snapshot and never frees the old one (a few dozen bytes per reload, a test hook), so
// a launch running on another host thread keeps reading a consistent set; a call that overlaps a reload may see the old
// set for its first launches and the new one after - routing only, every route gives the same bits.
std::atomic<const SmrfSwitches*> g_sw{new SmrfSwitches(read_switches())};
}  // namespace

const SmrfSwitches& smrf_sw() { return *g_sw.load(std::memory_order_acquire); }

int smrf_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" {

int smrf_abi_version(void) { return SMRF_ABI_VERSION; }

const char* smrf_last_error(void) { return g_err; }

void smrf_switches_reload(void) { g_sw.store(new SmrfSwitches(read_switches()), std::memory_order_release); }

int smrf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

}  // extern "C"
