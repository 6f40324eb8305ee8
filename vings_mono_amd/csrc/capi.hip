// Error reporting and version entry points of libvings_hip.so.
#include <stdarg.h>

#include "common.hpp"

namespace vings {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vings

extern "C" const char* vings_last_error(void) { return vings::g_err; }
namespace vings {
bool trace_sync() {
    static const bool on = [] {
        const char* e = getenv("VINGS_TRACE_SYNC");
        return e && e[0] == '1';
    }();
    return on;
}
}  // namespace vings

extern "C" int vings_abi_version(void) { return 4; }   // 4: record layout {Tu,x}{Tv,y}{Tw,n0}.., ordered passes

// ---------------------------------------------------------------------------
// Per-kernel timing with HIP events recorded on the caller's stream.
// Events come from a pool created by vings_profile_enable(1); nothing is
// allocated while timing is running.
// ---------------------------------------------------------------------------
#include <mutex>
#include <vector>

namespace vings {
namespace {
struct Pair { hipEvent_t a, b; int tag; bool closed; };
std::mutex g_mu;
std::vector<Pair> g_pairs;      // pool
size_t g_used = 0;
bool g_on = false;
uint32_t g_mask = ~0u;          // tags that are timed while g_on
int g_open[PROF_NTAGS];
const char* kTagNames[PROF_NTAGS] = {"project_fwd", "bin", "blend_fwd", "blend_bwd",
                                     "project_bwd", "adam", "dba_jacobian", "dba_schur",
                                     "dba_solve", "dba_retract", "aux", "loss", "splat_step"};
constexpr size_t kPool = 16384;
}  // namespace

bool prof_enabled() { return g_on; }

void prof_begin(int tag, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on || !((g_mask >> tag) & 1u) || g_used >= g_pairs.size()) { g_open[tag] = -1; return; }
    Pair& p = g_pairs[g_used];
    p.tag = tag; p.closed = false;
    if (hipEventRecord(p.a, s) != hipSuccess) { g_open[tag] = -1; return; }
    g_open[tag] = (int)g_used++;
}

void prof_end(int tag, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int i = g_open[tag];
    if (i < 0) return;
    if (hipEventRecord(g_pairs[i].b, s) == hipSuccess) g_pairs[i].closed = true;
    g_open[tag] = -1;
}
}  // namespace vings

extern "C" int vings_profile_enable(int on) {
    using namespace vings;
    std::lock_guard<std::mutex> lk(g_mu);
    if (on && g_pairs.empty()) {
        g_pairs.resize(kPool);
        for (auto& p : g_pairs) {
            VINGS_CHECK_HIP(hipEventCreate(&p.a));
            VINGS_CHECK_HIP(hipEventCreate(&p.b));
            p.closed = false;
        }
    }
    for (int t = 0; t < PROF_NTAGS; ++t) g_open[t] = -1;
    g_used = 0;
    g_on = on != 0;
    return VINGS_OK;
}

extern "C" int vings_profile_select(uint32_t tag_mask) {
    std::lock_guard<std::mutex> lk(vings::g_mu);
    vings::g_mask = tag_mask;
    return VINGS_OK;
}

extern "C" int vings_profile_num_tags(void) { return vings::PROF_NTAGS; }
extern "C" const char* vings_profile_tag_name(int tag) {
    return (tag >= 0 && tag < vings::PROF_NTAGS) ? vings::kTagNames[tag] : "";
}

// Sum of elapsed milliseconds and number of timed launches for `tag`.
// Synchronises on the recorded events.
extern "C" int vings_profile_read(int tag, double* total_ms, int64_t* count) {
    using namespace vings;
    VINGS_REQUIRE(total_ms && count && tag >= 0 && tag < PROF_NTAGS, "profile_read args");
    std::lock_guard<std::mutex> lk(g_mu);
    double ms = 0; int64_t n = 0;
    for (size_t i = 0; i < g_used; ++i) {
        const Pair& p = g_pairs[i];
        if (p.tag != tag || !p.closed) continue;
        VINGS_CHECK_HIP(hipEventSynchronize(p.b));
        float e = 0.f;
        VINGS_CHECK_HIP(hipEventElapsedTime(&e, p.a, p.b));
        ms += e; ++n;
    }
    *total_ms = ms; *count = n;
    return VINGS_OK;
}
