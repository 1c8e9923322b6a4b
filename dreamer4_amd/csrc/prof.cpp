// Glue-kernel launch timing (prof.h).  Single-threaded entry is assumed for the profiling switches (bench.py drives them).
#include "prof.h"
#include "common.h"
#include <mutex>
#include <vector>

namespace d4 {

namespace {
struct Rec { hipEvent_t a, b; int cls; double bytes, flops; };
std::mutex g_mu;
int g_mask = 0, g_stride = 1;
int g_tick[GL_N] = {0};
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
const char* const kNames[GL_N] = {"space_attn_kernel", "time_attn64_kernel", "time_kv_append_kernel", "pool_mix_kernel", "small_attn_kernel",
                                  "assemble_kernel", "splitk_reduce_kernel", "attn_wide_kernel", "frame_attn_out_kernel", "frame_pool_kernel"};
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool glue_prof_begin(int cls, double bytes, hipEvent_t* a, hipEvent_t* b, double flops) {
    if (!((g_mask >> cls) & 1)) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    if ((g_tick[cls]++ % g_stride) != 0) return false;
    *a = get_event(); *b = get_event();
    g_recs.push_back(Rec{*a, *b, cls, bytes, flops});
    return true;
}

bool glue_profile_active() { return g_mask != 0; }

int glue_profile_enable(int m) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_stride = (m >> 26) > 0 ? (m >> 26) : 1;
    g_mask = m & 0x3FFFFFF;
    for (int i = 0; i < GL_N; ++i) g_tick[i] = 0;
    return 0;
}

int glue_profile_read(double* ms, double* bytes, int64_t* count, int nclass) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < nclass; ++i) { ms[i] = 0; bytes[i] = 0; count[i] = 0; }
    for (auto& r : g_recs) {
        D4_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        D4_HIP(hipEventElapsedTime(&t, r.a, r.b));
        if (r.cls < nclass) { ms[r.cls] += t; bytes[r.cls] += r.bytes; count[r.cls] += 1; }
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.clear();
    return 0;
}

int glue_profile_read_flops(double* flops, int nclass) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < nclass; ++i) flops[i] = 0;
    for (auto& r : g_recs) if (r.cls < nclass) flops[r.cls] += r.flops;
    return 0;
}

const char* glue_class_name(int c) { return c >= 0 && c < GL_N ? kNames[c] : ""; }

}  // namespace d4
