#include "prof.h"
#include "common.h"
#include "../../../include/hrnet_hip.h"

#include <mutex>
#include <string>
#include <vector>
#include <string.h>

namespace {
struct Family { std::string name; long launches = 0; double ms = 0, flops = 0, bytes = 0; };
struct Rec { int fam; hipEvent_t a, b; double flops, bytes; };
std::mutex g_mu;
bool g_on = false;
std::vector<Family> g_fam;
std::vector<Rec> g_rec;          // pending (not yet folded into g_fam)
std::vector<hipEvent_t> g_pool;  // recycled events

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void fold_locked() {
    for (Rec& r : g_rec) {
        float ms = 0.f;
        if (r.a && r.b && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            Family& f = g_fam[r.fam];
            f.launches += 1; f.ms += ms; f.flops += r.flops; f.bytes += r.bytes;
        }
        if (r.a) g_pool.push_back(r.a);
        if (r.b) g_pool.push_back(r.b);
    }
    g_rec.clear();
}
}  // namespace

HrnProfScope::HrnProfScope(const char* family, double flops, double bytes, hipStream_t s) : rec(-1), stream(s) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    int fam = -1;
    for (size_t i = 0; i < g_fam.size(); ++i) if (g_fam[i].name == family) { fam = (int)i; break; }
    if (fam < 0) { g_fam.push_back(Family()); g_fam.back().name = family; fam = (int)g_fam.size() - 1; }
    Rec r; r.fam = fam; r.a = get_event(); r.b = get_event(); r.flops = flops; r.bytes = bytes;
    if (r.a) (void)hipEventRecord(r.a, s);
    g_rec.push_back(r);
    rec = (int)g_rec.size() - 1;
}
HrnProfScope::~HrnProfScope() {
    if (rec < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (rec < (int)g_rec.size() && g_rec[rec].b) (void)hipEventRecord(g_rec[rec].b, stream);
}

extern "C" {
int hrn_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (on) { fold_locked(); g_fam.clear(); g_on = true; }
    else { g_on = false; fold_locked(); }
    return 0;
}
int hrn_profile_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    fold_locked();
    return (int)g_fam.size();
}
int hrn_profile_get(int idx, char* name, int name_len, long* launches, double* total_ms, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    fold_locked();
    if (idx < 0 || idx >= (int)g_fam.size()) { hrn_set_error("hrn_profile_get: index %d out of range", idx); return -2; }
    const Family& f = g_fam[idx];
    if (name && name_len > 0) { strncpy(name, f.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (launches) *launches = f.launches;
    if (total_ms) *total_ms = f.ms;
    if (flops) *flops = f.flops;
    if (bytes) *bytes = f.bytes;
    return 0;
}
}

// ---- per-device opt-in to more than 64 KB of dynamic LDS for a kernel (hipFuncAttributeMaxDynamicSharedMemorySize is a
// property of the function ON THE CURRENT DEVICE: a process that switches devices must set it again there)
#include <set>
#include <utility>
int hrn_allow_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    HRN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return 0;
    HRN_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({dev, kernel});
    return 0;
}

// ---- compute units of the CURRENT device (persistent kernels launch one workgroup per CU): cached per device, thread-safe
int hrn_device_cus(void) {
    static std::mutex mu;
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(mu);
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}
