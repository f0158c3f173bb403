// C-ABI glue: error reporting, optional per-launch HIP-event profiling, conv entry points, self-test.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "conv_args.h"

static thread_local char g_err[512] = "";

void ieagan_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ieagan_last_error(void) { return g_err; }
extern "C" int ieagan_abi_version(void) { return IEAGAN_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------
// profiling: when enabled every launcher brackets its launches with two events on its stream
// ------------------------------------------------------------------------------------------------
struct ProfEvt {
    std::string name;
    hipEvent_t a, b;
    double flops, bytes, bytes_min;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfEvt> g_prof;

static bool g_prof_tags = false;
bool prof_tags_on() { return g_prof_on && g_prof_tags; }

static thread_local ConvPlanCtx g_plan = {false, 0};
ConvPlanCtx& conv_plan_ctx() { return g_plan; }

ProfScope::ProfScope(const char* nm, double fl, double by, hipStream_t s, const char* tag, double bmin)
    : active(false), stream(s), ev_a(nullptr), ev_b(nullptr), flops(fl), bytes(by), bytes_min(bmin < 0.0 ? by : bmin) {
    if (!g_prof_on || g_plan.on) return;
    if (g_prof_tags && tag != nullptr) snprintf(name, sizeof(name), "%s %s", nm, tag);
    else snprintf(name, sizeof(name), "%s", nm);
    if (hipEventCreate(&ev_a) != hipSuccess) return;
    if (hipEventCreate(&ev_b) != hipSuccess) {
        hipEventDestroy(ev_a);
        return;
    }
    hipEventRecord(ev_a, s);
    active = true;
}

ProfScope::~ProfScope() {
    if (!active) return;
    hipEventRecord(ev_b, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);        // the finished record changes hands here; collect / reset own it from now on
    g_prof.push_back(ProfEvt{std::string(name), ev_a, ev_b, flops, bytes, bytes_min});
}

extern "C" int ieagan_prof_enable(int on) {      // 0 off, 1 per kernel family, 2 per kernel family + shape tag
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    g_prof_tags = on == 2;
    return 0;
}

static void prof_clear_locked() {
    for (auto& e : g_prof) {
        hipEventDestroy(e.a);
        hipEventDestroy(e.b);
    }
    g_prof.clear();
}

extern "C" int ieagan_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_clear_locked();
    return 0;
}

extern "C" int ieagan_prof_collect(ieagan_prof_rec* out, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, ieagan_prof_rec> agg;
    for (auto& e : g_prof) {
        if (hipEventSynchronize(e.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) continue;
        auto& r = agg[e.name];
        if (r.launches == 0) {
            memset(&r, 0, sizeof(r));
            strncpy(r.name, e.name.c_str(), sizeof(r.name) - 1);
        }
        r.launches += 1;
        r.ms += ms;
        r.flops += e.flops;
        r.bytes += e.bytes;
        r.bytes_min += e.bytes_min;
    }
    prof_clear_locked();
    int n = 0;
    for (auto& kv : agg) {
        if (n >= cap) break;
        out[n++] = kv.second;
    }
    return n;
}

// ------------------------------------------------------------------------------------------------
extern "C" int ieagan_conv_forward(const ieagan_conv_desc* d, void* stream) {
    CHECK_ARG(d != nullptr && d->src.x != nullptr && d->w != nullptr && d->out != nullptr, "conv_forward: null pointer");
    return conv_gather_launch(*d, (hipStream_t)stream);
}

extern "C" int ieagan_conv_stats_slots(const ieagan_conv_desc* d) {
    CHECK_ARG(d != nullptr, "conv_stats_slots: null pointer");
    g_plan.on = true;
    g_plan.slots = -1;
    const int rc = conv_gather_launch(*d, nullptr);
    g_plan.on = false;
    if (rc < 0) return rc;
    CHECK_ARG(g_plan.slots > 0, "conv_stats_slots: the dispatch reached no launch site");
    return g_plan.slots;
}

extern "C" int ieagan_conv_wgrad(const ieagan_wgrad_desc* d, int use_tr_read, void* stream) {
    CHECK_ARG(d != nullptr && d->src.x != nullptr && d->g != nullptr && d->dw != nullptr, "conv_wgrad: null pointer");
    return conv_wgrad_launch(*d, (hipStream_t)stream, use_tr_read);
}

// ------------------------------------------------------------------------------------------------
// self-test of ds_read_b64_tr_b16 as used by conv_wgrad: in = [64 rows][16 cols] bf16, one wave;
// out[lane][j] must equal in[8*(lane>>4) + j][lane & 15]  (the MFMA B-operand fragment).
// ------------------------------------------------------------------------------------------------
__global__ void selftest_tr_kernel(const bf16* __restrict__ in, bf16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) bf16 lds[64 * 16];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 16; i += 64) lds[i] = in[i];
    __syncthreads();
    const int lr = l & 15, lg = l >> 4;
    const int q = lr >> 2, p = lr & 3;
    const bf16* p0 = lds + (8 * lg + q) * 16 + 4 * p;
    const bf16* p1 = p0 + 4 * 16;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        out[l * 8 + j] = lo[j];
        out[l * 8 + 4 + j] = hi[j];
    }
}

extern "C" int ieagan_selftest_tr_read(const void* in, void* out, void* stream) {
    hipLaunchKernelGGL(selftest_tr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)in, (bf16*)out);
    CHECK_LAUNCH("selftest_tr_read");
    return 0;
}
