// Library-level entry points: ABI version and the per-thread error text.
#include "common.hpp"

#include <mutex>
#include <vector>

namespace evi {

std::string& last_error_ref() {
    static thread_local std::string msg;
    return msg;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

// ---- timing ----------------------------------------------------------------------------------
namespace {
struct TimedSpan {
    int cls;
    hipEvent_t start, stop;
};
std::mutex g_timing_mu;
bool g_timing_on = false;
std::vector<TimedSpan> g_spans;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t take_event() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

bool timing_enabled() { return g_timing_on; }

int timing_begin(int cls, hipStream_t st) {
    if (!g_timing_on) return -1;
    std::lock_guard<std::mutex> lock(g_timing_mu);
    TimedSpan span{cls, take_event(), take_event()};
    if (!span.start || !span.stop) return -1;
    (void)hipEventRecord(span.start, st);
    g_spans.push_back(span);
    return (int)g_spans.size() - 1;
}

bool timing_kernel_events(int cls, hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!g_timing_on) return false;
    std::lock_guard<std::mutex> lock(g_timing_mu);
    TimedSpan span{cls, take_event(), take_event()};
    if (!span.start || !span.stop) return false;
    g_spans.push_back(span);
    *start = span.start;
    *stop = span.stop;
    return true;
}

void timing_end(int token, hipStream_t st) {
    if (token < 0) return;
    std::lock_guard<std::mutex> lock(g_timing_mu);
    if (token < (int)g_spans.size()) (void)hipEventRecord(g_spans[token].stop, st);
}

}  // namespace evi

extern "C" int evi_timing_enable(int on) {
    std::lock_guard<std::mutex> lock(evi::g_timing_mu);
    evi::g_timing_on = on != 0;
    return EVI_OK;
}

extern "C" int evi_timing_read(double* ms_host, int32_t* launches_host, int n_classes) {
    using namespace evi;
    EVI_REQUIRE(ms_host && launches_host && n_classes >= 1, "evi_timing_read: bad arguments");
    std::lock_guard<std::mutex> lock(g_timing_mu);
    for (int c = 0; c < n_classes; ++c) {
        ms_host[c] = 0.0;
        launches_host[c] = 0;
    }
    for (const TimedSpan& s : g_spans) {
        float ms = 0.f;
        EVI_HIP_CHECK(hipEventSynchronize(s.stop));
        EVI_HIP_CHECK(hipEventElapsedTime(&ms, s.start, s.stop));
        if (s.cls >= 0 && s.cls < n_classes) {
            ms_host[s.cls] += ms;
            launches_host[s.cls] += 1;
        }
        g_event_pool.push_back(s.start);
        g_event_pool.push_back(s.stop);
    }
    g_spans.clear();
    return EVI_OK;
}

extern "C" int evi_version(void) { return EVI_ABI_VERSION; }

extern "C" size_t evi_last_error(char* buf, size_t buf_bytes) {
    const std::string& msg = evi::last_error_ref();
    if (buf && buf_bytes > 0) {
        size_t n = msg.size() < buf_bytes - 1 ? msg.size() : buf_bytes - 1;
        __builtin_memcpy(buf, msg.data(), n);
        buf[n] = '\0';
    }
    return msg.size();
}
