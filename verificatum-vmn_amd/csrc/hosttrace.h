// hosttrace.h — wall-clock accounting of host-side sections (VMN_TRACE=1 prints a table to stderr at exit).
// A measurement aid for the small-N work: at N = 10^4 a proof is bound by what the HOST does between the
// kernels, which no GPU profile shows.  Disabled: one predictable branch per scope.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace vmn {
namespace trace {

struct Event {
    const char* label;
    long long t0, t1;          // ns, CLOCK_MONOTONIC (std::chrono::steady_clock): the clock of rocprofv3's kernel trace
    size_t tid;
};
struct Table {
    bool on = false;
    std::mutex mu;
    std::map<std::string, std::pair<long, double>> acc;
    const char* events_path = nullptr;         // VMN_TRACE_EVENTS=file: every scope as a line "label,t0_ns,t1_ns,thread"
    std::vector<Event> events;                 //   (tools/idle_gaps.py lays them over the kernel trace of the same run)
    Table() {
        events_path = getenv("VMN_TRACE_EVENTS");
        on = getenv("VMN_TRACE") != nullptr || events_path != nullptr;
    }
    ~Table() {
        if (!on) return;
        if (events_path) {
            if (FILE* f = fopen(events_path, "a")) {
                for (auto& e : events) fprintf(f, "%s,%lld,%lld,%zu\n", e.label, e.t0, e.t1, e.tid);
                fclose(f);
            }
        }
        if (!getenv("VMN_TRACE")) return;
        fprintf(stderr, "[vmn trace] %-36s %8s %12s\n", "section", "calls", "ms");
        for (auto& kv : acc) fprintf(stderr, "[vmn trace] %-36s %8ld %12.3f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    }
};
inline Table& table() {
    static Table t;
    return t;
}
struct Scope {
    const char* label;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit Scope(const char* l) : label(l), on(table().on) {
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~Scope() {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        Table& t = table();
        std::lock_guard<std::mutex> g(t.mu);
        auto& e = t.acc[label];
        e.first += 1;
        e.second += ms;
        if (t.events_path)
            t.events.push_back({label, (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(t0.time_since_epoch()).count(),
                                (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(t1.time_since_epoch()).count(),
                                std::hash<std::thread::id>()(std::this_thread::get_id())});
    }
};
inline void reset() {
    Table& t = table();
    std::lock_guard<std::mutex> g(t.mu);
    t.acc.clear();
}

}  // namespace trace
}  // namespace vmn
#define VMN_TRACE_CAT2(a, b) a##b
#define VMN_TRACE_CAT(a, b) VMN_TRACE_CAT2(a, b)
#define VMN_TRACE(label) vmn::trace::Scope VMN_TRACE_CAT(trace_scope_, __LINE__)(label)
