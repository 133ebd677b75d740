// Exception barrier of the HOST-side entry points (tmpnn_pdb.cpp, tmpnn_csv.cpp). A C++ exception must never leave an
// extern "C" function (the callers are ctypes / cgo / JNI frames: undefined behaviour, in practice std::terminate), and one that
// escapes a worker thread ends the whole process. Both are turned into the library's ordinary error return:
//   tm_host_guard("name", [&]() -> int { ... })     the body of an entry point; bad_alloc -> TMPNN_E_WORKSPACE, others -> TMPNN_E_INVALID
//   tm_run_pool(n_threads, work)                    work() on the caller's thread + up to n_threads - 1 others; a thread that cannot
//                                                   be started is not started (the rest take its share); the first exception thrown
//                                                   inside work() is rethrown on the caller's thread after every thread has joined
// (.hpp on purpose: bench.kernel_source_stamp() hashes csrc/*.hip and *.h — the DEVICE sources; this file is host-only.)
#pragma once
#include <exception>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/tmpnn.h"
int tm_set_error(int code, const char *fmt, ...);

template <class F>
inline int tm_host_guard(const char *what, F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return tm_set_error(TMPNN_E_WORKSPACE, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return tm_set_error(TMPNN_E_INVALID, "%s: %s", what, e.what());
    } catch (...) {
        return tm_set_error(TMPNN_E_INVALID, "%s: unknown C++ exception", what);
    }
}

template <class W>
inline void tm_run_pool(int n_threads, W &&work) {
    std::exception_ptr first;
    std::mutex mu;
    auto guarded = [&]() noexcept {
        try {
            work();
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            if (!first) first = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve(n_threads > 1 ? (size_t)n_threads - 1 : 0);
        for (int t = 1; t < n_threads; ++t) pool.emplace_back(guarded);
    } catch (const std::system_error &) {      // no more threads to be had: go on with the ones that started
    } catch (const std::bad_alloc &) {
    }
    guarded();
    for (auto &t : pool) t.join();
    if (first) std::rethrow_exception(first);
}
