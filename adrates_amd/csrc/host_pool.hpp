// Host-side range pool shared by the upload path (capi.hip) and the book compilers (book_host.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <system_error>
#include <thread>
#include <vector>

namespace adr {

inline int pool_threads(int64_t n, int64_t grain) {
    const int64_t hw = static_cast<int64_t>(std::thread::hardware_concurrency());
    return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({hw, 16, n / std::max<int64_t>(1, grain) + 1})));
}

// body(k, first, one past the last) over n_threads contiguous ranges of [0, n); range 0 runs on the calling thread.
// A thread that cannot be started (EAGAIN under a process / thread limit) is not an error: its range - and every
// later one - runs on the calling thread instead, after the threads that did start have been joined.  No exception
// leaves this function on account of thread creation, so none crosses the extern "C" boundary above it.
template <class Body>
void parallel_ranges(int64_t n, int n_threads, Body&& body) {
    std::vector<std::thread> pool;
    int started = 1;                                   // ranges handed to a thread (range 0: the caller)
    try {
        pool.reserve(static_cast<size_t>(std::max(0, n_threads - 1)));
        for (int k = 1; k < n_threads; ++k) {
            pool.emplace_back([&body, k, n, n_threads] { body(k, n * k / n_threads, n * (k + 1) / n_threads); });
            started = k + 1;
        }
    } catch (const std::system_error&) {
    } catch (const std::bad_alloc&) {
    }
    body(0, 0, n / n_threads);
    for (auto& th : pool) th.join();
    for (int k = started; k < n_threads; ++k) body(k, n * k / n_threads, n * (k + 1) / n_threads);
}

}  // namespace adr
