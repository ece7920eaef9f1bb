// parallel.hpp -- parallel_for on a persistent pool of host threads.
//
// The driver's stages call parallel_for a few times per 8-32 MB window of input (byte-range jobs of the MAF reader, the rows' copies, the
// writers' pieces), from up to three pipeline threads at once.  Until round 4 every call created and joined its threads: ~30-50 us per
// thread, i.e. ~0.7 ms per call on sixteen threads -- more than the work of a call whose jobs only walk their reference rows (measured on
// the MI355X box's host, tools/maf_direct_ab.sh: stage 1's "measure" phase took 0.023 s per 10 Mb whatever the thread count, its scan alone
// runs at 19 GB/s per thread).  Now the threads are created once and sleep on a condition variable between calls; a call publishes its
// index range, takes part itself, and returns when every index has run (the first exception is rethrown, as before).  Calls from
// different threads share the pool: an idle worker joins whichever open call still has indices and room (at most `threads` participants).
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

namespace pcsf_host {

class HostThreadPool {
public:
    static HostThreadPool& instance()
    {
        static HostThreadPool pool;
        return pool;
    }

    template <class Fn>
    void run(size_t n, int threads, Fn&& fn)
    {
        if (n == 0) return;
        const size_t nt = std::min<size_t>(n, static_cast<size_t>(std::max(threads, 1)));
        if (nt == 1) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        Call call;
        call.n = n;
        call.room = nt - 1;   // workers that may still join (the caller is the nt-th participant)
        call.ctx = &fn;
        call.invoke = [](void* ctx, size_t i) { (*static_cast<typename std::remove_reference<Fn>::type*>(ctx))(i); };
        {
            std::unique_lock<std::mutex> g(lock_);
            grow(nt - 1);
            open_.push_back(&call);
        }
        wake_.notify_all();
        work_on(call);
        {
            std::unique_lock<std::mutex> g(lock_);
            for (auto it = open_.begin(); it != open_.end(); ++it)
                if (*it == &call) {
                    open_.erase(it);
                    break;
                }
            call_done_.wait(g, [&] { return call.busy == 0; });   // workers still inside fn of this call
        }
        if (call.error) std::rethrow_exception(call.error);
    }

private:
    struct Call {
        size_t n = 0;
        std::atomic<size_t> next{0};
        size_t room = 0, busy = 0;   // under lock_
        void* ctx = nullptr;
        void (*invoke)(void*, size_t) = nullptr;
        std::exception_ptr error;     // under error_lock
        std::mutex error_lock;
    };

    HostThreadPool() = default;
    ~HostThreadPool()
    {
        {
            std::unique_lock<std::mutex> g(lock_);
            stop_ = true;
        }
        wake_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    HostThreadPool(const HostThreadPool&) = delete;
    HostThreadPool& operator=(const HostThreadPool&) = delete;

    void grow(size_t workers)   // lock_ held
    {
        while (workers_.size() < workers) workers_.emplace_back([this] { worker(); });
    }

    static void work_on(Call& call)
    {
        try {
            for (size_t i = call.next.fetch_add(1); i < call.n; i = call.next.fetch_add(1)) call.invoke(call.ctx, i);
        } catch (...) {
            std::lock_guard<std::mutex> g(call.error_lock);
            if (!call.error) call.error = std::current_exception();
            call.next.store(call.n);
        }
    }

    void worker()
    {
        std::unique_lock<std::mutex> g(lock_);
        for (;;) {
            Call* pick = nullptr;
            for (Call* c : open_)
                if (c->room > 0 && c->next.load(std::memory_order_relaxed) < c->n) {
                    pick = c;
                    break;
                }
            if (!pick) {
                if (stop_) return;
                wake_.wait(g);
                continue;
            }
            --pick->room;
            ++pick->busy;
            g.unlock();
            work_on(*pick);
            g.lock();
            ++pick->room;
            if (--pick->busy == 0) call_done_.notify_all();
        }
    }

    std::mutex lock_;
    std::condition_variable wake_, call_done_;
    std::deque<Call*> open_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};

// fn(i) for i in [0, n) on up to `threads` threads (dynamic schedule); the first exception is rethrown
template <class Fn>
inline void parallel_for(size_t n, int threads, Fn&& fn)
{
    HostThreadPool::instance().run(n, threads, std::forward<Fn>(fn));
}

}  // namespace pcsf_host
