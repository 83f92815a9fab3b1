// pimemb_hostcopy.h -- host-side memcpy of many pieces, spread over a few worker threads.
//
// The host-pointer lookup path (the reference's only calling convention, emb_host.h:234) packs the
// caller's index arrays into pinned staging and, for calls that run zero-copy, moves the pooled rows
// from pinned staging into the caller's per-table buffers.  One thread moves ~25-30 GB/s; for the
// mid-size calls (2-40 MB) that memcpy is as long as the kernel's PCIe-bound stores.  The reference
// post-processes results on SDK worker threads too (one dpu_callback per rank, emb_host.h:337).
// Workers start lazily on the first large copy and sleep on a condition variable in between.
#pragma once

#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace pimemb {

struct CopyPiece {
    void *dst;
    const void *src;
    size_t bytes;
};

class HostCopier {
public:
    // below this many bytes in total the caller's thread copies alone (waking workers costs ~10 us)
    static constexpr size_t kParallelBytes = 2u << 20;
    static constexpr size_t kChunk = 256u << 10;

    ~HostCopier() {
        forget_workers_after_fork();
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }

    // Copies every piece; returns when all bytes have landed.  One caller at a time (the engine's
    // host-call mutex guarantees it).
    void copy(const std::vector<CopyPiece> &pieces) {
        size_t total = 0;
        for (const CopyPiece &p : pieces) total += p.bytes;
        forget_workers_after_fork();
        if (total < kParallelBytes || !ensure_workers()) {
            for (const CopyPiece &p : pieces)
                if (p.bytes) memcpy(p.dst, p.src, p.bytes);
            return;
        }
        {
            // Workers enter drain() only after taking mu_ and seeing a new generation, and count
            // themselves in active_ while inside: with mu_ held and active_ == 0 nobody reads chunks_.
            std::unique_lock<std::mutex> lk(mu_);
            while (active_.load(std::memory_order_acquire) != 0) {   // a straggler leaving the previous round
                lk.unlock();
                std::this_thread::yield();
                lk.lock();
            }
            chunks_.clear();
            for (const CopyPiece &p : pieces)
                for (size_t off = 0; off < p.bytes; off += kChunk)
                    chunks_.push_back(CopyPiece{static_cast<char *>(p.dst) + off, static_cast<const char *>(p.src) + off,
                                                p.bytes - off < kChunk ? p.bytes - off : kChunk});
            next_.store(0, std::memory_order_relaxed);
            done_.store(0, std::memory_order_relaxed);
            generation_++;
        }
        cv_.notify_all();
        drain();                                           // the caller works too
        while (done_.load(std::memory_order_acquire) < chunks_.size()) std::this_thread::yield();
    }

private:
    // A fork()ed child inherits this object but not the threads: their handles must neither be
    // joined nor destroyed (std::terminate), so the child leaks them and starts its own on demand.
    void forget_workers_after_fork() {
        if (owner_pid_ == getpid()) return;
        if (!workers_.empty()) new std::vector<std::thread>(std::move(workers_));   // intentionally leaked
        workers_.clear();
        active_.store(0);
        owner_pid_ = getpid();
    }
    bool ensure_workers() {
        if (!workers_.empty()) return true;
        if (failed_) return false;
        unsigned hw = std::thread::hardware_concurrency();
        unsigned n = hw > 4 ? 3 : (hw > 1 ? hw - 1 : 0);   // caller + up to 3 workers
        try {
            for (unsigned i = 0; i < n; i++) workers_.emplace_back([this] { run(); });
        } catch (...) {
            failed_ = workers_.empty();
        }
        return !workers_.empty();
    }
    void drain() {
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= chunks_.size()) return;
            memcpy(chunks_[i].dst, chunks_[i].src, chunks_[i].bytes);
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void run() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                active_.fetch_add(1, std::memory_order_acq_rel);
            }
            drain();
            active_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }

    std::vector<std::thread> workers_;
    std::vector<CopyPiece> chunks_;
    std::atomic<size_t> next_{0}, done_{0};
    std::atomic<int> active_{0};
    std::mutex mu_;
    std::condition_variable cv_;
    uint64_t generation_ = 0;
    bool stop_ = false, failed_ = false;
    pid_t owner_pid_ = getpid();
};

}  // namespace pimemb
