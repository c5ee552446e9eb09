// dto_hostxfer.cpp -- see dto_hostxfer.h.  Host threads, a pinned ring and the packing launch; no arithmetic.
#include "dto_hostxfer.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <stdexcept>
#include <string>

namespace dto {

namespace {
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + " failed: " + hipGetErrorString(e));
}
}  // namespace

void XferPlan::finalize(int64_t chunk_doubles) {
    // merge runs that touch (the builders emit them per column / per term)
    std::vector<int64_t> s2, l2;
    for (size_t i = 0; i < start.size(); ++i) {
        if (len[i] <= 0) continue;
        if (!s2.empty() && s2.back() + l2.back() >= start[i]) {
            const int64_t end = std::max(s2.back() + l2.back(), start[i] + len[i]);
            l2.back() = end - s2.back();
        } else {
            s2.push_back(start[i]);
            l2.push_back(len[i]);
        }
    }
    start.swap(s2);
    len.swap(l2);
    poff.assign(start.size() + 1, 0);
    for (size_t i = 0; i < start.size(); ++i) poff[i + 1] = poff[i] + len[i];
    // a run longer than a ring slot is split so that every chunk fits
    chunk_run.clear();
    chunk_run.push_back(0);
    int64_t base = 0;
    for (size_t i = 0; i < start.size(); ++i) {
        if (len[i] > chunk_doubles) throw std::runtime_error("XferPlan: run longer than a ring slot");
        if (poff[i + 1] - base > chunk_doubles) {
            chunk_run.push_back(i);
            base = poff[i];
        }
    }
    chunk_run.push_back(start.size());
}

HostPool::HostPool(int n_threads) {
    for (int i = 0; i < n_threads; ++i) workers_.emplace_back([this] { loop(); });
}
HostPool::~HostPool() {
    {
        std::lock_guard<std::mutex> g(m_);
        stop_ = true;
    }
    cv_job_.notify_all();
    for (auto& t : workers_) t.join();
}
void HostPool::submit(std::function<void()> job) {
    {
        std::lock_guard<std::mutex> g(m_);
        jobs_.push_back(std::move(job));
        ++pending_;
    }
    cv_job_.notify_one();
}
void HostPool::wait_all() {
    std::unique_lock<std::mutex> g(m_);
    cv_done_.wait(g, [this] { return pending_ == 0; });
}
void HostPool::loop() {
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> g(m_);
            cv_job_.wait(g, [this] { return stop_ || !jobs_.empty(); });
            if (stop_ && jobs_.empty()) return;
            job = std::move(jobs_.front());
            jobs_.pop_front();
        }
        job();
        {
            std::lock_guard<std::mutex> g(m_);
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
}

static int pool_threads() {
    unsigned hc = std::thread::hardware_concurrency();
    int n = hc ? (int)hc : 4;
    return std::max(2, std::min(n, 12));  // the box's share for one GPU is 16 cores; leave some to the caller
}

HostXfer::HostXfer() : pool_(pool_threads()) {
    for (int i = 0; i < SLOTS; ++i) {
        hip_check(hipHostMalloc((void**)&pinned_[i], CHUNK_DOUBLES * sizeof(double)), "hipHostMalloc (D2H ring)");
        hip_check(hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming), "hipEventCreate");
    }
}
HostXfer::~HostXfer() {
    pool_.wait_all();
    for (int i = 0; i < SLOTS; ++i) {
        if (pinned_[i]) (void)hipHostFree(pinned_[i]);
        if (ev_[i]) (void)hipEventDestroy(ev_[i]);
    }
}

void HostXfer::fill_constants_async(const XferPlan& p, double* vals) {
    // gaps between the runs, cut into pieces of about equal size
    const size_t nr = p.start.size();
    const int pieces = std::max(1, pool_.size() * 4);
    const int64_t per = (p.total + pieces - 1) / pieces;
    for (int q = 0; q < pieces; ++q) {
        const int64_t lo = (int64_t)q * per, hi = std::min<int64_t>(p.total, lo + per);
        if (lo >= hi) break;
        pool_.submit([&p, vals, lo, hi, nr] {
            // first run that ends after lo
            size_t r = std::upper_bound(p.start.begin(), p.start.end(), lo) - p.start.begin();
            if (r > 0 && p.start[r - 1] + p.len[r - 1] > lo) --r;
            int64_t pos = lo;
            while (pos < hi) {
                const int64_t next_start = r < nr ? std::min<int64_t>(p.start[r], hi) : hi;
                if (next_start > pos) memset(vals + pos, 0, sizeof(double) * (size_t)(next_start - pos));
                if (r >= nr) break;
                pos = std::max(pos, p.start[r] + p.len[r]);
                ++r;
            }
            // constant non-zero entries of this piece
            const size_t a = std::lower_bound(p.one_pos.begin(), p.one_pos.end(), lo) - p.one_pos.begin();
            const size_t b = std::lower_bound(p.one_pos.begin(), p.one_pos.end(), hi) - p.one_pos.begin();
            for (size_t i = a; i < b; ++i) vals[p.one_pos[i]] = p.one_val[i];
        });
    }
}

void HostXfer::fetch(const XferPlan& p, const double* d_slab, double* vals, hipStream_t st) {
    const int64_t nr = (int64_t)p.start.size();
    if (nr > 0) launch_pack_runs(st, d_slab, p.d_start, p.d_len, p.d_poff, nr, p.d_packed);
    hip_check(hipGetLastError(), "pack launch");
    const size_t nchunks = p.chunk_run.empty() ? 0 : p.chunk_run.size() - 1;
    std::vector<std::atomic<int>> left(nchunks);
    std::mutex m;
    std::condition_variable cv;
    auto chunk_done = [&](size_t c) {
        if (left[c].fetch_sub(1) == 1) {
            std::lock_guard<std::mutex> g(m);
            cv.notify_all();
        }
    };
    // copy of chunk c into its ring slot (waits until the slot's previous tenant has been scattered)
    auto enqueue = [&](size_t c) {
        const size_t r0 = p.chunk_run[c], r1 = p.chunk_run[c + 1];
        left[c] = 0;
        if (r0 == r1) return;
        const int slot = (int)(c % SLOTS);
        if (c >= (size_t)SLOTS) {
            std::unique_lock<std::mutex> g(m);
            cv.wait(g, [&] { return left[c - SLOTS].load() == 0; });
        }
        const int64_t base = p.poff[r0], cnt = p.poff[r1] - base;
        left[c] = std::max(1, std::min<int>(pool_.size(), (int)((r1 - r0 + 255) / 256)));
        hip_check(hipMemcpyAsync(pinned_[slot], p.d_packed + base, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, st), "D2H chunk");
        hip_check(hipEventRecord(ev_[slot], st), "hipEventRecord");
    };
    if (nchunks > 0) enqueue(0);
    for (size_t c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) enqueue(c + 1);  // keeps the copy engine busy while chunk c is scattered
        const size_t r0 = p.chunk_run[c], r1 = p.chunk_run[c + 1];
        if (r0 == r1) continue;
        const int slot = (int)(c % SLOTS);
        hip_check(hipEventSynchronize(ev_[slot]), "hipEventSynchronize (D2H chunk)");
        const int parts = left[c].load();
        const size_t per = (r1 - r0 + parts - 1) / parts;
        const int64_t base = p.poff[r0];
        const double* src = pinned_[slot];
        for (int q = 0; q < parts; ++q) {
            const size_t a = std::min(r1, r0 + (size_t)q * per), b = std::min(r1, a + per);
            pool_.submit([&p, vals, a, b, base, src, c, &chunk_done] {
                for (size_t r = a; r < b; ++r)
                    memcpy(vals + p.start[r], src + (p.poff[r] - base), sizeof(double) * (size_t)p.len[r]);
                chunk_done(c);
            });
        }
    }
    pool_.wait_all();
}

}  // namespace dto
