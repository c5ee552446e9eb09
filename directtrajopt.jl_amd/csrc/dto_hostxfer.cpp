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

void XferPlan::finalize(int64_t n_int, int64_t max_run) {
    const bool has_early = !early.empty();
    // merge runs that touch (the builders emit them per column / per term) when they belong to the same class; cut a run
    // that would not fit a ring slot
    std::vector<int64_t> s2, l2;
    std::vector<int32_t> e2;
    for (size_t i = 0; i < start.size(); ++i) {
        if (len[i] <= 0) continue;
        const int32_t cls = has_early ? early[i] : -1;
        if (!s2.empty() && s2.back() + l2.back() >= start[i] && e2.back() == cls) {
            const int64_t end = std::max(s2.back() + l2.back(), start[i] + len[i]);
            l2.back() = end - s2.back();
        } else {
            s2.push_back(start[i]);
            l2.push_back(len[i]);
            e2.push_back(cls);
        }
        while (l2.back() > max_run) {
            const int64_t rest = l2.back() - max_run, at = s2.back() + max_run;
            l2.back() = max_run;
            s2.push_back(at);
            l2.push_back(rest);
            e2.push_back(cls);
        }
    }
    start.swap(s2);
    len.swap(l2);
    early.swap(e2);
    // packed order: the early runs interval by interval, then everything else in slab order
    const size_t nr = start.size();
    std::vector<size_t> order(nr);
    for (size_t i = 0; i < nr; ++i) order[i] = i;
    auto key = [&](size_t i) { return early[i] >= 0 ? (int64_t)early[i] : n_int; };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key(a) < key(b); });
    pk_start.resize(nr);
    pk_len.resize(nr);
    pk_poff.assign(nr + 1, 0);
    early_off.assign((size_t)n_int + 1, 0);
    for (size_t i = 0; i < nr; ++i) {
        pk_start[i] = start[order[i]];
        pk_len[i] = len[order[i]];
        pk_poff[i + 1] = pk_poff[i] + pk_len[i];
        if (early[order[i]] >= 0) {
            if (early[order[i]] >= n_int) throw std::runtime_error("XferPlan: early run of an interval the shard does not own");
            ++early_off[(size_t)early[order[i]] + 1];
        }
    }
    for (int64_t i = 0; i < n_int; ++i) early_off[(size_t)i + 1] += early_off[(size_t)i];
    built = true;
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

HostXfer::HostXfer() : pool_(pool_threads()), busy_(new std::atomic<int>[SLOTS]) {
    hip_check(hipGetDevice(&device_), "hipGetDevice");
    hip_check(hipStreamCreateWithFlags(&copy_, hipStreamNonBlocking), "hipStreamCreate (copy stream)");
    for (int i = 0; i < SLOTS; ++i) {
        busy_[i] = 0;
        hip_check(hipHostMalloc((void**)&pinned_[i], CHUNK_DOUBLES * sizeof(double)), "hipHostMalloc (D2H ring)");
        hip_check(hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming), "hipEventCreate");
    }
    drainer_ = std::thread([this] { drain_loop(); });
}
HostXfer::~HostXfer() {
    abort();
    {
        std::lock_guard<std::mutex> g(m_);
        stop_ = true;
    }
    cv_task_.notify_all();
    drainer_.join();
    for (int i = 0; i < SLOTS; ++i) {
        if (pinned_[i]) (void)hipHostFree(pinned_[i]);
        if (ev_[i]) (void)hipEventDestroy(ev_[i]);
    }
    for (hipEvent_t e : ready_pool_) (void)hipEventDestroy(e);
    for (hipEvent_t e : session_events_) (void)hipEventDestroy(e);
    if (copy_) (void)hipStreamDestroy(copy_);
}

void HostXfer::begin(const XferPlan& p, const double* d_slab, double* vals) {
    plan_ = &p;
    d_slab_ = d_slab;
    vals_ = vals;
    {
        std::lock_guard<std::mutex> g(m_);
        error_ = nullptr;
    }
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

void HostXfer::submit(int64_t r0, int64_t r1, hipStream_t producer) {
    if (!plan_ || r0 >= r1) return;
    hipEvent_t e;
    if (!ready_pool_.empty()) {
        e = ready_pool_.back();
        ready_pool_.pop_back();
    } else {
        hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    }
    session_events_.push_back(e);
    hip_check(hipEventRecord(e, producer), "hipEventRecord (slab slice ready)");
    {
        std::lock_guard<std::mutex> g(m_);
        tasks_.push_back(Task{r0, r1, e});
        working_ = true;
    }
    cv_task_.notify_all();
}

void HostXfer::finish() {
    {
        std::unique_lock<std::mutex> g(m_);
        cv_idle_.wait(g, [this] { return !working_; });
    }
    pool_.wait_all();
    for (hipEvent_t e : session_events_) ready_pool_.push_back(e);
    session_events_.clear();
    plan_ = nullptr;
    std::exception_ptr err;
    {
        std::lock_guard<std::mutex> g(m_);
        err = error_;
        error_ = nullptr;
    }
    if (err) std::rethrow_exception(err);
}

void HostXfer::abort() noexcept {
    try {
        finish();
    } catch (...) {
    }
}

void HostXfer::drain_loop() {
    (void)hipSetDevice(device_);
    for (;;) {
        Task t{};
        bool have = false;
        {
            std::unique_lock<std::mutex> g(m_);
            if (tasks_.empty() && inflight_.empty()) {
                working_ = false;
                cv_idle_.notify_all();
                cv_task_.wait(g, [this] { return stop_ || !tasks_.empty(); });
                if (tasks_.empty()) return;  // stop
            }
            if (!tasks_.empty()) {
                t = tasks_.front();
                tasks_.pop_front();
                have = true;
            }
        }
        try {
            bool failed;
            {
                std::lock_guard<std::mutex> g(m_);
                failed = (bool)error_;
            }
            if (failed) {
                inflight_.clear();  // a session that failed only has to come to rest
                continue;
            }
            if (have) run_task(t);
            else retire_one();
        } catch (...) {
            std::lock_guard<std::mutex> g(m_);
            if (!error_) error_ = std::current_exception();
            inflight_.clear();
        }
    }
}

void HostXfer::wait_slot(int slot) {
    std::unique_lock<std::mutex> g(m_);
    cv_slot_.wait(g, [&] { return busy_[slot].load() == 0; });
}

void HostXfer::run_task(const Task& t) {
    const XferPlan& p = *plan_;
    hip_check(hipStreamWaitEvent(copy_, t.ready, 0), "hipStreamWaitEvent (copy stream)");
    launch_pack_runs(copy_, d_slab_, p.d_start + t.r0, p.d_len + t.r0, p.d_poff + t.r0, t.r1 - t.r0, p.d_packed);
    hip_check(hipGetLastError(), "pack launch");
    int64_t r = t.r0;
    while (r < t.r1) {
        // as many whole runs as fit a ring slot
        const int64_t base = p.pk_poff[(size_t)r];
        int64_t e = (std::upper_bound(p.pk_poff.begin() + r, p.pk_poff.begin() + t.r1 + 1, base + CHUNK_DOUBLES) - p.pk_poff.begin()) - 1;
        if (e <= r) throw std::runtime_error("HostXfer: run longer than a ring slot");
        const int slot = (int)(piece_seq_++ % SLOTS);
        wait_slot(slot);  // the piece that used this slot SLOTS pieces ago has been scattered
        hip_check(hipMemcpyAsync(pinned_[slot], p.d_packed + base, sizeof(double) * (size_t)(p.pk_poff[(size_t)e] - base), hipMemcpyDeviceToHost, copy_),
                  "D2H piece");
        hip_check(hipEventRecord(ev_[slot], copy_), "hipEventRecord");
        inflight_.push_back(Piece{slot, r, e});
        while (inflight_.size() > 1) retire_one();  // scatter the previous piece while this one is on the wire
        r = e;
    }
}

void HostXfer::retire_one() {
    const Piece pc = inflight_.front();
    inflight_.pop_front();
    hip_check(hipEventSynchronize(ev_[pc.slot]), "hipEventSynchronize (D2H piece)");
    const XferPlan& p = *plan_;
    const int64_t nr = pc.r1 - pc.r0;
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(pool_.size(), (nr + 255) / 256));
    const int64_t per = (nr + parts - 1) / parts;
    const int64_t base = p.pk_poff[(size_t)pc.r0];
    const double* src = pinned_[pc.slot];
    double* vals = vals_;
    busy_[pc.slot] = parts;
    for (int q = 0; q < parts; ++q) {
        const int64_t a = std::min(pc.r1, pc.r0 + (int64_t)q * per), b = std::min(pc.r1, a + per);
        const int slot = pc.slot;
        pool_.submit([this, &p, vals, a, b, base, src, slot] {
            for (int64_t r = a; r < b; ++r)
                memcpy(vals + p.pk_start[(size_t)r], src + (p.pk_poff[(size_t)r] - base), sizeof(double) * (size_t)p.pk_len[(size_t)r]);
            if (busy_[slot].fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> g(m_);
                cv_slot_.notify_all();
            }
        });
    }
}

}  // namespace dto
