// dto_hostxfer.h -- device-to-host hand-off of a value slab for the host-pointer entry points (what Ipopt calls,
// src/solvers/evaluator.jl:368-401): only the entries that can differ from call to call cross PCIe, and they start
// crossing while the GPU is still computing.
//
// A Jacobian slab is half constants (the identity / zero z_{k+1} halves of the integrator blocks, structural zeros that
// `_fill_jacobian_values!` still has to store), a Hessian slab ~99 % structural zeros.  The plan lists the VARIABLE runs of a
// slab (everything a kernel may write a call-dependent value to) and the constant non-zero entries.  Per call host threads
// fill the constants into the caller's vector while the GPU computes; the variable runs are packed on the device into one
// dense buffer and copied in pieces through a small pinned ring, scattered by the same host threads as the pieces arrive.
//
// EARLY runs: the -E_k blocks of a lone bilinear integrator are final as soon as the propagator chain has finished the chunk
// of intervals they belong to -- long before the generator sweep and the assembly kernels have run.  They come first in the
// packed order, interval by interval, so the engine can hand over the slice of a chain chunk right behind that chunk's last
// squaring (`submit`): its packing and its PCIe copy run on a second stream next to the following chunk's GEMMs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace dto {

struct XferPlan {
    int64_t total = 0;                   // slab length (doubles)
    // --- filled by the builder (local slab positions; runs ascending and disjoint) ---
    std::vector<int64_t> start, len;     // variable runs
    std::vector<int32_t> early;          // per run: local interval whose chain chunk finishes it, or -1 (empty: none are early)
    std::vector<int64_t> one_pos;        // constant non-zero entries ...
    std::vector<double> one_val;         // ... and their values (everything else outside the runs is 0.0)
    // --- derived by finalize ---
    std::vector<int64_t> pk_start, pk_len, pk_poff;  // the runs in PACKED order (early runs first, by interval), poff = prefix sums
    std::vector<int64_t> early_off;      // packed-order index of the first early run of local interval i (size n_int + 1)
    bool built = false;
    int64_t* d_start = nullptr;          // device copies of pk_* (owned by the engine handle)
    int64_t* d_len = nullptr;
    int64_t* d_poff = nullptr;
    double* d_packed = nullptr;
    int64_t n_runs() const { return (int64_t)pk_start.size(); }
    int64_t n_early() const { return early_off.empty() ? 0 : early_off.back(); }
    int64_t packed_total() const { return pk_poff.empty() ? 0 : pk_poff.back(); }
    bool usable() const { return total > 0 && built && d_packed != nullptr; }
    // finish a plan whose start/len(/early) are filled: merge touching runs of the same class, packed order, prefix sums
    void finalize(int64_t n_int, int64_t max_run);
};

class HostPool {
public:
    explicit HostPool(int n_threads);
    ~HostPool();
    void submit(std::function<void()> job);
    void wait_all();
    int size() const { return (int)workers_.size(); }

private:
    void loop();
    std::vector<std::thread> workers_;
    std::deque<std::function<void()>> jobs_;
    std::mutex m_;
    std::condition_variable cv_job_, cv_done_;
    int pending_ = 0;
    bool stop_ = false;
};

class HostXfer {
public:
    HostXfer();
    ~HostXfer();
    // One session per host-pointer call: begin -> submit* -> finish (or abort).
    // begin: host threads start zeroing the gaps between the variable runs and writing the constant entries.
    void begin(const XferPlan& p, const double* d_slab, double* vals);
    // The runs [r0, r1) of the packed order are final once the work enqueued on `producer` so far has run: pack, copy and
    // scatter them behind it (asynchronous; another thread drives the copy stream).
    void submit(int64_t r0, int64_t r1, hipStream_t producer);
    // Returns when `vals` is whole; rethrows what went wrong on the way.
    void finish();
    // Error path: joins everything that is in flight, swallows secondary errors.
    void abort() noexcept;
    static constexpr int64_t CHUNK_DOUBLES = 8 << 20;  // 64 MB per ring slot

private:
    static constexpr int SLOTS = 4;
    struct Task { int64_t r0, r1; hipEvent_t ready; };
    struct Piece { int slot; int64_t r0, r1; };
    void drain_loop();
    void run_task(const Task& t);
    void retire_one();
    void wait_slot(int slot);

    HostPool pool_;
    int device_ = 0;
    hipStream_t copy_ = nullptr;
    double* pinned_[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    std::unique_ptr<std::atomic<int>[]> busy_;  // scatter jobs of a slot still running
    std::vector<hipEvent_t> ready_pool_;         // events of finished sessions, reused
    std::vector<hipEvent_t> session_events_;     // events handed to the drainer in this session
    // session
    const XferPlan* plan_ = nullptr;
    const double* d_slab_ = nullptr;
    double* vals_ = nullptr;
    // drainer
    std::thread drainer_;
    std::mutex m_;
    std::condition_variable cv_task_, cv_idle_, cv_slot_;
    std::deque<Task> tasks_;
    std::deque<Piece> inflight_;  // touched by the drainer only
    bool working_ = false, stop_ = false;
    int64_t piece_seq_ = 0;
    std::exception_ptr error_;
};

// packed[poff[r] + i] = slab[start[r] + i]  (dto_kernels.hip)
void launch_pack_runs(hipStream_t st, const double* slab, const int64_t* start, const int64_t* len, const int64_t* poff,
                      int64_t n_runs, double* packed);

}  // namespace dto
