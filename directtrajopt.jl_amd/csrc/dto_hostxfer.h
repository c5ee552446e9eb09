// dto_hostxfer.h -- device-to-host hand-off of a value slab for the host-pointer entry points (what Ipopt calls,
// src/solvers/evaluator.jl:368-401): only the entries that can differ from call to call cross PCIe.
//
// A Jacobian slab is half constants (the identity / zero z_{k+1} halves of the integrator blocks, structural zeros that
// `_fill_jacobian_values!` still has to store), a Hessian slab ~99 % structural zeros.  The plan lists the VARIABLE runs of a
// slab (everything a kernel may write a call-dependent value to) and the constant non-zero entries; per call the GPU packs
// the variable runs into one dense buffer, which is copied in chunks through a small pinned ring, while host threads fill the
// caller's vector: constants during the GPU's compute, variable runs as their chunks arrive.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace dto {

struct XferPlan {
    int64_t total = 0;                   // slab length (doubles)
    std::vector<int64_t> start, len;     // variable runs, ascending, disjoint (local slab positions)
    std::vector<int64_t> poff;           // packed offset of each run (prefix sums), size runs + 1
    std::vector<int64_t> one_pos;        // constant non-zero entries ...
    std::vector<double> one_val;         // ... and their values (everything else outside the runs is 0.0)
    std::vector<size_t> chunk_run;       // run index at which each D2H chunk starts, size chunks + 1
    int64_t* d_start = nullptr;          // device copies (owned by the engine handle)
    int64_t* d_len = nullptr;
    int64_t* d_poff = nullptr;
    double* d_packed = nullptr;
    int64_t packed_total() const { return poff.empty() ? 0 : poff.back(); }
    bool usable() const { return total > 0 && !poff.empty(); }
    // finish a plan whose start/len are filled: merge adjacent runs, prefix sums, chunk boundaries of at most `chunk` doubles
    void finalize(int64_t chunk_doubles);
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
    // phase 1 (before / while the GPU computes): zero the gaps between the variable runs and write the constant entries
    void fill_constants_async(const XferPlan& p, double* vals);
    // phase 2 (the slab is complete on `st` when the enqueued work has run): pack, copy, scatter; returns when `vals` is whole
    void fetch(const XferPlan& p, const double* d_slab, double* vals, hipStream_t st);
    static constexpr int64_t CHUNK_DOUBLES = 8 << 20;  // 64 MB per ring slot
private:
    static constexpr int SLOTS = 4;
    HostPool pool_;
    double* pinned_[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
};

// packed[poff[r] + i] = slab[start[r] + i]  (dto_kernels.hip)
void launch_pack_runs(hipStream_t st, const double* slab, const int64_t* start, const int64_t* len, const int64_t* poff,
                      int64_t n_runs, double* packed);

}  // namespace dto
