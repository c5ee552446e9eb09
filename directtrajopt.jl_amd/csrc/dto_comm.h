// dto_comm.h -- the engine's collectives (SURVEY.md §8e, §8b "Ownership": the engine owns the RCCL communicators).
//
// One process per GPU, each with a handle on its knot shard.  Every rank's Jacobian / Hessian / gradient output is ONE
// contiguous slab of the global value vector, so the only exchange step the path has is the gather of those slabs for a
// consumer that wants the whole vector on every GPU (MadNLP-GPU's KKT assembly; BASELINE configs[3]): one in-place
// ncclAllGather over xGMI on a vector allocated with a little padding, or -- for layouts without that shape -- one grouped
// set of in-place broadcasts.  RCCL is bound at run time (dlopen of librccl.so.1 at the first dto_comm_* call): a
// single-GPU user of the engine never loads it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

namespace dto {

struct Slab {
    int64_t lo, len;  // position and length inside the global vector
};

// How the slabs of one value vector lie across the communicator's ranks, and how a caller allocates the vector so that one
// equal-size all-gather moves every slab in place.
//
// The slabs of the interior ranks are equally long (n) and back to back; only the first and the last rank's are shorter
// (knot 1 has no z_{k+1} half, knot N no own half: a boundary half-block).  With front = n - len_0 doubles in front of the
// vector and back = n - len_last behind it, rank r's slab lies inside chunk r of a buffer of world * n doubles (at the chunk's
// end for rank 0, at its start for the others): ncclAllGather(buffer + rank * n, buffer, n) then moves every slab to its
// place, all links busy at once, no staging copy.  Layouts without that shape (a rank's knots over several handles, ranges
// that leave gaps) are served by one in-place broadcast per rank inside a group call.
struct GatherPlan {
    std::vector<Slab> slabs;  // per rank
    int64_t total = 0;        // length of the global vector
    int64_t n = 0;            // chunk length of the in-place all-gather (0: not applicable)
    int64_t front = 0, back = 0;
    bool in_place = false;
    int64_t padded_len() const { return in_place ? front + total + back : total; }
};
GatherPlan make_gather_plan(const std::vector<Slab>& slabs, int64_t total);

class Comm {
public:
    // ncclGetUniqueId: 128 bytes a caller hands to the other ranks by its own means (MPI, a file, torch.distributed)
    static std::string unique_id(void* out128);
    // ncclCommInitRank on the CURRENT device (collective over the ranks); err gets the text on failure
    static std::unique_ptr<Comm> create(const void* id128, int rank, int world, std::string& err);
    ~Comm();

    int rank() const { return rank_; }
    int world() const { return world_; }
    // every collective returns "" or the error text; all are enqueued on `st`
    std::string all_gather_in_place(double* buffer, int64_t n, hipStream_t st);
    std::string broadcast_slabs(double* full, const std::vector<Slab>& slabs, const std::vector<int>& root, hipStream_t st);
    std::string all_reduce_sum(double* d, int64_t count, hipStream_t st);
    std::string all_gather_i64(const int64_t* dsend, int64_t* drecv, int64_t count, hipStream_t st);

private:
    Comm() = default;
    void* comm_ = nullptr;
    int rank_ = 0, world_ = 1;
};

}  // namespace dto
