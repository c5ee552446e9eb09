// dto_comm.cpp -- RCCL-backed collectives of the engine (dto_comm.h).  RCCL is loaded lazily with dlopen: the types come from
// <rccl/rccl.h>, the entry points from the library already in the process (a host that initialised torch.distributed's
// backend "nccl" shares it) or from the ROCm installation next to libamdhip64.
#include "dto_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>

namespace dto {

GatherPlan make_gather_plan(const std::vector<Slab>& slabs, int64_t total) {
    GatherPlan p;
    p.slabs = slabs;
    p.total = total;
    const int world = (int)slabs.size();
    if (world == 0) return p;
    int64_t n = 0, sum = 0;
    for (auto& s : slabs) { n = std::max(n, s.len); sum += s.len; }
    if (n == 0 || sum != total || slabs[0].lo != 0) return p;
    for (int r = 1; r < world; ++r) {
        const Slab& s = slabs[r];
        if (s.lo != slabs[0].lo + slabs[0].len + (int64_t)(r - 1) * n) return p;
        if (s.len > n || (r < world - 1 && s.len != n)) return p;
    }
    p.n = n;
    p.front = n - slabs[0].len;
    p.back = world > 1 ? n - slabs[world - 1].len : 0;
    // a single rank, or a first slab that is the longest one with shorter ones behind it: world * n must cover the vector
    if (p.front + total + p.back != (int64_t)world * n) return p;
    p.in_place = true;
    return p;
}

namespace {

struct Rccl {
    void* lib = nullptr;
    std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.err = std::string("RCCL is not available (dlopen librccl.so.1: ") + dlerror() + ")";
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(r.lib, s);
            if (!p && r.err.empty()) r.err = std::string("RCCL: symbol missing: ") + s;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

std::string check(ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return "";
    return std::string(what) + " failed: " + rccl().GetErrorString(rc);
}

}  // namespace

std::string Comm::unique_id(void* out128) {
    Rccl& R = rccl();
    if (!R.err.empty()) return R.err;
    static_assert(sizeof(ncclUniqueId) == 128, "dto_engine.h hands the id over as DTO_COMM_ID_BYTES = 128 bytes");
    return check(R.GetUniqueId(reinterpret_cast<ncclUniqueId*>(out128)), "ncclGetUniqueId");
}

std::unique_ptr<Comm> Comm::create(const void* id128, int rank, int world, std::string& err) {
    Rccl& R = rccl();
    if (!R.err.empty()) { err = R.err; return nullptr; }
    if (world < 1 || rank < 0 || rank >= world) { err = "dto_comm_create: bad rank / world"; return nullptr; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    err = check(R.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    if (!err.empty()) return nullptr;
    std::unique_ptr<Comm> out(new Comm());
    out->comm_ = c;
    out->rank_ = rank;
    out->world_ = world;
    return out;
}

Comm::~Comm() {
    if (comm_) (void)rccl().CommDestroy((ncclComm_t)comm_);
}

std::string Comm::all_gather_in_place(double* buffer, int64_t n, hipStream_t st) {
    // in place: the send buffer is the rank's own chunk of the receive buffer
    return check(rccl().AllGather(buffer + (int64_t)rank_ * n, buffer, (size_t)n, ncclDouble, (ncclComm_t)comm_, st), "ncclAllGather");
}

std::string Comm::broadcast_slabs(double* full, const std::vector<Slab>& slabs, const std::vector<int>& root, hipStream_t st) {
    Rccl& R = rccl();
    std::string e = check(R.GroupStart(), "ncclGroupStart");
    if (!e.empty()) return e;
    for (size_t i = 0; i < slabs.size(); ++i) {
        if (slabs[i].len <= 0) continue;
        double* p = full + slabs[i].lo;
        e = check(R.Broadcast(p, p, (size_t)slabs[i].len, ncclDouble, root[i], (ncclComm_t)comm_, st), "ncclBroadcast");
        if (!e.empty()) { (void)R.GroupEnd(); return e; }
    }
    return check(R.GroupEnd(), "ncclGroupEnd");
}

std::string Comm::all_reduce_sum(double* d, int64_t count, hipStream_t st) {
    return check(rccl().AllReduce(d, d, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)comm_, st), "ncclAllReduce");
}

std::string Comm::all_gather_i64(const int64_t* dsend, int64_t* drecv, int64_t count, hipStream_t st) {
    return check(rccl().AllGather(dsend, drecv, (size_t)count, ncclInt64, (ncclComm_t)comm_, st), "ncclAllGather");
}

}  // namespace dto
