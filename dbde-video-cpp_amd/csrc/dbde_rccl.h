// dbde_rccl.h -- librccl, opened at run time (dlopen "librccl.so.1": in a PyTorch process that is the copy torch has
// loaded, so the process holds ONE RCCL), which keeps single-GPU users of libdbde_hip.so free of the dependency.
// Shared by the two exchange steps of the multi-GPU path: dbde_gather.cpp (encode side) and dbde_scatter.cpp (decode side).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace dbde_rccl {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string err;
};

inline Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.err = std::string("librccl not found: ") + dlerror(); return; }
#define DBDE_RCCL_SYM(field, name)                                             \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, #name));     \
    if (!r.field) { r.err = "librccl lacks " #name; r.handle = nullptr; return; }
        DBDE_RCCL_SYM(GetUniqueId, ncclGetUniqueId)
        DBDE_RCCL_SYM(CommInitRank, ncclCommInitRank)
        DBDE_RCCL_SYM(CommDestroy, ncclCommDestroy)
        DBDE_RCCL_SYM(AllGather, ncclAllGather)
        DBDE_RCCL_SYM(Broadcast, ncclBroadcast)
        DBDE_RCCL_SYM(Send, ncclSend)
        DBDE_RCCL_SYM(Recv, ncclRecv)
        DBDE_RCCL_SYM(GroupStart, ncclGroupStart)
        DBDE_RCCL_SYM(GroupEnd, ncclGroupEnd)
        DBDE_RCCL_SYM(GetErrorString, ncclGetErrorString)
        DBDE_RCCL_SYM(GetVersion, ncclGetVersion)
#undef DBDE_RCCL_SYM
    });
    return r.handle ? &r : nullptr;
}

}  // namespace dbde_rccl
