// RCCL, bound at run time.  The data-parallel epoch loop (rcn_hip_dp_*) needs five RCCL entry points; resolving them
// with dlopen keeps librcn_hip.so free of a link-time dependency on librccl (single-GPU users never load it) and, in a
// process that already loaded RCCL (PyTorch ships its own copy under the same SONAME), binds to THAT copy instead of a
// second one.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only

#include <mutex>
#include <string>

namespace rcn {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    bool ok = false;

    static Rccl& get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, [] { r.bind(); });
        return r;
    }

  private:
    void bind() {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return; }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(h, n);
            if (!p && err.empty()) err = std::string("librccl lacks ") + n;
            return p;
        };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        ok = err.empty();
    }
};

}  // namespace rcn
