// lsf_rccl.hpp -- librccl.so loaded on first use (lsfm::Rccl of lsf_multi.hpp): no link-time dependency on RCCL.  Included by lsf_api.hip.
#pragma once

namespace lsfm {
bool Rccl::load(std::string* err)
{
    if (lib) return true;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) {
        *err = std::string("RCCL transport requested but librccl.so cannot be loaded: ") + dlerror();
        return false;
    }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    GetVersion = (int (*)(int*))sym("ncclGetVersion");
    GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
    CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    GroupStart = (int (*)())sym("ncclGroupStart");
    GroupEnd = (int (*)())sym("ncclGroupEnd");
    Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
    Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
    if (!GetErrorString || !CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv) {
        *err = "librccl.so lacks a symbol of the point-to-point API";
        dlclose(lib);
        lib = nullptr;
        return false;
    }
    if (GetVersion) (void)GetVersion(&version);
    return true;
}
Rccl::~Rccl()
{
    for (void* c : comms)
        if (c && CommDestroy) (void)CommDestroy(c);
    comms.clear();
    // the library stays loaded: RCCL keeps threads and device state of its own
}
} // namespace lsfm
