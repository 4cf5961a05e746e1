// tip_internal.h -- shared plumbing of libtissue_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/tissue_hip.h"

namespace tip {

struct ProfRec { const char *name; hipEvent_t e0, e1; };

// One context per calling thread: the reference is driven from Qt worker threads (gui.py:1821-2137),
// ctypes releases the GIL, so entry points must be re-entrant.
struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    // workspace pool: cached device blocks, best-fit reuse
    struct Block { void *p; size_t bytes; bool used; };
    std::vector<Block> pool;
    int last_ws_labels = 0;   // marker count of this thread's last watershed (tip_last_watershed_labels)
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> free_events;
};

Ctx &ctx();                       // lazily initialised for device 0 unless tip_init() chose another
int fail(int code, const char *fmt, ...);
void *ws_alloc(size_t bytes);     // nullptr on failure (error text set)
void ws_free(void *p);

struct WsGuard {                  // frees workspaces at scope exit
    std::vector<void *> ptrs;
    template <typename T> T *get(size_t count) {
        void *p = ws_alloc(count * sizeof(T) ? count * sizeof(T) : 16);
        if (p) ptrs.push_back(p);
        return (T *)p;
    }
    ~WsGuard() { for (void *p : ptrs) ws_free(p); }
};

void prof_begin(const char *name);
void prof_end();

#define TIP_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return tip::fail(TIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                                 \
    } while (0)

// launch a kernel on the context stream, timed with HIP events when profiling is on
#define TIP_LAUNCH(name, kernel, grid, block, shmem, ...)                                  \
    do {                                                                                   \
        tip::prof_begin(name);                                                             \
        hipLaunchKernelGGL(kernel, grid, block, shmem, tip::ctx().stream, __VA_ARGS__);    \
        tip::prof_end();                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess)                                                             \
            return tip::fail(TIP_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e__));  \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Gaussian taps passed by value in the kernel-argument segment (scalar loads, no device copy).
struct Taps {
    double w[256];
    int n;
};
struct TapsF {   // float32 copy for the certified fast score passes
    float w[256];
    int n;
};

int make_taps(Taps &t, const double *w, int n);  // validates odd + symmetric (scipy's symmetric branch)
int libm_taps(double sigma, double truncate, double *w, int cap);

}  // namespace tip
