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
    long last_ws_other = 0;   // ... and its number of pixels that are neither the image's minimum nor its maximum
    hipEvent_t edge_event = nullptr;   // tip_wait_stream / tip_stream_wait_tip
    void *prep_ws = nullptr;           // order-statistic state of tip_unet_prepare_f64_dev (used on the caller's stream only)
    void *pin_buf = nullptr;           // pinned host staging (the watershed's marker-order stage: counts down, pop order up)
    size_t pin_bytes = 0;
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> free_events;
};

Ctx &ctx();                       // lazily initialised for device 0 unless tip_init() chose another
int fail(int code, const char *fmt, ...);
void *ws_alloc(size_t bytes);     // nullptr on failure (error text set)
void *pinned_scratch(size_t bytes);   // this thread's pinned host staging buffer, grown as needed (nullptr on failure)
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

// Process-wide tuning / test hooks.  Filled ONCE from the environment (TIP_* variables) when the library is loaded and
// changed afterwards only through tip_set_tuning(): entry points read plain ints, never the environment.
struct Tuning {
    int ws_ties = 1;            // TIP_WS_TIES: 1 = exact (serial (value, age) replay on tie landscapes), 0 = fast (device order)
    int ws_tile = -1;           // TIP_WS_TILE: everyday tile flavour (-1: the built-in default)
    int ws_open_a = -1, ws_open_b = -1;   // TIP_WS_OPEN=a,b
    int ws_cert_from = -1;      // TIP_WS_CERT_FROM
    int ws_no_skip = 0;         // TIP_WS_NO_SKIP
    int ws_lds_pad = 0;         // TIP_WS_LDS_PAD
    int ws_debug = 0;           // TIP_WS_DEBUG
    int ws_no_endgame = 0, ws_no_wide = 0;   // TIP_WS_NO_ENDGAME / TIP_WS_NO_WIDE (exercise the stall machinery)
    int mfma_blocks_per_cu = 2; // TIP_MFMA_BLOCKS_PER_CU
    int project_generic = 0, project_unfused_preblur = 0, project_unfused_mask = 0;
    int project_exact_score = 0, project_debug = 0;
    int fast_cfg_y = -1, fast_cfg_x = -1;    // TIP_FAST_CFG=y,x
    int unet_tail_unfused = 0;  // TIP_UNET_TAIL_UNFUSED: the tail's morphology as separate rank-filter launches (tests)
    int unet_xcd_map = 1;       // TIP_UNET_XCD_MAP: the channel blocks of one pixel tile side by side on one XCD (0: all workgroups in flight on one channel block)
    int unet_spb = 3;           // TIP_UNET_SPB: steps per barrier of the 3x3 16-row convolution kernel (1, 2 or 3)
    int uf_one_level = 0;       // TIP_UF_ONE_LEVEL: the one-level union-find (global atomics only) instead of tiles in LDS + borders (tests)
    int mb_small = -1;          // TIP_MB_SMALL: generations of the two-valued flood of at most this many pixels run in one workgroup (0: never; -1: the built-in default)
    int mb_batch = -1;          // TIP_MB_BATCH: generations queued between two looks at the device state (-1: the built-in default)
    int unet_tile8 = -1;        // TIP_UNET_TILE8: the U-Net convolution's tile rows: 1 = 8 everywhere, 0 = 16 where the grid allows, -1 (default) = 16 except for 3x3 layers with <= 128 input channels
};
const Tuning &tuning();

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
