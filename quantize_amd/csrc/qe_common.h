// qe_common.h -- shared host/device helpers of the gfx950 quant engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/quant_engine.h"

namespace qe {

// last failing hipError_t on this thread (qe_last_hip_error()).
extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return QE_ERR_HIP;
}

#define QE_HIP_TRY(expr)                                  \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) return ::qe::hip_fail(_e);  \
    } while (0)

// Kernel launches are followed by hipGetLastError() (the reference never checks).
#define QE_LAUNCH_CHECK() QE_HIP_TRY(hipGetLastError())

// Tuning knobs (QE_* environment variables) are read ONCE per process into a snapshot (a conv call used to make 18 getenv
// calls); env_get() answers from it.  qe_debug_reload_env() (not part of the public ABI; quantize_amd.capi.reload_env)
// re-reads the environment -- the test-suite and the A/B tools use it after changing a knob inside a live process.
const char *env_get(const char *name);

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumXCD = 8;

__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace qe
