// Shared host-side plumbing for the impdar HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/impdar_hip.h"

struct ncclComm;

struct impdar_ctx {
    int device = 0;
    hipStream_t stream = nullptr;   // compute stream (migration kernels, copies)
    hipStream_t aux = nullptr;      // producer stream: prep / table / all-gather of the NEXT radargram
    // RCCL communicator (null until impdar_comm_init)
    ncclComm *comm = nullptr;
    int rank = 0;
    int nranks = 1;
};

void impdar_set_error(const char *fmt, ...);

#define IMPDAR_HIP_CHECK(expr)                                                        \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            impdar_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                             hipGetErrorString(_e));                                  \
            return IMPDAR_ERR_HIP;                                                    \
        }                                                                             \
    } while (0)

#define IMPDAR_ARG_CHECK(cond, ...)                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            impdar_set_error(__VA_ARGS__);                                            \
            return IMPDAR_ERR_ARG;                                                    \
        }                                                                             \
    } while (0)

static inline size_t impdar_dtype_size(int dtype) { return dtype == IMPDAR_F64 ? 8 : 4; }

// RAII device buffer bound to a context's device.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};
