// Shared host/device helpers for libnns_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../include/nns.h"

#define NNS_API extern "C" __attribute__((visibility("default")))

namespace nns {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

// Every launch function ends with this: surfaces launch-configuration errors without a host sync.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return NNS_OK;
}

// bufs[i][0 .. bytes[i]) = 0 for every i, ONE launch per 24 buffers (csrc/optim_kernels.hip); entries with bytes <= 0 are skipped
int zero_buffers(void* const* bufs, const long* bytes, int count, hipStream_t s);

inline bool field_args_ok(int batch, int nx, int ny) { return batch >= 1 && nx >= 3 && ny >= 3; }

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kNumXCD = 8;         // MI355X: 8 XCDs, blocks dealt round-robin (blockIdx % 8 = XCD group)

// XCD-aware remap (bijective for any grid size): blocks that share an XCD (same id % 8) get a
// CONTIGUOUS range of logical tiles, so neighbouring tiles share that XCD's L2 (halo rows).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned xcd = bid % kNumXCD, q = nblocks / kNumXCD, r = nblocks % kNumXCD;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / kNumXCD;
}

// Device copy of a boundary list, in the field's arithmetic type.
template <typename T>
struct BcListDev {
    int n;
    int kind[NNS_MAX_BC];
    int side[NNS_MAX_BC];
    T value[NNS_MAX_BC];
    T dx[NNS_MAX_BC];
    T dy[NNS_MAX_BC];
};

template <typename T>
inline int make_bc_dev(const nns_bc_list* h, BcListDev<T>& d) {
    if (!h) return fail(NNS_ERR_INVALID_ARG, "bc list is NULL");
    if (h->n < 0 || h->n > NNS_MAX_BC) return fail(NNS_ERR_INVALID_ARG, "bc list length %d not in [0,%d]", h->n, NNS_MAX_BC);
    d.n = h->n;
    for (int i = 0; i < NNS_MAX_BC; ++i) {
        const bool live = i < h->n;
        d.kind[i] = live ? h->kind[i] : 0;
        d.side[i] = live ? h->side[i] : 0;
        d.value[i] = live ? (T)h->value[i] : (T)0;
        d.dx[i] = live ? (T)h->dx[i] : (T)0;
        d.dy[i] = live ? (T)h->dy[i] : (T)0;
        if (live && (h->kind[i] < 0 || h->kind[i] > 1 || h->side[i] < 0 || h->side[i] > 3))
            return fail(NNS_ERR_INVALID_ARG, "bc entry %d: kind %d / side %d invalid", i, h->kind[i], h->side[i]);
    }
    return NNS_OK;
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, I1)
template <int I0, int I1, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, I1>(f); }
}

}  // namespace nns
