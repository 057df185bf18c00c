// chorin_fd._get_pressure SOR loop (src/chorin_fd/simulate.py:190-200) on gfx950.
//
// The reference sweeps the interior in lexicographic order, IN PLACE (Gauss-Seidel + over-
// relaxation).  Point (i,j) therefore sees (i-1,j),(i,j-1) of the current sweep and (i+1,j),
// (i,j+1) of the previous one.  All points of an anti-diagonal d = i+j are independent and see
// exactly those values if the diagonals are visited in increasing d: the wavefront order is
// bitwise identical to the reference order (oracle/chorin_fd.py: sor_sweep_wavefront).
//
// One workgroup per grid (the solve is sequential across fronts: "replicas only" across GPUs).
// Parallelism beyond one front comes from keeping several SWEEPS in flight: wave w runs sweep
// s0+w, two fronts behind wave w-1 (the minimum lag for the in-place dependency, one workgroup
// barrier per step), so a batch of W sweeps costs nfronts + 2(W-1) steps instead of W*nfronts.
//
// The data-dependent stop (first sweep with max|p - pPrev| <= tol, :190,:198) is kept exact:
// every sweep records its own max update; if a sweep inside a batch meets the tolerance, the
// grid is restored from the snapshot taken at the start of the batch and exactly that many
// sweeps are replayed.  Compiled with -ffp-contract=off (reference operation order).
#include "nns_common.h"
#include "sor_device.h"
#include <cmath>

using namespace nns;
using namespace nns::sorlex;

namespace {

template <typename T, bool IN_LDS>
__global__ __launch_bounds__(kSorThreads) void sor_kernel(T* __restrict__ p, const T* __restrict__ C, T* __restrict__ info,
                                                           T* __restrict__ snap, const T* __restrict__ hint, int nx, int ny, int max_sweeps, SorK<T> k) {
    // all LDS in the dynamic region (16-byte aligned carve): [errs kSorBatch x 8 B][stop][pad][p][C]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* errs = reinterpret_cast<T*>(smem_raw);
    int* s_stop_p = reinterpret_cast<int*>(smem_raw + kSorBatch * 8);
    const int n = nx * ny, tid = threadIdx.x;
    T* pg = p + (size_t)blockIdx.x * n;
    const T* cg = C + (size_t)blockIdx.x * n;
    T* sg = snap + (size_t)blockIdx.x * n;
    T* pw = pg;
    const T* cw = cg;
    if (IN_LDS) {
        T* pl = reinterpret_cast<T*>(smem_raw + kSorHdr);
        T* cl = pl + n;
        for (int c = tid; c < n; c += kSorThreads) { pl[c] = pg[c]; cl[c] = cg[c]; }
        pw = pl; cw = cl;
        __syncthreads();
    }
    int done;
    T err;
    const int expect = hint ? (int)hint[2 * blockIdx.x] : 0;       // read before info is written: hint may BE the info buffer of the previous solve
    sor_solve<T, IN_LDS>(pw, cw, sg, nx, ny, max_sweeps, expect, k, errs, s_stop_p, done, err);
    if (IN_LDS) for (int c = tid; c < n; c += kSorThreads) pg[c] = pw[c];
    if (tid == 0) { info[2 * blockIdx.x] = (T)done; info[2 * blockIdx.x + 1] = err; }
}

template <typename T>
int sor(T* p, const T* C, T* info, const T* hint, void* work, int batch, int nx, int ny, double dx, double dy, double beta, double tol,
        int max_sweeps, hipStream_t s) {
    if (!p || !C || !info || !work || !field_args_ok(batch, nx, ny) || max_sweeps < 0)
        return fail(NNS_ERR_INVALID_ARG, "fd_sor: bad args (batch=%d nx=%d ny=%d max_sweeps=%d)", batch, nx, ny, max_sweeps);
    const SorK<T> k = make_sor_k<T>(dx, dy, beta, tol);
    const size_t lds = sor_lds_bytes(nx, ny, sizeof(T));
    T* snap = reinterpret_cast<T*>(work);
    if (lds <= kSorLdsMax) {
        static bool attr = false;                     // set once to the largest size used (keeps the launch path free of
        if (!attr) {                                  // non-stream API calls, e.g. under hipGraph capture)
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sor_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSorLdsMax);
            if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "fd_sor: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr = true;
        }
        hipLaunchKernelGGL((sor_kernel<T, true>), dim3(batch), dim3(kSorThreads), lds, s, p, C, info, snap, hint, nx, ny, max_sweeps, k);
    } else {
        hipLaunchKernelGGL((sor_kernel<T, false>), dim3(batch), dim3(kSorThreads), kSorHdr, s, p, C, info, snap, hint, nx, ny, max_sweeps, k);
    }
    return check_launch("fd_sor");
}


// ------------------------------------------------------------------------------------------------------------------
// Red-black SOR (oracle: get_pressure_redblack): the same update formula, relaxation
// factor, stopping rule and sweep cap as the reference's loop, but points with (i + j) even are relaxed first, then the
// odd ones.  A half-sweep only reads the other colour, so it is fully parallel: one workgroup per grid, every thread
// a few points, two barriers per sweep (the lexicographic order needs nx + ny fronts per sweep), p and C in LDS.  Grids
// that do not fit LDS take the chained chip-wide half-sweep launches further down (sor_redblack).  The error reduction
// is exact (max), so either way the result is bitwise the oracle's (no FMA contraction).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kRbThreads = 1024;

template <typename T>
__global__ __launch_bounds__(kRbThreads) void sor_redblack_kernel(T* __restrict__ p, const T* __restrict__ C, T* __restrict__ info,
                                                                   int nx, int ny, int max_sweeps, SorK<T> k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ T red[kRbThreads / kWave];
    __shared__ T err_sh;
    const int tid = threadIdx.x, n = nx * ny;
    T* pg = p + (size_t)blockIdx.x * n;
    const T* cg = C + (size_t)blockIdx.x * n;
    T* pw = reinterpret_cast<T*>(smem);                   // p and C live in LDS for the whole solve
    T* cw = pw + n;
    for (int c = tid; c < n; c += kRbThreads) { pw[c] = pg[c]; cw[c] = cg[c]; }
    __syncthreads();
    const int mx = nx - 2, my = ny - 2, hw = (my + 1) / 2, half = mx * hw;     // points of one colour: <= hw per interior row
    T err = (T)1;
    int done = 0;
    while (done < max_sweeps && err > k.tol) {
        T emax = (T)0;
#pragma unroll
        for (int colour = 0; colour < 2; ++colour) {
            for (int q = tid; q < half; q += kRbThreads) {
                const int i = q / hw + 1;
                const int j = 1 + 2 * (q % hw) + ((i + 1 + colour) & 1);        // (i + j) % 2 == colour
                if (j > my) continue;
                const int c = i * ny + j;
                const T old = pw[c];
                const T nw = k.beta * (k.dy2 * pw[c + ny] + k.dy2 * pw[c - ny] + k.dx2 * pw[c + 1] + k.dx2 * pw[c - 1] - cw[c]) / k.den + k.omb * old;
                pw[c] = nw;
                emax = nanmax<T>(emax, fabs(nw - old));
            }
            __syncthreads();
        }
        // workgroup max of |p - pPrev|
        for (int off = kWave / 2; off > 0; off >>= 1) emax = nanmax<T>(emax, __shfl_xor(emax, off));
        if ((tid & (kWave - 1)) == 0) red[tid / kWave] = emax;
        __syncthreads();
        if (tid == 0) {
            T e = red[0];
            for (int w = 1; w < kRbThreads / kWave; ++w) e = nanmax<T>(e, red[w]);
            err_sh = e;
        }
        __syncthreads();
        err = err_sh;
        ++done;
    }
    __syncthreads();
    for (int c = tid; c < n; c += kRbThreads) pg[c] = pw[c];
    if (tid == 0) { info[2 * blockIdx.x] = (T)done; info[2 * blockIdx.x + 1] = err; }
}

// ------------------------------------------------------------------------------------------------------------------
// One red-black HALF-sweep on a row slab (the opt-in sharded pressure solve).  p is
// [nxl][ny] with rows 0 and nxl-1 acting as halo / physical-boundary rows (never written); row i of the slab is global
// row gi0 + i, which fixes the colour of its points.  Multi-workgroup (one thread per colour point), so it also serves
// single-GPU grids too large for the LDS-resident kernel.  max|p_new - p_old| is folded into *err_bits with an
// unsigned atomic max on the IEEE bit pattern (order-preserving for non-negative values; the caller zeroes it).
// ------------------------------------------------------------------------------------------------------------------
template <typename T> struct BitsOf;
template <> struct BitsOf<float> { using U = unsigned int; };
template <> struct BitsOf<double> { using U = unsigned long long; };

// Chained use (red-black solve of grids too large for LDS, sor_redblack below): blockIdx.z = grid of the batch; the
// launches of ALL sweeps are enqueued up front and each one decides on the device whether it still has to run:
// sweep s is active iff the previous sweep's error (prev_bits, null for "always") is > tol -- the reference's
// `while err > tol` -- so no host round trip per sweep.  A skipped sweep marks its own slot NaN, which keeps every
// later sweep skipped whatever the sign of tol.
template <typename T>
__global__ __launch_bounds__(256) void sor_rb_halfsweep_kernel(T* __restrict__ p, const T* __restrict__ C, typename BitsOf<T>::U* __restrict__ err_bits,
                                                                const typename BitsOf<T>::U* __restrict__ prev_bits, long slot_stride,
                                                                int nxl, int ny, int gi0, int colour, SorK<T> k) {
    using U = typename BitsOf<T>::U;
    __shared__ T wave_e[256 / kWave];
    const size_t b = blockIdx.z;
    p += b * (size_t)nxl * ny;
    C += b * (size_t)nxl * ny;
    err_bits += b * slot_stride;
    if (prev_bits) {
        const T prev = __builtin_bit_cast(T, prev_bits[b * slot_stride]);
        if (!(prev > k.tol)) {
            if (colour == 1 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *err_bits = ~(U)0 >> 1;    // quiet NaN pattern
            return;
        }
    }
    const int hw = (ny - 2 + 1) / 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    T e = (T)0;
    if (q < hw) {
        for (int i = blockIdx.y + 1; i <= nxl - 2; i += gridDim.y) {   // interior rows 1 .. nxl-2, a strided share per block
            const int j = 1 + 2 * q + ((gi0 + i + 1 + colour) & 1);     // (global row + j) % 2 == colour
            if (j <= ny - 2) {
                const size_t c = (size_t)i * ny + j;
                const T old = p[c];
                const T nw = k.beta * (k.dy2 * p[c + ny] + k.dy2 * p[c - ny] + k.dx2 * p[c + 1] + k.dx2 * p[c - 1] - C[c]) / k.den + k.omb * old;
                p[c] = nw;
                e = nanmax<T>(e, fabs(nw - old));
            }
        }
    }
    // one atomic per block at most, and none when the running maximum already covers this block's value: same-address
    // atomics serialise in L2 (~11 ns each), which at one per wave cost 30x the sweep's memory time
    for (int off = kWave / 2; off > 0; off >>= 1) e = nanmax<T>(e, __shfl_xor(e, off));
    if ((threadIdx.x & (kWave - 1)) == 0) wave_e[threadIdx.x / kWave] = e;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; ++w) e = nanmax<T>(e, wave_e[w]);
        const U bits = __builtin_bit_cast(U, e);
        if (bits > __hip_atomic_load(err_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(err_bits, bits);
    }
}

template <typename T>
void launch_halfsweep(T* p, const T* C, typename BitsOf<T>::U* err, const typename BitsOf<T>::U* prev, long slot_stride, int batch,
                      int nxl, int ny, int gi0, int colour, const SorK<T>& k, hipStream_t s) {
    const int hw = (ny - 2 + 1) / 2;
    const int gx = (hw + 255) / 256, gy = std::min(nxl - 2, std::max(1, 2048 / gx));
    hipLaunchKernelGGL(sor_rb_halfsweep_kernel<T>, dim3(gx, gy, batch), dim3(256), 0, s, p, C, err, prev, slot_stride, nxl, ny, gi0, colour, k);
}

template <typename T>
int sor_rb_halfsweep(T* p, const T* C, void* err_bits, int nxl, int ny, int gi0, int colour, double dx, double dy, double beta, hipStream_t s) {
    if (!p || !C || !err_bits || nxl < 3 || ny < 3 || (colour != 0 && colour != 1) || gi0 < 0)
        return fail(NNS_ERR_INVALID_ARG, "fd_sor_redblack_halfsweep: bad args (nxl=%d ny=%d gi0=%d colour=%d)", nxl, ny, gi0, colour);
    SorK<T> k{(T)(dx * dx), (T)(dy * dy), (T)(2 * (dx * dx) + 2 * (dy * dy)), (T)beta, (T)(1 - beta), (T)0, (T)0};
    launch_halfsweep<T>(p, C, static_cast<typename BitsOf<T>::U*>(err_bits), nullptr, 0, 1, nxl, ny, gi0, colour, k, s);
    return check_launch("fd_sor_redblack_halfsweep");
}

// The chained form for slabs (nns/slab.py: SlabPressure): the half-sweep runs only if the PREVIOUS sweep's error *prev_bits
// (already max-reduced over the ranks by the caller, on the stream) is > tol, exactly like the single-GPU chain above, so
// a rank can enqueue many sweeps and their halo exchanges without reading anything back.
template <typename T>
int sor_rb_halfsweep_gated(T* p, const T* C, void* err_bits, const void* prev_bits, double tol, int nxl, int ny, int gi0, int colour,
                           double dx, double dy, double beta, hipStream_t s) {
    if (!p || !C || !err_bits || !prev_bits || nxl < 3 || ny < 3 || (colour != 0 && colour != 1) || gi0 < 0)
        return fail(NNS_ERR_INVALID_ARG, "fd_sor_redblack_halfsweep_gated: bad args (nxl=%d ny=%d gi0=%d colour=%d)", nxl, ny, gi0, colour);
    SorK<T> k{(T)(dx * dx), (T)(dy * dy), (T)(2 * (dx * dx) + 2 * (dy * dy)), (T)beta, (T)(1 - beta), (T)tol, (T)0};
    using U = typename BitsOf<T>::U;
    launch_halfsweep<T>(p, C, static_cast<U*>(err_bits), static_cast<const U*>(prev_bits), 0, 1, nxl, ny, gi0, colour, k, s);
    return check_launch("fd_sor_redblack_halfsweep_gated");
}

// slots[b][0] = the bit pattern of 1 (the reference's initial err, :183), slots[b][1 .. cap] = 0
template <typename T>
__global__ void sor_rb_init_kernel(typename BitsOf<T>::U* __restrict__ slots, long n, int per_grid) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = (i % per_grid == 0) ? __builtin_bit_cast(typename BitsOf<T>::U, (T)1) : 0;
}

// info[b] = (sweeps done, err of the last sweep done): walk the chain exactly as the host loop would
template <typename T>
__global__ void sor_rb_finish_kernel(const typename BitsOf<T>::U* __restrict__ slots, T* __restrict__ info, int batch, int cap, T tol) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const typename BitsOf<T>::U* sl = slots + (size_t)b * (cap + 1);
    T err = (T)1;
    int done = 0;
    while (done < cap && err > tol) { err = __builtin_bit_cast(T, sl[done + 1]); ++done; }
    info[2 * b] = (T)done;
    info[2 * b + 1] = err;
}

inline bool rb_fits_lds(int nx, int ny, size_t elem) { return 2 * (size_t)nx * ny * elem <= 150 * 1024; }

template <typename T>
int sor_redblack(T* p, const T* C, T* info, void* work, int batch, int nx, int ny, double dx, double dy, double beta, double tol, int max_sweeps, hipStream_t s) {
    if (!p || !C || !info || !field_args_ok(batch, nx, ny) || max_sweeps < 0)
        return fail(NNS_ERR_INVALID_ARG, "fd_sor_redblack: bad args (batch=%d nx=%d ny=%d max_sweeps=%d)", batch, nx, ny, max_sweeps);
    SorK<T> k{(T)(dx * dx), (T)(dy * dy), (T)(2 * (dx * dx) + 2 * (dy * dy)), (T)beta, (T)(1 - beta), (T)tol, (T)0};
    if (rb_fits_lds(nx, ny, sizeof(T))) {
        const size_t lds = 2 * (size_t)nx * ny * sizeof(T);
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sor_redblack_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSorLdsMax);
            if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "fd_sor_redblack: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr = true;
        }
        hipLaunchKernelGGL((sor_redblack_kernel<T>), dim3(batch), dim3(kRbThreads), lds, s, p, C, info, nx, ny, max_sweeps, k);
        return check_launch("fd_sor_redblack");
    }
    // Larger grids: every half-sweep is a chip-wide launch (one workgroup would leave 255 CUs idle); all 2 * max_sweeps
    // launches are enqueued at once and switch themselves off on the device once a sweep's error is <= tol.
    using U = typename BitsOf<T>::U;
    if (!work) return fail(NNS_ERR_INVALID_ARG, "fd_sor_redblack: a %dx%d grid needs the workspace of nns_fd_sor_redblack_workspace", nx, ny);
    if (batch > 65535) return fail(NNS_ERR_UNSUPPORTED, "fd_sor_redblack: batch %d of large grids exceeds the launch grid (65535)", batch);
    U* slots = static_cast<U*>(work);
    const int per = max_sweeps + 1;
    const long nslots = (long)batch * per;
    hipLaunchKernelGGL(sor_rb_init_kernel<T>, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, s, slots, nslots, per);
    for (int sw = 0; sw < max_sweeps; ++sw)
        for (int colour = 0; colour < 2; ++colour)
            launch_halfsweep<T>(p, C, slots + sw + 1, slots + sw, per, batch, nx, ny, 0, colour, k, s);
    hipLaunchKernelGGL(sor_rb_finish_kernel<T>, dim3((batch + 255) / 256), dim3(256), 0, s, slots, info, batch, max_sweeps, (T)tol);
    return check_launch("fd_sor_redblack");
}

}  // namespace

NNS_API size_t nns_fd_sor_workspace(int batch, int nx, int ny, int elem_size) {
    if (batch < 1 || nx < 3 || ny < 3 || (elem_size != 4 && elem_size != 8)) return 0;
    return (size_t)batch * nx * ny * elem_size;
}
NNS_API int nns_fd_sor_f32(float* p, const float* C, float* info, void* work, int batch, int nx, int ny, double dx, double dy,
                           double beta, double tol, int max_sweeps, void* stream) {
    return sor<float>(p, C, info, nullptr, work, batch, nx, ny, dx, dy, beta, tol, max_sweeps, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_sor_f64(double* p, const double* C, double* info, void* work, int batch, int nx, int ny, double dx, double dy,
                           double beta, double tol, int max_sweeps, void* stream) {
    return sor<double>(p, C, info, nullptr, work, batch, nx, ny, dx, dy, beta, tol, max_sweeps, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_sor_hint_f32(float* p, const float* C, float* info, const float* hint, void* work, int batch, int nx, int ny, double dx, double dy,
                                double beta, double tol, int max_sweeps, void* stream) {
    return sor<float>(p, C, info, hint, work, batch, nx, ny, dx, dy, beta, tol, max_sweeps, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_sor_hint_f64(double* p, const double* C, double* info, const double* hint, void* work, int batch, int nx, int ny, double dx, double dy,
                                double beta, double tol, int max_sweeps, void* stream) {
    return sor<double>(p, C, info, hint, work, batch, nx, ny, dx, dy, beta, tol, max_sweeps, reinterpret_cast<hipStream_t>(stream));
}

NNS_API size_t nns_fd_sor_redblack_workspace(int batch, int nx, int ny, int elem_size, int max_sweeps) {
    if (batch < 1 || nx < 3 || ny < 3 || (elem_size != 4 && elem_size != 8) || max_sweeps < 0) return 0;
    return rb_fits_lds(nx, ny, (size_t)elem_size) ? 0 : (size_t)batch * (max_sweeps + 1) * elem_size;
}
NNS_API int nns_fd_sor_redblack_f32(float* p, const float* C, float* info, void* work, int batch, int nx, int ny, double dx, double dy,
                                    double beta, double tol, int max_sweeps, void* stream) {
    return sor_redblack<float>(p, C, info, work, batch, nx, ny, dx, dy, beta, tol, max_sweeps, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_sor_redblack_f64(double* p, const double* C, double* info, void* work, int batch, int nx, int ny, double dx, double dy,
                                    double beta, double tol, int max_sweeps, void* stream) {
    return sor_redblack<double>(p, C, info, work, batch, nx, ny, dx, dy, beta, tol, max_sweeps, reinterpret_cast<hipStream_t>(stream));
}

NNS_API int nns_fd_sor_redblack_halfsweep_f32(float* p, const float* C, void* err_bits, int nxl, int ny, int gi0, int colour,
                                              double dx, double dy, double beta, void* stream) {
    return sor_rb_halfsweep<float>(p, C, err_bits, nxl, ny, gi0, colour, dx, dy, beta, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_sor_redblack_halfsweep_f64(double* p, const double* C, void* err_bits, int nxl, int ny, int gi0, int colour,
                                              double dx, double dy, double beta, void* stream) {
    return sor_rb_halfsweep<double>(p, C, err_bits, nxl, ny, gi0, colour, dx, dy, beta, reinterpret_cast<hipStream_t>(stream));
}

NNS_API int nns_fd_sor_redblack_halfsweep_gated_f32(float* p, const float* C, void* err_bits, const void* prev_err_bits, double tol,
                                                    int nxl, int ny, int gi0, int colour, double dx, double dy, double beta, void* stream) {
    return sor_rb_halfsweep_gated<float>(p, C, err_bits, prev_err_bits, tol, nxl, ny, gi0, colour, dx, dy, beta, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_sor_redblack_halfsweep_gated_f64(double* p, const double* C, void* err_bits, const void* prev_err_bits, double tol,
                                                    int nxl, int ny, int gi0, int colour, double dx, double dy, double beta, void* stream) {
    return sor_rb_halfsweep_gated<double>(p, C, err_bits, prev_err_bits, tol, nxl, ny, gi0, colour, dx, dy, beta, reinterpret_cast<hipStream_t>(stream));
}
