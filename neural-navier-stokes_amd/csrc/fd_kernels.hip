// Finite-difference operators of chorin_fd / direct_fd and boundary-condition application,
// as HIP kernels for gfx950.  Compiled with -ffp-contract=off: every expression keeps the
// reference's operation order (see oracle/chorin_fd.py, oracle/direct_fd.py), so the float64
// instantiations agree with the NumPy reference to rounding (no FMA contraction).
//
// All of these are HBM-bound streaming stencils (<= 3 FLOP/B): one thread per grid point,
// j (the contiguous axis) on threadIdx.x so every wave reads/writes 256 contiguous bytes per
// field; the i+-1 / j+-1 re-reads are served by L1/L2.  Algorithmic bytes per point are listed
// at each kernel (T = sizeof element).
#include "nns_common.h"
#include "fd_device.h"
#include <cstdlib>

using namespace nns;
using namespace nns::fd;

namespace {

constexpr int kTX = 256;   // threads along j (4 waves)

inline dim3 grid2d(int batch, int nx, int ny) { return dim3((ny + kTX - 1) / kTX, nx, batch); }

// ------------------------------------------------------------------------------------------
// Boundary conditions  (src/boundary.py:34-48, :56-86)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bc_apply_kernel(T* A, int nx, int ny, BcListDev<T> bcs) {
    T* g = A + (size_t)blockIdx.x * nx * ny;
    bc_apply_list<T>(g, nx, ny, bcs, threadIdx.x, blockDim.x);
}

template <typename T>
int bc_apply(T* A, int batch, int nx, int ny, const nns_bc_list* h, hipStream_t s) {
    if (!A || !field_args_ok(batch, nx, ny)) return fail(NNS_ERR_INVALID_ARG, "bc_apply: bad field args (batch=%d nx=%d ny=%d)", batch, nx, ny);
    BcListDev<T> d;
    if (int rc = make_bc_dev<T>(h, d)) return rc;
    if (d.n == 0) return NNS_OK;
    hipLaunchKernelGGL(bc_apply_kernel<T>, dim3(batch), dim3(256), 0, s, A, nx, ny, d);
    return check_launch("bc_apply");
}

// ------------------------------------------------------------------------------------------
// chorin_fd._explicit_predictor_step  (src/chorin_fd/simulate.py:63-91)
// algorithmic traffic: read un,vn,un1,vn1 + write ui,vi = 6T B/pt
// ------------------------------------------------------------------------------------------
// CORRECT = true: the build's "fixed y-advection" option (oracle:
// explicit_predictor_corrected): v d/dy differences along y.  false: the reference's form (x-difference twice).
template <typename T, bool CORRECT>
__global__ __launch_bounds__(kTX) void predictor_explicit_kernel(const T* __restrict__ un, const T* __restrict__ vn,
                                                                  const T* __restrict__ un1, const T* __restrict__ vn1,
                                                                  T* __restrict__ ui, T* __restrict__ vi,
                                                                  int nx, int ny, PredK<T> k) {
    const int j = blockIdx.x * kTX + threadIdx.x, i = blockIdx.y;
    if (j >= ny) return;
    const size_t base = (size_t)blockIdx.z * nx * ny;
    predictor_explicit_point<T, CORRECT>(un + base, vn + base, un1 + base, vn1 + base, ui + base, vi + base, i, j, nx, ny, k);
}

template <typename T>
int predictor_explicit(const T* un, const T* vn, const T* un1, const T* vn1, T* ui, T* vi, int batch, int nx, int ny,
                       double dt, double dx, double dy, double nu, hipStream_t s, bool corrected = false, bool column_slab = false) {
    if (!un || !vn || !un1 || !vn1 || !ui || !vi || !field_args_ok(batch, nx, ny))
        return fail(NNS_ERR_INVALID_ARG, "fd_predictor_explicit: bad args (batch=%d nx=%d ny=%d)", batch, nx, ny);
    const PredK<T> k = make_pred<T>(dt, dx, dy, nu);
    if (corrected) hipLaunchKernelGGL((predictor_explicit_kernel<T, true>), grid2d(batch, nx, ny), dim3(kTX), 0, s, un, vn, un1, vn1, ui, vi, nx, ny, k);
    else hipLaunchKernelGGL((predictor_explicit_kernel<T, false>), grid2d(batch, nx, ny), dim3(kTX), 0, s, un, vn, un1, vn1, ui, vi, nx, ny, k);
    return check_launch("fd_predictor_explicit");
}

// ------------------------------------------------------------------------------------------
// chorin_fd._semi_implicit_predictor_step  (src/chorin_fd/simulate.py:93-167)
// One thread per (field, column j): both tridiagonal solves run along axis 0 (reference quirk),
// so a column never needs another column's intermediate; all global accesses are coalesced
// across the threads of a wave (adjacent j).  The modified diagonal cp[i] of the constant-
// coefficient Thomas factorisation is the same for every column: each thread carries it in a
// register on the way down, thread 0 parks it in LDS for the way back up.
// algorithmic traffic: read 4 fields, write 2, + the ut/vt round trip (4T) = 10T B/pt.
// ------------------------------------------------------------------------------------------
template <typename T>
struct AdiK { T dt, two_dx, two_dy, dx2, dy2, dt_nu, half_dt, cx, cy, a_diag, b_diag; };

// FIRST_ONLY = true: stop after the first (axis-0) solve, leaving ut / vt in `work`: the corrected variant's second solve
// runs along axis 1 in adi_ysolve_kernel below.
template <typename T, bool FIRST_ONLY>
__global__ __launch_bounds__(64) void predictor_adi_kernel(const T* __restrict__ un, const T* __restrict__ vn,
                                                            const T* __restrict__ un1, const T* __restrict__ vn1,
                                                            T* __restrict__ ui, T* __restrict__ vi, T* __restrict__ work,
                                                            int nx, int ny, AdiK<T> k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* cpA = reinterpret_cast<T*>(smem_raw);       // [nx] modified diagonal, first solve
    T* cpB = cpA + nx;                             // [nx] second solve
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool is_v = blockIdx.y == 1;
    const size_t base = (size_t)blockIdx.z * nx * ny;
    const T* f = (is_v ? vn : un) + base;          // the field being advanced
    const T* f1 = (is_v ? vn1 : un1) + base;
    const T* a = un + base;  const T* b = vn + base;
    const T* a1 = un1 + base; const T* b1 = vn1 + base;
    T* out = (is_v ? vi : ui) + base;
    T* ft = work + ((size_t)blockIdx.z * 2 + (is_v ? 1 : 0)) * nx * ny;     // ut / vt
    const bool active = j < ny;
    const bool edge_col = active && (j == 0 || j == ny - 1);
    const bool solve = active && !edge_col;
    const T two = (T)2, three = (T)3;
    const T lo = -k.dt, up = -k.dt;

    if (!FIRST_ONLY && edge_col) {                  // ui = u.copy() on the untouched columns
        for (int i = 0; i < nx; ++i) out[(size_t)i * ny + j] = f[(size_t)i * ny + j];
    }
    // ---- first solve: A ut = (2/nu dx^2) (dt/2 (3H - H1) + dt nu lap f)        (:126-137)
    T cp = k.a_diag, yprev = 0;
    for (int i = 1; i <= nx - 2; ++i) {
        T l = 0;
        if (i > 1) { l = lo / cp; cp = k.a_diag - l * up; }
        if (threadIdx.x == 0) cpA[i] = cp;
        if (solve) {
            const size_t c = (size_t)i * ny + j;
            const T fc = f[c], fe = f[c + ny], fw = f[c - ny], fn = f[c + 1], fs = f[c - 1];
            const T H = a[c] * (fe - fw) / k.two_dx + b[c] * (fn - fs) / k.two_dy;
            const T H1 = a1[c] * (f1[c + ny] - f1[c - ny]) / k.two_dx + b1[c] * (f1[c + 1] - f1[c - 1]) / k.two_dy;
            const T C1 = k.half_dt * (three * H - H1);
            const T C2 = k.dt_nu * ((fe - two * fc + fw) / k.dx2 + (fn - two * fc + fs) / k.dy2);
            const T rhs = k.cx * (C1 + C2);
            const T y = i > 1 ? rhs - l * yprev : rhs;
            ft[c] = y;
            yprev = y;
        }
    }
    __syncthreads();
    if (solve) {
        T xnext = 0;
        for (int i = nx - 2; i >= 1; --i) {
            const size_t c = (size_t)i * ny + j;
            const T y = ft[c];
            const T x = i == nx - 2 ? y / cpA[i] : (y - up * xnext) / cpA[i];
            ft[c] = x;
            xnext = x;
        }
    }
    if (FIRST_ONLY) return;
    // ---- second solve: B ui = (2/nu dy^2)(ft + f) - dt d_yy f, again along axis 0   (:157-165)
    cp = k.b_diag; yprev = 0;
    for (int i = 1; i <= nx - 2; ++i) {
        T l = 0;
        if (i > 1) { l = lo / cp; cp = k.b_diag - l * up; }
        if (threadIdx.x == 0) cpB[i] = cp;
        if (solve) {
            const size_t c = (size_t)i * ny + j;
            const T fc = f[c];
            const T rhs = k.cy * (ft[c] + fc) - k.dt * (f[c + 1] - two * fc + f[c - 1]);
            const T y = i > 1 ? rhs - l * yprev : rhs;
            out[c] = y;
            yprev = y;
        }
    }
    __syncthreads();
    if (solve) {
        T xnext = 0;
        for (int i = nx - 2; i >= 1; --i) {
            const size_t c = (size_t)i * ny + j;
            const T y = out[c];
            const T x = i == nx - 2 ? y / cpB[i] : (y - up * xnext) / cpB[i];
            out[c] = x;
            xnext = x;
        }
        out[j] = f[j];                                                  // rows 0 and nx-1 copied
        out[(size_t)(nx - 1) * ny + j] = f[(size_t)(nx - 1) * ny + j];
    }
}

// ------------------------------------------------------------------------------------------
// The same predictor for grids whose two working fields fit one workgroup's LDS (round 4: the reference's 51 x 51, BASELINE config 1's 64 x 64).
// The kernel above is two lone waves walking 62 rows with a dozen global loads and a division chain per row in front of every recurrence step:
// 120 us of the semi-implicit step's 0.32 ms.  Here ONE workgroup of 512 threads owns a grid: every right-hand side is computed by all threads at
// once into LDS, the constant-coefficient factorisation (l[i] = lo / cp[i-1], cp[i], RN(1 / cp[i])) is tabulated ONCE per solve by one lane while
// the others do that, and a recurrence step is then  y = rhs - l[i] y_prev  /  x = (y - up x_next) / cp[i]  on LDS operands.  Same expressions
// on the same operands in the same order as the kernel above -- the division through div_exact, bitwise the IEEE quotient -- so the two kernels
// agree BITWISE (test).
// ------------------------------------------------------------------------------------------
constexpr int kAdiLdsThreads = 512;
#ifndef NNS_ADI_TIMING
#define NNS_ADI_TIMING 0            // 1: predictor_adi_lds_kernel prints the cycles of its phases (workgroup 0, threads 0 and 448)
#endif

// x / den from rcp = RN(1 / den) (Markstein; see div_den in sor_device.h and tools/fastdiv_check.hip); rcp = 0 or an operand out of the safe
// range: the plain division
template <typename T>
__device__ __forceinline__ T div_exact(T x, T den, T rcp) {
    constexpr T lo = sizeof(T) == 8 ? (T)1e-250 : (T)1e-25, hi = sizeof(T) == 8 ? (T)1e250 : (T)1e25;
    const T ax = fabs(x);
    const T q = x * rcp;
    T res = ax == (T)0 ? x : fma(fma(-q, den, x), rcp, q);                  // +-0 / den = +-0 (rcp != 0 only for den > 0)
    if (!(rcp != (T)0 && ((ax >= lo && ax <= hi) || ax == (T)0))) res = x / den;
    return res;
}

template <typename T>
__device__ __forceinline__ void adi_tables(T diag, T lo, T up, int nx, T* l, T* cp, T* rc) {
    constexpr T dlo = sizeof(T) == 8 ? (T)1e-50 : (T)1e-10, dhi = sizeof(T) == 8 ? (T)1e50 : (T)1e10;
    // ONE division per row: RN(1 / cp[i-1]) is needed for the back substitution anyway, and lo / cp[i-1] follows from it exactly (div_exact)
    // The recurrence c <- diag - (lo / c) up is a contraction (the matrix is diagonally dominant): in floating point it reaches a FIXED POINT after a
    // handful of rows (4 at the reference's cavity parameters), after which every row repeats the same three numbers bit for bit -- the chain of
    // divisions is cut there (it was 15 of this kernel's 48 us: a lone lane pays ~30 cycles per dependent float64 operation).
    T c = diag, r = (c >= dlo && c <= dhi) ? (T)1 / c : (T)0, li = 0;
    bool fixed = false;
    for (int i = 1; i <= nx - 2; ++i) {
        if (i > 1 && !fixed) {
            const T ln = div_exact<T>(lo, c, r);
            const T cn = diag - ln * up;
            fixed = i > 2 && ln == li && cn == c;
            if (!fixed) r = (cn >= dlo && cn <= dhi) ? (T)1 / cn : (T)0;
            li = ln; c = cn;
        }
        l[i] = li; cp[i] = c; rc[i] = r;
    }
}

// one column's forward elimination and back substitution in place on F[i * ny + j], i = 1 .. nx - 2: the streaming kernel's operations in its
// order (bitwise the same numbers).  Eight rows' operands are requested before their eight dependent steps.  (Measured and dropped: the back
// substitution re-associated as x = y / cp - (up / cp) x_next, two dependent operations per row instead of six -- 27.9 k cycles per solve against
// 24.6 k: the lone wave that owns 62 columns is bound by the ~30 instructions of a row step, not by the chain, and the last bits then differ from
// the streaming kernel that larger grids and other slab shapes take.)
template <typename T>
__device__ __forceinline__ void adi_column(T* F, int j, int nx, int ny, T up, const T* l, const T* cp, const T* rc) {
    constexpr int U = 8;
    T yprev = 0;
    for (int i0 = 1; i0 <= nx - 2; i0 += U) {
        T r[U], li[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = min(i0 + u, nx - 2); r[u] = F[i * ny + j]; li[u] = l[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            if (i <= nx - 2) {
                const T y = i > 1 ? r[u] - li[u] * yprev : r[u];
                F[i * ny + j] = y;
                yprev = y;
            }
        }
    }
    T xnext = 0;
    for (int i0 = nx - 2; i0 >= 1; i0 -= U) {
        T r[U], ci[U], ri[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = max(i0 - u, 1); r[u] = F[i * ny + j]; ci[u] = cp[i]; ri[u] = rc[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 - u;
            if (i >= 1) {
                const T x = i == nx - 2 ? div_exact<T>(r[u], ci[u], ri[u]) : div_exact<T>(r[u] - up * xnext, ci[u], ri[u]);
                F[i * ny + j] = x;
                xnext = x;
            }
        }
    }
}

// right-hand side of the first solve at interior point c of field fl (0: u, 1: v)   (:126-134)
template <typename T>
__device__ __forceinline__ T adi_rhs1_point(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ a1, const T* __restrict__ b1, int fl, int c,
                                            int ny, const AdiK<T>& k) {
    const T two = (T)2, three = (T)3;
    const T* f = fl ? b : a;
    const T* f1 = fl ? b1 : a1;
    const T fc = f[c], fe = f[c + ny], fw = f[c - ny], fn = f[c + 1], fs = f[c - 1];
    const T H = a[c] * (fe - fw) / k.two_dx + b[c] * (fn - fs) / k.two_dy;
    const T H1 = a1[c] * (f1[c + ny] - f1[c - ny]) / k.two_dx + b1[c] * (f1[c + 1] - f1[c - 1]) / k.two_dy;
    const T C1 = k.half_dt * (three * H - H1);
    const T C2 = k.dt_nu * ((fe - two * fc + fw) / k.dx2 + (fn - two * fc + fs) / k.dy2);
    return k.cx * (C1 + C2);
}

// The right-hand sides of the first solve by the WHOLE chip into `work` ([batch][2][nx][ny], interior points): eight float64 divisions per point
// are 27 us for one workgroup and a launch latency for 256 CUs.
template <typename T>
__global__ __launch_bounds__(kTX) void adi_rhs1_kernel(const T* __restrict__ un, const T* __restrict__ vn, const T* __restrict__ un1, const T* __restrict__ vn1,
                                                        T* __restrict__ work, int nx, int ny, AdiK<T> k) {
    const int j = blockIdx.x * kTX + threadIdx.x, i = blockIdx.y, fl = blockIdx.z & 1;
    if (j < 1 || j > ny - 2 || i < 1 || i > nx - 2) return;
    const size_t base = (size_t)(blockIdx.z >> 1) * nx * ny;
    work[(size_t)blockIdx.z * nx * ny + (size_t)i * ny + j] = adi_rhs1_point<T>(un + base, vn + base, un1 + base, vn1 + base, fl, i * ny + j, ny, k);
}

// RHS_READY: the first right-hand sides are in `work` (adi_rhs1_kernel); otherwise this workgroup computes them.
template <typename T, bool FIRST_ONLY, bool RHS_READY>
__global__ __launch_bounds__(kAdiLdsThreads) void predictor_adi_lds_kernel(const T* __restrict__ un, const T* __restrict__ vn,
                                                                            const T* __restrict__ un1, const T* __restrict__ vn1,
                                                                            T* __restrict__ ui, T* __restrict__ vi, T* __restrict__ work,
                                                                            int nx, int ny, AdiK<T> k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = nx * ny, tid = threadIdx.x;
    T* F = reinterpret_cast<T*>(smem_raw);          // [2][n]: the working fields of u and v
    T* tab = F + 2 * n;                             // l, cp, rc of the two solves: 6 x [nx]
    const size_t base = (size_t)blockIdx.x * n;
    const T* a = un + base;  const T* b = vn + base;
    const T* a1 = un1 + base; const T* b1 = vn1 + base;
    const T two = (T)2;
    const T lo = -k.dt, up = -k.dt;
    constexpr int UG = 4;                           // points per thread and round of the global-memory phases: their loads are in flight together
                                                    // (one point per round left every round waiting for its own loads: 16 x ~1000 cycles per phase)
    auto inner = [&](int c) { const int i = c / ny, j = c - i * ny; return i >= 1 && i <= nx - 2 && j >= 1 && j <= ny - 2; };
#if NNS_ADI_TIMING
    long tq[8]; tq[0] = clock64();
#endif
    // the factorisations, one lane each, while everyone else starts on the right-hand sides
    if (tid == 0) adi_tables<T>(k.a_diag, lo, up, nx, tab, tab + nx, tab + 2 * nx);
    if (tid == kWave && !FIRST_ONLY) adi_tables<T>(k.b_diag, lo, up, nx, tab + 3 * nx, tab + 4 * nx, tab + 5 * nx);
#if NNS_ADI_TIMING
    tq[1] = clock64();
#endif
    // ---- first solve: A ut = (2/nu dx^2) (dt/2 (3H - H1) + dt nu lap f)        (:126-137)
    for (int e0 = tid; e0 < 2 * n; e0 += UG * kAdiLdsThreads) {
        T v[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int e = min(e0 + u * kAdiLdsThreads, 2 * n - 1), fl = e >= n, c = e - fl * n;
            v[u] = RHS_READY ? work[(size_t)blockIdx.x * 2 * n + e] : (inner(c) ? adi_rhs1_point<T>(a, b, a1, b1, fl, c, ny, k) : (T)0);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) { const int e = e0 + u * kAdiLdsThreads; if (e < 2 * n) F[e] = v[u]; }
    }
    __syncthreads();
#if NNS_ADI_TIMING
    tq[2] = clock64();
#endif
    for (int q = tid; q < 2 * (ny - 2); q += kAdiLdsThreads) {
        const int fl = q >= ny - 2, j = 1 + q - fl * (ny - 2);
        adi_column<T>(F + fl * n, j, nx, ny, up, tab, tab + nx, tab + 2 * nx);
    }
    __syncthreads();
#if NNS_ADI_TIMING
    tq[3] = clock64();
#endif
    if (FIRST_ONLY) {                                                  // ut / vt to `work`: the corrected variant's second solve runs along axis 1
        T* ft = work + (size_t)blockIdx.x * 2 * n;
        for (int e = tid; e < 2 * n; e += kAdiLdsThreads) {
            const int fl = e >= n, c = e - fl * n;
            if (inner(c)) ft[e] = F[e];
        }
        return;
    }
    // ---- second solve: B ui = (2/nu dy^2)(ft + f) - dt d_yy f, again along axis 0   (:157-165)
    for (int e0 = tid; e0 < 2 * n; e0 += UG * kAdiLdsThreads) {
        T fc[UG], fp[UG], fm[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int e = min(e0 + u * kAdiLdsThreads, 2 * n - 1), fl = e >= n, c = e - fl * n;
            const T* f = fl ? b : a;
            const int cc = min(max(c, 1), n - 2);                       // (edge points are not used: any in-range address)
            fc[u] = f[cc]; fp[u] = f[cc + 1]; fm[u] = f[cc - 1];
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int e = e0 + u * kAdiLdsThreads;
            if (e < 2 * n && inner(e - (e >= n) * n)) F[e] = k.cy * (F[e] + fc[u]) - k.dt * (fp[u] - two * fc[u] + fm[u]);
        }
    }
    __syncthreads();
#if NNS_ADI_TIMING
    tq[4] = clock64();
#endif
    for (int q = tid; q < 2 * (ny - 2); q += kAdiLdsThreads) {
        const int fl = q >= ny - 2, j = 1 + q - fl * (ny - 2);
        adi_column<T>(F + fl * n, j, nx, ny, up, tab + 3 * nx, tab + 4 * nx, tab + 5 * nx);
    }
    __syncthreads();
#if NNS_ADI_TIMING
    tq[5] = clock64();
#endif
    for (int e0 = tid; e0 < 2 * n; e0 += UG * kAdiLdsThreads) {
        T fv[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) { const int e = min(e0 + u * kAdiLdsThreads, 2 * n - 1), fl = e >= n; fv[u] = (fl ? b : a)[e - fl * n]; }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int e = e0 + u * kAdiLdsThreads;
            if (e < 2 * n) { const int fl = e >= n, c = e - fl * n; ((fl ? vi : ui) + base)[c] = inner(c) ? F[e] : fv[u]; }     // ui = u.copy() on the edges
        }
    }
#if NNS_ADI_TIMING
    __syncthreads(); tq[6] = clock64();
    if ((tid == 0 || tid == 448) && blockIdx.x == 0) printf("adi lds, thread %d (cycles): tables %ld, rhs1 / load %ld, solve 1 %ld, rhs2 %ld, solve 2 %ld, write %ld\n", tid, tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4], tq[6] - tq[5]);
#endif
}

inline size_t adi_lds_bytes(int nx, int ny, size_t elem) { return (2 * (size_t)nx * ny + 6 * (size_t)nx) * elem; }

// ------------------------------------------------------------------------------------------
// Corrected option ("true y-direction ADI"; oracle: semi_implicit_predictor_corrected):
// the second solve  B ui = (2/nu dy^2)(ut + u) - dt d_yy u  along axis 1 (rows), as an ADI scheme means it.
// One thread per row would stride global memory by ny; instead a workgroup owns 64 rows and walks the columns in
// 64-wide tiles through LDS ([64][65]: both the coalesced tile copy, thread = column, and the per-row recurrence,
// thread = row, are conflict-free).  Down-sweep left to right storing the modified right-hand side in `out`, then the
// back substitution right to left; the constant-coefficient modified diagonal cp[j] is tabulated once in LDS.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void adi_ysolve_kernel(const T* __restrict__ un, const T* __restrict__ vn, const T* __restrict__ work,
                                                         T* __restrict__ ui, T* __restrict__ vi, int nx, int ny, AdiK<T> k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* cp = reinterpret_cast<T*>(smem_raw);            // [ny] modified diagonal
    T* tf = cp + ny;                                   // [64][65] tile of f (with one halo column each side handled by index)
    T* tt = tf + 64 * 65;                              // [64][65] tile of ft, then of the result
    const int t = threadIdx.x;
    const bool is_v = blockIdx.y == 1;
    const size_t base = (size_t)blockIdx.z * nx * ny;
    const T* f = (is_v ? vn : un) + base;
    const T* ft = work + ((size_t)blockIdx.z * 2 + (is_v ? 1 : 0)) * nx * ny;
    T* out = (is_v ? vi : ui) + base;
    const int i0 = blockIdx.x * 64, i = i0 + t;        // this thread's row in the recurrence phases
    const T lo = -k.dt, up = -k.dt, two = (T)2;
    if (t == 0) {
        T c = k.b_diag;
        for (int j = 1; j <= ny - 2; ++j) { if (j > 1) c = k.b_diag - (lo / c) * up; cp[j] = c; }
    }
    __syncthreads();
    const bool row_solve = i >= 1 && i <= nx - 2;
    // ---- down-sweep
    T yprev = 0, fprev = 0, cprev = k.b_diag;
    for (int j0 = 0; j0 < ny; j0 += 64) {
        for (int r = 0; r < 64; ++r) {                 // coalesced tile copy: thread = column
            const int ii = i0 + r, j = j0 + t;
            const bool in = ii < nx && j < ny;
            tf[r * 65 + t] = in ? f[(size_t)ii * ny + j] : (T)0;
            tt[r * 65 + t] = in ? ft[(size_t)ii * ny + j] : (T)0;
        }
        __syncthreads();
        if (i < nx) {
            const T fnext_tile = (j0 + 64 < ny) ? f[(size_t)i * ny + j0 + 64] : (T)0;      // right neighbour of the tile's last column
            for (int c = 0; c < 64; ++c) {
                const int j = j0 + c;
                if (j >= ny) break;
                const T fc = tf[t * 65 + c];
                if (row_solve && j >= 1 && j <= ny - 2) {
                    const T fn = c < 63 ? tf[t * 65 + c + 1] : fnext_tile;
                    const T rhs = k.cy * (tt[t * 65 + c] + fc) - k.dt * (fn - two * fc + fprev);
                    T y = rhs;
                    if (j > 1) { const T l = lo / cprev; y = rhs - l * yprev; }
                    cprev = cp[j];
                    tt[t * 65 + c] = y;
                    yprev = y;
                } else {
                    tt[t * 65 + c] = fc;                // edges: ui = u.copy()
                }
                fprev = fc;
            }
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r) {
            const int ii = i0 + r, j = j0 + t;
            if (ii < nx && j < ny) out[(size_t)ii * ny + j] = tt[r * 65 + t];
        }
        __syncthreads();
    }
    // ---- back substitution, right to left
    T xnext = 0;
    const int jlast = ((ny - 1) / 64) * 64;
    for (int j0 = jlast; j0 >= 0; j0 -= 64) {
        for (int r = 0; r < 64; ++r) {
            const int ii = i0 + r, j = j0 + t;
            tt[r * 65 + t] = (ii < nx && j < ny) ? out[(size_t)ii * ny + j] : (T)0;
        }
        __syncthreads();
        if (row_solve) {
            for (int c = 63; c >= 0; --c) {
                const int j = j0 + c;
                if (j > ny - 2 || j < 1) continue;
                const T y = tt[t * 65 + c];
                const T x = j == ny - 2 ? y / cp[j] : (y - up * xnext) / cp[j];
                tt[t * 65 + c] = x;
                xnext = x;
            }
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r) {
            const int ii = i0 + r, j = j0 + t;
            if (ii < nx && j < ny) out[(size_t)ii * ny + j] = tt[r * 65 + t];
        }
        __syncthreads();
    }
}

template <typename T>
int predictor_adi(const T* un, const T* vn, const T* un1, const T* vn1, T* ui, T* vi, T* work, int batch, int nx, int ny,
                  double dt, double dx, double dy, double nu, hipStream_t s, bool corrected = false, bool column_slab = false) {
    if (!un || !vn || !un1 || !vn1 || !ui || !vi || !work || !field_args_ok(batch, nx, ny))
        return fail(NNS_ERR_INVALID_ARG, "fd_predictor_adi: bad args (batch=%d nx=%d ny=%d)", batch, nx, ny);
    if (!corrected && !column_slab && nx != ny) return fail(NNS_ERR_INVALID_ARG, "fd_predictor_adi: the reference's second ADI solve acts along axis 0 (src/chorin_fd/simulate.py:159), which needs nx == ny (got %d x %d)", nx, ny);
    if (nu == 0) return fail(NNS_ERR_INVALID_ARG, "fd_predictor_adi: nu must be non-zero (2/nu)");
    AdiK<T> k;
    k.dt = (T)dt; k.two_dx = (T)(2 * dx); k.two_dy = (T)(2 * dy); k.dx2 = (T)(dx * dx); k.dy2 = (T)(dy * dy);
    k.dt_nu = (T)(dt * nu); k.half_dt = (T)(dt / 2.);
    k.cx = (T)(2 / nu * (dx * dx)); k.cy = (T)(2 / nu * (dy * dy));         // (2/nu)*dx^2  (:134, :157)
    k.a_diag = (T)(2 / nu * (dx * dx) + 2 * dt); k.b_diag = (T)(2 / nu * (dy * dy) + 2 * dt);   // :108, :117
    const int tpb = 64;
    const size_t shmem = 2 * (size_t)nx * sizeof(T);
    if (shmem > 64 * 1024) return fail(NNS_ERR_UNSUPPORTED, "fd_predictor_adi: nx=%d too large for the LDS diagonal cache", nx);
    static const bool lds_path = [] { const char* e = getenv("NNS_ADI_LDS"); return !e || atoi(e) != 0; }();     // NNS_ADI_LDS=0: the streaming kernel always (A/B)
    const size_t lds_all = adi_lds_bytes(nx, ny, sizeof(T));
    const bool small = lds_path && lds_all <= 150 * 1024;
    if (small) {
        // right-hand sides chip-wide, then one workgroup per grid for the two solves (a grid of one row of workgroups would spend 27 us on them)
        auto solve = [&](auto kern, int slot) -> int {
            static bool attr2[2] = {false, false};
            if (!attr2[slot]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "fd_predictor_adi: hipFuncSetAttribute: %s", hipGetErrorString(e));
                attr2[slot] = true;
            }
            hipLaunchKernelGGL(adi_rhs1_kernel<T>, dim3((ny + kTX - 1) / kTX, nx, 2 * batch), dim3(kTX), 0, s, un, vn, un1, vn1, work, nx, ny, k);
            hipLaunchKernelGGL(kern, dim3(batch), dim3(kAdiLdsThreads), lds_all, s, un, vn, un1, vn1, ui, vi, work, nx, ny, k);
            return NNS_OK;
        };
        if (!corrected) {
            if (int rc = solve(predictor_adi_lds_kernel<T, false, true>, 0)) return rc;
            return check_launch("fd_predictor_adi");
        }
        if (int rc = solve(predictor_adi_lds_kernel<T, true, true>, 1)) return rc;
        if (int rc = check_launch("fd_predictor_adi (x solve)")) return rc;
    } else {
        if (!corrected) {
            hipLaunchKernelGGL((predictor_adi_kernel<T, false>), dim3((ny + tpb - 1) / tpb, 2, batch), dim3(tpb), shmem, s,
                               un, vn, un1, vn1, ui, vi, work, nx, ny, k);
            return check_launch("fd_predictor_adi");
        }
        hipLaunchKernelGGL((predictor_adi_kernel<T, true>), dim3((ny + tpb - 1) / tpb, 2, batch), dim3(tpb), shmem, s,
                           un, vn, un1, vn1, ui, vi, work, nx, ny, k);
        if (int rc = check_launch("fd_predictor_adi (x solve)")) return rc;
    }
    const size_t shy = ((size_t)ny + 2 * 64 * 65) * sizeof(T);
    if (shy > 150 * 1024) return fail(NNS_ERR_UNSUPPORTED, "fd_predictor_adi_corrected: ny=%d too large for the LDS diagonal cache", ny);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(adi_ysolve_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "fd_predictor_adi_corrected: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
    hipLaunchKernelGGL(adi_ysolve_kernel<T>, dim3((nx + 63) / 64, 2, batch), dim3(64), shy, s, un, vn, work, ui, vi, nx, ny, k);
    return check_launch("fd_predictor_adi (y solve)");
}

// ------------------------------------------------------------------------------------------
// chorin_fd._get_pressure RHS (:186-188) and _correction_step (:204-210)
// traffic: rhs 3T B/pt, correction 5T B/pt
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kTX) void pressure_rhs_kernel(const T* __restrict__ ui, const T* __restrict__ vi, T* __restrict__ C,
                                                            int nx, int ny, T cu, T cv) {
    const int j = blockIdx.x * kTX + threadIdx.x, i = blockIdx.y;
    if (j >= ny) return;
    const size_t base = (size_t)blockIdx.z * nx * ny;
    C[base + (size_t)i * ny + j] = pressure_rhs_point<T>(ui + base, vi + base, i, j, nx, ny, cu, cv);
}

template <typename T>
int pressure_rhs(const T* ui, const T* vi, T* C, int batch, int nx, int ny, double dt, double dx, double dy, double rho, hipStream_t s) {
    if (!ui || !vi || !C || !field_args_ok(batch, nx, ny)) return fail(NNS_ERR_INVALID_ARG, "fd_pressure_rhs: bad args");
    T cu, cv;
    rhs_consts<T>(dt, dx, dy, rho, cu, cv);
    hipLaunchKernelGGL(pressure_rhs_kernel<T>, grid2d(batch, nx, ny), dim3(kTX), 0, s, ui, vi, C, nx, ny, cu, cv);
    return check_launch("fd_pressure_rhs");
}

template <typename T>
__global__ __launch_bounds__(kTX) void correction_kernel(const T* __restrict__ ui, const T* __restrict__ vi, const T* __restrict__ p,
                                                          T* __restrict__ u, T* __restrict__ v, int nx, int ny, T cx, T cy) {
    const int j = blockIdx.x * kTX + threadIdx.x, i = blockIdx.y;
    if (j >= ny) return;
    const size_t base = (size_t)blockIdx.z * nx * ny;
    correction_point<T>(ui + base, vi + base, p + base, ny, u + base, v + base, i, j, nx, ny, cx, cy);
}

template <typename T>
int correction(const T* ui, const T* vi, const T* p, T* u, T* v, int batch, int nx, int ny, double dt, double dx, double dy, hipStream_t s) {
    if (!ui || !vi || !p || !u || !v || !field_args_ok(batch, nx, ny)) return fail(NNS_ERR_INVALID_ARG, "fd_correction: bad args");
    hipLaunchKernelGGL(correction_kernel<T>, grid2d(batch, nx, ny), dim3(kTX), 0, s, ui, vi, p, u, v, nx, ny,
                       (T)(dt / (2 * dx)), (T)(dt / (2 * dy)));
    return check_launch("fd_correction");
}

// ------------------------------------------------------------------------------------------
// direct_fd  (src/direct_fd/simulate.py; axis 1 = x, axis 0 = y)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kTX) void build_b_kernel(const T* __restrict__ u, const T* __restrict__ v, T* __restrict__ b,
                                                       int nx, int ny, T rho, T inv_dt, T two_dx, T two_dy) {
    const int j = blockIdx.x * kTX + threadIdx.x, i = blockIdx.y;
    if (j >= ny) return;
    const size_t c = (size_t)blockIdx.z * nx * ny + (size_t)i * ny + j;
    if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) { b[c] = (T)0; return; }
    const T ux = (u[c + 1] - u[c - 1]) / two_dx;        // d u / dx  (:60)
    const T vy = (v[c + ny] - v[c - ny]) / two_dy;      // d v / dy  (:61)
    const T uy = (u[c + ny] - u[c - ny]) / two_dy;      // :63
    const T vx_num = (v[c + 1] - v[c - 1]);             // :64   (uy * vx_num) / two_dx keeps the reference order
    b[c] = (rho * (inv_dt * (ux + vy)) - ux * ux - (T)2 * (uy * vx_num / two_dx) - vy * vy);
}

template <typename T>
int build_b(const T* u, const T* v, T* b, int batch, int nx, int ny, double dt, double dx, double dy, double rho, hipStream_t s) {
    if (!u || !v || !b || !field_args_ok(batch, nx, ny)) return fail(NNS_ERR_INVALID_ARG, "fd_build_b: bad args");
    hipLaunchKernelGGL(build_b_kernel<T>, grid2d(batch, nx, ny), dim3(kTX), 0, s, u, v, b, nx, ny, (T)rho, (T)(1 / dt), (T)(2 * dx), (T)(2 * dy));
    return check_launch("fd_build_b");
}

template <typename T>
struct JacK { T dx2, dy2, den, cb, rcp; };     // rcp = RN(1 / den) where div_exact's short form applies, else 0 (make_jac)

template <typename T>
inline JacK<T> make_jac(double dx, double dy) {
    JacK<T> k{(T)(dx * dx), (T)(dy * dy), (T)(2 * (dx * dx + dy * dy)), (T)((dx * dx) * (dy * dy) / (2 * (dx * dx + dy * dy))), (T)0};
    const double ad = (double)k.den, dlo = sizeof(T) == 8 ? 1e-50 : 1e-10, dhi = sizeof(T) == 8 ? 1e50 : 1e10;
    if (ad >= dlo && ad <= dhi) k.rcp = (T)1 / k.den;
    return k;
}

template <typename T>
__device__ __forceinline__ T jacobi_point(T e, T w, T n, T s, T bb, const JacK<T>& k) {
    // (((pn[j+1] + pn[j-1]) * dy^2 + (pn[i+1] + pn[i-1]) * dx^2) / (2 (dx^2+dy^2)) - cb * b   (:78-82); the division by the constant
    // denominator through div_exact (bitwise the IEEE quotient, three instructions instead of thirteen)
    return div_exact<T>((e + w) * k.dy2 + (n + s) * k.dx2, k.den, k.rcp) - k.cb * bb;
}

// Multi-launch path (large grids): one sweep src -> dst, edges copied.  traffic 3T B/pt/sweep.
template <typename T>
__global__ __launch_bounds__(kTX) void jacobi_sweep_kernel(const T* __restrict__ src, T* __restrict__ dst, const T* __restrict__ b,
                                                            int nx, int ny, JacK<T> k) {
    const int j = blockIdx.x * kTX + threadIdx.x, i = blockIdx.y;
    if (j >= ny) return;
    const size_t c = (size_t)blockIdx.z * nx * ny + (size_t)i * ny + j;
    if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) { dst[c] = src[c]; return; }
    dst[c] = jacobi_point<T>(src[c + 1], src[c - 1], src[c + ny], src[c - ny], b[c], k);
}

// LDS-resident path (grids that fit): one workgroup per grid runs all nit sweeps, the BC list
// after each, without leaving the CU.  HBM traffic 3T B/pt for the whole solve.
template <typename T>
__global__ __launch_bounds__(1024) void jacobi_lds_kernel(T* __restrict__ p, const T* __restrict__ b, int nx, int ny, int nit,
                                                           JacK<T> k, BcListDev<T> bcs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = nx * ny;
    T* buf0 = reinterpret_cast<T*>(smem_raw);
    T* buf1 = buf0 + n;
    T* bl = buf1 + n;
    T* g = p + (size_t)blockIdx.x * n;
    const T* bg = b + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int c = tid; c < n; c += nt) { const T v = g[c]; buf0[c] = v; buf1[c] = v; bl[c] = bg[c]; }
    __syncthreads();
    T* cur = buf0; T* nxt = buf1;
    const int mi = nx - 2, mj = ny - 2;
    // a thread relaxes the same points in every sweep: their indices once, not two integer divisions per point and sweep (round 4: the sweep is
    // bound by instruction issue -- 1.9 us per sweep at 50 x 50 -- and t / mj, t % mj were as many instructions as the float64 update)
    constexpr int kOwn = 4;
    int own[kOwn];
#pragma unroll
    for (int u = 0; u < kOwn; ++u) { const int t = tid + u * nt; own[u] = t < mi * mj ? (1 + t / mj) * ny + 1 + t % mj : -1; }
    for (int q = 0; q < nit; ++q) {
#pragma unroll
        for (int u = 0; u < kOwn; ++u) {
            const int c = own[u];
            if (c >= 0) nxt[c] = jacobi_point<T>(cur[c + 1], cur[c - 1], cur[c + ny], cur[c - ny], bl[c], k);
        }
        for (int t = tid + kOwn * nt; t < mi * mj; t += nt) {               // (grids of more than 4096 interior points)
            const int i = 1 + t / mj, j = 1 + t % mj, c = i * ny + j;
            nxt[c] = jacobi_point<T>(cur[c + 1], cur[c - 1], cur[c + ny], cur[c - ny], bl[c], k);
        }
        // edges of nxt still hold the values after the previous BC application (p is updated in
        // place in the reference): copy them forward from cur before applying the BCs.
        for (int t = tid; t < 2 * (nx + ny); t += nt) {
            int i, j;
            if (t < ny) { i = 0; j = t; } else if (t < 2 * ny) { i = nx - 1; j = t - ny; }
            else if (t < 2 * ny + nx) { i = t - 2 * ny; j = 0; } else { i = t - 2 * ny - nx; j = ny - 1; }
            nxt[i * ny + j] = cur[i * ny + j];
        }
        __syncthreads();
        if (tid < kWave) bc_apply_list_one_wave<T>(nxt, nx, ny, bcs, tid);       // (round 4: one wave, no barrier per entry: 5-6 barriers per sweep became 2)
        __syncthreads();
        T* tmp = cur; cur = nxt; nxt = tmp;
    }
    for (int c = tid; c < n; c += nt) g[c] = cur[c];
}

template <typename T>
int jacobi(T* p, T* tmp, const T* b, int batch, int nx, int ny, double dx, double dy, int nit, const nns_bc_list* h, hipStream_t s) {
    if (!p || !b || !field_args_ok(batch, nx, ny) || nit < 0) return fail(NNS_ERR_INVALID_ARG, "fd_jacobi: bad args");
    BcListDev<T> d;
    if (int rc = make_bc_dev<T>(h, d)) return rc;
    if (nit == 0) return NNS_OK;
    const JacK<T> k = make_jac<T>(dx, dy);
    const size_t lds = 3 * (size_t)nx * ny * sizeof(T);
    if (lds <= 150 * 1024) {
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_lds_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "fd_jacobi: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr = true;
        }
        hipLaunchKernelGGL(jacobi_lds_kernel<T>, dim3(batch), dim3(1024), lds, s, p, b, nx, ny, nit, k, d);
        return check_launch("fd_jacobi(lds)");
    }
    if (!tmp) return fail(NNS_ERR_INVALID_ARG, "fd_jacobi: tmp scratch field required for %dx%d grids", nx, ny);
    T* src = p; T* dst = tmp;
    for (int q = 0; q < nit; ++q) {
        hipLaunchKernelGGL(jacobi_sweep_kernel<T>, grid2d(batch, nx, ny), dim3(kTX), 0, s, src, dst, b, nx, ny, k);
        if (d.n) hipLaunchKernelGGL(bc_apply_kernel<T>, dim3(batch), dim3(256), 0, s, dst, nx, ny, d);
        T* t = src; src = dst; dst = t;
    }
    if (src != p) {
        hipError_t e = hipMemcpyAsync(p, src, (size_t)batch * nx * ny * sizeof(T), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "fd_jacobi: copy-back: %s", hipGetErrorString(e));
    }
    return check_launch("fd_jacobi");
}

template <typename T>
struct DirK { T dt, dx, dy, c_px, c_py, nu, dt_dx2, dt_dy2; };

// direct_fd.step momentum update (:98-118).  traffic: read un,vn,p + write u,v = 5T B/pt
template <typename T>
__global__ __launch_bounds__(kTX) void direct_update_kernel(const T* __restrict__ un, const T* __restrict__ vn, const T* __restrict__ p,
                                                             T* __restrict__ u, T* __restrict__ v, int nx, int ny, DirK<T> k) {
    const int j = blockIdx.x * kTX + threadIdx.x, i = blockIdx.y;
    if (j >= ny) return;
    const size_t c = (size_t)blockIdx.z * nx * ny + (size_t)i * ny + j;
    const T uc = un[c], vc = vn[c];
    if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) { u[c] = uc; v[c] = vc; return; }
    const T two = (T)2;
    {
        const T e = un[c + 1], w = un[c - 1], n = un[c + ny], s = un[c - ny];
        u[c] = (uc - uc * k.dt / k.dx * (uc - w) - vc * k.dt / k.dy * (uc - s) - k.c_px * (p[c + 1] - p[c - 1]) +
                k.nu * (k.dt_dx2 * (e - two * uc + w) + k.dt_dy2 * (n - two * uc + s)));
    }
    {
        const T e = vn[c + 1], w = vn[c - 1], n = vn[c + ny], s = vn[c - ny];
        v[c] = (vc - uc * k.dt / k.dx * (vc - w) - vc * k.dt / k.dy * (vc - s) - k.c_py * (p[c + ny] - p[c - ny]) +
                k.nu * (k.dt_dx2 * (e - two * vc + w) + k.dt_dy2 * (n - two * vc + s)));
    }
}

template <typename T>
int direct_update(const T* un, const T* vn, const T* p, T* u, T* v, int batch, int nx, int ny, double dt, double dx, double dy,
                  double rho, double nu, hipStream_t s) {
    if (!un || !vn || !p || !u || !v || !field_args_ok(batch, nx, ny)) return fail(NNS_ERR_INVALID_ARG, "fd_direct_update: bad args");
    if (un == u || vn == v) return fail(NNS_ERR_INVALID_ARG, "fd_direct_update: in/out must not alias");
    DirK<T> k{(T)dt, (T)dx, (T)dy, (T)(dt / (2 * rho * dx)), (T)(dt / (2 * rho * dy)), (T)nu, (T)(dt / (dx * dx)), (T)(dt / (dy * dy))};
    hipLaunchKernelGGL(direct_update_kernel<T>, grid2d(batch, nx, ny), dim3(kTX), 0, s, un, vn, p, u, v, nx, ny, k);
    return check_launch("fd_direct_update");
}

}  // namespace

// ---------------------------------------------------------------------------- C ABI
#define S(stream) reinterpret_cast<hipStream_t>(stream)

NNS_API int nns_bc_apply_f32(float* A, int batch, int nx, int ny, const nns_bc_list* bcs, void* stream) { return bc_apply<float>(A, batch, nx, ny, bcs, S(stream)); }
NNS_API int nns_bc_apply_f64(double* A, int batch, int nx, int ny, const nns_bc_list* bcs, void* stream) { return bc_apply<double>(A, batch, nx, ny, bcs, S(stream)); }

NNS_API int nns_fd_predictor_explicit_f32(const float* un, const float* vn, const float* un1, const float* vn1, float* ui, float* vi,
                                          int batch, int nx, int ny, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_explicit<float>(un, vn, un1, vn1, ui, vi, batch, nx, ny, dt, dx, dy, nu, S(stream));
}
NNS_API int nns_fd_predictor_explicit_corrected_f32(const float* un, const float* vn, const float* un1, const float* vn1, float* ui, float* vi,
                                                    int batch, int nx, int ny, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_explicit<float>(un, vn, un1, vn1, ui, vi, batch, nx, ny, dt, dx, dy, nu, S(stream), true);
}
NNS_API int nns_fd_predictor_explicit_corrected_f64(const double* un, const double* vn, const double* un1, const double* vn1, double* ui, double* vi,
                                                    int batch, int nx, int ny, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_explicit<double>(un, vn, un1, vn1, ui, vi, batch, nx, ny, dt, dx, dy, nu, S(stream), true);
}
NNS_API int nns_fd_predictor_explicit_f64(const double* un, const double* vn, const double* un1, const double* vn1, double* ui, double* vi,
                                          int batch, int nx, int ny, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_explicit<double>(un, vn, un1, vn1, ui, vi, batch, nx, ny, dt, dx, dy, nu, S(stream));
}

NNS_API size_t nns_fd_predictor_adi_workspace(int batch, int nx, int ny, int elem_size) {
    if (batch < 1 || nx < 3 || ny < 3 || (elem_size != 4 && elem_size != 8)) return 0;
    return (size_t)4 * batch * nx * ny * elem_size;
}
NNS_API int nns_fd_predictor_adi_f32(const float* un, const float* vn, const float* un1, const float* vn1, float* ui, float* vi, float* work,
                                     int batch, int nx, int ny, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_adi<float>(un, vn, un1, vn1, ui, vi, work, batch, nx, ny, dt, dx, dy, nu, S(stream));
}
NNS_API int nns_fd_predictor_adi_corrected_f32(const float* un, const float* vn, const float* un1, const float* vn1, float* ui, float* vi, float* work,
                                               int batch, int nx, int ny, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_adi<float>(un, vn, un1, vn1, ui, vi, work, batch, nx, ny, dt, dx, dy, nu, S(stream), true);
}
NNS_API int nns_fd_predictor_adi_corrected_f64(const double* un, const double* vn, const double* un1, const double* vn1, double* ui, double* vi, double* work,
                                               int batch, int nx, int ny, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_adi<double>(un, vn, un1, vn1, ui, vi, work, batch, nx, ny, dt, dx, dy, nu, S(stream), true);
}
NNS_API int nns_fd_predictor_adi_f64(const double* un, const double* vn, const double* un1, const double* vn1, double* ui, double* vi, double* work,
                                     int batch, int nx, int ny, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_adi<double>(un, vn, un1, vn1, ui, vi, work, batch, nx, ny, dt, dx, dy, nu, S(stream));
}
NNS_API int nns_fd_predictor_adi_colslab_f32(const float* un, const float* vn, const float* un1, const float* vn1, float* ui, float* vi, float* work,
                                             int batch, int nx, int nyl, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_adi<float>(un, vn, un1, vn1, ui, vi, work, batch, nx, nyl, dt, dx, dy, nu, S(stream), false, true);
}
NNS_API int nns_fd_predictor_adi_colslab_f64(const double* un, const double* vn, const double* un1, const double* vn1, double* ui, double* vi, double* work,
                                             int batch, int nx, int nyl, double dt, double dx, double dy, double nu, void* stream) {
    return predictor_adi<double>(un, vn, un1, vn1, ui, vi, work, batch, nx, nyl, dt, dx, dy, nu, S(stream), false, true);
}

NNS_API int nns_fd_pressure_rhs_f32(const float* ui, const float* vi, float* C, int batch, int nx, int ny, double dt, double dx, double dy, double rho, void* stream) {
    return pressure_rhs<float>(ui, vi, C, batch, nx, ny, dt, dx, dy, rho, S(stream));
}
NNS_API int nns_fd_pressure_rhs_f64(const double* ui, const double* vi, double* C, int batch, int nx, int ny, double dt, double dx, double dy, double rho, void* stream) {
    return pressure_rhs<double>(ui, vi, C, batch, nx, ny, dt, dx, dy, rho, S(stream));
}

NNS_API int nns_fd_correction_f32(const float* ui, const float* vi, const float* p, float* u, float* v, int batch, int nx, int ny, double dt, double dx, double dy, void* stream) {
    return correction<float>(ui, vi, p, u, v, batch, nx, ny, dt, dx, dy, S(stream));
}
NNS_API int nns_fd_correction_f64(const double* ui, const double* vi, const double* p, double* u, double* v, int batch, int nx, int ny, double dt, double dx, double dy, void* stream) {
    return correction<double>(ui, vi, p, u, v, batch, nx, ny, dt, dx, dy, S(stream));
}

NNS_API int nns_fd_build_b_f32(const float* u, const float* v, float* b, int batch, int nx, int ny, double dt, double dx, double dy, double rho, void* stream) {
    return build_b<float>(u, v, b, batch, nx, ny, dt, dx, dy, rho, S(stream));
}
NNS_API int nns_fd_build_b_f64(const double* u, const double* v, double* b, int batch, int nx, int ny, double dt, double dx, double dy, double rho, void* stream) {
    return build_b<double>(u, v, b, batch, nx, ny, dt, dx, dy, rho, S(stream));
}

NNS_API int nns_fd_jacobi_f32(float* p, float* tmp, const float* b, int batch, int nx, int ny, double dx, double dy, int nit, const nns_bc_list* bc, void* stream) {
    return jacobi<float>(p, tmp, b, batch, nx, ny, dx, dy, nit, bc, S(stream));
}
NNS_API int nns_fd_jacobi_f64(double* p, double* tmp, const double* b, int batch, int nx, int ny, double dx, double dy, int nit, const nns_bc_list* bc, void* stream) {
    return jacobi<double>(p, tmp, b, batch, nx, ny, dx, dy, nit, bc, S(stream));
}

NNS_API int nns_fd_direct_update_f32(const float* un, const float* vn, const float* p, float* u, float* v, int batch, int nx, int ny,
                                     double dt, double dx, double dy, double rho, double nu, void* stream) {
    return direct_update<float>(un, vn, p, u, v, batch, nx, ny, dt, dx, dy, rho, nu, S(stream));
}
NNS_API int nns_fd_direct_update_f64(const double* un, const double* vn, const double* p, double* u, double* v, int batch, int nx, int ny,
                                     double dt, double dx, double dy, double rho, double nu, void* stream) {
    return direct_update<double>(un, vn, p, u, v, batch, nx, ny, dt, dx, dy, rho, nu, S(stream));
}
