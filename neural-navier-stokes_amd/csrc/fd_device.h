// Per-point forms of the chorin_fd operators and the boundary-condition list, shared by the one-operator kernels (csrc/fd_kernels.hip) and the
// fused explicit step (csrc/fd_step_kernels.hip): ONE definition of every expression, so that the two paths are bitwise the same (both
// translation units are compiled with -ffp-contract=off: the reference's operation order, oracle/chorin_fd.py).
#pragma once
#include "nns_common.h"

namespace nns {
namespace fd {

// ------------------------------------------------------------------------------------------
// Boundary conditions  (src/boundary.py:34-48, :56-86)
// ------------------------------------------------------------------------------------------
// One BC of the list applied by all threads of a block to the grid at A (global or LDS).
template <typename T, typename P>
__device__ __forceinline__ void bc_apply_one(P A, int nx, int ny, int kind, int side, T value, T dx, T dy,
                                             int tid, int nthreads) {
    if (side == NNS_SIDE_LEFT || side == NNS_SIDE_RIGHT) {
        const int i = side == NNS_SIDE_LEFT ? 0 : nx - 1;
        const int in = side == NNS_SIDE_LEFT ? 1 : nx - 2;
        for (int j = tid; j < ny; j += nthreads) {
            T r;
            if (kind == NNS_BC_DIRICHLET) r = value;
            else r = side == NNS_SIDE_LEFT ? A[(size_t)in * ny + j] - dx * value : A[(size_t)in * ny + j] + dx * value;
            A[(size_t)i * ny + j] = r;
        }
    } else {
        const int j = side == NNS_SIDE_BOTTOM ? 0 : ny - 1;
        const int jn = side == NNS_SIDE_BOTTOM ? 1 : ny - 2;
        for (int i = tid; i < nx; i += nthreads) {
            T r;
            if (kind == NNS_BC_DIRICHLET) r = value;
            else r = side == NNS_SIDE_BOTTOM ? A[(size_t)i * ny + jn] - dy * value : A[(size_t)i * ny + jn] + dy * value;
            A[(size_t)i * ny + j] = r;
        }
    }
}

// The whole list in list order (later entries win at corners, and a Neumann entry may read a
// corner an earlier entry wrote): one workgroup per grid, a barrier between entries.
template <typename T, typename P>
__device__ __forceinline__ void bc_apply_list(P A, int nx, int ny, const BcListDev<T>& bcs, int tid, int nthreads) {
    for (int k = 0; k < bcs.n; ++k) {
        bc_apply_one<T>(A, nx, ny, bcs.kind[k], bcs.side[k], bcs.value[k], bcs.dx[k], bcs.dy[k], tid, nthreads);
        __syncthreads();
    }
}


// The same list applied by ONE wave (the caller's other waves wait at the barrier that follows): inside a wave the entries are ordered by the LDS
// queue itself, so a list of four entries costs no workgroup barrier instead of four -- what a sweep-per-barrier loop (the Jacobi solve of
// direct_fd: the list after EVERY sweep) is made of.  A must be LDS (a wave's global stores are not ordered for its own later loads without a wait).
template <typename T, typename P>
__device__ __forceinline__ void bc_apply_list_one_wave(P A, int nx, int ny, const BcListDev<T>& bcs, int lane) {
    for (int k = 0; k < bcs.n; ++k) {
        bc_apply_one<T>(A, nx, ny, bcs.kind[k], bcs.side[k], bcs.value[k], bcs.dx[k], bcs.dy[k], lane, kWave);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------
// chorin_fd._explicit_predictor_step  (src/chorin_fd/simulate.py:63-91)
// ------------------------------------------------------------------------------------------
template <typename T>
struct PredK { T dt, two_dx, two_dy, dx2, dy2, dt_nu; };

template <typename T>
inline PredK<T> make_pred(double dt, double dx, double dy, double nu) { return PredK<T>{(T)dt, (T)(2 * dx), (T)(2 * dy), (T)(dx * dx), (T)(dy * dy), (T)(dt * nu)}; }

// Point (i, j) of ONE grid: ui, vi (boundary points: copies of un, vn).  CORRECT = true: v d/dy differenced along y (the build's option);
// false: the reference's form (x-difference twice, :73-76, :82-83).
template <typename T, bool CORRECT>
__device__ __forceinline__ void predictor_explicit_point(const T* __restrict__ un, const T* __restrict__ vn, const T* __restrict__ un1,
                                                         const T* __restrict__ vn1, T* __restrict__ ui, T* __restrict__ vi, int i, int j,
                                                         int nx, int ny, const PredK<T>& k) {
    const size_t c = (size_t)i * ny + j;
    const T uc = un[c], vc = vn[c];
    if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) { ui[c] = uc; vi[c] = vc; return; }
    const size_t xp = c + ny, xm = c - ny, yp = c + 1, ym = c - 1;
    const T u1c = un1[c], v1c = vn1[c];
    const T three_half = (T)1.5, half = (T)0.5, two = (T)2;
    {
        const T e = un[xp], w = un[xm], e1 = un1[xp], w1 = un1[xm];
        const T n = un[yp], so = un[ym], n1 = un1[yp], so1 = un1[ym];
        const T adv = uc * (e - w) / k.two_dx + vc * (CORRECT ? n - so : e - w) / k.two_dy;          // :73-74 (x-difference twice)
        const T adv1 = u1c * (e1 - w1) / k.two_dx + v1c * (CORRECT ? n1 - so1 : e1 - w1) / k.two_dy;  // :75-76
        const T lap = (e - two * uc + w) / k.dx2 + (n - two * uc + so) / k.dy2;
        const T lap1 = (e1 - two * u1c + w1) / k.dx2 + (n1 - two * u1c + so1) / k.dy2;
        ui[c] = uc - k.dt * (three_half * adv - half * adv1) + k.dt_nu * (three_half * lap - half * lap1);
    }
    {
        const T e = vn[xp], w = vn[xm], e1 = vn1[xp], w1 = vn1[xm];
        const T n = vn[yp], so = vn[ym], n1 = vn1[yp], so1 = vn1[ym];
        const T adv = uc * (e - w) / k.two_dx + vc * (CORRECT ? n - so : e - w) / k.two_dy;          // :82-83
        const T adv1 = u1c * (e1 - w1) / k.two_dx + v1c * (CORRECT ? n1 - so1 : e1 - w1) / k.two_dy;
        const T lap = (e - two * vc + w) / k.dx2 + (n - two * vc + so) / k.dy2;
        const T lap1 = (e1 - two * v1c + w1) / k.dx2 + (n1 - two * v1c + so1) / k.dy2;
        vi[c] = vc - k.dt * (three_half * adv - half * adv1) + k.dt_nu * (three_half * lap - half * lap1);
    }
}

// ------------------------------------------------------------------------------------------
// chorin_fd._get_pressure RHS (:186-188) and _correction_step (:204-210)
// ------------------------------------------------------------------------------------------
template <typename T>
inline void rhs_consts(double dt, double dx, double dy, double rho, T& cu, T& cv) { cu = (T)(dx * rho * (dy * dy) / dt); cv = (T)(dy * rho * (dx * dx) / dt); }      // :187-188

template <typename T>
__device__ __forceinline__ T pressure_rhs_point(const T* __restrict__ ui, const T* __restrict__ vi, int i, int j, int nx, int ny, T cu, T cv) {
    if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) return (T)0;
    const size_t c = (size_t)i * ny + j;
    return cu * (ui[c] - ui[c - ny]) + cv * (vi[c] - vi[c - 1]);
}

// p is addressed with its own row pitch ldp (the fused step keeps it in LDS); u, v may alias ui, vi (point c only reads ui[c], vi[c])
template <typename T, typename PP>
__device__ __forceinline__ void correction_point(const T* ui, const T* vi, PP p, int ldp, T* u, T* v, int i, int j, int nx, int ny, T cx, T cy) {
    const size_t c = (size_t)i * ny + j;
    const T uc = ui[c], vc = vi[c];
    if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) { u[c] = uc; v[c] = vc; return; }
    const size_t cp = (size_t)i * ldp + j;
    u[c] = uc - cx * (p[cp + ldp] - p[cp - ldp]);
    v[c] = vc - cy * (p[cp + 1] - p[cp - 1]);
}

}  // namespace fd
}  // namespace nns
