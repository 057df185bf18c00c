// One time step of chorin_fd's explicit method (src/chorin_fd/simulate.py:212-234: _explicit_predictor_step, boundary conditions, _get_pressure,
// boundary conditions, _correction_step) as ONE launch, one workgroup per grid.
//
// At the reference's grid sizes (51 x 51; 64 x 64 in BASELINE config 1) a step is seven kernels and three trajectory copies of ~5 us each around
// the pressure solve -- a quarter of the step (profiles/r04_c1_step.txt).  The solve already keeps p and its right-hand side in the LDS of one
// workgroup; here the same workgroup also computes the predictor (into the output velocity fields), applies the boundary lists, builds the
// right-hand side straight into LDS, and after the solve applies the pressure boundary list in LDS, writes p (and its trajectory copy) and
// corrects the velocities in place.  Every expression is the per-point function the one-operator kernels use (fd_device.h, sor_device.h,
// -ffp-contract=off): the step is BITWISE the sequence of separate launches.
#include "nns_common.h"
#include "fd_device.h"
#include "sor_device.h"

using namespace nns;
using namespace nns::fd;
using namespace nns::sorlex;

namespace {

#ifndef NNS_STEP_TIMING
#define NNS_STEP_TIMING 0           // 1: the kernel prints the cycles of its phases (workgroup 0)
#endif

template <typename T>
struct StepK { PredK<T> pred; T cu, cv, cx, cy; SorK<T> sor; };

// UV_LDS: the intermediate velocities live in LDS next to p and C (four fields: 64 x 64 float64, 96 x 96 float32 and below) -- the boundary lists
// (a barrier per entry), the right-hand side and the correction then never wait for global memory, which a lone workgroup cannot hide (with the
// intermediates in the output fields the fused step took as long as the separate launches: 193 us against 150 for the solve alone).  Otherwise they
// live in the output fields.
template <typename T, bool CORRECT, bool UV_LDS>
__global__ __launch_bounds__(kSorThreads) void fd_step_explicit_kernel(const T* __restrict__ un, const T* __restrict__ vn, const T* __restrict__ un1,
                                                                        const T* __restrict__ vn1, T* p, T* u_out, T* v_out, T* p_copy, T* info, const T* hint,
                                                                        T* __restrict__ snap, int nx, int ny, int max_sweeps, StepK<T> k,
                                                                        BcListDev<T> ubc, BcListDev<T> vbc, BcListDev<T> pbc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* errs = reinterpret_cast<T*>(smem_raw);
    int* s_stop_p = reinterpret_cast<int*>(smem_raw + kSorBatch * 8);
    const int n = nx * ny, tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * n;
    un += base; vn += base; un1 += base; vn1 += base;
    T* pg = p + base;
    T* pl = reinterpret_cast<T*>(smem_raw + kSorHdr);
    T* cl = pl + n;
    T* ui = UV_LDS ? cl + n : u_out + base;
    T* vi = UV_LDS ? cl + 2 * n : v_out + base;
#if NNS_STEP_TIMING
    long tq[7]; tq[0] = clock64();
#endif
    // 1. predictor (:63-91), then the velocity boundary lists (:219-220)
    for (int c = tid; c < n; c += kSorThreads) predictor_explicit_point<T, CORRECT>(un, vn, un1, vn1, ui, vi, c / ny, c % ny, nx, ny, k.pred);
    __syncthreads();
#if NNS_STEP_TIMING
    tq[1] = clock64();
#endif
    bc_apply_list<T>(ui, nx, ny, ubc, tid, kSorThreads);
    bc_apply_list<T>(vi, nx, ny, vbc, tid, kSorThreads);
#if NNS_STEP_TIMING
    tq[2] = clock64();
#endif
    // 2. right-hand side (:186-188) and p into LDS; the solve (:183-200)
    for (int c = tid; c < n; c += kSorThreads) { cl[c] = pressure_rhs_point<T>(ui, vi, c / ny, c % ny, nx, ny, k.cu, k.cv); pl[c] = pg[c]; }
    __syncthreads();
#if NNS_STEP_TIMING
    tq[3] = clock64();
#endif
    int done;
    T err;
    const int expect = hint ? (int)hint[2 * blockIdx.x] : 0;          // (before info is written: the two may be one buffer)
    sor_solve<T, true>(pl, cl, snap + base, nx, ny, max_sweeps, expect, k.sor, errs, s_stop_p, done, err);
#if NNS_STEP_TIMING
    tq[4] = clock64();
#endif
    // 3. the pressure boundary list (:222) on the LDS copy; p out; correction (:204-210)
    bc_apply_list<T>(pl, nx, ny, pbc, tid, kSorThreads);
#if NNS_STEP_TIMING
    tq[5] = clock64();
#endif
    T* pc = p_copy ? p_copy + base : nullptr;
    for (int c = tid; c < n; c += kSorThreads) {
        const T pv = pl[c];
        pg[c] = pv;
        if (pc) pc[c] = pv;
        correction_point<T>(ui, vi, pl, ny, u_out + base, v_out + base, c / ny, c % ny, nx, ny, k.cx, k.cy);
    }
#if NNS_STEP_TIMING
    __syncthreads(); tq[6] = clock64();
    if (tid == 0 && blockIdx.x == 0) printf("fused step (cycles): predictor %ld, velocity bcs %ld, rhs + p load %ld, solve %ld (%d sweeps), pressure bcs %ld, write + correction %ld\n", tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], done, tq[5] - tq[4], tq[6] - tq[5]);
#endif
    if (tid == 0) { info[2 * blockIdx.x] = (T)done; info[2 * blockIdx.x + 1] = err; }
}

template <typename T>
int step_explicit(const T* un, const T* vn, const T* un1, const T* vn1, T* p, const nns_bc_list* u_bc, const nns_bc_list* v_bc, const nns_bc_list* p_bc,
                  T* u_out, T* v_out, T* p_copy, T* info, const T* hint, void* work, int batch, int nx, int ny, double dt, double dx, double dy, double rho, double nu,
                  double beta, double tol, int max_sweeps, int corrected, hipStream_t s) {
    if (!un || !vn || !un1 || !vn1 || !p || !u_out || !v_out || !info || !work || !field_args_ok(batch, nx, ny) || max_sweeps < 0)
        return fail(NNS_ERR_INVALID_ARG, "fd_step_explicit: bad args (batch=%d nx=%d ny=%d max_sweeps=%d)", batch, nx, ny, max_sweeps);
    for (const T* in : {un, vn, un1, vn1})
        if (in == u_out || in == v_out) return fail(NNS_ERR_INVALID_ARG, "fd_step_explicit: the output velocity fields must not be input fields (the predictor reads neighbours)");
    if (u_out == v_out || p == u_out || p == v_out || p_copy == p) return fail(NNS_ERR_INVALID_ARG, "fd_step_explicit: aliased output fields");
    const size_t lds = sor_lds_bytes(nx, ny, sizeof(T));
    if (lds > kSorLdsMax) return fail(NNS_ERR_UNSUPPORTED, "fd_step_explicit: a %d x %d grid does not fit the workgroup's LDS (use the separate operators)", nx, ny);
    BcListDev<T> ub, vb, pb;
    if (int rc = make_bc_dev<T>(u_bc, ub)) return rc;
    if (int rc = make_bc_dev<T>(v_bc, vb)) return rc;
    if (int rc = make_bc_dev<T>(p_bc, pb)) return rc;
    StepK<T> k;
    k.pred = make_pred<T>(dt, dx, dy, nu);
    rhs_consts<T>(dt, dx, dy, rho, k.cu, k.cv);
    k.cx = (T)(dt / (2 * dx)); k.cy = (T)(dt / (2 * dy));
    k.sor = make_sor_k<T>(dx, dy, beta, tol);
    const size_t lds4 = lds + 2 * (size_t)nx * ny * sizeof(T);            // with the intermediate velocities in LDS as well
    const bool uv = lds4 <= kSorLdsMax;
    T* snap = reinterpret_cast<T*>(work);
    auto launch = [&](auto kern, int slot) -> int {
        static bool attr[4] = {false, false, false, false};
        if (!attr[slot]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSorLdsMax);
            if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "fd_step_explicit: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr[slot] = true;
        }
        hipLaunchKernelGGL(kern, dim3(batch), dim3(kSorThreads), uv ? lds4 : lds, s, un, vn, un1, vn1, p, u_out, v_out, p_copy, info, hint, snap, nx, ny, max_sweeps, k, ub, vb, pb);
        return NNS_OK;
    };
    int rc;
    if (corrected) rc = uv ? launch(fd_step_explicit_kernel<T, true, true>, 0) : launch(fd_step_explicit_kernel<T, true, false>, 1);
    else rc = uv ? launch(fd_step_explicit_kernel<T, false, true>, 2) : launch(fd_step_explicit_kernel<T, false, false>, 3);
    if (rc != NNS_OK) return rc;
    return check_launch("fd_step_explicit");
}

}  // namespace

NNS_API int nns_fd_step_explicit_fits(int nx, int ny, int elem_size) {
    return nx >= 3 && ny >= 3 && (elem_size == 4 || elem_size == 8) && sor_lds_bytes(nx, ny, (size_t)elem_size) <= kSorLdsMax;
}

NNS_API int nns_fd_step_explicit_f32(const float* un, const float* vn, const float* un1, const float* vn1, float* p, const nns_bc_list* u_bc, const nns_bc_list* v_bc,
                                     const nns_bc_list* p_bc, float* u_out, float* v_out, float* p_copy, float* info, const float* hint, void* work, int batch, int nx, int ny,
                                     double dt, double dx, double dy, double rho, double nu, double beta, double tol, int max_sweeps, int corrected, void* stream) {
    return step_explicit<float>(un, vn, un1, vn1, p, u_bc, v_bc, p_bc, u_out, v_out, p_copy, info, hint, work, batch, nx, ny, dt, dx, dy, rho, nu, beta, tol, max_sweeps,
                                corrected, (hipStream_t)stream);
}
NNS_API int nns_fd_step_explicit_f64(const double* un, const double* vn, const double* un1, const double* vn1, double* p, const nns_bc_list* u_bc,
                                     const nns_bc_list* v_bc, const nns_bc_list* p_bc, double* u_out, double* v_out, double* p_copy, double* info, const double* hint, void* work, int batch,
                                     int nx, int ny, double dt, double dx, double dy, double rho, double nu, double beta, double tol, int max_sweeps, int corrected,
                                     void* stream) {
    return step_explicit<double>(un, vn, un1, vn1, p, u_bc, v_bc, p_bc, u_out, v_out, p_copy, info, hint, work, batch, nx, ny, dt, dx, dy, rho, nu, beta, tol, max_sweeps,
                                 corrected, (hipStream_t)stream);
}
