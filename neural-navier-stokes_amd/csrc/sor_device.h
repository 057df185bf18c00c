// The lexicographic SOR pipelines of chorin_fd._get_pressure (src/chorin_fd/simulate.py:190-200) as device functions of ONE workgroup, shared by
// the solver kernel (csrc/sor_kernels.hip) and the fused explicit step (csrc/fd_step_kernels.hip).  Compile with -ffp-contract=off.
#pragma once
#include "nns_common.h"
#include <cmath>

namespace nns {
namespace sorlex {

constexpr int kSorThreads = 1024;                 // 16 waves
constexpr int kSorWaves = kSorThreads / kWave;
constexpr int kSorBatch = 128;                     // sweeps per speculative batch at most (row-per-lane pipeline; the LDS-exchange pipeline runs kSorWaves)
constexpr int kSorHdr = kSorBatch * 8 + 128;       // LDS header: per-sweep errs + stop flag

template <typename T>
struct SorK { T dx2, dy2, den, beta, omb, tol, rcp; };       // rcp: RN(1 / den) where div_den's short form applies, else 0 (make_sor_k)

template <typename T>
__device__ __forceinline__ T nanmax(T a, T b) { return (b > a || b != b) ? b : a; }

// Runs sweeps [0, nsw) pipelined on pw (LDS or global); errs[s] = max update of sweep s.
template <typename T>
__device__ __forceinline__ void sor_batch(T* pw, const T* cw, int nx, int ny, int nsw, const SorK<T>& k, T* errs) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int nfronts = nx + ny - 5;                       // d = 2 .. nx+ny-4
    const int nsteps = nfronts + 2 * (nsw - 1);
    T emax = (T)0;
    for (int t = 0; t < nsteps; ++t) {
        const int f = t - 2 * wave;
        if (wave < nsw && f >= 0 && f < nfronts) {
            const int d = f + 2;
            const int ilo = max(1, d - (ny - 2)), ihi = min(nx - 2, d - 1);
            for (int i = ilo + lane; i <= ihi; i += kWave) {
                const int c = i * ny + (d - i);
                const T old = pw[c];
                const T nw = (k.beta * (k.dy2 * pw[c + ny] + k.dy2 * pw[c - ny] + k.dx2 * pw[c + 1] + k.dx2 * pw[c - 1] - cw[c]) / k.den +
                              k.omb * old);                                           // :193-196
                pw[c] = nw;
                emax = nanmax<T>(emax, fabs(nw - old));
            }
        }
        __syncthreads();
    }
    for (int o = kWave / 2; o > 0; o >>= 1) emax = nanmax<T>(emax, __shfl_down(emax, o));
    if (lane == 0 && wave < nsw) errs[wave] = emax;
    __syncthreads();
}

// Round 4: the same pipeline with every ROW of the grid owned by one LANE (nx - 2 <= 64: the reference's 51 x 51 and 64 x 64 grids).
//   * Lane i - 1 marches along row i, one column per front, so both same-sweep neighbours of a point are already in registers: p[i][j-1] is the
//     lane's own previous result, p[i-1][j] the previous result of the lane before it (one DPP rotate).  What is left to LDS are the PREVIOUS
//     sweep's values (p[i][j+1], p[i+1][j], p[i][j]) and C -- and with a lag of three fronts between consecutive sweeps instead of two those were
//     written two steps ago, so they are requested a step AHEAD and the load latency leaves the critical path.
//   * A front of one sweep keeps on average HALF the lanes of its wave busy, and the kernel is bound by vector-instruction issue on its one CU
//     (16 waves x ~45 instructions, 28 of them float64, per step on four SIMDs: ~800 cycles per step measured).  A lane that has finished its row
//     of sweep s therefore goes straight on to the same row of sweep s + 16 (the wave's next one): the rows still open in sweep s are the HIGH
//     ones, those already open in sweep s + 16 the LOW ones -- complementary when the two are ny - 2 fronts apart -- so every lane computes a
//     point in every step and 49 sweeps take 3 (ny - 2) + nfronts steps instead of 4 (nfronts + 45).
// Same operations on the same operands in the same order as sor_batch: bitwise the same p, errs and sweep count (tools/sor_ab.py).
#ifndef NNS_SOR_ROWS
#define NNS_SOR_ROWS 1
#endif
#ifndef NNS_SOR_TIMING
#define NNS_SOR_TIMING 0            // 1: the row-per-lane pipeline prints the cycles of a step's parts (s_memtime stamps, waves 0 and 7)
#endif
constexpr int kSorLag = 3;

__device__ __forceinline__ float lane_before(float x) {       // lane l <- lane l - 1 (wave rotate right by one)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x13C, 0xF, 0xF, true));
}
__device__ __forceinline__ double lane_before(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    // (a whole-wave rotate reads a valid lane everywhere: no `old` operand to initialise)
    const int lo = __builtin_amdgcn_mov_dpp((int)b, 0x13C, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x13C, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// Sweep s runs on wave s % 16 and starts (its front 0) at step  T_s = (s / 16) P + (s % 16) kSorLag,  P = max(16 kSorLag, ny - 2, nx - 2):
// consecutive sweeps are >= kSorLag steps apart, consecutive sweeps of ONE wave P steps -- a lane needs ny - 2 of them for its row, and
// P >= nx - 2 keeps a lane's finished-row maximum in place until its wave has reduced that sweep's error.
// x / k.den for the row-per-lane pipeline.  The divisor is one constant per solve, so the IEEE quotient can be had from its correctly rounded
// reciprocal y = RN(1 / den) (computed on the host) by Markstein's correction  q = RN(x y), r = x - q den (exact in an FMA), RN(q + r y)  -- three
// instructions where the compiler's division is thirteen (v_div_scale x 2, v_rcp, two Newton steps, v_div_fmas, v_div_fixup), a quarter of the
// point update this issue-bound kernel is made of.  Used only where no intermediate can leave the normal range (|x| and den guarded; zeros,
// infinities and NaNs take the plain division: -0 / den must stay -0); tools/fastdiv_check.hip compared it with `/` BITWISE on 1.4e10 random
// operands per type over 51 divisors (the reference's grids, random ones, significands of nearly all ones): no mismatch.
#ifndef NNS_SOR_FASTDIV
#define NNS_SOR_FASTDIV 1
#endif
template <typename T>
__device__ __forceinline__ T div_den(T x, const SorK<T>& k) {
#if NNS_SOR_FASTDIV
    constexpr T lo = sizeof(T) == 8 ? (T)1e-250 : (T)1e-25, hi = sizeof(T) == 8 ? (T)1e250 : (T)1e25;
    const T ax = fabs(x);
    const T q = x * k.rcp;
    T res = ax == (T)0 ? x : fma(fma(-q, k.den, x), k.rcp, q);              // +-0 / den = +-0 (den > 0): a cavity at rest is zeros for many steps
    if (!(k.rcp != (T)0 && ((ax >= lo && ax <= hi) || ax == (T)0))) res = x / k.den;      // rare: the wave skips it when no lane needs it
    return res;
#else
    return x / k.den;
#endif
}

template <typename T>
__device__ __forceinline__ void sor_batch_rows(T* pw, const T* cw, int nx, int ny, int nsw, const SorK<T>& k, T* errs) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int i = lane + 1, row = i * ny, ncol = ny - 2;
    const bool has_row = i <= nx - 2;
    const int nfronts = nx + ny - 5;
    const int P = max(kSorWaves * kSorLag, max(ncol, nx - 2));
    const int nsteps = ((nsw - 1) / kSorWaves) * P + ((nsw - 1) % kSorWaves) * kSorLag + nfronts + 1;       // the last step reduces the last sweep's error
    // this lane's clock: pos < 0 waiting, 0 <= pos < ncol column pos + 1 of sweep `sweep`, then idle until pos == P starts the wave's next sweep
    int pos = -(i - 1) - wave * kSorLag, sweep = wave;
    // the wave's clock for the error reductions: sweep `esweep` is complete when epos reaches 0
    int epos = -wave * kSorLag - nfronts, esweep = wave;
    T emax = (T)0, edone = (T)0, own = (T)0;
    T qe = (T)0, qs = (T)0, qo = (T)0, qc = (T)0, qn = (T)0, qw = (T)0;       // the operands of this lane's next point, requested a step ahead
    // (pos < 0 is a huge unsigned number: one comparison for 0 <= pos < ncol)
    auto live = [&](int ps, int sw) { return has_row && sw < nsw && (unsigned)ps < (unsigned)ncol; };
    auto request = [&](int ps) {
        const int c = row + ps + 1;
        qe = pw[c + 1]; qs = pw[c + ny]; qo = pw[c]; qc = cw[c]; qn = pw[c - ny]; qw = pw[c - 1];
    };
    bool act = live(pos, sweep);                                               // this step's activity = what last step's request was issued under
    if (act) request(pos);
#if NNS_SOR_TIMING
    long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tcomp = 0, treq = 0, tbar = 0;
#endif
    for (int t = 0; t < nsteps; ++t) {
#if NNS_SOR_TIMING
        const bool timed = t >= 100 && t < 164 && blockIdx.x == 0;
        if (timed) tq0 = clock64();
#endif
        if (epos == 0) {                                                       // wave-uniform: every row of sweep esweep is done, no lane has finished another since
            if (esweep < nsw) {
                T e = edone;
                for (int o = kWave / 2; o > 0; o >>= 1) e = nanmax<T>(e, __shfl_down(e, o));
                if (lane == 0) errs[esweep] = e;
            }
            epos = -P; esweep += kSorWaves;
        }
        const T north = lane_before(own);                                      // p[i-1][j] of THIS sweep (before any lane moves on)
        if (act) {
            const T n_ = i == 1 ? qn : north, w_ = pos == 0 ? qw : own;          // boundary values come from the grid, interior ones from registers
            const T nw = (div_den<T>(k.beta * (k.dy2 * qs + k.dy2 * n_ + k.dx2 * qe + k.dx2 * w_ - qc), k) + k.omb * qo);  // :193-196
            pw[row + pos + 1] = nw;
            own = nw;
            emax = nanmax<T>(emax, fabs(nw - qo));
            if (pos == ncol - 1) { edone = emax; emax = (T)0; }                // the row is finished: its maximum waits for the wave's reduction
        }
        ++pos; ++epos;
        if (pos == P) { pos = 0; sweep += kSorWaves; }
#if NNS_SOR_TIMING
        if (timed) { __builtin_amdgcn_s_waitcnt(0xc07f); tq1 = clock64(); }
#endif
        act = live(pos, sweep);
        if (act) request(pos);
#if NNS_SOR_TIMING
        if (timed) { __builtin_amdgcn_s_waitcnt(0xc07f); tq2 = clock64(); }
#endif
        __syncthreads();
#if NNS_SOR_TIMING
        if (timed) { tq3 = clock64(); tcomp += tq1 - tq0; treq += tq2 - tq1; tbar += tq3 - tq2; }
#endif
    }
#if NNS_SOR_TIMING
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7) && nsw > 40)
        printf("sor rows, wave %d, mean of steps 100..163 (cycles): compute (to the store's completion) %ld, request + its wait %ld, barrier %ld\n", wave, tcomp / 64, treq / 64, tbar / 64);
#endif
    __syncthreads();
}

template <bool IN_LDS> __device__ __forceinline__ bool sor_rows_path(int nx) { return IN_LDS && NNS_SOR_ROWS && nx - 2 <= kWave; }
template <typename T, bool IN_LDS>
__device__ __forceinline__ void sor_run_batch(T* pw, const T* cw, int nx, int ny, int nsw, const SorK<T>& k, T* errs) {
    if (sor_rows_path<IN_LDS>(nx)) sor_batch_rows<T>(pw, cw, nx, ny, nsw, k, errs);
    else sor_batch<T>(pw, cw, nx, ny, nsw, k, errs);
}


// The whole solve of ONE grid by one workgroup of kSorThreads: pw / cw = the working copies of p and C (LDS when IN_LDS, else the global grid
// itself), sg = the grid's snapshot buffer (global, nx * ny), errs / s_stop_p = the LDS header.  Runs speculative batches of sweeps until the first
// sweep whose maximum update is <= tol (kept exact: restore + replay) or max_sweeps; returns the sweeps done and the last error (:183-200).
template <typename T, bool IN_LDS>
__device__ __forceinline__ void sor_solve(T* pw, const T* cw, T* sg, int nx, int ny, int max_sweeps, int expect, const SorK<T>& k, T* errs, int* s_stop_p,
                                          int& done_out, T& err_out) {
    const int n = nx * ny, tid = threadIdx.x;
    int done = 0, nbatch = 0;
    T err = (T)1;                                                     // :183
    while (done < max_sweeps) {
        // Sweeps run speculatively in batches (a stop inside a batch costs a restore + replay), so the batch sizes follow what is known: `expect`
        // (the sweep count of the previous solve of this grid, 0 = unknown) for the first one -- a time loop's counts change slowly, and a batch
        // that ends exactly at the stop needs no replay --, one sweep per wave for the second, whatever is left (up to the cap) after that.
        // The RESULT does not depend on the batch sizes.
        int want = nbatch == 0 ? (expect >= 1 ? expect : kSorWaves) : nbatch == 1 ? kSorWaves : kSorBatch;
        if (!sor_rows_path<IN_LDS>(nx)) want = kSorWaves;
        const int nsw = min(min(want, sor_rows_path<IN_LDS>(nx) ? kSorBatch : kSorWaves), max_sweeps - done);
        ++nbatch;
        for (int c = tid; c < n; c += kSorThreads) sg[c] = pw[c];      // snapshot for an exact early stop
        __syncthreads();
        sor_run_batch<T, IN_LDS>(pw, cw, nx, ny, nsw, k, errs);
        if (tid == 0) {
            int stop = -1;
            for (int s = 0; s < nsw; ++s) if (!(errs[s] > k.tol)) { stop = s; break; }    // loop runs while err > tol
            *s_stop_p = stop;
        }
        __syncthreads();
        const int stop = *s_stop_p;
        if (stop < 0) { done += nsw; err = errs[nsw - 1]; __syncthreads(); continue; }
        if (stop < nsw - 1) {                                          // overshoot: restore and replay stop+1 sweeps
            const T e_keep = errs[stop];
            __syncthreads();
            for (int c = tid; c < n; c += kSorThreads) pw[c] = sg[c];
            __syncthreads();
            sor_run_batch<T, IN_LDS>(pw, cw, nx, ny, stop + 1, k, errs);
            err = e_keep;
        } else {
            err = errs[stop];
        }
        done += stop + 1;
        break;
    }
    __syncthreads();
    done_out = done; err_out = err;
}

template <typename T>
inline SorK<T> make_sor_k(double dx, double dy, double beta, double tol) {
    SorK<T> k{(T)(dx * dx), (T)(dy * dy), (T)(2 * (dx * dx) + 2 * (dy * dy)), (T)beta, (T)(1 - beta), (T)tol, (T)0};
    // the short division needs a positive divisor well inside the normal range (its reciprocal and every x - q den too)
    const double ad = (double)k.den, dlo = sizeof(T) == 8 ? 1e-50 : 1e-10, dhi = sizeof(T) == 8 ? 1e50 : 1e10;
    if (ad >= dlo && ad <= dhi) k.rcp = (T)1 / k.den;
    return k;
}

inline size_t sor_lds_bytes(int nx, int ny, size_t elem) { return kSorHdr + 2 * (size_t)nx * ny * elem; }
constexpr size_t kSorLdsMax = 150 * 1024;

}  // namespace sorlex
}  // namespace nns
