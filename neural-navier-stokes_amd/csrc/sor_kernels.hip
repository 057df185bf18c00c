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
#include <cmath>

using namespace nns;

namespace {

constexpr int kSorThreads = 1024;                 // 16 waves
constexpr int kSorWaves = kSorThreads / kWave;
constexpr int kSorBatch = 128;                     // sweeps per speculative batch at most (row-per-lane pipeline; the LDS-exchange pipeline runs kSorWaves)
constexpr int kSorHdr = kSorBatch * 8 + 128;       // LDS header: per-sweep errs + stop flag

template <typename T>
struct SorK { T dx2, dy2, den, beta, omb, tol, rcp; };       // rcp: RN(1 / den) where div_den's short form applies, else 0 (set by sor())

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
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x13C, 0xF, 0xF, false));
}
__device__ __forceinline__ double lane_before(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x13C, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x13C, 0xF, 0xF, false);
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
__device__ __forceinline__ void sor_batch_rows(T* pw, const T* cw, int nx, int ny, int pitch, int nsw, const SorK<T>& k, T* errs) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int i = lane + 1, row = i * pitch, ncol = ny - 2;
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
    auto request = [&](int ps, int sw) {
        if (has_row && sw < nsw && ps >= 0 && ps < ncol) {
            const int c = row + ps + 1;
            qe = pw[c + 1]; qs = pw[c + pitch]; qo = pw[c]; qc = cw[c]; qn = pw[c - pitch]; qw = pw[c - 1];
        }
    };
    request(pos, sweep);
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
        if (has_row && sweep < nsw && pos >= 0 && pos < ncol) {
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
        request(pos, sweep);
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
// LDS row pitch of the row-per-lane pipeline: its lanes access ONE column of consecutive rows, so a pitch of 2^k elements (the 64 x 64 grid) would
// put a whole wave on one bank pair; an odd pitch spreads 8-byte elements over all 64 banks (4-byte ones over 32 of them at pitch = 1 mod 64 ...)
#ifndef NNS_SOR_PITCH
#define NNS_SOR_PITCH 0              // measured (round 4, 64 x 64 float64): even paddings change nothing (the kernel is bound by instruction issue, not by LDS banks), an ODD pitch is 3x slower (rows that are not 16-byte aligned: the adjacent-pair ds_read2_b64)
#endif
#ifndef NNS_SOR_PITCH_ADD
#define NNS_SOR_PITCH_ADD -1          // developer probe: >= 0 pads every row by that many elements instead of making the pitch odd
#endif
__host__ __device__ inline int sor_pitch(int nx, int ny, bool in_lds) {
    if (!(NNS_SOR_PITCH && in_lds && NNS_SOR_ROWS && nx - 2 <= kWave)) return ny;
    return NNS_SOR_PITCH_ADD >= 0 ? ny + NNS_SOR_PITCH_ADD : (ny | 1);
}

template <typename T, bool IN_LDS>
__device__ __forceinline__ void sor_run_batch(T* pw, const T* cw, int nx, int ny, int pitch, int nsw, const SorK<T>& k, T* errs) {
    if (sor_rows_path<IN_LDS>(nx)) sor_batch_rows<T>(pw, cw, nx, ny, pitch, nsw, k, errs);
    else sor_batch<T>(pw, cw, nx, ny, nsw, k, errs);
}

template <typename T, bool IN_LDS>
__global__ __launch_bounds__(kSorThreads) void sor_kernel(T* __restrict__ p, const T* __restrict__ C, T* __restrict__ info,
                                                           T* __restrict__ snap, int nx, int ny, int max_sweeps, SorK<T> k) {
    // all LDS in the dynamic region (16-byte aligned carve): [errs 16 x 8 B][stop][pad to 256][p][C]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* errs = reinterpret_cast<T*>(smem_raw);
    int* s_stop_p = reinterpret_cast<int*>(smem_raw + kSorBatch * 8);
    const int n = nx * ny, tid = threadIdx.x;
    T* pg = p + (size_t)blockIdx.x * n;
    const T* cg = C + (size_t)blockIdx.x * n;
    T* sg = snap + (size_t)blockIdx.x * n;
    T* pw = pg;
    const T* cw = cg;
    const int pitch = sor_pitch(nx, ny, IN_LDS);
    auto at = [&](int c) { return pitch == ny ? c : (c / ny) * pitch + c % ny; };      // element c of the dense grid in the working copy
    if (IN_LDS) {
        T* pl = reinterpret_cast<T*>(smem_raw + kSorHdr);
        T* cl = pl + nx * pitch;
        for (int c = tid; c < n; c += kSorThreads) { pl[at(c)] = pg[c]; cl[at(c)] = cg[c]; }
        pw = pl; cw = cl;
        __syncthreads();
    }
    int done = 0;
    T err = (T)1;                                                     // :183
    while (done < max_sweeps) {
        const int nsw = min(sor_rows_path<IN_LDS>(nx) ? kSorBatch : kSorWaves, max_sweeps - done);
        for (int c = tid; c < n; c += kSorThreads) sg[c] = pw[at(c)];  // snapshot for an exact early stop
        __syncthreads();
        sor_run_batch<T, IN_LDS>(pw, cw, nx, ny, pitch, nsw, k, errs);
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
            for (int c = tid; c < n; c += kSorThreads) pw[at(c)] = sg[c];
            __syncthreads();
            sor_run_batch<T, IN_LDS>(pw, cw, nx, ny, pitch, stop + 1, k, errs);
            err = e_keep;
        } else {
            err = errs[stop];
        }
        done += stop + 1;
        break;
    }
    __syncthreads();
    if (IN_LDS) for (int c = tid; c < n; c += kSorThreads) pg[c] = pw[at(c)];
    if (tid == 0) { info[2 * blockIdx.x] = (T)done; info[2 * blockIdx.x + 1] = err; }
}

template <typename T>
int sor(T* p, const T* C, T* info, void* work, int batch, int nx, int ny, double dx, double dy, double beta, double tol,
        int max_sweeps, hipStream_t s) {
    if (!p || !C || !info || !work || !field_args_ok(batch, nx, ny) || max_sweeps < 0)
        return fail(NNS_ERR_INVALID_ARG, "fd_sor: bad args (batch=%d nx=%d ny=%d max_sweeps=%d)", batch, nx, ny, max_sweeps);
    SorK<T> k{(T)(dx * dx), (T)(dy * dy), (T)(2 * (dx * dx) + 2 * (dy * dy)), (T)beta, (T)(1 - beta), (T)tol, (T)0};
    {   // the short division needs a divisor well inside the normal range (its reciprocal and every x - q den too)
        const double ad = std::fabs((double)k.den), dlo = sizeof(T) == 8 ? 1e-50 : 1e-10, dhi = sizeof(T) == 8 ? 1e50 : 1e10;
        if (ad >= dlo && ad <= dhi && k.den > (T)0) k.rcp = (T)1 / k.den;
    }
    size_t lds = kSorHdr + 2 * (size_t)nx * sor_pitch(nx, ny, true) * sizeof(T);
    if (lds > 150 * 1024) lds = kSorHdr + 2 * (size_t)nx * ny * sizeof(T);        // (cannot happen for nx <= 66; keeps the two sides of the choice in one place)
    T* snap = reinterpret_cast<T*>(work);
    if (kSorHdr + 2 * (size_t)nx * ny * sizeof(T) <= 150 * 1024 && lds <= 150 * 1024) {
        static bool attr = false;                     // set once to the largest size used (keeps the launch path free of
        if (!attr) {                                  // non-stream API calls, e.g. under hipGraph capture)
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sor_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "fd_sor: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr = true;
        }
        hipLaunchKernelGGL((sor_kernel<T, true>), dim3(batch), dim3(kSorThreads), lds, s, p, C, info, snap, nx, ny, max_sweeps, k);
    } else {
        hipLaunchKernelGGL((sor_kernel<T, false>), dim3(batch), dim3(kSorThreads), kSorHdr, s, p, C, info, snap, nx, ny, max_sweeps, k);
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
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sor_redblack_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
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
    return sor<float>(p, C, info, work, batch, nx, ny, dx, dy, beta, tol, max_sweeps, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_sor_f64(double* p, const double* C, double* info, void* work, int batch, int nx, int ny, double dx, double dy,
                           double beta, double tol, int max_sweeps, void* stream) {
    return sor<double>(p, C, info, work, batch, nx, ny, dx, dy, beta, tol, max_sweeps, reinterpret_cast<hipStream_t>(stream));
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
