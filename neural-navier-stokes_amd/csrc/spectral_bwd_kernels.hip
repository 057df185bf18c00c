// Backward (vector-Jacobian product) of the Fourier-spectral residual, gfx950: the conjugate spectral multiplies.
// Operator definition: oracle/periodic.py (spectral_residual_vjp); SURVEY.md section 8 (f) rank 2.
//
// With a = g_u, b = g_v, d = g_div, D^T = -D, L^T = L (see the oracle), the product separates by direction exactly
// like the forward:
//   x-pass (columns):  GU = a u_x + b v_x - [D_x(a u + d) + nu L_x a],  GV = -[D_x(b u) + nu L_x b],  GP = -(D_x a)/rho
//   y-pass (rows):     grad_u = GU + a/dt           - [D_y(a v)     + nu L_y a]
//                      grad_v = GV + b/dt + a u_y + b v_y - [D_y(b v + d) + nu L_y b]
//                      grad_p = GP - (D_y b)/rho,   grad_u_prev = -a/dt,  grad_v_prev = -b/dt (optional)
// Per line three packed forward transforms (float64, or float32 on forward-differenced lines: the forward's `precise` policy) and three
// inverse ones (float32), on the forward's FFT engine:
//   Z3 = FFT(f + i g)   ->  ifft(i k Z3)            = (f', g')            (f, g) = (u, v)
//   Z2 = FFT(a + i b)   ->  ifft(i k Z2)            = (a', b')            one component is used per pass
//   Z1 = FFT(s1 + i s2) ->  ifft(i k Z1 - nu k^2 Z2) = (D s1 + nu L a, D s2 + nu L b)
// HBM traffic (fp32): x-pass 5 in + 3 out = 32 B/pt, y-pass 8 in + 3..5 out = 44..52 B/pt.
#include "spectral_common.h"

using namespace nns;
using namespace nns::spec;

namespace {

struct AdjK {
    double c1;            // kscale / N
    double c2;            // nu kscale^2 / N
    float inv_rho, inv_dt;
};

// One line.  In: (f, g), (a, b), (s1, s2) as element tid + TPF*m in slot m.  Out: w = a f' + b g', da = a' (USE_A) or b',
// (cu, cv) = (D s1 + nu L a, D s2 + nu L b).
template <int N, typename TF, bool USE_A>
__device__ __forceinline__ void adj_core(const float (&ff)[16], const float (&gf)[16], const float (&af)[16], const float (&bf)[16],
                                         const float (&s1)[16], const float (&s2)[16],
                                         float (&w)[16], float (&da)[16], C2<float> (&c)[16],
                                         const C2<TF>* tabF, const C2<float>* tabI, unsigned char* xb_raw, int tid, const AdjK& k) {
    const C2<TF>* tabF2 = tabF + N / 2;
    const C2<float>* tabI2 = tabI + N / 2;
    C2<TF>* xbF = reinterpret_cast<C2<TF>*>(xb_raw);
    C2<float>* xbI = reinterpret_cast<C2<float>*>(xb_raw);
    C2<float> e[16];
#ifndef NNS_F32_DIFF
#define NNS_F32_DIFF 1
#endif
    if constexpr (sizeof(TF) == 4 && NNS_F32_DIFF) {
        // all-float32 mode, as the forward's (spectral_kernels.hip, deriv_core): every packed pair is forward-DIFFERENCED in physical space,
        // FFT(d) = (e^{i theta} - 1) FFT(f), and the spectral multiplies become the bounded filters
        //     i k   FFT(f) = M1 D,  M1 = (k / 2)(cot(theta / 2) - i);      -nu k^2 FFT(f) = (nu k / 2)(k + i k cot(theta / 2)) D
        const float* ctab = reinterpret_cast<const float*>(tabI + N / 2 + Pass2<N>::ENTRIES);
        const float c1h = (float)(0.5 * k.c1), c2h = (float)(0.5 * k.c2);
        auto diff_fft = [&](const float (&p)[16], const float (&q)[16], C2<float> (&zv)[16]) {
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                zv[m].x = right_of<m, N / 16>(p, tid) - p[m];
                zv[m].y = right_of<m, N / 16>(q, tid) - q[m];
            });
            fft_line<float, N, false>(zv, tabI, tabI2, xbI, tid);
        };
        C2<float> zv[16];
        // ---- (f', g')
        diff_fft(ff, gf, zv);
        int te = tid;
        asm volatile("" : "+v"(te), "+v"(zv[0].x));
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            int ko, ke;
            wavenumber<N, m>(te, ko, ke);
            const float ar = ctab[ke < 0 ? -ke : ke] * c1h, ai = (float)ko * c1h;
            e[m].x = ar * zv[m].x + ai * zv[m].y; e[m].y = ar * zv[m].y - ai * zv[m].x;
        });
        __builtin_amdgcn_sched_barrier(0);
        fft_line<float, N, true>(e, tabI, tabI2, xbI, tid);
#pragma unroll
        for (int m = 0; m < 16; ++m) w[m] = af[m] * e[m].x + bf[m] * e[m].y;
        __builtin_amdgcn_sched_barrier(0);
        // ---- a', b' and the viscous term  -nu k^2 Z2  (kept in c)
        diff_fft(af, bf, zv);
        te = tid;
        asm volatile("" : "+v"(te), "+v"(zv[0].x));
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            int ko, ke;
            wavenumber<N, m>(te, ko, ke);
            const float ct = ctab[ke < 0 ? -ke : ke];
            const float ar = ct * c1h, ai = (float)ko * c1h;
            e[m].x = ar * zv[m].x + ai * zv[m].y; e[m].y = ar * zv[m].y - ai * zv[m].x;
            const float kf = (float)ke * c2h;
            const float br = kf * (float)ke, bi = kf * ct;                   // -nu k^2 Z2 = (br + i bi) D
            c[m].x = br * zv[m].x - bi * zv[m].y; c[m].y = br * zv[m].y + bi * zv[m].x;
        });
        __builtin_amdgcn_sched_barrier(0);
        fft_line<float, N, true>(e, tabI, tabI2, xbI, tid);
#pragma unroll
        for (int m = 0; m < 16; ++m) da[m] = USE_A ? e[m].x : e[m].y;
        __builtin_amdgcn_sched_barrier(0);
        // ---- D s1 + nu L a,  D s2 + nu L b
        diff_fft(s1, s2, zv);
        te = tid;
        asm volatile("" : "+v"(te), "+v"(zv[0].x));
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            int ko, ke;
            wavenumber<N, m>(te, ko, ke);
            const float ar = ctab[ke < 0 ? -ke : ke] * c1h, ai = (float)ko * c1h;
            c[m].x += ar * zv[m].x + ai * zv[m].y; c[m].y += ar * zv[m].y - ai * zv[m].x;
        });
        __builtin_amdgcn_sched_barrier(0);
        fft_line<float, N, true>(c, tabI, tabI2, xbI, tid);
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    C2<TF> z[16];
    // ---- (f', g')
#pragma unroll
    for (int m = 0; m < 16; ++m) { z[m].x = (TF)ff[m]; z[m].y = (TF)gf[m]; }
    fft_line<TF, N, false>(z, tabF, tabF2, xbF, tid);
    int te = tid;
    asm volatile("" : "+v"(te), "+v"(z[0].x));
    static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        int ko, ke;
        wavenumber<N, m>(te, ko, ke);
        const TF k1 = (TF)((double)ko * k.c1);
        e[m].x = (float)(-k1 * z[m].y); e[m].y = (float)(k1 * z[m].x);
    });
    __builtin_amdgcn_sched_barrier(0);
    fft_line<float, N, true>(e, tabI, tabI2, xbI, tid);
#pragma unroll
    for (int m = 0; m < 16; ++m) w[m] = af[m] * e[m].x + bf[m] * e[m].y;
    __builtin_amdgcn_sched_barrier(0);
    // ---- a', b' and the viscous term  -nu k^2 Z2  (kept in c)
#pragma unroll
    for (int m = 0; m < 16; ++m) { z[m].x = (TF)af[m]; z[m].y = (TF)bf[m]; }
    fft_line<TF, N, false>(z, tabF, tabF2, xbF, tid);
    te = tid;
    asm volatile("" : "+v"(te), "+v"(z[0].x));
    static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        int ko, ke;
        wavenumber<N, m>(te, ko, ke);
        const TF k1 = (TF)((double)ko * k.c1);
        const TF k2 = (TF)((double)(ke * ke) * k.c2);
        e[m].x = (float)(-k1 * z[m].y); e[m].y = (float)(k1 * z[m].x);
        c[m].x = (float)(-k2 * z[m].x); c[m].y = (float)(-k2 * z[m].y);
    });
    __builtin_amdgcn_sched_barrier(0);
    fft_line<float, N, true>(e, tabI, tabI2, xbI, tid);
#pragma unroll
    for (int m = 0; m < 16; ++m) da[m] = USE_A ? e[m].x : e[m].y;
    __builtin_amdgcn_sched_barrier(0);
    // ---- D s1 + nu L a,  D s2 + nu L b
#pragma unroll
    for (int m = 0; m < 16; ++m) { z[m].x = (TF)s1[m]; z[m].y = (TF)s2[m]; }
    fft_line<TF, N, false>(z, tabF, tabF2, xbF, tid);
    te = tid;
    asm volatile("" : "+v"(te), "+v"(z[0].x));
    static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        int ko, ke;
        wavenumber<N, m>(te, ko, ke);
        const TF k1 = (TF)((double)ko * k.c1);
        c[m].x += (float)(-k1 * z[m].y); c[m].y += (float)(k1 * z[m].x);
    });
    __builtin_amdgcn_sched_barrier(0);
    fft_line<float, N, true>(c, tabI, tabI2, xbI, tid);
    __builtin_amdgcn_sched_barrier(0);
}

// ------------------------------------------------------------------------------------------------------------------
// y-pass: rows.  One line per TPF lanes; grid-stride over all batch*nx rows.  gu, gv, gp hold the x-pass partials on
// entry and the gradients on exit.
// ------------------------------------------------------------------------------------------------------------------
template <int N, typename TF>
__global__ __launch_bounds__(kSpecThreads) void spec_bwd_ypass_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                                       const float* __restrict__ ga, const float* __restrict__ gb, const float* __restrict__ gd,
                                                                       float* __restrict__ gu, float* __restrict__ gv, float* __restrict__ gp,
                                                                       float* __restrict__ gup, float* __restrict__ gvp, long nrows, AdjK k) {
    using L = SpecLds<N, TF>;
    constexpr int TPF = L::TPF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<TF>* tabF; C2<float>* tabI; unsigned char* lines;
    spec_setup<N, TF>(smem, tabF, tabI, lines);
    const long niter = (nrows + L::LINES - 1) / L::LINES;
    for (long it = blockIdx.x; it < niter; it += gridDim.x) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave;
        const int sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * L::LINE_BYTES;
        const long row_raw = it * L::LINES + line;
        const bool valid = row_raw < nrows;
        const long row = valid ? row_raw : nrows - 1;
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        const size_t base = (size_t)row * N + tidv;
        float vf[16], af[16], bf[16], s1[16], s2[16], w[16], db[16];
        C2<float> c[16];
        {
            float uf[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const size_t q = base + TPF * m;
                uf[m] = u[q]; vf[m] = v[q]; af[m] = ga[q]; bf[m] = gb[q]; s2[m] = gd[q];
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) { s1[m] = af[m] * vf[m]; s2[m] = bf[m] * vf[m] + s2[m]; }      // a v,  b v + d
            adj_core<N, TF, false>(uf, vf, af, bf, s1, s2, w, db, c, tabF, tabI, xb, tidv, k);
        }
        // epilogue in two halves: partials in, gradients out
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float pu[8], pv[8], pp[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const size_t q = base + TPF * (8 * h + i); pu[i] = gu[q]; pv[i] = gv[q]; pp[i] = gp[q]; }
            if (valid) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = 8 * h + i;
                    const size_t q = base + TPF * m;
                    gu[q] = pu[i] + af[m] * k.inv_dt - c[m].x;
                    gv[q] = pv[i] + bf[m] * k.inv_dt + w[m] - c[m].y;
                    gp[q] = pp[i] - db[m] * k.inv_rho;
                    if (gup) gup[q] = -af[m] * k.inv_dt;
                    if (gvp) gvp[q] = -bf[m] * k.inv_dt;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// x-pass: columns.  A workgroup owns LINES adjacent columns of one grid; the five input fields go through the LDS
// transpose stage in two rounds (u, v, a then b, d: the stage holds three fields), the three partials come back in one.
// ------------------------------------------------------------------------------------------------------------------
// WAVES (round 4): lines -- and waves -- per workgroup.  With 8 a CU holds ONE workgroup whose eight waves walk load / stage / transform / store in
// lockstep (the barriers of the staging steps): the vector pipe idles while the tile's 80 row pieces per thread are in flight and the memory system
// idles under the six transforms -- 1.7 TB/s, 34 % of wave-cycles issuing (profiles/r04_specbwd_summary.txt).  With 4 the LDS of a workgroup halves
// and a CU holds TWO that run out of step, one transforming while the other moves its tile -- but its tiles are 4 columns wide: 16-byte row pieces,
// twice the row pieces per byte.  MEASURED (profiles/r04_ab_specbwd_xwaves.txt, same box, three rounds): whole backward 2.10 ms with 8, 2.17 with 4:
// the access shape costs more than the overlap buys.  NNS_BWD_XWAVES=4 in the environment selects it for re-measurement.
template <int N, typename TF, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void spec_bwd_xpass_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                                       const float* __restrict__ ga, const float* __restrict__ gb, const float* __restrict__ gd,
                                                                       float* __restrict__ gu, float* __restrict__ gv, float* __restrict__ gp,
                                                                       int ny, int tiles_per_grid, long ntiles, AdjK k) {
    using L = SpecLds<N, TF, WAVES>;
    constexpr int TPF = L::TPF, CW = L::LINES, SF = L::STAGE_F;
    constexpr int ROWS_PER_IT = WAVES * kWave / CW;
    constexpr int NR = N / ROWS_PER_IT;
    static_assert(NR == 16, "staging geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<TF>* tabF; C2<float>* tabI; unsigned char* lines;
    spec_setup<N, TF, WAVES * kWave>(smem, tabF, tabI, lines);
#ifndef NNS_BWDX_TIMING
#define NNS_BWDX_TIMING 0          // 1: wave 0 of workgroup 0 prints the cycles (s_memtime) of the phases of its fourth tile
#endif
#if NNS_BWDX_TIMING
    long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BWDX_STAMP(i) if (t == blockIdx.x + 3 * (long)gridDim.x) { __builtin_amdgcn_s_waitcnt(0); tk[i] = clock64(); }
#else
#define BWDX_STAMP(i)
#endif
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave;
        const int sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * L::LINE_BYTES;
        float* my_stage = reinterpret_cast<float*>(xb) + (line % L::SKEW_MOD) * L::SKEW_DW;
        const long lt = ntiles - 1 - (long)xcd_remap((unsigned)t, (unsigned)ntiles);      // last grid first: see spectral_kernels.hip launch_xpass
        const int j0 = (int)(lt % tiles_per_grid) * CW;
        const size_t g = (size_t)(lt / tiles_per_grid) * N * ny;
        const int cc = tx % CW, cr = tx / CW;
        const int col = j0 + cc < ny ? j0 + cc : ny - 1;                               // clamped: loads need no mask
        float* cp_stage = reinterpret_cast<float*>(lines + (size_t)cc * L::LINE_BYTES) + (cc % L::SKEW_MOD) * L::SKEW_DW;
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        float uf[16], vf[16], af[16], bf[16], s1[16];
        BWDX_STAMP(0)
        // round 1: u, v, a
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = cr + ROWS_PER_IT * i;
            const size_t q = g + (size_t)r * ny + col;
            cp_stage[0 * SF + r] = u[q]; cp_stage[1 * SF + r] = v[q]; cp_stage[2 * SF + r] = ga[q];
        }
        BWDX_STAMP(1)
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            uf[m] = my_stage[0 * SF + tidv + TPF * m]; vf[m] = my_stage[1 * SF + tidv + TPF * m]; af[m] = my_stage[2 * SF + tidv + TPF * m];
        }
        __syncthreads();
        // round 2: b, d
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = cr + ROWS_PER_IT * i;
            const size_t q = g + (size_t)r * ny + col;
            cp_stage[0 * SF + r] = gb[q]; cp_stage[1 * SF + r] = gd[q];
        }
        BWDX_STAMP(2)
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; ++m) { bf[m] = my_stage[0 * SF + tidv + TPF * m]; s1[m] = my_stage[1 * SF + tidv + TPF * m]; }
        __syncthreads();                                                               // the stage aliases the exchange image
        BWDX_STAMP(3)
        float s2[16], w[16], dax[16];
        C2<float> c[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) { s1[m] = af[m] * uf[m] + s1[m]; s2[m] = bf[m] * uf[m]; }           // a u + d,  b u
        adj_core<N, TF, true>(uf, vf, af, bf, s1, s2, w, dax, c, tabF, tabI, xb, tidv, k);
        BWDX_STAMP(4)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            my_stage[0 * SF + tidv + TPF * m] = w[m] - c[m].x;                        // GU
            my_stage[1 * SF + tidv + TPF * m] = -c[m].y;                              // GV
            my_stage[2 * SF + tidv + TPF * m] = -dax[m] * k.inv_rho;                  // GP
        }
        __syncthreads();
        if (j0 + cc < ny) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = cr + ROWS_PER_IT * i;
                const size_t q = g + (size_t)r * ny + j0 + cc;
                gu[q] = cp_stage[0 * SF + r]; gv[q] = cp_stage[1 * SF + r]; gp[q] = cp_stage[2 * SF + r];
            }
        }
        BWDX_STAMP(5)
        __syncthreads();
        BWDX_STAMP(6)
    }
#if NNS_BWDX_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("backward column pass, one tile of wave 0 (cycles): round-1 loads + stage %ld, barrier + read + round-2 loads + stage %ld, barrier + read + barrier %ld, six transforms %ld, "
               "stage results + stores drained %ld, last barrier %ld, whole tile %ld\n", (long)(tk[1] - tk[0]), (long)(tk[2] - tk[1]), (long)(tk[3] - tk[2]), (long)(tk[4] - tk[3]),
               (long)(tk[5] - tk[4]), (long)(tk[6] - tk[5]), (long)(tk[6] - tk[0]));
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// x-pass, ROLE-SPLIT form (round 4; all-float32 mode).  spec_bwd_xpass_kernel above spends two thirds of a tile in memory phases during
// which no wave computes and one third in the six transforms during which nothing moves (profiles/r04_specbwd_xpass_phases.txt:
// 18 + 16 + 7 k cycles of loads and staging, 32 k of transforms, 25 k of stores).  Here, as in the forward column pass
// (spec_xpass_split_kernel), a workgroup has eight TRANSFORM waves with no global memory instruction and four MEMORY waves that own
// the tile traffic and hand the fields over through the LDS staging image.  The backward needs FIVE input fields where the staging image
// holds three, so a tile is fed in TWO steps, and the transform waves' work is cut where the second step's fields are first needed:
//
//     transform waves                                              memory waves (128 registers of tile data per lane)
//     read u, v, a from the image                                  store tile t-1's GU, GV;  load b, d of tile t
//     phase 1: (u_x, v_x) = T1(u, v);  w1 = a u_x
//     ---- barrier A ----                                          b, d -> image (fields 0, 1)
//     ---- barrier B ----
//     read b, d;  w = w1 + b v_x                                   store tile t-1's GP;  load u, v, a of tile t+1
//     phase 2: T2(a, b) -> a_x, viscous part;  T3(a u + d, b u);  GU, GV, GP -> image
//     ---- barrier C ----                                          exchange: GU, GV, GP out of the image, u, v, a of tile t+1 in, slot for slot
//     ---- barrier D ----
//
// Across barrier A / B a transform wave keeps u, a, w1, v_x (64 registers); the pieces of adj_core are re-ordered so that no phase holds
// more than ~100 registers of fields next to a transform's own state (w waits in LDS over phase 2): 3 waves per SIMD (<= 168 VGPRs), which is
// what gives the memory waves a home.  Same filters as adj_core<N, float, true>; c's two contributions are summed in the other order.
// ------------------------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kSplitThreads) void spec_bwd_xsplit_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                                         const float* __restrict__ ga, const float* __restrict__ gb, const float* __restrict__ gd,
                                                                         float* __restrict__ gu, float* __restrict__ gv, float* __restrict__ gp,
                                                                         int ny, int tiles_per_grid, long ntiles, AdjK k) {
    using L = SpecLds<N, float>;
    using SL = SplitLds<N, float>;
    constexpr int TPF = L::TPF, CW = L::LINES, SF = L::STAGE_F;
    constexpr int MROWS = 256 / CW;                       // rows of the tile one memory-wave instruction step covers
    constexpr int NR = N / MROWS;                         // elements per memory lane and field (= 32 for every N)
    static_assert(NR == 32, "tile geometry");
    static_assert(SL::TOTAL <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<float>* tabF; C2<float>* tabI; unsigned char* lines;
    spec_setup<N, float, kSplitThreads>(smem, tabF, tabI, lines);
    auto tile_coords = [&](long t, int& j0, size_t& g) {
        const long lt = ntiles - 1 - (long)xcd_remap((unsigned)t, (unsigned)ntiles);       // last grid first, neighbouring tiles on one XCD (as the forward)
        j0 = (int)(lt % tiles_per_grid) * CW;
        g = (size_t)(lt / tiles_per_grid) * (size_t)N * ny;
    };
    long t = blockIdx.x;
    if (t >= ntiles) return;                               // uniform over the workgroup (the launch never has more workgroups than tiles)
    if (threadIdx.x >= kSpecThreads) {
        // ================= memory waves =================
        int mt = threadIdx.x - kSpecThreads;
        asm volatile("" : "+v"(mt));
        const int cc = mt % CW, cr = mt / CW;
        float* stage = reinterpret_cast<float*>(lines + (size_t)cc * SL::LINE_BYTES) + (cc % 8) * SL::SKEW_DW + 4 * cr;       // [field][row]
        float R[4][NR];                                                        // fields 0..2: a tile's inputs / GU, GV; field 3: GP until its store window
        // a lane owns rows 4 cr .. 4 cr + 3 of every block of 4 MROWS rows (element i <-> row 4 cr + (i & 3) + 4 MROWS (i >> 2)): four consecutive
        // rows move through LDS as ONE 16-byte access; addresses = scalar grid base + 32-bit lane byte offset (spectral_fwd.h, the same scheme)
        auto off32 = [&](int i, unsigned col, unsigned crv) -> unsigned {
            const unsigned r0 = 4u * crv + 4u * MROWS * (unsigned)(i >> 2) + (unsigned)(i & 3);
            return (col + r0 * (unsigned)ny) * 4u;
        };
        auto at = [](const float* base, unsigned byte_off) -> const float& { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
        auto at_w = [](float* base, unsigned byte_off) -> float& { return *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off); };
        auto load3 = [&](long tt, const float* f0, const float* f1, const float* f2) {            // f2 == nullptr: two fields
            int j0; size_t g;
            tile_coords(tt, j0, g);
            unsigned col = (unsigned)(j0 + cc < ny ? j0 + cc : ny - 1);       // clamped column: no mask needed on a load
            unsigned crv = (unsigned)cr;
            asm volatile("" : "+v"(col), "+v"(crv));
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const unsigned c = off32(i, col, crv);
                R[0][i] = at(f0 + g, c); R[1][i] = at(f1 + g, c);
                if (f2) R[2][i] = at(f2 + g, c);
            }
        };
        // The tile's traffic is spread over the two windows the transform phases leave (phase 1 = 2 transforms, phase 2 = 4): GU, GV leave in the
        // first (with b, d coming in: 4 field moves), GP waits in a fourth register field and leaves in the second (with u, v, a of the next
        // tile coming in: 4 field moves)
        auto store_uv = [&](long tt) {
            int j0; size_t g;
            tile_coords(tt, j0, g);
            if (j0 + cc < ny) {
                unsigned col = (unsigned)(j0 + cc);
                unsigned crv = (unsigned)cr;
                asm volatile("" : "+v"(col), "+v"(crv));
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const unsigned c = off32(i, col, crv);
                    at_w(gu + g, c) = R[0][i]; at_w(gv + g, c) = R[1][i];
                }
            }
        };
        auto store_p = [&](long tt) {
            int j0; size_t g;
            tile_coords(tt, j0, g);
            if (j0 + cc < ny) {
                unsigned col = (unsigned)(j0 + cc);
                unsigned crv = (unsigned)cr;
                asm volatile("" : "+v"(col), "+v"(crv));
#pragma unroll
                for (int i = 0; i < NR; ++i) at_w(gp + g, off32(i, col, crv)) = R[3][i];
            }
        };
        load3(t, u, v, ga);
#pragma unroll
        for (int q = 0; q < NR / 4; ++q)
#pragma unroll
            for (int f = 0; f < 3; ++f)
                *reinterpret_cast<float4*>(stage + f * SF + 4 * MROWS * q) = make_float4(R[f][4 * q], R[f][4 * q + 1], R[f][4 * q + 2], R[f][4 * q + 3]);
        __syncthreads();                                                        // u, v, a of the first tile are staged
        long prev = -1;
        for (; t < ntiles; t += gridDim.x) {
            const long tn = t + gridDim.x;
            const bool has_next = tn < ntiles;
            if (prev >= 0) store_uv(prev);                                      // under phase 1: tile t-1's GU, GV out ...
            load3(t, gb, gd, nullptr);                                          // ... and this tile's b, d in (registers of fields 0, 1)
            __syncthreads();                                                    // (A) the transform waves have read u, v, a and are done with the exchange image
#pragma unroll
            for (int q = 0; q < NR / 4; ++q)
#pragma unroll
                for (int f = 0; f < 2; ++f)
                    *reinterpret_cast<float4*>(stage + f * SF + 4 * MROWS * q) = make_float4(R[f][4 * q], R[f][4 * q + 1], R[f][4 * q + 2], R[f][4 * q + 3]);
            __syncthreads();                                                    // (B) b, d are staged
            if (prev >= 0) store_p(prev);                                       // under phase 2: tile t-1's GP out ...
            if (has_next) load3(tn, u, v, ga);                                  // ... and the next tile's u, v, a in
            __syncthreads();                                                    // (C) the transform waves have written the tile's gradients to the image
#pragma unroll
            for (int q = 0; q < NR / 4; ++q) {
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    float4* slot = reinterpret_cast<float4*>(stage + f * SF + 4 * MROWS * q);
                    const float4 out = *slot;
                    if (has_next) *slot = make_float4(R[f][4 * q], R[f][4 * q + 1], R[f][4 * q + 2], R[f][4 * q + 3]);
                    float (&dst)[NR] = R[f == 2 ? 3 : f];                       // GP goes to the fourth field
                    dst[4 * q] = out.x; dst[4 * q + 1] = out.y; dst[4 * q + 2] = out.z; dst[4 * q + 3] = out.w;
                }
                if (q & 1) __builtin_amdgcn_sched_barrier(0);                  // two row groups in flight at a time (register pressure)
            }
            __syncthreads();                                                    // (D) the next tile's u, v, a are staged
            prev = t;
        }
        store_uv(prev);
        store_p(prev);
        return;
    }
    // ================= transform waves =================
    const C2<float>* tabI2 = tabI + N / 2;
    const float* ctab = reinterpret_cast<const float*>(tabI + N / 2 + Pass2<N>::ENTRIES);
    const float c1h = (float)(0.5 * k.c1), c2h = (float)(0.5 * k.c2);
    __syncthreads();                                                            // u, v, a of the first tile are staged
    for (; t < ntiles; t += gridDim.x) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave;
        const int sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * SL::LINE_BYTES;
        C2<float>* xbI = reinterpret_cast<C2<float>*>(xb);
        float* my_stage = reinterpret_cast<float*>(xb) + (line % 8) * SL::SKEW_DW;
        // behind the exchange image the line area has room for ONE lane-private array of 16 floats (LINE_BYTES - XB_BYTES >= 64 TPF bytes for every N): w waits
        // there over phase 2 -- 16 registers fewer at the kernel's register peak (52 bytes of scratch per lane at N = 1024 without it)
        static_assert(SL::LINE_BYTES - (L::XB_BYTES + 15) / 16 * 16 >= 64 * TPF, "park area");
        float4* parkw = reinterpret_cast<float4*>(xb + (L::XB_BYTES + 15) / 16 * 16) + tid;
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        // forward transform of the forward-DIFFERENCED packed pair (p, q): FFT(d) = (e^{i theta} - 1) FFT(p + i q) (adj_core, all-float32 mode)
        auto diff_fft = [&](const float (&p)[16], const float (&q)[16], C2<float> (&zv)[16]) {
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                zv[m].x = right_of<m, N / 16>(p, tidv) - p[m];
                zv[m].y = right_of<m, N / 16>(q, tidv) - q[m];
            });
            fft_line<float, N, false>(zv, tabI, tabI2, xbI, tidv);
        };
        float uf[16], af[16], w[16], vx[16];
        {   // ---- phase 1: (u_x, v_x), w1 = a u_x
            float vf[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                uf[m] = my_stage[0 * SF + tidv + TPF * m]; vf[m] = my_stage[1 * SF + tidv + TPF * m]; af[m] = my_stage[2 * SF + tidv + TPF * m];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            C2<float> zv[16], e[16];
            diff_fft(uf, vf, zv);
            int te = tidv;
            asm volatile("" : "+v"(te), "+v"(zv[0].x));
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                int ko, ke;
                wavenumber<N, m>(te, ko, ke);
                const float ar = ctab[ke < 0 ? -ke : ke] * c1h, ai = (float)ko * c1h;
                e[m].x = ar * zv[m].x + ai * zv[m].y; e[m].y = ar * zv[m].y - ai * zv[m].x;
            });
            __builtin_amdgcn_sched_barrier(0);
            fft_line<float, N, true>(e, tabI, tabI2, xbI, tidv);
#pragma unroll
            for (int m = 0; m < 16; ++m) { w[m] = af[m] * e[m].x; vx[m] = e[m].y; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __syncthreads();                                                        // (A)
        __syncthreads();                                                        // (B) b, d are staged
        float bf[16], s1[16], s2[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) { bf[m] = my_stage[0 * SF + tidv + TPF * m]; s1[m] = my_stage[1 * SF + tidv + TPF * m]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            w[m] = __builtin_fmaf(bf[m], vx[m], w[m]);                         // w = a u_x + b v_x
            s1[m] = af[m] * uf[m] + s1[m]; s2[m] = bf[m] * uf[m];              // a u + d,  b u
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) parkw[q * TPF] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
        C2<float> c[16];
        float gpo[16];
        {   // ---- phase 2a: c = D s1 + i D s2 in spectral space (the transform that needs u first: u, d are dead after it)
            C2<float> zv[16];
            diff_fft(s1, s2, zv);
            int te = tidv;
            asm volatile("" : "+v"(te), "+v"(zv[0].x));
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                int ko, ke;
                wavenumber<N, m>(te, ko, ke);
                const float ar = ctab[ke < 0 ? -ke : ke] * c1h, ai = (float)ko * c1h;
                c[m].x = ar * zv[m].x + ai * zv[m].y; c[m].y = ar * zv[m].y - ai * zv[m].x;
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // ---- phase 2b: a_x (-> GP); the viscous term -nu k^2 FFT(a + i b) joins c; then the one inverse transform of c
            // (adj_core adds the two contributions of c in the other order: the results differ from spec_bwd_xpass_kernel's in the last bit)
            C2<float> zv[16], e[16];
            diff_fft(af, bf, zv);
            int te = tidv;
            asm volatile("" : "+v"(te), "+v"(zv[0].x));
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                int ko, ke;
                wavenumber<N, m>(te, ko, ke);
                const float ct = ctab[ke < 0 ? -ke : ke];
                const float ar = ct * c1h, ai = (float)ko * c1h;
                e[m].x = ar * zv[m].x + ai * zv[m].y; e[m].y = ar * zv[m].y - ai * zv[m].x;
                const float kf = (float)ke * c2h;
                const float br = kf * (float)ke, bi = kf * ct;                   // -nu k^2 Z2 = (br + i bi) D
                c[m].x += br * zv[m].x - bi * zv[m].y; c[m].y += br * zv[m].y + bi * zv[m].x;
            });
            __builtin_amdgcn_sched_barrier(0);
            fft_line<float, N, true>(e, tabI, tabI2, xbI, tidv);
#pragma unroll
            for (int m = 0; m < 16; ++m) gpo[m] = -e[m].x * k.inv_rho;           // GP = -(D_x a) / rho
            __builtin_amdgcn_sched_barrier(0);
            fft_line<float, N, true>(c, tabI, tabI2, xbI, tidv);
        }
        {
            int tq = tid;
            asm volatile("" : "+v"(tq));                                        // a different address to the compiler: no forwarding of w across phase 2
            const float4* pr = reinterpret_cast<const float4*>(xb + (L::XB_BYTES + 15) / 16 * 16) + tq;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 x = pr[q * TPF]; w[4 * q] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            my_stage[0 * SF + tidv + TPF * m] = w[m] - c[m].x;                    // GU
            my_stage[1 * SF + tidv + TPF * m] = -c[m].y;                          // GV
            my_stage[2 * SF + tidv + TPF * m] = gpo[m];                           // GP
        }
        __syncthreads();                                                        // (C)
        __syncthreads();                                                        // (D)
    }
}

template <int N, typename TF>
int launch_bwd(const float* u, const float* v, const float* ga, const float* gb, const float* gd, float* gu, float* gv, float* gp,
               float* gup, float* gvp, int batch, int nx_or_ny_other, bool xpass, const AdjK& k, hipStream_t s) {
    using L = SpecLds<N, TF>;
    const long gmax = spec_grid_cap();
    if (xpass) {
#ifndef NNS_BWD_XSPLIT
#define NNS_BWD_XSPLIT 1           // 1: all-float32 mode: spec_bwd_xsplit_kernel (8 transform + 4 memory waves); 0: spec_bwd_xpass_kernel
#endif
        if constexpr (sizeof(TF) == 4) {
            static const bool use_split = [] { const char* e = getenv("NNS_BWD_XSPLIT"); return e ? atoi(e) != 0 : NNS_BWD_XSPLIT != 0; }();
            const int ny = nx_or_ny_other;
            // the memory waves address with a scalar grid base + a 32-bit byte offset per lane
            if (use_split && (unsigned long long)N * (unsigned long long)ny < (1ull << 30)) {
                using SL = SplitLds<N, float>;
                auto kern = spec_bwd_xsplit_kernel<N>;
                static bool attr = false;
                if (!attr) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SL::TOTAL);
                    if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spec bwd xpass: hipFuncSetAttribute(%d B): %s", SL::TOTAL, hipGetErrorString(e));
                    attr = true;
                }
                const int tiles_per_grid = (ny + L::LINES - 1) / L::LINES;
                const long ntiles = (long)batch * tiles_per_grid;
                hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < gmax ? ntiles : gmax)), dim3(kSplitThreads), SL::TOTAL, s, u, v, ga, gb, gd, gu, gv, gp, ny, tiles_per_grid, ntiles, k);
                return check_launch("spec_residual_bwd_xpass");
            }
        }
#ifndef NNS_BWD_XWAVES
#define NNS_BWD_XWAVES 8           // waves (= lines at N = 1024) per workgroup of the backward column pass: 8 = one workgroup per CU, 4 = two out of step (measured slower)
#endif
        static const int xw = [] { const char* e = getenv("NNS_BWD_XWAVES"); const int v = e ? atoi(e) : NNS_BWD_XWAVES; return v == 4 ? 4 : 8; }();
        auto go = [&](auto wc) -> int {
            constexpr int W = decltype(wc)::value;
            using LW = SpecLds<N, TF, W>;
            auto kern = spec_bwd_xpass_kernel<N, TF, W>;
            static bool attr = false;
            if (!attr) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LW::TOTAL);
                if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spec bwd xpass: hipFuncSetAttribute(%d B): %s", LW::TOTAL, hipGetErrorString(e));
                attr = true;
            }
            const int ny = nx_or_ny_other;
            const int tiles_per_grid = (ny + LW::LINES - 1) / LW::LINES;
            const long ntiles = (long)batch * tiles_per_grid;
            const long cap = gmax * (8 / W);
            hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(W * kWave), LW::TOTAL, s, u, v, ga, gb, gd, gu, gv, gp, ny, tiles_per_grid, ntiles, k);
            return check_launch("spec_residual_bwd_xpass");
        };
        return xw == 8 ? go(std::integral_constant<int, 8>{}) : go(std::integral_constant<int, 4>{});
    }
    auto kern = spec_bwd_ypass_kernel<N, TF>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spec bwd ypass: hipFuncSetAttribute(%d B): %s", L::TOTAL, hipGetErrorString(e));
        attr = true;
    }
    const long nrows = (long)batch * nx_or_ny_other;
    const long niter = (nrows + L::LINES - 1) / L::LINES;
    hipLaunchKernelGGL(kern, dim3((unsigned)(niter < gmax ? niter : gmax)), dim3(kSpecThreads), L::TOTAL, s, u, v, ga, gb, gd, gu, gv, gp, gup, gvp, nrows, k);
    return check_launch("spec_residual_bwd_ypass");
}

}  // namespace

NNS_API int nns_spec_residual_bwd_f32(const float* u, const float* v, const float* g_u, const float* g_v, const float* g_div,
                                      float* grad_u, float* grad_v, float* grad_p, float* grad_u_prev, float* grad_v_prev,
                                      int batch, int nx, int ny, double dt, double Lx, double Ly, double rho, double nu, int precise, void* stream) {
    if (!u || !v || !g_u || !g_v || !g_div || !grad_u || !grad_v || !grad_p || batch < 1)
        return fail(NNS_ERR_INVALID_ARG, "spec_residual_bwd: bad args");
    if (Lx == 0 || Ly == 0 || rho == 0 || dt == 0) return fail(NNS_ERR_INVALID_ARG, "spec_residual_bwd: Lx, Ly, rho, dt must be non-zero");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // per axis: the FFT engine for powers of two in [64, 1024], circulant matrices in float64 for any other length (spectral_dense.hip)
    if (!spec_len_ok(nx) || !spec_len_ok(ny))
        return fail(NNS_ERR_UNSUPPORTED, "spec_residual_bwd: nx=%d, ny=%d: powers of two in [64, 1024] (FFT engine) or any length 3 .. %d (dense fallback)", nx, ny, kDenseMaxLen);
    precise = spec_resolve_precise(precise, nu, nx, Lx, ny, Ly);                    // one arithmetic for both passes
    if (!pow2_in_range(nx)) {
        if (int rc0 = dense_bwd_xpass(u, v, g_u, g_v, g_div, grad_u, grad_v, grad_p, batch, nx, ny, Lx, rho, nu, s)) return rc0;
    }
    const bool x_done = !pow2_in_range(nx);
    const double kx = 2.0 * M_PI / Lx, ky = 2.0 * M_PI / Ly;
    const AdjK kxp{kx / nx, nu * kx * kx / nx, (float)(1.0 / rho), (float)(1.0 / dt)};
    const AdjK kyp{ky / ny, nu * ky * ky / ny, (float)(1.0 / rho), (float)(1.0 / dt)};
    int rc = x_done ? 0 : dispatch_n(nx, [&](auto n) {
        constexpr int N = decltype(n)::value;
        return !spec_f32_mode(precise, nu, nx, Lx) ? launch_bwd<N, double>(u, v, g_u, g_v, g_div, grad_u, grad_v, grad_p, nullptr, nullptr, batch, ny, true, kxp, s)
                                                  : launch_bwd<N, float>(u, v, g_u, g_v, g_div, grad_u, grad_v, grad_p, nullptr, nullptr, batch, ny, true, kxp, s);
    });
    if (rc) return rc;
    if (!pow2_in_range(ny)) return dense_bwd_ypass(u, v, g_u, g_v, g_div, grad_u, grad_v, grad_p, grad_u_prev, grad_v_prev, batch, nx, ny, dt, Ly, rho, nu, s);
    return dispatch_n(ny, [&](auto n) {
        constexpr int N = decltype(n)::value;
        return !spec_f32_mode(precise, nu, ny, Ly) ? launch_bwd<N, double>(u, v, g_u, g_v, g_div, grad_u, grad_v, grad_p, grad_u_prev, grad_v_prev, batch, nx, false, kyp, s)
                                                  : launch_bwd<N, float>(u, v, g_u, g_v, g_div, grad_u, grad_v, grad_p, grad_u_prev, grad_v_prev, batch, nx, false, kyp, s);
    });
}
