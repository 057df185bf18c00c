// LDS-resident power-of-two complex FFT for gfx950, 16 elements per lane.
//
// One FFT line of length N (64 <= N <= 1024) is owned by TPF = N/16 consecutive lanes of ONE
// wave (N = 1024: a whole wave64; N = 256: four lines per wave), so every exchange between
// passes is wave-synchronous (LDS queue order, no workgroup barrier).  Lane t of a line holds
// elements  t + TPF*m, m = 0..15  -- the same ownership before and after the transform, in
// natural order, so that
//   * global loads/stores of a contiguous line are coalesced (a wave instruction moves 256 B),
//   * the spectral multiply knows every element's wavenumber without any data movement, and
//   * an inverse transform can start directly from the registers the forward one left.
//
// Algorithm: Stockham autosort, decimation in frequency, radix plan (16, 16, N/256) or
// (16, N/16): a pass of radix R with sub-transform stride NS does, for "virtual thread" j,
//     y[s] = DFT_R( x[j + t*N/R] * W_N^{t * (j mod NS) * N/(NS*R)} ),  s,t = 0..R-1
// and scatters y[s] to  (j / NS)*NS*R + (j mod NS) + s*NS.  With 16 elements per lane a lane plays
// 16/R virtual threads j = tid + TPF*q, and the elements it needs are exactly the 16 it owns.
// The last pass (NS*R == N) scatters back onto the same lane: no exchange.
// LDS exchange image: element e at slot e + e/16 (see fft_slot).
//
// Twiddles W_N^k = exp(-2 pi i k / N) come from a half table (k < N/2, sign flip above) kept in
// LDS, one table per arithmetic type in use.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "nns_common.h"

#ifndef NNS_DFT_SB
#define NNS_DFT_SB 1      // scheduling barriers between the dft4 groups of a float64 dft16 (bounds register pressure)
#endif

namespace nns {

template <typename T> struct C2 { T x, y; };
// (Round 3, measured and rejected: complex arithmetic on the packed float32 instructions -- a complex multiply is two v_pk_mul/fma_f32 with the
//  swap and sign folded into op_sel / neg_lo, no moves (checked in the ISA), 22 % fewer vector instructions per line.  But on gfx950 every
//  v_pk_*_f32 costs a SIMD 4.3-4.5 cycles per wave64 instruction against 2.4 for v_add/v_mul_f32 and 2.7-3.0 for v_fma/v_fmac_f32 with distinct
//  sources (tools/valu_rate_bench.hip, profiles/r03_valu_rate.txt): two lanes' worth per packed instruction is no faster than two scalar
//  ones, and the column pass went 0.520 -> 0.555 ms, the fused row pass 0.752 -> 0.768 (profiles/r03_ab_packed_stagger_prio.log).)

template <typename T> __device__ __forceinline__ C2<T> operator+(C2<T> a, C2<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ C2<T> operator-(C2<T> a, C2<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> __device__ __forceinline__ C2<T> cmul(C2<T> a, C2<T> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// multiply by -i (forward) or +i (inverse)
template <typename T, bool INV> __device__ __forceinline__ C2<T> rot90(C2<T> z) {
    if constexpr (INV) return {-z.y, z.x}; else return {z.y, -z.x};
}
// multiply by the constant (cr, -ci) forward / (cr, +ci) inverse  [= W^m with cos = cr, sin = ci]
template <typename T, bool INV> __device__ __forceinline__ C2<T> mulw(C2<T> z, T cr, T ci) {
    if constexpr (INV) return {z.x * cr - z.y * ci, z.y * cr + z.x * ci};
    else return {z.x * cr + z.y * ci, z.y * cr - z.x * ci};
}

template <typename T, bool INV>
__device__ __forceinline__ void dft2(C2<T>& a, C2<T>& b) { const C2<T> t = a; a = t + b; b = t - b; }

template <typename T, bool INV>
__device__ __forceinline__ void dft4(C2<T>& a0, C2<T>& a1, C2<T>& a2, C2<T>& a3) {
    const C2<T> s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = rot90<T, INV>(a1 - a3);
    a0 = s02 + s13; a2 = s02 - s13; a1 = d02 + d13; a3 = d02 - d13;
}

template <typename T, bool INV>
__device__ __forceinline__ void dft8(C2<T> (&x)[8]) {
    constexpr T r = (T)0.70710678118654752440;
    dft4<T, INV>(x[0], x[2], x[4], x[6]);           // n2 = 0 : results k1 at x[2*k1]
    dft4<T, INV>(x[1], x[3], x[5], x[7]);           // n2 = 1 : results k1 at x[2*k1+1]
    x[3] = mulw<T, INV>(x[3], r, r);                // W8^1
    x[5] = rot90<T, INV>(x[5]);                     // W8^2
    x[7] = mulw<T, INV>(x[7], -r, r);               // W8^3
    dft2<T, INV>(x[0], x[1]); dft2<T, INV>(x[2], x[3]); dft2<T, INV>(x[4], x[5]); dft2<T, INV>(x[6], x[7]);
    // y[k1 + 4*k2] = x[2*k1 + k2]
    const C2<T> y1 = x[2], y2 = x[4], y3 = x[6], y4 = x[1], y5 = x[3], y6 = x[5];
    x[1] = y1; x[2] = y2; x[3] = y3; x[4] = y4; x[5] = y5; x[6] = y6;
}

template <typename T, bool INV>
__device__ __forceinline__ void dft16(C2<T> (&x)[16]) {
    constexpr T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173, r = (T)0.70710678118654752440;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        dft4<T, INV>(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);   // k1 at x[4*k1+n2]
        if constexpr (sizeof(T) == 8 && NNS_DFT_SB) __builtin_amdgcn_sched_barrier(0);
    }
    // x[4*k1 + n2] *= W16^{n2*k1}
    x[5] = mulw<T, INV>(x[5], c1, s1);      // 1
    x[6] = mulw<T, INV>(x[6], r, r);        // 2
    x[7] = mulw<T, INV>(x[7], s1, c1);      // 3
    x[9] = mulw<T, INV>(x[9], r, r);        // 2
    x[10] = rot90<T, INV>(x[10]);           // 4
    x[11] = mulw<T, INV>(x[11], -r, r);     // 6
    x[13] = mulw<T, INV>(x[13], s1, c1);    // 3
    x[14] = mulw<T, INV>(x[14], -r, r);     // 6
    x[15] = mulw<T, INV>(x[15], -c1, -s1);  // 9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        dft4<T, INV>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);  // k2 at x[4*k1+k2]
        if constexpr (sizeof(T) == 8 && NNS_DFT_SB) __builtin_amdgcn_sched_barrier(0);
    }
    // y[k1 + 4*k2] = x[4*k1 + k2]: 4x4 transpose
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) { const C2<T> t = x[4 * a + b]; x[4 * a + b] = x[4 * b + a]; x[4 * b + a] = t; }
}

template <typename T, int R, bool INV>
__device__ __forceinline__ void dftR(C2<T> (&y)[R]) {
    if constexpr (R == 2) dft2<T, INV>(y[0], y[1]);
    else if constexpr (R == 4) dft4<T, INV>(y[0], y[1], y[2], y[3]);
    else if constexpr (R == 8) dft8<T, INV>(y);
    else dft16<T, INV>(y);
}

// LDS slot of element e in the exchange image: one pad slot per 16 elements.  The first pass scatters with
// stride 16 (lane j writes 16j + t -> slot 17j + t: distinct banks for b64 and b128 stores), the read-back is
// contiguous across lanes; every address is a per-lane base plus an immediate.  (An XOR swizzle
// e ^ ((e>>4)&15) removes the residual 2-way conflict of the b128 read-back but costs ~250 extra integer
// instructions per line; measured equal on MI355X -- the kernel is issue-bound, not LDS-bound.)
__device__ __forceinline__ int fft_slot(int e) { return e + (e >> 4); }
template <int N> struct FftGeom {
    static constexpr int TPF = N / 16;             // lanes per line
    static constexpr int SLOTS = N + N / 16;       // padded line length in the exchange image
};

// Half twiddle table lookup: tab[k] = exp(-2 pi i k/N), k < N/2.
template <typename T, int N, bool INV>
__device__ __forceinline__ C2<T> twiddle(const C2<T>* __restrict__ tab, int k) {
    const bool neg = k >= N / 2;
    C2<T> w = tab[neg ? k - N / 2 : k];
    if (neg) { w.x = -w.x; w.y = -w.y; }
    if constexpr (INV) w.y = -w.y;
    return w;
}

// One pass.  x: the lane's 16 elements (element tid + TPF*m in x[m]); xb: this line's exchange
// image in LDS.  After a non-final pass x again holds elements tid + TPF*m of the partially
// transformed sequence.
template <typename T, int N, int R, int NS, bool INV>
__device__ __forceinline__ void fft_pass(C2<T> (&x)[16], const C2<T>* __restrict__ tab, const C2<T>* __restrict__ tab2,
                                         C2<T>* __restrict__ xb, int tid_in) {
    constexpr int TPF = N / 16, NB = 16 / R;
    constexpr bool LAST = (NS * R == N);
    // Everything below that depends only on the lane id (twiddle indices, table and exchange addresses) is
    // "free-floating" for instruction selection, which computes it thousands of instructions early and then
    // spills it.  Tie the lane id to this pass's input data so the index math is emitted HERE.
    int tid = tid_in;
    asm volatile("" : "+v"(tid), "+v"(x[0].x));
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const unsigned j = (unsigned)tid + TPF * q;
        C2<T> y[R];
#pragma unroll
        for (int t = 0; t < R; ++t) y[t] = x[q + NB * t];
#ifndef NNS_F32_PIN
#define NNS_F32_PIN 1      // pin float32 pass-2 table reads in groups of 4 (limits hoisting)
#endif
#ifndef NNS_TW_LOOKUP
#define NNS_TW_LOOKUP 0      // 0: float64 twiddles by running product from one table read; 1: grouped table reads
#endif
        if constexpr (NS == 16) {
            // second pass: twiddle W_N^{t * c * N/(16 R)}, c = j mod 16, from the pass-2 table laid out [t][c]:
            // for a given t the lanes of a wave read 16 CONSECUTIVE entries (conflict-free, broadcast across
            // the lane groups that share c); striding the main table by t*c*N/(16R) was up to 16-way conflicted.
            int c = (int)(j & 15u);
            if constexpr (sizeof(T) == 8 && !NNS_TW_LOOKUP) {
                // float64: ONE table read, the other R-2 twiddles by running product (14 roundings of 1e-16 are
                // irrelevant, and 15 hoisted double2 reads would cost 60 VGPRs).  Grouped table reads (pinned
                // three at a time) save 56 fp64 ops per pass but measured slower on MI355X (A/B on one box).
                C2<T> w = tab2[16 + c];
                if constexpr (INV) w.y = -w.y;
                C2<T> wt = w;
                y[1] = cmul<T>(y[1], wt);
#pragma unroll
                for (int t = 2; t < R; ++t) { wt = cmul<T>(wt, w); y[t] = cmul<T>(y[t], wt); }
            } else {
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    // table reads pinned in groups (3 for float64, 4 for float32) so they are not all hoisted
                    if constexpr (sizeof(T) == 8) { if (t % 3 == 1) asm volatile("" : "+v"(c), "+v"(y[t].x)); }
                    else if (NNS_F32_PIN) { if (t % 4 == 1) asm volatile("" : "+v"(c), "+v"(y[t].x)); }
                    C2<T> w = tab2[16 * t + c];
                    if constexpr (INV) w.y = -w.y;
                    y[t] = cmul<T>(y[t], w);
                }
            }
        } else if constexpr (NS > 1) {
            int jm = (int)((j & (unsigned)(NS - 1)) * (unsigned)(N / (NS * R)));
            if constexpr (sizeof(T) == 8 && !NNS_TW_LOOKUP) {
                const C2<T> w = twiddle<T, N, INV>(tab, jm);
                C2<T> wt = w;
                y[1] = cmul<T>(y[1], wt);
#pragma unroll
                for (int t = 2; t < R; ++t) { wt = cmul<T>(wt, w); y[t] = cmul<T>(y[t], wt); }
            } else {
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    if constexpr (sizeof(T) == 8) { if (t % 3 == 1) asm volatile("" : "+v"(jm), "+v"(y[t].x)); }
                    y[t] = cmul<T>(y[t], twiddle<T, N, INV>(tab, t * jm));
                }
            }
        }
        dftR<T, R, INV>(y);
        if constexpr (LAST) {
#pragma unroll
            for (int s = 0; s < R; ++s) x[q + NB * s] = y[s];
        } else {
            // slot(e0 + s NS) = slot(e0) + slot(s NS): s NS is a multiple of 16 or (NS = 1) e0 is: one base + constants
            const unsigned e0 = (j / NS) * (NS * R) + (j & (unsigned)(NS - 1));
            const unsigned wbase = e0 + (e0 >> 4);
#pragma unroll
            for (int s = 0; s < R; ++s) xb[wbase + (unsigned)fft_slot(s * NS)] = y[s];
        }
    }
    if constexpr (!LAST) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // slot(tid + TPF m) = slot(tid) + slot(TPF m): TPF m is a multiple of 16, or tid + (TPF m mod 16) < 16
        const unsigned rbase = (unsigned)tid + ((unsigned)tid >> 4);
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = xb[rbase + (unsigned)fft_slot(TPF * m)];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Full transform (unnormalised).  In and out: x[m] = element tid + TPF*m, natural order.
// `hook(std::integral_constant<int, SLOT0 + i>{})` is called after pass i: callers use these points to trickle
// global-memory instructions into the arithmetic (see spec_xpass_kernel).  FftPasses<N>::value passes per line.
template <int N> struct FftPasses { static constexpr int value = (N / 16 <= 16) ? 2 : 3; };
struct NoHook { template <typename S> __device__ __forceinline__ void operator()(S) const {} };

template <typename T, int N, bool INV, int SLOT0 = 0, typename Hook = NoHook>
__device__ __forceinline__ void fft_line(C2<T> (&x)[16], const C2<T>* __restrict__ tab, const C2<T>* __restrict__ tab2,
                                         C2<T>* __restrict__ xb, int tid, Hook&& hook = Hook{}) {
    static_assert(N == 64 || N == 128 || N == 256 || N == 512 || N == 1024, "supported line lengths");
    fft_pass<T, N, 16, 1, INV>(x, tab, tab2, xb, tid);
    hook(std::integral_constant<int, SLOT0>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (N / 16 < 16) {
        fft_pass<T, N, N / 16, 16, INV>(x, tab, tab2, xb, tid);
        hook(std::integral_constant<int, SLOT0 + 1>{});
    } else {
        fft_pass<T, N, 16, 16, INV>(x, tab, tab2, xb, tid);
        hook(std::integral_constant<int, SLOT0 + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (N / 256 > 1) {
            fft_pass<T, N, N / 256, 256, INV>(x, tab, tab2, xb, tid);
            hook(std::integral_constant<int, SLOT0 + 2>{});
        }
    }
}

// Fill a half twiddle table (N/2 entries) cooperatively; exact-ish sincospi in double.
template <typename T, int N>
__device__ __forceinline__ void fill_twiddles(C2<T>* tab, int tid, int nthreads) {
    for (int k = tid; k < N / 2; k += nthreads) {
        double s, c;
        sincospi(2.0 * (double)k / (double)N, &s, &c);
        tab[k].x = (T)c;
        tab[k].y = (T)(-s);
    }
}

// Pass-2 table: tab2[16*t + c] = W_N^{t * c * N/(16*R2)}, t < R2 = min(16, N/16), c < 16.
template <int N> struct Pass2 { static constexpr int R2 = (N / 16 < 16) ? N / 16 : 16; static constexpr int ENTRIES = 16 * R2; };
template <typename T, int N>
__device__ __forceinline__ void fill_twiddles2(C2<T>* tab2, int tid, int nthreads) {
    constexpr int R2 = Pass2<N>::R2;
    for (int e = tid; e < 16 * R2; e += nthreads) {
        const int t = e >> 4, c = e & 15;
        double s, co;
        sincospi(2.0 * (double)(t * c * (N / (16 * R2))) / (double)N, &s, &co);
        tab2[e].x = (T)co;
        tab2[e].y = (T)(-s);
    }
}

}  // namespace nns
