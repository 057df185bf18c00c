// Scaffolding shared by the spectral residual kernels (forward: spectral_kernels.hip, backward:
// spectral_bwd_kernels.hip): launch geometry, LDS layout, wavenumber helper, table setup, size dispatch.
#pragma once
#include "nns_common.h"
#include "fft_lds.h"
#include <type_traits>
#include <cstdlib>
#include <cmath>

namespace nns {
namespace spec {

#ifndef NNS_SPEC_THREADS
#define NNS_SPEC_THREADS 512
#endif
constexpr int kSpecThreads = NNS_SPEC_THREADS;      // 8 waves: 2 per SIMD, <= 256 VGPRs each
constexpr int kSpecWaves = kSpecThreads / kWave;

struct SpecK {
    // folded on the host (kscale = 2 pi / L of the transformed axis, 1/N = the inverse transform's normalisation):
    double c1;            // kscale / N            : first-derivative factor per unit wavenumber index
    double cs;            // kscale / (rho N)      : pressure-gradient factor per unit wavenumber index
    double c2;            // nu kscale^2 / N       : viscous factor per unit SQUARED wavenumber index
    float inv_dt;
};

template <int N, typename TF, int WAVES = kSpecWaves>
struct SpecLds {
    static constexpr int TPF = N / 16;
    static constexpr int FPW = kWave / TPF;                       // lines per wave
    static constexpr int LINES = WAVES * FPW;                     // lines per workgroup
    static constexpr int SLOTS = N + N / 16;
    static constexpr int STAGE_F = N + 16;                        // floats per staged field (padded)
    static constexpr int XB_BYTES = SLOTS * (int)sizeof(C2<TF>);
    static constexpr int SKEW_MOD = LINES < 32 ? LINES : 32;           // staging skew: line%SKEW_MOD * SKEW_DW dwords,
    static constexpr int SKEW_DW = 32 / SKEW_MOD;                       // so a 32-lane store group hits 32 banks
    static constexpr int STAGE_BYTES = 3 * STAGE_F * 4 + 128;
    static constexpr int LINE_BYTES = ((XB_BYTES > STAGE_BYTES ? XB_BYTES : STAGE_BYTES) + 127) / 128 * 128;
    static constexpr int TABF_BYTES = (N / 2 + Pass2<N>::ENTRIES) * (int)sizeof(C2<TF>);     // main half table + pass-2 table
    // float32 tables of the precise mode + the pressure filter table ctab[k] = k_odd cot(pi k / N), k = 0 .. N/2 (deriv_core)
    // (all-float32 mode: the float32 tables ARE tabF, only ctab follows them)
    static constexpr int CTAB_BYTES = ((N / 2 + 1) * 4 + 15) / 16 * 16;
    static constexpr int TABI_BYTES = (sizeof(TF) == 4 ? 0 : (N / 2 + Pass2<N>::ENTRIES) * (int)sizeof(C2<float>)) + CTAB_BYTES;
    static constexpr int TOTAL = TABF_BYTES + TABI_BYTES + LINES * LINE_BYTES;
};

// Role-split column passes (forward: spectral_fwd.h, backward: spectral_bwd_kernels.hip): eight transform waves + four memory waves per workgroup
constexpr int kSplitThreads = kSpecThreads + 256;
template <int N, typename TF>
struct SplitLds {
    using L = SpecLds<N, TF>;
    static constexpr int SKEW_DW = N == 64 ? 4 : 8;      // dwords of skew per line (mod 8 lines): the 16-lane groups of the memory waves' b128 exchange hit 64 distinct banks
                                                         // (N = 64: 128 lines per workgroup, half the skew keeps the image inside 160 KB)
    static constexpr int STAGE_BYTES = 3 * L::STAGE_F * 4 + 8 * SKEW_DW * 4;
    static constexpr int LINE_BYTES = ((L::XB_BYTES > STAGE_BYTES ? L::XB_BYTES : STAGE_BYTES) + 127) / 128 * 128;
    static constexpr int TOTAL = L::TABF_BYTES + L::TABI_BYTES + L::LINES * LINE_BYTES;
};

// Signed wavenumber index of the element in register slot m of lane `te` (element te + TPF m).  N/2 = 8 TPF, so
// slots 0..7 hold the non-negative wavenumbers and 8..15 the negative ones: no compare except for the Nyquist mode
// (slot 8 of lane 0), which odd derivatives drop, as in the oracle.
template <int N, int M>
__device__ __forceinline__ void wavenumber(int te, int& k_odd, int& k_even) {
    constexpr int TPF = N / 16;
    k_even = te + TPF * M - (M >= 8 ? N : 0);
    if constexpr (M == 8) k_odd = te == 0 ? 0 : k_even; else k_odd = k_even;
}

template <int N, typename TF, int NTHREADS = kSpecThreads>
__device__ __forceinline__ void spec_setup(unsigned char* smem, C2<TF>*& tabF, C2<float>*& tabI, unsigned char*& lines) {
    using L = SpecLds<N, TF>;
    tabF = reinterpret_cast<C2<TF>*>(smem);
    fill_twiddles<TF, N>(tabF, threadIdx.x, NTHREADS);
    fill_twiddles2<TF, N>(tabF + N / 2, threadIdx.x, NTHREADS);
    if constexpr (sizeof(TF) == 4) {
        tabI = reinterpret_cast<C2<float>*>(smem);
    } else {
        tabI = reinterpret_cast<C2<float>*>(smem + L::TABF_BYTES);
        fill_twiddles<float, N>(tabI, threadIdx.x, NTHREADS);
        fill_twiddles2<float, N>(tabI + N / 2, threadIdx.x, NTHREADS);
    }
    {
        float* ctab = reinterpret_cast<float*>(tabI + N / 2 + Pass2<N>::ENTRIES);
        for (int kk = threadIdx.x; kk <= N / 2; kk += NTHREADS) {
            double sn, cs;
            sincospi((double)kk / (double)N, &sn, &cs);
            ctab[kk] = (kk == 0 || kk == N / 2) ? 0.f : (float)((double)kk * cs / sn);
        }
    }
    lines = smem + L::TABF_BYTES + L::TABI_BYTES;
    __syncthreads();
}

__device__ __forceinline__ float wave_ror1(float x) {          // lane i <- lane i-1, lane 0 <- lane 63
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x13C, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_rol1(float x) {          // lane i <- lane i+1, lane 63 <- lane 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x134, 0xF, 0xF, false));
}
// left / right neighbour (column - 1 / + 1, periodic) of slot M of a row held as element tid + TPF m in slot m by the TPF lanes
// of one line -- DPP only, no LDS crossbar for any row length: the in-line neighbour is a whole-wave rotate by one lane (the lane
// it is wrong for -- the line's first / last -- takes the wrap value instead), the wrap value is the adjacent slot of the line's
// other end: a readlane at TPF = 64 (one line per wave), one ds_bpermute at TPF = 32, a rotate within the 16-lane DPP row by TPF - 1 below
// (TPF = 16, 8, 4 divide the row).  Round 2: the TPF < 64 forms used two ds_bpermute per neighbour (12 per point in the fused row pass).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_of(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
template <int M, int TPF>
__device__ __forceinline__ float left_of(const float (&x)[16], int tid) {
    const float l = wave_ror1(x[M]);
    const float e = x[(M + 15) & 15];
    float w;
    if constexpr (TPF == 64) w = lane_of(e, 63);
    else if constexpr (TPF == 32) w = __shfl(e, (int)((threadIdx.x % kWave) | 31));        // (two readlanes + a select measured slower than one ds_bpermute)
    else w = dpp_mov<0x120 + (17 - TPF)>(e);                         // row_ror: lane i <- lane i - (17 - TPF) = i + TPF - 1 (mod 16)
    return tid == 0 ? w : l;
}
template <int M, int TPF>
__device__ __forceinline__ float right_of(const float (&x)[16], int tid) {
    const float r = wave_rol1(x[M]);
    const float e = x[(M + 1) & 15];
    float w;
    if constexpr (TPF == 64) w = lane_of(e, 0);
    else if constexpr (TPF == 32) w = __shfl(e, (int)((threadIdx.x % kWave) & 32));
    else w = dpp_mov<0x120 + (TPF - 1)>(e);                          // row_ror: lane i <- lane i - (TPF - 1) (mod 16)
    return tid == TPF - 1 ? w : r;
}

// Which arithmetic a `precise` request gets.  The all-float32 mode (deriv_core, DIFF32) computes first derivatives at float32 accuracy
// for any input; its viscous term carries an amplification of nu |k| relative to them, rms nu pi N / (sqrt(3) L) over the spectrum
// (1.9 at the headline configuration: 1.1e-6 rel-L2 against the float64 oracle, 4e-6 worst case over nu <= 1 on resolved fields;
// tools/spec_accuracy_f32.py, profiles/r02_accuracy_f32diff.json).  precise = 1 takes it while that factor is <= 8 and the float64
// forward transform otherwise; precise = 0 always, precise >= 2 never (NNS_SPEC_F64=1 in the environment: as precise = 2).
inline bool pow2_in_range(int n) { return n >= 64 && n <= 1024 && (n & (n - 1)) == 0; }
constexpr double kF32AmpMax = 8.0;
inline bool spec_f32_mode(int precise, double nu, int n, double len) {
    if (!precise) return true;
    static const bool force64 = [] { const char* e = getenv("NNS_SPEC_F64"); return e && atoi(e) != 0; }();
    if (precise >= 2 || force64) return false;
    return std::fabs(nu) * M_PI * n / (1.7320508075688772 * std::fabs(len)) <= kF32AmpMax;
}

// A WHOLE evaluation (both directions) decides once: all-float32 only if BOTH transformed axes allow it -- an anisotropic grid must not run
// its two passes in different arithmetic.  Returns the `precise` value to hand to the per-axis passes: 0 or 2.  (Axes served by the
// dense circulant path do not take part: they are float64 throughout.)
inline int spec_resolve_precise(int precise, double nu, int nx, double Lx, int ny, double Ly) {
    const bool fx = !pow2_in_range(nx) || spec_f32_mode(precise, nu, nx, Lx);
    const bool fy = !pow2_in_range(ny) || spec_f32_mode(precise, nu, ny, Ly);
    return fx && fy ? 0 : 2;
}

// workgroups per launch (grid-stride over tiles): 2 generations per CU -- each generation pays the twiddle-table
// setup and one exposed first-tile load (2048 cost the x-pass 6 %, same-box sweep); NNS_SPEC_GRID overrides it for tuning
inline long spec_grid_cap() {
    static const long cap = [] { const char* e = getenv("NNS_SPEC_GRID"); const long v = e ? atol(e) : 0; return v > 0 ? v : 512L; }();
    return cap;
}


// spectral_dense.hip: the same operators on an axis of any length 3 .. 2048 as circulant matrices applied in float64 (O(n) per point;
// what every axis that is not a power of two in [64, 1024] gets).  Same partial-field convention as the FFT passes, so the two
// directions of one call may use different engines.
constexpr int kDenseMaxLen = 2048;
inline bool spec_len_ok(int n) { return pow2_in_range(n) || (n >= 3 && n <= kDenseMaxLen); }
int dense_xpass(const float* u, const float* v, const float* p, float* ru, float* rv, float* rd, int batch, int nx, int ny,
                double Lx, double rho, double nu, hipStream_t s);
int dense_ypass(const float* u, const float* v, const float* p, const float* up, const float* vp, float* ru, float* rv, float* rd,
                int batch, int nx, int ny, double dt, double Ly, double rho, double nu, hipStream_t s);
int dense_bwd_xpass(const float* u, const float* v, const float* ga, const float* gb, const float* gd, float* gu, float* gv, float* gp,
                    int batch, int nx, int ny, double Lx, double rho, double nu, hipStream_t s);
int dense_bwd_ypass(const float* u, const float* v, const float* ga, const float* gb, const float* gd, float* gu, float* gv, float* gp,
                    float* gup, float* gvp, int batch, int nx, int ny, double dt, double Ly, double rho, double nu, hipStream_t s);

template <typename F>
int dispatch_n(int n, F&& f) {
    switch (n) {
        case 64: return f(std::integral_constant<int, 64>{});
        case 128: return f(std::integral_constant<int, 128>{});
        case 256: return f(std::integral_constant<int, 256>{});
        case 512: return f(std::integral_constant<int, 512>{});
        case 1024: return f(std::integral_constant<int, 1024>{});
    }
    return fail(NNS_ERR_UNSUPPORTED, "spectral: axis length %d is not a power of two in [64, 1024] (the FFT engine's sizes)", n);
}


}  // namespace spec
}  // namespace nns
