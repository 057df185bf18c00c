// Kernel templates and launchers of the Fourier-spectral residual (forward direction), shared by the two translation units that
// instantiate them: spectral_kernels.hip (whole grids: the single-GPU entry points) and spectral_seg_kernels.hip (the slab-decomposed
// forms that read / write the all-to-all buffers in place).  Two units so that the instantiations of one do not disturb the register
// allocation of the other (the marching row pass runs at 252 of 256 VGPRs) and build in parallel.
//
// Fourier-spectral back-end of the periodic Navier-Stokes residual, gfx950.
// Operator definition: oracle/periodic.py (spectral_residual); no reference symbol exists
// (SURVEY.md section 8 row a17).
//
// The derivative operators are separable (d/dx touches axis 0 only, d/dy axis 1 only, and the
// Laplacian is their sum), so no 2-D transform is ever materialised in HBM:
//
//   x-pass (columns, axis 0):  P_u = u u_x + p_x/rho - nu u_xx,  P_v = u v_x - nu v_xx,  P_d = u_x
//   y-pass (rows,    axis 1):  r_u = (u-u_prev)/dt + P_u + v u_y        - nu u_yy
//                              r_v = (v-v_prev)/dt + P_v + v v_y + p_y/rho - nu v_yy
//                              r_div = P_d + v_y
//
// Each pass runs 1-D FFTs that live entirely in registers + LDS (fft_lds.h).  Per line it needs
// two forward and two inverse complex transforms, using linearity to pack real fields:
//   Z1 = FFT(u + i v), Z2 = FFT(p)        A = i k Z1                 -> ifft = u' + i v'
//                                         B = nu k^2 Z1 + (p-term)   -> ifft = L_u + i L_v
// (a derivative is a real linear operator, so it acts on the real and imaginary part of a packed
// signal independently; the Nyquist mode is dropped for odd derivatives, as in the oracle).
//
// Precision: with TF = double the forward transforms and the spectral multiply run in float64
// and the inverse in float32.  Forward rounding noise is white in k and the multiply amplifies it
// by k (k^2): all-float32 gives ~2e-4 rel-L2 at N = 1024, the mixed scheme 3e-7 (measured,
// DESIGN.md).  TF = float is the opt-in fast mode.
//
// HBM traffic (fp32 fields): x-pass 3 in + 3 out = 24 B/pt, y-pass 8 in + 3 out = 44 B/pt.
// Column access in the x-pass goes through an LDS transpose stage so that global accesses are
// 32..128-byte row pieces (a workgroup owns 8*FPW adjacent columns), and tiles that share
// 128-byte lines run on the same XCD (xcd_remap) so the line is fetched from HBM once.
#pragma once
#include "spectral_common.h"

namespace nns {
namespace spec {
namespace {

// The shared core: from the line's u, v, p (element tid + TPF*m in slot m) produce
//   a = (f_u', f_v')  and  b = (L_u, L_v)  with the pressure-gradient term added to the real part
//   (P_IN_REAL, x-pass) or the imaginary part (y-pass) of b.
// `hook` is called at the 4 * FftPasses<N> pass boundaries (slot numbers 0 .. 4 P - 1), see fft_line.
template <int N, typename TF, bool P_IN_REAL, typename Hook = NoHook>
__device__ __forceinline__ void deriv_core(const float (&uf)[16], const float (&vf)[16], const float (&pf)[16],
                                           C2<float> (&a)[16], C2<float> (&b)[16],
                                           const C2<TF>* tabF, const C2<float>* tabI, unsigned char* xb_raw, int tid,
                                           const SpecK& k, Hook&& hook = Hook{}) {
    const float* ctab = reinterpret_cast<const float*>(tabI + N / 2 + Pass2<N>::ENTRIES);      // behind the float32 tables (spec_setup)
    constexpr int P = FftPasses<N>::value;
    const C2<TF>* tabF2 = tabF + N / 2;
    const C2<float>* tabI2 = tabI + N / 2;
    C2<TF>* xbF = reinterpret_cast<C2<TF>*>(xb_raw);
    C2<float>* xbI = reinterpret_cast<C2<float>*>(xb_raw);
    C2<TF> z[16];
    // ---- pressure: Z2 = FFT(p); keep only its contribution to b, already scaled, in float
#ifndef NNS_P32
#define NNS_P32 1          // 1: the pressure transform in float32 on the forward-DIFFERENCED line (precise mode); 0: float64 FFT(p)
#endif
#ifndef NNS_F32_DIFF
#define NNS_F32_DIFF 1     // 1: the all-float32 mode transforms forward DIFFERENCES of u, v, p (bounded filters, see below); 0: the fields themselves
#endif
    constexpr bool DIFF32 = sizeof(TF) == 4 && NNS_F32_DIFF;
    if constexpr ((sizeof(TF) == 8 && NNS_P32) || DIFF32) {
        // p enters the residual only through its FIRST derivative.  FFT(p) in float32 would not do: its rounding noise, white
        // in k, is amplified by k (4e-5 of |p| at N = 1024).  The forward difference d_j = p_{j+1} - p_j (exact or correctly
        // rounded in float32) has FFT(d) = (e^{i theta} - 1) FFT(p), so
        //     (i k / rho) FFT(p) = M FFT(d),   M = (s / 2) (cot(theta / 2) - i),   s = k cs,   |M| <= (pi / 2) |k|_max cs / theta_max:
        // a BOUNDED filter on a float32 transform -- no amplification (3e-7 rel-L2), and one of the two
        // float64 forward transforms per line becomes a float32 one (measured 0.62 -> 0.58 ms column pass, 0.80 -> 0.77 row pass).
        C2<float> zf[16];
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            zf[m].x = right_of<m, N / 16>(pf, tid) - pf[m];
            zf[m].y = 0.f;
        });
        fft_line<float, N, false, 0>(zf, tabI, tabI2, xbI, tid, hook);
        int te = tid;
        asm volatile("" : "+v"(te), "+v"(zf[0].x));
        const float csh = (float)(0.5 * k.cs);
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            int ko, ke;
            wavenumber<N, m>(te, ko, ke);
            const float sh = (float)ko * csh;                               // s / 2 (0 at k = 0 and at the Nyquist mode, as the oracle drops it)
            const float c = ctab[ke < 0 ? -ke : ke] * csh;                  // (s / 2) cot(theta / 2), even in k; ctab[k] = k_odd cot(pi k / N)
            const float gx = c * zf[m].x + sh * zf[m].y, gy = c * zf[m].y - sh * zf[m].x;        // G = (i k / rho) FFT(p) / N
            if constexpr (P_IN_REAL) { b[m].x = gx; b[m].y = gy; } else { b[m].x = -gy; b[m].y = gx; }   // i G in the y-pass
        });
    } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) { z[m].x = (TF)pf[m]; z[m].y = (TF)0; }
    fft_line<TF, N, false, 0>(z, tabF, tabF2, xbF, tid, hook);
    // The per-element wavenumber factors depend only on the lane id: left alone, instruction selection
    // computes all 16 (fp64) during the butterflies above and spills them.  Tie the lane id to the
    // transform's output so they are computed here, where they are used.
    int te = tid;
    asm volatile("" : "+v"(te), "+v"(z[0].x));
    static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        int ko, ke;
        wavenumber<N, m>(te, ko, ke);
        const TF s = (TF)((double)ko * k.cs);
        if constexpr (P_IN_REAL) { b[m].x = (float)(-s * z[m].y); b[m].y = (float)(s * z[m].x); }   // (i k/rho) Z2
        else { b[m].x = (float)(-s * z[m].x); b[m].y = (float)(-s * z[m].y); }                       // i (i k/rho) Z2
    });
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DIFF32) {
        // ---- velocity, all-float32: D = FFT(d), d_j = (u + i v)_{j+1} - (u + i v)_j, so that  FFT(u + i v) = D / (e^{i theta} - 1)  and
        //     i k   FFT(u + i v) = M1 D,   M1 = (k / 2) (cot(theta / 2) - i)                       (bounded: no noise amplification)
        //     nu k^2 FFT(u + i v) = F2 D,  F2 = -(nu k / 2) (k + i k cot(theta / 2))  = -i nu k M1  (amplification nu k relative to M1)
        // The differences remove the large smooth part of the fields before the float32 transform: its white rounding noise is then
        // relative to |d| ~ h |u_x| instead of |u|, and the first derivatives come out at float32 accuracy.  The viscous term keeps an
        // amplification of nu |k| relative to the first derivatives (rms nu pi N / (sqrt(3) L)); the host picks this mode for a
        // `precise` request only while that factor is small (spec_precise_in_f32), and the float64 forward transform otherwise.
        C2<float> zv[16];
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            zv[m].x = right_of<m, N / 16>(uf, tid) - uf[m];
            zv[m].y = right_of<m, N / 16>(vf, tid) - vf[m];
        });
        fft_line<float, N, false, P>(zv, tabI, tabI2, xbI, tid, hook);
        int te = tid;
        asm volatile("" : "+v"(te), "+v"(zv[0].x));
        const float c1h = (float)(0.5 * k.c1), c2h = (float)(0.5 * k.c2);
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            int ko, ke;
            wavenumber<N, m>(te, ko, ke);
            const float ct = ctab[ke < 0 ? -ke : ke];                        // k cot(pi k / N), even in k, 0 at k = 0 and at the Nyquist mode
            const float ar = ct * c1h, ai = (float)ko * c1h;                 // M1 = ar - i ai
            a[m].x = ar * zv[m].x + ai * zv[m].y; a[m].y = ar * zv[m].y - ai * zv[m].x;
            const float kf = (float)ke * c2h;
            const float br = kf * (float)ke, bi = kf * ct;                   // F2 = -(br + i bi)
            b[m].x -= br * zv[m].x - bi * zv[m].y; b[m].y -= br * zv[m].y + bi * zv[m].x;
        });
    } else {
    // ---- velocity: Z1 = FFT(u + i v)
#pragma unroll
    for (int m = 0; m < 16; ++m) { z[m].x = (TF)uf[m]; z[m].y = (TF)vf[m]; }
    fft_line<TF, N, false, P>(z, tabF, tabF2, xbF, tid, hook);
    int te = tid;
    asm volatile("" : "+v"(te), "+v"(z[0].x));
    static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        int ko, ke;
        wavenumber<N, m>(te, ko, ke);
        const TF k1 = (TF)((double)ko * k.c1);
        const TF k2 = (TF)((double)(ke * ke) * k.c2);                         // ke^2 <= 2^18: exact
        a[m].x = (float)(-k1 * z[m].y); a[m].y = (float)(k1 * z[m].x);        // i k Z1
        b[m].x += (float)(k2 * z[m].x); b[m].y += (float)(k2 * z[m].y);       // nu k^2 Z1
    });
    }
    // ---- inverse transforms in float32
    __builtin_amdgcn_sched_barrier(0);
    fft_line<float, N, true, 2 * P>(a, tabI, tabI2, xbI, tid, hook);
    __builtin_amdgcn_sched_barrier(0);
    fft_line<float, N, true, 3 * P>(b, tabI, tabI2, xbI, tid, hook);
    __builtin_amdgcn_sched_barrier(0);
}

// Constants of the fused FD 5-point residual (nns_residual_both_f32): the y-pass owns whole rows, so the stencil's j-1 /
// j+1 neighbours of u and v are already in its registers (adjacent lane, or the adjacent 64-column slot at the wave's
// ends) and the rows i-1 / i+1 are re-read from L2 -- the stencil back-end then costs no second pass over the inputs.
struct FdK { float inv_2dx, inv_2dy, inv_rho, nu; double inv_dx2, inv_dy2; float inv_dx2f, inv_dy2f; };
// Row slabs of a grid sharded over ranks (nns/slab.py): the stencil's row above local row 0 / below local row nx-1 comes from
// the neighbour rank's edge rows, delivered as [u, v, p][grid][N] messages (top / bot; stride = grids * N).  NULL: the
// rows wrap around inside the local grid (single process).
struct HaloK { const float* top; const float* bot; long fstride; };
// The column pass's three partials when the grid is slab-decomposed (nns/slab.py): they arrive from the return all-to-all as
// [src][field][grid][nx_local][2^shift columns] -- a row of a local grid is P pieces of ny / P floats, one per source rank, `sstride`
// elements apart -- and the row pass reads them there (SEGP = true) instead of from a copy permuted into row slabs; the finished
// residuals still go to ru, rv, rd (row slabs).  pu, pv, pd: the three fields' blocks of source rank 0.
struct PartK { const float* pu; const float* pv; const float* pd; int shift; unsigned sstride_bytes; };
// BYTE offset of slot column jm (a multiple of TPF: wave-uniform, a scalar register) in the segmented layout, relative to the row's offset inside
// one source rank's block.  The three partial fields share it, and a load is  buffer_load_dword v, v_row, s[descriptor], s_col offen : base
// (descriptor) + lane's row offset (ONE vector register per row) + this scalar.  (First version: `global_load_dword v, v_row, s[pu + col]` -- the
// compiler hoisted the 16 slots x 3 fields of 64-bit scalar bases out of the row loop, 96 scalar registers that spilled into vector lanes: 64
// bytes of scratch per lane and a row pass of 0.84 instead of 0.72 ms at 1024^2 x 64.)
__device__ __forceinline__ int seg_col(const PartK& pk, int jm) { return (int)((unsigned)(jm >> pk.shift) * pk.sstride_bytes + (unsigned)(jm & ((1 << pk.shift) - 1)) * 4u); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t seg_rsrc(const float* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xffffffffu, 0x00020000); }
template <bool NT> __device__ __forceinline__ float ld_seg(__amdgpu_buffer_rsrc_t r, unsigned row_bytes, int col_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)row_bytes, col_bytes, NT ? 2 : 0));      // aux bit 1: nt
}
#ifndef NNS_YPASS_NT
#define NNS_YPASS_NT 1            // non-temporal hints on the fused row pass's write-once outputs and read-once streams (0.84 -> 0.78 ms)
#endif
#ifndef NNS_YPASS_NT_PLAIN
#define NNS_YPASS_NT_PLAIN 0      // the same hints in the plain (not fused) row pass
#endif
template <bool NT> __device__ __forceinline__ float ld_stream(const float* q) { if constexpr (NT) return __builtin_nontemporal_load(q); else return *q; }
template <bool NT> __device__ __forceinline__ void st_stream(float* q, float x) { if constexpr (NT) __builtin_nontemporal_store(x, q); else *q = x; }

// ------------------------------------------------------------------------------------------
// y-pass: rows (contiguous lines).  One line per TPF lanes; a workgroup iteration handles
// LINES rows; grid-stride over all batch*nx rows.
// ------------------------------------------------------------------------------------------
// FUSE_FD: also evaluates the FD 5-point residual of the same inputs into fu, fv, fd (fd_residual's
// formula, float64 Laplacian) -- "stencil + spectral residual on the same inputs" in one pass over the rows.
template <int N, typename TF, bool FUSE_FD = false, bool SEGP = false>
__global__ __launch_bounds__(kSpecThreads) void spec_ypass_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                                   const float* __restrict__ p, const float* __restrict__ up,
                                                                   const float* __restrict__ vp, float* __restrict__ ru,
                                                                   float* __restrict__ rv, float* __restrict__ rd,
                                                                   float* __restrict__ fu, float* __restrict__ fv, float* __restrict__ fd,
                                                                   int nx, FdK fk, long nrows, SpecK k, HaloK hk, PartK pk) {
    using L = SpecLds<N, TF>;
    constexpr int TPF = L::TPF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<TF>* tabF; C2<float>* tabI; unsigned char* lines;
    spec_setup<N, TF>(smem, tabF, tabI, lines);
    const long niter = (nrows + L::LINES - 1) / L::LINES;
    // element offset of this lane's first element of the line it owns in iteration `it` (row clamped into range)
    auto line_base = [&](long it) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave;
        const long row_raw = it * L::LINES + wave * L::FPW + lane / TPF;
        return (size_t)(row_raw < nrows ? row_raw : nrows - 1) * N + lane % TPF;
    };
    // Software pipeline: the NEXT line's u, v, p are requested when the inverse transforms start (registers are
    // slack there: the float64 spectra are dead) and are consumed at the top of the next iteration.
    float nu[16], nv[16], np[16];
#ifndef NNS_YPASS_XCD
#define NNS_YPASS_XCD 0
#endif
    long it = NNS_YPASS_XCD ? (long)xcd_remap(blockIdx.x, gridDim.x) : (long)blockIdx.x;
    if (it >= niter) return;
    {
        const size_t b0 = line_base(it);
#pragma unroll
        for (int m = 0; m < 16; ++m) { nu[m] = u[b0 + TPF * m]; nv[m] = v[b0 + TPF * m]; np[m] = p[b0 + TPF * m]; }
    }
    for (; it < niter; it += gridDim.x) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave;
        const int sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * L::LINE_BYTES;
        const bool valid = it * L::LINES + line < nrows;
        // Per-lane loop invariants (wavenumber factors, twiddle reads) would be hoisted out of this loop
        // by LICM and pinned in ~150 VGPRs for the whole body: make the lane id opaque per iteration.
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        const size_t base = line_base(it);
        // SEGP: byte offset of this lane's first column inside one source rank's block of the partials ([grid][nx][2^shift]: row index = the same
        // flattened batch * nx + i as `base`'s)
        unsigned pbyte = 0;
        if constexpr (SEGP) pbyte = (unsigned)((((base - (size_t)tidv) / N) << pk.shift) + (size_t)tidv) * 4u;
        // Memory phases are batched (all loads of a phase in flight together); the r_* arrays are read and written
        // through the same pointers, so an interleaved load/compute/store loop would be serialised by the compiler.
        float uf[16], vf[16], pf[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) { uf[m] = nu[m]; vf[m] = nv[m]; pf[m] = np[m]; }
        const size_t nbase = line_base(it + gridDim.x < niter ? it + gridDim.x : it);
        auto hook = [&](auto sc) {
            if constexpr (decltype(sc)::value == 2 * FftPasses<N>::value - 1) {          // after the last forward pass
#pragma unroll
                for (int m = 0; m < 16; ++m) { nu[m] = u[nbase + TPF * m]; nv[m] = v[nbase + TPF * m]; np[m] = p[nbase + TPF * m]; }
            }
        };
        C2<float> a[16], b[16];
        deriv_core<N, TF, false>(uf, vf, pf, a, b, tabF, tabI, xb, tidv, k, hook);
        // (touching the epilogue's five streams one transform ahead, one dword per 64 bytes, to have the lines in L2 when the
        //  epilogue asks: 0.783 -> 0.805 ms at every slot tried, round-2 A/B -- the second wave of the SIMD already covers that latency)
        // epilogue in two halves (bounds the registers in flight next to the prefetched line): u_prev, v_prev and the
        // x-pass partials in, residuals out
        float tu[16], tv[16];                                            // (u - u_prev)/dt, (v - v_prev)/dt: shared by both back-ends
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float pu[8], pv[8], pd[8], qu[8], qv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const size_t c = base + TPF * (8 * h + i);
                constexpr bool NT = (FUSE_FD || NNS_YPASS_NT_PLAIN) && NNS_YPASS_NT;
                if constexpr (SEGP) {
                    const int so = seg_col(pk, TPF * (8 * h + i));
                    pu[i] = ld_seg<NT>(seg_rsrc(pk.pu), pbyte, so); pv[i] = ld_seg<NT>(seg_rsrc(pk.pv), pbyte, so); pd[i] = ld_seg<NT>(seg_rsrc(pk.pd), pbyte, so);
                } else {
                pu[i] = ld_stream<NT>(ru + c); pv[i] = ld_stream<NT>(rv + c); pd[i] = ld_stream<NT>(rd + c);
                }
                qu[i] = ld_stream<NT>(up + c); qv[i] = ld_stream<NT>(vp + c);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = 8 * h + i;
                // explicitly rounded products: the fused and the plain instantiation must not differ by an FMA contraction here
                tu[m] = __fmul_rn(uf[m] - qu[i], k.inv_dt); tv[m] = __fmul_rn(vf[m] - qv[i], k.inv_dt);
            }
            if (valid) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = 8 * h + i;
                    const size_t c = base + TPF * m;
                    // one fixed rounding sequence (the compiler may not re-associate or contract differently per instantiation)
                    constexpr bool NT = (FUSE_FD || NNS_YPASS_NT_PLAIN) && NNS_YPASS_NT;
                    st_stream<NT>(ru + c, __fadd_rn(__fmaf_rn(vf[m], a[m].x, __fadd_rn(tu[m], pu[i])), b[m].x));
                    st_stream<NT>(rv + c, __fadd_rn(__fmaf_rn(vf[m], a[m].y, __fadd_rn(tv[m], pv[i])), b[m].y));
                    st_stream<NT>(rd + c, pd[i] + a[m].y);
                }
            }
        }
        if constexpr (FUSE_FD) {
            // The stencil back-end, AFTER the spectral epilogue: the spectral derivatives (64 registers) are dead by now,
            // which is what lets eight slots of eight streams be in flight per phase; the time-derivative terms are kept
            // from the spectral epilogue.
            __builtin_amdgcn_sched_barrier(0);
            const long row_raw = it * L::LINES + line;
            const long row = row_raw < nrows ? row_raw : nrows - 1;
            const long gi = row / nx, ii = row % nx;
            // one base pointer per stream (lane's first column included), slots at compile-time offsets TPF * m: the loads
            // then take immediate offsets instead of eighty precomputed 64-bit addresses
            const size_t bm = (size_t)(gi * nx + (ii == 0 ? nx - 1 : ii - 1)) * N + tidv;       // rows i-1, i+1 (periodic in the grid), i
            const size_t bp = (size_t)(gi * nx + (ii == nx - 1 ? 0 : ii + 1)) * N + tidv;
            const size_t bc = (size_t)row * N;
#ifdef NNS_ROWPASS_EXP              // timing experiment (wrong results): the stencil's rows i-1 / i+1 read from row i (no second touch of other rows)
            const float* um_p = u + bc + tidv; const float* un_p = um_p; const float* vm_p = v + bc + tidv; const float* vn_p = vm_p;
            const float* pm_p = p + bc + tidv; const float* pn_p = pm_p;
            (void)bm; (void)bp;
#else
            const float* um_p = u + bm; const float* un_p = u + bp; const float* vm_p = v + bm; const float* vn_p = v + bp;
            const float* pm_p = p + bm; const float* pn_p = p + bp;
#endif
            if (hk.top && ii == 0) { const float* h = hk.top + (size_t)gi * N + tidv; um_p = h; vm_p = h + hk.fstride; pm_p = h + 2 * hk.fstride; }
            if (hk.bot && ii == nx - 1) { const float* h = hk.bot + (size_t)gi * N + tidv; un_p = h; vn_p = h + hk.fstride; pn_p = h + 2 * hk.fstride; }
            const float* pl_p = p + bc + tidv - 1;                          // column - 1: wraps only for column 0 (slot 0 of lane 0)
            const float* pr_p = p + bc + tidv + 1;                          // column + 1: wraps only for column N-1 (slot 15 of lane 63)
            const float* pl0_p = p + bc + ((tidv + N - 1) & (N - 1));
            const float* pr15_p = p + bc + ((tidv + TPF * 15 + 1) & (N - 1));
            static_for<0, 2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                float um[8], un_[8], vm[8], vn_[8], pm[8], pn_[8], pl[8], pr[8];
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int m = 8 * h + i;
                    um[i] = um_p[TPF * m]; un_[i] = un_p[TPF * m]; vm[i] = vm_p[TPF * m]; vn_[i] = vn_p[TPF * m];
                    pm[i] = pm_p[TPF * m]; pn_[i] = pn_p[TPF * m];
                    // p's j-1 / j+1 by lane rotates of the row in registers, like u's and v's (two of the eight re-read streams gone).  With the
                    // float64 Laplacian this spilled 21 registers (0.78 -> 0.91 ms); with the float32 one it fits: 252 VGPRs, 0.765 -> 0.757 ms
#ifndef NNS_ROWPASS_PROT
#define NNS_ROWPASS_PROT 1
#endif
                    if constexpr (NNS_ROWPASS_PROT) { pl[i] = left_of<m, TPF>(pf, tidv); pr[i] = right_of<m, TPF>(pf, tidv); }
                    else {
                    if constexpr (m == 0) pl[i] = pl0_p[0]; else pl[i] = pl_p[TPF * m];
                    if constexpr (m == 15) pr[i] = pr15_p[0]; else pr[i] = pr_p[TPF * m];
                    }
                });
                if (valid) {
                    static_for<0, 8>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        constexpr int m = 8 * h + i;
                        const size_t c = base + TPF * m;
                        const float ucc = uf[m], vcc = vf[m];
                        const float ul = left_of<m, TPF>(uf, tidv), ur = right_of<m, TPF>(uf, tidv);
                        const float vl = left_of<m, TPF>(vf, tidv), vr = right_of<m, TPF>(vf, tidv);
                        const float ux = (un_[i] - um[i]) * fk.inv_2dx, uy = (ur - ul) * fk.inv_2dy;
                        const float vx = (vn_[i] - vm[i]) * fk.inv_2dx, vy = (vr - vl) * fk.inv_2dy;
                        const float px = (pn_[i] - pm[i]) * fk.inv_2dx, py = (pr[i] - pl[i]) * fk.inv_2dy;
#ifndef NNS_FUSED_LAP32
#define NNS_FUSED_LAP32 1         // 1: second differences as differences of (exact) first differences in float32; 0: float64 sums
#endif
                        float lu, lv;
                        if constexpr (NNS_FUSED_LAP32) {
                            // (a - c) - (c - b): neighbouring values of a resolved field are within a factor 2 of each other, so both
                            // first differences are EXACT in float32 (Sterbenz) and the only rounding is relative to the second
                            // difference itself -- the accuracy of the float64 sum without ~28 double-rate instructions per point
                            // (this pass is VALU-issue-bound, unlike the standalone stencil kernel, which keeps the float64 form)
                            lu = ((un_[i] - ucc) - (ucc - um[i])) * fk.inv_dx2f + ((ur - ucc) - (ucc - ul)) * fk.inv_dy2f;
                            lv = ((vn_[i] - vcc) - (vcc - vm[i])) * fk.inv_dx2f + ((vr - vcc) - (vcc - vl)) * fk.inv_dy2f;
                        } else {
                            lu = (float)(((double)un_[i] - 2.0 * ucc + (double)um[i]) * fk.inv_dx2 + ((double)ur - 2.0 * ucc + (double)ul) * fk.inv_dy2);
                            lv = (float)(((double)vn_[i] - 2.0 * vcc + (double)vm[i]) * fk.inv_dx2 + ((double)vr - 2.0 * vcc + (double)vl) * fk.inv_dy2);
                        }
                        st_stream<NNS_YPASS_NT>(fu + c, tu[m] + ucc * ux + vcc * uy + px * fk.inv_rho - fk.nu * lu);
                        st_stream<NNS_YPASS_NT>(fv + c, tv[m] + ucc * vx + vcc * vy + py * fk.inv_rho - fk.nu * lv);
                        st_stream<NNS_YPASS_NT>(fd + c, ux + vy);
                    });
                }
            });
        }
    }
}

// ------------------------------------------------------------------------------------------
// The fused row pass, MARCHING form (all-float32 mode).  spec_ypass_kernel<N, TF, true> hands a workgroup eight consecutive rows per
// iteration and re-reads the stencil's rows i-1 / i+1 of u, v, p from memory one and a half iterations after they were first
// touched -- by then they have left the 4 MB L2 of the XCD (its contents turn over every ~5 us), so six of the pass's twenty
// streams are fetched twice (measured 1.31x the algorithmic bytes).  Here every LINE walks down its own chunk of R consecutive rows
// of one grid: the row below is the next row's prefetch, which the pass issues anyway, and the row above is the row the line has
// just finished -- u and v parked in a lane-private LDS image behind the exchange buffer (the all-float32 mode leaves the room:
// 8.5 + 8 KB per line), p in 16 registers.  The stencil phase then issues no global load at all; per chunk two extra rows are read
// (2 / R of three streams).  Lines are independent, no workgroup barrier anywhere, as before.
// ------------------------------------------------------------------------------------------
#ifndef NNS_MARCH_NT_IN
#define NNS_MARCH_NT_IN 1          // non-temporal hint on the row prefetch (each row is read once, bar the chunk edges): +0.5 % (same-box A/B)
#endif
template <int N>
struct MarchLds {
    using L = SpecLds<N, float>;
    static constexpr int XB = (L::XB_BYTES + 127) / 128 * 128;
    static constexpr int PARK = 2 * N * 4;                          // u, v of the row above: float4 [8][TPF], lane-private
    static constexpr int LINE_BYTES = XB + PARK;
    static constexpr int TOTAL = L::TABF_BYTES + L::TABI_BYTES + L::LINES * LINE_BYTES;
};
template <int N, bool SEGP = false>
__global__ __launch_bounds__(kSpecThreads) void spec_rowmarch_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                                      const float* __restrict__ p, const float* __restrict__ up,
                                                                      const float* __restrict__ vp, float* __restrict__ ru,
                                                                      float* __restrict__ rv, float* __restrict__ rd,
                                                                      float* __restrict__ fu, float* __restrict__ fv, float* __restrict__ fd,
                                                                      int nx, FdK fk, SpecK k, HaloK hk, int R, int chunks_per_grid, long nchunks, PartK pk) {
    using L = SpecLds<N, float>;
    using ML = MarchLds<N>;
    constexpr int TPF = L::TPF;
    static_assert(ML::TOTAL <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<float>* tabF; C2<float>* tabI; unsigned char* lines;
    spec_setup<N, float>(smem, tabF, tabI, lines);
    const long TL = (long)gridDim.x * L::LINES;
    const long njt = (nchunks + TL - 1) / TL;
    for (long j = 0; j < njt; ++j) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave;
        const int sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * ML::LINE_BYTES;
        float4* park = reinterpret_cast<float4*>(xb + ML::XB) + tid;           // [q] at park[q * TPF]: q = 0..3 u, 4..7 v (slots 4 q' .. 4 q' + 3)
        const long c_raw = (long)blockIdx.x * L::LINES + line + j * TL;
        const bool active = c_raw < nchunks;
        const long c = active ? c_raw : nchunks - 1;
        const long g = c / chunks_per_grid;
        const int r0 = (int)(c % chunks_per_grid) * R;
        const int len = active ? (nx - r0 < R ? nx - r0 : R) : 0;
        const size_t gbase = (size_t)g * nx * N + tid;
        // u, v, p pointers of row ii of this grid (lane's first column included); ii = -1 / nx: the periodic wrap or the neighbour rank's edge row
        auto row_ptrs = [&](int ii, const float*& a, const float*& b, const float*& cq) {
            const int iw = ii < 0 ? nx - 1 : (ii >= nx ? 0 : ii);
            const size_t o = gbase + (size_t)iw * N;
            a = u + o; b = v + o; cq = p + o;
            if (hk.top && ii < 0) { const float* h = hk.top + (size_t)g * N + tid; a = h; b = h + hk.fstride; cq = h + 2 * hk.fstride; }
            if (hk.bot && ii >= nx) { const float* h = hk.bot + (size_t)g * N + tid; a = h; b = h + hk.fstride; cq = h + 2 * hk.fstride; }
        };
        float nu[16], nv[16], np[16], pm[16];
        {   // chunk prologue: the row above the chunk (parked) and the chunk's first row (as the "next" row)
            const float *a, *b, *cq;
            row_ptrs(r0 - 1, a, b, cq);
            float t0[16], t1[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { t0[m] = a[TPF * m]; t1[m] = b[TPF * m]; pm[m] = cq[TPF * m]; }
            row_ptrs(r0, a, b, cq);
#pragma unroll
            for (int m = 0; m < 16; ++m) { nu[m] = a[TPF * m]; nv[m] = b[TPF * m]; np[m] = cq[TPF * m]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                park[q * TPF] = make_float4(t0[4 * q], t0[4 * q + 1], t0[4 * q + 2], t0[4 * q + 3]);
                park[(4 + q) * TPF] = make_float4(t1[4 * q], t1[4 * q + 1], t1[4 * q + 2], t1[4 * q + 3]);
            }
        }
        for (int r = 0; r < R; ++r) {
            const bool valid = r < len;
            const int ii = r0 + (valid ? r : (len > 0 ? len - 1 : 0));
            int tidv = tid;
            asm volatile("" : "+v"(tidv));
            float uf[16], vf[16], pf[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { uf[m] = nu[m]; vf[m] = nv[m]; pf[m] = np[m]; }
            const float *na, *nb, *nc;
            row_ptrs(ii + 1, na, nb, nc);
            auto hook = [&](auto sc) {
                if constexpr (decltype(sc)::value == 2 * FftPasses<N>::value - 1) {          // after the last forward pass: the row below
#pragma unroll
                    for (int m = 0; m < 16; ++m) { nu[m] = ld_stream<NNS_MARCH_NT_IN>(na + TPF * m); nv[m] = ld_stream<NNS_MARCH_NT_IN>(nb + TPF * m); np[m] = ld_stream<NNS_MARCH_NT_IN>(nc + TPF * m); }
                }
            };
            C2<float> a[16], b[16];
            deriv_core<N, float, false>(uf, vf, pf, a, b, tabF, tabI, xb, tidv, k, hook);
            // p of this row and of the row above are needed by the stencil only: over the epilogue (the register peak of the iteration) they
            // wait in the exchange image, which is idle until the next row's transforms
            float4* pk2 = reinterpret_cast<float4*>(xb) + tidv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pk2[q * TPF] = make_float4(pf[4 * q], pf[4 * q + 1], pf[4 * q + 2], pf[4 * q + 3]);
                pk2[(4 + q) * TPF] = make_float4(pm[4 * q], pm[4 * q + 1], pm[4 * q + 2], pm[4 * q + 3]);
            }
            const size_t base = gbase + (size_t)ii * N;
            unsigned pbyte = 0;                                                 // SEGP: this lane's first column of row ii inside one source rank's block
            if constexpr (SEGP) pbyte = (unsigned)((((size_t)g * nx + (size_t)ii) << pk.shift) + (size_t)tid) * 4u;
            float tu[16], tv[16];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float pu[8], pv[8], pd[8], qu[8], qv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const size_t c2 = base + TPF * (8 * h + i);
                    if constexpr (SEGP) {
                        const int so = seg_col(pk, TPF * (8 * h + i));
                        pu[i] = ld_seg<NNS_YPASS_NT>(seg_rsrc(pk.pu), pbyte, so); pv[i] = ld_seg<NNS_YPASS_NT>(seg_rsrc(pk.pv), pbyte, so); pd[i] = ld_seg<NNS_YPASS_NT>(seg_rsrc(pk.pd), pbyte, so);
                    } else {
                    pu[i] = ld_stream<NNS_YPASS_NT>(ru + c2); pv[i] = ld_stream<NNS_YPASS_NT>(rv + c2); pd[i] = ld_stream<NNS_YPASS_NT>(rd + c2);
                    }
                    qu[i] = ld_stream<NNS_YPASS_NT>(up + c2); qv[i] = ld_stream<NNS_YPASS_NT>(vp + c2);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = 8 * h + i;
                    tu[m] = __fmul_rn(uf[m] - qu[i], k.inv_dt); tv[m] = __fmul_rn(vf[m] - qv[i], k.inv_dt);
                }
                if (valid) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int m = 8 * h + i;
                        const size_t c2 = base + TPF * m;
                        st_stream<NNS_YPASS_NT>(ru + c2, __fadd_rn(__fmaf_rn(vf[m], a[m].x, __fadd_rn(tu[m], pu[i])), b[m].x));
                        st_stream<NNS_YPASS_NT>(rv + c2, __fadd_rn(__fmaf_rn(vf[m], a[m].y, __fadd_rn(tv[m], pv[i])), b[m].y));
                        st_stream<NNS_YPASS_NT>(rd + c2, pd[i] + a[m].y);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the stencil: row i-1 from the park image / pm, row i+1 = the prefetched next row, columns j-1 / j+1 by lane rotates
            {
                int tq = tid;
                asm volatile("" : "+v"(tq));                                    // a different address to the compiler: no store-to-load forwarding across the epilogue
                const float4* rk = reinterpret_cast<const float4*>(xb) + tq;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x = rk[q * TPF], y = rk[(4 + q) * TPF];
                    pf[4 * q] = x.x; pf[4 * q + 1] = x.y; pf[4 * q + 2] = x.z; pf[4 * q + 3] = x.w;
                    pm[4 * q] = y.x; pm[4 * q + 1] = y.y; pm[4 * q + 2] = y.z; pm[4 * q + 3] = y.w;
                }
            }
            static_for<0, 2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                const float4 u0 = park[(2 * h) * TPF], u1 = park[(2 * h + 1) * TPF], v0 = park[(4 + 2 * h) * TPF], v1 = park[(5 + 2 * h) * TPF];
                const float um[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
                const float vm[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (valid) static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int m = 8 * h + i;
                    const size_t c2 = base + TPF * m;
                    const float ucc = uf[m], vcc = vf[m];
                    const float ul = left_of<m, TPF>(uf, tidv), ur = right_of<m, TPF>(uf, tidv);
                    const float vl = left_of<m, TPF>(vf, tidv), vr = right_of<m, TPF>(vf, tidv);
                    const float pl = left_of<m, TPF>(pf, tidv), pr = right_of<m, TPF>(pf, tidv);
                    const float ux = (nu[m] - um[i]) * fk.inv_2dx, uy = (ur - ul) * fk.inv_2dy;
                    const float vx = (nv[m] - vm[i]) * fk.inv_2dx, vy = (vr - vl) * fk.inv_2dy;
                    const float px = (np[m] - pm[m]) * fk.inv_2dx, py = (pr - pl) * fk.inv_2dy;
                    const float lu = ((nu[m] - ucc) - (ucc - um[i])) * fk.inv_dx2f + ((ur - ucc) - (ucc - ul)) * fk.inv_dy2f;
                    const float lv = ((nv[m] - vcc) - (vcc - vm[i])) * fk.inv_dx2f + ((vr - vcc) - (vcc - vl)) * fk.inv_dy2f;
                    st_stream<NNS_YPASS_NT>(fu + c2, tu[m] + ucc * ux + vcc * uy + px * fk.inv_rho - fk.nu * lu);
                    st_stream<NNS_YPASS_NT>(fv + c2, tv[m] + ucc * vx + vcc * vy + py * fk.inv_rho - fk.nu * lv);
                    st_stream<NNS_YPASS_NT>(fd + c2, ux + vy);
                });
            });
            // this row becomes the row above
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                park[q * TPF] = make_float4(uf[4 * q], uf[4 * q + 1], uf[4 * q + 2], uf[4 * q + 3]);
                park[(4 + q) * TPF] = make_float4(vf[4 * q], vf[4 * q + 1], vf[4 * q + 2], vf[4 * q + 3]);
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) pm[m] = pf[m];
        }
    }
}

// ------------------------------------------------------------------------------------------
// x-pass: columns.  A workgroup owns LINES adjacent columns of one grid; the tile is staged
// through LDS so that global accesses are row pieces of LINES*4 bytes.
// ------------------------------------------------------------------------------------------
// PREFETCH = false falls back to loading each tile at the top of its own iteration (instantiations whose
// register allocation does not fit the extra 48 staging registers without spilling).
// SEG (slab-decomposed grids, nns/slab.py): the rows of a column slab arrive from the all-to-all in blocks of 2^shift rows
// per source rank, [src][field][grid][2^shift][ny]; row r of a grid then starts at (r >> shift) * stride + (r & mask) * ny
// and consecutive grids are 2^shift * ny apart.  The column pass reads that layout and writes its partials in the same
// layout (= the send buffer of the return all-to-all), so no permuting copy stands on either side of it.
struct SegK { int shift; long stride; };
template <int N, typename TF, bool PREFETCH, bool SEG = false>
__global__ __launch_bounds__(kSpecThreads) void spec_xpass_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                                   const float* __restrict__ p, float* __restrict__ ru,
                                                                   float* __restrict__ rv, float* __restrict__ rd,
                                                                   int ny, int tiles_per_grid, long ntiles, SpecK k, SegK sg) {
    auto row_off = [&](int r) -> size_t {
        if constexpr (SEG) return (size_t)(r >> sg.shift) * sg.stride + (size_t)(r & ((1 << sg.shift) - 1)) * ny;
        else return (size_t)r * ny;
    };
    using L = SpecLds<N, TF>;
    constexpr int TPF = L::TPF, CW = L::LINES, SF = L::STAGE_F;
    constexpr int ROWS_PER_IT = kSpecThreads / CW;
    constexpr int NR = N / ROWS_PER_IT;                 // staged elements per thread and field (= 16 for every N)
    static_assert(NR == 16, "staging geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<TF>* tabF; C2<float>* tabI; unsigned char* lines;
    spec_setup<N, TF>(smem, tabF, tabI, lines);

    // Software pipeline over tiles (issue early / write late): the raw u, v, p of tile t+1 are loaded into
    // registers (su, sv, sp) while tile t is being transformed and are written to the LDS staging image only
    // after the transforms have released it -- global-load latency hides under the FFT phase.
    float su[NR], sv[NR], sp[NR];
    auto tile_coords = [&](long t, int& j0, size_t& g) {
#ifndef NNS_XPASS_REVERSE
#define NNS_XPASS_REVERSE 1
#endif
        // neighbouring tiles -> same XCD (shared lines); tiles are walked from the LAST grid to the first (see launch_xpass)
        const long lt = NNS_XPASS_REVERSE ? ntiles - 1 - (long)xcd_remap((unsigned)t, (unsigned)ntiles) : (long)xcd_remap((unsigned)t, (unsigned)ntiles);
        j0 = (int)(lt % tiles_per_grid) * CW;
        g = (size_t)(lt / tiles_per_grid) * (SEG ? ((size_t)ny << sg.shift) : (size_t)N * ny);
    };
    auto issue_loads = [&](long t) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int cc = tx % CW, cr = tx / CW;
        int j0; size_t g;
        tile_coords(t, j0, g);
        const bool ok = j0 + cc < ny;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const size_t c = g + row_off(cr + ROWS_PER_IT * i) + j0 + cc;
            su[i] = ok ? u[c] : 0.f; sv[i] = ok ? v[c] : 0.f; sp[i] = ok ? p[c] : 0.f;
        }
    };
    long t = blockIdx.x;
    if (PREFETCH && t < ntiles) issue_loads(t);
    for (; t < ntiles; t += gridDim.x) {
        if (!PREFETCH) issue_loads(t);
        // thread-constant indices are recomputed per tile from an opaque copy of the thread id: cheaper than the
        // spill/reload the compiler otherwise chooses for values that are live across the whole tile body
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave;
        const int sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * L::LINE_BYTES;
        float* my_stage = reinterpret_cast<float*>(xb) + (line % L::SKEW_MOD) * L::SKEW_DW;   // skewed: conflict-free staging
        int j0; size_t g;
        tile_coords(t, j0, g);
        {   // ---- staged registers -> LDS [line][field][row]
            const int cc = tx % CW, cr = tx / CW;
            float* cp_stage = reinterpret_cast<float*>(lines + (size_t)cc * L::LINE_BYTES) + (cc % L::SKEW_MOD) * L::SKEW_DW;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = cr + ROWS_PER_IT * i;
                cp_stage[0 * SF + r] = su[i]; cp_stage[1 * SF + r] = sv[i]; cp_stage[2 * SF + r] = sp[i];
            }
        }
        __syncthreads();
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        float uf[16], vf[16], pf[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            uf[m] = my_stage[0 * SF + tidv + TPF * m]; vf[m] = my_stage[1 * SF + tidv + TPF * m]; pf[m] = my_stage[2 * SF + tidv + TPF * m];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Prefetch of the next tile, TRICKLED: a burst of 48 loads per thread (each wave-instruction touching 8
        // separate 32-byte row pieces) keeps every wave of every CU in vector-memory issue at the same time --
        // measured 0.17 ms of a 0.79 ms kernel at 1024^2 x 64.  Instead a few row pieces are requested at each of
        // the 4 P FFT-pass boundaries, where other waves' arithmetic covers the issue time.
        const long tn = t + gridDim.x < ntiles ? t + gridDim.x : t;          // clamped: the last prefetch re-reads this tile
        int j0n; size_t gn;
        tile_coords(tn, j0n, gn);
        auto load_chunk = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            int ty = threadIdx.x;
            asm volatile("" : "+v"(ty));
            const int cc = ty % CW, cr = ty / CW;
            const int col = j0n + cc < ny ? j0n + cc : ny - 1;                // clamped column: no mask needed on a load
            const size_t c = gn + row_off(cr + ROWS_PER_IT * i) + col;
            su[i] = u[c]; sv[i] = v[c]; sp[i] = p[c];
        };
        auto hook = [&](auto sc) {
            if constexpr (PREFETCH) {
                constexpr int NSLOT = 4 * FftPasses<N>::value;
                constexpr int s = decltype(sc)::value;
                constexpr int c0 = NR * s / NSLOT, c1 = NR * (s + 1) / NSLOT;
                static_assert(c1 - c0 <= 2, "at most two chunks per slot");
                if constexpr (c1 > c0) load_chunk(std::integral_constant<int, c0>{});
                if constexpr (c1 > c0 + 1) load_chunk(std::integral_constant<int, c0 + 1>{});
            }
        };
        C2<float> a[16], b2[16];
        deriv_core<N, TF, true>(uf, vf, pf, a, b2, tabF, tabI, xb, tidv, k, hook);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            my_stage[0 * SF + tidv + TPF * m] = uf[m] * a[m].x + b2[m].x;     // P_u = u u_x + p_x/rho - nu u_xx
            my_stage[1 * SF + tidv + TPF * m] = uf[m] * a[m].y + b2[m].y;     // P_v = u v_x - nu v_xx
            my_stage[2 * SF + tidv + TPF * m] = a[m].x;                       // P_d = u_x
        }
        __syncthreads();
        {
            int ty = threadIdx.x;
            asm volatile("" : "+v"(ty));
            const int cc2 = ty % CW, cr2 = ty / CW;
            const float* st = reinterpret_cast<const float*>(lines + (size_t)cc2 * L::LINE_BYTES) + (cc2 % L::SKEW_MOD) * L::SKEW_DW;
            if (j0 + cc2 < ny) {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int r = cr2 + ROWS_PER_IT * i;
                    const size_t c = g + row_off(r) + j0 + cc2;
                    ru[c] = st[0 * SF + r];
                    rv[c] = st[1 * SF + r];
                    rd[c] = st[2 * SF + r];
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// x-pass, ROLE-SPLIT form (round 2).  spec_xpass_kernel above lets its eight transform waves also move the tile: loads trickled
// into the transforms, but the 3072 store row pieces of a tile are a burst in which no wave computes (7 of a tile's 19 us), and
// keeping the next tile in registers costs every wave 48 VGPRs.  Here a workgroup has TWELVE waves: eight TRANSFORM waves (one line
// each, as before, now with no global memory instruction at all) and four MEMORY waves that own the tile traffic: while the
// transform waves work on tile t they store tile t-1's partials from their registers and load tile t+1's inputs into the same
// registers (96 per lane); after the transforms ONE exchange step swaps LDS staging contents -- the memory waves take tile t's
// results out of the staging image and put tile t+1's inputs in, element for element (same slot) -- between two workgroup
// barriers.  The row-piece traffic (the 0.43 ms skeleton of this pass) then runs entirely under the transforms.  Without the
// prefetch registers the transform waves fit 3 waves per SIMD (<= 168 VGPRs), which is what gives the four extra waves a home.
// ------------------------------------------------------------------------------------------
// (kSplitThreads, SplitLds: spectral_common.h -- shared with the backward column pass)
template <int N, typename TF, bool SEG>
__global__ __launch_bounds__(kSplitThreads) void spec_xpass_split_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                                          const float* __restrict__ p, float* __restrict__ ru,
                                                                          float* __restrict__ rv, float* __restrict__ rd,
                                                                          int ny, int tiles_per_grid, long ntiles, SpecK k, SegK sg) {
    using L = SpecLds<N, TF>;
    using SL = SplitLds<N, TF>;
    constexpr int TPF = L::TPF, CW = L::LINES, SF = L::STAGE_F;
    constexpr int MROWS = 256 / CW;                       // rows of the tile one memory-wave instruction step covers
    constexpr int NR = N / MROWS;                         // elements per memory lane and field (= 32 for every N)
    static_assert(NR == 32, "tile geometry");
    // a memory lane owns rows 4 cr .. 4 cr + 3 of every block of 4 MROWS rows: element i <-> row 4 cr + (i & 3) + 4 MROWS (i >> 2), so
    // that its four consecutive rows move through LDS as ONE 16-byte access (24 + 24 LDS instructions per exchange instead of 96 + 96)
    static_assert(SL::TOTAL <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<TF>* tabF; C2<float>* tabI; unsigned char* lines;
    spec_setup<N, TF, kSplitThreads>(smem, tabF, tabI, lines);
#ifndef NNS_SPLIT_EXP
#define NNS_SPLIT_EXP 0            // timing experiments (wrong results): 1 = no transforms, 2 = no global traffic
#endif
    auto tile_coords = [&](long t, int& j0, size_t& g) {
        const long lt = NNS_XPASS_REVERSE ? ntiles - 1 - (long)xcd_remap((unsigned)t, (unsigned)ntiles) : (long)xcd_remap((unsigned)t, (unsigned)ntiles);
        j0 = (int)(lt % tiles_per_grid) * CW;
        g = (size_t)(lt / tiles_per_grid) * (SEG ? ((size_t)ny << sg.shift) : (size_t)N * ny);
    };
    long t = blockIdx.x;
    if (t >= ntiles) return;                               // uniform over the workgroup (the launch never has more workgroups than tiles)
    if (threadIdx.x >= kSpecThreads) {
        // ================= memory waves =================
        int mt = threadIdx.x - kSpecThreads;
        asm volatile("" : "+v"(mt));
        const int cc = mt % CW, cr = mt / CW;
        float* stage = reinterpret_cast<float*>(lines + (size_t)cc * SL::LINE_BYTES) + (cc % 8) * SL::SKEW_DW + 4 * cr;       // [field][row]
        float R[3][NR];
        // addresses: a uniform grid base (scalar registers) + a 32-bit BYTE offset per lane, recomputed per tile from an opaque
        // seed (left alone, the compiler keeps 32 precomputed 64-bit row offsets alive across the whole loop and spills them)
        // (a lane's four consecutive rows never straddle a segment: seg_rows is a power of two >= 4, checked on the host)
        // Round 3: BYTE offsets, so that a load is `global_load_dword v, v_off, s[base]` -- with element offsets every access carried a
        // 64-bit shift-and-add (96 v_lshl_add_u64 + 32 v_lshlrev_b64 per tile and direction, on the SIMDs the transform waves compute on).
        auto off32 = [&](int i, unsigned seed, unsigned crv) -> unsigned {
            const unsigned r0 = 4u * crv + 4u * MROWS * (unsigned)(i >> 2);
            unsigned o;
            if constexpr (SEG) o = (r0 >> sg.shift) * (unsigned)sg.stride + (r0 & ((1u << sg.shift) - 1u)) * (unsigned)ny;
            else o = r0 * (unsigned)ny;
            return (seed + o + (unsigned)(i & 3) * (unsigned)ny) * 4u;
        };
        auto at = [](const float* base, unsigned byte_off) -> const float& { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
        auto at_w = [](float* base, unsigned byte_off) -> float& { return *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off); };
#ifndef NNS_SPLIT_BUF
#define NNS_SPLIT_BUF 0            // 1: whole grids (not SEG): buffer loads / stores -- descriptor of the grid + ONE lane offset per tile + a scalar row offset per access
#endif
        // Round 4: with `global_load_dword v, v_off, s[base]` every row piece still costs two vector instructions (add the row
        // stride, shift to bytes: 128 per tile, direction and memory wave, on the SIMDs the transform waves compute on).  A buffer access takes
        // base (descriptor, 4 SGPRs) + lane offset (ONE VGPR per tile) + scalar offset (the row: scalar unit) -- no vector instruction per access
        // (checked in the ISA: 192 buffer_load_dword / 192 buffer_store_dword per tile loop, `s_off offen`, no v_add / v_lshl between them).
        // MEASURED, same box, three rounds (profiles/r04_ab_xpass_buffer_addressing.log): column pass 0.493 ms against 0.490 with the global_
        // form, step 5.17e10 against 5.22e10 -- no gain: the memory waves' vector instructions were not what the roles cost each other.  Kept
        // behind the macro, off.
        constexpr bool BUF = NNS_SPLIT_BUF && !SEG;
        auto rsrc = [&](const float* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)((unsigned)N * (unsigned)ny * 4u), 0x00020000); };
        auto row_soff = [&](int i) -> int { return (4 * MROWS * (i >> 2) + (i & 3)) * ny * 4; };      // wave-uniform: scalar registers
        auto load_tile = [&](long tt) {
            int j0; size_t g;
            tile_coords(tt, j0, g);
            unsigned col = (unsigned)(j0 + cc < ny ? j0 + cc : ny - 1);       // clamped column: no mask needed on a load
            unsigned crv = (unsigned)cr;
            asm volatile("" : "+v"(col), "+v"(crv));
            const float* ug = u + g; const float* vg = v + g; const float* pg = p + g;
            if constexpr (BUF) {
                const auto r0 = rsrc(ug), r1 = rsrc(vg), r2 = rsrc(pg);
                const int voff = (int)((col + 4u * crv * (unsigned)ny) * 4u);
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int so = row_soff(i);
                    R[0][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, voff, so, 0));
                    R[1][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, voff, so, 0));
                    R[2][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, voff, so, 0));
                }
            } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const unsigned c = off32(i, col, crv);
                R[0][i] = at(ug, c); R[1][i] = at(vg, c); R[2][i] = at(pg, c);
            }
            }
        };
        auto store_tile = [&](long tt) {
            int j0; size_t g;
            tile_coords(tt, j0, g);
            if (j0 + cc < ny) {
                unsigned col = (unsigned)(j0 + cc);
                unsigned crv = (unsigned)cr;
                asm volatile("" : "+v"(col), "+v"(crv));
                float* ug = ru + g; float* vg = rv + g; float* pg = rd + g;
                if constexpr (BUF) {
                    const auto r0 = rsrc(ug), r1 = rsrc(vg), r2 = rsrc(pg);
                    const int voff = (int)((col + 4u * crv * (unsigned)ny) * 4u);
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const int so = row_soff(i);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, R[0][i]), r0, voff, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, R[1][i]), r1, voff, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, R[2][i]), r2, voff, so, 0);
                    }
                } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const unsigned c = off32(i, col, crv);
                    at_w(ug, c) = R[0][i]; at_w(vg, c) = R[1][i]; at_w(pg, c) = R[2][i];
                }
                }
            }
        };
        load_tile(t);
#pragma unroll
        for (int q = 0; q < NR / 4; ++q)
#pragma unroll
            for (int f = 0; f < 3; ++f)
                *reinterpret_cast<float4*>(stage + f * SF + 4 * MROWS * q) = make_float4(R[f][4 * q], R[f][4 * q + 1], R[f][4 * q + 2], R[f][4 * q + 3]);
        __syncthreads();                                                        // inputs of the first tile are staged
        long prev = -1;
        for (; t < ntiles; t += gridDim.x) {
            const long tn = t + gridDim.x;
            const bool has_next = tn < ntiles;
            if (NNS_SPLIT_EXP != 2) {
#ifndef NNS_SPLIT_NOST
#define NNS_SPLIT_NOST 0           // timing experiments (wrong results): no partial stores / no input loads after the first tile
#endif
#ifndef NNS_SPLIT_NOLD
#define NNS_SPLIT_NOLD 0
#endif
            if (prev >= 0 && !NNS_SPLIT_NOST) store_tile(prev);                 // under the transforms of tile t: tile t-1's partials out ...
            if (has_next && !NNS_SPLIT_NOLD) load_tile(tn);                     // ... and tile t+1's inputs in (same registers)
            }
            __syncthreads();                                                    // (1) the transform waves have written tile t's partials to the staging image
            // the exchange step: results out of the image, next inputs into it, slot for slot
#pragma unroll
            for (int q = 0; q < NR / 4; ++q) {
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    float4* slot = reinterpret_cast<float4*>(stage + f * SF + 4 * MROWS * q);
                    const float4 out = *slot;
                    if (has_next) *slot = make_float4(R[f][4 * q], R[f][4 * q + 1], R[f][4 * q + 2], R[f][4 * q + 3]);
                    R[f][4 * q] = out.x; R[f][4 * q + 1] = out.y; R[f][4 * q + 2] = out.z; R[f][4 * q + 3] = out.w;
                }
                if (q & 1) __builtin_amdgcn_sched_barrier(0);                  // two row groups in flight at a time: all 24 reads hoisted above the writes would double the live registers
            }
            __syncthreads();                                                    // (2) tile t+1's inputs are staged
            prev = t;
        }
        store_tile(prev);
        return;
    }
    // ================= transform waves =================
    // (round 3, rejected by same-box A/B, profiles/r03_ab_packed_stagger_prio.log: waves 4..7 delayed by s_sleep 4 / 8 at every tile start so that
    //  the two transform waves of a SIMD run out of step: 0.520 -> 0.535 ms; static s_setprio 1 for waves 4..7: 0.520 -> 0.535 ms)
    __syncthreads();                                                            // inputs of the first tile are staged
    for (; t < ntiles; t += gridDim.x) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave;
        const int sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * SL::LINE_BYTES;
        float* my_stage = reinterpret_cast<float*>(xb) + (line % 8) * SL::SKEW_DW;
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        float uf[16], vf[16], pf[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            uf[m] = my_stage[0 * SF + tidv + TPF * m]; vf[m] = my_stage[1 * SF + tidv + TPF * m]; pf[m] = my_stage[2 * SF + tidv + TPF * m];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        C2<float> a[16], b2[16];
        if (NNS_SPLIT_EXP == 1) {
#pragma unroll
            for (int m = 0; m < 16; ++m) { a[m].x = vf[m]; a[m].y = pf[m]; b2[m].x = uf[m]; b2[m].y = vf[m]; }
        } else
        deriv_core<N, TF, true>(uf, vf, pf, a, b2, tabF, tabI, xb, tidv, k);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            my_stage[0 * SF + tidv + TPF * m] = uf[m] * a[m].x + b2[m].x;     // P_u = u u_x + p_x/rho - nu u_xx
            my_stage[1 * SF + tidv + TPF * m] = uf[m] * a[m].y + b2[m].y;     // P_v = u v_x - nu v_xx
            my_stage[2 * SF + tidv + TPF * m] = a[m].x;                       // P_d = u_x
        }
        __syncthreads();                                                        // (1)
        __syncthreads();                                                        // (2)
    }
}

// Tile order: the x-pass walks the grids from the LAST to the first and the y-pass from the first to the last, so the
// y-pass starts on the partials (and inputs) the x-pass touched last -- part of them is still in the 256 MB Infinity
// Cache -- and an x-pass that follows a forward-streaming kernel over the same inputs (the FD residual in bench.py)
// starts on what that kernel read last.  Same-box A/B at 1024^2 x 64: y-pass 0.619 -> 0.589 ms, x-pass 0.614 -> 0.606.
template <int N, typename TF, bool SEG = false>
int launch_xpass(const float* u, const float* v, const float* p, float* ru, float* rv, float* rd, int batch, int ny, const SpecK& k, hipStream_t s,
                 const SegK& sg = SegK{}) {
    using L = SpecLds<N, TF>;
    const int tiles_per_grid = (ny + L::LINES - 1) / L::LINES;
    const long ntiles = (long)batch * tiles_per_grid;
    // prefetch is disabled where hipcc (ROCm 7.2) spills with it: checked with -Rpass-analysis=kernel-resource-usage
#ifndef NNS_XPASS_SPLIT
#define NNS_XPASS_SPLIT 1          // 1: spec_xpass_split_kernel (8 transform waves + 4 memory waves), 0: spec_xpass_kernel
#endif
    // the role-split kernel's memory waves address with a scalar grid base + a 32-BIT BYTE offset per lane: the largest element offset inside
    // one grid (SEG: across all source-rank segments) must stay below 2^30; beyond that the older kernel (size_t row offsets) takes over
    const unsigned long long max_off = SEG ? (unsigned long long)((N >> sg.shift) - 1) * (unsigned long long)sg.stride + ((unsigned long long)ny << sg.shift)
                                           : (unsigned long long)N * (unsigned long long)ny;
    const bool off32_ok = max_off < (1ull << 30);
    if (NNS_XPASS_SPLIT && off32_ok) {
        auto kern = spec_xpass_split_kernel<N, TF, SEG>;
        using SL = SplitLds<N, TF>;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SL::TOTAL);
            if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spec xpass: hipFuncSetAttribute(%d B): %s", SL::TOTAL, hipGetErrorString(e));
            attr_set = true;
        }
        const long gmax = spec_grid_cap();
        const unsigned grid = (unsigned)(ntiles < gmax ? ntiles : gmax);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kSplitThreads), SL::TOTAL, s, u, v, p, ru, rv, rd, ny, tiles_per_grid, ntiles, k, sg);
        return check_launch("spec_residual_xpass");
    }
    constexpr bool PF = !((N == 128 && sizeof(TF) == 4) || N == 256);
    auto kern = spec_xpass_kernel<N, TF, PF, SEG>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spec xpass: hipFuncSetAttribute(%d B): %s", L::TOTAL, hipGetErrorString(e));
        attr_set = true;
    }
    const long gmax = spec_grid_cap();
    const unsigned grid = (unsigned)(ntiles < gmax ? ntiles : gmax);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kSpecThreads), L::TOTAL, s, u, v, p, ru, rv, rd, ny, tiles_per_grid, ntiles, k, sg);
    return check_launch("spec_residual_xpass");
}

template <int N, typename TF, bool FUSE_FD = false, bool SEGP = false>
int launch_ypass(const float* u, const float* v, const float* p, const float* up, const float* vp, float* ru, float* rv, float* rd,
                 long nrows, const SpecK& k, hipStream_t s, float* fu = nullptr, float* fv = nullptr, float* fd = nullptr, int nx = 1,
                 const FdK& fk = FdK{}, const HaloK& hk = HaloK{}, const PartK& pk = PartK{}) {
    using L = SpecLds<N, TF>;
    auto kern = spec_ypass_kernel<N, TF, FUSE_FD, SEGP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spec ypass: hipFuncSetAttribute(%d B): %s", L::TOTAL, hipGetErrorString(e));
        attr_set = true;
    }
    const long niter = (nrows + L::LINES - 1) / L::LINES;
    const long gmax = spec_grid_cap();
    const unsigned grid = (unsigned)(niter < gmax ? niter : gmax);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kSpecThreads), L::TOTAL, s, u, v, p, up, vp, ru, rv, rd, fu, fv, fd, nx, fk, nrows, k, hk, pk);
    return check_launch("spec_residual_ypass");
}

inline int device_cus() {
    static const int n = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256; return v; }();
    return n;
}
#ifndef NNS_ROWMARCH
#define NNS_ROWMARCH 1             // 1: the all-float32 fused row pass marches (spec_rowmarch_kernel); 0: spec_ypass_kernel<N, float, true>
#endif
// rows per chunk of the marching row pass: one chunk per line once every CU is busy, a power of two in [1, 64].  Small batches get
// short chunks (R = 1: three row reads per row, as the non-marching kernel, but those fit the caches); the SAME kernel serves every
// batch size, so grid b of a large batch equals the same grid evaluated alone bit for bit.
template <int N>
int march_chunk_rows(long nrows, int nx) {
    static const int forced = [] { const char* e = getenv("NNS_MARCH_R"); return e ? atoi(e) : 0; }();      // tuning override (a power of two)
    if (forced > 0) return forced > nx ? nx : forced;
    const long tl = (long)device_cus() * SpecLds<N, float>::LINES;
    const long r = nrows / tl;
    int R = 1;
    while (R * 2 <= r && R < 64) R *= 2;
    while (R > nx) R /= 2;
    return R;
}
template <int N, bool SEGP = false>
int launch_rowmarch(const float* u, const float* v, const float* p, const float* up, const float* vp, float* ru, float* rv, float* rd,
                    float* fu, float* fv, float* fd, int batch, int nx, int R, const SpecK& k, const FdK& fk, const HaloK& hk, hipStream_t s,
                    const PartK& pk = PartK{}) {
    using ML = MarchLds<N>;
    auto kern = spec_rowmarch_kernel<N, SEGP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ML::TOTAL);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spec rowmarch: hipFuncSetAttribute(%d B): %s", ML::TOTAL, hipGetErrorString(e));
        attr_set = true;
    }
    const int chunks_per_grid = (nx + R - 1) / R;
    const long nchunks = (long)batch * chunks_per_grid;
    const long wgs = (nchunks + ML::L::LINES - 1) / ML::L::LINES;
    const unsigned grid = (unsigned)(wgs < device_cus() ? wgs : device_cus());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kSpecThreads), ML::TOTAL, s, u, v, p, up, vp, ru, rv, rd, fu, fv, fd, nx, fk, k, hk, R, chunks_per_grid, nchunks, pk);
    return check_launch("residual_both_rowpass");
}

}  // namespace
}  // namespace spec
}  // namespace nns
