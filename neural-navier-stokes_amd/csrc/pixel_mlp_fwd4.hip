// Per-pixel MLP forward, width 33..64, bf16 operands: the one-wave-per-SIMD, four-tile kernel (reference: BasisFunc,
// src/neural_spectral/spectral_ode.py:100-119, generalised as in pixel_mlp_kernels.hip).  A translation unit of its own: it is compiled with
// -mllvm -amdgpu-mfma-vgpr-form=1.  Left to its heuristics hipcc (ROCm 7.2) keeps the 128 accumulator registers of the four tiles in AGPRs, which the
// vector pipe cannot read: every conversion then pays a v_accvgpr_read per value and every bias load a v_accvgpr_write (192 extra vector
// instructions per layer, checked in the ISA) -- more than the 32-cycle MFMA gaps can hide.  With the VGPR form the layer body is the intended
// stream: MFMA, four conversion instructions, MFMA, ...
#include "pixel_mlp_common.h"

using namespace nns;
using namespace nns::pm;

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Forward, width 33..64, bf16, ONE wave per SIMD with FOUR pixel tiles (round 3).  The two-tile kernel above overlaps MFMAs and conversions
// inside a wave, but two such waves per SIMD still leave the matrix pipe ~60 % idle: they contend for it and for the LDS, whose fragment
// traffic (one 1-KB read per two MFMAs and wave, 128 B/clk per CU) sits at half the LDS peak.  With 512 registers a single wave holds four
// tiles: every weight fragment feeds FOUR MFMAs (half the LDS traffic per MFMA), tiles {0, 1} run their MFMAs while tiles {2, 3} convert and
// vice versa (four vector instructions per 32-cycle MFMA gap: inside the issue budget of one wave), nothing else competes for the SIMD, and the
// next 128 pixels' inputs are prefetched into registers under the layers.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kP4Threads = 256;
__global__ __launch_bounds__(kP4Threads) __attribute__((amdgpu_waves_per_eu(1, 1))) void pixel_mlp_fwd_pipe4_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ Bv,
                                                                        float* __restrict__ y, long npix_total, int P, PixelMlpDesc d, unsigned pmagic, int pshift) {
    constexpr int OT = 2, SS = 4, NF = OT * SS, NT = 4;
    using U = UniLds<OT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    stage_uniform<OT>(d, W, Bv, lds, threadIdx.x, kP4Threads);
    __syncthreads();
#ifndef NNS_P4_EXP
#define NNS_P4_EXP 0               // timing probes (wrong results): 1 = weights staged, nothing else; 2 = no layers (inputs in, accumulators = bias out); 3 = first and last layer only
#endif
    if (NNS_P4_EXP == 1) return;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, r = lane & 31, h = lane >> 5;
    const int nl = d.nlayers, cin0 = d.cin[0], coutL = d.cout[nl - 1];       // cin0, coutL <= 4 (checked on the host)
    const long ngroups = (npix_total + 32 * NT - 1) / (32 * NT);
    const long gstride = (long)gridDim.x * (kP4Threads / kWave);
    const unsigned char* bias0 = lds + nl * U::W_BYTES;
    const bf16x8* wl0 = reinterpret_cast<const bf16x8*>(lds) + lane;
    bf16x8 w[NF];
#pragma unroll
    for (int idx = 0; idx < NF; ++idx) w[idx] = wl0[idx * 64];
    // pixel index -> (image, pixel in image): division by the runtime P with a host-made multiplier (exact for every 32-bit index; the 64-bit
    // `gp / P`, `gp % P` of the first version were ~150 emulated-division instructions per tile, eight times per group of 128 pixels)
    auto split = [&](long gp, long& img, long& pix) {
        if (pshift >= 0) {
            const unsigned n = (unsigned)gp, t = __umulhi(pmagic, n);
            const unsigned q = (t + ((n - t) >> 1)) >> pshift;
            img = q; pix = n - q * (unsigned)P;
        } else { img = gp / P; pix = gp % P; }
    };
    // raw inputs of a group: channel j < cin0 of this lane's pixel of every tile (unconditional, clamped loads)
    float xin[NT][4];
    auto load_raw = [&](long g) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long gp = (g * NT + t) * 32 + r;
            const long gc = gp < npix_total ? gp : npix_total - 1;
            long bimg, pimg;
            split(gc, bimg, pimg);
            const float* xb = x + (size_t)bimg * cin0 * P + pimg;
#pragma unroll
            for (int j = 0; j < 4; ++j) xin[t][j] = xb[(size_t)(j < cin0 ? j : 0) * P];
        }
    };
    long g = (long)blockIdx.x * (kP4Threads / kWave) + wave;
    if (g < ngroups) load_raw(g);
    for (; g < ngroups; g += gstride) {
        bf16x8 fr[NT][SS];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool ok = (g * NT + t) * 32 + r < npix_total;
#pragma unroll
            for (int s2 = 0; s2 < SS; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) fr[t][s2][j] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) fr[t][0][j] = (short)f2bf((ok && h == 0 && j < cin0) ? xin[t][j] : 0.f);      // channels 0..3: elements 0..3 of fragment 0, lane half 0
        }
        { const long gn = g + gstride; if (gn < ngroups) load_raw(gn); }     // the next group's inputs travel under this group's layers
        f32x16 acc[NT][OT];
        auto load_bias = [&](int t, int ot, int l) {
            const float* bl = reinterpret_cast<const float*>(bias0 + l * U::B_BYTES);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][ot][i] = bl[32 * ot + acc_row(i, h)];
        };
        // conversion unit u (0 .. 15) of tile pair q: tile 2 q + (u >> 3), four accumulator values -> two packed registers of its fragment (u & 7) >> 1, ReLU
        auto conv_unit = [&](auto qc, auto uc) {
            constexpr int t = 2 * decltype(qc)::value + (decltype(uc)::value >> 3), u = decltype(uc)::value & 7;
            constexpr int sfr = u >> 1, half = u & 1, b0 = 8 * (sfr & 1) + 4 * half;
            i32x4v f = __builtin_bit_cast(i32x4v, fr[t][sfr]);
            const f32x16& a = acc[t][sfr >> 1];
            f[2 * half] = pack2<true>(a[b0], a[b0 + 1]);
            f[2 * half + 1] = pack2<true>(a[b0 + 2], a[b0 + 3]);
            fr[t][sfr] = __builtin_bit_cast(bf16x8, f);
        };
#pragma unroll
        for (int t = 0; t < NT; ++t) { load_bias(t, 0, 0); load_bias(t, 1, 0); }
        using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
        auto layer = [&](int l, auto firstc, auto lastc) {
            constexpr bool FIRST = decltype(firstc)::value, LAST = decltype(lastc)::value;
            const bf16x8* wn = LAST ? wl0 : wl0 + (size_t)(l + 1) * (NF * 64);
            // ---- tiles 0, 1: MFMAs of layer l; tiles 2, 3: conversion of their layer l-1 result, then this layer's bias into their accumulators
            // (edge layers: the input has <= 4 channels, so the first layer's operand fragments 1 .. 3 are zero -- only k-step 0 is multiplied;
            //  the output has <= 4 channels, so the last layer needs output tile 0 only)
            static_for<0, NF>([&](auto ic) {
                constexpr int idx = decltype(ic)::value, ot = idx / SS, s2 = idx % SS;
                constexpr bool DO = !(FIRST && s2 != 0) && !(LAST && ot != 0);
                if constexpr (DO) acc[0][ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[idx], fr[0][s2], acc[0][ot], 0, 0, 0);
                if constexpr (!FIRST) conv_unit(Q1{}, std::integral_constant<int, 2 * idx>{});
                if constexpr (DO) acc[1][ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[idx], fr[1][s2], acc[1][ot], 0, 0, 0);
                if constexpr (!FIRST) {
                    conv_unit(Q1{}, std::integral_constant<int, 2 * idx + 1>{});
                    if constexpr (idx == 1) load_bias(2, 0, l);              // units 0 .. 3 = acc[2][0], 4 .. 7 = acc[2][1], 8 .. 11 = acc[3][0], 12 .. 15 = acc[3][1]
                    if constexpr (idx == 3) load_bias(2, 1, l);
                    if constexpr (idx == 5) load_bias(3, 0, l);
                    if constexpr (idx == 7) load_bias(3, 1, l);
                }
                // pins: MFMA, 4 conversion instructions, MFMA, 4 conversion instructions, and -- where a bias load was just requested -- its four
                // LDS reads HERE (left free, the scheduler sinks them to their first use in the next phase and the MFMA there waits for lgkmcnt(0))
                __builtin_amdgcn_sched_group_barrier(0x008, DO ? 1 : 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, FIRST ? 0 : 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, DO ? 1 : 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, FIRST ? 0 : 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, (!FIRST && (idx & 1)) ? 4 : 0, 0);
            });
            // ---- tiles 2, 3: MFMAs of layer l (same fragments, then refilled in place with the next layer's); tiles 0, 1: conversion + next bias
            static_for<0, NF>([&](auto ic) {
                constexpr int idx = decltype(ic)::value, ot = idx / SS, s2 = idx % SS;
                constexpr bool DO = !(FIRST && s2 != 0) && !(LAST && ot != 0);
                if constexpr (DO) acc[2][ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[idx], fr[2][s2], acc[2][ot], 0, 0, 0);
                if constexpr (!LAST) conv_unit(Q0{}, std::integral_constant<int, 2 * idx>{});
                if constexpr (DO) acc[3][ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[idx], fr[3][s2], acc[3][ot], 0, 0, 0);
                w[idx] = wn[idx * 64];
                if constexpr (!LAST) {
                    conv_unit(Q0{}, std::integral_constant<int, 2 * idx + 1>{});
                    if constexpr (idx == 1) load_bias(0, 0, l + 1);
                    if constexpr (idx == 3) load_bias(0, 1, l + 1);
                    if constexpr (idx == 5) load_bias(1, 0, l + 1);
                    if constexpr (idx == 7) load_bias(1, 1, l + 1);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, DO ? 1 : 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, LAST ? 0 : 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, DO ? 1 : 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1 + ((!LAST && (idx & 1)) ? 4 : 0), 0);
                __builtin_amdgcn_sched_group_barrier(0x002, LAST ? 0 : 4, 0);
            });
        };
        using T_ = std::true_type; using F_ = std::false_type;
        if (NNS_P4_EXP == 2) {}
        else if (nl == 1) layer(0, T_{}, T_{});
        else {
            layer(0, T_{}, F_{});
            if (NNS_P4_EXP != 3) for (int l = 1; l + 1 < nl; ++l) layer(l, F_{}, F_{});
            layer(nl - 1, F_{}, T_{});
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long gp = (g * NT + t) * 32 + r;
            if (gp < npix_total) {
                long bimg, pimg;
                split(gp, bimg, pimg);
                store_acc<OT, true>(y + (size_t)bimg * coutL * P + pimg, (size_t)P, coutL, h, acc[t]);
            }
        }
    }
}

}  // namespace

int nns::pm::launch_fwd_pipe4(const float* x, const float* weights, const float* biases, float* y, long npix, int P, const PixelMlpDesc& d, hipStream_t s) {
    const int lds = UniLds<2>::total(d.nlayers);
    if (lds > 160 * 1024) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_fwd: weights need %d B of LDS (> 160 KiB)", lds);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_mlp_fwd_pipe4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "pixel_mlp_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const long ngroups = (npix + 127) / 128;
    int cus = 256;
    { int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    long blocks = (ngroups + 3) / 4; if (blocks > cus) blocks = cus;       // persistent: one workgroup (four waves, one per SIMD) per CU
    // n / P for 32-bit n:  t = umulhi(m, n);  q = (t + ((n - t) >> 1)) >> (l - 1)  with l = ceil(log2 P), m = floor(2^32 (2^l - P) / P) + 1
    unsigned pmagic = 0; int pshift = -1;
    if (npix < (1L << 32) && P >= 2) {
        int l = 0; while ((1L << l) < P) ++l;
        pmagic = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << l) - (unsigned long long)P)) / (unsigned long long)P + 1);
        pshift = l - 1;
    }
    hipLaunchKernelGGL(pixel_mlp_fwd_pipe4_kernel, dim3((unsigned)blocks), dim3(kP4Threads), lds, s, x, weights, biases, y, npix, P, d, pmagic, pshift);
    return check_launch("pixel_mlp_fwd");
}

