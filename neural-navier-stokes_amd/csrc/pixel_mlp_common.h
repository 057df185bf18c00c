// Pieces of the per-pixel MLP kernels shared by pixel_mlp_kernels.hip and pixel_mlp_fwd4.hip (the four-tile forward lives in a translation unit
// of its own because it is compiled with -mllvm -amdgpu-mfma-vgpr-form=1, see csrc/Makefile).
#pragma once
#include "nns_common.h"

namespace nns {
namespace pm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;      // 8 bf16 in 4 VGPRs
using bf16x4 = __attribute__((ext_vector_type(4))) short;

constexpr int kMaxLayers = 8;
constexpr int kMaxWidth = 64;

struct PixelMlpDesc {
    int nlayers;
    int cin[kMaxLayers], cout[kMaxLayers];
    int woff[kMaxLayers], boff[kMaxLayers];      // offsets (floats) into the packed weight / bias arrays
    int lds_off[kMaxLayers];                     // offset (bytes) of the layer's pre-permuted fragments in LDS
    int lds_bias[kMaxLayers];                    // offset (bytes) of the layer's padded bias
};

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }     // C/D map, 32x32

__device__ __forceinline__ unsigned short f2bf(float x) {          // round-to-nearest-even; NaN-safe via the plain cast
    return __builtin_bit_cast(unsigned short, (__bf16)x);
}
// Accumulator registers -> packed bf16 operand elements, TWO values per v_cvt_pk_bf16_f32 (round 3: converting element by element and
// assembling short vectors compiled to one single-value convert per element plus a v_perm_b32 per pair -- 128 instead of 64 vector
// instructions per 64 x 64 layer and tile pair, checked in the ISA).  RELU: packed int16 maximum with 0 (bf16 is sign-magnitude).
using f32x2v = __attribute__((ext_vector_type(2))) float;
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using s16x2v = __attribute__((ext_vector_type(2))) short;
using i32x4v = __attribute__((ext_vector_type(4))) int;
template <bool RELU>
__device__ __forceinline__ int pack2(float a, float b) {
    s16x2v q = __builtin_bit_cast(s16x2v, __builtin_convertvector((f32x2v){a, b}, bf16x2v));
    if constexpr (RELU) q = __builtin_elementwise_max(q, (s16x2v){0, 0});
    return __builtin_bit_cast(int, q);
}
// ReLU' mask of the backward chain: keep the halves of `v` (two packed bf16) whose activation in `act` (two packed bf16, >= 0) is non-zero.
// Written as (0 - act) >> 15 on packed int16 (v_pk_sub_i16, v_pk_ashrrev_i16, v_and: three instructions per pair); the min / negate
// and multiply forms are "recognised" by the compiler as selects and come out as two compares, two conditional moves and a v_perm per pair.
__device__ __forceinline__ int mask2(int v, int act) {
    const s16x2v m = ((s16x2v){0, 0} - __builtin_bit_cast(s16x2v, act)) >> (s16x2v){15, 15};
    return v & __builtin_bit_cast(int, m);
}
template <bool RELU_UNUSED = false>
__device__ __forceinline__ bf16x8 pack8_masked(const f32x16& a, int base, bf16x8 act) {
    const i32x4v m = __builtin_bit_cast(i32x4v, act);
    const i32x4v r = {mask2(pack2<false>(a[base], a[base + 1]), m[0]), mask2(pack2<false>(a[base + 2], a[base + 3]), m[1]),
                      mask2(pack2<false>(a[base + 4], a[base + 5]), m[2]), mask2(pack2<false>(a[base + 6], a[base + 7]), m[3])};
    return __builtin_bit_cast(bf16x8, r);
}
// fragment s of the next layer (channels 16 s .. 16 s + 15 in operand order) from the accumulator tile that holds them: registers base .. base + 7
template <bool RELU>
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int base) {
    const i32x4v r = {pack2<RELU>(a[base], a[base + 1]), pack2<RELU>(a[base + 2], a[base + 3]), pack2<RELU>(a[base + 4], a[base + 5]), pack2<RELU>(a[base + 6], a[base + 7])};
    return __builtin_bit_cast(bf16x8, r);
}


template <int OT>
struct UniLds {
    static constexpr int SS = 2 * OT;
    static constexpr int W_BYTES = OT * SS * 64 * 16;            // fragments of one layer
    static constexpr int B_BYTES = OT * 32 * 4;
    __host__ __device__ static int total(int nlayers) { return nlayers * (W_BYTES + B_BYTES); }
};

// Staging walks the REAL [out][in] matrices with consecutive threads on consecutive input channels (coalesced reads, no
// div/mod per element) and scatters into the zero-filled fragment image.  (Walking the image and gathering from global
// memory instead cost ~100 us per workgroup -- more than the whole tile loop at depth 8 / width 64.)
template <int OT>
__device__ inline void stage_uniform(const PixelMlpDesc& d, const float* __restrict__ W, const float* __restrict__ B, unsigned char* lds, int tid, int nthreads) {
    using U = UniLds<OT>;
    constexpr int SS = U::SS;
    const int total = U::total(d.nlayers);
    for (int e = tid; e < total / 16; e += nthreads) reinterpret_cast<uint4*>(lds)[e] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int l = 0; l < d.nlayers; ++l) {
        const int cin = d.cin[l], cout = d.cout[l], n = cin * cout;
        const float* Wl = W + d.woff[l];
        unsigned short* dst = reinterpret_cast<unsigned short*>(lds + l * U::W_BYTES);
        // UNR elements per thread and round, all their global reads in flight before the first LDS write (round 3: one read per loop
        // iteration, each waited for before its 2-byte LDS write, made the staging of a depth-8 width-64 stack 128 dependent L2 round trips:
        // 32 us of a 280 us kernel, tools/pm_kernel_times.sh).
        constexpr int UNR = 8;            // (32 per round measured the same)
        const unsigned magic = ((1u << 20) + (unsigned)cin - 1u) / (unsigned)cin;        // e / cin = (e * magic) >> 20, exact for e < 64 cin (cin <= 64)
        for (int e0 = tid; e0 < n; e0 += UNR * nthreads) {
            float v[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) { const int e = e0 + q * nthreads; v[q] = Wl[e < n ? e : n - 1]; }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int e = e0 + q * nthreads;
                if (e < n) {
                    const int row = (int)(((unsigned)e * magic) >> 20), k = e - row * cin;
                    const int ot = row >> 5, r = row & 31, s2 = k >> 4, kk = k & 15;
                    const int lane = r + 32 * ((kk >> 2) & 1), j = 4 * (kk >> 3) + (kk & 3);
                    dst[(((ot * SS + s2) * 64 + lane) << 3) + j] = f2bf(v[q]);
                }
            }
        }
        float* bl = reinterpret_cast<float*>(lds + d.nlayers * U::W_BYTES + l * U::B_BYTES);
        for (int e = tid; e < cout; e += nthreads) bl[e] = B[d.boff[l] + e];
    }
}

// Tile I/O without per-site branches.  Fragment element (s, j) of lane half h holds channel 16 s + 8 (j>>2) + 4 h + (j&3)
// and accumulator register i of tile t holds channel 32 t + (i&3) + 8 (i>>2) + 4 h: both increase with the site index,
// so the sites are visited in channel order and the walk stops (uniformly) at the first site beyond the channel count --
// 3 channels touch 3 sites, not 64.  Loads are unconditional (pixel and channel clamped into range) and zeroed by a
// select; stores are masked per lane.
template <int SS, bool SMALL>
__device__ __forceinline__ void load_frags(const float* __restrict__ xb, size_t P, int cin0, bool ok, int h, bf16x8 (&f)[SS]) {
#pragma unroll
    for (int s = 0; s < SS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) f[s][j] = 0;
    if constexpr (SMALL) {      // cin0 <= 4, the usual (u, v, p) input: channels 0..3 sit in elements 0..3 of fragment 0, lane half 0 only
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = xb[(size_t)(j < cin0 ? j : 0) * P];
            f[0][j] = (short)f2bf((ok && h == 0 && j < cin0) ? v : 0.f);
        }
        return;
    } else {
#pragma unroll
    for (int s = 0; s < SS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cmin = 16 * s + 8 * (j >> 2) + (j & 3);
            if (cmin >= cin0) return;
            const int c = cmin + 4 * h;
            const float v = xb[(size_t)(c < cin0 ? c : cin0 - 1) * P];
            f[s][j] = (short)f2bf((ok && c < cin0) ? v : 0.f);
        }
    }
}

template <int OT, bool SMALL>
__device__ __forceinline__ void store_acc(float* __restrict__ yb, size_t P, int cout, int h, const f32x16 (&a)[OT]) {
    if constexpr (SMALL) {            // channels 0..3 = registers 0..3 of tile 0, lane half 0
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < cout) yb[(size_t)i * P] = a[0][i];
        }
        return;
    } else {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rmin = 32 * t + (i & 3) + 8 * (i >> 2);
            if (rmin >= cout) return;
            const int c = rmin + 4 * h;
            if (c < cout) yb[(size_t)c * P] = a[t][i];
        }
    }
}


// pixel_mlp_fwd4.hip
int launch_fwd_pipe4(const float* x, const float* weights, const float* biases, float* y, long npix, int P, const PixelMlpDesc& d, hipStream_t s);

}  // namespace pm
}  // namespace nns
