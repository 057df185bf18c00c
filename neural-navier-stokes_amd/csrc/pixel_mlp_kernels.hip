// Per-pixel MLP (1x1-convolution stack) forward on gfx950: the reference's BasisFunc
// (src/neural_spectral/spectral_ode.py:100-119: Conv2d(3,16,1)-ReLU-Conv2d(16,32,1)-ReLU-...-Conv2d(16,3,1)),
// generalised to any depth <= 8 and width <= 64 (BASELINE configs 2 and 3: depth-4 width-32 float32, depth-8
// width-64 bfloat16 weights/activations with float32 accumulation).
//
// GEMM-shaped (M = pixels, N = K = channels) -> matrix cores.  The layers are CHAINED IN REGISTERS:
// a wave owns a tile of 32 pixels and computes the transposed product  Y^T[out x pix] = W[out x in] X^T[in x pix]
// with A = W (from LDS) and B = X^T.  The 32x32 accumulator has the pixel on the lane and the channels in its 16
// registers, which is exactly what the NEXT layer's B operand needs ("an accumulator tile as the next MFMA's
// operand", cdna_hip_programming.md section 3) -- provided the weights' k order is permuted to the accumulator's
// register order.  That permutation is applied once, when the weights are staged into LDS, so no activation
// ever goes through LDS or HBM between layers: HBM traffic is C_in + C_out floats per pixel.
//   float32 : v_mfma_f32_32x32x2_f32   (exact fp32 fma chain; k-step i uses channels {c(i), c(i)+4},
//             c(i) = (i&3) + 8*(i>>2): the two rows register i holds in lane halves 0 / 1)
//   bfloat16: v_mfma_f32_32x32x16_bf16 (k-step s uses registers 8s..8s+7 packed pairwise to bf16; element j of
//             lane half h is channel 16s + 8*(j>>2) + 4h + (j&3))
#include "pixel_mlp_common.h"

using namespace nns;
using namespace nns::pm;

namespace {

// LDS image of a layer's weights, one entry per (out tile ot, k block, lane):
//   float32 : [ot][kb = in/32][i = 0..15][lane 64]  float   = W[32 ot + (lane&31)][32 kb + c(i) + 4 (lane>>5)]
//   bfloat16: [ot][s  = in/16][lane 64][8]          bf16    = W[32 ot + (lane&31)][16 s + 8 (j>>2) + 4 (lane>>5) + (j&3)]
template <bool BF16>
__device__ void stage_weights(const PixelMlpDesc& d, const float* __restrict__ W, const float* __restrict__ B, unsigned char* lds, int tid, int nthreads) {
    for (int l = 0; l < d.nlayers; ++l) {
        const int cin = d.cin[l], cout = d.cout[l];
        const int ots = (cout + 31) / 32;
        const float* Wl = W + d.woff[l];
        if constexpr (BF16) {
            const int ss = (cin + 15) / 16;
            unsigned short* dst = reinterpret_cast<unsigned short*>(lds + d.lds_off[l]);
            for (int e = tid; e < ots * ss * 64 * 8; e += nthreads) {
                const int j = e & 7, lane = (e >> 3) & 63, s = (e >> 9) % ss, ot = (e >> 9) / ss;
                const int row = 32 * ot + (lane & 31), k = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
                dst[e] = f2bf((row < cout && k < cin) ? Wl[row * cin + k] : 0.f);
            }
        } else {
            const int kbs = (cin + 31) / 32;
            float* dst = reinterpret_cast<float*>(lds + d.lds_off[l]);
            for (int e = tid; e < ots * kbs * 16 * 64; e += nthreads) {
                const int lane = e & 63, i = (e >> 6) & 15, kb = (e >> 10) % kbs, ot = (e >> 10) / kbs;
                const int row = 32 * ot + (lane & 31), k = 32 * kb + acc_row(i, lane >> 5);
                dst[e] = (row < cout && k < cin) ? Wl[row * cin + k] : 0.f;
            }
        }
        float* bl = reinterpret_cast<float*>(lds + d.lds_bias[l]);
        for (int e = tid; e < ots * 32; e += nthreads) bl[e] = e < cout ? B[d.boff[l] + e] : 0.f;
    }
}

// Forward launch geometry: kFwdThreads / 64 waves share one LDS copy of the weights.  With 50-60 KB of fragments
// (depth 8, width 64, bf16) two workgroups fit a CU, so 6 waves per workgroup give 3 waves per SIMD (<= 170 VGPRs)
// instead of 2; the next tile's input pixels are loaded while the current tile runs through the layers.
// Round 3, same-box A/B at depth 8 / width 64 / 16 x 512^2 (profiles/r03_ab_pixel_mlp_fwd.log): 256 threads 0.374 ms, 384 threads 0.43 ms, 512 threads
// 0.355 ms (one staging of the weights per eight waves); requesting the weight fragments 2 or 4 ahead instead of 1: no change (0.372 / 0.376) --
// the fragment latency is not what keeps the matrix pipe at 34 % busy.
#ifndef NNS_PM_THREADS
#define NNS_PM_THREADS 512
#endif
constexpr int kFwdThreads = NNS_PM_THREADS, kFwdWaves = kFwdThreads / 64;

#ifndef NNS_PM_GEN_THREADS
#define NNS_PM_GEN_THREADS 512
#endif
constexpr int kGenThreads = NNS_PM_GEN_THREADS, kGenWaves = kGenThreads / 64;      // the runtime-shaped (float32) kernel: its weights fill LDS, so ONE workgroup per CU -- eight waves of it

template <bool BF16>
__global__ __launch_bounds__(kGenThreads) void pixel_mlp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ Bv,
                                                                     float* __restrict__ y, long npix_total, int P, PixelMlpDesc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    stage_weights<BF16>(d, W, Bv, lds, threadIdx.x, kGenThreads);
    __syncthreads();
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, r = lane & 31, h = lane >> 5;
    const int cin0 = d.cin[0], coutL = d.cout[d.nlayers - 1];
    const long ntiles = (npix_total + 31) / 32;
    const long tstride = (long)gridDim.x * kGenWaves;
    // input tile: up to 64 channels x 32 pixels = two 32-row accumulator tiles; `nxt` is the prefetched next tile
    auto load_tile = [&](long tile, f32x16 (&a)[2]) {
        const long gp = tile * 32 + r;
        const bool ok = tile < ntiles && gp < npix_total;
        const long b = ok ? gp / P : 0, p = ok ? gp % P : 0;
        const float* xb = x + (size_t)b * cin0 * P + p;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int c = 32 * t + acc_row(i, h); a[t][i] = (ok && c < cin0) ? xb[(size_t)c * P] : 0.f; }
    };
    f32x16 nxt[2];
    long tile = (long)blockIdx.x * kGenWaves + wave;
    load_tile(tile, nxt);
    for (; tile < ntiles; tile += tstride) {
        const long gp = tile * 32 + r;
        const bool ok = gp < npix_total;
        const long b = ok ? gp / P : 0, p = ok ? gp % P : 0;
        f32x16 act[2];
        act[0] = nxt[0]; act[1] = nxt[1];
        load_tile(tile + tstride, nxt);
        if constexpr (BF16) {
            // The loop-carried state is the PACKED bf16 operand fragments, not float32 activations: after a layer's MFMAs
            // the accumulators are converted pairwise (v_cvt_pk_bf16_f32) and ReLU is a packed integer max with 0
            // (bf16 is sign-magnitude: max as int16 with 0 zeroes exactly the negative values) -- 32 VALU per layer
            // instead of ~150 (the kernel was VALU-bound: 1600 VALU per tile for 54 MFMAs).
            bf16x8 fr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) fr[s] = pack8<false>(act[s >> 1], 8 * (s & 1));
            const int nl = d.nlayers;
            for (int l = 0; l < nl; ++l) {
                const int cin = d.cin[l], cout = d.cout[l];
                const int ots = (cout + 31) / 32, ss = (cin + 15) / 16;
                const float* bl = reinterpret_cast<const float*>(lds + d.lds_bias[l]);
                const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds + d.lds_off[l]);
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    if (ot < ots) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) act[ot][i] = bl[32 * ot + acc_row(i, h)];       // bias (0 in the padding rows)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            if (s < ss) act[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[(ot * ss + s) * 64 + lane], fr[s], act[ot], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) act[ot][i] = 0.f;
                    }
                }
                if (l + 1 < nl) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) fr[s] = pack8<true>(act[s >> 1], 8 * (s & 1));
                }
            }
        } else {
        // float32 operands (v_mfma_f32_32x32x2_f32, 64 cycles each).  Round 3: a layer is a straight run of OTS * KBS * 16 MFMAs for its
        // compile-time tile counts (four forms, chosen per layer at run time), the two output tiles' accumulator chains ALTERNATE, and the
        // weight operand -- one dword per lane and MFMA -- comes through a ring of eight requested ahead.  Before, every MFMA sat behind a
        // run-time `if`, its own ds_read_b32 and an lgkmcnt(0), one accumulator chain at a time: the matrix pipe 57 % busy.
        auto layer_f32 = [&](auto otc, auto kbc, int l) {
            constexpr int OTS = decltype(otc)::value, KBS = decltype(kbc)::value, NST = OTS * KBS * 16, RD = 8;
            const float* bl = reinterpret_cast<const float*>(lds + d.lds_bias[l]);
            const float* wl = reinterpret_cast<const float*>(lds + d.lds_off[l]) + lane;
            f32x16 out[2];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int i = 0; i < 16; ++i) out[ot][i] = ot < OTS ? bl[32 * ot + acc_row(i, h)] : 0.f;    // bias as the initial accumulator
            // step n: k-block n / (16 OTS), k-pair (n / OTS) % 16, output tile n % OTS; its weights: [ot][kb][i][lane]
            auto widx = [](int n) constexpr { return (((n % OTS) * KBS + n / (16 * OTS)) * 16 + (n / OTS) % 16) * 64; };
            float ring[RD];
#pragma unroll
            for (int q = 0; q < RD; ++q) ring[q] = wl[widx(q)];
            static_for<0, NST>([&](auto nc) {
                constexpr int n = decltype(nc)::value, ot = n % OTS, kb = n / (16 * OTS), i = (n / OTS) % 16;
                out[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[n % RD], act[kb][i], out[ot], 0, 0, 0);
                if constexpr (n + RD < NST) ring[n % RD] = wl[widx(n + RD)];
            });
            const bool relu = l + 1 < d.nlayers;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) act[t][i] = relu ? fmaxf(out[t][i], 0.f) : out[t][i];
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        for (int l = 0; l < d.nlayers; ++l) {
            const bool o2 = d.cout[l] > 32, k2 = d.cin[l] > 32;
            if (o2 && k2) layer_f32(I2{}, I2{}, l);
            else if (o2) layer_f32(I2{}, I1{}, l);
            else if (k2) layer_f32(I1{}, I2{}, l);
            else layer_f32(I1{}, I1{}, l);
        }
        }
        if (ok) {
            float* yb = y + (size_t)b * coutL * P + p;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) { const int c = 32 * t + acc_row(i, h); if (c < coutL) yb[(size_t)c * P] = act[t][i]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 forward, UNIFORM layer shape.  Every layer is padded (with zero weights) to the same square shape
// 32 OT x 32 OT (OT = 1: widths <= 32, OT = 2: widths <= 64), so the layer body has no data-dependent branch: all
// 2 OT^2 weight fragments of a layer are fetched from LDS up front and its 2 OT^2 MFMAs issue back to back, the
// accumulators are converted pairwise to the next layer's operand fragments (v_cvt_pk_bf16_f32) and ReLU is a packed
// integer max with 0 (bf16 is sign-magnitude).  (The runtime-shaped kernel above put an `if` and a full
// `s_waitcnt lgkmcnt(0)` in front of every MFMA: 12 % MFMA utilisation.)  Padding costs 64 instead of 54 MFMAs per
// tile at depth 8 / width 64 / 3 in / 3 out.
// LDS image: [layer][ot][s][lane 64][8] bf16 fragments, then [layer][32 OT] float biases.
// PT = 2 pixel tiles (64 pixels) per wave and pass: every weight fragment read from LDS feeds two MFMAs and the bias
// is read once for both tiles.  With one tile per pass the kernel is LDS-bandwidth-bound: a 64x64 layer re-reads 8 KB of
// fragments + 8 KB of bias per 256 MFMA cycles of ONE SIMD, i.e. the CU's whole 128 B/clk, twice over.
constexpr int kPT = 2;

template <int OT, bool SMALLIO>
__global__ __launch_bounds__(kFwdThreads) __attribute__((amdgpu_waves_per_eu(2, 3))) void pixel_mlp_fwd_uniform_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ Bv,
                                                                             float* __restrict__ y, long npix_total, int P, PixelMlpDesc d) {
    using U = UniLds<OT>;
    constexpr int SS = U::SS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    stage_uniform<OT>(d, W, Bv, lds, threadIdx.x, kFwdThreads);
    __syncthreads();
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, r = lane & 31, h = lane >> 5;
    const int nl = d.nlayers, cin0 = d.cin[0], coutL = d.cout[nl - 1];
    const long ngroups = (npix_total + 32 * kPT - 1) / (32 * kPT);
    const long gstride = (long)gridDim.x * kFwdWaves;
    const unsigned char* bias0 = lds + nl * U::W_BYTES;
#ifndef NNS_PM_STAGGER
#define NNS_PM_STAGGER 0           // s_sleep argument (64-cycle units) for the second half of the workgroup's waves, once, before the tile loop
#endif
    if (NNS_PM_STAGGER && wave >= kFwdWaves / 2) __builtin_amdgcn_s_sleep(NNS_PM_STAGGER);
    // input pixels of group g as bf16 operand fragments (zero beyond cin0 / the last pixel)
    auto load_group = [&](long g, bf16x8 (&f)[kPT][SS]) {
#pragma unroll
        for (int pt = 0; pt < kPT; ++pt) {
            const long gp = (g * kPT + pt) * 32 + r;
            const bool ok = g < ngroups && gp < npix_total;
            const long gc = ok ? gp : npix_total - 1;                       // clamped: the load itself is unconditional
            load_frags<SS, SMALLIO>(x + (size_t)(gc / P) * cin0 * P + gc % P, (size_t)P, cin0, ok, h, f[pt]);
        }
    };
    for (long g = (long)blockIdx.x * kFwdWaves + wave; g < ngroups; g += gstride) {
        bf16x8 fr[kPT][SS];
        load_group(g, fr);                 // no register prefetch of the next group: the 32 VGPRs buy the second wave per SIMD
        // Layer loop, software-pipelined by hand: fragment n+1 is requested before the kPT MFMAs of fragment n (the
        // compiler alone serialises ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma: ~17 k wait cycles per 64 pixels), the
        // NEXT layer's first fragment and bias are requested before this layer's conversion phase, and the bias goes
        // straight into the first MFMA of each chain as its accumulator input.
        f32x16 acc[kPT][OT], bn[OT];
        const bf16x8* wl0 = reinterpret_cast<const bf16x8*>(lds) + lane;
        auto fetch_bias = [&](int l) {
            const float* bl = reinterpret_cast<const float*>(bias0 + l * U::B_BYTES);
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                for (int i = 0; i < 16; ++i) bn[ot][i] = bl[32 * ot + acc_row(i, h)];
        };
        fetch_bias(0);
#ifndef NNS_PM_PF
#define NNS_PM_PF 1                // weight fragments requested ahead of use (1, 2 or 4): the fragments of all layers are one contiguous stream in LDS
#endif
        constexpr int NF = OT * SS, PF = (NNS_PM_PF <= NF && NF % NNS_PM_PF == 0) ? NNS_PM_PF : 1;      // the queue is indexed at compile time
        const int nfr = nl * NF;
        bf16x8 q[PF];
#pragma unroll
        for (int dq = 0; dq < PF; ++dq) q[dq] = wl0[(dq < nfr ? dq : nfr - 1) * 64];
        for (int l = 0; l < nl; ++l) {
#pragma unroll
            for (int idx = 0; idx < NF; ++idx) {
                const int ot = idx / SS, s2 = idx % SS;
                const bf16x8 w = q[idx % PF];
                int nn = l * NF + idx + PF;
                nn = nn < nfr ? nn : nfr - 1;
                q[idx % PF] = wl0[nn * 64];
#ifndef NNS_PM_EXP
#define NNS_PM_EXP 0               // timing probes (wrong results): 1 = no accumulator -> operand conversion between layers, 2 = no MFMAs, 3 = no bias reads
#endif
#pragma unroll
                for (int pt = 0; pt < kPT; ++pt) {
                    if (NNS_PM_EXP == 2) { asm volatile("" :: "v"(w), "v"(fr[pt][s2])); if (s2 == 0) acc[pt][ot] = bn[ot]; }
                    else acc[pt][ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, fr[pt][s2], s2 == 0 ? bn[ot] : acc[pt][ot], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // one LDS read (fragment n + PF) ...
                __builtin_amdgcn_sched_group_barrier(0x008, kPT, 0);      // ... ahead of the MFMAs of fragment n
            }
            if (l + 1 < nl && NNS_PM_EXP != 1) {
                if (NNS_PM_EXP != 3) fetch_bias(l + 1);
#pragma unroll
                for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
                    for (int s = 0; s < SS; ++s) fr[pt][s] = pack8<true>(acc[pt][s >> 1], 8 * (s & 1));
            }
        }
#pragma unroll
        for (int pt = 0; pt < kPT; ++pt) {
            const long gp = (g * kPT + pt) * 32 + r;
            if (gp < npix_total) store_acc<OT, SMALLIO>(y + (size_t)(gp / P) * coutL * P + gp % P, (size_t)P, coutL, h, acc[pt]);
        }
    }
}

#ifndef NNS_PM_PIPE
#define NNS_PM_PIPE 1              // widths 33..64 with <= 4 channels in and out: 1 = pixel_mlp_fwd_pipe4_kernel (pixel_mlp_fwd4.hip), 0 = pixel_mlp_fwd_uniform_kernel<2, true>
#endif
template <int OT, bool SMALLIO>
int launch_fwd_uniform(const float* x, const float* weights, const float* biases, float* y, long npix, int P, const PixelMlpDesc& d, hipStream_t s) {
    const int lds = UniLds<OT>::total(d.nlayers);
    if (lds > 160 * 1024) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_fwd: weights need %d B of LDS (> 160 KiB)", lds);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_mlp_fwd_uniform_kernel<OT, SMALLIO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "pixel_mlp_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const long ngroups = (npix + 32 * kPT - 1) / (32 * kPT);
    // persistent: one generation of workgroups (2 per CU fit by LDS), so the weights are staged once per workgroup
    const long cap = kFwdThreads > 256 ? 256 : 512;          // workgroups resident at once (8-wave workgroups: one per CU by registers)
    long blocks = (ngroups + kFwdWaves - 1) / kFwdWaves; if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((pixel_mlp_fwd_uniform_kernel<OT, SMALLIO>), dim3((unsigned)blocks), dim3(kFwdThreads), lds, s, x, weights, biases, y, npix, P, d);
    return check_launch("pixel_mlp_fwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Backward (bfloat16 operands, float32 accumulation).  One launch computes, for the same stack,
//     gx = dL/dx,   gW_l = sum_pix delta_l a_{l-1}^T,   gb_l = sum_pix delta_l          given gy = dL/dy,
// WITHOUT activations saved by the forward: a workgroup (4 waves, one 32-pixel tile each = a 128-pixel super-tile)
// recomputes the forward in registers, keeping every layer's INPUT as the bf16 operand fragments the forward MFMAs
// consume (8 OT VGPRs per layer), then walks the layers backwards:
//   * data chain   delta_{l-1}^T[in x pix] = W_l^T[in x out] delta_l^T[out x pix]  -- the same accumulator-as-operand
//     chaining as the forward; its MFMAs are issued FIRST so that they run under the image writes below; ReLU mask
//     from the stored input fragments (a_{l-1} != 0);
//   * weight grads sum over the PIXEL index, which sits on the lanes of both delta_l and a_{l-1}: one transpose is
//     unavoidable.  Each wave stores its tiles as bf16 rows of two [128 pix][32 OT ch] LDS images (8-byte stores) and,
//     after a barrier, reads A = delta and B = activation fragments over the 128 pixels with ds_read_b64_tr_b16
//     (hardware transpose).  OT = 2: each wave owns one 32x32 block of the 64x64 gW_l; OT = 1: the single block's K
//     range (128 pixels) is split over the four waves.  Accumulated in registers across all super-tiles;
//   * bias grads   v_dot2c_f32_bf16 with ones on the delta fragments already read for the weight grads.
// Like the forward, every layer is padded to one compile-time square shape (32 OT)^2, so nothing inside a layer
// branches.  The weights live in LDS ONCE, as plain bf16 [out][in] matrices (padded rows): the forward's A fragments
// are two 8-byte reads of a row, the transposed product's A fragments two transposing reads of a 4 x 16 block.
// Per layer and tile: 8 + 8 + 8 (OT = 2) MFMA 32x32x16.  Partial gradients go to per-workgroup (OT = 2) or per-wave
// (OT = 1) workspace slices; pixel_mlp_reduce_kernel adds the slices in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------------------------
// LDS images of the bf16 backward kernels (round 4).  Four access patterns meet on them:
//   (a) frag_w   the forward product's A fragments: lane (row r, half h) reads two 8-byte pieces of one weight row (ds_read2_b64);
//   (b) frag_t   the transposed product's A fragments: ds_read_b64_tr_b16 over 4 rows x 16 columns per 16-lane group;
//   (c) the chain waves' 8-byte stores of delta / activation rows into the [pixel][channel] images (ds_write_b64);
//   (d) frag_pix the gradient waves' transposing reads of those images (the contraction over pixels).
// Rounds 2-3 kept plain rows of 2 CH + 8 bytes: (a) and (c) conflict-free, but a transposing read's 32-lane half takes four rows at
// 34-dword spacing -- rows q and q + 2 overlap on 12 of their 16 banks -- so EVERY (b) and (d) read took two LDS passes: 28 % of the
// kernel's LDS-active cycles were bank conflicts (profiles/r03_mfma_pmc_final.csv) with the LDS 64 % busy.  Now an image is cut into
// SUB-IMAGES of 16 columns, [column block][row][32 bytes]:
//     byte(sub, row, slot) = sub * SUB + 32 row + 16 (row >> 3) + 8 (slot ^ ((row >> 2) & 1)),     slot = the 8-byte piece 0..3 of the row
//   * a transposing read's four rows are 128 contiguous bytes, and the half's second 16-lane group reads the NEXT sub-image, SUB = 32 dwords
//     (mod 64) further: 64 distinct banks;
//   * 16 consecutive rows at one slot -- (a) and (c) -- hit 16 distinct bank pairs: 4 (row & 3) from the 32-byte rows, the XOR with row
//     bit 2 and the 16-byte pad per 8 rows supply the other two bits;
//   * every compile-time quantity (k-step, output tile, first / second piece, layer) stays an IMMEDIATE offset on one lane address.
// tools/lds_banks.py replays the four patterns under the hardware's bank rules (old layout: 4 LDS cycles per transposing read, new: 2).
//
// NNS_PM_LAYOUT = 2 (second step of round 4): the two 8-byte pieces a forward fragment takes from a row (slots h and 2 + h) sit NEXT to each
// other, so that (a) is ONE ds_read_b128 (4 LDS cycles per KB) instead of a ds_read2_b64 (8) and (c) one 16-byte store:
//     byte(sub, row, slot) = sub * SUB + 32 row + 8 (2 ((slot & 1) ^ row bit 2 ^ row bit 3) + (slot >> 1)),        no pad rows
// The 16-lane groups of a 16-byte read are non-contiguous ({0-3, 12-15, 20-27}, ...): rows 8 apart must differ in their 16-byte half -- row
// bit 3 in the XOR -- and 8 consecutive rows of a 16-byte store need row bit 2 in it.  In a transposing read row bit 3 is the compile-time
// "second read" bit, so its XOR cannot be an immediate: such reads keep TWO lane addresses (base, base ^ 16) -- one register more.
#ifndef NNS_PM_LAYOUT
#define NNS_PM_LAYOUT 2
#endif
template <int OT>
struct BwdLds {
    static constexpr int SS = 2 * OT, CH = 32 * OT;
    static constexpr int NSUB = 2 * OT;                            // 16-column sub-images per image
    static constexpr int sub_bytes(int nrows) {
        const int b = nrows * 32 + (NNS_PM_LAYOUT == 2 ? 0 : (nrows / 8) * 16);
        return b + ((32 - (b / 4) % 64 + 64) % 64) * 4;           // consecutive sub-images 32 banks apart
    }
    static constexpr int W_SUB = sub_bytes(CH), IMG_SUB = sub_bytes(128);
    static constexpr int W_BYTES = NSUB * W_SUB;
    static constexpr int B_BYTES = CH * 4;
    static constexpr int IMG_BYTES = NSUB * IMG_SUB;
#ifndef NNS_PM_IMGSETS
#define NNS_PM_IMGSETS 2                                       // 2: the images are double-buffered over the layers (one barrier per layer)
#endif
    __host__ __device__ static int total(int nl) { return nl * (W_BYTES + B_BYTES) + NNS_PM_IMGSETS * 2 * IMG_BYTES; }
    // byte offset of the 8-byte piece `slot` of row `row` inside one sub-image
    __host__ __device__ static constexpr int piece(int row, int slot) {
        if (NNS_PM_LAYOUT == 2) return 32 * row + 8 * (2 * ((slot & 1) ^ ((row >> 2) & 1) ^ ((row >> 3) & 1)) + (slot >> 1));
        return 32 * row + 16 * (row >> 3) + 8 * (slot ^ ((row >> 2) & 1));
    }
    // weight element (row = out, col = in) of a layer's image
    __host__ __device__ static constexpr int w_elem(int row, int col) { return (col >> 4) * W_SUB + piece(row, (col & 15) >> 2) + 2 * (col & 3); }
};

__device__ __forceinline__ bf16x8 join8(bf16x4 lo, bf16x4 hi) {
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// A fragment of the forward product: rows 32 ot + r of the weight image, k-step s (columns 16 s ..: sub-image s); lane half h takes the
// pieces h and 2 + h of the row (channels 16 s + 4 h + 0..3 and 16 s + 8 + 4 h + 0..3: the accumulator-as-operand k order).  One ds_read2_b64.
template <int OT>
__device__ __forceinline__ bf16x8 frag_w(const unsigned char* wimg, int r, int h, int ot, int s) {
    using U = BwdLds<OT>;
    if constexpr (NNS_PM_LAYOUT == 2) {
        const unsigned char* a = wimg + (32 * r + 16 * (h ^ ((r >> 2) & 1) ^ ((r >> 3) & 1))) + (s * U::W_SUB + ot * (32 * 32));
        return *reinterpret_cast<const bf16x8*>(a);                               // pieces h and 2 + h, adjacent: one ds_read_b128
    } else {
    const unsigned char* a = wimg + (32 * r + 16 * (r >> 3) + 8 * (h ^ ((r >> 2) & 1))) + (s * U::W_SUB + ot * (32 * 32 + 4 * 16));
    return join8(*reinterpret_cast<const bf16x4*>(a), *reinterpret_cast<const bf16x4*>(a + 16));
    }
}

// The lane part of a transposing read's address (frag_t, frag_pix): lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the
// group's 4 x 16 block; the half's second group (lane bit 4) reads the next sub-image; lane half h' takes rows +4.
template <int SUB>
__device__ __forceinline__ int tr_lane(int lane, int second = 0) {
    const int gl = lane & 15, q = gl >> 2, pp = gl & 3, hp = lane >> 5;
    if constexpr (NNS_PM_LAYOUT == 2)       // rows 4 hp + q (+ 8 for the second read: row bit 3 joins the XOR): piece pp at 2 ((pp & 1) ^ hp ^ second) + (pp >> 1)
        return ((lane >> 4) & 1) * SUB + (4 * hp + q) * 32 + 8 * (2 * ((pp & 1) ^ hp ^ second) + (pp >> 1));
    return ((lane >> 4) & 1) * SUB + (4 * hp + q) * 32 + 8 * (pp ^ hp);
}
// Transposing fragment read: element j of lane (r, h) = M[16 s + 8 (j>>2) + 4 h + (j&3)][col_block + r] of a weight image (rows = out).
// Lane i of a group receives column i of the four rows.  EXEC must be all ones here.
template <int OT>
__device__ __forceinline__ bf16x8 frag_t(const unsigned char* img, int lane, int s, int col_block) {
    using U = BwdLds<OT>;
    using lds_v4 = __attribute__((address_space(3))) bf16x4;
    if constexpr (NNS_PM_LAYOUT == 2) {
        const int off = (col_block >> 4) * U::W_SUB + s * (16 * 32);
        return join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(img + tr_lane<U::W_SUB>(lane, 0) + off)),
                     __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(img + tr_lane<U::W_SUB>(lane, 1) + off + 8 * 32)));
    } else {
    const unsigned char* a0 = img + tr_lane<U::W_SUB>(lane) + ((col_block >> 4) * U::W_SUB + s * (16 * 32 + 2 * 16));
    return join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + (8 * 32 + 16))));
    }
}

// 8 pixels of channel ch_block + r from a [pix][ch] image for the contraction over pixels: element j of lane (r, h) = pixel
// 16 s + 8 (j>>2) + 4 h + (j&3) (the same k order for both operands of the product; round 3 used 16 s + 8 h + j)
template <int OT>
__device__ __forceinline__ bf16x8 frag_pix(const unsigned char* img, int lane, int s, int ch_block) {
    using U = BwdLds<OT>;
    using lds_v4 = __attribute__((address_space(3))) bf16x4;
    if constexpr (NNS_PM_LAYOUT == 2) {
        const int off = (ch_block >> 4) * U::IMG_SUB + s * (16 * 32);
        return join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(img + tr_lane<U::IMG_SUB>(lane, 0) + off)),
                     __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(img + tr_lane<U::IMG_SUB>(lane, 1) + off + 8 * 32)));
    } else {
    const unsigned char* a0 = img + tr_lane<U::IMG_SUB>(lane) + (ch_block >> 4) * U::IMG_SUB + s * (16 * 32 + 2 * 16);
    return join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + (8 * 32 + 16))));
    }
}
// The lane part of the chain waves' image row stores: row 32 wave + r, piece h (+ 2 for the fragment's second half), sub-image = k-step
template <int OT>
__device__ __forceinline__ int img_row_lane(int wave, int r, int h) {
    const int row = 32 * wave + r;
    if constexpr (NNS_PM_LAYOUT == 2) return 32 * row + 16 * (h ^ ((row >> 2) & 1) ^ ((row >> 3) & 1));     // the fragment's two pieces: 16 contiguous bytes
    return 32 * row + 16 * (row >> 3) + 8 * (h ^ ((row >> 2) & 1));
}

template <int OT, bool SMALL>
__device__ __forceinline__ void load_acc(const float* __restrict__ gb, size_t P, int cout, bool ok, int h, f32x16 (&a)[OT]) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[t][i] = 0.f;
    if constexpr (SMALL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = gb[(size_t)(i < cout ? i : 0) * P];
            a[0][i] = (ok && h == 0 && i < cout) ? v : 0.f;
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
            const float v = gb[(size_t)(c < cout ? c : cout - 1) * P];
            a[t][i] = (ok && c < cout) ? v : 0.f;
        }
    }
}

template <int OT, bool SMALLIO>
__global__ __launch_bounds__(256) void pixel_mlp_bwd_uniform_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                     const float* __restrict__ W, const float* __restrict__ Bv,
                                                                     float* __restrict__ gx, float* __restrict__ ws,
                                                                     long npix_total, int P, PixelMlpDesc d, int nparams_w, int nparams) {
    using U = BwdLds<OT>;
    constexpr int SS = U::SS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nl = d.nlayers;
    {   // stage: zero everything (pads, images), then scatter the real matrices (coalesced reads)
        const int total = U::total(nl);
        for (int e = threadIdx.x; e < total / 16; e += 256) reinterpret_cast<uint4*>(lds)[e] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (int l = 0; l < nl; ++l) {
            const int cin = d.cin[l], cout = d.cout[l], n = cin * cout;
            const float* Wl = W + d.woff[l];
            unsigned char* dst = lds + l * U::W_BYTES;
            int row = threadIdx.x / cin, k = threadIdx.x - row * cin;
            const int drow = 256 / cin, dk = 256 - drow * cin;
            for (int e = threadIdx.x; e < n; e += 256) {
                *reinterpret_cast<unsigned short*>(dst + U::w_elem(row, k)) = f2bf(Wl[e]);
                row += drow; k += dk;
                if (k >= cin) { k -= cin; ++row; }
            }
            float* bl = reinterpret_cast<float*>(lds + nl * U::W_BYTES + l * U::B_BYTES);
            for (int e = threadIdx.x; e < cout; e += 256) bl[e] = Bv[d.boff[l] + e];
        }
        __syncthreads();
    }
    const unsigned char* bias0 = lds + nl * U::W_BYTES;
    unsigned char* img0 = lds + nl * (U::W_BYTES + U::B_BYTES);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, r = lane & 31, h = lane >> 5;
    const int bo = OT == 2 ? wave >> 1 : 0, bi = OT == 2 ? wave & 1 : 0;       // this wave's gW block
    constexpr int KS = OT == 2 ? 8 : 2;                                       // its k-steps (of 8 x 16 pixels)
    const int ks0 = OT == 2 ? 0 : 2 * wave;
    const bool do_gb = OT == 2 ? bi == 0 : true;
    const int cin0 = d.cin[0], coutL = d.cout[nl - 1];

    f32x16 gw[kMaxLayers];
    float gbp[kMaxLayers];
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        gbp[l] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) gw[l][i] = 0.f;
    }
    const long nsuper = (npix_total + 127) / 128;
    for (long sup = blockIdx.x; sup < nsuper; sup += gridDim.x) {
        const long gp = sup * 128 + wave * 32 + r;
        const bool ok = gp < npix_total;
        const long gc = ok ? gp : npix_total - 1;
        const long b = gc / P, p = gc % P;
        // ---------------- forward: afrag[l] = input fragments of layer l
        bf16x8 afrag[kMaxLayers][SS];
        load_frags<SS, SMALLIO>(x + (size_t)b * cin0 * P + p, (size_t)P, cin0, ok, h, afrag[0]);
#pragma unroll
        for (int l = 0; l + 1 < kMaxLayers; ++l) {
            if (l + 1 < nl) {
                __builtin_amdgcn_sched_barrier(0);            // phase boundaries: keep fragment loads from being hoisted across
                const unsigned char* wimg = lds + l * U::W_BYTES;
                const float* bl = reinterpret_cast<const float*>(bias0 + l * U::B_BYTES);
                f32x16 acc[OT];
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ot][i] = bl[32 * ot + acc_row(i, h)];
#pragma unroll
                    for (int s = 0; s < SS; ++s) acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_w<OT>(wimg, r, h, ot, s), afrag[l][s], acc[ot], 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < SS; ++s) afrag[l + 1][s] = pack8<true>(acc[s >> 1], 8 * (s & 1));
            }
        }
        // ---------------- backward
        bf16x8 dfrag[SS];
        {
            f32x16 dl[OT];
            load_acc<OT, SMALLIO>(gy + (size_t)b * coutL * P + p, (size_t)P, coutL, ok, h, dl);
#pragma unroll
            for (int s = 0; s < SS; ++s) dfrag[s] = pack8<false>(dl[s >> 1], 8 * (s & 1));
        }
#pragma unroll
        for (int l = kMaxLayers - 1; l >= 0; --l) {
            if (l < nl) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* wimg = lds + l * U::W_BYTES;
                // data chain first: its MFMAs run while the images are written
                f32x16 nd[OT];
#pragma unroll
                for (int it = 0; it < OT; ++it) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) nd[it][i] = 0.f;
#pragma unroll
                    for (int s = 0; s < SS; ++s) nd[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_t<OT>(wimg, lane, s, 32 * it), dfrag[s], nd[it], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // Consecutive layers use alternate image sets: a wave may write layer l-1's images while slower waves still read
                // layer l's, so a layer needs ONE workgroup barrier (write -> read), not two; the set of layer l+1 is free
                // again because every wave passed this layer's barrier after reading it.  (One more barrier per super-tile.)
                unsigned char* imgD = img0 + (NNS_PM_IMGSETS == 2 ? (l & 1) : 0) * 2 * U::IMG_BYTES;
                unsigned char* imgA = imgD + U::IMG_BYTES;
                {   // delta_l and a_{l-1} as bf16 rows [32 wave + r] of the images
                    unsigned char* rowD = imgD + img_row_lane<OT>(wave, r, h);
                    unsigned char* rowA = imgA + img_row_lane<OT>(wave, r, h);
#pragma unroll
                    for (int s = 0; s < SS; ++s) {
                        // fragment elements 0..3 = channels 16 s + 4 h + (0..3): piece h of sub-image s; elements 4..7 = channels 16 s + 8 + 4 h + (0..3): piece 2 + h
                        constexpr int P2 = NNS_PM_LAYOUT == 2 ? 8 : 16;          // byte distance of the fragment's second piece
                        *reinterpret_cast<bf16x4*>(rowD + s * U::IMG_SUB) = __builtin_shufflevector(dfrag[s], dfrag[s], 0, 1, 2, 3);
                        *reinterpret_cast<bf16x4*>(rowD + s * U::IMG_SUB + P2) = __builtin_shufflevector(dfrag[s], dfrag[s], 4, 5, 6, 7);
                        *reinterpret_cast<bf16x4*>(rowA + s * U::IMG_SUB) = __builtin_shufflevector(afrag[l][s], afrag[l][s], 0, 1, 2, 3);
                        *reinterpret_cast<bf16x4*>(rowA + s * U::IMG_SUB + P2) = __builtin_shufflevector(afrag[l][s], afrag[l][s], 4, 5, 6, 7);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const bf16x8 fa = frag_pix<OT>(imgD, lane, ks0 + kk, 32 * bo);
                    const bf16x8 fb = frag_pix<OT>(imgA, lane, ks0 + kk, 32 * bi);
                    gw[l] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, gw[l], 0, 0, 0);
                    if (kk & 1) __builtin_amdgcn_sched_barrier(0);
                    if (do_gb) {
                        const bf16x2v ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            const unsigned pr = (unsigned)(unsigned short)fa[j] | ((unsigned)(unsigned short)fa[j + 1] << 16);
                            gbp[l] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, pr), ones, gbp[l], false);
                        }
                    }
                }
                if (NNS_PM_IMGSETS != 2) __syncthreads();
                if (l > 0) {
                    // ReLU mask a_{l-1} != 0 (activations are >= 0), then the next layer's operand fragments
#pragma unroll
                    for (int s = 0; s < SS; ++s) dfrag[s] = pack8_masked(nd[s >> 1], 8 * (s & 1), afrag[l][s]);
                } else if (ok) {
                    store_acc<OT, SMALLIO>(gx + (size_t)b * cin0 * P + p, (size_t)P, cin0, h, nd);
                }
            }
        }
        if (NNS_PM_IMGSETS == 2) __syncthreads();          // the next super-tile's first layer may reuse the set layer 0 just read
    }
    // ---------------- partial gradients: workspace slice per workgroup (OT = 2) or per wave (OT = 1)
    float* wsb = ws + (size_t)(OT == 2 ? blockIdx.x : blockIdx.x * 4 + wave) * nparams;
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        if (l < nl) {
            const int cin = d.cin[l], cout = d.cout[l];
            const int in = 32 * bi + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int out = 32 * bo + acc_row(i, h);
                if (out < cout && in < cin) wsb[d.woff[l] + out * cin + in] = gw[l][i];
            }
            if (do_gb) {
                const float tot = gbp[l] + __shfl_xor(gbp[l], 32);
                if (h == 0 && 32 * bo + r < cout) wsb[nparams_w + d.boff[l] + 32 * bo + r] = tot;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same backward with SPLIT ROLES (round 2): pixel_mlp_bwd_uniform_kernel serialises, in ONE wave per SIMD (455 registers), the
// forward recompute, the data chain, the image writes, a barrier and the pixel contraction of the weight gradients -- 4500 cycles per
// layer and tile against 770 of MFMA.  Here a workgroup has EIGHT waves, two per SIMD, each under 256 registers:
//   * four CHAIN waves (one 32-pixel tile each): forward recompute, data chain, delta_l / a_{l-1} images -- no weight-gradient
//     accumulators (128 registers less);
//   * four GRADIENT waves (one 32x32 block of every layer's gW each; OT = 1: a quarter of the pixels each): after the layer's barrier
//     they contract the images over the 128 pixels while the chain waves are already on the next layer -- no activations kept
//     (128 registers less).
// Per layer still ONE workgroup barrier: the chain waves write layer l-1's images into the set the gradient waves finished reading
// before that barrier (layer l+1's).  The two role bodies are separate code paths (separate loops with matching barriers), so that the
// register allocator sees two small live sets instead of their union.
// ------------------------------------------------------------------------------------------------------------------
#ifndef NNS_PMB_DEPTH
#define NNS_PMB_DEPTH 4            // operand fragments in flight per MFMA stream of the split backward
#endif
constexpr int kBwdDepth = NNS_PMB_DEPTH;
#ifndef NNS_PMB_TIMING
#define NNS_PMB_TIMING 0           // 1: the two-tile kernel prints the cycles of one super-tile's forward and backward halves (wave 0 of workgroup 0)
#endif
#ifndef NNS_PMB_TIMING
#define NNS_PMB_TIMING 0           // 1: the split backward prints the cycles of one super-tile's phases (s_memtime stamps in wave 0 of workgroup 0)
#endif
#ifndef NNS_PMB_PRIO
#define NNS_PMB_PRIO 0
#endif
#ifndef NNS_PMB_ALT
#define NNS_PMB_ALT 0              // 1: the data chain's MFMAs alternate between the two accumulators
#endif
constexpr bool kBwdAlt = NNS_PMB_ALT != 0;
#ifndef NNS_PMB_OVERLAP
#define NNS_PMB_OVERLAP 0          // bit 0 (forward recompute) / bit 1 (backward walk): the chain waves convert the FIRST output tile's accumulators (mask / pack,
                                   // 4 - 8 vector instructions per MFMA gap) under the MFMAs of the second tile instead of after the layer's last MFMA (OT = 2 only).
                                   // Round 4, same-box A/B (profiles/r04_ab_pixel_mlp_bwd.log): both 0.96 -> 1.04 ms -- the gap's issue slots are taken (2 LDS reads,
                                   // a store pair and the MFMA's own 8 cycles), the conversion only stretches the MFMA loop
#endif
#ifndef NNS_PMB_EXP
#define NNS_PMB_EXP 0              // timing probes of the split backward (wrong results): 1 = chain waves read no weight fragments from LDS, 2 = no per-layer barriers, 3 = no weight-gradient MFMAs,
                                   // 4 = gradient waves only keep the barriers, 5 = no forward recompute, 6 = no image writes, 7 = no ReLU' mask
#endif
#define PMB(k) (((NNS_PMB_EXP) >> ((k) - 1)) & 1)     // probe k is bit k-1 of NNS_PMB_EXP, so that probes combine; 8 / 9: forward / backward conversion replaced by a register reinterpretation
__device__ __forceinline__ bf16x8 raw8(const f32x16& a, int base) { const i32x4v r = {__builtin_bit_cast(int, a[base]), __builtin_bit_cast(int, a[base + 1]), __builtin_bit_cast(int, a[base + 2]), __builtin_bit_cast(int, a[base + 3])}; return __builtin_bit_cast(bf16x8, r); }
// Staging of the split backward kernels: zero everything (pads, images), then scatter the real matrices (coalesced reads).  512 threads.
template <int OT>
__device__ __forceinline__ void bwd_stage(unsigned char* lds, const float* __restrict__ W, const float* __restrict__ Bv, const PixelMlpDesc& d) {
    using U = BwdLds<OT>;
    const int nl = d.nlayers;
    {
        const int total = U::total(nl);
        for (int e = threadIdx.x; e < total / 16; e += 512) reinterpret_cast<uint4*>(lds)[e] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (int l = 0; l < nl; ++l) {
            const int cin = d.cin[l], cout = d.cout[l], n = cin * cout;
            const float* Wl = W + d.woff[l];
            unsigned char* dst = lds + l * U::W_BYTES;
            {   // eight reads in flight per thread and round (a dependent L2 round trip per element otherwise: see stage_uniform)
                const unsigned magic = ((1u << 20) + (unsigned)cin - 1u) / (unsigned)cin;     // e / cin = (e * magic) >> 20, exact for e < 64 cin
                for (int e0 = threadIdx.x; e0 < n; e0 += 8 * 512) {
                    float vq[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const int e = e0 + q * 512; vq[q] = Wl[e < n ? e : n - 1]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = e0 + q * 512;
                        if (e < n) { const int row = (int)(((unsigned)e * magic) >> 20); *reinterpret_cast<unsigned short*>(dst + U::w_elem(row, e - row * cin)) = f2bf(vq[q]); }
                    }
                }
            }
            float* bl = reinterpret_cast<float*>(lds + nl * U::W_BYTES + l * U::B_BYTES);
            for (int e = threadIdx.x; e < cout; e += 512) bl[e] = Bv[d.boff[l] + e];
        }
        __syncthreads();
    }
}

// The GRADIENT waves of the split backward kernels (waves 4..7 of the workgroup): see pixel_mlp_bwd_split_kernel.
template <int OT>
__device__ __forceinline__ void bwd_gradient_waves(unsigned char* img0, int wave, int lane, long nsuper, const PixelMlpDesc& d,
                                                   float* __restrict__ ws, int nparams_w, int nparams) {
    using U = BwdLds<OT>;
    const int nl = d.nlayers, r = lane & 31, h = lane >> 5;
    // ================= gradient waves =================
    const int gwv = wave - 4;
    const int bo = OT == 2 ? gwv >> 1 : 0, bi = OT == 2 ? gwv & 1 : 0;     // this wave's gW block
    constexpr int KS = OT == 2 ? 8 : 2;                                     // its k-steps (of 8 x 16 pixels)
    const int ks0 = OT == 2 ? 0 : 2 * gwv;
    const bool do_gb = OT == 2 ? bi == 0 : true;
    f32x16 gw[kMaxLayers];
    float gbp[kMaxLayers];
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        gbp[l] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) gw[l][i] = 0.f;
    }
    for (long sup = blockIdx.x; sup < nsuper; sup += gridDim.x) {
#pragma unroll
        for (int l = kMaxLayers - 1; l >= 0; --l) {
            if (l < nl) {
                const unsigned char* imgD = img0 + (l & 1) * 2 * U::IMG_BYTES;
                const unsigned char* imgA = imgD + U::IMG_BYTES;
                if (!PMB(2)) __syncthreads();                      // layer l's images are written
                if (PMB(4)) continue;
                // the operand fragments of k-step kk + GD are requested before the MFMA of k-step kk (round 3: an LDS read takes longer than
                // one MFMA, so one step ahead still left every MFMA waiting)
                constexpr int GD = KS < kBwdDepth ? KS : kBwdDepth;
                // (two forms behind a wave-uniform branch: written as a condition inside the loop the bias sums were computed by every wave and
                // selected away; four partial sums: one accumulator made the 32 dot products of a layer a dependent chain)
                auto contract = [&](auto gbc) {
                    constexpr bool with_gb = decltype(gbc)::value;
                    bf16x8 fa_r[GD], fb_r[GD];
                    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < GD; ++q) { fa_r[q] = frag_pix<OT>(imgD, lane, ks0 + q, 32 * bo); fb_r[q] = frag_pix<OT>(imgA, lane, ks0 + q, 32 * bi); }
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) {
                        const bf16x8 fa = fa_r[kk % GD], fb = fb_r[kk % GD];
                        if (!PMB(3)) gw[l] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, gw[l], 0, 0, 0);
                        if (kk + GD < KS) { fa_r[kk % GD] = frag_pix<OT>(imgD, lane, ks0 + kk + GD, 32 * bo); fb_r[kk % GD] = frag_pix<OT>(imgA, lane, ks0 + kk + GD, 32 * bi); }
                        if (kk & 1) __builtin_amdgcn_sched_barrier(0);
                        if constexpr (with_gb) {
                            const bf16x2v ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const unsigned pr = (unsigned)(unsigned short)fa[2 * j] | ((unsigned)(unsigned short)fa[2 * j + 1] << 16);
                                part[j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, pr), ones, part[j], false);
                            }
                        }
                    }
                    if constexpr (with_gb) gbp[l] += (part[0] + part[1]) + (part[2] + part[3]);
                };
                if (do_gb) contract(std::true_type{}); else contract(std::false_type{});
            }
        }
        if (nl & 1) __syncthreads();                                        // end of the super-tile: see the chain waves
    }
    float* wsb = ws + (size_t)(OT == 2 ? blockIdx.x : blockIdx.x * 4 + gwv) * nparams;
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        if (l < nl) {
            const int cin = d.cin[l], cout = d.cout[l];
            const int in = 32 * bi + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int out = 32 * bo + acc_row(i, h);
                if (out < cout && in < cin) wsb[d.woff[l] + out * cin + in] = gw[l][i];
            }
            if (do_gb) {
                const float tot = gbp[l] + __shfl_xor(gbp[l], 32);
                if (h == 0 && 32 * bo + r < cout) wsb[nparams_w + d.boff[l] + 32 * bo + r] = tot;
            }
        }
    }
}

template <int OT, bool SMALLIO>
__global__ __launch_bounds__(512) void pixel_mlp_bwd_split_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                   const float* __restrict__ W, const float* __restrict__ Bv,
                                                                   float* __restrict__ gx, float* __restrict__ ws,
                                                                   long npix_total, int P, PixelMlpDesc d, int nparams_w, int nparams) {
    using U = BwdLds<OT>;
    constexpr int SS = U::SS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nl = d.nlayers;
    bwd_stage<OT>(lds, W, Bv, d);
    const unsigned char* bias0 = lds + nl * U::W_BYTES;
    unsigned char* img0 = lds + nl * (U::W_BYTES + U::B_BYTES);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, r = lane & 31, h = lane >> 5;
    const long nsuper = (npix_total + 127) / 128;
    if (wave >= 4) {
        bwd_gradient_waves<OT>(img0, wave, lane, nsuper, d, ws, nparams_w, nparams);
        return;
    }
    // ================= chain waves =================
    if (NNS_PMB_PRIO) __builtin_amdgcn_s_setprio(NNS_PMB_PRIO);                 // the chain is the critical path; the gradient waves fill its gaps
    const int cin0 = d.cin[0], coutL = d.cout[nl - 1];
    // this lane's pixel in a super-tile: in range?, batch entry, pixel within the field
    auto locate = [&](long sup_, bool& ok_, int& b_, int& p_) {
        const long gp = sup_ * 128 + wave * 32 + r;
        ok_ = gp < npix_total;
        const long gc = ok_ ? gp : npix_total - 1;
        if (npix_total <= 0x7fffffffL) { const unsigned q = (unsigned)gc / (unsigned)P; b_ = (int)q; p_ = (int)((unsigned)gc - q * (unsigned)P); }
        else { b_ = (int)(gc / P); p_ = (int)(gc % P); }
    };
    // SMALLIO: channels 0..3 of a [channel][pixel] field, unconverted, so that the load can stay in flight (lane half 0 holds them)
    // (the values are masked where they are USED: a select next to the load would wait for it)
    auto raw4 = [&](const float* __restrict__ fb, int nch, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fb[(size_t)(i < nch ? i : 0) * P];
    };
    // Round 3: a super-tile used to START with its x loads and its backward walk with its gy loads -- two exposed HBM round trips
    // (~10 k of ~21 k cycles per super-tile, measured with s_memtime).  Now gy is requested before the forward recompute and the NEXT
    // super-tile's x before the backward walk.
    bool okn = false; int bn_ = 0, pn_ = 0; float xr[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SMALLIO) { locate(blockIdx.x, okn, bn_, pn_); raw4(x + (size_t)bn_ * cin0 * P + pn_, cin0, xr); }
#if NNS_PMB_TIMING
    long tk[4] = {0, 0, 0, 0}, tbody = 0, tbar = 0;
#endif
    for (long sup = blockIdx.x; sup < nsuper; sup += gridDim.x) {
#if NNS_PMB_TIMING
        const bool timed = sup == blockIdx.x + 3 * (long)gridDim.x;
        if (timed) tk[0] = clock64();
#endif
        bool ok; int b, p;
        if constexpr (SMALLIO) { ok = okn; b = bn_; p = pn_; } else locate(sup, ok, b, p);
        // ---------------- forward: afrag[l] = input fragments of layer l
        bf16x8 afrag[kMaxLayers][SS];
        float gr[4];
        if constexpr (SMALLIO) {
#pragma unroll
            for (int s = 0; s < SS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) afrag[0][s][j] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) afrag[0][0][j] = (short)f2bf((ok && h == 0 && j < cin0) ? xr[j] : 0.f);
            raw4(gy + (size_t)b * coutL * P + p, coutL, gr);
        } else
        load_frags<SS, SMALLIO>(x + (size_t)b * cin0 * P + p, (size_t)P, cin0, ok, h, afrag[0]);
        // Weight fragments are requested kBwdDepth MFMAs AHEAD of their use, across layer boundaries too (round 3; probe: with no fragment
        // reads at all the kernel takes 1.06 instead of 1.37 ms -- an LDS read issued by one wave per SIMD takes ~100 cycles, three MFMAs).
        constexpr int NM = OT * SS, D = kBwdDepth < NM ? kBwdDepth : NM;
        static_assert(NM % D == 0, "ring slots must line up across layers");
        bf16x8 wr[D];
#pragma unroll
        for (int q = 0; q < D; ++q) wr[q] = frag_w<OT>(lds, r, h, q / SS, q % SS);
        f32x16 accb[2][OT];                                                                        // accumulators by layer parity: the other set takes the next bias
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int i = 0; i < 16; ++i) accb[0][ot][i] = reinterpret_cast<const float*>(bias0)[32 * ot + acc_row(i, h)];
#pragma unroll
        for (int l = 0; l + 1 < kMaxLayers; ++l) {
            if (l + 1 < nl) {
                if (PMB(5)) {
#pragma unroll
                    for (int s = 0; s < SS; ++s) afrag[l + 1][s] = afrag[l][s];
                    continue;
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* wimg = lds + l * U::W_BYTES;
                const unsigned char* wnext = lds + (l + 2 < nl ? l + 1 : l) * U::W_BYTES;          // the next recomputed layer's image (clamped)
                f32x16 (&acc)[OT] = accb[l & 1];                                                   // holds this layer's bias already
                f32x16 (&accn)[OT] = accb[(l + 1) & 1];
                i32x4v cv[2];                                                                      // NNS_PMB_OVERLAP: tile 0's two fragments, converted under tile 1's MFMAs
                static_for<0, NM>([&](auto ic) {
                    constexpr int idx = decltype(ic)::value, ot = idx / SS, s2 = idx % SS, nx = idx + D;
                    acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PMB(1) ? afrag[l][(s2 + 1) % SS] : wr[idx % D], afrag[l][s2], acc[ot], 0, 0, 0);
                    if constexpr (nx < NM) wr[idx % D] = frag_w<OT>(wimg, r, h, nx / SS, nx % SS);
                    else wr[idx % D] = frag_w<OT>(wnext, r, h, (nx - NM) / SS, (nx - NM) % SS);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                              // a frag_w is one ds_read2_b64
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr ((NNS_PMB_OVERLAP & 1) && OT == 2 && idx >= SS && !PMB(8)) {
                        // tile 0's accumulator is complete: a quarter of its conversion per MFMA of tile 1 (fragment (idx - SS) >> 1, its ints 2 q, 2 q + 1)
                        constexpr int pc = idx - SS, fs = pc >> 1, q = pc & 1;
                        cv[fs][2 * q] = pack2<true>(acc[0][8 * fs + 4 * q], acc[0][8 * fs + 4 * q + 1]);
                        cv[fs][2 * q + 1] = pack2<true>(acc[0][8 * fs + 4 * q + 2], acc[0][8 * fs + 4 * q + 3]);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                });
                {   // the next layer's bias, requested under this layer's conversion (a read placed at its use is the youngest in the queue: lgkmcnt(0))
                    const float* bn = reinterpret_cast<const float*>(bias0 + (l + 2 < nl ? l + 1 : l) * U::B_BYTES);
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                        for (int i = 0; i < 16; ++i) accn[ot][i] = bn[32 * ot + acc_row(i, h)];
                    __builtin_amdgcn_sched_group_barrier(0x100, 4 * OT, 0);
                }
#pragma unroll
                for (int s = 0; s < SS; ++s) {
                    if ((NNS_PMB_OVERLAP & 1) && OT == 2 && !PMB(8) && s < 2) afrag[l + 1][s] = __builtin_bit_cast(bf16x8, cv[s]);
                    else afrag[l + 1][s] = PMB(8) ? raw8(acc[s >> 1], 8 * (s & 1)) : pack8<true>(acc[s >> 1], 8 * (s & 1));
                }
            }
        }
#if NNS_PMB_TIMING
        if (timed) tk[1] = clock64();
#endif
        // ---------------- backward
        bf16x8 dfrag[SS];
        {
            f32x16 dl[OT];
            if constexpr (SMALLIO) {
                // (unconditional -- past the end the current pixel again: behind a branch the waits for gy below would also wait for these)
                locate(sup + gridDim.x < nsuper ? sup + gridDim.x : sup, okn, bn_, pn_);
                raw4(x + (size_t)bn_ * cin0 * P + pn_, cin0, xr);
#pragma unroll
                for (int t = 0; t < OT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) dl[t][i] = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) dl[0][i] = (ok && h == 0 && i < coutL) ? gr[i] : 0.f;
            } else
            load_acc<OT, SMALLIO>(gy + (size_t)b * coutL * P + p, (size_t)P, coutL, ok, h, dl);
#pragma unroll
            for (int s = 0; s < SS; ++s) dfrag[s] = pack8<false>(dl[s >> 1], 8 * (s & 1));
        }
#if NNS_PMB_TIMING
        if (timed) tk[2] = clock64();
#endif
        bf16x8 tr[D];
#pragma unroll
        for (int q = 0; q < D; ++q) tr[q] = frag_t<OT>(lds + (nl - 1) * U::W_BYTES, lane, kBwdAlt ? q / OT : q % SS, 32 * (kBwdAlt ? q % OT : q / SS));
#pragma unroll
        for (int l = kMaxLayers - 1; l >= 0; --l) {
            if (l < nl) {
#if NNS_PMB_TIMING
                long tl0 = 0, tl1 = 0;
                if (timed) tl0 = clock64();
#endif
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* wimg = lds + l * U::W_BYTES;
                f32x16 nd[OT];
#pragma unroll
                for (int it = 0; it < OT; ++it)
#pragma unroll
                    for (int i = 0; i < 16; ++i) nd[it][i] = 0.f;
                const unsigned char* wprev = lds + (l > 0 ? l - 1 : 0) * U::W_BYTES;               // the next layer of the walk
                // delta_l and a_{l-1} are known on entry: their image rows are stored UNDER the layer's MFMAs (the set is free once the barrier of
                // layer l + 1 is behind this wave), so that only the barrier itself stands between the last MFMA and the mask / convert step.
                unsigned char* imgD = img0 + (l & 1) * 2 * U::IMG_BYTES;
                unsigned char* rowD = imgD + img_row_lane<OT>(wave, r, h);
                unsigned char* rowA = rowD + U::IMG_BYTES;
                i32x4v dn[2];                                                                      // NNS_PMB_OVERLAP: the next delta's first two fragments (from nd[0])
                constexpr bool kOvl = (NNS_PMB_OVERLAP & 2) && OT == 2 && !kBwdAlt && !PMB(7) && !PMB(9);
                static_for<0, NM>([&](auto ic) {
                    constexpr int idx = decltype(ic)::value, it = kBwdAlt ? idx % OT : idx / SS, s2 = kBwdAlt ? idx / OT : idx % SS, nx = idx + D;
                    constexpr int nit = kBwdAlt ? nx % OT : (nx % NM) / SS, ns2 = kBwdAlt ? (nx % NM) / OT : nx % SS;
                    nd[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PMB(1) ? dfrag[(s2 + 1) % SS] : tr[idx % D], dfrag[s2], nd[it], 0, 0, 0);
                    tr[idx % D] = frag_t<OT>(nx < NM ? wimg : wprev, lane, ns2, 32 * nit);
                    constexpr int per = 4 * SS / NM;                                                // 8-byte image stores per MFMA
                    if (!PMB(6)) {
#pragma unroll
                        for (int j = 0; j < per; ++j) {
                            const int k = idx * per + j, s = k >> 2;
                            const bf16x8 src = (k & 2) ? afrag[l][s] : dfrag[s];
                            unsigned char* row = (k & 2) ? rowA : rowD;
                            *reinterpret_cast<bf16x4*>(row + s * U::IMG_SUB + (NNS_PM_LAYOUT == 2 ? 8 : 16) * (k & 1)) = (k & 1) ? __builtin_shufflevector(src, src, 4, 5, 6, 7) : __builtin_shufflevector(src, src, 0, 1, 2, 3);
                        }
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (!PMB(6)) __builtin_amdgcn_sched_group_barrier(0x200, per, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (kOvl && idx >= SS) {
                        // nd[0] is complete: a quarter of its mask / pack step per MFMA of nd[1] (the old delta fragments are still operands and
                        // image rows of this layer: the new ones go to registers of their own)
                        if (l > 0) {
                            constexpr int pc = idx - SS, fs = pc >> 1, q = pc & 1;
                            const i32x4v m = __builtin_bit_cast(i32x4v, afrag[l][fs]);
                            dn[fs][2 * q] = mask2(pack2<false>(nd[0][8 * fs + 4 * q], nd[0][8 * fs + 4 * q + 1]), m[2 * q]);
                            dn[fs][2 * q + 1] = mask2(pack2<false>(nd[0][8 * fs + 4 * q + 2], nd[0][8 * fs + 4 * q + 3]), m[2 * q + 1]);
                            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                        }
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
                // the mask / convert step does not need the barrier: it runs while the image stores drain and the other waves arrive
                if (l > 0) {
#pragma unroll
                    for (int s = 0; s < SS; ++s) {
                        if (kOvl && s < 2) dfrag[s] = __builtin_bit_cast(bf16x8, dn[s]);
                        else dfrag[s] = PMB(9) ? raw8(nd[s >> 1], 8 * (s & 1)) : PMB(7) ? pack8<false>(nd[s >> 1], 8 * (s & 1)) : pack8_masked(nd[s >> 1], 8 * (s & 1), afrag[l][s]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#if NNS_PMB_TIMING
                if (timed) { __builtin_amdgcn_s_waitcnt(0xc07f); tl1 = clock64(); tbody += tl1 - tl0; }
#endif
                if (!PMB(2)) __syncthreads();                          // layer l's images are written: over to the gradient waves
#if NNS_PMB_TIMING
                if (timed) tbar += clock64() - tl1;
#endif
                if (l == 0 && ok) {
                    store_acc<OT, SMALLIO>(gx + (size_t)b * cin0 * P + p, (size_t)P, cin0, h, nd);
                }
            }
        }
        // With an EVEN number of layers the next super-tile's first images (layer nl-1, odd set) never meet the set layer 0 is still being
        // read from (even), and its second ones are written after a barrier the gradient waves only reach once they are done with layer 0:
        // no barrier between super-tiles, the chain waves run ahead into the next forward recompute.  Odd nl: both layers share a set.
        if (nl & 1) __syncthreads();
#if NNS_PMB_TIMING
        if (timed) tk[3] = clock64();
#endif
    }
#if NNS_PMB_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("split backward, one super-tile of wave 0 (cycles): forward recompute %ld, gy convert %ld, backward walk %ld (MFMA loops + mask / convert %ld, barrier waits %ld), whole %ld\n",
               (long)(tk[1] - tk[0]), (long)(tk[2] - tk[1]), (long)(tk[3] - tk[2]), tbody, tbar, (long)(tk[3] - tk[0]));
#endif
}

// (Round 3 also built a TWO-TILE chain on v_mfma_f32_16x16x32_bf16 -- two 16-pixel tiles per chain wave so that one tile's conversion hides
// under the other's MFMAs, in two forms -- correct and slower, 1.47 - 1.51 against 1.01 ms: profiles/r03_ab_pixel_mlp_bwd.log, DESIGN.md 7.2;
// the code is in the history at the commit "MLP backward: gy requested before the forward recompute ...".)
// ------------------------------------------------------------------------------------------------------------------
// float32-operand backward (widths <= 32: e.g. BASELINE config 2's depth-4 width-32 stack), v_mfma_f32_32x32x2_f32.
// Same structure as the bf16 kernel with OT = 1, but nothing is rounded: the accumulator registers ARE the next
// product's B operand (k-step i of a 32x32x2 MFMA takes channels {c(i), c(i) + 4}, c(i) = (i&3) + 8 (i>>2), i.e. exactly
// register i of the two lane halves), each layer's input is kept as its 16 accumulator registers, and because a lane
// supplies ONE element per MFMA the pixel-contraction operands are plain 4-byte LDS reads (no transpose instruction).
// The single 32x32 gW block's K range (128 pixels) is split over the four waves.
// ------------------------------------------------------------------------------------------------------------------
struct BwdLdsF32 {
    static constexpr int CH = 32;
    static constexpr int ROWF = CH + 1;                       // floats per row: +1 keeps column reads conflict-free
    static constexpr int W_BYTES = CH * ROWF * 4;
    static constexpr int B_BYTES = CH * 4;
    static constexpr int IMG_BYTES = 128 * ROWF * 4;
    __host__ __device__ static int total(int nl) { return nl * (W_BYTES + B_BYTES) + 2 * IMG_BYTES; }
};

template <bool SMALLIO>
__global__ __launch_bounds__(256) void pixel_mlp_bwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                 const float* __restrict__ W, const float* __restrict__ Bv,
                                                                 float* __restrict__ gx, float* __restrict__ ws,
                                                                 long npix_total, int P, PixelMlpDesc d, int nparams_w, int nparams) {
    using U = BwdLdsF32;
    constexpr int ROWF = U::ROWF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nl = d.nlayers;
    for (int e = threadIdx.x; e < U::total(nl) / 4; e += 256) reinterpret_cast<float*>(lds)[e] = 0.f;
    __syncthreads();
    for (int l = 0; l < nl; ++l) {
        const int cin = d.cin[l], cout = d.cout[l], n = cin * cout;
        const float* Wl = W + d.woff[l];
        float* dst = reinterpret_cast<float*>(lds + l * U::W_BYTES);
        for (int e = threadIdx.x; e < n; e += 256) dst[(e / cin) * ROWF + e % cin] = Wl[e];
        float* bl = reinterpret_cast<float*>(lds + nl * U::W_BYTES + l * U::B_BYTES);
        for (int e = threadIdx.x; e < cout; e += 256) bl[e] = Bv[d.boff[l] + e];
    }
    __syncthreads();
    const unsigned char* bias0 = lds + nl * U::W_BYTES;
    float* imgD = reinterpret_cast<float*>(lds + nl * (U::W_BYTES + U::B_BYTES));
    float* imgA = imgD + 128 * ROWF;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, r = lane & 31, h = lane >> 5;
    const int cin0 = d.cin[0], coutL = d.cout[nl - 1];
    f32x16 gw[kMaxLayers];
    float gbp[kMaxLayers];
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        gbp[l] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) gw[l][i] = 0.f;
    }
    const long nsuper = (npix_total + 127) / 128;
    for (long sup = blockIdx.x; sup < nsuper; sup += gridDim.x) {
        const long gp = sup * 128 + wave * 32 + r;
        const bool ok = gp < npix_total;
        const long gc = ok ? gp : npix_total - 1;
        const long b = gc / P, p = gc % P;
        // ---------------- forward: ain[l] = input of layer l as accumulator-layout registers
        f32x16 ain[kMaxLayers];
        {
            f32x16 t[1];
            load_acc<1, SMALLIO>(x + (size_t)b * cin0 * P + p, (size_t)P, cin0, ok, h, t);
            ain[0] = t[0];
        }
#pragma unroll
        for (int l = 0; l + 1 < kMaxLayers; ++l) {
            if (l + 1 < nl) {
                __builtin_amdgcn_sched_barrier(0);
                const float* wimg = reinterpret_cast<const float*>(lds + l * U::W_BYTES);
                const float* bl = reinterpret_cast<const float*>(bias0 + l * U::B_BYTES);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = bl[acc_row(i, h)];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wimg[r * ROWF + acc_row(i, h)], ain[l][i], acc, 0, 0, 0);     // A = W[out r][in c(i) + 4h]
#pragma unroll
                for (int i = 0; i < 16; ++i) ain[l + 1][i] = fmaxf(acc[i], 0.f);
            }
        }
        // ---------------- backward
        f32x16 dl;
        {
            f32x16 t[1];
            load_acc<1, SMALLIO>(gy + (size_t)b * coutL * P + p, (size_t)P, coutL, ok, h, t);
            dl = t[0];
        }
#pragma unroll
        for (int l = kMaxLayers - 1; l >= 0; --l) {
            if (l < nl) {
                __builtin_amdgcn_sched_barrier(0);
                const float* wimg = reinterpret_cast<const float*>(lds + l * U::W_BYTES);
                f32x16 nd;
#pragma unroll
                for (int i = 0; i < 16; ++i) nd[i] = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    nd = __builtin_amdgcn_mfma_f32_32x32x2f32(wimg[acc_row(i, h) * ROWF + r], dl[i], nd, 0, 0, 0);           // A = W^T[in r][out c(i) + 4h]
                __builtin_amdgcn_sched_barrier(0);
                {   // delta_l and a_{l-1} as rows [32 wave + r] of the [pix][ch] images (register i = channel c(i) + 4h)
                    float* rowD = imgD + (32 * wave + r) * ROWF;
                    float* rowA = imgA + (32 * wave + r) * ROWF;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { rowD[acc_row(i, h)] = dl[i]; rowA[acc_row(i, h)] = ain[l][i]; }
                }
                __syncthreads();
                // this wave's share of the pixel contraction: pixels 32 wave .. 32 wave + 31, two per MFMA
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const int pix = 32 * wave + 2 * kk + h;
                    const float fa = imgD[pix * ROWF + r], fb = imgA[pix * ROWF + r];
                    gw[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, gw[l], 0, 0, 0);
                    gbp[l] += fa;
                }
                __syncthreads();
                if (l > 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) dl[i] = ain[l][i] > 0.f ? nd[i] : 0.f;
                } else if (ok) {
                    f32x16 t[1]; t[0] = nd;
                    store_acc<1, SMALLIO>(gx + (size_t)b * cin0 * P + p, (size_t)P, cin0, h, t);
                }
            }
        }
    }
    // per-wave workspace slices (the block's K range is split over the waves)
    float* wsb = ws + (size_t)(blockIdx.x * 4 + wave) * nparams;
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        if (l < nl) {
            const int cin = d.cin[l], cout = d.cout[l];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int out = acc_row(i, h);
                if (out < cout && r < cin) wsb[d.woff[l] + out * cin + r] = gw[l][i];
            }
            const float tot = gbp[l] + __shfl_xor(gbp[l], 32);
            if (h == 0 && r < cout) wsb[nparams_w + d.boff[l] + r] = tot;
        }
    }
}

// Sum of the per-workgroup gradient slices in a FIXED order (deterministic): a workgroup owns 64 parameters; thread (part q, parameter i) adds
// slices q, q + 16, q + 32, ... (loads of a round independent of each other: in flight together), then the 16 partial sums are added in order
// of q.  (Round 3: one thread per parameter walking all 256 slices one dependent load at a time took 62 us for 34 MB.)
constexpr int kRedParts = 16, kRedParams = 64;
__global__ __launch_bounds__(kRedParts * kRedParams) void pixel_mlp_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gW, float* __restrict__ gB,
                                                                               int nslices, int nparams_w, int nparams) {
    __shared__ float part[kRedParts][kRedParams];
    const int il = threadIdx.x % kRedParams, q = threadIdx.x / kRedParams;
    const int i = blockIdx.x * kRedParams + il;
    float acc = 0.f;
    if (i < nparams) {
        int k = q;
        for (; k + 3 * kRedParts < nslices; k += 4 * kRedParts) {
            const float a0 = ws[(size_t)k * nparams + i], a1 = ws[(size_t)(k + kRedParts) * nparams + i];
            const float a2 = ws[(size_t)(k + 2 * kRedParts) * nparams + i], a3 = ws[(size_t)(k + 3 * kRedParts) * nparams + i];
            acc = (((acc + a0) + a1) + a2) + a3;
        }
        for (; k < nslices; k += kRedParts) acc += ws[(size_t)k * nparams + i];
    }
    part[q][il] = acc;
    __syncthreads();
    if (q == 0 && i < nparams) {
        float t = part[0][il];
#pragma unroll
        for (int j = 1; j < kRedParts; ++j) t += part[j][il];
        if (i < nparams_w) gW[i] = t; else gB[i - nparams_w] = t;
    }
}

constexpr int kBwdMaxBlocks = 256;

int build_bwd_desc(const int* widths_host, int nlayers, PixelMlpDesc& d, int& nparams_w, int& nparams, int& maxw) {
    if (nlayers < 1 || nlayers > kMaxLayers) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_bwd: %d layers (1..%d supported)", nlayers, kMaxLayers);
    d.nlayers = nlayers;
    int woff = 0, boff = 0;
    maxw = 0;
    for (int l = 0; l < kMaxLayers; ++l) { d.cin[l] = d.cout[l] = 1; d.woff[l] = d.boff[l] = d.lds_off[l] = d.lds_bias[l] = 0; }
    for (int l = 0; l < nlayers; ++l) {
        const int cin = widths_host[l], cout = widths_host[l + 1];
        if (cin < 1 || cout < 1 || cin > kMaxWidth || cout > kMaxWidth) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_bwd: layer %d is %d -> %d (widths 1..%d supported)", l, cin, cout, kMaxWidth);
        d.cin[l] = cin; d.cout[l] = cout; d.woff[l] = woff; d.boff[l] = boff;
        woff += cin * cout; boff += cout;
        maxw = cin > maxw ? cin : maxw; maxw = cout > maxw ? cout : maxw;
    }
    nparams_w = woff; nparams = woff + boff;
    return NNS_OK;
}

template <bool SMALLIO>
int launch_bwd_f32(const float* x, const float* gy, const float* weights, const float* biases, float* gx, float* gW, float* gB,
                   long npix, int P, const PixelMlpDesc& d, int nparams_w, int nparams, float* ws, hipStream_t s) {
    const int lds = BwdLdsF32::total(d.nlayers);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_mlp_bwd_f32_kernel<SMALLIO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "pixel_mlp_bwd: hipFuncSetAttribute(%d B): %s", lds, hipGetErrorString(e));
    const long nsuper = (npix + 127) / 128;
    const int blocks = (int)(nsuper < kBwdMaxBlocks ? nsuper : kBwdMaxBlocks);
    hipLaunchKernelGGL((pixel_mlp_bwd_f32_kernel<SMALLIO>), dim3(blocks), dim3(256), lds, s, x, gy, weights, biases, gx, ws, npix, P, d, nparams_w, nparams);
    if (int rc = check_launch("pixel_mlp_bwd")) return rc;
    hipLaunchKernelGGL(pixel_mlp_reduce_kernel, dim3((nparams + kRedParams - 1) / kRedParams), dim3(kRedParts * kRedParams), 0, s, ws, gW, gB, blocks * 4, nparams_w, nparams);
    return check_launch("pixel_mlp_reduce");
}

template <int OT, bool SMALLIO>
int launch_bwd_uniform(const float* x, const float* gy, const float* weights, const float* biases, float* gx, float* gW, float* gB,
                       long npix, int P, const PixelMlpDesc& d, int nparams_w, int nparams, float* ws, hipStream_t s) {
    const int lds = BwdLds<OT>::total(d.nlayers);
    if (lds > 160 * 1024) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_bwd: needs %d B of LDS (> 160 KiB)", lds);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_mlp_bwd_uniform_kernel<OT, SMALLIO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "pixel_mlp_bwd: hipFuncSetAttribute(%d B): %s", lds, hipGetErrorString(e));
    const long nsuper = (npix + 127) / 128;
    const int blocks = (int)(nsuper < kBwdMaxBlocks ? nsuper : kBwdMaxBlocks);
#ifndef NNS_PM_SPLIT
#define NNS_PM_SPLIT 1             // 1: pixel_mlp_bwd_split_kernel (4 chain waves + 4 gradient waves), 0: pixel_mlp_bwd_uniform_kernel
#endif
    if (NNS_PM_SPLIT && BwdLds<OT>::total(d.nlayers) == lds && NNS_PM_IMGSETS == 2) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_mlp_bwd_split_kernel<OT, SMALLIO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "pixel_mlp_bwd: hipFuncSetAttribute(%d B): %s", lds, hipGetErrorString(e));
        hipLaunchKernelGGL((pixel_mlp_bwd_split_kernel<OT, SMALLIO>), dim3(blocks), dim3(512), lds, s, x, gy, weights, biases, gx, ws, npix, P, d, nparams_w, nparams);
    } else
    hipLaunchKernelGGL((pixel_mlp_bwd_uniform_kernel<OT, SMALLIO>), dim3(blocks), dim3(256), lds, s, x, gy, weights, biases, gx, ws, npix, P, d, nparams_w, nparams);
    if (int rc = check_launch("pixel_mlp_bwd")) return rc;
    const int nslices = blocks * (OT == 2 ? 1 : 4);
    hipLaunchKernelGGL(pixel_mlp_reduce_kernel, dim3((nparams + kRedParams - 1) / kRedParams), dim3(kRedParts * kRedParams), 0, s, ws, gW, gB, nslices, nparams_w, nparams);
    return check_launch("pixel_mlp_reduce");
}

}  // namespace

// x [mb, C_in, P], y [mb, C_out, P] (NCHW with P = nx*ny, as the reference's Conv2d stack); weights packed layer after
// layer in torch layout [C_out_l][C_in_l] (a 1x1 Conv2d weight squeezed), biases packed likewise; widths[0..nlayers]
// = C_in, hidden..., C_out.  ReLU between layers, none after the last (spectral_ode.py:106-116).
// bf16 != 0: weights and inter-layer activations rounded to bfloat16, float32 accumulation (config 3).
NNS_API int nns_pixel_mlp_fwd_f32(const float* x, const float* weights, const float* biases, float* y, int mb, int P,
                                  const int* widths_host, int nlayers, int bf16, void* stream) {
    if (!x || !weights || !biases || !y || !widths_host || mb < 1 || P < 1) return fail(NNS_ERR_INVALID_ARG, "pixel_mlp_fwd: bad args");
    if (nlayers < 1 || nlayers > kMaxLayers) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_fwd: %d layers (1..%d supported)", nlayers, kMaxLayers);
    PixelMlpDesc d;
    d.nlayers = nlayers;
    int woff = 0, boff = 0, lds = 0;
    for (int l = 0; l < kMaxLayers; ++l) { d.cin[l] = d.cout[l] = 1; d.woff[l] = d.boff[l] = d.lds_off[l] = d.lds_bias[l] = 0; }
    for (int l = 0; l < nlayers; ++l) {
        const int cin = widths_host[l], cout = widths_host[l + 1];
        if (cin < 1 || cout < 1 || cin > kMaxWidth || cout > kMaxWidth) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_fwd: layer %d is %d -> %d (widths 1..%d supported)", l, cin, cout, kMaxWidth);
        d.cin[l] = cin; d.cout[l] = cout; d.woff[l] = woff; d.boff[l] = boff;
        woff += cin * cout; boff += cout;
        const int ots = (cout + 31) / 32;
        d.lds_off[l] = lds;
        lds += bf16 ? ots * ((cin + 15) / 16) * 64 * 16 : ots * ((cin + 31) / 32) * 16 * 64 * 4;
        d.lds_bias[l] = lds;
        lds += ots * 32 * 4;
    }
    if (!bf16 && lds > 160 * 1024) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_fwd: weights need %d B of LDS (> 160 KiB)", lds);
    const long npix = (long)mb * P;
    const long ntiles = (npix + 31) / 32;
    long blocks = (ntiles + kGenWaves - 1) / kGenWaves;
    {   // persistent: the weights are staged once per workgroup (128 KB at depth 8, width 64: a workgroup per CU)
        int cus = 256, dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        const long cap = (long)cus * (lds > 80 * 1024 ? 1 : 2);
        if (blocks > cap) blocks = cap;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    if (bf16) {
        int maxw = 0;
        for (int l = 0; l <= nlayers; ++l) maxw = widths_host[l] > maxw ? widths_host[l] : maxw;
        const bool small = widths_host[0] <= 4 && widths_host[nlayers] <= 4;          // (u, v, p)-sized input and output: straight-line tile I/O
        if (maxw <= 32) return small ? launch_fwd_uniform<1, true>(x, weights, biases, y, npix, P, d, s) : launch_fwd_uniform<1, false>(x, weights, biases, y, npix, P, d, s);
        if (NNS_PM_PIPE && small) return launch_fwd_pipe4(x, weights, biases, y, npix, P, d, s);             // (generic I/O keeps the kernel above)
        return small ? launch_fwd_uniform<2, true>(x, weights, biases, y, npix, P, d, s) : launch_fwd_uniform<2, false>(x, weights, biases, y, npix, P, d, s);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_mlp_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "pixel_mlp_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(pixel_mlp_fwd_kernel<false>, dim3((unsigned)blocks), dim3(kGenThreads), lds, s, x, weights, biases, y, npix, P, d);
    }
    return check_launch("pixel_mlp_fwd");
}

// Backward of nns_pixel_mlp_fwd_f32 (see the kernel comments).  gy [mb, C_out, P] in; gx [mb, C_in, P], gW (packed like
// weights) and gB (packed like biases) out -- overwritten, not accumulated.  bf16 != 0: bf16 operands, any supported shape;
// bf16 == 0: float32 operands, widths <= 32 (wider stacks fail with NNS_ERR_UNSUPPORTED).
NNS_API int nns_pixel_mlp_bwd_workspace(const int* widths_host, int nlayers, size_t* bytes) {
    if (!widths_host || !bytes) return fail(NNS_ERR_INVALID_ARG, "pixel_mlp_bwd_workspace: bad args");
    PixelMlpDesc d; int nw, np, maxw;
    if (int rc = build_bwd_desc(widths_host, nlayers, d, nw, np, maxw)) return rc;
    *bytes = (size_t)kBwdMaxBlocks * (maxw <= 32 ? 4 : 1) * np * sizeof(float);
    return NNS_OK;
}

NNS_API int nns_pixel_mlp_bwd_f32(const float* x, const float* gy, const float* weights, const float* biases,
                                  float* gx, float* gW, float* gB, int mb, int P, const int* widths_host, int nlayers, int bf16,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !gy || !weights || !biases || !gx || !gW || !gB || !widths_host || !workspace || mb < 1 || P < 1)
        return fail(NNS_ERR_INVALID_ARG, "pixel_mlp_bwd: bad args");
    PixelMlpDesc d; int nw, np, maxw;
    if (int rc = build_bwd_desc(widths_host, nlayers, d, nw, np, maxw)) return rc;
    if (!bf16 && maxw > 32) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_bwd: the float32-operand backward supports widths <= 32 (got %d); use bf16", maxw);
    if (workspace_bytes < (size_t)kBwdMaxBlocks * (maxw <= 32 ? 4 : 1) * np * sizeof(float))
        return fail(NNS_ERR_WORKSPACE, "pixel_mlp_bwd: workspace too small (%zu B, see nns_pixel_mlp_bwd_workspace)", workspace_bytes);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long npix = (long)mb * P;
    float* ws = static_cast<float*>(workspace);
    const bool small = widths_host[0] <= 4 && widths_host[nlayers] <= 4;
    if (!bf16)
        return small ? launch_bwd_f32<true>(x, gy, weights, biases, gx, gW, gB, npix, P, d, nw, np, ws, s)
                     : launch_bwd_f32<false>(x, gy, weights, biases, gx, gW, gB, npix, P, d, nw, np, ws, s);
    if (maxw <= 32)
        return small ? launch_bwd_uniform<1, true>(x, gy, weights, biases, gx, gW, gB, npix, P, d, nw, np, ws, s)
                     : launch_bwd_uniform<1, false>(x, gy, weights, biases, gx, gW, gB, npix, P, d, nw, np, ws, s);
    return small ? launch_bwd_uniform<2, true>(x, gy, weights, biases, gx, gW, gB, npix, P, d, nw, np, ws, s)
                 : launch_bwd_uniform<2, false>(x, gy, weights, biases, gx, gW, gB, npix, P, d, nw, np, ws, s);
}
