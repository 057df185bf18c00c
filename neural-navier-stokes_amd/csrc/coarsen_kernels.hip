// Spatial coarsening of field sequences: block means over agg_x x agg_y cells of [T][nx][ny] sequences
// (reference: spatial_coarsen, src/utils.py:13-60 -- data preparation for the neural models; SURVEY.md section 8 (f)
// rank 4).  u, v and p are coarsened by ONE launch.  HBM-bound: reads 3 x 8|4 B per fine point once, writes
// 1/(agg_x agg_y) of that.
//
// Reference semantics kept:
//   * the block of one output cell is flattened row-major ([agg_x][agg_y]) and reduced by numpy.mean, whose float add
//     order is PAIRWISE: for n < 8 a running sum; for 8 <= n <= 128 eight interleaved partial sums r[k] += a[i + k]
//     combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and then the n % 8 tail added in order; longer blocks split
//     recursively at n/2 rounded down to a multiple of 8.  The kernel follows that order exactly (the recursion with an
//     explicit stack), so the float64 result is bit-identical to the reference's;
//   * the reference's column loop runs over ny // agg_x cells (src/utils.py:49, not ny // agg_y): cells with
//     j >= ny // agg_x are left at their initial 0.  `jfill` carries that count; the host side raises the reference's
//     IndexError for jfill > ny / agg_y before any launch.
#include "nns_common.h"

using namespace nns;

namespace {

// sum of the n block elements e0 .. e0 + n - 1 (flattened index e -> row e / ay, column e % ay), numpy's pairwise order
template <typename T>
__device__ __forceinline__ T block_elem(const T* __restrict__ blk, int ny, int ay, int e) {
    return blk[(size_t)(e / ay) * ny + e % ay];
}

template <typename T>
__device__ T pairwise_leaf(const T* __restrict__ blk, int ny, int ay, int e0, int n) {
    if (n < 8) {
        T res = (T)-0.0;                                  // numpy starts the short sum at -0 (keeps an all -0 block)
        for (int i = 0; i < n; ++i) res += block_elem(blk, ny, ay, e0 + i);
        return res;
    }
    T r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = block_elem(blk, ny, ay, e0 + k);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += block_elem(blk, ny, ay, e0 + i + k);
    }
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += block_elem(blk, ny, ay, e0 + i);
    return res;
}

constexpr int kLeaf = 128;            // numpy's PW_BLOCKSIZE
constexpr int kStack = 24;            // depth of the split recursion: n <= 128 * 2^24

template <typename T, bool BIG>
__device__ T pairwise_sum(const T* __restrict__ blk, int ny, int ay, int n) {
    if constexpr (!BIG) return pairwise_leaf(blk, ny, ay, 0, n);
    // post-order walk of the split tree: a frame is (e0, n, state); value stack holds finished left sums
    int fe[kStack], fn[kStack], fs[kStack];
    T fv[kStack];
    int sp = 0;
    fe[0] = 0; fn[0] = n; fs[0] = 0;
    T ret = (T)0;
    while (sp >= 0) {
        const int e0 = fe[sp], m = fn[sp];
        if (m <= kLeaf) { ret = pairwise_leaf(blk, ny, ay, e0, m); --sp; continue; }
        int n2 = m / 2; n2 -= n2 % 8;
        if (fs[sp] == 0) { fs[sp] = 1; ++sp; fe[sp] = e0; fn[sp] = n2; fs[sp] = 0; }
        else if (fs[sp] == 1) { fv[sp] = ret; fs[sp] = 2; ++sp; fe[sp] = e0 + n2; fn[sp] = m - n2; fs[sp] = 0; }
        else { ret = fv[sp] + ret; --sp; }
    }
    return ret;
}

// one thread per output cell; consecutive threads = consecutive j, so a wave reads 64 * ay contiguous elements per row
// BIG: blocks of more than 128 cells (the split recursion's explicit stack lives in scratch: kept out of the usual path)
template <typename T, bool BIG>
__global__ __launch_bounds__(256) void coarsen_kernel(const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ p,
                                                      T* __restrict__ cu, T* __restrict__ cv, T* __restrict__ cp,
                                                      int nx, int ny, int ax, int ay, int cnx, int cny, int jfill) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cny) return;
    const int i = blockIdx.y;
    const size_t t = blockIdx.z;
    const size_t o = (t * cnx + i) * cny + j;
    if (j >= jfill) { cu[o] = (T)0; cv[o] = (T)0; cp[o] = (T)0; return; }
    const size_t b = (t * nx + (size_t)i * ax) * ny + (size_t)j * ay;
    const int n = ax * ay;
    cu[o] = pairwise_sum<T, BIG>(u + b, ny, ay, n) / (T)n;
    cv[o] = pairwise_sum<T, BIG>(v + b, ny, ay, n) / (T)n;
    cp[o] = pairwise_sum<T, BIG>(p + b, ny, ay, n) / (T)n;
}

template <typename T>
int coarsen(const T* u, const T* v, const T* p, T* cu, T* cv, T* cp, int nt, int nx, int ny, int ax, int ay, int jfill, hipStream_t s) {
    if (!u || !v || !p || !cu || !cv || !cp || nt < 1 || nx < 1 || ny < 1 || ax < 1 || ay < 1)
        return fail(NNS_ERR_INVALID_ARG, "coarsen: bad args (nt=%d nx=%d ny=%d agg=%dx%d)", nt, nx, ny, ax, ay);
    if (nx % ax || ny % ay) return fail(NNS_ERR_INVALID_ARG, "coarsen: nx=%d, ny=%d must be multiples of agg_x=%d, agg_y=%d", nx, ny, ax, ay);
    const int cnx = nx / ax, cny = ny / ay;
    if (jfill < 0 || jfill > cny) return fail(NNS_ERR_INVALID_ARG, "coarsen: jfill=%d outside [0, %d]", jfill, cny);
    if (nt > 65535 || cnx > 65535) return fail(NNS_ERR_UNSUPPORTED, "coarsen: nt=%d / nx/agg_x=%d exceed the launch grid (65535)", nt, cnx);
    if ((long)ax * ay > (long)kLeaf << (kStack - 2)) return fail(NNS_ERR_UNSUPPORTED, "coarsen: block of %ld cells is too large", (long)ax * ay);
    const dim3 grid((cny + 255) / 256, cnx, nt);
    if (ax * ay <= kLeaf) hipLaunchKernelGGL((coarsen_kernel<T, false>), grid, dim3(256), 0, s, u, v, p, cu, cv, cp, nx, ny, ax, ay, cnx, cny, jfill);
    else hipLaunchKernelGGL((coarsen_kernel<T, true>), grid, dim3(256), 0, s, u, v, p, cu, cv, cp, nx, ny, ax, ay, cnx, cny, jfill);
    return check_launch("coarsen");
}

}  // namespace

NNS_API int nns_coarsen_f32(const float* u, const float* v, const float* p, float* cu, float* cv, float* cp, int nt, int nx, int ny,
                            int agg_x, int agg_y, int jfill, void* stream) {
    return coarsen<float>(u, v, p, cu, cv, cp, nt, nx, ny, agg_x, agg_y, jfill, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_coarsen_f64(const double* u, const double* v, const double* p, double* cu, double* cv, double* cp, int nt, int nx, int ny,
                            int agg_x, int agg_y, int jfill, void* stream) {
    return coarsen<double>(u, v, p, cu, cv, cp, nt, nx, ny, agg_x, agg_y, jfill, reinterpret_cast<hipStream_t>(stream));
}
