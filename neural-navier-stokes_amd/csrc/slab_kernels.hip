// Device-side packing for the slab decomposition of ONE grid over the GPUs of a node (nns/slab.py; SURVEY.md
// section 8 (e): halo rows for the stencils, an all-to-all transpose for the spectral column pass).  The reference has
// no multi-device path (src/neural_spectral/spectral_ode.py:155-156: one device); the north star is the spec.
//
// All four kernels are pure copies (HBM-bound, 8 B moved per element); what they buy is that a halo exchange or a
// transpose costs ONE launch on each side of the collective instead of torch stack / permute / contiguous / cat passes:
//   * gather_lines / scatter_lines: line `line_off + e * elem_stride`, e < len, of every outer block of up to 4 fields
//     <-> one contiguous message [field][outer][len].  A row of a [B, nloc, ny] slab is (outer = B, stride 1); a column of
//     a [nx, nyl] slab is (outer = 1, stride nyl).
//   * transpose_pack: row slabs [F][B][nloc][ny] -> the all-to-all send buffer [dest][F][B][nloc][ny/P] (the column block of
//     every destination contiguous); transpose_unpack: the received [src][F][B][nloc][ny/P] -> row slabs.
//     (The column pass itself reads and writes the [src|dest][F][B][nloc][ny/P] layout in place: segmented rows,
//     nns_spec_residual_xpass_seg_f32.)
#include "nns_common.h"
#include <cstdint>

using namespace nns;

namespace {

template <typename T> struct Ptr4 { const T* p[4]; };
template <typename T> struct MPtr4 { T* p[4]; };

template <typename T>
__global__ __launch_bounds__(256) void gather_lines_kernel(Ptr4<T> src, T* __restrict__ dst, long nouter, long outer_stride, long line_off,
                                                           long len, long elem_stride) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= len) return;
    const long o = blockIdx.y, f = blockIdx.z;
    dst[(f * nouter + o) * len + e] = src.p[f][o * outer_stride + line_off + e * elem_stride];
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_lines_kernel(const T* __restrict__ src, MPtr4<T> dst, long nouter, long outer_stride, long line_off,
                                                            long len, long elem_stride) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= len) return;
    const long o = blockIdx.y, f = blockIdx.z;
    dst.p[f][o * outer_stride + line_off + e * elem_stride] = src[(f * nouter + o) * len + e];
}

template <typename T>
int lines(bool gather, const T* const* fields, int nfields, T* buf, long nouter, long outer_stride, long line_off, long len, long elem_stride,
          hipStream_t s) {
    if (!fields || !buf || nfields < 1 || nfields > 4 || nouter < 1 || nouter > 65535 || len < 1 || line_off < 0 || elem_stride < 1 || outer_stride < 0)
        return fail(NNS_ERR_INVALID_ARG, "slab lines: bad args (nfields=%d nouter=%ld len=%ld)", nfields, nouter, len);
    for (int f = 0; f < nfields; ++f)
        if (!fields[f]) return fail(NNS_ERR_INVALID_ARG, "slab lines: field %d is NULL", f);
    const dim3 grid((unsigned)((len + 255) / 256), (unsigned)nouter, (unsigned)nfields);
    if (gather) {
        Ptr4<T> q{};
        for (int f = 0; f < nfields; ++f) q.p[f] = fields[f];
        hipLaunchKernelGGL(gather_lines_kernel<T>, grid, dim3(256), 0, s, q, buf, nouter, outer_stride, line_off, len, elem_stride);
    } else {
        MPtr4<T> q{};
        for (int f = 0; f < nfields; ++f) q.p[f] = const_cast<T*>(fields[f]);
        hipLaunchKernelGGL(scatter_lines_kernel<T>, grid, dim3(256), 0, s, buf, q, nouter, outer_stride, line_off, len, elem_stride);
    }
    return check_launch(gather ? "slab_gather_lines" : "slab_scatter_lines");
}

// One thread per element of a row-slab field; consecutive threads = consecutive columns, so both sides move
// contiguous runs of ny/P elements (512 B at 1024 / 8 in float32).
template <typename T, bool PACK>
__global__ __launch_bounds__(256) void transpose_kernel(Ptr4<T> rows_in, MPtr4<T> rows_out, T* __restrict__ buf, long B, long nloc, long ny, long P) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= ny) return;
    const long bi = blockIdx.y;                     // b * nloc + i
    const long f = blockIdx.z, F = gridDim.z;
    const long nyl = ny / P, d = j / nyl, jj = j - d * nyl;
    const long row = bi * ny + j;                                            // [B][nloc][ny]
    const long pk = ((d * F + f) * B * nloc + bi) * nyl + jj;                // [P][F][B][nloc][nyl]
    if constexpr (PACK) buf[pk] = rows_in.p[f][row]; else rows_out.p[f][row] = buf[pk];
}

// The same copy, 16 bytes per thread and several row-lines per workgroup (round 3: one 4-byte element per thread and one 1-KB row piece per
// workgroup moved the transposes of the 1024^2 x 64 slab at 2.9 TB/s -- 786 k tiny workgroups).  VW = elements per 16-byte vector; needs
// ny / P a multiple of VW and 16-byte aligned fields and buffer (checked on the host).
template <typename T, bool PACK>
__global__ __launch_bounds__(256) void transpose_vec_kernel(Ptr4<T> rows_in, MPtr4<T> rows_out, T* __restrict__ buf, long B, long nloc, long ny, long P, long nlines) {
    constexpr int VW = 16 / (int)sizeof(T);
    using V = __attribute__((ext_vector_type(VW))) T;
    const long f = blockIdx.z, F = gridDim.z;
    const long nyl = ny / P, vpr = ny / VW;                                  // vectors per row-line
    const long total = nlines * vpr;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long bi = e / vpr, j = (e - bi * vpr) * VW;
        const long d = j / nyl, jj = j - d * nyl;
        const long row = bi * ny + j;
        const long pk = ((d * F + f) * B * nloc + bi) * nyl + jj;
        if constexpr (PACK) *reinterpret_cast<V*>(buf + pk) = *reinterpret_cast<const V*>(rows_in.p[f] + row);
        else *reinterpret_cast<V*>(rows_out.p[f] + row) = *reinterpret_cast<const V*>(buf + pk);
    }
}

// transpose_pack of the batch chunk [g0, g0 + Bc) AND the halo messages of the WHOLE local batch in one launch (round 4: the slab step's
// two gather_lines launches folded into the first chunk's pack): the trailing workgroups of the grid copy row 0 of every grid into
// first[f][b][ny] and row nloc - 1 into last[f][b][ny].
template <typename T>
__global__ __launch_bounds__(256) void pack_halo_vec_kernel(Ptr4<T> rows_in, T* __restrict__ buf, T* __restrict__ first, T* __restrict__ last,
                                                            long Btot, long g0, long Bc, long nloc, long ny, long P, unsigned pack_blocks) {
    constexpr int VW = 16 / (int)sizeof(T);
    using V = __attribute__((ext_vector_type(VW))) T;
    const long f = blockIdx.z, F = gridDim.z;
    const long vpr = ny / VW;
    if (blockIdx.x < pack_blocks) {
        const long nyl = ny / P, nlines = Bc * nloc, total = nlines * vpr;
        const T* src = rows_in.p[f] + g0 * nloc * ny;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)pack_blocks * 256) {
            const long bi = e / vpr, j = (e - bi * vpr) * VW;
            const long d = j / nyl, jj = j - d * nyl;
            *reinterpret_cast<V*>(buf + ((d * F + f) * Bc * nloc + bi) * nyl + jj) = *reinterpret_cast<const V*>(src + bi * ny + j);
        }
    } else {
        const long total = 2 * Btot * vpr, hb = gridDim.x - pack_blocks;
        for (long e = (long)(blockIdx.x - pack_blocks) * 256 + threadIdx.x; e < total; e += hb * 256) {
            const long which = e / (Btot * vpr), r = e - which * Btot * vpr;
            const long b = r / vpr, j = (r - b * vpr) * VW;
            T* dst = (which ? last : first) + (f * Btot + b) * ny + j;
            *reinterpret_cast<V*>(dst) = *reinterpret_cast<const V*>(rows_in.p[f] + (b * nloc + (which ? nloc - 1 : 0)) * ny + j);
        }
    }
}

template <typename T>
int pack_halo(const T* const* fields, int nfields, T* send, T* first, T* last, int Btot, int g0, int Bc, int nloc, int ny, int P, hipStream_t s) {
    constexpr int VW = 16 / (int)sizeof(T);
    if (!fields || !send || nfields < 1 || nfields > 4 || Btot < 1 || g0 < 0 || Bc < 1 || g0 + Bc > Btot || nloc < 1 || ny < 1 || P < 1 || ny % P || (first == nullptr) != (last == nullptr))
        return fail(NNS_ERR_INVALID_ARG, "slab pack_halo: bad args (nfields=%d Btot=%d g0=%d Bc=%d nloc=%d ny=%d P=%d)", nfields, Btot, g0, Bc, nloc, ny, P);
    bool vec = (ny / P) % VW == 0 && reinterpret_cast<uintptr_t>(send) % 16 == 0 && reinterpret_cast<uintptr_t>(first) % 16 == 0 && reinterpret_cast<uintptr_t>(last) % 16 == 0;
    for (int f = 0; f < nfields; ++f) {
        if (!fields[f]) return fail(NNS_ERR_INVALID_ARG, "slab pack_halo: field %d is NULL", f);
        vec = vec && reinterpret_cast<uintptr_t>(fields[f]) % 16 == 0;
    }
    if (!vec) return fail(NNS_ERR_UNSUPPORTED, "slab pack_halo: needs 16-byte aligned fields / buffers and ny / P a multiple of %d (use the separate pack and gather calls)", VW);
    Ptr4<T> in{};
    for (int f = 0; f < nfields; ++f) in.p[f] = fields[f];
    const long total = (long)Bc * nloc * (ny / VW);
    long pb = (total + 255) / 256; if (pb > 16384) pb = 16384;
    long hb = first ? (2L * Btot * (ny / VW) + 255) / 256 : 0; if (hb > 1024) hb = 1024;
    hipLaunchKernelGGL(pack_halo_vec_kernel<T>, dim3((unsigned)(pb + hb), 1, (unsigned)nfields), dim3(256), 0, s, in, send, first, last,
                       (long)Btot, (long)g0, (long)Bc, (long)nloc, (long)ny, (long)P, (unsigned)pb);
    return check_launch("slab_pack_halo");
}

template <typename T>
int transpose(bool pack, const T* const* fields, int nfields, T* buf, int B, int nloc, int ny, int P, hipStream_t s) {
    if (!fields || !buf || nfields < 1 || nfields > 4 || B < 1 || nloc < 1 || ny < 1 || P < 1 || ny % P || (long)B * nloc > 0x7fffffffL)
        return fail(NNS_ERR_INVALID_ARG, "slab transpose: bad args (nfields=%d B=%d nloc=%d ny=%d P=%d)", nfields, B, nloc, ny, P);
    for (int f = 0; f < nfields; ++f)
        if (!fields[f]) return fail(NNS_ERR_INVALID_ARG, "slab transpose: field %d is NULL", f);
    const long bi = (long)B * nloc;
    if (bi > 65535L * 32768L) return fail(NNS_ERR_UNSUPPORTED, "slab transpose: slab too tall");
    {   // vector path
        constexpr int VW = 16 / (int)sizeof(T);
        bool vec = (ny / P) % VW == 0 && reinterpret_cast<uintptr_t>(buf) % 16 == 0;
        for (int f = 0; f < nfields; ++f) vec = vec && reinterpret_cast<uintptr_t>(fields[f]) % 16 == 0;
        if (vec) {
            Ptr4<T> in{}; MPtr4<T> out{};
            for (int f = 0; f < nfields; ++f) { in.p[f] = fields[f]; out.p[f] = const_cast<T*>(fields[f]); }
            const long total = bi * (ny / VW);
            long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;       // grid-stride: ~64 workgroups per CU and field
            const dim3 grid((unsigned)blocks, 1, (unsigned)nfields);
            if (pack) hipLaunchKernelGGL((transpose_vec_kernel<T, true>), grid, dim3(256), 0, s, in, out, buf, (long)B, (long)nloc, (long)ny, (long)P, bi);
            else hipLaunchKernelGGL((transpose_vec_kernel<T, false>), grid, dim3(256), 0, s, in, out, buf, (long)B, (long)nloc, (long)ny, (long)P, bi);
            return check_launch(pack ? "slab_transpose_pack" : "slab_transpose_unpack");
        }
    }
    // blockIdx.y is limited to 65535: fold the excess of B * nloc into blockIdx.x strides is not needed below 65535 rows;
    // taller slabs go in chunks of 65535 row-lines
    for (long r0 = 0; r0 < bi; r0 += 65535) {
        const long nr = bi - r0 < 65535 ? bi - r0 : 65535;
        Ptr4<T> in{}; MPtr4<T> out{};
        for (int f = 0; f < nfields; ++f) { in.p[f] = fields[f] + r0 * ny; out.p[f] = const_cast<T*>(fields[f]) + r0 * ny; }
        const dim3 grid((unsigned)((ny + 255) / 256), (unsigned)nr, (unsigned)nfields);
        // the packed buffer is indexed by (b * nloc + i) too: shift its base by r0 lines of nyl elements
        T* bshift = buf + r0 * (ny / P);
        if (pack) hipLaunchKernelGGL((transpose_kernel<T, true>), grid, dim3(256), 0, s, in, out, bshift, (long)B, (long)nloc, (long)ny, (long)P);
        else hipLaunchKernelGGL((transpose_kernel<T, false>), grid, dim3(256), 0, s, in, out, bshift, (long)B, (long)nloc, (long)ny, (long)P);
    }
    return check_launch(pack ? "slab_transpose_pack" : "slab_transpose_unpack");
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

NNS_API int nns_slab_gather_lines_f32(const float* const* fields_host, int nfields, float* msg, long nouter, long outer_stride, long line_off,
                                      long len, long elem_stride, void* stream) {
    return lines<float>(true, fields_host, nfields, msg, nouter, outer_stride, line_off, len, elem_stride, S(stream));
}
NNS_API int nns_slab_gather_lines_f64(const double* const* fields_host, int nfields, double* msg, long nouter, long outer_stride, long line_off,
                                      long len, long elem_stride, void* stream) {
    return lines<double>(true, fields_host, nfields, msg, nouter, outer_stride, line_off, len, elem_stride, S(stream));
}
NNS_API int nns_slab_scatter_lines_f32(const float* msg, float* const* fields_host, int nfields, long nouter, long outer_stride, long line_off,
                                       long len, long elem_stride, void* stream) {
    return lines<float>(false, fields_host, nfields, const_cast<float*>(msg), nouter, outer_stride, line_off, len, elem_stride, S(stream));
}
NNS_API int nns_slab_scatter_lines_f64(const double* msg, double* const* fields_host, int nfields, long nouter, long outer_stride, long line_off,
                                       long len, long elem_stride, void* stream) {
    return lines<double>(false, fields_host, nfields, const_cast<double*>(msg), nouter, outer_stride, line_off, len, elem_stride, S(stream));
}
NNS_API int nns_slab_transpose_pack_f32(const float* const* fields_host, int nfields, float* send, int batch, int nloc, int ny, int nranks, void* stream) {
    return transpose<float>(true, fields_host, nfields, send, batch, nloc, ny, nranks, S(stream));
}
NNS_API int nns_slab_transpose_pack_f64(const double* const* fields_host, int nfields, double* send, int batch, int nloc, int ny, int nranks, void* stream) {
    return transpose<double>(true, fields_host, nfields, send, batch, nloc, ny, nranks, S(stream));
}
NNS_API int nns_slab_pack_halo_f32(const float* const* fields_host, int nfields, float* send, float* first, float* last, int batch_total, int grid0, int batch,
                                   int nloc, int ny, int nranks, void* stream) {
    return pack_halo<float>(fields_host, nfields, send, first, last, batch_total, grid0, batch, nloc, ny, nranks, S(stream));
}
NNS_API int nns_slab_pack_halo_f64(const double* const* fields_host, int nfields, double* send, double* first, double* last, int batch_total, int grid0, int batch,
                                   int nloc, int ny, int nranks, void* stream) {
    return pack_halo<double>(fields_host, nfields, send, first, last, batch_total, grid0, batch, nloc, ny, nranks, S(stream));
}
NNS_API int nns_slab_transpose_unpack_f32(const float* recv, float* const* fields_host, int nfields, int batch, int nloc, int ny, int nranks, void* stream) {
    return transpose<float>(false, fields_host, nfields, const_cast<float*>(recv), batch, nloc, ny, nranks, S(stream));
}
NNS_API int nns_slab_transpose_unpack_f64(const double* recv, double* const* fields_host, int nfields, int batch, int nloc, int ny, int nranks, void* stream) {
    return transpose<double>(false, fields_host, nfields, const_cast<double*>(recv), batch, nloc, ny, nranks, S(stream));
}
