// Periodic-box Navier-Stokes residual, finite-difference back-end (5- and 9-point), gfx950.
// Operator definition: oracle/periodic.py (no reference symbol exists).
//
//   r_u = (u-u_prev)/dt + u u_x + v u_y + p_x/rho - nu lap u,  r_v likewise,  r_div = u_x + v_y
//
// HBM-bound: compulsory traffic is 5 fields in + 3 out = 8T B/pt (32 B/pt in fp32) against
// ~60 FLOP/pt.  Design for that bound:
//   * every global access is a 16-byte vector per lane (float4 / double2), a wave moves 1 KiB
//     contiguous per instruction;
//   * a workgroup owns a band of R rows x (256 lanes x V columns) and MARCHES down the rows
//     keeping rows i-1, i, i+1 of u, v, p in registers (rolling window): each input row is
//     loaded once per band, halo overhead (R+2)/R on 3 of the 8 streams;
//   * the j+-1 neighbours come from the adjacent lane by cross-lane shuffle, only the two edge
//     lanes of a wave issue a (predicated) scalar load; periodic wrap is index arithmetic;
//   * bands of one grid are laid out so that consecutive bands run on the same XCD (xcd_remap):
//     the two halo rows of a band are L2 hits in that XCD, not extra HBM reads;
//   * R is chosen on the host so that the launch has >= ~2048 workgroups when the problem
//     allows (256 CUs x 8 resident blocks).
#include "nns_common.h"

using namespace nns;

namespace {

template <typename T> struct VecT;
template <> struct VecT<float> { using type = float4; using native = __attribute__((ext_vector_type(4))) float; static constexpr int V = 4; };
template <> struct VecT<double> { using type = double2; using native = __attribute__((ext_vector_type(2))) double; static constexpr int V = 2; };

// Second differences (and the 9-point cross term) cancel O(1) values down to O(h^2): in float32 the
// rounding of those sums, multiplied by 1/h^2 ~ 3e4 at 1024^2, is what limits the residual to ~2e-5
// rel-L2.  They are therefore accumulated in float64 (exact for float32 inputs) -- ~25 fp64 ops per
// point, free under the HBM bound.  First differences and products stay in the field type.
template <typename T>
struct ResK { T inv_dt, inv_2dx, inv_2dy, inv_rho, nu; double inv_dx2, inv_dy2, c9; };

template <typename T>
inline ResK<T> make_resk(double dt, double dx, double dy, double rho, double nu) {
    ResK<T> k;
    k.inv_dt = (T)(1.0 / dt); k.inv_2dx = (T)(1.0 / (2 * dx)); k.inv_2dy = (T)(1.0 / (2 * dy));
    k.inv_dx2 = 1.0 / (dx * dx); k.inv_dy2 = 1.0 / (dy * dy); k.inv_rho = (T)(1.0 / rho); k.nu = (T)nu;
    k.c9 = (dx * dx + dy * dy) / 12.0 / (dx * dx * dy * dy);
    return k;
}

template <typename T, int V>
struct Row { T v[V]; T l, r; };          // V consecutive columns of one row + the two neighbours

// Row slabs of a grid sharded over ranks (nns/slab.py): rows -1 and nx of the local slab are the neighbour ranks' edge rows,
// delivered as [u, v, p][grid][ny] messages (fstride = grids * ny).  NULL = the rows wrap around inside the local grid.
// [r0, r1) = the local rows this launch evaluates (interior rows first, the two edge rows once the halos have arrived).
template <typename T>
struct HaloRows { const T* top; const T* bot; long fstride; int r0, r1; };
template <typename T>
__device__ __forceinline__ const T* halo_row(const T* grid_base, const HaloRows<T>& hr, int field, int b, int i, int nx, int ny) {
    if (i < 0) return hr.top ? hr.top + field * hr.fstride + (size_t)b * ny : grid_base + (size_t)(nx - 1) * ny;
    if (i >= nx) return hr.bot ? hr.bot + field * hr.fstride + (size_t)b * ny : grid_base;
    return grid_base + (size_t)i * ny;
}

#ifndef NNS_FD_NT
#define NNS_FD_NT 0        // 1: non-temporal hint on the write-once outputs and read-once u_prev / v_prev streams
#endif
#ifndef NNS_FD_RMAX
#define NNS_FD_RMAX 32     // tallest band a workgroup marches down
#endif
template <typename T>
__device__ __forceinline__ void load_vec(const T* p, T (&out)[VecT<T>::V], bool nt = false) {
    using VT = typename VecT<T>::type;
    if (NNS_FD_NT && nt) {
        using NV = typename VecT<T>::native;
        const NV n = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
#pragma unroll
        for (int e = 0; e < VecT<T>::V; ++e) out[e] = n[e];
        return;
    }
    const VT t = *reinterpret_cast<const VT*>(p);
    if constexpr (VecT<T>::V == 4) { out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w; }
    else { out[0] = t.x; out[1] = t.y; }
}

template <typename T>
__device__ __forceinline__ void store_vec(T* p, const T (&in)[VecT<T>::V]) {
    using VT = typename VecT<T>::type;
    VT t;
    if constexpr (VecT<T>::V == 4) { t.x = in[0]; t.y = in[1]; t.z = in[2]; t.w = in[3]; }
    else { t.x = in[0]; t.y = in[1]; }
    if (NNS_FD_NT) {
        using NV = typename VecT<T>::native;
        NV n;
#pragma unroll
        for (int e = 0; e < VecT<T>::V; ++e) n[e] = in[e];
        __builtin_nontemporal_store(n, reinterpret_cast<NV*>(p));
    } else *reinterpret_cast<VT*>(p) = t;
}

// One row of one field: the lane's vector, plus left/right neighbours taken from the adjacent
// lanes; lanes whose neighbour lives in another wave (or wraps around the box) load it.
template <typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ rowp, int j0, int jl, int jr, bool need_l, bool need_r,
                                         Row<T, VecT<T>::V>& R) {
    constexpr int V = VecT<T>::V;
    load_vec<T>(rowp + j0, R.v);
    T l = __shfl_up(R.v[V - 1], 1);
    T r = __shfl_down(R.v[0], 1);
    if (need_l) l = rowp[jl];
    if (need_r) r = rowp[jr];
    R.l = l; R.r = r;
}

template <typename T, int STENCIL>
__global__ __launch_bounds__(256) void fd_residual_vec_kernel(const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ p,
                                                               const T* __restrict__ up, const T* __restrict__ vp,
                                                               T* __restrict__ ru, T* __restrict__ rv, T* __restrict__ rd,
                                                               int nx, int ny, int R, int nbands, int nstrips, ResK<T> k, HaloRows<T> hr) {
    constexpr int V = VecT<T>::V;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = lb % nstrips, band = (lb / nstrips) % nbands, b = lb / (nstrips * nbands);
    const int nvec = ny / V;
    const int jv_raw = strip * 256 + (int)threadIdx.x;
    const bool valid = jv_raw < nvec;
    const int jv = valid ? jv_raw : nvec - 1;
    const int j0 = jv * V;
    const int jl = j0 == 0 ? ny - 1 : j0 - 1;
    const int jr = j0 + V == ny ? 0 : j0 + V;
    const int lane = threadIdx.x & (kWave - 1);
    const bool need_l = lane == 0;
    const bool need_r = lane == kWave - 1 || jv_raw >= nvec - 1;
    const size_t g = (size_t)b * nx * ny;
    const T* ug = u + g; const T* vg = v + g; const T* pg = p + g;
    const int i0 = hr.r0 + band * R, i1 = min(hr.r1, i0 + R);
    if (i0 >= hr.r1) return;

    Row<T, V> um, uc, un, vm, vc, vn, pm, pc, pn;
    {
        const size_t rc = (size_t)i0 * ny;
        load_row<T>(halo_row<T>(ug, hr, 0, b, i0 - 1, nx, ny), j0, jl, jr, need_l, need_r, um); load_row<T>(ug + rc, j0, jl, jr, need_l, need_r, uc);
        load_row<T>(halo_row<T>(vg, hr, 1, b, i0 - 1, nx, ny), j0, jl, jr, need_l, need_r, vm); load_row<T>(vg + rc, j0, jl, jr, need_l, need_r, vc);
        load_row<T>(halo_row<T>(pg, hr, 2, b, i0 - 1, nx, ny), j0, jl, jr, need_l, need_r, pm); load_row<T>(pg + rc, j0, jl, jr, need_l, need_r, pc);
    }
    for (int i = i0; i < i1; ++i) {
        const size_t rc = g + (size_t)i * ny + j0;
        load_row<T>(halo_row<T>(ug, hr, 0, b, i + 1, nx, ny), j0, jl, jr, need_l, need_r, un);
        load_row<T>(halo_row<T>(vg, hr, 1, b, i + 1, nx, ny), j0, jl, jr, need_l, need_r, vn);
        load_row<T>(halo_row<T>(pg, hr, 2, b, i + 1, nx, ny), j0, jl, jr, need_l, need_r, pn);
        T upv[V], vpv[V], o_u[V], o_v[V], o_d[V];
        load_vec<T>(up + rc, upv, true);
        load_vec<T>(vp + rc, vpv, true);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T ucc = uc.v[e], vcc = vc.v[e];
            const T ul = e == 0 ? uc.l : uc.v[e > 0 ? e - 1 : 0], ur = e == V - 1 ? uc.r : uc.v[e < V - 1 ? e + 1 : 0];
            const T vl = e == 0 ? vc.l : vc.v[e > 0 ? e - 1 : 0], vr = e == V - 1 ? vc.r : vc.v[e < V - 1 ? e + 1 : 0];
            const T pl = e == 0 ? pc.l : pc.v[e > 0 ? e - 1 : 0], pr = e == V - 1 ? pc.r : pc.v[e < V - 1 ? e + 1 : 0];
            const T ux = (un.v[e] - um.v[e]) * k.inv_2dx, uy = (ur - ul) * k.inv_2dy;
            const T vx = (vn.v[e] - vm.v[e]) * k.inv_2dx, vy = (vr - vl) * k.inv_2dy;
            const T px = (pn.v[e] - pm.v[e]) * k.inv_2dx, py = (pr - pl) * k.inv_2dy;
            double lu = ((double)un.v[e] - 2.0 * ucc + (double)um.v[e]) * k.inv_dx2 + ((double)ur - 2.0 * ucc + (double)ul) * k.inv_dy2;
            double lv = ((double)vn.v[e] - 2.0 * vcc + (double)vm.v[e]) * k.inv_dx2 + ((double)vr - 2.0 * vcc + (double)vl) * k.inv_dy2;
            if constexpr (STENCIL == 9) {
                const T uml = e == 0 ? um.l : um.v[e > 0 ? e - 1 : 0], umr = e == V - 1 ? um.r : um.v[e < V - 1 ? e + 1 : 0];
                const T unl = e == 0 ? un.l : un.v[e > 0 ? e - 1 : 0], unr = e == V - 1 ? un.r : un.v[e < V - 1 ? e + 1 : 0];
                const T vml = e == 0 ? vm.l : vm.v[e > 0 ? e - 1 : 0], vmr = e == V - 1 ? vm.r : vm.v[e < V - 1 ? e + 1 : 0];
                const T vnl = e == 0 ? vn.l : vn.v[e > 0 ? e - 1 : 0], vnr = e == V - 1 ? vn.r : vn.v[e < V - 1 ? e + 1 : 0];
                lu += k.c9 * (((double)uml + umr + unl + unr) - 2.0 * ((double)um.v[e] + un.v[e] + ul + ur) + 4.0 * ucc);
                lv += k.c9 * (((double)vml + vmr + vnl + vnr) - 2.0 * ((double)vm.v[e] + vn.v[e] + vl + vr) + 4.0 * vcc);
            }
            o_u[e] = (ucc - upv[e]) * k.inv_dt + ucc * ux + vcc * uy + px * k.inv_rho - k.nu * (T)lu;
            o_v[e] = (vcc - vpv[e]) * k.inv_dt + ucc * vx + vcc * vy + py * k.inv_rho - k.nu * (T)lv;
            o_d[e] = ux + vy;
        }
        if (valid) { store_vec<T>(ru + rc, o_u); store_vec<T>(rv + rc, o_v); store_vec<T>(rd + rc, o_d); }
        um = uc; uc = un; vm = vc; vc = vn; pm = pc; pc = pn;
    }
}

// Any-size fallback (ny not a multiple of the vector width): one thread per point.
template <typename T, int STENCIL>
__global__ __launch_bounds__(256) void fd_residual_generic_kernel(const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ p,
                                                                   const T* __restrict__ up, const T* __restrict__ vp,
                                                                   T* __restrict__ ru, T* __restrict__ rv, T* __restrict__ rd,
                                                                   int nx, int ny, ResK<T> k, HaloRows<T> hr) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = hr.r0 + blockIdx.y;
    if (j >= ny) return;
    const size_t g = (size_t)blockIdx.z * nx * ny;
    const int im = i - 1, in = i + 1;                       // -1 / nx: the halo rows (or the periodic wrap), see halo_row
    const int jm = j == 0 ? ny - 1 : j - 1, jn = j + 1 == ny ? 0 : j + 1;
    auto at = [&](const T* f, int a, int b) { return halo_row<T>(f + g, hr, f == u ? 0 : f == v ? 1 : 2, (int)blockIdx.z, a, nx, ny)[b]; };
    const T ucc = at(u, i, j), vcc = at(v, i, j);
    const T ue = at(u, in, j), uw = at(u, im, j), ur = at(u, i, jn), ul = at(u, i, jm);
    const T ve = at(v, in, j), vw = at(v, im, j), vr = at(v, i, jn), vl = at(v, i, jm);
    const T ux = (ue - uw) * k.inv_2dx, uy = (ur - ul) * k.inv_2dy;
    const T vx = (ve - vw) * k.inv_2dx, vy = (vr - vl) * k.inv_2dy;
    const T px = (at(p, in, j) - at(p, im, j)) * k.inv_2dx, py = (at(p, i, jn) - at(p, i, jm)) * k.inv_2dy;
    double lu = ((double)ue - 2.0 * ucc + (double)uw) * k.inv_dx2 + ((double)ur - 2.0 * ucc + (double)ul) * k.inv_dy2;
    double lv = ((double)ve - 2.0 * vcc + (double)vw) * k.inv_dx2 + ((double)vr - 2.0 * vcc + (double)vl) * k.inv_dy2;
    if constexpr (STENCIL == 9) {
        lu += k.c9 * (((double)at(u, im, jm) + at(u, im, jn) + at(u, in, jm) + at(u, in, jn)) - 2.0 * ((double)uw + ue + ul + ur) + 4.0 * ucc);
        lv += k.c9 * (((double)at(v, im, jm) + at(v, im, jn) + at(v, in, jm) + at(v, in, jn)) - 2.0 * ((double)vw + ve + vl + vr) + 4.0 * vcc);
    }
    const size_t c = g + (size_t)i * ny + j;
    ru[c] = (ucc - up[c]) * k.inv_dt + ucc * ux + vcc * uy + px * k.inv_rho - k.nu * (T)lu;
    rv[c] = (vcc - vp[c]) * k.inv_dt + ucc * vx + vcc * vy + py * k.inv_rho - k.nu * (T)lv;
    rd[c] = ux + vy;
}

template <typename T>
int fd_residual(const T* u, const T* v, const T* p, const T* up, const T* vp, T* ru, T* rv, T* rd, int batch, int nx, int ny,
                double dt, double dx, double dy, double rho, double nu, int stencil, hipStream_t s,
                const T* halo_top = nullptr, const T* halo_bot = nullptr, int r0 = 0, int r1 = -1) {
    if (!u || !v || !p || !up || !vp || !ru || !rv || !rd || !field_args_ok(batch, nx, ny))
        return fail(NNS_ERR_INVALID_ARG, "fd_residual: bad args (batch=%d nx=%d ny=%d)", batch, nx, ny);
    if (r1 < 0) r1 = nx;
    if (r0 < 0 || r1 > nx || r0 > r1) return fail(NNS_ERR_INVALID_ARG, "fd_residual: row range [%d, %d) not inside [0, %d)", r0, r1, nx);
    if ((halo_top == nullptr) != (halo_bot == nullptr)) return fail(NNS_ERR_INVALID_ARG, "fd_residual: halo_top and halo_bot go together");
    if (r0 == r1) return NNS_OK;
    const HaloRows<T> hr{halo_top, halo_bot, (long)batch * ny, r0, r1};
    const int nr = r1 - r0;
    if (stencil != 5 && stencil != 9) return fail(NNS_ERR_INVALID_ARG, "fd_residual: stencil must be 5 or 9 (got %d)", stencil);
    if (dt == 0 || dx == 0 || dy == 0 || rho == 0) return fail(NNS_ERR_INVALID_ARG, "fd_residual: dt, dx, dy, rho must be non-zero");
    const ResK<T> k = make_resk<T>(dt, dx, dy, rho, nu);
    constexpr int V = VecT<T>::V;
    const bool aligned = ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(p) |
                           reinterpret_cast<uintptr_t>(up) | reinterpret_cast<uintptr_t>(vp) | reinterpret_cast<uintptr_t>(ru) |
                           reinterpret_cast<uintptr_t>(rv) | reinterpret_cast<uintptr_t>(rd)) & 15) == 0;
    if (ny % V == 0 && ny >= 2 * V && aligned) {
        const int nvec = ny / V, nstrips = (nvec + 255) / 256;
        // rows per band: aim for >= 2048 workgroups, keep the halo overhead (R+2)/R small
        long want = 2048;
        int R = (int)(((long)batch * nr * nstrips + want - 1) / want);
        R = R < 4 ? 4 : (R > NNS_FD_RMAX ? NNS_FD_RMAX : R);
        const int nbands = (nr + R - 1) / R;
        const long nblocks = (long)batch * nbands * nstrips;
        if (nblocks > 0x7fffffffL) return fail(NNS_ERR_UNSUPPORTED, "fd_residual: grid too large");
        if (stencil == 5)
            hipLaunchKernelGGL((fd_residual_vec_kernel<T, 5>), dim3((unsigned)nblocks), dim3(256), 0, s, u, v, p, up, vp, ru, rv, rd, nx, ny, R, nbands, nstrips, k, hr);
        else
            hipLaunchKernelGGL((fd_residual_vec_kernel<T, 9>), dim3((unsigned)nblocks), dim3(256), 0, s, u, v, p, up, vp, ru, rv, rd, nx, ny, R, nbands, nstrips, k, hr);
    } else {
        if (nr > 65535 || batch > 65535) return fail(NNS_ERR_UNSUPPORTED, "fd_residual: ragged-size path is limited to 65535 rows and grids per launch");
        const dim3 grid((ny + 255) / 256, nr, batch);
        if (stencil == 5)
            hipLaunchKernelGGL((fd_residual_generic_kernel<T, 5>), grid, dim3(256), 0, s, u, v, p, up, vp, ru, rv, rd, nx, ny, k, hr);
        else
            hipLaunchKernelGGL((fd_residual_generic_kernel<T, 9>), grid, dim3(256), 0, s, u, v, p, up, vp, ru, rv, rd, nx, ny, k, hr);
    }
    return check_launch("fd_residual");
}


// ------------------------------------------------------------------------------------------------------------------
// Backward (vector-Jacobian product) of the FD residual: the adjoint stencils (oracle/periodic.py: residual_vjp).
// With a = g_u, b = g_v, d = g_div (upstream gradients), D = central difference (D^T = -D), L = the 5/9-point
// Laplacian (L^T = L):
//   grad_u = a/dt + a u_x + b v_x - D_x(a u + d) - D_y(a v) - nu L a
//   grad_v = b/dt + a u_y + b v_y - D_x(b u) - D_y(b v + d) - nu L b
//   grad_p = -(D_x a + D_y b)/rho,     grad_u_prev = -a/dt,  grad_v_prev = -b/dt   (optional outputs)
// Same design as the forward: 5 input streams (u, v, a, b, d) in a three-row rolling window, 3..5 output streams,
// 16-byte lanes, j+-1 neighbours by shuffle; 32..40 B/pt of compulsory traffic, HBM-bound.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int V>
__device__ __forceinline__ T left_of(const Row<T, V>& r, int e) { return e == 0 ? r.l : r.v[e > 0 ? e - 1 : 0]; }
template <typename T, int V>
__device__ __forceinline__ T right_of(const Row<T, V>& r, int e) { return e == V - 1 ? r.r : r.v[e < V - 1 ? e + 1 : 0]; }

template <typename T, int STENCIL>
__global__ __launch_bounds__(256) void fd_residual_bwd_vec_kernel(const T* __restrict__ u, const T* __restrict__ v,
                                                                   const T* __restrict__ ga, const T* __restrict__ gb, const T* __restrict__ gd,
                                                                   T* __restrict__ gu, T* __restrict__ gv, T* __restrict__ gp,
                                                                   T* __restrict__ gup, T* __restrict__ gvp,
                                                                   int nx, int ny, int R, int nbands, int nstrips, ResK<T> k) {
    constexpr int V = VecT<T>::V;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = lb % nstrips, band = (lb / nstrips) % nbands, b = lb / (nstrips * nbands);
    const int nvec = ny / V;
    const int jv_raw = strip * 256 + (int)threadIdx.x;
    const bool valid = jv_raw < nvec;
    const int jv = valid ? jv_raw : nvec - 1;
    const int j0 = jv * V;
    const int jl = j0 == 0 ? ny - 1 : j0 - 1;
    const int jr = j0 + V == ny ? 0 : j0 + V;
    const int lane = threadIdx.x & (kWave - 1);
    const bool need_l = lane == 0;
    const bool need_r = lane == kWave - 1 || jv_raw >= nvec - 1;
    const size_t g = (size_t)b * nx * ny;
    const T* ug = u + g; const T* vg = v + g; const T* ag = ga + g; const T* bg = gb + g; const T* dg = gd + g;
    const int i0 = band * R, i1 = min(nx, i0 + R);
    if (i0 >= nx) return;

    // rows i-1, i, i+1 of u, v, a, b;  d only needs its vector on rows i+-1 and its l / r on row i
    Row<T, V> um, uc, un, vm, vc, vn, am, ac, an, bm, bc, bn, dm, dc, dn;
    {
        const size_t rm = (size_t)(i0 == 0 ? nx - 1 : i0 - 1) * ny, rc = (size_t)i0 * ny;
        load_row<T>(ug + rm, j0, jl, jr, need_l, need_r, um); load_row<T>(ug + rc, j0, jl, jr, need_l, need_r, uc);
        load_row<T>(vg + rm, j0, jl, jr, need_l, need_r, vm); load_row<T>(vg + rc, j0, jl, jr, need_l, need_r, vc);
        load_row<T>(ag + rm, j0, jl, jr, need_l, need_r, am); load_row<T>(ag + rc, j0, jl, jr, need_l, need_r, ac);
        load_row<T>(bg + rm, j0, jl, jr, need_l, need_r, bm); load_row<T>(bg + rc, j0, jl, jr, need_l, need_r, bc);
        load_row<T>(dg + rm, j0, jl, jr, need_l, need_r, dm); load_row<T>(dg + rc, j0, jl, jr, need_l, need_r, dc);
    }
    for (int i = i0; i < i1; ++i) {
        const size_t rn = (size_t)(i + 1 == nx ? 0 : i + 1) * ny, rc = g + (size_t)i * ny + j0;
        load_row<T>(ug + rn, j0, jl, jr, need_l, need_r, un);
        load_row<T>(vg + rn, j0, jl, jr, need_l, need_r, vn);
        load_row<T>(ag + rn, j0, jl, jr, need_l, need_r, an);
        load_row<T>(bg + rn, j0, jl, jr, need_l, need_r, bn);
        load_row<T>(dg + rn, j0, jl, jr, need_l, need_r, dn);
        T o_u[V], o_v[V], o_p[V], o_up[V], o_vp[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T a = ac.v[e], bb = bc.v[e];
            const T ul = left_of<T, V>(uc, e), ur = right_of<T, V>(uc, e), vl = left_of<T, V>(vc, e), vr = right_of<T, V>(vc, e);
            const T al = left_of<T, V>(ac, e), ar = right_of<T, V>(ac, e), bl = left_of<T, V>(bc, e), br = right_of<T, V>(bc, e);
            const T dl = left_of<T, V>(dc, e), dr = right_of<T, V>(dc, e);
            const T ux = (un.v[e] - um.v[e]) * k.inv_2dx, uy = (ur - ul) * k.inv_2dy;
            const T vx = (vn.v[e] - vm.v[e]) * k.inv_2dx, vy = (vr - vl) * k.inv_2dy;
            // D_x(a u + d), D_y(a v), D_x(b u), D_y(b v + d), D_x a, D_y b
            const T dx_aud = ((an.v[e] * un.v[e] + dn.v[e]) - (am.v[e] * um.v[e] + dm.v[e])) * k.inv_2dx;
            const T dy_av = (ar * vr - al * vl) * k.inv_2dy;
            const T dx_bu = (bn.v[e] * un.v[e] - bm.v[e] * um.v[e]) * k.inv_2dx;
            const T dy_bvd = ((br * vr + dr) - (bl * vl + dl)) * k.inv_2dy;
            const T dx_a = (an.v[e] - am.v[e]) * k.inv_2dx, dy_b = (br - bl) * k.inv_2dy;
            double la = ((double)an.v[e] - 2.0 * a + (double)am.v[e]) * k.inv_dx2 + ((double)ar - 2.0 * a + (double)al) * k.inv_dy2;
            double lb2 = ((double)bn.v[e] - 2.0 * bb + (double)bm.v[e]) * k.inv_dx2 + ((double)br - 2.0 * bb + (double)bl) * k.inv_dy2;
            if constexpr (STENCIL == 9) {
                la += k.c9 * (((double)left_of<T, V>(am, e) + right_of<T, V>(am, e) + left_of<T, V>(an, e) + right_of<T, V>(an, e))
                              - 2.0 * ((double)am.v[e] + an.v[e] + al + ar) + 4.0 * a);
                lb2 += k.c9 * (((double)left_of<T, V>(bm, e) + right_of<T, V>(bm, e) + left_of<T, V>(bn, e) + right_of<T, V>(bn, e))
                               - 2.0 * ((double)bm.v[e] + bn.v[e] + bl + br) + 4.0 * bb);
            }
            o_u[e] = a * k.inv_dt + a * ux + bb * vx - dx_aud - dy_av - k.nu * (T)la;
            o_v[e] = bb * k.inv_dt + a * uy + bb * vy - dx_bu - dy_bvd - k.nu * (T)lb2;
            o_p[e] = -(dx_a + dy_b) * k.inv_rho;
            o_up[e] = -a * k.inv_dt; o_vp[e] = -bb * k.inv_dt;
        }
        if (valid) {
            store_vec<T>(gu + rc, o_u); store_vec<T>(gv + rc, o_v); store_vec<T>(gp + rc, o_p);
            if (gup) store_vec<T>(gup + rc, o_up);
            if (gvp) store_vec<T>(gvp + rc, o_vp);
        }
        um = uc; uc = un; vm = vc; vc = vn; am = ac; ac = an; bm = bc; bc = bn; dm = dc; dc = dn;
    }
}

template <typename T, int STENCIL>
__global__ __launch_bounds__(256) void fd_residual_bwd_generic_kernel(const T* __restrict__ u, const T* __restrict__ v,
                                                                       const T* __restrict__ ga, const T* __restrict__ gb, const T* __restrict__ gd,
                                                                       T* __restrict__ gu, T* __restrict__ gv, T* __restrict__ gp,
                                                                       T* __restrict__ gup, T* __restrict__ gvp, int nx, int ny, ResK<T> k) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= ny) return;
    const size_t g = (size_t)blockIdx.z * nx * ny;
    const int im = i == 0 ? nx - 1 : i - 1, in = i + 1 == nx ? 0 : i + 1;
    const int jm = j == 0 ? ny - 1 : j - 1, jn = j + 1 == ny ? 0 : j + 1;
    auto at = [&](const T* f, int a, int b) { return f[g + (size_t)a * ny + b]; };
    const T a = at(ga, i, j), bb = at(gb, i, j);
    const T ux = (at(u, in, j) - at(u, im, j)) * k.inv_2dx, uy = (at(u, i, jn) - at(u, i, jm)) * k.inv_2dy;
    const T vx = (at(v, in, j) - at(v, im, j)) * k.inv_2dx, vy = (at(v, i, jn) - at(v, i, jm)) * k.inv_2dy;
    const T dx_aud = ((at(ga, in, j) * at(u, in, j) + at(gd, in, j)) - (at(ga, im, j) * at(u, im, j) + at(gd, im, j))) * k.inv_2dx;
    const T dy_av = (at(ga, i, jn) * at(v, i, jn) - at(ga, i, jm) * at(v, i, jm)) * k.inv_2dy;
    const T dx_bu = (at(gb, in, j) * at(u, in, j) - at(gb, im, j) * at(u, im, j)) * k.inv_2dx;
    const T dy_bvd = ((at(gb, i, jn) * at(v, i, jn) + at(gd, i, jn)) - (at(gb, i, jm) * at(v, i, jm) + at(gd, i, jm))) * k.inv_2dy;
    const T dx_a = (at(ga, in, j) - at(ga, im, j)) * k.inv_2dx, dy_b = (at(gb, i, jn) - at(gb, i, jm)) * k.inv_2dy;
    auto lap = [&](const T* f, T c) {
        double l = ((double)at(f, in, j) - 2.0 * c + (double)at(f, im, j)) * k.inv_dx2 + ((double)at(f, i, jn) - 2.0 * c + (double)at(f, i, jm)) * k.inv_dy2;
        if constexpr (STENCIL == 9)
            l += k.c9 * (((double)at(f, im, jm) + at(f, im, jn) + at(f, in, jm) + at(f, in, jn))
                         - 2.0 * ((double)at(f, im, j) + at(f, in, j) + at(f, i, jm) + at(f, i, jn)) + 4.0 * c);
        return l;
    };
    const size_t c = g + (size_t)i * ny + j;
    gu[c] = a * k.inv_dt + a * ux + bb * vx - dx_aud - dy_av - k.nu * (T)lap(ga, a);
    gv[c] = bb * k.inv_dt + a * uy + bb * vy - dx_bu - dy_bvd - k.nu * (T)lap(gb, bb);
    gp[c] = -(dx_a + dy_b) * k.inv_rho;
    if (gup) gup[c] = -a * k.inv_dt;
    if (gvp) gvp[c] = -bb * k.inv_dt;
}

template <typename T>
int fd_residual_bwd(const T* u, const T* v, const T* ga, const T* gb, const T* gd, T* gu, T* gv, T* gp, T* gup, T* gvp,
                    int batch, int nx, int ny, double dt, double dx, double dy, double rho, double nu, int stencil, hipStream_t s) {
    if (!u || !v || !ga || !gb || !gd || !gu || !gv || !gp || !field_args_ok(batch, nx, ny))
        return fail(NNS_ERR_INVALID_ARG, "fd_residual_bwd: bad args (batch=%d nx=%d ny=%d)", batch, nx, ny);
    if (stencil != 5 && stencil != 9) return fail(NNS_ERR_INVALID_ARG, "fd_residual_bwd: stencil must be 5 or 9 (got %d)", stencil);
    if (dt == 0 || dx == 0 || dy == 0 || rho == 0) return fail(NNS_ERR_INVALID_ARG, "fd_residual_bwd: dt, dx, dy, rho must be non-zero");
    const ResK<T> k = make_resk<T>(dt, dx, dy, rho, nu);
    constexpr int V = VecT<T>::V;
    uintptr_t bits = 0;
    for (const void* q : {(const void*)u, (const void*)v, (const void*)ga, (const void*)gb, (const void*)gd, (const void*)gu, (const void*)gv, (const void*)gp, (const void*)gup, (const void*)gvp})
        bits |= reinterpret_cast<uintptr_t>(q);
    if (ny % V == 0 && ny >= 2 * V && (bits & 15) == 0) {
        const int nvec = ny / V, nstrips = (nvec + 255) / 256;
        long want = 2048;
        int R = (int)(((long)batch * nx * nstrips + want - 1) / want);
        R = R < 4 ? 4 : (R > NNS_FD_RMAX ? NNS_FD_RMAX : R);
        const int nbands = (nx + R - 1) / R;
        const long nblocks = (long)batch * nbands * nstrips;
        if (nblocks > 0x7fffffffL) return fail(NNS_ERR_UNSUPPORTED, "fd_residual_bwd: grid too large");
        if (stencil == 5)
            hipLaunchKernelGGL((fd_residual_bwd_vec_kernel<T, 5>), dim3((unsigned)nblocks), dim3(256), 0, s, u, v, ga, gb, gd, gu, gv, gp, gup, gvp, nx, ny, R, nbands, nstrips, k);
        else
            hipLaunchKernelGGL((fd_residual_bwd_vec_kernel<T, 9>), dim3((unsigned)nblocks), dim3(256), 0, s, u, v, ga, gb, gd, gu, gv, gp, gup, gvp, nx, ny, R, nbands, nstrips, k);
    } else {
        const dim3 grid((ny + 255) / 256, nx, batch);
        if (stencil == 5)
            hipLaunchKernelGGL((fd_residual_bwd_generic_kernel<T, 5>), grid, dim3(256), 0, s, u, v, ga, gb, gd, gu, gv, gp, gup, gvp, nx, ny, k);
        else
            hipLaunchKernelGGL((fd_residual_bwd_generic_kernel<T, 9>), grid, dim3(256), 0, s, u, v, ga, gb, gd, gu, gv, gp, gup, gvp, nx, ny, k);
    }
    return check_launch("fd_residual_bwd");
}

}  // namespace

NNS_API int nns_fd_residual_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                float* r_u, float* r_v, float* r_div, int batch, int nx, int ny, double dt, double dx, double dy,
                                double rho, double nu, int stencil, void* stream) {
    return fd_residual<float>(u, v, p, u_prev, v_prev, r_u, r_v, r_div, batch, nx, ny, dt, dx, dy, rho, nu, stencil, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_residual_f64(const double* u, const double* v, const double* p, const double* u_prev, const double* v_prev,
                                double* r_u, double* r_v, double* r_div, int batch, int nx, int ny, double dt, double dx, double dy,
                                double rho, double nu, int stencil, void* stream) {
    return fd_residual<double>(u, v, p, u_prev, v_prev, r_u, r_v, r_div, batch, nx, ny, dt, dx, dy, rho, nu, stencil, reinterpret_cast<hipStream_t>(stream));
}

NNS_API int nns_fd_residual_halo_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                     const float* halo_top, const float* halo_bot, float* r_u, float* r_v, float* r_div,
                                     int batch, int nx_local, int ny, int row_begin, int row_end, double dt, double dx, double dy,
                                     double rho, double nu, int stencil, void* stream) {
    if (!halo_top || !halo_bot) return fail(NNS_ERR_INVALID_ARG, "fd_residual_halo: halo_top and halo_bot are required");
    return fd_residual<float>(u, v, p, u_prev, v_prev, r_u, r_v, r_div, batch, nx_local, ny, dt, dx, dy, rho, nu, stencil, reinterpret_cast<hipStream_t>(stream),
                              halo_top, halo_bot, row_begin, row_end);
}
NNS_API int nns_fd_residual_halo_f64(const double* u, const double* v, const double* p, const double* u_prev, const double* v_prev,
                                     const double* halo_top, const double* halo_bot, double* r_u, double* r_v, double* r_div,
                                     int batch, int nx_local, int ny, int row_begin, int row_end, double dt, double dx, double dy,
                                     double rho, double nu, int stencil, void* stream) {
    if (!halo_top || !halo_bot) return fail(NNS_ERR_INVALID_ARG, "fd_residual_halo: halo_top and halo_bot are required");
    return fd_residual<double>(u, v, p, u_prev, v_prev, r_u, r_v, r_div, batch, nx_local, ny, dt, dx, dy, rho, nu, stencil, reinterpret_cast<hipStream_t>(stream),
                               halo_top, halo_bot, row_begin, row_end);
}

NNS_API int nns_fd_residual_bwd_f32(const float* u, const float* v, const float* g_u, const float* g_v, const float* g_div,
                                    float* grad_u, float* grad_v, float* grad_p, float* grad_u_prev, float* grad_v_prev,
                                    int batch, int nx, int ny, double dt, double dx, double dy, double rho, double nu, int stencil, void* stream) {
    return fd_residual_bwd<float>(u, v, g_u, g_v, g_div, grad_u, grad_v, grad_p, grad_u_prev, grad_v_prev, batch, nx, ny, dt, dx, dy, rho, nu, stencil, reinterpret_cast<hipStream_t>(stream));
}
NNS_API int nns_fd_residual_bwd_f64(const double* u, const double* v, const double* g_u, const double* g_v, const double* g_div,
                                    double* grad_u, double* grad_v, double* grad_p, double* grad_u_prev, double* grad_v_prev,
                                    int batch, int nx, int ny, double dt, double dx, double dy, double rho, double nu, int stencil, void* stream) {
    return fd_residual_bwd<double>(u, v, g_u, g_v, g_div, grad_u, grad_v, grad_p, grad_u_prev, grad_v_prev, batch, nx, ny, dt, dx, dy, rho, nu, stencil, reinterpret_cast<hipStream_t>(stream));
}
