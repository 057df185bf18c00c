"""CPU stand-ins (tests only) for the compute back-ends of nns.slab: the oracle's NumPy operators behind the same
protocol as HipCompute / HipSorCompute / HipChorinCompute, so the gloo tests exercise the decomposition logic (index
ranges, message layouts, posting order, device-side stopping) against the single-process oracle."""
import numpy as np
import torch

from oracle import periodic as OP


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class OracleLines(object):
    def gather_lines(self, fields, msg, nouter, outer_stride, line_off, length, elem_stride=1):
        m = msg.view(len(fields), nouter, length)
        for f, t in enumerate(fields):
            m[f].copy_(torch.as_strided(t, (nouter, length), (outer_stride, elem_stride), t.storage_offset() + line_off))
        return msg

    def scatter_lines(self, msg, fields, nouter, outer_stride, line_off, length, elem_stride=1):
        m = msg.view(len(fields), nouter, length)
        for f, t in enumerate(fields):
            torch.as_strided(t, (nouter, length), (outer_stride, elem_stride), t.storage_offset() + line_off).copy_(m[f])
        return fields


class OracleCompute(OracleLines):
    """Stand-in for nns.slab.HipCompute."""
    fused_dtypes = (torch.float32, torch.float64)

    def transpose_pack(self, fields, send, P):
        nyl = fields[0].shape[2] // P
        for d in range(P):
            for f, t in enumerate(fields):
                send[d, f].copy_(t[:, :, d * nyl:(d + 1) * nyl])
        return send

    def transpose_unpack(self, recv, fields, P):
        nyl = fields[0].shape[2] // P
        for s in range(P):
            for f, t in enumerate(fields):
                t[:, :, s * nyl:(s + 1) * nyl].copy_(recv[s, f])
        return fields

    def pack_halo(self, fields, send, first, last, g0, P):
        """The fused pack + halo launch of HipCompute (nns_slab_pack_halo_*): grids [g0, g0 + Bc) into the send buffer, the edge rows of ALL grids
        into the two halo messages."""
        self.transpose_pack([t[g0:g0 + send.shape[2]] for t in fields], send, P)
        if first is not None:
            for f, t in enumerate(fields):
                first[f].copy_(t[:, 0]); last[f].copy_(t[:, -1])
        return send

    @staticmethod
    def _rows_of(got):
        """[src][3][B][nloc][nyl] (the return all-to-all's receive buffer) -> the three partial fields as row slabs [B][nloc][P * nyl]: what the
        segmented row passes read in place."""
        P, _, B, nloc, nyl = got.shape
        return [got[:, f].permute(1, 2, 0, 3).reshape(B, nloc, P * nyl).clone() for f in range(3)]

    def spec_ypass_seg(self, u, v, p, up, vp, got, out, dt, Ly, rho, nu, precise):
        r = OP.spectral_ypart(*[t.numpy() for t in (u, v, p, up, vp)], *[t.numpy() for t in self._rows_of(got)], dt, Ly, rho, nu)
        for o, a in zip(out, r):
            o.copy_(_t(a))
        return out

    def both_rowpass_halo_seg(self, u, v, p, up, vp, top, bot, got, dt, dx, Ly, rho, nu, precise, out_fd, out_spec, halo_grid0=0):
        g = slice(halo_grid0, halo_grid0 + u.shape[0])               # this call's grids of the whole-batch halo messages
        self.fd_residual_halo(u, v, p, up, vp, top[:, g], bot[:, g], dt, dx, Ly / u.shape[2], rho, nu, 5, None, out_fd)
        return out_fd, self.spec_ypass_seg(u, v, p, up, vp, got, out_spec, dt, Ly, rho, nu, precise)

    def fd_residual_halo(self, u, v, p, up, vp, top, bot, dt, dx, dy, rho, nu, stencil, rows, out):
        pad = lambda f, k: np.concatenate([top[k].numpy()[:, None, :], f.numpy(), bot[k].numpy()[:, None, :]], axis=1)
        same = lambda f: np.concatenate([f.numpy()[:, :1], f.numpy(), f.numpy()[:, -1:]], axis=1)      # halo values of *_prev are never used
        r = OP.fd_residual(pad(u, 0), pad(v, 1), pad(p, 2), same(up), same(vp), dt, dx, dy, rho, nu, stencil)
        r0, r1 = rows if rows is not None else (0, u.shape[1])
        for o, a in zip(out, r):
            o[:, r0:r1].copy_(_t(a[:, r0 + 1:r1 + 1]))
        return out

    def spec_xpass_seg(self, recv, send, B, nx, nyl, seg_rows, Lx, rho, nu, precise):
        P = recv.shape[0]
        cols = recv.permute(1, 2, 0, 3, 4).reshape(3, B, nx, nyl).numpy()
        parts = np.stack(OP.spectral_xpart(cols[0], cols[1], cols[2], Lx, rho, nu))
        send.copy_(_t(parts).reshape(3, B, P, seg_rows, nyl).permute(2, 0, 1, 3, 4))
        return send

    def spec_ypass(self, u, v, p, up, vp, ru, rv, rd, dt, Ly, rho, nu, precise):
        r = OP.spectral_ypart(*[t.numpy() for t in (u, v, p, up, vp, ru, rv, rd)], dt, Ly, rho, nu)
        for o, a in zip((ru, rv, rd), r):                    # in place, as nns_spec_residual_ypass_f32 finishes the partials
            o.copy_(_t(a))
        return ru, rv, rd

    def both_rowpass_halo(self, u, v, p, up, vp, top, bot, partials, dt, dx, Ly, rho, nu, precise, out_fd=None, halo_grid0=0):
        fd = out_fd if out_fd is not None else tuple(torch.empty_like(u) for _ in range(3))
        g = slice(halo_grid0, halo_grid0 + u.shape[0])               # this call's grids of the whole-batch halo messages
        top, bot = top[:, g], bot[:, g]
        self.fd_residual_halo(u, v, p, up, vp, top, bot, dt, dx, Ly / u.shape[2], rho, nu, 5, None, fd)
        return fd, self.spec_ypass(u, v, p, up, vp, *partials, dt, Ly, rho, nu, precise)


class OracleSor(OracleLines):
    """Stand-in for HipSorCompute: one colour of oracle.chorin_fd.sor_sweep_redblack on the slab, with the device-side gate of
    nns_fd_sor_redblack_halfsweep_gated_* (runs only if the previous sweep's err > tol; a skipped colour-1 half-sweep marks
    its slot NaN)."""

    def halfsweep_gated(self, p, C, err, prev_err, tol, gi0, colour, dx, dy, beta):
        if not (float(prev_err[0]) > tol):
            if colour == 1:
                err[0] = float('nan')
            return err
        a = p.numpy()
        nxl, ny = a.shape
        I, J = np.meshgrid(np.arange(1, nxl - 1), np.arange(1, ny - 1), indexing='ij')
        m = ((I + gi0 + J) % 2) == colour
        i, j = I[m], J[m]
        c = C.numpy()
        new = (beta * (dy**2 * a[i + 1, j] + dy**2 * a[i - 1, j] + dx**2 * a[i, j + 1] + dx**2 * a[i, j - 1] - c[i, j]) / (2 * dx**2 + 2 * dy**2)
               + (1 - beta) * a[i, j])
        if new.size:
            err[0] = max(float(err[0]), float(np.max(np.abs(new - a[i, j]))))
        a[i, j] = new
        return err


class OracleChorin(OracleSor):
    """Stand-in for HipChorinCompute: the oracle's operators applied to the slab arrays."""

    def predictor(self, un, vn, un1, vn1, dt, dx, dy, nu, corrected):
        from oracle import chorin_fd as O
        f = O.explicit_predictor_corrected if corrected else O.explicit_predictor
        ui, vi = f(un.numpy(), vn.numpy(), un1.numpy(), vn1.numpy(), dt, dx, dy, nu)
        return torch.from_numpy(ui), torch.from_numpy(vi)

    def predictor_adi(self, un, vn, un1, vn1, dt, dx, dy, nu):
        from oracle import chorin_fd as O
        ui, vi = O.semi_implicit_predictor(un.numpy(), vn.numpy(), un1.numpy(), vn1.numpy(), dt, dx, dy, nu, column_slab=True)
        return torch.from_numpy(ui), torch.from_numpy(vi)

    def bc_apply_(self, A, bcs):
        from oracle.boundary import apply_bc_list
        apply_bc_list(A.numpy(), bcs)
        return A

    def rhs(self, ui, vi, dt, dx, dy, rho):
        from oracle import chorin_fd as O
        return torch.from_numpy(O.pressure_rhs(ui.numpy(), vi.numpy(), dt, dx, dy, rho))

    def correction(self, ui, vi, p, dt, dx, dy):
        from oracle import chorin_fd as O
        u, v = O.correction(ui.numpy(), vi.numpy(), p.numpy(), dt, dx, dy)
        return torch.from_numpy(u), torch.from_numpy(v)
