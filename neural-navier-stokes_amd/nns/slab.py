"""Slab decomposition of ONE periodic grid across the GPUs of a node.

Rank r owns rows [r*nx/P, (r+1)*nx/P) of every field ([B, nx/P, ny] local tensors).  One process per
GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

  * FD residual: one nearest-neighbour exchange -- each rank sends its first / last row of u, v, p to
    its ring neighbours (periodic wrap closes the ring; xGMI is point-to-point, so this uses exactly the
    two direct links to the neighbours).  The three fields are packed into ONE message per direction.
    The stencil kernel then runs on the halo-padded slab and the two halo rows of output are dropped.
  * spectral residual: the y-pass (rows) is local.  The x-pass needs complete columns: ONE all-to-all
    transposes u, v, p (packed) to column slabs [B, nx, ny/P], the x-pass kernel runs there, and ONE
    all-to-all brings the three partial fields back -- 2 collectives per evaluation instead of 10 for a
    library-style 2-D FFT per derivative.  An all-to-all uses all 7 xGMI links of a GPU concurrently.
  * SOR in the reference's lexicographic order does not shard (sequential fronts): replicas only.  The opt-in
    RED-BLACK order does (SlabPressure below): one halo exchange per half-sweep and one all-reduce(max) per sweep.

The compute callables default to the HIP ops; the CPU tests inject oracle-based ones to check the
decomposition logic (index ranges, packing, wrap-around) against the single-process result.
"""
import math

import torch
import torch.distributed as dist


class HipCompute(object):
    """The product's compute back-end: hand-written HIP kernels through the C ABI (nns.ops)."""

    def fd_residual(self, u, v, p, up, vp, dt, dx, dy, rho, nu, stencil):
        from . import ops
        return ops.fd_residual(u, v, p, up, vp, dt, dx, dy, rho, nu, stencil)

    def spec_xpass(self, u, v, p, Lx, rho, nu, precise):
        from . import ops
        return ops.spec_residual_xpass(u, v, p, Lx, rho, nu, precise)

    def spec_ypass(self, u, v, p, up, vp, ru, rv, rd, dt, Ly, rho, nu, precise):
        from . import ops
        return ops.spec_residual_ypass_(u, v, p, up, vp, ru, rv, rd, dt, Ly, rho, nu, precise)


class SlabResidual(object):
    def __init__(self, nx, ny, dt, rho, nu, Lx=2 * math.pi, Ly=2 * math.pi, group=None, compute=None, precise=True):
        self.group = group
        self.P = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if nx % self.P or ny % self.P:
            raise ValueError("nx and ny must be divisible by the number of ranks (%d x %d over %d)" % (nx, ny, self.P))
        self.nx, self.ny, self.nloc, self.nyloc = nx, ny, nx // self.P, ny // self.P
        self.dt, self.rho, self.nu, self.Lx, self.Ly = dt, rho, nu, Lx, Ly
        self.dx, self.dy = Lx / nx, Ly / ny
        self.compute = compute if compute is not None else HipCompute()
        self.precise = precise

    # ------------------------------------------------------------------ FD: halo rows
    def exchange_halo(self, fields):
        """fields: list of [B, nloc, ny] tensors.  Returns them padded to [B, nloc+2, ny] with the periodic
        neighbours' edge rows (one packed message per direction)."""
        P, r = self.P, self.rank
        up_rank, down_rank = (r - 1) % P, (r + 1) % P            # "up" owns the rows before ours
        first = torch.stack([f[:, 0, :] for f in fields]).contiguous()       # goes to up_rank (their bottom halo)
        last = torch.stack([f[:, -1, :] for f in fields]).contiguous()       # goes to down_rank (their top halo)
        top_halo, bot_halo = torch.empty_like(last), torch.empty_like(first)
        if P == 1:
            top_halo.copy_(last), bot_halo.copy_(first)
        else:
            # With P == 2 both neighbours are the same peer and messages between a pair are matched in posting
            # order: the peer sends (its first row -> our bottom halo, its last row -> our top halo), so the
            # receives are posted in that order.
            ops = [dist.P2POp(dist.isend, first, up_rank, self.group), dist.P2POp(dist.isend, last, down_rank, self.group),
                   dist.P2POp(dist.irecv, bot_halo, down_rank, self.group), dist.P2POp(dist.irecv, top_halo, up_rank, self.group)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        out = []
        for i, f in enumerate(fields):
            out.append(torch.cat([top_halo[i][:, None, :], f, bot_halo[i][:, None, :]], dim=1).contiguous())
        return out

    def fd(self, u, v, p, u_prev, v_prev, stencil=5):
        pu, pv, pp = self.exchange_halo([u, v, p])
        pad = lambda f: torch.cat([f[:, :1], f, f[:, -1:]], dim=1).contiguous()      # halo values of *_prev are never used
        r = self.compute.fd_residual(pu, pv, pp, pad(u_prev), pad(v_prev), self.dt, self.dx, self.dy, self.rho, self.nu, stencil)
        return tuple(t[:, 1:-1, :].contiguous() for t in r)

    # ------------------------------------------------------------------ spectral: all-to-all transpose
    def _to_columns(self, fields):
        """list of F [B, nloc, ny] row slabs -> [F, B, nx, nyloc] column slab (one all-to-all)."""
        P = self.P
        x = torch.stack(fields)                                                       # [F, B, nloc, ny]
        F, B = x.shape[0], x.shape[1]
        send = x.reshape(F, B, self.nloc, P, self.nyloc).permute(3, 0, 1, 2, 4).contiguous()   # [P(dest), F, B, nloc, nyloc]
        recv = torch.empty_like(send)
        if P == 1:
            recv.copy_(send)
        else:
            dist.all_to_all_single(recv, send, group=self.group)
        # recv[src] holds rows of rank src: concatenate along x
        return recv.permute(1, 2, 0, 3, 4).reshape(F, B, self.nx, self.nyloc).contiguous()

    def _to_rows(self, cols):
        """[F, B, nx, nyloc] column slab -> list of F [B, nloc, ny] row slabs (one all-to-all)."""
        P = self.P
        F, B = cols.shape[0], cols.shape[1]
        send = cols.reshape(F, B, P, self.nloc, self.nyloc).permute(2, 0, 1, 3, 4).contiguous()  # [P(dest), F, B, nloc, nyloc]
        recv = torch.empty_like(send)
        if P == 1:
            recv.copy_(send)
        else:
            dist.all_to_all_single(recv, send, group=self.group)
        x = recv.permute(1, 2, 3, 0, 4).reshape(F, B, self.nloc, self.ny)                        # columns of rank src side by side
        return [x[i].contiguous() for i in range(F)]

    def spectral(self, u, v, p, u_prev, v_prev):
        cols = self._to_columns([u, v, p])
        pu, pv, pd = self.compute.spec_xpass(cols[0], cols[1], cols[2], self.Lx, self.rho, self.nu, self.precise)
        ru, rv, rd = self._to_rows(torch.stack([pu, pv, pd]))
        return self.compute.spec_ypass(u, v, p, u_prev, v_prev, ru, rv, rd, self.dt, self.Ly, self.rho, self.nu, self.precise)

    def both(self, u, v, p, u_prev, v_prev, stencil=5):
        return self.fd(u, v, p, u_prev, v_prev, stencil), self.spectral(u, v, p, u_prev, v_prev)


class HipSorCompute(object):
    """Half-sweep back-end of SlabPressure: the HIP kernel nns_fd_sor_redblack_halfsweep_*."""

    def halfsweep(self, p, C, err, gi0, colour, dx, dy, beta):
        from . import ops
        return ops.fd_sor_redblack_halfsweep_(p, C, err, gi0, colour, dx, dy, beta)

    def err_value(self, err):
        return err          # the kernel max-accumulates the IEEE bit pattern of a non-negative value: it reads back as that value


class SlabPressure(object):
    """Red-black SOR for the pressure Poisson problem of chorin_fd on a NON-periodic [nx, ny] grid, rows sharded over
    ranks ("opt-in red-black shards like Jacobi; global err needs all-reduce(max)").

    Rank r owns interior-or-boundary rows [lo, hi) of the global grid; its working slab is those rows plus one halo row
    on each side that has a neighbour (the physical boundary rows 0 and nx-1 belong to the first / last rank and are
    never updated, exactly as in the single-process solver).  Per half-sweep: refresh the halo rows from the ring
    neighbours (no wrap: the domain is not periodic), relax one colour; per sweep: all-reduce(max) of the local
    max|p - pPrev|.  Same formula, relaxation factor, stopping rule and sweep cap as nns_fd_sor_redblack; the result is
    bitwise the single-process red-black solve (a half-sweep only reads the other colour)."""

    def __init__(self, nx, ny, dx, dy, beta, tol=5e-6, group=None, compute=None, axis=0):
        """axis = 0: row slabs (the description above).  axis = 1: COLUMN slabs -- columns [lo, hi) of ny with halo
        columns; "rows" below then reads "columns" (the halo lines are strided, so they are packed for the exchange)."""
        assert axis in (0, 1)
        self.axis = axis
        self.group = group
        self.P = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        n_split = nx if axis == 0 else ny
        if n_split < 3 * self.P:
            raise ValueError("need at least 3 lines per rank (%d over %d ranks)" % (n_split, self.P))
        base, rem = divmod(n_split, self.P)
        counts = [base + (1 if r < rem else 0) for r in range(self.P)]
        self.lo = sum(counts[:self.rank]); self.hi = self.lo + counts[self.rank]
        self.nx, self.ny, self.dx, self.dy, self.beta, self.tol = nx, ny, dx, dy, beta, tol
        self.has_up, self.has_down = self.rank > 0, self.rank < self.P - 1
        self.compute = compute if compute is not None else HipSorCompute()

    def local_rows(self, full):
        """This rank's rows (axis 0) or columns (axis 1) of a global [nx, ny] array."""
        return full[self.lo:self.hi] if self.axis == 0 else full[:, self.lo:self.hi]

    def _line(self, a, i):
        """Line i along the split axis (a view): row i, or column i."""
        return a[i] if self.axis == 0 else a[:, i]

    def exchange_halo(self, *slabs):
        """Refresh the halo rows of the given slabs ([nloc + has_up + has_down, ny] each) from the neighbours' edge rows:
        ONE message per direction carrying all the fields (no wrap: the domain is not periodic)."""
        if self.P == 1:
            return
        up, down = self.has_up, self.has_down
        reqs = []
        if up:
            first = torch.stack([self._line(a, 1) for a in slabs])
            top = torch.empty_like(first)
            reqs += [dist.isend(first, self.rank - 1, group=self.group), dist.irecv(top, self.rank - 1, group=self.group)]
        if down:
            last = torch.stack([self._line(a, -2) for a in slabs])
            bot = torch.empty_like(last)
            reqs += [dist.isend(last, self.rank + 1, group=self.group), dist.irecv(bot, self.rank + 1, group=self.group)]
        for q in reqs:
            q.wait()
        for i, a in enumerate(slabs):
            if up:
                self._line(a, 0).copy_(top[i])
            if down:
                self._line(a, -1).copy_(bot[i])

    def to_slab(self, rows):
        """Owned rows [hi - lo, ny] -> working slab with (zeroed) halo rows."""
        up, halo = int(self.has_up), int(self.has_up) + int(self.has_down)
        if self.axis == 0:
            slab = torch.zeros(rows.shape[0] + halo, self.ny, dtype=rows.dtype, device=rows.device)
            slab[up:up + rows.shape[0]].copy_(rows)
        else:
            slab = torch.zeros(self.nx, rows.shape[1] + halo, dtype=rows.dtype, device=rows.device)
            slab[:, up:up + rows.shape[1]].copy_(rows)
        return slab

    def owned(self, slab):
        """View of the owned rows of a working slab."""
        up = int(self.has_up)
        return slab[up:up + (self.hi - self.lo)] if self.axis == 0 else slab[:, up:up + (self.hi - self.lo)]

    def solve_slab_(self, slab, cs, max_sweeps):
        """Red-black solve in place on the working slab (halo rows are refreshed here).  Returns (sweeps, last err)."""
        gi0 = self.lo - int(self.has_up)                      # global index of slab line 0: the colour of a point is (offset + i + j) % 2
        err_buf = torch.zeros(1, dtype=slab.dtype, device=slab.device)
        err, done = 1.0, 0
        while done < max_sweeps and err > self.tol:
            err_buf.zero_()
            for colour in (0, 1):
                self.exchange_halo(slab)
                self.compute.halfsweep(slab, cs, err_buf, gi0, colour, self.dx, self.dy, self.beta)
            e = self.compute.err_value(err_buf).clone()
            if self.P > 1:
                dist.all_reduce(e, op=dist.ReduceOp.MAX, group=self.group)
            err = float(e.item())
            done += 1
        return done, err

    def solve_(self, p_loc, C_loc, max_sweeps):
        """p_loc, C_loc: this rank's rows [hi - lo, ny] (p is updated in place).  Returns (sweeps done, last err)."""
        slab, cs = self.to_slab(p_loc), self.to_slab(C_loc)
        done, err = self.solve_slab_(slab, cs, max_sweeps)
        p_loc.copy_(self.owned(slab))
        return done, err


class HipChorinCompute(HipSorCompute):
    """Operator back-end of SlabChorinFD: the HIP kernels of the single-GPU chorin_fd mirror, applied to a row slab."""

    def predictor(self, un, vn, un1, vn1, dt, dx, dy, nu, corrected):
        from . import ops
        f = ops.fd_predictor_explicit_corrected if corrected else ops.fd_predictor_explicit
        return f(un, vn, un1, vn1, dt, dx, dy, nu)

    def predictor_adi(self, un, vn, un1, vn1, dt, dx, dy, nu):
        from . import ops
        return ops.fd_predictor_adi(un, vn, un1, vn1, dt, dx, dy, nu, column_slab=True)

    def bc_apply_(self, A, bcs):
        from . import ops
        if bcs:
            ops.bc_apply_(A, bcs)
        return A

    def rhs(self, ui, vi, dt, dx, dy, rho):
        from . import ops
        return ops.fd_pressure_rhs(ui, vi, dt, dx, dy, rho)

    def correction(self, ui, vi, p, dt, dx, dy):
        from . import ops
        return ops.fd_correction(ui, vi, p, dt, dx, dy)


def _bc_tuple(b):
    return tuple(b) if isinstance(b, (tuple, list)) else (b.type, b.boundary, b.value, b.dx, b.dy)


class SlabChorinFD(object):
    """The chorin_fd projection step (src/chorin_fd/simulate.py:212-234, explicit predictor) on ONE [nx, ny] cavity grid
    sharded by rows over the ranks -- SURVEY.md section 8 (e), rows 1-2 ("cavity (non-periodic) = same without wrap";
    "opt-in red-black shards like Jacobi").  The pressure solve is the red-black option (the reference's lexicographic
    order is sequential across slabs and does not shard); everything else is the reference's step:

        exchange (u, v)^n halos [done at the end of the previous step]
        predictor on the slab -> u*, v*      (interior rows = owned rows; halo / boundary rows copy u^n as in the kernel)
        u / v boundary conditions            ('left' only on the first rank, 'right' only on the last, columns everywhere;
                                              list order kept, so corners resolve as in the single-process run)
        exchange (u*, v*) halos              (the RHS differences backwards along x)
        C = RHS;  red-black SOR (SlabPressure.solve_slab_: halo exchange per half-sweep, all-reduce(max) per sweep)
        p boundary conditions;  exchange p halos   (the correction differences p centrally along x)
        correction -> u, v^{n+1};  exchange their halos

    = 3 packed neighbour exchanges per step outside the pressure solve.  In float64 the owned rows are bitwise those of
    the single-process run with pressure_solver='redblack' (same kernels, same operation order per point)."""

    def __init__(self, u_bc, v_bc, p_bc, nit, nx, ny, dt, rho, nu, beta, advection='reference', group=None, compute=None,
                 method='explicit'):
        """method='semi_implicit' (the reference's ADI predictor, :93-167): both of its tridiagonal solves run along axis
        0, so the grid is split into COLUMN slabs and the solves need no communication at all (SURVEY.md section 8 (e),
        ADI row); the halo columns serve the explicit terms.  Then 'bottom' / 'top' are the sides owned by the first /
        last rank.  It needs nx == ny globally, as in the reference."""
        assert advection in ['reference', 'corrected'] and method in ['explicit', 'semi_implicit']
        if method == 'semi_implicit' and (advection != 'reference' or nx != ny):
            raise ValueError("semi_implicit slabs: the reference ADI (advection='reference') on a square grid")
        self.method = method
        axis = 1 if method == 'semi_implicit' else 0
        self.compute = compute if compute is not None else HipChorinCompute()
        self.nx, self.ny, self.dt, self.rho, self.nu, self.beta, self.nit = nx, ny, dt, rho, nu, beta, nit
        self.dx, self.dy = 2. / (nx - 1), 2. / (ny - 1)                    # src/chorin_fd/simulate.py:58
        self.corrected = advection == 'corrected'
        self.press = SlabPressure(nx, ny, self.dx, self.dy, beta, tol=5e-6, group=group, compute=self.compute, axis=axis)
        lo_side, hi_side = ('left', 'right') if axis == 0 else ('bottom', 'top')       # A[0, :], A[-1, :] / A[:, 0], A[:, -1]
        keep = lambda b: not ((b[1] == lo_side and self.press.has_up) or (b[1] == hi_side and self.press.has_down))
        self.u_bc, self.v_bc, self.p_bc = ([b for b in map(_bc_tuple, l) if keep(b)] for l in (u_bc, v_bc, p_bc))
        self.last_sor = None

    def init_slabs(self, u_ic, v_ic, p_ic):
        """Global initial fields (every rank passes the same arrays) -> this rank's working slabs with the boundary
        conditions applied and the halos filled (src/chorin_fd/simulate.py:236-249)."""
        sp = self.press
        slabs = [sp.to_slab(sp.local_rows(a)) for a in (u_ic, v_ic, p_ic)]
        for a, bc in zip(slabs, (self.u_bc, self.v_bc, self.p_bc)):
            self.compute.bc_apply_(a, bc)
        sp.exchange_halo(*slabs)
        return slabs

    def step(self, un, vn, un1, vn1, p):
        """One step on working slabs (halos of un, vn, un1, vn1 valid on entry).  p is updated in place; returns (u, v, p)
        with valid halos."""
        c, sp = self.compute, self.press
        if self.method == 'semi_implicit':
            ui, vi = c.predictor_adi(un, vn, un1, vn1, self.dt, self.dx, self.dy, self.nu)
        else:
            ui, vi = c.predictor(un, vn, un1, vn1, self.dt, self.dx, self.dy, self.nu, self.corrected)
        c.bc_apply_(ui, self.u_bc)
        c.bc_apply_(vi, self.v_bc)
        sp.exchange_halo(ui, vi)
        C = c.rhs(ui, vi, self.dt, self.dx, self.dy, self.rho)
        self.last_sor = sp.solve_slab_(p, C, max(int(self.nit) - 1, 0))
        c.bc_apply_(p, self.p_bc)
        sp.exchange_halo(p)
        u, v = c.correction(ui, vi, p, self.dt, self.dx, self.dy)
        sp.exchange_halo(u, v)
        return u, v, p

    def simulate(self, u_ic, v_ic, p_ic, nt):
        """nt steps from the global initial fields; returns this rank's OWNED rows (columns for semi_implicit) of the
        trajectory, three tensors [nt, hi - lo, ny] ([nt, nx, hi - lo]); rank order = line order: concatenate along
        axis 1 (2) for the global fields."""
        u, v, p = self.init_slabs(u_ic, v_ic, p_ic)
        u1, v1 = u.clone(), v.clone()                                       # first step: u^{-1} = u^0 (:256)
        sp = self.press
        us, vs, ps = [], [], []
        for _ in range(nt):
            nu_, nv_, p = self.step(u, v, u1, v1, p)
            u1, v1, u, v = u, v, nu_, nv_
            us.append(sp.owned(u).clone()), vs.append(sp.owned(v).clone()), ps.append(sp.owned(p).clone())
        return torch.stack(us), torch.stack(vs), torch.stack(ps)
