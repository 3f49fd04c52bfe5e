"""The arithmetic fact the exact active set of the HJB solve rests on (csrc/hjb.hip, k_stageV / build_work_list), checked on
the CPU with the oracle's restatement of the reference's RHS (optimals.py:144-164) and SciPy's RK45 stages (rk.py:14-72):

  where phi is one constant power of two u on a tile and its halo (6 rows / 8 columns), K0 = 0 and the potential term is 0
  or wall, ONE Runge-Kutta step leaves the tile exactly as it was -- every stage derivative exactly 0, y_new = u bit for bit;

  and a tile that is trivial in that sense TOGETHER WITH ITS 8 NEIGHBOURS is still trivial, halo included, after the step
  (so it needs no block in the next attempt): the stencil carries the targets' influence 6 cells per step, a tile is 20 x 112.

No GPU: this is the claim; tests/test_hip_active_set.py checks the kernels against computing every tile."""
import numpy as np

from oracle import oc_oracle as oc                     # (checker only)

TY, TX, HY, HX = 20, 112, 6, 8                        # k_stageV's tile and halo


def _rk_step(y, k0, h, f):
    """one Dormand-Prince step in the reference's order of operations (scipy rk.py:14-72): K[0..6], y_new"""
    K = [k0]
    for s in range(1, 6):
        dy = sum(a * k for a, k in zip(oc.RK_A[s][:s], K)) * h
        K.append(f(y + dy))
    y_new = y + h * sum(b * k for b, k in zip(oc.RK_B, K))
    K.append(f(y_new))
    return K, y_new


def _tile_flags(y, k0, coef_zero_or_wall, u, Ny, Nx):
    """per tile: the whole region (tile + halo, clipped to the grid) has y == u, K0 == 0, potential term 0 or wall"""
    cell = (y.reshape(Ny, Nx) == u) & (k0.reshape(Ny, Nx) == 0) & coef_zero_or_wall
    nby, nbx = -(-Ny // TY), -(-Nx // TX)
    fl = np.zeros((nby, nbx), bool)
    for bi in range(nby):
        for bj in range(nbx):
            r0, r1 = max(bi * TY - HY, 0), min(bi * TY + TY + HY, Ny)
            c0, c1 = max(bj * TX - HX, 0), min(bj * TX + TX + HX, Nx)
            fl[bi, bj] = cell[r0:r1, c0:c1].all()
    return fl


def test_trivial_tiles_stay_bit_identical_and_quiet_neighbourhoods_stay_trivial():
    Ny, Nx, dx = 200, 448, 0.1
    V = np.zeros((Ny, Nx))
    V[0], V[-1], V[:, 0], V[:, -1] = -100, -100, -100, -100
    V[60:70, 200:203] = -100                             # a wall inside
    V[95:105, -1] = 1.0                                  # the target: the only place where the potential term is not 0
    m = np.zeros((Ny, Nx))
    hp = oc.DEFAULT_CONFIG["hjb_params"]
    f = lambda y: oc.hjb_rhs(y, V, m, hp["sigma"], hp["mu"], hp["g"], dx, dx)      # noqa: E731
    term = V + hp["g"] * m
    coef_ok = (term == 0) | (V < 0)
    u = 2.0 ** -3                                        # any power of two (the float32 state is phi * 2^-e)
    y = np.full(Ny * Nx, u)
    k0 = f(y)
    h = -0.9 * 3.3 / (8 * (hp["sigma"] ** 2 / 2) / dx ** 2)       # near the stability limit, as the controller runs it
    nby, nbx = -(-Ny // TY), -(-Nx // TX)
    seen_active, seen_quiet = 0, 0
    for step in range(25):
        fl = _tile_flags(y, k0, coef_ok, u, Ny, Nx)
        K, y_new = _rk_step(y, k0, h, f)
        Y2, K6 = y_new.reshape(Ny, Nx), K[6].reshape(Ny, Nx)
        fl_next = _tile_flags(y_new, K[6], coef_ok, u, Ny, Nx)
        for bi in range(nby):
            for bj in range(nbx):
                rows, cols = slice(bi * TY, min(bi * TY + TY, Ny)), slice(bj * TX, min(bj * TX + TX, Nx))
                if fl[bi, bj]:
                    # trivial region -> the tile's outputs are exactly trivial: every stage derivative 0, y_new = u
                    for k in K[1:]:
                        assert not k.reshape(Ny, Nx)[rows, cols].any()
                    assert (Y2[rows, cols] == u).all() and not K6[rows, cols].any()
                    nb = fl[max(bi - 1, 0):bi + 2, max(bj - 1, 0):bj + 2]
                    if nb.all():                         # ... and with 8 trivial neighbours the REGION is trivial next time
                        assert fl_next[bi, bj]
                        seen_quiet += 1
                else:
                    seen_active += 1
        y, k0 = y_new, K[6]
        if step == 12:                                   # a rescale by a power of two keeps everything a power of two
            y, k0, u = y * 2.0 ** -5, k0 * 2.0 ** -5, u * 2.0 ** -5
    # the front moved: tiles became active over the steps, and the far side of the room never did
    assert seen_active > 25 and seen_quiet > 25 * (nby * nbx) // 3
    assert _tile_flags(y, k0, coef_ok, u, Ny, Nx)[:, 0].all()


def test_the_stencil_of_a_constant_is_exactly_zero():
    """((u + u) + u) + u - 4u in the kernels' order of operations: exact for a power of two at every step (2u, 3u, 4u are
    representable), and exact in the end for any u in round-to-nearest (the rounding of 3u is absorbed by 4u's coarser
    grid) -- the kernel still asks for a power of two (is_pow2): that is what phi = 1 carried as phi * 2^-e is, and it keeps
    the argument independent of the rounding mode."""
    rng = np.random.RandomState(0)
    for dt in (np.float32, np.float64):
        for e in range(-40, 41):
            v = dt(2.0 ** e)
            assert dt(v + v) == 2 * v and dt(dt(v + v) + v) == 3 * v          # exact partial sums
            assert dt(dt(dt(v + v) + v) + v) - dt(4) * v == 0
        for v in rng.uniform(0.5, 2.0, 2000).astype(dt):
            assert dt(dt(dt(v + v) + v) + v) - dt(4) * v == 0


def _region_trivial(t, y_seen, k_seen, u, coef_ok, nbx, Ny, Nx):
    bi, bj = divmod(t, nbx)
    r0, r1 = max(bi * TY - HY, 0), min(bi * TY + TY + HY, Ny)
    c0, c1 = max(bj * TX - HX, 0), min(bj * TX + TX + HX, Nx)
    return bool((y_seen[r0:r1, c0:c1] == u).all() and not k_seen[r0:r1, c0:c1].any() and coef_ok[r0:r1, c0:c1].all())


def test_far_field_substitution_protocol_never_reads_an_unmaintained_cell():
    """The bookkeeping of k_stageV's far-field substitution (ring solves; csrc/hjb.hip: Stage3Args::undef), restated with
    NumPy and run against a solver that maintains every cell.  The state buffers Y[b] / K[b] are POISONED (NaN) wherever the
    protocol does not write them; a mark undef[b][t] says that tile t's cells of buffer b stand for "y = the far-field
    value, K0 = 0"; a listed tile loads its region with the marked tiles' cells substituted, tests it, and either computes
    (writes its cells of the other buffer, mark 0) or declares itself trivial (writes nothing, mark 1); the builder drops a
    tile that was trivial with its 8 neighbours and marks dropped tiles in both buffers in the first two builds.  Checked
    over a sequence with rejected attempts (the buffer that stays y keeps its own marks) and changes of scale (nothing is
    rewritten): every region a listed tile loads equals the fully maintained state, bit for bit, and no NaN is ever read."""
    Ny, Nx, dx = 200, 448, 0.1
    V = np.zeros((Ny, Nx))
    V[0], V[-1], V[:, 0], V[:, -1] = -100, -100, -100, -100
    V[95:105, -1] = 1.0
    hp = oc.DEFAULT_CONFIG["hjb_params"]
    m = np.zeros((Ny, Nx))
    f = lambda y: oc.hjb_rhs(y, V, m, hp["sigma"], hp["mu"], hp["g"], dx, dx)      # noqa: E731
    coef_ok = ((V + hp["g"] * m) == 0) | (V < 0)
    h = -0.9 * 3.3 / (8 * (hp["sigma"] ** 2 / 2) / dx ** 2)
    nby, nbx = -(-Ny // TY), -(-Nx // TX)
    tile_of = (np.arange(Ny)[:, None] // TY) * nbx + np.arange(Nx)[None, :] // TX      # the tile a cell lies in
    rng = np.random.RandomState(5)

    # the solver that maintains everything: the raw content a fully maintained y / K0 buffer would have
    u_bg = 1.0
    y_true = np.full(Ny * Nx, u_bg)
    k_true = f(y_true)
    # the protocol's buffers: buffer `par` holds the initial state, the other one nothing yet
    Y = [y_true.reshape(Ny, Nx).copy(), np.full((Ny, Nx), np.nan)]
    K = [k_true.reshape(Ny, Nx).copy(), np.full((Ny, Nx), np.nan)]
    undef = np.zeros((2, nby * nbx), bool)
    par, scale = 0, 1.0                                  # `scale`: the pending in_scale, applied as the next attempt loads
    work = list(range(nby * nbx))                        # (the first attempt looks at every tile)
    builds = rejects = rescales = declared = accepted = computed_first_time_in_a_reject = 0
    ever_computed = np.zeros(nby * nbx, bool)
    forced = set()
    for attempt in range(60):
        # what every listed tile sees: the buffer being read, marked tiles replaced by what they stand for
        sub = undef[par][tile_of]
        y_seen = np.where(sub, u_bg, Y[par])
        k_seen = np.where(sub, 0.0, K[par])
        Kst, y_new = _rk_step(y_true * scale, k_true * scale, h, f)            # (the maintained solver's attempt)
        accept = rng.rand() < 0.75 or attempt < 2
        # (the corner the marks-per-BUFFER exist for, forced: the attempt in which a tile is computed for the first time is
        # rejected once -- the retry reads the buffer that stays y, where that tile is still only a mark)
        newly = [t for t in work if attempt >= 2 and not ever_computed[t] and not _region_trivial(t, y_seen, k_seen, u_bg, coef_ok, nbx, Ny, Nx)]
        if newly and tuple(newly) not in forced:
            forced.add(tuple(newly))
            accept = False
        flags = np.ones(nby * nbx, bool)                 # trivial at this attempt (tiles off the list: trivial by the rule)
        for t in work:
            bi, bj = divmod(t, nbx)
            r0, r1 = max(bi * TY - HY, 0), min(bi * TY + TY + HY, Ny)
            c0, c1 = max(bj * TX - HX, 0), min(bj * TX + TX + HX, Nx)
            reg_y, reg_k = y_seen[r0:r1, c0:c1], k_seen[r0:r1, c0:c1]
            assert not np.isnan(reg_y).any() and not np.isnan(reg_k).any(), (attempt, t)
            assert np.array_equal(reg_y, y_true.reshape(Ny, Nx)[r0:r1, c0:c1]), (attempt, t)
            assert np.array_equal(reg_k, k_true.reshape(Ny, Nx)[r0:r1, c0:c1]), (attempt, t)
            trivial = _region_trivial(t, y_seen, k_seen, u_bg, coef_ok, nbx, Ny, Nx)
            rows, cols = slice(bi * TY, min(bi * TY + TY, Ny)), slice(bj * TX, min(bj * TX + TX, Nx))
            if trivial:
                undef[par ^ 1][t] = True                 # nothing is stored
                declared += 1
            else:
                Y[par ^ 1][rows, cols] = y_new.reshape(Ny, Nx)[rows, cols]
                K[par ^ 1][rows, cols] = Kst[6].reshape(Ny, Nx)[rows, cols]
                undef[par ^ 1][t] = False
                flags[t] = False
                if not ever_computed[t] and not accept:
                    computed_first_time_in_a_reject += 1
                ever_computed[t] = True
        # the builder: a tile stays off the next list if it was trivial with all 8 neighbours -- whatever the decision was
        fl2 = flags.reshape(nby, nbx)
        work = [bi * nbx + bj for bi in range(nby) for bj in range(nbx)
                if not fl2[max(bi - 1, 0):bi + 2, max(bj - 1, 0):bj + 2].all()]
        builds += 1
        if builds <= 2:
            off = np.ones(nby * nbx, bool)
            off[work] = False
            undef[0][off] = undef[1][off] = True
        if accept:
            y_true, k_true = y_new, Kst[6]
            u_bg *= scale                                # (what the far field reads in the buffer that is y from now on)
            par ^= 1
            scale = 1.0
            accepted += 1
            if accepted in (6, 15, 27):                  # a change of scale: nothing is rewritten
                scale = 2.0 ** -4
                rescales += 1
        else:
            rejects += 1
    assert rejects >= 8 and rescales == 3 and declared > 50 and computed_first_time_in_a_reject >= 1
    assert len(work) < nby * nbx                         # tiles stayed off the list to the end
    # what is marked stands for the far field, what is not is maintained
    for t in range(nby * nbx):
        bi, bj = divmod(t, nbx)
        rows, cols = slice(bi * TY, min(bi * TY + TY, Ny)), slice(bj * TX, min(bj * TX + TX, Nx))
        yt, kt = y_true.reshape(Ny, Nx)[rows, cols], k_true.reshape(Ny, Nx)[rows, cols]
        if undef[par][t]:
            assert (yt == u_bg).all() and not kt.any()
        else:
            assert np.array_equal(Y[par][rows, cols], yt) and np.array_equal(K[par][rows, cols], kt)
