/* oc_hip.h -- C-ABI of liboc_hip.so: the MI355X (gfx950) implementation of the optimal_crowds hot path.
 *
 * The reference (matteobutano/optimal_crowds) has no FFI: the path sits behind Python methods.  Each entry
 * point below names the reference interface it replaces (file:line into the reference tree).  The host side
 * (the Python modules of optimal_crowds_amd/) binds these with ctypes; INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative oc_status otherwise; oc_last_error() gives the text.
 *   - all array arguments are DEVICE pointers owned by the caller (e.g. torch tensors' data_ptr()), valid until
 *     the stream has drained; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - calls are asynchronous on `stream` unless the documentation says the function returns host values.
 *   - one handle per device and per simulation; a handle is not thread-safe.
 *   - grids are row-major (Ny, Nx): cell (i, j) = row i (y index), column j (x index), as in the reference.
 *   - `dtype` selects the arithmetic of the HJB state and direction field: OC_F32 (product) or OC_F64 (audit);
 *     oc_hjb_solve / oc_hjb_solve_rows also take OC_F64_F32: float64 state and coef, float32 direction field (the
 *     reference's field to 1e-6 without the float64 field's memory; register the field with oc_set_field(OC_F32)).
 */
#ifndef OC_HIP_H
#define OC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OC_ABI_VERSION 3

typedef enum { OC_OK = 0, OC_ERR_ARG = -1, OC_ERR_HIP = -2, OC_ERR_STATE = -3, OC_ERR_SOLVER = -4, OC_ERR_CAPACITY = -5 } oc_status;
typedef enum { OC_F32 = 0, OC_F64 = 1, OC_F64_F32 = 2 } oc_dtype;

typedef struct oc_handle oc_handle;

/* Scalars of config.json + the room extent (reference simulations.py:54-97, optimals.py:48-74). */
typedef struct oc_config {
    int32_t Ny, Nx;               /* int(H//grid_step+1), int(L//grid_step+1)      simulations.py:63-64 */
    double dx, dy;                /* = grid_step                                   simulations.py:66-67 */
    double room_length, room_height;
    double dt;                    /* config "dt" */
    /* HJB (config "hjb_params") */
    double sigma, mu, g;
    double lim;                   /* 10e-3, optimals.py:95 */
    double rtol, atol;            /* SciPy RK45 defaults 1e-3 / 1e-6 (scipy rk.py:85-86) */
    /* GCFM */
    double a_min;                 /* = config b_min (quirk: simulations.py:88, pedestrians.py:74) */
    double tau_a, b_min, b_max, eta, eta_walls;
    double repulsion_cutoff, v_max, relaxation;
    double noise_amp;             /* hjb sigma / 2, simulations.py:303 */
    double wall_search_radius;    /* <=0: unbounded exact search; else rows beyond this distance are skipped */
    int32_t max_agents;           /* capacity for scratch */
    int32_t max_keys;             /* number of distinct target keys */
} oc_config;

/* ---- lifetime ------------------------------------------------------------------------------------------- */
int oc_abi_version(void);
const char* oc_last_error(void);
/* X0 (Nx) and Y0 (Ny): HOST float64 axes = np.linspace(0, L, Nx) / (0, H, Ny) (simulations.py:69-70). */
int oc_create(int device, const oc_config* cfg, const double* X0_host, const double* Y0_host, oc_handle** out);
int oc_destroy(oc_handle* h);

/* ---- HJB: reference optimals.py:124-206 ------------------------------------------------------------------ */
/* coef = V + g*m on free cells, NaN on walls (V < 0)  -- the (V + g*m) term and the wall mask of
 * optimals.py:159-162.  V: float32 (Ny,Nx) finalised potential {-100,0,1}; m: dtype (Ny,Nx) or NULL (= 0). */
int oc_hjb_make_coef(oc_handle* h, int dtype, const float* V, const void* m, void* coef, void* stream);

/* out = hjb(t, phi, m): replaces the closure `hjb` at optimals.py:144-164.  phi/out: dtype (Ny*Nx). */
int oc_hjb_rhs(oc_handle* h, int dtype, const void* phi, const void* coef, void* out, void* stream);

/* The step controller's power err^(-1/5) (SciPy rk.py: factor = SAFETY * error_norm ** ERROR_EXPONENT, ERROR_EXPONENT = -1/5
 * for RK45) as the DEVICE controller forms it -- the hardware's float log2 / exp2 and two float64 Newton steps (hjb.hip:
 * inv_fifth_root) -- for n values x[i] in [1e-6, 1e6] (outside that range the controller's clamps decide without it).
 * x, out: device pointers to float64.  A test hook: the solve's parity with SciPy rests on this being good to a few ulp. */
int oc_hjb_ctl_power(oc_handle* h, const double* x, int64_t n, double* out, void* stream);

typedef struct oc_hjb_stats {
    int32_t nfev, n_attempt, n_accept, n_rescale;
    double h_abs0, h_min, h_max;
    int32_t final_exponent;       /* state carried as phi * 2^-exponent (fp32 range management) */
    int32_t n_slices_written;
    double seconds;               /* host wall time of the call (synchronous) */
    /* the persistent launch (oc_set_tuning "hjb_persist"): attempts that ran inside it, its grid, and how it ended --
     * 0 not used, 1 the solve finished inside, 2 handed back to the launch chain (every tile active / list too long),
     * 3 given up at a barrier deadline (grid not co-resident); the chain continues from the controller's state either way */
    int32_t persist_attempts, persist_blocks, persist_exit;
    int32_t grid_resyncs;         /* sparse attempts whose launch was too small for the work list (drained and re-enqueued) */
    int32_t lead_attempts;        /* attempts launched with the leading controller (oc_set_tuning "hjb_lead") */
    int32_t reserved_;
} oc_hjb_stats;

/* Replaces compute_optimal_velocity's solve_ivp(RK45, t_eval=linspace(T,0,nt)) + vels() fill loop,
 * optimals.py:193-204 (+ SciPy RK45 rk.py:14-180, 552-575; common.py:63-134; ivp.py:598-723).
 *   nt        = round((T - t_now)/dt)   (optimals.py:139-142); slices s = 0..nt-2 of vx_out/vy_out are written:
 *               vx_out[s] = direction field of phi(t = s*T/(nt-1)).  Layout: (nslice, Ny, Nx) of `dtype`, i.e. each
 *               slice padded to the full grid so rows are 16-byte aligned; the reference's vx_opt[s] (Ny-2, Nx-2)
 *               is the interior block [1:-1, 1:-1], the border ring is written as 0.
 *   keep_idx  : optional HOST list of n_keep t_eval indices k whose phi(t_eval[k]) is stored unscaled as
 *               float64 (n_keep, Ny, Nx) in phi_keep (device) -- audit/tests only.
 *   t_steps   : optional HOST buffer (cap t_steps_cap) receiving accepted step end times.
 * Synchronous: returns after the solve has drained; stats are host values. */
int oc_hjb_solve(oc_handle* h, int dtype, const void* coef, double T, int32_t nt,
                 void* vx_out, void* vy_out,
                 const int32_t* keep_idx, int32_t n_keep, double* phi_keep,
                 double* t_steps, int32_t t_steps_cap,
                 oc_hjb_stats* stats, void* stream);

/* ---- row-block (domain-decomposed) HJB solve: one rank of `world` ranks that split the grid rows ------------------
 * The same solve as oc_hjb_solve (optimals.py:193-204) on rows [row0, row0 + Ny_local) of the grid.  The local arrays
 * hold the rank's owned rows plus `halo` rows of each neighbour:
 *     local rows [0, own_lo)  upper halo (has_up)   [own_lo, own_hi)  owned   [own_hi, Ny_local)  lower halo (has_dn)
 * One RK step applies the 5-point stencil 6 times (5 stages + the FSAL derivative), so with y and K0 valid `halo`
 * rows deep at the start of a step, K1..K5, y_new, K6 are valid halo-1 .. halo-6 rows deep: that keeps the owned rows
 * exact.  The dense output's coefficients Q = K^T P (scipy rk.py:178-180) are formed inside the stage launch, from the
 * LOCAL K6, and the direction field of an owned row differentiates phi across the row beyond it: halo >= 7, and the
 * direction field is exact on the owned rows +- (halo - 7).  After every ACCEPTED step the rank
 * sends its `halo` owned rows next to each neighbour of (y_new, K6) and receives the neighbour's: one message of
 * 2*halo*Nx elements per neighbour and step (instead of one row per RHS evaluation).  The error norm of an attempt
 * (scipy rk.py:146-147) is summed over owned rows and completed by an all-gather of (sum, max|y|) that every rank adds
 * in rank order, so all ranks take identical step-size decisions.
 * The library does the packing/unpacking and the controller; the two collectives are the caller's (host callbacks that
 * enqueue on `stream`, e.g. torch.distributed over RCCL):
 *   allgather_red(user): red_all[r][0..1] = rank r's red_local[0..1]
 *   exchange_halo(user): send_up -> upper neighbour's recv_dn, send_dn -> lower neighbour's recv_up
 * Buffers (device, caller-owned): send_up/recv_up/send_dn/recv_dn: dtype (2, halo, Nx); red_local: float64 (2);
 * red_all: float64 (world, 2).  coef_rows / vx_out / vy_out / phi_keep cover the LOCAL rows: coef_rows (Ny_local, Nx),
 * vx_out (nslice, Ny_local, Nx), phi_keep (n_keep, Ny_local, Nx).  Ny_global*Nx is the RMS norm's cell count. */
#define OC_MAX_WORLD 64
typedef struct oc_hjb_shard {
    int32_t Ny_local, Ny_global;
    int32_t own_lo, own_hi;
    int32_t halo;
    int32_t has_up, has_dn;
    int32_t world;
    void *send_up, *recv_up, *send_dn, *recv_dn;
    double *red_local, *red_all;
    int (*allgather_red)(void* user);
    int (*exchange_halo)(void* user);
    void* user;
    /* >= 0: the two collectives are issued by the library itself over the RCCL communicator of this slot (oc_comm_create;
     * its ranks = the row blocks in order, this rank = block `own` index) and the callbacks are not called; -1: callbacks */
    int32_t comm_slot, reserved;
} oc_hjb_shard;
int oc_hjb_solve_rows(oc_handle* h, int dtype, const void* coef_rows, double T, int32_t nt,
                      void* vx_out, void* vy_out, const oc_hjb_shard* shard,
                      const int32_t* keep_idx, int32_t n_keep, double* phi_keep,
                      oc_hjb_stats* stats, void* stream);

/* ---- the interpolant ring: the direction field evaluated where it is looked up ---------------------------------------
 * The reference fills every slice of vx_opt / vy_opt because NumPy has to (optimals.py:200-204), but step() reads ONE
 * slice per step and only one or two cells of it per agent (optimals.py:231-248), and with recompute=True every re-solve
 * overwrites ~950 slices of which 50 were looked up (simulations.py:432-435).  oc_hjb_solve_ring runs the same solve as
 * oc_hjb_solve and keeps, instead of slices, what SciPy's dense output of each ACCEPTED step is made of (rk.py:178-180,
 * 560-575): y_old, Q0 = K0, Q1 .. Q3 -- five values per cell and step (2048^2, T = 20: 93 steps x 5 arrays = 7.8 GB per
 * key instead of 999 slices x 2 = 33.5 GB, and no extra array pass at all: the stage launch stores them in the place of
 * Q1 .. Q3; a slot is five planes of rows x Nx values, y | K0 | Q1 | Q2 | Q3) -- plus a table slice -> (step, h x^1..4).  oc_set_field_ring registers it for the agent step, whose lookups
 * then evaluate phi(t_eval) at the cell and its four neighbours and `vels` (optimals.py:168-186) on the spot, with the FMA
 * chain of the slice writer: bit-identical to looking the value up in the slices oc_hjb_solve writes.
 * oc_hjb_ring_slices materialises slices from a ring with the slice writer itself (for optimals.vx_opt / tests).
 *   data : `dtype` state values, slots * oc_hjb_ring_slot_elems(rows) of them (OC_F64_F32: float64)
 *   meta : oc_hjb_ring_meta_bytes(...) bytes (per-slot records, the slice table, tile flags of the exact active set)
 *   slots: capacity in accepted steps (oc_hjb_ring_slots_hint); a solve that needs more returns OC_ERR_CAPACITY
 *   nslice: slices the table covers (nt0 - 1);  rows: grid rows held (Ny, or Ny_local of a row block)
 * Needs Nx % 4 == 0.  All three buffers are device memory owned by the caller. */
typedef struct oc_hjb_ring {
    void* data;
    void* meta;
    int32_t slots, nslice, rows, reserved;
} oc_hjb_ring;
int64_t oc_hjb_ring_slot_elems(const oc_handle* h, int32_t rows);
int64_t oc_hjb_ring_meta_bytes(const oc_handle* h, int dtype, int32_t slots, int32_t nslice, int32_t rows);
int32_t oc_hjb_ring_slots_hint(const oc_handle* h, double T);
/* Several independent whole-grid ring solves at the same time -- the reference solves its keys one after the other
 * (simulations.py:419-420: `for key in self.targets: self.targets[key].compute_optimal_velocity(...)`); they share nothing.
 * One job = oc_hjb_make_coef(h, coef_dtype, V, m, coef) + oc_hjb_solve_ring(h, dtype, coef, T, nt, ring, stats) on `stream`;
 * every job needs its own handle (solver scratch), coef buffer and stream.  The library runs the jobs on host threads of its
 * own (job 0 on the caller's) and returns when all have drained; rc / error / stats are per job.  Returns the first failing
 * job's code. */
typedef struct oc_hjb_ring_job {
    oc_handle* h;
    int32_t coef_dtype, dtype;      /* oc_hjb_make_coef's / oc_hjb_solve_ring's dtype */
    const float* V;                 /* the key's finalised potential (Ny, Nx) */
    const void* m;                  /* density term (Ny, Nx) of coef_dtype's element type, or NULL */
    void* coef;                     /* (Ny, Nx) scratch of the job */
    double T;
    int32_t nt, rc;
    const oc_hjb_ring* ring;
    void* stream;
    oc_hjb_stats stats;
    char error[256];
} oc_hjb_ring_job;
int oc_hjb_solve_ring_many(oc_hjb_ring_job* jobs, int32_t n);

/* oc_hjb_solve / oc_hjb_solve_rows with a ring in the place of vx_out / vy_out (same arguments otherwise; phi_keep is
 * produced from the ring after the solve). */
int oc_hjb_solve_ring(oc_handle* h, int dtype, const void* coef, double T, int32_t nt, const oc_hjb_ring* ring,
                      const int32_t* keep_idx, int32_t n_keep, double* phi_keep,
                      double* t_steps, int32_t t_steps_cap, oc_hjb_stats* stats, void* stream);
int oc_hjb_solve_rows_ring(oc_handle* h, int dtype, const void* coef_rows, double T, int32_t nt, const oc_hjb_ring* ring,
                           const oc_hjb_shard* shard, const int32_t* keep_idx, int32_t n_keep, double* phi_keep,
                           oc_hjb_stats* stats, void* stream);
/* Slices [s_lo, s_hi) (s_hi <= nt - 1 of the ring's solve) as oc_hjb_solve would have written them: vx_out / vy_out are
 * (s_hi - s_lo, rows, Nx) of the field type of `dtype` (OC_F32 / OC_F64_F32: float32, OC_F64: float64).  Synchronises. */
int oc_hjb_ring_slices(oc_handle* h, int dtype, const oc_hjb_ring* ring, int32_t s_lo, int32_t s_hi, void* vx_out,
                       void* vy_out, void* stream);

/* Direction field of one phi (float64 input (Ny,Nx), dtype output padded (Ny,Nx)): replaces `vels`, optimals.py:168-186. */
int oc_hjb_vels(oc_handle* h, int dtype, const void* phi, double scale, void* vx, void* vy, void* stream);

/* ---- per-key data used by the agent step ------------------------------------------------------------------ */
/* Register the direction field of key `key`: arrays as written by oc_hjb_solve; nt_opt = current horizon
 * (optimals.nt_opt), nslice_alloc = allocated slices (nt0-1).  Replaces attribute access optimals.vx_opt. */
int oc_set_field(oc_handle* h, int key, int dtype, const void* vx, const void* vy, int32_t nslice_alloc, int32_t nt_opt);
/* The same for a ROW BLOCK of the field, as written by oc_hjb_solve_rows: arrays (nslice_alloc, rows, Nx) whose local
 * row 0 is grid row `row0`; this rank answers lookups whose first cell lies in grid rows [own_lo, own_hi) (the rows
 * it owns; the block carries the margin the second cell may need) and returns zero for the others -- see
 * oc_gcfm_step_sharded.  simulations.py:281 / optimals.py:212-250 on a domain-decomposed field. */
int oc_set_field_rows(oc_handle* h, int key, int dtype, const void* vx, const void* vy, int32_t nslice_alloc,
                      int32_t nt_opt, int32_t row0, int32_t rows, int32_t own_lo, int32_t own_hi);
/* The same with an interpolant ring (oc_hjb_solve_ring / oc_hjb_solve_rows_ring, `dtype` as passed there) in the place of
 * the slices: optimals.py:231-248 reads slice simu_step -- here that slice's value is evaluated from the ring.  row0 < 0:
 * the ring covers the whole grid; else it is the row block [row0, row0 + ring->rows) and this rank answers the lookups
 * whose first cell lies in grid rows [own_lo, own_hi), as with oc_set_field_rows. */
int oc_set_field_ring(oc_handle* h, int key, int dtype, const oc_hjb_ring* ring, int32_t nt_opt, int32_t row0,
                      int32_t own_lo, int32_t own_hi);
/* Exit rectangles (cx,cy,w,h) of a key's possible targets: pedestrians.py:132-136. rects: HOST (n,4) float64. */
int oc_set_key_targets(oc_handle* h, int key, const double* rects, int32_t n);
/* Room description -> potential on the grid: replaces simulation.create_potential (simulations.py:516-576) and, with
 * finalise != 0, the in-place finalisation of optimals.__init__ (optimals.py:89-91).  Shapes are HOST float64 arrays in
 * the room file's own form: walls / holes / targets (n,4) = (cx, cy, width, height), cylinders (n,3) = (cx, cy, radius);
 * targets = the rectangles of THIS key's targets.  V_out: float32 device (Ny,Nx), bit-identical to the reference's
 * array (values -k, 0, 1; or wall_potential, 0, target_potential when finalised). */
int oc_rasterise_potential(oc_handle* h, const double* walls, int32_t n_walls, const double* holes, int32_t n_holes,
                           const double* cylinders, int32_t n_cyl, const double* targets, int32_t n_targets,
                           double wall_potential, double target_potential, int32_t finalise, float* V_out,
                           void* stream);
/* Nearest-"wall"-cell structure of Vs[key] for pedestrians.py:311-315.  V: float32 device (Ny,Nx);
 * vmin = min(V) (host value): cells with V == vmin are the candidate set of the reference's argmin. */
int oc_wallmap_build(oc_handle* h, int key, const float* V, float vmin, void* stream);

/* ---- GCFM agent update: reference simulations.py:252-342, pedestrians.py:216-334 ----------------------------- */
/* Agent state (SoA, device, length N): x,y,vx,vy,v_des float64; status uint8 (1 = inside); key int32;
 * agent_time float64 (ped.time). */
typedef struct oc_agents {
    int32_t N;
    double *x, *y, *vx, *vy;
    const double* v_des;
    uint8_t* status;
    const int32_t* key;
    double* agent_time;
} oc_agents;

/* floor-field lookup for every agent: des = v_des * choose_optimal_velocity(pos, step)  (optimals.py:212-250,
 * simulations.py:281).  des: float64 (N,2).  err_flag (device int32[2], may be NULL): [0] set to 1 + agent index
 * if an agent is in the reference's IndexError band (quirk Q8). */
int oc_field_lookup(oc_handle* h, const oc_agents* a, int32_t simu_step, double* des, int32_t* err_flag, void* stream);
/* index of the reference's argmin (flat row-major, int64 (N)) and wall force float64 (N,2): pedestrians.py:282-334 */
int oc_wall_force(oc_handle* h, const oc_agents* a, int64_t* index_out, double* force_out, void* stream);
/* the cell the LAST step's (warm-started) search found for each agent: int32 (N), -1 = none within its search radius / no
 * step yet.  Introspection for tests: equals oc_wall_force's index at the positions that step started from. */
int oc_wall_last_index(oc_handle* h, const oc_agents* a, int32_t* index_out, void* stream);
/* sum of agents_repulsion over active j != i within the cutoff: float64 (N,2).  simulations.py:285-295 */
int oc_pair_force(oc_handle* h, const oc_agents* a, double* force_out, void* stream);

/* One synchronous (frozen-snapshot) step of every active agent, in place: simulations.py:277-332.
 *   noise : float64 (N,2) standard normals (noise[i] = agent i's draw) or NULL -> device Philox(seed, step).
 *   counters (device int32[4], reset by the call): [0] number of agents that exited this step,
 *            [1] 1+index of an agent in the IndexError band (0 = none), [2] active agents after the step.
 * Rebuilds the spatial order every `resort_every` steps (0 = library default). */
int oc_gcfm_step(oc_handle* h, const oc_agents* a, int32_t simu_step, const double* noise, uint64_t seed,
                 int32_t* counters, void* stream);
/* One SEQUENTIAL (Gauss-Seidel) step in the reference's own update order -- simulations.py:268-332 with
 * pedestrians.py:140-162 returning the already advanced trajectory point: agents are visited one by one in the
 * permutation `perm` (device int32 (N): perm[0] moves first; the host draws it like simulations.py:271 does), and
 * agent i sees the NEW position / velocity / status of every agent visited before it.  The library level-schedules the
 * permutation: an agent moves as soon as all earlier agents within cutoff + 2 M of it have moved (M = "seq_margin_mm",
 * default 250 mm, a bound on one step's displacement that every round verifies; a violation is OC_ERR_STATE): every agent's
 * number of earlier agents within reach is counted once per step and released as they move, one pair of launches per
 * level, so the result equals the one-by-one loop up to float32 pair arithmetic and summation order.
 * noise as in oc_gcfm_step (noise[i] = agent i's draw, whatever its place in the permutation).  rounds_out (host, may be
 * NULL): number of levels.  The levels are looped on the device: the host enqueues as many rounds as the previous step
 * needed (plus a margin), rounds past the last level return at once, and it synchronises ONCE per step (again only if the
 * step turns out deeper than the last one).  For trajectory-level parity with the reference; oc_gcfm_step (synchronous)
 * is the fast path. */
int oc_gcfm_step_sequential(oc_handle* h, const oc_agents* a, int32_t simu_step, const int32_t* perm, const double* noise,
                            uint64_t seed, int32_t* counters, int32_t* rounds_out, void* stream);
/* The step on `world` GPUs with REPLICATED agents: every rank holds the whole agent state and calls this
 * with the same arguments.  Rank r computes the pair forces and the wall forces of ITS SHARE of the tiles (tile = 64 agents
 * of the spatial order; share = ceil(tiles / world) consecutive tiles) against all agents, and the desired velocities of
 * the agents whose floor field it can answer.  The results travel through the caller's exchange buffer `xchg`
 * (oc_xchg_doubles(N, world) float64 values):
 *   - `world` blocks, one per rank, of the share's tiles: per tile [pair sum x 64][wall force x 64]([desired velocity
 *     x 64]) double2, then 8 doubles whose first is the rank's IndexError-band word.  `allgather(user, xchg, n)` must
 *     gather them IN PLACE on `stream`: block r (n doubles at xchg + r * n) is rank r's -- one RCCL all-gather, 32 or 48
 *     bytes per agent in total (the zero-padded sum-all-reduce of round 2 moved 48 B per agent TWICE over the wire);
 *   - when the floor fields are sharded -- row blocks (oc_set_field_rows / oc_set_field_ring with a row block) or keys
 *     dealt to different ranks (no field registered here for some key) -- a desired velocity is answered by the rank that
 *     holds the field rows, not by the tile's owner: those 16 bytes per agent go through `allreduce_sum(user, buf, count)`
 *     (in place SUM on `stream`; one non-zero contribution per element, so the order of the sum does not matter).  With
 *     every field whole on every rank the tile's owner answers, they ride in its block and allreduce_sum is not called.
 * Then every rank finishes ALL agents from the gathered buffer: the replicated states stay bit-identical without another
 * exchange, and counters[1] (IndexError band) is the same on every rank. */
typedef struct oc_step_comm {
    int32_t rank, world;
    double* xchg;
    int (*allgather)(void* user, double* buf, int64_t doubles_per_rank);
    int (*allreduce_sum)(void* user, double* buf, int64_t count);
    void* user;
    /* >= 0: both collectives are issued by the library itself over the RCCL communicator of this slot (oc_comm_create:
     * rank / world must be the communicator's) and the callbacks may be NULL; -1: the callbacks */
    int32_t comm_slot, reserved;
} oc_step_comm;
int64_t oc_padded_agents(int32_t N);
int64_t oc_xchg_doubles(int32_t N, int32_t world);
int oc_gcfm_step_sharded(oc_handle* h, const oc_agents* a, int32_t simu_step, const double* noise, uint64_t seed,
                         int32_t* counters, const oc_step_comm* comm, void* stream);
/* ---- RCCL inside the library -------------------------------------------------------------------------
 * Communicators the library owns, so that the collectives above leave from C on the launch stream -- no host callback on
 * the launch path.  RCCL is bound at run time (dlopen librccl.so.1); a single-GPU user never needs it.  Per group of
 * ranks: one of them calls oc_comm_unique_id and hands the 128 bytes to the others by any means (torch.distributed
 * broadcast, MPI, a file); then EVERY member calls oc_comm_create (collective) with its rank in the group.  `slot`
 * (0 .. 15) names the communicator in oc_hjb_shard.comm_slot / oc_step_comm.comm_slot.  Halo rows go as one grouped
 * send/recv pair per neighbour (xGMI is point-to-point). */
#define OC_COMM_ID_BYTES 128
/* OC_OK when librccl can be bound in this process (dlopen + symbols only: no id is drawn, no bootstrap listener started) */
int oc_comm_available(void);
int oc_comm_unique_id(void* id_out);
int oc_comm_create(oc_handle* h, int slot, const void* id, int rank, int world);
int oc_comm_destroy(oc_handle* h, int slot);

/* For a driver that does not read `counters` after every step (run() enqueues steps and looks every 16th):
 * out4 (host) = { active agents now, 1 + the step in which the last active agent left (0: somebody is inside),
 * 1 + index of an agent that was in the IndexError band (0: none), the first step in which it was }.  Synchronises;
 * reset != 0 clears the last three afterwards.  simulations.py:427 (loop condition), optimals.py:234-239 (IndexError). */
int oc_run_status(oc_handle* h, int32_t* out4, int reset, void* stream);
/* oc_set_tuning(h, "lazy_state", 1): oc_gcfm_step no longer scatters x / y / vx / vy / agent_time into the caller's arrays in
 * every step (five 8-byte stores per agent at random addresses: more than half of the step's second launch at 500 000
 * agents) -- the state lives in the library's slot-ordered snapshot, an agent that leaves is written when it does, and the
 * arrays are brought up to date by oc_agents_sync, by the periodic re-sort and by every other entry point that reads them
 * (oc_density, the diagnostic exports, the sequential and the sharded step).  `status` is always current.  Call
 * oc_agents_sync BEFORE reading the arrays yourself, and before writing them (then oc_agents_changed, as ever).  Default 0:
 * every step leaves the arrays current (simulations.py:303-326 writes the agent's own fields). */
int oc_agents_sync(oc_handle* h, void* stream);
int oc_set_resort_every(oc_handle* h, int32_t steps);
/* Between consecutive oc_gcfm_step calls the library reuses the packed snapshot it wrote itself; call this after
 * modifying the state arrays in place from outside (new array pointers are detected automatically). */
int oc_agents_changed(oc_handle* h);

/* d = sum over active agents of exp(-r^2/(2 sigma^2)) / (2 pi sigma), zero where V_all < 0:
 * simulations.py:453-487.  out: dtype (Ny,Nx); V_all float32 (Ny,Nx) (simulation.V) or NULL.
 * trunc_sigmas: contributions beyond this many sigmas are dropped (<= 0: 6.0). */
int oc_density(oc_handle* h, int dtype, const oc_agents* a, double sigma, const float* V_all, double trunc_sigmas,
               void* out, void* stream);
/* The same for grid rows [row_lo, row_hi) only -- `out` is still the (Ny, Nx) array, the other rows are left as they are: a
 * rank of a row-block solve (oc_hjb_solve_rows) needs the density under its own rows (+ halo) only; the agents are
 * replicated, so this needs no communication. */
int oc_density_rows(oc_handle* h, int dtype, const oc_agents* a, double sigma, const float* V_all, double trunc_sigmas,
                    int32_t row_lo, int32_t row_hi, void* out, void* stream);

/* ---- introspection for bench.py / tests --------------------------------------------------------------------- */
/* With profiling enabled every launch of a named kernel class is bracketed by HIP events on its own stream.
 * oc_kernel_stats returns, since the last reset: number of launches, summed event time (ms) and summed
 * bytes the launch is DESIGNED to move (DESIGN.md section "Rooflines").  Classes: "hjb_stage" (the stage launch(es) of
 * one RK attempt: 8 array passes -- 3 read, 5 written; with the interpolant ring 10 -- 3 read, 7 written), "hjb_list" (ring solves:
 * the work-list builder's launch), "hjb_out" (dense output + direction-field writer), "gcfm_forces", "gcfm_finish" (the
 * two launches of a step), "density".  The HJB classes are sampled (every 8th attempt).  Synchronises. */
/* mode bits: 1 = HJB kernels, 2 = GCFM kernels every step, 4 = GCFM kernels every 16th step; 0 = off */
int oc_set_profiling(oc_handle* h, int mode);
int oc_kernel_stats(oc_handle* h, const char* name, int64_t* launches, double* total_ms, double* total_bytes);
int oc_reset_kernel_stats(oc_handle* h);
/* 0 = the product path (default): all six stages of an RK attempt in one launch, step controller on the device (no host
 *     round trip per attempt); 1 = simple one-array-pass kernels, 2 = one fused launch per stage -- both with the step
 *     controller on the host, kept as cross-checks (the arithmetic per cell of all three is identical). */
int oc_set_hjb_variant(oc_handle* h, int variant);
/* Launch-shape knobs of the step kernels (results do not depend on them beyond float64 summation order):
 *   "pair_split"  1, 2 or 4 pair wavefronts per tile of 64 agents (default 4)
 *   "prep_blocks" upper limit of the blocks of the force kernel that do the wall search / field lookup of the tiles
 *                 (default 256; the launch uses the fewest blocks that give every prep wavefront the same tile count)
 *   "resort_every" same as oc_set_resort_every
 *   "seq_margin_mm" displacement bound M of oc_gcfm_step_sequential in millimetres (default 250)
 *   "out_tile"     tile of the dense-output / direction-field writer: -1 = by grid size (default), 0 = 64 x 16,
 *                  1 = 128 x 8, 2 = 256 x 4 cells
 *   "hjb_fuse"     stage launches of an RK attempt of the product path: 0 = one launch of six stages (default),
 *                  1 = two launches of three
 *   "lazy_state"   see oc_agents_sync (default 0)
 *   "wall_warm"    1 (default) = the exact nearest-wall search of an agent (pedestrians.py:311-315) starts from the cell the
 *                  previous step's search found -- a valid candidate, so the result is the same cell; 0 = always cold
 *   "hjb_tail"     ring solves (oc_hjb_solve_ring): 1 (default) = while few blocks take part in an attempt -- small grids, or
 *                  the active set's short lists -- the LAST block of the stage launch to finish is the attempt's step
 *                  controller and list builder (one launch per attempt instead of three); 0 = always separate launches;
 *                  2 = the fused launch at any number of blocks (two-level tickets; measured slower when every tile is
 *                  computed, kept for that comparison).  Same decisions, same results either way.
 *   "hjb_lead"     sparse ring solves (exact active set, far-field substitution): 1 = an attempt's step controller runs at the
 *                  START of the next attempt's launch, redundantly in every block, from the listed tiles' records -- no
 *                  ticket, no last block; the block at list position 0 writes the controller block, the ring's records and
 *                  the mailbox.  0 (default): measured equal to the fused tail at C4 (1.88 against 1.85 ms per solve).  Same
 *                  decisions, same results either way.
 *   "hjb_subst"    ring solves with the exact active set: 1 (default) = the state buffers do not maintain the cells of trivial
 *                  tiles -- a load of such a cell yields the far-field value the tile stands for -- so a change of scale
 *                  does not put every tile on the work list to rewrite the far field; 0 = every buffer holds every tile.
 *                  Same decisions, same results either way.
 *   "hjb_active"   exact active set of oc_hjb_solve: tiles of the grid on which phi is one constant power of two, K0 = 0
 *                  and coef = 0 -- the far field of a solve with m = 0, where optimals.py:144-164 evaluates to exactly 0
 *                  -- are not computed; every slice of the direction field is still written, and fields, step sequence
 *                  and statistics are bit-identical to computing every tile.  0 = off, 1 = on for grids of at least
 *                  96 stage tiles of 112 x 20 cells (default), 2 = on at any size (both: up to 12 000 tiles).  Row-block
 *                  solves (oc_hjb_solve_rows*) use it per rank ("hjb_active_rows", default 1). */
int oc_set_tuning(oc_handle* h, const char* name, int value);

/* Opt-in corrections of reference quirks (SURVEY.md quirk register, section 8f-4).  Default 0: every entry point is
 * bug-compatible with the reference.  Each bit replaces one documented oddity by what the model intends:
 *   OC_FIX_LOOKUP        optimals.py:234-248 (Q7, Q8): the desired direction at (x, y) is the bilinear interpolation of
 *                        the direction field over the four grid NODES around the position (node j at X0[j]); the
 *                        reference indexes the interior array with grid indices (a one-cell shift), averages the
 *                        diagonal pair only, and raises IndexError in a band along the far borders -- here positions are
 *                        clamped to the interior nodes and no error flag is raised.
 *   OC_FIX_WALL_FORCE    pedestrians.py:328,334 (Q12): q_i = 1/sqrt(.) (the reference omits the root), force
 *                        -rep * e along the UNIT vector to the nearest wall cell, without the factor 3.
 *   OC_FIX_PAIR_FORCE    pedestrians.py:280 (Q12): agent repulsion -rep * e along the unit vector (reference: -rep * R).
 *   OC_FIX_DENSITY_NORM  simulations.py:482 (Q11): Gaussian normaliser 2 pi sigma^2 (reference: 2 pi sigma).
 * (Q9, the shrinking horizon of a recompute, is host logic: the caller passes the full horizon to oc_hjb_solve --
 * optimal_crowds_amd.simulations.simulation(fixes=...) does.) */
#define OC_FIX_LOOKUP 1u
#define OC_FIX_WALL_FORCE 2u
#define OC_FIX_PAIR_FORCE 4u
#define OC_FIX_DENSITY_NORM 8u
#define OC_FIX_ALL 15u
int oc_set_fix_flags(oc_handle* h, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif
