/*
 * safe_control_amd.h -- C-ABI of the MI355X batched CBF-QP / MPC-CBF solve engine.
 *
 * The reference (tkkim-robot/safe_control) has no FFI: its hot path sits
 * behind a duck-typed Python plugin protocol (tracking.py:140-154 picks
 * `pos_controller`; tracking.py:611-616 calls
 * `pos_controller.solve_control_problem(robot.X, control_ref, obs)`).  This
 * header is therefore the *new* boundary a maintainer binds with ctypes (see
 * INTEGRATION.md); every entry point cites the reference interface it
 * replaces.  Plain pointers and sizes only; no exceptions cross the ABI; all
 * functions return an `sc_error` (0 = ok).
 *
 * Memory: all `*_batch` entry points take DEVICE pointers (HBM resident, as
 * handed out by hipMalloc / torch) and enqueue on `stream` (a hipStream_t, may
 * be NULL = default stream) without synchronising.  The `*_host` twins take
 * HOST pointers, stage through device buffers and synchronise before
 * returning (single-agent drop-in use).  Caller owns every buffer.
 * Thread safety: calls on distinct streams are independent; the library
 * keeps no mutable global state besides a thread-local last-error string.
 *
 * Array layouts are exactly the reference's numpy layouts, row-major:
 *   X      [B, 4]     state  [x, y, theta, v]          (robot.X, robots/robot.py:38; [B, 6] for Quad2D)
 *   u_ref  [B, 2]     nominal input                    (control_ref['u_ref'], tracking.py:607-609)
 *   obs    [B, K, 7]  obstacle rows [x,y,r,vx,vy,-,flag] or [ox,oy,a,b,e,theta,1]
 *                     (nearest_multi_obs, tracking.py:584); [K, 7] when obs_shared != 0
 *   n_obs  [B] int32  optional per-agent count of valid rows (<= K); NULL = all K.
 *                     Rows >= n_obs[i] are "0*u + 0 >= 0" like the reference's
 *                     zero-initialised A1/b1 rows (position_control/cbf_qp.py:110-111).
 */
#ifndef SAFE_CONTROL_AMD_H
#define SAFE_CONTROL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: the minor number goes up with EVERY change of a struct layout or an entry point's signature (round 4 grew
 * sc_resto_params and put slack_reset into sc_mpccbf_params / sc_mpclin_params: 0.2; round 5 added sc_ipopt_params and sc_mpcvtol_ms_solve_batch: 0.3; the continuation entry points of the optimal-decay families: 0.4; sc_odmpcvtol_ms_solve_batch: 0.5; round 6 added sc_mpccbf_ms_solve_batch: 0.7; sc_mpccbf_ms_workspace_bytes and sc_ipopt_params.floor_iter in the slot that was `reserved`: 0.8; sc_mpccbf_params.v_min / rear_ax_dist for the KinematicBicycle2D instantiation of sc_mpccbf_ms_solve_batch: 0.9).  A binding compares sc_version() with the
 * version its struct mirrors were written for before the first call (safe_control_amd/_lib.py: ABI_VERSION). */
#define SC_VERSION_MAJOR 0
#define SC_VERSION_MINOR 9

/* ---- return codes ------------------------------------------------------ */
typedef enum sc_error {
    SC_OK = 0,
    SC_ERR_INVALID_ARGUMENT = 1,   /* NULL pointer, bad K/B/dtype/model id            */
    SC_ERR_UNSUPPORTED = 2,        /* K (or horizon) above the compiled maximum       */
    SC_ERR_HIP = 3,                /* a HIP runtime call failed (see sc_last_error)   */
    SC_ERR_NO_DEVICE = 4           /* no gfx950 device visible                        */
} sc_error;

/* ---- per-problem solve status (int32 in status_out) --------------------
 * Mapped by the Python shim to the strings tracking.py:628 compares with
 * 'optimal' ('optimal' / 'infeasible' / 'optimal_inaccurate').             */
#define SC_STATUS_OPTIMAL      0
#define SC_STATUS_INFEASIBLE   1   /* MPC-CBF: certified by a converged feasibility restoration (sc_resto_params) */
#define SC_STATUS_INACCURATE   2   /* MPC-CBF only: stopped without convergence (iteration limit, line search)    */
/* (-1 is used INSIDE sc_cbfqp_solve_batch between the two launches of its large-batch DynamicUnicycle2D path -- "pending: left to the
 * generic pass" -- and is overwritten by the second launch; if that launch fails the call returns an error code and status_out,
 * u_out and h_out are undefined, like after any failed call.)                                                                     */
#define SC_STATUS_BAD_OBSTACLE 3   /* obstacle flag not 0/1 for a model that needs it
                                      (the reference raises inside agent_barrier,
                                      robots/dynamic_unicycle2D.py:133-136)            */

/* ---- robot models: robots/robot.py:65-175 dispatch ---------------------- */
#define SC_MODEL_DYNAMIC_UNICYCLE2D        0  /* robots/dynamic_unicycle2D.py                    */
#define SC_MODEL_KINEMATIC_BICYCLE2D       1  /* robots/kinematic_bicycle2D.py                   */
#define SC_MODEL_KINEMATIC_BICYCLE2D_C3BF  2  /* dynamic_env/kinematic_bicycle2D_c3bf.py         */
#define SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF 3  /* dynamic_env/kinematic_bicycle2D_dpcbf.py        */
#define SC_MODEL_SINGLE_INTEGRATOR2D       4  /* robots/single_integrator2D.py: X = [x, y, -, -], U = [vx, vy]   */
#define SC_MODEL_DOUBLE_INTEGRATOR2D       5  /* robots/double_integrator2D.py: X = [x, y, vx, vy], U = [ax, ay] */
#define SC_MODEL_QUAD2D                    6  /* robots/quad2D.py: X = [x, z, theta, vx, vz, theta_dot] (state_dim 6), U = [F_right, F_left] */
#define SC_MODEL_UNICYCLE2D                 7  /* robots/unicycle2D.py: X = [x, y, theta, -], U = [v, omega] (rel-deg 1) */
#define SC_MODEL_COUNT                     8

#define SC_DTYPE_F32 0
#define SC_DTYPE_F64 1

#define SC_CBF_MODE_CBF  0   /* robot_spec['cbf_mode'] == 'cbf'  (cbf_qp.py:120)  */
#define SC_CBF_MODE_HARD 1   /* robot_spec['cbf_mode'] == 'hard' (cbf_qp.py:158-161,170-177) */

#define SC_CBFQP_MAX_OBS 32  /* largest K the CBF-QP kernels are instantiated for */

/* Parameters of one CBFQP controller instance: what CBFQP.__init__ /
 * setup_control_problem read from robot_spec and cbf_param
 * (position_control/cbf_qp.py:5-45, 47-106).                               */
typedef struct sc_cbfqp_params {
    int32_t model_id;        /* SC_MODEL_*                                            */
    int32_t io_dtype;        /* SC_DTYPE_*: element type of X,u_ref,obs,u_out,h_out   */
    int32_t compute_dtype;   /* SC_DTYPE_*: arithmetic type inside the kernel         */
    int32_t cbf_mode;        /* SC_CBF_MODE_*                                         */
    int32_t obs_shared;      /* 0: obs is [B,K,7]; 1: one [K,7] table for all agents  */
    int32_t state_dim;       /* row length of X: 0 or 4 for the 4-state models, 6 for Quad2D */
    double  robot_radius;    /* robot.robot_radius (robots/robot.py:49-50)            */
    double  dt;              /* robot.dt, used by 'hard' mode only                    */
    double  alpha1;          /* cbf_param['alpha1'] (rel-deg 2) or ['alpha'] (rel-deg 1) */
    double  alpha2;          /* cbf_param['alpha2'] (rel-deg 2), unused otherwise     */
    double  u_min[2];        /* input box, cbf_qp.py:62-65 / :70-73                   */
    double  u_max[2];
    double  rear_ax_dist;    /* robot_spec['rear_ax_dist'] (KinematicBicycle2D family) */
    double  mass;            /* robot_spec['mass'] (Quad2D, robots/quad2D.py:41)      */
} sc_cbfqp_params;

/* ---- library ------------------------------------------------------------ */
int         sc_version(void);                 /* major*1000 + minor                        */
const char* sc_last_error(void);              /* thread-local message of the last failure   */
int         sc_device_count(int* count_out);  /* number of visible HIP devices              */

/* ---- CBF-QP -------------------------------------------------------------
 * Replaces, for a whole batch of agents in one launch:
 *   CBFQP.solve_control_problem(robot_state, control_ref, obs_list)
 *     position_control/cbf_qp.py:108-199  (row assembly :120-183, solve :190, status :195)
 *   robot.f()/g()/agent_barrier(obs)   robots/robot.py:389-436 -> robots/<model>.py
 *   cvxpy -> GUROBI solve of  min ||u-u_ref||^2  s.t. A1 u + b1 >= 0, box
 *
 * u_out [B,2]: minimiser (NaN where status != OPTIMAL; the reference returns None).
 * status_out [B] int32: SC_STATUS_*.
 * h_out [B,K] or NULL: barrier value h(x) per obstacle row (0 for rows >= n_obs[i]).
 * Element type of X/u_ref/obs/u_out/h_out is params->io_dtype.
 */
int sc_cbfqp_solve_batch(const sc_cbfqp_params* params, int64_t B, int32_t K,
                         const void* X, const void* u_ref, const void* obs,
                         const int32_t* n_obs,
                         void* u_out, int32_t* status_out, void* h_out,
                         void* stream);

/* Same computation, HOST pointers; copies in/out and synchronises. */
int sc_cbfqp_solve_batch_host(const sc_cbfqp_params* params, int64_t B, int32_t K,
                              const void* X, const void* u_ref, const void* obs,
                              const int32_t* n_obs,
                              void* u_out, int32_t* status_out, void* h_out,
                              int device);

/* ---- MPC-CBF ------------------------------------------------------------
 * Parameters of one MPCCBF controller instance: what MPCCBF.__init__ reads
 * from robot_spec (position_control/mpc_cbf.py:7-100).                     */
#define SC_MPCCBF_MAX_HORIZON 32

/* Feasibility restoration of the interior point behind every sc_mpc*_solve_batch (Waechter & Biegler 2006, section 3.3, on
 * the condensed problem; the reference's solver is IPOPT via do-mpc, position_control/mpc_cbf.py:163,384, whose restoration
 * phase this stands in for).  When the regular iteration cannot continue at an infeasible iterate z_R (line search failed,
 * multipliers above 1e10) the kernels minimise  rho * sum_i t_i + sqrt(mu)/2 |z - z_R|^2  s.t.  g_i(z) + t_i >= 0, t_i >= 0
 * over the CBF rows i (state bounds and the input box stay hard) with the same primal-dual iteration, return to the regular
 * phase once the l1 violation has dropped to kappa * violation(z_R), and report SC_STATUS_INFEASIBLE only when the restoration
 * CONVERGES with a violation above theta_tol: the returned input is then that minimiser of the violation -- or when it STALLS
 * with violation left (retry_max, stall_iter, stall_theta below; round 4).  Any other
 * unsuccessful exit is SC_STATUS_INACCURATE.  oracle/mpc_cbf.py: solve is the float64 statement of the same steps.       */
typedef struct sc_resto_params {
    double  rho;             /* l1 penalty of the elastic variables (IPOPT: 1000)                                    */
    double  kappa;           /* first entry: back to the regular phase at violation <= kappa * violation(z_R) (0.1)           */
    double  theta_tol;       /* a converged restoration with l1 violation above this is SC_STATUS_INFEASIBLE (1e-6) */
    double  tol;             /* KKT tolerance of the restoration problem (1e-2, in units of its objective rho * violation:
                              * |grad violation| <= 1e-5); the certificate also needs violation > 10 * error / rho      */
    double  small_alpha;     /* the regular phase also hands over after small_iter consecutive accepted steps shorter  */
    int32_t small_iter;      /*   than small_alpha at an infeasible iterate (0.02, 4): IPOPT's alpha_min rule          */
    int32_t max_entries;     /* restoration may be entered this many times per solve (2); 0 disables it             */
    int32_t slack_reset;     /* 1: the restoration's line search sets the slack of a row to g + t where that is >= mu / nu, the
                              * minimiser of its merit function in s for fixed z, t (default; turns most restorations that crawled
                              * to the iteration limit into certificates); 0: off                                     */
    int32_t retry_max;       /* a restoration step whose line search fails is retried from the same iterate with a Levenberg-damped
                              * Newton system, delta >= 1, 1e2, 1e4, ... (each retry is one iteration), this many times before the
                              * solve gives up (3); 0: give up at once (round 3)                                       */
    double  stall_theta;     /* a restoration that has not lowered the l1 violation by 1 % within stall_iter iterations while the  */
    int32_t stall_iter;      /*   violation is above stall_theta stops with SC_STATUS_INFEASIBLE (below it: stops too, SC_STATUS_INACCURATE): a local minimiser of the violation
                              *   at a kink of the rows, where no KKT error goes to zero (1e-3, 40); stall_iter = 0: off.
                              *   The one-NLP-per-lane VTOL2D cross-check kernel supports neither (retry_max = stall_iter = 0).  */
    int32_t gauss_newton;    /* 1: the restoration's Newton system is J' Sigma J + zeta I -- the second-order terms of the rows (multipliers
                              * near rho) and of the dynamics are dropped.  VTOL2D (sc_mpcvtol_*) only, where the exact Hessian of the
                              * restoration's Lagrangian is so indefinite that its steps shrink to 1e-3 (default there: 1); the other
                              * entry points require 0                                                                   */
} sc_resto_params;

/* Continuation launches of the interior point (sc_mpc*_solve_batch_sliced; csrc/mpc_cont.hpp).  The reference hands its NLPs to
 * IPOPT with default options (position_control/mpc_cbf.py:163-173: max_iter = 3000); one launch ends with its slowest problem, so
 * a budget of that size is only affordable when the launch that serves the whole batch stops at a much smaller cap and hands the
 * few unfinished solves -- WITH their solver state (iterate, slacks, multipliers, elastic variables, barrier parameter, merit
 * penalty, restoration flags, counters: one record per problem in `workspace`) -- to the next launch, which continues them with
 * the instructions an uninterrupted solve would have executed (resumed == uninterrupted bit for bit).
 *   launches:  [classify]  ->  solve to it_stop[0]  ->  ... ->  solve to it_stop[n_caps - 1]  ->  solve to params->max_iter
 * `order`: pending solves whose CBF rows are violated at the hand-over (the ones that crawl or restore feasibility) start first in
 * the next launch.  `classify_first`: a launch that only evaluates the initial guess sorts the batch the same way before the first
 * solve launch (a launch ends with its slowest problem: the long solves have to start in its first scheduling round).
 * Every status the caller sees is final: SC_STATUS_PENDING is only written between the launches of one call.                     */
#define SC_STATUS_PENDING    (-1)
#define SC_MPC_MAX_SLICES    8
typedef struct sc_mpc_slices {
    int32_t n_caps;                       /* launches before the last one: 0 .. SC_MPC_MAX_SLICES                              */
    int32_t it_stop[SC_MPC_MAX_SLICES];   /* their iteration caps (cumulative, strictly increasing, >= 1; caps >= max_iter are dropped) */
    int32_t order;                        /* 0 / 1                                                                               */
    int32_t classify_first;               /* 0 / 1                                                                               */
    int32_t reserved;
    void*   workspace;                    /* device memory of sc_mpc*_slices_workspace_bytes(...) bytes, no initialisation needed */
    size_t  workspace_bytes;
} sc_mpc_slices;

typedef struct sc_mpccbf_params {
    int32_t model_id;        /* SC_MODEL_DYNAMIC_UNICYCLE2D or SC_MODEL_UNICYCLE2D (others: SC_ERR_UNSUPPORTED).
                              * Unicycle2D (robots/unicycle2D.py): inputs [v, omega], u_max = (v_max, w_max),
                              * one-step CBF rows h(p_k+1) - (1 - alpha1) h(p_k) >= 0 (mpc_cbf.py:312-315,
                              * unicycle2D.py:127-145), Q[3], alpha2, v_max and X[:,3] unused               */
    int32_t io_dtype;        /* SC_DTYPE_*: element type of X,u_prev,goal,obs,u_out,z_out          */
    int32_t horizon;         /* robot_spec['mpc_horizon'], default 10 (mpc_cbf.py:15)              */
    int32_t max_iter;        /* interior-point iteration limit (-> SC_STATUS_INACCURATE)           */
    int32_t obs_shared;      /* 0: obs is [B,K,7]; 1: one [K,7] table for all agents               */
    int32_t acceptable_iter; /* stop after this many consecutive iterations within acceptable_tol (IPOPT's
                              * acceptable_iter, 15); 0 = 15                                       */
    int32_t slack_reset;     /* line search of the regular phase: 0 off (MPCCBF of both unicycles and the DynamicUnicycle2D
                              * optimal-decay class: measured to change nothing there); 2: s = g where g >= mu / nu after a
                              * trial step -- the setting of the config-5 extension (Unicycle2D optimal decay, N = 20,
                              * superellipsoids), oracle/od_mpc_rd1.py                              */
    int32_t superellipsoid_rows; /* sc_mpccbf_ms_solve_batch only: 0 = every obstacle row is a circle (flag column < 0.5; the kernel does not look);
                              * 1 = rows may be superellipsoids [ox, oy, a, b, e, theta, 1] -- DynamicUnicycle2D and DoubleIntegrator2D, whose
                              * DT barriers have that branch (dynamic_unicycle2D.py:204-220): a slower instantiation                       */
    double  dt;              /* robot.dt                                                           */
    double  Q[4];            /* diagonal state weights, DU: 50,50,.01,30 (mpc_cbf.py:25-27)        */
    double  R[2];            /* input-rate weights of mpc.set_rterm, DU: .5,.5 (mpc_cbf.py:180)    */
    double  alpha1, alpha2;  /* DT-CBF gains, DU: .15,.15 (mpc_cbf.py:56-59)                       */
    double  v_max;           /* |x[3]| <= v_max on every stage (mpc_cbf.py:193-195)                */
    double  u_max[2];        /* |a| <= a_max, |omega| <= w_max (mpc_cbf.py:196-199)                */
    double  robot_radius;
    double  beta;            /* barrier inflation, 1.01 (dynamic_unicycle2D.py:188)                */
    double  tol;             /* KKT tolerance on the gradient-scaled problem (1e-6)                */
    double  acceptable_tol;  /* accepted when the iteration stalls at the precision limit (1e-5)   */
    double  mu_init;         /* initial barrier parameter (0.1, IPOPT's default)                   */
    double  mu_min;          /* smallest barrier parameter (1e-9)                                  */
    sc_resto_params resto;   /* feasibility restoration (not used by the optimal-decay entry points) */
    double  v_min;           /* sc_mpccbf_ms_solve_batch with SC_MODEL_KINEMATIC_BICYCLE2D: lower end of robot.step's speed clip
                              * (kinematic_bicycle2D.py:116-121; v_max is the upper end and the state bound); unused elsewhere */
    double  rear_ax_dist;    /* ... and L_r of x+ = x + dt (.., v beta / L_r, a) (kinematic_bicycle2D.py:67-110)            */
} sc_mpccbf_params;

/* Replaces, for a whole batch of agents in one launch:
 *   MPCCBF.solve_control_problem(robot_state, control_ref, nearest_obs)   mpc_cbf.py:366-402
 *     mpc.x0 = x; mpc.set_initial_guess(); update_tvp(goal, obs); mpc.make_step(x)  (IPOPT)
 * X [B,4], u_prev [B,2] (last applied input = do-mpc's u0, zeros on the first call),
 * goal [B,2], obs [B,K,7] already padded with [1000,1000,0,...] rows like update_tvp
 * (mpc_cbf.py:338-364).  u_out [B,2] first move of the horizon; status_out [B] SC_STATUS_*;
 * iters_out [B] or NULL; z_out [B, 2*horizon] or NULL (whole input sequence).
 * One NLP per wavefront; arithmetic is always f64.
 */
int sc_mpccbf_solve_batch(const sc_mpccbf_params* params, int64_t B, int32_t K,
                          const void* X, const void* u_prev, const void* goal, const void* obs,
                          void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out,
                          void* stream);

/* The same solve as a sequence of continuation launches (sc_mpc_slices above); slices == NULL or an empty schedule: one launch. */
size_t sc_mpccbf_slices_workspace_bytes(const sc_mpccbf_params* params, int64_t B, int32_t K);
int sc_mpccbf_solve_batch_sliced(const sc_mpccbf_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K,
                                 const void* X, const void* u_prev, const void* goal, const void* obs,
                                 void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out,
                                 void* stream);

int sc_mpccbf_solve_batch_host(const sc_mpccbf_params* params, int64_t B, int32_t K,
                               const void* X, const void* u_prev, const void* goal, const void* obs,
                               void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out,
                               int device);

/* ---- MPC-CBF for the reference's linear models (SingleIntegrator2D, Quad3D) ---------------
 * MPCCBF (position_control/mpc_cbf.py:7-402) over a robot whose f(x) = A x and g(x) = B are constant
 * (robots/single_integrator2D.py:45-62; robots/quad3D.py:77-119, MPC only: its CBF-QP barrier raises, quad3D.py:269-273):
 *   prediction  x+ = x + (f + g u) dt = Ae x + Be u                                   (mpc_cbf.py:135-141)
 *   cost        sum_{k=1..N} (x_k - xg)' diag(Q) (x_k - xg), xg = [goal, 0, ..] (:144,176-178,267) + r-term R on delta u (:180)
 *   CBF rows    h(step(x_k, u_k)) - (1 - alpha) h(x_k) >= 0 per stage and obstacle  (:312-315), step = the robot's own
 *               one-step map  As x + Bs u  (Euler for SI; RK4 of the linear system for Quad3D, quad3D.py:121-158);
 *               h = circle distance barrier (Quad3D: circles only, quad3D.py:283-291; SI also superellipsoids)
 *   bounds      u_lo <= u <= u_hi                                                     (:183-187, :219-223)
 * Every barrier point is affine in the inputs, so the condensed cost Hessian and the point Jacobian are constants of
 * the controller: sc_mpclin_build_model computes them once on the HOST from (Ae, Be, As, Bs) (row-major nx x nx,
 * nx x nu, nx x nx, nx x nu) into a blob of sc_mpclin_model_doubles doubles that the caller keeps in DEVICE memory
 * for sc_mpclin_solve_batch (HOST memory for the _host twin).
 * X [B,nx], u_prev [B,nu], goal [B,ng], obs [B,K,7] (or [K,7]) padded like update_tvp; u_out [B,nu], status_out [B],
 * iters_out [B] or NULL, z_out [B, nu*horizon] or NULL.  One NLP per wavefront, f64 arithmetic.
 */
typedef struct sc_mpclin_params {
    int32_t io_dtype;        /* SC_DTYPE_*: element type of X,u_prev,goal,obs,u_out,z_out                 */
    int32_t nx, nu, ng;      /* states (<= 12), inputs (<= 4), goal entries (2: SI, 3: Quad3D, mpc_cbf.py:80-81) */
    int32_t horizon;         /* robot_spec['mpc_horizon'], default 10; nu * horizon <= 128 and the LDS fit (Quad3D: N <= 20) */
    int32_t max_iter, obs_shared, acceptable_iter;   /* as sc_mpccbf_params                              */
    int32_t circles_only;    /* 1: the model's barrier has no superellipsoid branch (Quad3D)               */
    int32_t optimal_decay;   /* 0: MPCCBF.  1: EXTENSION (BASELINE config 5, no reference counterpart; Quad3D only): one
                                decay variable rho_k per stage, rows h(step) - (1 - alpha rho_k) h(x_k) >= 0 -- the
                                rel-degree-1 form of optimal_decay_cbf_qp.py:96-101,113-125 -- cost + od_p_sb (rho_k -
                                od_omega_ref)^2, r-term R u^2 (optimal_decay_mpc_cbf.py:178-184); oracle/od_mpc_rd1.py.
                                The model blob must be built with the same flag (its cost Hessian differs).
                                2: the semantics the reference's OptimalDecayMPCCBF gives the models of its rel-degree-1 branch
                                (Quad3D: optimal_decay_mpc_cbf.py:284-287): the PLAIN row d_h + alpha h_k -- the decay inputs omega1,
                                omega2 exist in the model but touch no row, their penalty keeps them at their reference -- with that
                                class's input term R u^2 (:173-179) instead of MPCCBF's delta-u penalty; solved through
                                sc_mpclin_solve_batch (restoration and continuation launches included).                        */
    int32_t slack_reset;     /* line search of the regular phase: 0 off; 2: s = g where g >= mu / nu after a trial step (as in
                                sc_mpcgn_params; the oracle's setting for the linear models since round 4: Quad3D at N = 20 crawled
                                for hundreds of iterations without it; optimal_decay = 1, the config-5 extension, uses it too).     */
    int32_t reserved;
    double  alpha;           /* DT-CBF gain: SI 0.05 (mpc_cbf.py:48-50), Quad3D 0.15 (:77-78)              */
    double  robot_radius, beta, tol, acceptable_tol, mu_init, mu_min;   /* as sc_mpccbf_params            */
    double  Q[12];           /* diagonal state weights (mpc_cbf.py:19-20, :37-38)                          */
    double  R[4];            /* input-rate weights (mpc_cbf.py:21, :39)                                    */
    double  u_lo[4], u_hi[4];
    double  od_omega_ref, od_p_sb;   /* optimal_decay = 1: reference 1.0 and penalty 10.0 (optimal_decay_mpc_cbf.py:88-89) */
    sc_resto_params resto;   /* feasibility restoration (optimal_decay = 0 only)                           */
} sc_mpclin_params;

size_t sc_mpclin_model_doubles(int32_t nx, int32_t nu, int32_t horizon);
int sc_mpclin_build_model(const sc_mpclin_params* params, const double* Ae, const double* Be, const double* As,
                          const double* Bs, double* model_out);
int sc_mpclin_solve_batch(const sc_mpclin_params* params, const double* model, int64_t B, int32_t K,
                          const void* X, const void* u_prev, const void* goal, const void* obs,
                          void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);
/* continuation launches (sc_mpc_slices, above sc_mpccbf_params); optimal_decay = 0 only */
size_t sc_mpclin_slices_workspace_bytes(const sc_mpclin_params* params, int64_t B, int32_t K);
int sc_mpclin_solve_batch_sliced(const sc_mpclin_params* params, const sc_mpc_slices* slices, const double* model, int64_t B, int32_t K,
                                 const void* X, const void* u_prev, const void* goal, const void* obs,
                                 void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);
int sc_mpclin_solve_batch_host(const sc_mpclin_params* params, const double* model, int64_t B, int32_t K,
                               const void* X, const void* u_prev, const void* goal, const void* obs,
                               void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, int device);
/* optimal_decay = 1 (see sc_mpclin_params): as sc_mpclin_solve_batch plus rho_out [B, horizon] (or NULL), the decay variables. */
int sc_odmpclin_solve_batch(const sc_mpclin_params* params, const double* model, int64_t B, int32_t K,
                            const void* X, const void* u_prev, const void* goal, const void* obs,
                            void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);

/* ---- MPC-CBF for DoubleIntegrator2D, Quad2D and the KinematicBicycle2D family ----------------
 * MPCCBF (position_control/mpc_cbf.py:7-402) for the planar models whose rel-deg-2 DT-CBF steps the state with the robot's
 * own step(): x1 = step(x_k, u_k), x2 = step(x1, u_k) (double_integrator2D.py:222-272 with the speed rescaled to v_max
 * :79-107; quad2D.py:179-206), so the barrier points are functions of (x_k, u_k) and not predicted positions.  Weights /
 * gains / bounds of mpc_cbf.py: DI Q = diag(50,50,20,20), R = (.5,.5), alpha .2, |a| <= (ax_max, ay_max); Quad2D Q =
 * diag(25,25,50,10,10,50), R = (.5,.5), alpha .15, f_min <= u <= f_max.  Interior point as sc_mpccbf_solve_batch (exact
 * Hessian for Quad2D, Gauss-Newton for the linear DoubleIntegrator2D; oracle/mpc_gn.py).
 * KinematicBicycle2D (mpc_cbf.py:31-33,64-67,205-211; kinematic_bicycle2D.py:113-123,175-199): Q = diag(50,50,1,1), R = (.5, 5000),
 * alpha .1, beta 1.1, |a| <= a_max, |beta| <= beta_max, |v_k| <= v_max as state rows; step() clips the speed to [v_min, v_max].
 * KinematicBicycle2D_C3BF / _DPCBF (mpc_cbf.py:68-73,312-315; kinematic_bicycle2D_c3bf.py:77-118, ..._dpcbf.py:86-142): the same
 * dynamics, cost and bounds with the row d_h + alpha h_k of a barrier of the FULL state; the one gain (0.15) travels in alpha1,
 * alpha2 = 0, beta unused (1.01 / 1.05 are fixed in the barriers).  As in the reference's MPC the obstacle's velocity columns are
 * not read (the barrier gets a 1 x 7 row there: `obs.shape[0] > 3` is False); oracle/mpc_kb_state.py.
 * X [B,nx] (nx = 4, or 6 for Quad2D), u_prev [B,2], goal [B,2], obs [B,K,7] (or [K,7]) padded like update_tvp;
 * u_out [B,2], status_out [B], iters_out [B] or NULL, z_out [B, 2*horizon] or NULL.  One NLP per wavefront, f64 arithmetic.
 */
typedef struct sc_mpcgn_params {
    int32_t model_id;        /* SC_MODEL_DOUBLE_INTEGRATOR2D, SC_MODEL_QUAD2D, SC_MODEL_KINEMATIC_BICYCLE2D[_C3BF|_DPCBF] */
    int32_t io_dtype, horizon, max_iter, obs_shared, acceptable_iter;      /* as sc_mpccbf_params                  */
    int32_t circles_only;    /* 1: no superellipsoid branch in the model's DT barrier (KB, Quad2D)                 */
    int32_t slack_reset;     /* line search: 0 off; 2: s = g where g >= mu / nu after a trial step (the oracle's setting for the
                                KinematicBicycle2D family since round 3; as sc_mpcvtol_params, mode 1 is not offered here) */
    double  dt;
    double  Q[6], R[2];      /* mpc_cbf.py:28-36                                                                   */
    double  alpha1, alpha2;  /* mpc_cbf.py:60-76 (C3BF / DPCBF: alpha, 0)                                          */
    double  u_lo[2], u_hi[2];
    double  v_min, v_max;    /* DI: speed rescaling of step() (v_max); bicycles: clip of step(), state bound |v| <= v_max */
    double  rear_ax_dist;    /* bicycles: L_r                                                                      */
    double  mass, inertia;   /* Quad2D                                                                             */
    double  robot_radius;    /* barrier radius (and the rotor arm of Quad2D, quad2D.py:72)                         */
    double  beta;            /* barrier inflation: 1.01                                                            */
    double  tol, acceptable_tol, mu_init, mu_min;
    sc_resto_params resto;   /* feasibility restoration, as sc_mpccbf_params                                       */
} sc_mpcgn_params;

int sc_mpcgn_solve_batch(const sc_mpcgn_params* params, int64_t B, int32_t K,
                         const void* X, const void* u_prev, const void* goal, const void* obs,
                         void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);
size_t sc_mpcgn_slices_workspace_bytes(const sc_mpcgn_params* params, int64_t B, int32_t K);
int sc_mpcgn_solve_batch_sliced(const sc_mpcgn_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K,
                                const void* X, const void* u_prev, const void* goal, const void* obs,
                                void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);
int sc_mpcgn_solve_batch_host(const sc_mpcgn_params* params, int64_t B, int32_t K,
                              const void* X, const void* u_prev, const void* goal, const void* obs,
                              void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, int device);

/* Optimal-decay MPC-CBF for the two models of optimal_decay_mpc_cbf.py:19 whose DT barrier steps the state with the
 * robot's own step(): KinematicBicycle2D and Quad2D.  Two decay variables per stage (the reference's omega1 / omega2 inputs, :123-124)
 * scale the DT-CBF gains of that stage's rows:  dd_h + (a1 rho1 + a2 rho2) d_h + a1 a2 rho1 rho2 h >= 0  (:291-297); cost + p_sb (rho -
 * omega_ref)^2 per stage and variable (:181-186), input term R u^2 (:178-179, not do-mpc's delta-u penalty); weights KB Q =
 * diag(50,50,1,1), R = (0.5, 50), gains 0.05; Quad2D as MPCCBF (:37-42,66-74).  The reference copy is stale (five 5-wide obstacle
 * slots) and its solver stack absent: oracle-only parity (oracle/od_mpc_gn.py), obstacle rows as in sc_mpcgn_solve_batch.
 * `mpc` is the sc_mpcgn_params of the model (alpha1 / alpha2 = the optimal-decay gains); no restoration phase (mpc.resto unused).
 * rho_out [B, 2 * horizon] or NULL: the decay variables (rho1_0, rho2_0, rho1_1, ...).                                          */
typedef struct sc_odmpcgn_params {
    sc_mpcgn_params mpc;
    double omega_ref[2];     /* cbf_param['omega1'], ['omega2'] = 1.0  (optimal_decay_mpc_cbf.py:88,90) */
    double p_sb[2];          /* cbf_param['p_sb1'], ['p_sb2'] = 10     (:89,91)                          */
} sc_odmpcgn_params;

int sc_odmpcgn_solve_batch(const sc_odmpcgn_params* params, int64_t B, int32_t K,
                           const void* X, const void* u_prev, const void* goal, const void* obs,
                           void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);

/* ---- MPC-CBF for VTOL2D ----------------------------------------------
 * MPCCBF (position_control/mpc_cbf.py:40-43,83-87,135-141,222-233,316-321) for the tilt-rotor of robots/vtol2D.py:118-311: 6 states
 * (x, z, theta, x_dot, z_dot, theta_dot), 4 inputs (front / rear / pusher throttle, elevator), prediction x + (f + g u) dt, horizon 30,
 * Q = diag(10, 10, 250, 10, 10, 50), delta-u weights R = (0.5, 0.5, 0.5, 50000), rel-degree-2 DT-CBF rows through step o step against K
 * discs (vtol2D.py:475-497, beta = 1.01, alpha1 = alpha2 = 0.05), bounds |x_dot| <= v_max, z_dot >= -descent_speed_max,
 * |theta| <= pitch_max, throttles in [0, 1], |elevator| <= 0.5.  Same interior point and statuses as sc_mpccbf_solve_batch
 * (restoration included), with the exact Hessian of the aero model and the slack reset of the line search that this model needs
 * (oracle/mpc_vtol.py: params).  The Newton system is solved stage by stage (Riccati recursion over 14 x 14 blocks): one NLP per
 * wavefront, one stage per lane, everything in registers and LDS (no workspace: sc_mpcvtol_workspace_bytes() returns 0 and
 * `workspace` may be NULL).  kernel = 1 selects the one-NLP-per-lane kernel the wave kernel was checked against; its work arrays
 * live in `workspace` (device memory, no initialisation needed).  K <= 16, horizon <= 64.
 * X [B,6], u_prev [B,4], goal [B,2], obs [B,K,7] (or [K,7] with obs_shared; columns 0..2 used: x, z, radius), u_out [B,4],
 * status_out [B], iters_out [B] or NULL, z_out [B, 4*horizon] or NULL.  f64 arithmetic; io_dtype f32 or f64.                       */
typedef struct sc_mpcvtol_params {
    int32_t io_dtype, horizon, max_iter, obs_shared, acceptable_iter;
    int32_t slack_reset;     /* 0: off, 1: s = max(s, g) after a trial step, 2: s = g where g >= mu / nu (oracle default for VTOL2D) */
    int32_t kernel;          /* 0 / 2: one NLP per wavefront (one stage per lane); 1: one NLP per lane (needs the workspace)         */
    int32_t reserved;
    double  dt;
    double  Q[6], R[4];
    double  alpha1, alpha2;
    double  u_lo[4], u_hi[4];
    double  v_max, descent_speed_max, pitch_max;      /* pitch_max in rad (mpc_cbf.py:232: degrees * 3.14159 / 180) */
    double  robot_radius, beta;
    double  tol, acceptable_tol, mu_init, mu_min;
    double  airframe[21];    /* vtol2D.py:56-111: mass, inertia, S_wing, rho, C_L0, C_Lalpha, M, alpha_0, C_Ldelta_e, C_D0, C_Dalpha,
                                C_Ddelta_e, C_m0, C_malpha, C_mdelta_e, chord, k_front, k_rear, k_pusher, ell_f, ell_r            */
    sc_resto_params resto;
} sc_mpcvtol_params;

size_t sc_mpcvtol_workspace_bytes(const sc_mpcvtol_params* params, int64_t B, int32_t K);
int sc_mpcvtol_solve_batch(const sc_mpcvtol_params* params, int64_t B, int32_t K,
                           const void* X, const void* u_prev, const void* goal, const void* obs,
                           void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out,
                           void* workspace, size_t workspace_bytes, void* stream);
/* continuation launches (sc_mpc_slices): the wave-per-problem kernel only (kernel = 0 / 2) */
size_t sc_mpcvtol_slices_workspace_bytes(const sc_mpcvtol_params* params, int64_t B, int32_t K);
int sc_mpcvtol_solve_batch_sliced(const sc_mpcvtol_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K,
                                  const void* X, const void* u_prev, const void* goal, const void* obs,
                                  void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);
int sc_mpcvtol_solve_batch_host(const sc_mpcvtol_params* params, int64_t B, int32_t K,
                                const void* X, const void* u_prev, const void* goal, const void* obs,
                                void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, int device);

/* MPC-CBF for VTOL2D AS DO-MPC POSES IT -- multiple shooting, IPOPT's algorithm (csrc/mpc_vtol_ms.hip, DESIGN.md kernel 12; round 5).
 * position_control/mpc_cbf.py:162-174 hands IPOPT the states x_0 .. x_N as variables with the dynamics as equality rows and
 * set_initial_guess (:366-369) starts every stage at x0; sc_mpcvtol_solve_batch solves the condensed single-shooting problem instead and
 * loses the reference's own example flight to one diverging rollout.  This entry point solves the multiple-shooting NLP with the filter
 * line-search interior point of Waechter & Biegler (2006) at IPOPT's documented option defaults (sc_ipopt_params; oracle/ms_ipopt.py is the
 * float64 statement, iterate for iterate), one NLP per wavefront, one stage per lane, Riccati recursion with defects.
 *   params      the model / problem fields of sc_mpcvtol_params (horizon, dt, Q, R, alpha1/2, bounds, radius, beta, airframe, io_dtype,
 *               obs_shared); its solver fields (tol .. resto, slack_reset, kernel, max_iter) are NOT read -- sc_ipopt_params has them
 *   status_out  SC_STATUS_OPTIMAL (tol or IPOPT's acceptable rule), SC_STATUS_INACCURATE (iteration limit), SC_STATUS_NEEDS_RESTO: the
 *               line search ended below alpha_min (or the inertia correction ran out) -- IPOPT would enter its restoration phase here;
 *               the caller re-solves these problems with sc_mpcvtol_solve_batch and its restoration (BatchedVtolMPCCBF does)
 *   plan_out    [B, (horizon + 1) * 6 + horizon * 4] or NULL: x_0 .. x_N, then u_0 .. u_{N-1}
 *   trace_out   [B, max_iter + 1, 8] float64 or NULL: per iteration E_0, dual / primal infeasibility, complementarity, mu, theta,
 *               delta_w, alpha (a debugging aid: the columns of oracle/ms_ipopt.py's trace)                                                 */
#define SC_STATUS_NEEDS_RESTO 4
typedef struct sc_ipopt_params {
    int32_t max_iter, acceptable_iter;
    double  tol, dual_inf_tol, constr_viol_tol, compl_inf_tol;
    double  acceptable_tol, acceptable_dual_inf_tol, acceptable_constr_viol_tol, acceptable_compl_inf_tol;
    double  nlp_scaling_max_gradient, nlp_scaling_min_value, bound_relax_factor, bound_push, bound_frac, constr_mult_init_max;
    double  mu_init, mu_linear_decrease_factor, mu_superlinear_decrease_power, barrier_tol_factor, tau_min;
    double  kappa_sigma, kappa_d, s_max;
    double  theta_max_fact, theta_min_fact, eta_phi, delta, s_phi, s_theta, gamma_phi, gamma_theta, alpha_min_frac, alpha_red_factor, obj_max_inc;
    double  first_hessian_perturbation, min_hessian_perturbation, max_hessian_perturbation, perturb_inc_fact_first, perturb_inc_fact,
            perturb_dec_fact;
    /* restoration phase (section 3.3 of the paper; IPOPT's option names and defaults: 1000, 1, 0.9, 1e3, 1e-6, 1e8).  It runs inside the kernel when
     * resto_workspace points at sc_mpcvtol_ms_workspace_bytes(B, K) bytes of device memory (row state of the elastic problem, one slab per NLP);
     * with resto_workspace = NULL a solve that needs it ends with SC_STATUS_NEEDS_RESTO.  The elastic variables sit on the inequality rows only
     * (the dynamics rows stay hard): oracle/ms_ipopt.py, resto_elastic = "ineq".                                                          */
    double  resto_penalty_parameter, resto_proximity_weight, required_infeasibility_reduction, bound_mult_reset_threshold,
            resto_failure_feasibility_threshold, resto_theta_max_fact;
    void*   resto_workspace;
    int64_t resto_workspace_bytes;
    /* NOT an IPOPT option (IPOPT's watchdog and tiny-step heuristics are not restated): a solve whose accepted step length stays below
     * stall_alpha for stall_iter consecutive iterations of one phase (regular or restoration) ends SC_STATUS_INACCURATE there instead of crawling to max_iter (the fraction-to-
     * the-boundary rule pins every step against a bound: measured on 2 of 4096 optimal-decay bench problems, after their restoration).
     * stall_iter = 0 disables the rule.  Oracle: oracle/ms_ipopt.py, options stall_iter / stall_alpha.                                    */
    double  stall_alpha;
    int32_t stall_iter;
    /* NOT an IPOPT option either (round 6; the slot was `reserved`: no layout change): a regular-phase solve that has sat floor_iter consecutive
     * iterations at the smallest barrier parameter with everything but the dual infeasibility inside the 'acceptable' tolerances ends
     * SC_STATUS_INACCURATE there.  Found with the 256-aircraft flight: six of ~80 k VTOL2D solves reach their optimum (the float64 oracle calls
     * them optimal after ~95 iterations with the same input to 3e-8) and then sit on a precision floor of the kernel's multiplier recovery
     * (multipliers of 1e8, a dual infeasibility that stays ~1) until max_iter: 0.7 s each in a closed loop.  30 in the Python classes; 0 disables
     * the rule.  Oracle: oracle/ms_ipopt.py, option floor_iter.                                                                             */
    int32_t floor_iter;
} sc_ipopt_params;

size_t sc_mpcvtol_ms_workspace_bytes(int64_t B, int32_t K);

int sc_mpcvtol_ms_solve_batch(const sc_mpcvtol_params* params, const sc_ipopt_params* ipopt, int64_t B, int32_t K,
                              const void* X, const void* u_prev, const void* goal, const void* obs,
                              void* u_out, int32_t* status_out, int32_t* iters_out, void* plan_out, double* trace_out, void* stream);

/* Optimal-decay MPC-CBF for VTOL2D -- the last model of the reference's accept list (optimal_decay_mpc_cbf.py:19): the NLP of
 * sc_mpcvtol_solve_batch with two decay variables per stage scaling the DT-CBF gains of that stage's rows,
 *   dd_h + (a1 rho1 + a2 rho2) d_h + a1 a2 rho1 rho2 h >= 0   (:288-296; gains 0.35: :83-86),
 * cost + p_sb (rho - omega_ref)^2 per stage and variable (:175-176), input term R u^2 (:173-174; not the delta-u penalty).  Same kernel
 * (one NLP per wavefront, one stage per lane, Riccati recursion); the 2 x 2 decay block of a stage is eliminated from the stage block
 * before the recursion.  No restoration phase (mpc.resto unused), one launch (no continuation entry).  The reference copy is stale and its
 * solver absent: oracle-only parity (oracle/od_mpc_vtol.py).  rho_out [B, 2 * horizon] or NULL.                                        */
typedef struct sc_odmpcvtol_params {
    sc_mpcvtol_params mpc;   /* alpha1 / alpha2 = the optimal-decay gains (0.35); kernel must be 0 or 2 */
    double omega_ref[2];     /* cbf_param['omega1'], ['omega2'] = 1.0 */
    double p_sb[2];          /* cbf_param['p_sb1'], ['p_sb2'] = 10    */
} sc_odmpcvtol_params;

int sc_odmpcvtol_solve_batch(const sc_odmpcvtol_params* params, int64_t B, int32_t K,
                             const void* X, const void* u_prev, const void* goal, const void* obs,
                             void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);

/* the same NLP in the multiple-shooting form under IPOPT's algorithm (kernel 12 with the decay rates as two more inputs of a stage, which is
 * what they are in the reference: optimal_decay_mpc_cbf.py:123-124); params->mpc as for sc_mpcvtol_ms_solve_batch, status / plan / trace too */
int sc_odmpcvtol_ms_solve_batch(const sc_odmpcvtol_params* params, const sc_ipopt_params* ipopt, int64_t B, int32_t K,
                                const void* X, const void* u_prev, const void* goal, const void* obs,
                                void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* plan_out, double* trace_out, void* stream);
size_t sc_odmpcvtol_slices_workspace_bytes(const sc_odmpcvtol_params* params, int64_t B, int32_t K);
int sc_odmpcvtol_solve_batch_sliced(const sc_odmpcvtol_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K,
                                    const void* X, const void* u_prev, const void* goal, const void* obs,
                                    void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);

/* MPC-CBF for DynamicUnicycle2D, Unicycle2D, SingleIntegrator2D, DoubleIntegrator2D and KinematicBicycle2D AS DO-MPC POSES IT (csrc/mpc_du_ms.hip, DESIGN.md
 * kernel 13; round 6): BASELINE configs[2] in the reference's own formulation.  Replaces MPCCBF.solve_control_problem (position_control/mpc_cbf.py:366-402: mpc.x0 = x; set_initial_guess(); update_tvp;
 * make_step -> do-mpc multiple shooting -> IPOPT, :162-174) for a batch: states x_0 .. x_N as variables, dynamics as equality rows, every
 * stage started at x0, IPOPT's filter line-search interior point at its documented defaults (sc_ipopt_params; oracle/ms_ipopt.py with
 * du_model() is the float64 statement, iterate for iterate), restoration phase inside the kernel (its state lives in LDS;
 * sc_ipopt_params.resto_workspace is the optional launch-order workspace here: sc_mpccbf_ms_workspace_bytes).  sc_mpccbf_solve_batch solves the condensed single-shooting form of the same NLP with an
 * l1-merit interior point: same optimum where there is one (4086 of 4096 config-3 draws), a different last iterate where the NLP has no
 * feasible point -- and the reference APPLIES that iterate (mpc_cbf.py:384, status hard-wired 'optimal', :10).
 *   params      the problem fields of sc_mpccbf_params (horizon 1 .. 62, dt, Q, R, alpha1/2, v_max, u_max, robot_radius, beta, io_dtype,
 *               obs_shared); its solver fields (tol .. resto, slack_reset, max_iter) are NOT read.  model_id:
 *                 SC_MODEL_DYNAMIC_UNICYCLE2D   x = (px, py, theta, v), u = (a, omega), |v_k| <= v_max, two-step rows (alpha1, alpha2)
 *                 SC_MODEL_UNICYCLE2D           X rows [px, py, theta, unused], u = (v, omega), u_max = (v_max, w_max), ONE-step rows
 *                                               h(p_k+1) - (1 - alpha1) h(p_k) >= 0 (mpc_cbf.py:312-315), Q[3] / alpha2 / v_max unused
 *                 SC_MODEL_SINGLE_INTEGRATOR2D  X rows [px, py, unused, unused] (four columns), u = (vx, vy), u_max = (v_max, v_max), one-step rows
 *                 SC_MODEL_DOUBLE_INTEGRATOR2D  x = (px, py, vx, vy), u = (ax, ay), no state bound; v_max = the norm robot.step rescales the
 *                                               velocity to inside the barrier (double_integrator2D.py:79-107,225-226)
 *                 SC_MODEL_KINEMATIC_BICYCLE2D  x = (px, py, theta, v), u = (a, beta), |v_k| <= v_max; robot.step clips the speed to
 *                                               [v_min, v_max] inside the barrier (kinematic_bicycle2D.py:112-123,175-199); rear_ax_dist.
 *                                               Where a plan slows down to v_min the clip's kink sits on the solution and the iteration
 *                                               cycles to max_iter (kernel 13): bound it
 *               ; obstacle rows: circles (flag column < 0.5) unless params->superellipsoid_rows = 1 (DynamicUnicycle2D and
 *               DoubleIntegrator2D; the host-side classes set it from the rows' flags), 1 <= K <= 16
 *   status_out  SC_STATUS_OPTIMAL (tol or IPOPT's acceptable rule), SC_STATUS_INFEASIBLE (the restoration phase converged to a stationary
 *               point of the violation: IPOPT's "converged to a point of local infeasibility"), SC_STATUS_INACCURATE (iteration limit,
 *               restoration failed, stall rule); u_out is the last iterate's u_0 in every case, as in the reference
 *   plan_out    [B, (horizon + 1) * 4 + horizon * 2] or NULL: x_0 .. x_N, then u_0 .. u_{N-1}
 *   trace_out   [B, max_iter + 1, 8] float64 or NULL: per iteration E_0, dual / primal infeasibility, complementarity, mu, theta,
 *               delta_w, alpha (negative inside the restoration phase)                                                                   */
int sc_mpccbf_ms_solve_batch(const sc_mpccbf_params* params, const sc_ipopt_params* ipopt, int64_t B, int32_t K,
                             const void* X, const void* u_prev, const void* goal, const void* obs,
                             void* u_out, int32_t* status_out, int32_t* iters_out, void* plan_out, double* trace_out, void* stream);
size_t sc_mpccbf_ms_lds_bytes(int32_t horizon, int32_t K);      /* LDS per NLP (one wavefront): 160 KiB / this = NLPs resident per CU */
/* Optional launch-order workspace (device memory, no initialisation needed): passed as ipopt->resto_workspace / resto_workspace_bytes -- the
 * fields the VTOL2D entry uses for its restoration state, which this kernel keeps in LDS.  With it, a pre-pass sends the problems whose start
 * point violates a CBF row (every NLP without a feasible point is one: the long solves) to the front of the grid, and a launch of more than 1024
 * problems ends with its longest solve instead of ~15 % later (configs[2]: 5.0 -> 4.3 ms).  Results do not depend on it.                       */
size_t sc_mpccbf_ms_workspace_bytes(int64_t B);

/* ---- optimal-decay MPC-CBF ---------------------------------------
 * OptimalDecayMPCCBF (position_control/optimal_decay_mpc_cbf.py:15-330) for DynamicUnicycle2D: the MPC-CBF NLP with
 * two decay variables per stage (omega1_k, omega2_k, model inputs at :123-124) that scale the DT-CBF gains,
 *   dd_h + (alpha1 omega1 + alpha2 omega2) d_h + alpha1 alpha2 omega1 omega2 h >= 0                  (:291-297)
 * and are pulled to their references by p_sb (omega - omega_ref)^2 (:181-184); the input term is R u^2 (:178-179,
 * an expression r-term, not the delta-u penalty of MPCCBF).  `mpc.alpha1/alpha2` carry the optimal-decay gains
 * (0.01 for DU, :59-60); `mpc.R` the weights of u^2.  The reference copy is stale (5-wide obstacle rows, 5 fixed
 * slots): here obstacles are the K 7-wide rows of sc_mpccbf_solve_batch.  No runnable reference exists; parity is
 * against oracle/od_mpc_cbf.py.
 * A reading the build cannot settle: the reference calls mpc.set_rterm(R u^2) and then mpc.set_rterm(p_sb terms) (:177-178).  If do-mpc's
 * set_rterm(expression) ASSIGNS the r-term (do-mpc is not installed here; its documentation suggests it does), the second call replaces the
 * first and the reference's objective has no R u^2 at all.  The kernels implement the sum of both, which is what the code says it means;
 * `mpc.R = {0, 0}` gives the other reading with no other change (the Python classes pass their public attribute R: set ctl.R to zeros).
 * Outputs as sc_mpccbf_solve_batch plus rho_out [B, 2*horizon] or NULL: (omega1_k, omega2_k) for every stage.
 */
typedef struct sc_odmpccbf_params {
    sc_mpccbf_params mpc;
    double omega_ref[2];     /* cbf_param['omega1'], ['omega2'] = 1 (:88,90)   */
    double p_sb[2];          /* cbf_param['p_sb1'], ['p_sb2'] = 10 (:89,91)    */
} sc_odmpccbf_params;

int sc_odmpccbf_solve_batch(const sc_odmpccbf_params* params, int64_t B, int32_t K,
                            const void* X, const void* u_prev, const void* goal, const void* obs,
                            void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out,
                            void* stream);

/* continuation launches (sc_mpc_slices) of the optimal-decay families -- round 5; the decay variables travel with the solver state:
 * resumed == uninterrupted bit for bit (tests/test_mpc_slices_gpu.py).  sc_odmpclin: sc_mpclin_slices_workspace_bytes sizes the workspace. */
size_t sc_odmpccbf_slices_workspace_bytes(const sc_odmpccbf_params* params, int64_t B, int32_t K);
int sc_odmpccbf_solve_batch_sliced(const sc_odmpccbf_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K,
                                   const void* X, const void* u_prev, const void* goal, const void* obs,
                                   void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);
size_t sc_odmpcgn_slices_workspace_bytes(const sc_odmpcgn_params* params, int64_t B, int32_t K);
int sc_odmpcgn_solve_batch_sliced(const sc_odmpcgn_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K,
                                  const void* X, const void* u_prev, const void* goal, const void* obs,
                                  void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);
int sc_odmpclin_solve_batch_sliced(const sc_mpclin_params* params, const sc_mpc_slices* slices, const double* model, int64_t B, int32_t K,
                                   const void* X, const void* u_prev, const void* goal, const void* obs,
                                   void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream);
int sc_odmpccbf_solve_batch_host(const sc_odmpccbf_params* params, int64_t B, int32_t K,
                                 const void* X, const void* u_prev, const void* goal, const void* obs,
                                 void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out,
                                 int device);

/* ---- optimal-decay CBF-QP --------------------------------------
 * OptimalDecayCBFQP (position_control/optimal_decay_cbf_qp.py:13-158): decision variables u (2) and the
 * decay multipliers omega1, omega2 with penalties p_sb (omega - omega_ref)^2 (:72-76); ONE obstacle row
 *   rel-deg 2 (DU, KB, Quad2D):  A u + b + (alpha1+alpha2) omega1 h_dot + alpha1 alpha2 h omega2 >= 0   (:83-90,105-115;
 *              Quad2D: X is [B,6], qp.state_dim = 6, box f_min <= u <= f_max in qp.u_min/u_max, qp.mass)
 *   rel-deg 1 (C3BF/DPCBF): A u + b + alpha h omega1 >= 0, no omega2 term                        (:99-104)
 * plus the input box.  `qp.alpha1/alpha2` carry the optimal-decay gains (0.5, :19-20).  The reference
 * copy is stale (it is handed a (k,7) array): here the obstacle is one 7-wide row per
 * agent (the nearest), has_obs[i] == 0 reproduces the `nearest_obs is None` branch (:133-137).
 */
typedef struct sc_odcbfqp_params {
    sc_cbfqp_params qp;      /* model, dtypes, radius, alpha1/alpha2 (or alpha), input box, rear_ax_dist */
    double omega_ref[2];     /* cbf_param['omega1'], ['omega2'] = 1.0                                 */
    double p_sb[2];          /* cbf_param['p_sb1'], ['p_sb2'] = 1e4                                   */
} sc_odcbfqp_params;

/* X [B,4] ([B,6] for Quad2D), u_ref [B,2], obs [B,7], has_obs [B] int32 or NULL (all present);
 * u_out [B,2] (NaN if not optimal), omega_out [B,2] (omega2 = omega_ref[1] for rel-deg-1 models),
 * status_out [B], h_out [B] or NULL. */
int sc_odcbfqp_solve_batch(const sc_odcbfqp_params* params, int64_t B,
                           const void* X, const void* u_ref, const void* obs, const int32_t* has_obs,
                           void* u_out, void* omega_out, int32_t* status_out, void* h_out, void* stream);

int sc_odcbfqp_solve_batch_host(const sc_odcbfqp_params* params, int64_t B,
                                const void* X, const void* u_ref, const void* obs, const int32_t* has_obs,
                                void* u_out, void* omega_out, int32_t* status_out, void* h_out, int device);

/* ---- Manipulator2D CBF-QP -------------------------------------
 * The Manipulator2D branches of CBFQP (position_control/cbf_qp.py:34-35 alpha = 1.0, :94-104 three inputs with
 * |u_i| <= w_max, :130-151 row loop) over robots/manipulator2D.py: a planar 3-joint arm, state X = [theta1..3], input
 * U = joint velocities, f = 0, g = I.  get_link_circles (:129-152) discretises link i into link_steps[i] + 1 circles
 * (the reference evaluates int(np.ceil(link_len / (10/60))) in float64: 8, 8, 6 for its 80/70/50-pixel links -- the
 * host passes the counts so the device never re-derives that rounding); agent_barrier (:185-224) returns one
 * (h, dh/dq) per circle: h = |c - o|^2 - beta (R + r)^2, dh/dq = 2 (c - o)^T J_c with beta = 1.3.  Rows are written
 * obstacle by obstacle, circle by circle, until `num_rows` (= CBFQP's num_obs; 150 by default, tracking.py:134-138)
 * are used: A = dh/dq, b = alpha h ('cbf') or h / dt ('hard').
 * X [B,3], u_ref [B,3], obs [B,K,7] (or [K,7]) circles [x, y, r, ...], n_obs [B] or NULL;
 * u_out [B,3] (NaN where not optimal), status_out [B], h_out [B,num_rows] or NULL (0 for unused rows).
 * One QP (3 variables, up to 250 + 6 rows) per wavefront, dual active set; arithmetic is f64.
 */
#define SC_MANIP_MAX_ROWS 250

typedef struct sc_manip_cbfqp_params {
    int32_t io_dtype;        /* SC_DTYPE_*: element type of X,u_ref,obs,u_out,h_out                     */
    int32_t cbf_mode;        /* SC_CBF_MODE_*                                                           */
    int32_t obs_shared;      /* 0: obs is [B,K,7]; 1: one [K,7] table for all agents                    */
    int32_t num_rows;        /* CBFQP(num_obs=...): row cap, 1..SC_MANIP_MAX_ROWS                       */
    int32_t link_steps[3];   /* circles per link minus one (manipulator2D.py:143), each >= 1            */
    int32_t reserved;
    double  robot_radius;    /* robot.robot_radius: radius of every link circle                         */
    double  dt;              /* robot.dt ('hard' mode)                                                  */
    double  alpha;           /* cbf_param['alpha'] = 1.0 (cbf_qp.py:34-35)                              */
    double  w_max;           /* robot_spec['w_max'] = 2.0 (manipulator2D.py:21)                         */
    double  beta;            /* barrier inflation, 1.3 (manipulator2D.py:185)                           */
    double  link_lengths[3]; /* manipulator2D.py:18                                                     */
    double  base_pos[2];     /* manipulator2D.py:24 (examples/test_tracking.py:163-165 moves it)        */
} sc_manip_cbfqp_params;

int sc_manip_cbfqp_solve_batch(const sc_manip_cbfqp_params* params, int64_t B, int32_t K,
                               const void* X, const void* u_ref, const void* obs, const int32_t* n_obs,
                               void* u_out, int32_t* status_out, void* h_out, void* stream);

int sc_manip_cbfqp_solve_batch_host(const sc_manip_cbfqp_params* params, int64_t B, int32_t K,
                                    const void* X, const void* u_ref, const void* obs, const int32_t* n_obs,
                                    void* u_out, int32_t* status_out, void* h_out, int device);

/* ---- neighbour agents as moving obstacles (extension) -----------------
 * The reference's robots never see each other (examples/test_multi_robot.py:77-80); BASELINE
 * config 4 asks for it.  After the per-step all-gather of agent states (RCCL over xGMI when the
 * batch is sharded), every local agent takes its K nearest OTHER agents as circular moving
 * obstacles [x, y, r, vx, vy, 0, 0] with r = neighbour_radius, (vx, vy) = v (cos theta, sin theta),
 * ordered by centre distance like tracking.py:393-403.  Rows beyond the number of other agents are
 * far-away dummies [1e3, 1e3, 0, ...] (mpc_cbf.py:343).  X_all [B_all,4]; the local agents are
 * X_all[first_local : first_local + B_local]; obs_out [B_local, K, 7], K <= 32.
 */
int sc_neighbor_obstacles_batch(int32_t io_dtype, int64_t B_all, int64_t first_local, int64_t B_local,
                                int32_t K, double neighbour_radius, const void* X_all, void* obs_out,
                                void* stream);

/* Same result, bit for bit, with the candidate range cut into slices (one wave per 64 agents x slice, then a merge):
 * the single-scan form above takes the same time whatever the shard size, this one shortens with the shard, which is
 * what sharding the fleet over several GPUs needs.  `workspace` is device memory of at least
 * sc_neighbor_workspace_bytes(io_dtype, B_all, B_local, K) bytes (0 on invalid arguments).
 */
size_t sc_neighbor_workspace_bytes(int32_t io_dtype, int64_t B_all, int64_t B_local, int32_t K);
int sc_neighbor_obstacles_batch_ws(int32_t io_dtype, int64_t B_all, int64_t first_local, int64_t B_local,
                                   int32_t K, double neighbour_radius, const void* X_all, void* obs_out,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- closed-loop control_step, fused -----------------------
 * Runs `n_steps` iterations of LocalTrackingController.control_step (tracking.py:559-668; moving
 * obstacles: dynamic_env/main.py:126-236) for B agents in ONE launch, with the CBF-QP solve behind
 * the boundary: goal / state machine (tracking.py:497-535, :569-577), nearest-unpassed obstacle
 * selection (:345-403), nominal input choice (:589-604; robots/<model>.py nominal_input / stop /
 * rotate_to), CBFQP.solve_control_problem, collision checks (:445-495, :627-646), robot.step (:637)
 * and the return code (:666-668).  Agent state stays in registers between steps.
 * Models: DynamicUnicycle2D, Unicycle2D, the KinematicBicycle2D family, and SingleIntegrator2D / DoubleIntegrator2D with
 * enable_rotation = 0 (their rotate state runs the attitude controllers, which are outside this library; without them the
 * heading of an integrator never changes and only decides the first state-machine state, which the caller sets).
 */
#define SC_SM_IDLE   0
#define SC_SM_TRACK  1
#define SC_SM_STOP   2
#define SC_SM_ROTATE 3

#define SC_TRACKING_MAX_CONSTRAINTS 16   /* num_constraints (reference default 10, tracking.py:134-138) */

typedef struct sc_tracking_params {
    sc_cbfqp_params qp;          /* the position controller (io_dtype applies to every float array) */
    int32_t n_steps;             /* control steps in this launch                                    */
    int32_t max_waypoints;       /* W: row count of `waypoints` per agent                           */
    int32_t waypoints_shared;    /* 1: waypoints is [W,2] and n_wp is [1]; 0: [B,W,2] and [B]       */
    int32_t enable_rotation;     /* LocalTrackingController(enable_rotation=...)                    */
    int32_t dyn_obs;             /* 1: obstacle table moves obs[:,0:2] += obs[:,3:5]*dt after the
                                    selection of each step (dynamic_env/main.py:54-58,147)          */
    int32_t num_constraints;     /* K: rows of the CBF-QP (tracking.py:138)                         */
    int32_t step_offset;         /* control steps already run before this launch: ret_step stores
                                    step_offset + the index inside the launch                       */
    int32_t reserved0;           /* keep 0                                                          */
    double  reached_threshold;   /* 0.3  (tracking.py:49)                                           */
    double  rotation_threshold;  /* 0.1  (tracking.py:46)                                           */
    double  v_max, v_min;        /* robot_spec: nominal-input saturation; KB step clips v to both   */
    double  k_omega, k_a, k_v;   /* nominal_input gains as forwarded by BaseRobot (2, 1, 1)         */
    double  delta_max;           /* KB family                                                       */
    double  wheel_base;          /* KB family                                                       */
} sc_tracking_params;

/* X [B,4] in/out; waypoints [B,W,2] (or [W,2]); n_wp [B] (or [1]); wp_index [B] in/out;
 * state_machine [B] in/out (SC_SM_*); goal [B,3] in/out = (gx, gy, valid);
 * obs_table [M,7] in/out (shared by all agents; when dyn_obs every block of the launch reads the table
 * as it was at launch and a stream-ordered follow-up kernel writes the table advanced by n_steps);
 * u_last [B,2] in/out: the last input an agent applied (kept across launches for frozen agents);
 * ret [B] in/out: 0 running, -1 all waypoints reached, -2 infeasible or collision (sticky: an agent
 * whose ret != 0 is frozen); ret_step [B] in/out: step_offset + step index inside the launch at which
 * ret turned non-zero (initialise to -1; untouched for agents that were already frozen); traj_X [n_steps,B,4], traj_U [n_steps,B,2] optional (NULL to skip):
 * state AFTER each step and the input applied (rows of frozen agents repeat their last state).
 */
int sc_tracking_rollout_batch(const sc_tracking_params* params, int64_t B, int32_t M,
                              void* X, const void* waypoints, const int32_t* n_wp,
                              int32_t* wp_index, int32_t* state_machine, void* goal,
                              void* obs_table, void* u_last, int32_t* ret, int32_t* ret_step,
                              void* traj_X, void* traj_U, void* stream);

/* control_step split around the solve, for position controllers that are their own launch (MPC-CBF, optimal-decay
 * MPC-CBF; the reference's default `--algo mpc_cbf`, examples/test_tracking.py:15):
 *   sc_tracking_select_batch = tracking.py:569-609: state machine / goal update, nearest-unpassed selection
 *     (num_constraints rows, missing rows padded [1000,1000,0,..] like MPCCBF.update_tvp, mpc_cbf.py:338-364) and the
 *     nominal input.  Outputs obs_out [B,K,7], goal_out [B,2] (the agent's own position when it has no goal),
 *     u_ref_out [B,2], track_out [B] = 1 where control_ref['state_machine'] == 'track' (the controllers pass u_ref
 *     through otherwise, mpc_cbf.py:379-381).  wp_index / state_machine / goal are updated in place.
 *   sc_tracking_apply_batch = tracking.py:627-668: collision checks, robot.step(u), return codes.  u [B,2] is the
 *     input to apply; u_status [B] (SC_STATUS_*) or NULL when the controller never reports failure (MPCCBF.status is
 *     hard-wired to 'optimal', mpc_cbf.py:10).  DynamicUnicycle2D and the KinematicBicycle2D family; static tables.
 */
int sc_tracking_select_batch(const sc_tracking_params* params, int64_t B, int32_t M,
                             const void* X, const void* waypoints, const int32_t* n_wp,
                             int32_t* wp_index, int32_t* state_machine, void* goal,
                             const void* obs_table, const int32_t* ret,
                             void* obs_out, void* goal_out, void* u_ref_out, int32_t* track_out, void* stream);

int sc_tracking_apply_batch(const sc_tracking_params* params, int64_t B, int32_t M, int32_t step_index,
                            void* X, const int32_t* state_machine, const void* goal, const void* obs_table,
                            const void* u, const int32_t* u_status, void* u_last,
                            int32_t* ret, int32_t* ret_step, void* stream);

/* Closed loop for the arm (SURVEY 8f-1 for Manipulator2D): `n_steps` iterations of LocalTrackingController.control_step
 * (tracking.py:559-668; goal_reached on the end effector :263-268; update_goal :497-535) with nominal_input / step of
 * robots/manipulator2D.py:38-41,110-127 and the CBF-QP above, one arm per wavefront, joint angles in registers.
 * obs_table [M,7]: the known obstacles ALREADY ORDERED by distance to the base (what get_nearest_unpassed_obs returns
 * for this model: robot.get_position() is the fixed base, robots/robot.py:354-356); the first min(M, num_rows) go to the
 * controller.  X [B,3] in/out; waypoints [B,W,2] (or [W,2]); n_wp, wp_index, state_machine, goal [B,3] = (gx, gy, valid),
 * ret, ret_step as in sc_tracking_rollout_batch; u_last [B,3]; traj_X / traj_U [n_steps,B,3] or NULL.
 */
typedef struct sc_manip_tracking_params {
    sc_manip_cbfqp_params qp;      /* num_rows = num_constraints (150 by default, tracking.py:134-138)          */
    int32_t n_steps, max_waypoints, waypoints_shared, enable_rotation;
    int32_t step_offset, reserved0; /* as in sc_tracking_params: ret_step = step_offset + index in the launch    */
    double  Kp;                    /* robot_spec['Kp'] (manipulator2D.py:22; examples use 5.0)                  */
    double  reached_threshold;     /* robot_spec['reached_threshold'] (0.3; examples/test_tracking.py:130: 0.5) */
    double  rotation_threshold;    /* 0.1 (tracking.py:46)                                                      */
} sc_manip_tracking_params;

int sc_manip_tracking_rollout_batch(const sc_manip_tracking_params* params, int64_t B, int32_t M,
                                    void* X, const void* waypoints, const int32_t* n_wp,
                                    int32_t* wp_index, int32_t* state_machine, void* goal,
                                    const void* obs_table, void* u_last, int32_t* ret, int32_t* ret_step,
                                    void* traj_X, void* traj_U, void* stream);

/* ---- Backup-CBF QP --------------------------------------------------------------------------------
 * BackupCBF.solve_control_problem (position_control/backup_cbf_qp.py:563-794) for B agents per launch on the scenario the
 * reference ships for it (examples/evade/test_evade.py --algo backupcbf): DoubleIntegrator2D
 * (robots/double_integrator2D.py:46-107), EvadeBackupController (position_control/backup_controller.py:420-572) and
 * EvadeEnv with its constant-speed rectangular bullet (envs/evade_env.py:30-83,360-406).  Per agent: rollout of the backup
 * controller over n_steps = int(backup_horizon / dt) states with robot.step, sensitivities by forward differences
 * (:236-320), one row per backup step from the finite-difference gradient of _h_safety (:343-461, :620-665) plus the
 * terminal row (_h_terminal :463-561, :668-676), the QP in scaled inputs (:678-716) solved EXACTLY (the reference calls
 * OSQP, :717-726), and the fallback rules (:737-774).  DriftingCar / LaneChangeController (examples/drift_car) are not served.
 * status_out: -1 no row survived |lhs| > 1e-6 (u = the reference input as given), 0 QP solved, 1 QP infeasible or
 * non-finite rows (u = clipped reference input when min h > 0.01, else the backup input).
 */
typedef struct sc_backupcbf_params {
    int32_t io_dtype;            /* SC_DTYPE_F32 / SC_DTYPE_F64: element type of X, u_nom, bullet_x, u_out, h_min_out    */
    int32_t n_steps;             /* int(backup_horizon / dt) (backup_cbf_qp.py:55), 2 <= n_steps <= 128                  */
    int32_t bullet_shared;       /* 1: bullet_x is [1] (one environment for every agent), 0: [B]                         */
    int32_t reserved;            /* keep 0                                                                               */
    double  dt, backup_horizon;  /* backup_horizon also dates the safety value inside _h_terminal (:537-541)             */
    double  fd_eps;              /* 1e-5 (:282, :451, :551)                                                              */
    double  robot_radius, a_max, v_max, safety_margin;   /* robot_spec (test_evade.py:75-88,298-299)                     */
    double  alpha, alpha_terminal;                       /* 1.0, 2.0 (:93-94)                                            */
    double  backup_kp, backup_kd;                        /* 2.0, 2.0 (backup_controller.py:449-450)                      */
    double  hallway_length, half_width;                  /* EvadeEnv geometry (envs/evade_env.py:51-79)                  */
    double  pocket_x_min, pocket_x_max, pocket_y_min, pocket_y_max;
    double  goal_x_min, goal_x_max;
    double  bullet_speed, bullet_length, bullet_width, bullet_start_x;
} sc_backupcbf_params;

/* X [B,4] = (x, y, vx, vy); u_nom [B,2] the nominal input (nominal_u_traj[0], :179-181) or NULL for the example's
 * EvadeNominalController (test_evade.py:128-166); bullet_x [B] or [1]: EvadeEnv.bullet_x (the box the barrier sees is
 * get_bullet_state()'s: centre + length / 6, length 4/3); u_out [B,2]; status_out [B] as above; using_backup_out [B]
 * (BackupCBF.is_using_backup()), h_min_out [B] (_last_h_min), n_rows_out [B] and rows_out [B, n_steps, 3] (float64, the
 * kept rows "(G S) us >= h" in the order the reference appends them, first n_rows_out[b] entries) may be NULL. */
int sc_backupcbf_solve_batch(const sc_backupcbf_params* params, int64_t B, const void* X, const void* u_nom,
                             const void* bullet_x, void* u_out, int32_t* status_out, int32_t* using_backup_out,
                             void* h_min_out, int32_t* n_rows_out, double* rows_out, void* stream);

/* The example's closed loop (test_evade.py:425-500), n_ctrl control steps in one launch: nominal controller -> Backup-CBF
 * QP -> robot.step -> speed clamp -> step_bullet (respawn past the hallway) -> collision / goal checks.  X [B,4] and
 * bullet_x ([B]; with bullet_shared the caller advances the shared bullet) in/out; ret [B] in/out: 0 running, 1 goal
 * reached, -2 collision with the bullet (sticky: frozen afterwards); ret_step [B] in/out: step_offset + index inside the
 * launch at which ret turned non-zero; u_out / status_out / using_backup_out / h_min_out: values of the last step. */
int sc_backupcbf_rollout_batch(const sc_backupcbf_params* params, int64_t B, int32_t n_ctrl, int32_t step_offset,
                               void* X, void* bullet_x, void* u_out, int32_t* status_out, int32_t* using_backup_out,
                               void* h_min_out, int32_t* ret, int32_t* ret_step, void* stream);

/* ---- control_step around the solve for Quad2D / Quad3D (SURVEY 8f-1 over the 8f-3 models) ----------------------------------
 * The split of sc_tracking_select_batch / sc_tracking_apply_batch for the two quadrotor models, whose states (6 / 12), inputs
 * (2 / 4) and goals (2-D / 3-D) do not fit the 4-state kernels: LocalTrackingController.control_step (tracking.py:559-668) with
 * update_goal (:497-535; Quad2D skips 'rotate', Quad3D waypoints are [x, y, z]), the K nearest obstacles
 * (get_nearest_unpassed_obs :345-403, angle_unpassed = 2 pi for both), nominal_input / stop / has_stopped / rotate_to / step
 * of robots/quad2D.py:83-164 and robots/quad3D.py:113-257.  Between the two calls the caller runs the position controller
 * (sc_mpcgn_solve_batch for Quad2D, sc_mpclin_solve_batch for Quad3D) and passes u_ref through for agents with track_out == 0
 * (mpc_cbf.py:379-381).
 */
#define SC_QUADTRACK_QUAD2D 0
#define SC_QUADTRACK_QUAD3D 1
#define SC_QUADTRACK_VTOL2D 2   /* X [B,6], u [B,4], goal_out [B,2]; 'rotate' skipped, reference input zero (vtol2D.py:459-465), obstacles
                                  inside the 1.2 pi cone about the pitch angle first (tracking.py:354-355), ground and pitch tests (:490-495) */

typedef struct sc_quadtrack_params {
    int32_t model;               /* SC_QUADTRACK_QUAD2D: X [B,6], u [B,2], goal_out [B,2];  _QUAD3D: X [B,12], u [B,4], goal_out [B,3] */
    int32_t io_dtype;            /* SC_DTYPE_F32 / SC_DTYPE_F64: element type of every float array                          */
    int32_t max_waypoints;       /* W: rows of `waypoints` per agent, [B,W,3] (or [W,3] when waypoints_shared)              */
    int32_t waypoints_shared;
    int32_t enable_rotation;     /* LocalTrackingController(enable_rotation=...)                                            */
    int32_t num_constraints;     /* K <= SC_TRACKING_MAX_CONSTRAINTS obstacle rows handed to the controller                 */
    int32_t reserved0, reserved1;
    double  dt, reached_threshold, rotation_threshold, robot_radius;   /* 0.05, 0.3, 0.1 (tracking.py:46-54), robot_spec     */
    double  mass;                /* Quad2D 1.0 (quad2D.py:41), Quad3D 3.0 (quad3D.py:50)                                    */
    double  inertia, f_min, f_max;                                   /* Quad2D (quad2D.py:42-44)                            */
    double  Ix, Iy, Iz, L, nu, u_min, u_max;                         /* Quad3D (quad3D.py:51-59)                            */
    double  airframe[21];        /* VTOL2D: as sc_mpcvtol_params.airframe (mass and inertia are read from here for this model)   */
    double  pitch_limit;         /* VTOL2D: what tracking.py:493 compares |theta| with -- robot_spec['pitch_max'] AS GIVEN (degrees: 15) */
} sc_quadtrack_params;

/* X [B,nx]; waypoints, n_wp, wp_index, state_machine, ret as in sc_tracking_select_batch; goal [B,4] in/out = (gx, gy, gz,
 * valid); obs_table [M,7]; obs_out [B,K,7] (missing rows padded [1000,1000,0,..]); goal_out [B,ng] (the own position when
 * the agent has no goal); u_ref_out [B,nu]; track_out [B]. */
int sc_quadtrack_select_batch(const sc_quadtrack_params* params, int64_t B, int32_t M,
                              const void* X, const void* waypoints, const int32_t* n_wp,
                              int32_t* wp_index, int32_t* state_machine, void* goal,
                              const void* obs_table, const int32_t* ret,
                              void* obs_out, void* goal_out, void* u_ref_out, int32_t* track_out, void* stream);

/* X [B,nx] in/out; u [B,nu] the input to apply; u_last [B,nu]; ret / ret_step as in sc_tracking_apply_batch. */
int sc_quadtrack_apply_batch(const sc_quadtrack_params* params, int64_t B, int32_t M, int32_t step_index,
                             void* X, const int32_t* state_machine, const void* goal, const void* obs_table,
                             const void* u, void* u_last, int32_t* ret, int32_t* ret_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAFE_CONTROL_AMD_H */
