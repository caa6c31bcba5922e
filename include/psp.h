/*
 * psp.h -- C ABI of the MI355X-native path-space hot path (libpsp_hip.so).
 *
 * The reference (lorenzrichter/path-space-PDE-solver) is pure Python and exposes no
 * FFI; its boundary is the duck-typed Solver / Problem / FunctionSpace API.  This
 * library sits underneath this repo's Python mirror of that API and replaces the
 * body of the training iteration for the supported catalogue:
 *   Solver.train (reference solver.py:420-557)
 *     psp_hjb_*   approx_method='control', time_approx='inner', one tanh MLP with two hidden layers
 *                 (function_space.py:177-195); losses log-variance / moment (in the kernels), variance /
 *                 cross_entropy (caller-supplied trajectory weights), relative_entropy; detach_forward True, or
 *                 False through psp_hjb_adjoint_sweep (gradients through the state path); importance-sampling
 *                 evaluation (psp_hjb_rollout_eval, utilities.py:287-359)
 *     psp_dnet_*  DenseNet controls (function_space.py:116-140): time_approx='outer' (one net per time step) and a
 *                 DenseNet(d+1 -> d) swapped into z_n; forward rollout AND hand-written parameter gradient
 *                 (psp_dnet_rollout_bwd; instances whose accumulators do not fit report bwd_supported = 0);
 *                 detach_forward False and relative_entropy through psp_dnet_adjoint_sweep
 *   GeneralSolver.train / EllipticSolver.train (solver.py:1001-1206, :628-826)
 *     psp_gen_*   diffusion / BSDE loss on unbounded, sphere and box domains, V = DenseNet(d+1 -> 1) / DenseNet(d -> 1)
 *   shared: psp_adam_step (per-net Adam, function_space.py:185), psp_allreduce + psp_comm_* (trajectory sharding over
 *   the GPUs of a node), diagnostics.
 *
 * Conventions
 *  - plain pointers and sizes only; every device buffer is owned by the caller
 *    (torch tensors on the Python side); nothing here allocates or frees.
 *  - all functions are asynchronous on the given hipStream_t (passed as void*),
 *    return 0 on success and a negative code on failure; psp_last_error() gives text.
 *  - everything is fp32 ("dtype f32"); partial sums of D are fp64.
 *  - parameters / gradients use the torch nn.Linear flat layout of the control net:
 *        [W1 (H x (d+1)), b1 (H), W2 (H x H), b2 (H), W3 (d x H), b3 (d)]
 *    where input column 0 of W1 is the time feature (reference solver.py:355).
 */
#ifndef PSP_H_
#define PSP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSP_VERSION 400 /* 0.3.0: range_flag in psp_hjb_config / psp_gen_config (guarded split-product mode); 0.3.1: store_path 4;
                         * 0.4.0: PSP_DOM_ANNULUS; psp_genl_config.activation / linear_layout, psp_genl_rollout_bwd replaces
                         *        psp_genl_adjoints (hand-written weight gradient), psp_genl_sizes re-laid out; the backward side of
                         *        the range guard; psp_hjb_sizes.fwd_coop_tiles (the former `reserved`); the on-device noise is
                         *        Philox4x32-7 (psp_philox_normal_fill and every rollout kernel: 10 rounds before) */

/* drift b(x): reference problems.py:36-37,154-155 (dense), :311-315 (double well) */
enum { PSP_DRIFT_ZERO = 0, PSP_DRIFT_DENSE = 1, PSP_DRIFT_DIAG = 2, PSP_DRIFT_DOUBLE_WELL = 3 };
/* sigma(x) = B constant: problems.py:39-40,157-158,317-318 */
enum { PSP_SIGMA_IDENTITY = 0, PSP_SIGMA_DENSE = 1, PSP_SIGMA_SCALED_IDENTITY = 2 };
/* running cost f(x) inside h = -0.5|z|^2 - f(x): problems.py:46 (zero), :161,167 (x'Px, diagonal P) */
enum { PSP_RUNCOST_ZERO = 0, PSP_RUNCOST_DIAG_QUAD = 1 };
/* terminal cost g(x): problems.py:49 (alpha.x), :164 (x'Rx, diagonal R), :334 (sum eta_j (x_j-1)^2) */
enum { PSP_TERM_LINEAR = 0, PSP_TERM_DIAG_QUAD = 1, PSP_TERM_SHIFTED_QUAD = 2 };
/* loss: solver.py:167-168 (log-variance), :165-166 (moment) */
/* PSP_LOSS_WEIGHTS: psp_hjb_rollout_bwd takes w_k = dLoss/dY_k itself in its D argument (losses whose
 * weights are not affine in D: variance :171-172, cross_entropy :183-186) */
/* PSP_LOSS_REL_ENTROPY (solver.py:179-180, 484-486): the forward kernel accumulates Y = -Zsum =
 * -sum_n (|Z_n|^2 / 2 + f(X_{n+1})) dt, so D = Y - g(X_N) = -(Zsum + g) and the loss is -mean D = -sums[0] / K. */
enum { PSP_LOSS_LOG_VARIANCE = 0, PSP_LOSS_MOMENT = 1, PSP_LOSS_WEIGHTS = 2, PSP_LOSS_REL_ENTROPY = 3 };
/* Brownian increments: supplied = the reference's host-generated xi (solver.py:381), philox = on device */
enum { PSP_NOISE_SUPPLIED = 0, PSP_NOISE_PHILOX = 1 };

/* arithmetic type of the matrix products (operands; accumulation is always fp32) */
enum { PSP_MLP_FP32 = 0, PSP_MLP_BF16_FWD = 1, PSP_MLP_BF16 = 2, PSP_MLP_F16X3 = 3 };

/* POD description of one HJB rollout problem on one rank. */
typedef struct psp_hjb_config {
    int32_t d;            /* state dimension                                    */
    int32_t H;            /* hidden width of the control MLP (two hidden layers) */
    int32_t K_local;      /* trajectories on this rank                           */
    int32_t N;            /* time steps, floor(T/dt) (solver.py:41)              */
    int64_t K_global;     /* trajectories over all ranks (loss normalisation)    */
    int64_t k_offset;     /* global index of local trajectory 0 (Philox counter) */
    float dt;             /* fp32 step (solver.py:39)                            */
    float sqrt_dt;        /* fp32 sqrt(dt) (solver.py:40)                        */
    int32_t drift_kind;
    int32_t sigma_kind;
    int32_t runcost_kind;
    int32_t term_kind;
    int32_t adaptive;     /* 1: c = -Z (solver.py:456), 0: c = 0 (solver.py:451) */
    int32_t loss_kind;
    int32_t noise_mode;
    int32_t store_path;   /* 0: forward only; 1: keep X_n, h1, h2, xi for the backward pass;
                           * 2 / 3: same with xi - sqrt(dt) Z / Z in the xi slot, for psp_hjb_adjoint_sweep;
                           * 4: keep X_n, h1, h2 only -- psp_hjb_rollout_bwd regenerates xi from the Philox counters
                           *    (seed, iter of the call = those of the forward call).  Narrow family, PSP_NOISE_PHILOX,
                           *    adaptive = 1 (the image of mode 1 is then xi itself); the block layout is that of mode 1
                           *    with the xi slot unused (same path_bytes).  Not for psp_hjb_rollout_bwd_step.        */
    float sigma_scale;    /* PSP_SIGMA_SCALED_IDENTITY                            */
    int32_t reserved;
    const float* drift;   /* DENSE: A (d*d row-major); DIAG: a (d); DOUBLE_WELL: kappa (d) */
    const float* sigma;   /* DENSE: B (d*d row-major); else NULL                  */
    const float* runcost; /* DIAG_QUAD: p (d); else NULL                          */
    const float* term;    /* alpha / r / eta (d)                                  */
    const float* u_ref;   /* optional (N, d) + 16 floats of slack: reference control u*(t_n) of a solution that does not depend on x
                           * (LLGC, problems.py:51-53); enables the u_L2 log of solver.py:491-494 inside the
                           * forward kernels.  NULL: no logging                                             */
    float* u_l2_out;      /* (K_local): sum_n |-Z_n(X_n) - u*(t_n)|^2 dt per trajectory (mean = u_L2_loss)  */
    int32_t mlp_dtype;    /* how the matrix products are computed (accumulation is fp32 in every mode):
                           * PSP_MLP_FP32 (0): v_mfma_f32_16x16x4_f32.
                           * PSP_MLP_F16X3: fp32-GRADE split products on the f16 matrix pipe -- every operand x = hi + lo / 2048 as
                           * two f16 numbers, a.b = hi.hi + (hi.lo + lo.hi) / 2048 as three v_mfma_f32_16x16x32_f16 (16x the
                           * fp32 matrix rate); same parity bounds as PSP_MLP_FP32 (D within 2e-5, gradient 2e-4, loss 1e-4 of the
                           * reference; observed 1e-6), operands must stay below 65504 in magnitude.  Narrow family: forward
                           * (hjb_fwd_kernel mode 2, tile-per-wave kernel only), adjoint sweep and backward (hjb_bwd3_kernel); wide
                           * family: forward, adjoint sweep and backward (d > 256: hjbw_bwd_x3_kernel; d <= 256: hjbw_bwd2x_kernel); DenseNet
                           * controls (psp_dnet_*): forward and adjoint sweep; -3 where an instance does not have the mode or its
                           * tables do not fit the LDS.  The weight-carrying operands of the backward passes (~1 / K) are scaled by
                           * a power of two inside the kernels and the results scaled back (exact).
                           * PSP_MLP_BF16_FWD: the three products of the control net in the FORWARD rollout on
                           * v_mfma_f32_16x16x32_bf16 (bf16 operands: its OWN tolerance, not the 1e-4 bar); drift / sigma
                           * products, state, sums and the backward pass stay fp32.  Narrow kernel family only (-3 otherwise) */
    int32_t reserved2;
    const uint32_t* iter_dev; /* optional DEVICE-resident iteration counter (the `iter` member of a psp_iter_state): when set, the
                           * forward kernels key Philox with *iter_dev instead of the `iter` argument, so that a captured
                           * hipGraph of the iteration can be replayed without per-iteration host arguments.  NULL: `iter` */
    int32_t* range_flag;  /* optional DEVICE int32[4] (8-byte aligned; [2..3] are scratch of the library), read only with mlp_dtype == PSP_MLP_F16X3: the RANGE GUARD of the split-product
                           * mode.  The reference computes in fp32 (solver.py:39-40); an f16x3 operand beyond 65504 has hi = +inf and
                           * lo = -inf, so main and correction chains of every product it enters meet as inf - inf: every output of
                           * that trajectory is NaN from that step on and D_k is NaN -- an overflow can never go unnoticed, and it
                           * costs nothing to detect.  With range_flag set,
                           *   psp_hjb_rollout_fwd / psp_dnet_rollout_fwd run the split kernel, set range_flag[0] = 1 iff a per-workgroup
                           *     partial of (sum D, sum D^2) is non-finite (else 0; range_flag[1] counts the 1s), and enqueue the
                           *     fp32-MFMA kernel of the same launch PREDICATED on range_flag[0] (its workgroups return at once when it
                           *     is 0; when it runs it overwrites D, the partials and the path store, whose format the two share);
                           *   psp_hjb_adjoint_sweep, psp_hjb_rollout_bwd(_step), psp_dnet_adjoint_sweep enqueue the split kernel
                           *     predicated on range_flag[0] == 0 and its fp32-MFMA twin predicated on range_flag[0] == 1.
                           * A guarded iteration therefore returns the fp32-MFMA result wherever the split kernels left their range
                           * (and the same non-finite loss as the reference when fp32 itself overflows), with no host sync.
                           * NULL: unguarded split kernels.                                                                       */
} psp_hjb_config;

/* Sizes of the caller-owned scratch buffers for a config. */
typedef struct psp_hjb_sizes {
    int64_t path_bytes;       /* X_n store for the backward pass (0 if !store_path)  */
    int64_t fwd_partial_bytes;/* per-workgroup (sum D, sum D^2) fp64 pairs            */
    int64_t grad_partial_bytes;/* per-workgroup partial gradients                     */
    int32_t n_params;         /* p = (d+1)H+H + H*H+H + H*d+d                         */
    int32_t fwd_workgroups;
    int32_t bwd_workgroups;
    int32_t fwd_coop_tiles;   /* 0.4.0 (the former `reserved`): 2 or 4 when psp_hjb_rollout_fwd runs the cooperative wide forward
                               * (hjbc_fwd_kernel: that many 16-trajectory tiles per 512-thread workgroup), else 0 */
} psp_hjb_sizes;

int psp_version(void);
const char* psp_last_error(void);
/* sizeof() of the six structs above / below, in declaration order (psp_hjb_config, psp_hjb_sizes, psp_gen_config,
 * psp_gen_sizes, psp_dnet_config, psp_dnet_sizes): lets a binding check its own struct declarations at load time. */
int psp_abi_struct_sizes(int32_t out[6]);
/* ... and of psp_genl_config, psp_genl_sizes (added in 0.3.0). */
int psp_abi_struct_sizes2(int32_t out[2]);

/* 1 if a compiled kernel instantiation exists for (d, H), else 0. */
int psp_hjb_supported(int32_t d, int32_t H);
/* Kernel family that serves (d, H): 0 none, 1 narrow (state panel in registers, tables in LDS; any flag
 * combination of psp_hjb_config), 2 wide (large d: tables in global memory). */
int psp_hjb_family(int32_t d, int32_t H);
/* Enumeration of the compiled (d, H) instances (family as above).  A configuration whose (d, H) is not in the
 * list runs EXACTLY on any instance with larger d and H after zero padding (padded state components never couple
 * back: zero weight rows / columns, zero drift, sigma, running- and terminal-cost entries); the host mirror
 * (native_shapes.py) picks the cheapest one and keeps the index map between the real and the padded flat
 * parameter / gradient vectors. */
int psp_hjb_instance_count(void);
int psp_hjb_instance_get(int32_t i, int32_t* d, int32_t* H, int32_t* family);

/* Fills *out; returns <0 if the config is not supported. */
int psp_hjb_query(const psp_hjb_config* cfg, psp_hjb_sizes* out);

/*
 * Forward rollout: replaces the n-loop of Solver.train (solver.py:440-478) plus
 * initialize_training_data (:364-382) and D = Y - g(X_N) of loss_function (:167-168).
 *   params   : flat control-net parameters (device)
 *   x0       : initial states, x0_stride = 0 -> one (d) vector broadcast (solver.py:365),
 *              x0_stride = d -> (K_local, d) row-major (random_X_0, solver.py:367)
 *   y0       : device pointer to the learnable scalar Y_0 (solver.py:372-373) or NULL (Y=0)
 *   xi       : PSP_NOISE_SUPPLIED: (N+1, K_local, d) fp32, slice n+1 drives step n
 *              (the reference's (K,d,N+1) tensor permuted); ignored for PHILOX
 *   seed,iter: Philox key / iteration counter
 *   path     : X_n store (path_bytes) or NULL
 *   D_out    : (K_local) fp32, D_k = Y_k - g(X_N,k)
 *   XN_out   : optional (K_local, d) final states or NULL
 *   Y_out    : optional (K_local) Y_N (losses that need Y and g separately) or NULL
 *   fwd_partial: fwd_partial_bytes scratch
 * Three kernels implement this call for the narrow family (d <= 112), chosen from the trajectory count of the launch so that
 * the chip stays busy: one 16-trajectory tile per wave (hjb_fwd_kernel), a tile split over the eight waves of a workgroup
 * when there are at most two tiles per CU (hjbs_fwd_kernel, K <= 8192 on 256 CUs), four trajectories per workgroup on
 * v_mfma_f32_4x4x1 when there are at most CUs / 4 tiles (hjbq_fwd_kernel, K <= 1024).  Same Philox counters and path-store
 * format; D agrees to summation order.  The environment variable PSP_FWD_VARIANT=1 / 2 / 3 forces one (tests, A/B timing).
 */
int psp_hjb_rollout_fwd(const psp_hjb_config* cfg, const float* params, const float* x0, int32_t x0_stride,
                        const float* y0, const float* xi, uint64_t seed, uint32_t iter, float* path,
                        float* D_out, float* XN_out, float* Y_out, double* fwd_partial, void* stream);

/*
 * Forward-only controlled rollout for importance-sampling evaluation: replaces the n-loop of
 * utilities.do_importance_sampling_me (reference utilities.py:309-328) with u = -Z_n (control='approx').
 * Same kernel as psp_hjb_rollout_fwd with store_path = 0 and
 *   tfeat    : (N) fp32 network time input per step -- the reference maps t = n*delta_t to
 *              ceil(t / model.delta_t) * model.delta_t (solver.py:360-362); NULL -> n*dt
 *   D_out    : Y_N - g(X_N) with Y accumulated as in Solver.train; the Girsanov log-weight of
 *              utilities.py:330-336 is  D_out - 2*Fint_out  (-int f - g - int u.dW - 0.5 int |u|^2)
 *   Fint_out : optional (K_local) sum_n f(X_{n+1}) dt
 */
int psp_hjb_rollout_eval(const psp_hjb_config* cfg, const float* params, const float* x0, int32_t x0_stride,
                         const float* xi, uint64_t seed, uint32_t iter, const float* tfeat, float* D_out,
                         float* Fint_out, float* XN_out, double* fwd_partial, void* stream);

/* Sums the per-workgroup partials in a fixed order: sums_out[0] = sum_k D_k, sums_out[1] = sum_k D_k^2
 * over this rank's trajectories (fp64, device).  The caller all-reduces sums_out across ranks. */
int psp_hjb_terminal_reduce(const psp_hjb_config* cfg, const double* fwd_partial, double* sums_out, void* stream);

/*
 * Backward pass: replaces loss.backward() of solver.py:221 with the analytic gradient
 * (detach_forward=True: dL/dZ_n[k] = w_k ((Z_n + c) dt + xi_{n+1} sqrt(dt)),
 *  w_k = (2/K)(D_k - mean D) for log-variance, (2/K) D_k for moment).
 *   path     : the path store written by psp_hjb_rollout_fwd with store_path != 0 (register images of
 *              X_n, h1, h2 and of the Brownian increment xi_{n+1} per step and 16-trajectory tile)
 *   xi, seed, iter : accepted for symmetry with the forward call and ignored -- the increments are read
 *              back from `path`, whatever the noise mode (a VALU cycle next to the fp32 MFMA stream
 *              costs more than the extra 448 B per trajectory-step of path store)
 *   sums     : GLOBAL (sum D, sum D^2), fp64 on device (after the all-reduce)
 *   grad_partial : grad_partial_bytes scratch (per-workgroup partial gradients; the wide kernel family
 *              also keeps an operand table there)
 *   grad_out : flat gradient (n_params fp32) for this rank's trajectories, summed in a fixed
 *              order (bitwise reproducible).  The gradient of the learnable Y_0 (moment loss)
 *              is (2/K) sum_k D_k and is formed by the caller from `sums`.
 */
int psp_hjb_rollout_bwd(const psp_hjb_config* cfg, const float* params, const float* xi, uint64_t seed,
                        uint32_t iter, const float* path, const float* D, const double* sums,
                        float* grad_partial, float* grad_out, void* stream);

/*
 * Gradients THROUGH the state path: adaptive_forward_process=True with detach_forward=False, the reference's default
 * flags (solver.py:451-469: c = -Z_n(X_n) is not detached, so loss.backward() of solver.py:221 also differentiates the
 * Euler-Maruyama recursion :471-478).  Call sequence per iteration:
 *     psp_hjb_rollout_fwd   with store_path = 2 (losses of Y_N - g(X_N)) or 3 (PSP_LOSS_REL_ENTROPY), XN_out given
 *     [loss and per-trajectory weights  mu_k = dL/dY_N[k],  nu_k = dL/dZsum_N[k]  from D on the caller's side]
 *     psp_hjb_adjoint_sweep  reverse-time adjoint recursion per trajectory (kernel: csrc/hjba_kernels.h); rewrites the
 *                            xi slot of `path` with dL/dZ_n / sqrt(dt)
 *     psp_hjb_rollout_bwd   with loss_kind = PSP_LOSS_WEIGHTS and D = 1 for every trajectory
 *   XN : (K_local, d) terminal states from the forward call;  mu, nu : K_local floats each (nu may be NULL = 0);
 *   wT : K_local weights of grad g(X_N) in lambda_N, or NULL for nu - mu (losses that depend on X_N only through
 *        Y_N - g(X_N) resp. Zsum_N + g(X_N)); cross_entropy passes -Y_N exp(D) / K (solver.py:183-185);
 *   fwd_partial : the forward call's scratch (the large-d kernel family rebuilds its operand tables in it).
 */
int psp_hjb_adjoint_sweep(const psp_hjb_config* cfg, const float* params, float* path, const float* XN,
                          const float* mu, const float* nu, const float* wT, double* fwd_partial, void* stream);

/* torch.optim.Adam(lr, betas=(b1,b2), eps, weight_decay=0, amsgrad=False) on a flat buffer
 * (function_space.py:185, solver.py:198-200).  step is 1-based. */
int psp_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  int32_t step, float lr, float beta1, float beta2, float eps, void* stream);

/*
 * Launch-bound regime (BASELINE.json configs[1]: K = 1024, N = 50 -- a whole iteration is ~0.3 ms): the six launches of an
 * iteration are captured ONCE into a hipGraph by the caller (any stream-capture API: the entry points only enqueue
 * kernels on the given stream) and replayed.  Everything that changes from one iteration to the next then has to live in
 * device memory: psp_iter_state holds the Philox iteration index and Adam's step count / running beta powers;
 *   psp_hjb_config.iter_dev = &state->iter         forward kernels
 *   psp_hjb_terminal_reduce_loss(..., &state->iter) partial sums -> (sum D, sum D^2) AND the loss value of solver.py:167-168 /
 *                                                   :165-166 / :179-180 into loss_log[state->iter] (single rank: the local
 *                                                   sums are the global ones; replaces six tiny element-wise launches)
 *   psp_adam_step_dev                               Adam with bias corrections 1 - beta^step taken from the state
 *   psp_iter_state_advance                          iter += 1, step += 1, beta powers *= beta (last node of the graph)
 */
typedef struct psp_iter_state {
    uint32_t iter;        /* iteration index l of Solver.train (Philox counter, index into the loss log) */
    uint32_t step;        /* 1-based Adam step the NEXT psp_adam_step_dev applies                        */
    double beta1_pow;     /* beta1 ** step, beta2 ** step (fp64, as torch forms its bias corrections)    */
    double beta2_pow;
} psp_iter_state;
/* Fills a HOST copy (the caller uploads it): state for iteration `iter` whose Adam step is `step` (1-based). */
int psp_iter_state_init(psp_iter_state* host_out, uint32_t iter, int32_t step, float beta1, float beta2);
int psp_iter_state_advance(psp_iter_state* dev_state, float beta1, float beta2, void* stream);
/* psp_hjb_terminal_reduce + the loss of cfg->loss_kind (log-variance / moment / relative entropy) from these sums with
 * K = cfg->K_global, written as fp32 to loss_log[index_dev ? *index_dev : 0].  Only valid when the rank's sums ARE the
 * global sums (one rank); with several ranks call psp_hjb_terminal_reduce, all-reduce, and form the loss from the result. */
int psp_hjb_terminal_reduce_loss(const psp_hjb_config* cfg, const double* fwd_partial, double* sums_out, float* loss_log,
                                 const uint32_t* index_dev, void* stream);
/* psp_hjb_rollout_bwd + gradient reduction + Adam + psp_iter_state_advance as TWO launches instead of four (launch-bound sizes:
 * each tiny launch costs ~4 us inside a replayed graph): the reduction kernel applies the Adam update to the parameter it has
 * just summed and the last of its workgroups advances the state.  Only for the exact (unpadded) parameter layout, one rank,
 * loss kinds the kernels weight themselves; `ticket` is one zero-initialised device uint32 owned by the caller. */
int psp_hjb_rollout_bwd_step(const psp_hjb_config* cfg, float* params, const float* path, const float* D, const double* sums,
                             float* grad_partial, float* grad_out, float* exp_avg, float* exp_avg_sq, psp_iter_state* dev_state,
                             uint32_t* ticket, float lr, float beta1, float beta2, float eps, void* stream);
/* psp_adam_step with step / bias corrections read from a device psp_iter_state. */
int psp_adam_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const psp_iter_state* dev_state, float lr, float beta1, float beta2, float eps, void* stream);

/* Materialises the device noise stream exactly as the rollout kernels consume it:
 * out is (N+1, K_local, d) fp32 with slice 0 zero.  Test / diagnostics helper. */
int psp_philox_normal_fill(float* out, int32_t N, int32_t K_local, int32_t d, int64_t k_offset,
                           uint64_t seed, uint32_t iter, void* stream);

/* Control evaluation u = -Z on a batch (solver.py:349-362): out (K, d) = -MLP([t, X]). */
int psp_hjb_control_eval(int32_t d, int32_t H, const float* params, const float* X, int32_t K, float t,
                         float* minus_Z_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GeneralSolver.train hot path (reference solver.py:1001-1206): diffusion / BSDE loss on unbounded and bounded
 * (sphere / box) domains, V = DenseNet(d+1 -> 1, two hidden layers of width H, relu^2; function_space.py:116-140).
 * EllipticSolver.train (solver.py:628-826, V = DenseNet(d -> 1)) runs through the same entry points with T = +inf
 * and a zero time row in the parameter map (the padded-shape index map of the host adds it).
 * Flat parameter layout = the DenseNet's registration order, weights stored (in, out), input [x, t]:
 *     [W1 ((d+1) x H), b1 (H), W2 ((d+1+H) x H), b2 (H), W3 (d+1+2H), b3 (1)]
 * ------------------------------------------------------------------------------------------------ */
/* nonlinearity h(t,x,y,z): problems.py:1755 (0), :519 (-|z|^2/2), :1204 (y - y^3), and the exponential-on-the-ball
 * family  h = -2 al y (2 al |x|^2 + d) - e y + nl,  nl = 0 (LIN: problems.py:985, :1130 with e = 1),
 * E - y^2 (SQ: :1022), sin(E - y^2) (SIN: :1058, :1166 with e = 1 and the time term), E = exp(2 al |x|^2 + 2 tau t_n);
 * h_par = {al, d, e, tau} */
enum { PSP_GH_ZERO = 0, PSP_GH_QUAD = 1, PSP_GH_ALLEN_CAHN = 2, PSP_GH_EXPBALL_LIN = 3, PSP_GH_EXPBALL_SQ = 4,
       PSP_GH_EXPBALL_SIN = 5 };
/* exit test of a bounded domain: a trajectory stays active while (solver.py:1119-1129; EllipticSolver :758-767)
 *   SPHERE        |X_n| < dom_a               (the state BEFORE the move, as the reference tests it)
 *   BOX           dom_a <= X_proposal <= dom_b in every coordinate
 *   BOX_UPPER_ALL X_proposal <= dom_b in every coordinate   (EllipticSolver, one_boundary)
 *   BOX_UPPER_ANY X_proposal <= dom_b in some coordinate    (GeneralSolver, one_boundary; also the exit test of
 *                                                            'square-corner', solver.py:759-760)
 *   ANNULUS       dom_a < |X_n| < dom_b       ('two_spheres', solver.py:1122-1123, :752-753; the state before the move) */
enum { PSP_DOM_NONE = 0, PSP_DOM_SPHERE = 1, PSP_DOM_BOX = 2, PSP_DOM_BOX_UPPER_ALL = 3, PSP_DOM_BOX_UPPER_ANY = 4,
       PSP_DOM_ANNULUS = 5 };

typedef struct psp_gen_config {
    int32_t d, H, K_local, N;
    int64_t k_offset;     /* global index of local trajectory 0 (Philox counter)            */
    float dt, sqrt_dt;    /* fp32 step and its fp32 square root (solver.py:950-951)          */
    float T;              /* terminal time: a trajectory freezes once t + dt > T (:1131)     */
    float sigma_scale;    /* sigma = s I (problems.py:493 s = 1; :1183,1740 s = sqrt 2)      */
    int32_t drift_kind;   /* PSP_DRIFT_ZERO, PSP_DRIFT_DIAG or PSP_DRIFT_DOUBLE_WELL               */
    int32_t h_kind;
    int32_t adaptive;     /* 1: c = -Z detached (solver.py:1111-1114), 0: c = 0              */
    int32_t noise_mode;   /* PSP_NOISE_SUPPLIED: xi is (N, K_local, d); PSP_NOISE_PHILOX      */
    int32_t store_path;   /* 1: keep what the backward pass needs                            */
    int32_t domain_kind;  /* PSP_DOM_*                                                         */
    const float* drift;   /* DOUBLE_WELL: kappa (d); DIAG: a (d); else NULL                  */
    float dom_a, dom_b;   /* sphere radius (dom_a), box bounds X_l, X_r, or annulus radii r_1, r_2 */
    float h_par[4];       /* PSP_GH_EXPBALL_*: al, d (the REAL dimension, not a padded one), e, tau */
    int32_t d_real;       /* components the exit test and |x|^2 read when d is a zero-padded instance (0: all d);
                           * the padding carries device noise, which nothing else ever reads                */
    int32_t mlp_dtype;    /* PSP_MLP_FP32 (0); PSP_MLP_BF16_FWD: the value-net products of the forward rollout (V, grad_x V,
                           * tangent pass) on v_mfma_f32_16x16x32_bf16 -- bf16 operands, fp32 accumulate; PSP_MLP_BF16: also the
                           * adjoint products and the weight-gradient outer products of the backward kernel.  State, Y,
                           * accumulators and every element-wise step stay fp32 (BASELINE.json configs[2]); with PSP_MLP_BF16 the
                           * path store holds the six images as bf16 pairs (960 instead of 1 920 bytes per sample:
                           * psp_gen_query reports the size), with PSP_MLP_BF16_FWD it stays fp32.
                           * PSP_MLP_F16X3: fp32-GRADE split products (psp_hjb_config.mlp_dtype) in the forward rollout and -- with
                           * shared trajectory weights -- in the backward kernel (per_sample_weights keeps fp32 MFMA there); fp32
                           * path store, the 1e-4 bounds of PSP_MLP_FP32; the network factors multiplying the scaled weights in
                           * the backward pass (w3 phi', W2 products) must stay below 256 in magnitude                        */
    /* Solver.train with approx_method='value_function' (solver.py:93-97, 334-339, 438-440: Z = sigma grad_x Y_n(X), loss +
     * mean_k sum_{n>=1} (Y_n(X_n) - Y)^2) runs on these kernels too (plan_value_native.py):                              */
    float* v_steps_out;   /* optional (N, 16*ceil(K_local/16)): V(X_n, t_n) at every step, written by psp_gen_rollout_fwd       */
    float* y_steps_out;   /* optional, same shape: the running Y before the increment of step n (both or neither)               */
    int32_t per_sample_weights; /* psp_gen_rollout_bwd: 1 -> wY is (N+1, 16*ceil(K_local/16)) = weight of the TANGENT part of every
                           * sample (0 in slot N) and `ahat` holds the coefficient of grad_theta V of every sample itself
                           * (wV is ignored): losses with a term at every step.  0: per-trajectory wY, wV as described below     */
    int32_t reserved;
    int32_t* range_flag;  /* optional DEVICE int32[4]: range guard of PSP_MLP_F16X3 as in psp_hjb_config.range_flag -- the forward sets
                           * range_flag[0] iff V(X_N) or Y_N of any trajectory is non-finite and enqueues the fp32-MFMA forward
                           * predicated on it; psp_gen_rollout_bwd enqueues both backward kernels, predicated.  NULL: unguarded    */
} psp_gen_config;

typedef struct psp_gen_sizes {
    int64_t path_bytes;        /* (N+1) sample slots x K/16 blocks of register images               */
    int64_t ahat_bytes;        /* (N+1) x 16*ceil(K/16) floats                                       */
    int64_t grad_partial_bytes;
    int32_t n_params, fwd_workgroups, bwd_workgroups, reserved;
} psp_gen_sizes;

int psp_gen_supported(int32_t d, int32_t H);
/* enumeration of the compiled GeneralSolver instances; zero padding works as for the HJB kernels (the time input stays
 * the LAST row of W1 / the x-t block of W2, W3: the host index map moves it from row d to row d_pad) */
int psp_gen_instance_count(void);
int psp_gen_instance_get(int32_t i, int32_t* d, int32_t* H);
int psp_gen_query(const psp_gen_config* cfg, psp_gen_sizes* out);

/* Forward rollout (solver.py:1076-1160 and V(X_N,t_N) of :1163): x0 (K_local,d), t0 (K_local) initial
 * points/times; outputs per trajectory V(X_N,t_N), Y_N, X_N (K_local,d), t_N and the active-step count
 * (K_log, :1152,1168) accumulated into *kcount (device u64, zeroed by the caller). */
int psp_gen_rollout_fwd(const psp_gen_config* cfg, const float* params, const float* x0, const float* t0,
                        const float* xi, uint64_t seed, uint32_t iter, float* path, float* ahat, float* VN,
                        float* YN, float* XN, float* tN, unsigned long long* kcount, void* stream);

/* Backward pass (replaces loss.backward() of solver.py:1187 for the domain part of the loss):
 *   grad_out = sum_k [ wV_k dV(X_N,t_N)/dtheta + wY_k dY_N/dtheta ]   over this rank's trajectories,
 * wY = dLoss/dY_N, wV = dLoss/dV(X_N,t_N) per trajectory, each ZERO-PADDED to 16*ceil(K_local/16)
 * floats (the kernel reads them with unconditional 16-byte loads), formed by the caller from
 * VN, YN (diffusion: wV = 2 a0 (VN - YN)/K = -wY ; BSDE: wV = 0, wY = 2 (YN - f(XN))/K). */
int psp_gen_rollout_bwd(const psp_gen_config* cfg, const float* params, const float* path, const float* ahat,
                        const float* wY, const float* wV, float* grad_partial, float* grad_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GeneralSolver / EllipticSolver with a value net of ANY depth: V = dense-concat net (d [+ 1] -> 1, arch = [H_1 .. H_L]),
 * 1 <= L <= 4, H_i <= 128, d + 1 <= 112 (reference function_space.py:116-140 DenseNet, relu^2; :143-158 DenseNet_tanh, tanh with
 * nn.Linear weights; the nets the diffusion-loss notebooks swap into model.V: Allen-Cahn.ipynb:72 arch = [110, 110, 50],
 * [30, 30, 30, 30], `Committor function.ipynb` cell 1: [d + 10, d, d, d] with tanh^2).  Shapes and activation are run-time
 * arguments: activations in per-tile LDS images, weights as A-operand tables in global memory (csrc/genl_kernels.h).  The
 * step is the one of psp_gen_rollout_fwd (same noise counters, exit tests, h kinds, outputs); a tile whose trajectories have
 * all stopped leaves the time loop early (solver.py:1093-1097 / :742-744) and the backward pass skips what it did not execute.
 */
enum { PSP_ACT_RELU2 = 0,   /* h = relu(z)^2   (function_space.py:138)                       */
       PSP_ACT_TANH2 = 1,   /* h = tanh(z)^2   (Committor function.ipynb, DenseNet_tanh_2)    */
       PSP_ACT_TANH = 2 };  /* h = tanh(z)     (function_space.py:157)                        */

typedef struct psp_genl_config {
    psp_gen_config base;      /* d = state dimension; H, mlp_dtype, d_real, range_flag unused */
    int32_t has_time;         /* 1: network input [x, t] (GeneralSolver), 0: [x] (EllipticSolver; base.T = +inf)                 */
    int32_t n_hidden;         /* L                                                                                              */
    int32_t widths[4];        /* H_1 .. H_L                                                                                     */
    int32_t activation;       /* PSP_ACT_*                                                                                      */
    int32_t linear_layout;    /* 0: weights stored (in, out) (DenseNet); 1: (out, in) (nn.Linear, DenseNet_tanh)                */
    int32_t time_first;       /* 1: the net's input is [t, x] (Solver.Y_n, solver.py:338) instead of [x, t]: only the parameter
                               * index map changes, the kernels keep the time in their last input row                          */
    float time_scale;         /* the net sees time_scale * t (0 = 1): Solver's value-function ansatz feeds the STEP INDEX
                               * n = t / dt as the time (solver.py:336, 439)                                                    */
} psp_genl_config;
/* base.v_steps_out / y_steps_out / per_sample_weights of the embedded psp_gen_config are honoured by psp_genl_rollout_fwd /
 * psp_genl_rollout_bwd exactly as by the psp_gen_* entry points (Solver(approx_method='value_function'), plan_value_native.py) */

typedef struct psp_genl_sizes {
    int64_t table_bytes;      /* scratch for the operand tables (rebuilt by every forward call)                                 */
    int64_t path_bytes;       /* (N + 1) x ceil(K/16) blocks of x and s u^ images                                               */
    int64_t ahat_bytes;       /* (N + 1) x 16 ceil(K/16) floats, then ceil(K/16) int32: the step count of every tile            */
    int64_t n_params;         /* registration order W_1, b_1, .., W_out, b_out                                                  */
    int64_t grad_partial_bytes;     /* per-workgroup partial gradients of psp_genl_rollout_bwd                                  */
    int32_t n_blocks;         /* (N + 1) x ceil(K/16) sample blocks                                                              */
    int32_t fwd_workgroups, bwd_workgroups;
    int32_t waves_per_tile;   /* 1 (small nets, no barriers) or 8                                                               */
    int32_t seg_block_offset[5];    /* first padded 16-feature block of segment s (0: input, s: h_s); [L] + ceil(H_L/16) = TB    */
    int32_t reserved;
} psp_genl_sizes;

/* <0: shape outside the limits above, or the activation images exceed 160 KiB. */
int psp_genl_query(const psp_genl_config* cfg, psp_genl_sizes* out);
/* Forward rollout; arguments as psp_gen_rollout_fwd plus the table scratch.  `ahat` is required (it carries the tiles' step
 * counts behind the coefficients). */
int psp_genl_rollout_fwd(const psp_genl_config* cfg, const float* params, const float* x0, const float* t0, const float* xi,
                         uint64_t seed, uint32_t iter, float* tables, float* path, float* ahat, float* VN, float* YN,
                         float* XN, float* tN, unsigned long long* kcount, void* stream);
/* Backward pass (replaces loss.backward() of solver.py:1187 / :814 for the domain part of the loss), arguments as
 * psp_gen_rollout_bwd: wY, wV per-trajectory loss weights zero padded to 16 ceil(K/16).  One kernel recomputes the activations
 * of every executed sample block, runs the adjoint sweep and accumulates all parameter gradients (weight tiles as MFMA outer
 * products over the samples of the block); grad_partial (psp_genl_sizes.grad_partial_bytes) is summed in a fixed order.
 * `tables` must hold the tables of the SAME parameters (psp_genl_rollout_fwd leaves them there). */
int psp_genl_rollout_bwd(const psp_genl_config* cfg, const float* params, const float* tables, const float* path,
                         const float* ahat, const float* wY, const float* wV, float* grad_partial, float* grad_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Solver.train with a DenseNet control (function_space.py:116-140: dense-concat layers, relu^2, weights (in, out)):
 *   time_approx='outer' (the constructor default, solver.py:88, 352-353): N nets DenseNet(d -> d), one per time step
 *                                                              -> per_step = 1, time_input = 0
 *   a DenseNet(d+1 -> d) swapped into z_n with time_approx='inner' (solver.py:142-162, 355: input [t, x])
 *                                                              -> per_step = 0, time_input = 1
 * Forward rollout (psp_dnet_rollout_fwd) and parameter gradient (psp_dnet_rollout_bwd, a hand-written kernel over the register
 * images the rollout leaves in psp_dnet_config.images_out).  Instances whose accumulator tiles do not fit a wave's
 * registers (psp_dnet_sizes.bwd_supported = 0: d = 256) get the rollout's row-major stores instead (px, pxi, r1_out, r2_out)
 * and the caller forms the same gradient with library GEMMs, see plan_dense_native.py.
 * base.d / base.H name the compiled instance (psp_dnet_instance_get; whole 16-blocks); d_real <= d, H_real <= H are
 * the net's sizes.  params = per_step ? N : 1 consecutive parameter sets in the DenseNet's registration order
 *     [W1 (di x H_real), b1, W2 ((di+H_real) x H_real), b2, W3 ((di+2 H_real) x d_real), b3],  di = d_real + time_input,
 * read with their real strides (no padded copy).  Problem vectors / matrices, x0 and supplied noise are padded to
 * base.d as for the other entry points.  Gradients through the state path and the relative-entropy loss run through
 * psp_dnet_adjoint_sweep (below) on the instances with bwd_supported = 1.
 * ------------------------------------------------------------------------------------------------ */
typedef struct psp_dnet_config {
    psp_hjb_config base;
    int32_t d_real, H_real;
    int32_t time_input;   /* 1: the net's input is [t, x] (time = column 0) */
    int32_t per_step;     /* 1: one parameter set per time step            */
    float* r1_out;        /* optional (N, K_local, H_real): relu(z1) and relu(z2) of every sample, row-major -- spares the   */
    float* r2_out;        /* gradient pass the recomputation of the two hidden layers (both NULL: not stored)                */
    float* images_out;    /* optional (psp_dnet_sizes.image_bytes): X_n, relu(z1), relu(z2) and the xi image as register images
                           * for psp_dnet_rollout_bwd; when set they REPLACE the row-major stores (px, pxi, r1_out, r2_out) */
} psp_dnet_config;

typedef struct psp_dnet_sizes {
    int64_t table_bytes;       /* scratch for the A-operand tables the call writes (L2-resident)   */
    int64_t fwd_partial_bytes;
    int64_t n_params_per_set;
    int32_t fwd_workgroups, reserved;
    /* hand-written backward (psp_dnet_rollout_bwd); bwd_supported = 0 when the instance's accumulators do not fit */
    int64_t image_bytes;       /* N x ceil(K/16) image blocks                                          */
    int64_t partial_bytes;     /* N x slices partial gradients of padded_params floats                 */
    int32_t bwd_supported, slices, padded_params, bwd_workgroups;
} psp_dnet_sizes;

int psp_dnet_instance_count(void);
int psp_dnet_instance_get(int32_t i, int32_t* d, int32_t* H);
int psp_dnet_query(const psp_dnet_config* cfg, psp_dnet_sizes* out);
int psp_dnet_terminal_reduce(const psp_dnet_config* cfg, const double* fwd_partial, double* sums_out, void* stream);
/* px, pxi: (N, K_local, d_real) stores of X_n and of the xi image (xi, or xi + sqrt(dt) Z when the forward process is
 * not adaptive: dL/dZ_n = w_k sqrt(dt) * image either way); written when base.store_path = 1.  tfeat: optional (N)
 * time feature per step (importance-sampling grids, utilities.py:296-299), NULL -> n * dt.  Outputs as for
 * psp_hjb_rollout_fwd / psp_hjb_rollout_eval; psp_dnet_terminal_reduce reduces fwd_partial (this family's own
 * workgroup count) to the global (sum D, sum D^2). */
int psp_dnet_rollout_fwd(const psp_dnet_config* cfg, const float* params, const float* x0, int32_t x0_stride,
                         const float* y0, const float* xi, uint64_t seed, uint32_t iter, const float* tfeat,
                         float* px, float* pxi, float* D_out, float* Fint_out, float* XN_out, float* Y_out,
                         double* fwd_partial, float* tables, void* stream);

/* Gradients THROUGH the state path for a DenseNet control (adaptive_forward_process=True with detach_forward=False -- the
 * reference's constructor defaults, solver.py:23-24, 451-469) and the relative-entropy loss (:179-180, 484-486): the
 * counterpart of psp_hjb_adjoint_sweep.  Call sequence per iteration:
 *     psp_dnet_rollout_fwd   with base.store_path = 2 (losses of Y_N - g(X_N)) or 3 (PSP_LOSS_REL_ENTROPY), images_out and
 *                            XN_out given
 *     [loss and per-trajectory weights mu_k = dL/dY_N[k], nu_k = dL/dZsum_N[k] from D on the caller's side]
 *     psp_dnet_adjoint_sweep reverse-time adjoint recursion per trajectory tile (csrc/hjbd_kernels.h: hjbd_adj_kernel, with the
 *                            Jacobian of the dense-concat net from the stored relu images); rewrites the xi slot of `images`
 *                            with dL/dZ_n / sqrt(dt)
 *     psp_dnet_rollout_bwd   with w = 1 for every trajectory
 *   XN : (K_local, base.d) terminal states;  mu, nu, wT as for psp_hjb_adjoint_sweep (nu / wT may be NULL);
 *   tables : the forward call's table scratch (the sweep rebuilds it with the transposed orientations). */
int psp_dnet_adjoint_sweep(const psp_dnet_config* cfg, const float* params, float* images, const float* XN,
                           const float* mu, const float* nu, const float* wT, float* tables, void* stream);

/* Parameter gradient of sum_{n,k} w_k sqrt(dt) image_n[k] . Z_n(X_n[k]) (= dL/dtheta for a detached forward process) from the
 * images the forward wrote (cfg->images_out).  w: per-trajectory weights dL/dD_k, zero padded to 16 * ceil(K_local/16).
 * partial: (N * slices, padded_params) -- per work item (time step, slice of its tiles) in the PADDED layout of the instance
 * (d x H etc., no time rows):  [W1 (d x H), b1 (H), W2 ((d+H) x H), b2, W3 ((d+2H) x d), b3 (d)];  the caller sums the slices of a
 * step, cuts the real rows / columns out and -- with time_input -- forms the time rows as sum_n t_n * (bias gradient of step n). */
int psp_dnet_rollout_bwd(const psp_dnet_config* cfg, const float* params, const float* images, const float* w,
                         float* partial, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, trajectories sharded in contiguous blocks, parameters replicated.
 * Per iteration two in-place SUM all-reduces on the caller's stream: (sum D, sum D^2) after psp_hjb_terminal_reduce
 * (2 x fp64; the log-variance loss needs the GLOBAL mean) and the flat gradient after psp_hjb_rollout_bwd (n_params
 * fp32), then the same psp_adam_step on every rank.  The reference has no distributed code; these entry points are
 * what a C / ctypes consumer of this library uses instead of torch.distributed.  They drive RCCL (librccl.so.1 is
 * bound at first use, so the library loads on machines without it): both messages are latency-bound (16 B, 69 KB at
 * d=100), far from the per-link xGMI bandwidth.
 *   psp_comm_unique_id : rank 0 creates the 128-byte id and hands it to the other ranks by any host-side means
 *   psp_comm_init      : collective over all ranks, after hipSetDevice; *comm_out is an ncclComm_t
 *   psp_allreduce      : in-place SUM of n elements (dtype PSP_DT_F32 / PSP_DT_F64), asynchronous on `stream`
 * ------------------------------------------------------------------------------------------------ */
#define PSP_COMM_ID_BYTES 128
enum { PSP_DT_F32 = 0, PSP_DT_F64 = 1 };
int psp_comm_unique_id(unsigned char id_out[PSP_COMM_ID_BYTES]);
int psp_comm_init(void** comm_out, int32_t nranks, int32_t rank, const unsigned char id[PSP_COMM_ID_BYTES]);
int psp_comm_destroy(void* comm);
int psp_allreduce(void* buf, int64_t n, int32_t dtype, void* comm, void* stream);

/* Diagnostics: device buffer that receives per-wave phase cycle sums (8 u64 per wave of the
 * backward kernel).  Returns 1 if the library was built with -DPSP_STAMPS (diagnostic build,
 * never the shipped one), 0 otherwise (the pointer is then ignored). NULL clears it. */
int psp_debug_set_stamp_buffer(unsigned long long* buf, int64_t n_entries);

#ifdef __cplusplus
}
#endif
#endif /* PSP_H_ */
