/* openpystruct_amd -- C ABI of the MI355X-native batched beam FE solve.
 *
 * This is the drop-in boundary of the hot path.  The reference has no FFI of its own for
 * it: the boundary it crosses is the `openseespy.opensees` command API that
 * `setup_model` + `generate_sample` drive, one case at a time
 * (/root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py):
 *
 *   ops.wipe / model / node / fix / geomTransf / element('elasticBeamColumn') /
 *   timeSeries / pattern / load / eleLoad('-beamUniform')            :93-117   (model build)
 *   ops.system('BandSPD') / numberer / constraints('Plain') / integrator / algorithm /
 *   ops.analysis('Static') ; ops.analyze(1)                          :120-124, :180-182
 *   ops.eleResponse(e,'forces')[1], [2]                              :189-190
 *   ops.nodeDisp(n, 2), ops.nodeDisp(n, 3)                           :224-232
 *
 * One call of `ops_beam_solve_batched_f64` replaces that whole command sequence for B
 * independent cases.  All pointers are DEVICE pointers owned by the caller; nothing is
 * allocated, copied or synchronised inside; the work is enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream).
 *
 * Binding examples (ctypes / cgo / JNI): INTEGRATION.md.
 */
#ifndef OPENPYSTRUCT_AMD_H
#define OPENPYSTRUCT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPS_AMD_ABI_VERSION 13

/* return codes of the launch functions (per-beam results are in `status`) */
#define OPS_AMD_OK 0
#define OPS_AMD_ERR_INVALID_ARG 1   /* null pointer, B < 0, Ne out of range, bad stride */
#define OPS_AMD_ERR_UNSUPPORTED 2   /* Ne larger than the largest compiled tiling */
#define OPS_AMD_ERR_LAUNCH 3        /* HIP reported a launch error (see ops_amd_last_error) */

/* fix[] bits per node -- `ops.fix(node, fx, fy, rz)` (SingleCore.py:100-102).  The x
 * translation decouples from bending on a straight horizontal beam and is ignored. */
#define OPS_AMD_FIX_UY 1u
#define OPS_AMD_FIX_RZ 2u

/* Batched static solve of B straight Euler-Bernoulli beams with Ne elements (N = Ne+1 nodes).
 *
 * Replaces, per beam: setup_model (SingleCore.py:89-124) + analysis/analyze (:180-182) +
 * 2*Ne eleResponse calls (:189-190) + 2*N nodeDisp calls (:224-232).
 *
 * Batch strides are in ELEMENTS of the array's type; a stride of 0 means "one value /
 * one row shared by every beam".
 *
 *   x      [N] (x_bstride 0) or [B,N]     node coordinates along the beam   (`ops.node`)
 *   E      scalar (E_bstride 0) or [B,Ne] Young's modulus per element       (`ops.element` arg E)
 *   I      [B,Ne]  (I_bstride >= Ne)      second moment of area per element (`ops.element` arg Iz)
 *   fix    [N] (fix_bstride 0) or [B,N]   OPS_AMD_FIX_* bits per node       (`ops.fix`)
 *   Fy     [B,N]   (Fy_bstride >= N)      nodal transverse point loads      (`ops.load(node,0,F,0)`)
 *   wy     scalar (wy_bstride 0) or [B,Ne] transverse UDL per element       (`ops.eleLoad -beamUniform Wy`)
 *
 * Outputs (dense, row-major):
 *   v      [B,N]   u_y per node            (`ops.nodeDisp(n, 2)`)
 *   theta  [B,N]   theta_z per node        (`ops.nodeDisp(n, 3)`)
 *   V      [B,Ne]  eleResponse(e,'forces')[1]  (global Fy at element node I)
 *   M      [B,Ne]  eleResponse(e,'forces')[2]  (Mz at element node I)
 *   status [B]     0 = solved; non-zero = stiffness matrix not positive definite
 *                  (what `ops.analyze(1)` reports with a non-zero return,
 *                  MultiCore.py:182-186); outputs of such a beam are NaN.  May be NULL.
 *
 * `tiling`: 0 = choose from B and Ne; otherwise lanes-per-beam P in {8, 16, 32, 64}
 * (64 = one wavefront per beam); OR-ed with OPS_AMD_TILING_STREAM_OUT the results are written with
 * non-temporal stores: for callers that cycle through more output than the 256 MiB Infinity Cache
 * holds (measured at 10^4 beams: cache-resident buffers 13.8 us default / 15.4 us streaming, HBM-resident
 * buffers 18.4 / 15.8 us; profiles/r02_notes.md).  Returns OPS_AMD_OK or an OPS_AMD_ERR_* code.
 * Never throws, never blocks.
 */
#define OPS_AMD_TILING_STREAM_OUT 0x100
/* OR-ed into `tiling` (with P = 16 or 8): the row-staged variant of that tiling (csrc/beam_fat.hip; shared geometry and
 * constraint mask only, else OPS_AMD_ERR_UNSUPPORTED).  P = 6, the fat-wave tiling, exists only in that form. */
#define OPS_AMD_TILING_ROWS 0x200
int ops_beam_solve_batched_f64(int B, int Ne,
                               const double* x, long x_bstride,
                               const double* E, long E_bstride,
                               const double* I, long I_bstride,
                               const uint8_t* fix, long fix_bstride,
                               const double* Fy, long Fy_bstride,
                               const double* wy, long wy_bstride,
                               double* v, double* theta, double* V, double* M,
                               int32_t* status, int tiling, void* stream);

/* The sizing loop's per-epoch solve (SingleCore.py:176-190 reads only eleResponse inside the loop; the displacements are
 * read once, after it, :224-232): same arguments, element end forces only (no v / theta traffic), and an optional
 * `active` mask [B] -- a wavefront whose beams are all inactive (early-stopped cases) returns immediately. */
int ops_beam_solve_forces_f64(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                              const double* I, long I_bstride, const uint8_t* fix, long fix_bstride,
                              const double* Fy, long Fy_bstride, const double* wy, long wy_bstride, double* V, double* M,
                              int32_t* status, const uint8_t* active, int tiling, void* stream);
/* ... and with the forces already rounded to float32 rows (what SingleCore.py:189-190 does with them): half the bytes. */
int ops_beam_solve_forces_f32(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                              const double* I, long I_bstride, const uint8_t* fix, long fix_bstride,
                              const double* Fy, long Fy_bstride, const double* wy, long wy_bstride, float* V32, float* M32,
                              int32_t* status, const uint8_t* active, int tiling, void* stream);

/* Hyper-parameters of the per-case sizing optimiser (module-level constants of the reference,
 * SingleCore.py:20-44): E, G = E / 2.6, alpha_moment = alpha_shear = 1e-2, lr = 0.01, gamma = 0.98,
 * tolerance = 5e-3, patience = 5 (MultiCore: 10, GPU script: 1e-2 / 100), max_epochs = 600,
 * clamp_min = 1e-8, bend_eps = 1e-6 (:195), area_coef = 0.03 (:196); Adam defaults of torch. */
typedef struct ops_sizing_params {
  double E, G;
  double alpha_moment, alpha_shear;
  double lr, gamma;
  double beta1, beta2, adam_eps;
  double clamp_min, bend_eps, area_coef;
  double tolerance;
  int32_t patience, max_epochs;
} ops_sizing_params;

/* One optimiser epoch for B cases (everything after the FE solve in the reference's epoch loop,
 * SingleCore.py:189-219): float32 rounding of shear/moment, loss, gradient of the explicit I terms,
 * Adam + ExponentialLR + clamp, early-stop bookkeeping.  Cases with active[b] == 0 are skipped.
 *
 *   I            [B,Ne] float32 in/out   the reference's I_tensor (:163)
 *   I64          [B,Ne] float64 out      widened copy read by the next ops_beam_solve_batched_f64; NOT
 *                                        refreshed for a case that stops in this call, so the solver keeps
 *                                        reproducing that case's last solve (records lag I by one step, :239)
 *   V, M         [B,Ne] float64 in       outputs of the solve of this epoch
 *   exp_avg, exp_avg_sq [B,Ne] float32   Adam moments (zero-initialised by the caller)
 *   best_loss    [B] float32 (init +inf), patience_cnt [B] int32 (init 0), epochs_run [B] int32 (init 0),
 *   active       [B] uint8 (init 1): cleared when patience runs out or max_epochs is reached
 *   last_loss    [B] float32 out, V32/M32 [B,Ne] float32 out: the recorded `shear_forces` / `bending_moments`
 *                (both may be NULL: a caller that re-solves once after the loop rounds V / M itself)
 * Device pointers, asynchronous on `stream`, nothing allocated.  Ne <= 512. */
int ops_beam_sizing_step_f32(int B, int Ne, float* I, double* I64, const double* V, const double* M,
                             float* exp_avg, float* exp_avg_sq, float* best_loss, int32_t* patience_cnt,
                             int32_t* epochs_run, uint8_t* active, float* last_loss, float* V32, float* M32,
                             const ops_sizing_params* hp, void* stream);
/* The same step reading the float32 rows written by ops_beam_solve_forces_f32 (they ARE the recorded V32 / M32).
 * `schedule` (device, [max_epochs, 2] float32, or NULL): per-epoch step size and sqrt(1 - beta2^(t+1)) as tabulated on
 * the host by ops_sizing_schedule_f32 -- without it every wavefront evaluates three double-precision pow(). */
int ops_beam_sizing_step_vm32_f32(int B, int Ne, float* I, double* I64, const float* V32, const float* M32,
                                  float* exp_avg, float* exp_avg_sq, float* best_loss, int32_t* patience_cnt,
                                  int32_t* epochs_run, uint8_t* active, float* last_loss, const ops_sizing_params* hp,
                                  const float* schedule, void* stream);
void ops_sizing_schedule_f32(const ops_sizing_params* hp, float* schedule_host /* [max_epochs, 2] */);

/* Case randomisation of the dataset generator (SingleCore.py:133-160 draws, :100-113 `ops.fix` / `ops.load`) as one launch,
 * C-ABI version 8: cases first_case .. first_case + B - 1 of the list that `seed` defines (case i is a pure function of
 * (seed, i): counter-based stream, csrc/case_draw.hip).  random_bridge = 1: L = L_min + U(0, 1) L_max and 1 .. n_rollers_max
 * distinct rollers among nodes 2 .. num_nodes - 1; otherwise L = L_max and the `n_fixed` rollers of `fixed_rollers` (HOST
 * array, 1-based node ids).  Then 1 .. m_forces_max distinct loaded nodes that are not rollers, loads U(max_force, min_force).
 * Device outputs: L [B], roller_nodes [B, R] (R = n_rollers_max or n_fixed; 1-based, 0 = unused slot), n_rollers [B],
 * force_nodes / force_values [B, m_forces_max], n_forces [B], fix [B, num_nodes] (node 1 and the rollers), Fy [B, num_nodes]. */
#define OPS_CASE_MAX_PICKS 8
int ops_sizing_draw_cases_f64(long B, unsigned long long first_case, unsigned long long seed, int num_nodes, int n_rollers_max,
                              int m_forces_max, int random_bridge, const int32_t* fixed_rollers, int n_fixed, double L_min,
                              double L_max, double max_force, double min_force, double* L, long long* roller_nodes,
                              long long* n_rollers, long long* force_nodes, long long* n_forces, double* force_values,
                              uint8_t* fix, double* Fy, void* stream);

/* One WHOLE epoch of the reference's per-sample loop for B cases in one launch (SingleCore.py:176-219): the FE solve on
 * the float32 inertias I (widened to double while they are staged: `I_tensor[i].item()`, :107) followed, in the same
 * wavefront and on the forces it still holds in LDS, by the optimiser step above -- shear and moment never travel through
 * HBM and no widened copy of I is kept (ABI 2; the epoch moves ~32 bytes per element: I 4+4, moments 8+8, loads 8).
 *   I_last  [B,Ne] float32 out: written ONCE per case, in the call in which it stops (patience or max_epochs): the
 *           inertias its last solve ran on.  The reference records shear / moment / displacements of that solve next to
 *           the I of one Adam step later (:189-208 vs :239); a caller reproduces them with one solve on (double)I_last.
 * Other arguments as in ops_beam_solve_batched_f64 (geometry, supports, loads) and ops_beam_sizing_step_vm32_f32 (state). */
int ops_beam_sizing_epoch_f32(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                              const uint8_t* fix, long fix_bstride, const double* Fy, long Fy_bstride, const double* wy,
                              long wy_bstride, float* I, float* I_last, float* exp_avg, float* exp_avg_sq, float* best_loss,
                              int32_t* patience_cnt, int32_t* epochs_run, uint8_t* active, float* last_loss,
                              const ops_sizing_params* hp, const float* schedule, int32_t* status, int tiling, void* stream);

/* Matrix-free FE residual for physics losses (an addition: the reference's "PINN" has no FE operator).
 *   r = D (K(I) u - f):  rv, rt [B,N]; D zeroes the fixed DOFs; f = Fy + consistent beamUniform loads.
 * Same argument meaning / strides as ops_beam_solve_batched_f64; I, Fy, v, theta, rv, rt dense. */
int ops_beam_residual_f64(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                          const double* I, const uint8_t* fix, long fix_bstride, const double* Fy,
                          const double* wy, long wy_bstride, const double* v, const double* theta,
                          double* rv, double* rt, void* stream);

/* Vector-Jacobian product of the residual: given g = dL/dr (gv, gt [B,N]) returns
 *   dv, dt [B,N] = K(I) D g   (dL/du)      dI [B,Ne] = (D g)_e^T (dk_e/dI_e) u_e   (dL/dI)
 * scratch_v, scratch_t [B,N]: caller-provided work space (holds D g). */
int ops_beam_residual_vjp_f64(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                              const double* I, const uint8_t* fix, long fix_bstride, const double* v,
                              const double* theta, const double* gv, const double* gt, double* scratch_v,
                              double* scratch_t, double* dv, double* dt, double* dI, void* stream);

/* The surrogates' FE-residual TERM (r04, ABI 10) in three launches -- what physics.py fe_residual_loss built from ~60 framework nodes around
 * the two entry points above:
 *   I_e = max(preds[b, e] * I_scale[e] + I_mean[e], I_min);   u = (v_rec, t_rec)[rows[b]]  (I-only models; rows NULL: row b)   or
 *   v_n = preds[b, Ne + n] * v_scale[n] + v_mean[n], theta_n = preds[b, Ne + N + n] * t_scale[n] + t_mean[n]   (v_rec == NULL: the PINN);
 *   r = D (K(I) u - f) with Fy [.., N] gathered by rows, shared x [N] / fix [N] / E / wy;   e = r / diag K(I)  (no gradient through the diagonal);
 *   value[0] = weight * (mean e_v^2 / (mean v^2 + 1e-30) + mean e_t^2 / (mean theta^2 + 1e-30)),  value_sum[0] += it (optional).
 * ops_physics_loss_fwd: residual + scaled errors (ev, et [B, N] float64, kept for the backward launch) + the value.
 * ops_physics_loss_bwd: dpreds [B, ldp] (the predictions' dtype) = d value / d preds, ASSIGNED on the columns the term reads (Ne, or
 * Ne + 2 N); the caller zeroes the rest.  part: ops_physics_loss_part_doubles(B, Ne) doubles shared by the two calls. */
typedef struct ops_physics_loss_args {
  int32_t B, Ne;
  const void* preds; int32_t preds_bf16, ldp;
  const float* I_scale; const float* I_mean; float I_min;
  const double* v_rec; const double* t_rec;
  const float* v_scale; const float* v_mean; const float* t_scale; const float* t_mean;
  const long long* rows;
  const double* Fy;
  const double* x; const uint8_t* fix; double E, wy;
  float weight;
  double* ev; double* et; double* part;
  float* value; float* value_sum;
  void* dpreds;
} ops_physics_loss_args;
size_t ops_physics_loss_part_doubles(int B, int Ne);
int ops_physics_loss_fwd(const ops_physics_loss_args* args, void* stream);
int ops_physics_loss_bwd(const ops_physics_loss_args* args, void* stream);


/* Batched 2-D frame solve (3 DOF per node; SURVEY 8(f1), BASELINE config 5): replaces, for B frames that share one
 * topology, `setup_frame_model` + `ops.analyze(1)` + `ops.eleResponse(e,'forces')` of
 * OpenPyStruct_FrameOpt_Discrete_Beta.py:75-139, :151, :181-183.  Host-prepared, shared by the batch:
 *   elem_geo [Ne,3] (L, cos, sin), elem_EA [Ne], elem_E [Ne], elem_w [Ne,2] (beamUniform Wy, Wx: local transverse, axial),
 *   elem_eq [Ne,6] / node_eq [Nn,3]: equation numbers (PlainHandler: -1 for constrained DOFs), n_eq, half_bandwidth.
 * Per frame: I [B,Ne]; loads [Nn,3] (loads_bstride 0) or [B,Nn,3] (`ops.load(node, Fx, Fy, Mz)`).
 * Outputs: disp [B,Nn,3], forces [B,Ne,6] (global resisting forces = eleResponse 'forces'), V / M [B,Ne] = forces[..,1] /
 * forces[..,2] (FR:151-153), status [B] (non-zero: not positive definite, outputs NaN).
 * The factor lives in a caller-provided device workspace of ops_frame_workspace_bytes(B, n_eq, half_bandwidth) bytes: the assembly plan at its
 * start, then about (n_eq + 4) * (window width) * 8 per frame (window width: the half bandwidth + 1 rounded up to 6 / 10 / 12 / 16 / 18 / 22 / 24 / 28 / 30 /
 * 36 / 52 / 56; the kernels keep the sliding window in registers; the packed kernel's share counts whole wavefronts: B rounded up to 4 or 2 frames).  ERR_INVALID_ARG when it is NULL or too small.
 * Small batches (up to the batch at which the kernel families meet: 256 .. 4 000 frames by frame size; the reference's one frame per
 * epoch) take the workgroup-per-frame kernels, which keep the band in LDS when it fits (ops_frame_workspace_bytes is then 0 and
 * `workspace` may be NULL), or -- r06, csrc/frame_coop.hpp -- four wavefronts per frame with the window in registers where that band does not
 * fit or fits a CU only once (workspace: plan + factor, as for the tuned kernels) -- always size the workspace with the B of the call.
 * half_bandwidth <= 29 (98 of the 100 (bays, stories) draws of FR:17-18): 16 or 32 lanes per frame, 4 or 2 frames per wavefront, persistent waves;
 * 30..55: a wavefront per frame; 56..63: the workgroup-per-frame kernels; 64..1024 (more than 20 bays and stories: beyond
 * the reference's range): a plain column-by-column fallback on the band in the workspace, milliseconds per frame;
 * ERR_UNSUPPORTED beyond that or when one right-hand side and one column do not fit 160 KB of LDS. */
int ops_frame_solve_batched_f64(int B, int n_nodes, int n_elems, int n_eq, int half_bandwidth,
                                const double* elem_geo, const double* elem_EA, const double* elem_E,
                                const double* elem_w, const int32_t* elem_eq, const int32_t* node_eq,
                                const double* I, const double* loads, long loads_bstride, double* disp,
                                double* forces, double* V, double* M, int32_t* status, void* workspace,
                                size_t workspace_bytes, void* stream);
size_t ops_frame_workspace_bytes(int B, int n_eq, int half_bandwidth);
/* r06 (ABI 13).  The tuned kernels (a wavefront, or for half_bandwidth <= 29 -- 98 of the 100 (bays, stories) draws of FR:17-18 -- 16 or 32 lanes
 * per frame) assemble each frame's rows from an ASSEMBLY PLAN: the topology-only part of `setup_frame_model` (which element entry goes where,
 * FR:84-131), built by a small kernel at the start of the workspace.  ops_frame_solve_batched_f64 rebuilds it on every call (20-25 us: it cannot
 * know whether the topology arrays changed).  A caller that KNOWS they did not -- same elem_* / node_eq contents, same n_eq / half_bandwidth, same
 * workspace buffer, and ops_frame_plan_signature(B, n_eq, half_bandwidth) equal to that of the call that built it (non-zero; it changes where
 * the batch size changes the kernel family) -- passes OPS_FRAME_REUSE_PLAN and the solve is one launch.  flags = 0: exactly the call above. */
#define OPS_FRAME_REUSE_PLAN 1u
int ops_frame_solve_batched_f64_ex(int B, int n_nodes, int n_elems, int n_eq, int half_bandwidth,
                                   const double* elem_geo, const double* elem_EA, const double* elem_E,
                                   const double* elem_w, const int32_t* elem_eq, const int32_t* node_eq,
                                   const double* I, const double* loads, long loads_bstride, double* disp,
                                   double* forces, double* V, double* M, int32_t* status, void* workspace,
                                   size_t workspace_bytes, void* stream, unsigned flags);
long ops_frame_plan_signature(int B, int n_eq, int half_bandwidth);

/* Fused 3-tap stencil + single-channel batch normalisation of a [B,F] float32 tensor: the PINN ResidualBlock's
 * `bn1(conv1(x.unsqueeze(1))).squeeze(1)` (Conv1d(1,1,3,padding=1) + BatchNorm1d(1),
 * OpenPyStruct_PINN_MultiCase.py:425-452) in two launches forward, three backward; B*F <= 2^26.
 * `workspace`: ops_stencil3_bn1_workspace_bytes() bytes of device memory (per-workgroup partial sums; the backward call
 * may reuse the forward call's).  z / grad_z are float32 or, with the *_is_bf16 flags, bfloat16 (the autocast dtype the
 * framework pair returns): x, parameters, statistics and dx stay float32.
 *   fwd: z = gamma (y - mean) invstd + beta, y = w0 x[i-1] + w1 x[i] + w2 x[i+1] + b; training != 0: batch statistics,
 *        running_mean / running_var momentum update (unbiased variance), *num_batches_tracked += 1 (may be NULL);
 *        training == 0: running statistics.  save[2] receives (mean, invstd) for the backward pass.
 *   bwd: dx [B,F] and dparams[6] = d(w0, w1, w2, b, gamma, beta) from grad_z; `training` as in the forward pass. */
int ops_stencil3_bn1_fwd_f32(int B, int F, const float* x, const float* conv_w, const float* conv_b, const float* gamma,
                             const float* beta, float eps, float momentum, int training, float* running_mean,
                             float* running_var, long long* num_batches_tracked, void* z, int z_is_bf16, float* save,
                             void* workspace, void* stream);
int ops_stencil3_bn1_bwd_f32(int B, int F, const float* x, const void* grad_z, int grad_is_bf16, const float* conv_w, const float* conv_b,
                             const float* gamma, const float* save, int training, float* dx, float* dparams, void* workspace,
                             void* stream);
size_t ops_stencil3_bn1_workspace_bytes(void);

/* Gradient clipping + Adam over flat float32 buffers of n values (the optimiser step of the surrogate training loops:
 * `clip_grad_norm_(params, max_norm)` + `optim.Adam(lr, weight_decay)`, OpenPyStruct_PINN_MultiCase.py:696, :766-768) in
 * two launches.  grads are scaled by grad_scale first (1 / world_size after a sum all-reduce); `lr` and `step` are device
 * scalars (step is advanced by the call); max_norm <= 0 disables clipping; weight decay is torch's L2 form (g += wd p)
 * or, with decoupled_weight_decay & OPS_ADAM_DECOUPLED, AdamW's (p *= 1 - lr wd); decoupled_weight_decay & OPS_ADAM_ZERO_GRADS: `grads` is
 * ZEROED after use (the next step's optimizer.zero_grad(); the pointer is then written through despite its const).  params_bf16 (optional, n bfloat16 values): refreshed with the
 * rounded new parameters, so that bf16 GEMMs of the next step need no per-step cast kernels.
 * `workspace`: ops_flat_adam_workspace_bytes() bytes. */
#define OPS_ADAM_DECOUPLED 1
#define OPS_ADAM_ZERO_GRADS 2
/* r05 (ABI 12): the workspace ALREADY holds the step's (decoupled_weight_decay >> 16) partial sums of (g * grad_scale)^2, the advanced
 * step count and its bias corrections -- written by the producers of the gradients (ops_mlp_wgrad_group_norm) -- so the call skips its
 * norm launch (one kernel node and ~4.5 us less per step of the PINN's training graph). */
#define OPS_ADAM_NORM_READY 4
#define OPS_FLAT_ADAM_MAX_PARTS 1024   /* partial sums a workspace holds; the two bias corrections follow them */
int ops_flat_clip_adam_step_f32(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* lr,
                                int32_t* step, float max_norm, float grad_scale, float beta1, float beta2, float eps,
                                float weight_decay, int decoupled_weight_decay, void* params_bf16, void* workspace, void* stream);
size_t ops_flat_adam_workspace_bytes(void);

/* Training loss of the surrogates, value and gradient w.r.t. the predictions in one pass (two launches):
 * TrainableL1L2Loss on the first nI columns of preds/targets [B,C] (OpenPyStruct_PINN_MultiCase.py:549-601) -- alpha is a
 * device scalar, clamped to [1e-6, 1]; min/max_constraint device scalars or NULL; box_weight = penalty_weight --, plus
 * CompositeLoss's relative-L1 terms (PINN:603-653, weight rel_penalty, eps 1e-8) over the next nD and the remaining
 * C - nI - nD columns, plus (alpha0 - alpha)^2 (TFD:743; pass NaN for none).  preds / grad: float32, or bfloat16 with preds_is_bf16;
 * `grad` = d loss / d preds; `workspace`: ops_surrogate_loss_workspace_bytes() bytes. */
int ops_surrogate_loss_grad_f32(int B, int C, int nI, int nD, const void* preds, int preds_is_bf16, const float* targets,
                                const float* alpha, float alpha0, const float* min_constraint, const float* max_constraint,
                                float box_weight, float rel_penalty, float* loss, void* grad, void* workspace, void* stream);
/* ... that also ADDS the loss value to *loss_sum (a training loop's per-epoch running sum; may be NULL). */
int ops_surrogate_loss_grad_sum_f32(int B, int C, int nI, int nD, const void* preds, int preds_is_bf16, const float* targets,
                                    const float* alpha, float alpha0, const float* min_constraint, const float* max_constraint,
                                    float box_weight, float rel_penalty, float* loss, float* loss_sum, void* grad, void* workspace, void* stream);
size_t ops_surrogate_loss_workspace_bytes(void);


/* Largest Ne a build supports, ABI version, and the text of the last HIP error seen by
 * this thread (empty string if none). */
int ops_amd_max_elements(void);
int ops_amd_abi_version(void);
const char* ops_amd_last_error(void);

/* Library options (r06, ABI 13): the ONE way to steer the dispatch from outside -- the library reads no environment variable.  Process-wide,
 * thread-safe, effective from the next call.  Unknown name: ERR_INVALID_ARG / -2.
 *   "frame_latency_batch"  -1 (default): batches of up to 256 .. 4 000 frames (by frame size) take the workgroup-per-frame kernels, which answer
 *                          a handful of frames sooner; N >= 0: that threshold is N (0: the tuned kernels for every batch)
 *   "frame_pack"           1 (default): half bandwidths up to 29 take 16 or 32 lanes per frame; 0: one wave per frame for every band (A/B)
 *   "frame_coop"           1 (default): small batches take a workgroup of four wavefronts per frame where that was measured faster than the
 *                          workgroup-per-frame kernels (their band not LDS-resident: half bandwidth 44 .. 55 up to 512 frames; or resident only once
 *                          per CU: e.g. 10 x 10 at 257 .. 768 frames); 0: never; 2: for every small batch (A/B, tests)
 *   "deterministic"        0 (default); 1: the Transformer-Diffusion step's gradient launches reduce in a fixed order -- one row split per weight-
 *                          gradient product and one workgroup per column-sum strip (ops_linear_wgrad_accumulate*), one workgroup for the [CLS]
 *                          sums (ops_tfd_front_bwd), the head's LayerNorm sums in workgroup order (ops_tfd_head_bwd) -- so that two runs of one
 *                          seed give the same bits (float atomics otherwise land in arrival order); slower.  Takes effect at launch: a captured
 *                          graph keeps the mode it was captured in.  (The PINN step and the FE kernels have no float atomics.) */
int ops_amd_set_option(const char* name, long value);
long ops_amd_get_option(const char* name);

/* Name of the kernel symbol a given (B, Ne, tiling) call dispatches to -- lets profilers
 * and bench.py find the right row in a rocprofv3 kernel trace. */
const char* ops_beam_solve_kernel_name(int B, int Ne, int tiling);

/* Fused [x1 + x2 + x3] -> BatchNorm1d (train: batch statistics over the B rows, running statistics updated with the unbiased
 * variance, num_batches_tracked += 1; eval: running statistics) -> optional LeakyReLU(slope) -> optional dropout(p), one
 * launch per direction: the elementwise tails of the PINN's residual MLP (PINN_MultiCase.py:425-452, :519-541).
 * Activations [B,F] float32 or bfloat16 (act_is_bf16); gamma == NULL switches the normalisation off (activation + dropout
 * only).  x2 / x3 may be NULL.  Training saves z_save [B,F] (the summed input; may be NULL when x2 == x3 == NULL: x1 is it),
 * mean_save / rstd_save [F] and the keep mask [B,F] bytes.  call_counter: two uint64 of device memory (zero-initialised)
 * that the kernel advances itself, so that a replayed HIP graph draws fresh dropout masks.
 * Backward: dz [B,F] = gradient w.r.t. every addend; dgamma / dbeta [F] are ASSIGNED (point them at the parameters'
 * gradient slices: no accumulate kernels). */
int ops_fused_bn_act_fwd(int B, int F, const void* x1, const void* x2, const void* x3, int act_is_bf16,
                         const float* gamma, const float* beta, float eps, float momentum, int training,
                         float* running_mean, float* running_var, long long* num_batches_tracked,
                         float slope, int use_act, float p_drop, unsigned long long seed,
                         unsigned long long* call_counter, void* y, void* z_save, float* mean_save,
                         float* rstd_save, uint8_t* mask, void* stream);
int ops_fused_bn_act_bwd(int B, int F, const void* dy, int act_is_bf16, const void* z, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, float slope, int use_act, float p_drop, const uint8_t* mask,
                         void* dz, float* dgamma, float* dbeta, void* stream);

/* Batch assembly of the surrogate training loops in one launch: out[b, :] = X[idx[b], :] + (*sigma) * N(0, 1) for b < B, X
 * [N,F] float32 (F = product of the trailing dimensions), idx [B] int64, out float32 or bfloat16 -- DataLoader gather + input
 * noise + autocast cast (PINN_MultiCase.py:743-756).  sigma: device scalar (NULL or 0: no noise); counter: TWO uint64 of
 * zero-initialised device memory ([0] = calls so far, advanced once per call by the last workgroup to finish, [1] = its
 * tally; csrc/call_counter.hpp) (NULL: a fixed stream). */
int ops_gather_rows_noise_f32(int B, long F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                              unsigned long long* counter, void* out, int out_is_bf16, void* stream);
/* ... and the batch's targets in the same launch: Yout [B, C] = Y[idx] (float32, untouched; Y may be NULL). */
int ops_gather_rows_noise_targets_f32(int B, long F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                      unsigned long long* counter, void* out, int out_is_bf16, const float* Y, int C, float* Yout, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Layer blocks of the PINN's residual MLP (PINN_MultiCase.py:395-541) for batches of up to 128 rows: ONE launch per
 * Linear -> [+ Conv1d/BatchNorm1d(1) stencil + residual] -> BatchNorm1d -> LeakyReLU -> dropout, and one per backward
 * counterpart (bf16 MFMA, fp32 accumulation and statistics).  csrc/mlp_block.hip; host side: openpystruct_amd/pinn_fused.py.
 *
 * Layout contract.  Operands are stored FRAGMENT-TILED: a [rows, K] bfloat16 matrix (rows padded to 16, K to 32, KS = K / 32) is
 * a sequence of 1 KB tiles, tile (row >> 4, k >> 5) at tile index (row >> 4) * KS + (k >> 5), each in MFMA lane order:
 *     element (row, k)  ->  tile * 512 + ((k >> 3) & 3) * 128 + (row & 15) * 8 + (k & 7)          [elements]
 * so that one wave-wide 16-byte load is the A / B fragment of v_mfma_f32_16x16x32_bf16 (lane = (k >> 3 & 3) << 4 | row & 15).
 * Every activation / gradient matrix X [B, F] lives twice, zero outside the live B x F corner:
 *   X  : rows = batch row (128), K = ld columns, ld >= F rounded up to 32          ("[128, ld]" below)
 *   Xt : rows = column of X (F rounded up to 32), K = 128 batch rows               ("[F r.u. 32, 128]" below)
 * and every weight W [N, K] twice as well: Wp (rows = N r.u. 16, K r.u. 32) and Wtp (rows = K r.u. 16, N r.u. 32);
 * ops_mlp_repack_weights and ops_flat_clip_adam_step_repack_f32 write both from the float32 parameters.  All `ld*` arguments
 * are the padded K of the tiled matrix (multiples of 32).  openpystruct_amd/pinn_fused.py: to_tiled / from_tiled.
 * -----------------------------------------------------------------------------------------------------------------*/
#define OPS_MLP_MAX_ROWS 128
/* tails (`tail`): what follows the product in the same launch */
#define OPS_MLP_TAIL_NONE 0            /* Y = A W^T + bias                                            (output layer) */
#define OPS_MLP_TAIL_ACT_DROP 1        /* dropout(LeakyReLU(.))                                       (block fc1) */
#define OPS_MLP_TAIL_BN_ACT_DROP 2     /* dropout(LeakyReLU(BatchNorm1d(.)))                          (input layer) */
#define OPS_MLP_TAIL_BN 3              /* BatchNorm1d(.)                                              (block fc2 + norm) */
#define OPS_MLP_TAIL_BWD_ACT_DROP 4    /* gradient through TAIL_ACT_DROP of the layer below; dbias = column sums */
#define OPS_MLP_TAIL_BWD_BN 5          /* gradient through TAIL_BN of the layer below; dgamma, dbeta, dbias */
#define OPS_MLP_TAIL_BWD_BN_ACT_DROP 6 /* gradient through TAIL_BN_ACT_DROP of the layer below; dgamma, dbeta, dbias */
#define OPS_MLP_TAIL_LOSS 7            /* output layer + training loss: predictions -> P, Y / Yt = d loss / d predictions, dbias;
                                          the loss VALUE is completed by the next launch (loss_finish_rows) */
/* addends (`add_mode`) joined to the product before the tail */
#define OPS_MLP_ADD_NONE 0
#define OPS_MLP_ADD_FWD_BLOCK 1        /* + bn1(conv1(O)) + O: the ResidualBlock's stencil path and identity (O = block input) */
#define OPS_MLP_ADD_BWD_BLOCK 2        /* + dZ + conv1^T(bn1 backward(dZ)): the gradient the block input receives from them */
/* side jobs (`side`): partial sums for the whole-tensor BatchNorm1d(1), collected by extra workgroups of this launch (one per 8
 * columns of O) and consumed by the NEXT launch's ADD_* */
#define OPS_MLP_SIDE_NONE 0
#define OPS_MLP_SIDE_FWD_STENCIL_STATS 1
#define OPS_MLP_SIDE_BWD_STENCIL_SUMS 2

typedef struct ops_mlp_strip_args {
  int32_t B, N, K;                 /* live rows (1..128), output columns, reduction length */
  int32_t tail, add_mode, side;
  const void* A; int32_t lda;      /* [128, lda] bfloat16 */
  const void* W; int32_t ldw;      /* [N r.u. 16, ldw] bfloat16: forward Wp of the layer, backward Wtp of the layer above */
  const float* bias;               /* [N] float32 or NULL (forward tails; rounded to bfloat16 as autocast does) */
  void* Y; int32_t ldy; void* Yt;  /* result, both layouts (Yt may be NULL) */
  /* BatchNorm1d of the tail: forward writes mean / rstd / Zt (the pre-normalisation values, transposed), backward reads them */
  const float* gamma; const float* beta; float eps, momentum;
  float* running_mean; float* running_var; long long* num_batches_tracked;
  float* mean; float* rstd; void* Zt;
  const void* Yref_t;              /* backward of an activation/dropout tail: that layer's forward OUTPUT, transposed */
  float slope, p_drop;
  unsigned long long seed; unsigned long long* call_counter;   /* dropout stream (device counter advanced by the launch) */
  float* dgamma; float* dbeta; float* dbias;                    /* ASSIGNED */
  /* stencil path: Conv1d(1,1,3,padding=1) + BatchNorm1d(1) over the block input O [B, No] */
  const void* Ot; int32_t No;      /* transposed block input */
  const void* dZt;                 /* ADD_BWD_BLOCK / SIDE_BWD: gradient at the block's sum, transposed */
  const float* conv_w; const float* conv_b; const float* sgamma; const float* sbeta; float seps, smomentum;
  float* srunning_mean; float* srunning_var; long long* snum_batches_tracked;
  float* ssave;                    /* [2] mean, 1/std of the stencil normalisation (forward writes, backward reads) */
  double* spart;                   /* ops_mlp_spart_doubles(No) doubles: SIDE_* writes them, ADD_* of the next launch reads them */
  float* sdparams;                 /* [6] ASSIGNED by ADD_BWD_BLOCK: d conv_w[3], d conv_b, d gamma, d beta */
  /* TAIL_LOSS: the loss of ops_surrogate_loss_grad_f32 on the N = C columns of this product */
  void* P; int32_t ldp;            /* predictions [128, ldp] bfloat16 (may be NULL) */
  const float* targets_t;          /* [N, 128] float32, TRANSPOSED targets (ops_mlp_gather_noise writes them), rows >= B ignored */
  int32_t nI, nD;
  const float* alpha; float alpha0; const float* min_constraint; const float* max_constraint; float box_weight, rel_penalty;
  void* loss_ws;                   /* ops_mlp_loss_workspace_bytes() bytes: TAIL_LOSS leaves per-strip partial sums there ... */
  /* ... and the NEXT launch (any tail) adds them when loss_finish_rows = strips of the loss launch = (C + 15) / 16 (with the same
   * nI, nD, alpha, alpha0, box_weight, rel_penalty and loss_C = C): loss[0] = value, loss_sum[0] += value (optional running total) */
  int32_t loss_finish_rows, loss_C;
  float* loss; float* loss_sum;
  /* evaluation pass (model.eval()): forward BatchNorm tails and the stencil's BatchNorm1d(1) normalise with the RUNNING statistics
   * (read only: nothing is updated or saved); pass p_drop = 0 and side = NONE with it */
  int32_t eval_stats;
  /* evaluation slots, C-ABI version 9 (n_slots = 0: none).  The launch runs n_slots independent batches side by side (grid.y):
   * slot s reads and writes every per-batch buffer -- A, Y, Yt, Zt, Ot, P, targets_t, loss_ws, loss, loss_sum -- at the given
   * pointer + s * slot_stride BYTES (one arena per slot with the same layout; weights, biases, statistics are shared), and holds
   * min(B, slot_total_rows - s * B) rows.  Evaluation / inference launches only (eval_stats, or a tail without statistics): a
   * validation pass of 14 batches is 7 launches instead of 14 x 8 (profiles/r03_notes.md section 9). */
  int32_t n_slots;
  int32_t slot_total_rows;
  int64_t slot_stride;
} ops_mlp_strip_args;

/* One strip launch: workgroup = 128 rows x 16 output columns.  Returns OPS_AMD_ERR_INVALID_ARG on a broken layout contract. */
int ops_mlp_strip_launch(const ops_mlp_strip_args* args, void* stream);
/* doubles of `spart` for a block input of No columns (No <= 512) */
size_t ops_mlp_spart_doubles(int No);

/* All weight gradients of a step in one launch: out_i [N_i, K_i] float32 (ASSIGNED, row stride ldo_i) = At_i [N_i r.u. 32, 128] x
 * Bt_i [K_i r.u. 32, 128]^T, the transposed layouts of the layer's output gradient and of its input. */
typedef struct ops_mlp_wgrad_problem {
  const void* At; const void* Bt; float* out; int32_t N, K, ldo;
} ops_mlp_wgrad_problem;
#define OPS_MLP_MAX_WGRAD 8
#define OPS_MLP_MAX_REPACK 16   /* matrices per ops_flat_clip_adam_step_repack_f32 (ops_mlp_repack_weights: OPS_MLP_MAX_WGRAD per call) */
int ops_mlp_wgrad_group(int nprob, const ops_mlp_wgrad_problem* problems, void* stream);
/* r05: the same launch also leaves what ops_flat_clip_adam_step_*_f32 needs of the gradient norm in the optimiser's `workspace`
 * (ops_flat_adam_workspace_bytes()): every tile's workgroup the sum of the squares it stores, `nranges` extra one-wave workgroups the
 * sums over the float32 ranges (range_ptr[i], range_len[i]) -- the gradients the matrices do not cover (biases, normalisation
 * parameters: complete when this launch starts) -- each times grad_scale^2; workgroup 0 advances `step` and tabulates its bias
 * corrections.  Returns the number of partial sums through *nparts (pass it on as OPS_ADAM_NORM_READY | nparts << 16). */
#define OPS_MLP_MAX_NORM_RANGES 32
int ops_mlp_wgrad_group_norm(int nprob, const ops_mlp_wgrad_problem* problems, int nranges, const float* const* range_ptr,
                             const int32_t* range_len, float grad_scale, void* workspace, int32_t* step, float beta1, float beta2,
                             int32_t* nparts, void* stream);

/* Wp / Wtp of up to 8 weight matrices from the float32 parameters W_i [N_i, K_i] in one launch. */
typedef struct ops_mlp_repack_entry {
  const float* W; int32_t N, K; void* Wp; int32_t ldw; void* Wtp; int32_t ldwt;
} ops_mlp_repack_entry;
int ops_mlp_repack_weights(int nmat, const ops_mlp_repack_entry* entries, void* stream);
/* ops_flat_clip_adam_step_f32 that also refreshes those copies (every entry's W must point into `params`; up to OPS_MLP_MAX_REPACK
 * matrices): one coalesced launch behind the update, one wave per 1 KB tile. */
int ops_flat_clip_adam_step_repack_f32(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* lr,
                                       int32_t* step, float max_norm, float grad_scale, float beta1, float beta2, float eps,
                                       float weight_decay, int decoupled_weight_decay, void* params_bf16, void* workspace,
                                       int nmat, const ops_mlp_repack_entry* entries, void* stream);

/* ops_gather_rows_noise_f32 writing the layout above: out (rows = batch, K = ld) and out_t (rows = F r.u. 32, K = 128) bfloat16,
 * rows >= B zeroed; and, when Y [n, C] float32 is given, the batch's targets TRANSPOSED: targets_t [C, 128] = Y[idx]^T. */
int ops_mlp_gather_noise(int B, int F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                         unsigned long long* counter, void* out, int ld, void* out_t, const float* Y, int C, float* targets_t,
                         void* stream);
/* r05 (ABI 12): the same launch also rebuilds the tiled bf16 weight copies of `nmat` matrices from the float32 parameters (entries as in
 * ops_flat_clip_adam_step_repack_f32: every W inside [params, params + n_params)) in extra workgroups -- the previous optimiser step's
 * copies and this step's batch depend on nothing of each other, so the two adjacent launches are one.  The optimiser is then called
 * WITHOUT its repack (ops_flat_clip_adam_step_f32); whoever reads the copies outside a training step (an evaluation pass) calls
 * ops_mlp_repack_weights first. */
int ops_mlp_gather_noise_repack(int B, int F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                unsigned long long* counter, void* out, int ld, void* out_t, const float* Y, int C, float* targets_t,
                                long n_params, const float* params, int nmat, const ops_mlp_repack_entry* entries, void* stream);

/* bytes of ops_mlp_strip_args.loss_ws (TAIL_LOSS) */
size_t ops_mlp_loss_workspace_bytes(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Row-wise blocks of the Transformer-Diffusion surrogate's encoder layers (TransformerDiffusionModule_MultiCase.py:539-575,
 * nn.TransformerEncoderLayer: post-norm, ReLU, dropout) for the training step, one launch per direction each (csrc/seq_block.hip).
 * Dropout: counter-based hash of (seed, *counter, element); `counter` (one uint64 of device memory) is only READ -- the caller
 * advances it between steps -- and the value a forward launch used is left in `used_call` (one uint64), from which the backward
 * launch regenerates the mask.  All activations bfloat16 unless said.
 * -----------------------------------------------------------------------------------------------------------------*/
/* softmax(q k^T / sqrt(dh)) -> dropout(p) -> @ v per sample and head, for sequences of S <= 8 tokens: qkv [Bn * S, 3 * H * dh]
 * (the in-projection's output: q | k | v per row), ctx [Bn * S, H * dh]; backward: dqkv from dctx. */
int ops_seq_attention_fwd(int Bn, int S, int H, int dh, const void* qkv, void* ctx, float p_drop, unsigned long long seed,
                          unsigned long long* counter, unsigned long long* used_call, void* stream);
int ops_seq_attention_bwd(int Bn, int S, int H, int dh, const void* qkv, const void* dctx, void* dqkv, float p_drop,
                          unsigned long long seed, unsigned long long* used_call, void* stream);
/* y = LayerNorm(res + dropout(x)) over the d <= 256 columns of T rows: x bfloat16, res float32 or bfloat16, y in BOTH float32
 * (the residual stream) and bfloat16 (the next product's operand); saves z = res + dropout(x) [T, d] float32, mean / rstd [T].
 * Backward: dy32 and / or dy16 (summed) -> dx (bfloat16, through the dropout) and dres (float32); dgamma / dbeta [d] are ADDED to
 * with float atomics -- the caller zeroes them (the training loop's flat gradient buffer is zeroed every step). */
int ops_dropout_add_layernorm_fwd(int T, int d, const void* x, const void* res, int res_is_bf16, const float* gamma, const float* beta,
                                  float eps, float p_drop, unsigned long long seed, unsigned long long* counter,
                                  unsigned long long* used_call, float* y32, void* y16, float* z, float* mean, float* rstd, void* stream);
int ops_dropout_add_layernorm_bwd(int T, int d, const float* dy32, const void* dy16, const float* z, const float* mean, const float* rstd,
                                  const float* gamma, float p_drop, unsigned long long seed, unsigned long long* used_call, void* dx,
                                  float* dres, float* dgamma, float* dbeta, void* stream);
/* y = dropout(LeakyReLU_slope(x)) on n bfloat16 elements (slope 0: ReLU); backward from x and dy. */
int ops_act_dropout_fwd(long n, const void* x, void* y, float slope, float p_drop, unsigned long long seed, unsigned long long* counter,
                        unsigned long long* used_call, void* stream);
int ops_act_dropout_bwd(long n, const void* x, const void* dy, void* dx, float slope, float p_drop, unsigned long long seed,
                        unsigned long long* used_call, void* stream);

/* Weight AND bias gradient of a Linear over many rows: dW [N, K] float32 += dY^T X, dbias [N] += column sums of dY (optional); dY
 * [T, N] and X [T, K] bfloat16 row-major.  The rows are split over the grid and the partial results added with float atomics
 * (dW / dbias: the caller's zeroed flat gradient slices). */
int ops_linear_wgrad_accumulate(int T, int N, int K, const void* dY, const void* X, float* dW, float* dbias, void* stream);
/* ... and up to 16 such products in ONE launch (every weight gradient of a backward pass, once all output gradients exist). */
typedef struct ops_wgrad_problem {
  int32_t T, N, K;
  const void* dY; const void* X; float* dW; float* dbias;
  int32_t ldy, ldx;                                   /* row strides of dY / X in elements; 0: N / K (contiguous rows) */
} ops_wgrad_problem;
/* K = 0: a COLUMN-SUM job instead of a product: dW [N] += column sums of the FLOAT32 matrix dY [T, N] (row stride ldy; X, dbias unused) --
 * per-workgroup partial sums of another launch, reduced without a launch of their own. */
#define OPS_WGRAD_MAX_GROUP 24
int ops_linear_wgrad_accumulate_group(int nprob, const ops_wgrad_problem* problems, void* stream);

/* Diffusion front end of the Transformer-Diffusion surrogate (TransformerDiffusionModule_MultiCase.py:443-478, :563-567) around its
 * two-layer MLP: noise: x_noisy = sqrt(acp[t]) x + sqrt(1 - acp[t]) eps for `rows` = B * Nc rows of d features (t int64 [rows] and eps
 * [rows, d] drawn by the caller), outputs float32 AND bfloat16 plus sa = sqrt(acp[t]), sb = sqrt(1 - acp[t]) [rows];
 * combine: z [B, 1 + Nc, d] float32 = [cls | (x_noisy - sb mlp) / sa] + pe[:1 + Nc] with mlp [rows, d] bfloat16; its backward: dm
 * (bfloat16) from g [B, 1 + Nc, d] and dcls [d] += column sums of g[:, 0, :] (float atomics; may be NULL). */
int ops_diffusion_noise(long rows, int d, const float* x, const long long* t, const float* eps, const float* alpha_cumprod, float* xn32,
                        void* xn16, float* sa, float* sb, void* stream);
/* ops_diffusion_noise with t (uniform in [0, T)) and eps (standard normal) drawn inside the launch from the counter-based stream of
 * (seed, *counter) -- `counter` is only read; t_out [rows] / eps_out [rows, d]: optional copies of the draws. */
int ops_diffusion_noise_draw(long rows, int d, int T, const float* x, const float* alpha_cumprod, unsigned long long seed,
                             const unsigned long long* counter, float* xn32, void* xn16, float* sa, float* sb, long long* t_out, float* eps_out,
                             void* stream);
int ops_diffusion_combine_fwd(int B, int Nc, int d, const void* m, const float* xn32, const float* sa, const float* sb, const float* cls,
                              const float* pe, float* z, void* z16 /* optional bfloat16 copy of z */, void* stream);
int ops_diffusion_combine_bwd(int B, int Nc, int d, const float* g /* float32, may be NULL */, const void* g16 /* bfloat16, may be NULL: the
                              gradient is g + g16 */, const float* sa, const float* sb, void* dm, float* dcls, void* stream);

/* One post-norm encoder layer of the Transformer-Diffusion surrogate (TransformerDiffusionModule_MultiCase.py:539-575) for the training
 * step as ONE launch per direction (csrc/seq_layer.hip).  Forward: in-projection, attention over S <= 8 tokens, out-projection,
 * dropout + add + LayerNorm, feed-forward with ReLU + dropout, dropout + add + LayerNorm.  T = Bn * S rows; d = H * dh <= 128, dh <= 16,
 * ff <= 256, d and ff multiples of 8.
 * Weights: bfloat16 copies in the FRAGMENT-TILED layout of ops_mlp_repack_weights (ops_flat_clip_adam_step_repack_f32 refreshes them
 * inside the optimiser launch): for W [N, K], `Wp` has ldw = K rounded up to 32 and `Wtp` (the transpose's tiles) ldwt = N rounded up
 * to 32, zero padded, 16-byte aligned.  The forward pass reads the four Wp, the backward pass the four Wtp; biases: bfloat16 vectors.
 * Everything the backward launch and the weight-gradient products read is saved by the forward launch; dropout masks: the stream
 * of csrc/dropout_stream.hpp, one seed per site, `counter` only read, `used_call` = the value used (the backward launch reads it). */
typedef struct ops_tfd_layer_args {
  int32_t Bn, S, H, dh, d, ff;
  const float* x32;                                   /* [T, d] layer input (residual stream) */
  const void* W_in; const void* b_in;                 /* Wp of [3 d, d], [3 d] */
  const void* W_out; const void* b_out;               /* Wp of [d, d], [d] */
  const void* W_1; const void* b_1;                   /* Wp of [ff, d], [ff] */
  const void* W_2; const void* b_2;                   /* Wp of [d, ff], [d] */
  const float* gamma1; const float* beta1; float eps1;
  const float* gamma2; const float* beta2; float eps2;
  float p_attn, p_1, p_act, p_2;
  unsigned long long seed_attn, seed_1, seed_act, seed_2;
  const unsigned long long* counter; unsigned long long* used_call;
  void* qkv;                                          /* [T, 3 d] bf16 */
  void* ctx;                                          /* [T, d] bf16: attention output */
  float* z1; float* mean1; float* rstd1; void* y1_16; /* LayerNorm1: input [T, d] f32, statistics [T], output bf16 */
  void* u; void* h;                                   /* [T, ff] bf16: feed-forward pre-activation, and ReLU + dropout of it */
  float* z2; float* mean2; float* rstd2;
  float* y32; void* y16;                              /* [T, d] layer output, both precisions */
  unsigned long long* trace;                          /* diagnostics: NULL, or 16 stage stamps per workgroup (100 MHz clock) */
  int32_t identity_act;                               /* r04, verification only: 1 = the feed-forward activation is the identity instead of ReLU (a smooth
                                                         network whose gradients can be compared with float64 to bf16 rounding); 0 = the reference */
} ops_tfd_layer_args;
int ops_tfd_encoder_layer_fwd(const ops_tfd_layer_args* args, void* stream);
/* Two consecutive layers (b->x32 == a->y32, same Bn / S / d) as ONE launch: a workgroup owns whole samples, so it runs layer b on the rows it
 * has just written as layer a's output.  Same results as the two launches. */
int ops_tfd_encoder_layer_pair_fwd(const ops_tfd_layer_args* a, const ops_tfd_layer_args* b, void* stream);

/* The layer's BACKWARD pass as one launch: LayerNorm2 backward, d_h = d_f W_2, ReLU + dropout backward, d_y1 = d_u W_1 (+ the residual
 * branch), LayerNorm1 backward, d_ctx = d_a W_out, attention backward, dx = residual branch + dqkv W_in.  The incoming gradient is
 * g32 + g16 (either may be NULL, not both).  Written for the weight-gradient products (ops_linear_wgrad_accumulate_group): d_f, d_u,
 * d_a, dqkv (bfloat16, gradients at the four products' outputs).  dgamma / dbeta: ADDED to (float atomics; the caller zeroes them). */
typedef struct ops_tfd_layer_bwd_args {
  int32_t Bn, S, H, dh, d, ff;
  const float* g32; const void* g16;                  /* [T, d] gradient at the layer output */
  const void* Wt_in; const void* Wt_out; const void* Wt_1; const void* Wt_2;   /* Wtp of the four weights */
  const float* gamma1; const float* gamma2;
  float p_attn, p_1, p_act, p_2;
  unsigned long long seed_attn, seed_1, seed_act, seed_2;
  const unsigned long long* used_call;                /* the forward launch's */
  const void* qkv; const float* z1; const float* mean1; const float* rstd1; const void* u; const float* z2; const float* mean2; const float* rstd2;
  void* d_f; void* d_u; void* d_a; void* dqkv;        /* [T, d], [T, ff], [T, d], [T, 3 d] bf16 */
  float* dx32;                                        /* [T, d] gradient at the layer input (residual + in-projection branch) */
  float* dgamma1; float* dbeta1; float* dgamma2; float* dbeta2;
  unsigned long long* trace;
  float* ln_part;                                     /* NULL, or [workgroups = ceil(Bn / (16 / S))][4][128] float32: the launch stores every
                                                         workgroup's column sums (dgamma2 | dbeta2 | dgamma1 | dbeta1) there INSTEAD of adding
                                                         them with atomics (224 workgroups on the same 480 addresses stalled the launch's memory
                                                         pipeline for ~9 us); the caller sums the rows (ops_linear_wgrad_accumulate_group, K = 0) */
  int32_t identity_act;                               /* as ops_tfd_layer_args */
} ops_tfd_layer_bwd_args;
int ops_tfd_encoder_layer_bwd(const ops_tfd_layer_bwd_args* args, void* stream);
/* The backward passes of two consecutive layers as ONE launch: `later` first, then `earlier` on the float32 dx the workgroup has just written
 * (earlier->g32 == later->dx32, earlier->g16 == NULL).  Same results as the two launches. */
int ops_tfd_encoder_layer_pair_bwd(const ops_tfd_layer_bwd_args* later, const ops_tfd_layer_bwd_args* earlier, void* stream);

/* The Transformer-Diffusion model's head (TransformerDiffusionModule_MultiCase.py:568-575) as one launch per direction: rows = the
 * [CLS] rows of the last encoder layer's bf16 output (row b S of y16 [B S, d]), a = fc1 rows + b1 (bf16), LayerNorm over `hid` columns,
 * ReLU, dropout, out = fc2 + b2 (bf16 [B, C]).  d <= 128, hid <= 256 (multiples of 8), C <= 128 (a multiple of 4); weights: fragment-tiled
 * copies as for the layer launches (forward Wp, backward Wtp).  Saved for the backward launch and the weight-gradient products: a16,
 * mean, rstd, h.  Backward: g [B, C] bf16 = d loss / d out -> d_a [B, hid] bf16 (gradient at fc1's output), dcls = d_a W_1 written into the
 * [CLS] rows (row stride S d elements) of `dcls_rows` (a [B S, d] bf16 tensor whose other rows the caller keeps zero), dgamma / dbeta ADDED. */
typedef struct ops_tfd_head_args {
  int32_t B, S, d, hid, C;
  const void* y16;
  const void* W1; const void* b1; const float* gamma; const float* beta; float eps; const void* W2; const void* b2;
  float p_drop; unsigned long long seed; const unsigned long long* counter; unsigned long long* used_call;
  void* a16; float* mean; float* rstd; void* h; void* out;
  /* r04 (ABI 10): the training loss ON the output tile (TrainableL1L2Loss, TFD:581-633: what ops_surrogate_loss_grad_sum_f32 computes for
   * nI = C -- two launches of the step less).  targets != NULL: grad [B, C] bf16 = d loss / d out, loss_part [5 ceil(B / 16)] doubles =
   * this launch's per-workgroup partial sums (|d|, d^2, box penalty, 0, 0); the NEXT launch of the step (ops_tfd_head_bwd) adds them
   * up.  alpha: device scalar (clamped to [1e-6, 1] as the loss does); min_constraint / max_constraint: device scalars or NULL. */
  const float* targets; void* grad; double* loss_part; const float* alpha; const float* min_constraint; const float* max_constraint;
  float box_weight;
  int32_t identity_act;                               /* as ops_tfd_layer_args (the head's ReLU) */
  const long long* target_rows;                       /* NULL: row b of `targets`; else row target_rows[b] (the front end's idx_out) */
} ops_tfd_head_args;
int ops_tfd_head_fwd(const ops_tfd_head_args* args, void* stream);
typedef struct ops_tfd_head_bwd_args {
  int32_t B, S, d, hid, C;
  const void* g; const void* Wt2; const void* Wt1; const float* gamma;
  float p_drop;
  const void* a16; const float* mean; const float* rstd; const void* h;
  void* d_a; void* dcls_rows; float* dgamma; float* dbeta;
  /* r04 (ABI 10): loss_part != NULL: workgroup 0 finishes the loss of the forward launch: loss[0] = the value (+ (alpha0 - alpha)^2 unless
   * alpha0 is NaN), loss_sum[0] += it (may be NULL). */
  const double* loss_part; const float* alpha; float alpha0; float box_weight; float* loss; float* loss_sum;
  /* g2 != NULL: the gradient is g + g2 (both bfloat16 [B, C], the sum rounded to bfloat16 as the framework's addition rounds it): the
   * launch's own loss gradient plus what another term (the FE-residual one) put on the predictions; g_sum (may be NULL, may be g)
   * receives the sum -- the weight-gradient product of the output layer reads it. */
  const void* g2; void* g_sum;
} ops_tfd_head_bwd_args;
int ops_tfd_head_bwd(const ops_tfd_head_bwd_args* args, void* stream);

/* The diffusion front end of the Transformer-Diffusion model (TransformerDiffusionModule_MultiCase.py:443-478, :563-567) as one launch
 * per direction: step indices and noise drawn in the launch (ops_diffusion_noise_draw's stream), x_noisy, the two-layer MLP
 * m = W_2 relu(W_0 x_noisy + b_0) + b_2 on bf16 MFMA (fragment-tiled weights: Wp forward, Wtp of W_2 backward), and the combine
 * z[b, 0] = cls + pe[0], z[b, 1 + n] = (x_noisy - sb m) / sa + pe[1 + n] (float32 AND bfloat16).  rows = B Nc, d <= 128, hid <= 256
 * (multiples of 8).  Saved for the backward launch / the weight-gradient products: xn16, h, sa, sb.  Backward: g32 + g16 [B, 1 + Nc, d]
 * (either may be NULL) -> dm [rows, d] = -(sb / sa) g[:, 1:, :] and d_h [rows, hid] = relu'(h) (dm W_2), both bfloat16; dcls [d] +=
 * column sums of g[:, 0, :] (float atomics; may be NULL). */
typedef struct ops_tfd_front_args {
  int32_t B, Nc, d, hid, T;
  const float* x; const float* alpha_cumprod;
  unsigned long long seed; const unsigned long long* counter;
  const void* W0; const void* b0; const void* W2; const void* b2;
  const float* cls; const float* pe;
  void* xn16; void* h; float* sa; float* sb; float* z; void* z16;
  long long* t_out; float* eps_out;
  int32_t identity_act;                               /* as ops_tfd_layer_args (the diffusion MLP's ReLU) */
  /* r04 (ABI 10): the launch assembles its own batch (what ops_gather_rows_noise_targets_f32 did in a launch of its own per step).
   * order != NULL: sample b is row order[*cursor + b] of `src` [.., Nc d] float32, plus sigma[0] * N(0, 1) from the assembly's stream
   * (in_seed, the counter's value c at entry; csrc/input_noise.hpp) -- `x` is not read; the diffusion draws use c + 1 (as if the
   * assembly launch had run and advanced the counter); idx_out[b] = that row (the head's loss reads its targets by it); the last
   * workgroup advances counter[0] (layout [calls, tally]: csrc/call_counter.hpp) and *cursor += B. */
  const float* src; const long long* order; long long* cursor; long long* idx_out; const float* sigma; unsigned long long in_seed;
  long long n_order;                                  /* r05 (ABI 11): entries of `order`; positions cursor + b >= n_order wrap modulo n_order (0: unchecked) */
} ops_tfd_front_args;
int ops_tfd_front_fwd(const ops_tfd_front_args* args, void* stream);
typedef struct ops_tfd_front_bwd_args {
  int32_t B, Nc, d, hid;
  const float* g32; const void* g16; const float* sa; const float* sb; const void* h; const void* Wt2;
  void* dm; void* d_h; float* dcls;
} ops_tfd_front_bwd_args;
int ops_tfd_front_bwd(const ops_tfd_front_bwd_args* args, void* stream);

/* Measurement aid of bench.py, not a product call: device-to-device copy of `bytes` (a multiple of 16, both pointers 16-byte
 * aligned) with one 16-byte access per lane and instruction -- the achievable HBM rate the roofline records quote next to the
 * nominal 8 TB/s.  non_temporal != 0: nt stores. */
int ops_hbm_copy16(const void* src, void* dst, size_t bytes, int non_temporal, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENPYSTRUCT_AMD_H */
