/*
 * climsim_hip.h - C ABI of the MI355X (gfx950) engine for ClimSim's baseline column emulators.
 *
 * The reference (leap-stc/ClimSim) has no FFI: its hot path is Keras `Model.fit` / `Model.predict`
 * on a functional MLP.  Each entry point below names the reference interface it stands in for, so a
 * maintainer can bind it from Python with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 (CS_OK) or a negative cs_status; nothing throws or aborts across the
 *     ABI; cs_last_error() gives a thread-local message for the last failure.
 *   - `*_dev` pointers are HIP device pointers owned by the CALLER (e.g. tensor.data_ptr()); the
 *     library owns weights, optimiser state and workspace (sized by cfg.max_batch at create time).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is enqueued
 *     asynchronously on it, nothing synchronises the device except get_weights/get_opt_state.
 *   - a handle is not thread-safe; handles on distinct devices are independent.
 *   - feature layout: x rows are 124 float32 (v1 inputs: state_t[60], state_q0001[60], state_ps,
 *     pbuf_SOLIN, pbuf_LHFLX, pbuf_SHFLX), y rows 128 float32 (ptend_t[60], ptend_q0001[60], 8
 *     scalars), both C-contiguous - the `.npy` layout of climsim_utils/data_utils.py:906-925.
 */
#ifndef CLIMSIM_HIP_H
#define CLIMSIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cs_mlp cs_mlp_t;

typedef enum { CS_OK = 0, CS_ERR_INVALID = -1, CS_ERR_HIP = -2, CS_ERR_NOMEM = -3, CS_ERR_STATE = -4 } cs_status;

/* keras.layers.ReLU / ELU / LeakyReLU(alpha) - step2_retrain.py:104-110 */
typedef enum { CS_ACT_RELU = 0, CS_ACT_ELU = 1, CS_ACT_LEAKYRELU = 2 } cs_act;
/* keras.optimizers.Adam / tfa RectifiedAdam / keras RMSprop / SGD - step2_retrain.py:150-157 */
typedef enum { CS_OPT_ADAM = 0, CS_OPT_RADAM = 1, CS_OPT_RMSPROP = 2, CS_OPT_SGD = 3,
               /* torch.optim.Adam(lr) - online_testing/baseline_models/MLP_v2rh/training/train_mlp_h5loader.py:210-211
                * (bias corrections as Python doubles, eps added to sqrt(v)/sqrt(1-b2^t); default eps 1e-8) */
               CS_OPT_ADAM_TORCH = 4 } cs_opt;
/* compile(loss='mse') step2_retrain.py:160-162 | nn.MSELoss / nn.L1Loss / nn.SmoothL1Loss
 * (train_mlp_h5loader.py:226-236); every one is a mean over batch x outputs */
typedef enum { CS_LOSS_MSE = 0, CS_LOSS_MAE = 1, CS_LOSS_HUBER = 2 } cs_loss;

#define CS_MAX_HIDDEN 16

/* Hyper-parameters of build_model() - step2_retrain.py:79-126 (hpo_baseline_v1.py:64-103). */
typedef struct cs_mlp_cfg {
    int32_t n_in;                 /* input_length = 124                                         */
    int32_t n_hidden;             /* hp num_layers (2..12 in the HPO space)                      */
    int32_t hidden[CS_MAX_HIDDEN];/* hp units_k, multiples of 128 (128..1024 in the HPO space)   */
    int32_t n_out_lin;            /* output_length_lin  = 120 (linear head)                      */
    int32_t n_out_relu;           /* output_length_relu = 8   (relu head); both multiples of 4, lin+relu <= 1024;
                                     128 (v1) runs on the tuned layer-chain kernels, other widths (368 = v2) on the wide chain */
    int32_t act;                  /* cs_act                                                       */
    float   alpha;                /* LeakyReLU slope (0.15)                                       */
    int32_t optimizer;            /* cs_opt                                                       */
    int32_t max_batch;            /* largest n accepted by forward / loss_grads                   */
    int32_t device;               /* HIP device ordinal                                           */
    int32_t flags;                /* CS_FLAG_*                                                    */
    /* Optimiser hyper-parameters as the Python floats (doubles) Keras receives: 0.9, 0.999, 1e-7,
     * rho 0.9.  They are cast to float32 exactly where TensorFlow casts them (see k_optimizer). */
    double  beta1, beta2, eps, rho;
} cs_mlp_cfg;

#define CS_FLAG_NO_TR_READ 1      /* wgrad operands by 16-bit LDS gathers instead of ds_read_b64_tr_b16 */
#define CS_FLAG_NO_CHAIN   2      /* one GEMM launch per layer instead of the fused layer-chain kernels    */
#define CS_FLAG_CHAIN_BM64  4     /* force 64-row chain tiles  (default: by batch size)                    */
#define CS_FLAG_CHAIN_BM128 8     /* force 128-row chain tiles                                             */
#define CS_FLAG_CHAIN_BM32  16    /* force 32-row chain tiles                                              */
#define CS_FLAG_GEMM_V1 64         /* per-layer path on the first (register-staged) GEMM kernels: A/B and parity runs */
#define CS_FLAG_DIRECT_HEAD 128   /* online_testing MLP (MLP_v2rh/training/mlp.py:41-67): no Dense(output_length)+act between the
                                     hidden stack and the heads - the output layer [n_out_lin linear || n_out_relu relu] sits on
                                     the last hidden layer (torch `final_linear` + relu on the last 8 columns)              */
#define CS_FLAG_COOP 512           /* training steps of up to 4096 columns on the COOPERATIVE chain (csrc/coop.h): a 32-row tile is split over
                                    * 8 / 4 / 2 workgroups that exchange layer outputs inside the launch (1.4x at 1024 columns).  The
                                    * workgroups of such a launch wait for one another, so it needs every one of them resident: the caller
                                    * guarantees that NO OTHER cooperative launch (another stream, another process) runs on the device
                                    * at the same time - and no other kernel that occupies compute units (a side-stream loader).  A wait
                                    * that runs out is COUNTED in host-mapped memory and the kernel runs on (never a hang): every later
                                    * cs_mlp_* compute call on the handle fails with CS_ERR_STATE as soon as the host sees the count,
                                    * cs_mlp_check() forces the test behind a stream synchronisation.  The error is sticky. */
#define CS_FLAG_NO_CHAIN_FB 256    /* forward and backward chain as two launches (default: one launch on 32-row tiles)  */
#define CS_FLAG_CHAIN_BWD32_ON_FWD64 32   /* tests: 64-row forward tiles (with CHAIN_BM64), 32-row backward tiles    */

/* keras.Model(...) + compile(): allocates weights (zero), optimiser state, workspace. */
int  cs_mlp_create(cs_mlp_t** out, const cs_mlp_cfg* cfg);
void cs_mlp_destroy(cs_mlp_t* h);
/* model.count_params() */
int64_t cs_mlp_num_params(const cs_mlp_t* h);

/* data_utils.save_norm() vectors (data_utils.py:954-988): x_hat = (x - sub)/div, inf/nan -> 0
 * (data_utils.py:807-809, :894-897).  Host pointers, copied.  Only used when a call passes
 * normalise != 0; the .npy splits of save_as_npy are already normalised. */
int cs_mlp_set_norm(cs_mlp_t* h, const float* input_sub, const float* input_div);

/* Loss the gradients are taken of (default CS_LOSS_MSE) and output pruning: keep_host[n_out] (n = n_out) holds 1 for
 * the output columns the model produces and 0 for the ones it forces to zero - `x[:, 60:60+strato_lev_out] = 0` ... in
 * MLP.forward (online_testing/baseline_models/MLP_v2rh/training/mlp.py:56-61); pruned columns stay in the loss mean
 * with prediction 0 and pass no gradient.  keep_host == NULL: no pruning.  With CS_LOSS_HUBER the first loss sum
 * (loss_sums[0]) is the sum of SmoothL1 terms instead of squared errors; loss_sums[1] is always the sum of |e|. */
int cs_mlp_set_head_options(cs_mlp_t* h, int loss_kind, const float* keep_host, int64_t n);

/* nn.Dropout(p) behind every hidden Linear while TRAINING (online_testing/baseline_models/MLP_v2rh/training/mlp.py:39-52:
 * Sequential(Linear, Dropout) then relu; the shipped configuration uses p = 0).  Kept activations are scaled by 1/(1-p),
 * prediction and evaluation run in eval mode.  torch's Philox stream cannot be reproduced by anyone else: the mask is a
 * counter hash of (seed, optimiser step, layer, row, column) that the oracle shares.  ReLU stacks on the wide layer-chain
 * kernels only (a model on the tuned chain is moved there). */
int cs_mlp_set_dropout(cs_mlp_t* h, double rate, uint64_t seed);

/* model.set_weights / model.get_weights: one flat float32 host buffer in Keras order
 * [W0(in,out), b0, ..., W_up(.,128), b_up, W_lin(128,120), b_lin, W_relu(128,8), b_relu].
 * set_weights also refreshes the bf16 operand copies.  get_* synchronise `stream`. */
int cs_mlp_set_weights(cs_mlp_t* h, const float* host, int64_t n, void* stream);
int cs_mlp_get_weights(cs_mlp_t* h, float* host, int64_t n, void* stream);
/* optimizer.get_weights(): first/second moments (Keras order, like the weights) and iteration. */
int cs_mlp_get_opt_state(cs_mlp_t* h, float* host_m, float* host_v, int64_t n, int64_t* iterations, void* stream);
int cs_mlp_set_opt_state(cs_mlp_t* h, const float* host_m, const float* host_v, int64_t n, int64_t iterations, void* stream);

/* model.predict / model.evaluate (step3_inference.ipynb cell 2; validation pass of model.fit):
 *   x_dev      (rows, 124) float32; row i of the batch is x_dev[row_idx ? row_idx[i] : i]
 *   yhat_dev   (n, 128) float32 output in scaled space, may be NULL
 *   y_dev      targets (indexed like x_dev) or NULL; when given, loss_dev[0] += sum (yhat-y)^2,
 *              loss_dev[1] += sum |yhat-y| over n*128 elements (loss_dev is zeroed first unless
 *              accumulate != 0).  mse = loss_dev[0]/(128 n), mae = loss_dev[1]/(128 n). */
/* Rows ONE cs_mlp_forward call may take: max_batch sizes the training buffers; prediction / evaluation on the layer-chain paths keep
 * no per-row state of their own, so they take up to 2^22 rows per call (128-row tiles fill the chip from 32768 rows: 390 M columns/s
 * against 200 M in calls of 8192).  max_batch for models on the per-layer path. */
int64_t cs_mlp_forward_limit(const cs_mlp_t* h);
int cs_mlp_forward(cs_mlp_t* h, const float* x_dev, const int64_t* row_idx_dev, int64_t n, int normalise,
                   float* yhat_dev, const float* y_dev, float* loss_dev, int accumulate, void* stream);

/* Forward + backward of one batch (the autodiff half of Model.train_step): fills the flat float32
 * gradient buffer with d(sum of squared errors)/d(param) = UNSCALED sums; the 1/(n_out n) of the
 * 'mse' mean (and 1/world for data parallel) is applied by cs_mlp_apply's grad_scale.
 * accumulate != 0 adds to the existing gradient/loss (micro-batching). */
int cs_mlp_loss_grads(cs_mlp_t* h, const float* x_dev, const float* y_dev, const int64_t* row_idx_dev,
                      int64_t n, int normalise, float* loss_dev, int accumulate, void* stream);

/* Flat gradient buffer (internal parameter order: Keras order with the heads fused and the two output layers padded
 * to a multiple of 128 columns; *n_floats >= num_params) for an external all-reduce (RCCL via torch.distributed on a
 * tensor aliasing it), or rebind it to caller memory of that many floats.  cs_mlp_get_grads copies it out in Keras
 * order (n = num_params) for inspection. */
int cs_mlp_grad_buffer(cs_mlp_t* h, void** dev_ptr, int64_t* n_floats);
int cs_mlp_set_grad_buffer(cs_mlp_t* h, void* dev_ptr);
int cs_mlp_get_grads(cs_mlp_t* h, float* host, int64_t n, void* stream);

/* optimizer.apply_gradients: g = grad * grad_scale; Keras-2.11 Adam / RMSprop / SGD or tfa-0.19
 * RectifiedAdam update with learning rate `lr`; increments optimizer.iterations; re-casts bf16
 * operand copies of the weights. */
int cs_mlp_apply(cs_mlp_t* h, float lr, float grad_scale, void* stream);

/* Model.train_step: loss_grads + apply(lr, 1/(128 n)).  What the gradient buffer holds AFTER a train_step is unspecified (the step may
 * keep its row splits' partial sums in separate buffers and leave the buffer un-zeroed: 4 bytes per parameter less per step); a
 * later cs_mlp_loss_grads starts from a clean buffer either way - ALSO with accumulate != 0: gradients that were already applied
 * (or a buffer that was just rebound) are never the first micro-batch of an accumulation. */
int cs_mlp_train_step(cs_mlp_t* h, const float* x_dev, const float* y_dev, const int64_t* row_idx_dev,
                      int64_t n, int normalise, float lr, float* loss_dev, void* stream);

/* Cooperative-chain health (CS_FLAG_COOP; the counterpart of nothing in Keras - a safety net of this engine):
 * cs_mlp_check synchronises `stream` and returns CS_ERR_STATE if any bounded wait of a cooperative launch has run out
 * since the handle was created (the results since then, optimiser state included, are invalid); call it where the host
 * synchronises anyway (end of an epoch, end of a timed block, before a checkpoint).  cs_mlp_coop_timeouts reads the
 * counter without synchronising (0 for a healthy handle and for handles without the cooperative chain). */
int cs_mlp_check(cs_mlp_t* h, void* stream);
int64_t cs_mlp_coop_timeouts(const cs_mlp_t* h);

/* Diagnostics: bytes of device memory held by the handle. */
int64_t cs_mlp_device_bytes(const cs_mlp_t* h);

/* Measurement aid (bench.py's roofline leg): one cs_mlp_train_step with a hipEvent pair recorded
 * on `stream` around every kernel launch; returns summed milliseconds and launch counts per
 * kernel kind.  Synchronises the stream.  Same arithmetic as cs_mlp_train_step. */
enum { CS_K_PREPARE = 0, CS_K_GEMM_FWD = 1, CS_K_GEMM_DGRAD = 2, CS_K_WGRAD = 3, CS_K_OPTIMIZER = 4,
       CS_K_MEMSET = 5, CS_K_CHAIN_FWD = 6, CS_K_CHAIN_BWD = 7,
       CS_K_CHAIN_FB = 8 /* forward + backward chain in one launch */, CS_K_COUNT = 9 };
typedef struct cs_kernel_times { float ms[CS_K_COUNT]; int32_t launches[CS_K_COUNT]; } cs_kernel_times;
/* The same over many steps with no synchronisation in between (the regime of a timed training loop): every kernel the
 * engine launches on this thread between begin and end carries its own start / stop events (hipExtLaunchKernelGGL: the
 * dispatch packet's timestamps, what rocprofv3 reports); end synchronises `stream` and returns sums and launch counts. */
int cs_profile_begin(void* stream);
int cs_profile_end(cs_kernel_times* out);
int cs_mlp_profile_step(cs_mlp_t* h, const float* x_dev, const float* y_dev, const int64_t* row_idx_dev,
                        int64_t n, int normalise, float lr, float* loss_dev, void* stream,
                        cs_kernel_times* out);

/* Development aid: s_memtime stamps of the chain kernels (needs CS_CHAIN_DBG=1 at create time):
 * [fwd|bwd][workgroup][64 slots] shader-clock ticks.  Not for production use. */
int cs_mlp_debug_stamps(cs_mlp_t* h, unsigned long long* host, int64_t n_words);
/* ... and of the last k_wgrad3 launch: [workgroup][8] words (csrc/wgrad2.h: entry, stage 0 landed, contraction done, stores issued,
 * stores acknowledged in shader clocks; 100 MHz entry / exit; stages); *grid = its workgroups. */
int cs_mlp_debug_stamps_wgrad(cs_mlp_t* h, unsigned long long* host, int64_t n_words, int32_t* grid);

/* The shuffle of the input pipeline (`.unbatch().shuffle(buffer, reshuffle_each_iteration).batch(bs)`, step2_retrain.py:266-277) for
 * rows that live in HBM: out_dev[0..n) = a permutation of 0..n-1 keyed by `seed` (4-round Feistel network over the index bits,
 * cycle-walked into range: a bijection by construction, one pass, no sort).  The row indices cs_mlp_train_step / cs_mlp_loss_grads
 * gather by.  n <= 2^31. */
int cs_permutation(int64_t n, uint64_t seed, int64_t* out_dev, void* stream);

/* Stand-alone loader-path kernel (data_utils.py:807-809 + :894-897 on device): out = (x-sub)/div,
 * inf/nan -> 0, float32 -> float32; rows gathered through row_idx when given. */
int cs_normalise_rows(const float* x_dev, const int64_t* row_idx_dev, int64_t n, int32_t width,
                      const float* sub_dev, const float* div_dev, float* out_dev, void* stream);

/* ---- level-axis 1-D CNN (baseline_models/CNN/training/hpo_train.py:124-236) ----
 * ResNet-style: depth x { Conv1D(C,3,'same')+ReLU+Dropout, Conv1D(C,3,'same')+ReLU+Dropout, + Conv1D(C,1)(block input) },
 * Conv1D(10,1)+ELU, per-level Dense(10->n_lin) || Dense(10->10-n_lin, relu).  Weights, optimiser slots and
 * gradients are flat float32 buffers in Keras order (kernels (k, c_in, c_out); per block conv a, conv b,
 * projection).  Inputs are (n,124) flat rows (x3d = 0; the reshape of data_utils.reshape_input_for_cnn,
 * data_utils.py:1692-1712, happens in the first kernel) or (n,60,6) (x3d = 1); targets (n,128) flat
 * (y3d = 0; reshape_target_for_cnn :1714-1738 applied on the fly) or (n,60,10) (y3d = 1). */
typedef struct cs_cnn cs_cnn_t;
typedef enum { CS_CNN_LOSS_MAE_ADJUSTED = 0, CS_CNN_LOSS_MSE_ADJUSTED = 1 } cs_cnn_loss_kind;   /* hpo_train.py:114-121 */
/* Raw timestep fields -> normalised float32 training rows on the device: the per-file work of
 * data_utils.load_ncdata_with_generator + save_as_npy (data_utils.py:698-711, 807-809, 815-820, 894-897, 906).
 * mli_dev [n_steps][n_in][ncol], mlo_dev [n_steps][n_out][ncol] (float64 when src_f64 else float32), feature rows in
 * stacking order; tend_src_dev[f] >= 0 marks a tendency target (mlo row f minus mli row tend_src[f], over 1200 s).
 * x_out (n_steps*ncol, n_in), y_out (n_steps*ncol, n_out); either may be NULL.  Arithmetic in float64. */
int cs_loader_stack(const void* mli_dev, const void* mlo_dev, int32_t src_f64, int64_t n_steps, int32_t ncol, int32_t n_in,
                    const double* in_sub_dev, const double* in_div_dev, int32_t n_out, const int32_t* tend_src_dev,
                    const double* out_scale_dev, float* x_out_dev, float* y_out_dev, void* stream);

/* Evaluation metrics on the device: data_utils.output_weighting + calc_MAE / calc_RMSE / calc_R2 / calc_bias with
 * avg_grid=False (data_utils.py:1112-1362, 1432-1497).  pred/target (n_steps*ncol, n_out) float32 rows, row = t*ncol + c;
 * weight(row, f) = (wa[f] + wb[f]*ps[row]) * area[c]  (constants folded on the host, see climsim_amd/metrics.py).
 * stats_dev [ncol][n_out][6] float64: slots 0..3 = MAE, RMSE, R2, bias per grid column and output (4, 5: scratch). */
int cs_metrics_columns(const float* pred_dev, const float* target_dev, int64_t n_steps, int32_t ncol, int32_t n_out,
                       const double* ps_dev, const double* wa_dev, const double* wb_dev, const double* area_dev,
                       double* stats_dev, void* stream);
/* The same with the surface pressure taken inside the kernel from the (normalised) input rows x_dev (n_steps*ncol, n_in) float32:
 * ps[row] = x[row][ps_index] * ps_mul + ps_add in float64 - data_utils.set_pressure_grid reads state_ps from the input rows
 * (data_utils.py:1037-1086); saves the caller's gather + conversion passes over the rows.  Needs n_out % 4 == 0 and 16-byte aligned rows. */
int cs_metrics_columns_x(const float* pred_dev, const float* target_dev, int64_t n_steps, int32_t ncol, int32_t n_out,
                         const float* x_dev, int32_t n_in, int32_t ps_index, double ps_mul, double ps_add,
                         const double* wa_dev, const double* wb_dev, const double* area_dev, double* stats_dev, void* stream);

/* ---- many trials per GPU / ensembles: K members, ONE grouped launch per kernel kind --------------------------------
 * The reference searches ~8k small MLPs, five worker processes per GPU (hpo_baseline_v1.py:221-245, 255-260, batch
 * 48..3072), and trains a 32-member ensemble of one shape (baseline_models/RPN/training/rpn_model_v1_data.py:71-163).
 * A small-batch step of one model occupies n/32 of the 256 CUs; a group runs the step of K members - each with its own
 * handle, weights, optimiser rule, learning rate, step count and batch - as three launches (layer chains, weight
 * gradients, optimisers) whose grids are the members' grids laid end to end.  Members must be of one kernel family
 * (cs_mlp_kernel_family: 1 = hidden widths 128/256/512 and 128 outputs, 2 = any widths up to 1024; +16 = ELU) on one
 * device; they need not share an architecture.  Results per member are those of cs_mlp_train_step on that member alone
 * (same kernels bodies, same arithmetic). */
typedef struct cs_mlp_group cs_mlp_group_t;
#define CS_GROUP_MAX_MEMBERS 32
int cs_mlp_kernel_family(const cs_mlp_t* h);
int cs_mlp_group_create(cs_mlp_group_t** g, cs_mlp_t* const* members, int32_t k);   /* members are borrowed, not owned */
void cs_mlp_group_destroy(cs_mlp_group_t* g);
int32_t cs_mlp_group_size(const cs_mlp_group_t* g);
/* One optimiser step of every member i with n[i] > 0 (n[i] == 0 leaves a member out of this step).  Host arrays of k
 * entries: x_dev[i] / y_dev[i] device rows (members may share a split), row_idx_dev[i] gathered rows or NULL (the array
 * itself may be NULL), n[i] batch size, lr[i] learning rate; loss_dev (device, k x 2 floats) receives each member's
 * [sum of squared errors, sum of absolute errors].  Asynchronous on `stream`. */
int cs_mlp_group_train_step(cs_mlp_group_t* g, const float* const* x_dev, const float* const* y_dev,
                            const int64_t* const* row_idx_dev, const int64_t* n, int normalise, const float* lr,
                            float* loss_dev, void* stream);
/* model.predict / model.evaluate of every member i with n[i] > 0 in ONE launch (the validation pass of model.fit: the
 * reference runs `validation_data` every epoch for every trial, hpo_baseline_v1.py:139-150, 223-245).  Arguments as
 * cs_mlp_forward, one entry per member: yhat_dev[i] (n[i], n_out) float32 or NULL (the array itself may be NULL), y_dev[i]
 * targets or NULL (the array may be NULL); with targets, loss_dev[2 i] += sum (yhat-y)^2 (or the member's loss terms),
 * loss_dev[2 i + 1] += sum |yhat-y|; loss_dev (k x 2 floats) is zeroed first unless accumulate != 0.  Members built after a
 * later cs_mlp_set_head_options are picked up; a member moved to another kernel family fails the call (CS_ERR_STATE). */
int cs_mlp_group_forward(cs_mlp_group_t* g, const float* const* x_dev, const int64_t* const* row_idx_dev, const int64_t* n,
                         int normalise, float* const* yhat_dev, const float* const* y_dev, float* loss_dev, int accumulate,
                         void* stream);
/* cs_mlp_profile_step for a group: event pairs around the three launches. */
int cs_mlp_group_profile_step(cs_mlp_group_t* g, const float* const* x_dev, const float* const* y_dev,
                              const int64_t* const* row_idx_dev, const int64_t* n, int normalise, const float* lr,
                              float* loss_dev, void* stream, cs_kernel_times* out);

/* Training-pass `accuracy` (the CSVLogger column `accuracy`, step2_retrain.py:160-162,262): with a non-null count_dev every later
 * cs_mlp_train_step / cs_mlp_loss_grads also writes the predictions of its batch into an internal buffer and adds the number of rows
 * with argmax(y_true) == argmax(y_pred) to *count_dev (the caller zeroes it per epoch; accuracy = count / rows seen).  Null switches
 * it off again (the default: a regression has no use for it, and it costs 512 B of stores per column). */
int cs_mlp_set_train_accuracy(cs_mlp_t* h, unsigned long long* count_dev);

/* Keras' `accuracy` metric for a (B, width) regression target (compile(metrics=['mse','mae','accuracy']),
 * step2_retrain.py:160-162; for a multi-column target Keras resolves it to categorical_accuracy:
 * argmax(y_true, -1) == argmax(y_pred, -1), first maximum on ties).  *count_dev (+)= number of matching rows of
 * pred/target (n, width) float32; the CSVLogger columns `accuracy` / `val_accuracy` are count / n. */
int cs_categorical_accuracy(const float* pred_dev, const float* target_dev, int64_t n, int32_t width,
                            unsigned long long* count_dev, int accumulate, void* stream);

#define CS_CNN_FLAG_TILE128 1   /* development: run every conv on the 128x128-tile kernel (A/B runs, parity cross-check) */
typedef struct cs_cnn_cfg {
    int32_t depth;       /* hp_depth = 12            */
    int32_t channels;    /* hp_channel_width = 406   */
    int32_t kernel;      /* hp_kernel_width = 3      */
    int32_t seq;         /* 60 levels                */
    int32_t c_in;        /* 6  input channels        */
    int32_t c_out;       /* 10 output channels       */
    int32_t n_lin;       /* 2 linear heads, rest relu */
    int32_t max_batch;   /* columns per call          */
    int32_t device;
    int32_t flags;
    int32_t train;       /* 1: allocate activations, gradients and optimiser slots */
    int32_t optimizer;   /* CS_OPT_ADAM | CS_OPT_SGD   (hpo_train.py:215-219)      */
    int32_t loss;        /* cs_cnn_loss_kind; hp_loss "mean_absolute_error" -> mae_adjusted */
    int32_t reserved;
    double dropout;      /* hp_dropout = 0.175 */
    double beta1, beta2, eps;   /* keras.optimizers.Adam defaults 0.9, 0.999, 1e-7 */
    uint64_t seed;       /* dropout stream: call k of cs_cnn_loss_grads uses seed + k */
} cs_cnn_cfg;
/* The trunk's convolutions run as launches of up to 12 convs with an in-launch hand-off between the workgroups of a row tile (bounded
 * waits; csrc/conv2.h, CS_CNN_FUSE).  A wait that runs out - or workgroups of a row tile found on different XCDs - is counted by the
 * kernel in host-mapped memory: every later cs_cnn_predict / cs_cnn_loss_grads / cs_cnn_train_step on the handle fails with
 * CS_ERR_STATE (what such a launch computed is not trusted); recreate the model, with CS_CNN_FUSE=1 for one conv per launch. */
int  cs_cnn_create(cs_cnn_t** out, const cs_cnn_cfg* cfg);            /* CNNHyperModel().build()   */
void cs_cnn_destroy(cs_cnn_t* h);
int64_t cs_cnn_num_params(const cs_cnn_t* h);                          /* model.count_params()      */
int  cs_cnn_set_weights(cs_cnn_t* h, const float* host, int64_t n, void* stream);   /* model.set_weights */
int  cs_cnn_get_weights(cs_cnn_t* h, float* host, int64_t n, void* stream);         /* model.get_weights */
int  cs_cnn_get_opt_state(cs_cnn_t* h, float* host_m, float* host_v, int64_t n, int64_t* iterations, void* stream);
int  cs_cnn_set_opt_state(cs_cnn_t* h, const float* host_m, const float* host_v, int64_t n, int64_t iterations, void* stream);
/* model.predict (dropout = identity).  Outputs: (n,60,10) f32 and/or the flat (n,128) form of
 * data_utils.reshape_target_from_cnn (data_utils.py:1740-1761); either may be NULL. */
int  cs_cnn_forward(cs_cnn_t* h, const float* x_dev, int x3d, int64_t n, float* out3d_dev, float* out_flat_dev,
                    void* stream);
/* model.evaluate / validation pass: loss_dev[4] (+)= [sum|e| profile channels, sum|e| scalar channels,
 * sum e^2 profile, sum e^2 scalar] over the n columns;  mae_adjusted = s0/(n*seq*n_lin)*(120/128) +
 * s1/(n*seq*(10-n_lin))*(8/128), mse_adjusted likewise from s2, s3. */
int  cs_cnn_evaluate(cs_cnn_t* h, const float* x_dev, int x3d, const float* y_dev, int y3d, const int64_t* row_idx_dev,
                     int64_t n, float* loss_dev, int accumulate, void* stream);
/* Training-mode forward (dropout on) + backward.  Gradients (of n*seq*loss, i.e. unscaled sums) overwrite
 * the flat gradient buffer; loss_dev[4] as above. */
int  cs_cnn_loss_grads(cs_cnn_t* h, const float* x_dev, int x3d, const float* y_dev, int y3d, const int64_t* row_idx_dev,
                       int64_t n, float* loss_dev, void* stream);
/* The two remaining entries of the reference's compile(metrics=[...]) list (hpo_train.py:83-111, 231): while `dev2` (two floats of
 * device memory owned by the caller, or NULL to switch this off) is set, every cs_cnn_evaluate / cs_cnn_loss_grads / cs_cnn_train_step
 * ADDS [sum over (column, level) of the continuous_ranked_probability_score terms mean_j|p_j - y_j| - 1/2 mean_ij|p_i - p_j| over the
 * 10 channels, number of (column, level) pairs whose argmax over the channels agrees between target and prediction ("accuracy" =
 * Keras categorical accuracy)] to it; divide by n*seq.  The caller zeroes it (once per epoch: Keras metrics are running means). */
int  cs_cnn_set_metrics_buffer(cs_cnn_t* h, float* dev2);
/* Restart the dropout stream: call k of cs_cnn_loss_grads uses seed + k.  Under data parallelism every rank takes its
 * own stream (climsim_amd.cnn.CNNEmulator.fit: seed + 1000003 * rank). */
int  cs_cnn_set_seed(cs_cnn_t* h, uint64_t seed);
/* Development aid (CS_CNN_DBG=1 at create time): stamps of the last conv weight-gradient launch, [workgroup][16] words
 * (csrc/conv_wgrad2.h: entry, set-up, ring fill, loop, flush in shader clocks; slabs; 100 MHz entry / exit; hardware id);
 * *grid = workgroups of that launch.  Not for production use. */
int  cs_cnn_debug_stamps(cs_cnn_t* h, unsigned long long* host, int64_t n_words, int32_t* grid);
int  cs_cnn_grad_buffer(cs_cnn_t* h, void** grad_dev, int64_t* n_floats);     /* payload of the DP all-reduce */
int  cs_cnn_set_grad_buffer(cs_cnn_t* h, float* grad_dev, int64_t n_floats);  /* bind a caller-owned buffer (NULL: own) */
/* optimizer.apply_gradients: w -= update(grad_scale * G); grad_scale = 1/(n*seq*world_size); zeroes G. */
int  cs_cnn_apply(cs_cnn_t* h, float lr, float grad_scale, void* stream);
/* one Model.train_step of model.fit (hpo_train.py:231-236) */
int  cs_cnn_train_step(cs_cnn_t* h, const float* x_dev, int x3d, const float* y_dev, int y3d, const int64_t* row_idx_dev,
                       int64_t n, float lr, float* loss_dev, void* stream);

/* ---- data parallel: one all-reduce(SUM) of a flat float32 buffer over RCCL (xGMI), on the CALLER'S stream ----------
 * The only gradient-synchronous training in the reference is torch DDP over NCCL
 * (online_testing/baseline_models/MLP_v2rh/training/train_mlp_h5loader.py:195-207).  Here the collective is issued
 * on the stream the step's kernels run on, so it is ordered with them without events or a second stream.
 * The RCCL library is bound at run time (`rccl_path` = the librccl.so the process already uses, e.g. torch/lib/librccl.so;
 * NULL: "librccl.so.1" from the loader path): libclimsim_hip.so itself has no link-time dependency on it.
 *   rank 0:     cs_dp_unique_id(rccl_path, id)  ->  128 opaque bytes, handed to the other ranks by the launcher's
 *               rendezvous (torch.distributed broadcast in climsim_amd/dp.py)
 *   every rank: cs_dp_init(&comm, rccl_path, id, world, rank, device)       (collective)
 *   every step: cs_dp_allreduce(comm, grad_dev, n, stream)                  (in place; grad_dev of cs_*_grad_buffer)
 * A communicator is used from one host thread; destroy it before the device is reset. */
typedef struct cs_dp cs_dp_t;
#define CS_DP_UNIQUE_ID_BYTES 128
int  cs_dp_unique_id(const char* rccl_path, void* id_out);
int  cs_dp_init(cs_dp_t** comm, const char* rccl_path, const void* id, int world, int rank, int device);
int  cs_dp_allreduce(cs_dp_t* comm, float* buf_dev, int64_t n_floats, void* stream);
/* Same collective with HALF the bytes on the links: buf_dev is rounded to bf16 (nearest-even) into a buffer the
 * communicator owns, summed over the ranks as ncclBfloat16, and widened back into buf_dev - three operations on `stream`,
 * nothing else changes (the optimiser keeps its float32 moments and master weights).  An option, not the default: DDP in
 * the reference reduces float32 (train_mlp_h5loader.py:195-207); the reduced sums carry bf16 rounding (2^-9 relative per
 * hop), every rank still receives the SAME buffer.  buf_dev must be 16-byte aligned. */
int  cs_dp_allreduce_bf16(cs_dp_t* comm, float* buf_dev, int64_t n_floats, void* stream);
/* What RCCL itself says about the communicator: ncclCommCount / ncclCommUserRank (bench.py prints them beside the
 * launcher's WORLD_SIZE, so a scaling figure cannot come from a group of the wrong size). */
int  cs_dp_comm_info(cs_dp_t* comm, int* nranks, int* rank);
void cs_dp_destroy(cs_dp_t* comm);

/* ---- one-shot all-reduce over peer-mapped buffers: an OPTION beside cs_dp_allreduce (same role: the one gradient
 * all-reduce of a DDP step, train_mlp_h5loader.py:195-207).  Every rank owns an exchange buffer that the other ranks of the
 * NODE map through HIP IPC handles; one kernel per rank pulls + sums its 1/world slice from all buffers and pushes the sum
 * into every buffer (two flag exchanges, every wait bounded).  The engine's gradient buffer is rebound to the exchange
 * buffer (cs_mlp_set_grad_buffer / cs_cnn_set_grad_buffer), so nothing is copied.
 *   every rank: cs_dp_ipc_create(&c, world, rank, device, n_floats)        n_floats = gradient length rounded up to 4
 *               cs_dp_ipc_export(c, record)                                CS_DP_IPC_HANDLE_BYTES opaque bytes
 *               ... all-gather the records through the launcher's rendezvous ...
 *               cs_dp_ipc_connect(c, records_of_all_ranks)                 world x CS_DP_IPC_HANDLE_BYTES, rank order
 *               cs_dp_ipc_buffer(c, &ptr, &n)  ->  cs_mlp_set_grad_buffer(h, ptr)
 *   every step: cs_dp_ipc_allreduce(c, n_floats, stream)                   in place on the exchange buffer
 * At most 8 ranks, one node.  Tested with two processes on one device; NOT yet run over xGMI links: RCCL stays the default. */
typedef struct cs_dp_ipc cs_dp_ipc_t;
#define CS_DP_IPC_HANDLE_BYTES 128
int  cs_dp_ipc_create(cs_dp_ipc_t** out, int world, int rank, int device, int64_t n_floats);
int  cs_dp_ipc_export(cs_dp_ipc_t* c, void* record_out);
int  cs_dp_ipc_connect(cs_dp_ipc_t* c, const void* all_records);
int  cs_dp_ipc_buffer(cs_dp_ipc_t* c, void** dev_ptr, int64_t* n_floats);
int  cs_dp_ipc_allreduce(cs_dp_ipc_t* c, int64_t n_floats, void* stream);
int64_t cs_dp_ipc_timeouts(const cs_dp_ipc_t* c);
/* Bound of every wait inside cs_dp_ipc_allreduce in wall-clock milliseconds (default 30000, environment CS_DP_IPC_TIMEOUT_MS at
 * creation): generous on purpose - rank skew from a checkpoint, a validation pass or a first-step build on one rank is not a
 * fault of the collective; a wait that does run out is counted (cs_dp_ipc_timeouts) and fails the next call, never hangs. */
int  cs_dp_ipc_set_timeout_ms(cs_dp_ipc_t* c, double ms);
void cs_dp_ipc_destroy(cs_dp_ipc_t* c);

const char* cs_last_error(void);
const char* cs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CLIMSIM_HIP_H */
