/* libunet_hip.so -- C ABI of the MI355X-native UNet3d conv engine.
 *
 * This is the drop-in boundary under the reference's UNet3d operator surface: a host (the C++
 * `UNet3dImpl` of include/unet.hpp, or the Python mirror in unet-studio_amd/) owns parameters,
 * gradients and the workspace as device memory and drives the engine through these calls.  No
 * torch types cross this boundary: plain pointers, sizes and a hipStream_t (passed as void*).
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *   unet_init / unet_device_info     cuda.cu:34-74 (check_cuda device enumeration)
 *   unet_plan_create                 UNet3dImpl::UNet3dImpl + create_layer        unet.cpp:24-166
 *   unet_plan_param_* / buffer_*     parameters()/buffers() order and shapes      unet.cpp:130,160-164; main.cpp:193-204
 *   unet_forward                     UNet3dImpl::forward (libtorch conv/norm/...)  unet.cpp:168-193; callers train.cpp:628,840,
 *                                    evaluate.cpp:226, qc.cpp:88
 *   unet_backward                    autograd backward of forward                 train.cpp:706
 *   unet_loss                        calc_losses + deep-supervision loop          train.cpp:501-552,634-706
 *   unet_sgd_step                    /batch_size, clip_grad_norm_(12), SGD step   train.cpp:759-766; unet.cpp:246-277
 *   unet_op_*                        single libtorch modules (unit-test surface)  unet.cpp:38-98
 *
 * All functions return 0 on success, non-zero on error; unet_last_error() returns a thread-local
 * message (the C++ host rethrows it as std::runtime_error so the reference's catch blocks at
 * train.cpp:709-721,1134 keep working).  All device pointers must live on the plan's device.
 * Calls on one plan are re-entrant as long as each concurrent call has its own workspace and stream
 * (qc.cpp:273-297 calls forward on one model from several threads).
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_DTYPE_F32 0  /* activations, gradients and arithmetic in fp32 (parity configuration) */
#define UNET_DTYPE_BF16 1 /* activations/gradients bf16, fp32 accumulate, fp32 master parameters */

#define UNET_IMPL_AUTO 0    /* MFMA kernels where the shape allows, direct kernels elsewhere */
#define UNET_IMPL_DIRECT 1  /* direct (non-MFMA) HIP kernels only */

typedef struct unet_plan unet_plan;

const char* unet_last_error(void);
void unet_set_error(const char* msg); /* internal: lets the library's other translation units report through unet_last_error */

/* device enumeration (cuda.cu:34-74) */
int unet_init(int* n_devices);
int unet_device_info(int device, char* name, size_t name_len, size_t* total_mem, int* compute_units, int* gcn_arch_is_gfx950);

/* Plan: the parsed architecture DSL bound to one input size, dtype and device.
 * arch_dsl: text of UNet3dImpl::architecture; in_c/out_c: in_count/out_count; D,H,W: input volume
 * (tensor {1,in_c,D,H,W}, x fastest).  Fails with the reference's messages for DSL errors. */
int unet_plan_create(const char* arch_dsl, int in_c, int out_c, int D, int H, int W, int dtype, int device, int impl,
                     unet_plan** out);
void unet_plan_destroy(unet_plan* plan);

/* parameters() order (unet.cpp:130,160-164).  dims has room for 5 entries. */
int unet_plan_param_count(const unet_plan* plan, int* n);
int unet_plan_param_shape(const unet_plan* plan, int i, int64_t dims[5], int* ndim);
/* named_parameters() key of parameter i, e.g. "encode0.3.weight" (module registration names of unet.cpp:130,160-164) */
int unet_plan_param_name(const unet_plan* plan, int i, char* name, size_t name_len);
/* 1 if weight decay applies to parameter i (unet.cpp:254: dim > 1 and no "bias" in the name) */
int unet_plan_param_decay(const unet_plan* plan, int i, int* decay);
/* fan_in of parameter i's owning conv (0 for norm affine): default-init bound 1/sqrt(fan_in) */
int unet_plan_param_fan_in(const unet_plan* plan, int i, int64_t* fan_in, int* is_norm_weight);
/* buffers() order: per bnorm layer running_mean[C], running_var[C] (fp32) ; num_batches_tracked is host-side */
int unet_plan_buffer_count(const unet_plan* plan, int* n);
int unet_plan_buffer_shape(const unet_plan* plan, int i, int64_t* numel);
/* outputs: one per decoder level, [0] = full resolution; dims = {1,C,D,H,W}; C = 0 when the level has no head */
int unet_plan_output_count(const unet_plan* plan, int* n);
int unet_plan_output_shape(const unet_plan* plan, int level, int64_t dims[5]);
int unet_plan_workspace_bytes(const unet_plan* plan, size_t* bytes);
/* algorithmic conv + conv-transpose FLOPs (2*MAC) of one forward, and of backward (dgrad + wgrad) */
int unet_plan_flops(const unet_plan* plan, double* fwd, double* bwd);
/* human-readable op list of the lowered graph (for print_layers()/debugging); returns needed length */
size_t unet_plan_describe(const unet_plan* plan, char* buf, size_t len);

/* lowered op list (forward order): kind 0 pack_input, 1 conv, 2 conv_trans, 3 norm, 4 materialize, 5 max_pool, 6 upsample,
 * 7 export; dims are {D,H,W} of the first source / the destination tensor (0 where there is none) */
int unet_plan_op_count(const unet_plan* plan, int* n);
int unet_plan_op_info(const unet_plan* plan, int i, int* kind, int* cin, int* cout, int* ks, int* stride, int64_t in_dims[3],
                      int64_t out_dims[3], char* name, size_t name_len);

/* Per-op timing for the measurement harness (bench.py: conv_mfma_frac, the time-dominant kernel).  Between begin and end every
 * unet_forward / unet_backward issued by THIS host thread runs on the caller's stream only and brackets each op's launches with
 * HIP events; end synchronises and returns one record per bracket: the op index (-1: the batched filter pack), a category and
 * the elapsed milliseconds.  No reference counterpart (the reference has no profiler). */
#define UNET_PROF_CONV_FWD 0 /* conv / conv_trans forward kernels (+ their per-op filter packs) */
#define UNET_PROF_DGRAD 1    /* input-gradient kernels */
#define UNET_PROF_WGRAD 2    /* weight/bias-gradient kernels and their slab reduces */
#define UNET_PROF_NORM_FWD 3 /* norm finalize + activated copy */
#define UNET_PROF_NORM_BWD 4 /* norm / activation backward */
#define UNET_PROF_OTHER 5    /* input pack, pool, upsample, heads' backward, export, batched filter pack */
int unet_profile_begin(void);
int unet_profile_end(int max_records, int* op_index, int* category, float* ms, int* n_records);

/* forward.  params: host array of device pointers (fp32, parameters() order).  buffers: host array of
 * device pointers (fp32 running_mean/running_var pairs) or NULL when the architecture has no bnorm.
 * x: fp32 {1,in_c,D,H,W}.  outs: host array of device pointers, fp32 {1,out_c,D>>l,H>>l,W>>l} per level
 * (NULL entries are skipped).  mode: 0 = eval (bnorm uses running stats: train.cpp:834-840, and after
 * prepare_for_inference y = gamma*x+beta), 1 = train (batch statistics, running stats updated).
 * workspace: unet_plan_workspace_bytes() bytes; after a mode-1 forward it holds what backward needs. */
/* mode | UNET_MODE_PACKS_CURRENT (either mode): the caller asserts that the filter packs this workspace holds are current -- i.e. a
 * forward OF THE SAME MODE has run on THIS workspace since the parameters last changed (micro-steps 2..batch_size of one optimizer
 * step, train.cpp:604-606: the parameters only change at :765; volumes 2.. of an inference run, evaluate.cpp:211-246).  The engine
 * then skips the repack of the filters (the batched bf16 pack, the fp32 engine's per-layer packs). */
#define UNET_MODE_PACKS_CURRENT 2
int unet_forward(const unet_plan* plan, const float* const* params, float* const* buffers, const float* x,
                 float* const* outs, void* workspace, int mode, void* stream);

/* backward of the last mode-1 forward on this workspace.  grad_outs: fp32 dL/d(outs[l]) or NULL (= no
 * loss on that level).  grad_params: fp32, parameters() order, ACCUMULATED (+=) as .grad is across the
 * batch_size micro-steps of one optimizer step (train.cpp:604-606,706).  grad_x: must be NULL (the reference never asks for
 * dL/dx -- the input is a leaf without requires_grad, train.cpp:619-628 -- and the engine does not compute it; non-NULL is an error).
 * The parameter-gradient kernels run on a side stream the plan owns, forked from and joined back into `stream`
 * before the call returns its work to the caller: stream order is all a caller needs, but two host threads must
 * not run unet_backward on the SAME plan concurrently (the reference trains one model per thread, train.cpp:573-579;
 * unet_forward stays re-entrant per workspace, qc.cpp:273-297). */
int unet_backward(const unet_plan* plan, const float* const* params, const float* const* grad_outs,
                  float* const* grad_params, float* grad_x, void* workspace, void* stream);

/* The same backward issued in parts, so that a host can start the all-reduce of a finished gradient bucket while the rest of
 * the backward runs (replaces the serial reduce-to-root of add_gradient_from, unet.cpp:224-244; train.cpp:756-757).
 * unet_plan_backward_buckets: bucket k covers ops [op_lo[k], op_lo[k-1]) (op_lo[-1] = "all", pass a large op_hi) and, once run,
 * leaves the gradients of the flat parameter elements [elem_lo[k], elem_lo[k-1]) final (elem_lo[-1] = parameter count).
 * unet_backward_part runs ops [op_lo, op_hi) of the backward; calling it for the buckets in order equals unet_backward. */
int unet_plan_backward_buckets(const unet_plan* plan, int max_buckets, int* n_buckets, int* op_lo, int64_t* elem_lo);
int unet_backward_part(const unet_plan* plan, const float* const* params, const float* const* grad_outs, float* const* grad_params,
                       float* grad_x, void* workspace, int op_hi, int op_lo, void* stream);

/* calc_losses over all deep-supervision levels (train.cpp:501-552,634-706).
 * target: int64 {1,D,H,W} labels (values >= out_c are masked out); cost_mask bit0 ce, bit1 dice, bit2 mse
 * (0 behaves as ce only: train.cpp:696-697); collapse_before as calc_losses.
 * losses_out (device, 4 floats): {total, ce0, dice0, mse0}.  grad_outs[l] (may be NULL = no gradient
 * wanted) receives dL_total/d(outs[l]).  scratch: unet_loss_scratch_bytes() bytes. */
int unet_loss_scratch_bytes(const unet_plan* plan, size_t* bytes);
int unet_loss(const unet_plan* plan, const float* const* outs, const int64_t* target, int cost_mask, int collapse_before,
              float* const* grad_outs, float* losses_out, void* scratch, void* stream);

/* unet_forward (mode 1) + unet_loss in one call.  Same results as the two calls; the engine may issue the loss of the coarse
 * deep-supervision levels beside the rest of the decoder (train.cpp:628 followed by :634-706 for one sample). */
int unet_forward_loss(const unet_plan* plan, const float* const* params, float* const* buffers, const float* x, float* const* outs,
                      const int64_t* target, int cost_mask, int collapse_before, float* const* grad_outs, float* losses_out,
                      void* loss_scratch, void* workspace, void* stream);
/* the same with the forward's mode spelled out: 1, or 1 | UNET_MODE_PACKS_CURRENT */
int unet_forward_loss_mode(const unet_plan* plan, const float* const* params, float* const* buffers, const float* x, float* const* outs,
                           const int64_t* target, int cost_mask, int collapse_before, float* const* grad_outs, float* losses_out,
                           void* loss_scratch, void* workspace, int mode, void* stream);

/* EXPERIMENTAL (micro-steps in flight on disjoint parts of the chip; measured in DESIGN.md section 6): a stream confined to the CUs
 * [cu_first, cu_first + cu_count) of EVERY XCD (32 per XCD on MI355X; the indices wrap), and the same for a plan's side stream.  Such a
 * stream synchronizes with the NULL stream (hipExtStreamCreateWithCUMask takes no flags): use it with callers on non-NULL streams. */
int unet_stream_create_cu_range(int device, int cu_first, int cu_count, void** stream);
int unet_stream_destroy(void* stream);
int unet_plan_side_cu_range(unet_plan* plan, int cu_first, int cu_count);

/* The filter packs of `workspace` made from the current parameter values, on `stream` (what a mode-1 forward without
 * UNET_MODE_PACKS_CURRENT does first, on the plan's side stream, beside its first kernels).  A trainer may call it right behind
 * unet_sgd_step -- the parameters are still in the last-level cache -- and give the next forward on this workspace
 * UNET_MODE_PACKS_CURRENT.  with_dgrad = 0: the forward packs only (all an inference forward reads).  *made = 1 when the packs were
 * made, 0 when this plan has no batched pack (fp32 engine, parameters not in one flat buffer): the caller must NOT claim
 * UNET_MODE_PACKS_CURRENT then. */
int unet_pack_filters(const unet_plan* plan, const float* const* params, void* workspace, int with_dgrad, int* made, void* stream);

/* Gradient buffers of micro-steps that ran side by side (each into a buffer of its own) -> the ONE buffer the step epilogue reads:
 * out[i] = ((bufs[0][i] + bufs[1][i]) + bufs[2][i]) + ... in fp32, exactly the association a single buffer ends up with when the
 * micro-steps accumulate into it one after the other (train.cpp:604-606,706: .grad accumulates; 0 + x == x), so the update is
 * bit-identical to the sequential order.  bufs: host array of n device pointers (n <= UNET_SUM_MAX_BUFFERS; out may be bufs[0]);
 * all 16-byte aligned, count floats each.  zero_inputs != 0: the inputs other than `out` are cleared for the next step. */
#define UNET_SUM_MAX_BUFFERS 64
int unet_sum_buffers(const float* const* bufs, int n, float* out, int64_t count, int zero_inputs, void* stream);

/* step epilogue over flat buffers (train.cpp:759-766 + SGD(momentum, nesterov, weight decay) of
 * unet.cpp:254-275): g *= grad_scale (1/batch_size); coef = min(1, clip_norm/(||g||+1e-6));
 * d = coef*g + wd*p (wd only on decay parameters); m = momentum*m + d; p -= lr*(d + momentum*m) (nesterov)
 * or lr*m; g = 0.  params/grads/momentum are flat fp32 buffers holding the tensors in parameters()
 * order, contiguous.  norm_out (device, 1 float) receives ||g|| before clipping.  scratch: 64 KiB. */
int unet_sgd_step(const unet_plan* plan, float* params_flat, float* grads_flat, float* momentum_flat, float lr,
                  float momentum, int nesterov, float weight_decay, float clip_norm, float grad_scale, float* norm_out,
                  void* scratch, void* stream);

/* ---- single-op surface (unit tests, parity per kernel).  Activations are channels-last [D][H][W][C]
 * in the element type of `dtype`; weights/bias/grads fp32 in torch layout ([Cout,Cin,k,k,k]; conv_trans
 * [Cin,Cout,2,2,2]).  impl: UNET_IMPL_*.  scratch: at least unet_op_scratch_bytes() bytes. ---- */
int unet_op_scratch_bytes(int cin, int cout, int D, int H, int W, size_t* bytes);
int unet_op_conv3d_fwd(int dtype, int impl, const void* x, const float* w, const float* b, void* y, int cin, int cout, int D,
                       int H, int W, int ks, int stride, void* scratch, void* stream);
/* conv3d with the read-side fusion of the engine: the input is seen as act(x*scale[c]+shift[c]) (scale/shift fp32 [cin] or
 * NULL, act 0 none 1 relu 2 leaky_relu(0.01) 3 elu), zero padding applied after it; stats (may be NULL) receives the
 * per-channel {sum, sum of squares} of the stored output, fp32 [cout][2] (what the following norm layer needs). */
int unet_op_conv3d_fwd_fused(int dtype, int impl, const void* x, const float* scale, const float* shift, int act, const float* w,
                             const float* b, void* y, float* stats, int cin, int cout, int D, int H, int W, int ks, int stride,
                             void* scratch, void* stream);
/* The two halves of unet_op_conv3d_fwd_fused as a plan runs them (unet_forward packs every filter once per call, then
 * launches one kernel per torch::nn::Conv3d, unet.cpp:59-72): pack the fp32 filter into MFMA fragment order, then launch
 * ONLY the convolution kernel.  wpacked and stats_partials (may be NULL; per-block {sum, sum of squares} partials of the
 * stored output) each need unet_op_scratch_bytes() bytes.  Fails for shapes the MFMA kernels do not cover. */
int unet_op_conv3d_pack(int dtype, const float* w, void* wpacked, int cin, int cout, int D, int H, int W, int ks, int stride,
                        void* stream);
int unet_op_conv3d_fwd_packed(int dtype, const void* x, const void* wpacked, const float* b, void* y, float* stats_partials,
                              int cin, int cout, int D, int H, int W, int ks, int stride, void* stream);
int unet_op_conv3d_bwd_data(int dtype, int impl, const void* dy, const float* w, void* dx, int cin, int cout, int D, int H,
                            int W, int ks, int stride, void* scratch, void* stream);
int unet_op_conv3d_bwd_weight(int dtype, int impl, const void* x, const void* dy, float* dw, float* db, int cin, int cout,
                              int D, int H, int W, int ks, int stride, void* scratch, void* stream);
int unet_op_convt_fwd(int dtype, int impl, const void* x, const float* w, const float* b, void* y, int cin, int cout, int D,
                      int H, int W, void* scratch, void* stream);
int unet_op_convt_bwd_data(int dtype, int impl, const void* dy, const float* w, void* dx, int cin, int cout, int D, int H,
                           int W, void* scratch, void* stream);
int unet_op_convt_bwd_weight(int dtype, int impl, const void* x, const void* dy, float* dw, float* db, int cin, int cout,
                             int D, int H, int W, void* scratch, void* stream);
/* layout helpers: fp32 NCDHW <-> channels-last element type */
int unet_op_pack_ndhwc(int dtype, const float* x_ncdhw, void* y_ndhwc, int C, int64_t S, void* stream);
int unet_op_unpack_ncdhw(int dtype, const void* x_ndhwc, float* y_ncdhw, int C, int64_t S, void* stream);

/* ---- gradient collectives over RCCL / xGMI ----
 * Replaces the reference's replica synchronisation: UNet3dImpl::add_gradient_from (unet.cpp:224-244; call site train.cpp:756-757:
 * per parameter `grad.to(device0)` + `add_` on the root) becomes ONE sum all-reduce of the flat fp32 gradient buffer (or of its
 * finished buckets, under the rest of the backward -- unet_backward_part / unet_plan_backward_buckets), after which every rank
 * applies the identical unet_sgd_step; the per-step weight broadcast of copy_from (unet.cpp:195-222, train.cpp:573-579) is then
 * only needed once, at start (unet_comm_broadcast).
 * One process per GPU: rank 0 calls unet_comm_unique_id and hands the 128 bytes to the other ranks out of band (a file, a socket,
 * MPI, torch.distributed's store), every rank calls unet_comm_create.  One process driving several GPUs from threads (the
 * reference's model, train.cpp:592-600): unet_comm_create_all + unet_allreduce_grads_all from one thread.
 * Collectives are enqueued on a stream the communicator owns, ordered after `stream` (where the buffer became final) by an
 * event; unet_comm_join makes `stream` wait for everything enqueued so far.  librccl is bound at run time (no link dependency). */
#define UNET_COMM_ID_BYTES 128
typedef struct unet_comm unet_comm;
int unet_comm_unique_id(void* id_bytes /* UNET_COMM_ID_BYTES */);
int unet_comm_create(int rank, int world, const void* id_bytes, int device, unet_comm** out);
int unet_comm_create_all(int n, const int* devices, unet_comm** out /* n entries */);
int unet_comm_destroy(unet_comm* comm);
int unet_comm_rank(const unet_comm* comm, int* rank, int* world);
/* flat[elem_lo:elem_hi] <- sum over ranks, in place (fp32) */
int unet_allreduce_grads(unet_comm* comm, float* flat, int64_t elem_lo, int64_t elem_hi, void* stream);
int unet_allreduce_grads_all(unet_comm* const* comms, int n, float* const* flats, int64_t elem_lo, int64_t elem_hi, void* const* streams);
/* buf[0:n] <- root's buf (initial parameters; bnorm running statistics follow the root like copy_from's buffers, unet.cpp:207-215) */
int unet_comm_broadcast(unet_comm* comm, float* buf, int64_t n, int root, void* stream);
int unet_comm_join(unet_comm* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif
