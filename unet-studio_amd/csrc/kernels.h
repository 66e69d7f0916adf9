// Launchers of the HIP kernels (gfx950).  Every function enqueues on `stream` and returns; errors are
// reported through hipGetLastError() by the caller.  dtype: 0 fp32, 1 bf16 element type of the
// channels-last activation/gradient tensors.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/unet_augment.h"
#include "../../include/unet_hip.h"

namespace unet {

// A tensor as a consumer sees it: act(x * scale[c] + shift[c]) (scale == nullptr: identity affine).
struct SrcDesc {
    const void* ptr = nullptr;
    int C = 0;
    const float* scale = nullptr;
    const float* shift = nullptr;
    int act = 0;
};
// Gradient destination of one source: written (accumulate == 0) or added to (accumulate == 1).
struct DstGrad {
    void* ptr = nullptr;  // nullptr: this source needs no gradient
    int C = 0;
    int accumulate = 0;
};

struct ConvGeom {
    int Cin = 0, Cout = 0;
    int D = 0, H = 0, W = 0;     // input volume
    int Do = 0, Ho = 0, Wo = 0;  // output volume
    int ks = 3, stride = 1;
};

__host__ __device__ static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---- weight repacking (fp32 torch layout -> kernel layouts), once per parameter update ----
// conv  [Cout][Cin][k3]  -> fwd  [k3][Cin][CoutP]   and dgrad [k3][Cout][CinP]   (P = padded to 8)
void launch_pack_conv_w(const float* w, float* w_fwd, float* w_dgrad, int Cin, int Cout, int k3, hipStream_t s);
// convT [Cin][Cout][8]   -> fwd  [8][Cin][CoutP]    and dgrad [8][Cout][CinP]
void launch_pack_convt_w(const float* w, float* w_fwd, float* w_dgrad, int Cin, int Cout, hipStream_t s);

// ---- layout ----
void launch_pack_input(int dtype, const float* x_ncdhw, void* y, int C, int64_t S, hipStream_t s);
void launch_export(int dtype, SrcDesc src, float* y_ncdhw, int64_t S, hipStream_t s);                // view -> fp32 NCDHW
void launch_import_grad(int dtype, const float* g_ncdhw, void* g, int C, int64_t S, int accumulate, hipStream_t s);

// ---- direct (non-MFMA) conv family ----
// out: channels-last element type, or (out_ncdhw != nullptr) fp32 NCDHW external output
void launch_conv_fwd_direct(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const float* w_fwd, const float* bias,
                            void* out, float* out_ncdhw, hipStream_t s);
void launch_conv_dgrad_direct(int dtype, const ConvGeom& g, const void* dy, const float* w_dgrad, const DstGrad* dst, int ndst,
                              hipStream_t s);
// dw/db in torch layout, accumulated (+=).  scratch (wgrad_direct_scratch_bytes, may be nullptr) lets the kernel split
// the voxel range over more blocks (partial slabs summed in a fixed order)
size_t wgrad_direct_scratch_bytes(const ConvGeom& g, int transposed);
void launch_conv_wgrad_direct(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db,
                              void* scratch, hipStream_t s);
void launch_convt_fwd_direct(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const float* w_fwd, const float* bias,
                             void* out, hipStream_t s);
void launch_convt_dgrad_direct(int dtype, const ConvGeom& g, const void* dy, const float* w_dgrad, const DstGrad* dst, int ndst,
                               hipStream_t s);
void launch_convt_wgrad_direct(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db,
                               void* scratch, hipStream_t s);

// ---- normalisation ----
// number of partial blocks the statistics kernels use for S voxels (plan-time constant)
int stats_blocks(int64_t S);
// per-block partial {sum, sumsq} of a raw tensor: partial[blk][c][2]
// block partials: float for bf16 tensors, DOUBLE for fp32 tensors (then pass dbl = true to the finalize / sum that reads them)
void launch_stats_partial(int dtype, const void* x, int C, int64_t S, float* partial, hipStream_t s);
// partials -> stat[0..C) mean, [C..2C) rstd, [2C..3C) scale = gamma*rstd, [3C..4C) shift = beta - mean*scale;
// running stats (bnorm, may be nullptr): rm = (1-m)*rm + m*mean, rv = (1-m)*rv + m*unbiased var
void launch_norm_finalize(const float* partial, int nblk, int C, int64_t S, const float* gamma, const float* beta, double eps,
                          float* stat, float* running_mean, float* running_var, double momentum, hipStream_t s, bool dbl = false);
// partials -> out[c][2] = {sum, sum of squares}
void launch_stats_sum(const float* partial, int nblk, int C, float* out, hipStream_t s, bool dbl = false);
// eval-mode bnorm: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale (mean := rm, rstd := 1/sqrt(rv+eps))
void launch_norm_eval(int C, const float* gamma, const float* beta, const float* rm, const float* rv, double eps, float* stat,
                      hipStream_t s);

// ---- view backward: g holds dL/d(act(norm(u))) on entry ----
// no norm: g *= act'(u)
void launch_act_bwd(int dtype, void* g, const void* u, int act, int64_t n, hipStream_t s);
// with norm: pass 1: g <- dv = g*act'(v), partial[blk][c] = {sum dv, sum dv*xhat}
void launch_norm_bwd_partial(int dtype, void* g, const void* u, int C, int64_t S, const float* stat, int act, float* partial,
                             hipStream_t s);
// pass 2: coef[0..C) = gamma*rstd, [C..2C) = mean(dv), [2C..3C) = mean(dv*xhat); dgamma += sum dv*xhat, dbeta += sum dv
void launch_norm_bwd_finalize(const float* partial, int nblk, int C, int64_t S, const float* gamma, const float* stat, float* coef,
                              float* dgamma, float* dbeta, hipStream_t s, bool dbl = false);
// finalize + activated copy in ONE launch where the partial rows are few (bf16, <= 128 rows, C <= 512): returns false when the
// shape does not qualify (the caller then uses launch_norm_finalize + launch_apply_view)
bool launch_norm_finalize_apply(int dtype, const float* partial, int nblk, int C, int64_t S, const float* gamma, const float* beta, double eps,
                                float* stat, float* running_mean, float* running_var, double momentum, const void* raw, int act, void* out,
                                hipStream_t s);
// the same for passes 2 + 3 of the backward
bool launch_norm_bwd_finalize_apply(int dtype, const float* partial, int nblk, int C, int64_t S, const float* gamma, const float* stat,
                                    float* coef, float* dgamma, float* dbeta, void* g, const void* u, int act, hipStream_t s);
// pass 3: g <- du = coef0 * (dv - m1 - xhat*m2)
void launch_norm_bwd_apply(int dtype, void* g, const void* u, int C, int64_t S, const float* stat, const float* coef, int act,
                           hipStream_t s);

// ---- pooling / resampling / copies ----
void launch_maxpool_fwd(int dtype, SrcDesc src, void* out, int D, int H, int W, hipStream_t s);
void launch_maxpool_bwd(int dtype, SrcDesc src, const void* gout, DstGrad dst, int D, int H, int W, hipStream_t s);
void launch_upsample_fwd(int dtype, SrcDesc src, void* out, int D, int H, int W, hipStream_t s);
void launch_upsample_bwd(int dtype, const void* gout, DstGrad dst, int D, int H, int W, hipStream_t s);
void launch_materialize(int dtype, const SrcDesc* src, int nsrc, void* out, int64_t S, hipStream_t s);
// out = act(src*scale+shift) for a whole tensor (vectorised for bf16): the activated copy that consumers read
void launch_apply_view(int dtype, SrcDesc src, void* out, int64_t S, hipStream_t s);
void launch_materialize_bwd(int dtype, const void* gout, const DstGrad* dst, int ndst, int64_t S, hipStream_t s);
void launch_export_bwd(int dtype, const float* g_ncdhw, DstGrad dst, int64_t S, hipStream_t s);
void launch_unpack_ncdhw(int dtype, const void* g, float* out_ncdhw, int C, int64_t S, hipStream_t s);

// ---- losses (train.cpp:501-552, 634-706) ----
void launch_target_half(const int64_t* t, int64_t* o, int D, int H, int W, hipStream_t s);
int loss_blocks(int64_t S);
// pass 1: per-block partials [blk][3 + 2*oc]: {ce, mse, nvalid, inter[oc], card[oc]}
void launch_loss_partial(const float* logits, const int64_t* target, int C, int64_t S, int collapse, float* partial, hipStream_t s);
// pass 2: level_out[0..2] = ce, dice, mse; level_out[3 ..] = n, inter[oc], card[oc]; totals[0] += weight*(selected), totals[1..3] = stats when set_stats
void launch_loss_finalize(const float* partial, int nblk, int oc, float level_weight, int cost_mask, float* level_out, float* totals,
                          int set_stats, hipStream_t s);
// pass 3: dlogits = level_weight * d(selected losses)/dlogits
void launch_loss_grad(const float* logits, const int64_t* target, int C, int64_t S, int collapse, const float* level_out,
                      float level_weight, int cost_mask, float* dlogits, hipStream_t s);

// out = ((b0 + b1) + b2) + ... element-wise in exactly that order (n <= UNET_SUM_MAX_BUFFERS, 16-B aligned); zero_inputs clears b* (not out)
void launch_sum_buffers(const float* const* bufs, int n, float* out, int64_t count, int zero_inputs, hipStream_t s);

// ---- step epilogue ----
struct SgdSeg { int64_t offset, count; float wd; };  // wd: 1 when weight decay applies to the tensor, else 0
void launch_sumsq_partial(const float* g, int64_t n, float scale, float* partial, int nblk, hipStream_t s);
void launch_sgd(float* p, float* g, float* m, int64_t n, const SgdSeg* segs_dev, int nseg, const float* partial, int nblk,
                float lr, float momentum, int nesterov, float weight_decay, float clip_norm, float grad_scale, float* norm_out,
                hipStream_t s);

// ---- MFMA implicit-GEMM family (kernels_mfma_conv.hip, kernels_mfma_wgrad.hip), bf16 only ----
// one filter pack of the batched pack kernel: fp32 parameter (src_off floats from the flat parameter base) ->
// bf16 fragments at dst_off bytes into the workspace; blk0 = first 256-thread block of the job in the launch
struct PackJob { int64_t src_off, dst_off, total, blk0; int Ci, Co, CK, T, mode, A, B, pad; };
int mfma_conv_pack_jobs(const ConvGeom& g, bool want_dgrad, PackJob* out2);
int mfma_convt_pack_jobs(const ConvGeom& g, PackJob* out2);
// one launch serves the pack units [blk_base, blk_base + nblocks) of the table (njobs = the whole table); max_grid > 0 bounds the
// number of blocks (a block then walks several units); zero / nzero: ints the launch also clears (the deep levels' arrival counters)
void launch_mfma_pack_batched(const float* params_base, void* ws, const PackJob* jobs_dev, int njobs, int64_t nblocks, hipStream_t s,
                              int64_t blk_base = 0, int max_grid = 0, int* zero = nullptr, int nzero = 0);
bool mfma_conv_fwd_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
size_t mfma_conv_w_bytes(const ConvGeom& g);
void launch_mfma_pack_conv_w(const float* w, void* w_mfma_fwd, void* w_mfma_dgrad, const ConvGeom& g, hipStream_t s);
// returns the number of statistics partial rows written to stats_partial ([rows][Cout][2], one per persistent block)
int launch_mfma_conv_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out,
                         float* stats_partial, hipStream_t s);
// the first conv (Cin = 1, 3x3x3 stride 1, Cout 16 or 32, plain bf16 input) on the matrix cores, filter read as fp32 torch layout;
// returns the number of statistics partial rows (<= conv_first_mfma_blocks)
bool conv_first_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
int conv_first_mfma_blocks(const ConvGeom& g);
int launch_conv_first_mfma(const ConvGeom& g, const SrcDesc* src, const float* w, const float* bias, void* out, float* stats_partial,
                           hipStream_t s);
// upper bound of that row count (tiles of the geometry): sizes the partials buffer
int mfma_conv_blocks(const ConvGeom& g);
// wgrad (+ bias grad) of a 3x3x3 conv, stride 1 or 2; dw/db fp32 torch layout, accumulated (+=); db may be nullptr
// sliding-window wgrad of the 3x3x3 stride-1 convs (kernels_mfma_wgrad_z.hip): kernel only, returns the number of slab rows
// written at `scratch` ([rows][Cout][Cin][27], then [rows][Cout] bias partials when want_bias); 0 = shape not covered
bool mfma_wgrad_z_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
size_t mfma_wgrad_z_scratch_bytes(const ConvGeom& g, int polite = 0);
int mfma_wgrad_z_splits(const ConvGeom& g, int polite = 0);
int launch_mfma_wgrad_z(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, bool want_bias, void* scratch, hipStream_t s,
                        int polite = 0);
// dw[i] += sum over rows of slab[row][i] (n % 4 == 0), db[c] += sum of bias_slab[row][c]; fixed order, no atomics
void wgrad_reduce(const float* slab, const float* bias_slab, int nsplit, int64_t n, int Cb, float* dw, float* db, hipStream_t s);
// one launch for many layers' slabs: offsets in floats from the workspace base (slab, bias partials; bias_off < 0: none) and
// from the flat gradient buffer (dw, db)
struct WgradReduceJob { long long slab_off, bias_off, dw_off, db_off, n; int nsplit, Cb, ly, blk0, nblk, op; };
int wgrad_reduce_job_blocks(WgradReduceJob& j, int blk0);
void launch_wgrad_reduce_batched(const WgradReduceJob* jobs_dev, int job0, int njobs, int blk_base, int nblocks, const void* ws, float* gflat,
                                 hipStream_t s);
bool mfma_wgrad_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
// polite = 1 (weight-gradient launches only): the layer's gradient runs on the side stream beside the caller's stream's latency-bound small
// levels -- one 4-wave block per CU (one wave per SIMD), so that the caller's kernels always find registers and LDS.  The slab row count
// depends on it: the same value must be given to *_scratch_bytes, *_splits and the launch (engine.cpp: Plan::side_polite).
size_t mfma_wgrad_scratch_bytes(const ConvGeom& g, int polite = 0);
// defer_reduce: only the slab ([rows][n] + [rows][Cout] bias partials at `scratch`) is written; rows = *_wgrad_splits(g)
int mfma_conv_wgrad_splits(const ConvGeom& g, int polite = 0);
int mfma_convt_wgrad_splits(const ConvGeom& g);
// small volumes: the launch ADDS its result into dw / db itself (output-stationary blocks; no slab, defer_reduce is ignored)
bool mfma_conv_wgrad_direct(const ConvGeom& g);
bool mfma_convt_wgrad_direct(const ConvGeom& g);
int conv_first_wgrad_splits(const ConvGeom& g);
void launch_mfma_conv_wgrad(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db, void* scratch,
                            hipStream_t s, bool defer_reduce = false, int polite = 0);
// wgrad of layers with <= 1024 weights (Cin = 1 first conv, 6-channel heads): row-staged, HBM-bound
// 1x1x1 heads (Cout <= 8, Cin = 16 * 2^k <= 256, one plain or viewed source): forward writes results[level] (fp32 NCDHW) and/or
// the channels-last tensor; backward = dL/dW (+=), dL/db (+=) and dL/d(source view) in one pass, dy as fp32 NCDHW or channels-last
bool head_supported(const ConvGeom& g, int nsrc);
size_t head_bwd_scratch_bytes(const ConvGeom& g);
void launch_head_fwd(int dtype, const ConvGeom& g, const SrcDesc& src, const float* w, const float* bias, void* y, float* out_ncdhw,
                     hipStream_t s);
// defer_reduce: only the slab is written (scratch must stay untouched until launch_head_bwd_reduce has run on a stream ordered after `s`)
void launch_head_bwd(int dtype, const ConvGeom& g, const SrcDesc& src, const float* dy_ncdhw, const void* dy_cl, const float* w,
                     DstGrad dst, float* dw, float* db, void* scratch, hipStream_t s, bool defer_reduce = false);
void launch_head_bwd_reduce(const ConvGeom& g, float* dw, float* db, const void* scratch, hipStream_t s);
// wgrad (+ bias grad) of the first conv (Cin = 1, 3x3x3 stride 1, Cout 16 or 32, plain bf16 input) on the matrix cores
bool conv_first_wgrad_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
size_t conv_first_wgrad_mfma_scratch_bytes(const ConvGeom& g);
// nb_fuse: dy is dL/d(activated view); the norm backward's element-wise pass (k_norm_bwd_apply8's arithmetic with the coefficients
// k_norm_bwd_finalize left) is applied as the tiles are staged -- the same bits as running that pass first, without its 3 tensor transfers
struct NormBwdFuse { const void* u; const float* stat; const float* coef; int act; };
void launch_conv_first_wgrad_mfma(const ConvGeom& g, const SrcDesc* src, const void* dy, float* dw, float* db, void* scratch, hipStream_t s,
                                  bool defer_reduce = false, const NormBwdFuse* nb_fuse = nullptr);
bool wgrad_small_supported(const ConvGeom& g, int nsrc);
size_t wgrad_small_scratch_bytes(const ConvGeom& g);
void launch_conv_wgrad_small(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db,
                             void* scratch, hipStream_t s);
// conv_trans wgrad (single source); db (optional): the bias gradient rides along (sums of dy as the kernel stages it: [rows][Cout]
// partials behind the slab, or += db itself in the direct form) -- no pass of its own over dy
bool mfma_convt_wgrad_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
size_t mfma_convt_wgrad_scratch_bytes(const ConvGeom& g);
void launch_mfma_convt_wgrad(const ConvGeom& g, const SrcDesc* src, const void* dy, float* dw, void* scratch, hipStream_t s,
                             bool defer_reduce = false, float* db = nullptr, int polite = 0);
// vectorised per-block column sums partial[blk][C] of a bf16 [S][C] tensor; returns #blocks (0: not applicable)
int launch_colsum_partial8(int dtype, const void* x, int C, int64_t S, float* partial, hipStream_t s);
size_t bias_grad_scratch_bytes(int C, int64_t S);
void launch_bias_grad(int dtype, const void* dy, int C, int64_t S, float* db, void* scratch, hipStream_t s);
// dgrad of a 3x3x3 conv, stride 1 or 2 (g = forward geometry)
bool mfma_conv_dgrad_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
size_t mfma_conv_dgrad_w_bytes(const ConvGeom& g);
// bn (optional): also leave the norm-backward statistics of the destination tensor's view (what launch_norm_bwd_partial computes) in
// bn->partial; returns the number of partial rows written, 0 when the shape is not served by a kernel with that epilogue (the caller
// then runs launch_norm_bwd_partial as usual).  Only for a single destination that is written, not accumulated.
struct BnBwdStats { const void* u; const float* stat; float* partial; int act, C; };
int launch_mfma_conv_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, hipStream_t s,
                           const BnBwdStats* bn = nullptr);
// kernels_mfma_s2.hip: sliding-window / LDS-DMA kernels for the contractions that cross a resolution boundary, coarse grid >= 16 wide
// (the launchers below try them first; 0 / false = shape not served, the halo-tile kernel k_mfma_conv_p runs instead).
// launch_s2_conv_fwd returns the number of statistics rows; launch_s2_conv_dgrad the number of norm-backward partial rows it left in
// bn->partial (its destination may be ACCUMULATED into: the old values and the raw tensor travel by LDS-DMA too), -1 without bn
int launch_s2_conv_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out, float* stats_partial,
                       hipStream_t s);
int launch_s2_conv_dgrad(const ConvGeom& g, const void* dy, const void* w_s2_dgrad, const DstGrad* dst, int ndst, hipStream_t s,
                         const BnBwdStats* bn);
int s2_conv_dgrad_rows_max();
bool launch_s2_convt_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out, hipStream_t s);
bool launch_s2_convt_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, hipStream_t s);
// kernels_mfma_s2_wgrad.hip: sliding-window weight gradient of Conv3d(k3, s2) (ks = 3: fine = the input, a concat when fine2 != nullptr,
// coarse = dy) and ConvTranspose3d(k2, s2) (ks = 2: fine = dy, coarse = the input; bias_from_a: the bias partials are sums of the fine
// tensor).  Kernel only: slab [rows][Cb][Ca][ks^3] (+ [rows][Cb or Ca] bias partials behind it) at `scratch`; returns rows, 0 = not served
int s2_wgrad_splits(int Ca, int Cb, int cD, int cH, int cW);
int launch_s2_wgrad(int ks, const void* fine, const void* fine2, int C0, int Ca, int fD, int fH, int fW, const void* coarse, int Cb, int cD, int cH,
                    int cW, bool want_bias, int bias_from_a, void* scratch, hipStream_t s, int polite);
// kernels_mfma_deep.hip: the deep levels (output grids of DEEP_MAX_VOXELS voxels or fewer): split-K implicit GEMM straight from global
// memory, finished by the block that arrives last; with nf / nb the norm layer behind (forward) or in front of (backward) the conv runs
// in that epilogue too.  The launchers below are tried first by the engine; false / 0 = shape not served.  DeepScratch: fp32 partial tiles
// and the arrival counters (ints, ZERO before the first launch; every launch leaves them zero) of one workspace.
constexpr int64_t DEEP_MAX_VOXELS = 512;
struct DeepScratch { float* part = nullptr; size_t part_bytes = 0; int* cnt = nullptr; int ncnt = 0; };
struct DeepNormFwd {      // the norm + activation on the conv's output: statistics of this launch (use_running == 0) or the running ones
    const float* gamma; const float* beta; double eps; float* stat; float* rm; float* rv; double momentum; int use_running; int act;
    void* act_out;        // the activated copy [voxel][Cout] bf16
};
struct DeepNormBwd {      // the norm + activation whose view the dgrad's destination is: u = the raw tensor, stat = {mean, rstd, scale, shift}
    const void* u; const float* stat; const float* gamma; float* coef; float* dgamma; float* dbeta; int act;
};
bool deep_conv_applies(int dtype, int64_t out_voxels, int cin, int cout);
bool launch_deep_conv_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out,
                          const DeepNormFwd* nf, const DeepScratch& sc, hipStream_t s);
bool launch_deep_convt_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out,
                           const DeepScratch& sc, hipStream_t s);
// 0: not served; 1: gradient written / accumulated; 2: ... and the destination's norm backward done in place (nb given, one destination)
int launch_deep_conv_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, const DeepNormBwd* nb,
                           const DeepScratch& sc, hipStream_t s);
bool launch_deep_convt_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, const DeepNormBwd* nb,
                             const DeepScratch& sc, int* norm_done, hipStream_t s);
// ConvTranspose3d 2x2x2 stride 2: forward (1x1 GEMM + depth-to-space scatter) and dgrad (2x2x2 stride-2 conv of dL/dy)
bool mfma_convt_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
size_t mfma_convt_w_bytes(const ConvGeom& g);
size_t mfma_convt_dgrad_w_bytes(const ConvGeom& g);
void launch_mfma_pack_convt_w(const float* w, void* w_mfma_fwd, void* w_mfma_dgrad, const ConvGeom& g, hipStream_t s);
void launch_mfma_convt_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out, hipStream_t s);
void launch_mfma_convt_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, hipStream_t s);

// kernels_mfma_f32.hip: fp32 3x3x3 stride-1 conv on the fp32 matrix cores (the fp32 engine with impl == AUTO)
bool conv_f32_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
// stats_partial (optional): [rows][Cout][2] fp64 {sum, sum of squares} of the output per block, rows = the return value
// (= conv_f32_mfma_stat_rows): the norm layer's statistics without a pass of its own (read them with dbl = true)
int conv_f32_mfma_stat_rows(const ConvGeom& g, const SrcDesc* src);
int launch_conv_f32_mfma(const ConvGeom& g, const SrcDesc* src, int nsrc, const float* w_fwd, const float* bias, float* out,
                         hipStream_t s, double* stats_partial = nullptr);
// the fp32 engine's first conv (Cin = 1, 3x3x3 stride 1, Cout 16 or 32, plain fp32 input) on the fp32 matrix cores, filter read in torch
// layout; stats_partial (optional): fp64 {sum, sum of squares} rows, one per block (the return value)
bool conv_first_f32_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
int conv_first_f32_mfma_blocks(const ConvGeom& g);
int launch_conv_first_f32_mfma(const ConvGeom& g, const SrcDesc* src, const float* w, const float* bias, float* out, double* stats_partial,
                               hipStream_t s);
// fp32 ConvTranspose3d(k2, s2) forward on the fp32 matrix cores (w_fwd: launch_pack_convt_w's [8][Cin][CoutP])
bool convt_f32_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
void launch_convt_f32_mfma(const ConvGeom& g, const SrcDesc* src, const float* w_fwd, const float* bias, float* out, hipStream_t s);
bool wgrad_f32_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc);
size_t wgrad_f32_mfma_scratch_bytes(const ConvGeom& g);
void launch_wgrad_f32_mfma(const ConvGeom& g, const SrcDesc* src, int nsrc, const float* dy, float* dw, float* db, void* scratch,
                           hipStream_t s);
// out[i] += sum over the nsplit slabs of slab[k][i], fixed order, fp64 (kernels_direct.hip)
void slab_reduce_public(const float* slab, int nsplit, int64_t n, float* out, hipStream_t s);
bool conv_f32_mfma_dgrad_supported(int dtype, const ConvGeom& g, const DstGrad* dst, int ndst);
void launch_conv_f32_mfma_dgrad(const ConvGeom& g, const float* dy, const float* w_dgrad, const DstGrad* dst, int ndst, hipStream_t s);

// kernels_augment.hip: on-GPU sample augmentation (include/unet_augment.h)
size_t augment_scratch_bytes(const UnetAugmentRecipe& r);
void launch_augment(const UnetAugmentRecipe& r, float* image, float* label, void* scratch, hipStream_t st);
size_t simulate_scratch_bytes(const UnetSimulateRecipe& r);
void launch_simulate_modality(const UnetSimulateRecipe& r, float* t1w, const float* label, void* scratch, hipStream_t st);

}  // namespace unet
