// Plan + executor + C ABI (include/unet_hip.h).  The plan is immutable after creation; every call
// works on caller-owned device memory (parameters, gradients, workspace) and a caller-owned stream,
// so concurrent calls on one plan are safe when they use different workspaces (qc.cpp:273-297).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/unet_hip.h"
#include "graph.hpp"
#include "kernels.h"

using namespace unet;

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return 1; }

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        HIP_OK(hipGetDevice(&prev));
        if (prev != dev) HIP_OK(hipSetDevice(dev));
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ---- per-op timing (unet_profile_begin / unet_profile_end) ----
// While a host thread has a profile open, every forward / backward it issues runs on the caller's stream alone (no side
// stream) and each op's launches are bracketed by a pair of HIP events on that stream, tagged (op index, category).
struct ProfRec { int op, cat; hipEvent_t e0, e1; };
struct ProfSink { std::vector<ProfRec> recs; };
thread_local ProfSink* g_prof = nullptr;
struct ProfScope {
    ProfRec r{};
    hipStream_t s = nullptr;
    bool on = false;
    ProfScope(int op, int cat, hipStream_t st) {
        if (!g_prof) return;
        on = true; s = st; r.op = op; r.cat = cat;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(r.e0, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.e1, s);
        g_prof->recs.push_back(r);
    }
};

}  // namespace

struct unet_plan {
    Graph g;
    int dtype = 0, device = 0, impl = 0;
    size_t elsize = 4;
    // workspace layout (byte offsets)
    std::vector<size_t> t_off, g_off;        // tensor storage / gradient storage (SIZE_MAX: none)
    std::vector<size_t> a_off;               // activated copy act(norm(tensor)) that consumers read (SIZE_MAX: none)
    std::vector<size_t> n_stat, n_coef;      // per norm: 4C / 3C floats
    std::vector<size_t> w_fwd, w_dgrad;      // per op (conv / conv_trans): packed fp32 weights
    std::vector<size_t> wm_fwd, wm_dgrad;    // per op: MFMA fragment-order bf16 filters (SIZE_MAX: op not on the MFMA path)
    std::vector<int> n_consumers;            // per tensor: ops that read it
    std::vector<int> first_consumer;         // per tensor: lowest op index that reads it (-1: none) -- in the backward its LAST gradient writer
    // norm-backward partial rows a dgrad epilogue left in a workspace's partial() for the tensor's view_backward, which may run in a later
    // unet_backward_part call on the same workspace (the bucketed backward must make the same choices as the whole one): workspace -> {tensor, rows}
    mutable std::mutex bn_mu;
    mutable std::unordered_map<const void*, std::pair<int, int>> bn_pending;
    // ... and the tensors whose whole norm backward a deep-level dgrad already ran in its epilogue (kernels_mfma_deep.hip): workspace -> tensors
    mutable std::unordered_map<const void*, std::vector<int>> bn_done;
    std::vector<char> use_mfma;              // per op: forward runs on the MFMA kernel
    std::vector<char> dgrad_mfma;            // per op: dgrad runs on the MFMA kernel
    std::vector<char> wgrad_mfma;            // per op: wgrad runs on the MFMA kernel
    size_t wgrad_off = 0;
    // sliding-window wgrads keep their slabs until ONE batched reduce per backward (part): per-op slab regions + the job table
    std::vector<size_t> wz_off;              // per op: slab region (SIZE_MAX: op does not use k_mfma_wgrad_z)
    std::vector<int> wz_job_of_op;           // per op: index into wz_jobs or -1
    std::vector<WgradReduceJob> wz_jobs;     // ascending op index
    WgradReduceJob* wz_jobs_dev = nullptr;
    size_t partial_off = 0, partial_bytes = 0;
    size_t ws_bytes = 0;
    // loss scratch layout
    size_t loss_bytes = 0;
    // batched MFMA filter pack (used when the caller's parameters are one flat contiguous buffer)
    std::vector<int64_t> p_off;              // element offset of parameter i in a flat buffer
    std::vector<PackJob> pack_jobs;
    int64_t pack_blocks = 0;
    PackJob* jobs_dev = nullptr;
    // sgd
    SgdSeg* segs_dev = nullptr;
    int nseg = 0;
    int64_t n_param_elems = 0;

    // backward side stream: the parameter-gradient kernels (wgrad, its reduce, bias grad) of a layer run beside the
    // dgrad -> norm-backward chain of the next one (they only share read-only inputs); forked / joined with events
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pack = nullptr, ev_packd = nullptr;
    // the training forward packs the filters in three launches on the side stream.  The job table holds every op's FORWARD pack first
    // (op order), then every DGRAD pack: [0, pack_split_blocks) = forward packs of the ops before pack_split_op (the encoder's top
    // levels: a few hundred KB), [pack_split_blocks, pack_fwd_blocks) = the other forward packs (the deep levels and the decoder; only
    // awaited by the first op that reads one of them, ~0.3 ms into the forward), [pack_fwd_blocks, pack_blocks) = the dgrad packs (half
    // of the bytes): nothing reads them before the backward, which waits for ev_packd.  An inference forward packs [0, pack_fwd_blocks) only.
    int pack_split_op = 0;
    int64_t pack_split_blocks = 0, pack_fwd_blocks = 0;
    size_t head_off = 0;                     // scratch of the fused head backward (stays on the main stream)
    // the deep levels' split-K kernels (kernels_mfma_deep.hip): fp32 partial tiles + arrival counters, used on the caller's stream only
    size_t deep_part_off = 0, deep_part_bytes = 0, deep_cnt_off = 0;
    static constexpr int deep_ncnt = 4096;
    std::vector<size_t> head_op_off;         // per op: a head's own slab region (its reduce runs on the side stream, later) or SIZE_MAX

    ~unet_plan() {
        if (segs_dev) (void)hipFree(segs_dev);
        if (wz_jobs_dev) (void)hipFree(wz_jobs_dev);
        if (jobs_dev) (void)hipFree(jobs_dev);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_pack) (void)hipEventDestroy(ev_pack);
        if (ev_packd) (void)hipEventDestroy(ev_packd);
        if (side) (void)hipStreamDestroy(side);
    }

    ConvGeom op_geom_of(const Op& op) const {
        const Tensor& a = g.tensors[op.src[0]];
        const Tensor& o = g.tensors[op.dst];
        ConvGeom cg;
        cg.Cin = op.cin; cg.Cout = op.cout; cg.D = a.D; cg.H = a.H; cg.W = a.W; cg.Do = o.D; cg.Ho = o.H; cg.Wo = o.W;
        cg.ks = op.ks; cg.stride = op.stride;
        return cg;
    }
    // Which weight-gradient launches are "polite" (one 4-wave block per CU: mfma_util.h polite_lds): every one that runs on the side
    // stream.  The backward walks the ops from the last to the first -- the decoder's top levels (their gradients are HELD, see
    // backward()), then the small levels, then the encoder's top levels; polite launches for the tail of the step as well measured as
    // fast or faster than giving those the whole chip (profiles/r10e_ab_polite_policy.txt: 2.915-2.93 vs 2.94 ms).
    std::vector<int> side_polite;
    // Weight gradients that run on the CALLER's stream (full occupancy, slab summed there too): the stride-2 convs above 32^3 (the top of
    // the encoder).  Such a kernel becomes ready together with its own dgrad at the tail of the backward, where nothing latency-bound is
    // left to hide it behind, and the two side by side took longer than one after the other (154 us against 55 + 49: both stream the same
    // 64-MB tensors through the same L2s; profiles/r10h_ab_tail_on_main.txt: step 2.92 -> 2.88 ms; the stride-1 layer after it loses 0.03 ms).
    std::vector<char> wgrad_on_main;
    void choose_polite() {
        side_polite.assign(g.ops.size(), 0);
        wgrad_on_main.assign(g.ops.size(), 0);
        bool first = true, any_deep = false;
        for (size_t i = 0; i < g.ops.size(); ++i) {
            const Op& op = g.ops[i];
            if (op.kind != OP_CONV && op.kind != OP_CONVT) continue;
            if (g.tensors[op.dst].voxels() <= (int64_t)32 * 32 * 32) { any_deep = true; break; }
            if (!first && op.kind == OP_CONV && op.stride == 2) wgrad_on_main[i] = 1;
            first = false;
        }
        if (!any_deep) return;      // a network without small levels has nothing latency-bound to be polite to
        for (size_t i = 0; i < g.ops.size(); ++i) {
            const Op& op = g.ops[i];
            if (op.kind == OP_CONV || op.kind == OP_CONVT) side_polite[i] = wgrad_on_main[i] ? 0 : 1;
        }
    }

    // tensor t is read by fused heads only (bf16): see layout()
    bool head_only(size_t t) const {
        if (dtype != UNET_DTYPE_BF16 || g.tensors[t].norm < 0 || g.tensors[t].C % 16) return false;
        int readers = 0;
        for (const Op& op : g.ops) {
            if (op.kind == OP_NORM) continue;
            for (int k = 0; k < op.nsrc; ++k) {
                if (op.src[k] != (int)t) continue;
                if (!(op.kind == OP_CONV && op.out_level >= 0 && op.nsrc == 1 && head_supported(op_geom_of(op), 1))) return false;
                ++readers;
            }
        }
        return readers > 0;
    }
    void layout() {
        choose_polite();
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
        t_off.assign(g.tensors.size(), SIZE_MAX);
        g_off.assign(g.tensors.size(), SIZE_MAX);
        n_consumers.assign(g.tensors.size(), 0);
        first_consumer.assign(g.tensors.size(), -1);
        for (size_t oi = 0; oi < g.ops.size(); ++oi) {
            const Op& op = g.ops[oi];
            for (int k = 0; k < op.nsrc && op.kind != OP_NORM; ++k)     // (a norm op names the tensor it normalises: not a reader of the view)
                if (op.src[k] >= 0) { ++n_consumers[op.src[k]]; if (first_consumer[op.src[k]] < 0) first_consumer[op.src[k]] = (int)oi; }
        }
        a_off.assign(g.tensors.size(), SIZE_MAX);
        for (size_t i = 0; i < g.tensors.size(); ++i) {
            t_off[i] = take((size_t)g.tensors[i].numel() * elsize);
            if (g.tensors[i].needs_grad) g_off[i] = take((size_t)g.tensors[i].numel() * elsize);
            // Activated copy act(norm(u)): one extra write + the consumers read it as is.  Measured on the 32->16 conv at
            // 128^3: transforming in the conv's staging loop costs +85 % of the kernel (VALU-bound, repeated for the
            // 2.5x halo re-reads), the separate 2-pass copy ~0.03 ms.  288 GB of HBM makes the extra tensor free.
            // ... except for a tensor whose only readers are fused heads (the decoder's last tensor at full resolution): the head kernels are
            // bandwidth-bound element-wise passes that transform as they load, so the copy (read + write of 64 MB at 128^3) is never made.
            if (impl == UNET_IMPL_AUTO && (g.tensors[i].norm >= 0 || g.tensors[i].act != ACT_NONE) && !head_only(i))
                a_off[i] = take((size_t)g.tensors[i].numel() * elsize);
        }
        n_stat.resize(g.norms.size()); n_coef.resize(g.norms.size());
        size_t pmax = 0;
        for (size_t i = 0; i < g.norms.size(); ++i) {
            n_stat[i] = take(4 * (size_t)g.norms[i].C * 4);
            n_coef[i] = take(3 * (size_t)g.norms[i].C * 4);
            size_t pb = (size_t)stats_blocks(g.tensors[g.norms[i].tensor].voxels()) * g.norms[i].C * 2 * (dtype == UNET_DTYPE_F32 ? 8 : 4);
            if (pb > pmax) pmax = pb;
        }
        w_fwd.assign(g.ops.size(), SIZE_MAX); w_dgrad.assign(g.ops.size(), SIZE_MAX);
        wm_fwd.assign(g.ops.size(), SIZE_MAX); wm_dgrad.assign(g.ops.size(), SIZE_MAX);
        use_mfma.assign(g.ops.size(), 0);
        dgrad_mfma.assign(g.ops.size(), 0);
        for (size_t i = 0; i < g.ops.size(); ++i) {
            const Op& op = g.ops[i];
            if (op.kind != OP_CONV && op.kind != OP_CONVT) continue;
            int k3 = op.kind == OP_CONV ? op.ks * op.ks * op.ks : 8;
            w_fwd[i] = take((size_t)k3 * op.cin * round_up(op.cout, 8) * 4);
            w_dgrad[i] = take((size_t)k3 * op.cout * round_up(op.cin, 8) * 4);
            if (op.kind == OP_CONV && impl == UNET_IMPL_AUTO && op.out_level < 0) {
                ConvGeom cg = op_geom_of(op);
                SrcDesc sd[2];
                for (int k = 0; k < op.nsrc; ++k) sd[k].C = g.tensors[op.src[k]].C;
                if (mfma_conv_fwd_supported(dtype, cg, sd, op.nsrc)) {
                    use_mfma[i] = 1;
                    wm_fwd[i] = take(mfma_conv_w_bytes(cg));
                    if (g.tensors[op.dst].norm >= 0) {
                        size_t pb = (size_t)mfma_conv_blocks(cg) * op.cout * 2 * 4;
                        if (pb > pmax) pmax = pb;
                    }
                }
                if (conv_first_f32_mfma_supported(dtype, cg, sd, op.nsrc) && g.tensors[op.dst].norm >= 0) {
                    size_t pb = (size_t)conv_first_f32_mfma_blocks(cg) * op.cout * 2 * 8;
                    if (pb > pmax) pmax = pb;
                }
                if (conv_f32_mfma_supported(dtype, cg, sd, op.nsrc) && g.tensors[op.dst].norm >= 0) {   // fp64 statistics rows of the fp32 conv
                    size_t pb = (size_t)conv_f32_mfma_stat_rows(cg, sd) * op.cout * 2 * 8;
                    if (pb > pmax) pmax = pb;
                }
                if (mfma_conv_dgrad_supported(dtype, cg, sd, op.nsrc)) {
                    dgrad_mfma[i] = 1;
                    wm_dgrad[i] = take(mfma_conv_dgrad_w_bytes(cg));
                    if (op.stride == 2) {      // norm-backward partial rows of the stride-2 dgrad's epilogue (kernels_mfma_s2.hip)
                        size_t pb = (size_t)s2_conv_dgrad_rows_max() * op.cin * 2 * 4;
                        if (pb > pmax) pmax = pb;
                    }
                }
            }
            if (op.kind == OP_CONVT && impl == UNET_IMPL_AUTO) {
                ConvGeom cg = op_geom_of(op);
                SrcDesc sd[2];
                for (int k = 0; k < op.nsrc; ++k) sd[k].C = g.tensors[op.src[k]].C;
                if (mfma_convt_supported(dtype, cg, sd, op.nsrc)) {
                    use_mfma[i] = 1; dgrad_mfma[i] = 1;
                    wm_fwd[i] = take(mfma_convt_w_bytes(cg));
                    wm_dgrad[i] = take(mfma_convt_dgrad_w_bytes(cg));
                }
            }
        }
        partial_bytes = pmax ? pmax : 256;
        partial_off = take(partial_bytes);
        // MFMA wgrad: one shared slab scratch (ops run one after another on the stream)
        wgrad_mfma.assign(g.ops.size(), 0);
        size_t wmax = 0, hmax = 0;
        for (size_t i = 0; i < g.ops.size(); ++i) {
            const Op& op = g.ops[i];
            if (op.kind != OP_CONV || impl != UNET_IMPL_AUTO) continue;
            ConvGeom cg = op_geom_of(op);
            SrcDesc sd[2];
            for (int k = 0; k < op.nsrc; ++k) sd[k].C = g.tensors[op.src[k]].C;
            if (mfma_wgrad_supported(dtype, cg, sd, op.nsrc)) {
                wgrad_mfma[i] = 1;
                size_t b = mfma_wgrad_scratch_bytes(cg, side_polite[i]);
                if (b > wmax) wmax = b;
            }
        }
        for (size_t i = 0; i < g.ops.size(); ++i) {
            const Op& op = g.ops[i];
            if (op.kind != OP_CONV && op.kind != OP_CONVT) continue;
            ConvGeom cg = op_geom_of(op);
            size_t b = wgrad_direct_scratch_bytes(cg, op.kind == OP_CONVT);
            if (b > wmax) wmax = b;
            if (op.kind == OP_CONV && wgrad_small_supported(cg, op.nsrc)) {
                b = wgrad_small_scratch_bytes(cg);
                if (b > wmax) wmax = b;
            }
            if (op.kind == OP_CONV && impl == UNET_IMPL_AUTO) {
                SrcDesc sd[2];
                for (int k = 0; k < op.nsrc; ++k) sd[k].C = g.tensors[op.src[k]].C;
                if (wgrad_f32_mfma_supported(dtype, cg, sd, op.nsrc)) {
                    b = wgrad_f32_mfma_scratch_bytes(cg);
                    if (b > wmax) wmax = b;
                }
            }
            if (op.kind == OP_CONV && cg.Cin == 1 && (cg.Cout == 16 || cg.Cout == 32)) {
                b = conv_first_wgrad_mfma_scratch_bytes(cg);
                if (b > wmax) wmax = b;
            }
            if (op.kind == OP_CONV && head_supported(cg, op.nsrc)) {
                b = head_bwd_scratch_bytes(cg);
                if (b > hmax) hmax = b;
            }
            if (op.kind == OP_CONVT && impl == UNET_IMPL_AUTO) {
                SrcDesc sd[2];
                for (int k = 0; k < op.nsrc; ++k) sd[k].C = g.tensors[op.src[k]].C;
                if (mfma_convt_wgrad_supported(dtype, cg, sd, op.nsrc)) {
                    wgrad_mfma[i] = 1;
                    b = mfma_convt_wgrad_scratch_bytes(cg);
                    if (b > wmax) wmax = b;
                    b = bias_grad_scratch_bytes(cg.Cout, (int64_t)cg.Do * cg.Ho * cg.Wo);
                    if (b > wmax) wmax = b;
                }
            }
        }
        wgrad_off = take(wmax ? wmax : 256);
        head_off = take(hmax ? hmax : 256);
        if (dtype == UNET_DTYPE_BF16 && impl == UNET_IMPL_AUTO) {
            deep_part_bytes = (size_t)8 << 20;
            deep_part_off = take(deep_part_bytes);
            deep_cnt_off = take((size_t)deep_ncnt * 4);
        }
        head_op_off.assign(g.ops.size(), SIZE_MAX);
        for (size_t i = 0; i < g.ops.size(); ++i) {
            const Op& op = g.ops[i];
            if (op.kind != OP_CONV || op.out_level < 0 || impl != UNET_IMPL_AUTO) continue;
            ConvGeom cg = op_geom_of(op);
            if (head_supported(cg, op.nsrc)) head_op_off[i] = take(head_bwd_scratch_bytes(cg));
        }
        // every matrix-core weight gradient keeps a slab region of its own until the batched reduce of the backward (part)
        wz_off.assign(g.ops.size(), SIZE_MAX);
        for (size_t i = 0; i < g.ops.size(); ++i) {
            const Op& op = g.ops[i];
            if ((op.kind != OP_CONV && op.kind != OP_CONVT) || impl != UNET_IMPL_AUTO || op.out_level >= 0) continue;
            ConvGeom cg = op_geom_of(op);
            SrcDesc sd[2];
            for (int k = 0; k < op.nsrc; ++k) sd[k].C = g.tensors[op.src[k]].C;
            if (op.kind == OP_CONV && conv_first_wgrad_mfma_supported(dtype, cg, sd, op.nsrc)) wz_off[i] = take(conv_first_wgrad_mfma_scratch_bytes(cg));
            else if (op.kind == OP_CONV && wgrad_mfma[i]) { if (!mfma_conv_wgrad_direct(cg)) wz_off[i] = take(mfma_wgrad_scratch_bytes(cg, side_polite[i])); }
            else if (op.kind == OP_CONVT && wgrad_mfma[i]) { if (!mfma_convt_wgrad_direct(cg)) wz_off[i] = take(mfma_convt_wgrad_scratch_bytes(cg)); }
        }
        ws_bytes = off;
        // batched filter pack: one job per MFMA filter pack, sources as offsets into a flat parameter buffer
        p_off.assign(g.params.size() + 1, 0);
        for (size_t i = 0; i < g.params.size(); ++i) {
            int64_t n = 1;
            for (auto d : g.params[i].shape) n *= d;
            p_off[i + 1] = p_off[i] + n;
        }
        wz_jobs.clear();
        wz_job_of_op.assign(g.ops.size(), -1);
        int wz_blk = 0;
        for (size_t i = 0; i < g.ops.size(); ++i) {
            if (wz_off[i] == SIZE_MAX) continue;
            const Op& op = g.ops[i];
            ConvGeom cg = op_geom_of(op);
            WgradReduceJob j;
            j.op = (int)i;
            j.slab_off = (long long)(wz_off[i] / 4);
            j.dw_off = p_off[op.weight];
            j.Cb = op.cout;
            if (op.kind == OP_CONVT) {   // the bias partials (sums of dy over the fine grid) sit behind the slab, [nsplit][Cout]
                j.nsplit = mfma_convt_wgrad_splits(cg);
                j.n = (long long)8 * op.cin * op.cout;
                j.bias_off = op.bias >= 0 ? j.slab_off + (long long)j.nsplit * j.n : -1;
                j.db_off = op.bias >= 0 ? p_off[op.bias] : -1;
            } else {
                j.nsplit = (op.cin == 1) ? conv_first_wgrad_splits(cg) : mfma_conv_wgrad_splits(cg, side_polite[i]);
                j.n = (long long)27 * op.cin * op.cout;
                j.bias_off = op.bias >= 0 ? j.slab_off + (long long)j.nsplit * j.n : -1;
                j.db_off = op.bias >= 0 ? p_off[op.bias] : -1;
            }
            wz_blk += wgrad_reduce_job_blocks(j, wz_blk);
            wz_job_of_op[i] = (int)wz_jobs.size();
            wz_jobs.push_back(j);
        }
        pack_jobs.clear();
        pack_blocks = 0;
        pack_split_op = -1; pack_split_blocks = 0; pack_fwd_blocks = 0;
        for (int which = 0; which < 2; ++which) {          // forward packs first, then the dgrad packs
            for (size_t i = 0; i < g.ops.size(); ++i) {
                const Op& op = g.ops[i];
                if ((op.kind != OP_CONV && op.kind != OP_CONVT) || !use_mfma[i]) continue;
                // first matrix-core op whose output is 16^3 voxels or smaller: its forward pack and the later ops' go into the second launch
                if (which == 0 && pack_split_op < 0 && g.tensors[op.dst].voxels() <= (int64_t)16 * 16 * 16) { pack_split_op = (int)i; pack_split_blocks = pack_blocks; }
                PackJob jb[2];
                int n = op.kind == OP_CONV ? mfma_conv_pack_jobs(op_geom_of(op), dgrad_mfma[i] != 0, jb) : mfma_convt_pack_jobs(op_geom_of(op), jb);
                if (which >= n) continue;
                jb[which].src_off = p_off[op.weight];
                jb[which].dst_off = (int64_t)(which == 0 ? wm_fwd[i] : wm_dgrad[i]);
                jb[which].blk0 = pack_blocks;
                pack_blocks += jb[which].total;
                pack_jobs.push_back(jb[which]);
            }
            if (which == 0) pack_fwd_blocks = pack_blocks;
        }
    }
};

namespace {

struct Exec {
    const unet_plan& p;
    char* ws;
    hipStream_t s;
    // the deep levels' kernels may run in this call: the workspace's arrival counters are known to be zero (cleared by the batched filter
    // pack of this forward, or of the forward whose packs / activations this call works on)
    bool deep_on = false;
    Exec(const unet_plan& plan, void* workspace, void* stream) : p(plan), ws((char*)workspace), s((hipStream_t)stream) {}
    DeepScratch deep() const {
        DeepScratch d;
        if (deep_on && p.deep_part_bytes) { d.part = (float*)(ws + p.deep_part_off); d.part_bytes = p.deep_part_bytes; d.cnt = (int*)(ws + p.deep_cnt_off); d.ncnt = p.deep_ncnt; }
        return d;
    }
    int* deep_cnt() const { return p.deep_part_bytes ? (int*)(ws + p.deep_cnt_off) : nullptr; }

    void* tptr(int t) const { return ws + p.t_off[t]; }
    void* gptr(int t) const { return p.g_off[t] == SIZE_MAX ? nullptr : ws + p.g_off[t]; }
    float* stat(int n) const { return (float*)(ws + p.n_stat[n]); }
    float* coef(int n) const { return (float*)(ws + p.n_coef[n]); }
    float* partial() const { return (float*)(ws + p.partial_off); }

    // the raw tensor with its recorded norm + activation applied by the reader
    SrcDesc raw_src(int t) const {
        const Tensor& T = p.g.tensors[t];
        SrcDesc d;
        d.ptr = tptr(t); d.C = T.C; d.act = T.act;
        if (T.norm >= 0) { d.scale = stat(T.norm) + 2 * T.C; d.shift = stat(T.norm) + 3 * T.C; }
        return d;
    }
    // what consumers read: the activated copy when the plan keeps one (no per-read transform), else the raw tensor + transform
    SrcDesc src(int t) const {
        if (p.a_off[t] == SIZE_MAX) return raw_src(t);
        SrcDesc d;
        d.ptr = ws + p.a_off[t]; d.C = p.g.tensors[t].C;
        return d;
    }
    // after the producer (and its norm statistics) are done: write the activated copy
    void apply_view(int t) const {
        if (p.a_off[t] != SIZE_MAX) launch_apply_view(p.dtype, raw_src(t), ws + p.a_off[t], p.g.tensors[t].voxels(), s);
    }
    ConvGeom geom(const Op& op) const {
        const Tensor& a = p.g.tensors[op.src[0]];
        const Tensor& o = p.g.tensors[op.dst];
        ConvGeom g;
        g.Cin = op.cin; g.Cout = op.cout; g.D = a.D; g.H = a.H; g.W = a.W; g.Do = o.D; g.Ho = o.H; g.Wo = o.W;
        g.ks = op.ks; g.stride = op.stride;
        return g;
    }

    // on_head(level): called right after the launches that produce results[level] (a fused forward + loss issues that level's loss there)
    void forward(const float* const* params, float* const* buffers, const float* x, float* const* outs, int mode,
                 const std::function<void(int)>* on_head = nullptr) {
        const Graph& g = p.g;
        // UNET_MODE_PACKS_CURRENT: the filter packs this workspace holds were made from these parameter values (an earlier mode-1
        // forward on it since the last update): micro-steps 2..batch_size of an optimizer step skip the ~0.1 ms / 190 MB repack
        const bool packs_current = (mode & UNET_MODE_PACKS_CURRENT) != 0;
        mode &= 1;
        std::vector<int> fused_blocks(g.norms.size(), 0);   // > 0: the producing conv already wrote the statistics partials
        std::vector<char> fused_dbl(g.norms.size(), 0);     // ... as fp64 rows (the fp32 engine)
        std::vector<char> fused_done(g.norms.size(), 0);    // the producing conv ran the whole norm layer in its epilogue (kernels_mfma_deep.hip)
        // parameters in one flat contiguous buffer (the hosts allocate them so): every MFMA filter pack in ONE launch
        bool packed = false, pack_pending = false, pack2_pending = false;
        if (p.jobs_dev && packs_current) { packed = true; deep_on = true; }
        else if (p.jobs_dev) {
            bool flat = true;
            for (size_t i = 0; i < g.params.size() && flat; ++i) flat = params[i] == params[0] + p.p_off[i];
            if (flat) {
                // training forward: the pack runs on the plan's side stream beside the input pack, the first conv (which reads the
                // fp32 filter) and its norm; the first kernel that needs packed filters waits for it.  (Eval forwards stay on the
                // caller's stream: they are re-entrant per workspace, the side stream and its events are per plan.)
                static const bool no_side = getenv("UNET_NO_SIDE_STREAM") != nullptr;
                const int njobs = (int)p.pack_jobs.size();
                if (mode == 1 && p.side && !no_side && !g_prof) {
                    HIP_OK(hipEventRecord(p.ev_fork, s));
                    HIP_OK(hipStreamWaitEvent(p.side, p.ev_fork, 0));
                    if (p.pack_split_op > 0 && p.pack_split_blocks > 0) {
                        // three launches: the top levels' forward packs (a few hundred KB: the first MFMA conv waits for these only), the
                        // other forward packs, the dgrad packs (nothing reads them before the backward).  Measured and not kept: a bounded
                        // grid for the later launches (no gain), the dgrad packs launched when the caller's stream reaches the 16^3 level
                        // (2.925-2.935 ms against 2.915: their 5800 short blocks delay the small levels' latency-bound kernels by more
                        // than they cost the bandwidth-bound ones), one launch for everything (a 20-us bubble in front of the first MFMA conv).
                        launch_mfma_pack_batched(params[0], ws, p.jobs_dev, njobs, p.pack_split_blocks, p.side, 0, 0, deep_cnt(), p.deep_ncnt);
                        HIP_OK(hipEventRecord(p.ev_join, p.side));
                        launch_mfma_pack_batched(params[0], ws, p.jobs_dev, njobs, p.pack_fwd_blocks - p.pack_split_blocks, p.side, p.pack_split_blocks);
                        HIP_OK(hipEventRecord(p.ev_pack, p.side));
                        pack2_pending = true;
                        launch_mfma_pack_batched(params[0], ws, p.jobs_dev, njobs, p.pack_blocks - p.pack_fwd_blocks, p.side, p.pack_fwd_blocks);
                        HIP_OK(hipEventRecord(p.ev_packd, p.side));
                    } else {
                        launch_mfma_pack_batched(params[0], ws, p.jobs_dev, njobs, p.pack_blocks, p.side, 0, 0, deep_cnt(), p.deep_ncnt);
                        HIP_OK(hipEventRecord(p.ev_join, p.side));
                    }
                    pack_pending = true;
                } else {
                    // an inference forward never reads a dgrad pack
                    ProfScope ps(-1, UNET_PROF_OTHER, s);
                    launch_mfma_pack_batched(params[0], ws, p.jobs_dev, njobs, mode == 1 ? p.pack_blocks : p.pack_fwd_blocks, s, 0, 0, deep_cnt(), p.deep_ncnt);
                }
                packed = true;
                deep_on = true;
            }
        }
        auto need_packs = [&](int op_index = 1 << 30) {
            if (pack_pending) { HIP_OK(hipStreamWaitEvent(s, p.ev_join, 0)); pack_pending = false; }
            if (pack2_pending && op_index >= p.pack_split_op) { HIP_OK(hipStreamWaitEvent(s, p.ev_pack, 0)); pack2_pending = false; }
        };
        for (size_t i = 0; i < g.ops.size(); ++i) {
            const Op& op = g.ops[i];
            ProfScope ps((int)i, (op.kind == OP_CONV || op.kind == OP_CONVT) ? UNET_PROF_CONV_FWD : op.kind == OP_NORM ? UNET_PROF_NORM_FWD : UNET_PROF_OTHER, s);
            switch (op.kind) {
                case OP_PACK_INPUT:
                    launch_pack_input(p.dtype, x, tptr(op.dst), g.in_c, g.tensors[op.dst].voxels(), s);
                    break;
                case OP_CONV:
                case OP_CONVT: {
                    if (op.out_level >= 0 && !(outs && outs[op.out_level])) break;  // result not wanted
                    SrcDesc sd[2] = {src(op.src[0]), op.nsrc > 1 ? src(op.src[1]) : SrcDesc()};
                    ConvGeom cg = geom(op);
                    float* wf = (float*)(ws + p.w_fwd[i]);
                    float* wd = (float*)(ws + p.w_dgrad[i]);
                    if (p.use_mfma[i]) need_packs((int)i);
                    if (op.kind == OP_CONV && p.use_mfma[i]) {
                        if (!packed)
                            launch_mfma_pack_conv_w(params[op.weight], ws + p.wm_fwd[i],
                                                    (mode == 1 && p.dgrad_mfma[i]) ? ws + p.wm_dgrad[i] : nullptr, cg, s);
                        if (mode == 1 && !p.dgrad_mfma[i] && !packs_current)
                            launch_pack_conv_w(params[op.weight], wf, wd, op.cin, op.cout, op.ks * op.ks * op.ks, s);
                        const Tensor& T = g.tensors[op.dst];
                        bool want_stats = T.norm >= 0 && !(g.norms[T.norm].batch && mode == 0);
                        // the deep levels: split-K kernel, the norm layer behind the conv in its epilogue (statistics, running statistics,
                        // activated copy: the OP_NORM that follows has nothing left to do)
                        if (deep_on && deep_conv_applies(p.dtype, T.voxels(), op.cin, op.cout)) {
                            DeepNormFwd nf;
                            const bool fuse = T.norm >= 0 && p.a_off[op.dst] != SIZE_MAX;
                            if (fuse) {
                                const Norm& n = g.norms[T.norm];
                                nf = {params[n.gamma], params[n.beta], n.eps, stat(T.norm), n.batch ? buffers[n.buffer] : nullptr,
                                      n.batch ? buffers[n.buffer + 1] : nullptr, 0.1, (n.batch && mode == 0) ? 1 : 0, T.act, ws + p.a_off[op.dst]};
                            }
                            if (launch_deep_conv_fwd(cg, sd, op.nsrc, ws + p.wm_fwd[i], params[op.bias], tptr(op.dst), fuse ? &nf : nullptr, deep(), s)) {
                                if (fuse) fused_done[T.norm] = 1;
                                break;
                            }
                        }
                        int rows = launch_mfma_conv_fwd(cg, sd, op.nsrc, ws + p.wm_fwd[i], params[op.bias], tptr(op.dst),
                                                        want_stats ? partial() : nullptr, s);
                        if (want_stats) fused_blocks[T.norm] = rows;
                    } else if (op.kind == OP_CONV && p.impl == UNET_IMPL_AUTO && op.out_level >= 0 && head_supported(cg, op.nsrc)) {
                        // a head: results[level] straight from the source tensor (the channels-last copy only if nobody asked for the level)
                        float* o = outs[op.out_level];
                        launch_head_fwd(p.dtype, cg, sd[0], params[op.weight], params[op.bias], o ? nullptr : tptr(op.dst), o, s);
                    } else if (op.kind == OP_CONV && p.impl == UNET_IMPL_AUTO && op.out_level < 0 &&
                               conv_first_mfma_supported(p.dtype, cg, sd, op.nsrc)) {
                        const Tensor& T = g.tensors[op.dst];
                        bool want_stats = T.norm >= 0 && !(g.norms[T.norm].batch && mode == 0);
                        int rows = launch_conv_first_mfma(cg, sd, params[op.weight], params[op.bias], tptr(op.dst),
                                                          want_stats ? partial() : nullptr, s);
                        if (want_stats) fused_blocks[T.norm] = rows;
                        // the fp32 [tap][cin][cout] copies are read by the direct dgrad only: not made when the input needs no gradient
                        if (mode == 1 && !packs_current && g.tensors[op.src[0]].needs_grad)
                            launch_pack_conv_w(params[op.weight], wf, wd, op.cin, op.cout, op.ks * op.ks * op.ks, s);
                    } else if (op.kind == OP_CONV && p.impl == UNET_IMPL_AUTO && op.out_level < 0 &&
                               conv_first_f32_mfma_supported(p.dtype, cg, sd, op.nsrc)) {
                        // fp32 engine, Cin = 1: the first conv on the fp32 matrix cores (filter read in torch layout), statistics in its epilogue
                        const Tensor& T = g.tensors[op.dst];
                        const bool want_stats = T.norm >= 0 && !(g.norms[T.norm].batch && mode == 0);
                        const int rows = launch_conv_first_f32_mfma(cg, sd, params[op.weight], params[op.bias], (float*)tptr(op.dst),
                                                                    want_stats ? (double*)partial() : nullptr, s);
                        if (want_stats) { fused_blocks[T.norm] = rows; fused_dbl[T.norm] = 1; }
                        if (mode == 1 && !packs_current) launch_pack_conv_w(params[op.weight], wf, wd, op.cin, op.cout, op.ks * op.ks * op.ks, s);
                    } else if (op.kind == OP_CONV && p.impl == UNET_IMPL_AUTO && op.out_level < 0 &&
                               conv_f32_mfma_supported(p.dtype, cg, sd, op.nsrc)) {
                        // fp32 engine: the same IEEE fp32 products and sums as the VALU kernel below, on the fp32 matrix cores
                        if (!packs_current) launch_pack_conv_w(params[op.weight], wf, wd, op.cin, op.cout, op.ks * op.ks * op.ks, s);
                        const Tensor& T = g.tensors[op.dst];
                        const bool want_stats = T.norm >= 0 && !(g.norms[T.norm].batch && mode == 0);
                        const int rows = launch_conv_f32_mfma(cg, sd, op.nsrc, wf, params[op.bias], (float*)tptr(op.dst), s,
                                                              want_stats ? (double*)partial() : nullptr);
                        if (want_stats) { fused_blocks[T.norm] = rows; fused_dbl[T.norm] = 1; }
                    } else if (op.kind == OP_CONV) {
                        if (!packs_current) launch_pack_conv_w(params[op.weight], wf, wd, op.cin, op.cout, op.ks * op.ks * op.ks, s);
                        launch_conv_fwd_direct(p.dtype, cg, sd, op.nsrc, wf, params[op.bias], tptr(op.dst),
                                               op.out_level >= 0 ? outs[op.out_level] : nullptr, s);
                    } else if (p.use_mfma[i]) {
                        if (!packed) launch_mfma_pack_convt_w(params[op.weight], ws + p.wm_fwd[i], mode == 1 ? ws + p.wm_dgrad[i] : nullptr, cg, s);
                        if (!(deep_on && launch_deep_convt_fwd(cg, sd, op.nsrc, ws + p.wm_fwd[i], params[op.bias], tptr(op.dst), deep(), s)))
                            launch_mfma_convt_fwd(cg, sd, op.nsrc, ws + p.wm_fwd[i], params[op.bias], tptr(op.dst), s);
                    } else {
                        if (!packs_current) launch_pack_convt_w(params[op.weight], wf, wd, op.cin, op.cout, s);
                        if (p.impl == UNET_IMPL_AUTO && convt_f32_mfma_supported(p.dtype, cg, sd, op.nsrc))
                            launch_convt_f32_mfma(cg, sd, wf, params[op.bias], (float*)tptr(op.dst), s);
                        else
                            launch_convt_fwd_direct(p.dtype, cg, sd, op.nsrc, wf, params[op.bias], tptr(op.dst), s);
                    }
                    break;
                }
                case OP_NORM: {
                    const Norm& n = g.norms[op.norm];
                    const Tensor& T = g.tensors[n.tensor];
                    if (fused_done[op.norm]) break;
                    if (n.batch && mode == 0) {
                        launch_norm_eval(n.C, params[n.gamma], params[n.beta], buffers[n.buffer], buffers[n.buffer + 1], n.eps,
                                         stat(op.norm), s);
                    } else {
                        int nb = fused_blocks[op.norm];
                        bool dbl = fused_dbl[op.norm] != 0;   // fp32 tensors leave fp64 block partials (k_stats_partial, k_conv_f32_mfma)
                        if (!nb) {
                            launch_stats_partial(p.dtype, tptr(n.tensor), n.C, T.voxels(), partial(), s);
                            nb = stats_blocks(T.voxels());
                            dbl = p.dtype == UNET_DTYPE_F32;
                        }
                        // few partial rows (32^3 and deeper): finalize + activated copy in one launch
                        if (!dbl && p.a_off[n.tensor] != SIZE_MAX &&
                            launch_norm_finalize_apply(p.dtype, partial(), nb, n.C, T.voxels(), params[n.gamma], params[n.beta], n.eps,
                                                       stat(op.norm), n.batch ? buffers[n.buffer] : nullptr,
                                                       n.batch ? buffers[n.buffer + 1] : nullptr, 0.1, tptr(n.tensor), T.act,
                                                       ws + p.a_off[n.tensor], s))
                            break;
                        launch_norm_finalize(partial(), nb, n.C, T.voxels(), params[n.gamma], params[n.beta],
                                             n.eps, stat(op.norm), n.batch ? buffers[n.buffer] : nullptr,
                                             n.batch ? buffers[n.buffer + 1] : nullptr, 0.1, s, dbl);
                    }
                    apply_view(n.tensor);
                    break;
                }
                case OP_MATERIALIZE: {
                    SrcDesc sd[2] = {src(op.src[0]), op.nsrc > 1 ? src(op.src[1]) : SrcDesc()};
                    launch_materialize(p.dtype, sd, op.nsrc, tptr(op.dst), g.tensors[op.dst].voxels(), s);
                    break;
                }
                case OP_MAXPOOL: {
                    const Tensor& a = g.tensors[op.src[0]];
                    launch_maxpool_fwd(p.dtype, src(op.src[0]), tptr(op.dst), a.D, a.H, a.W, s);
                    break;
                }
                case OP_UPSAMPLE: {
                    const Tensor& a = g.tensors[op.src[0]];
                    launch_upsample_fwd(p.dtype, src(op.src[0]), tptr(op.dst), a.D, a.H, a.W, s);
                    break;
                }
                case OP_EXPORT:
                    if (outs && outs[op.out_level])
                        launch_export(p.dtype, src(op.src[0]), outs[op.out_level], g.tensors[op.src[0]].voxels(), s);
                    break;
            }
            // an activation recorded on a tensor that has no norm (e.g. "conv8,relu"): its copy is due right after the producer
            if (op.kind != OP_NORM && op.kind != OP_EXPORT && op.dst >= 0) {
                const Tensor& T = g.tensors[op.dst];
                if (T.norm < 0 && T.act != ACT_NONE) apply_view(op.dst);
            }
            if (on_head && op.out_level >= 0 && outs && outs[op.out_level] && (op.kind == OP_CONV || op.kind == OP_CONVT || op.kind == OP_EXPORT))
                (*on_head)(op.out_level);
        }
        need_packs();   // nothing consumed the packs (no MFMA op): still order the caller's stream after the side stream
    }

    // g[t] holds dL/d(view of t); turn it into dL/d(raw t) (and accumulate the norm's affine gradients)
    // no_apply: stop after the finalize (coef(T.norm) is final; dgamma / dbeta accumulated) -- the element-wise pass is fused into the one
    // consumer of dL/d(raw t) (the first conv's weight gradient)
    void view_backward(int t, const float* const* params, float* const* gparams, bool no_apply = false) {
        const Tensor& T = p.g.tensors[t];
        if (T.norm >= 0) {
            const Norm& n = p.g.norms[T.norm];
            // the statistics pass, unless the dgrad that produced this gradient already left its partial rows (bn_rows_tensor == t)
            int rows = stats_blocks(T.voxels()), have = 0;
            {
                std::lock_guard<std::mutex> lk(p.bn_mu);
                auto d = p.bn_done.find(ws);
                if (d != p.bn_done.end()) {
                    auto f = std::find(d->second.begin(), d->second.end(), t);
                    if (f != d->second.end()) { d->second.erase(f); return; }    // dL/d(raw), the affine gradients and coef() are already there
                }
                auto it = p.bn_pending.find(ws);
                if (it != p.bn_pending.end()) { if (it->second.first == t) have = it->second.second; p.bn_pending.erase(it); }
            }
            if (have > 0) rows = have;
            else launch_norm_bwd_partial(p.dtype, gptr(t), tptr(t), T.C, T.voxels(), stat(T.norm), T.act, partial(), s);
            if (!no_apply && launch_norm_bwd_finalize_apply(p.dtype, partial(), rows, T.C, T.voxels(), params[n.gamma], stat(T.norm),
                                                            coef(T.norm), gparams[n.gamma], gparams[n.beta], gptr(t), tptr(t), T.act, s))
                return;
            launch_norm_bwd_finalize(partial(), rows, T.C, T.voxels(), params[n.gamma], stat(T.norm), coef(T.norm),
                                     gparams[n.gamma], gparams[n.beta], s, p.dtype == UNET_DTYPE_F32);
            if (no_apply) return;
            launch_norm_bwd_apply(p.dtype, gptr(t), tptr(t), T.C, T.voxels(), stat(T.norm), coef(T.norm), T.act, s);
        } else if (T.act != ACT_NONE) {
            launch_act_bwd(p.dtype, gptr(t), tptr(t), T.act, T.numel(), s);
        }
    }

    // Ops [op_lo, op_hi) are run (in reverse); ops >= op_hi are only replayed on the host (dry) to rebuild which gradient buffers
    // already hold a value: that state depends on the graph and on which grad_outs are given, not on earlier calls, so the
    // backward can be issued in parts (unet_backward_part) with collectives of finished gradient buckets in between.
    void backward(const float* const* params, const float* const* grad_outs, float* const* gparams, float* grad_x, int op_hi = 1 << 30,
                  int op_lo = 0) {
        const Graph& g = p.g;
        // the dgrad filter packs of this step were launched on the side stream in the middle of the forward.  The wait is unconditional:
        // the packs belong to a (workspace, stream) pair but the event is the plan's, and with two workspaces interleaved
        // (A.forward, B.forward, B.backward, A.backward) a consumed-once flag let A's backward run ahead of A's packs.  The event's latest
        // record is behind every earlier pack on the side stream, and waiting on a never-recorded or completed event costs nothing.
        if (p.side) HIP_OK(hipStreamWaitEvent(s, p.ev_packd, 0));
        {   // the deep levels' kernels: only behind a forward that made its packs with the batched launch (which cleared the counters)
            bool flat = p.jobs_dev != nullptr;
            for (size_t k = 0; k < g.params.size() && flat; ++k) flat = params[k] == params[0] + p.p_off[k];
            deep_on = flat;
        }
        std::vector<char> init(g.tensors.size(), 0);
        auto dst_of = [&](int t) {
            DstGrad d;
            d.C = g.tensors[t].C;
            d.ptr = g.tensors[t].needs_grad ? gptr(t) : nullptr;
            d.accumulate = init[t];
            return d;
        };
        auto mark = [&](const Op& op) {
            for (int k = 0; k < op.nsrc; ++k)
                if (g.tensors[op.src[k]].needs_grad) init[op.src[k]] = 1;
        };
        // sb: where the parameter-gradient kernels go.  fork() orders them after everything issued so far on the caller's stream
        // (dL/d(raw output) of the layer is final); the join at the end orders the caller's stream after them.
        static const bool no_side = getenv("UNET_NO_SIDE_STREAM") != nullptr;
        const hipStream_t sb = (p.side && !no_side && !g_prof) ? p.side : s;
        auto fork = [&]() {
            if (sb == s) return;
            HIP_OK(hipEventRecord(p.ev_fork, s));
            HIP_OK(hipStreamWaitEvent(sb, p.ev_fork, 0));
        };
        if (op_lo < 0) op_lo = 0;
        if (op_hi >= (int)g.ops.size()) { std::lock_guard<std::mutex> lk(p.bn_mu); p.bn_pending.erase(ws); p.bn_done.erase(ws); }   // a new backward starts
        // gradients in one flat buffer (both hosts allocate them so): the sliding-window wgrads only write their slabs here and ONE
        // batched reduce at the end of this call adds them all into the gradients
        bool gflat = p.wz_jobs_dev != nullptr;
        for (size_t k = 0; k < g.params.size() && gflat; ++k) gflat = gparams[k] == gparams[0] + p.p_off[k];
        std::vector<char> wz_ran(p.wz_jobs.size(), 0);
        int wz_pending = 0;
        // slabs of the weight gradients launched so far -> gradients: one launch per run of consecutive jobs (normally one).  Flushed
        // every few layers, not only at the end: a single reduce of everything would sit behind the last layer on the side stream
        // and the caller's stream waits for it at the join.
        auto flush_wz = [&](hipStream_t st) {
            for (size_t j = 0; j < wz_ran.size();) {
                if (!wz_ran[j]) { ++j; continue; }
                size_t e = j;
                int nblk = 0;
                while (e < wz_ran.size() && wz_ran[e]) { nblk += p.wz_jobs[e].nblk; wz_ran[e] = 0; ++e; }
                ProfScope pr(-1, UNET_PROF_WGRAD, st);
                launch_wgrad_reduce_batched(p.wz_jobs_dev, (int)j, (int)(e - j), p.wz_jobs[j].blk0, nblk, ws, gparams[0], st);
                j = e;
            }
            wz_pending = 0;
        };
        // weight / bias gradient of op i (its dL/d(raw output) is final), enqueued on sb
        auto do_wgrad = [&](int i) {
            const Op& op = g.ops[i];
            const int t = op.dst;
            SrcDesc sd[2] = {src(op.src[0]), op.nsrc > 1 ? src(op.src[1]) : SrcDesc()};
            ConvGeom cg = geom(op);
            {   // the op's own bracket closes before flush_wz opens the batched reduce's (they would nest and count the reduce twice)
            ProfScope pw(i, UNET_PROF_WGRAD, sb);
            if (op.kind == OP_CONV) {
                if (p.impl == UNET_IMPL_AUTO && conv_first_wgrad_mfma_supported(p.dtype, cg, sd, op.nsrc)) {
                    const bool defer = gflat && p.wz_job_of_op[i] >= 0;
                    launch_conv_first_wgrad_mfma(cg, sd, gptr(t), gparams[op.weight], gparams[op.bias],
                                                 ws + (defer ? p.wz_off[i] : p.wgrad_off), sb, defer);
                    if (defer) { wz_ran[p.wz_job_of_op[i]] = 1; ++wz_pending; }
                } else if (p.wgrad_mfma[i] && p.wgrad_on_main[i] && p.wz_off[i] != SIZE_MAX) {
                    // experiment: kernel and its slab sum on the caller's stream (slab in the op's own region: the shared scratch is the side stream's)
                    launch_mfma_conv_wgrad(cg, sd, op.nsrc, gptr(t), gparams[op.weight], gparams[op.bias], ws + p.wz_off[i], s, false, 0);
                } else if (p.wgrad_mfma[i]) {
                    const bool defer = gflat && p.wz_job_of_op[i] >= 0;   // slab only: summed by the batched reduce below
                    launch_mfma_conv_wgrad(cg, sd, op.nsrc, gptr(t), gparams[op.weight], gparams[op.bias],
                                           ws + (p.wz_off[i] != SIZE_MAX ? p.wz_off[i] : p.wgrad_off), sb, defer, p.side_polite[i]);
                    if (defer) { wz_ran[p.wz_job_of_op[i]] = 1; ++wz_pending; }
                }
                else if (p.impl == UNET_IMPL_AUTO && wgrad_f32_mfma_supported(p.dtype, cg, sd, op.nsrc))
                    launch_wgrad_f32_mfma(cg, sd, op.nsrc, (const float*)gptr(t), gparams[op.weight], gparams[op.bias], ws + p.wgrad_off, sb);
                else if (p.impl == UNET_IMPL_AUTO && wgrad_small_supported(cg, op.nsrc))
                    launch_conv_wgrad_small(p.dtype, cg, sd, op.nsrc, gptr(t), gparams[op.weight], gparams[op.bias], ws + p.wgrad_off, sb);
                else
                    launch_conv_wgrad_direct(p.dtype, cg, sd, op.nsrc, gptr(t), gparams[op.weight], gparams[op.bias], ws + p.wgrad_off, sb);
            } else if (p.wgrad_mfma[i]) {
                const bool defer = gflat && p.wz_job_of_op[i] >= 0;
                launch_mfma_convt_wgrad(cg, sd, gptr(t), gparams[op.weight], ws + (defer ? p.wz_off[i] : p.wgrad_off), sb, defer, gparams[op.bias], p.side_polite[i]);
                if (defer) { wz_ran[p.wz_job_of_op[i]] = 1; ++wz_pending; }
            } else {
                launch_convt_wgrad_direct(p.dtype, cg, sd, op.nsrc, gptr(t), gparams[op.weight], gparams[op.bias], ws + p.wgrad_off, sb);
            }
            }
            if (wz_pending >= 3) flush_wz(sb);      // (every layer and one flush at the end both measured slower: profiles/r06 A/Bs)
        };
        // The backward starts at full resolution, where the caller's stream is bandwidth-bound, and then spends a long stretch in the
        // small levels, where it is launch-latency bound and the memory system idles.  The big weight gradients of the decoder's top
        // levels (everything they read stays in the workspace: each tensor has a gradient buffer of its own) are therefore HELD until
        // the caller's stream reaches the small levels and run beside those, instead of competing with the top levels' dgrad / norm
        // kernels for bandwidth.
        const int64_t hold_from = (int64_t)64 * 64 * 64, hold_below = (int64_t)32 * 32 * 32;
        std::vector<int> held;
        bool deep_seen = false;
        // An event record on the caller's stream is not free: the kernel behind it starts ~6 us late (time line of a step,
        // profiles/r07_timeline.txt: every dgrad that follows a fork).  The weight gradient of a layer can start any time after its
        // dL/d(raw output) is final, so forks are shared: layers wait in `pending` and ONE fork serves `fork_every` of them.
        constexpr int fork_every = 3;
        std::vector<int> pending;
        auto issue_pending = [&]() {
            if (pending.empty()) return;
            fork();
            for (int h : pending) do_wgrad(h);
            pending.clear();
        };
        for (int i = (int)g.ops.size() - 1; i >= op_lo; --i) {
            const Op& op = g.ops[i];
            const bool dry = i >= op_hi;
            if (op.kind == OP_NORM) continue;
            if (op.kind == OP_EXPORT) {
                if (grad_outs && grad_outs[op.out_level] && g.tensors[op.src[0]].needs_grad) {
                    if (!dry) launch_export_bwd(p.dtype, grad_outs[op.out_level], dst_of(op.src[0]), g.tensors[op.src[0]].voxels(), s);
                    mark(op);
                }
                continue;
            }
            int t = op.dst;
            if ((op.kind == OP_CONV) && op.out_level >= 0) {
                if (!(grad_outs && grad_outs[op.out_level])) continue;
                if (p.impl == UNET_IMPL_AUTO && head_supported(geom(op), op.nsrc)) {
                    // fused head backward: dL/dW, dL/db and dL/d(source view) in one pass over (source, dL/dresults[level])
                    DstGrad dgh = dst_of(op.src[0]);
                    ProfScope ph(dry ? -2 : i, UNET_PROF_OTHER, s);
                    if (!dry) {
                        // the slab sum that finishes the head's dW / db feeds nothing on the caller's chain: on the side stream (a slab region
                        // per head, so the next level's head does not overwrite rows that have not been summed yet)
                        const bool defer = sb != s && p.head_op_off[i] != SIZE_MAX;
                        char* hs = ws + (defer ? p.head_op_off[i] : p.head_off);
                        launch_head_bwd(p.dtype, geom(op), src(op.src[0]), grad_outs[op.out_level], nullptr, params[op.weight], dgh,
                                        gparams[op.weight], gparams[op.bias], hs, s, defer);
                        if (defer) { fork(); launch_head_bwd_reduce(geom(op), gparams[op.weight], gparams[op.bias], hs, sb); }
                    }
                    if (dgh.ptr) mark(op);
                    continue;
                }
                if (!dry) launch_import_grad(p.dtype, grad_outs[op.out_level], gptr(t), g.tensors[t].C, g.tensors[t].voxels(), 0, s);
                init[t] = 1;
            }
            if (!g.tensors[t].needs_grad || !init[t]) continue;
            // The network's first conv with a norm behind it: dL/d(raw output) is only read by its weight gradient (the input needs no
            // gradient), which applies the norm backward's element-wise pass itself (NormBwdFuse) -- see the OP_CONV case below
            bool first_fused = false;
            if (!dry && op.kind == OP_CONV && p.impl == UNET_IMPL_AUTO && p.dtype == UNET_DTYPE_BF16 && g.tensors[t].norm >= 0 && op.nsrc == 1 &&
                !g.tensors[op.src[0]].needs_grad && sb != s && p.wz_off[i] != SIZE_MAX && g.tensors[t].C % 8 == 0) {
                static const bool off = getenv("UNET_NO_FIRST_WGRAD_FUSE") != nullptr;
                SrcDesc sd0 = src(op.src[0]);
                first_fused = !off && conv_first_wgrad_mfma_supported(p.dtype, geom(op), &sd0, 1);
            }
            if (!dry) { ProfScope ps(i, UNET_PROF_NORM_BWD, s); view_backward(t, params, gparams, first_fused); }
            switch (op.kind) {
                case OP_CONV:
                case OP_CONVT: {
                    SrcDesc sd[2] = {src(op.src[0]), op.nsrc > 1 ? src(op.src[1]) : SrcDesc()};
                    DstGrad dg[2] = {dst_of(op.src[0]), op.nsrc > 1 ? dst_of(op.src[1]) : DstGrad()};
                    ConvGeom cg = geom(op);
                    const float* wd = (const float*)(ws + p.w_dgrad[i]);
                    bool any = dg[0].ptr || (op.nsrc > 1 && dg[1].ptr);
                    // parameter gradients: on the side stream now, or held back (see `held`)
                    // The network's first conv (its input needs no gradient: nothing follows on the caller's stream) -- its weight gradient is
                    // the last kernel of the backward whichever stream it is on; on the caller's stream it starts without waiting for a fork
                    // and its reduce is not behind the side stream's queue.  Slab in the op's own region (the shared scratch is the side stream's).
                    if (!dry && !any && sb != s && op.kind == OP_CONV && p.impl == UNET_IMPL_AUTO && p.wz_off[i] != SIZE_MAX &&
                        conv_first_wgrad_mfma_supported(p.dtype, cg, sd, op.nsrc)) {
                        ProfScope pw(i, UNET_PROF_WGRAD, s);
                        const Tensor& To = g.tensors[t];
                        NormBwdFuse nf = {tptr(t), first_fused ? stat(To.norm) : nullptr, first_fused ? coef(To.norm) : nullptr, To.act};
                        launch_conv_first_wgrad_mfma(cg, sd, gptr(t), gparams[op.weight], gparams[op.bias], ws + p.wz_off[i], s, false,
                                                     first_fused ? &nf : nullptr);
                    } else
                    if (!dry) {
                        const int64_t vox = (int64_t)cg.Do * cg.Ho * cg.Wo;
                        if (!deep_seen && vox <= hold_below && !held.empty()) {     // the small levels begin: the held launches run beside them
                            deep_seen = true;
                            pending.insert(pending.begin(), held.begin(), held.end());
                            held.clear();
                            pending.push_back(i);
                            issue_pending();
                        } else if (!deep_seen && sb != s && vox >= hold_from) held.push_back(i);
                        else {
                            // forks are shared among the small levels' layers only (many short launches, the side stream has slack); a layer of
                            // 32^3 voxels or more forks at once: held back, the encoder's last weight gradients would start after the caller's
                            // stream has finished and lengthen the tail of the step (time line r07f2: +60 us before the join)
                            pending.push_back(i);
                            if ((int)pending.size() >= fork_every || sb == s || vox >= hold_below) issue_pending();
                        }
                    }
                    if (op.kind == OP_CONV) {
                        ProfScope pd(i, UNET_PROF_DGRAD, s);
                        if (!dry && (any && p.dgrad_mfma[i])) {
                            // The source is a norm layer's view read by this conv alone: its gradient is complete when this dgrad
                            // has written it, so the statistics pass of that norm's backward (a second read of the gradient and of the
                            // raw tensor) moves into the dgrad's epilogue where the kernel has one (k_mfma_conv_z16).  The rows wait in
                            // partial(): the next thing the caller's stream runs is that tensor's view_backward.
                            const int ts = op.src[0];
                            const Tensor& Ts = g.tensors[ts];
                            BnBwdStats bn = {tptr(ts), Ts.norm >= 0 ? stat(Ts.norm) : nullptr, partial(), Ts.act, Ts.C};
                            // With several consumers the one with the lowest op index writes last -- accumulating -- and sees the complete
                            // gradient: for the skip tensors that is the stride-2 conv, whose dgrad (k_s2_scatter) fetches the old gradient
                            // and the raw tensor by LDS-DMA and has the epilogue too.  (Stride-1 kernels only take it when they WRITE.)
                            const bool can = op.nsrc == 1 && Ts.norm >= 0 && p.first_consumer[ts] == i && p.dtype == UNET_DTYPE_BF16;
                            int rows = 0, served = 0;
                            if (deep_on) {      // the deep levels: split-K kernel; as the last writer of a norm layer's view it runs that norm's backward too
                                DeepNormBwd nb;
                                if (can) {
                                    const Norm& n = g.norms[Ts.norm];
                                    nb = {tptr(ts), stat(Ts.norm), params[n.gamma], coef(Ts.norm), gparams[n.gamma], gparams[n.beta], Ts.act};
                                }
                                served = launch_deep_conv_dgrad(cg, gptr(t), ws + p.wm_dgrad[i], dg, op.nsrc, can ? &nb : nullptr, deep(), s);
                                if (served == 2) { std::lock_guard<std::mutex> lk(p.bn_mu); p.bn_done[ws].push_back(ts); }
                            }
                            if (!served) rows = launch_mfma_conv_dgrad(cg, gptr(t), ws + p.wm_dgrad[i], dg, op.nsrc, s, can ? &bn : nullptr);
                            if (rows > 0) { std::lock_guard<std::mutex> lk(p.bn_mu); p.bn_pending[ws] = {ts, rows}; }
                        }
                        else if (!dry && any && p.impl == UNET_IMPL_AUTO && conv_f32_mfma_dgrad_supported(p.dtype, cg, dg, op.nsrc))
                            launch_conv_f32_mfma_dgrad(cg, (const float*)gptr(t), wd, dg, op.nsrc, s);
                        else if (!dry && (any)) launch_conv_dgrad_direct(p.dtype, cg, gptr(t), wd, dg, op.nsrc, s);
                    } else {
                        ProfScope pd(i, UNET_PROF_DGRAD, s);
                        if (!dry && (any && p.dgrad_mfma[i])) {
                            const int ts = op.src[0];
                            const Tensor& Ts = g.tensors[ts];
                            const bool can = op.nsrc == 1 && Ts.norm >= 0 && p.first_consumer[ts] == i && p.dtype == UNET_DTYPE_BF16;
                            DeepNormBwd nb;
                            if (can) {
                                const Norm& n = g.norms[Ts.norm];
                                nb = {tptr(ts), stat(Ts.norm), params[n.gamma], coef(Ts.norm), gparams[n.gamma], gparams[n.beta], Ts.act};
                            }
                            int norm_done = 0;
                            if (deep_on && launch_deep_convt_dgrad(cg, gptr(t), ws + p.wm_dgrad[i], dg, op.nsrc, can ? &nb : nullptr, deep(), &norm_done, s)) {
                                if (norm_done) { std::lock_guard<std::mutex> lk(p.bn_mu); p.bn_done[ws].push_back(ts); }
                            } else
                                launch_mfma_convt_dgrad(cg, gptr(t), ws + p.wm_dgrad[i], dg, op.nsrc, s);
                        }
                        else if (!dry && (any)) launch_convt_dgrad_direct(p.dtype, cg, gptr(t), wd, dg, op.nsrc, s);
                    }
                    if (any) mark(op);
                    break;
                }
                case OP_MATERIALIZE: {
                    DstGrad dg[2] = {dst_of(op.src[0]), op.nsrc > 1 ? dst_of(op.src[1]) : DstGrad()};
                    if (dg[0].ptr || (op.nsrc > 1 && dg[1].ptr)) {
                        // the materialized tensor is act(norm(src)): its gradient passes to the view of src unchanged
                        if (!dry) launch_materialize_bwd(p.dtype, gptr(t), dg, op.nsrc, g.tensors[t].voxels(), s);
                        mark(op);
                    }
                    break;
                }
                case OP_MAXPOOL: {
                    const Tensor& a = g.tensors[op.src[0]];
                    DstGrad d = dst_of(op.src[0]);
                    if (d.ptr) { if (!dry) launch_maxpool_bwd(p.dtype, src(op.src[0]), gptr(t), d, a.D, a.H, a.W, s); mark(op); }
                    break;
                }
                case OP_UPSAMPLE: {
                    const Tensor& a = g.tensors[op.src[0]];
                    DstGrad d = dst_of(op.src[0]);
                    if (d.ptr) { if (!dry) launch_upsample_bwd(p.dtype, gptr(t), d, a.D, a.H, a.W, s); mark(op); }
                    break;
                }
                case OP_PACK_INPUT:
                    if (!dry && (grad_x)) launch_unpack_ncdhw(p.dtype, gptr(t), grad_x, g.in_c, g.tensors[t].voxels(), s);
                    break;
                default: break;
            }
        }
        pending.insert(pending.begin(), held.begin(), held.end());
        held.clear();
        issue_pending();
        flush_wz(sb);
        if (sb != s) {   // join: whatever the caller enqueues next (optimizer step, next forward) sees every gradient
            HIP_OK(hipEventRecord(p.ev_join, sb));
            HIP_OK(hipStreamWaitEvent(s, p.ev_join, 0));
        }
    }
};

void check_launch() { HIP_OK(hipGetLastError()); }

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* unet_last_error(void) { return g_err.c_str(); }
void unet_set_error(const char* msg) { g_err = msg ? msg : ""; }   // comm.cpp reports through the same thread-local message

int unet_init(int* n_devices) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { if (n_devices) *n_devices = 0; return fail(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    if (n_devices) *n_devices = n;
    return 0;
}

int unet_device_info(int device, char* name, size_t name_len, size_t* total_mem, int* compute_units, int* is_gfx950) {
    hipDeviceProp_t pr;
    hipError_t e = hipGetDeviceProperties(&pr, device);
    if (e != hipSuccess) return fail(std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (name && name_len) { std::strncpy(name, pr.name, name_len - 1); name[name_len - 1] = 0; }
    if (total_mem) *total_mem = pr.totalGlobalMem;
    if (compute_units) *compute_units = pr.multiProcessorCount;
    if (is_gfx950) *is_gfx950 = std::strncmp(pr.gcnArchName, "gfx950", 6) == 0;
    return 0;
}

static bool create_cu_range_stream(int device, int cu_first, int cu_count, hipStream_t* out);
int unet_plan_create(const char* arch, int in_c, int out_c, int D, int H, int W, int dtype, int device, int impl, unet_plan** out) {
    if (!arch || !out) return fail("unet_plan_create: null argument");
    if (dtype != UNET_DTYPE_F32 && dtype != UNET_DTYPE_BF16) return fail("unet_plan_create: unknown dtype");
    unet_plan* p = nullptr;
    try {
        p = new unet_plan();
        p->g = Graph::build(arch, in_c, out_c, D, H, W);
        // a norm's affine gradients need the gradient of the tensor it sits on, even on the network input
        for (auto& n : p->g.norms) p->g.tensors[n.tensor].needs_grad = true;
        p->dtype = dtype; p->device = device; p->impl = impl; p->elsize = dtype == UNET_DTYPE_F32 ? 4 : 2;
        p->layout();
        // loss scratch: target pyramid + partials + per-level results
        {
            size_t off = 0;
            int oc = out_c;
            for (size_t l = 0; l < p->g.outputs.size(); ++l) {
                const auto& o = p->g.outputs[l];
                int64_t S = (int64_t)(D >> l) * (H >> l) * (W >> l);
                if (S <= 0) S = 1;
                off = align_up(off + (size_t)S * 8);                               // target level l (l >= 1)
                off = align_up(off + (size_t)(4 + 2 * oc) * 4);                    // level_out
                (void)o;
            }
            off = align_up(off + (size_t)1024 * (3 + 2 * oc) * 4);                // partials (level 0 ...)
            off = align_up(off + (size_t)1024 * (3 + 2 * oc) * 4);                // ... and the coarse levels, which may run on another stream
            p->loss_bytes = off;
        }
        // sgd segment table
        std::vector<SgdSeg> segs;
        int64_t o = 0;
        for (auto& pr : p->g.params) {
            int64_t n = 1;
            for (auto d : pr.shape) n *= d;
            SgdSeg sg;
            sg.offset = o; sg.count = n; sg.wd = pr.decay ? 1.f : 0.f;
            segs.push_back(sg);
            o += n;
        }
        p->n_param_elems = o;
        p->nseg = (int)segs.size();
        int nd = 0;
        if (hipGetDeviceCount(&nd) == hipSuccess && nd > 0) {
            DeviceGuard dg(device);
            int pr_least = 0, pr_greatest = 0;   // the side stream yields to the caller's (critical-path) stream
            (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
            // (a CU-masked side stream -- unet_plan_side_cu_range -- measured slower than occupancy politeness: profiles/r10d_ab_cu_mask.txt)
            HIP_OK(hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, pr_least));
            // the plan's events only order its two streams on ONE device: no host ever waits on them, so the system-scope fence a
            // default event performs when it is recorded (cache write-back for host visibility) is not needed
            const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
            HIP_OK(hipEventCreateWithFlags(&p->ev_fork, evf));
            HIP_OK(hipEventCreateWithFlags(&p->ev_join, evf));
            HIP_OK(hipEventCreateWithFlags(&p->ev_pack, evf));
            HIP_OK(hipEventCreateWithFlags(&p->ev_packd, evf));
            HIP_OK(hipMalloc((void**)&p->segs_dev, segs.size() * sizeof(SgdSeg)));
            HIP_OK(hipMemcpy(p->segs_dev, segs.data(), segs.size() * sizeof(SgdSeg), hipMemcpyHostToDevice));
            if (!p->wz_jobs.empty()) {
                HIP_OK(hipMalloc((void**)&p->wz_jobs_dev, p->wz_jobs.size() * sizeof(WgradReduceJob)));
                HIP_OK(hipMemcpy(p->wz_jobs_dev, p->wz_jobs.data(), p->wz_jobs.size() * sizeof(WgradReduceJob), hipMemcpyHostToDevice));
            }
            if (!p->pack_jobs.empty()) {
                HIP_OK(hipMalloc((void**)&p->jobs_dev, p->pack_jobs.size() * sizeof(PackJob)));
                HIP_OK(hipMemcpy(p->jobs_dev, p->pack_jobs.data(), p->pack_jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
            }
        }
        *out = p;
        return 0;
    } catch (const std::exception& e) {
        delete p;
        return fail(e.what());
    }
}

void unet_plan_destroy(unet_plan* p) { delete p; }

int unet_plan_param_count(const unet_plan* p, int* n) { *n = (int)p->g.params.size(); return 0; }
int unet_plan_param_shape(const unet_plan* p, int i, int64_t dims[5], int* ndim) {
    if (i < 0 || i >= (int)p->g.params.size()) return fail("parameter index out of range");
    const auto& s = p->g.params[i].shape;
    *ndim = (int)s.size();
    for (size_t k = 0; k < s.size(); ++k) dims[k] = s[k];
    return 0;
}
int unet_plan_param_name(const unet_plan* p, int i, char* name, size_t name_len) {
    if (i < 0 || i >= (int)p->g.params.size()) return fail("parameter index out of range");
    if (name && name_len) { std::strncpy(name, p->g.params[i].name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    return 0;
}
int unet_plan_param_decay(const unet_plan* p, int i, int* decay) {
    if (i < 0 || i >= (int)p->g.params.size()) return fail("parameter index out of range");
    *decay = p->g.params[i].decay;
    return 0;
}
int unet_plan_param_fan_in(const unet_plan* p, int i, int64_t* fan_in, int* is_norm_weight) {
    if (i < 0 || i >= (int)p->g.params.size()) return fail("parameter index out of range");
    *fan_in = p->g.params[i].fan_in; *is_norm_weight = p->g.params[i].norm_weight;
    return 0;
}
int unet_plan_buffer_count(const unet_plan* p, int* n) { *n = (int)p->g.buffers.size(); return 0; }
int unet_plan_buffer_shape(const unet_plan* p, int i, int64_t* numel) {
    if (i < 0 || i >= (int)p->g.buffers.size()) return fail("buffer index out of range");
    *numel = p->g.buffers[i];
    return 0;
}
int unet_plan_output_count(const unet_plan* p, int* n) { *n = (int)p->g.outputs.size(); return 0; }
int unet_plan_output_shape(const unet_plan* p, int l, int64_t dims[5]) {
    if (l < 0 || l >= (int)p->g.outputs.size()) return fail("output level out of range");
    const auto& o = p->g.outputs[l];
    dims[0] = 1; dims[1] = o.C; dims[2] = o.D; dims[3] = o.H; dims[4] = o.W;
    return 0;
}
int unet_plan_workspace_bytes(const unet_plan* p, size_t* bytes) { *bytes = p->ws_bytes; return 0; }
int unet_plan_flops(const unet_plan* p, double* fwd, double* bwd) { *fwd = p->g.flops_fwd; *bwd = p->g.flops_bwd; return 0; }
size_t unet_plan_describe(const unet_plan* p, char* buf, size_t len) {
    std::string d = p->g.describe();
    if (buf && len) { std::strncpy(buf, d.c_str(), len - 1); buf[len - 1] = 0; }
    return d.size() + 1;
}

int unet_forward(const unet_plan* p, const float* const* params, float* const* buffers, const float* x, float* const* outs,
                 void* workspace, int mode, void* stream) {
    try {
        if (!p || !params || !x || !workspace) throw std::runtime_error("unet_forward: null argument");
        if (mode & ~(1 | UNET_MODE_PACKS_CURRENT)) throw std::runtime_error("unet_forward: unknown mode bits (0 = eval, 1 = train, optionally | UNET_MODE_PACKS_CURRENT)");
        if (!p->g.buffers.empty() && !buffers) throw std::runtime_error("unet_forward: architecture has bnorm layers but buffers is null");
        DeviceGuard dg(p->device);
        Exec ex(*p, workspace, stream);
        ex.forward(params, buffers, x, outs, mode);
        check_launch();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int unet_backward(const unet_plan* p, const float* const* params, const float* const* grad_outs, float* const* grad_params,
                  float* grad_x, void* workspace, void* stream) {
    try {
        if (!p || !params || !grad_params || !workspace) throw std::runtime_error("unet_backward: null argument");
        if (grad_x) throw std::runtime_error("unet_backward: grad_x must be NULL -- dL/dx is not computed (the input carries no gradient, as in train.cpp:619-628)");
        DeviceGuard dg(p->device);
        Exec ex(*p, workspace, stream);
        ex.backward(params, grad_outs, grad_params, grad_x);
        check_launch();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int unet_backward_part(const unet_plan* p, const float* const* params, const float* const* grad_outs, float* const* grad_params,
                       float* grad_x, void* workspace, int op_hi, int op_lo, void* stream) {
    try {
        if (!p || !params || !grad_params || !workspace) throw std::runtime_error("unet_backward_part: null argument");
        if (op_lo < 0 || op_hi < op_lo) throw std::runtime_error("unet_backward_part: invalid op range");
        if (grad_x) throw std::runtime_error("unet_backward_part: grad_x must be NULL -- dL/dx is not computed");
        DeviceGuard dg(p->device);
        Exec ex(*p, workspace, stream);
        ex.backward(params, grad_outs, grad_params, grad_x, op_hi, op_lo);
        check_launch();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

// Buckets of the backward for overlapping the gradient all-reduce with it: bucket k = ops [op_lo[k], op_lo[k-1]) (op_lo[-1] = number
// of ops); when it has run, the gradients of the flat parameter elements [elem_lo[k], elem_lo[k-1]) are final.  A cut is made when the
// parameters finished since the last cut reach total/(max_buckets - 0.5) elements; the last bucket ends at op 0 / element 0.
int unet_plan_backward_buckets(const unet_plan* p, int max_buckets, int* n_buckets, int* op_lo, int64_t* elem_lo) {
    try {
        if (!p || !n_buckets || !op_lo || !elem_lo || max_buckets < 1) throw std::runtime_error("unet_plan_backward_buckets: bad argument");
        const Graph& g = p->g;
        const int nops = (int)g.ops.size();
        // first flat element of the parameters an op's backward finishes (conv/conv_trans: weight, bias and the norm layer recorded
        // on its output); INT64_MAX for ops without parameters
        std::vector<int64_t> first(nops, INT64_MAX);
        bool monotone = true;
        int64_t prev = -1;
        for (int i = 0; i < nops; ++i) {
            const Op& op = g.ops[i];
            if (op.kind != OP_CONV && op.kind != OP_CONVT) continue;
            int64_t f = p->p_off[op.weight];
            if (op.bias >= 0 && p->p_off[op.bias] < f) f = p->p_off[op.bias];
            first[i] = f;
            if (f <= prev) monotone = false;
            prev = f;
            const Tensor& T = g.tensors[op.dst];
            if (T.norm >= 0) {
                const Norm& n = g.norms[T.norm];
                if (p->p_off[n.gamma] < f || p->p_off[n.beta] < f) monotone = false;   // must lie behind the conv's own parameters
            }
        }
        int nb = 0;
        if (monotone && max_buckets > 1) {
            const double thr = (double)p->n_param_elems / ((double)max_buckets - 0.5);
            int64_t hi = p->n_param_elems;
            for (int i = nops - 1; i > 0 && nb < max_buckets - 1; --i) {
                if (first[i] == INT64_MAX) continue;
                if ((double)(hi - first[i]) >= thr) { op_lo[nb] = i; elem_lo[nb] = first[i]; hi = first[i]; ++nb; }
            }
        }
        op_lo[nb] = 0; elem_lo[nb] = 0; ++nb;
        *n_buckets = nb;
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int unet_plan_op_count(const unet_plan* p, int* n) { *n = (int)p->g.ops.size(); return 0; }
int unet_plan_op_info(const unet_plan* p, int i, int* kind, int* cin, int* cout, int* ks, int* stride, int64_t in_dims[3],
                      int64_t out_dims[3], char* name, size_t name_len) {
    if (i < 0 || i >= (int)p->g.ops.size()) return fail("op index out of range");
    const Op& op = p->g.ops[i];
    if (kind) *kind = (int)op.kind;
    if (cin) *cin = op.cin;
    if (cout) *cout = op.cout;
    if (ks) *ks = op.ks;
    if (stride) *stride = op.stride;
    const Tensor* a = op.nsrc > 0 && op.src[0] >= 0 ? &p->g.tensors[op.src[0]] : nullptr;
    const Tensor* o = op.dst >= 0 ? &p->g.tensors[op.dst] : nullptr;
    if (in_dims) { in_dims[0] = a ? a->D : 0; in_dims[1] = a ? a->H : 0; in_dims[2] = a ? a->W : 0; }
    if (out_dims) { out_dims[0] = o ? o->D : 0; out_dims[1] = o ? o->H : 0; out_dims[2] = o ? o->W : 0; }
    if (name && name_len) { std::strncpy(name, op.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    return 0;
}

int unet_profile_begin(void) {
    if (g_prof) return fail("unet_profile_begin: a profile is already open on this thread");
    g_prof = new ProfSink();
    return 0;
}
int unet_profile_end(int max_records, int* op_index, int* category, float* ms, int* n_records) {
    if (!g_prof) return fail("unet_profile_end: no open profile on this thread");
    ProfSink* ps = g_prof;
    g_prof = nullptr;
    int n = 0, rc = 0;
    for (auto& r : ps->recs) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(r.e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, r.e0, r.e1);
        if (e != hipSuccess) rc = fail(std::string("unet_profile_end: ") + hipGetErrorString(e));
        if (r.op != -2 && n < max_records && op_index && category && ms) { op_index[n] = r.op; category[n] = r.cat; ms[n] = t; ++n; }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    if (n_records) *n_records = n;
    delete ps;
    return rc;
}

int unet_loss_scratch_bytes(const unet_plan* p, size_t* bytes) { *bytes = p->loss_bytes; return 0; }

namespace {
// calc_losses + the deep-supervision loop (train.cpp:501-552,634-706) in pieces, so that a fused forward + loss can issue a level's
// kernels as soon as its head is computed, on another stream: prepare() = totals reset + target pyramid; level_partial /
// level_finish = one level.  Levels that may run concurrently must use different partial areas (`area`).
struct LossRun {
    const unet_plan* p;
    const float* const* outs; const int64_t* target; float* const* grad_outs; float* losses_out;
    int C, oc, collapse, cost_mask;
    float inv;
    std::vector<int64_t*> tgt;
    std::vector<float*> lvl;
    float* partial[2];
    LossRun(const unet_plan* plan, const float* const* outs_, const int64_t* target_, int cost_mask_, int collapse_before, float* const* grad_outs_,
            float* losses_out_, void* scratch)
        : p(plan), outs(outs_), target(target_), grad_outs(grad_outs_), losses_out(losses_out_), collapse(collapse_before), cost_mask(cost_mask_) {
        if (!p || !outs || !target || !losses_out || !scratch) throw std::runtime_error("unet_loss: null argument");
        const Graph& g = p->g;
        C = g.out_c;
        if (collapse_before < 0 || collapse_before >= C) throw std::runtime_error("invalid collapse_before");
        oc = collapse_before ? C - collapse_before + 1 : C;
        const size_t nl = g.outputs.size();
        float wsum = 0.f;
        for (size_t k = 0; k < nl; ++k) wsum += 1.0f / (float)(1 << k);
        inv = 1.0f / wsum;
        char* sc = (char*)scratch;
        size_t off = 0;
        tgt.resize(nl); lvl.resize(nl);
        for (size_t l = 0; l < nl; ++l) {
            int64_t S = (int64_t)(g.D >> l) * (g.H >> l) * (g.W >> l);
            if (S <= 0) S = 1;
            tgt[l] = (int64_t*)(sc + off); off = align_up(off + (size_t)S * 8);
            lvl[l] = (float*)(sc + off); off = align_up(off + (size_t)(4 + 2 * C) * 4);
        }
        partial[0] = (float*)(sc + off); off = align_up(off + (size_t)1024 * (3 + 2 * C) * 4);
        partial[1] = (float*)(sc + off);
        int D = g.D, H = g.H, W = g.W;
        for (size_t k = 0; k < nl; ++k) {      // the reference's run-time checks (train.cpp:651-652,664-671), before anything is launched
            if (k > 0) {
                if ((D >> 1) <= 0 || (H >> 1) <= 0 || (W >> 1) <= 0) throw std::runtime_error("deep supervision target size became zero");
                D >>= 1; H >>= 1; W >>= 1;
            }
            const auto& o = g.outputs[k];
            if (o.C == 0 || !outs[k]) throw std::runtime_error("undefined deep supervision output at level " + std::to_string(k));
            if (o.C != C)
                throw std::runtime_error("output channel mismatch at level " + std::to_string(k) + ": tensor has " + std::to_string(o.C) +
                                         ", out_count is " + std::to_string(C));
            if (o.D != D || o.H != H || o.W != W) throw std::runtime_error("deep supervision output/target size mismatch at level " + std::to_string(k));
        }
    }
    size_t levels() const { return p->g.outputs.size(); }
    const int64_t* target_of(size_t k) const { return k == 0 ? target : tgt[k]; }
    int64_t voxels(size_t k) const { const auto& o = p->g.outputs[k]; return (int64_t)o.D * o.H * o.W; }
    void prepare(hipStream_t s) {
        HIP_OK(hipMemsetAsync(losses_out, 0, 4 * sizeof(float), s));
        const Graph& g = p->g;
        int D = g.D, H = g.H, W = g.W;
        for (size_t k = 1; k < levels(); ++k) {
            launch_target_half(target_of(k - 1), tgt[k], D, H, W, s);
            D >>= 1; H >>= 1; W >>= 1;
        }
    }
    void level_partial(size_t k, int area, hipStream_t s) { launch_loss_partial(outs[k], target_of(k), C, voxels(k), collapse, partial[area], s); }
    void level_finish(size_t k, int area, hipStream_t s) {
        const float w = (1.0f / (float)(1 << k)) * inv;
        launch_loss_finalize(partial[area], loss_blocks(voxels(k)), oc, w, cost_mask, lvl[k], losses_out, k == 0, s);
        if (grad_outs && grad_outs[k]) launch_loss_grad(outs[k], target_of(k), C, voxels(k), collapse, lvl[k], w, cost_mask, grad_outs[k], s);
    }
};
}  // namespace

int unet_loss(const unet_plan* p, const float* const* outs, const int64_t* target, int cost_mask, int collapse_before,
              float* const* grad_outs, float* losses_out, void* scratch, void* stream) {
    try {
        LossRun lr(p, outs, target, cost_mask, collapse_before, grad_outs, losses_out, scratch);
        DeviceGuard dg(p->device);
        hipStream_t s = (hipStream_t)stream;
        lr.prepare(s);
        for (size_t k = 0; k < lr.levels(); ++k) { lr.level_partial(k, 0, s); lr.level_finish(k, 0, s); }
        check_launch();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

// unet_forward (train mode) + unet_loss in one call, which lets the loss of the coarse levels leave the critical path: the target
// pyramid is built on the plan's side stream while the encoder runs, and the loss kernels of level k >= 1 are issued there as soon as
// head k exists (the decoder's remaining levels run meanwhile on the caller's stream).  Level 0 stays on the caller's stream; its
// finalize comes after the join, so totals[0] is summed in a fixed order (levels 1.. then 0) and stays bit-reproducible.
int unet_forward_loss(const unet_plan* p, const float* const* params, float* const* buffers, const float* x, float* const* outs,
                      const int64_t* target, int cost_mask, int collapse_before, float* const* grad_outs, float* losses_out,
                      void* loss_scratch, void* workspace, void* stream) {
    return unet_forward_loss_mode(p, params, buffers, x, outs, target, cost_mask, collapse_before, grad_outs, losses_out, loss_scratch, workspace, 1,
                                  stream);
}

int unet_forward_loss_mode(const unet_plan* p, const float* const* params, float* const* buffers, const float* x, float* const* outs,
                           const int64_t* target, int cost_mask, int collapse_before, float* const* grad_outs, float* losses_out,
                           void* loss_scratch, void* workspace, int mode, void* stream) {
    try {
        if ((mode & 1) != 1 || (mode & ~(1 | UNET_MODE_PACKS_CURRENT)))
            throw std::runtime_error("unet_forward_loss_mode: mode must be 1 (train), optionally | UNET_MODE_PACKS_CURRENT");
        if (!p || !params || !x || !workspace || !outs) throw std::runtime_error("unet_forward_loss: null argument");
        if (!p->g.buffers.empty() && !buffers) throw std::runtime_error("unet_forward_loss: architecture has bnorm layers but buffers is null");
        LossRun lr(p, outs, target, cost_mask, collapse_before, grad_outs, losses_out, loss_scratch);
        DeviceGuard dg(p->device);
        hipStream_t s = (hipStream_t)stream;
        static const bool no_side = getenv("UNET_NO_SIDE_STREAM") != nullptr;
        const bool side = p->side && !no_side && !g_prof;
        Exec ex(*p, workspace, stream);
        if (!side) {
            ex.forward(params, buffers, x, outs, mode);
            lr.prepare(s);
            for (size_t k = 0; k < lr.levels(); ++k) { lr.level_partial(k, 0, s); lr.level_finish(k, 0, s); }
            check_launch();
            return 0;
        }
        hipStream_t sd = p->side;
        bool prepared = false;
        // ONE fork for all coarse levels, taken when the last of them (level 1) has its head: an event record costs the caller's stream
        // ~6 us, the coarse levels' loss kernels ~0.1 ms in all, and the full-resolution decoder level that follows (~0.25 ms) covers them
        constexpr bool per_level = false;     // (a fork per level, as in round 2, measured no faster: profiles/r07_ab_stream_knobs.txt)
        std::function<void(int)> on_head = [&](int level) {
            if (level < 1 || (size_t)level >= lr.levels()) return;
            if (!per_level && level != 1) return;
            HIP_OK(hipEventRecord(p->ev_fork, s));             // results[level..] are final on the caller's stream here (and so is `target`)
            HIP_OK(hipStreamWaitEvent(sd, p->ev_fork, 0));
            if (!prepared) { lr.prepare(sd); prepared = true; }   // not at entry: the forward's filter pack goes first on the side stream
            for (size_t k = per_level ? (size_t)level : lr.levels() - 1; k >= (size_t)level; --k) {   // coarsest first: the order of the totals' sum
                lr.level_partial(k, 1, sd);
                lr.level_finish(k, 1, sd);
            }
        };
        ex.forward(params, buffers, x, outs, mode, &on_head);
        if (!prepared) {   // a single-level architecture: nothing went to the side stream
            HIP_OK(hipEventRecord(p->ev_fork, s));
            HIP_OK(hipStreamWaitEvent(sd, p->ev_fork, 0));
            lr.prepare(sd);
        }
        lr.level_partial(0, 0, s);
        HIP_OK(hipEventRecord(p->ev_join, sd));
        HIP_OK(hipStreamWaitEvent(s, p->ev_join, 0));
        lr.level_finish(0, 0, s);
        check_launch();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

// A stream confined to CUs [cu_first, cu_first + cu_count) of EVERY XCD (indices wrap inside the XCD).  Bit i of the HIP CU mask is CU
// i / 8 of XCD i % 8 on this part (profiles/tools/cu_mask_probe.hip); every XCD keeps at least one CU, or the driver ignores the mask.
// Such a stream is created without hipStreamNonBlocking (the API has no flags): it synchronizes with the NULL stream like any blocking stream.
static bool create_cu_range_stream(int device, int cu_first, int cu_count, hipStream_t* out) {
    hipDeviceProp_t prop;
    if (cu_count <= 0 || hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount % 8 != 0) return false;
    const int per_xcd = prop.multiProcessorCount / 8;
    if (cu_count >= per_xcd) return false;
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0u);
    for (int c = 0; c < cu_count; ++c) {
        const int cu = ((cu_first + c) % per_xcd + per_xcd) % per_xcd;
        for (int x = 0; x < 8; ++x) { const int bit = cu * 8 + x; mask[bit / 32] |= 1u << (bit % 32); }
    }
    if (hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}

int unet_stream_create_cu_range(int device, int cu_first, int cu_count, void** stream) {
    try {
        if (!stream) throw std::runtime_error("unet_stream_create_cu_range: null argument");
        DeviceGuard dg(device);
        hipStream_t st = nullptr;
        if (!create_cu_range_stream(device, cu_first, cu_count, &st)) throw std::runtime_error("unet_stream_create_cu_range: the device / driver does not take this CU range");
        *stream = (void*)st;
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}
int unet_stream_destroy(void* stream) {
    if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) return fail("hipStreamDestroy failed");
    return 0;
}
int unet_plan_side_cu_range(unet_plan* p, int cu_first, int cu_count) {
    try {
        if (!p) throw std::runtime_error("unet_plan_side_cu_range: null plan");
        if (!p->side) return 0;
        DeviceGuard dg(p->device);
        hipStream_t st = nullptr;
        if (!create_cu_range_stream(p->device, cu_first, cu_count, &st)) throw std::runtime_error("unet_plan_side_cu_range: the device / driver does not take this CU range");
        HIP_OK(hipStreamSynchronize(p->side));
        HIP_OK(hipStreamDestroy(p->side));
        p->side = st;
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int unet_pack_filters(const unet_plan* p, const float* const* params, void* workspace, int with_dgrad, int* made, void* stream) {
    try {
        if (!p || !params || !workspace || !made) throw std::runtime_error("unet_pack_filters: null argument");
        *made = 0;
        if (!p->jobs_dev || p->pack_jobs.empty()) return 0;
        for (size_t i = 0; i < p->g.params.size(); ++i)
            if (params[i] != params[0] + p->p_off[i]) return 0;
        DeviceGuard dg(p->device);
        launch_mfma_pack_batched(params[0], workspace, p->jobs_dev, (int)p->pack_jobs.size(), with_dgrad ? p->pack_blocks : p->pack_fwd_blocks,
                                 (hipStream_t)stream, 0, 0, p->deep_part_bytes ? (int*)((char*)workspace + p->deep_cnt_off) : nullptr, p->deep_ncnt);
        check_launch();
        *made = 1;
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int unet_sum_buffers(const float* const* bufs, int n, float* out, int64_t count, int zero_inputs, void* stream) {
    try {
        if (!bufs || !out || n < 1 || n > UNET_SUM_MAX_BUFFERS || count < 0) throw std::runtime_error("unet_sum_buffers: bad argument");
        for (int k = 0; k < n; ++k)
            if (!bufs[k] || (reinterpret_cast<uintptr_t>(bufs[k]) & 15)) throw std::runtime_error("unet_sum_buffers: null or misaligned buffer");
        if (reinterpret_cast<uintptr_t>(out) & 15) throw std::runtime_error("unet_sum_buffers: misaligned output");
        if (count == 0) return 0;
        launch_sum_buffers(bufs, n, out, count, zero_inputs, (hipStream_t)stream);
        check_launch();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int unet_sgd_step(const unet_plan* p, float* params, float* grads, float* mom, float lr, float momentum, int nesterov, float wd,
                  float clip_norm, float grad_scale, float* norm_out, void* scratch, void* stream) {
    try {
        if (!p || !params || !grads || !mom || !scratch) throw std::runtime_error("unet_sgd_step: null argument");
        if (!p->segs_dev) throw std::runtime_error("unet_sgd_step: plan was created without a device");
        DeviceGuard dg(p->device);
        hipStream_t s = (hipStream_t)stream;
        const int nblk = 1024;           // partial sums of squares (scratch: 64 KiB)
        float* partial = (float*)scratch;
        launch_sumsq_partial(grads, p->n_param_elems, grad_scale, partial, nblk, s);
        launch_sgd(params, grads, mom, p->n_param_elems, p->segs_dev, p->nseg, partial, nblk, lr, momentum, nesterov, wd, clip_norm,
                   grad_scale, norm_out, s);
        check_launch();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

// ---- single-op surface ----
int unet_op_scratch_bytes(int cin, int cout, int D, int H, int W, size_t* bytes) {
    size_t b = 2 * align_up((size_t)27 * round_up(cin, 8) * round_up(cout, 8) * 4) + 4096 +
               align_up((size_t)160 * round_up(cin, 32) * round_up(cout, 32));   // largest MFMA filter pack: stride-2 dgrad, 128 B per Cin*Cout
    if (D > 0 && H > 0 && W > 0) {   // statistics partials of unet_op_conv3d_fwd_fused, behind the filter packs
        int64_t S = (int64_t)D * H * W;
        size_t blocks = (size_t)(S / 32 + 4096);   // >= any conv tile count (>= 64 voxels per tile, ragged edges) and >= stats_blocks(S)
        b += align_up(blocks * cout * 2 * 8);   // fp64 partials for fp32 tensors
    }
    if ((int64_t)cin * cout <= 1024) {   // small-weight wgrad slabs: <= 1024 row blocks x <= 1024 weights, + bias partials
        size_t w = ((size_t)1024 * 1024 + (size_t)1024 * cout) * 4 + 1024;
        if (w > b) b = w;
    }
    if (cout <= 8 || cin == 1) {   // register-accumulating small wgrads: <= 512 blocks x (weights + bias sums)
        size_t w = (size_t)512 * ((size_t)27 * cin * cout + cout) * 4 + 1024;
        if (w > b) b = w;
    }
    if (cin % 16 == 0 && cout % 16 == 0 && D > 0 && H > 0 && W > 0) {
        ConvGeom gf;   // fp32 matrix-core wgrad: per-wave slabs (<= 64 MB) + the bias partials
        gf.Cin = cin; gf.Cout = cout; gf.D = gf.Do = D; gf.H = gf.Ho = H; gf.W = gf.Wo = W; gf.ks = 3; gf.stride = 1;
        size_t wf32 = wgrad_f32_mfma_scratch_bytes(gf);
        if (wf32 > b) b = wf32;
    }
    if (cin % 16 == 0 && cout % 16 == 0 && D > 0 && H > 0 && W > 0) {
        ConvGeom g;   // MFMA wgrad slabs: stride-1 geometry has the most tiles
        g.Cin = cin; g.Cout = cout; g.D = g.Do = D; g.H = g.Ho = H; g.W = g.Wo = W; g.ks = 3; g.stride = 1;
        size_t w = std::max(mfma_wgrad_scratch_bytes(g, 0), mfma_wgrad_scratch_bytes(g, 1));   // unet_op_conv3d_bwd_weight may launch politely (UNET_OP_POLITE): more slab rows
        if (w > b) b = w;
        g.stride = 2; g.Do = (D - 1) / 2 + 1; g.Ho = (H - 1) / 2 + 1; g.Wo = (W - 1) / 2 + 1;
        w = std::max(mfma_wgrad_scratch_bytes(g, 0), mfma_wgrad_scratch_bytes(g, 1));
        if (w > b) b = w;
        g.Do = 2 * D; g.Ho = 2 * H; g.Wo = 2 * W;   // conv_trans
        w = mfma_convt_wgrad_scratch_bytes(g);
        if (w > b) b = w;
    }
    *bytes = b;
    return 0;
}

static ConvGeom op_geom(int cin, int cout, int D, int H, int W, int ks, int stride, bool transposed) {
    ConvGeom g;
    g.Cin = cin; g.Cout = cout; g.D = D; g.H = H; g.W = W; g.ks = ks; g.stride = stride;
    if (transposed) { g.Do = 2 * D; g.Ho = 2 * H; g.Wo = 2 * W; }
    else {
        int pad = (ks - 1) / 2;
        g.Do = (D + 2 * pad - ks) / stride + 1; g.Ho = (H + 2 * pad - ks) / stride + 1; g.Wo = (W + 2 * pad - ks) / stride + 1;
    }
    return g;
}
static void op_pack(const float* w, int cin, int cout, int k3, bool transposed, void* scratch, float** wf, float** wd, hipStream_t s) {
    size_t half = align_up((size_t)27 * round_up(cin, 8) * round_up(cout, 8) * 4);
    *wf = (float*)scratch; *wd = (float*)((char*)scratch + half);
    if (transposed) launch_pack_convt_w(w, *wf, *wd, cin, cout, s);
    else launch_pack_conv_w(w, *wf, *wd, cin, cout, k3, s);
}
#define OP_TRY(...) try { __VA_ARGS__; check_launch(); return 0; } catch (const std::exception& e) { return fail(e.what()); }
// the single-op surface runs the deep levels' kernels (kernels_mfma_deep.hip) as the engine does; their partial tiles and arrival counters
// live in one allocation per device, made (and cleared) on first use -- every launch leaves the counters zero.  Ops of this surface
// that use it must not run concurrently on two streams of one device (the tests and tools that call it are single-stream).
static DeepScratch op_deep() {
    static std::mutex mu;
    static std::unordered_map<int, DeepScratch> per_dev;
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_dev.find(dev);
    if (it != per_dev.end()) return it->second;
    DeepScratch d;
    d.part_bytes = (size_t)8 << 20; d.ncnt = 4096;
    char* base = nullptr;
    HIP_OK(hipMalloc((void**)&base, d.part_bytes + (size_t)d.ncnt * 4));
    HIP_OK(hipMemset(base + d.part_bytes, 0, (size_t)d.ncnt * 4));
    d.part = (float*)base; d.cnt = (int*)(base + d.part_bytes);
    per_dev[dev] = d;
    return d;
}

int unet_op_conv3d_fwd(int dtype, int impl, const void* x, const float* w, const float* b, void* y, int cin, int cout, int D, int H,
                       int W, int ks, int stride, void* scratch, void* stream) {
    OP_TRY({
        hipStream_t s = (hipStream_t)stream;
        ConvGeom g = op_geom(cin, cout, D, H, W, ks, stride, false);
        float *wf, *wd;
        SrcDesc sd; sd.ptr = x; sd.C = cin;
        if (impl == UNET_IMPL_AUTO && mfma_conv_fwd_supported(dtype, g, &sd, 1)) {
            void* wm = (char*)scratch + 2 * align_up((size_t)27 * round_up(cin, 8) * round_up(cout, 8) * 4);
            launch_mfma_pack_conv_w(w, wm, nullptr, g, s);
            if (!launch_deep_conv_fwd(g, &sd, 1, wm, b, y, nullptr, op_deep(), s)) launch_mfma_conv_fwd(g, &sd, 1, wm, b, y, nullptr, s);
        } else if (impl == UNET_IMPL_AUTO && conv_first_mfma_supported(dtype, g, &sd, 1)) {
            launch_conv_first_mfma(g, &sd, w, b, y, nullptr, s);
        } else if (impl == UNET_IMPL_AUTO && conv_first_f32_mfma_supported(dtype, g, &sd, 1)) {
            (void)launch_conv_first_f32_mfma(g, &sd, w, b, (float*)y, nullptr, s);
        } else {
            op_pack(w, cin, cout, ks * ks * ks, false, scratch, &wf, &wd, s);
            if (impl == UNET_IMPL_AUTO && conv_f32_mfma_supported(dtype, g, &sd, 1)) (void)launch_conv_f32_mfma(g, &sd, 1, wf, b, (float*)y, s);
            else launch_conv_fwd_direct(dtype, g, &sd, 1, wf, b, y, nullptr, s);
        }
    })
}
int unet_op_conv3d_fwd_fused(int dtype, int impl, const void* x, const float* scale, const float* shift, int act, const float* w,
                             const float* b, void* y, float* stats, int cin, int cout, int D, int H, int W, int ks, int stride,
                             void* scratch, void* stream) {
    OP_TRY({
        hipStream_t s = (hipStream_t)stream;
        ConvGeom g = op_geom(cin, cout, D, H, W, ks, stride, false);
        float *wf, *wd;
        SrcDesc sd; sd.ptr = x; sd.C = cin; sd.scale = scale; sd.shift = shift; sd.act = act;
        int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
        // partials live behind the filter packs
        size_t poff = 2 * align_up((size_t)27 * round_up(cin, 8) * round_up(cout, 8) * 4) + align_up((size_t)160 * round_up(cin, 32) * round_up(cout, 32));
        float* part = (float*)((char*)scratch + poff);
        if (impl == UNET_IMPL_AUTO && mfma_conv_fwd_supported(dtype, g, &sd, 1)) {
            void* wm = (char*)scratch + 2 * align_up((size_t)27 * round_up(cin, 8) * round_up(cout, 8) * 4);
            launch_mfma_pack_conv_w(w, wm, nullptr, g, s);
            int rows = launch_mfma_conv_fwd(g, &sd, 1, wm, b, y, stats ? part : nullptr, s);
            if (stats) launch_stats_sum(part, rows, cout, stats, s);
        } else if (impl == UNET_IMPL_AUTO && conv_first_mfma_supported(dtype, g, &sd, 1)) {
            int rows = launch_conv_first_mfma(g, &sd, w, b, y, stats ? part : nullptr, s);
            if (stats) launch_stats_sum(part, rows, cout, stats, s);
        } else {
            op_pack(w, cin, cout, ks * ks * ks, false, scratch, &wf, &wd, s);
            if (impl == UNET_IMPL_AUTO && conv_f32_mfma_supported(dtype, g, &sd, 1)) {
                // the fp32 matrix-core conv leaves its own fp64 statistics rows (one per tile)
                const int rows = launch_conv_f32_mfma(g, &sd, 1, wf, b, (float*)y, s, stats ? (double*)part : nullptr);
                if (stats) launch_stats_sum(part, rows, cout, stats, s, true);
            } else {
                launch_conv_fwd_direct(dtype, g, &sd, 1, wf, b, y, nullptr, s);
                if (stats) {
                    launch_stats_partial(dtype, y, cout, So, part, s);
                    launch_stats_sum(part, stats_blocks(So), cout, stats, s, dtype == UNET_DTYPE_F32);
                }
            }
        }
    })
}
int unet_op_conv3d_pack(int dtype, const float* w, void* wpacked, int cin, int cout, int D, int H, int W, int ks, int stride,
                        void* stream) {
    OP_TRY({
        ConvGeom g = op_geom(cin, cout, D, H, W, ks, stride, false);
        SrcDesc sd; sd.C = cin;
        if (!mfma_conv_fwd_supported(dtype, g, &sd, 1)) throw std::runtime_error("unet_op_conv3d_pack: shape not covered by the MFMA kernels");
        launch_mfma_pack_conv_w(w, wpacked, nullptr, g, (hipStream_t)stream);
    })
}
int unet_op_conv3d_fwd_packed(int dtype, const void* x, const void* wpacked, const float* b, void* y, float* stats_partials,
                              int cin, int cout, int D, int H, int W, int ks, int stride, void* stream) {
    OP_TRY({
        ConvGeom g = op_geom(cin, cout, D, H, W, ks, stride, false);
        SrcDesc sd; sd.ptr = x; sd.C = cin;
        if (!mfma_conv_fwd_supported(dtype, g, &sd, 1)) throw std::runtime_error("unet_op_conv3d_fwd_packed: shape not covered by the MFMA kernels");
        launch_mfma_conv_fwd(g, &sd, 1, wpacked, b, y, stats_partials, (hipStream_t)stream);
    })
}
int unet_op_conv3d_bwd_data(int dtype, int impl, const void* dy, const float* w, void* dx, int cin, int cout, int D, int H, int W,
                            int ks, int stride, void* scratch, void* stream) {
    OP_TRY({
        hipStream_t s = (hipStream_t)stream;
        ConvGeom g = op_geom(cin, cout, D, H, W, ks, stride, false);
        float *wf, *wd;
        DstGrad d; d.ptr = dx; d.C = cin; d.accumulate = 0;
        SrcDesc sd; sd.C = cin;
        if (impl == UNET_IMPL_AUTO && mfma_conv_dgrad_supported(dtype, g, &sd, 1)) {
            void* wm = (char*)scratch + 2 * align_up((size_t)27 * round_up(cin, 8) * round_up(cout, 8) * 4);
            launch_mfma_pack_conv_w(w, nullptr, wm, g, s);
            if (!launch_deep_conv_dgrad(g, dy, wm, &d, 1, nullptr, op_deep(), s)) launch_mfma_conv_dgrad(g, dy, wm, &d, 1, s);
        } else {
            op_pack(w, cin, cout, ks * ks * ks, false, scratch, &wf, &wd, s);
            if (impl == UNET_IMPL_AUTO && conv_f32_mfma_dgrad_supported(dtype, g, &d, 1)) launch_conv_f32_mfma_dgrad(g, (const float*)dy, wd, &d, 1, s);
            else launch_conv_dgrad_direct(dtype, g, dy, wd, &d, 1, s);
        }
    })
}
int unet_op_conv3d_bwd_weight(int dtype, int impl, const void* x, const void* dy, float* dw, float* db, int cin, int cout, int D,
                              int H, int W, int ks, int stride, void* scratch, void* stream) {
    OP_TRY({
        ConvGeom g = op_geom(cin, cout, D, H, W, ks, stride, false);
        SrcDesc sd; sd.ptr = x; sd.C = cin;
        // UNET_OP_POLITE=1: the launch configuration the engine gives this layer's gradient on its side stream (one 4-wave block per
        // CU, single (ca, cb) pairs) -- for micro-benchmarks and counter collection of the kernel as the train step runs it
        static const bool op_polite = getenv("UNET_OP_POLITE") != nullptr;
        if (impl == UNET_IMPL_AUTO && mfma_wgrad_supported(dtype, g, &sd, 1))
            launch_mfma_conv_wgrad(g, &sd, 1, dy, dw, db, scratch, (hipStream_t)stream, false, op_polite ? 1 : 0);
        else if (impl == UNET_IMPL_AUTO && conv_first_wgrad_mfma_supported(dtype, g, &sd, 1))
            launch_conv_first_wgrad_mfma(g, &sd, dy, dw, db, scratch, (hipStream_t)stream);
        else if (impl == UNET_IMPL_AUTO && wgrad_f32_mfma_supported(dtype, g, &sd, 1))
            launch_wgrad_f32_mfma(g, &sd, 1, (const float*)dy, dw, db, scratch, (hipStream_t)stream);
        else if (impl == UNET_IMPL_AUTO && wgrad_small_supported(g, 1))
            launch_conv_wgrad_small(dtype, g, &sd, 1, dy, dw, db, scratch, (hipStream_t)stream);
        else
            launch_conv_wgrad_direct(dtype, g, &sd, 1, dy, dw, db, nullptr, (hipStream_t)stream);
    })
}
int unet_op_convt_fwd(int dtype, int impl, const void* x, const float* w, const float* b, void* y, int cin, int cout, int D, int H,
                      int W, void* scratch, void* stream) {
    OP_TRY({
        hipStream_t s = (hipStream_t)stream;
        ConvGeom g = op_geom(cin, cout, D, H, W, 2, 2, true);
        float *wf, *wd;
        SrcDesc sd; sd.ptr = x; sd.C = cin;
        if (impl == UNET_IMPL_AUTO && mfma_convt_supported(dtype, g, &sd, 1)) {
            void* wm = (char*)scratch + 2 * align_up((size_t)27 * round_up(cin, 8) * round_up(cout, 8) * 4);
            launch_mfma_pack_convt_w(w, wm, nullptr, g, s);
            if (!launch_deep_convt_fwd(g, &sd, 1, wm, b, y, op_deep(), s)) launch_mfma_convt_fwd(g, &sd, 1, wm, b, y, s);
        } else {
            op_pack(w, cin, cout, 8, true, scratch, &wf, &wd, s);
            if (impl == UNET_IMPL_AUTO && convt_f32_mfma_supported(dtype, g, &sd, 1)) launch_convt_f32_mfma(g, &sd, wf, b, (float*)y, s);
            else launch_convt_fwd_direct(dtype, g, &sd, 1, wf, b, y, s);
        }
    })
}
int unet_op_convt_bwd_data(int dtype, int impl, const void* dy, const float* w, void* dx, int cin, int cout, int D, int H, int W,
                           void* scratch, void* stream) {
    OP_TRY({
        hipStream_t s = (hipStream_t)stream;
        ConvGeom g = op_geom(cin, cout, D, H, W, 2, 2, true);
        float *wf, *wd;
        DstGrad d; d.ptr = dx; d.C = cin; d.accumulate = 0;
        SrcDesc sd; sd.C = cin;
        if (impl == UNET_IMPL_AUTO && mfma_convt_supported(dtype, g, &sd, 1)) {
            void* wm = (char*)scratch + 2 * align_up((size_t)27 * round_up(cin, 8) * round_up(cout, 8) * 4);
            launch_mfma_pack_convt_w(w, nullptr, wm, g, s);
            if (!launch_deep_convt_dgrad(g, dy, wm, &d, 1, nullptr, op_deep(), nullptr, s)) launch_mfma_convt_dgrad(g, dy, wm, &d, 1, s);
        } else {
            op_pack(w, cin, cout, 8, true, scratch, &wf, &wd, s);
            launch_convt_dgrad_direct(dtype, g, dy, wd, &d, 1, s);
        }
    })
}
int unet_op_convt_bwd_weight(int dtype, int impl, const void* x, const void* dy, float* dw, float* db, int cin, int cout, int D, int H,
                             int W, void* scratch, void* stream) {
    OP_TRY({
        ConvGeom g = op_geom(cin, cout, D, H, W, 2, 2, true);
        SrcDesc sd; sd.ptr = x; sd.C = cin;
        if (impl == UNET_IMPL_AUTO && mfma_convt_wgrad_supported(dtype, g, &sd, 1)) {
            launch_mfma_convt_wgrad(g, &sd, dy, dw, scratch, (hipStream_t)stream, false, db);
        } else {
            launch_convt_wgrad_direct(dtype, g, &sd, 1, dy, dw, db, nullptr, (hipStream_t)stream);
        }
    })
}
int unet_op_pack_ndhwc(int dtype, const float* x, void* y, int C, int64_t S, void* stream) {
    OP_TRY({ launch_pack_input(dtype, x, y, C, S, (hipStream_t)stream); })
}
int unet_op_unpack_ncdhw(int dtype, const void* x, float* y, int C, int64_t S, void* stream) {
    OP_TRY({ launch_unpack_ncdhw(dtype, x, y, C, S, (hipStream_t)stream); })
}


// ---- on-GPU sample augmentation (include/unet_augment.h) ----
static const char* augment_recipe_error(const UnetAugmentRecipe* r) {
    if (!r) return "unet_augment: null recipe";
    for (int d = 0; d < 3; ++d)
        if (r->dims[d] < 1) return "unet_augment: dims must be positive";
    if (r->channels < 1 || r->channels > UNET_AUG_MAX_CHANNELS) return "unet_augment: channels out of range (1..UNET_AUG_MAX_CHANNELS)";
    if (r->n_foci < 0 || r->n_foci > UNET_AUG_MAX_FOCI) return "unet_augment: n_foci out of range (0..UNET_AUG_MAX_FOCI)";
    if (r->trunc_top < 0 || r->trunc_bottom < 0) return "unet_augment: negative truncation";
    if (r->downsample)
        for (int d = 0; d < 3; ++d)
            if (r->low_dims[d] < 1 || r->low_dims[d] > r->dims[d]) return "unet_augment: low_dims must be in 1..dims";
    for (int k = 0; k < r->n_foci; ++k)
        if (!(r->foci_radius[k] > 0.f)) return "unet_augment: distortion radius must be positive";
    return nullptr;
}
int unet_augment_scratch_bytes(const UnetAugmentRecipe* recipe, size_t* bytes) {
    if (const char* e = augment_recipe_error(recipe)) return fail(e);
    if (!bytes) return fail("unet_augment_scratch_bytes: null output");
    *bytes = augment_scratch_bytes(*recipe);
    return 0;
}
int unet_augment_run(const UnetAugmentRecipe* recipe, float* image, float* label, void* scratch, size_t scratch_bytes, void* stream) {
    if (const char* e = augment_recipe_error(recipe)) return fail(e);
    if (!image || !label || !scratch) return fail("unet_augment_run: null device pointer");
    if (scratch_bytes < augment_scratch_bytes(*recipe)) return fail("unet_augment_run: scratch too small (see unet_augment_scratch_bytes)");
    OP_TRY({
        // the volumes' device is the one to launch on, whatever the calling thread's current device is
        hipPointerAttribute_t at;
        HIP_OK(hipPointerGetAttributes(&at, image));
        DeviceGuard guard(at.device);
        launch_augment(*recipe, image, label, scratch, (hipStream_t)stream);
    })
}

// ---- simulate_modality (include/unet_augment.h) ----
static const char* simulate_recipe_error(const UnetSimulateRecipe* r) {
    if (!r) return "unet_simulate_modality: null recipe";
    for (int d = 0; d < 3; ++d)
        if (r->dims[d] < 1) return "unet_simulate_modality: dims must be positive";
    if (r->with_label && (r->max_label < 0 || r->max_label >= UNET_SIM_MAX_LABELS)) return "unet_simulate_modality: max_label out of range";
    for (int k = 0; k < UNET_SIM_TERMS; ++k)
        if (r->term_a[k] > 3 || r->term_b[k] > 3 || r->term_c[k] > 3 || r->term_d[k] > 3) return "unet_simulate_modality: exponents must be 0..3";
    return nullptr;
}
int unet_simulate_modality_scratch_bytes(const UnetSimulateRecipe* recipe, size_t* bytes) {
    if (const char* e = simulate_recipe_error(recipe)) return fail(e);
    if (!bytes) return fail("unet_simulate_modality_scratch_bytes: null output");
    *bytes = simulate_scratch_bytes(*recipe);
    return 0;
}
int unet_simulate_modality_run(const UnetSimulateRecipe* recipe, float* t1w, const float* label, void* scratch, size_t scratch_bytes,
                               void* stream) {
    if (const char* e = simulate_recipe_error(recipe)) return fail(e);
    if (!t1w || !scratch || (recipe->with_label && !label)) return fail("unet_simulate_modality_run: null device pointer");
    if (scratch_bytes < simulate_scratch_bytes(*recipe)) return fail("unet_simulate_modality_run: scratch too small");
    OP_TRY({
        hipPointerAttribute_t at;
        HIP_OK(hipPointerGetAttributes(&at, t1w));
        DeviceGuard guard(at.device);
        launch_simulate_modality(*recipe, t1w, label, scratch, (hipStream_t)stream);
    })
}

}  // extern "C"
