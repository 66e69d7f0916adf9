// DSL -> lowered graph.  Follows UNet3dImpl::create_layer (unet.cpp:24-101), the constructor
// (unet.cpp:103-166) for channel bookkeeping and parameter order, and forward (unet.cpp:168-193)
// for op order.
#include "graph.hpp"

#include <sstream>
#include <stdexcept>
#include <unordered_map>

namespace unet {
namespace {

enum LayerKind { L_CONV, L_CONVT, L_NORM, L_BNORM, L_MAXPOOL, L_UPSAMPLE, L_ACT };
struct Layer {
    LayerKind kind;
    int cin = 0, cout = 0, ks = 0, stride = 0, act = ACT_NONE;
    int weight = -1, bias = -1, buffer = -1;
};
typedef std::vector<Layer> Seq;

std::vector<std::string> split(const std::string& s, char ch) {
    std::vector<std::string> out;
    std::string cur;
    std::istringstream in(s);
    while (std::getline(in, cur, ch)) out.push_back(cur);
    return out;
}

// tipl::split_by_line_breaks is not in the reference tree (TIPL): strip '\r', skip blank lines.
std::vector<std::string> split_lines(const std::string& s) {
    std::vector<std::string> out;
    for (auto l : split(s, '\n')) {
        while (!l.empty() && (l.back() == '\r' || l.back() == ' ')) l.pop_back();
        if (!l.empty()) out.push_back(l);
    }
    return out;
}

struct Builder {
    Graph g;
    std::vector<int> producer;  // tensor id -> op index that wrote it (-1: none)

    int add_param(std::vector<int64_t> shape, bool decay, int64_t fan_in, bool norm_w, const std::string& name) {
        Param p;
        p.shape = shape; p.decay = decay; p.fan_in = fan_in; p.norm_weight = norm_w; p.name = name;
        g.params.push_back(p);
        return (int)g.params.size() - 1;
    }

    // unet.cpp:24-101
    int create_layer(Seq& layers, const std::string& def, int in_c, const std::string& prefix) {
        std::unordered_map<std::string, std::string> params;
        std::string first_key;
        for (const auto& arg : split(def, ',')) {
            size_t pos = arg.find_first_of("0123456789");
            std::string key = pos != std::string::npos ? arg.substr(0, pos) : arg;
            params[key] = pos != std::string::npos ? arg.substr(pos) : "1";
            if (first_key.empty()) first_key = key;
        }
        int out_c = in_c;
        std::string nm = prefix + "." + std::to_string(layers.size());
        Layer l;
        if (params.count("max_pool")) {
            l.kind = L_MAXPOOL;
        } else if (params.count("upsample")) {
            l.kind = L_UPSAMPLE;
        } else if (params.count("conv_trans")) {
            out_c = std::stoi(params["conv_trans"]);
            int ks = params.count("ks") ? std::stoi(params["ks"]) : 2;
            int stride = params.count("stride") ? std::stoi(params["stride"]) : 2;
            if (ks != 2 || stride != 2) throw std::runtime_error("conv_trans supports only ks2 stride2");
            l.kind = L_CONVT; l.cin = in_c; l.cout = out_c; l.ks = 2; l.stride = 2;
            // ConvTranspose3d weight [Cin,Cout,2,2,2]; torch computes fan_in from size(1)*k^3
            l.weight = add_param({in_c, out_c, 2, 2, 2}, true, (int64_t)out_c * 8, false, nm + ".weight");
            l.bias = add_param({out_c}, false, (int64_t)out_c * 8, false, nm + ".bias");
        } else if (params.count("conv")) {
            out_c = std::stoi(params["conv"]);
            int ks = params.count("ks") ? std::stoi(params["ks"]) : 3;
            int stride = params.count("stride") ? std::stoi(params["stride"]) : 1;
            if (!((ks == 1 && stride == 1) || (ks == 3 && (stride == 1 || stride == 2))))
                throw std::runtime_error("conv supports only ks1 stride1, ks3 stride1, and ks3 stride2");
            l.kind = L_CONV; l.cin = in_c; l.cout = out_c; l.ks = ks; l.stride = stride;
            int64_t fan = (int64_t)in_c * ks * ks * ks;
            l.weight = add_param({out_c, in_c, ks, ks, ks}, true, fan, false, nm + ".weight");
            l.bias = add_param({out_c}, false, fan, false, nm + ".bias");
        } else if (params.count("norm") || params.count("bnorm")) {
            l.kind = params.count("norm") ? L_NORM : L_BNORM;
            l.cin = l.cout = in_c;
            l.weight = add_param({in_c}, false, 0, true, nm + ".weight");
            l.bias = add_param({in_c}, false, 0, false, nm + ".bias");
            if (l.kind == L_BNORM) {
                l.buffer = (int)g.buffers.size();
                g.buffers.push_back(in_c);
                g.buffers.push_back(in_c);
            }
        } else {
            throw std::runtime_error("unknown layer: " + (params.empty() ? def : first_key));
        }
        if (out_c <= 0 || in_c <= 0) throw std::runtime_error("invalid channel count in layer: " + def);
        layers.push_back(l);
        Layer a;
        a.kind = L_ACT;
        if (params.count("relu")) a.act = ACT_RELU;
        else if (params.count("leaky_relu")) a.act = ACT_LEAKY;
        else if (params.count("elu")) a.act = ACT_ELU;
        if (a.act != ACT_NONE) layers.push_back(a);
        return out_c;
    }

    // ---- lowering ----
    struct Cursor { int t0 = -1, t1 = -1; };  // t1 >= 0: pending channel concat {t0, t1}

    int new_tensor(int C, int D, int H, int W) {
        if (C <= 0 || D <= 0 || H <= 0 || W <= 0) throw std::runtime_error("tensor size became zero inside the network");
        Tensor t;
        t.C = C; t.D = D; t.H = H; t.W = W;
        g.tensors.push_back(t);
        producer.push_back(-1);
        return (int)g.tensors.size() - 1;
    }
    int channels(const Cursor& c) const { return g.tensors[c.t0].C + (c.t1 >= 0 ? g.tensors[c.t1].C : 0); }
    void set_srcs(Op& op, const Cursor& c) {
        op.nsrc = c.t1 >= 0 ? 2 : 1;
        op.src[0] = c.t0; op.src[1] = c.t1;
    }
    int push(Op op) {
        g.ops.push_back(op);
        if (op.dst >= 0) producer[op.dst] = (int)g.ops.size() - 1;
        return (int)g.ops.size() - 1;
    }
    Cursor materialize(const Cursor& c, const std::string& why) {
        const Tensor& a = g.tensors[c.t0];
        Op op;
        op.kind = OP_MATERIALIZE;
        set_srcs(op, c);
        op.dst = new_tensor(channels(c), a.D, a.H, a.W);
        op.name = "materialize(" + why + ")";
        bool ng = g.tensors[c.t0].needs_grad || (c.t1 >= 0 && g.tensors[c.t1].needs_grad);
        g.tensors[op.dst].needs_grad = ng;
        push(op);
        Cursor r; r.t0 = op.dst;
        return r;
    }
    void freeze(const Cursor& c) {
        g.tensors[c.t0].frozen = true;
        if (c.t1 >= 0) g.tensors[c.t1].frozen = true;
    }

    Cursor run_seq(const Seq& layers, Cursor cur, const std::string& prefix) {
        int li = 0;
        for (const Layer& l : layers) {
            std::string nm = prefix + "." + std::to_string(li++);
            const Tensor a = g.tensors[cur.t0];
            if (l.kind == L_CONV || l.kind == L_CONVT) {
                if (channels(cur) != l.cin)
                    throw std::runtime_error("channel mismatch at " + nm + ": layer expects " + std::to_string(l.cin) +
                                             " channels, tensor has " + std::to_string(channels(cur)));
                Op op;
                op.kind = l.kind == L_CONV ? OP_CONV : OP_CONVT;
                set_srcs(op, cur);
                op.weight = l.weight; op.bias = l.bias; op.cin = l.cin; op.cout = l.cout; op.ks = l.ks; op.stride = l.stride;
                int Do, Ho, Wo;
                if (l.kind == L_CONV) {
                    int pad = (l.ks - 1) / 2;
                    Do = (a.D + 2 * pad - l.ks) / l.stride + 1; Ho = (a.H + 2 * pad - l.ks) / l.stride + 1;
                    Wo = (a.W + 2 * pad - l.ks) / l.stride + 1;
                    g.flops_fwd += 2.0 * l.cin * l.cout * l.ks * l.ks * l.ks * (double)Do * Ho * Wo;
                } else {
                    Do = 2 * a.D; Ho = 2 * a.H; Wo = 2 * a.W;
                    g.flops_fwd += 2.0 * l.cin * l.cout * 8.0 * (double)a.D * a.H * a.W;
                }
                double f = l.kind == L_CONV ? 2.0 * l.cin * l.cout * l.ks * l.ks * l.ks * (double)Do * Ho * Wo
                                            : 2.0 * l.cin * l.cout * 8.0 * (double)a.D * a.H * a.W;
                bool ng = g.tensors[cur.t0].needs_grad || (cur.t1 >= 0 && g.tensors[cur.t1].needs_grad);
                g.flops_bwd += f + (ng ? f : 0.0);
                op.dst = new_tensor(l.cout, Do, Ho, Wo);
                op.name = nm + (l.kind == L_CONV ? ":conv" : ":conv_trans") + std::to_string(l.cout) + ",ks" + std::to_string(l.ks) +
                          ",stride" + std::to_string(l.stride);
                push(op);
                cur = Cursor(); cur.t0 = op.dst;
            } else if (l.kind == L_NORM || l.kind == L_BNORM) {
                if (channels(cur) != l.cin) throw std::runtime_error("channel mismatch at " + nm);
                const Tensor& t = g.tensors[cur.t0];
                if (cur.t1 >= 0 || t.frozen || t.norm >= 0 || t.act != ACT_NONE) cur = materialize(cur, nm);
                Norm n;
                n.tensor = cur.t0; n.C = l.cin; n.batch = l.kind == L_BNORM; n.gamma = l.weight; n.beta = l.bias;
                n.buffer = l.buffer; n.eps = n.batch ? 0.0 : 1e-5;
                g.norms.push_back(n);
                g.tensors[cur.t0].norm = (int)g.norms.size() - 1;
                Op op;
                op.kind = OP_NORM; op.nsrc = 1; op.src[0] = cur.t0; op.norm = (int)g.norms.size() - 1;
                op.name = nm + (n.batch ? ":bnorm" : ":norm");
                push(op);
            } else if (l.kind == L_ACT) {
                const Tensor& t = g.tensors[cur.t0];
                if (cur.t1 >= 0 || t.frozen || t.act != ACT_NONE) cur = materialize(cur, nm);
                g.tensors[cur.t0].act = l.act;
            } else {  // max_pool / upsample
                if (cur.t1 >= 0) cur = materialize(cur, nm);
                const Tensor s = g.tensors[cur.t0];
                Op op;
                op.kind = l.kind == L_MAXPOOL ? OP_MAXPOOL : OP_UPSAMPLE;
                op.nsrc = 1; op.src[0] = cur.t0;
                op.dst = l.kind == L_MAXPOOL ? new_tensor(s.C, s.D / 2, s.H / 2, s.W / 2) : new_tensor(s.C, 2 * s.D, 2 * s.H, 2 * s.W);
                g.tensors[op.dst].needs_grad = s.needs_grad;
                op.name = nm + (l.kind == L_MAXPOOL ? ":max_pool" : ":upsample");
                push(op);
                cur = Cursor(); cur.t0 = op.dst;
            }
        }
        return cur;
    }
};

}  // namespace

Graph Graph::build(const std::string& arch, int in_c, int out_c, int D, int H, int W) {
    if (in_c <= 0 || out_c <= 0 || D <= 0 || H <= 0 || W <= 0) throw std::runtime_error("invalid u-net input size");
    Builder b;
    Graph& g = b.g;
    g.in_c = in_c; g.out_c = out_c; g.D = D; g.H = H; g.W = W;

    // ---- constructor pass, unet.cpp:103-166: token split, channel bookkeeping, parameter order ----
    std::vector<std::vector<std::string>> enc_tokens, dec_tokens;
    {
        std::vector<std::string> all_lines = split_lines(arch);
        if (all_lines.size() < 3) throw std::runtime_error("invalid u-net structure");
        if (all_lines.size() % 2 == 0)
            throw std::runtime_error("invalid u-net structure: an even number of lines leaves an encoder level without a decoder");
        size_t enc_count = all_lines.size() / 2 + 1;
        for (size_t i = 0; i < all_lines.size(); ++i) (i < enc_count ? enc_tokens : dec_tokens).push_back(split(all_lines[i], '+'));
    }
    std::vector<Seq> encoding(enc_tokens.size());
    int channel = in_c;
    std::vector<int> skip_channels(enc_tokens.size());
    for (size_t level = 0; level < enc_tokens.size(); ++level) {
        for (const auto& token : enc_tokens[level])
            channel = b.create_layer(encoding[level], token, channel, "encode" + std::to_string(level));
        skip_channels[level] = channel;
    }
    size_t nd = dec_tokens.size();
    std::vector<Seq> decoding(nd), output(nd), tail(nd);
    if (dec_tokens.back().empty()) throw std::runtime_error("invalid u-net structure");
    std::string out_token = dec_tokens.back().back();
    for (int level = (int)nd - 1; level >= 0; --level) {
        const auto& tokens = dec_tokens[nd - 1 - level];
        bool after_out = false;
        channel += skip_channels[level];
        // registration order is decode, output, decode_tail (unet.cpp:160-164): create in that order so that
        // parameter indices follow parameters()
        Seq d, o, tl;
        std::vector<std::pair<int, std::string>> plan;  // (which seq, token)
        for (const auto& t : tokens) {
            if (t == out_token) { plan.push_back({1, t}); after_out = true; continue; }
            plan.push_back({after_out ? 2 : 0, t});
        }
        int ch_dec = channel;
        for (auto& p : plan) if (p.first == 0) ch_dec = b.create_layer(d, p.second, ch_dec, "decode" + std::to_string(level));
        {   // channel seen by each output token = running channel at its position in the line
            int ch = channel;
            std::vector<int> ch_at;
            Seq scratch_d, scratch_t;
            Builder tmp;  // dry run for channel values only
            for (auto& p : plan) {
                ch_at.push_back(ch);
                if (p.first != 1) ch = tmp.create_layer(p.first == 0 ? scratch_d : scratch_t, p.second, ch, "dry");
            }
            for (size_t i = 0; i < plan.size(); ++i)
                if (plan[i].first == 1) b.create_layer(o, plan[i].second, ch_at[i], "output" + std::to_string(level));
            int ch_t = ch_dec;
            for (auto& p : plan) if (p.first == 2) ch_t = b.create_layer(tl, p.second, ch_t, "decode_tail" + std::to_string(level));
            channel = ch;
        }
        decoding[level] = d; output[level] = o; tail[level] = tl;
    }

    // ---- forward pass, unet.cpp:168-193 ----
    Builder::Cursor cur;
    cur.t0 = b.new_tensor(in_c, D, H, W);
    g.tensors[cur.t0].needs_grad = false;
    {
        Op op;
        op.kind = OP_PACK_INPUT; op.dst = cur.t0; op.name = "input";
        b.push(op);
    }
    std::vector<int> skips(encoding.size() - 1, -1);
    for (size_t level = 0; level < encoding.size(); ++level) {
        cur = b.run_seq(encoding[level], cur, "encode" + std::to_string(level));
        b.freeze(cur);
        if (level < encoding.size() - 1) skips[level] = cur.t0;
    }
    g.outputs.resize(nd);
    for (int level = (int)encoding.size() - 2; level >= 0; --level) {
        if (cur.t1 >= 0) cur = b.materialize(cur, "cat");
        const Tensor &s = g.tensors[skips[level]], &x = g.tensors[cur.t0];
        if (s.D != x.D || s.H != x.H || s.W != x.W)
            throw std::runtime_error("torch.cat size mismatch at decoder level " + std::to_string(level) + ": skip is " +
                                     std::to_string(s.D) + "x" + std::to_string(s.H) + "x" + std::to_string(s.W) + ", x is " +
                                     std::to_string(x.D) + "x" + std::to_string(x.H) + "x" + std::to_string(x.W));
        Builder::Cursor cat;
        cat.t0 = skips[level]; cat.t1 = cur.t0;
        cur = b.run_seq(decoding[level], cat, "decode" + std::to_string(level));
        b.freeze(cur);
        if (!output[level].empty()) {
            size_t first_op = g.ops.size();
            Builder::Cursor o = b.run_seq(output[level], cur, "output" + std::to_string(level));
            if (o.t1 >= 0) o = b.materialize(o, "output cat");
            Tensor& t = g.tensors[o.t0];
            Graph::Out oo;
            oo.C = t.C; oo.D = t.D; oo.H = t.H; oo.W = t.W; oo.tensor = o.t0;
            g.outputs[level] = oo;
            int p = b.producer[o.t0];
            if (p >= (int)first_op && p == (int)g.ops.size() - 1 && g.ops[p].kind == OP_CONV && t.norm < 0 && t.act == ACT_NONE && !t.frozen) {
                g.ops[p].out_level = level;  // conv writes results[level] directly
            } else {
                Op op;
                op.kind = OP_EXPORT; op.nsrc = 1; op.src[0] = o.t0; op.out_level = level;
                op.name = "output" + std::to_string(level) + ":export";
                b.push(op);
            }
            t.frozen = true;
        }
        if (!tail[level].empty()) {
            cur = b.run_seq(tail[level], cur, "decode_tail" + std::to_string(level));
            b.freeze(cur);
        }
    }
    return g;
}

std::string Graph::describe() const {
    static const char* kn[] = {"pack_input", "conv", "conv_trans", "norm", "materialize", "max_pool", "upsample", "export"};
    static const char* an[] = {"", "+relu", "+leaky_relu", "+elu"};
    std::ostringstream o;
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op& op = ops[i];
        o << i << " " << kn[op.kind] << " " << op.name << " src[";
        for (int s = 0; s < op.nsrc; ++s) {
            const Tensor& t = tensors[op.src[s]];
            o << (s ? "," : "") << "t" << op.src[s] << (t.norm >= 0 ? "+norm" : "") << an[t.act];
        }
        o << "]";
        if (op.dst >= 0) {
            const Tensor& t = tensors[op.dst];
            o << " -> t" << op.dst << " {" << t.C << "," << t.D << "," << t.H << "," << t.W << "}";
        }
        if (op.out_level >= 0) o << " => results[" << op.out_level << "]";
        o << "\n";
    }
    return o.str();
}

}  // namespace unet
