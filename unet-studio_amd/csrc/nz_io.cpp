// `.nz` network files for the C++ host: the reference's free functions
//     bool save_to_file(UNet3d& model, const char* file_name);      main.cpp:207-233   (declared train.hpp:32)
//     bool load_from_file(UNet3d& model, const char* file_name);    main.cpp:157-206   (declared train.hpp:33, evaluate.cpp:460)
// with the same signatures, record names, order and shapes.  The reference's container code is TIPL's gz_mat_read / gz_mat_write
// (absent); this is the same restatement as unet-studio_amd/nz.py -- a gzip stream of little-endian MATLAB Level-4 records
// {int32 type, mrows, ncols, imagf, namlen; name; column-major data} -- and the two are tested against each other.  PARITY UNPINNED
// (no .nz file or TIPL source in the tree).  Tensors are written as plain float; a tensor stored in TIPL's `sloped` encoding
// (main.cpp:223-229) is refused with a message instead of being dequantised by guesswork.  Errors go to model->error_msg and the
// functions return false, like the reference's `return tipl::error() << ..., false`.
#include <zlib.h>

#include <cstring>
#include <map>
#include <sstream>

#include "../../include/unet.hpp"

namespace {

struct Rec { int p = 0, text = 0, rows = 0, cols = 0; std::vector<char> data; };
const int kElem[6] = {8, 4, 4, 2, 2, 1};

bool put(gzFile f, const std::string& name, int p, int text, int rows, int cols, const void* data)
{
    int32_t h[5] = {p * 10 + text, rows, cols, 0, (int32_t)name.size() + 1};
    if (gzwrite(f, h, sizeof(h)) != (int)sizeof(h)) return false;
    if (gzwrite(f, name.c_str(), (unsigned)name.size() + 1) != (int)name.size() + 1) return false;
    size_t n = (size_t)rows * cols * kElem[p];
    const char* d = (const char*)data;
    while (n) {   // gzwrite takes an unsigned length
        unsigned c = n > (1u << 30) ? (1u << 30) : (unsigned)n;
        if (gzwrite(f, d, c) != (int)c) return false;
        d += c; n -= c;
    }
    return true;
}
bool put_text(gzFile f, const std::string& name, const std::string& s) { return put(f, name, 5, 1, 1, (int)s.size(), s.data()); }

struct GzFile {   // closes on every path out of read_all, exceptions included
    gzFile f;
    explicit GzFile(gzFile g) : f(g) {}
    ~GzFile() { if (f) gzclose(f); }
    GzFile(const GzFile&) = delete;
    GzFile& operator=(const GzFile&) = delete;
};
// the largest record a network file can hold: the whole default architecture is 60 MB; 16 GiB leaves room for any DSL-legal network
// and still rejects a corrupt or crafted header before anything is allocated for it
const uint64_t kMaxRecordBytes = 1ull << 34;

bool read_all(const char* file_name, std::map<std::string, Rec>& recs, std::string& err)
{
    GzFile gz(gzopen(file_name, "rb"));
    gzFile f = gz.f;
    if (!f) { err = std::string("cannot open ") + file_name; return false; }
    for (;;) {
        int32_t h[5];
        int got = gzread(f, h, sizeof(h));
        if (got == 0) break;
        if (got != (int)sizeof(h)) { err = "truncated record header"; return false; }
        int typ = h[0], m = typ / 1000, o = (typ % 1000) / 100, p = (typ % 100) / 10, t = typ % 10;
        if (typ < 0 || m != 0 || o != 0 || p < 0 || p > 5 || t > 1 || h[1] < 0 || h[2] < 0 || h[4] < 1 || h[4] > 4096 || (h[3] != 0 && h[3] != 1)) {
            err = "not a little-endian Level-4 MAT record"; return false;
        }
        std::string name((size_t)h[4], '\0');
        if (gzread(f, &name[0], (unsigned)h[4]) != h[4]) { err = "truncated record name"; return false; }
        name = name.c_str();
        Rec r;
        r.p = p; r.text = t; r.rows = h[1]; r.cols = h[2];
        // rows * cols * element size: both factors are < 2^31, so the product of the first two fits 64 bits; the cap bounds the rest
        const uint64_t cells = (uint64_t)h[1] * (uint64_t)h[2];
        const uint64_t per = (uint64_t)kElem[p] * (h[3] ? 2 : 1);
        if (cells > kMaxRecordBytes / per) { err = "record " + name + " declares an implausible size"; return false; }
        const size_t n = (size_t)(cells * per);
        // read in pieces and grow as the bytes arrive: a header that promises more than the stream holds fails on the short read
        // without the promised amount ever having been allocated
        size_t off = 0;
        while (off < n) {
            const size_t piece = n - off > ((size_t)64 << 20) ? ((size_t)64 << 20) : n - off;
            r.data.resize(off + piece);
            if (gzread(f, r.data.data() + off, (unsigned)piece) != (int)piece) { err = "record " + name + " is truncated"; return false; }
            off += piece;
        }
        recs[name] = std::move(r);
    }
    return true;
}

double elem(const Rec& r, size_t i)
{
    const char* d = r.data.data();
    switch (r.p) {
        case 0: { double v; memcpy(&v, d + 8 * i, 8); return v; }
        case 1: { float v; memcpy(&v, d + 4 * i, 4); return v; }
        case 2: { int32_t v; memcpy(&v, d + 4 * i, 4); return v; }
        case 3: { int16_t v; memcpy(&v, d + 2 * i, 2); return v; }
        case 4: { uint16_t v; memcpy(&v, d + 2 * i, 2); return v; }
        default: return (unsigned char)d[i];
    }
}
std::string text_of(const Rec& r)
{
    std::string s;
    for (size_t i = 0; i < (size_t)r.rows * r.cols; ++i) { char c = (char)(int)elem(r, i); if (!c) break; s.push_back(c); }
    return s;
}

}  // namespace

bool save_to_file(UNet3d& model, const char* file_name)
{
    gzFile f = gzopen(file_name, "wb");
    if (!f) return false;
    bool ok = true;
    int32_t ch[2] = {model->in_count, model->out_count};
    ok = ok && put(f, "channels", 2, 0, 1, 2, ch);
    ok = ok && put_text(f, "architecture", model->architecture);
    // (the reference hands TIPL a uint32 shape, main.cpp:214; Level 4 has no 32-bit unsigned type code, and which code TIPL picks for
    // it is not visible here: written as int32 -- identical bytes for every real volume size -- and read back through elem() whatever the code)
    int32_t dim[3] = {(int32_t)model->dim[0], (int32_t)model->dim[1], (int32_t)model->dim[2]};
    ok = ok && put(f, "dimension", 2, 0, 1, 3, dim);
    float vs[3] = {model->voxel_size[0], model->voxel_size[1], model->voxel_size[2]};
    ok = ok && put(f, "voxel_size", 1, 0, 1, 3, vs);
    ok = ok && put_text(f, "fov_strategy", model->fov_strategy) && put_text(f, "preproc", model->preproc) &&
         put_text(f, "orientation", model->orientation) && put_text(f, "postproc", model->postproc);
    auto tr = model->get_training_errors(), te = model->get_testing_errors();
    ok = ok && put(f, "training_errors", 1, 0, 3, (int)(tr.size() / 3), tr.data()) && put(f, "testing_errors", 1, 0, 3, (int)(te.size() / 3), te.data());
    int id = 0;
    for (const auto& tensor : model->parameters()) {
        auto cpu_tensor = tensor.detach().to(torch::kCPU).to(torch::kFloat32).contiguous();
        int cols = (int)cpu_tensor.sizes().front();
        ok = ok && put(f, "tensor" + std::to_string(id), 1, 0, (int)(cpu_tensor.numel() / cols), cols, cpu_tensor.data_ptr<float>());
        ++id;
    }
    ok = (gzclose(f) == Z_OK) && ok;
    return ok;
}

static bool load_from_file_impl(UNet3d& model, const char* file_name);

// contract of main.cpp:157-206: false + error_msg, never an exception (a corrupt file must not take the caller down with bad_alloc)
bool load_from_file(UNet3d& model, const char* file_name)
{
    try { return load_from_file_impl(model, file_name); }
    catch (const std::exception& e) {
        if (!model.is_empty()) model->error_msg = e.what();
        std::cerr << e.what() << std::endl;
        return false;
    }
}

static bool load_from_file_impl(UNet3d& model, const char* file_name)
{
    std::map<std::string, Rec> recs;
    std::string err;
    auto fail = [&](const std::string& m) { if (!model.is_empty()) model->error_msg = m; std::cerr << m << std::endl; return false; };
    if (!read_all(file_name, recs, err)) return fail(err);
    if (!recs.count("channels") || !recs.count("architecture") || (size_t)recs["channels"].rows * recs["channels"].cols < 2) return fail("invalid format");
    const int in_c = (int)elem(recs["channels"], 0), out_c = (int)elem(recs["channels"], 1);
    const std::string architecture = text_of(recs["architecture"]);
    try { model = UNet3d(in_c, out_c, architecture); }
    catch (const std::exception& e) { return fail(e.what()); }
    if (!recs.count("dimension") || !recs.count("voxel_size") || (size_t)recs["dimension"].rows * recs["dimension"].cols < 3 ||
        (size_t)recs["voxel_size"].rows * recs["voxel_size"].cols < 3)
        return fail("invalid format");
    for (int k = 0; k < 3; ++k) {
        model->dim[k] = (unsigned)elem(recs["dimension"], k);
        model->voxel_size[k] = (float)elem(recs["voxel_size"], k);
    }
    if (recs.count("fov_strategy")) model->fov_strategy = text_of(recs["fov_strategy"]);
    if (recs.count("preproc")) model->preproc = text_of(recs["preproc"]);
    if (recs.count("orientation")) model->orientation = text_of(recs["orientation"]);
    if (recs.count("postproc")) model->postproc = text_of(recs["postproc"]);
    auto as_vector = [&](const char* key, auto& v) {
        v.clear();
        if (!recs.count(key)) return;
        const Rec& r = recs[key];
        for (size_t i = 0; i < (size_t)r.rows * r.cols; ++i) v.push_back((typename std::decay_t<decltype(v)>::value_type)elem(r, i));
    };
    as_vector("single_component_label", model->single_component_label);
    as_vector("testing_errors", model->testing_errors);
    as_vector("training_errors", model->training_errors);
    model->training_errors.resize(model->testing_errors.size());   // main.cpp:188
    model->train();
    int id = 0;
    torch::NoGradGuard ng;
    for (auto& tensor : model->parameters()) {
        const std::string key = "tensor" + std::to_string(id);
        if (recs.count(key + ".slope") || recs.count(key + ".inter") || (recs.count(key) && recs[key].p > 1))
            return fail(key + " is stored in TIPL's sloped (quantised) encoding (main.cpp:223-229), whose layout is defined by TIPL and not "
                              "available here: re-save the network with plain float tensors");
        const size_t have = recs.count(key) ? (size_t)recs[key].rows * recs[key].cols : 0;
        if (!recs.count(key) || have != (size_t)tensor.numel()) {
            std::ostringstream m;
            m << "tensor size mismatch at " << key << " " << have << " not the expected of size " << tensor.numel();
            return fail(m.str());
        }
        auto host = torch::empty({tensor.numel()}, torch::kFloat32);
        float* d = host.data_ptr<float>();
        const Rec& r = recs[key];
        if (r.p == 1) memcpy(d, r.data.data(), (size_t)tensor.numel() * 4);
        else for (int64_t i = 0; i < tensor.numel(); ++i) d[i] = (float)elem(r, (size_t)i);
        tensor.copy_(host.view(tensor.sizes()));
        ++id;
    }
    return true;
}
