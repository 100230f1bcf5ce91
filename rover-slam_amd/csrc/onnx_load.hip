// onnx_load.hip -- `.onnx` in, from C++.  Host code only (no kernels): the reference constructs its sessions from
// onnxmodel/superpoint.onnx (src/Extractors/SPextractor.cc:92-94) and the hard-coded onnxmodel/lightglue_sim.onnx
// (src/Matchers/lightglue_onnx.cpp:38); with this file rfe_load_weights takes those paths as they are -- no Python step, no `.rfew` file.
//
// A C++ restatement of rover-slam_amd/onnx_weights.py, function for function (the Python module stays the specification and the test oracle:
// tests/test_onnx_cpp.py requires the blob and the hyper-parameters of both to be BIT-identical on every exported graph, and the same
// refusals):
//   * a protobuf WIRE-FORMAT reader for ModelProto.graph.{initializer, node(+attributes)} -- no protobuf / onnx library; external-data
//     tensors are refused;
//   * weights: SuperPoint by parameter name (conv1a.weight ...) or by shape + file order; LightGlue by parameter name (cvg naming, Linear
//     weights that constant folding turned into anonymous transposed MatMul constants recovered through MatMul -> Add(<prefix>.bias) / Gemm)
//     or, when no names survive (onnx-simplifier output), by order of first use against the shape sequence of the published graph -- nothing
//     is guessed past the first disagreement; the interleaved Wqkv rows (head, dim, q|k|v) are re-ordered to the canonical (q|k|v, head, dim);
//   * hyper-parameters: NMS radius from the stride-1 MaxPools (5 of them = the published simple_nms), detection threshold from the one
//     Greater(x, 0 < c < 1), max_num_keypoints from TopK's k (through Min(k, count) -> topk_always), border from the Slice bounds of the index
//     vectors the border ScatterNDs write through, GridSample attributes; LightGlue depth / heads / filter threshold, control flow and
//     per-layer confidence heads reported.  What cannot be read is an ERROR: a loaded model never carries guessed hyper-parameters.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>
#include "rfe_internal.h"

namespace rfe {
namespace {

// ------------------------------------------------------------------------------------------------ canonical layouts (weights.py manifests)
struct SpLayer { const char* name; int cin, cout, k; };
const SpLayer SP_LAYERS_[] = {{"conv1a", 1, 64, 3}, {"conv1b", 64, 64, 3}, {"conv2a", 64, 64, 3}, {"conv2b", 64, 64, 3}, {"conv3a", 64, 128, 3}, {"conv3b", 128, 128, 3},
                              {"conv4a", 128, 128, 3}, {"conv4b", 128, 128, 3}, {"convPa", 128, 256, 3}, {"convPb", 256, 65, 1}, {"convDa", 128, 256, 3}, {"convDb", 256, 256, 1}};
struct ManEntry { std::string name; size_t off; std::vector<int64_t> shape; };
std::vector<ManEntry> lg_manifest(size_t* total) {
    std::vector<ManEntry> out;
    size_t off = 0;
    auto take = [&](const std::string& n, std::vector<int64_t> shape) {
        size_t c = 1;
        for (auto d : shape) c *= (size_t)d;
        out.push_back({n, off, shape});
        off += c;
    };
    take("posenc.Wr", {32, 2});
    for (int l = 0; l < LG_LAYERS; ++l) {
        const std::string p = "layers." + std::to_string(l) + ".";
        take(p + "self.Wqkv", {768, 256}); take(p + "self.bqkv", {768});
        take(p + "self.Wo", {256, 256}); take(p + "self.bo", {256});
        take(p + "self.W1", {512, 512}); take(p + "self.b1", {512});
        take(p + "self.ln_g", {512}); take(p + "self.ln_b", {512});
        take(p + "self.W2", {256, 512}); take(p + "self.b2", {256});
        take(p + "cross.Wqk", {256, 256}); take(p + "cross.bqk", {256});
        take(p + "cross.Wv", {256, 256}); take(p + "cross.bv", {256});
        take(p + "cross.Wo", {256, 256}); take(p + "cross.bo", {256});
        take(p + "cross.W1", {512, 512}); take(p + "cross.b1", {512});
        take(p + "cross.ln_g", {512}); take(p + "cross.ln_b", {512});
        take(p + "cross.W2", {256, 512}); take(p + "cross.b2", {256});
    }
    take("final_proj.W", {256, 256}); take("final_proj.b", {256});
    take("matchability.w", {256}); take("matchability.b", {1});
    *total = off;
    return out;
}

// ------------------------------------------------------------------------------------------------ protobuf wire format
struct Buf { const unsigned char* p; size_t n; };
struct Field { int fn, wt; uint64_t v; Buf b; };
bool varint(const Buf& b, size_t& i, uint64_t& r) {
    r = 0;
    for (int shift = 0; i < b.n && shift < 70; shift += 7) {
        const unsigned char c = b.p[i++];
        r |= (uint64_t)(c & 0x7F) << shift;
        if (!(c & 0x80)) return true;
    }
    return false;
}
// iterate the fields of one message; false on a malformed buffer
template <typename F>
bool fields(const Buf& b, F&& f) {
    size_t i = 0;
    while (i < b.n) {
        uint64_t key;
        if (!varint(b, i, key)) return false;
        Field fd{(int)(key >> 3), (int)(key & 7), 0, {nullptr, 0}};
        if (fd.wt == 0) { if (!varint(b, i, fd.v)) return false; }
        // every bound is written as "length > bytes left" (i <= b.n holds here): a 10-byte varint length near 2^64 must not wrap `i + ln`
        else if (fd.wt == 1) { if (b.n - i < 8) return false; fd.b = {b.p + i, 8}; i += 8; }
        else if (fd.wt == 2) { uint64_t ln; if (!varint(b, i, ln) || ln > (uint64_t)(b.n - i)) return false; fd.b = {b.p + i, (size_t)ln}; i += (size_t)ln; }
        else if (fd.wt == 5) { if (b.n - i < 4) return false; fd.b = {b.p + i, 4}; i += 4; }
        else return false;
        f(fd);
    }
    return true;
}
std::string str(const Buf& b) { return std::string((const char*)b.p, b.n); }
int64_t sgn(uint64_t v) { return (int64_t)v; }   // two's complement, as protobuf int64 varints are

// A tensor: float32 payloads in f (what the weights are), every integer / bool payload widened to int64 in i, float64 in d.
struct Tensor {
    std::string name; int dtype = 1; std::vector<int64_t> dims; bool has_dims = false;
    std::vector<float> f; std::vector<int64_t> i; std::vector<double> d;
    bool supported = true; bool external = false;
    size_t size() const { return dtype == 1 ? f.size() : dtype == 11 ? d.size() : i.size(); }
    int ndim() const { return has_dims ? (int)dims.size() : (size() == 1 ? 0 : 1); }   // no dims: rank 0 when one element (Constant scalars), else 1-D
    bool is_float() const { return dtype == 1 || dtype == 11 || dtype == 10; }
    bool shape_is(std::initializer_list<int64_t> s) const {
        if (!has_dims) return s.size() == 1 && (int64_t)size() == *s.begin() && size() != 1;
        return dims.size() == s.size() && std::equal(dims.begin(), dims.end(), s.begin());
    }
    double scalar() const { return dtype == 1 ? (double)f[0] : dtype == 11 ? d[0] : (double)i[0]; }
};

bool parse_tensor(const Buf& b, Tensor& t) {
    Buf raw{nullptr, 0};
    bool has_raw = false;
    std::vector<float> floats; std::vector<int64_t> int64s, int32s; std::vector<double> doubles;
    auto packed = [](const Field& fd, std::vector<int64_t>& out) {
        if (fd.wt == 2) { size_t j = 0; uint64_t v; while (j < fd.b.n && varint(fd.b, j, v)) out.push_back(sgn(v)); }
        else out.push_back(sgn(fd.v));
    };
    const bool ok = fields(b, [&](const Field& fd) {
        switch (fd.fn) {
            case 1: packed(fd, t.dims); t.has_dims = true; break;
            case 2: t.dtype = (int)fd.v; break;
            case 8: t.name = str(fd.b); break;
            case 9: raw = fd.b; has_raw = true; break;
            case 4: if (fd.wt == 2 || fd.wt == 5) { const size_t c = fd.b.n / 4; const size_t o = floats.size(); floats.resize(o + c); memcpy(floats.data() + o, fd.b.p, c * 4); } break;
            case 7: packed(fd, int64s); break;
            case 5: packed(fd, int32s); break;
            case 10: if (fd.wt == 2 || fd.wt == 1) { const size_t c = fd.b.n / 8; const size_t o = doubles.size(); doubles.resize(o + c); memcpy(doubles.data() + o, fd.b.p, c * 8); } break;
            case 13: t.external = true; break;
            default: break;
        }
    });
    if (!ok) return false;
    auto from_raw = [&](size_t width, auto conv) { const size_t c = raw.n / width; for (size_t k = 0; k < c; ++k) conv(raw.p + k * width); };
    switch (t.dtype) {
        case 1: if (has_raw) { t.f.resize(raw.n / 4); memcpy(t.f.data(), raw.p, t.f.size() * 4); } else t.f = floats; break;
        case 11: if (has_raw) { t.d.resize(raw.n / 8); memcpy(t.d.data(), raw.p, t.d.size() * 8); } else t.d = doubles; break;
        case 7: if (has_raw) from_raw(8, [&](const unsigned char* p) { int64_t v; memcpy(&v, p, 8); t.i.push_back(v); }); else t.i = int64s; break;
        case 6: if (has_raw) from_raw(4, [&](const unsigned char* p) { int32_t v; memcpy(&v, p, 4); t.i.push_back(v); }); else t.i = int32s; break;
        case 9: case 2: if (has_raw) from_raw(1, [&](const unsigned char* p) { t.i.push_back(*p); }); else t.i = int32s; break;
        case 3: if (has_raw) from_raw(1, [&](const unsigned char* p) { t.i.push_back((signed char)*p); }); else t.i = int32s; break;
        case 5: if (has_raw) from_raw(2, [&](const unsigned char* p) { int16_t v; memcpy(&v, p, 2); t.i.push_back(v); }); else t.i = int32s; break;
        case 12: if (has_raw) from_raw(4, [&](const unsigned char* p) { uint32_t v; memcpy(&v, p, 4); t.i.push_back(v); }); break;
        case 13: if (has_raw) from_raw(8, [&](const unsigned char* p) { uint64_t v; memcpy(&v, p, 8); t.i.push_back((int64_t)v); }); break;
        case 10: t.supported = true; break;   // float16: carried without values (never a weight or a hyper-parameter of the fp32 exports)
        default: t.supported = false; break;
    }
    // a tensor whose element count is not the product of its dims (a damaged or hostile file) is dropped: every later shape test may then trust dims
    if (t.has_dims && t.dtype != 10 && t.supported) {
        unsigned long long prod = 1;
        bool ok_dims = true;
        for (auto d : t.dims) { if (d < 0 || (d > 0 && prod > (1ull << 40) / (unsigned long long)d)) { ok_dims = false; break; } prod *= (unsigned long long)d; }
        if (!ok_dims || prod != (unsigned long long)t.size()) t.supported = false;
    }
    return true;
}

struct Attr {
    std::string name;
    enum Kind { NONE, INT, FLOAT, STR, TENSOR, INTS, FLOATS, GRAPH } kind = NONE;
    int64_t i = 0; float f = 0.f; std::string s; Tensor t; std::vector<int64_t> ints; std::vector<float> floats;
};
bool parse_attr(const Buf& b, Attr& a) {
    bool terr = false;
    const bool ok = fields(b, [&](const Field& fd) {
        if (fd.fn == 1) a.name = str(fd.b);
        else if (fd.fn == 2 && fd.wt == 5) { memcpy(&a.f, fd.b.p, 4); a.kind = Attr::FLOAT; }
        else if (fd.fn == 3 && fd.wt == 0) { a.i = sgn(fd.v); a.kind = Attr::INT; }
        else if (fd.fn == 4 && fd.wt == 2) { a.s = str(fd.b); a.kind = Attr::STR; }
        else if (fd.fn == 5 && fd.wt == 2) { if (!parse_tensor(fd.b, a.t)) terr = true; a.kind = Attr::TENSOR; }
        else if (fd.fn == 6 && fd.wt == 2) { a.kind = Attr::GRAPH; }
        else if (fd.fn == 7) { if (fd.wt == 2) { const size_t c = fd.b.n / 4, o = a.floats.size(); a.floats.resize(o + c); memcpy(a.floats.data() + o, fd.b.p, c * 4); } else if (fd.wt == 5) { float v; memcpy(&v, fd.b.p, 4); a.floats.push_back(v); } }
        else if (fd.fn == 8) { if (fd.wt == 2) { size_t j = 0; uint64_t v; while (j < fd.b.n && varint(fd.b, j, v)) a.ints.push_back(sgn(v)); } else a.ints.push_back(sgn(fd.v)); }
    });
    if (a.kind == Attr::NONE && !a.ints.empty()) a.kind = Attr::INTS;
    if (a.kind == Attr::NONE && !a.floats.empty()) a.kind = Attr::FLOATS;
    return ok && !terr;
}

struct Node {
    std::string op, name; std::vector<std::string> inputs, outputs; std::vector<Attr> attrs;
    const Attr* attr(const char* n) const { for (const auto& a : attrs) if (a.name == n && a.kind != Attr::NONE) return &a; return nullptr; }
    int64_t attr_int(const char* n, int64_t dflt) const { const Attr* a = attr(n); return a && a->kind == Attr::INT ? a->i : dflt; }
    std::string attr_str(const char* n, const char* dflt) const { const Attr* a = attr(n); return a && a->kind == Attr::STR ? a->s : std::string(dflt); }
};
struct Model {
    std::map<std::string, Tensor> inits; std::vector<std::string> order;   // initializers by name, and in file order (the name-free fallbacks walk it)
    std::vector<Node> nodes;
    const Tensor* init(const std::string& n) const { auto it = inits.find(n); return it == inits.end() ? nullptr : &it->second; }
};

bool read_model(const std::string& path, Model& m, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open model file " + path; return false; }
    std::vector<unsigned char> data;
    unsigned char chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) data.insert(data.end(), chunk, chunk + got);
    fclose(f);
    bool bad = false, saw_graph = false;
    std::string ext;
    const bool ok = fields(Buf{data.data(), data.size()}, [&](const Field& fd) {
        if (fd.fn != 7 || fd.wt != 2) return;                    // ModelProto.graph
        saw_graph = true;
        if (!fields(fd.b, [&](const Field& g) {
                if (g.fn == 5 && g.wt == 2) {                    // GraphProto.initializer
                    Tensor t;
                    if (!parse_tensor(g.b, t)) { bad = true; return; }
                    if (t.external) { ext = t.name; return; }
                    if (t.supported) { if (!m.inits.count(t.name)) m.order.push_back(t.name); m.inits[t.name] = std::move(t); }
                } else if (g.fn == 1 && g.wt == 2) {             // GraphProto.node
                    Node n;
                    if (!fields(g.b, [&](const Field& nf) {
                            if (nf.fn == 1) n.inputs.push_back(str(nf.b));
                            else if (nf.fn == 2) n.outputs.push_back(str(nf.b));
                            else if (nf.fn == 3) n.name = str(nf.b);
                            else if (nf.fn == 4) n.op = str(nf.b);
                            else if (nf.fn == 5 && nf.wt == 2) { Attr a; if (!parse_attr(nf.b, a)) bad = true; if (a.kind != Attr::NONE) n.attrs.push_back(std::move(a)); }
                        })) bad = true;
                    m.nodes.push_back(std::move(n));
                }
            })) bad = true;
    });
    if (!ext.empty()) { err = path + ": tensor '" + ext + "': external data is not supported"; return false; }
    if (!ok || bad || !saw_graph) { err = path + ": not an ONNX ModelProto (protobuf wire format damaged or no graph)"; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------------ graph view
const char* PASS_THROUGH[] = {"Cast", "Reshape", "Unsqueeze", "Squeeze", "Identity", "Flatten"};
bool pass_through(const std::string& op) { for (auto p : PASS_THROUGH) if (op == p) return true; return false; }
struct View {
    const Model& m;
    std::map<std::string, const Tensor*> consts;
    std::map<std::string, const Node*> producer;
    std::vector<Tensor> made;   // value_float / value_int / value_ints ... attributes turned into tensors
    explicit View(const Model& mm) : m(mm) {
        made.reserve(mm.nodes.size());
        for (const auto& kv : m.inits) consts[kv.first] = &kv.second;
        for (const auto& n : m.nodes) {
            if (n.op == "Constant" && !n.outputs.empty()) {
                for (const char* key : {"value", "value_float", "value_int", "value_floats", "value_ints"}) {
                    const Attr* a = n.attr(key);
                    if (!a || a->kind == Attr::STR || a->kind == Attr::GRAPH) continue;
                    if (a->kind == Attr::TENSOR) { if (a->t.supported) consts[n.outputs[0]] = &a->t; }
                    else {
                        Tensor t;
                        if (a->kind == Attr::FLOAT) { t.dtype = 1; t.f = {a->f}; }
                        else if (a->kind == Attr::INT) { t.dtype = 7; t.i = {a->i}; }
                        else if (a->kind == Attr::FLOATS) { t.dtype = 1; t.f = a->floats; t.has_dims = true; t.dims = {(int64_t)a->floats.size()}; }
                        else { t.dtype = 7; t.i = a->ints; t.has_dims = true; t.dims = {(int64_t)a->ints.size()}; }
                        made.push_back(std::move(t));
                        consts[n.outputs[0]] = &made.back();
                    }
                    break;
                }
            }
            for (const auto& o : n.outputs) if (!o.empty()) producer[o] = &n;
        }
    }
    const Node* prod(const std::string& n) const { auto it = producer.find(n); return it == producer.end() ? nullptr : it->second; }
    // the constant a tensor name resolves to through shape-only pass-through ops, or null
    const Tensor* const_of(std::string name, int depth = 6) const {
        while (depth >= 0) {
            auto it = consts.find(name);
            if (it != consts.end()) return it->second;
            const Node* n = prod(name);
            if (!n || !pass_through(n->op) || n->inputs.empty()) return nullptr;
            name = n->inputs[0]; --depth;
        }
        return nullptr;
    }
};
// a one-element constant: (found, is_float, value)
struct Scalar { bool ok = false, is_float = false; double v = 0; };
Scalar scalar_of(const Tensor* t) {
    Scalar s;
    if (!t || t->size() != 1 || t->dtype == 10) return s;
    s.ok = true; s.is_float = t->dtype == 1 || t->dtype == 11; s.v = t->scalar();
    return s;
}
std::vector<const Node*> ancestors(const View& v, const std::string& name, size_t limit = 400) {
    std::set<const Node*> seen;
    std::vector<const Node*> out;
    std::vector<std::string> todo{name};
    size_t head = 0;
    while (head < todo.size() && out.size() < limit) {
        const std::string t = todo[head++];
        const Node* n = v.prod(t);
        if (!n || seen.count(n)) continue;
        seen.insert(n);
        out.push_back(n);
        for (const auto& i : n->inputs) if (!i.empty()) todo.push_back(i);
    }
    return out;
}
bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
std::string fmt(double v) { char b[64]; snprintf(b, sizeof(b), "%.9g", v); return b; }

// ------------------------------------------------------------------------------------------------ hyper-parameters
struct SpRead { bool has_k = false, has_thr = false, has_r = false, has_b = false, has_always = false; int k = 0, r = 0, b = 0, always = 0; float thr = 0.f; };
std::vector<std::string> read_sp_hparams(const Model& m, SpRead& hp) {
    View v(m);
    std::vector<std::string> problems;
    // ---- NMS radius
    struct Pool { int64_t kh, kw; std::vector<int64_t> pads; bool operator<(const Pool& o) const { return std::tie(kh, kw, pads) < std::tie(o.kh, o.kw, o.pads); } };
    std::vector<Pool> nms;
    for (const auto& n : m.nodes) {
        if (n.op != "MaxPool") continue;
        const Attr* ks = n.attr("kernel_shape");
        const Attr* st = n.attr("strides");
        std::vector<int64_t> strides = st && st->kind == Attr::INTS ? st->ints : std::vector<int64_t>{1, 1};
        if (ks && ks->kind == Attr::INTS && ks->ints.size() == 2 && strides == std::vector<int64_t>{1, 1}) {
            const Attr* pd = n.attr("pads");
            nms.push_back({ks->ints[0], ks->ints[1], pd && pd->kind == Attr::INTS ? pd->ints : std::vector<int64_t>{0, 0, 0, 0}});
        }
    }
    if (nms.empty()) problems.push_back("nms_radius: no stride-1 MaxPool in the graph (NMS not exported as max_pool2d?)");
    else {
        std::set<Pool> kinds(nms.begin(), nms.end());
        const Pool& p0 = nms[0];
        const int64_t r = (p0.kh - 1) / 2;
        bool pads_r = !p0.pads.empty();
        for (auto q : p0.pads) pads_r &= q == r;
        if (kinds.size() != 1 || p0.kh != p0.kw || p0.kh % 2 == 0 || !pads_r)
            problems.push_back("nms_radius: the stride-1 MaxPools disagree or are not (2r+1) windows padded by r");
        else if (nms.size() != 5)
            problems.push_back("nms_radius: " + std::to_string(nms.size()) + " stride-1 MaxPools, the published simple_nms (2 suppression rounds) has 5 -- the kernels implement exactly that recurrence");
        else { hp.has_r = true; hp.r = (int)r; }
    }
    // ---- detection threshold
    std::set<float> thr;
    for (const auto& n : m.nodes)
        if (n.op == "Greater" && n.inputs.size() == 2) {
            const Scalar s = scalar_of(v.const_of(n.inputs[1]));
            if (s.ok && s.is_float && s.v > 0.0 && s.v < 1.0) thr.insert((float)s.v);
        }
    if (thr.size() == 1) { hp.has_thr = true; hp.thr = *thr.begin(); }
    else {
        std::string l;
        for (float t : thr) l += (l.empty() ? "" : ", ") + fmt(t);
        problems.push_back("detection_threshold: expected exactly one Greater(x, c) with 0 < c < 1, found constants [" + l + "]");
    }
    // ---- max_num_keypoints
    std::vector<const Node*> topk;
    for (const auto& n : m.nodes) if (n.op == "TopK") topk.push_back(&n);
    if (topk.size() != 1)
        problems.push_back("max_keypoints: " + std::to_string(topk.size()) + " TopK nodes (an export without max_num_keypoints returns every candidate; the C ABI needs a capacity)");
    else {
        const Node* t = topk[0];
        if (t->attr_int("largest", 1) != 1) problems.push_back("max_keypoints: TopK with largest = 0");
        std::string name = t->inputs.size() > 1 ? t->inputs[1] : "";
        bool have = false; int64_t k = 0;
        for (int it = 0; it < 8; ++it) {                 // K input: constant, or Min(constant, dynamic count) behind shape ops
            const Scalar c = scalar_of(v.const_of(name));
            if (c.ok) { have = true; k = (int64_t)c.v; break; }
            const Node* n = v.prod(name);
            if (!n) break;
            if (n->op == "Min") {                          // torch.topk(scores, min(k, n)): applied whatever the candidate count -> always sorted
                std::vector<int64_t> cs;
                for (const auto& i : n->inputs) { const Scalar c2 = scalar_of(v.const_of(i)); if (c2.ok) cs.push_back((int64_t)c2.v); }
                if (cs.size() == 1) { have = true; k = cs[0]; }
                hp.has_always = true; hp.always = 1;
                break;
            }
            if (pass_through(n->op) && !n->inputs.empty()) name = n->inputs[0];
            else break;
        }
        if (!have || k < 1) problems.push_back("max_keypoints: the K input of TopK does not resolve to a constant (or Min(constant, count))");
        else {
            hp.has_k = true; hp.k = (int)k;
            if (!hp.has_always) { hp.has_always = true; hp.always = 0; }   // constant k: the published top_k_keypoints
        }
    }
    // ---- border
    std::set<int64_t> pads;
    int nscatter = 0;
    for (const auto& n : m.nodes) {
        if (n.op != "ScatterND" || n.inputs.size() < 3) continue;
        const Tensor* upd = v.const_of(n.inputs[2]);
        std::set<int64_t> here;
        for (const Node* a : ancestors(v, n.inputs[1])) {
            if (a->op != "Slice" || a->inputs.size() < 3) continue;
            const Node* src = v.prod(a->inputs[0]);
            while (src && pass_through(src->op) && !src->inputs.empty()) src = v.prod(src->inputs[0]);
            if (!src || src->op != "Range") continue;      // only slices of an index vector (arange(H) / arange(W)) describe the border
            const Scalar st = scalar_of(v.const_of(a->inputs[1])), en = scalar_of(v.const_of(a->inputs[2]));
            if (st.ok && st.v == 0 && en.ok && en.v > 0 && en.v <= 64) here.insert((int64_t)en.v);
            else if (st.ok && st.v >= -64 && st.v < 0) here.insert((int64_t)-st.v);
        }
        if (!here.empty()) {
            ++nscatter;
            pads.insert(here.begin(), here.end());
            if (upd && upd->size()) {
                bool all_m1 = true;
                for (size_t q = 0; q < upd->size(); ++q) all_m1 &= (upd->dtype == 1 ? (double)upd->f[q] : upd->dtype == 11 ? upd->d[q] : (double)upd->i[q]) == -1.0;
                if (!all_m1) problems.push_back("remove_borders: a border ScatterND writes something other than -1");
            }
        }
    }
    if (nscatter == 0) problems.push_back("remove_borders: no ScatterND over sliced index vectors found (border handled in another form, or not at all)");
    else if (pads.size() != 1) problems.push_back("remove_borders: the border slices disagree");
    else { hp.has_b = true; hp.b = (int)*pads.begin(); }
    // ---- descriptor sampling
    std::vector<const Node*> gs;
    for (const auto& n : m.nodes) if (n.op == "GridSample") gs.push_back(&n);
    if (gs.size() != 1) problems.push_back("grid_sample: " + std::to_string(gs.size()) + " GridSample nodes (descriptor sampling exported in another form)");
    else {
        const std::string mode = gs[0]->attr_str("mode", "bilinear"), pm = gs[0]->attr_str("padding_mode", "zeros");
        const int64_t ac = gs[0]->attr_int("align_corners", 0);
        if ((mode != "bilinear" && mode != "linear") || ac != 1 || pm != "zeros")
            problems.push_back("grid_sample: mode / align_corners / padding_mode = " + mode + " / " + std::to_string(ac) + " / " + pm + "; the kernels implement bilinear / 1 / zeros");
    }
    return problems;
}

// every Linear parameter set, once, in order of first use: (W [out,in] row-major, bias or empty, where)
struct Lin { std::vector<float> w; int64_t out = 0, in = 0; std::vector<float> b; bool has_b = false; std::string where; };
std::vector<Lin> linears_in_order_of_use(const Model& m) {
    std::map<std::string, std::vector<const Node*>> consumers;
    for (const auto& n : m.nodes) for (const auto& i : n.inputs) consumers[i].push_back(&n);
    std::set<std::string> seen;
    std::vector<Lin> out;
    auto transposed = [](const Tensor& t) {   // [r,c] -> [c,r]
        const int64_t r = t.dims[0], c = t.dims[1];
        std::vector<float> o((size_t)(r * c));
        for (int64_t a = 0; a < r; ++a) for (int64_t b = 0; b < c; ++b) o[(size_t)(b * r + a)] = t.f[(size_t)(a * c + b)];
        return o;
    };
    for (size_t idx = 0; idx < m.nodes.size(); ++idx) {
        const Node& n = m.nodes[idx];
        if (n.op == "Gemm" && n.inputs.size() >= 2) {
            const Tensor* w = m.init(n.inputs[1]);
            if (!w || w->dtype != 1 || !w->has_dims || w->dims.size() != 2 || seen.count(n.inputs[1])) continue;
            seen.insert(n.inputs[1]);
            Lin l;
            if (n.attr_int("transB", 0) == 1) { l.w = w->f; l.out = w->dims[0]; l.in = w->dims[1]; }
            else { l.w = transposed(*w); l.out = w->dims[1]; l.in = w->dims[0]; }
            if (n.inputs.size() >= 3) { const Tensor* b = m.init(n.inputs[2]); if (b && b->dtype == 1) { l.b = b->f; l.has_b = true; } }
            l.where = "node " + std::to_string(idx) + " Gemm(" + n.inputs[1] + ")";
            out.push_back(std::move(l));
        } else if (n.op == "MatMul") {
            std::vector<std::string> wn;
            for (const auto& i : n.inputs) { const Tensor* t = m.init(i); if (t && t->dtype == 1 && t->has_dims && t->dims.size() == 2) wn.push_back(i); }
            if (wn.size() != 1 || seen.count(wn[0])) continue;
            seen.insert(wn[0]);
            const Tensor* w = m.init(wn[0]);
            Lin l;
            l.w = transposed(*w); l.out = w->dims[1]; l.in = w->dims[0];
            for (const Node* c : consumers[n.outputs.empty() ? "" : n.outputs[0]])
                if (c->op == "Add") {
                    std::vector<const Tensor*> cb;
                    for (const auto& i : c->inputs) { const Tensor* t = m.init(i); if (t && t->ndim() == 1) cb.push_back(t); }
                    if (cb.size() == 1 && cb[0]->dtype == 1) { l.b = cb[0]->f; l.has_b = true; }
                }
            l.where = "node " + std::to_string(idx) + " MatMul(" + wn[0] + ")";
            out.push_back(std::move(l));
        }
    }
    return out;
}
int distinct_linears(const std::vector<Lin>& lins, int64_t out, int64_t in) {
    std::set<std::string> s;
    for (const auto& l : lins) if (l.out == out && l.in == in) s.insert(std::string((const char*)l.w.data(), l.w.size() * 4));
    return (int)s.size();
}

struct LgRead { bool has_layers = false, has_heads = false, has_thr = false; int layers = 0, heads = 0; float thr = 0.f; };
std::vector<std::string> read_lg_hparams(const Model& m, LgRead& hp) {
    View v(m);
    std::vector<std::string> problems;
    std::set<int> idx;
    for (const auto& kv : m.inits)
        if (kv.first.rfind("transformers.", 0) == 0) {
            const size_t a = 13, b = kv.first.find('.', a);
            const std::string d = kv.first.substr(a, b == std::string::npos ? std::string::npos : b - a);
            if (!d.empty() && std::all_of(d.begin(), d.end(), [](char c) { return c >= '0' && c <= '9'; })) idx.insert(atoi(d.c_str()));
        }
    const std::vector<Lin> lins = linears_in_order_of_use(m);
    const int n_qkv = distinct_linears(lins, 768, 256);
    if (!idx.empty()) {
        hp.has_layers = true; hp.layers = *idx.rbegin() + 1;
        if (n_qkv && n_qkv != hp.layers) problems.push_back("layers: parameter names say " + std::to_string(hp.layers) + ", the graph holds " + std::to_string(n_qkv) + " Wqkv (768 x 256) Linears");
    } else if (n_qkv) { hp.has_layers = true; hp.layers = n_qkv; }
    else problems.push_back("layers: neither `transformers.{i}.` names nor 768 x 256 Linears found");
    std::set<int64_t> heads;
    for (const auto& n : m.nodes) {
        if (n.op != "Reshape" || n.inputs.size() < 2) continue;
        std::vector<int64_t> shp;
        const Tensor* st = v.const_of(n.inputs[1]);
        if (st) { if (st->dtype == 1 || st->dtype == 11) continue; shp = st->i; }
        else {                                            // shape assembled from dynamic dims: Concat of constants and Gather(Shape) pieces
            const Node* c = v.prod(n.inputs[1]);
            if (!c || c->op != "Concat") continue;
            for (const auto& i : c->inputs) {
                const Tensor* p = v.const_of(i);
                if (p && p->dtype != 1 && p->dtype != 11) shp.insert(shp.end(), p->i.begin(), p->i.end());
                else shp.push_back(-7);
            }
        }
        const size_t L = shp.size();
        if (L >= 4 && shp[L - 1] == 3 && shp[L - 3] > 0 && (shp[L - 2] == 64 || shp[L - 2] == -1)) heads.insert(shp[L - 3]);   // Wqkv(x).unflatten(-1, (heads, -1, 3))
        else if (L >= 4 && shp[L - 1] == 64 && shp[L - 2] > 0) heads.insert(shp[L - 2]);                                          // to_qk / to_v: unflatten(-1, (heads, -1))
    }
    if (heads.size() == 1) { hp.has_heads = true; hp.heads = (int)*heads.begin(); }
    else problems.push_back("heads: head-split Reshape constants are ambiguous or absent");
    std::set<float> thr;
    for (const auto& n : m.nodes)
        if (n.op == "Greater" && n.inputs.size() == 2) {
            const Scalar s = scalar_of(v.const_of(n.inputs[1]));
            if (s.ok && s.is_float && s.v > 0.0 && s.v < 1.0) thr.insert((float)s.v);
        }
    if (thr.size() == 1) { hp.has_thr = true; hp.thr = *thr.begin(); }
    else problems.push_back("filter_threshold: expected exactly one Greater(x, c) with 0 < c < 1, found " + std::to_string(thr.size()));
    std::set<std::string> ctl;
    for (const auto& n : m.nodes) if (n.op == "If" || n.op == "Loop" || n.op == "Scan") ctl.insert(n.op);
    if (!ctl.empty()) {
        std::string l;
        for (const auto& c : ctl) l += (l.empty() ? "'" : ", '") + c + "'";
        problems.push_back("control flow [" + l + "] in the graph (early exit / point pruning): only the fixed-depth export is implemented");
    }
    const int n_conf = distinct_linears(lins, 1, 256);
    if (n_conf > 1) problems.push_back(std::to_string(n_conf) + " Linear(256 -> 1) heads: per-layer token-confidence heads of the early-exit graph are present (the fixed-depth export keeps the last matchability head only)");
    return problems;
}

// ------------------------------------------------------------------------------------------------ weights
bool convert_sp(const Model& m, const std::string& path, std::vector<float>& blob, std::string& err) {
    blob.assign((size_t)SP_COUNT, 0.f);
    std::vector<const Tensor*> convs, biases;               // fallback when parameter names were not preserved: in file order
    for (const auto& nm : m.order) {
        const Tensor& t = m.inits.at(nm);
        if (t.dtype != 1) continue;
        if (t.ndim() == 4) convs.push_back(&t);
        else if (t.ndim() == 1) biases.push_back(&t);
    }
    std::string missing;
    size_t off = 0;
    for (size_t li = 0; li < 12; ++li) {
        const SpLayer& L = SP_LAYERS_[li];
        for (int leaf = 0; leaf < 2; ++leaf) {
            const std::vector<int64_t> shape = leaf == 0 ? std::vector<int64_t>{L.cout, L.cin, L.k, L.k} : std::vector<int64_t>{L.cout};
            size_t cnt = 1;
            for (auto d : shape) cnt *= (size_t)d;
            const std::string key = std::string(L.name) + (leaf == 0 ? ".weight" : ".bias");
            const Tensor* arr = m.init(key);
            if (arr && arr->dtype != 1) arr = nullptr;
            if (!arr) {
                // layers sharing a shape (conv1b, conv2a, conv2b ...) are told apart by their order in the file
                size_t same_before = 0;
                for (size_t q = 0; q < li; ++q) {
                    const SpLayer& Q = SP_LAYERS_[q];
                    const std::vector<int64_t> qs = leaf == 0 ? std::vector<int64_t>{Q.cout, Q.cin, Q.k, Q.k} : std::vector<int64_t>{Q.cout};
                    same_before += qs == shape;
                }
                std::vector<const Tensor*> cand;
                for (const Tensor* t : (leaf == 0 ? convs : biases)) if (t->has_dims ? t->dims == shape : (shape.size() == 1 && (int64_t)t->size() == shape[0])) cand.push_back(t);
                arr = same_before < cand.size() ? cand[same_before] : nullptr;
            }
            const bool shape_ok = arr && arr->f.size() == cnt && (arr->has_dims ? arr->dims == shape : shape.size() == 1);
            if (!shape_ok) missing += (missing.empty() ? "" : ", ") + key;
            else memcpy(blob.data() + off, arr->f.data(), cnt * 4);
            off += cnt;
        }
    }
    if (!missing.empty()) { err = path + ": cannot place SuperPoint tensors [" + missing + "]"; return false; }
    return true;
}

void deinterleave_qkv(std::vector<float>& w, std::vector<float>& b) {
    // published LightGlue SelfBlock: output row h*192 + d*3 + t  ->  canonical row t*256 + h*64 + d
    std::vector<float> w2(w.size()), b2(b.size());
    for (int h = 0; h < 4; ++h) for (int d = 0; d < 64; ++d) for (int t = 0; t < 3; ++t) {
        const int src = h * 192 + d * 3 + t, dst = t * 256 + h * 64 + d;
        memcpy(w2.data() + (size_t)dst * 256, w.data() + (size_t)src * 256, 256 * 4);
        b2[dst] = b[src];
    }
    w.swap(w2); b.swap(b2);
}

struct Placed { std::vector<float> v; std::vector<int64_t> shape; };
bool fill_lg_blob(const std::map<std::string, Placed>& t, std::vector<float>& blob, std::vector<std::string>& missing) {
    size_t total;
    const std::vector<ManEntry> man = lg_manifest(&total);
    blob.assign(total, 0.f);
    for (const auto& e : man) {
        auto it = t.find(e.name);
        size_t cnt = 1;
        for (auto d : e.shape) cnt *= (size_t)d;
        // the copy is sized by the MANIFEST; a placed tensor must have exactly that shape and exactly that many values
        if (it == t.end() || it->second.shape != e.shape || it->second.v.size() != cnt) { missing.push_back(e.name + (it == t.end() ? "" : " (shape)")); continue; }
        memcpy(blob.data() + e.off, it->second.v.data(), cnt * 4);
    }
    return missing.empty();
}

bool convert_lg_by_structure(const Model& m, const std::string& path, std::vector<float>& blob, std::string& err) {
    const std::vector<Lin> lins = linears_in_order_of_use(m);
    // LayerNorm(512) (gamma, beta) pairs in order of first use
    std::map<std::string, std::vector<const Node*>> consumers;
    for (const auto& n : m.nodes) for (const auto& i : n.inputs) consumers[i].push_back(&n);
    std::vector<std::pair<const Tensor*, const Tensor*>> lns;
    std::set<std::string> seen;
    auto is512 = [&](const std::string& nm) { const Tensor* t = m.init(nm); return t && t->dtype == 1 && t->ndim() == 1 && t->size() == 512 ? t : nullptr; };
    for (const auto& n : m.nodes) {
        if (n.op == "LayerNormalization" && n.inputs.size() >= 3 && is512(n.inputs[1]) && is512(n.inputs[2])) {   // beta validated exactly like gamma
            if (!seen.count(n.inputs[1])) { seen.insert(n.inputs[1]); lns.push_back({m.init(n.inputs[1]), m.init(n.inputs[2])}); }
        } else if (n.op == "Mul") {
            std::vector<std::string> g;
            for (const auto& i : n.inputs) if (is512(i)) g.push_back(i);
            if (g.size() == 1 && !seen.count(g[0]))
                for (const Node* c : consumers[n.outputs.empty() ? "" : n.outputs[0]]) {
                    std::vector<std::string> bb;
                    if (c->op == "Add") for (const auto& i : c->inputs) if (is512(i)) bb.push_back(i);
                    if (bb.size() == 1) { seen.insert(g[0]); lns.push_back({m.init(g[0]), m.init(bb[0])}); break; }
                }
        }
    }
    struct Exp { std::string w, b; int64_t out, in; bool has_bias; };
    std::vector<Exp> expect{{"posenc.Wr", "", 32, 2, false}};
    for (int l = 0; l < LG_LAYERS; ++l) {
        const std::string p = "layers." + std::to_string(l) + ".";
        expect.push_back({p + "self.Wqkv", p + "self.bqkv", 768, 256, true}); expect.push_back({p + "self.Wo", p + "self.bo", 256, 256, true});
        expect.push_back({p + "self.W1", p + "self.b1", 512, 512, true}); expect.push_back({p + "self.W2", p + "self.b2", 256, 512, true});
        expect.push_back({p + "cross.Wqk", p + "cross.bqk", 256, 256, true}); expect.push_back({p + "cross.Wv", p + "cross.bv", 256, 256, true});
        expect.push_back({p + "cross.Wo", p + "cross.bo", 256, 256, true}); expect.push_back({p + "cross.W1", p + "cross.b1", 512, 512, true});
        expect.push_back({p + "cross.W2", p + "cross.b2", 256, 512, true});
    }
    expect.push_back({"final_proj.W", "final_proj.b", 256, 256, true}); expect.push_back({"matchability.w", "matchability.b", 1, 256, true});
    std::vector<std::string> problems;
    if (lins.size() != expect.size())
        problems.push_back(std::to_string(lins.size()) + " Linear layers in the graph, the published " + std::to_string(LG_LAYERS) + "-layer LightGlue has " + std::to_string(expect.size()));
    std::map<std::string, Placed> t;
    for (size_t pos = 0; pos < expect.size(); ++pos) {
        const Exp& e = expect[pos];
        if (pos >= lins.size()) { problems.push_back("Linear #" + std::to_string(pos) + " (" + e.w + "): graph has no more Linear layers"); break; }
        Lin l = lins[pos];
        if (l.out != e.out || l.in != e.in || (e.has_bias && (!l.has_b || (int64_t)l.b.size() != e.out))) {
            problems.push_back("Linear #" + std::to_string(pos) + ": expected " + e.w + " (" + std::to_string(e.out) + ", " + std::to_string(e.in) + ")" + (e.has_bias ? " + bias" : "") +
                               ", found (" + std::to_string(l.out) + ", " + std::to_string(l.in) + ") at " + l.where);
            break;                                        // everything after a disagreement would be guesswork
        }
        if (ends_with(e.w, "self.Wqkv")) deinterleave_qkv(l.w, l.b);
        t[e.w] = e.w == "matchability.w" ? Placed{l.w, {256}} : Placed{l.w, {e.out, e.in}};
        if (!e.b.empty()) t[e.b] = Placed{l.b, {(int64_t)l.b.size()}};
    }
    if ((int)lns.size() != 2 * LG_LAYERS) problems.push_back(std::to_string(lns.size()) + " LayerNorm(512) parameter pairs found, expected " + std::to_string(2 * LG_LAYERS));
    else
        for (int l = 0; l < LG_LAYERS; ++l)
            for (int j = 0; j < 2; ++j) {
                const std::string p = "layers." + std::to_string(l) + (j ? ".cross" : ".self");
                t[p + ".ln_g"] = Placed{lns[2 * l + j].first->f, {512}};
                t[p + ".ln_b"] = Placed{lns[2 * l + j].second->f, {512}};
            }
    std::vector<std::string> missing;
    if (problems.empty() && !fill_lg_blob(t, blob, missing)) for (const auto& q : missing) problems.push_back(q + ": not placed");
    if (!problems.empty()) {
        err = path + ": cannot place LightGlue tensors by structure (no parameter names in the file): ";
        for (size_t q = 0; q < problems.size() && q < 6; ++q) err += (q ? "; " : "") + problems[q];
        return false;
    }
    return true;
}

bool convert_lg(const Model& m, const std::string& path, std::vector<float>& blob, std::string& err) {
    bool named = false;
    for (const auto& kv : m.inits) named |= kv.first.rfind("transformers.", 0) == 0;
    if (!named) return convert_lg_by_structure(m, path, blob, err);
    // Linear layers exported as MatMul(x, W^T) + Add(<prefix>.bias) or Gemm: prefix -> (W [out,in], b)
    std::map<std::string, const Node*> producer;
    for (const auto& n : m.nodes) for (const auto& o : n.outputs) producer[o] = &n;
    std::map<std::string, std::pair<Placed, Placed>> lin;
    for (const auto& n : m.nodes) {
        if (n.op == "Add") {
            std::vector<std::string> bias, other;
            for (const auto& i : n.inputs) { if (m.init(i) && ends_with(i, ".bias")) bias.push_back(i); else if (!m.init(i)) other.push_back(i); }
            if (bias.size() == 1 && other.size() == 1 && producer.count(other[0]) && producer[other[0]]->op == "MatMul") {
                const Node* mm = producer[other[0]];
                std::vector<const Tensor*> w;
                for (const auto& i : mm->inputs) if (m.init(i)) w.push_back(m.init(i));
                if (w.size() == 1 && w[0]->dtype == 1 && w[0]->has_dims && w[0]->dims.size() == 2) {
                    const int64_t r = w[0]->dims[0], c = w[0]->dims[1];
                    Placed W{std::vector<float>((size_t)(r * c)), {c, r}};
                    for (int64_t a = 0; a < r; ++a) for (int64_t b = 0; b < c; ++b) W.v[(size_t)(b * r + a)] = w[0]->f[(size_t)(a * c + b)];
                    const Tensor* bt = m.init(bias[0]);
                    lin[bias[0].substr(0, bias[0].size() - 5)] = {W, Placed{bt->f, {(int64_t)bt->f.size()}}};
                }
            }
        } else if (n.op == "Gemm" && n.inputs.size() >= 3 && m.init(n.inputs[2]) && ends_with(n.inputs[2], ".bias")) {
            const Tensor* w = m.init(n.inputs[1]);
            const Tensor* b = m.init(n.inputs[2]);
            if (w && w->dtype == 1 && w->has_dims && w->dims.size() == 2) {
                Placed W;
                if (w->dims[0] == (int64_t)b->f.size()) W = Placed{w->f, {w->dims[0], w->dims[1]}};      // PyTorch's [out,in] with transB = 1
                else {
                    const int64_t r = w->dims[0], c = w->dims[1];
                    W = Placed{std::vector<float>((size_t)(r * c)), {c, r}};
                    for (int64_t a = 0; a < r; ++a) for (int64_t q = 0; q < c; ++q) W.v[(size_t)(q * r + a)] = w->f[(size_t)(a * c + q)];
                }
                lin[n.inputs[2].substr(0, n.inputs[2].size() - 5)] = {W, Placed{b->f, {(int64_t)b->f.size()}}};
            }
        }
    }
    std::vector<std::string> missing;
    auto linear = [&](const std::string& prefix, Placed& W, Placed& B) {
        const Tensor* w = m.init(prefix + ".weight");
        const Tensor* b = m.init(prefix + ".bias");
        if (w && b && w->dtype == 1 && b->dtype == 1) { W = Placed{w->f, w->has_dims ? w->dims : std::vector<int64_t>{(int64_t)w->f.size()}}; B = Placed{b->f, {(int64_t)b->f.size()}}; return true; }
        auto it = lin.find(prefix);
        if (it != lin.end()) { W = it->second.first; B = it->second.second; return true; }
        missing.push_back("'" + prefix + "'");
        return false;
    };
    std::map<std::string, Placed> t;
    {
        const Tensor* wr = m.init("posenc.Wr.weight");
        if (wr && !(wr->dtype == 1 && (wr->shape_is({32, 2}) || wr->shape_is({2, 32})))) wr = nullptr;    // a tensor of that name with another type / shape is not it
        if (!wr)                                          // bias-free Linear: anonymous MatMul constant [2,32]
            for (const auto& nm : m.order) { const Tensor& c = m.inits.at(nm); if (c.dtype == 1 && (c.shape_is({2, 32}) || c.shape_is({32, 2}))) { wr = &c; break; } }
        if (!wr) missing.push_back("posenc.Wr");
        else if (wr->shape_is({32, 2})) t["posenc.Wr"] = Placed{wr->f, {32, 2}};
        else { Placed W{std::vector<float>(64), {32, 2}}; for (int a = 0; a < 2; ++a) for (int b = 0; b < 32; ++b) W.v[b * 2 + a] = wr->f[a * 32 + b]; t["posenc.Wr"] = W; }
    }
    for (int l = 0; l < LG_LAYERS; ++l) {
        const std::string p = "layers." + std::to_string(l) + ".", s = "transformers." + std::to_string(l) + ".self_attn.", c = "transformers." + std::to_string(l) + ".cross_attn.";
        Placed W, B;
        if (!linear(s + "Wqkv", W, B)) continue;
        if (W.v.size() == 768 * 256 && B.v.size() == 768) deinterleave_qkv(W.v, B.v);
        t[p + "self.Wqkv"] = W; t[p + "self.bqkv"] = B;
        const char* map_[][3] = {{"out_proj", "self.Wo", "self.bo"}, {"ffn.0", "self.W1", "self.b1"}, {"ffn.3", "self.W2", "self.b2"}};
        bool ok = true;
        for (auto& e : map_) { if (!linear(s + e[0], W, B)) { ok = false; break; } t[p + e[1]] = W; t[p + e[2]] = B; }
        if (!ok) continue;
        const char* mapc[][3] = {{"to_qk", "cross.Wqk", "cross.bqk"}, {"to_v", "cross.Wv", "cross.bv"}, {"to_out", "cross.Wo", "cross.bo"}, {"ffn.0", "cross.W1", "cross.b1"}, {"ffn.3", "cross.W2", "cross.b2"}};
        for (auto& e : mapc) { if (!linear(c + e[0], W, B)) { ok = false; break; } t[p + e[1]] = W; t[p + e[2]] = B; }
        if (!ok) continue;
        for (int j = 0; j < 2; ++j) {                      // LayerNorm(512)
            const std::string src = (j ? c : s) + "ffn.1";
            const Tensor* g = m.init(src + ".weight");
            const Tensor* b = m.init(src + ".bias");
            if (!g || !b || g->dtype != 1 || b->dtype != 1) { missing.push_back("'" + src + ".weight'"); break; }
            t[p + (j ? "cross" : "self") + ".ln_g"] = Placed{g->f, {(int64_t)g->f.size()}};
            t[p + (j ? "cross" : "self") + ".ln_b"] = Placed{b->f, {(int64_t)b->f.size()}};
        }
    }
    {   // no early exit in the fused export: only the last assignment head is live
        Placed W, B;
        const std::string a = "log_assignment." + std::to_string(LG_LAYERS - 1) + ".";
        if (linear(a + "final_proj", W, B)) {
            t["final_proj.W"] = W; t["final_proj.b"] = B;
            if (linear(a + "matchability", W, B)) { t["matchability.w"] = Placed{W.v, {(int64_t)W.v.size()}}; t["matchability.b"] = Placed{B.v, {(int64_t)B.v.size()}}; }
        }
    }
    std::vector<std::string> unplaced;
    fill_lg_blob(t, blob, unplaced);
    missing.insert(missing.end(), unplaced.begin(), unplaced.end());
    if (!missing.empty()) {
        err = path + ": cannot place LightGlue tensors [";
        for (size_t q = 0; q < missing.size() && q < 12; ++q) err += (q ? ", " : "") + missing[q];
        err += "]...";
        return false;
    }
    return true;
}

}  // namespace

// kind 1 / 2 -> canonical blob + the graph's hyper-parameters written into *hp (only this kind's fields).  Nothing is guessed: any problem of
// the readers is an error (the C entry takes no `assume`; a caller who must state a value converts with the Python tool's --assume-* and loads
// the RFEW file, or calls rfe_set_hparams afterwards).
static bool onnx_convert_impl(const std::string& path, int kind, std::vector<float>& blob, rfe_hparams* hp, std::string& err) {
    Model m;
    if (!read_model(path, m, err)) return false;
    std::vector<std::string> errors;
    if (kind == RFE_KIND_SUPERPOINT) {
        SpRead r;
        std::vector<std::string> problems = read_sp_hparams(m, r);
        if (!r.has_always && r.has_k) { r.has_always = true; r.always = 0; }
        errors = problems;
        if (errors.empty() && !(r.has_r && r.has_b && r.has_k && r.has_thr)) errors.push_back("hyper-parameters incomplete");
        if (errors.empty() && !(1 <= r.r && r.r <= NMS_MAX_RADIUS && 0 <= r.b && r.b <= 64 && 1 <= r.k && r.k <= 4096))
            errors.push_back("SuperPoint hyper-parameters outside the library's range (radius 1..8, border 0..64, keypoints 1..4096)");
        if (errors.empty()) {
            hp->sp_max_keypoints = r.k; hp->sp_detection_threshold = r.thr; hp->sp_nms_radius = r.r; hp->sp_remove_borders = r.b; hp->sp_topk_always = r.always;
        }
    } else if (kind == RFE_KIND_LIGHTGLUE) {
        LgRead r;
        errors = read_lg_hparams(m, r);
        if (errors.empty() && !(r.has_layers && r.has_heads && r.has_thr)) errors.push_back("hyper-parameters incomplete");
        if (errors.empty() && (r.layers != LG_LAYERS || r.heads != 4))
            errors.push_back("LightGlue with " + std::to_string(r.layers) + " layers of " + std::to_string(r.heads) + " heads: the kernels are built for " + std::to_string(LG_LAYERS) + " layers of 4 heads x 64");
        if (errors.empty()) { hp->lg_layers = r.layers; hp->lg_heads = r.heads; hp->lg_filter_threshold = r.thr; }
    } else { err = "onnx_convert: unknown model kind"; return false; }
    if (!errors.empty()) {
        err = path + ": graph hyper-parameters refused -- ";
        for (size_t q = 0; q < errors.size(); ++q) err += (q ? "; " : "") + errors[q];
        return false;
    }
    return kind == RFE_KIND_SUPERPOINT ? convert_sp(m, path, blob, err) : convert_lg(m, path, blob, err);
}

// the weights alone (tools that only need them; the Python convert_superpoint / convert_lightglue)
static bool onnx_convert_weights_only_impl(const std::string& path, int kind, std::vector<float>& blob, std::string& err) {
    Model m;
    if (!read_model(path, m, err)) return false;
    return kind == RFE_KIND_SUPERPOINT ? convert_sp(m, path, blob, err) : kind == RFE_KIND_LIGHTGLUE ? convert_lg(m, path, blob, err) : false;
}

// Nothing thrown by the reader (std::bad_alloc / std::length_error of a container sized by a hostile file, std::out_of_range) crosses into the
// extern "C" entries that call these two: an unreadable graph is `false` + a reason, which the callers turn into RFE_ERR_IO.
bool onnx_convert(const std::string& path, int kind, std::vector<float>& blob, rfe_hparams* hp, std::string& err) {
    try { return onnx_convert_impl(path, kind, blob, hp, err); }
    catch (const std::exception& e) { err = path + ": cannot be read as an ONNX graph (" + e.what() + ")"; return false; }
    catch (...) { err = path + ": cannot be read as an ONNX graph"; return false; }
}
bool onnx_convert_weights_only(const std::string& path, int kind, std::vector<float>& blob, std::string& err) {
    try { return onnx_convert_weights_only_impl(path, kind, blob, err); }
    catch (const std::exception& e) { err = path + ": cannot be read as an ONNX graph (" + e.what() + ")"; return false; }
    catch (...) { err = path + ": cannot be read as an ONNX graph"; return false; }
}

}  // namespace rfe

// Test hook without a ctx or a GPU: convert `path` (kind 1 / 2) into blob [rfe_weight_count(kind)] and the kind's fields of *hp.
// weights_only != 0 skips the hyper-parameter readers.  Returns RFE_OK or RFE_ERR_IO with the reason in err (NUL-terminated, truncated).
extern "C" int rfe_k_onnx_convert(const char* path, int kind, int weights_only, float* blob, rfe_hparams* hp, char* err, int errlen) {
    if (!path || !blob || (kind != RFE_KIND_SUPERPOINT && kind != RFE_KIND_LIGHTGLUE)) return RFE_ERR_INVALID;
    std::vector<float> b;
    std::string e;
    rfe_hparams h = rfe_default_hparams();
    if (hp) h = *hp;
    const bool ok = weights_only ? rfe::onnx_convert_weights_only(path, kind, b, e) : rfe::onnx_convert(path, kind, b, &h, e);
    if (err && errlen > 0) { snprintf(err, (size_t)errlen, "%s", e.c_str()); }
    if (!ok) return RFE_ERR_IO;
    if ((int64_t)b.size() != rfe_weight_count(kind)) return RFE_ERR_IO;
    memcpy(blob, b.data(), b.size() * sizeof(float));
    if (hp) *hp = h;
    return RFE_OK;
}
