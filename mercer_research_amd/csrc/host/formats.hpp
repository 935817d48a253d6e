// formats.hpp -- the data formats either side of the hot path, for the C++ host mirror:
//   * the reference's `rcn.bin`: bincode 1.x default options of `struct RCN` (rcn.rs:13-25) with the hand-written serde
//     impls of Weights / Bias (utils/serialization.rs:11-151);
//   * a minimal PNG decoder (zlib inflate + the five scan-line filters) with the `image` crate's grayscale semantics
//     that `RCN::classify` / `load_data` rely on (rcn.rs:83, 394-398; lib.rs:27-41).
// Same behaviour as mercer_research_amd/checkpoint.py and png.py (tests compare the two byte for byte).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rcn {
namespace host {

struct FormatError : std::runtime_error { using std::runtime_error::runtime_error; };
// errors.rs:1-13: "InvalidGrayscaleImageError: Image provided was not Luma8 (grayscaled image)"
struct InvalidGrayscaleImageError : FormatError { using FormatError::FormatError; };

// ---------------------------------------------------------------------------------------------------- rcn.bin
struct Checkpoint {
    uint64_t classes = 0;
    std::vector<std::pair<uint32_t, uint32_t>> convpool_cfg;   // (variant, inner variant)
    std::vector<uint64_t> feedforward_cfg;
    struct Matrix { uint64_t rows = 0, cols = 0; std::vector<double> data; };   // column-major, as DMatrix iterates
    std::vector<Matrix> layer_weights;
    std::vector<std::vector<double>> layer_bias;
    double scale_mean = 1.0, scale_sd = 1.0;                   // RCN::new: (1, 1), rcn.rs:71
    std::string training_path, testing_path;
};

namespace detail {
inline void put_u64(std::string& o, uint64_t v) { char b[8]; for (int i = 0; i < 8; ++i) b[i] = (char)(v >> (8 * i)); o.append(b, 8); }
inline void put_u32(std::string& o, uint32_t v) { char b[4]; for (int i = 0; i < 4; ++i) b[i] = (char)(v >> (8 * i)); o.append(b, 4); }
inline void put_f64(std::string& o, double d) { uint64_t v; std::memcpy(&v, &d, 8); put_u64(o, v); }
struct Reader {
    const std::string& d; size_t o = 0;
    explicit Reader(const std::string& s) : d(s) {}
    const unsigned char* take(size_t n) {
        if (n > d.size() - o) throw FormatError("unexpected end of rcn.bin (bincode: UnexpectedEof)");
        const unsigned char* p = reinterpret_cast<const unsigned char*>(d.data()) + o; o += n; return p;
    }
    uint64_t u64() { const unsigned char* p = take(8); uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }
    uint32_t u32() { const unsigned char* p = take(4); uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)p[i] << (8 * i); return v; }
    double f64() { uint64_t v = u64(); double x; std::memcpy(&x, &v, 8); return x; }
    std::vector<double> f64s(uint64_t n) {
        if (n > (d.size() - o) / 8) throw FormatError("sequence length exceeds the file");
        std::vector<double> v(n); for (auto& x : v) x = f64(); return v;
    }
};
}  // namespace detail

inline std::string checkpoint_dumps(const Checkpoint& c) {
    using namespace detail;
    std::string o;
    put_u64(o, c.classes);
    put_u64(o, c.convpool_cfg.size());
    for (auto& l : c.convpool_cfg) { put_u32(o, l.first); put_u32(o, l.second); }
    put_u64(o, c.feedforward_cfg.size());
    for (auto h : c.feedforward_cfg) put_u64(o, h);
    put_u64(o, c.layer_weights.size());
    for (auto& w : c.layer_weights) {
        put_u64(o, w.rows); put_u64(o, w.cols); put_u64(o, w.data.size());
        for (double v : w.data) put_f64(o, v);
    }
    put_u64(o, c.layer_bias.size());
    for (auto& b : c.layer_bias) { put_u64(o, b.size()); for (double v : b) put_f64(o, v); }
    put_f64(o, c.scale_mean); put_f64(o, c.scale_sd);
    for (const std::string* s : {&c.training_path, &c.testing_path}) { put_u64(o, s->size()); o += *s; }
    return o;
}

inline Checkpoint checkpoint_loads(const std::string& bytes) {
    detail::Reader r(bytes);
    Checkpoint c;
    c.classes = r.u64();
    for (uint64_t n = r.u64(), i = 0; i < n; ++i) {
        const uint32_t k = r.u32(), a = r.u32();
        if (k > 1 || a > 1) throw FormatError("invalid enum variant index in convpool_cfg");
        c.convpool_cfg.push_back({k, a});
    }
    for (uint64_t n = r.u64(), i = 0; i < n; ++i) c.feedforward_cfg.push_back(r.u64());
    for (uint64_t n = r.u64(), i = 0; i < n; ++i) {
        Checkpoint::Matrix m;
        m.rows = r.u64(); m.cols = r.u64();
        m.data = r.f64s(r.u64());
        if (m.rows * m.cols != m.data.size()) throw FormatError("Weights: dims do not match data length (DMatrix::from_vec panics)");
        c.layer_weights.push_back(std::move(m));
    }
    for (uint64_t n = r.u64(), i = 0; i < n; ++i) c.layer_bias.push_back(r.f64s(r.u64()));
    c.scale_mean = r.f64(); c.scale_sd = r.f64();
    for (std::string* s : {&c.training_path, &c.testing_path}) { const uint64_t n = r.u64(); const unsigned char* p = r.take(n); s->assign(reinterpret_cast<const char*>(p), n); }
    return c;
}

inline std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw FormatError("cannot open " + path);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
inline void write_file(const std::string& path, const std::string& bytes) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw FormatError("cannot write " + path);
    f.write(bytes.data(), (std::streamsize)bytes.size());
}

// ---------------------------------------------------------------------------------------------------- PNG -> pixel matrix
struct GrayImage { int h = 0, w = 0; std::vector<uint8_t> px; };   // row-major [y][x]: what get_pixel_matrix reads

inline GrayImage png_to_pixel_matrix(const std::string& data) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (data.size() < 8 || std::memcmp(data.data(), sig, 8) != 0) throw FormatError("not a PNG file");
    auto be32 = [&](size_t o) { const unsigned char* p = (const unsigned char*)data.data() + o; return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; };
    uint32_t W = 0, H = 0; int depth = 0, ctype = -1, interlace = 0;
    std::string idat; std::vector<uint8_t> plte;
    for (size_t pos = 8; pos + 8 <= data.size();) {
        const uint32_t len = be32(pos);
        const std::string typ = data.substr(pos + 4, 4);
        if (pos + 12 + (size_t)len > data.size()) throw FormatError("truncated chunk");
        const char* body = data.data() + pos + 8;
        if (typ == "IHDR") { W = be32(pos + 8); H = be32(pos + 12); depth = (unsigned char)body[8]; ctype = (unsigned char)body[9]; interlace = (unsigned char)body[12]; }
        else if (typ == "PLTE") plte.assign(body, body + len);
        else if (typ == "IDAT") idat.append(body, len);
        else if (typ == "IEND") break;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || idat.empty()) throw FormatError("missing IHDR / IDAT");
    if (interlace) throw FormatError("interlaced PNG not supported");
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: throw FormatError("bad colour type"); }
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) throw FormatError("bad bit depth");
    const size_t bpp = std::max<size_t>(1, (size_t)ch * depth / 8), stride = ((size_t)W * ch * depth + 7) / 8;
    std::vector<uint8_t> raw(H * (stride + 1));
    uLongf out_len = raw.size();
    if (uncompress(raw.data(), &out_len, (const Bytef*)idat.data(), idat.size()) != Z_OK || out_len < raw.size()) throw FormatError("IDAT inflate failed / too short");
    std::vector<uint8_t> rows(H * stride), zero(stride, 0);
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t* in = &raw[y * (stride + 1) + 1];
        uint8_t* cur = &rows[y * stride];
        const uint8_t* prev = y ? &rows[(y - 1) * stride] : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) throw FormatError("bad filter type");
            cur[i] = (uint8_t)(in[i] + pred);
        }
    }
    if (depth == 16) throw InvalidGrayscaleImageError("InvalidGrayscaleImageError: Image provided was not Luma8 (grayscaled image)");
    GrayImage g; g.h = (int)H; g.w = (int)W; g.px.resize((size_t)H * W);
    auto sample = [&](uint32_t y, size_t idx) -> int {       // idx-th sample of row y, expanded from `depth` bits
        if (depth == 8) return rows[y * stride + idx];
        const size_t bit = idx * depth; const int shift = 8 - depth - (int)(bit % 8);
        return (rows[y * stride + bit / 8] >> shift) & ((1 << depth) - 1);
    };
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            int v;
            if (ctype == 0) { v = sample(y, x); if (depth < 8) v = v * 255 / ((1 << depth) - 1); }
            else if (ctype == 4) v = sample(y, (size_t)x * 2);
            else {
                int r, gg, b;
                if (ctype == 3) { const size_t k = (size_t)sample(y, x) * 3; if (k + 2 >= plte.size()) throw FormatError("palette index out of range"); r = plte[k]; gg = plte[k + 1]; b = plte[k + 2]; }
                else { r = sample(y, (size_t)x * ch); gg = sample(y, (size_t)x * ch + 1); b = sample(y, (size_t)x * ch + 2); }
                v = (2126 * r + 7152 * gg + 722 * b) / 10000;      // image 0.24 rgb_to_luma (.grayscale(), rcn.rs:83,398)
            }
            g.px[(size_t)y * W + x] = (uint8_t)v;
        }
    return g;
}

}  // namespace host
}  // namespace rcn
