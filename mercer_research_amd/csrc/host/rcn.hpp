// rcn.hpp -- C++17 host-side mirror of the reference's Rust API (`rcn::rcn::RCN`, rcn/src/rcn.rs) over the C ABI of
// include/rcn_hip.h.  The reference is compiled (Rust) code and no Rust toolchain exists in this environment, so the
// host side above the boundary is C++; the Rust binding a maintainer would use is rust/rcn-hip-sys.
// Header-only; link with -lrcn_hip.  Names, argument meaning and failure behaviour follow the reference: conditions on
// which the reference panics surface as rcn::Panic, everything else as rcn::Error.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <filesystem>

#include "../../../include/rcn_hip.h"
#include "formats.hpp"

namespace rcn {
namespace host {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};
struct Panic : Error { using Error::Error; };   // the Rust reference panics here (shape / "Not implemented")

enum class Padding : int32_t { None = 0, Same = 1 };        // utils/kernel.rs:25-28
enum class Pooling : int32_t { Average = 0, Max = 1 };      // utils/kernel.rs:32-35

struct RCNLayer {                                           // rcn.rs:35-38
    int32_t kind, arg;
    static RCNLayer Convolve2D(Padding p) { return {RCN_HIP_LAYER_CONVOLVE2D, (int32_t)p}; }
    static RCNLayer Pool2D(Pooling p) { return {RCN_HIP_LAYER_POOL2D, (int32_t)p}; }
};

// InputSet (rcn.rs:49): features and one-hot expectation, stored sample-major for a whole set
struct DataSet {
    size_t n = 0, features = 0, classes = 0;
    std::vector<double> x, y;
};

class RCN {
public:
    // RCN::new (rcn.rs:58-75); in_h/in_w, dtype and device are what a device context needs on top of it
    RCN(size_t classes, std::vector<RCNLayer> convpool_cfg, std::vector<size_t> feedforward_cfg, std::string training_path,
        std::string testing_path, int in_h = 28, int in_w = 28, int dtype = RCN_HIP_F32, int device = 0)
        : classes_(classes), convpool_cfg_(std::move(convpool_cfg)), feedforward_cfg_(std::move(feedforward_cfg)),
          training_path_(std::move(training_path)), testing_path_(std::move(testing_path)), in_h_(in_h), in_w_(in_w) {
        std::vector<rcn_hip_layer> layers;
        for (const auto& l : convpool_cfg_) layers.push_back({l.kind, l.arg});
        std::vector<int32_t> hidden(feedforward_cfg_.begin(), feedforward_cfg_.end());
        rcn_hip_cfg cfg{};
        cfg.struct_size = sizeof(cfg); cfg.device = device; cfg.dtype = dtype; cfg.in_h = in_h; cfg.in_w = in_w;
        cfg.n_convpool = (int32_t)layers.size(); cfg.convpool = layers.data();
        cfg.n_hidden = (int32_t)hidden.size(); cfg.hidden = hidden.data(); cfg.classes = (int32_t)classes; cfg.stream = nullptr;
        const int st = rcn_hip_create(&cfg, &ctx_);
        if (st != RCN_HIP_OK) {
            const std::string msg = ctx_ ? rcn_hip_last_error(ctx_) : rcn_hip_status_string(st);
            if (ctx_) rcn_hip_destroy(ctx_);
            ctx_ = nullptr;
            raise(st, msg);
        }
        int64_t f = 0;
        rcn_hip_feature_len(ctx_, &f);
        feature_len_ = (size_t)f;
    }
    ~RCN() { if (ctx_) rcn_hip_destroy(ctx_); }
    RCN(const RCN&) = delete;
    RCN& operator=(const RCN&) = delete;

    size_t feature_len() const { return feature_len_; }
    size_t classes() const { return classes_; }
    rcn_hip_ctx* ctx() { return ctx_; }
    bool weights_loaded() const { return weights_loaded_; }
    std::pair<double, double> scale_set() const { double m, s; rcn_hip_get_scale(ctx_, &m, &s); return {m, s}; }

    // load_weights_and_bias (rcn.rs:425-457)
    void load_weights_and_bias(uint64_t seed = 0) { check(rcn_hip_init_params(ctx_, seed)); weights_loaded_ = true; }
    // Weights.0 / Bias.0 of layer l, column-major like DMatrix (rcn.rs:28,31)
    void set_params(int layer, const std::vector<double>& w_colmajor, const std::vector<double>& b) {
        check(rcn_hip_set_params(ctx_, layer, w_colmajor.data(), b.data()));
        weights_loaded_ = true;
    }
    void get_params(int layer, std::vector<double>& w_colmajor, std::vector<double>& b) {
        int32_t r, c;
        check(rcn_hip_layer_dims(ctx_, layer, &r, &c));
        w_colmajor.resize((size_t)r * c); b.resize(r);
        check(rcn_hip_get_params(ctx_, layer, w_colmajor.data(), b.data()));
    }

    // flatten_feature_set (rcn.rs:317-356) for n decoded grayscale images (row-major u8 pixels)
    std::vector<double> flatten_feature_set(const uint8_t* pixels, size_t n) {
        std::vector<double> out(n * feature_len_);
        check(rcn_hip_features(ctx_, pixels, n, out.data()));
        return out;
    }

    // the arithmetic of load_data after the PNG decode (rcn.rs:399-414): features, gen_scales (overwrites scale_set),
    // standardise + clamp, one-hot expectations (rcn.rs:466-471)
    DataSet load_data(const uint8_t* pixels, const int32_t* labels, size_t n) {
        DataSet d;
        d.n = n; d.features = feature_len_; d.classes = classes_;
        d.x = flatten_feature_set(pixels, n);
        double mean, sd;
        check(rcn_hip_gen_scales(ctx_, d.x.data(), n, &mean, &sd));
        check(rcn_hip_standardize(ctx_, d.x.data(), d.x.size()));
        d.y.assign(n * classes_, 0.0);
        for (size_t i = 0; i < n; ++i) d.y[i * classes_ + (size_t)labels[i]] = 1.0;
        return d;
    }

    // train_batch (rcn.rs:176-223)
    double train_batch(const double* x, const double* y, size_t batch, double eta, bool want_loss = false) {
        double loss = 0.0;
        check(rcn_hip_train_batch(ctx_, x, y, batch, eta, want_loss ? &loss : nullptr));
        return loss;
    }
    // classify_test (rcn.rs:105-116)
    std::vector<double> classify_test(const double* x, size_t n) {
        std::vector<double> out(n * classes_);
        check(rcn_hip_forward(ctx_, x, n, out.data()));
        return out;
    }
    // RCN::classify minus the PNG decode (rcn.rs:82-98)
    size_t classify(const uint8_t* pixels) {
        int32_t cls = 0;
        check(rcn_hip_classify_images(ctx_, pixels, 1, &cls));
        return (size_t)cls;
    }

    // RCN::train (rcn.rs:126-167) on decoded images: both sets are loaded (scale_set ends up holding the TEST set's
    // statistics, rcn.rs:134-137), weights are drawn if empty, then per epoch: shuffle, chunks_exact, train_batch, and
    // the accuracy line of rcn.rs:158-164.  Returns the accepted count per epoch.
    std::vector<int64_t> train(const uint8_t* train_px, const int32_t* train_lab, size_t n_train, const uint8_t* test_px,
                               const int32_t* test_lab, size_t n_test, size_t batch_size, size_t epochs, double eta,
                               uint64_t shuffle_seed = 0, bool print = true) {
        // both sets stay resident in HBM (rcn_hip_load_data); an epoch is ONE call (device shuffle or the order drawn here,
        // then chunks_exact over it), the evaluation another -- no per-step traffic over PCIe
        double mean, sd;
        check(rcn_hip_load_data(ctx_, 0, train_px, train_lab, n_train, &mean, &sd));      // rcn.rs:134-135
        check(rcn_hip_load_data(ctx_, 1, test_px, test_lab, n_test, &mean, &sd));         // rcn.rs:136-137: scale_set = the TEST statistics
        // rcn.rs:139-141; a seeded run (tests, --seed) draws its parameters from the seed too -- the reference has no seeds at all
        if (!weights_loaded_) load_weights_and_bias(shuffle_seed ? (shuffle_seed ^ 0x9E3779B97F4A7C15ULL) | 1ULL : 0);
        std::mt19937_64 rng(shuffle_seed ? shuffle_seed : std::random_device{}());
        std::vector<int32_t> order(n_train);
        std::iota(order.begin(), order.end(), 0);
        std::vector<int64_t> accepted;
        for (size_t e = 0; e < epochs; ++e) {                                // rcn.rs:144
            std::shuffle(order.begin(), order.end(), rng);                    // rcn.rs:146
            check(rcn_hip_train_set_epoch(ctx_, 0, order.data(), 0, batch_size, eta, nullptr));   // chunks_exact + train_batch, rcn.rs:147-149
            int64_t acc = 0;
            check(rcn_hip_evaluate_set(ctx_, 1, &acc));                       // rcn.rs:152-157
            accepted.push_back(acc);
            if (print) std::printf("Epoch %zu: %lld/%zu [%.2f%%]\n", e, (long long)acc, n_test, (double)acc / (double)n_test * 100.0);
        }
        return accepted;
    }

    // ---- the reference's on-disk model (rcn/src/main.rs:47-50,77): bincode rcn.bin
    void save(const std::string& path) {
        Checkpoint c;
        c.classes = classes_;
        for (const auto& l : convpool_cfg_) c.convpool_cfg.push_back({(uint32_t)l.kind, (uint32_t)l.arg});
        c.feedforward_cfg.assign(feedforward_cfg_.begin(), feedforward_cfg_.end());
        if (weights_loaded_)
            for (int l = 0; l < rcn_hip_num_layers(ctx_); ++l) {
                Checkpoint::Matrix m; std::vector<double> b; int32_t r, cc;
                check(rcn_hip_layer_dims(ctx_, l, &r, &cc));
                m.rows = r; m.cols = cc;
                get_params(l, m.data, b);
                c.layer_weights.push_back(std::move(m)); c.layer_bias.push_back(std::move(b));
            }
        auto sc = scale_set(); c.scale_mean = sc.first; c.scale_sd = sc.second;
        c.training_path = training_path_; c.testing_path = testing_path_;
        write_file(path, checkpoint_dumps(c));
    }
    static std::unique_ptr<RCN> load(const std::string& path, int in_h = 28, int in_w = 28, int dtype = RCN_HIP_F32, int device = 0) {
        const Checkpoint c = checkpoint_loads(read_file(path));
        std::vector<RCNLayer> cfg;
        for (auto& l : c.convpool_cfg) cfg.push_back({(int32_t)l.first, (int32_t)l.second});
        auto m = std::make_unique<RCN>((size_t)c.classes, cfg, std::vector<size_t>(c.feedforward_cfg.begin(), c.feedforward_cfg.end()),
                                       c.training_path, c.testing_path, in_h, in_w, dtype, device);
        for (size_t l = 0; l < c.layer_weights.size(); ++l) m->set_params((int)l, c.layer_weights[l].data, c.layer_bias[l]);   // non-empty => skip init, rcn.rs:139
        m->check(rcn_hip_set_scale(m->ctx_, c.scale_mean, c.scale_sd));
        return m;
    }
    void set_paths(std::string training, std::string testing) { training_path_ = std::move(training); testing_path_ = std::move(testing); }

    // ---- load_data's file handling (rcn.rs:367-404): class directories in sorted order, `csl` files drawn per class
    // without replacement; panics if a class holds fewer.  -> pixels [n][h][w] u8, class indices
    struct ImageSet { std::vector<uint8_t> px; std::vector<int32_t> labels; size_t n = 0, n_classes = 0; };
    ImageSet read_image_set(const std::string& path, size_t class_size_limit, std::mt19937_64& rng) const {
        namespace fs = std::filesystem;
        std::vector<fs::path> classes;
        for (auto& e : fs::directory_iterator(path)) classes.push_back(e.path());
        std::sort(classes.begin(), classes.end());                                   // rcn.rs:374
        ImageSet s; s.n_classes = classes.size();
        for (size_t i = 0; i < classes.size(); ++i) {
            std::vector<fs::path> files;
            for (auto& e : fs::directory_iterator(classes[i])) files.push_back(e.path());
            if (class_size_limit > files.size())                                          // rcn.rs:383-390
                throw Panic(RCN_HIP_ERR_SHAPE, "provided class_size_limit for " + path + " too large! expected " + std::to_string(class_size_limit) +
                                                   " <= " + std::to_string(files.size()));
            for (size_t k = 0; k < class_size_limit; ++k) {                              // rcn.rs:392-394
                const size_t idx = std::uniform_int_distribution<size_t>(0, files.size() - 1)(rng);
                const GrayImage g = png_to_pixel_matrix(read_file(files[idx].string()));
                files.erase(files.begin() + (long)idx);
                if (g.h != in_h_ || g.w != in_w_) throw Error(RCN_HIP_ERR_INVALID_ARG, "image shape differs from the context's input shape");
                s.px.insert(s.px.end(), g.px.begin(), g.px.end());
                s.labels.push_back((int32_t)i);
            }
        }
        s.n = s.labels.size();
        return s;
    }
    // RCN::train (rcn.rs:126-133) from the two directories
    std::vector<int64_t> train(size_t batch_size, size_t epochs, double eta, size_t training_class_size_limit, size_t testing_class_size_limit,
                               uint64_t seed = 0, bool print = true) {
        std::mt19937_64 rng(seed ? seed : std::random_device{}());
        const ImageSet tr = read_image_set(training_path_, training_class_size_limit, rng);
        const ImageSet te = read_image_set(testing_path_, testing_class_size_limit, rng);
        if (tr.n_classes != classes_ || te.n_classes != classes_)
            throw Panic(RCN_HIP_ERR_SHAPE, "number of class directories differs from the network's outputs (gemv / vector comparison panics)");
        return train(tr.px.data(), tr.labels.data(), tr.n, te.px.data(), te.labels.data(), te.n, batch_size, epochs, eta, rng(), print);
    }
    // RCN::classify(img_path) (rcn.rs:82-98)
    size_t classify(const std::string& img_path) {
        const GrayImage g = png_to_pixel_matrix(read_file(img_path));
        if (g.h != in_h_ || g.w != in_w_) throw Error(RCN_HIP_ERR_INVALID_ARG, "image shape differs from the context's input shape");
        return classify(g.px.data());
    }

private:
    [[noreturn]] static void raise(int st, const std::string& msg) {
        if (st == RCN_HIP_ERR_SHAPE || st == RCN_HIP_ERR_UNSUPPORTED) throw Panic(st, msg);
        throw Error(st, msg);
    }
    void check(int st) const {
        if (st != RCN_HIP_OK) raise(st, rcn_hip_last_error(ctx_));
    }
    size_t classes_;
    std::vector<RCNLayer> convpool_cfg_;
    std::vector<size_t> feedforward_cfg_;
    std::string training_path_, testing_path_;
    int in_h_, in_w_;
    rcn_hip_ctx* ctx_ = nullptr;
    size_t feature_len_ = 0;
    bool weights_loaded_ = false;
};

}  // namespace host
}  // namespace rcn
