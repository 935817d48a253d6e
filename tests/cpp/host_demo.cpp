// Exercises the C++ host mirror (mercer_research_amd/csrc/host/rcn.hpp) end to end on a GPU: RCN::new with the
// architecture of rcn/src/main.rs:53-59, train on a small synthetic set, classify.  Prints the reference's epoch lines.
// Build: g++ -std=c++17 -O2 tests/cpp/host_demo.cpp -Lmercer_research_amd -lrcn_hip -Wl,-rpath,$PWD/mercer_research_amd
#include <cstdio>
#include <random>
#include <vector>

#include "../../mercer_research_amd/csrc/host/rcn.hpp"

using namespace rcn::host;

int main() {
    const int H = 28, W = 28, classes = 10;
    const size_t n_train = 2000, n_test = 500;
    std::mt19937 rng(7);
    // ten fixed random prototypes + noise: learnable, so accuracy must rise well above chance
    std::vector<std::vector<uint8_t>> proto(classes, std::vector<uint8_t>(H * W, 0));
    for (auto& p : proto)
        for (int y = 4; y < H - 4; ++y)
            for (int x = 4; x < W - 4; ++x) p[y * W + x] = (rng() % 100 < 25) ? (uint8_t)(64 + rng() % 192) : 0;
    auto make = [&](size_t n, std::vector<uint8_t>& px, std::vector<int32_t>& lab) {
        px.resize(n * H * W); lab.resize(n);
        for (size_t i = 0; i < n; ++i) {
            lab[i] = (int32_t)(rng() % classes);
            for (int k = 0; k < H * W; ++k) {
                int v = proto[lab[i]][k];
                if (v && rng() % 100 < 15) v = 0;
                if (!v && rng() % 100 < 3) v = 100;
                px[i * H * W + k] = (uint8_t)v;
            }
        }
    };
    std::vector<uint8_t> trp, tep;
    std::vector<int32_t> trl, tel;
    make(n_train, trp, trl);
    make(n_test, tep, tel);
    try {
        RCN model(classes, {RCNLayer::Convolve2D(Padding::Same), RCNLayer::Pool2D(Pooling::Max), RCNLayer::Convolve2D(Padding::Same), RCNLayer::Pool2D(Pooling::Max)},
                  {30}, "images/mnist_png/training", "images/mnist_png/testing", H, W, RCN_HIP_F32, 0);
        model.load_weights_and_bias(42);
        auto acc = model.train(trp.data(), trl.data(), n_train, tep.data(), tel.data(), n_test, 10, 3, 3.0, 11, true);
        const size_t cls = model.classify(tep.data());
        std::printf("classify(first test image) = %zu (label %d)\n", cls, tel[0]);
        // the reference's Average pooling panics ("Not implemented", kernel.rs:283): same here
        bool panicked = false;
        try { RCN bad(classes, {RCNLayer::Convolve2D(Padding::Same), RCNLayer::Pool2D(Pooling::Average)}, {30}, "", "", H, W); }
        catch (const Panic&) { panicked = true; }
        if (!panicked) { std::printf("FAIL: Average pooling did not panic\n"); return 1; }
        if (acc.back() < (int64_t)(n_test * 0.5)) { std::printf("FAIL: accuracy %lld/%zu too low\n", (long long)acc.back(), n_test); return 1; }
        std::printf("host_demo ok\n");
    } catch (const Error& e) {
        std::printf("error %d: %s\n", e.status, e.what());
        return 2;
    }
    return 0;
}
