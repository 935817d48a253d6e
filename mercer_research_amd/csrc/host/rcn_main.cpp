// rcn_main.cpp -- the reference CLI (rcn/src/main.rs:8-79) over the gfx950 C ABI: same flags and defaults, the
// hard-coded architecture conv(Same), pool(Max), conv(Same), pool(Max) + [30], loads ./rcn.bin when present, trains,
// prints "Epoch {}: {}/{} [{:.2}%]" per epoch (rcn.rs:158-164) and writes ./rcn.bin in the reference's bincode format.
// Build: g++ -std=c++17 -O2 rcn_main.cpp -L<dir of librcn_hip.so> -lrcn_hip -lz -o rcn_hip_cli
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <string>

#include "rcn.hpp"

using namespace rcn::host;

int main(int argc, char** argv) {
    size_t num_classes = 10, training_class_size = 500, testing_class_size = 500, batches = 10, epochs = 30;   // main.rs:8-42
    std::string training_path = "images/mnist_png/training", testing_path = "images/mnist_png/testing", model_path = "./rcn.bin";
    double learning_rate = 3.0;
    int in_h = 28, in_w = 28, dtype = RCN_HIP_F32;
    uint64_t seed = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "error: a value is required for '%s'\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "-n" || a == "--num-classes") num_classes = std::strtoull(val(), nullptr, 10);
        else if (a == "--training-path") training_path = val();
        else if (a == "--testing-path") testing_path = val();
        else if (a == "--training-class-size") training_class_size = std::strtoull(val(), nullptr, 10);
        else if (a == "--testing-class-size") testing_class_size = std::strtoull(val(), nullptr, 10);
        else if (a == "-l" || a == "--learning-rate") learning_rate = std::strtod(val(), nullptr);
        else if (a == "-b" || a == "--batches") batches = std::strtoull(val(), nullptr, 10);
        else if (a == "-e" || a == "--epochs") epochs = std::strtoull(val(), nullptr, 10);
        else if (a == "--model-path") model_path = val();                    // the remaining flags are not in the reference
        else if (a == "--input-shape") { in_h = std::atoi(val()); in_w = std::atoi(val()); }
        else if (a == "--dtype") dtype = std::strcmp(val(), "f64") == 0 ? RCN_HIP_F64 : RCN_HIP_F32;
        else if (a == "--seed") seed = std::strtoull(val(), nullptr, 10);
        else { std::fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
    }
    try {
        std::unique_ptr<RCN> model;
        if (std::filesystem::exists(model_path)) {                             // main.rs:47-50
            model = RCN::load(model_path, in_h, in_w, dtype, 0);
            model->set_paths(training_path, testing_path);
        } else {                                                               // main.rs:51-62
            model = std::make_unique<RCN>(num_classes, std::vector<RCNLayer>{RCNLayer::Convolve2D(Padding::Same), RCNLayer::Pool2D(Pooling::Max),
                                                                             RCNLayer::Convolve2D(Padding::Same), RCNLayer::Pool2D(Pooling::Max)},
                                          std::vector<size_t>{30}, training_path, testing_path, in_h, in_w, dtype, 0);
        }
        // main.rs:65-74 matches on train's Result, but load_data unwrap()s every I/O and decode error (rcn.rs:369-398), so a
        // bad file is a panic there, not an Err: FormatError therefore takes the panic exit below
        model->train(batches, epochs, learning_rate, training_class_size, testing_class_size, seed, true);
        model->save(model_path);                                               // main.rs:77
    } catch (const Panic& e) {
        std::fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
        return 101;                                                            // a Rust panic exits with 101
    } catch (const FormatError& e) {
        std::fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
        return 101;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
