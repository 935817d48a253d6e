// Test helper for mercer_research_amd/csrc/host/formats.hpp (no GPU needed):
//   format_check bincode <in> <out>   load rcn.bin, re-serialise
//   format_check png <file>           print "H W" then the pixel matrix bytes in hex
#include <cstdio>
#include <string>

#include "../../mercer_research_amd/csrc/host/formats.hpp"

using namespace rcn::host;

int main(int argc, char** argv) {
    try {
        if (argc == 4 && std::string(argv[1]) == "bincode") {
            write_file(argv[3], checkpoint_dumps(checkpoint_loads(read_file(argv[2]))));
            return 0;
        }
        if (argc == 3 && std::string(argv[1]) == "png") {
            const GrayImage g = png_to_pixel_matrix(read_file(argv[2]));
            std::printf("%d %d\n", g.h, g.w);
            for (uint8_t v : g.px) std::printf("%02x", v);
            std::printf("\n");
            return 0;
        }
    } catch (const InvalidGrayscaleImageError& e) { std::printf("InvalidGrayscaleImageError\n"); return 3; }
    catch (const FormatError& e) { std::printf("FormatError: %s\n", e.what()); return 4; }
    return 2;
}
