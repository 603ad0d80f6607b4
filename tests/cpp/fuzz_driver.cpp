// tests/cpp/fuzz_driver.cpp -- runs a libFuzzer-style target (LLVMFuzzerTestOneInput: the reference's own fuzz/target_*.cpp, compiled unchanged
// against this repository's headers and host sources) under AddressSanitizer + UndefinedBehaviorSanitizer: first every file of a corpus
// directory as it is, then random mutations of them (bit flips, byte writes, truncations, duplications, splices of two files) from a fixed
// seed for a given number of seconds.  The reference builds these targets with -fsanitize=address,undefined,fuzzer (CMakeLists.txt:65-66); this
// driver replaces libFuzzer's main (no coverage feedback), which keeps the run bounded and reproducible inside a test.
//   fuzz_driver <corpus dir> <seconds> [seed]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>

extern "C" int LLVMFuzzerTestOneInput(const std::uint8_t *data, std::size_t size);

int main(int argc, char **argv) {
    if(argc < 3) {
        std::fprintf(stderr, "usage: %s <corpus dir> <seconds> [seed]\n", argv[0]);
        return 2;
    }
    std::vector<std::vector<std::uint8_t>> corpus;
    for(const auto &entry : std::filesystem::directory_iterator(argv[1])) {
        std::ifstream f(entry.path(), std::ios::binary);
        corpus.emplace_back((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    }
    if(corpus.empty()) {
        std::fprintf(stderr, "empty corpus\n");
        return 2;
    }
    for(const auto &input : corpus) {
        LLVMFuzzerTestOneInput(input.data(), input.size());
    }
    std::mt19937_64 rng(argc > 3 ? std::strtoull(argv[3], nullptr, 0) : 1);
    const double seconds = std::atof(argv[2]);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long runs = corpus.size();
    while(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        std::vector<std::uint8_t> input = corpus[rng() % corpus.size()];
        const int edits = 1 + static_cast<int>(rng() % 6);
        for(int e = 0; e < edits; e++) {
            switch(rng() % 6) {
                case 0:
                    if(!input.empty()) {
                        input[rng() % input.size()] ^= static_cast<std::uint8_t>(1U << (rng() % 8));
                    }
                    break;
                case 1:
                    if(!input.empty()) {
                        input[rng() % input.size()] = static_cast<std::uint8_t>(rng());
                    }
                    break;
                case 2:
                    if(!input.empty()) {
                        input.resize(rng() % input.size());
                    }
                    break;
                case 3:
                    if(!input.empty() && input.size() < (1U << 16)) {
                        const std::size_t from = rng() % input.size(), len = 1 + rng() % (input.size() - from);
                        const std::vector<std::uint8_t> piece(input.begin() + static_cast<long>(from), input.begin() + static_cast<long>(from + len));
                        input.insert(input.begin() + static_cast<long>(rng() % (input.size() + 1)), piece.begin(), piece.end());
                    }
                    break;
                case 4: {
                    const auto &other = corpus[rng() % corpus.size()];
                    if(!other.empty() && input.size() < (1U << 16)) {
                        const std::size_t from = rng() % other.size();
                        input.insert(input.begin() + static_cast<long>(rng() % (input.size() + 1)), other.begin() + static_cast<long>(from), other.end());
                    }
                    break;
                }
                default:
                    input.insert(input.begin() + static_cast<long>(rng() % (input.size() + 1)), static_cast<std::uint8_t>("0123456789.-e \n/vf"[rng() % 18]));
                    break;
            }
        }
        LLVMFuzzerTestOneInput(input.data(), input.size());
        runs++;
    }
    std::printf("%llu inputs, no sanitizer report\n", runs);
    return 0;
}
