// benchmark/benchmark.h -- a small stand-in for Google Benchmark (not installed in this image), written for this repository, so that the
// reference's benchmark/main.cpp compiles and runs UNCHANGED against libPathTrace.so.  It covers what that file uses: State (range-for
// timing loop, SetItemsProcessed), RegisterBenchmark(name, fn)->UseRealTime()->Unit(), Initialize, ReportUnrecognizedArguments,
// RunSpecifiedBenchmarks, DoNotOptimize, ClobberMemory.  Like the original it calls the benchmark function repeatedly with growing
// iteration counts until a run lasts --benchmark_min_time seconds (default 0.5) and reports real time per iteration and items per second.
// Flags: --benchmark_filter=<substring or regex>, --benchmark_min_time=<seconds>[s].
#ifndef PT_SHIM_BENCHMARK_H
#define PT_SHIM_BENCHMARK_H

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <regex>
#include <string>
#include <vector>

namespace benchmark {

    enum TimeUnit { kNanosecond, kMicrosecond, kMillisecond, kSecond };

    class State {
      public:
        explicit State(int64_t max_iterations) : max_iterations_(max_iterations) {}

        struct Iterator {
            State *state;
            int64_t left;
            bool operator!=(const Iterator &) {
                if(left > 0) {
                    return true;
                }
                state->finish();
                return false;
            }
            void operator++() { left--; }
            struct Value {};
            Value operator*() const { return Value{}; }
        };
        Iterator begin() {
            start_ = std::chrono::steady_clock::now();
            running_ = true;
            return Iterator{this, max_iterations_};
        }
        Iterator end() { return Iterator{this, 0}; }

        void SetItemsProcessed(int64_t items) { items_ = items; }
        int64_t items_processed() const { return items_; }
        void SetBytesProcessed(int64_t bytes) { bytes_ = bytes; }
        int64_t iterations() const { return max_iterations_; }
        double elapsed_seconds() const { return elapsed_; }

      private:
        void finish() {
            if(running_) {
                elapsed_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - start_).count();
                running_ = false;
            }
        }
        int64_t max_iterations_;
        int64_t items_ = 0, bytes_ = 0;
        double elapsed_ = 0.0;
        bool running_ = false;
        std::chrono::steady_clock::time_point start_;
    };

    namespace internal {
        class Benchmark {
          public:
            Benchmark(std::string name, std::function<void(State &)> fn) : name_(std::move(name)), fn_(std::move(fn)) {}
            Benchmark *UseRealTime() { return this; }
            Benchmark *Unit(TimeUnit unit) {
                unit_ = unit;
                return this;
            }
            std::string name_;
            std::function<void(State &)> fn_;
            TimeUnit unit_ = kNanosecond;
        };
        inline std::vector<std::unique_ptr<Benchmark>> &registry() {
            static std::vector<std::unique_ptr<Benchmark>> all;
            return all;
        }
        inline std::string &filter() {
            static std::string f;
            return f;
        }
        inline double &min_time() {
            static double t = 0.5;
            return t;
        }
    } // namespace internal

    template<class Fn>
    internal::Benchmark *RegisterBenchmark(const char *name, Fn &&fn) {
        internal::registry().push_back(std::make_unique<internal::Benchmark>(name, std::function<void(State &)>(fn)));
        return internal::registry().back().get();
    }

    inline void Initialize(int *argc, char **argv) {
        int kept = 1;
        for(int i = 1; i < *argc; i++) {
            const std::string arg = argv[i];
            if(arg.rfind("--benchmark_filter=", 0) == 0) {
                internal::filter() = arg.substr(19);
            }
            else if(arg.rfind("--benchmark_min_time=", 0) == 0) {
                internal::min_time() = std::atof(arg.substr(21).c_str());
            }
            else {
                argv[kept++] = argv[i];
            }
        }
        *argc = kept;
    }

    inline bool ReportUnrecognizedArguments(int argc, char **argv) {
        for(int i = 1; i < argc; i++) {
            std::fprintf(stderr, "%s: error: unrecognized command-line flag: %s\n", argv[0], argv[i]);
        }
        return argc > 1;
    }

    inline size_t RunSpecifiedBenchmarks() {
        size_t ran = 0;
        std::printf("%-32s %15s %12s %18s\n", "Benchmark", "Time", "Iterations", "items_per_second");
        for(const auto &b : internal::registry()) {
            if(!internal::filter().empty() && !std::regex_search(b->name_, std::regex(internal::filter()))) {
                continue;
            }
            int64_t iterations = 1;
            for(;;) {
                State state(iterations);
                b->fn_(state);
                const double seconds = state.elapsed_seconds();
                if(seconds >= internal::min_time() || iterations >= (int64_t(1) << 30)) {
                    const double scale = b->unit_ == kSecond ? 1.0 : b->unit_ == kMillisecond ? 1e3 : b->unit_ == kMicrosecond ? 1e6 : 1e9;
                    const char *unit = b->unit_ == kSecond ? "s" : b->unit_ == kMillisecond ? "ms" : b->unit_ == kMicrosecond ? "us" : "ns";
                    // Google Benchmark divides the items set by the benchmark by the run's total time; the reference sets the items of ONE
                    // iteration (benchmark/main.cpp:30), so the per-iteration rate is printed too
                    std::printf("%-32s %12.3f %s %12lld %18.6g  items_per_second_per_iteration=%.6g\n", (b->name_ + "/real_time").c_str(), seconds / double(iterations) * scale, unit,
                                static_cast<long long>(iterations), seconds > 0.0 ? double(state.items_processed()) / seconds : 0.0,
                                seconds > 0.0 ? double(state.items_processed()) * double(iterations) / seconds : 0.0);
                    std::fflush(stdout);
                    break;
                }
                const double want = internal::min_time() * 1.4 / std::max(seconds / double(iterations), 1e-9);
                iterations = std::max<int64_t>(iterations + 1, std::min<int64_t>(static_cast<int64_t>(want), iterations * 10));
            }
            ran++;
        }
        return ran;
    }

    template<class T>
    inline void DoNotOptimize(T const &value) {
        asm volatile("" : : "r,m"(value) : "memory");
    }
    inline void ClobberMemory() { asm volatile("" : : : "memory"); }

} // namespace benchmark

#endif
