// Minimal stand-in for the parts of GoogleTest/GoogleMock that the PathTrace test programs use (neither is installed in this
// image): TEST, EXPECT_THAT with streaming, the matchers Eq Ge Gt Le Lt NotNull FloatEq FloatNear, InitGoogleTest,
// RUN_ALL_TESTS.  Test infrastructure only.
#ifndef PT_GTEST_SHIM_H
#define PT_GTEST_SHIM_H

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace testing {

    struct TestCase {
        std::string name;
        std::function<void()> body;
    };
    inline std::vector<TestCase> &registry() {
        static std::vector<TestCase> tests;
        return tests;
    }
    inline int &failures() {
        static int count = 0;
        return count;
    }
    struct Registrar {
        Registrar(const char *suite, const char *name, std::function<void()> body) { registry().push_back({std::string(suite) + "." + name, std::move(body)}); }
    };

    inline void InitGoogleTest(int *, char **) {}

    inline int RunAllTests() {
        int failed_tests = 0;
        // PT_TEST_SKIP: comma-separated substrings of test names to skip (e.g. the GPU tests on a machine without one)
        const char *skip_env = std::getenv("PT_TEST_SKIP");
        const std::string skip = skip_env != nullptr ? skip_env : "";
        for(const auto &t : registry()) {
            bool skipped = false;
            for(size_t start = 0; start < skip.size();) {
                const size_t end = skip.find(',', start);
                const std::string word = skip.substr(start, end == std::string::npos ? std::string::npos : end - start);
                skipped = skipped || (!word.empty() && t.name.find(word) != std::string::npos);
                start = end == std::string::npos ? skip.size() : end + 1;
            }
            if(skipped) {
                std::cout << "[ SKIPPED  ] " << t.name << std::endl;
                continue;
            }
            const int before = failures();
            std::cout << "[ RUN      ] " << t.name << std::endl;
            try {
                t.body();
            }
            catch(const std::exception &e) {
                std::cout << "  exception: " << e.what() << std::endl;
                failures()++;
            }
            const bool ok = failures() == before;
            failed_tests += ok ? 0 : 1;
            std::cout << (ok ? "[       OK ] " : "[  FAILED  ] ") << t.name << std::endl;
        }
        std::cout << "[==========] " << registry().size() << " tests, " << failed_tests << " failed" << std::endl;
        return failed_tests == 0 ? 0 : 1;
    }

    // prints a value if it can be streamed
    template<typename T, typename = void>
    struct Printer {
        static void print(std::ostream &os, const T &) { os << "<value>"; }
    };
    template<typename T>
    struct Printer<T, std::void_t<decltype(std::declval<std::ostream &>() << std::declval<const T &>())>> {
        static void print(std::ostream &os, const T &v) { os << v; }
    };

    template<typename Expected, typename Compare>
    struct ComparisonMatcher {
        Expected expected;
        const char *description;
        template<typename Actual>
        bool matches(const Actual &actual) const {
            return Compare()(actual, expected);
        }
        void describe(std::ostream &os) const {
            os << description << " ";
            Printer<Expected>::print(os, expected);
        }
    };
    struct EqOp {
        template<typename A, typename B>
        bool operator()(const A &a, const B &b) const { return a == b; }
    };
    struct GeOp {
        template<typename A, typename B>
        bool operator()(const A &a, const B &b) const { return a >= b; }
    };
    struct GtOp {
        template<typename A, typename B>
        bool operator()(const A &a, const B &b) const { return a > b; }
    };
    struct LeOp {
        template<typename A, typename B>
        bool operator()(const A &a, const B &b) const { return a <= b; }
    };
    struct LtOp {
        template<typename A, typename B>
        bool operator()(const A &a, const B &b) const { return a < b; }
    };
    template<typename T>
    ComparisonMatcher<T, EqOp> Eq(T v) { return {v, "is equal to"}; }
    template<typename T>
    ComparisonMatcher<T, GeOp> Ge(T v) { return {v, "is >="}; }
    template<typename T>
    ComparisonMatcher<T, GtOp> Gt(T v) { return {v, "is >"}; }
    template<typename T>
    ComparisonMatcher<T, LeOp> Le(T v) { return {v, "is <="}; }
    template<typename T>
    ComparisonMatcher<T, LtOp> Lt(T v) { return {v, "is <"}; }

    struct NotNullMatcher {
        template<typename P>
        bool matches(const P &p) const { return p != nullptr; }
        void describe(std::ostream &os) const { os << "isn't NULL"; }
    };
    inline NotNullMatcher NotNull() { return {}; }

    // equal within 4 units in the last place, like GoogleTest's FloatEq
    struct FloatEqMatcher {
        float expected;
        bool matches(float actual) const {
            if(std::isnan(actual) || std::isnan(expected)) {
                return false;
            }
            auto biased = [](float f) {
                uint32_t u;
                std::memcpy(&u, &f, 4);
                return (u & 0x80000000U) ? ~u + 1 : u | 0x80000000U;
            };
            const uint32_t a = biased(actual), b = biased(expected);
            return (a > b ? a - b : b - a) <= 4;
        }
        void describe(std::ostream &os) const { os << "is approximately " << expected; }
    };
    inline FloatEqMatcher FloatEq(float v) { return {v}; }

    struct FloatNearMatcher {
        float expected, tolerance;
        bool matches(float actual) const { return std::fabs(actual - expected) <= tolerance; }
        void describe(std::ostream &os) const { os << "is within " << tolerance << " of " << expected; }
    };
    inline FloatNearMatcher FloatNear(float v, float tolerance) { return {v, tolerance}; }

    // collects the user's << message and reports when it goes out of scope
    class Failure {
      public:
        Failure(const char *file, int line, std::string what) : active(true) { text << file << ":" << line << ": Failure\n" << what << "\n"; }
        Failure() : active(false) {}
        Failure(Failure &&o) noexcept : active(o.active) {
            text << o.text.str();
            o.active = false;
        }
        ~Failure() {
            if(active) {
                failures()++;
                std::cout << text.str() << std::endl;
            }
        }
        template<typename T>
        Failure &operator<<(const T &v) {
            if(active) {
                text << v;
            }
            return *this;
        }

      private:
        bool active;
        std::ostringstream text;
    };

    template<typename Actual, typename Matcher>
    Failure expectThat(const char *file, int line, const char *expression, const Actual &actual, const Matcher &matcher) {
        if(matcher.matches(actual)) {
            return Failure();
        }
        std::ostringstream what;
        what << "Value of: " << expression << "\nExpected: ";
        matcher.describe(what);
        what << "\n  Actual: ";
        Printer<Actual>::print(what, actual);
        return Failure(file, line, what.str());
    }

} // namespace testing

#define TEST(suite, name)                                                                    \
    static void pt_test_##suite##_##name();                                                  \
    static ::testing::Registrar pt_registrar_##suite##_##name(#suite, #name, pt_test_##suite##_##name); \
    static void pt_test_##suite##_##name()

#define EXPECT_THAT(value, matcher) ::testing::expectThat(__FILE__, __LINE__, #value, (value), (matcher))
#define RUN_ALL_TESTS() ::testing::RunAllTests()

#endif
