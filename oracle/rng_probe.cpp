// rng_probe.cpp -- TEST INFRASTRUCTURE ONLY.
// Prints the stream that libstdc++'s std::mt19937_64 + std::uniform_real_distribution<T>(-1,1)
// produce for a given seed, and the complex sample polar(|draw|, pi*draw) with g++'s
// argument evaluation order -- the generator recipe of the reference's getter_random
// (include/dlaf/util_matrix.h:148-179), written against the C++ standard library only.
// Used by oracle/gen_golden.py to make tests/golden/rng_stream.json.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>

template <class T>
struct Getter {
  explicit Getter(long seed) : engine(static_cast<std::size_t>(seed)) {}
  T operator()() { return sampler(engine); }
  std::mt19937_64 engine;
  std::uniform_real_distribution<T> sampler{-1, 1};
};

template <class T>
std::complex<T> complex_sample(Getter<T>& g) {
  // both draws are function arguments: evaluation order is the compiler's choice
  return std::polar<T>(std::abs(g()), static_cast<T>(M_PI) * g());
}

int main(int argc, char** argv) {
  long seed = argc > 1 ? std::atol(argv[1]) : 0;
  int count = argc > 2 ? std::atoi(argv[2]) : 8;
  {
    std::mt19937_64 e(static_cast<std::size_t>(seed));
    std::printf("raw");
    for (int i = 0; i < count; ++i) {
      unsigned long long v = e();
      std::printf(" %llu", v);
    }
    std::printf("\n");
  }
  {
    Getter<double> g(seed);
    std::printf("d");
    for (int i = 0; i < count; ++i) {
      double v = g();
      std::printf(" %.17g", v);
    }
    std::printf("\n");
  }
  {
    Getter<float> g(seed);
    std::printf("s");
    for (int i = 0; i < count; ++i) {
      float v = g();
      std::printf(" %.9g", v);
    }
    std::printf("\n");
  }
  {
    Getter<double> g(seed);
    std::printf("z");
    for (int i = 0; i < count / 2; ++i) {
      std::complex<double> v = complex_sample(g);
      std::printf(" %.17g %.17g", v.real(), v.imag());
    }
    std::printf("\n");
  }
  {
    Getter<float> g(seed);
    std::printf("c");
    for (int i = 0; i < count / 2; ++i) {
      std::complex<float> v = complex_sample(g);
      std::printf(" %.9g %.9g", v.real(), v.imag());
    }
    std::printf("\n");
  }
  return 0;
}
