// Developer probe (GPU box): is  q' = fma(x - q den, rcp, q), q = x rcp, rcp = RN(1 / den)  (Markstein's correction with a correctly rounded
// reciprocal) BITWISE the IEEE quotient x / den?  Random significands and exponents of x (|x| in the guarded range of csrc/sor_kernels.hip),
// a list of divisors (the 2 dx^2 + 2 dy^2 of the reference's grids, random ones, significands near all-ones), float64 and float32.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/fastdiv_check.hip -o /tmp/fastdiv_check && /tmp/fastdiv_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

template <typename T> struct Bits;
template <> struct Bits<double> { using U = uint64_t; static constexpr int MB = 52, EB = 1023, ER = 890; };
template <> struct Bits<float> { using U = uint32_t; static constexpr int MB = 23, EB = 127, ER = 95; };

template <typename T>
__global__ void check(T den, T rcp, uint64_t seed, int per_thread, unsigned long long* bad, unsigned long long* shown) {
    using U = typename Bits<T>::U;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0;
    for (int it = 0; it < per_thread; ++it) {
        const uint64_t r = mix(seed + tid * 0x100000001b3ull + (uint64_t)it * 0x9e3779b97f4a7c15ull);
        const U mant = (U)(r & (((uint64_t)1 << Bits<T>::MB) - 1));
        const int e = (int)((r >> 53) % (2 * Bits<T>::ER + 1)) - Bits<T>::ER + Bits<T>::EB;       // exponent field: |x| in 2^-ER .. 2^ER
        const U sign = (U)((r >> 52) & 1) << (sizeof(T) * 8 - 1);
        U xb = sign | ((U)e << Bits<T>::MB) | (it % 7 == 0 ? (mant | (mant >> 1) | (mant >> 2)) : it % 11 == 0 ? (mant & (mant >> 1) & (mant >> 3)) : mant);
        T x; memcpy(&x, &xb, sizeof(T));
        const T q = x * rcp;
        const T rr = fma(-q, den, x);
        const T fast = fma(rr, rcp, q);
        const T slow = x / den;
        U fb, sb; memcpy(&fb, &fast, sizeof(T)); memcpy(&sb, &slow, sizeof(T));
        if (fb != sb) { ++nb; if (atomicAdd(shown, 1ull) < 8) printf("MISMATCH x=%a den=%a fast=%a slow=%a\n", (double)x, (double)den, (double)fast, (double)slow); }
    }
    if (nb) atomicAdd(bad, nb);
}

template <typename T>
unsigned long long run(const std::vector<double>& dens, const char* name) {
    unsigned long long *bad, *shown, total = 0, samples = 0;
    hipMalloc(&bad, 8); hipMalloc(&shown, 8);
    for (size_t d = 0; d < dens.size(); ++d) {
        const T den = (T)dens[d], rcp = (T)1 / den;
        hipMemset(bad, 0, 8); hipMemset(shown, 0, 8);
        const int blocks = 4096, threads = 256, per = 256;
        hipLaunchKernelGGL(check<T>, dim3(blocks), dim3(threads), 0, 0, den, rcp, 0x1234567ull * (d + 1), per, bad, shown);
        unsigned long long h = 0; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        total += h; samples += (unsigned long long)blocks * threads * per;
        if (h) printf("%s den=%a: %llu mismatches\n", name, (double)den, h);
    }
    printf("%s: %zu divisors, %llu samples, %llu mismatches\n", name, dens.size(), samples, total);
    hipFree(bad); hipFree(shown);
    return total;
}

int main() {
    std::vector<double> dens;
    for (int n : {51, 50, 64, 41, 101, 33, 129, 17}) { const double h = 2.0 / (n - 1); dens.push_back(2 * h * h + 2 * h * h); dens.push_back(2 * h * h + 2 * (1.5 * h) * (1.5 * h)); }
    uint64_t s = 42;
    for (int i = 0; i < 24; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; dens.push_back(ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 40) - 20)); }
    for (int i = 0; i < 6; ++i) dens.push_back(ldexp(2.0 - ldexp(1.0, -52 + i) - ldexp(1.0, -30 - i), i - 3));      // significands of (nearly) all ones
    dens.push_back(1.0); dens.push_back(3.0); dens.push_back(0.1); dens.push_back(1e-3); dens.push_back(7e5);
    const unsigned long long b64 = run<double>(dens, "float64"), b32 = run<float>(dens, "float32");
    return (b64 || b32) ? 1 : 0;
}
