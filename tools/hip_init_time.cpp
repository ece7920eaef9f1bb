// Floor of any HIP process on this box: runtime initialisation, first context, first allocation, first kernel-less sync.
// hipcc -O2 -o tools/_build/hip_init_time tools/hip_init_time.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main()
{
    auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    hipInit(0);
    const double a = since();
    hipSetDevice(0);
    hipFree(nullptr);
    const double b = since();
    void* p = nullptr;
    hipMalloc(&p, 1 << 20);
    hipStream_t s;
    hipStreamCreate(&s);
    hipStreamSynchronize(s);
    const double c = since();
    void* h = nullptr;
    hipHostMalloc(&h, 64 << 20, hipHostMallocDefault);
    const double d = since();
    printf("hipInit %.3f s, first context %.3f s, malloc + stream %.3f s, 64 MB page-locked %.3f s\n", a, b - a, c - b, d - c);
    return 0;
}
