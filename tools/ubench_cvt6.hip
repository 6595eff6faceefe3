// Semantics + issue cost of v_cvt_scalef32_pk32_f16_{fp6,bf6} on gfx950 (developer tool): 32 six-bit floats
// (6 VGPRs) -> 32 fp16 (16 VGPRs) in one instruction.  Was a candidate for the wide fill's table expansion (one
// conversion per 16 slots instead of one per slot).  Measured on MI355X: the instruction occupies the VALU for
// ~64 cycles per wave (16 passes, one per destination register) -- exactly what 16 v_cvt_scalef32_pk_f16_bf8
// cost, so nothing is gained; kept as the record of that (the semantics dump below does not decode the element
// order, which was not pursued further).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_cvt6.hip -o tools/ubench_cvt6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>

typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef unsigned v16u __attribute__((ext_vector_type(16)));

template <int BF>
__device__ __forceinline__ v16u cvt32(v6u a, float sc) {
    v16u r;
    if (BF) asm volatile("v_cvt_scalef32_pk32_f16_bf6 %0, %1, %2" : "=v"(r) : "v"(a), "v"(sc));
    else asm volatile("v_cvt_scalef32_pk32_f16_fp6 %0, %1, %2" : "=v"(r) : "v"(a), "v"(sc));
    return r;
}
template <int BF>
__global__ void sem(const unsigned* in, unsigned* out, float sc) {
    v6u a;
    for (int i = 0; i < 6; ++i) a[i] = in[i];
    v16u r = cvt32<BF>(a, sc);
    if (threadIdx.x == 0) for (int i = 0; i < 16; ++i) out[i] = r[i];
}
template <int BF>
__global__ void rate(unsigned* out, int iters, unsigned seed) {
    v6u a;
    for (int i = 0; i < 6; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v16u v = cvt32<BF>(a, 1.0f);
            acc ^= v[0] ^ v[15];
            a[0] += acc & 1;
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
__global__ void rate_ref(unsigned* out, int iters, unsigned seed) {   // 8 x (xor, xor, and, add) without the cvt
    unsigned a = seed + threadIdx.x, acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile("" : "+v"(a));
            acc ^= a ^ (a >> 3);
            a += acc & 1;
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
static float h2f(uint16_t h) {
    int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? std::ldexp((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : std::ldexp(1.0f + m / 1024.0f, e - 15);
    return s ? -v : v;
}
int main() {
    unsigned h_in[6], h_out[16];
    unsigned *d_in, *d_out;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, 4096);
    // value i (0..31) = its own index as the 6-bit pattern: shows the element order and the decoding
    unsigned long long bits[3] = {0, 0, 0};
    for (int i = 0; i < 32; ++i) {
        const unsigned long long pat = (unsigned)i & 63u;
        const int pos = 6 * i;
        bits[pos / 64] |= pat << (pos % 64);
        if (pos % 64 > 58) bits[pos / 64 + 1] |= pat >> (64 - pos % 64);
    }
    for (int i = 0; i < 6; ++i) h_in[i] = (unsigned)(bits[i / 2] >> (32 * (i & 1)));
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    for (int bf = 0; bf < 2; ++bf) {
        if (bf) hipLaunchKernelGGL(sem<1>, dim3(1), dim3(64), 0, 0, d_in, d_out, 1.0f);
        else hipLaunchKernelGGL(sem<0>, dim3(1), dim3(64), 0, 0, d_in, d_out, 1.0f);
        hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
        printf("%s, patterns 0..31 in elements 0..31, scale 1.0:\n ", bf ? "bf6 (E3M2)" : "fp6 (E2M3)");
        for (int i = 0; i < 16; ++i) printf(" %g %g", h2f(h_out[i] & 0xffff), h2f(h_out[i] >> 16));
        printf("\n");
    }
    // patterns 32..63 (negative half)
    for (int i = 0; i < 3; ++i) bits[i] = 0;
    for (int i = 0; i < 32; ++i) {
        const unsigned long long pat = (unsigned)(32 + i) & 63u;
        const int pos = 6 * i;
        bits[pos / 64] |= pat << (pos % 64);
        if (pos % 64 > 58) bits[pos / 64 + 1] |= pat >> (64 - pos % 64);
    }
    for (int i = 0; i < 6; ++i) h_in[i] = (unsigned)(bits[i / 2] >> (32 * (i & 1)));
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem<1>, dim3(1), dim3(64), 0, 0, d_in, d_out, 1.0f);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    printf("bf6 patterns 32..63:\n ");
    for (int i = 0; i < 16; ++i) printf(" %g %g", h2f(h_out[i] & 0xffff), h2f(h_out[i] >> 16));
    printf("\n");
    for (int w : {1, 2, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000, blocks = 256 * w;
        float ms[3];
        for (int k = 0; k < 3; ++k) {
            auto launch = [&](int it) {
                if (k == 0) hipLaunchKernelGGL(rate_ref, dim3(blocks), dim3(256), 0, 0, d_out, it, 3u);
                else if (k == 1) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d_out, it, 3u);
                else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d_out, it, 3u);
            };
            launch(10);
            hipEventRecord(e0);
            launch(iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[k], e0, e1);
        }
        const double n = (double)iters * 8 * w;   // conversions per SIMD
        printf("waves/SIMD=%d: loop without cvt %.2f ns/iter, with fp6 cvt %.2f, with bf6 cvt %.2f  -> cvt = %.1f / %.1f cycles @2.4GHz\n",
               w, ms[0] * 1e6 / n, ms[1] * 1e6 / n, ms[2] * 1e6 / n, (ms[1] - ms[0]) * 1e6 / n * 2.4, (ms[2] - ms[0]) * 1e6 / n * 2.4);
    }
    return 0;
}
