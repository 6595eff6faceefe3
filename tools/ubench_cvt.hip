// Semantics + issue rate of v_cvt_scalef32_pk_f16_bf8 on gfx950 (developer tool).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_cvt.hip -o tools/ubench_cvt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>

__device__ __forceinline__ unsigned cvt_lo(unsigned a, float sc) {
    unsigned r;
    asm("v_cvt_scalef32_pk_f16_bf8 %0, %1, %2" : "=v"(r) : "v"(a), "v"(sc));
    return r;
}
__device__ __forceinline__ unsigned cvt_hi(unsigned a, float sc) {
    unsigned r;
    asm("v_cvt_scalef32_pk_f16_bf8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(r) : "v"(a), "v"(sc));
    return r;
}
__global__ void sem(const unsigned* in, unsigned* out) {
    out[2 * threadIdx.x] = cvt_lo(in[threadIdx.x], 1.0f);
    out[2 * threadIdx.x + 1] = cvt_hi(in[threadIdx.x], 1.0f);
}
__global__ void rate(unsigned* out, int iters, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = cvt_lo(a[i], 1.0f) + 0x01010101u;
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
static uint8_t bf8(float v) {  // E5M2 encode of small exact values
    if (std::isinf(v)) return v < 0 ? 0xFC : 0x7C;
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    v = std::fabs(v);
    int e = (int)std::floor(std::log2(v));
    float m = v / std::ldexp(1.0f, e) - 1.0f;  // [0,1)
    return s | (uint8_t)((e + 15) << 2) | (uint8_t)(m * 4);
}
static float h2f(uint16_t h) {
    int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? std::ldexp((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : std::ldexp(1.0f + m / 1024.0f, e - 15);
    return s ? -v : v;
}
int main() {
    const float vals[8] = {0, 1, 2, 3, 7, -1, -6, -INFINITY};
    unsigned h_in[64] = {0}, h_out[128];
    for (int i = 0; i < 4; ++i)
        h_in[i] = bf8(vals[2 * i]) | (bf8(vals[2 * i + 1]) << 8) | (bf8(vals[7 - 2 * i]) << 16) | (bf8(vals[6 - 2 * i]) << 24);
    unsigned *d_in, *d_out;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, sizeof h_out);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d_in, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i)
        printf("in %08x -> lo {%g, %g}  hi {%g, %g}\n", h_in[i], h2f(h_out[2 * i] & 0xffff), h2f(h_out[2 * i] >> 16),
               h2f(h_out[2 * i + 1] & 0xffff), h2f(h_out[2 * i + 1] >> 16));
    for (int w : {1, 2, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000, blocks = 256 * w;
        hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, d_out, 10, 3u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, d_out, iters, 3u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)iters * 16 * 8 * 2 * w;  // cvt + add per wave, per SIMD
        printf("waves/SIMD=%d: %.2f ns per wave-op (= %.2f cycles @2.4GHz) [cvt+add pairs]\n", w, ms * 1e6 / ops, ms * 1e6 / ops * 2.4);
    }
    return 0;
}
