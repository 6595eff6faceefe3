// Micro-benchmark of VALU issue rates on gfx950 for the instructions the fill kernel is made of.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)));
}
__device__ __forceinline__ unsigned pk_adds(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)));
}

__device__ __forceinline__ unsigned pk_max3h(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned pk_maxh(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_addh(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// MODE 7: 8 independent v_pk_maximum3_f16; 8: one dependent v_pk_maximum3_f16 chain;
// 9: 8 independent chains of pk_max_f16 / pk_add_f16 / pk_maximum3_f16 (the 3-op cell update)
// MODE 0: 8 independent pk_max chains; 1: 8 independent i32 max chains; 2: one dependent pk_max chain
// 3: 8 independent chains alternating pk_max / pk_add(clamp); 4: one dependent i32 chain
// 5: 8 independent DPP row_shr max (i32); 6: ds_bpermute chain
template <int MODE>
__global__ void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    unsigned b = seed ^ 0x10001u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = pk_max(a[i], b + i);
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = (unsigned)max((int)a[i], (int)(b + i));
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[0] = pk_max(a[0], b + i) + 0;
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = (r & 1) ? pk_max(a[i], b + i) : pk_adds(a[i], b);
            } else if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[0] = (unsigned)max((int)a[0], (int)(b + i));
            } else if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    a[i] = (unsigned)max((int)a[i], __builtin_amdgcn_update_dpp(0, (int)a[i], 0x111, 0xf, 0xf, false));
            } else if (MODE == 7) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = pk_max3h(a[i], b, a[(i + 1) & 7]);
            } else if (MODE == 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[0] = pk_max3h(a[0], b, a[1 + (i & 3)]);
            } else if (MODE == 9) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned u = pk_maxh(a[i], b);
                    const unsigned v = pk_addh(u, a[(i + 3) & 7]);
                    a[i] = pk_max3h(a[(i + 1) & 7], v, a[i]);
                }
            } else if (MODE == 6) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[0] = (unsigned)__shfl_up((int)a[0], 1) + i;
            }
            b += 0x10001u;
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
    unsigned* d;
    hipMalloc(&d, 4096);
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks: 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 3u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 16 * 8;           // counted ops per wave
    const double per_simd = instr * waves_per_simd;        // ops issued per SIMD
    const double ns_per_op = ms * 1e6 / per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f ns per wave-op per SIMD (= %.2f cycles @2.4GHz)\n", name,
           waves_per_simd, ms, ns_per_op, ns_per_op * 2.4);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("pk_max_i16 indep x8", w);
        run<1>("max_i32 indep x8", w);
        run<3>("pk_max/pk_add_sat indep x8", w);
        run<2>("pk_max_i16 dependent", w);
        run<4>("max_i32 dependent", w);
        run<5>("max_i32 dpp row_shr indep", w);
        run<6>("ds_bpermute dependent", w);
        run<7>("pk_maximum3_f16 indep x8", w);
        run<8>("pk_maximum3_f16 dependent", w);
        run<9>("f16 max/add/max3 (x3 ops)", w);
    }
    return 0;
}
