// ubench_u16max.hip -- do the fp16 maximum instructions of gfx950 act as UNSIGNED 16-bit integer maxima on the
// bit patterns 0x0000..0x7BFF (all non-negative finite fp16 values, denormals included)?  Exhaustive over all
// pairs for v_pk_max_f16 / v_max_f16 (DPP/SDWA forms use the same ALU), all pairs x 64 third operands for
// v_pk_maximum3_f16, and v_cmp_eq_f16 as a pattern equality.  Prints the number of mismatches per instruction.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_u16max.hip -o tools/ubench_u16max
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void check(unsigned long long* bad) {
    const unsigned a = blockIdx.x;                       // 0 .. 0x7BFF
    unsigned long long e_pk = 0, e_max = 0, e_m3 = 0, e_cmp = 0, e_sdwa = 0;
    for (unsigned b = threadIdx.x; b <= 0x7BFFu; b += blockDim.x) {
        const unsigned want = a > b ? a : b;
        const unsigned pa = a | (b << 16), pb = b | (a << 16);
        unsigned r;
        asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(pa), "v"(pb));
        if ((r & 0xffffu) != want || (r >> 16) != want) ++e_pk;
        asm volatile("v_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        if ((r & 0xffffu) != want) ++e_max;
        asm volatile("v_max_f16_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(pa));
        if ((r & 0xffffu) != want) ++e_sdwa;
        unsigned long long m;
        asm volatile("v_cmp_eq_f16_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
        const bool eq = (m >> (threadIdx.x & 63)) & 1ull;
        if (eq != (a == b)) ++e_cmp;
        for (unsigned k = 0; k < 64; ++k) {
            const unsigned c = (k * 509u + b * 7u + a) % 0x7C00u;
            const unsigned pc = c | (c << 16);
            const unsigned w3 = want > c ? want : c;
            asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(pa), "v"(pb), "v"(pc));
            if ((r & 0xffffu) != w3 || (r >> 16) != w3) ++e_m3;
        }
    }
    if (e_pk) atomicAdd(&bad[0], e_pk);
    if (e_max) atomicAdd(&bad[1], e_max);
    if (e_m3) atomicAdd(&bad[2], e_m3);
    if (e_cmp) atomicAdd(&bad[3], e_cmp);
    if (e_sdwa) atomicAdd(&bad[4], e_sdwa);
}

// v_pk_sub_u16 clamp = per-field saturating subtraction; v_add_u32 of (hi << 16) + lo with a signed pair
__global__ void check_arith(unsigned long long* bad) {
    const unsigned a = blockIdx.x * 4 + 3;               // sample of the low field
    unsigned long long e_sub = 0, e_add = 0;
    for (unsigned b = threadIdx.x; b <= 0xFFFFu; b += blockDim.x) {
        const unsigned pa = a | ((b ^ 0x1234u) << 16), pb = b | ((a >> 1) << 16);
        unsigned r;
        asm volatile("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(pa), "v"(pb));
        const unsigned lo = a > b ? a - b : 0u, hi = (b ^ 0x1234u) > (a >> 1) ? (b ^ 0x1234u) - (a >> 1) : 0u;
        if (r != (lo | (hi << 16))) ++e_sub;
        // signed pair {tl, th} in [-8, 7] added to fields that stay inside 16 bits
        const int tl = (int)(b & 15) - 8, th = (int)((b >> 4) & 15) - 8;
        const unsigned fl = 0x0400u + (a & 0x3fffu), fh = 0x0400u + ((b * 3u) & 0x3fffu);
        const unsigned t = (unsigned)(th * 65536 + tl);
        const unsigned s = (fl | (fh << 16)) + t;
        if ((s & 0xffffu) != (unsigned)((int)fl + tl) || (s >> 16) != (unsigned)((int)fh + th)) ++e_add;
    }
    if (e_sub) atomicAdd(&bad[5], e_sub);
    if (e_add) atomicAdd(&bad[6], e_add);
}

int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 64);
    (void)hipMemset(d, 0, 64);
    hipLaunchKernelGGL(check, dim3(0x7C00), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(check_arith, dim3(0x4000), dim3(256), 0, 0, d);
    unsigned long long h[8];
    (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("patterns 0x0000..0x7BFF as unsigned integers -- mismatches (0 = the instruction is an exact u16 maximum / equality):\n");
    printf("  v_pk_max_f16       %llu\n  v_max_f16          %llu\n  v_pk_maximum3_f16  %llu\n  v_cmp_eq_f16       %llu\n  v_max_f16_sdwa     %llu\n",
           h[0], h[1], h[2], h[3], h[4]);
    printf("  v_pk_sub_u16 clamp %llu\n  32-bit add of a signed pair %llu\n", h[5], h[6]);
    return (h[0] | h[1] | h[2] | h[3] | h[4] | h[5] | h[6]) ? 1 : 0;
}
