// ubench_issue.hip -- how many cycles does one SIMD of gfx950 need per wave64 VALU instruction, per opcode
// and per number of resident waves?  ("weak" 1: is the ceiling of the fill kernels one
// wave-instruction per SIMD per 4 cycles or per 2?)
//
// Every kernel below is a stream of INDEPENDENT instructions of one opcode (8 accumulators, 16 x 8 = 128
// instructions per loop trip, written as `asm volatile` so that nothing is folded, reordered or dual-packed
// by the compiler), or one dependent chain of it (latency).  Each wave stamps s_memtime before and after and
// records which SIMD it ran on (HW_REG_HW_ID / HW_REG_XCC_ID); the host groups the waves by SIMD and
// reports, over the SIMDs that really held W waves,
//     cycles per wave-instruction per SIMD = (last end - first start) / (W * instructions per wave)
// next to the wall-clock figure of the whole launch (all 1024 SIMDs busy, so DVFS is included).
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct Stamp {
    unsigned long long t0, t1;
    unsigned hwid, xcc;
};

#define HEAD                                                                                       \
    unsigned a[8];                                                                                 \
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;                               \
    unsigned b = seed ^ 0x3c003c00u, c = seed + 0x40004000u;                                       \
    unsigned long long a64[8];                                                                     \
    for (int i = 0; i < 8; ++i) a64[i] = a[i];                                                     \
    (void)a64; (void)b; (void)c;                                                                   \
    __syncthreads();                                                                               \
    const unsigned long long t0 = __builtin_readcyclecounter();

#define TAIL                                                                                       \
    const unsigned long long t1 = __builtin_readcyclecounter();                                    \
    unsigned s = 0;                                                                                \
    for (int i = 0; i < 8; ++i) s ^= a[i] ^ (unsigned)a64[i];                                      \
    if (s == 0x12345678u) out[threadIdx.x] = s;                                                    \
    if ((threadIdx.x & 63) == 0) {                                                                 \
        unsigned hw, xc;                                                                           \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                           \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));                          \
        Stamp st{t0, t1, hw, xc};                                                                  \
        stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;                                 \
    }

// An instruction is a macro I(d, b, c): d = accumulator (read + write), b = a wave-uniform VGPR operand,
// c = another accumulator (read only).  The 8 independent instructions of a group sit in ONE asm block
// (the compiler puts an `s_nop` between two asm blocks that define VGPRs).
#define GROUP8(I) I("%0", "%8", "%3") I("%1", "%8", "%4") I("%2", "%8", "%5") I("%3", "%8", "%6")             \
                  I("%4", "%8", "%7") I("%5", "%8", "%0") I("%6", "%8", "%1") I("%7", "%8", "%2")
#define KERNEL_INDEP(NAME, I)                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, Stamp* stamps, int iters, unsigned seed) { \
        HEAD                                                                                       \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                         \
                asm volatile(GROUP8(I) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), \
                             "+v"(a[6]), "+v"(a[7]) : "v"(b) : "s20", "s21", "s22", "s23", "vcc");  \
        }                                                                                          \
        TAIL                                                                                       \
    }
// one dependent chain (latency): every instruction reads the previous result
#define DEP8(I) I("%0", "%1", "%2") I("%0", "%1", "%2") I("%0", "%1", "%2") I("%0", "%1", "%2")                \
                I("%0", "%1", "%2") I("%0", "%1", "%2") I("%0", "%1", "%2") I("%0", "%1", "%2")
#define KERNEL_DEP(NAME, I)                                                                        \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, Stamp* stamps, int iters, unsigned seed) { \
        HEAD                                                                                       \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                         \
                asm volatile(DEP8(I) : "+v"(a[0]) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23", "vcc"); \
        }                                                                                          \
        TAIL                                                                                       \
    }
// 64-bit accumulators
#define KERNEL_INDEP64(NAME, I)                                                                    \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, Stamp* stamps, int iters, unsigned seed) { \
        HEAD                                                                                       \
        unsigned long long b64 = ((unsigned long long)b << 32) | c;                                \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                         \
                asm volatile(GROUP8(I) : "+v"(a64[0]), "+v"(a64[1]), "+v"(a64[2]), "+v"(a64[3]), "+v"(a64[4]), \
                             "+v"(a64[5]), "+v"(a64[6]), "+v"(a64[7]) : "v"(b64) : "vcc");          \
        }                                                                                          \
        TAIL                                                                                       \
    }

#define I_PK_MAX_F16(d, b, c) "v_pk_max_f16 " d ", " d ", " b "\n\t"
#define I_PK_ADD_F16(d, b, c) "v_pk_add_f16 " d ", " d ", " b "\n\t"
#define I_PK_MAXIMUM3(d, b, c) "v_pk_maximum3_f16 " d ", " d ", " b ", " c "\n\t"
#define I_PK_FMA_F16(d, b, c) "v_pk_fma_f16 " d ", " d ", " b ", " c "\n\t"
#define I_MAX_F16(d, b, c) "v_max_f16 " d ", " d ", " b "\n\t"
#define I_ADD_F32(d, b, c) "v_add_f32 " d ", " d ", " b "\n\t"
#define I_FMA_F32(d, b, c) "v_fma_f32 " d ", " d ", " b ", " c "\n\t"
#define I_ADD_U32(d, b, c) "v_add_u32 " d ", " d ", " b "\n\t"
#define I_MOV_B32(d, b, c) "v_mov_b32 " d ", " c "\n\t"
#define I_MAX_I32(d, b, c) "v_max_i32 " d ", " d ", " b "\n\t"
#define I_MAX3_I32(d, b, c) "v_max3_i32 " d ", " d ", " b ", " c "\n\t"
#define I_PK_MAX_I16(d, b, c) "v_pk_max_i16 " d ", " d ", " b "\n\t"
#define I_PK_ADD_I16(d, b, c) "v_pk_add_i16 " d ", " d ", " b " clamp\n\t"
#define I_MAX_F16_DPP(d, b, c) "v_max_f16_dpp " d ", " d ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_MOV_DPP_WSHR(d, b, c) "v_mov_b32_dpp " d ", " c " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_MOV_DPP_BCAST(d, b, c) "v_mov_b32_dpp " d ", " c " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
#define I_MAX_F16_SDWA(d, b, c) "v_max_f16_sdwa " d ", " d ", " b " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
#define I_BITOP3(d, b, c) "v_bitop3_b32 " d ", " d ", " b ", " c " bitop3:0x96\n\t"
#define I_BFI(d, b, c) "v_bfi_b32 " d ", " b ", " d ", " c "\n\t"
#define I_PERM(d, b, c) "v_perm_b32 " d ", " d ", " b ", " c "\n\t"
#define I_AND_OR(d, b, c) "v_and_or_b32 " d ", " d ", " b ", " c "\n\t"
#define I_ALIGNBIT(d, b, c) "v_alignbit_b32 " d ", " d ", " b ", 3\n\t"
#define I_CVT_BF8(d, b, c) "v_cvt_scalef32_pk_f16_bf8 " d ", " c ", 1.0\n\t"
#define I_CVT_F16_F32(d, b, c) "v_cvt_f16_f32 " d ", " c "\n\t"
#define I_LSHL_OR(d, b, c) "v_lshl_or_b32 " d ", " d ", 1, " b "\n\t"
#define I_CELL3(d, b, c) "v_pk_max_f16 " d ", " d ", " b "\n\tv_pk_add_f16 " d ", " d ", " c "\n\tv_pk_maximum3_f16 " d ", " d ", " b ", " c "\n\t"
#define I_VALU_SALU(d, b, c) "v_pk_max_f16 " d ", " d ", " b "\n\ts_add_u32 s20, s20, 1\n\t"
#define I_VALU_SNOP(d, b, c) "v_pk_max_f16 " d ", " d ", " b "\n\ts_nop 0\n\t"
#define I_SALU(d, b, c) "s_add_u32 s20, s20, 1\n\t"
#define I_SNOP(d, b, c) "s_nop 0\n\t"
#define I_DEP_DPP(d, b, c) "s_nop 1\n\tv_max_f16_dpp " d ", " d ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_LSHL_ADD_U64(d, b, c) "v_lshl_add_u64 " d ", " d ", 0, " b "\n\t"
#define I_ADD_F64(d, b, c) "v_add_f64 " d ", " d ", " b "\n\t"
#define I_PK_ADD_F32(d, b, c) "v_pk_add_f32 " d ", " d ", " b "\n\t"
#define I_CMP_EQ_F16(d, b, c) "v_cmp_eq_f16_e64 s[22:23], " d ", " b "\n\t"
#define I_CMP_EQ_F16_VCC(d, b, c) "v_cmp_eq_f16_e32 vcc, " d ", " b "\n\t"
#define I_READLANE(d, b, c) "v_readlane_b32 s21, " d ", 63\n\t"
#define I_READFIRSTLANE(d, b, c) "v_readfirstlane_b32 s21, " d "\n\t"

#define I_MAX_F32(d, b, c) "v_max_f32 " d ", " d ", " b "\n\t"
#define I_MUL_F32(d, b, c) "v_mul_f32 " d ", " d ", " b "\n\t"
#define I_ADD_F16(d, b, c) "v_add_f16 " d ", " d ", " b "\n\t"
#define I_MAX3_F16(d, b, c) "v_max3_f16 " d ", " d ", " b ", " c "\n\t"
#define I_MAX3_F32(d, b, c) "v_max3_f32 " d ", " d ", " b ", " c "\n\t"
#define I_MAXIMUM3_F32(d, b, c) "v_maximum3_f32 " d ", " d ", " b ", " c "\n\t"
#define I_MAX_U16(d, b, c) "v_max_u16 " d ", " d ", " b "\n\t"
#define I_MAX_I16(d, b, c) "v_max_i16 " d ", " d ", " b "\n\t"
#define I_MAX_U32(d, b, c) "v_max_u32 " d ", " d ", " b "\n\t"
#define I_MIN_U32(d, b, c) "v_min_u32 " d ", " d ", " b "\n\t"
#define I_SUB_U32(d, b, c) "v_sub_u32 " d ", " d ", " b "\n\t"
#define I_AND_B32(d, b, c) "v_and_b32 " d ", " d ", " b "\n\t"
#define I_OR_B32(d, b, c) "v_or_b32 " d ", " d ", " b "\n\t"
#define I_XOR_B32(d, b, c) "v_xor_b32 " d ", " d ", " b "\n\t"
#define I_LSHLREV(d, b, c) "v_lshlrev_b32 " d ", 1, " d "\n\t"
#define I_LSHRREV(d, b, c) "v_lshrrev_b32 " d ", 1, " d "\n\t"
#define I_ASHRREV(d, b, c) "v_ashrrev_i32 " d ", 1, " d "\n\t"
#define I_CNDMASK(d, b, c) "v_cndmask_b32 " d ", " d ", " b ", vcc\n\t"
#define I_ADD3_U32(d, b, c) "v_add3_u32 " d ", " d ", " b ", " c "\n\t"
#define I_ADD_LSHL(d, b, c) "v_add_lshl_u32 " d ", " d ", " b ", 1\n\t"
#define I_LSHL_ADD(d, b, c) "v_lshl_add_u32 " d ", " d ", 1, " b "\n\t"
#define I_MED3_F32(d, b, c) "v_med3_f32 " d ", " d ", " b ", " c "\n\t"
#define I_PK_ADD_U16(d, b, c) "v_pk_add_u16 " d ", " d ", " b "\n\t"
#define I_PK_SUB_U16_CLAMP(d, b, c) "v_pk_sub_u16 " d ", " d ", " b " clamp\n\t"
#define I_PK_MAX_U16(d, b, c) "v_pk_max_u16 " d ", " d ", " b "\n\t"
#define I_PK_MUL_F16(d, b, c) "v_pk_mul_f16 " d ", " d ", " b "\n\t"
#define I_PK_MAX_SCALAR(d, b, c) "v_pk_max_f16 " d ", " d ", s20 op_sel_hi:[1,0]\n\t"
#define I_ADD_U32_SCALAR(d, b, c) "v_add_u32 " d ", s20, " d "\n\t"
#define I_ADD_U32_E64(d, b, c) "v_add_u32_e64 " d ", " d ", " b "\n\t"
#define I_ADD_CO_U32(d, b, c) "v_add_co_u32 " d ", vcc, " d ", " b "\n\t"
#define I_MBCNT(d, b, c) "v_mbcnt_lo_u32_b32 " d ", " b ", " d "\n\t"
#define I_SAD_U8(d, b, c) "v_sad_u8 " d ", " d ", " b ", " c "\n\t"
#define I_DOT2_F16(d, b, c) "v_dot2_f32_f16 " d ", " b ", " c ", " d "\n\t"
// mixes: the proposed cell update = integer add on the packed pair (both 16-bit fields at once) + packed max3
#define I_MIX_ADDU32_MAX3(d, b, c) "v_add_u32 " d ", " d ", " b "\n\tv_pk_maximum3_f16 " d ", " d ", " b ", " c "\n\t"
#define I_MIX_ADDU32_PKMAX(d, b, c) "v_add_u32 " d ", " d ", " b "\n\tv_pk_max_f16 " d ", " d ", " c "\n\t"
#define I_MIX_ADDF32_MAXI32(d, b, c) "v_add_f32 " d ", " d ", " b "\n\tv_max_i32 " d ", " d ", " c "\n\t"
#define I_MIX_3(d, b, c) "v_pk_max_f16 " d ", " d ", " b "\n\tv_add_u32 " d ", " d ", " c "\n\tv_pk_maximum3_f16 " d ", " d ", " b ", " c "\n\t"
#define I_MIX_MOV_MAX3(d, b, c) "v_mov_b32 " d ", " c "\n\tv_pk_maximum3_f16 " d ", " d ", " b ", " c "\n\t"
#define I_MIX_ADD2_MAX3(d, b, c) "v_add_u32 " d ", " d ", " b "\n\tv_add_u32 " d ", " d ", " b "\n\tv_pk_maximum3_f16 " d ", " d ", " b ", " c "\n\t"

KERNEL_INDEP(k_pk_max_f16, I_PK_MAX_F16)
KERNEL_INDEP(k_max_f32, I_MAX_F32)
KERNEL_INDEP(k_mul_f32, I_MUL_F32)
KERNEL_INDEP(k_add_f16, I_ADD_F16)
KERNEL_INDEP(k_max3_f16, I_MAX3_F16)
KERNEL_INDEP(k_max3_f32, I_MAX3_F32)
KERNEL_INDEP(k_maximum3_f32, I_MAXIMUM3_F32)
KERNEL_INDEP(k_max_u16, I_MAX_U16)
KERNEL_INDEP(k_max_i16, I_MAX_I16)
KERNEL_INDEP(k_max_u32, I_MAX_U32)
KERNEL_INDEP(k_min_u32, I_MIN_U32)
KERNEL_INDEP(k_sub_u32, I_SUB_U32)
KERNEL_INDEP(k_and_b32, I_AND_B32)
KERNEL_INDEP(k_or_b32, I_OR_B32)
KERNEL_INDEP(k_xor_b32, I_XOR_B32)
KERNEL_INDEP(k_lshlrev, I_LSHLREV)
KERNEL_INDEP(k_lshrrev, I_LSHRREV)
KERNEL_INDEP(k_ashrrev, I_ASHRREV)
KERNEL_INDEP(k_cndmask, I_CNDMASK)
KERNEL_INDEP(k_add3_u32, I_ADD3_U32)
KERNEL_INDEP(k_add_lshl, I_ADD_LSHL)
KERNEL_INDEP(k_lshl_add, I_LSHL_ADD)
KERNEL_INDEP(k_med3_f32, I_MED3_F32)
KERNEL_INDEP(k_pk_add_u16, I_PK_ADD_U16)
KERNEL_INDEP(k_pk_sub_u16_clamp, I_PK_SUB_U16_CLAMP)
KERNEL_INDEP(k_pk_max_u16, I_PK_MAX_U16)
KERNEL_INDEP(k_pk_mul_f16, I_PK_MUL_F16)
KERNEL_INDEP(k_pk_max_scalar, I_PK_MAX_SCALAR)
KERNEL_INDEP(k_add_u32_scalar, I_ADD_U32_SCALAR)
KERNEL_INDEP(k_add_u32_e64, I_ADD_U32_E64)
KERNEL_INDEP(k_add_co_u32, I_ADD_CO_U32)
KERNEL_INDEP(k_mbcnt, I_MBCNT)
KERNEL_INDEP(k_sad_u8, I_SAD_U8)
KERNEL_INDEP(k_dot2_f16, I_DOT2_F16)
KERNEL_INDEP(k_mix_addu32_max3, I_MIX_ADDU32_MAX3)
KERNEL_INDEP(k_mix_addu32_pkmax, I_MIX_ADDU32_PKMAX)
KERNEL_INDEP(k_mix_addf32_maxi32, I_MIX_ADDF32_MAXI32)
KERNEL_INDEP(k_mix_3, I_MIX_3)
KERNEL_INDEP(k_mix_mov_max3, I_MIX_MOV_MAX3)
KERNEL_INDEP(k_mix_add2_max3, I_MIX_ADD2_MAX3)

KERNEL_INDEP(k_pk_add_f16, I_PK_ADD_F16)
KERNEL_INDEP(k_pk_maximum3_f16, I_PK_MAXIMUM3)
KERNEL_INDEP(k_pk_fma_f16, I_PK_FMA_F16)
KERNEL_INDEP(k_max_f16, I_MAX_F16)
KERNEL_INDEP(k_add_f32, I_ADD_F32)
KERNEL_INDEP(k_fma_f32, I_FMA_F32)
KERNEL_INDEP(k_add_u32, I_ADD_U32)
KERNEL_INDEP(k_mov_b32, I_MOV_B32)
KERNEL_INDEP(k_max_i32, I_MAX_I32)
KERNEL_INDEP(k_max3_i32, I_MAX3_I32)
KERNEL_INDEP(k_pk_max_i16, I_PK_MAX_I16)
KERNEL_INDEP(k_pk_add_i16, I_PK_ADD_I16)
KERNEL_INDEP(k_max_f16_dpp_row, I_MAX_F16_DPP)
KERNEL_INDEP(k_mov_dpp_wave_shr, I_MOV_DPP_WSHR)
KERNEL_INDEP(k_mov_dpp_row_bcast, I_MOV_DPP_BCAST)
KERNEL_INDEP(k_max_f16_sdwa, I_MAX_F16_SDWA)
KERNEL_INDEP(k_bitop3, I_BITOP3)
KERNEL_INDEP(k_bfi, I_BFI)
KERNEL_INDEP(k_perm, I_PERM)
KERNEL_INDEP(k_and_or, I_AND_OR)
KERNEL_INDEP(k_alignbit, I_ALIGNBIT)
KERNEL_INDEP(k_cvt_bf8, I_CVT_BF8)
KERNEL_INDEP(k_cvt_f16_f32, I_CVT_F16_F32)
KERNEL_INDEP(k_lshl_or, I_LSHL_OR)
KERNEL_INDEP(k_cell3, I_CELL3)
KERNEL_INDEP(k_valu_salu, I_VALU_SALU)
KERNEL_INDEP(k_valu_snop, I_VALU_SNOP)
KERNEL_INDEP(k_salu_only, I_SALU)
KERNEL_INDEP(k_snop_only, I_SNOP)
KERNEL_INDEP(k_cmp_eq_f16, I_CMP_EQ_F16)
KERNEL_INDEP(k_cmp_eq_f16_vcc, I_CMP_EQ_F16_VCC)
KERNEL_INDEP(k_readlane, I_READLANE)
KERNEL_INDEP(k_readfirstlane, I_READFIRSTLANE)
KERNEL_DEP(k_dep_pk_max_f16, I_PK_MAX_F16)
KERNEL_DEP(k_dep_pk_maximum3, I_PK_MAXIMUM3)
KERNEL_DEP(k_dep_add_f32, I_ADD_F32)
KERNEL_DEP(k_dep_max_i32, I_MAX_I32)
KERNEL_DEP(k_dep_dpp, I_DEP_DPP)
KERNEL_INDEP64(k_lshl_add_u64, I_LSHL_ADD_U64)
KERNEL_INDEP64(k_add_f64, I_ADD_F64)
KERNEL_INDEP64(k_pk_add_f32, I_PK_ADD_F32)

typedef void (*kern_t)(unsigned*, Stamp*, int, unsigned);

static void run(const char* name, kern_t k, int inst_per_trip, int W, FILE* csv) {
    const int blocks = 256 * W;   // 256-thread blocks: the dispatcher puts the 4 waves on the 4 SIMDs of a CU
    const int waves = blocks * 4;
    const int iters = 400;
    unsigned* d;
    Stamp* ds;
    hipMalloc(&d, 4096);
    hipMalloc(&ds, sizeof(Stamp) * waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, ds, 20, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, ds, iters, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(waves);
    hipMemcpy(st.data(), ds, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
    const double n_inst = (double)iters * inst_per_trip;
    // group by SIMD: xcc, se (15:13), sh(12), cu (11:8), simd (5:4)
    struct G { unsigned long long t0 = ~0ull, t1 = 0; int n = 0; double sumdur = 0; };
    std::map<unsigned, G> groups;
    for (const Stamp& s : st) {
        const unsigned key = ((s.xcc & 0xf) << 16) | (s.hwid & 0xff30u);
        G& g = groups[key];
        g.t0 = std::min(g.t0, s.t0);
        g.t1 = std::max(g.t1, s.t1);
        g.n++;
        g.sumdur += (double)(s.t1 - s.t0);
    }
    std::vector<double> cyc, wavecyc;
    int exact = 0;
    for (auto& kv : groups) {
        const G& g = kv.second;
        if (g.n != W) continue;
        ++exact;
        cyc.push_back((double)(g.t1 - g.t0) / (n_inst * W));
        wavecyc.push_back(g.sumdur / g.n / n_inst);
    }
    double med = 0, medw = 0;
    if (!cyc.empty()) {
        std::sort(cyc.begin(), cyc.end());
        std::sort(wavecyc.begin(), wavecyc.end());
        med = cyc[cyc.size() / 2];
        medw = wavecyc[wavecyc.size() / 2];
    }
    // wall clock: all SIMDs hold W waves -> ns per wave-instruction per SIMD
    const double ns = ms * 1e6 / (n_inst * W);
    printf("%-26s W=%d  simds=%4zu (with exactly W: %4d)  cyc/inst/SIMD=%6.2f  cyc/inst/wave=%6.2f  wall %.3f ms = %.3f ns/inst/SIMD (%.2f GHz implied)\n",
           name, W, groups.size(), exact, med, medw, ms, ns, med > 0 ? med / ns : 0.0);
    if (csv) fprintf(csv, "%s,%d,%zu,%d,%.3f,%.3f,%.4f,%.4f\n", name, W, groups.size(), exact, med, medw, ms, ns);
    hipFree(d);
    hipFree(ds);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    FILE* csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) fprintf(csv, "kernel,waves_per_simd,simds_seen,simds_with_W,cyc_per_inst_per_simd,cyc_per_inst_per_wave,wall_ms,wall_ns_per_inst_per_simd\n");
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    struct E { const char* n; kern_t k; int per; };
    const E list[] = {
        {"v_pk_max_f16", k_pk_max_f16, 128}, {"v_pk_add_f16", k_pk_add_f16, 128},
        {"v_pk_maximum3_f16", k_pk_maximum3_f16, 128}, {"v_pk_fma_f16", k_pk_fma_f16, 128},
        {"v_max_f16", k_max_f16, 128}, {"v_add_f32", k_add_f32, 128}, {"v_fma_f32", k_fma_f32, 128},
        {"v_add_u32", k_add_u32, 128}, {"v_mov_b32", k_mov_b32, 128}, {"v_max_i32", k_max_i32, 128},
        {"v_max3_i32", k_max3_i32, 128}, {"v_pk_max_i16", k_pk_max_i16, 128},
        {"v_pk_add_i16 clamp", k_pk_add_i16, 128}, {"v_max_f16_dpp row_shr:1", k_max_f16_dpp_row, 128},
        {"v_mov_b32_dpp wave_shr:1", k_mov_dpp_wave_shr, 128}, {"v_mov_b32_dpp row_bcast31", k_mov_dpp_row_bcast, 128},
        {"v_max_f16_sdwa", k_max_f16_sdwa, 128}, {"v_bitop3_b32", k_bitop3, 128}, {"v_bfi_b32", k_bfi, 128},
        {"v_perm_b32", k_perm, 128}, {"v_and_or_b32", k_and_or, 128}, {"v_alignbit_b32", k_alignbit, 128},
        {"v_cvt_scalef32_pk_f16_bf8", k_cvt_bf8, 128}, {"v_cvt_f16_f32", k_cvt_f16_f32, 128},
        {"v_lshl_or_b32", k_lshl_or, 128}, {"cell: max/add/maximum3", k_cell3, 384},
        {"v_pk_max_f16 + s_add_u32", k_valu_salu, 128}, {"v_pk_max_f16 + s_nop 0", k_valu_snop, 128},
        {"s_add_u32 only", k_salu_only, 128}, {"s_nop 0 only", k_snop_only, 128},
        {"v_max_f32", k_max_f32, 128}, {"v_mul_f32", k_mul_f32, 128}, {"v_add_f16", k_add_f16, 128},
        {"v_max3_f16", k_max3_f16, 128}, {"v_max3_f32", k_max3_f32, 128}, {"v_maximum3_f32", k_maximum3_f32, 128},
        {"v_max_u16", k_max_u16, 128}, {"v_max_i16", k_max_i16, 128}, {"v_max_u32", k_max_u32, 128},
        {"v_min_u32", k_min_u32, 128}, {"v_sub_u32", k_sub_u32, 128}, {"v_and_b32", k_and_b32, 128},
        {"v_or_b32", k_or_b32, 128}, {"v_xor_b32", k_xor_b32, 128}, {"v_lshlrev_b32", k_lshlrev, 128},
        {"v_lshrrev_b32", k_lshrrev, 128}, {"v_ashrrev_i32", k_ashrrev, 128}, {"v_cndmask_b32", k_cndmask, 128},
        {"v_add3_u32", k_add3_u32, 128}, {"v_add_lshl_u32", k_add_lshl, 128}, {"v_lshl_add_u32", k_lshl_add, 128},
        {"v_med3_f32", k_med3_f32, 128}, {"v_pk_add_u16", k_pk_add_u16, 128},
        {"v_pk_sub_u16 clamp", k_pk_sub_u16_clamp, 128}, {"v_pk_max_u16", k_pk_max_u16, 128},
        {"v_pk_mul_f16", k_pk_mul_f16, 128}, {"v_pk_max_f16 sgpr operand", k_pk_max_scalar, 128},
        {"v_add_u32 sgpr operand", k_add_u32_scalar, 128}, {"v_add_u32_e64", k_add_u32_e64, 128},
        {"v_add_co_u32", k_add_co_u32, 128}, {"v_mbcnt_lo", k_mbcnt, 128}, {"v_sad_u8", k_sad_u8, 128},
        {"v_dot2_f32_f16", k_dot2_f16, 128},
        {"mix: add_u32 + pk_maximum3", k_mix_addu32_max3, 256}, {"mix: add_u32 + pk_max_f16", k_mix_addu32_pkmax, 256},
        {"mix: add_f32 + max_i32", k_mix_addf32_maxi32, 256}, {"mix: pk_max + add_u32 + pk_max3", k_mix_3, 384},
        {"mix: mov + pk_maximum3", k_mix_mov_max3, 256}, {"mix: 2 add_u32 + pk_maximum3", k_mix_add2_max3, 384},
        {"v_lshl_add_u64", k_lshl_add_u64, 128}, {"v_add_f64", k_add_f64, 128}, {"v_pk_add_f32", k_pk_add_f32, 128},
        {"v_cmp_eq_f16 (sgpr dst)", k_cmp_eq_f16, 128}, {"v_cmp_eq_f16 (vcc)", k_cmp_eq_f16_vcc, 128},
        {"v_readlane_b32", k_readlane, 128}, {"v_readfirstlane_b32", k_readfirstlane, 128},
        {"dep v_pk_max_f16", k_dep_pk_max_f16, 128}, {"dep v_pk_maximum3_f16", k_dep_pk_maximum3, 128},
        {"dep v_add_f32", k_dep_add_f32, 128}, {"dep v_max_i32", k_dep_max_i32, 128},
        {"dep s_nop1+v_max_f16_dpp", k_dep_dpp, 128},
    };
    for (int W : {1, 2, 4, 8})
        for (const E& e : list) run(e.n, e.k, e.per, W, csv);
    if (csv) fclose(csv);
    return 0;
}
