// Hardware probe: which register hazards around an inline-asm MFMA are real on gfx950?  (tools/isa_hazard.py encodes the answers.)
// hipcc pads nothing around an `asm` MFMA, so every sequence below is issued exactly as written — ONE asm statement on literally named
// registers (v152-v247, declared clobbered) — and its result is compared bit for bit with the same products issued by the compiler builtin.
// Per probe: the number of (lane, iteration) trials whose result differs.
//   WAR-AB/ind k   independent back-to-back MFMAs (the pipe backs up), the last one on A1, B1; k wait states; VALU overwrite of A1[0], B1[0]
//   WAR-AB/dep k   eight back-to-back MFMAs on ONE accumulator (each waits for its predecessor's D), the last on A1, B1; the same overwrite
//   RAW-AB k       VALU writes of A1[0:1], B1[0:1] (over junk); k wait states; the MFMA that reads them
//   RAW-C k        VALU write of C[0] (over junk); k wait states; the MFMA that accumulates into C
//   D-read k       an MFMA; k wait states; a VALU read of D[0] / of D's LAST register (the smallest clean k = the pad an asm MFMA needs in
//                  front of a reader: the result registers are written in order, the last one P + 3 states behind the issue)
//   RAW-AB v_mov_b32 / v_mov_b32_e64 / v_xor_b32 / v_cvt_pk_f16_f32 / + v_nop / + s_waitcnt: other writers and other fillers of the one state;
//   RAW-AB register i: the write goes to register i of the 4-register operand instead of register 0
// for the two shapes the library issues from asm: v_mfma_f32_16x16x32_f16 (4 passes) and v_mfma_f32_32x32x16_f16 (8 passes).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_hazards mfma_hazards.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ h8 mk(unsigned seed) {
    h8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (_Float16)((float)(hsh(seed * 8u + i) & 1023u) / 512.0f - 1.0f);
    return r;
}
struct Q { u64 lo, hi; };
__device__ __forceinline__ Q halves(h8 v) { return __builtin_bit_cast(Q, v); }

#define K0 ""
#define K1 "s_nop 0\n\t"
#define K2 "s_nop 1\n\t"
#define K3 "s_nop 2\n\t"
#define K4 "s_nop 3\n\t"
#define K6 "s_nop 5\n\t"
#define K5 "s_nop 4\n\t"
#define K7 "s_nop 6\n\t"
#define K8 "s_nop 7\n\t"
#define K9 "s_nop 7\n\ts_nop 0\n\t"
#define K11 "s_nop 7\n\ts_nop 2\n\t"
#define K10 "s_nop 7\n\ts_nop 1\n\t"
#define K12 "s_nop 7\n\ts_nop 3\n\t"
#define K14 "s_nop 7\n\ts_nop 5\n\t"
#define K16 "s_nop 7\n\ts_nop 7\n\t"
#define K20 "s_nop 7\n\ts_nop 7\n\ts_nop 3\n\t"
#define DRAIN "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
// A1 = v[200:203], B1 = v[204:207], A0 = v[208:211], B0 = v[212:215]; accumulators from v216 down / up (below)
#define LOAD_OPERANDS "v_mov_b64 v[200:201], %4\n\tv_mov_b64 v[202:203], %5\n\tv_mov_b64 v[204:205], %6\n\tv_mov_b64 v[206:207], %7\n\t" \
                      "v_mov_b64 v[208:209], %8\n\tv_mov_b64 v[210:211], %9\n\tv_mov_b64 v[212:213], %10\n\tv_mov_b64 v[214:215], %11\n\t"
#define ZERO_ACCS ".irp r,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183," \
                  "184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231," \
                  "232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
#define JUNK_A1B1 "v_mov_b32 v200, %12\n\tv_mov_b32 v201, %12\n\tv_mov_b32 v204, %12\n\tv_mov_b32 v205, %12\n\t"
#define OUT4 "v_mov_b32 %0, v216\n\tv_mov_b32 %1, v217\n\tv_mov_b32 %2, v218\n\tv_mov_b32 %3, v219\n\t"
#define CLOBBERS "v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171", \
    "v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193", \
    "v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215", \
    "v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237", \
    "v238","v239","v240","v241","v242","v243","v244","v245","v246","v247"
#define OPERANDS : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(qa1.lo), "v"(qa1.hi), "v"(qb1.lo), "v"(qb1.hi), "v"(qa0.lo), "v"(qa0.hi), "v"(qb0.lo), "v"(qb0.hi), "v"(junk), "v"(cval), "v"(a1lo32), "v"(b1lo32), "v"(cval64), "v"(fa0), "v"(fa1), "v"(fb0), "v"(fb1), "v"(a1d1), "v"(a1d2), "v"(a1d3), "v"(b1d1), "v"(b1d2), "v"(b1d3) : CLOBBERS

// the MFMA of shape SH on literal registers: d = c accumulator range, a / b operand ranges
#define MF16(acc, a, b) "v_mfma_f32_16x16x32_f16 " acc ", " a ", " b ", " acc "\n\t"
#define MF32(acc, a, b) "v_mfma_f32_32x32x16_f16 " acc ", " a ", " b ", " acc "\n\t"
#define A1 "v[200:203]"
#define B1 "v[204:207]"
#define A0 "v[208:211]"
#define B0 "v[212:215]"
// the LAST accumulator (compared): 16x16: v[216:219]; 32x32: v[216:231].  The independent ones in front of it:
#define IND16 MF16("v[220:223]", A0, B0) MF16("v[224:227]", A0, B0) MF16("v[228:231]", A0, B0) MF16("v[232:235]", A0, B0) MF16("v[236:239]", A0, B0) MF16("v[240:243]", A0, B0) MF16("v[244:247]", A0, B0)
#define IND32 MF32("v[232:247]", A0, B0) MF32("v[184:199]", A0, B0) MF32("v[168:183]", A0, B0) MF32("v[152:167]", A0, B0)
#define DEP16 MF16("v[216:219]", A0, B0) MF16("v[216:219]", A0, B0) MF16("v[216:219]", A0, B0) MF16("v[216:219]", A0, B0) MF16("v[216:219]", A0, B0) MF16("v[216:219]", A0, B0) MF16("v[216:219]", A0, B0)
#define DEP32 MF32("v[216:231]", A0, B0) MF32("v[216:231]", A0, B0) MF32("v[216:231]", A0, B0) MF32("v[216:231]", A0, B0) MF32("v[216:231]", A0, B0) MF32("v[216:231]", A0, B0) MF32("v[216:231]", A0, B0)
#define LAST16 MF16("v[216:219]", A1, B1)
#define LAST32 MF32("v[216:231]", A1, B1)
#define OVERWRITE "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t"

template <int SH> struct Shape;
template <> struct Shape<16> { typedef f4 acc_t; static constexpr int NR = 4; };
template <> struct Shape<32> { typedef f16v acc_t; static constexpr int NR = 16; };
template <int SH> __device__ __forceinline__ typename Shape<SH>::acc_t ref_mfma(h8 a, h8 b, typename Shape<SH>::acc_t c) {
    if constexpr (SH == 16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

constexpr int NPROBE = 62;

template <int SH>
__global__ __launch_bounds__(256) void probe(unsigned* cnt, int iters) {
    typedef typename Shape<SH>::acc_t acc_t;
    constexpr int NR = Shape<SH>::NR;
    const unsigned tid = blockIdx.x * 256u + threadIdx.x;
    unsigned bad[NPROBE];
#pragma unroll
    for (int i = 0; i < NPROBE; ++i) bad[i] = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned seed = tid * 977u + it * 131071u;
        const h8 a0 = mk(seed), b0 = mk(seed + 1), a1 = mk(seed + 2), b1 = mk(seed + 3);
        const Q qa0 = halves(a0), qb0 = halves(b0), qa1 = halves(a1), qb1 = halves(b1);
        const unsigned junk = 0x7bff7bffu;                          // 65504 in both halves: a stale or early read shows
        acc_t z;
#pragma unroll
        for (int r = 0; r < NR; ++r) z[r] = 0.0f;
        const acc_t ref_last = ref_mfma<SH>(a1, b1, z);             // a lone a1.b1 into a zero accumulator
        acc_t ref_dep = z;
#pragma unroll
        for (int i = 0; i < 7; ++i) ref_dep = ref_mfma<SH>(a0, b0, ref_dep);
        ref_dep = ref_mfma<SH>(a1, b1, ref_dep);
        acc_t c_in = z;
        c_in[0] = 3.25f;
        const acc_t ref_c = ref_mfma<SH>(a1, b1, c_in);             // RAW-C: C[0] = 3.25 written by a VALU right in front
        const float cval = 3.25f;
        const unsigned a1lo32 = (unsigned)qa1.lo, b1lo32 = (unsigned)qb1.lo;
        const float fa0 = (float)a1[0], fa1 = (float)a1[1], fb0 = (float)b1[0], fb1 = (float)b1[1];      // v_cvt_pk_f16_f32 of them = the operands' first dwords
        const unsigned a1d1 = (unsigned)(qa1.lo >> 32), a1d2 = (unsigned)qa1.hi, a1d3 = (unsigned)(qa1.hi >> 32);
        const unsigned b1d1 = (unsigned)(qb1.lo >> 32), b1d2 = (unsigned)qb1.hi, b1d3 = (unsigned)(qb1.hi >> 32);
        const u64 cval64 = (u64)__builtin_bit_cast(unsigned, cval);                // C[0] = 3.25, C[1] = 0 in one 64-bit move
        unsigned o0, o1, o2, o3;
        auto cmp4 = [&](const acc_t& r) {
            // (scalar copies first: __builtin_bit_cast on a vector ELEMENT expression reads element 0 under hipcc 7.2)
            const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
            return (o0 != __builtin_bit_cast(unsigned, r0) || o1 != __builtin_bit_cast(unsigned, r1) || o2 != __builtin_bit_cast(unsigned, r2) ||
                    o3 != __builtin_bit_cast(unsigned, r3)) ? 1u : 0u;
        };
        auto cmp_last = [&](const acc_t& r) {
            const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[NR - 1];
            return (o0 != __builtin_bit_cast(unsigned, r0) || o1 != __builtin_bit_cast(unsigned, r1) || o2 != __builtin_bit_cast(unsigned, r2) ||
                    o3 != __builtin_bit_cast(unsigned, r3)) ? 1u : 0u;
        };
#define RUN(SLOT, REF, BODY) { asm volatile(BODY OPERANDS); bad[SLOT] += (SLOT >= 40 && SLOT < 48) ? cmp_last(REF) : cmp4(REF); }
#define SEL(x16, x32) (SH == 16 ? x16 : x32)
        if constexpr (SH == 16) {
#define WAR_IND(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS DRAIN IND16 LAST16 K OVERWRITE DRAIN OUT4)
#define WAR_DEP(SLOT, K) RUN(SLOT, ref_dep, LOAD_OPERANDS ZERO_ACCS DRAIN DEP16 LAST16 K OVERWRITE DRAIN OUT4)
#define RAW_AB(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS JUNK_A1B1 DRAIN "v_mov_b64 v[200:201], %4\n\tv_mov_b64 v[204:205], %6\n\t" K LAST16 DRAIN OUT4)
#define RAW_C(SLOT, K) RUN(SLOT, ref_c, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v216, %12\n\t" DRAIN "v_mov_b32 v216, %13\n\t" K LAST16 DRAIN OUT4)
#define D_READ(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS DRAIN LAST16 K "v_mov_b32 %0, v216\n\t" DRAIN "v_mov_b32 %1, v217\n\tv_mov_b32 %2, v218\n\tv_mov_b32 %3, v219\n\t")
#define RAW_AB32(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_mov_b32 v200, %14\n\tv_mov_b32 v204, %15\n\t" K LAST16 DRAIN OUT4)
#define RAW_ABW(SLOT) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_mov_b32 v200, %14\n\tv_mov_b32 v204, %15\n\ts_waitcnt lgkmcnt(0)\n\t" LAST16 DRAIN OUT4)
#define RAW_CVT(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_cvt_pk_f16_f32 v200, %17, %18\n\tv_cvt_pk_f16_f32 v204, %19, %20\n\t" K LAST16 DRAIN OUT4)
#define RAW_E64(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_mov_b32_e64 v200, %14\n\tv_mov_b32_e64 v204, %15\n\t" K LAST16 DRAIN OUT4)
#define RAW_XOR(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_xor_b32 v200, 0, %14\n\tv_xor_b32 v204, 0, %15\n\t" K LAST16 DRAIN OUT4)
#define RAW_VNOP(SLOT) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_mov_b32 v200, %14\n\tv_mov_b32 v204, %15\n\tv_nop\n\t" LAST16 DRAIN OUT4)
#define RAW_REG(SLOT, RA, RB, IA, IB, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 " RA ", %12\n\tv_mov_b32 " RB ", %12\n\t" DRAIN "v_mov_b32 " RA ", " IA "\n\tv_mov_b32 " RB ", " IB "\n\t" K LAST16 DRAIN OUT4)
#define RAW_C64(SLOT, K) RUN(SLOT, ref_c, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v216, %12\n\tv_mov_b32 v217, %12\n\t" DRAIN "v_mov_b64 v[216:217], %16\n\t" K LAST16 DRAIN OUT4)
#define D_LAST(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS DRAIN LAST16 K "v_mov_b32 %3, v219\n\t" DRAIN "v_mov_b32 %0, v216\n\tv_mov_b32 %1, v217\n\tv_mov_b32 %2, v218\n\t")
            WAR_IND(0, K0) WAR_IND(1, K1) WAR_IND(2, K2) WAR_IND(3, K4) WAR_IND(4, K8)
            WAR_DEP(5, K0) WAR_DEP(6, K1) WAR_DEP(7, K2) WAR_DEP(8, K4) WAR_DEP(9, K8)
            RAW_AB(10, K0) RAW_AB(11, K1) RAW_AB(12, K2) RAW_AB(13, K4)
            RAW_C(14, K0) RAW_C(15, K1) RAW_C(16, K2) RAW_C(17, K4)
            D_READ(18, K0) D_READ(19, K1) D_READ(20, K2) D_READ(21, K3) D_READ(22, K4) D_READ(23, K6) D_READ(24, K8) D_READ(25, K10) D_READ(26, K12) D_READ(27, K14) D_READ(28, K16) D_READ(29, K20)
            RAW_AB32(30, K0) RAW_AB32(31, K1) RAW_AB32(32, K2) RAW_ABW(33) RAW_C64(34, K0) RAW_C64(35, K1) RAW_C64(36, K2) D_READ(37, K5) D_READ(38, K7) D_READ(39, K9)
            D_LAST(40, K4) D_LAST(41, K6) D_LAST(42, K7) D_LAST(43, K8) D_LAST(44, K9) D_LAST(45, K10) D_LAST(46, K11) D_LAST(47, K12)
            RAW_CVT(48, K0) RAW_CVT(49, K1) RAW_CVT(50, K2) RAW_E64(51, K0) RAW_E64(52, K1) RAW_XOR(53, K0) RAW_XOR(54, K1) RAW_VNOP(55)
            RAW_REG(56, "v201", "v205", "%21", "%24", K0) RAW_REG(57, "v201", "v205", "%21", "%24", K1) RAW_REG(58, "v202", "v206", "%22", "%25", K0) RAW_REG(59, "v202", "v206", "%22", "%25", K1)
            RAW_REG(60, "v203", "v207", "%23", "%26", K0) RAW_REG(61, "v203", "v207", "%23", "%26", K1)
#undef WAR_IND
#undef WAR_DEP
#undef RAW_AB
#undef RAW_C
#undef D_READ
#undef RAW_AB32
#undef RAW_ABW
#undef RAW_C64
#undef RAW_REG
#undef RAW_CVT
#undef RAW_E64
#undef RAW_XOR
#undef RAW_VNOP
#undef D_LAST
        } else {
#define WAR_IND(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS DRAIN IND32 LAST32 K OVERWRITE DRAIN OUT4)
#define WAR_DEP(SLOT, K) RUN(SLOT, ref_dep, LOAD_OPERANDS ZERO_ACCS DRAIN DEP32 LAST32 K OVERWRITE DRAIN OUT4)
#define RAW_AB(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS JUNK_A1B1 DRAIN "v_mov_b64 v[200:201], %4\n\tv_mov_b64 v[204:205], %6\n\t" K LAST32 DRAIN OUT4)
#define RAW_C(SLOT, K) RUN(SLOT, ref_c, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v216, %12\n\t" DRAIN "v_mov_b32 v216, %13\n\t" K LAST32 DRAIN OUT4)
#define D_READ(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS DRAIN LAST32 K "v_mov_b32 %0, v216\n\t" DRAIN "v_mov_b32 %1, v217\n\tv_mov_b32 %2, v218\n\tv_mov_b32 %3, v219\n\t")
#define RAW_AB32(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_mov_b32 v200, %14\n\tv_mov_b32 v204, %15\n\t" K LAST32 DRAIN OUT4)
#define RAW_ABW(SLOT) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_mov_b32 v200, %14\n\tv_mov_b32 v204, %15\n\ts_waitcnt lgkmcnt(0)\n\t" LAST32 DRAIN OUT4)
#define RAW_CVT(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_cvt_pk_f16_f32 v200, %17, %18\n\tv_cvt_pk_f16_f32 v204, %19, %20\n\t" K LAST32 DRAIN OUT4)
#define RAW_E64(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_mov_b32_e64 v200, %14\n\tv_mov_b32_e64 v204, %15\n\t" K LAST32 DRAIN OUT4)
#define RAW_XOR(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_xor_b32 v200, 0, %14\n\tv_xor_b32 v204, 0, %15\n\t" K LAST32 DRAIN OUT4)
#define RAW_VNOP(SLOT) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v200, %12\n\tv_mov_b32 v204, %12\n\t" DRAIN "v_mov_b32 v200, %14\n\tv_mov_b32 v204, %15\n\tv_nop\n\t" LAST32 DRAIN OUT4)
#define RAW_REG(SLOT, RA, RB, IA, IB, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 " RA ", %12\n\tv_mov_b32 " RB ", %12\n\t" DRAIN "v_mov_b32 " RA ", " IA "\n\tv_mov_b32 " RB ", " IB "\n\t" K LAST32 DRAIN OUT4)
#define RAW_C64(SLOT, K) RUN(SLOT, ref_c, LOAD_OPERANDS ZERO_ACCS "v_mov_b32 v216, %12\n\tv_mov_b32 v217, %12\n\t" DRAIN "v_mov_b64 v[216:217], %16\n\t" K LAST32 DRAIN OUT4)
#define D_LAST(SLOT, K) RUN(SLOT, ref_last, LOAD_OPERANDS ZERO_ACCS DRAIN LAST32 K "v_mov_b32 %3, v231\n\t" DRAIN "v_mov_b32 %0, v216\n\tv_mov_b32 %1, v217\n\tv_mov_b32 %2, v218\n\t")
            WAR_IND(0, K0) WAR_IND(1, K1) WAR_IND(2, K2) WAR_IND(3, K4) WAR_IND(4, K8)
            WAR_DEP(5, K0) WAR_DEP(6, K1) WAR_DEP(7, K2) WAR_DEP(8, K4) WAR_DEP(9, K8)
            RAW_AB(10, K0) RAW_AB(11, K1) RAW_AB(12, K2) RAW_AB(13, K4)
            RAW_C(14, K0) RAW_C(15, K1) RAW_C(16, K2) RAW_C(17, K4)
            D_READ(18, K0) D_READ(19, K1) D_READ(20, K2) D_READ(21, K3) D_READ(22, K4) D_READ(23, K6) D_READ(24, K8) D_READ(25, K10) D_READ(26, K12) D_READ(27, K14) D_READ(28, K16) D_READ(29, K20)
            RAW_AB32(30, K0) RAW_AB32(31, K1) RAW_AB32(32, K2) RAW_ABW(33) RAW_C64(34, K0) RAW_C64(35, K1) RAW_C64(36, K2) D_READ(37, K5) D_READ(38, K7) D_READ(39, K9)
            D_LAST(40, K4) D_LAST(41, K6) D_LAST(42, K7) D_LAST(43, K8) D_LAST(44, K9) D_LAST(45, K10) D_LAST(46, K11) D_LAST(47, K12)
            RAW_CVT(48, K0) RAW_CVT(49, K1) RAW_CVT(50, K2) RAW_E64(51, K0) RAW_E64(52, K1) RAW_XOR(53, K0) RAW_XOR(54, K1) RAW_VNOP(55)
            RAW_REG(56, "v201", "v205", "%21", "%24", K0) RAW_REG(57, "v201", "v205", "%21", "%24", K1) RAW_REG(58, "v202", "v206", "%22", "%25", K0) RAW_REG(59, "v202", "v206", "%22", "%25", K1)
            RAW_REG(60, "v203", "v207", "%23", "%26", K0) RAW_REG(61, "v203", "v207", "%23", "%26", K1)
        }
    }
#pragma unroll
    for (int i = 0; i < NPROBE; ++i) if (bad[i]) atomicAdd(&cnt[i], bad[i]);
}

int main() {
    unsigned* cnt;
    if (hipMalloc(&cnt, NPROBE * 4) != hipSuccess) { printf("no device\n"); return 1; }
    const char* names[62] = {"WAR-AB/ind k=0", "WAR-AB/ind k=1", "WAR-AB/ind k=2", "WAR-AB/ind k=4", "WAR-AB/ind k=8",
                             "WAR-AB/dep k=0", "WAR-AB/dep k=1", "WAR-AB/dep k=2", "WAR-AB/dep k=4", "WAR-AB/dep k=8",
                             "RAW-AB k=0", "RAW-AB k=1", "RAW-AB k=2", "RAW-AB k=4", "RAW-C k=0", "RAW-C k=1", "RAW-C k=2", "RAW-C k=4",
                             "D-read k=0", "D-read k=1", "D-read k=2", "D-read k=3", "D-read k=4", "D-read k=6", "D-read k=8", "D-read k=10", "D-read k=12", "D-read k=14",
                             "D-read k=16", "D-read k=20",
                             "RAW-AB v_mov_b32 k=0", "RAW-AB v_mov_b32 k=1", "RAW-AB v_mov_b32 k=2", "RAW-AB b32 + s_waitcnt", "RAW-C v_mov_b64 k=0", "RAW-C v_mov_b64 k=1", "RAW-C v_mov_b64 k=2",
                             "D-read k=5", "D-read k=7", "D-read k=9",
                             "D-read LAST reg k=4", "D-read LAST reg k=6", "D-read LAST reg k=7", "D-read LAST reg k=8", "D-read LAST reg k=9", "D-read LAST reg k=10", "D-read LAST reg k=11", "D-read LAST reg k=12",
                             "RAW-AB v_cvt_pk_f16_f32 k=0", "RAW-AB v_cvt_pk_f16_f32 k=1", "RAW-AB v_cvt_pk_f16_f32 k=2", "RAW-AB v_mov_b32_e64 k=0", "RAW-AB v_mov_b32_e64 k=1", "RAW-AB v_xor_b32 k=0", "RAW-AB v_xor_b32 k=1", "RAW-AB v_mov_b32 + v_nop",
                             "RAW-AB register 1 k=0", "RAW-AB register 1 k=1", "RAW-AB register 2 k=0", "RAW-AB register 2 k=1", "RAW-AB register 3 (last) k=0", "RAW-AB register 3 (last) k=1"};
    for (int sh = 0; sh < 2; ++sh)
        for (int wg = 256; wg <= 2048; wg *= 8) {
            (void)hipMemset(cnt, 0, NPROBE * 4);
            const int iters = 100;
            if (sh == 0) hipLaunchKernelGGL(probe<16>, dim3(wg), dim3(256), 0, 0, cnt, iters);
            else hipLaunchKernelGGL(probe<32>, dim3(wg), dim3(256), 0, 0, cnt, iters);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            unsigned h[NPROBE];
            (void)hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost);
            printf("%s, %d workgroups x 256 threads x %d iterations = %.0f lane-trials per probe\n",
                   sh == 0 ? "v_mfma_f32_16x16x32_f16 (4 passes)" : "v_mfma_f32_32x32x16_f16 (8 passes)", wg, iters, (double)wg * 256 * iters);
            for (int i = 0; i < 62; ++i) printf("   %-30s mismatching lane-trials: %u\n", names[i], h[i]);
        }
    return 0;
}
