// Step-0 roofline denominator: measured VALU issue rates of the integer/fp64 instructions a
// big-number kernel can be built from, on the actual MI355X.  Standalone diagnostic (not part of
// the product library).  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

#define REP8(X) X X X X X X X X
#define ITER_BODY 64   // instructions per loop iteration per chain-set

// Each kernel: `iters` iterations of a 64-instruction body.  Results are folded into out[] so the
// compiler keeps them; the asm volatile keeps every instruction.
template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(u32* out, int iters, u32 seed, u64* stamps) {
    extern __shared__ u32 lds_pad[];             // dynamic LDS only limits blocks per CU (forces an even spread)
    if (iters < 0) lds_pad[threadIdx.x] = seed;
    u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    u32 a = seed + threadIdx.x, b = seed * 3 + threadIdx.x * 7 + 1;
    u64 c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
    u32 d0 = a, d1 = b, d2 = a ^ b, d3 = a + b, d4 = 1, d5 = 2, d6 = 3, d7 = 4;
    double f0 = a, f1 = b, f2 = 1.5, f3 = 2.5, f4 = 3.5, f5 = 4.5, f6 = 5.5, f7 = 6.5;
    double fa = 1.0000001, fb = 0.5;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // v_mad_u64_u32, 8 independent accumulators
            REP8(asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                : "v"(a), "v"(b) : "vcc");)
        } else if constexpr (MODE == 1) {   // v_mad_u64_u32, one dependent chain
            REP8(asm volatile(
                "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0"
                : "+v"(c0) : "v"(a), "v"(b) : "vcc");)
        } else if constexpr (MODE == 2) {   // v_mul_lo_u32 independent
            REP8(asm volatile(
                "v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\t"
                "v_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\t"
                "v_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                : "v"(b));)
        } else if constexpr (MODE == 3) {   // v_mul_hi_u32 independent
            REP8(asm volatile(
                "v_mul_hi_u32 %0, %0, %8\n\tv_mul_hi_u32 %1, %1, %8\n\tv_mul_hi_u32 %2, %2, %8\n\t"
                "v_mul_hi_u32 %3, %3, %8\n\tv_mul_hi_u32 %4, %4, %8\n\tv_mul_hi_u32 %5, %5, %8\n\t"
                "v_mul_hi_u32 %6, %6, %8\n\tv_mul_hi_u32 %7, %7, %8"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                : "v"(b));)
        } else if constexpr (MODE == 4) {   // v_mad_u32_u24 independent
            REP8(asm volatile(
                "v_mad_u32_u24 %0, %8, %9, %0\n\tv_mad_u32_u24 %1, %8, %9, %1\n\tv_mad_u32_u24 %2, %8, %9, %2\n\t"
                "v_mad_u32_u24 %3, %8, %9, %3\n\tv_mad_u32_u24 %4, %8, %9, %4\n\tv_mad_u32_u24 %5, %8, %9, %5\n\t"
                "v_mad_u32_u24 %6, %8, %9, %6\n\tv_mad_u32_u24 %7, %8, %9, %7"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                : "v"(a), "v"(b));)
        } else if constexpr (MODE == 5) {   // v_mul_hi_u32_u24 independent
            REP8(asm volatile(
                "v_mul_hi_u32_u24 %0, %0, %8\n\tv_mul_hi_u32_u24 %1, %1, %8\n\tv_mul_hi_u32_u24 %2, %2, %8\n\t"
                "v_mul_hi_u32_u24 %3, %3, %8\n\tv_mul_hi_u32_u24 %4, %4, %8\n\tv_mul_hi_u32_u24 %5, %5, %8\n\t"
                "v_mul_hi_u32_u24 %6, %6, %8\n\tv_mul_hi_u32_u24 %7, %7, %8"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                : "v"(b));)
        } else if constexpr (MODE == 6) {   // v_add_co_u32 + v_addc_co_u32 pairs (carry chain)
            REP8(asm volatile(
                "v_add_co_u32 %0, vcc, %0, %8\n\tv_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
                "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\tv_addc_co_u32 %3, vcc, %3, %9, vcc\n\t"
                "v_addc_co_u32 %4, vcc, %4, %8, vcc\n\tv_addc_co_u32 %5, vcc, %5, %9, vcc\n\t"
                "v_addc_co_u32 %6, vcc, %6, %8, vcc\n\tv_addc_co_u32 %7, vcc, %7, %9, vcc"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                : "v"(a), "v"(b) : "vcc");)
        } else if constexpr (MODE == 7) {   // v_lshl_add_u64 independent
            REP8(asm volatile(
                "v_lshl_add_u64 %0, %0, 0, %8\n\tv_lshl_add_u64 %1, %1, 0, %8\n\tv_lshl_add_u64 %2, %2, 0, %8\n\t"
                "v_lshl_add_u64 %3, %3, 0, %8\n\tv_lshl_add_u64 %4, %4, 0, %8\n\tv_lshl_add_u64 %5, %5, 0, %8\n\t"
                "v_lshl_add_u64 %6, %6, 0, %8\n\tv_lshl_add_u64 %7, %7, 0, %8"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                : "v"((u64)a));)
        } else if constexpr (MODE == 8) {   // v_fma_f64 independent
            REP8(asm volatile(
                "v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\t"
                "v_fma_f64 %3, %8, %9, %3\n\tv_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\t"
                "v_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7"
                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                : "v"(fa), "v"(fb));)
        } else if constexpr (MODE == 9) {   // v_add3_u32 independent (full-rate calibration)
            REP8(asm volatile(
                "v_add3_u32 %0, %0, %8, %9\n\tv_add3_u32 %1, %1, %8, %9\n\tv_add3_u32 %2, %2, %8, %9\n\t"
                "v_add3_u32 %3, %3, %8, %9\n\tv_add3_u32 %4, %4, %8, %9\n\tv_add3_u32 %5, %5, %8, %9\n\t"
                "v_add3_u32 %6, %6, %8, %9\n\tv_add3_u32 %7, %7, %8, %9"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                : "v"(a), "v"(b));)
        } else if constexpr (MODE == 10) {  // mix: 4 x (v_mad_u64_u32 -> sgpr carry ; v_addc_co_u32 on a carry word)
            REP8(asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
                "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
                "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                : "v"(a), "v"(b) : "vcc");)
        } else if constexpr (MODE == 11) {  // v_mad_u64_u32 with an SGPR multiplicand (modulus limbs live in SGPRs)
            u32 sb = __builtin_amdgcn_readfirstlane(b);
            REP8(asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                : "v"(a), "s"(sb) : "vcc");)
        } else if constexpr (MODE == 12) {  // v_mul_f64 independent
            REP8(asm volatile(
                "v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\t"
                "v_mul_f64 %3, %3, %8\n\tv_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\t"
                "v_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8"
                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                : "v"(fa));)
        } else if constexpr (MODE == 13) {  // v_fma_f32 independent (full-rate fp calibration)
            float g0 = d0, g1 = d1, g2 = d2, g3 = d3, g4 = d4, g5 = d5, g6 = d6, g7 = d7, ga = 1.0001f, gb = 0.5f;
            REP8(asm volatile(
                "v_fma_f32 %0, %8, %9, %0\n\tv_fma_f32 %1, %8, %9, %1\n\tv_fma_f32 %2, %8, %9, %2\n\t"
                "v_fma_f32 %3, %8, %9, %3\n\tv_fma_f32 %4, %8, %9, %4\n\tv_fma_f32 %5, %8, %9, %5\n\t"
                "v_fma_f32 %6, %8, %9, %6\n\tv_fma_f32 %7, %8, %9, %7"
                : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7)
                : "v"(ga), "v"(gb));)
            d0 = g0; d1 = g1; d2 = g2; d3 = g3; d4 = g4; d5 = g5; d6 = g6; d7 = g7;
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (stamps && threadIdx.x == 0) {
        u32 hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[4 * blockIdx.x] = t1 - t0; stamps[4 * blockIdx.x + 1] = r1 - r0;
        stamps[4 * blockIdx.x + 2] = ((u64)(xcc & 0xf) << 32) | hwid; stamps[4 * blockIdx.x + 3] = t0;
    }
    u64 s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
    u32 r = (u32)s ^ (u32)(s >> 32) ^ d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7;
    double fs = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    r ^= (u32)__double_as_longlong(fs);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

struct Mode { int id; const char* name; };

template <int MODE>
static void run(const char* name, u32* dout, int waves_per_simd, FILE* csv) {
    int iters = 20000;
    static u64* dst = nullptr; if (!dst) CK(hipMalloc(&dst, 2048 * 4 * sizeof(u64)));
    int blocks = 256 * waves_per_simd;   // 256-thread block = 4 waves = 1 wave per SIMD of one CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t lds = (160 * 1024 / waves_per_simd) & ~(size_t)1023;
    CK(hipFuncSetAttribute((const void*)rate_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    rate_kernel<MODE><<<blocks, 256, lds>>>(dout, 50, 1u, nullptr);           // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        rate_kernel<MODE><<<blocks, 256, lds>>>(dout, iters, 1u + r, dst);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double wave_instr = (double)blocks * 4.0 * iters * 64.0;   // wave-instructions executed
    double lane_ops = wave_instr * 64.0;
    double per_s = lane_ops / (best * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    double simd_cycles = best * 1e-3 * 2.4e9;                  // cycles available per SIMD
    double instr_per_simd = (double)waves_per_simd * iters * 64.0;
    std::vector<u64> hs(4 * blocks); CK(hipMemcpy(hs.data(), dst, hs.size() * sizeof(u64), hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; std::vector<int> percu(8 * 4096, 0); int used = 0, maxper = 0;
    for (int b = 0; b < blocks; ++b) {
        cyc += hs[4 * b]; rt += hs[4 * b + 1];
        u32 hw = (u32)hs[4 * b + 2], xcc = (u32)(hs[4 * b + 2] >> 32);
        int key = xcc * 4096 + ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 0x1) + 32 * ((hw >> 13) & 0x7);   // cu_id, sh_id, se_id
        if (percu[key]++ == 0) ++used; if (percu[key] > maxper) maxper = percu[key];
    }
    cyc /= blocks; rt /= blocks;
    double clk_ghz = cyc / (rt * 10.0);                       // s_memrealtime ticks at 100 MHz
    double cyc_per_instr = cyc * 1.0 / instr_per_simd;         // shader cycles of a resident block / instrs issued on its SIMD
    printf("%-32s w/SIMD=%d %8.3f ms %7.3f Tlane-op/s  %5.2f cyc/instr/SIMD (in-kernel)  clk %.3f GHz (wall@2.4: %.2f) CUs used %d max blocks/CU %d\n",
           name, waves_per_simd, best, per_s / 1e12, cyc_per_instr, clk_ghz, simd_cycles / instr_per_simd, used, maxper);
    if (csv) fprintf(csv, "%s,%d,%.4f,%.4f,%.3f,%.3f\n", name, waves_per_simd, best, per_s / 1e12, cyc_per_instr, clk_ghz);
}

int main(int argc, char** argv) {
    u32* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(u32)));
    FILE* csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) fprintf(csv, "instr,waves_per_simd,ms,Tlaneops_per_s,shader_cycles_per_wave_instr_per_simd,clock_GHz\n");
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s  CUs=%d  clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2, 3, 4}) {
        run<0>("v_mad_u64_u32 indep x8", dout, w, csv);
        run<1>("v_mad_u64_u32 dependent", dout, w, csv);
        run<11>("v_mad_u64_u32 sgpr-operand", dout, w, csv);
        run<10>("v_mad_u64_u32 + v_addc_co pair", dout, w, csv);
        run<2>("v_mul_lo_u32", dout, w, csv);
        run<3>("v_mul_hi_u32", dout, w, csv);
        run<4>("v_mad_u32_u24", dout, w, csv);
        run<5>("v_mul_hi_u32_u24", dout, w, csv);
        run<6>("v_add_co/addc_co chain", dout, w, csv);
        run<7>("v_lshl_add_u64", dout, w, csv);
        run<9>("v_add3_u32", dout, w, csv);
        run<8>("v_fma_f64", dout, w, csv);
        run<12>("v_mul_f64", dout, w, csv);
        run<13>("v_fma_f32", dout, w, csv);
    }
    if (csv) fclose(csv);
    return 0;
}
