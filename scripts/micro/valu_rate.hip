// valu_rate.hip -- issue rate of wave64 FP32 VALU instructions on gfx950, scalar vs packed, at 1 / 2 / 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE> __global__ void k(float* out, int iters, float s) {
    const unsigned long long mask = 0x5555555555555555ull + (unsigned long long)iters;
    float a[16];
    v2f p[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x + i; p[i] = (v2f){(float)threadIdx.x, (float)i}; }
    v2f ps = (v2f){s, s};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 2) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ps));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 3) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(ps));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 4) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ps));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 5) {   // dependent chain of scalar adds (one register)
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 6) {   // v_mul with an SGPR operand
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 7) {   // DPP add
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 8) {   // v_cndmask
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 9) {   // v_cndmask with an SGPR-pair mask (VOP3)
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "s"(mask));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 10) {  // v_mul_f32, VGPR operands only
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 11) {  // v_permlane32_swap
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 15]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 12) {  // v_permlane16_swap
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 15]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 13) {  // packed fma with op_sel / neg modifiers (complex multiply second half)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(p[i]) : "v"(ps));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 14) {  // v_mov_b32
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 15) {  // dpp add with a bank mask (partial write)
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 16) {  // v_sub_f32
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 17) {  // v_fmac_f32 (VOP2 fma)
#define X(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(s));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 18) {  // add with an inline constant
#define X(i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (MODE == 19) {  // mul by a literal
#define X(i) asm volatile("v_mul_f32 %0, 0x3f3504f3, %0" : "+v"(a[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        }
    }
    float r = 0;
    for (int i = 0; i < 16; i++) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, float* d) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int iters = 4000, grid = 256 * 4 * wps;     // wps waves per SIMD (one 64-thread block = one wave)
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = 64.0 * iters;
        // cycles per wave-instruction per SIMD at a nominal 2.4 GHz (the clock under this load is lower: compare rows)
        printf("%-28s %d waves/SIMD: %.3f ms  -> %.2f ns per instr per SIMD (= %.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / (instr_per_wave * wps), ms * 1e6 / (instr_per_wave * wps) * 2.4);
    }
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 4 * 64 * sizeof(float));
    run<0>("v_add_f32", d);
    run<1>("v_fma_f32", d);
    run<2>("v_pk_add_f32", d);
    run<3>("v_pk_fma_f32", d);
    run<4>("v_pk_mul_f32", d);
    run<5>("v_add_f32 dependent", d);
    run<6>("v_mul_f32 sgpr", d);
    run<7>("v_add_f32_dpp", d);
    run<8>("v_cndmask_b32 vcc", d);
    run<9>("v_cndmask_b32_e64 sgpr", d);
    run<10>("v_mul_f32 vgpr", d);
    run<11>("v_permlane32_swap", d);
    run<12>("v_permlane16_swap", d);
    run<13>("v_pk_fma_f32 opsel/neg", d);
    run<14>("v_mov_b32", d);
    run<15>("v_add_f32_dpp bank_mask", d);
    run<16>("v_sub_f32", d);
    run<17>("v_fmac_f32", d);
    run<18>("v_add_f32 inline const", d);
    run<19>("v_mul_f32 literal", d);
    return 0;
}
