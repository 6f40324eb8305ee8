// Measures how many cycles one SIMD of gfx950 needs per wave64 v_fma_f32 as a function of the
// number of resident waves per SIMD and of the number of independent dependency chains per
// wave.  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CHAINS>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, float a, float b) {
    float v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x * 0.001f + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64 / CHAINS; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    if (s == 12345.678f) out[0] = s;
}

// instruction classes the blend kernels are made of, 8 independent chains per wave
template <int KIND>
__global__ __launch_bounds__(256) void mix_kernel(float* out, int iters, float a, float b) {
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = threadIdx.x * 0.001f + c;
    const float sa = __builtin_amdgcn_readfirstlane(a), sb = __builtin_amdgcn_readfirstlane(b);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w2[4], k2 = {a, b};
#pragma unroll
    for (int c = 0; c < 4; ++c) w2[c] = f2{v[2 * c], v[2 * c + 1]};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "s"(sa), "s"(sa));
                if (KIND == 1) asm volatile("v_mul_f32 %0, 0x3f800347, %0" : "+v"(v[c]));
                if (KIND == 2) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[c]));
                if (KIND == 3) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(a) : "vcc");
                if (KIND == 4) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[c]),
                                                                    __builtin_bit_cast(unsigned, v[c]), false, false);
                    unsigned r0 = r[0];
                    asm volatile("" : "+v"(r0));
                    v[c] = __builtin_bit_cast(float, r0);
                }
                if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[c]));
                if (KIND == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(v[c]));
                if (KIND == 7 && c < 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w2[c]) : "v"(k2));
                if (KIND == 8 && c < 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w2[c]) : "v"(k2));
                if (KIND == 9 && c < 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w2[c]) : "v"(k2));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += v[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) s += w2[c].x + w2[c].y;
    if (s == 12345.678f) out[0] = s + sb;
}

template <int KIND>
static void run_mix(const char* name, int waves_per_simd, float* out, double ghz) {
    const int cus = 256, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mix_kernel<KIND><<<cus * waves_per_simd, 256>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    mix_kernel<KIND><<<cus * waves_per_simd, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)waves_per_simd * iters * 64 * (KIND == 3 ? 2.0 : KIND >= 7 ? 0.5 : 1.0);
    printf("%-28s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd,
           ms * 1e-3 * ghz * 1e9 / n);
}

template <int CHAINS>
static void run(int waves_per_simd, float* out, double ghz) {
    // one workgroup of 256 lanes = 4 waves = one wave per SIMD of a CU; w workgroups per CU
    const int cus = 256, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    fma_kernel<CHAINS><<<cus * waves_per_simd, 256>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    fma_kernel<CHAINS><<<cus * waves_per_simd, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)waves_per_simd * iters * 64;
    printf("chains=%2d waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n",
           CHAINS, waves_per_simd, ms, ms * 1e-3 * ghz * 1e9 / inst_per_simd, ghz);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    printf("%s, %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    float* out; hipMalloc(&out, 4);
    for (int w : {1, 2, 4, 5, 8}) run<1>(w, out, ghz);
    for (int w : {1, 2, 4, 5, 8}) run<4>(w, out, ghz);
    for (int w : {1, 2, 4, 5, 8}) run<16>(w, out, ghz);
    for (int w : {2, 4}) {
        run_mix<0>("v_fma_f32 two SGPR operands", w, out, ghz);
        run_mix<1>("v_mul_f32 literal", w, out, ghz);
        run_mix<2>("v_add_f32_dpp quad_perm", w, out, ghz);
        run_mix<3>("v_cmp + v_cndmask (vcc)", w, out, ghz);
        run_mix<4>("v_permlane32_swap", w, out, ghz);
        run_mix<5>("v_rcp_f32", w, out, ghz);
        run_mix<6>("v_exp_f32", w, out, ghz);
        run_mix<7>("v_pk_fma_f32", w, out, ghz);
        run_mix<8>("v_pk_mul_f32", w, out, ghz);
        run_mix<9>("v_pk_add_f32", w, out, ghz);
    }
    hipFree(out);
    return 0;
}
