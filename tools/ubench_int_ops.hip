// instruction-rate microbenchmark for the integer ops the codec kernels lean on (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define N_IT 256
#define UNR 32
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 77u;
    unsigned b0 = seed ^ 0x01020304u; a[0] ^= b0;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            unsigned &x = a[u & 7]; const unsigned b = a[(u + 3) & 7];
            if (OP == 0) x = x + b;                                                                  // v_add_u32
            else if (OP == 1) x = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2, x) + __builtin_bit_cast(s16x2, b));   // v_pk_add_u16
            else if (OP == 2) x = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2, x) - __builtin_bit_cast(s16x2, b));   // v_pk_sub_i16
            else if (OP == 3) x = __builtin_amdgcn_perm(x, b, 0x0c050c01u);                           // v_perm_b32
            else if (OP == 4) x = __builtin_amdgcn_sad_u8(x, b, x);                                   // v_sad_u8
            else if (OP == 5) x = __builtin_amdgcn_alignbyte(x, b, 1);                                // v_alignbyte_b32
            else if (OP == 6) x = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true) + b; // dpp mov + add
            else if (OP == 7) x = (unsigned)((int)x * 58 + (int)b);                                   // v_mad (mul_lo?)
            else if (OP == 8) x = (unsigned)__builtin_amdgcn_ds_swizzle((int)x, 0x1F | (4 << 10));    // ds_swizzle
            else if (OP == 9) { s16x2 t = __builtin_bit_cast(s16x2, x); t = __builtin_elementwise_max(t, __builtin_bit_cast(s16x2, b)); x = __builtin_bit_cast(unsigned, t); }  // v_pk_max_i16
            else if (OP == 10) x = (unsigned)(__builtin_amdgcn_sdot4((int)x, (int)b, (int)x, false)); // v_dot4_i32_i8
            else if (OP == 11) x = (unsigned)((((int)x << 8) >> 8) * (int)(short)b + (int)x);
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, unsigned *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double winstr = (double)blocks * 4 * N_IT * UNR;              // wave-instructions
    double per_simd = winstr / (256.0 * 4);                        // per SIMD
    printf("%-14s %8.3f ms  -> %.2f clk per wave-instr per SIMD @2.4GHz (%.1f G wave-instr/s)\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd, winstr / ms / 1e6);
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_pk_add_u16", d); run<2>("v_pk_sub_i16", d); run<3>("v_perm_b32", d); run<4>("v_sad_u8", d);
    run<5>("v_alignbyte", d); run<6>("dpp+add", d); run<7>("mul+add", d); run<8>("ds_swizzle", d); run<9>("v_pk_max_i16", d);
    run<10>("v_dot4_i32_i8", d); run<11>("mad_i24ish", d);
    return 0;
}
