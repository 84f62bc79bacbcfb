// Micro-benchmark: issue rate of scalar vs packed FP32 VALU ops on gfx950 (wave64), 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f2 ss = {s, s};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) { a0 = a0 * s; a1 = a1 * s; a2 = a2 * s; a3 = a3 * s; a4 = a4 * s; a5 = a5 * s; a6 = a6 * s; a7 = a7 * s; }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(ss));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(ss));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2) : "v"(ss));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p3) : "v"(ss));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<2048, 256>>>(d, iters, 1.0000001f); else k<1><<<2048, 256>>>(d, iters, 1.0000001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions: mode0 64 v_mul per iter, mode1 32 v_pk_mul per iter (same flops)
            double winst = (double)2048 * 4 * iters * (mode == 0 ? 64 : 32);
            printf("mode %d (%s): %.3f ms, %.2f Gwave-instr/s, per SIMD-cycle at 2.4GHz x1024 SIMDs: %.3f instr/cycle -> %.2f cycles/instr\n", mode,
                   mode == 0 ? "v_mul_f32" : "v_pk_mul_f32", ms, winst / ms / 1e6, winst / (ms * 1e-3) / (2.4e9 * 1024), (2.4e9 * 1024) / (winst / (ms * 1e-3)));
        }
    }
    return 0;
}
