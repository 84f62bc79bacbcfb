// Micro-benchmark: cost of per-lane gathers from a small (cache-resident) table on gfx950, the access pattern of BVH
// node fetches.  Reports cycles per wave-instruction per CU for global_load_dword/x2/x4 and ds_read_b128 with random
// or uniform lane addresses and with part of the lanes masked off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: 4 x dwordx4 from one 64-B record per lane (a node fetch)   MODE 1: 1 x dwordx4   MODE 2: 1 x dword
// MODE 3: 4 x ds_read_b128 from an LDS copy (64-B records)           MODE 4: 2 x dwordx4 (32-B record)
// MODE 7: 4 x dwordx4, the 4 lanes of a quad reading the 4 words of one 64-B record (coalesced 64-B requests)
// MODE 8: MODE 0 with a 32-bit byte offset from a wave-uniform base (global_load ... v_off, s[base])   MODE 9: MODE 0 as raw buffer loads
// MODE 10: MODE 0 with the first word's load completed (s_waitcnt) before the other three are issued (do in-flight misses to one line merge?)
// MODE 11: odd lanes 4 x ds_read_b128 from the LDS copy, even lanes 4 x dwordx4 from the table (do the two pipes overlap?)
// MODE 12: as 11 with three lanes of four on the LDS copy
// MODE 5: 8 x dwordx4 from one 128-B record per lane (a whole cache line: a 4-wide node)   MODE 6: 3 x dwordx4 at 48-B stride (a triangle record)
__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ table, uint32_t nrec, int iters, int activeLanes, int uniform, float* out) {
    __shared__ float4 lds[(MODE == 3 || MODE == 11 || MODE == 12) ? 2048 * 4 : 1];
    if (MODE == 3 || MODE == 11 || MODE == 12) { for (uint32_t i = threadIdx.x; i < 2048 * 4; i += 256) lds[i] = table[i % (nrec * 4)]; __syncthreads(); }
    uint32_t lane = threadIdx.x & 63;
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    if ((int)lane < activeLanes) {
        for (int i = 0; i < iters; i++) {
            x = x * 1664525u + 1013904223u;
            uint32_t r = uniform ? (uint32_t)(i * 7919) % nrec : (x >> 8) % nrec;
            if (MODE == 3) r &= 2047;
            float4 a, b, c, d;
            if (MODE == 0) { const float4* p = table + (size_t)r * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; acc += sum4(a) + sum4(b) + sum4(c) + sum4(d); }
            else if (MODE == 1) { a = table[(size_t)r * 4]; acc += sum4(a); }
            else if (MODE == 2) { acc += reinterpret_cast<const float*>(table)[(size_t)r * 16]; }
            else if (MODE == 3) { const float4* p = lds + r * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; acc += sum4(a) + sum4(b) + sum4(c) + sum4(d); }
            else if (MODE == 4) { const float4* p = table + (size_t)r * 4; a = p[0]; b = p[1]; acc += sum4(a) + sum4(b); }
            else if (MODE == 5) { const float4* p = table + (size_t)(r >> 1) * 8; a = p[0]; b = p[1]; c = p[2]; d = p[3]; float4 e = p[4], f = p[5], g = p[6], h = p[7]; acc += sum4(a) + sum4(b) + sum4(c) + sum4(d) + sum4(e) + sum4(f) + sum4(g) + sum4(h); }
            else if (MODE == 7) {   // quad-cooperative: the 4 lanes of a quad fetch the four 16-B words of ONE 64-B record per instruction
                uint32_t q = lane & 3u;
                uint32_t r0 = __shfl(r, (lane & ~3u) + 0, 64), r1 = __shfl(r, (lane & ~3u) + 1, 64), r2 = __shfl(r, (lane & ~3u) + 2, 64), r3 = __shfl(r, (lane & ~3u) + 3, 64);
                a = table[(size_t)r0 * 4 + q]; b = table[(size_t)r1 * 4 + q]; c = table[(size_t)r2 * 4 + q]; d = table[(size_t)r3 * 4 + q];
                acc += sum4(a) + sum4(b) + sum4(c) + sum4(d); }
            else if (MODE == 8) { const char* base = reinterpret_cast<const char*>(table); uint32_t off = r * 64u;
                a = *reinterpret_cast<const float4*>(base + off); b = *reinterpret_cast<const float4*>(base + off + 16);
                c = *reinterpret_cast<const float4*>(base + off + 32); d = *reinterpret_cast<const float4*>(base + off + 48); acc += sum4(a) + sum4(b) + sum4(c) + sum4(d); }
            else if (MODE == 9) { __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(table), 0, (int)(nrec * 64u), 0x00020000);
                uint32_t off = r * 64u;
                auto ld = [&](uint32_t o) { auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + o), 0, 0); float4 f; __builtin_memcpy(&f, &v, 16); return f; };
                a = ld(0); b = ld(16); c = ld(32); d = ld(48); acc += sum4(a) + sum4(b) + sum4(c) + sum4(d); }
            else if (MODE == 10) { const float4* p = table + (size_t)r * 4; a = p[0]; asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) :: "memory");
                b = p[1]; c = p[2]; d = p[3]; acc += sum4(a) + sum4(b) + sum4(c) + sum4(d); }
            else if (MODE == 11 || MODE == 12) {
                const bool inLds = MODE == 11 ? (lane & 1u) != 0u : (lane & 3u) != 0u;
                if (inLds) { const float4* p = lds + (r & 2047) * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
                else { const float4* p = table + (size_t)r * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
                acc += sum4(a) + sum4(b) + sum4(c) + sum4(d); }
            else if (MODE == 6) { const float4* p = table + (size_t)r * 3; a = p[0]; b = p[1]; c = p[2]; acc += sum4(a) + sum4(b) + sum4(c); }
            // dependent chain like a traversal: next index depends on the data
            x += (uint32_t)(acc != 12345.678f);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const float4* table, uint32_t nrec, float* out, int instrPerIter) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 8, iters = 2000;
    for (int uniform = 0; uniform < 2; uniform++)
        for (int lanes : {64, 32, 16, 4}) {
            k<MODE><<<blocks, 256>>>(table, nrec, 10, lanes, uniform, out);
            CHK(hipEventRecord(e0));
            k<MODE><<<blocks, 256>>>(table, nrec, iters, lanes, uniform, out);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            double waveIters = (double)blocks * 4 * iters;               // wave-level iterations
            double cyclesPerCU = ms * 1e-3 * 2.4e9;                       // assumed 2.4 GHz
            double perIter = cyclesPerCU / (waveIters / 256.0);          // CU-cycles per wave-iteration
            printf("%-28s %s lanes %2d: %8.3f ms  %7.1f CU-cycles per wave-iteration (%d vmem/lds instr) -> %6.1f per instr, %5.2f per active lane-instr\n",
                   name, uniform ? "uniform" : "random ", lanes, ms, perIter, instrPerIter, perIter / instrPerIter, perIter / instrPerIter / lanes);
        }
}

int main() {
    const uint32_t nrec = 1600;          // ~100 KB of 64-B records: scene-4-sized node array
    std::vector<float> h(nrec * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 97) * 0.5f;
    float4* table; float* out;
    CHK(hipMalloc(&table, nrec * 64)); CHK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHK(hipMemcpy(table, h.data(), nrec * 64, hipMemcpyHostToDevice));
    if (getenv("UBENCH_MIXED_ONLY")) {
        run<0>("4 x global_load_dwordx4", table, nrec, out, 4);
        run<3>("4 x ds_read_b128", table, nrec, out, 4);
        run<11>("mixed: 1/2 LDS, 1/2 global", table, nrec, out, 4);
        run<12>("mixed: 3/4 LDS, 1/4 global", table, nrec, out, 4);
        return 0;
    }
    run<0>("4 x global_load_dwordx4", table, nrec, out, 4);
    run<4>("2 x global_load_dwordx4", table, nrec, out, 2);
    run<1>("1 x global_load_dwordx4", table, nrec, out, 1);
    run<2>("1 x global_load_dword", table, nrec, out, 1);
    run<5>("8 x dwordx4 (128-B record)", table, nrec, out, 8);
    run<6>("3 x dwordx4 (48-B stride)", table, nrec, out, 3);
    run<7>("4 x dwordx4 quad-cooperative", table, nrec, out, 4);
    run<10>("1 + wait + 3 x dwordx4", table, nrec, out, 4);
    run<8>("4 x dwordx4 saddr + voffset", table, nrec, out, 4);
    run<9>("4 x buffer_load_dwordx4", table, nrec, out, 4);
    run<3>("4 x ds_read_b128", table, nrec, out, 4);
    return 0;
}
