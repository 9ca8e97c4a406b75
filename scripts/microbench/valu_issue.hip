// VALU issue-rate microbenchmark for gfx950: how many cycles does one wave64 v_fma_f32 occupy a SIMD, with 1..8 waves per SIMD?
// Settles the "x4 or x2" question behind bench.py's valu block (round-1 VERDICT): SQ_ACTIVE_INST_VALU counts 4 cycles per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run: ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PACKED>
__global__ __launch_bounds__(64) void fma_chain(float* out, int iters, float a, float b)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = { x0, x1 }, p1 = { x2, x3 }, p2 = { x4, x5 }, p3 = { x6, x7 }, pa = { a, a }, pb = { b, b };
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (PACKED) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        } else {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        }
    }
    if (PACKED) { x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y; }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(t1 - t0) * 0.0f;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (size_t)gridDim.x * 64)[0] = t1 - t0;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    float* d; hipMalloc(&d, (size_t)cus * 4 * 16 * 64 * 4 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int packed = 0; packed < 2; ++packed)
        for (int wps : { 1, 2, 3, 4, 5, 8 }) {
            const int blocks = cus * 4 * wps;      // one 64-thread block = one wave; the dispatcher spreads them over the SIMDs
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (packed) hipLaunchKernelGGL(fma_chain<1>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(fma_chain<0>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long cyc; hipMemcpy(&cyc, d + (size_t)blocks * 64, 8, hipMemcpyDeviceToHost);
            const double instPerWave = (double)iters * (packed ? 4 : 8);
            // wall-time view: SIMD-cycles per wave-instruction = time * clock / (instructions issued per SIMD)
            const double simdCycles = ms * 1e-3 * p.clockRate * 1e3 / (instPerWave * wps);
            printf("%s waves/SIMD %d: %.3f ms; wave 0 saw %.2f shader-clock ticks per instruction; SIMD cycles per wave64 instruction (wall clock x %d kHz) %.2f\n",
                   packed ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, (double)cyc / instPerWave, p.clockRate, simdCycles);
        }
    return 0;
}
