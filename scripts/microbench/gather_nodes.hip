// Gather microbenchmark for gfx950: what does one BVH "node step" cost the memory pipeline (TA / TCP / L2) when every lane of a wave fetches
// its own 128-byte node with R x global_load_dwordx4 (all R rows in the same 128-byte line), the next node index depending on the data?
// This is the access pattern of wf_extend on trees in global memory (config 4: 7 row loads per step, 41 of 64 lanes active, TA busy 66 % of
// the kernel, L2 read latency 214 cycles, TCC hit rate 0.79: gpurun_out/r03/pmcmem_config4). Questions the node-format design depends on:
//   * is the cost per wave-INSTRUCTION (then fewer, wider loads / smaller nodes pay) or per active lane (then lane utilisation pays)?
//   * how does it change with the table size (L1-, L2-, MALL-resident)?
//   * how much do dwordx2 / dword loads cost against dwordx4?
// Persistent waves (6 per SIMD, 256-thread blocks), a dependent random walk per lane; reports node steps per microsecond per CU and
// cycles of CU time per wave-step. Build: hipcc --offload-arch=gfx950 -O3 -o gather_nodes gather_nodes.hip ; run: ./gather_nodes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// Tree-like access skew: every step picks a level 0..8 uniformly and a node of that level uniformly (level l has 4^l nodes, breadth-first
// numbering): the top levels are hot (L1 / L2 hits) like the top of a BVH; 87 381 nodes = 10.7 MB in all. nodeMask = 0xFFFFFFFF selects it.
__device__ __forceinline__ uint32_t pick(uint32_t r, uint32_t nodeMask)
{
    if (nodeMask != 0xFFFFFFFFu) return r & nodeMask;
    const uint32_t l = (r >> 24) % 9u, n = 1u << (2u * l);
    return (n - 1u) / 3u + ((r >> 3) & (n - 1u));
}
// ROWS: 16-byte rows fetched per step (1..8); WIDTH: bytes per load instruction (16, 8, 4); LANEMASK: which lanes walk (others idle, exec-masked)
template <int ROWS, int WIDTH>
__global__ __launch_bounds__(256) void walk(const uint4* __restrict__ nodes, uint32_t nodeMask, int steps, unsigned long long laneMask, uint32_t* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    if ((laneMask >> lane) & 1ull) {
        for (int s = 0; s < steps; ++s) {
            const uint4* p = nodes + (size_t)pick(idx, nodeMask) * 8u;
            uint32_t v = 0;
            if (WIDTH == 16) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) { uint4 q = p[r]; v ^= q.x + q.y + q.z + q.w; }
            } else if (WIDTH == 8) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) { uint2 q = reinterpret_cast<const uint2*>(p)[r * 2]; v ^= q.x + q.y; }
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) { v ^= reinterpret_cast<const uint32_t*>(p)[r * 4]; }
            }
            acc += v;
            idx = idx * 1664525u + 1013904223u + v;           // depends on the data: the next fetch cannot start before this one returned
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

// COOPERATIVE fetch: the 8 lanes of a group fetch ONE node per instruction (lane j of the group loads 16-byte row j: 128 contiguous bytes
// per group), 8 rounds per step (round k fetches the node of group member k), staged through LDS (ds_write_b128 at the lane-linear slot,
// then every lane reads the 7 rows of ITS node back with ds_read_b128): 8 load instructions of 8 lines each per wave-step instead of 7
// instructions of 64 lines each.
template <int ROWS>
__global__ __launch_bounds__(256) void walk_coop(const uint4* __restrict__ nodes, uint32_t nodeMask, int steps, unsigned long long laneMask, uint32_t* out)
{
    __shared__ uint4 stage[4][8][64];                        // [wave][round][lane]: 32 KB per block
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, grp = lane >> 3, sub = lane & 7u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    const bool walking = (laneMask >> lane) & 1ull;
    for (int s = 0; s < steps; ++s) {
        const uint32_t mine = pick(idx, nodeMask);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t owner = grp * 8u + (uint32_t)k;
            const uint32_t node = (uint32_t)__shfl((int)mine, (int)owner, 64);
            const bool ownerWalks = (laneMask >> owner) & 1ull;
            if (ownerWalks) stage[wave][k][lane] = nodes[(size_t)node * 8u + sub];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint32_t v = 0;
        if (walking) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { uint4 q = stage[wave][sub][grp * 8u + (uint32_t)r]; v ^= q.x + q.y + q.z + q.w; }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        acc += v;
        idx = idx * 1664525u + 1013904223u + v;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

typedef void (*Kernel)(const uint4*, uint32_t, int, unsigned long long, uint32_t*);
struct Case { const char* name; Kernel k; int rows, width; };

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t maxNodes = 1u << 21;                       // 256 MB
    uint4* dNodes; CHECK(hipMalloc(&dNodes, maxNodes * 128));
    {
        std::vector<uint32_t> h(maxNodes * 32);
        uint32_t x = 12345u; for (auto& w : h) { x = x * 1664525u + 1013904223u; w = x >> 3; }
        CHECK(hipMemcpy(dNodes, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    const int blocksPerCu = 6, blocks = cus * blocksPerCu;
    uint32_t* dOut; CHECK(hipMalloc(&dOut, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const Case cases[] = {
        { "1 x dwordx4", walk<1, 16>, 1, 16 }, { "2 x dwordx4", walk<2, 16>, 2, 16 }, { "4 x dwordx4", walk<4, 16>, 4, 16 }, { "5 x dwordx4", walk<5, 16>, 5, 16 },
        { "7 x dwordx4", walk<7, 16>, 7, 16 }, { "8 x dwordx4", walk<8, 16>, 8, 16 }, { "7 x dwordx2", walk<7, 8>, 7, 8 }, { "7 x dword", walk<7, 4>, 7, 4 }, { "coop 8 x 16B", walk_coop<7>, 77, 16 },
    };
    struct Mask { const char* name; unsigned long long m; } masks[] = {
        { "64 lanes", ~0ull }, { "41 lanes (scattered)", 0xB6DB6DB6DB6DB6DBull & 0xFFFFFFFFFFFFFFFFull }, { "32 lanes (every other)", 0x5555555555555555ull },
        { "32 lanes (low half)", 0x00000000FFFFFFFFull }, { "16 lanes (one per quad)", 0x1111111111111111ull }, { "16 lanes (low quarter)", 0xFFFFull },
    };
    printf("device %s, %d CUs, %d blocks of 256 per CU (6 waves per SIMD); dependent random walk over 128-byte nodes\n", prop.gcnArchName, cus, blocksPerCu);
    printf("%-14s %-26s %10s | %12s %14s %16s %14s\n", "loads / step", "active lanes", "table", "ms", "Gsteps/s lane", "wave-steps/us/CU", "CU cyc / wave-step");
    for (size_t nodes : { (size_t)128, (size_t)(1u << 13), (size_t)(1u << 16), (size_t)0, (size_t)(1u << 21) })       // 16 KB (L1), 1 MB (L2), 8 MB (config 4's tree), 256 MB (beyond MALL)
        for (const Case& c : cases)
            for (const Mask& m : masks) {
                if (nodes != (1u << 16) && nodes != 0 && (m.m != ~0ull || (c.rows != 7 && c.rows != 4 && c.rows != 1 && c.rows != 77) || c.width != 16)) continue;
                if (nodes == 0 && ((c.rows != 7 && c.rows != 4 && c.rows != 5 && c.rows != 77) || c.width != 16 || (m.m != ~0ull && m.m != 0xB6DB6DB6DB6DB6DBull))) continue;      // the full matrix only at the 8 MB size
                int steps = 2000; float ms = 0.0f;
                for (int pass = 0; pass < 3; ++pass) {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, dNodes, (uint32_t)(nodes - 1), steps, m.m, dOut);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (pass == 0) steps = (int)(steps * 20.0f / (ms > 0.01f ? ms : 0.01f));
                    if (steps < 2000) steps = 2000;
                }
                const double lanes = (double)__builtin_popcountll(m.m);
                const double laneSteps = (double)blocks * 4.0 * lanes * steps, waveSteps = (double)blocks * 4.0 * steps;
                const double clock = 2.2e9;         // approximate clock under this load, for the cycles column only
                printf("%-14s %-26s %7.1f MB | %12.3f %14.2f %16.2f %14.1f\n", c.name, m.name, (nodes ? nodes : (size_t)87381) * 128.0 / 1048576.0 * (nodes ? 1.0 : -1.0), ms, laneSteps / (ms * 1e-3) / 1e9,
                       waveSteps / (ms * 1e3) / cus, ms * 1e-3 * clock / (waveSteps / cus));
                fflush(stdout);
            }
    return 0;
}
