// extend_budget.hip -- disassembly-derived instruction budget of wf_extend's building blocks: each kernel below holds
// exactly one block of the traversal kernel between a load of its inputs and a store of its results, compiled with the product's flags;
// scripts/microbench/extend_budget.py counts its VALU instructions by issue class (profiles/r03_valu_issue_microbench.txt). Not run: only compiled.
//   hipcc <product flags> --offload-arch=gfx950 -I../../hobbyrenderer_amd/csrc -S --cuda-device-only extend_budget.hip -o extend_budget.s
#include <hip/hip_runtime.h>
#include "pt_device.h"
using namespace hrt;
namespace {
constexpr uint32_t kBlk = 256;
struct Stack {          // the LDS stack of pt_wavefront.hip (16 rows, no spill)
    int32_t* base;
    HRT_DEV void push(int sp, int32_t v) { base[(sp & 15) * kBlk] = v; }
    HRT_DEV int32_t pop(int sp) { return base[(sp & 15) * kBlk]; }
};
struct LdsTree {        // the LDS copy of the 4-wide tree (pt_wavefront.hip LdsBvh<4>)
    static constexpr int kWidth = 4; static constexpr bool kTwoLevel = false; static constexpr bool kLds = true;
    const float4* nodes; const float4* tris;
    typedef __attribute__((address_space(3))) const char* LdsPtr;
    HRT_DEV uint32_t rowoff(int i, uint32_t byteOffset) const { return (uint32_t)(uintptr_t)(LdsPtr) reinterpret_cast<const char*>(nodes) + (uint32_t)i * 128u + byteOffset; }
    HRT_DEV float4 load(uint32_t off) const { return *reinterpret_cast<const float4*>((const char*)(LdsPtr)(uintptr_t)off); }
    HRT_DEV void tri(uint32_t i, float4& a, float4& b, float4& c) const { const float4* p = tris + 3 * i; a = p[0]; b = p[1]; c = p[2]; }
};
}
extern __shared__ __attribute__((aligned(128))) char smem[];

// one node step over an LDS tree: in = (cur, noi, inv, tmin, tlim, sp), out = (next, sp)
extern "C" __global__ __launch_bounds__(256) void budget_node_step_lds(const float4* in, int4* out)
{
    Stack st; st.base = reinterpret_cast<int32_t*>(smem) + threadIdx.x;
    LdsTree t; t.nodes = reinterpret_cast<const float4*>(smem + 16 * kBlk * 4); t.tris = t.nodes;
    const float4 a = in[threadIdx.x * 3], b = in[threadIdx.x * 3 + 1], c = in[threadIdx.x * 3 + 2];
    int sp = __float_as_int(c.z);
    const int32_t next = inner_step(t, __float_as_int(c.w), mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, b.w, st, sp);
    out[threadIdx.x] = make_int4(next, sp, 0, 0);
}
// the same step over fp32 nodes in global memory and over quantised nodes
extern "C" __global__ __launch_bounds__(256) void budget_node_step_global(const float4* in, int4* out, const GpuNode4* nodes)
{
    Stack st; st.base = reinterpret_cast<int32_t*>(smem) + threadIdx.x;
    GlobalBvh4 t; t.nodes = nodes; t.tris = nullptr;
    const float4 a = in[threadIdx.x * 3], b = in[threadIdx.x * 3 + 1], c = in[threadIdx.x * 3 + 2];
    int sp = __float_as_int(c.z);
    const int32_t next = inner_step(t, __float_as_int(c.w), mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, b.w, st, sp);
    out[threadIdx.x] = make_int4(next, sp, 0, 0);
}
extern "C" __global__ __launch_bounds__(256) void budget_node_step_quantised(const float4* in, int4* out, const GpuNodeQ* nodes)
{
    Stack st; st.base = reinterpret_cast<int32_t*>(smem) + threadIdx.x;
    GlobalBvhQ t; t.nodes = nodes; t.tris = nullptr;
    const float4 a = in[threadIdx.x * 3], b = in[threadIdx.x * 3 + 1], c = in[threadIdx.x * 3 + 2];
    int sp = __float_as_int(c.z);
    const int32_t next = inner_step(t, __float_as_int(c.w), mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, b.w, st, sp);
    out[threadIdx.x] = make_int4(next, sp, 0, 0);
}
// one watertight triangle test + the closest-hit update (selects), triangle from LDS
extern "C" __global__ __launch_bounds__(256) void budget_triangle_test(const float4* in, float4* out)
{
    const float4* tri = reinterpret_cast<const float4*>(smem);
    const float4 o = in[threadIdx.x * 4], d = in[threadIdx.x * 4 + 1], sh0 = in[threadIdx.x * 4 + 2], bst = in[threadIdx.x * 4 + 3];
    Ray r; r.o = mk3(o.x, o.y, o.z); r.d = mk3(d.x, d.y, d.z); r.tmin = o.w; r.tmax = d.w;
    RayShear s; s.kx = __float_as_int(sh0.x) & 3; s.ky = (__float_as_int(sh0.x) >> 2) & 3; s.kz = (__float_as_int(sh0.x) >> 4) & 3; s.Sx = sh0.y; s.Sy = sh0.z; s.Sz = sh0.w;
    const uint32_t first = __float_as_uint(bst.w);
    const float4 ta = tri[first * 3], tb = tri[first * 3 + 1], tc = tri[first * 3 + 2];
    float t, u, v;
    float bt = bst.x, bu = bst.y, bv = bst.z; uint32_t binst = 0, bprim = 0; bool valid = bst.x < 1e9f;
    if (tri_test(mk3(ta.x, ta.y, ta.z), mk3(tb.x, tb.y, tb.z), mk3(tc.x, tc.y, tc.z), r, s, t, u, v)) {
        const uint32_t inst = __float_as_uint(ta.w), prim = __float_as_uint(tb.w);
        const bool take = !valid || key_less(t, inst, prim, bt, binst, bprim);
        bt = take ? t : bt; bu = take ? u : bu; bv = take ? v : bv; binst = take ? inst : binst; bprim = take ? prim : bprim; valid = valid || take;
    }
    out[threadIdx.x] = make_float4(bt, bu, bv, __uint_as_float(binst + bprim + (valid ? 1u : 0u)));
}
// what a refilled lane computes before it can traverse: shear constants (three IEEE divisions), reciprocal direction, origin term
extern "C" __global__ __launch_bounds__(256) void budget_refill_setup(const float4* in, float4* out)
{
    const float4 o = in[threadIdx.x * 2], d = in[threadIdx.x * 2 + 1];
    const f3 dir = mk3(d.x, d.y, d.z);
    const RayShear s = make_shear(dir);
    const f3 inv = traversal_rcp(dir), noi = slab_origin_term(mk3(o.x, o.y, o.z), inv);
    out[threadIdx.x * 3] = make_float4(s.Sx, s.Sy, s.Sz, __int_as_float(s.kx | (s.ky << 2) | (s.kz << 4)));
    out[threadIdx.x * 3 + 1] = make_float4(inv.x, inv.y, inv.z, 0.0f);
    out[threadIdx.x * 3 + 2] = make_float4(noi.x, noi.y, noi.z, 0.0f);
}
