// gv_probe.hip — measurement aids that are not part of the visibility path (kept out of gv_cull.hip so that its source
// hash, which ties profiles/traffic.json to the cull kernels, does not move with them).
#include "gv_device.hpp"

namespace gv {

// Read-stream probe (gv_debug_stream_peak): the cull kernel's input streams of a flat, exactly paired pool — mesh.a 16 B,
// mesh.b 8 B, xf.ab 32 B, xf.c 8 B and the active bits (one 8-byte word per wave, as the kernel reads them; the 65 B per
// entry of the accounting count the flag byte they stand for) — with the same nontemporal loads, workgroup size and tile
// mapping, nothing computed and nothing written: what this box's HBM delivers to this access pattern (SURVEY.md §8d asks
// for the measured read-stream peak beside the vendor figure). A reference, not a bound: the cull kernel itself now and
// then comes out a few percent above it.
__global__ __launch_bounds__(kCullBlock) void stream_probe_kernel(const MeshMirror mesh, const TransformMirror xf, uint32_t nblocks,
                                                                 uint32_t xcd_run, float* __restrict__ sink)
{
    const uint32_t lb = tile_of_workgroup(blockIdx.x, xcd_run);
    if (lb >= nblocks)
        return;
    const uint32_t i = lb * kCullBlock + threadIdx.x;
    if (i >= mesh.count || i >= xf.count)
        return;
    const float4 ma = stream_load(&mesh.a[i]);
    const float2 mb = stream_load(&mesh.b[i]);
    const float4 xa = stream_load(&xf.ab[i].a);
    const float4 xb = stream_load(&xf.ab[i].b);
    const float2 xc = stream_load(&xf.c[i]);
    const uint32_t f = (uint32_t)((xf.active_bits[i >> 6] >> (i & 63u)) & 1ull);
    const float s = ma.x + ma.y + ma.z + ma.w + mb.x + mb.y + xa.x + xa.y + xa.z + xa.w + xb.x + xb.y + xb.z + xb.w + xc.x + xc.y + (float)f;
    if (s == 12345.678f)  // never true for real pools: keeps the loads alive
        sink[0] = s;
}

hipError_t launch_stream_probe(const MeshMirror& mesh, const TransformMirror& xf, float* sink, hipStream_t stream)
{
    const uint32_t n = std::min(mesh.count, xf.count);
    if (n == 0)
        return hipSuccess;
    const uint32_t nblocks = (n + kCullBlock - 1) / kCullBlock;
    const uint32_t run = xcd_run_for_tiles(nblocks);
    hipLaunchKernelGGL(stream_probe_kernel, dim3(grid_for_tiles(nblocks, run)), dim3(kCullBlock), 0, stream, mesh, xf, nblocks, run, sink);
    return hipGetLastError();
}

}  // namespace gv
