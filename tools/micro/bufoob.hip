// Probe: do raw buffer loads / stores whose per-lane offset lies beyond the descriptor's num_records
// return zero / get dropped on this GPU, with and without a scalar offset?  The allocation is larger than
// every address the probe can form, so a "no" shows up as a sentinel value, never as a fault.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef int vi4 __attribute__((ext_vector_type(4)));
__device__ vf4 raw_buffer_load_f4(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void raw_buffer_store_f4(vf4 v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ inline vi4 buf_of(const void *p, unsigned nrec)
{
    union { struct { const void *p; unsigned n, f; } s; vi4 v; } r;
    r.s.p = p; r.s.n = nrec; r.s.f = 0x00020000u;
    return r.v;
}
__global__ void probe(float *base, float *res, unsigned soff)
{
    const vi4 b = buf_of(base, 0x7fffffffu);
    const int lane = threadIdx.x;
    // even lanes in range, odd lanes out of range
    const unsigned off = (lane & 1) ? 0x80000000u + 16u * lane : 16u * lane;
    const vf4 v = raw_buffer_load_f4(b, (int)off, (int)soff, 0);
    res[4 * lane + 0] = v.x; res[4 * lane + 1] = v.y; res[4 * lane + 2] = v.z; res[4 * lane + 3] = v.w;
    vf4 w = {-1.f, -2.f, -3.f, -4.f};
    raw_buffer_store_f4(w, b, (int)off, (int)soff, 2);
}
int main()
{
    const size_t bytes = 3ull << 30;   // 3 GiB: covers base + 0x80000000 + everything below
    float *d, *r;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&r, 64 * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const size_t n = bytes / 4;
    std::vector<float> h(1 << 20, 7.0f);
    for (size_t o = 0; o < bytes; o += h.size() * 4) hipMemcpy((char *)d + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (unsigned soff : {0u, 1024u, 0x30000000u}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, r, soff);
        hipDeviceSynchronize();
        float res[256];
        hipMemcpy(res, r, sizeof(res), hipMemcpyDeviceToHost);
        float in[4], oob[4];
        // what the stores did: in-range lanes wrote at soff + 16*lane; out-of-range lanes would have written at 2 GiB + ...
        hipMemcpy(in, (char *)d + soff + 16 * 2, 16, hipMemcpyDeviceToHost);
        hipMemcpy(oob, (char *)d + 0x80000000ull + soff + 16 * 3, 16, hipMemcpyDeviceToHost);
        printf("soff %#x: in-range lane 2 loaded %g (7 expected), out-of-range lane 3 loaded %g (0 expected); "
               "in-range store landed %g (-1 expected), out-of-range store target holds %g (7 expected)\n",
               soff, res[8], res[12], in[0], oob[0]);
        bad += !(res[8] == 7.f && res[12] == 0.f && in[0] == -1.f && oob[0] == 7.f);
        // restore
        for (int l = 0; l < 64; l += 2) hipMemcpy((char *)d + soff + 16 * l, h.data(), 16, hipMemcpyHostToDevice);
    }
    printf(bad ? "RANGE CHECK NOT AS ASSUMED\n" : "range check ok\n");
    (void)n;
    return bad;
}
