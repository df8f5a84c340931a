// Micro-benchmark: what does one dependent "time step" over 8 x 92 400 floats cost at minimum?
// Chains of N kernels replayed from a HIP graph; variants differ in what each step does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}

// out = f(cur, old, m, eta): one float4 per thread, streaming
__global__ __launch_bounds__(256) void k_stream(const float4 *cur, const float4 *old, const float4 *m, const float4 *eta,
                                                float4 *out, int n4, int n4_model)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 c = cur[i], o = old[i], mm = m[i % n4_model], e = eta[i % n4_model];
    out[i] = make_float4(2 * c.x - o.x + mm.x * e.x, 2 * c.y - o.y + mm.y * e.y, 2 * c.z - o.z + mm.z * e.z,
                         2 * c.w - o.w + mm.w * e.w);
}

// same plus `extra` more float4 loads of cur per thread at scattered tile-like offsets (halo cost model)
template <int EXTRA>
__global__ __launch_bounds__(256) void k_halo(const float4 *cur, const float4 *old, const float4 *m, const float4 *eta,
                                              float4 *out, int n4, int n4_model)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 c = cur[i];
    const float4 o = old[i], mm = m[i % n4_model], e = eta[i % n4_model];
#pragma unroll
    for (int j = 1; j <= EXTRA; ++j) {
        const float4 h = cur[(i + j * 7919) % n4];
        c.x += 1e-9f * h.x, c.y += 1e-9f * h.y, c.z += 1e-9f * h.z, c.w += 1e-9f * h.w;
    }
    out[i] = make_float4(2 * c.x - o.x + mm.x * e.x, 2 * c.y - o.y + mm.y * e.y, 2 * c.z - o.z + mm.z * e.z,
                         2 * c.w - o.w + mm.w * e.w);
}

int main()
{
    const int shots = 8, field = 420 * 220 + 4096, n = shots * field, n4 = n / 4, n4m = field / 4, steps = 1500;
    float *buf[3], *m, *eta;
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&buf[i], n * 4)); CK(hipMemset(buf[i], 0, n * 4)); }
    CK(hipMalloc(&m, field * 4)); CK(hipMalloc(&eta, field * 4));
    CK(hipMemset(m, 0, field * 4)); CK(hipMemset(eta, 0, field * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nb = (n4 + 255) / 256;
    for (int variant = 0; variant < 5; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < steps; ++t) {
            const float4 *c = (const float4 *)buf[t % 3], *o = (const float4 *)buf[(t + 2) % 3];
            float4 *out = (float4 *)buf[(t + 1) % 3];
            switch (variant) {
                case 0: hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); break;
                case 1: hipLaunchKernelGGL(k_empty, dim3(nb), dim3(256), 0, s); break;
                case 2: hipLaunchKernelGGL(k_stream, dim3(nb), dim3(256), 0, s, c, o, (const float4 *)m, (const float4 *)eta, out, n4, n4m); break;
                case 3: hipLaunchKernelGGL(k_halo<2>, dim3(nb), dim3(256), 0, s, c, o, (const float4 *)m, (const float4 *)eta, out, n4, n4m); break;
                case 4: hipLaunchKernelGGL(k_halo<5>, dim3(nb), dim3(256), 0, s, c, o, (const float4 *)m, (const float4 *)eta, out, n4, n4m); break;
            }
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const char *names[] = {"empty 1 block", "empty 816-ish blocks", "stream 3 reads + 1 write", "stream + 2 extra reads", "stream + 5 extra reads"};
        printf("%-28s blocks %4d : %.3f us per step\n", names[variant], variant == 0 ? 1 : nb, best * 1e3 / steps);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
