// developer aid (microbenchmark, gfx950): what does a DEVICE-SCOPE barrier between W co-resident workgroups cost when it sits in a dependent chain -- the price a
// single-stream step (configs[1]: 17 encoder / 22 decoder dependent stages) would pay per stage if a stage's weight stream were sliced over W
// workgroups instead of one.  Every workgroup: publish 128 B ("its slice of the stage's activations"), release fence, arrive on a monotonic counter, poll it with
// s_sleep until all W arrived, acquire fence, read every workgroup's 128 B.  N such rounds back to back; the chain time / N is the per-stage synchronisation price.
// build: hipcc --offload-arch=gfx950 -O3 -o wg_barrier_chain wg_barrier_chain.hip ; run: ./wg_barrier_chain [W = 8] [rounds = 2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void k(unsigned *counter, float *slab, int W, int rounds, long long *cycles, float *sink)
{
    const int w = blockIdx.x, tid = threadIdx.x;
    float acc = 0.0f;
    const long long t0 = clock64();
    for (int r = 1; r <= rounds; r++) {
        if (tid < 32) slab[((size_t)(r & 1) * W + w) * 32 + tid] = (float)(r + w) + acc * 1e-30f;
        __syncthreads();
        if (tid == 0) {
            __threadfence();                                                       // release: this workgroup's slice is visible device-wide
            atomicAdd(counter, 1u);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r * W)) __builtin_amdgcn_s_sleep(1);
            __threadfence();                                                       // acquire
        }
        __syncthreads();
        if (tid < 32 * W && tid < 512) acc += __builtin_nontemporal_load(&slab[((size_t)(r & 1) * W + tid / 32) * 32 + (tid & 31)]);
    }
    if (tid == 0) cycles[w] = clock64() - t0;
    sink[w * 512 + tid] = acc;
}
int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 8, rounds = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned *cnt; float *slab, *sink; long long *cyc;
    hipMalloc(&cnt, 4); hipMalloc(&slab, sizeof(float) * 2 * W * 32); hipMalloc(&sink, sizeof(float) * W * 512); hipMalloc(&cyc, 8 * W);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(cnt, 0, 4); hipMemset(slab, 0, sizeof(float) * 2 * W * 32);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k, dim3(W), dim3(512), 0, 0, cnt, slab, W, rounds, cyc, sink);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("W = %d workgroups, %d dependent rounds: %.3f ms -> %.2f us per round (publish 128 B, device-scope barrier, read W x 128 B)\n", W, rounds, ms, 1e3 * ms / rounds);
    }
    return 0;
}
