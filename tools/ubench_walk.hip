// Time of the frame kernel's running-energy walk (np.cumsum of squares, float32, one frame per lane, 16 lanes) on an
// otherwise idle CU: the product's energy_walk against the bare dependent chain (the retired 8-sample-block walk it
// replaced is recorded in profiles/r2_ubench_walk.txt: 23.7 cycles per add).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I spectrogram-midi_amd/csrc tools/ubench_walk.hip -o tools/_build/ubench_walk
#include "../spectrogram-midi_amd/csrc/kernels.hip"
#include <cstdio>
namespace aegis {
// no LDS in the loop at all: the chain alone (values from registers)
__device__ __forceinline__ float chain_only(float e, float x, int n) {
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(e) : "v"(x)); }
    }
    return e;
}
template <int V>
__global__ __launch_bounds__(256) void walk_bench(long long *out, float *sink, int mp, int en_stride) {
    extern __shared__ __align__(16) unsigned char sm[];
    float *stage = reinterpret_cast<float *>(sm);               // 10240 floats
    float *en = stage + 10240;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 10240; i += 256) stage[i] = 1e-3f * (float)((i * 37) & 255);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);
    const long long t0 = clock64();
    float acc = 0.f;
    if (wid == 0 && lane < 16) {
        float *row = en + lane * en_stride;
        const float *srow = stage + lane * (512 + 4);
        auto fetch = [&](int j, float4 &a, float4 &b) {
            const float *q = srow + j + 4 * (j >> 9);
            a = *reinterpret_cast<const float4 *>(q);
            b = *reinterpret_cast<const float4 *>(q + 4);
        };
        if (V == 0) energy_walk<true>(fetch, row, mp);
        if (V == 2) acc = chain_only(0.f, stage[lane], 1568);
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = clock64();
    __syncthreads();
    if (tid == 0) out[V] = t1 - t0;
    if (tid < 16) sink[tid] = en[tid * en_stride + 5] + acc;
}
}  // namespace aegis
namespace aegis { hipError_t viterbi_set_lds_limits() { return hipSuccess; } }
int main() {
    long long *d, h[4] = {0};
    float *sink;
    hipMalloc(&d, 32); hipMalloc(&sink, 64 * 4); hipMemset(d, 0, 32);
    const int mp = 536, stride = aegis::frame_en_stride(mp);
    const size_t lds = 10240 * 4 + 16 * stride * 4;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(aegis::walk_bench<0>, dim3(1), dim3(256), lds, 0, d, sink, mp, stride);
        hipLaunchKernelGGL(aegis::walk_bench<2>, dim3(1), dim3(256), lds, 0, d, sink, mp, stride);
    }
    hipDeviceSynchronize();
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("energy_walk (product)   %8lld ticks = %.1f per add\n", h[0], h[0] / 1568.0);
    printf("chain only (registers)  %8lld ticks = %.1f per add\n", h[2], h[2] / 1568.0);
    return 0;
}
