// Stand-alone probe for the round-1 crash "rocprofv3 --kernel-trace -- python3 tools/bench_stream.py" (SIGSEGV with
// aegis_stream_push on the stack, first hipGraph capture / launch).  It replays the same graph SHAPE without any of the
// library's code: optional H2D copy from pinned memory, kernels with a small or a ~2 KB by-value argument block (the
// Viterbi kernel's kernarg-resident transition row), optional D2H copy to pinned memory, captured with
// hipStreamCaptureModeThreadLocal on a non-blocking stream, instantiated and launched.  Run each variant plain and
// under rocprofv3: a variant that only dies under the profiler is the tool's problem, not the library's.
//   graph_probe <mask>   bit0 H2D node, bit1 D2H node, bit2 large kernarg, bit3 five kernels instead of one,
//                        bit4 host code touches the pinned result right after the sync (as aegis_stream_push does)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Big { double v[4][51]; double lmax[4]; double all; };
struct Small { double a; };

template <typename A>
__global__ void k(const float *in, float *out, int n, A arg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + (float)reinterpret_cast<const double *>(&arg)[i % (sizeof(A) / 8)];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)

int main(int argc, char **argv) {
    const int mask = argc > 1 ? atoi(argv[1]) : 31;
    const int n = 2048;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *d_in, *d_out, *pin_in, *pin_out;
    CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_out, n * 4));
    CK(hipHostMalloc(reinterpret_cast<void **>(&pin_in), n * 4, hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void **>(&pin_out), n * 4, hipHostMallocDefault));
    for (int i = 0; i < n; ++i) pin_in[i] = (float)i;
    CK(hipMemset(d_in, 0, n * 4));
    Big big; std::memset(&big, 0, sizeof(big));
    Small small{0.0};
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    if (mask & 1) CK(hipMemcpyAsync(d_in, pin_in, n * 4, hipMemcpyHostToDevice, s));
    const int nk = (mask & 8) ? 5 : 1;
    for (int q = 0; q < nk; ++q) {
        if (mask & 4) hipLaunchKernelGGL(k<Big>, dim3(n / 256), dim3(256), 0, s, d_in, d_out, n, big);
        else hipLaunchKernelGGL(k<Small>, dim3(n / 256), dim3(256), 0, s, d_in, d_out, n, small);
    }
    if (mask & 2) CK(hipMemcpyAsync(pin_out, d_out, n * 4, hipMemcpyDeviceToHost, s));
    hipGraph_t g = nullptr;
    CK(hipStreamEndCapture(s, &g));
    hipGraphExec_t ex = nullptr;
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    double acc = 0;
    for (int it = 0; it < 200; ++it) {
        pin_in[0] = (float)it;
        CK(hipGraphLaunch(ex, s));
        CK(hipStreamSynchronize(s));
        if (mask & 16) acc += pin_out[1];
    }
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    printf("graph_probe mask=%d OK (%g)\n", mask, acc);
    return 0;
}
