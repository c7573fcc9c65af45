// What bounds the unvoiced-source chain of viterbi_band_kernel: NW waves of one workgroup on one CU each run the
// 51-candidate group-maximum chain (ds_read_b64 with immediate offsets + v_add_f64 with a scalar table operand +
// v_max_f64) REP times; cycles per chain and wave for NW = 1..16 and several instruction orders.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/ubench_chain.hip -o tools/_build/ubench_chain && tools/_build/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

constexpr int W = 51, GS = 7, NG = 8, REP = 256;
struct Row { double v[W]; };

// MODE 0: add, max interleaved as the compiler emits the production chain (each max waits for the add before it)
// MODE 1: a group's 7 adds first, then a max tree
// MODE 2: MODE 0 without LDS reads (operands in registers)
// MODE 3: the 51 LDS reads only (summed with integer adds so that they are not dead)
// MODE 4: two groups interleaved (two independent add/max chains in flight)
template <int MODE>
__global__ __launch_bounds__(1024) void chain(long long *out, double *sink, Row row) {
    extern __shared__ double val[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 2048; i += blockDim.x) val[i] = -1.0 - 0.001 * ((i * 37) & 255);
    double k[26];
#pragma unroll
    for (int i = 0; i < 26; ++i) {
        const double t = row.v[i];
        k[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t)), __builtin_amdgcn_readfirstlane(__double2loint(t)));
    }
    auto kat = [&](int d) { return k[d <= 25 ? d : W - 1 - d]; };
    __syncthreads();
    const double *vi = val + wid * 64 + lane;
    double acc = 0;
    long long iacc = 0;
    __builtin_amdgcn_s_waitcnt(0);
    const long long t0 = clock64();
#pragma nounroll
    for (int r = 0; r < REP; ++r) {
        int off = r & 7;                    // opaque index: the reads of one repetition must not be hoisted or merged
        asm volatile("" : "+v"(off));
        const double *v = vi + off;
        double gm[NG];
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int d0 = g * GS < W - GS ? g * GS : W - GS;
                double m = (MODE == 2 ? acc + d0 : v[d0]) + kat(W - 1 - d0);
#pragma unroll
                for (int q = 1; q < GS; ++q) m = fmax(m, (MODE == 2 ? acc + (d0 + q) : v[d0 + q]) + kat(W - 1 - d0 - q));
                gm[g] = m;
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int d0 = g * GS < W - GS ? g * GS : W - GS;
                double c[GS];
#pragma unroll
                for (int q = 0; q < GS; ++q) c[q] = v[d0 + q] + kat(W - 1 - d0 - q);
                gm[g] = fmax(fmax(fmax(c[0], c[1]), fmax(c[2], c[3])), fmax(fmax(c[4], c[5]), c[6]));
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int d = 0; d < W; ++d) iacc += __double_as_longlong(v[d]);
#pragma unroll
            for (int g = 0; g < NG; ++g) gm[g] = 0;
        } else {
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                const int d0 = g * GS, d1 = (g + 1) * GS < W - GS ? (g + 1) * GS : W - GS;
                double m0 = v[d0] + kat(W - 1 - d0), m1 = v[d1] + kat(W - 1 - d1);
#pragma unroll
                for (int q = 1; q < GS; ++q) {
                    m0 = fmax(m0, v[d0 + q] + kat(W - 1 - d0 - q));
                    m1 = fmax(m1, v[d1 + q] + kat(W - 1 - d1 - q));
                }
                gm[g] = m0; gm[g + 1] = m1;
            }
        }
        double b = gm[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) b = fmax(b, gm[g]);
        acc += b;
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = clock64();
    if (lane == 0) out[wid] = t1 - t0;
    sink[tid] = acc + (double)iacc;
}

template <int MODE>
void run(const char *name, long long *d, double *sink, const Row &row) {
    printf("%-44s", name);
    for (int nw : {1, 2, 4, 7, 8, 14, 16}) {
        long long h[16];
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64 * nw), 2048 * 8 + 1024 * 8, 0, d, sink, row);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int w = 0; w < nw; ++w) mx = h[w] > mx ? h[w] : mx;
        printf(" %2dw:%6.0f", nw, (double)mx / REP);
    }
    printf("   cycles per chain (slowest wave)\n");
}

int main() {
    long long *d; double *sink;
    hipMalloc(&d, 128); hipMalloc(&sink, 1024 * 8);
    Row row;
    for (int i = 0; i < W; ++i) row.v[i] = std::log(0.99 * (26 - std::abs(i - 25)) / 676.0);
    run<0>("add/max interleaved (production order)", d, sink, row);
    run<1>("7 adds then max tree", d, sink, row);
    run<4>("two groups interleaved", d, sink, row);
    run<2>("production order, operands in registers", d, sink, row);
    run<3>("51 ds_read_b64 only", d, sink, row);
    return 0;
}
