// What does one stage of the split-fp16 main loop cost on gfx950, piece by piece?
// One workgroup per CU, RS = 8 (64 x 128 tile, 24 v_mfma_f32_16x16x32_f16 + 18 ds_read_b128 per wave-stage).
//   0: MFMAs only, operands in registers                        (calibrates the clock: 16 cycles each)
//   1: + the 18 software-pipelined ds_read_b128 of the stage, 4 waves, no barrier
//   2: + one s_barrier per stage (4 waves)
//   3: + 4 idle partner waves that only take the barrier (8 waves, as the kernel's loader waves do)
//   4: the partner waves also stream a stage of operands into LDS by global_load_lds (L2-resident source)
//   5: like 4 but the consumers run no MFMA (reads + barrier only): the LDS-fill stream by itself
//   /tmp/bench_f16_loop iters
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int RS = 8, ROWS = 64 + 16 * RS, TILE_F4 = ROWS * 8, NBUF = 4;

__device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int VAR>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* out, int iters, int stream, int nblk, unsigned long long* clk, int rndlds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* stage = reinterpret_cast<float4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int x = tid; x < NBUF * TILE_F4; x += blockDim.x) {
        if (rndlds) {  // random fp16 pairs, |x| < 2^14
            unsigned h = x * 2654435761u + blockIdx.x * 40503u;
            unsigned w[4];
            for (int c = 0; c < 4; ++c) {
                h = h * 1664525u + 1013904223u;
                const unsigned lo = (h >> 4) & 0xffff, hi = (h >> 16) & 0xffff;
                w[c] = ((lo & 0x8000) | (0x3000 + (lo & 0x3fff))) | (((hi & 0x8000) | (0x3000 + (hi & 0x3fff))) << 16);
            }
            stage[x] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
        } else {
            stage[x] = make_float4(1e-3f * (x & 255), 1.f, -1e-3f, 0.5f);
        }
    }
    __syncthreads();
    if (wave >= 4) {
        if (VAR < 3) return;
        const int lw = wave - 4;
        // 192 rows x 128 B per stage = 24 wave-instructions of 1 KB, 6 per loader wave
        unsigned voff[6];
        for (int m = 0; m < 6; ++m) {
            const int R = 8 * (lw + 4 * m) + (lane >> 3);
            voff[m] = ((unsigned)R * 512u + ((lane & 7) ^ ((R >> 1) & 7)) * 4) * 4u;
        }
        // resident mode: 32 row blocks, each shared by the 8 workgroups b % 32 of one XCD
        // stream mode : like the kernel: the 64 query rows of a block stay, the 128 support rows move to a
        //               new block every 16 stages (shared by the 8 workgroups of the XCD with equal b/8 % 4)
        const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
        for (int it = 0; it < iters; ++it) {
            if (VAR >= 4) {
                float4* buf = stage + ((it + 3) & 3) * TILE_F4;
                size_t blk = blockIdx.x % 32;
                if (stream) blk = ((size_t)(it >> 4) * 32 + (cu >> 3) * 8 + xcd) % nblk;
                const char* b = reinterpret_cast<const char*>(src) + blk * (ROWS * 2048) + (size_t)(it & 15) * 128;
#pragma unroll
                for (int m = 0; m < 6; ++m)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + voff[m]),
                                                     (__attribute__((address_space(3))) void*)(buf + 64 * (lw + 4 * m)), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            bar();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int i = lane & 15, g = lane >> 4, rsw = (i >> 1) & 7;
    const int qrow = 16 * wave + i, sh = g ^ rsw, sl = (4 + g) ^ rsw;
    struct Frag { float4 bh, bl, ah[RS], al[RS]; };
    auto load = [&](Frag& f, int buf) {
        const float4* Qs = stage + (buf & 3) * TILE_F4;
        const float4* Ss = Qs + 64 * 8;
        f.bh = Qs[qrow * 8 + sh]; f.bl = Qs[qrow * 8 + sl];
#pragma unroll
        for (int r = 0; r < RS; ++r) { f.ah[r] = Ss[(16 * r + i) * 8 + sh]; f.al[r] = Ss[(16 * r + i) * 8 + sl]; }
    };
    auto mm = [](const float4& a, const float4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    };
    f32x4 acc[RS];
    for (int r = 0; r < RS; ++r) acc[r] = f32x4{0, 0, 0, 0};
    auto mf = [&](const Frag& f) {
        if (VAR == 5) return;
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = mm(f.al[r], f.bh, acc[r]);
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = mm(f.ah[r], f.bl, acc[r]);
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = mm(f.ah[r], f.bh, acc[r]);
    };
    auto il = [&]() {
        if (VAR == 0 || VAR == 5) return;
#pragma unroll
        for (int x = 0; x < 2 * (RS + 1); ++x) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * RS - 2 * (RS + 1), 0);
    };
    Frag f0, f1;
    load(f0, 0);
    load(f1, 1);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
        if (VAR >= 1) load(f1, it + 1);
        mf(f0);
        il();
        if (VAR >= 2) bar(); else __builtin_amdgcn_sched_barrier(0);
        if (VAR >= 1) load(f0, it + 2);
        mf(f1);
        il();
        if (VAR >= 2) bar(); else __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) {  // clock held in the loop: shader-clock ticks per 100 MHz tick (stamps go to a buffer of their own)
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0;
    for (int r = 0; r < RS; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    if (VAR == 5) s += f0.bh.x + f0.ah[3].y + f1.al[7].z + f1.bl.w;
    if (s == 12345.678f) out[tid] = s;
}

static int g_stream = 0, g_nblk = 32;
static unsigned long long* g_clk = nullptr;
static int g_rndlds = 0;
template <int VAR>
void run(const float* src, float* out, int iters, const char* what) {
    const size_t lds = (size_t)NBUF * TILE_F4 * 16;
    hipFuncSetAttribute((const void*)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 512;
    hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(threads), lds, 0, src, out, iters, g_stream, g_nblk, g_clk, g_rndlds);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(threads), lds, 0, src, out, iters, g_stream, g_nblk, g_clk, g_rndlds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    unsigned long long h[512];
    hipMemcpy(h, g_clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < 256; ++b) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    printf("var %d: %8.1f ns/stage  %7.1f cycles/stage at %.2f GHz  (fp32-equivalent %6.1f TFLOP/s)   %s\n", VAR, ns,
           cyc / 256 / iters, cyc / real * 0.1, 256.0 * 2 * 64 * 128 * 32 / ns / 1e3, what);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int rnd = argc > 2 ? atoi(argv[2]) : 0;
    g_stream = argc > 3 ? atoi(argv[3]) : 0;
    g_nblk = g_stream ? 256 : 32;   // 256 blocks x 384 KB = 100 MB
    float *src, *out;
    const size_t bytes = (size_t)g_nblk * ROWS * 2048 + 4096;
    hipMalloc(&src, bytes); hipMalloc(&out, 4096); hipMalloc(&g_clk, 4096);
    hipMemset(src, 0, bytes);
    if (rnd) {   // random fp16 pairs (|x| < 2^14) instead of zeros
        std::vector<unsigned short> h(bytes / 2);
        srand(3);
        for (auto& v : h) v = (unsigned short)((rand() & 0x8000) | (0x3000 + rand() % 0x4000));
        hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    }
    printf("source: %s, %s\n", rnd ? "random fp16" : "zeros", g_stream ? "streamed support rows (100 MB)" : "L2-resident (12 MB)");
    g_rndlds = argc > 4 ? atoi(argv[4]) : 0;
    printf("LDS image before the loop: %s\n", g_rndlds ? "random fp16" : "constants");
    run<0>(src, out, iters, "MFMA only");
    run<1>(src, out, iters, "+ ds_read_b128 x18");
    run<2>(src, out, iters, "+ barrier (4 waves)");
    run<3>(src, out, iters, "+ 4 partner waves at the barrier");
    run<4>(src, out, iters, "+ LDS-DMA stream by the partner waves");
    run<5>(src, out, iters, "LDS-DMA stream + reads, no MFMA");
    return 0;
}
