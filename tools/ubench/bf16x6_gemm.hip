// Second stage of the split-bf16 experiment: a whole 1x1 layer -- Y[p][co] = relu(sum_k X[p][k] W[co][k] + b) --
// through v_mfma_f32_32x32x16_bf16 with operands split into three bf16 planes (six products), staging, barriers and
// epilogue included, next to the same kernel structure on v_mfma_f32_32x32x2_f32.  The split of X is its own pass (what a
// producing layer's epilogue would do) and is timed separately; W is split once.
//   workgroup = 4 waves (2 x 2), wave tile 64 x 64, workgroup tile 128 pixels x 128 output channels, 16 channels per stage,
//   next stage prefetched into registers while the current one is multiplied.
//   hipcc --offload-arch=gfx950 -O2 bf16x6_gemm.hip -o bf16x6_gemm && ./bf16x6_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __host__ inline unsigned f2u(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
__device__ __host__ inline float u2f(unsigned u) { float x; memcpy(&x, &u, 4); return x; }
__device__ __host__ inline float bf16_rne(float x)
{
    unsigned u = f2u(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u2f(u & 0xFFFF0000u);
}

// X [rows][K] fp32 -> three planes [3][rows][K] of bf16 (h, m, l)
__global__ void split_kernel(const float* __restrict__ x, u16* __restrict__ planes, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = x[i];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float h = bf16_rne(v);
            planes[(size_t)p * n + i] = (u16)(f2u(h) >> 16);
            v -= h;
        }
    }
}

#define ROWB 48  // LDS bytes per row of 16 bf16 (32 B) + pad: conflict-free ds_read_b128 over 16 lanes
// planes: A [3][P][K], B [3][Co][K] (bf16 bits); Y [P][Co]
__global__ __launch_bounds__(256) void gemm_bf16x6(const u16* __restrict__ A, const u16* __restrict__ B, const float* __restrict__ bias,
                                                   float* __restrict__ Y, int P, int Co, int K)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * 128 * ROWB];  // [side][plane][row][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int p0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
    const int lrow = tid >> 1, lhalf = tid & 1;  // staging: thread -> (row, 8-channel half) of both sides
    const size_t nA = (size_t)P * K, nB = (size_t)Co * K;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    uint4 ra[3], rb[3];
    auto gload = [&](int k) {
        const int pr = p0 + lrow < P ? p0 + lrow : P - 1, cr = c0 + lrow < Co ? c0 + lrow : Co - 1;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            ra[pl] = *reinterpret_cast<const uint4*>(A + pl * nA + (size_t)pr * K + k + lhalf * 8);
            rb[pl] = *reinterpret_cast<const uint4*>(B + pl * nB + (size_t)cr * K + k + lhalf * 8);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<uint4*>(lds + ((0 * 3 + pl) * 128 + lrow) * ROWB + lhalf * 16) = ra[pl];
            *reinterpret_cast<uint4*>(lds + ((1 * 3 + pl) * 128 + lrow) * ROWB + lhalf * 16) = rb[pl];
        }
    };
    gload(0);
    for (int k = 0; k < K; k += 16) {
        __syncthreads();  // the previous stage's fragments have been read
        lstore();
        __syncthreads();
        if (k + 16 < K) gload(k + 16);
        bf16x8 a[2][3], b[2][3];
        const int r31 = lane & 31, kb = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[i][pl] = *reinterpret_cast<const bf16x8*>(lds + ((0 * 3 + pl) * 128 + wm * 64 + i * 32 + r31) * ROWB + kb * 16);
                b[i][pl] = *reinterpret_cast<const bf16x8*>(lds + ((1 * 3 + pl) * 128 + wn * 64 + i * 32 + r31) * ROWB + kb * 16);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            }
    }
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = c0 + wn * 64 + j * 32 + l31;
            const float bv = co < Co ? bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p < P && co < Co) Y[(size_t)p * Co + co] = fmaxf(acc[i][j][r] + bv, 0.f);
            }
        }
}

#define ROWF 80  // LDS bytes per row of 16 floats (64 B) + pad
__global__ __launch_bounds__(256) void gemm_f32(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                float* __restrict__ Y, int P, int Co, int K)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * ROWF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int p0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
    const int lrow = tid >> 1, lhalf = tid & 1;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[2], rb[2];
    auto gload = [&](int k) {
        const int pr = p0 + lrow < P ? p0 + lrow : P - 1, cr = c0 + lrow < Co ? c0 + lrow : Co - 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra[q] = *reinterpret_cast<const f32x4*>(A + (size_t)pr * K + k + lhalf * 8 + q * 4);
            rb[q] = *reinterpret_cast<const f32x4*>(B + (size_t)cr * K + k + lhalf * 8 + q * 4);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<f32x4*>(lds + (0 * 128 + lrow) * ROWF + lhalf * 32 + q * 16) = ra[q];
            *reinterpret_cast<f32x4*>(lds + (1 * 128 + lrow) * ROWF + lhalf * 32 + q * 16) = rb[q];
        }
    };
    gload(0);
    for (int k = 0; k < K; k += 16) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (k + 16 < K) gload(k + 16);
        const int r31 = lane & 31, hh = lane >> 5;
        // lane half hh supplies channel 2 s + hh of step s (8 steps of v_mfma_f32_32x32x2_f32 per 16 channels)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {  // channels 4 s4 .. 4 s4 + 3: one b128 per row gives two steps
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(lds + (0 * 128 + wm * 64 + i * 32 + r31) * ROWF + s4 * 16);
                b[i] = *reinterpret_cast<const f32x4*>(lds + (1 * 128 + wn * 64 + i * 32 + r31) * ROWF + s4 * 16);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][2 * st + hh], b[j][2 * st + hh], acc[i][j], 0, 0, 0);
        }
    }
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = c0 + wn * 64 + j * 32 + l31;
            const float bv = co < Co ? bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p < P && co < Co) Y[(size_t)p * Co + co] = fmaxf(acc[i][j][r] + bv, 0.f);
            }
        }
}


// ---- the same layer through a pipelined structure: 32 channels per stage, operands by LDS DMA (global_load_lds, 16 bytes per
// lane, a wave's 64 lanes land in 1 KB of consecutive LDS) into two alternating buffers, XOR-swizzled 16-byte chunks so that
// the fragment reads are conflict-free; the DMA of stage k + 1 runs under the MFMAs of stage k.
// X6: planes A [3][P][K], B [3][Co][K] bf16;  else: A [P][K], B [Co][K] fp32.
template <bool X6>
__global__ __launch_bounds__(256) void gemm_pipe(const void* __restrict__ Av, const void* __restrict__ Bv, const float* __restrict__ bias,
                                                 float* __restrict__ Y, int P, int Co, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int ROW = X6 ? 64 : 128;         // bytes per row and stage (32 channels)
    constexpr int SIDE = 128 * ROW;            // one operand tile of one plane
    constexpr int NPL = X6 ? 3 : 1;
    constexpr int STAGE = 2 * NPL * SIDE;      // A planes, then B planes
    constexpr int CPR = ROW / 16;              // 16-byte chunks per row
    constexpr int RPW = 64 / CPR;              // rows per wave-load (1 KB)
    constexpr int ESZ = X6 ? 2 : 4;
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int p0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
    const char* A = (const char*)Av;
    const char* B = (const char*)Bv;
    const size_t nA = (size_t)P * K * ESZ, nB = (size_t)Co * K * ESZ;
    // DMA geometry of this lane: row within a wave-load, chunk position, and the source chunk that belongs there
    const int drow = lane / CPR, dpos = lane % CPR;
    auto swz = [&](int row) { return X6 ? ((row >> 2) & 3) : ((row >> 1) & 7); };
    auto dma = [&](int k, int buf) {
        // wave-loads of a stage: [side][plane][128 / RPW]; spread over the four waves
        constexpr int LOADS = 2 * NPL * (128 / RPW);
#pragma unroll
        for (int u = 0; u < LOADS / 4; ++u) {
            const int w = wave + 4 * u;  // wave-uniform
            const int side = w / (NPL * (128 / RPW)), rem = w % (NPL * (128 / RPW)), pl = rem / (128 / RPW), blk = rem % (128 / RPW);
            const int row = blk * RPW + drow;
            const int src_chunk = dpos ^ swz(row);
            const int grow = side ? (c0 + row < Co ? c0 + row : Co - 1) : (p0 + row < P ? p0 + row : P - 1);
            const char* g = (side ? B + pl * nB : A + pl * nA) + ((size_t)grow * K + k) * ESZ + src_chunk * 16;
            unsigned char* l = lds + buf * STAGE + (side * NPL + pl) * SIDE + blk * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int r31 = lane & 31, hh = lane >> 5;
    dma(0, 0);
    int buf = 0;
    for (int k = 0; k < K; k += 32, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // stage k has landed for everybody, and everybody is done reading the other buffer
        if (k + 32 < K) dma(k + 32, buf ^ 1);
        const unsigned char* sa = lds + buf * STAGE;
        const unsigned char* sb = sa + NPL * SIDE;
        if (X6) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {  // two MFMA steps of 16 channels
                bf16x8 a[2][3], b[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const int ra_ = wm * 64 + i * 32 + r31, rb_ = wn * 64 + i * 32 + r31;
                        a[i][pl] = *reinterpret_cast<const bf16x8*>(sa + pl * SIDE + ra_ * ROW + (((2 * s + hh) ^ swz(ra_)) * 16));
                        b[i][pl] = *reinterpret_cast<const bf16x8*>(sb + pl * SIDE + rb_ * ROW + (((2 * s + hh) ^ swz(rb_)) * 16));
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {  // 4 channels per chunk: two steps of v_mfma_f32_32x32x2_f32
                f32x4 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ra_ = wm * 64 + i * 32 + r31, rb_ = wn * 64 + i * 32 + r31;
                    a[i] = *reinterpret_cast<const f32x4*>(sa + ra_ * ROW + ((q ^ swz(ra_)) * 16));
                    b[i] = *reinterpret_cast<const f32x4*>(sb + rb_ * ROW + ((q ^ swz(rb_)) * 16));
                }
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][2 * st + hh], b[j][2 * st + hh], acc[i][j], 0, 0, 0);
            }
        }
    }
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = c0 + wn * 64 + j * 32 + l31;
            const float bv = co < Co ? bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p < P && co < Co) Y[(size_t)p * Co + co] = fmaxf(acc[i][j][r] + bv, 0.f);
            }
        }
}

static float timeit(void (*fn)(void*), void* ctx, int reps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    fn(ctx);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) fn(ctx);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
struct Ctx {
    const float *X, *W, *b;
    u16 *Xp, *Wp;
    float* Y;
    int P, Co, K;
};
static void run_split(void* v)
{
    Ctx* c = (Ctx*)v;
    hipLaunchKernelGGL(split_kernel, dim3(2048), dim3(256), 0, 0, c->X, c->Xp, (size_t)c->P * c->K);
}
static void run_x6(void* v)
{
    Ctx* c = (Ctx*)v;
    hipLaunchKernelGGL(gemm_bf16x6, dim3((c->P + 127) / 128, (c->Co + 127) / 128), dim3(256), 0, 0, c->Xp, c->Wp, c->b, c->Y, c->P, c->Co, c->K);
}
static void run_f32(void* v)
{
    Ctx* c = (Ctx*)v;
    hipLaunchKernelGGL(gemm_f32, dim3((c->P + 127) / 128, (c->Co + 127) / 128), dim3(256), 0, 0, c->X, c->W, c->b, c->Y, c->P, c->Co, c->K);
}

static void run_px6(void* v)
{
    Ctx* c = (Ctx*)v;
    hipLaunchKernelGGL(gemm_pipe<true>, dim3((c->P + 127) / 128, (c->Co + 127) / 128), dim3(256), 2 * 6 * 128 * 64, 0, (const void*)c->Xp, (const void*)c->Wp,
                       c->b, c->Y, c->P, c->Co, c->K);
}
static void run_pf32(void* v)
{
    Ctx* c = (Ctx*)v;
    hipLaunchKernelGGL(gemm_pipe<false>, dim3((c->P + 127) / 128, (c->Co + 127) / 128), dim3(256), 2 * 2 * 128 * 128, 0, (const void*)c->X, (const void*)c->W,
                       c->b, c->Y, c->P, c->Co, c->K);
}

int main()
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pipe<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 6 * 128 * 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pipe<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 128 * 128);
    struct { const char* name; int P, K, Co; } shapes[] = {
        {"128x160 x4: 1x1 192 -> 96", 4 * 128 * 160, 192, 96}, {"128x160 x4: 1x1 96 -> 192", 4 * 128 * 160, 96, 192},
        {"256x320 x4: 1x1 384 -> 192", 4 * 256 * 320, 384, 192}, {"32x40 x4: 1x1 1280 -> 213", 4 * 32 * 40, 1280, 213},
        {"32x40 x4: 1x1 2816 -> 469", 4 * 32 * 40, 2816, 469}, {"32x40 x4: 1x1 320 -> 160", 4 * 32 * 40, 320, 160}};
    srand(2);
    for (auto& s : shapes) {
        const size_t nX = (size_t)s.P * s.K, nW = (size_t)s.Co * s.K, nY = (size_t)s.P * s.Co;
        float *hX = (float*)malloc(nX * 4), *hW = (float*)malloc(nW * 4), *hb = (float*)malloc(s.Co * 4), *h1 = (float*)malloc(nY * 4),
              *h2 = (float*)malloc(nY * 4);
        for (size_t i = 0; i < nX; ++i) hX[i] = (float)((rand() % 2001) - 1000) / 500.f * (1.f + (rand() % 97) * 1e-4f);
        for (size_t i = 0; i < nW; ++i) hW[i] = (float)((rand() % 2001) - 1000) / 20000.f * (1.f + (rand() % 89) * 1e-4f);
        for (int i = 0; i < s.Co; ++i) hb[i] = 0.01f * i;
        Ctx c;
        c.P = s.P; c.K = s.K; c.Co = s.Co;
        float *X, *W, *b, *Y;
        (void)hipMalloc(&X, nX * 4); (void)hipMalloc(&W, nW * 4); (void)hipMalloc(&b, s.Co * 4); (void)hipMalloc(&Y, nY * 4);
        (void)hipMalloc(&c.Xp, nX * 6); (void)hipMalloc(&c.Wp, nW * 6);
        (void)hipMemcpy(X, hX, nX * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, hW, nW * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(b, hb, s.Co * 4, hipMemcpyHostToDevice);
        c.X = X; c.W = W; c.b = b; c.Y = Y;
        hipLaunchKernelGGL(split_kernel, dim3(2048), dim3(256), 0, 0, W, c.Wp, nW);
        const float t_split = timeit(run_split, &c, 20);
        const float t_x6 = timeit(run_x6, &c, 20);
        (void)hipMemcpy(h1, Y, nY * 4, hipMemcpyDeviceToHost);
        const float t_f32 = timeit(run_f32, &c, 20);
        (void)hipMemcpy(h2, Y, nY * 4, hipMemcpyDeviceToHost);
        // accuracy on a sample of outputs against double
        double e6 = 0, e32 = 0;
        for (int t = 0; t < 2000; ++t) {
            const size_t p = (size_t)rand() % s.P;
            const int co = rand() % s.Co;
            double ref = hb[co], mag = 0;
            for (int k = 0; k < s.K; ++k) {
                ref += (double)hX[p * s.K + k] * hW[(size_t)co * s.K + k];
                mag += fabs((double)hX[p * s.K + k] * hW[(size_t)co * s.K + k]);
            }
            ref = ref > 0 ? ref : 0;
            e6 = fmax(e6, fabs(h1[p * s.Co + co] - ref) / mag);
            e32 = fmax(e32, fabs(h2[p * s.Co + co] - ref) / mag);
        }
        const double gf = 2.0 * s.P * s.K * s.Co / 1e9;
        const float t_px6 = timeit(run_px6, &c, 20);
        (void)hipMemcpy(h1, Y, nY * 4, hipMemcpyDeviceToHost);
        const float t_pf32 = timeit(run_pf32, &c, 20);
        (void)hipMemcpy(h2, Y, nY * 4, hipMemcpyDeviceToHost);
        double p6 = 0, p32 = 0;
        srand(7);
        for (int t = 0; t < 2000; ++t) {
            const size_t p = (size_t)rand() % s.P;
            const int co = rand() % s.Co;
            double ref = hb[co], mag = 0;
            for (int k = 0; k < s.K; ++k) {
                ref += (double)hX[p * s.K + k] * hW[(size_t)co * s.K + k];
                mag += fabs((double)hX[p * s.K + k] * hW[(size_t)co * s.K + k]);
            }
            ref = ref > 0 ? ref : 0;
            p6 = fmax(p6, fabs(h1[p * s.Co + co] - ref) / mag);
            p32 = fmax(p32, fabs(h2[p * s.Co + co] - ref) / mag);
        }
        printf("%-30s fp32 %7.1f us %6.1f TF/s (err %.1e) | bf16x6 %7.1f us %6.1f TF/s (err %.1e) + split of X %6.1f us | %.2fx (%.2fx with the split)\n", s.name,
               t_f32 * 1e3, gf / t_f32, e32, t_x6 * 1e3, gf / t_x6, e6, t_split * 1e3, t_f32 / t_x6, t_f32 / (t_x6 + t_split));
        printf("%-30s pipelined: fp32 %7.1f us %6.1f TF/s (err %.1e) | bf16x6 %7.1f us %6.1f TF/s (err %.1e) | %.2fx (%.2fx with the split)\n", "", t_pf32 * 1e3,
               gf / t_pf32, p32, t_px6 * 1e3, gf / t_px6, p6, t_pf32 / t_px6, t_pf32 / (t_px6 + t_split));
        (void)hipFree(X); (void)hipFree(W); (void)hipFree(b); (void)hipFree(Y); (void)hipFree(c.Xp); (void)hipFree(c.Wp);
        free(hX); free(hW); free(hb); free(h1); free(h2);
    }
    return 0;
}
