// ckr_train.hip -- the training step of the reference's network in hand-written HIP.
//
// Reference: train_nn (training_pipeline.py:123-179) fits create_nn's model (:59-114) with Keras: float32
// arithmetic, loss = w_p * categorical cross-entropy + w_v * MSE + l2 penalties, Adam.  This file holds the
// device side of one optimisation step on a batch of B boards (P = 64 B positions, activations [P][C] float32,
// channels last): forward in training mode (BatchNormalization on batch statistics), backward, Adam.
// The host side (train_hip.py) owns the buffers and the order of the launches.
//
//   * the 3x3 convolutions are IMPLICIT GEMMs (k = tap * 128 + c) on the float32 matrix pipe
//     (v_mfma_f32_32x32x2_f32: exact float32 products, float32 accumulation, the arithmetic Keras uses),
//     128 x 128 tiles, K chunks of 32 staged through LDS, split-K with a deterministic reduction:
//       forward        Z[p][o]   = sum_{tap,c} X[p + off(tap)][c] W[o][tap,c]        k_gemm_nt<+1>, rows gathered
//       data gradient  dX[p][c]  = sum_{tap,o} dZ[p - off(tap)][o] Wt[c][tap,o]      k_gemm_nt<-1>, Wt from k_wflip
//       weight grad.   dW[o][tap,c] = sum_p dZ[p][o] X[p + off(tap)][c]              k_wgrad_tn, K = positions
//     no im2col matrix exists for the 128 -> 128 layers; the 14-plane first layer (K = 126) uses a small one;
//     by default the three run on the bf16 matrix pipe with every float32 operand split into three bfloat16 pieces
//     (k_gemm_nt6 / k_wgrad_tn6: six products per multiply-add, float32-grade, 2.65 x the float32 matrix rate);
//   * the split-K reductions carry the next elementwise step (bias + ReLU + BatchNorm partial sums forward,
//     the BatchNorm-backward partial sums on the way back), and the BatchNorm apply kernels finish the
//     per-channel sums in their prologue, so a conv block is 3 launches forward and 6 backward;
//   * everything else (1x1 convolutions and the dense layers of the two heads, losses, Adam with the l2 terms)
//     is bandwidth- or latency-bound elementwise / reduction work in plain float32.
#include "ckr_host.h"
#include <cstring>
#include <hip/hip_runtime.h>

namespace ckrt {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ------------------------------------------------------------------------------------------------ GEMM (NT)
constexpr int BM = 128, BN = 128, BK = 32, GT = 256;
constexpr int PITCH = BK + 4;                                     // floats per LDS row: 144 B, conflict-free b128 reads

// C[M][N] (ldc) = sum_k A[m][k] * Bt[n][k]; M % 128 == 0, N % 128 == 0, K % (32 * slices) == 0.
// gridDim = (N / 128, M / 128, slices); slice z covers k in [z * K / slices, (z + 1) * K / slices) and writes
// C + z * M * ldc (the caller reduces the slices).
// GATHER = 0: A is a plain [M][lda] matrix.  GATHER = +1 / -1: A is an activation [M = positions][128] and
// column k = tap * 128 + c of the virtual matrix is A[p + GATHER * off(tap)][c], 0 outside the 8x8 board
// (off(tap) = 8 dy + dx, tap = 3 (dy + 1) + (dx + 1)); a chunk of 32 columns never straddles a tap.
template <int GATHER>
__global__ __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_nt(const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb,
                                                float* __restrict__ C, int ldc, int M, int K) {
    __shared__ __attribute__((aligned(16))) float As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[BN * PITCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kper = K / gridDim.z, kbeg = blockIdx.z * kper;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    // staging registers of the next chunk: named scalars, not arrays (an indexed float4 array captured by a lambda is
    // moved to LDS by the compiler's alloca promotion here, which puts a vmcnt(0) wait behind every load)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const int frow = tid >> 3, fc4 = tid & 7;                     // this thread stages rows frow + 32 i, columns 4 fc4 .. 4 fc4 + 3
    unsigned on_board = 0xf;                                      // GATHER: bit i = the tap of staged row i lies on the board
    auto load_a = [&](int k0, int i) -> float4 {
        const int row = frow + 32 * i;
        if (GATHER == 0) return *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * lda + k0 + 4 * fc4);
        const int tap = k0 >> 7, dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int p = m0 + row, y = ((p >> 3) & 7) + GATHER * dy, x = (p & 7) + GATHER * dx;
        const bool in = (unsigned)y < 8u && (unsigned)x < 8u;     // branch-free: an off-board tap reads its own row, zeroed when staged
        on_board = (on_board & ~(1u << i)) | ((unsigned)in << i);
        return *reinterpret_cast<const float4*>(A + (size_t)(p + (in ? GATHER * (8 * dy + dx) : 0)) * 128 + (k0 & 127) + 4 * fc4);
    };
    auto load_b = [&](int k0, int i) -> float4 {
        return *reinterpret_cast<const float4*>(Bt + (size_t)(n0 + frow + 32 * i) * ldb + k0 + 4 * fc4);
    };
#define CKR_NT_FETCH(k0)                                                                              \
    ra0 = load_a(k0, 0); ra1 = load_a(k0, 1); ra2 = load_a(k0, 2); ra3 = load_a(k0, 3);               \
    rb0 = load_b(k0, 0); rb1 = load_b(k0, 1); rb2 = load_b(k0, 2); rb3 = load_b(k0, 3);
#define CKR_NT_STAGE(buf, r, i) *reinterpret_cast<float4*>(buf + (frow + 32 * i) * PITCH + 4 * fc4) = r;
#define CKR_NT_STAGE_A(r, i)                                                                          \
    { const bool in = GATHER == 0 || ((on_board >> i) & 1u);                                          \
      *reinterpret_cast<float4*>(As + (frow + 32 * i) * PITCH + 4 * fc4) = make_float4(in ? r.x : 0.f, in ? r.y : 0.f, in ? r.z : 0.f, in ? r.w : 0.f); }
    CKR_NT_FETCH(kbeg)
    for (int k0 = kbeg; k0 < kbeg + kper; k0 += BK) {
        __syncthreads();                                          // the previous chunk's fragments have been read
        CKR_NT_STAGE_A(ra0, 0) CKR_NT_STAGE_A(ra1, 1) CKR_NT_STAGE_A(ra2, 2) CKR_NT_STAGE_A(ra3, 3)
        CKR_NT_STAGE(Bs, rb0, 0) CKR_NT_STAGE(Bs, rb1, 1) CKR_NT_STAGE(Bs, rb2, 2) CKR_NT_STAGE(Bs, rb3, 3)
        __syncthreads();
        {
            const int kn = min(k0 + BK, kbeg + kper - BK);        // next chunk in flight under the MFMAs (the last one re-reads itself:
            CKR_NT_FETCH(kn)                                      // no branch, and the loads stay ahead of the MFMAs)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k8 = 0; k8 < BK; k8 += 8) {                      // lanes 0-31 own k8 + 0..3, lanes 32-63 k8 + 4..7
            float4 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *reinterpret_cast<const float4*>(As + (64 * wm + 32 * t + l31) * PITCH + k8 + 4 * half);
                fb[t] = *reinterpret_cast<const float4*>(Bs + (64 * wn + 32 * t + l31) * PITCH + k8 + 4 * half);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
                }
        }
    }
    float* Cz = C + (size_t)blockIdx.z * (size_t)M * ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + 64 * wm + 32 * a + 8 * g + 4 * half + i, col = n0 + 64 * wn + 32 * b + l31;
                    Cz[(size_t)row * ldc + col] = acc[a][b][4 * g + i];
                }
#undef CKR_NT_FETCH
#undef CKR_NT_STAGE
#undef CKR_NT_STAGE_A
}

// ------------------------------------------------------------------------------------------------ the same GEMMs on the bf16 pipe
// float32 operands split into THREE bfloat16 pieces, x = b1 + b2 + b3 (each rounded to nearest even, residuals exact in
// float32: 3 x 8 significant bits cover the 24 of a float32), and the six products that matter
//     b1 b1' + b1 b2' + b2 b1' + b2 b2' + b1 b3' + b3 b1'        (dropped: b2 b3', b3 b2', b3 b3' <= 2^-26 of the product)
// accumulated in the float32 accumulators of v_mfma_f32_32x32x16_bf16 (bf16 products are exact in float32): float32-grade
// results with float32's exponent range (no scaling) at 1/6 of the 2.5 PFLOP/s bf16 rate = 2.65 x the float32 matrix rate.
// Pieces are made when a chunk is staged into LDS (v_cvt_pk_bf16_f32); LDS row = [b1 x 32 | b2 x 32 | b3 x 32] + 16 B pad.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
struct Split4 { uint2 p1, p2, p3; };                               // 4 floats -> 3 x (4 bf16)
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 t = {a, b};
    const bf16x2 r = __builtin_convertvector(t, bf16x2);
    return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ Split4 split3(float4 v) {
    Split4 o;
    o.p1 = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    v.x -= __uint_as_float(o.p1.x << 16); v.y -= __uint_as_float(o.p1.x & 0xffff0000u);
    v.z -= __uint_as_float(o.p1.y << 16); v.w -= __uint_as_float(o.p1.y & 0xffff0000u);
    o.p2 = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    v.x -= __uint_as_float(o.p2.x << 16); v.y -= __uint_as_float(o.p2.x & 0xffff0000u);
    v.z -= __uint_as_float(o.p2.y << 16); v.w -= __uint_as_float(o.p2.y & 0xffff0000u);
    o.p3 = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    return o;
}
#define CKR_MFMA6(acc, A, B)                                                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], acc, 0, 0, 0);                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], acc, 0, 0, 0);                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], acc, 0, 0, 0);                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], acc, 0, 0, 0);                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], acc, 0, 0, 0);                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], acc, 0, 0, 0);

constexpr int P6 = 26;                                             // LDS row pitch in 8-byte units: 3 x 64 B + 16 B
template <int GATHER>
__global__ __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_nt6(const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb,
                                                float* __restrict__ C, int ldc, int M, int K) {
    __shared__ __attribute__((aligned(16))) uint2 As[BM * P6];
    __shared__ __attribute__((aligned(16))) uint2 Bs[BN * P6];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kper = K / gridDim.z, kbeg = blockIdx.z * kper;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;                // named scalars: see k_gemm_nt
    const int frow = tid >> 3, fc4 = tid & 7;
    unsigned on_board = 0xf;
    auto load_a = [&](int k0, int i) -> float4 {
        const int row = frow + 32 * i;
        if (GATHER == 0) return *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * lda + k0 + 4 * fc4);
        const int tap = k0 >> 7, dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int p = m0 + row, y = ((p >> 3) & 7) + GATHER * dy, x = (p & 7) + GATHER * dx;
        const bool in = (unsigned)y < 8u && (unsigned)x < 8u;
        on_board = (on_board & ~(1u << i)) | ((unsigned)in << i);
        return *reinterpret_cast<const float4*>(A + (size_t)(p + (in ? GATHER * (8 * dy + dx) : 0)) * 128 + (k0 & 127) + 4 * fc4);
    };
    auto load_b = [&](int k0, int i) -> float4 {
        return *reinterpret_cast<const float4*>(Bt + (size_t)(n0 + frow + 32 * i) * ldb + k0 + 4 * fc4);
    };
#define CKR_NT6_FETCH(k0)                                                                             \
    ra0 = load_a(k0, 0); ra1 = load_a(k0, 1); ra2 = load_a(k0, 2); ra3 = load_a(k0, 3);               \
    rb0 = load_b(k0, 0); rb1 = load_b(k0, 1); rb2 = load_b(k0, 2); rb3 = load_b(k0, 3);
#define CKR_NT6_STAGE(buf, r, i, keep)                                                                \
    { const bool in = keep;                                                                           \
      const Split4 sp = split3(make_float4(in ? r.x : 0.f, in ? r.y : 0.f, in ? r.z : 0.f, in ? r.w : 0.f)); \
      uint2* dst = buf + (frow + 32 * i) * P6 + fc4;                                                  \
      dst[0] = sp.p1; dst[8] = sp.p2; dst[16] = sp.p3; }
    CKR_NT6_FETCH(kbeg)
    for (int k0 = kbeg; k0 < kbeg + kper; k0 += BK) {
        __syncthreads();
        CKR_NT6_STAGE(As, ra0, 0, GATHER == 0 || (on_board & 1u)) CKR_NT6_STAGE(As, ra1, 1, GATHER == 0 || (on_board & 2u))
        CKR_NT6_STAGE(As, ra2, 2, GATHER == 0 || (on_board & 4u)) CKR_NT6_STAGE(As, ra3, 3, GATHER == 0 || (on_board & 8u))
        CKR_NT6_STAGE(Bs, rb0, 0, true) CKR_NT6_STAGE(Bs, rb1, 1, true) CKR_NT6_STAGE(Bs, rb2, 2, true) CKR_NT6_STAGE(Bs, rb3, 3, true)
        __syncthreads();
        {
            const int kn = min(k0 + BK, kbeg + kper - BK);
            CKR_NT6_FETCH(kn)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                           // lanes 0-31 own k = 16 kk + 0..7, lanes 32-63 16 kk + 8..15
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    fa[t][q] = *reinterpret_cast<const bf16x8*>(As + (64 * wm + 32 * t + l31) * P6 + 8 * q + 4 * kk + 2 * half);
                    fb[t][q] = *reinterpret_cast<const bf16x8*>(Bs + (64 * wn + 32 * t + l31) * P6 + 8 * q + 4 * kk + 2 * half);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) { CKR_MFMA6(acc[a][b], fa[a], fb[b]) }
        }
    }
    float* Cz = C + (size_t)blockIdx.z * (size_t)M * ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + 64 * wm + 32 * a + 8 * g + 4 * half + i, col = n0 + 64 * wn + 32 * b + l31;
                    Cz[(size_t)row * ldc + col] = acc[a][b][4 * g + i];
                }
#undef CKR_NT6_FETCH
#undef CKR_NT6_STAGE
}

// ---- the same GEMMs on operands split ONCE where they are produced (round 4).  k_gemm_nt6 splits every float32 operand into
// its three bfloat16 pieces while it stages a chunk -- the weights 64 times per GEMM (once per row tile), an activation nine times
// (once per tap) and again in the weight gradient -- and that VALU work, between two barriers with the matrix pipe idle, is half of
// the kernel's time (PMC: matrix pipe busy 46-50 %).  Here the PRODUCER of a tensor (k_bn_apply128, k_bn_bwd_apply128, k_wsplit)
// also stores its pieces, in the order a GEMM stages them: a row of C floats becomes C / 32 blocks of 192 bytes,
//     [chunk of 32 columns][piece 1 | piece 2 | piece 3][32 bf16]        (768 B per 128-float row, 6 912 B per 1 152-float row)
// so that the LDS image of a row's K chunk ([b1 x 32 | b2 x 32 | b3 x 32], as in k_gemm_nt6) is ONE contiguous 192-byte block of
// global memory and staging is twelve 16-byte copies per thread and chunk: no conversion, no select (a tap that leaves the board
// reads row `zero_row`, an all-zero row behind the tensor).  Same pieces, same products, same order: bit-identical to k_gemm_nt6.
__device__ __forceinline__ void store_pieces4(void* __restrict__ base, size_t row, int row_bytes, int c, const float4 v) {
    const Split4 sp = split3(v);
    char* dst = reinterpret_cast<char*>(base) + row * (size_t)row_bytes + (c >> 5) * 192 + (c & 31) * 2;
    *reinterpret_cast<uint2*>(dst) = sp.p1;
    *reinterpret_cast<uint2*>(dst + 64) = sp.p2;
    *reinterpret_cast<uint2*>(dst + 128) = sp.p3;
}
// x[rows][cols] float32 -> pieces (cols % 32 == 0); one thread per float4
__global__ __launch_bounds__(256) void k_split_rows(const float* __restrict__ x, long long rows, int cols, void* __restrict__ out3) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, n4 = rows * (cols / 4);
    if (t >= n4) return;
    const long long r = t / (cols / 4);
    const int c = (int)(t % (cols / 4)) * 4;
    store_pieces4(out3, (size_t)r, cols / 32 * 192, c, *reinterpret_cast<const float4*>(x + r * cols + c));
}

constexpr int P4 = 13;                                             // LDS row pitch in 16-byte units: 192 B + 16 B (as P6)
template <int GATHER>
__global__ __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_p6(const uint4* __restrict__ A3, const uint4* __restrict__ B3,
                                                float* __restrict__ C, int ldc, int M, int K, int zero_row) {
    __shared__ __attribute__((aligned(16))) uint4 As4[BM * P4];
    __shared__ __attribute__((aligned(16))) uint4 Bs4[BN * P4];
    const uint2* As = reinterpret_cast<const uint2*>(As4);
    const uint2* Bs = reinterpret_cast<const uint2*>(Bs4);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kper = K / gridDim.z, kbeg = blockIdx.z * kper;
    const int row_units = K / 32 * 12;                            // 16-byte units per operand row (GATHER: 48 per activation row)
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    // unit u = tid + 256 i (i < 6) of a tile's chunk: row u / 12, 16-byte unit u % 12 of the row's 192-byte block -- twelve
    // consecutive lanes read one contiguous block
    int srow[6], sj[6];
    unsigned taps_ok[6];                                          // GATHER: bit tap = that tap of the staged row lies on the board
    unsigned boff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int u = tid + 256 * i;
        srow[i] = u / 12; sj[i] = u - 12 * srow[i];
        boff[i] = (unsigned)((n0 + srow[i]) * row_units + sj[i]);
        taps_ok[i] = 0u;
        if (GATHER != 0) {
            const int p = m0 + srow[i], y = (p >> 3) & 7, x = p & 7;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + GATHER * (tap / 3 - 1), xx = x + GATHER * (tap % 3 - 1);
                taps_ok[i] |= (unsigned)((unsigned)yy < 8u && (unsigned)xx < 8u) << tap;
            }
        }
    }
    uint4 ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5;   // named scalars: see k_gemm_nt
    auto load_a = [&](int k0, int i) -> uint4 {
        if (GATHER == 0) return A3[(size_t)(m0 + srow[i]) * row_units + (k0 >> 5) * 12 + sj[i]];
        const int tap = k0 >> 7, shift = GATHER * (8 * (tap / 3 - 1) + (tap % 3 - 1));
        const int on = -(int)((taps_ok[i] >> tap) & 1u);         // all ones / zero: arithmetic, so that the loads stay branch-free
        const int p = zero_row + ((m0 + srow[i] + shift - zero_row) & on);
        return A3[(unsigned)(p * 48 + ((k0 & 127) >> 5) * 12 + sj[i])];
    };
    auto load_b = [&](int k0, int i) -> uint4 { return B3[boff[i] + (k0 >> 5) * 12]; };
#define CKR_P6_FETCH(k0)                                                                              \
    ra0 = load_a(k0, 0); ra1 = load_a(k0, 1); ra2 = load_a(k0, 2); ra3 = load_a(k0, 3); ra4 = load_a(k0, 4); ra5 = load_a(k0, 5); \
    rb0 = load_b(k0, 0); rb1 = load_b(k0, 1); rb2 = load_b(k0, 2); rb3 = load_b(k0, 3); rb4 = load_b(k0, 4); rb5 = load_b(k0, 5);
#define CKR_P6_PUT(buf, r, i) buf[srow[i] * P4 + sj[i]] = r;
    CKR_P6_FETCH(kbeg)
    for (int k0 = kbeg; k0 < kbeg + kper; k0 += BK) {
        __syncthreads();
        CKR_P6_PUT(As4, ra0, 0) CKR_P6_PUT(As4, ra1, 1) CKR_P6_PUT(As4, ra2, 2) CKR_P6_PUT(As4, ra3, 3) CKR_P6_PUT(As4, ra4, 4) CKR_P6_PUT(As4, ra5, 5)
        CKR_P6_PUT(Bs4, rb0, 0) CKR_P6_PUT(Bs4, rb1, 1) CKR_P6_PUT(Bs4, rb2, 2) CKR_P6_PUT(Bs4, rb3, 3) CKR_P6_PUT(Bs4, rb4, 4) CKR_P6_PUT(Bs4, rb5, 5)
        __syncthreads();
        {
            const int kn = min(k0 + BK, kbeg + kper - BK);        // next chunk in flight under the MFMAs (the last one re-reads itself)
            CKR_P6_FETCH(kn)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                           // lanes 0-31 own k = 16 kk + 0..7, lanes 32-63 16 kk + 8..15
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    fa[t][q] = *reinterpret_cast<const bf16x8*>(As + (64 * wm + 32 * t + l31) * P6 + 8 * q + 4 * kk + 2 * half);
                    fb[t][q] = *reinterpret_cast<const bf16x8*>(Bs + (64 * wn + 32 * t + l31) * P6 + 8 * q + 4 * kk + 2 * half);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) { CKR_MFMA6(acc[a][b], fa[a], fb[b]) }
        }
    }
    float* Cz = C + (size_t)blockIdx.z * (size_t)M * ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + 64 * wm + 32 * a + 8 * g + 4 * half + i, col = n0 + 64 * wn + 32 * b + l31;
                    Cz[(size_t)row * ldc + col] = acc[a][b][4 * g + i];
                }
#undef CKR_P6_FETCH
#undef CKR_P6_PUT
}

// k_gemm_p6 with TWO chunk buffers in LDS and one barrier per chunk: while the 48 MFMAs of chunk c run out of one buffer, the
// registers holding chunk c + 1 (fetched an iteration ago) are stored into the other and refilled with chunk c + 2 -- every global
// load has a whole iteration to arrive and no wave waits at a barrier with the matrix pipe idle.  WM = 2: 128 x 128 tile, 4 waves,
// 106 KB of LDS (one workgroup per CU, one wave per SIMD); WM = 4: 256 x 128 tile, 8 waves, 160 KB (two waves per SIMD).
template <int GATHER, int WM>
__global__ __launch_bounds__(128 * WM) void k_gemm_p6d(const uint4* __restrict__ A3, const uint4* __restrict__ B3,
                                                        float* __restrict__ C, int ldc, int M, int K, int zero_row) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];  // [2][A rows | B rows] x P4
    constexpr int TM = 64 * WM, NT = 128 * WM, NB = 128 * 12 / NT, BUF = (TM + BN) * P4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * BN;
    const int kper = K / gridDim.z, kbeg = blockIdx.z * kper, kend = kbeg + kper;
    const int row_units = K / 32 * 12;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    int alds[6], blds[NB];                                        // LDS unit of staged unit i (same in both buffers)
    unsigned aoff[6], taps_ok[6], boff[NB];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int u = tid + NT * i, r = u / 12, j = u - 12 * r;
        alds[i] = r * P4 + j;
        taps_ok[i] = 0u;
        if (GATHER != 0) {
            const int p = m0 + r, y = (p >> 3) & 7, x = p & 7;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + GATHER * (tap / 3 - 1), xx = x + GATHER * (tap % 3 - 1);
                taps_ok[i] |= (unsigned)((unsigned)yy < 8u && (unsigned)xx < 8u) << tap;
            }
            aoff[i] = (unsigned)((m0 + r) * 48 + j);
        } else aoff[i] = (unsigned)((m0 + r) * row_units + j);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int u = tid + NT * i, r = u / 12, j = u - 12 * r;
        blds[i] = TM * P4 + r * P4 + j;
        boff[i] = (unsigned)((n0 + r) * row_units + j);
    }
    // staging registers: named scalars (arrays filled through a lambda end up in scratch memory here: every global load would be
    // followed by a scratch store, and every LDS store by a scratch load and a wait)
    uint4 ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5;
    rb3 = rb4 = rb5 = make_uint4(0u, 0u, 0u, 0u);
    auto load_a = [&](int k0, int i) -> uint4 {
        if (GATHER == 0) return A3[aoff[i] + (k0 >> 5) * 12];
        const int tap = k0 >> 7, shift = GATHER * (8 * (tap / 3 - 1) + (tap % 3 - 1)) * 48, cu = ((k0 & 127) >> 5) * 12;
        const int on = -(int)((taps_ok[i] >> tap) & 1u);
        const int z = zero_row * 48 + (int)(aoff[i] % 48u);
        return A3[(unsigned)(z + (((int)aoff[i] + shift - z) & on) + cu)];
    };
    auto load_b = [&](int k0, int i) -> uint4 { return B3[boff[i] + (k0 >> 5) * 12]; };
#define CKR_P6D_FETCH(k0)                                                                             \
    ra0 = load_a(k0, 0); ra1 = load_a(k0, 1); ra2 = load_a(k0, 2); ra3 = load_a(k0, 3); ra4 = load_a(k0, 4); ra5 = load_a(k0, 5); \
    rb0 = load_b(k0, 0); rb1 = load_b(k0, 1); rb2 = load_b(k0, 2);                                    \
    if constexpr (NB > 3) { rb3 = load_b(k0, 3); rb4 = load_b(k0, 4); rb5 = load_b(k0, 5); }
#define CKR_P6D_PUT(buf)                                                                              \
    (buf)[alds[0]] = ra0; (buf)[alds[1]] = ra1; (buf)[alds[2]] = ra2; (buf)[alds[3]] = ra3; (buf)[alds[4]] = ra4; (buf)[alds[5]] = ra5; \
    (buf)[blds[0]] = rb0; (buf)[blds[1]] = rb1; (buf)[blds[2]] = rb2;                                  \
    if constexpr (NB > 3) { (buf)[blds[3]] = rb3; (buf)[blds[4]] = rb4; (buf)[blds[5]] = rb5; }
    CKR_P6D_FETCH(kbeg)
    CKR_P6D_PUT(lds4)
    { const int k1 = min(kbeg + BK, kend - BK); CKR_P6D_FETCH(k1) }
    __syncthreads();
    // Per chunk: the first half runs the 24 MFMAs of k = 0..15 while the fragments of k = 16..31 are read and the staged chunk
    // c + 1 is stored into the other buffer; ONE barrier (LDS traffic only: the global loads in flight are not waited for); the
    // second half runs the other 24 MFMAs while chunk c + 2 is requested from memory and the k = 0..15 fragments of chunk c + 1
    // are read from the buffer the barrier has just completed -- the next iteration starts on its MFMAs at once.
    bf16x8 f0a[2][3], f0b[2][3], f1a[2][3], f1b[2][3];
#define CKR_P6D_FRAGS(fa, fb, As, Bs, kk)                                                             \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                     \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                   \
        fa[t][q] = *reinterpret_cast<const bf16x8*>(As + (64 * wm + 32 * t + l31) * P6 + 8 * q + 4 * kk + 2 * half); \
        fb[t][q] = *reinterpret_cast<const bf16x8*>(Bs + (64 * wn + 32 * t + l31) * P6 + 8 * q + 4 * kk + 2 * half); }
    // product-major: consecutive MFMAs are independent; per accumulator the order of CKR_MFMA6
#define CKR_P6D_MFMAS(fa, fb)                                                                         \
    _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                                   \
        constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};                        \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                 \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                 \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][QA[q]], fb[b][QB[q]], acc[a][b], 0, 0, 0); }
    {
        const uint2* As = reinterpret_cast<const uint2*>(lds4);
        const uint2* Bs = As + TM * P6;
        CKR_P6D_FRAGS(f0a, f0b, As, Bs, 0)
    }
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK, cur ^= 1) {
        const uint2* As = reinterpret_cast<const uint2*>(lds4 + cur * BUF);
        const uint2* Bs = As + TM * P6;
        uint4* nxt = lds4 + (cur ^ 1) * BUF;
        const uint2* An = reinterpret_cast<const uint2*>(nxt);
        const uint2* Bn = An + TM * P6;
        __builtin_amdgcn_sched_barrier(0);
        CKR_P6D_FRAGS(f1a, f1b, As, Bs, 1)
        CKR_P6D_PUT(nxt)                                           // (the last iteration stores a clamped re-fetch: no branch in the body)
        CKR_P6D_MFMAS(f0a, f0b)
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 6 + NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 24 - 12 - 6 - NB, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        { const int k2 = min(k0 + 2 * BK, kend - BK); CKR_P6D_FETCH(k2) }
        CKR_P6D_FRAGS(f0a, f0b, An, Bn, 0)
        CKR_P6D_MFMAS(f1a, f1b)
#pragma unroll
        for (int i = 0; i < 6 + NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CKR_P6D_FRAGS
#undef CKR_P6D_MFMAS
    float* Cz = C + (size_t)blockIdx.z * (size_t)M * ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + 64 * wm + 32 * a + 8 * g + 4 * half + i, col = n0 + 64 * wn + 32 * b + l31;
                    Cz[(size_t)row * ldc + col] = acc[a][b][4 * g + i];
                }
}
#undef CKR_P6D_FETCH
#undef CKR_P6D_PUT
template <int WM> constexpr int gemm_p6d_lds() { return 2 * (64 * WM + BN) * P4 * 16; }

// Weight gradient of a 3x3 convolution with 128 kernels: C[z][o][n0 + c] = sum_{p in slice z} dZ[p][o] * X[p + off(tap)][c]
// (0 outside the board); gridDim = (taps, 1, slices), n0 = 128 * blockIdx.x, tap = tap0 + blockIdx.x (tap0 = 4 and one
// block column: a plain dZ^T . X).  Both operands arrive as rows of 128 floats (one position); a thread loads the same
// four columns of four consecutive positions, transposes the 4 x 4 block in registers and stores, per column m, the four
// positions as one 16-byte slot: LDS tile [8 position groups][128 columns][4 positions], slot index of column m
// swizzled (m ^ ((m >> 4) & 3)) so that the 16 lanes of a b128 store pass hit 16 different slots.  The MFMA operand of
// lane (m, half) for position group 2 j + half is then ONE ds_read_b128 (four k steps), as in k_gemm_nt.
__device__ __forceinline__ int tn_slot(int kg, int m) { return kg * 128 + (m ^ ((m >> 4) & 3)); }
__global__ __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wgrad_tn(const float* __restrict__ dZ, const float* __restrict__ X, int P, int tap0,
                                                float* __restrict__ C, int ldc) {
    __shared__ float4 As[8 * 128];
    __shared__ float4 Bs[8 * 128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int tap = tap0 + blockIdx.x, dy = tap / 3 - 1, dx = tap % 3 - 1, off = 8 * dy + dx, n0 = 128 * blockIdx.x;
    const int nchunk = P / BK;                                    // slice z: chunks [z n / Z, (z + 1) n / Z) of 32 positions
    const int pbeg = (int)((long long)blockIdx.z * nchunk / gridDim.z) * BK, pend = (int)((long long)(blockIdx.z + 1) * nchunk / gridDim.z) * BK;
    const int c4 = tid & 31, rg = tid >> 5;                       // this thread: columns 4 c4 .. 4 c4 + 3 of positions 4 rg .. 4 rg + 3
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;                // named scalars: see k_gemm_nt
    unsigned on_board = 0;                                        // bit i = the tap of position 4 rg + i lies on the board
    auto load_x = [&](int p0, int i) -> float4 {                  // branch-free: an off-board tap reads its own row, zeroed when staged
        const int p = p0 + 4 * rg + i, y = ((p >> 3) & 7) + dy, x = (p & 7) + dx;
        const bool in = (unsigned)y < 8u && (unsigned)x < 8u;
        on_board = (on_board & ~(1u << i)) | ((unsigned)in << i);
        return *reinterpret_cast<const float4*>(X + (size_t)(p + (in ? off : 0)) * 128 + 4 * c4);
    };
#define CKR_TN_FETCH(p0)                                                                               \
    ra0 = *reinterpret_cast<const float4*>(dZ + (size_t)(p0 + 4 * rg + 0) * 128 + 4 * c4);             \
    ra1 = *reinterpret_cast<const float4*>(dZ + (size_t)(p0 + 4 * rg + 1) * 128 + 4 * c4);             \
    ra2 = *reinterpret_cast<const float4*>(dZ + (size_t)(p0 + 4 * rg + 2) * 128 + 4 * c4);             \
    ra3 = *reinterpret_cast<const float4*>(dZ + (size_t)(p0 + 4 * rg + 3) * 128 + 4 * c4);             \
    rb0 = load_x(p0, 0); rb1 = load_x(p0, 1); rb2 = load_x(p0, 2); rb3 = load_x(p0, 3);
    CKR_TN_FETCH(pbeg)
    for (int p0 = pbeg; p0 < pend; p0 += BK) {
        __syncthreads();
        As[tn_slot(rg, 4 * c4 + 0)] = make_float4(ra0.x, ra1.x, ra2.x, ra3.x);
        As[tn_slot(rg, 4 * c4 + 1)] = make_float4(ra0.y, ra1.y, ra2.y, ra3.y);
        As[tn_slot(rg, 4 * c4 + 2)] = make_float4(ra0.z, ra1.z, ra2.z, ra3.z);
        As[tn_slot(rg, 4 * c4 + 3)] = make_float4(ra0.w, ra1.w, ra2.w, ra3.w);
        {
            const bool i0 = on_board & 1u, i1 = on_board & 2u, i2 = on_board & 4u, i3 = on_board & 8u;
            Bs[tn_slot(rg, 4 * c4 + 0)] = make_float4(i0 ? rb0.x : 0.f, i1 ? rb1.x : 0.f, i2 ? rb2.x : 0.f, i3 ? rb3.x : 0.f);
            Bs[tn_slot(rg, 4 * c4 + 1)] = make_float4(i0 ? rb0.y : 0.f, i1 ? rb1.y : 0.f, i2 ? rb2.y : 0.f, i3 ? rb3.y : 0.f);
            Bs[tn_slot(rg, 4 * c4 + 2)] = make_float4(i0 ? rb0.z : 0.f, i1 ? rb1.z : 0.f, i2 ? rb2.z : 0.f, i3 ? rb3.z : 0.f);
            Bs[tn_slot(rg, 4 * c4 + 3)] = make_float4(i0 ? rb0.w : 0.f, i1 ? rb1.w : 0.f, i2 ? rb2.w : 0.f, i3 ? rb3.w : 0.f);
        }
        __syncthreads();
        {
            const int pn = min(p0 + BK, pend - BK);
            CKR_TN_FETCH(pn)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {                             // lanes 0-31 own position group 2 j, lanes 32-63 group 2 j + 1
            float4 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = As[tn_slot(2 * j + half, 64 * wm + 32 * t + l31)];
                fb[t] = Bs[tn_slot(2 * j + half, 64 * wn + 32 * t + l31)];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
                }
        }
    }
    float* Cz = C + (size_t)blockIdx.z * (size_t)128 * ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 64 * wm + 32 * a + 8 * g + 4 * half + i, col = n0 + 64 * wn + 32 * b + l31;
                    Cz[(size_t)row * ldc + col] = acc[a][b][4 * g + i];
                }
}

#undef CKR_TN_FETCH

// The weight gradient on the bf16 pipe (see k_gemm_nt6): K = positions.  A thread holds 4 columns x 4 consecutive positions;
// per column and piece the four positions are one 8-byte half of a 16-byte slot of 8 positions (threads rg = 2 j, 2 j + 1
// fill the two halves): LDS tile [3 pieces][4 position groups of 8][128 columns] slots, column index swizzled as in
// k_wgrad_tn.  The MFMA operand of lane (m, half) for k step kk is the slot of position group 2 kk + half.
__device__ __forceinline__ int tn6_slot(int piece, int kg8, int m) { return (piece * 4 + kg8) * 128 + (m ^ ((m >> 4) & 3)); }
__global__ __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wgrad_tn6(const float* __restrict__ dZ, const float* __restrict__ X, int P, int tap0,
                                                float* __restrict__ C, int ldc) {
    __shared__ __attribute__((aligned(16))) uint2 As[2 * 3 * 4 * 128];
    __shared__ __attribute__((aligned(16))) uint2 Bs[2 * 3 * 4 * 128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int tap = tap0 + blockIdx.x, dy = tap / 3 - 1, dx = tap % 3 - 1, off = 8 * dy + dx, n0 = 128 * blockIdx.x;
    const int nchunk = P / BK;
    const int pbeg = (int)((long long)blockIdx.z * nchunk / gridDim.z) * BK, pend = (int)((long long)(blockIdx.z + 1) * nchunk / gridDim.z) * BK;
    const int c4 = tid & 31, rg = tid >> 5;                       // columns 4 c4 .. 4 c4 + 3 of positions 4 rg .. 4 rg + 3
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    unsigned on_board = 0;
    auto load_x = [&](int p0, int i) -> float4 {
        const int p = p0 + 4 * rg + i, y = ((p >> 3) & 7) + dy, x = (p & 7) + dx;
        const bool in = (unsigned)y < 8u && (unsigned)x < 8u;
        on_board = (on_board & ~(1u << i)) | ((unsigned)in << i);
        return *reinterpret_cast<const float4*>(X + (size_t)(p + (in ? off : 0)) * 128 + 4 * c4);
    };
#define CKR_TN6_FETCH(p0)                                                                              \
    ra0 = *reinterpret_cast<const float4*>(dZ + (size_t)(p0 + 4 * rg + 0) * 128 + 4 * c4);             \
    ra1 = *reinterpret_cast<const float4*>(dZ + (size_t)(p0 + 4 * rg + 1) * 128 + 4 * c4);             \
    ra2 = *reinterpret_cast<const float4*>(dZ + (size_t)(p0 + 4 * rg + 2) * 128 + 4 * c4);             \
    ra3 = *reinterpret_cast<const float4*>(dZ + (size_t)(p0 + 4 * rg + 3) * 128 + 4 * c4);             \
    rb0 = load_x(p0, 0); rb1 = load_x(p0, 1); rb2 = load_x(p0, 2); rb3 = load_x(p0, 3);
    // column j of the thread's 4 x 4 block: positions 4 rg .. 4 rg + 3 -> one half slot per piece
#define CKR_TN6_STAGE(buf, j, v0, v1, v2, v3)                                                          \
    { const Split4 sp = split3(make_float4(v0, v1, v2, v3));                                           \
      const int hs = rg & 1;                                                                           \
      buf[2 * tn6_slot(0, rg >> 1, 4 * c4 + j) + hs] = sp.p1;                                          \
      buf[2 * tn6_slot(1, rg >> 1, 4 * c4 + j) + hs] = sp.p2;                                          \
      buf[2 * tn6_slot(2, rg >> 1, 4 * c4 + j) + hs] = sp.p3; }
    CKR_TN6_FETCH(pbeg)
    for (int p0 = pbeg; p0 < pend; p0 += BK) {
        __syncthreads();
        CKR_TN6_STAGE(As, 0, ra0.x, ra1.x, ra2.x, ra3.x) CKR_TN6_STAGE(As, 1, ra0.y, ra1.y, ra2.y, ra3.y)
        CKR_TN6_STAGE(As, 2, ra0.z, ra1.z, ra2.z, ra3.z) CKR_TN6_STAGE(As, 3, ra0.w, ra1.w, ra2.w, ra3.w)
        {
            const bool i0 = on_board & 1u, i1 = on_board & 2u, i2 = on_board & 4u, i3 = on_board & 8u;
            CKR_TN6_STAGE(Bs, 0, i0 ? rb0.x : 0.f, i1 ? rb1.x : 0.f, i2 ? rb2.x : 0.f, i3 ? rb3.x : 0.f)
            CKR_TN6_STAGE(Bs, 1, i0 ? rb0.y : 0.f, i1 ? rb1.y : 0.f, i2 ? rb2.y : 0.f, i3 ? rb3.y : 0.f)
            CKR_TN6_STAGE(Bs, 2, i0 ? rb0.z : 0.f, i1 ? rb1.z : 0.f, i2 ? rb2.z : 0.f, i3 ? rb3.z : 0.f)
            CKR_TN6_STAGE(Bs, 3, i0 ? rb0.w : 0.f, i1 ? rb1.w : 0.f, i2 ? rb2.w : 0.f, i3 ? rb3.w : 0.f)
        }
        __syncthreads();
        {
            const int pn = min(p0 + BK, pend - BK);
            CKR_TN6_FETCH(pn)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                           // lanes 0-31: positions 16 kk + 0..7, lanes 32-63: 16 kk + 8..15
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    fa[t][q] = *reinterpret_cast<const bf16x8*>(As + 2 * tn6_slot(q, 2 * kk + half, 64 * wm + 32 * t + l31));
                    fb[t][q] = *reinterpret_cast<const bf16x8*>(Bs + 2 * tn6_slot(q, 2 * kk + half, 64 * wn + 32 * t + l31));
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) { CKR_MFMA6(acc[a][b], fa[a], fb[b]) }
        }
    }
    float* Cz = C + (size_t)blockIdx.z * (size_t)128 * ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 64 * wm + 32 * a + 8 * g + 4 * half + i, col = n0 + 64 * wn + 32 * b + l31;
                    Cz[(size_t)row * ldc + col] = acc[a][b][4 * g + i];
                }
#undef CKR_TN6_FETCH
#undef CKR_TN6_STAGE
}
// Wt[l][c][tap * 128 + o] = W[l][o][tap * 128 + c] for the seven 128 -> 128 layers (operand of the data-gradient GEMM);
// W[l] at w + offs[l] floats.  gridDim = (9 * 128 * 128 / 256, layers).
struct LayerOffsets { long long off[8]; };
__global__ void k_wflip(const float* __restrict__ w, LayerOffsets offs, float* __restrict__ wt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // t = (c * 9 + tap) * 128 + o
    const int o = t & 127, tap = (t >> 7) % 9, c = t / 1152;
    wt[(size_t)blockIdx.y * 147456 + t] = w[offs.off[blockIdx.y] + (size_t)o * 1152 + tap * 128 + c];
}

// The pieces (see k_gemm_p6) of the seven 128 -> 128 kernels, as the forward GEMM reads them (w3[l][o][k], k = tap * 128 + c) and
// flipped for the data-gradient GEMM (wt3[l][c][tap * 128 + o] = W[l][o][tap * 128 + c]).  gridDim = (144, layers, 2); either
// output may be NULL.
__global__ __launch_bounds__(256) void k_wsplit(const float* __restrict__ w, LayerOffsets offs, void* __restrict__ w3, void* __restrict__ wt3) {
    const int t = blockIdx.x * 256 + threadIdx.x;                 // float4 index within [128][1152]
    const int row = t / 288, k = (t - row * 288) * 4;
    const float* W = w + offs.off[blockIdx.y];
    if (blockIdx.z == 0) {
        if (w3) store_pieces4(reinterpret_cast<char*>(w3) + (size_t)blockIdx.y * 128 * 6912, (size_t)row, 6912, k, *reinterpret_cast<const float4*>(W + (size_t)row * 1152 + k));
    } else if (wt3) {
        const int tap = k >> 7, o = k & 127, c = row;
        const float4 v = make_float4(W[(size_t)o * 1152 + tap * 128 + c], W[(size_t)(o + 1) * 1152 + tap * 128 + c],
                                     W[(size_t)(o + 2) * 1152 + tap * 128 + c], W[(size_t)(o + 3) * 1152 + tap * 128 + c]);
        store_pieces4(reinterpret_cast<char*>(wt3) + (size_t)blockIdx.y * 128 * 6912, (size_t)row, 6912, k, v);
    }
}

// out[i] = sum_z part[z][i] (+ add[i]); float4 granularity.  Block = 64 columns x 4 slice groups (group g adds slices
// g, g + 4, ... in order; the four group sums are then added in order): deterministic, 4 x the loads in flight.
__global__ __launch_bounds__(256) void k_sum_slices(const float4* __restrict__ part, int slices, long long n4, const float4* __restrict__ add, float4* __restrict__ out) {
    __shared__ float4 red[3][64];
    const int g = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        if (g == 0 && add) s = add[i];
#pragma unroll 4
        for (int z = g; z < slices; z += 4) { const float4 v = part[(size_t)z * n4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (g) red[g - 1][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { const float4 v = red[j][threadIdx.x]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        out[i] = s;
    }
}

// Small matrices (1x1 convolutions with 8 / 1 kernels, the heads' dense layers, their gradients):
// C[m][n] = sum_k A[m * am + k * ak] * B[k * bk + n * bn] (+ C if accumulate); one thread per output, float32 FMAs
// in k order.
__global__ void k_gemm_small(const float* __restrict__ A, long long am, long long ak, const float* __restrict__ B, long long bk, long long bn,
                             float* __restrict__ C, long long ldc, int M, int N, int K, int accumulate) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)M * N) return;
    const int m = (int)(t / N), n = (int)(t % N);
    const float* a = A + m * am;
    const float* b = B + n * bn;
    float s = 0.0f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {                                  // 16 loads in flight, FMAs in k order
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { av[j] = a[(k + j) * ak]; bv[j] = b[(k + j) * bk]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(av[j], bv[j], s);
    }
    for (; k < K; ++k) s = fmaf(a[k * ak], b[k * bk], s);
    float* c = C + m * ldc + n;
    *c = accumulate ? *c + s : s;
}

// ------------------------------------------------------------------------------------------------ im2col (first layer)
// col[p][tap * cin + c] = x[p + off(tap)][c] (0 outside the 8x8 board), columns [9 cin, kpad) zero.  One thread per (p, k).
__global__ void k_im2col(const float* __restrict__ x, int P, int cin, int kpad, float* __restrict__ col) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)P * kpad) return;
    const int p = (int)(t / kpad), k = (int)(t % kpad);
    float v = 0.0f;
    if (k < 9 * cin) {
        const int tap = k / cin, c = k % cin, dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int y = (p >> 3) & 7, xx = p & 7;
        if ((unsigned)(y + dy) < 8u && (unsigned)(xx + dx) < 8u) v = x[(size_t)(p + 8 * dy + dx) * cin + c];
    }
    col[t] = v;
}

// Tall products of the heads: part[blk][m][n] = sum_{p in block} A[p][m] * B[p][n] (M <= 8, N <= 256, N | 256);
// the caller sums the blocks (k_sum_rows).  Thread = (row lane, n); 64-row blocks.
constexpr int TALL_ROWS = 64;
template <int M>
__global__ void k_tall_tn(const float* __restrict__ A, const float* __restrict__ B, int P, int N, float* __restrict__ part) {
    __shared__ float red[256];
    const int n = threadIdx.x % N, rl = threadIdx.x / N, nrl = blockDim.x / N;
    const int r0 = blockIdx.x * TALL_ROWS, r1 = min(P, r0 + TALL_ROWS);
    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.0f;
    for (int r = r0 + rl; r < r1; r += nrl) {
        const float b = B[(size_t)r * N + n];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = fmaf(A[(size_t)r * M + m], b, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        __syncthreads();
        red[threadIdx.x] = acc[m];
        __syncthreads();
        if (rl == 0) {
            float sum = acc[m];
            for (int j = 1; j < nrl; ++j) sum += red[j * N + n];
            part[((size_t)blockIdx.x * M + m) * N + n] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------ conv block glue
// Keras block: a = ReLU(z + bias); out = gamma * (a - mean) / sqrt(var + eps) + beta with batch statistics.
// Pass 1: a (in place over z) and per-block partial sums of a, a^2 per channel: part[blk][2][C].
// blockDim = 256 threads = (256 / C) row lanes x C channels (C <= 256, power of two); ROWS rows per block.
constexpr int ROWS = 64;
// The sums are SHIFTED by the channel's moving mean (shift; a good estimate of the batch mean once training runs): sum (a - c0)
// and sum (a - c0)^2, so that var = E[(a - c0)^2] - (E[a - c0])^2 does not cancel when |mean| >> std.  Block 0 leaves the
// shift it used behind the partials (part[2 C nblk + c]) for the finalising pass -- the moving mean itself is updated there.
__global__ void k_bias_relu_stats(float* __restrict__ z, const float* __restrict__ bias, int P, int Cc, int relu, float* __restrict__ part,
                                  const float* __restrict__ shift) {
    __shared__ float red[2][256];
    const int c = threadIdx.x % Cc, rl = threadIdx.x / Cc, nrl = blockDim.x / Cc;
    const int r0 = blockIdx.x * ROWS;
    float s = 0.0f, s2 = 0.0f;
    const float b = bias ? bias[c] : 0.0f;
    const float c0 = shift ? shift[c] : 0.0f;
    if (blockIdx.x == 0 && rl == 0) part[(size_t)gridDim.x * 2 * Cc + c] = c0;
    for (int r = r0 + rl; r < min(P, r0 + ROWS); r += nrl) {
        float v = z[(size_t)r * Cc + c] + b;
        if (relu) v = fmaxf(v, 0.0f);
        z[(size_t)r * Cc + c] = v;
        const float d = v - c0;
        s += d; s2 += d * d;
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < nrl; ++j) { s += red[0][j * Cc + c]; s2 += red[1][j * Cc + c]; }
        part[((size_t)blockIdx.x * 2 + 0) * Cc + c] = s;
        part[((size_t)blockIdx.x * 2 + 1) * Cc + c] = s2;
    }
}

// Sums of the per-block partials part[b][V] (V <= 256 values) by one block: blockDim / V groups of threads take
// every (blockDim / V)-th block, then the groups are added in order (double; deterministic).  Result in red[0 .. V).
__device__ inline void sum_partials(const float* __restrict__ part, int nblk, int V, double* red) {
    const int G = blockDim.x / V, v = threadIdx.x % V, g = threadIdx.x / V;
    double s = 0.0;
    if (g < G) {
#pragma unroll 8
        for (int b = g; b < nblk; b += G) s += (double)part[(size_t)b * V + v];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < V) {
        for (int j = 1; j < G; ++j) s += red[j * V + v];
    }
    __syncthreads();
    if ((int)threadIdx.x < V) red[v] = s;
    __syncthreads();
}

// Pass 2 (one block): mean, biased variance, 1 / sqrt(var + eps); moving statistics (momentum; the moving variance takes the
// UNBIASED batch variance -- what tf.keras' fused BatchNormalization of 4-D inputs and torch do -- or, biased_moving_var != 0,
// the biased one: tf.keras' non-fused BatchNormalization behind a Dense layer, training_pipeline.py:109).
// stats[0][C] = mean, stats[1][C] = inv_std.  C <= 128.
__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ part, int nblk, int P, int Cc, float eps, float momentum,
                              float* __restrict__ stats, float* __restrict__ run_mean, float* __restrict__ run_var, int biased_moving_var) {
    __shared__ double red[256];
    sum_partials(part, nblk, 2 * Cc, red);
    const int c = threadIdx.x;
    if (c >= Cc) return;
    const double d1 = red[c] / P, mean = (double)part[(size_t)nblk * 2 * Cc + c] + d1;          // shift + E[a - shift]
    double var = red[Cc + c] / P - d1 * d1;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[Cc + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double corr = biased_moving_var ? 1.0 : (double)P / (double)(P > 1 ? P - 1 : 1);
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)(var * corr);
    }
}

// Pass 3: out = gamma * (a - mean) * inv_std + beta
__global__ void k_bn_apply(const float* __restrict__ a, const float* __restrict__ stats, const float* __restrict__ gamma,
                           const float* __restrict__ beta, long long n, int Cc, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int c = (int)(t % Cc);
    out[t] = gamma[c] * ((a[t] - stats[c]) * stats[Cc + c]) + beta[c];
}

// Backward pass 1: per-block partial sums of dout and dout * ahat per channel: part[blk][2][C]
__global__ void k_bn_bwd_stats(const float* __restrict__ dout, const float* __restrict__ a, const float* __restrict__ stats,
                               int P, int Cc, float* __restrict__ part) {
    __shared__ float red[2][256];
    const int c = threadIdx.x % Cc, rl = threadIdx.x / Cc, nrl = blockDim.x / Cc;
    const int r0 = blockIdx.x * ROWS;
    const float mean = stats[c], inv = stats[Cc + c];
    float s = 0.0f, s2 = 0.0f;
    for (int r = r0 + rl; r < min(P, r0 + ROWS); r += nrl) {
        const float d = dout[(size_t)r * Cc + c];
        s += d; s2 += d * ((a[(size_t)r * Cc + c] - mean) * inv);
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < nrl; ++j) { s += red[0][j * Cc + c]; s2 += red[1][j * Cc + c]; }
        part[((size_t)blockIdx.x * 2 + 0) * Cc + c] = s;
        part[((size_t)blockIdx.x * 2 + 1) * Cc + c] = s2;
    }
}

// Backward pass 2 (one block): dbeta = sum dout, dgamma = sum dout * ahat -> sums[0][C], sums[1][C] and the gradients
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ part, int nblk, int Cc, float* __restrict__ sums,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[256];
    sum_partials(part, nblk, 2 * Cc, red);
    const int c = threadIdx.x;
    if (c >= Cc) return;
    sums[c] = (float)red[c]; sums[Cc + c] = (float)red[Cc + c];
    dbeta[c] = (float)red[c]; dgamma[c] = (float)red[Cc + c];
}

// Backward pass 3: da = gamma * inv_std * (dout - dbeta / P - ahat * dgamma / P); dz = da * [a > 0] (ReLU, if any),
// written over dout; per-block partial sums of dz per channel (the bias gradient): part[blk][C]
__global__ void k_bn_bwd_apply(float* __restrict__ dout, const float* __restrict__ a, const float* __restrict__ stats,
                               const float* __restrict__ sums, const float* __restrict__ gamma, int P, int Cc, int relu,
                               float* __restrict__ part) {
    __shared__ float red[256];
    const int c = threadIdx.x % Cc, rl = threadIdx.x / Cc, nrl = blockDim.x / Cc;
    const int r0 = blockIdx.x * ROWS;
    const float mean = stats[c], inv = stats[Cc + c], g = gamma[c] * inv, db = sums[c] / (float)P, dg = sums[Cc + c] / (float)P;
    float s = 0.0f;
    for (int r = r0 + rl; r < min(P, r0 + ROWS); r += nrl) {
        const size_t i = (size_t)r * Cc + c;
        const float av = a[i];
        float d = g * (dout[i] - db - ((av - mean) * inv) * dg);
        if (relu && !(av > 0.0f)) d = 0.0f;
        dout[i] = d;
        s += d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < nrl; ++j) s += red[j * Cc + c];
        part[(size_t)blockIdx.x * Cc + c] = s;
    }
}

// out[c] = sum_b part[b][c] (double accumulation); blocks of 64 columns x 4 row groups, groups added in order
__global__ __launch_bounds__(256) void k_sum_rows(const float* __restrict__ part, int nblk, int Cc, float* __restrict__ out) {
    __shared__ double red[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double s = 0.0;
    if (c < Cc) {
#pragma unroll 8
        for (int b = g; b < nblk; b += 4) s += (double)part[(size_t)b * Cc + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && c < Cc) out[c] = (float)(((s + red[64 + threadIdx.x]) + red[128 + threadIdx.x]) + red[192 + threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------ 128-channel conv blocks
// The same three passes for the [P][128] activations of the conv blocks, float4 per thread (32 threads per row, 32 rows
// at a time), fused with the split-K reduction of the producing GEMM; the apply kernels finish the per-channel sums
// of the previous pass in their prologue (every block repeats the small sum; block 0 stores the results).
//
// Forward pass 1: a = ReLU(sum_z ws[z] + bias) and the partial sums of a, a^2: part[blk][2][128]
// (shifted sums as in k_bias_relu_stats: part[256 nblk + c] = the shift)
__global__ __launch_bounds__(1024) void k_fwd_reduce128(const float* __restrict__ ws, int slices, const float* __restrict__ bias, int P, int rpb,
                                                       float* __restrict__ a, float* __restrict__ part, const float* __restrict__ shift) {
    __shared__ __attribute__((aligned(16))) float red[2][32][128];
    const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int r0 = blockIdx.x * rpb, r1 = min(P, r0 + rpb);
    const size_t n = (size_t)P * 128;
    const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c4);
    const float4 c0 = *reinterpret_cast<const float4*>(shift + 4 * c4);
    if (blockIdx.x == 0 && rl == 0) *reinterpret_cast<float4*>(part + (size_t)gridDim.x * 256 + 4 * c4) = c0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s;
    for (int r = r0 + rl; r < r1; r += 32) {
        const size_t i = (size_t)r * 128 + 4 * c4;
        float4 v = b;
#pragma unroll 4
        for (int z = 0; z < slices; ++z) { const float4 w = *reinterpret_cast<const float4*>(ws + z * n + i); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<float4*>(a + i) = v;
        const float4 d = make_float4(v.x - c0.x, v.y - c0.y, v.z - c0.z, v.w - c0.w);
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
        s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
    }
    *reinterpret_cast<float4*>(&red[0][rl][4 * c4]) = s;
    *reinterpret_cast<float4*>(&red[1][rl][4 * c4]) = s2;
    __syncthreads();
    if (threadIdx.x < 256) {
        const int st = threadIdx.x >> 7, c = threadIdx.x & 127;
        float t = red[st][0][c];
        for (int j = 1; j < 32; ++j) t += red[st][j][c];
        part[((size_t)blockIdx.x * 2 + st) * 128 + c] = t;
    }
}

// Forward pass 2 + 3: statistics from the partials, out = gamma * (a - mean) * inv_std + beta
__global__ __launch_bounds__(1024) void k_bn_apply128(const float* __restrict__ a, const float* __restrict__ part, int npart, int P, int rpb,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                     float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ stats,
                                                     float* __restrict__ out, void* __restrict__ out3) {
    __shared__ double red[1024];
    __shared__ __attribute__((aligned(16))) float sc[128], sh[128];          // out = a * sc + sh
    sum_partials(part, npart, 256, red);
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        const double d1 = red[c] / P, mean = (double)part[(size_t)npart * 256 + c] + d1;      // shift + E[a - shift]
        double var = red[128 + c] / P - d1 * d1;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean, inv = (float)(1.0 / sqrt(var + (double)eps));
        sc[c] = inv; sh[c] = mf;
        if (blockIdx.x == 0) {
            stats[c] = mf; stats[128 + c] = inv;
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mf;
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)(var * (double)P / (double)(P > 1 ? P - 1 : 1));
        }
    }
    __syncthreads();
    const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const float4 inv = *reinterpret_cast<const float4*>(sc + 4 * c4), mean = *reinterpret_cast<const float4*>(sh + 4 * c4);
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c4), bt = *reinterpret_cast<const float4*>(beta + 4 * c4);
    const int r0 = blockIdx.x * rpb, r1 = min(P, r0 + rpb);
    for (int r = r0 + rl; r < r1; r += 32) {
        const size_t i = (size_t)r * 128 + 4 * c4;
        const float4 v = *reinterpret_cast<const float4*>(a + i);
        float4 o;
        o.x = g.x * ((v.x - mean.x) * inv.x) + bt.x; o.y = g.y * ((v.y - mean.y) * inv.y) + bt.y;
        o.z = g.z * ((v.z - mean.z) * inv.z) + bt.z; o.w = g.w * ((v.w - mean.w) * inv.w) + bt.w;
        *reinterpret_cast<float4*>(out + i) = o;
        if (out3) store_pieces4(out3, (size_t)r, 768, 4 * c4, o);       // the operand of the next block's GEMM, split once (k_gemm_p6)
    }
}

// Backward pass 1: dout = sum_z ws[z] (+ add) (the data gradient of the layer above, reduced over its K slices) and the
// partial sums of dout, dout * ahat of THIS block's BatchNorm: part[blk][2][128]
__global__ __launch_bounds__(1024) void k_bwd_reduce128(const float* __restrict__ ws, int slices, const float* __restrict__ add, const float* __restrict__ a,
                                                       const float* __restrict__ stats, int P, int rpb, float* __restrict__ dout, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float red[2][32][128];
    const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int r0 = blockIdx.x * rpb, r1 = min(P, r0 + rpb);
    const size_t n = (size_t)P * 128;
    const float4 mean = *reinterpret_cast<const float4*>(stats + 4 * c4), inv = *reinterpret_cast<const float4*>(stats + 128 + 4 * c4);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s;
    for (int r = r0 + rl; r < r1; r += 32) {
        const size_t i = (size_t)r * 128 + 4 * c4;
        float4 v = add ? *reinterpret_cast<const float4*>(add + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int z = 0; z < slices; ++z) { const float4 w = *reinterpret_cast<const float4*>(ws + z * n + i); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        if (slices > 0 || add) *reinterpret_cast<float4*>(dout + i) = v; else v = *reinterpret_cast<const float4*>(dout + i);
        const float4 av = *reinterpret_cast<const float4*>(a + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        s2.x += v.x * ((av.x - mean.x) * inv.x); s2.y += v.y * ((av.y - mean.y) * inv.y);
        s2.z += v.z * ((av.z - mean.z) * inv.z); s2.w += v.w * ((av.w - mean.w) * inv.w);
    }
    *reinterpret_cast<float4*>(&red[0][rl][4 * c4]) = s;
    *reinterpret_cast<float4*>(&red[1][rl][4 * c4]) = s2;
    __syncthreads();
    if (threadIdx.x < 256) {
        const int st = threadIdx.x >> 7, c = threadIdx.x & 127;
        float t = red[st][0][c];
        for (int j = 1; j < 32; ++j) t += red[st][j][c];
        part[((size_t)blockIdx.x * 2 + st) * 128 + c] = t;
    }
}

// Backward pass 2 + 3: dbeta, dgamma from the partials; dz = gamma * inv_std * (dout - dbeta / P - ahat * dgamma / P) * [a > 0]
// written over dout; partial sums of dz per channel (the conv bias gradient): part2[blk][128]
__global__ __launch_bounds__(1024) void k_bn_bwd_apply128(float* __restrict__ dout, const float* __restrict__ a, const float* __restrict__ stats,
                                                         const float* __restrict__ part, int npart, const float* __restrict__ gamma, int P, int rpb,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ part2, void* __restrict__ dz3) {
    __shared__ double red[1024];
    __shared__ __attribute__((aligned(16))) float sdb[128], sdg[128];
    __shared__ __attribute__((aligned(16))) float r2[32][128];
    sum_partials(part, npart, 256, red);
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        const float db = (float)red[c], dg = (float)red[128 + c];
        sdb[c] = db / (float)P; sdg[c] = dg / (float)P;
        if (blockIdx.x == 0) { dbeta[c] = db; dgamma[c] = dg; }
    }
    __syncthreads();
    const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const float4 mean = *reinterpret_cast<const float4*>(stats + 4 * c4), inv = *reinterpret_cast<const float4*>(stats + 128 + 4 * c4);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * c4);
    const float4 db = *reinterpret_cast<const float4*>(sdb + 4 * c4), dg = *reinterpret_cast<const float4*>(sdg + 4 * c4);
    const float4 g = make_float4(gm.x * inv.x, gm.y * inv.y, gm.z * inv.z, gm.w * inv.w);
    const int r0 = blockIdx.x * rpb, r1 = min(P, r0 + rpb);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0 + rl; r < r1; r += 32) {
        const size_t i = (size_t)r * 128 + 4 * c4;
        const float4 av = *reinterpret_cast<const float4*>(a + i), d = *reinterpret_cast<const float4*>(dout + i);
        float4 o;
        o.x = av.x > 0.f ? g.x * (d.x - db.x - ((av.x - mean.x) * inv.x) * dg.x) : 0.f;
        o.y = av.y > 0.f ? g.y * (d.y - db.y - ((av.y - mean.y) * inv.y) * dg.y) : 0.f;
        o.z = av.z > 0.f ? g.z * (d.z - db.z - ((av.z - mean.z) * inv.z) * dg.z) : 0.f;
        o.w = av.w > 0.f ? g.w * (d.w - db.w - ((av.w - mean.w) * inv.w) * dg.w) : 0.f;
        *reinterpret_cast<float4*>(dout + i) = o;
        if (dz3) store_pieces4(dz3, (size_t)r, 768, 4 * c4, o);         // the operand of the data-gradient GEMM, split once
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *reinterpret_cast<float4*>(&r2[rl][4 * c4]) = s;
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = r2[0][threadIdx.x];
        for (int j = 1; j < 32; ++j) t += r2[j][threadIdx.x];
        part2[(size_t)blockIdx.x * 128 + threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------------ losses
// Policy head: logits[B][512] (+ bias) -> p = softmax; Keras categorical cross-entropy on p / sum(p) clipped to
// [1e-7, 1 - 1e-7]: ce_b = -sum_i pi_i log(clip p_i); dlogit_j = w / B * (p_j * sum_i pi_i u_i - pi_j u_j),
// u_i = [1e-7 < p_i < 1 - 1e-7] (the derivative torch takes through clamp).  One wave per board; lane owns 8 logits.
__global__ __launch_bounds__(256) void k_policy_loss(const float* __restrict__ logits, const float* __restrict__ bias,
                                                     const float* __restrict__ pi, int B, float weight,
                                                     float* __restrict__ dlogits, float* __restrict__ ce_out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float z[8], t[8];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        z[j] = logits[(size_t)b * 512 + i] + bias[i];
        t[j] = pi[(size_t)b * 512 + i];
        mx = fmaxf(mx, z[j]);
    }
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { z[j] = expf(z[j] - mx); sum += z[j]; }
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    float ce = 0.0f, tu = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        z[j] = z[j] / sum;                                        // p
        const bool u = z[j] > 1e-7f && z[j] < 1.0f - 1e-7f;
        const float pc = fminf(fmaxf(z[j], 1e-7f), 1.0f - 1e-7f);
        ce -= t[j] * logf(pc);
        tu += u ? t[j] : 0.0f;
        t[j] = u ? t[j] : 0.0f;
    }
    for (int d = 32; d >= 1; d >>= 1) { ce += __shfl_xor(ce, d); tu += __shfl_xor(tu, d); }
    const float sc = weight / (float)B;
#pragma unroll
    for (int j = 0; j < 8; ++j) dlogits[(size_t)b * 512 + lane + 64 * j] = sc * (z[j] * tu - t[j]);
    if (lane == 0) ce_out[b] = ce;
}

// Value head: v = tanh(z + bias); mse_b = (v - t)^2; dz = w * 2 (v - t) / B * (1 - v^2)
__global__ void k_value_loss(const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ target, int B, float weight,
                             float* __restrict__ dz, float* __restrict__ se_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float v = tanhf(z[b] + *bias), e = v - target[b];
    se_out[b] = e * e;
    dz[b] = weight * 2.0f * e / (float)B * (1.0f - v * v);
}

// acc[0..2] += n_rows * {w_p * mean ce + w_v * mean se + penalty, mean ce, mean se} (float64 running sums of an epoch);
// penalty = the sum of k_adam's ADAM_BLOCKS partials.  One block of 256 threads, sums in a fixed order.
constexpr int ADAM_BLOCKS = 512;
__global__ __launch_bounds__(256) void k_loss_sums(const float* __restrict__ ce, const float* __restrict__ se, int B, float wp, float wv,
                                                   const double* __restrict__ penalty_parts, double n_rows, double* __restrict__ acc) {
    __shared__ double red[3][256];
    double c = 0.0, s = 0.0, pen = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) { c += ce[b]; s += se[b]; }
    if (penalty_parts) for (int b = threadIdx.x; b < ADAM_BLOCKS; b += 256) pen += penalty_parts[b];
    red[0][threadIdx.x] = c; red[1][threadIdx.x] = s; red[2][threadIdx.x] = pen;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) for (int j = 0; j < 3; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x) return;
    c = red[0][0] / B; s = red[1][0] / B;
    acc[0] += n_rows * ((double)wp * c + (double)wv * s + red[2][0]);
    acc[1] += n_rows * c;
    acc[2] += n_rows * s;
}

// ------------------------------------------------------------------------------------------------ Adam + l2
struct LossArgs { const float* ce; const float* se; double* acc; double n_rows; int B; float wp, wv; int pad; };
// One flat parameter vector; reg[i] = the l2 coefficient of element i (CONV_REG / DENSE_REG on kernels and biases,
// 0 on BatchNorm parameters).  g = grad + 2 reg w;  torch.optim.Adam arithmetic (bias-corrected step size,
// eps outside the square root), lr and the step counter read from device memory (captured in a HIP graph).
// ADAM_BLOCKS blocks stride over the vector; block b also leaves its share of the penalty sum_i reg[i] w[i]^2 of the
// weights BEFORE the update (Keras adds it to the loss it prints) in penalty_parts[b].
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ w, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                       const float* __restrict__ reg, long long n, const float* __restrict__ lr, float beta1, float beta2, float eps,
                       float* __restrict__ step, double* __restrict__ penalty_parts, const LossArgs la) {
    __shared__ double red[256];
    __shared__ unsigned last;
    const float t = *step + 1.0f;                                 // the last block to finish stores it (every block has read *step by then)
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float rate = *lr / bc1, rs = sqrtf(bc2);
    double pen = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float wi = w[i], r = reg[i];
        pen += (double)r * (double)wi * (double)wi;
        const float g = grad[i] + 2.0f * r * wi;
        const float mi = beta1 * m[i] + (1.0f - beta1) * g;
        const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / rs + eps;
        w[i] = wi - rate * (mi / denom);
    }
    if (penalty_parts) {
        red[threadIdx.x] = pen;
        __syncthreads();
        for (int d = 128; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    }
    // the block that finishes last advances the step counter and adds the batch's losses to the running sums: the work of
    // two more launches at the end of the step's critical path.  step[1] holds the ticket (an unsigned, left at 0).
    // (No agent-scope release / acquire fences: each would write back and invalidate the XCD's L2, 512 times -- measured
    // 21 us instead of 11.  The partial sums travel as device-scope atomic stores / loads, which go to the coherent level
    // themselves; the wait between the store and the ticket keeps their order.)
    if (!threadIdx.x) {
        if (penalty_parts) {
            __hip_atomic_store(penalty_parts + blockIdx.x, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);                        // the store has been acknowledged before the ticket is taken
        }
        const unsigned ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(step + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = ticket == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if (!threadIdx.x) { step[0] = t; __hip_atomic_store(reinterpret_cast<unsigned*>(step + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (!la.acc) return;
    __shared__ double red3[3][256];
    double c = 0.0, q = 0.0, pn = 0.0;
    for (int b = threadIdx.x; b < la.B; b += 256) { c += la.ce[b]; q += la.se[b]; }
    if (penalty_parts)
        for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) pn += __hip_atomic_load(penalty_parts + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red3[0][threadIdx.x] = c; red3[1][threadIdx.x] = q; red3[2][threadIdx.x] = pn;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) for (int j = 0; j < 3; ++j) red3[j][threadIdx.x] += red3[j][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x) return;
    c = red3[0][0] / la.B; q = red3[1][0] / la.B;
    la.acc[0] += la.n_rows * ((double)la.wp * c + (double)la.wv * q + red3[2][0]);
    la.acc[1] += la.n_rows * c;
    la.acc[2] += la.n_rows * q;
}


// ------------------------------------------------------------------------------------------------ the value head in four launches
// 1x1 conv (1 kernel) + ReLU + BN -> flatten -> Dense(64) + ReLU + BN -> Dense(1) -> tanh, its loss and its whole backward pass
// (training_pipeline.py:102-112).  At the reference's batch (128) every matrix here is tiny and the 26 launches of the
// layer-by-layer sequence cost 4.5 us each whatever they compute (profiles/r03_train_step_timeline_before.txt: the graph issues
// ~119 nodes per step at ~8 us per node); so: k_vh_conv (positions in parallel: the 1x1 convolution, ReLU, BatchNorm partial
// sums), k_vh_mlp (ONE workgroup walks the rest of the forward pass, the loss and the backward pass down to the gradient w.r.t.
// the convolution's output; the arithmetic and its order follow the layer kernels above), k_vh_conv_bwd (positions in parallel:
// gradient w.r.t. the body's output, partial sums of the 1x1 kernel's gradient) + k_sum_rows.
struct ValueHead {
    const float* body; const float* target;                       // [P][128], [B]
    const float* v1_w; const float* v1_b; const float* v1_g; const float* v1_beta;       // [128], [1], [1], [1]
    const float* f1_w; const float* f1_b; const float* vbn_g; const float* vbn_beta;     // [64 out][64 in], [64], [64], [64]
    const float* f2_w; const float* f2_b;                         // [64], [1]
    float* v1_rm; float* v1_rv; float* vbn_rm; float* vbn_rv;      // moving statistics
    float* stats_v1; float* stats_vbn;                            // [2][1], [2][64]: mean, 1 / sqrt(var + eps)
    float* g_v1_w; float* g_v1_b; float* g_v1_g; float* g_v1_beta; float* g_f1_w; float* g_f1_b; float* g_vbn_g; float* g_vbn_beta;
    float* g_f2_w; float* g_f2_b;
    float* a_v1; float* out_v1; float* a_f1; float* out_f1; float* dz_f2; float* d_f1; float* d_v1;      // kept activations / scratch
    float* d_body; float* se; float* part;                        // [P][128] gradient w.r.t. the body's output; [B]; workspace
    int P, B; float eps, momentum, weight;
};

__global__ __launch_bounds__(256) void k_vh_conv(const ValueHead A) {                 // 64 positions per block, 4 lanes (32 channels each) per position
    __shared__ float red[2][64];
    const int tid = threadIdx.x, pos = tid >> 2, q = tid & 3, p = blockIdx.x * 64 + pos;
    const float c0 = *A.v1_rm;
    const float4* x = reinterpret_cast<const float4*>(A.body + (size_t)p * 128 + 32 * q);
    const float4* w = reinterpret_cast<const float4*>(A.v1_w + 32 * q);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 xv = x[k], wv = w[k];
        s = fmaf(xv.x, wv.x, s); s = fmaf(xv.y, wv.y, s); s = fmaf(xv.z, wv.z, s); s = fmaf(xv.w, wv.w, s);
    }
    const float s1 = __shfl(s, (tid & ~3) + 1), s2 = __shfl(s, (tid & ~3) + 2), s3 = __shfl(s, (tid & ~3) + 3);
    if (q == 0) {
        const float a = fmaxf((((s + s1) + s2) + s3) + *A.v1_b, 0.0f), d = a - c0;
        A.a_v1[p] = a;
        red[0][pos] = d; red[1][pos] = d * d;
    }
    __syncthreads();
    if (tid < 2) {
        float t = red[tid][0];
        for (int k = 1; k < 64; ++k) t += red[tid][k];
        A.part[blockIdx.x * 2 + tid] = t;
        if (blockIdx.x == 0 && tid == 0) A.part[gridDim.x * 2] = c0;
    }
}

// sums of two per-thread doubles over the workgroup (1 024 threads); every thread gets both totals
__device__ inline void vh_block_sum2(double& a, double& b, double* red) {
    __syncthreads();
    red[threadIdx.x] = a; red[1024 + threadIdx.x] = b;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; red[1024 + threadIdx.x] += red[1024 + threadIdx.x + o]; }
        __syncthreads();
    }
    a = red[0]; b = red[1024];
    __syncthreads();
}

// B <= 128.  Everything between the two convolution-sized passes lives in LDS (dynamic, VH_MLP_LDS bytes): the flattened
// BatchNorm output f [B][64] (later the gradient w.r.t. it), the Dense(64) activations a [B][64] and their gradient d [B][64].
constexpr int VH_MLP_LDS = 2048 * 8 + (3 * 128 * 64 + 64 * 65 + 128 + 6 * 64) * 4;
__global__ __launch_bounds__(1024) void k_vh_mlp(const ValueHead A, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vh_lds[];
    double* red = reinterpret_cast<double*>(vh_lds);               // [2048]
    float* f = reinterpret_cast<float*>(red + 2048);               // [128][64]
    float* a = f + 128 * 64;                                       // [128][64]
    float* d = a + 128 * 64;                                       // [128][64]
    float* w1 = d + 128 * 64;                                      // Dense(64) kernel, rows padded to 65: lane = output unit reads conflict-free
    float* dz2 = w1 + 64 * 65;                                     // [128]
    float (*cst)[64] = reinterpret_cast<float (*)[64]>(dz2 + 128); // per hidden unit: mean, inv_std, gamma, beta, dbeta / B, dgamma / B
    const int tid = threadIdx.x, P = A.P, B = A.B;
    const int j = tid & 63, g = tid >> 6;                          // 64 units x 16 row groups
    for (int i = tid; i < 4096; i += 1024) w1[(i >> 6) * 65 + (i & 63)] = A.f1_w[i];
    if (tid < 64) { cst[2][tid] = A.vbn_g[tid]; cst[3][tid] = A.vbn_beta[tid]; }
    // ---- BatchNorm of the convolution's output: statistics from k_vh_conv's partial sums, apply
    double s1 = 0.0, s2 = 0.0;
    for (int b = tid; b < nblk; b += 1024) { s1 += (double)A.part[2 * b]; s2 += (double)A.part[2 * b + 1]; }
    vh_block_sum2(s1, s2, red);
    const double d1 = s1 / P, mean_d = (double)A.part[2 * nblk] + d1;
    double var = s2 / P - d1 * d1;
    if (var < 0.0) var = 0.0;
    const float mean1 = (float)mean_d, inv1 = (float)(1.0 / sqrt(var + (double)A.eps));
    if (tid == 0) {
        A.stats_v1[0] = mean1; A.stats_v1[1] = inv1;
        *A.v1_rm = (1.0f - A.momentum) * *A.v1_rm + A.momentum * mean1;
        *A.v1_rv = (1.0f - A.momentum) * *A.v1_rv + A.momentum * (float)(var * (double)P / (double)(P > 1 ? P - 1 : 1));
    }
    const float g1 = *A.v1_g, be1 = *A.v1_beta;
    for (int p = tid; p < P; p += 1024) f[p] = g1 * ((A.a_v1[p] - mean1) * inv1) + be1;
    __syncthreads();
    // ---- Dense(64) + bias + ReLU
    {
        const float bj = A.f1_b[j];
        for (int b = g; b < B; b += 16) {
            const float* fb = f + b * 64;
            float s = 0.0f;
#pragma unroll 16
            for (int k = 0; k < 64; ++k) s = fmaf(fb[k], w1[j * 65 + k], s);
            const float av = fmaxf(s + bj, 0.0f);
            a[b * 64 + j] = av; A.a_f1[b * 64 + j] = av;           // kept post-ReLU activations (global copy: for inspection)
        }
    }
    __syncthreads();
    // ---- BatchNorm behind the Dense layer (Keras non-fused: biased moving variance)
    {
        const float c0 = A.vbn_rm[j];
        float s = 0.0f, q = 0.0f;
        for (int b = g; b < B; b += 16) { const float e = a[b * 64 + j] - c0; s += e; q += e * e; }
        red[tid] = s; red[1024 + tid] = q;
        __syncthreads();
        if (tid < 64) {
            double t1 = 0.0, t2 = 0.0;
            for (int k = 0; k < 16; ++k) { t1 += red[k * 64 + tid]; t2 += red[1024 + k * 64 + tid]; }
            const double e1 = t1 / B, m = (double)c0 + e1;
            double v = t2 / B - e1 * e1;
            if (v < 0.0) v = 0.0;
            const float mf = (float)m, inv = (float)(1.0 / sqrt(v + (double)A.eps));
            cst[0][tid] = mf; cst[1][tid] = inv;
            A.stats_vbn[tid] = mf; A.stats_vbn[64 + tid] = inv;
            A.vbn_rm[tid] = (1.0f - A.momentum) * c0 + A.momentum * mf;
            A.vbn_rv[tid] = (1.0f - A.momentum) * A.vbn_rv[tid] + A.momentum * (float)v;
        }
        __syncthreads();
    }
    // ---- Dense(1) + tanh, squared error, gradient w.r.t. the pre-activation: 8 lanes per row, 8 units each, added in order
    //      h = BatchNorm output, recomputed from a where it is needed
    {
        const int b = tid >> 3, q = tid & 7;
        float s = 0.0f;
        if (b < B) {
#pragma unroll
            for (int k = 8 * q; k < 8 * q + 8; ++k) s = fmaf(cst[2][k] * ((a[b * 64 + k] - cst[0][k]) * cst[1][k]) + cst[3][k], A.f2_w[k], s);
        }
        float t = s;                                               // ordered sum of the eight partial dot products
#pragma unroll
        for (int k = 1; k < 8; ++k) { const float o = __shfl(s, (tid & ~7) + k); if (q == 0) t += o; }
        if (b < B && q == 0) {
            const float v = tanhf(t + *A.f2_b), e = v - A.target[b];
            A.se[b] = e * e;
            const float dzb = A.weight * 2.0f * e / (float)B * (1.0f - v * v);
            dz2[b] = dzb; A.dz_f2[b] = dzb;
        }
    }
    __syncthreads();
    // ---- backward: Dense(1); BatchNorm (64 units) statistics
    {
        const float w2 = A.f2_w[j];
        float sw = 0.0f, s = 0.0f, q = 0.0f;
        for (int b = g; b < B; b += 16) {
            const float ah = (a[b * 64 + j] - cst[0][j]) * cst[1][j], h = cst[2][j] * ah + cst[3][j], dh = dz2[b] * w2;
            sw = fmaf(dz2[b], h, sw);
            d[b * 64 + j] = dh;
            s += dh; q += dh * ah;
        }
        red[tid] = s; red[1024 + tid] = q;
        __syncthreads();
        if (tid < 64) {
            double t1 = 0.0, t2 = 0.0;
            for (int k = 0; k < 16; ++k) { t1 += red[k * 64 + tid]; t2 += red[1024 + k * 64 + tid]; }
            const float db = (float)t1, dg = (float)t2;
            A.g_vbn_beta[tid] = db; A.g_vbn_g[tid] = dg;
            cst[4][tid] = db / (float)B; cst[5][tid] = dg / (float)B;
        }
        __syncthreads();
        red[tid] = sw;
        __syncthreads();
        if (tid < 64) {
            double t = 0.0;
            for (int k = 0; k < 16; ++k) t += red[k * 64 + tid];
            A.g_f2_w[tid] = (float)t;
        } else if (tid == 64) {
            double t = 0.0;
            for (int b = 0; b < B; ++b) t += (double)dz2[b];
            *A.g_f2_b = (float)t;
        }
        __syncthreads();
        // ---- BatchNorm + ReLU backward; bias gradient of Dense(64)
        float sb = 0.0f;
        for (int b = g; b < B; b += 16) {
            const int i = b * 64 + j;
            const float av = a[i];
            float e = (cst[2][j] * cst[1][j]) * (d[i] - cst[4][j] - ((av - cst[0][j]) * cst[1][j]) * cst[5][j]);
            if (!(av > 0.0f)) e = 0.0f;
            d[i] = e;
            sb += e;
        }
        red[tid] = sb;
        __syncthreads();
        if (tid < 64) {
            double t = 0.0;
            for (int k = 0; k < 16; ++k) t += red[k * 64 + tid];
            A.g_f1_b[tid] = (float)t;
        }
        __syncthreads();
    }
    // ---- backward: Dense(64): kernel gradient dW1[jj][i] = sum_b dz[b][jj] f[b][i] (4 per thread); then the gradient w.r.t. f, written over f
    {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int b = 0; b < B; ++b) {
            const float fv = f[b * 64 + j];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(d[b * 64 + g + 16 * q], fv, acc[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) A.g_f1_w[(g + 16 * q) * 64 + j] = acc[q];
    }
    __syncthreads();
    for (int b = g; b < B; b += 16) {
        const float* db = d + b * 64;
        float s = 0.0f;
#pragma unroll 16
        for (int k = 0; k < 64; ++k) s = fmaf(db[k], w1[k * 65 + j], s);
        f[b * 64 + j] = s;
    }
    __syncthreads();
    // ---- backward: BatchNorm (1 channel, P values) + ReLU of the convolution; its bias gradient
    double t1 = 0.0, t2 = 0.0;
    for (int p = tid; p < P; p += 1024) { const float e = f[p]; t1 += (double)e; t2 += (double)(e * ((A.a_v1[p] - mean1) * inv1)); }
    vh_block_sum2(t1, t2, red);
    const float dbeta = (float)t1, dgamma = (float)t2;
    if (tid == 0) { *A.g_v1_beta = dbeta; *A.g_v1_g = dgamma; }
    const float gi = g1 * inv1, dbp = dbeta / (float)P, dgp = dgamma / (float)P;
    double sb = 0.0, zero = 0.0;
    for (int p = tid; p < P; p += 1024) {
        const float av = A.a_v1[p];
        float e = gi * (f[p] - dbp - ((av - mean1) * inv1) * dgp);
        if (!(av > 0.0f)) e = 0.0f;
        A.d_v1[p] = e;
        sb += (double)e;
    }
    vh_block_sum2(sb, zero, red);
    if (tid == 0) *A.g_v1_b = (float)sb;
}

// d_body[p][c] = dz[p] w[c]; part[blk][c] = sum over the block's 64 positions of dz[p] body[p][c] (256 threads = 2 row lanes x 128)
__global__ __launch_bounds__(256) void k_vh_conv_bwd(const ValueHead A, float* __restrict__ part) {
    __shared__ float red[256];
    const int c = threadIdx.x & 127, rl = threadIdx.x >> 7;
    const int r0 = blockIdx.x * 64, r1 = min(A.P, r0 + 64);
    const float w = A.v1_w[c];
    const float* __restrict__ body = A.body;
    const float* __restrict__ dv = A.d_v1;
    float* __restrict__ dbody = A.d_body;
    float acc = 0.0f;
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 2) {
        const float dz = dv[r];
        dbody[(size_t)r * 128 + c] = dz * w;
        acc = fmaf(dz, body[(size_t)r * 128 + c], acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) part[(size_t)blockIdx.x * 128 + c] = acc + red[128 + c];
}


// ------------------------------------------------------------------------------------------------ the policy head in few launches
// 1x1 conv (8 kernels) + ReLU + BN -> flatten (H, W, C) -> Dense(512) -> softmax, Keras' clipped cross-entropy, backward
// (training_pipeline.py:93-100).  It sits ON the step's critical path (between the policy conv block's forward and backward
// pass): seven launches there instead of seventeen -- k_ph_conv, k_ph_bn_apply, the logits GEMM (split-K partials left in the
// workspace), k_policy_loss (adds the partials), the GEMM of the gradient w.r.t. the features (on the transposed kernel),
// k_ph_bn_bwd_stats (adds its partials), k_ph_bn_bwd_conv -- and the parameter gradients nobody waits for (Dense kernel and
// bias, 1x1 kernel and bias) on the side stream.  Arithmetic and summation orders of the layer kernels above.
struct PolicyHead {
    const float* x; const float* pi;                              // [P][128] policy conv block output; [B][512] targets
    const float* p2_w; const float* p2_b; const float* p2_g; const float* p2_beta;       // [8][128], [8], [8], [8]
    const float* fc_w; const float* fc_b;                         // [512 out][512 in], [512]
    float* fc_wt;                                                 // [512 in][512 out]: transposed copy (phase 3)
    float* p2_rm; float* p2_rv; float* stats_p2;                  // moving statistics; [2][8]
    float* g_p2_w; float* g_p2_b; float* g_p2_g; float* g_p2_beta; float* g_fc_w; float* g_fc_b;
    float* a_p2; float* out_p2; float* dlogits; float* d_f; float* ce;       // [P][8], [P][8], [B][512], [B][512] = [P][8], [B]
    float* d_x;                                                   // [P][128] gradient w.r.t. x
    float* ws; float* part; float* tall;                          // split-K workspace (4 B 512); 48 P / 64 + 64 floats; 16 P floats
    int P, B; float eps, momentum, weight;
};

__global__ __launch_bounds__(256) void k_ph_conv(const PolicyHead A) {
    __shared__ __attribute__((aligned(16))) float w[8 * 128];
    __shared__ float red[2][8][64];
    const int tid = threadIdx.x, pos = tid >> 2, og = tid & 3, p = blockIdx.x * 64 + pos;
    for (int i = tid; i < 1024; i += 256) w[i] = A.p2_w[i];
    __syncthreads();
    const float4* x = reinterpret_cast<const float4*>(A.x + (size_t)p * 128);
    const float4* w0 = reinterpret_cast<const float4*>(w + (2 * og) * 128);
    const float4* w1 = reinterpret_cast<const float4*>(w + (2 * og + 1) * 128);
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const float4 xv = x[k], a = w0[k], b = w1[k];
        s0 = fmaf(xv.x, a.x, s0); s0 = fmaf(xv.y, a.y, s0); s0 = fmaf(xv.z, a.z, s0); s0 = fmaf(xv.w, a.w, s0);
        s1 = fmaf(xv.x, b.x, s1); s1 = fmaf(xv.y, b.y, s1); s1 = fmaf(xv.z, b.z, s1); s1 = fmaf(xv.w, b.w, s1);
    }
    const int o = 2 * og;
    const float a0 = fmaxf(s0 + A.p2_b[o], 0.0f), a1 = fmaxf(s1 + A.p2_b[o + 1], 0.0f);
    *reinterpret_cast<float2*>(A.a_p2 + (size_t)p * 8 + o) = make_float2(a0, a1);
    const float d0 = a0 - A.p2_rm[o], d1 = a1 - A.p2_rm[o + 1];               // sums shifted by the moving mean
    red[0][o][pos] = d0; red[1][o][pos] = d0 * d0; red[0][o + 1][pos] = d1; red[1][o + 1][pos] = d1 * d1;
    __syncthreads();
    if (tid < 16) {
        const int st = tid >> 3, c = tid & 7;
        float t = red[st][c][0];
        for (int j = 1; j < 64; ++j) t += red[st][c][j];
        A.part[((size_t)blockIdx.x * 2 + st) * 8 + c] = t;
        if (blockIdx.x == 0 && st == 0) A.part[(size_t)gridDim.x * 16 + c] = A.p2_rm[c];
    }
}

__global__ __launch_bounds__(256) void k_ph_bn_apply(const PolicyHead A, int nblk) {
    __shared__ double red[256];
    __shared__ float mi[2][8];
    sum_partials(A.part, nblk, 16, red);
    if (threadIdx.x < 8) {
        const int c = threadIdx.x;
        const double d1 = red[c] / A.P, mean = (double)A.part[(size_t)nblk * 16 + c] + d1;
        double var = red[8 + c] / A.P - d1 * d1;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean, inv = (float)(1.0 / sqrt(var + (double)A.eps));
        mi[0][c] = mf; mi[1][c] = inv;
        if (blockIdx.x == 0) {
            A.stats_p2[c] = mf; A.stats_p2[8 + c] = inv;
            A.p2_rm[c] = (1.0f - A.momentum) * A.p2_rm[c] + A.momentum * mf;
            A.p2_rv[c] = (1.0f - A.momentum) * A.p2_rv[c] + A.momentum * (float)(var * (double)A.P / (double)(A.P > 1 ? A.P - 1 : 1));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const size_t t = (size_t)blockIdx.x * 512 + i;
        const int c = i & 7;
        A.out_p2[t] = A.p2_g[c] * ((A.a_p2[t] - mi[0][c]) * mi[1][c]) + A.p2_beta[c];
    }
}

// k_policy_loss on split-K partial logits: logits = sum of `slices` partials of [B][512] + bias
template <int SLICES>
__global__ __launch_bounds__(256) void k_policy_loss_s(const float* __restrict__ parts, const float* __restrict__ bias,
                                                       const float* __restrict__ pi, int B, float weight,
                                                       float* __restrict__ dlogits, float* __restrict__ ce_out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const size_t n = (size_t)B * 512;
    float z[8], t[8];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        float l = parts[(size_t)b * 512 + i];
#pragma unroll
        for (int q = 1; q < SLICES; ++q) l += parts[q * n + (size_t)b * 512 + i];
        z[j] = l + bias[i];
        t[j] = pi[(size_t)b * 512 + i];
        mx = fmaxf(mx, z[j]);
    }
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { z[j] = expf(z[j] - mx); sum += z[j]; }
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    float ce = 0.0f, tu = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        z[j] = z[j] / sum;
        const bool u = z[j] > 1e-7f && z[j] < 1.0f - 1e-7f;
        const float pc = fminf(fmaxf(z[j], 1e-7f), 1.0f - 1e-7f);
        ce -= t[j] * logf(pc);
        tu += u ? t[j] : 0.0f;
        t[j] = u ? t[j] : 0.0f;
    }
    for (int d = 32; d >= 1; d >>= 1) { ce += __shfl_xor(ce, d); tu += __shfl_xor(tu, d); }
    const float sc = weight / (float)B;
#pragma unroll
    for (int j = 0; j < 8; ++j) dlogits[(size_t)b * 512 + lane + 64 * j] = sc * (z[j] * tu - t[j]);
    if (lane == 0) ce_out[b] = ce;
}

// d_f = sum of the split-K partials of dlogits . W (kept: the gradient w.r.t. the BatchNorm output, [P][8]); partial sums of
// d_f and d_f * ahat per channel: partB[blk][2][8]
template <int SLICES>
__global__ __launch_bounds__(256) void k_ph_bn_bwd_stats(const PolicyHead A, float* __restrict__ partB) {
    __shared__ float red[2][8][64];
    const size_t n = (size_t)A.P * 8;
    for (int i = threadIdx.x; i < 512; i += 256) {
        const size_t t = (size_t)blockIdx.x * 512 + i;
        float d = A.ws[t];
#pragma unroll
        for (int q = 1; q < SLICES; ++q) d += A.ws[q * n + t];
        A.d_f[t] = d;
        const int c = i & 7, pos = i >> 3;
        red[0][c][pos] = d; red[1][c][pos] = d * ((A.a_p2[t] - A.stats_p2[c]) * A.stats_p2[8 + c]);
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int st = threadIdx.x >> 3, c = threadIdx.x & 7;
        float t = red[st][c][0];
        for (int j = 1; j < 64; ++j) t += red[st][c][j];
        partB[((size_t)blockIdx.x * 2 + st) * 8 + c] = t;
    }
}

// BatchNorm + ReLU backward (sums finished in the prologue), then, with the block's dz[64][8] in LDS: d_x[p][c] = sum_o dz[p][o] W[o][c],
// partial kernel gradient tall[blk][o][c] = sum_p dz[p][o] x[p][c], partial bias gradient part2[blk][o] = sum_p dz[p][o]
__global__ __launch_bounds__(256) void k_ph_bn_bwd_conv(const PolicyHead A, int nblk, const float* __restrict__ partB, float* __restrict__ part2) {
    __shared__ double red[256];
    __shared__ float cs[4][8];                                    // mean, g * inv, dbeta / P, dgamma / P
    __shared__ float dz[64][8];
    __shared__ float w[8][128];
    __shared__ float acc2[8][128];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 256) w[i >> 7][i & 127] = A.p2_w[i];
    sum_partials(partB, nblk, 16, red);
    if (tid < 8) {
        const float db = (float)red[tid], dg = (float)red[8 + tid];
        if (blockIdx.x == 0) { A.g_p2_beta[tid] = db; A.g_p2_g[tid] = dg; }
        cs[0][tid] = A.stats_p2[tid]; cs[1][tid] = A.stats_p2[8 + tid];
        cs[2][tid] = db / (float)A.P; cs[3][tid] = dg / (float)A.P;
    }
    __syncthreads();
    for (int i = tid; i < 512; i += 256) {
        const size_t t = (size_t)blockIdx.x * 512 + i;
        const int c = i & 7;
        const float av = A.a_p2[t], inv = cs[1][c];
        float d = (A.p2_g[c] * inv) * (A.d_f[t] - cs[2][c] - ((av - cs[0][c]) * inv) * cs[3][c]);
        if (!(av > 0.0f)) d = 0.0f;
        A.d_f[t] = d;
        dz[i >> 3][c] = d;
    }
    __syncthreads();
    if (tid < 8) {
        float t = 0.0f;
        for (int j = 0; j < 64; ++j) t += dz[j][tid];
        part2[(size_t)blockIdx.x * 8 + tid] = t;
    }
    const int c = tid & 127, rl = tid >> 7;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.0f;
    float wc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) wc[o] = w[o][c];
    const float* __restrict__ xin = A.x + (size_t)blockIdx.x * 64 * 128 + c;
    float* __restrict__ dx = A.d_x + (size_t)blockIdx.x * 64 * 128 + c;
    for (int r0 = rl; r0 < 64; r0 += 16) {                         // 8 rows per pass: the loads first
        float xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = xin[(size_t)(r0 + 2 * u) * 128];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 2 * u;
            float s = 0.0f;
#pragma unroll
            for (int o = 0; o < 8; ++o) { const float d = dz[r][o]; s = fmaf(d, wc[o], s); acc[o] = fmaf(d, xv[u], acc[o]); }
            dx[(size_t)r * 128] = s;
        }
    }
    if (rl == 1) {
#pragma unroll
        for (int o = 0; o < 8; ++o) acc2[o][c] = acc[o];
    }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int o = 0; o < 8; ++o) A.tall[((size_t)blockIdx.x * 8 + o) * 128 + c] = acc[o] + acc2[o][c];
    }
}

__global__ __launch_bounds__(256) void k_transpose512(const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x & 15, by = blockIdx.x >> 4, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) t[j][tx] = in[(size_t)(by * 32 + j) * 512 + bx * 32 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8) out[(size_t)(bx * 32 + j) * 512 + by * 32 + tx] = t[tx][j];
}


}  // namespace ckrt

using namespace ckrt;

#define LAUNCH1D(kernel, n, st, ...) hipLaunchKernelGGL(kernel, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, (hipStream_t)(st), __VA_ARGS__)

extern "C" {

static int rows_per_block(int P) { return P <= 16384 ? 128 : 128 * ((P + 16383) / 16384); }   // <= 128 partials per reduction

int ckr_gemm_nt(const float* A, int32_t lda, const float* Bt, int32_t ldb, float* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                int32_t slices, float* workspace, const float* add, void* stream) {
    if (!A || !Bt || (!C && slices < 2) || M <= 0 || N <= 0 || K <= 0 || M % BM || N % BN || slices < 1 || K % (BK * slices) || (ldc % 4) || (lda % 4) || (ldb % 4))
        return ckr::fail(CKR_ERR_INVALID, "ckr_gemm_nt: M, N multiples of 128, K a multiple of 32 * slices, leading dimensions of 4");
    if (slices > 1 && (!workspace || ldc != N)) return ckr::fail(CKR_ERR_INVALID, "ckr_gemm_nt: split-K needs a workspace and ldc == N");
    if (int rc = ckr::require_device()) return rc;
    float* dst = slices > 1 ? workspace : C;
    hipLaunchKernelGGL(k_gemm_nt<0>, dim3(N / BN, M / BM, slices), dim3(GT), 0, (hipStream_t)stream, A, (int)lda, Bt, (int)ldb, dst, (int)ldc, (int)M, (int)K);
    if (C && (slices > 1 || add)) {                                 // C == NULL: the caller's next kernel adds the slices
        const long long n4 = (long long)M * N / 4;
        if (slices == 1) {                                        // C = C + add
            hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const float4*)C, 1, n4, (const float4*)add, (float4*)C);
        } else {
            hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const float4*)workspace, (int)slices, n4, (const float4*)add, (float4*)C);
        }
    }
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// The GEMM of a 3x3 convolution with 128 input and 128 output planes on activations act[P][128] (P = 64 * boards):
// direction +1: workspace[z][p][o] = sum_{k in slice z} act[p + off(tap)][c] * w[o][tap * 128 + c]      (forward; w = the kernel)
// direction -1: workspace[z][p][c] = sum_{k in slice z} act[p - off(tap)][o] * w[c][tap * 128 + o]      (data gradient; w = ckr_conv_wflip's)
// The caller's next kernel (ckr_conv_bias_relu_bn / ckr_conv_bn_relu_backward) adds the `slices` partial products.
int ckr_conv_gemm(const float* act, const float* w, int32_t P, int32_t direction, int32_t slices, int32_t pipe, float* workspace, void* stream) {
    if (!act || !w || !workspace || P <= 0 || P % 128 || (direction != 1 && direction != -1) || slices < 1 || 1152 % (BK * slices))
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_gemm: P must be a multiple of 128, direction +1 or -1, slices a divisor of 36");
    if (int rc = ckr::require_device()) return rc;
    if (pipe != 0 && pipe != 1) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_gemm: pipe must be 0 (float32 MFMA) or 1 (split bfloat16 x 6)");
    const dim3 grid(1, P / BM, slices);
    if (pipe == 0) {
        if (direction > 0) hipLaunchKernelGGL(k_gemm_nt<1>, grid, dim3(GT), 0, (hipStream_t)stream, act, 128, w, 1152, workspace, 128, (int)P, 1152);
        else hipLaunchKernelGGL(k_gemm_nt<-1>, grid, dim3(GT), 0, (hipStream_t)stream, act, 128, w, 1152, workspace, 128, (int)P, 1152);
    } else {
        if (direction > 0) hipLaunchKernelGGL(k_gemm_nt6<1>, grid, dim3(GT), 0, (hipStream_t)stream, act, 128, w, 1152, workspace, 128, (int)P, 1152);
        else hipLaunchKernelGGL(k_gemm_nt6<-1>, grid, dim3(GT), 0, (hipStream_t)stream, act, 128, w, 1152, workspace, 128, (int)P, 1152);
    }
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// ckr_conv_gemm on operands that were split into their three bfloat16 pieces where they were produced (k_gemm_p6): act3 = the
// pieces of an activation [P + 1][128] whose row P is all zero (768 bytes per row; written by ckr_conv_bias_relu_bn /
// ckr_conv_bn_relu_backward, or ckr_split_pieces), w3 = the pieces of the kernel rows [128][1152] (ckr_conv_wsplit).
int ckr_conv_gemm_pieces(const void* act3, const void* w3, int32_t P, int32_t direction, int32_t slices, float* workspace, void* stream) {
    if (!act3 || !w3 || !workspace || P <= 0 || P % 128 || (direction != 1 && direction != -1) || slices < 1 || 1152 % (BK * slices))
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_gemm_pieces: P must be a multiple of 128, direction +1 or -1, slices a divisor of 36");
    if (int rc = ckr::require_device()) return rc;
    // Kernel by batch size (profiles/r04_train_gemm_presplit.txt): the double-buffered kernels win where their grid is one full
    // round of workgroups -- 128 x 128 tiles (one workgroup of four waves per CU) for batches up to 256 boards with split-K,
    // 256 x 128 tiles of eight waves from 1 024 boards on; in between the single-buffered kernel keeps two workgroups per CU.
    // (-DCKR_EXPERIMENTS builds: CKR_P6_VARIANT = 0 | 2 | 4 forces one.)
    int variant = P <= 16384 ? 2 : (P >= 65536 && P % 256 == 0) ? 4 : 0;
#ifdef CKR_EXPERIMENTS
    if (const char* v = getenv("CKR_P6_VARIANT")) variant = atoi(v);
    if (variant == 4 && P % 256) variant = 2;
#endif
    static bool lds_set = false;
    if (!lds_set) {
        CKR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_p6d<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, gemm_p6d_lds<2>()));
        CKR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_p6d<-1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, gemm_p6d_lds<2>()));
        CKR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_p6d<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, gemm_p6d_lds<4>()));
        CKR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_p6d<-1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, gemm_p6d_lds<4>()));
        lds_set = true;
    }
    const uint4* a3 = (const uint4*)act3;
    const uint4* b3 = (const uint4*)w3;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 2) {
        const dim3 grid(1, P / 128, slices);
        if (direction > 0) hipLaunchKernelGGL((k_gemm_p6d<1, 2>), grid, dim3(256), gemm_p6d_lds<2>(), st, a3, b3, workspace, 128, (int)P, 1152, (int)P);
        else hipLaunchKernelGGL((k_gemm_p6d<-1, 2>), grid, dim3(256), gemm_p6d_lds<2>(), st, a3, b3, workspace, 128, (int)P, 1152, (int)P);
    } else if (variant == 4) {
        const dim3 grid(1, P / 256, slices);
        if (direction > 0) hipLaunchKernelGGL((k_gemm_p6d<1, 4>), grid, dim3(512), gemm_p6d_lds<4>(), st, a3, b3, workspace, 128, (int)P, 1152, (int)P);
        else hipLaunchKernelGGL((k_gemm_p6d<-1, 4>), grid, dim3(512), gemm_p6d_lds<4>(), st, a3, b3, workspace, 128, (int)P, 1152, (int)P);
    } else {
        const dim3 grid(1, P / BM, slices);
        if (direction > 0) hipLaunchKernelGGL(k_gemm_p6<1>, grid, dim3(GT), 0, st, a3, b3, workspace, 128, (int)P, 1152, (int)P);
        else hipLaunchKernelGGL(k_gemm_p6<-1>, grid, dim3(GT), 0, st, a3, b3, workspace, 128, (int)P, 1152, (int)P);
    }
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// The pieces of x[rows][cols] float32 (cols a multiple of 32): out3[rows][cols / 32][3][32] bfloat16.
int ckr_split_pieces(const float* x, int64_t rows, int32_t cols, void* out3, void* stream) {
    if (!x || !out3 || rows <= 0 || cols <= 0 || cols % 32) return ckr::fail(CKR_ERR_INVALID, "ckr_split_pieces: cols must be a multiple of 32");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_split_rows, rows * (cols / 4), stream, x, (long long)rows, (int)cols, out3);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// w3[l] = the pieces of the kernel of layer l ([128][1152] float32 at w + offsets[l]), wt3[l] = the pieces of its flipped copy
// (ckr_conv_wflip's layout); either may be NULL.  l < layers <= 8.
int ckr_conv_wsplit(const float* w, const int64_t* offsets, int32_t layers, void* w3, void* wt3, void* stream) {
    if (!w || !offsets || (!w3 && !wt3) || layers < 1 || layers > 8) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_wsplit: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LayerOffsets lo;
    for (int l = 0; l < 8; ++l) lo.off[l] = l < layers ? offsets[l] : 0;
    hipLaunchKernelGGL(k_wsplit, dim3(144, layers, 2), dim3(256), 0, (hipStream_t)stream, w, lo, w3, wt3);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// Weight gradient: dw[o][tap * 128 + c] (taps = 9, ld = 1152) = sum_p dz[p][o] * x[p + off(tap)][c], or with taps = 1 the plain
// product dw[o][c] = sum_p dz[p][o] * x[p][c] (ld = 128; the first layer on its im2col matrix).  Split over `slices` ranges of
// positions (P % (32 * slices) == 0), reduced deterministically through workspace[slices][128][ld].
int ckr_conv_wgrad(const float* dz, const float* x, int32_t P, int32_t taps, int32_t slices, int32_t pipe, float* workspace, float* dw, void* stream) {
    if (!dz || !x || !dw || !workspace || P <= 0 || (taps != 9 && taps != 1) || slices < 1 || P % BK || slices > P / BK)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_wgrad: taps 9 or 1, P a multiple of 32, 1 <= slices <= P / 32");
    if (int rc = ckr::require_device()) return rc;
    const int ld = 128 * taps;
    if (pipe != 0 && pipe != 1) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_wgrad: pipe must be 0 (float32 MFMA) or 1 (split bfloat16 x 6)");
    if (pipe == 0) hipLaunchKernelGGL(k_wgrad_tn, dim3(taps, 1, slices), dim3(GT), 0, (hipStream_t)stream, dz, x, (int)P, taps == 9 ? 0 : 4, workspace, ld);
    else hipLaunchKernelGGL(k_wgrad_tn6, dim3(taps, 1, slices), dim3(GT), 0, (hipStream_t)stream, dz, x, (int)P, taps == 9 ? 0 : 4, workspace, ld);
    const long long n4 = 128LL * ld / 4;
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const float4*)workspace, (int)slices, n4, (const float4*)nullptr, (float4*)dw);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// wt[l][c][tap * 128 + o] = w[offsets[l] + o * 1152 + tap * 128 + c], l < layers <= 8 (offsets in floats, host array)
int ckr_conv_wflip(const float* w, const int64_t* offsets, int32_t layers, float* wt, void* stream) {
    if (!w || !offsets || !wt || layers < 1 || layers > 8) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_wflip: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LayerOffsets lo;
    for (int l = 0; l < 8; ++l) lo.off[l] = l < layers ? offsets[l] : 0;
    hipLaunchKernelGGL(k_wflip, dim3(147456 / 256, layers), dim3(256), 0, (hipStream_t)stream, w, lo, wt);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// Forward half of a conv block after its GEMM: a = ReLU(sum of the workspace slices + bias) (kept for the backward pass),
// batch statistics -> stats[2][128] (mean, 1 / sqrt(var + eps)), moving statistics updated, out = BatchNorm(a).
// part: workspace of 256 * ceil(P / 128) + 128 floats.  The slices may alias a (slices == 1, workspace == a).
int ckr_conv_bias_relu_bn(const float* workspace, int32_t slices, const float* bias, int32_t P, const float* gamma, const float* beta, float eps,
                          float momentum, float* run_mean, float* run_var, float* stats, float* a, float* out, float* part, void* out_pieces,
                          void* stream) {
    if (!workspace || slices < 1 || !bias || !gamma || !beta || !run_mean || !run_var || !stats || !a || !out || !part || P <= 0)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_bias_relu_bn: bad argument");
    if (int rc = ckr::require_device()) return rc;
    const int rpb = rows_per_block(P), nblk = (P + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_fwd_reduce128, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, workspace, (int)slices, bias, (int)P, rpb, a, part, (const float*)run_mean);
    hipLaunchKernelGGL(k_bn_apply128, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, (const float*)a, (const float*)part, nblk, (int)P, rpb, gamma, beta,
                       eps, momentum, run_mean, run_var, stats, out, out_pieces);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// Backward half of a conv block before its GEMMs.  dout = sum of the workspace slices (the data-gradient GEMM of the block
// above; slices == 0: dout is given) + add (optional: a second consumer's gradient); then dz = d loss / d (conv output) written
// over dout, dgamma, dbeta, dbias.  a, stats: the block's kept activation and statistics.  part: 384 * ceil(P / 128) floats.
int ckr_conv_bn_relu_backward(const float* workspace, int32_t slices, const float* add, float* dout, const float* a, const float* stats,
                              const float* gamma, int32_t P, float* dgamma, float* dbeta, float* dbias, float* part, void* dz_pieces, void* stream) {
    if ((slices > 0 && !workspace) || slices < 0 || !dout || !a || !stats || !gamma || !dgamma || !dbeta || !part || P <= 0)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_bn_relu_backward: bad argument");
    if (int rc = ckr::require_device()) return rc;
    const int rpb = rows_per_block(P), nblk = (P + rpb - 1) / rpb;
    float* part2 = part + (size_t)256 * nblk;
    hipLaunchKernelGGL(k_bwd_reduce128, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, workspace, (int)slices, add, a, stats, (int)P, rpb, dout, part);
    hipLaunchKernelGGL(k_bn_bwd_apply128, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, dout, a, stats, (const float*)part, nblk, gamma, (int)P, rpb,
                       dgamma, dbeta, part2, dz_pieces);
    if (dbias) hipLaunchKernelGGL(k_sum_rows, dim3(2), dim3(256), 0, (hipStream_t)stream, (const float*)part2, nblk, 128, dbias);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// The conv bias gradient from the partial sums ckr_conv_bn_relu_backward (called with dbias == NULL) left in `part`:
// a separate call so that it can run on another stream, off the critical path of the backward chain.
int ckr_conv_bias_grad(const float* part, int32_t P, float* dbias, void* stream) {
    if (!part || !dbias || P <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_bias_grad: bad argument");
    if (int rc = ckr::require_device()) return rc;
    const int rpb = rows_per_block(P), nblk = (P + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_sum_rows, dim3(2), dim3(256), 0, (hipStream_t)stream, part + (size_t)256 * nblk, nblk, 128, dbias);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_gemm_small(const float* A, int64_t am, int64_t ak, const float* B, int64_t bk, int64_t bn, float* C, int64_t ldc,
                   int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_gemm_small: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_gemm_small, (long long)M * N, stream, A, (long long)am, (long long)ak, B, (long long)bk, (long long)bn, C, (long long)ldc, (int)M, (int)N, (int)K, (int)accumulate);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// C[m][n] = sum_p A[p][m] * B[p][n]: the weight gradients of the heads' 1x1 convolutions (M = 8 or 1 kernels, N = 128 planes,
// P positions).  part: workspace of M * N * ceil(P / 64) floats.
int ckr_gemm_tall(const float* A, const float* B, int32_t P, int32_t M, int32_t N, float* C, float* part, void* stream) {
    if (!A || !B || !C || !part || P <= 0 || (M != 1 && M != 8) || N < 1 || N > 256 || 256 % N) return ckr::fail(CKR_ERR_INVALID, "ckr_gemm_tall: M 1 or 8, N | 256");
    if (int rc = ckr::require_device()) return rc;
    const int nblk = (P + TALL_ROWS - 1) / TALL_ROWS;
    if (M == 1) hipLaunchKernelGGL(k_tall_tn<1>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, A, B, (int)P, (int)N, part);
    else hipLaunchKernelGGL(k_tall_tn<8>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, A, B, (int)P, (int)N, part);
    hipLaunchKernelGGL(k_sum_rows, dim3((M * N + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float*)part, nblk, (int)(M * N), C);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_im2col(const float* x, int32_t P, int32_t cin, int32_t kpad, float* col, void* stream) {
    if (!x || !col || P <= 0 || P % 64 || cin <= 0 || kpad < 9 * cin) return ckr::fail(CKR_ERR_INVALID, "ckr_im2col: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_im2col, (long long)P * kpad, stream, x, (int)P, (int)cin, (int)kpad, col);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

static bool chan_ok(int Cc) { return Cc >= 1 && Cc <= 128 && (128 % Cc) == 0; }

// a = act(z + bias) in place, batch statistics -> stats[2][C], moving statistics updated, out = BatchNorm(a).
// part: workspace of 2 * C * ceil(P / 64) + C floats.  (The heads' small layers; the conv blocks use ckr_conv_bias_relu_bn.)
int ckr_bn_forward(float* z, const float* bias, int32_t P, int32_t Cc, int32_t relu, const float* gamma, const float* beta, float eps,
                   float momentum, float* run_mean, float* run_var, float* stats, float* out, float* part, int32_t biased_moving_var,
                   void* stream) {
    if (!z || !gamma || !beta || !stats || !out || !part || P <= 0 || !chan_ok(Cc)) return ckr::fail(CKR_ERR_INVALID, "ckr_bn_forward: bad argument");
    if (int rc = ckr::require_device()) return rc;
    const int nblk = (P + ROWS - 1) / ROWS;
    hipLaunchKernelGGL(k_bias_relu_stats, dim3(nblk), dim3(256), 0, (hipStream_t)stream, z, bias, (int)P, (int)Cc, (int)relu, part, (const float*)run_mean);
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)part, nblk, (int)P, (int)Cc, eps, momentum, stats, run_mean, run_var,
                       (int)biased_moving_var);
    LAUNCH1D(k_bn_apply, (long long)P * Cc, stream, (const float*)z, (const float*)stats, gamma, beta, (long long)P * Cc, (int)Cc, out);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// dout (gradient w.r.t. the BatchNorm output) -> dz (gradient w.r.t. the pre-activation z) in place; dgamma, dbeta, dbias.
int ckr_bn_backward(float* dout, const float* a, const float* stats, const float* gamma, int32_t P, int32_t Cc, int32_t relu,
                    float* dgamma, float* dbeta, float* dbias, float* part, float* sums, void* stream) {
    if (!dout || !a || !stats || !gamma || !dgamma || !dbeta || !part || !sums || P <= 0 || !chan_ok(Cc))
        return ckr::fail(CKR_ERR_INVALID, "ckr_bn_backward: bad argument");
    if (int rc = ckr::require_device()) return rc;
    const int nblk = (P + ROWS - 1) / ROWS;
    hipLaunchKernelGGL(k_bn_bwd_stats, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)dout, a, stats, (int)P, (int)Cc, part);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)part, nblk, (int)Cc, sums, dgamma, dbeta);
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dout, a, stats, (const float*)sums, gamma, (int)P, (int)Cc, (int)relu, part);
    if (dbias) hipLaunchKernelGGL(k_sum_rows, dim3((Cc + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float*)part, nblk, (int)Cc, dbias);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_policy_loss(const float* logits, const float* bias, const float* pi, int32_t B, float weight, float* dlogits, float* ce, void* stream) {
    if (!logits || !bias || !pi || !dlogits || !ce || B <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_policy_loss: bad argument");
    if (int rc = ckr::require_device()) return rc;
    hipLaunchKernelGGL(k_policy_loss, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, bias, pi, (int)B, weight, dlogits, ce);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_value_loss(const float* z, const float* bias, const float* target, int32_t B, float weight, float* dz, float* se, void* stream) {
    if (!z || !bias || !target || !dz || !se || B <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_value_loss: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_value_loss, B, stream, z, bias, target, (int)B, weight, dz, se);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// penalty_parts: the 512 partial sums ckr_adam_step left (or null)
int ckr_loss_sums(const float* ce, const float* se, int32_t B, float wp, float wv, const double* penalty_parts, double n_rows, double* acc, void* stream) {
    if (!ce || !se || !acc || B <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_loss_sums: bad argument");
    if (int rc = ckr::require_device()) return rc;
    hipLaunchKernelGGL(k_loss_sums, dim3(1), dim3(256), 0, (hipStream_t)stream, ce, se, (int)B, wp, wv, penalty_parts, n_rows, acc);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// d_penalty_parts: 512 doubles (or null): the l2 penalty of the weights before this update, in 512 partial sums.
// d_step: 2 floats -- the step counter and a scratch word that is 0 between calls.  losses (may be null): the sums of
// ckr_loss_sums, added by the same launch.
int ckr_adam_step(float* w, const float* grad, float* m, float* v, const float* reg, int64_t n, const float* d_lr, float beta1, float beta2,
                  float eps, float* d_step, double* d_penalty_parts, const ckr_loss_args* losses, void* stream) {
    if (!w || !grad || !m || !v || !reg || !d_lr || !d_step || n <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_adam_step: bad argument");
    if (losses && (!losses->ce || !losses->se || !losses->acc || losses->B <= 0)) return ckr::fail(CKR_ERR_INVALID, "ckr_adam_step: bad loss arguments");
    if (int rc = ckr::require_device()) return rc;
    static_assert(sizeof(ckr_loss_args) == sizeof(LossArgs), "ckr_loss_args mirrors LossArgs");
    LossArgs la;
    memset(&la, 0, sizeof(la));
    if (losses) memcpy(&la, losses, sizeof(la));
    hipLaunchKernelGGL(k_adam, dim3(ADAM_BLOCKS), dim3(256), 0, (hipStream_t)stream, w, grad, m, v, reg, (long long)n, d_lr, beta1, beta2, eps,
                       d_step, d_penalty_parts, la);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_value_head_step(const ckr_value_head* h, void* stream) {
    if (!h) return ckr::fail(CKR_ERR_INVALID, "ckr_value_head_step: null argument");
    const void* const* ptrs = reinterpret_cast<const void* const*>(h);
    for (int i = 0; i < 38; ++i)
        if (!ptrs[i]) return ckr::fail(CKR_ERR_INVALID, "ckr_value_head_step: null pointer (field %d)", i);
    if (h->B <= 0 || h->B > 128 || h->P != 64 * h->B) return ckr::fail(CKR_ERR_INVALID, "ckr_value_head_step: 1 <= B <= 128, P = 64 B");
    if (int rc = ckr::require_device()) return rc;
    static_assert(sizeof(ckr_value_head) == sizeof(ValueHead), "ckr_value_head mirrors ValueHead");
    ValueHead A;
    memcpy(&A, h, sizeof(A));
    const int nblk = h->P / 64, nb2 = nblk;
    float* part2 = h->part + 2 * nblk + 8;
    static bool lds_set = false;                                  // one workgroup keeps three [128][64] float arrays in LDS
    if (!lds_set) {
        CKR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_vh_mlp), hipFuncAttributeMaxDynamicSharedMemorySize, VH_MLP_LDS));
        lds_set = true;
    }
    hipLaunchKernelGGL(k_vh_conv, dim3(nblk), dim3(256), 0, (hipStream_t)stream, A);
    hipLaunchKernelGGL(k_vh_mlp, dim3(1), dim3(1024), VH_MLP_LDS, (hipStream_t)stream, A, nblk);
    hipLaunchKernelGGL(k_vh_conv_bwd, dim3(nb2), dim3(256), 0, (hipStream_t)stream, A, part2);
    hipLaunchKernelGGL(k_sum_rows, dim3(2), dim3(256), 0, (hipStream_t)stream, (const float*)part2, nb2, 128, h->g_v1_w);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_policy_head_step(const ckr_train_policy_head* h, int32_t phase, void* stream) {
    if (!h) return ckr::fail(CKR_ERR_INVALID, "ckr_policy_head_step: null argument");
    const void* const* ptrs = reinterpret_cast<const void* const*>(h);
    for (int i = 0; i < 27; ++i)
        if (!ptrs[i]) return ckr::fail(CKR_ERR_INVALID, "ckr_policy_head_step: null pointer (field %d)", i);
    if (h->B <= 0 || h->B % 128 || h->P != 64 * h->B) return ckr::fail(CKR_ERR_INVALID, "ckr_policy_head_step: B a multiple of 128, P = 64 B");
    if (int rc = ckr::require_device()) return rc;
    static_assert(sizeof(ckr_train_policy_head) == sizeof(PolicyHead), "ckr_train_policy_head mirrors PolicyHead");
    PolicyHead A;
    memcpy(&A, h, sizeof(A));
    hipStream_t st = (hipStream_t)stream;
    constexpr int SL = 4;
    const int nblk = h->P / 64;
    float* partB = h->part + (size_t)nblk * 16 + 16;
    float* part2 = partB + (size_t)nblk * 16;
    if (phase == 3) {                                             // the Dense kernel transposed (it is fixed during a step)
        hipLaunchKernelGGL(k_transpose512, dim3(256), dim3(256), 0, st, h->fc_w, h->fc_wt);
    } else if (phase == 0) {                                      // forward + loss
        hipLaunchKernelGGL(k_ph_conv, dim3(nblk), dim3(256), 0, st, A);
        hipLaunchKernelGGL(k_ph_bn_apply, dim3(nblk), dim3(256), 0, st, A, nblk);
        if (int rc = ckr_gemm_nt(h->out_p2, 512, h->fc_w, 512, nullptr, 512, h->B, 512, 512, SL, h->ws, nullptr, stream)) return rc;
        hipLaunchKernelGGL(k_policy_loss_s<SL>, dim3((h->B + 3) / 4), dim3(256), 0, st, (const float*)h->ws, h->fc_b, h->pi, (int)h->B, h->weight,
                           h->dlogits, h->ce);
    } else if (phase == 1) {                                      // backward, critical path: down to the gradient w.r.t. x
        if (int rc = ckr_gemm_nt(h->dlogits, 512, h->fc_wt, 512, nullptr, 512, h->B, 512, 512, SL, h->ws, nullptr, stream)) return rc;
        hipLaunchKernelGGL(k_ph_bn_bwd_stats<SL>, dim3(nblk), dim3(256), 0, st, A, partB);
        hipLaunchKernelGGL(k_ph_bn_bwd_conv, dim3(nblk), dim3(256), 0, st, A, nblk, (const float*)partB, part2);
    } else if (phase == 2) {                                      // parameter gradients (off the critical path)
        hipLaunchKernelGGL(k_sum_rows, dim3(8), dim3(256), 0, st, (const float*)h->dlogits, (int)h->B, 512, h->g_fc_b);
        LAUNCH1D(k_gemm_small, 512ll * 512, stream, (const float*)h->dlogits, 1ll, 512ll, (const float*)h->out_p2, 512ll, 1ll, h->g_fc_w, 512ll, 512, 512, (int)h->B, 0);
        hipLaunchKernelGGL(k_sum_rows, dim3(16), dim3(256), 0, st, (const float*)h->tall, nblk, 1024, h->g_p2_w);
        hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, st, (const float*)part2, nblk, 8, h->g_p2_b);
    } else {
        return ckr::fail(CKR_ERR_INVALID, "ckr_policy_head_step: phase must be 0 .. 3");
    }
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_sum_rows(const float* in, int32_t rows, int32_t cols, float* out, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_sum_rows: bad argument");
    if (int rc = ckr::require_device()) return rc;
    hipLaunchKernelGGL(k_sum_rows, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, (int)rows, (int)cols, out);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

}  // extern "C"
