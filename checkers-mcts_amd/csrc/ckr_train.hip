// ckr_train.hip -- the training step of the reference's network in hand-written HIP.
//
// Reference: train_nn (training_pipeline.py:123-179) fits create_nn's model (:59-114) with Keras: float32
// arithmetic, loss = w_p * categorical cross-entropy + w_v * MSE + l2 penalties, Adam.  This file holds the
// device side of one optimisation step on a batch of B boards (P = 64 B positions, activations [P][C] float32,
// channels last): forward in training mode (BatchNormalization on batch statistics), backward, Adam.
// The host side (train_hip.py) owns the buffers and the order of the launches.
//
//   * the 3x3 convolutions are GEMMs on an explicit im2col matrix (k = tap * Cin + c): forward
//     Z = COL . W^T, data gradient dCOL = dZ . W, weight gradient dW = dZ^T . COL -- all three through ONE
//     "NT" GEMM kernel (C = A . Bt^T, both operands K-contiguous) on the float32 matrix pipe
//     (v_mfma_f32_32x32x2_f32: exact float32 products, float32 accumulation, the arithmetic Keras uses);
//     128 x 128 tiles, K chunks of 32 staged through LDS, optional split-K with a deterministic reduction;
//   * everything else (bias + ReLU + BatchNorm statistics / apply / backward, 1x1 convolutions and the
//     dense layers of the two heads, losses, Adam with the l2 terms) is bandwidth- or latency-bound
//     elementwise / reduction work in plain float32.
#include "ckr_host.h"
#include <hip/hip_runtime.h>

namespace ckrt {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ------------------------------------------------------------------------------------------------ GEMM (NT)
constexpr int BM = 128, BN = 128, BK = 32, GT = 256;
constexpr int PITCH = BK + 4;                                     // floats per LDS row: 144 B, conflict-free b128 reads

// C[M][N] (ldc) = sum_k A[m][k] * Bt[n][k]; M % 128 == 0, N % 128 == 0, K % (32 * slices) == 0.
// gridDim = (N / 128, M / 128, slices); slice z covers k in [z * K / slices, (z + 1) * K / slices) and writes
// C + z * M * ldc (the caller reduces the slices).
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
    float4 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + GT * i, row = idx >> 3, c4 = idx & 7;
            ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * lda + k0 + 4 * c4);
            rb[i] = *reinterpret_cast<const float4*>(Bt + (size_t)(n0 + row) * ldb + k0 + 4 * c4);
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kbeg + kper; k0 += BK) {
        __syncthreads();                                          // the previous chunk's fragments have been read
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + GT * i, row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<float4*>(As + row * PITCH + 4 * c4) = ra[i];
            *reinterpret_cast<float4*>(Bs + row * PITCH + 4 * c4) = rb[i];
        }
        __syncthreads();
        if (k0 + BK < kbeg + kper) fetch(k0 + BK);                // next chunk in flight under the MFMAs
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
}

// out[i] = sum_z part[z][i] (+ add[i]); float4 granularity
__global__ void k_sum_slices(const float4* __restrict__ part, int slices, long long n4, const float4* __restrict__ add, float4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = add ? add[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < slices; ++z) { const float4 v = part[(size_t)z * n4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    out[i] = s;
}

// Small matrices (1x1 convolutions with 8 / 1 kernels, the heads' dense layers, their gradients):
// C[m][n] = sum_k A[m * am + k * ak] * B[k * bk + n * bn] (+ C if accumulate); one thread per output, float32 FMAs
// in k order.
__global__ void k_gemm_small(const float* __restrict__ A, long long am, long long ak, const float* __restrict__ B, long long bk, long long bn,
                             float* __restrict__ C, long long ldc, int M, int N, int K, int accumulate) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)M * N) return;
    const int m = (int)(t / N), n = (int)(t % N);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s = fmaf(A[m * am + k * ak], B[k * bk + n * bn], s);
    float* c = C + m * ldc + n;
    *c = accumulate ? *c + s : s;
}

// ------------------------------------------------------------------------------------------------ im2col / col2im
// col[p][tap * cin + c] = x[p + off(tap)][c] (0 outside the 8x8 board), columns [9 cin, kpad) zero;
// colT[k][p] the same transposed (operand of the weight-gradient GEMM).  One thread per (p, k).
__global__ void k_im2col(const float* __restrict__ x, int P, int cin, int kpad, float* __restrict__ col, float* __restrict__ colT) {
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
    if (colT) colT[(size_t)k * P + p] = v;
}

// dx[p][c] = sum_tap dcol[p - off(tap)][tap * cin + c] over the positions whose tap lands on p
__global__ void k_col2im(const float* __restrict__ dcol, int P, int cin, int kpad, float* __restrict__ dx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)P * cin) return;
    const int p = (int)(t / cin), c = (int)(t % cin), y = (p >> 3) & 7, xx = p & 7;
    float s = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dxx = tap % 3 - 1, qy = y - dy, qx = xx - dxx;       // q + off(tap) = p
        if ((unsigned)qy < 8u && (unsigned)qx < 8u) s += dcol[(size_t)(p - 8 * dy - dxx) * kpad + tap * cin + c];
    }
    dx[t] = s;
}

__global__ void k_transpose(const float* __restrict__ in, int R, int Cc, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < Cc) tile[j][tx] = in[(size_t)(r0 + j) * Cc + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < Cc && r0 + tx < R) out[(size_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

// ------------------------------------------------------------------------------------------------ conv block glue
// Keras block: a = ReLU(z + bias); out = gamma * (a - mean) / sqrt(var + eps) + beta with batch statistics.
// Pass 1: a (in place over z) and per-block partial sums of a, a^2 per channel: part[blk][2][C].
// blockDim = 256 threads = (256 / C) row lanes x C channels (C <= 256, power of two); ROWS rows per block.
constexpr int ROWS = 64;
__global__ void k_bias_relu_stats(float* __restrict__ z, const float* __restrict__ bias, int P, int Cc, int relu, float* __restrict__ part) {
    __shared__ float red[2][256];
    const int c = threadIdx.x % Cc, rl = threadIdx.x / Cc, nrl = blockDim.x / Cc;
    const int r0 = blockIdx.x * ROWS;
    float s = 0.0f, s2 = 0.0f;
    const float b = bias ? bias[c] : 0.0f;
    for (int r = r0 + rl; r < min(P, r0 + ROWS); r += nrl) {
        float v = z[(size_t)r * Cc + c] + b;
        if (relu) v = fmaxf(v, 0.0f);
        z[(size_t)r * Cc + c] = v;
        s += v; s2 += v * v;
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < nrl; ++j) { s += red[0][j * Cc + c]; s2 += red[1][j * Cc + c]; }
        part[((size_t)blockIdx.x * 2 + 0) * Cc + c] = s;
        part[((size_t)blockIdx.x * 2 + 1) * Cc + c] = s2;
    }
}

// Pass 2 (one block, C threads): mean, biased variance, 1 / sqrt(var + eps); moving statistics with torch's
// convention (momentum, unbiased variance).  stats[0][C] = mean, stats[1][C] = inv_std.
__global__ void k_bn_finalize(const float* __restrict__ part, int nblk, int P, int Cc, float eps, float momentum,
                              float* __restrict__ stats, float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = threadIdx.x;
    if (c >= Cc) return;
    double s = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) { s += part[((size_t)b * 2 + 0) * Cc + c]; s2 += part[((size_t)b * 2 + 1) * Cc + c]; }
    const double mean = s / P;
    double var = s2 / P - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[Cc + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)(var * (double)P / (double)(P > 1 ? P - 1 : 1));
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
__global__ void k_bn_bwd_finalize(const float* __restrict__ part, int nblk, int Cc, float* __restrict__ sums,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = threadIdx.x;
    if (c >= Cc) return;
    double s = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) { s += part[((size_t)b * 2 + 0) * Cc + c]; s2 += part[((size_t)b * 2 + 1) * Cc + c]; }
    sums[c] = (float)s; sums[Cc + c] = (float)s2;
    dbeta[c] = (float)s; dgamma[c] = (float)s2;
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

// out[c] = sum_b part[b][c] (double accumulation); one thread per column
__global__ void k_sum_rows(const float* __restrict__ part, int nblk, int Cc, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cc) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * Cc + c];
    out[c] = (float)s;
}

// y = a + b (elementwise): the body's output gradient is the sum of the policy and the value branch
__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ y) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = a[t] + b[t];
}

// dz = dy * [a > 0]  /  y = max(x + bias, 0) for the dense layer of the value head (no BatchNorm statistics needed here)
__global__ void k_relu_bwd(const float* __restrict__ a, long long n, float* __restrict__ d) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && !(a[t] > 0.0f)) d[t] = 0.0f;
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

// acc[0..2] += n_rows * {w_p * mean ce + w_v * mean se + penalty, mean ce, mean se} (float64 running sums of an epoch)
__global__ void k_loss_sums(const float* __restrict__ ce, const float* __restrict__ se, int B, float wp, float wv,
                            const float* __restrict__ penalty, double n_rows, double* __restrict__ acc) {
    if (threadIdx.x || blockIdx.x) return;
    double c = 0.0, s = 0.0;
    for (int b = 0; b < B; ++b) { c += ce[b]; s += se[b]; }
    c /= B; s /= B;
    acc[0] += n_rows * ((double)wp * c + (double)wv * s + (penalty ? (double)*penalty : 0.0));
    acc[1] += n_rows * c;
    acc[2] += n_rows * s;
}

// ------------------------------------------------------------------------------------------------ Adam + l2
// One flat parameter vector; reg[i] = the l2 coefficient of element i (CONV_REG / DENSE_REG on kernels and biases,
// 0 on BatchNorm parameters).  g = grad + 2 reg w;  torch.optim.Adam arithmetic (bias-corrected step size,
// eps outside the square root), lr and the step counter read from device memory (captured in a HIP graph).
__global__ void k_adam(float* __restrict__ w, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                       const float* __restrict__ reg, long long n, const float* __restrict__ lr, float beta1, float beta2, float eps,
                       const float* __restrict__ step) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = *step;
    const float g = grad[i] + 2.0f * reg[i] * w[i];
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    m[i] = mi; v[i] = vi;
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    w[i] -= (*lr / bc1) * (mi / denom);
}

__global__ void k_step_inc(float* step) { if (!threadIdx.x && !blockIdx.x) *step += 1.0f; }

// penalty = sum_i reg[i] w[i]^2 (reporting: Keras adds it to the loss it prints); one block, deterministic
__global__ __launch_bounds__(1024) void k_penalty(const float* __restrict__ w, const float* __restrict__ reg, long long n, float* __restrict__ out) {
    __shared__ double red[1024];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) s += (double)reg[i] * (double)w[i] * (double)w[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    if (!threadIdx.x) *out = (float)red[0];
}

}  // namespace ckrt

using namespace ckrt;

#define LAUNCH1D(kernel, n, st, ...) hipLaunchKernelGGL(kernel, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, (hipStream_t)(st), __VA_ARGS__)

extern "C" {

int ckr_gemm_nt(const float* A, int32_t lda, const float* Bt, int32_t ldb, float* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                int32_t slices, float* workspace, const float* add, void* stream) {
    if (!A || !Bt || !C || M <= 0 || N <= 0 || K <= 0 || M % BM || N % BN || slices < 1 || K % (BK * slices) || (ldc % 4) || (lda % 4) || (ldb % 4))
        return ckr::fail(CKR_ERR_INVALID, "ckr_gemm_nt: M, N multiples of 128, K a multiple of 32 * slices, leading dimensions of 4");
    if (slices > 1 && (!workspace || ldc != N)) return ckr::fail(CKR_ERR_INVALID, "ckr_gemm_nt: split-K needs a workspace and ldc == N");
    if (int rc = ckr::require_device()) return rc;
    float* dst = slices > 1 ? workspace : C;
    hipLaunchKernelGGL(k_gemm_nt, dim3(N / BN, M / BM, slices), dim3(GT), 0, (hipStream_t)stream, A, (int)lda, Bt, (int)ldb, dst, (int)ldc, (int)M, (int)K);
    if (slices > 1 || add) {
        const long long n4 = (long long)M * N / 4;
        if (slices == 1) {                                        // C = C + add
            LAUNCH1D(k_sum_slices, n4, stream, (const float4*)C, 1, n4, (const float4*)add, (float4*)C);
        } else {
            LAUNCH1D(k_sum_slices, n4, stream, (const float4*)workspace, (int)slices, n4, (const float4*)add, (float4*)C);
        }
    }
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

int ckr_im2col(const float* x, int32_t P, int32_t cin, int32_t kpad, float* col, float* colT, void* stream) {
    if (!x || !col || P <= 0 || P % 64 || cin <= 0 || kpad < 9 * cin) return ckr::fail(CKR_ERR_INVALID, "ckr_im2col: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_im2col, (long long)P * kpad, stream, x, (int)P, (int)cin, (int)kpad, col, colT);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_col2im(const float* dcol, int32_t P, int32_t cin, int32_t kpad, float* dx, void* stream) {
    if (!dcol || !dx || P <= 0 || P % 64 || cin <= 0 || kpad < 9 * cin) return ckr::fail(CKR_ERR_INVALID, "ckr_col2im: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_col2im, (long long)P * cin, stream, dcol, (int)P, (int)cin, (int)kpad, dx);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_transpose(const float* in, int32_t R, int32_t Cc, float* out, void* stream) {
    if (!in || !out || R <= 0 || Cc <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_transpose: bad argument");
    if (int rc = ckr::require_device()) return rc;
    hipLaunchKernelGGL(k_transpose, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, (int)R, (int)Cc, out);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

static bool chan_ok(int Cc) { return Cc >= 1 && Cc <= 256 && (256 % Cc) == 0; }

// a = act(z + bias) in place, batch statistics -> stats[2][C], moving statistics updated, out = BatchNorm(a).
// part: workspace of 2 * C * ceil(P / 64) floats.
int ckr_bn_forward(float* z, const float* bias, int32_t P, int32_t Cc, int32_t relu, const float* gamma, const float* beta, float eps,
                   float momentum, float* run_mean, float* run_var, float* stats, float* out, float* part, void* stream) {
    if (!z || !gamma || !beta || !stats || !out || !part || P <= 0 || !chan_ok(Cc)) return ckr::fail(CKR_ERR_INVALID, "ckr_bn_forward: bad argument");
    if (int rc = ckr::require_device()) return rc;
    const int nblk = (P + ROWS - 1) / ROWS;
    hipLaunchKernelGGL(k_bias_relu_stats, dim3(nblk), dim3(256), 0, (hipStream_t)stream, z, bias, (int)P, (int)Cc, (int)relu, part);
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)part, nblk, (int)P, (int)Cc, eps, momentum, stats, run_mean, run_var);
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
    if (dbias) hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)part, nblk, (int)Cc, dbias);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_add(const float* a, const float* b, int64_t n, float* y, void* stream) {
    if (!a || !b || !y || n <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_add: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_add, (long long)n, stream, a, b, (long long)n, y);
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

int ckr_loss_sums(const float* ce, const float* se, int32_t B, float wp, float wv, const float* penalty, double n_rows, double* acc, void* stream) {
    if (!ce || !se || !acc || B <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_loss_sums: bad argument");
    if (int rc = ckr::require_device()) return rc;
    hipLaunchKernelGGL(k_loss_sums, dim3(1), dim3(64), 0, (hipStream_t)stream, ce, se, (int)B, wp, wv, penalty, n_rows, acc);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_adam_step(float* w, const float* grad, float* m, float* v, const float* reg, int64_t n, const float* d_lr, float beta1, float beta2,
                  float eps, float* d_step, float* d_penalty, void* stream) {
    if (!w || !grad || !m || !v || !reg || !d_lr || !d_step || n <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_adam_step: bad argument");
    if (int rc = ckr::require_device()) return rc;
    if (d_penalty) hipLaunchKernelGGL(k_penalty, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)w, reg, (long long)n, d_penalty);
    hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(64), 0, (hipStream_t)stream, d_step);
    LAUNCH1D(k_adam, (long long)n, stream, w, grad, m, v, reg, (long long)n, d_lr, beta1, beta2, eps, (const float*)d_step);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_sum_rows(const float* in, int32_t rows, int32_t cols, float* out, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_sum_rows: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_sum_rows, cols, stream, in, (int)rows, (int)cols, out);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_relu_backward(const float* a, int64_t n, float* d, void* stream) {
    if (!a || !d || n <= 0) return ckr::fail(CKR_ERR_INVALID, "ckr_relu_backward: bad argument");
    if (int rc = ckr::require_device()) return rc;
    LAUNCH1D(k_relu_bwd, (long long)n, stream, a, (long long)n, d);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

}  // extern "C"
