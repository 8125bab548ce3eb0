// AttentionBlock (CBAM-style channel + spatial attention) forward / backward, gfx950.
// Restates /root/reference models/dehazing/base_model.py:43-78 as HBM-streaming kernels over NHWC:
//   forward  = pool pass (read x) + tiny MLP + spatial-stat pass (read x) + apply pass (read x, write out)
//   backward = 3 streaming passes over (g, x) + two tiny kernels.
// Reductions: per-pixel channel reductions use 8 lanes per pixel (128-B coalesced segments) and
// wavefront shuffles; per-channel spatial reductions keep a channel quad per thread and combine
// blocks in a fixed order (deterministic).  Arg-max ties resolve to the FIRST index like
// torch.max / adaptive_max_pool2d on CPU (ReLU outputs contain exact zeros, so ties do occur).
#include "common.h"

#ifndef POOL_PPB
#define POOL_PPB 512      // pixels per block of cbam_pool_partial_kernel and cbam_bwd_c_kernel (1024 before: cbam_bwd_c 1.61 -> 1.41 ms per step, cbam_pool 0.85 -> 0.80)
#endif
#ifndef CBAM_SCALE_MAXBLK
#define CBAM_SCALE_MAXBLK 65536    // blocks per image of the two-stream kernels (cbam_scale_kernel, cbam_bwd_e_kernel): one trip per thread
                                   // (swept at the end of round 4, profiles/r04b_sweep_cbam_grid.txt: cbam_apply 1.78 -> 1.69 ms per step, cbam_bwd_e 1.57 -> 1.52
                                   // against the cap of 1024)
#endif
#ifndef CBAM_ROW_MAXBLK
#define CBAM_ROW_MAXBLK 2048       // blocks per image of the per-pixel kernels (cbam_spatial_stats_kernel, cbam_bwd_a_kernel)
#endif

extern "C" int adh_cbam_pool_num_blocks(int HW) { return adh_ceil_div(HW, POOL_PPB); }

// partial[n][blk][2][C] (sum, max), partial_idx[n][blk][C]
__global__ __launch_bounds__(256) void cbam_pool_partial_kernel(const float* __restrict__ x, int x_cs, int HW, int C,
                                                                float* __restrict__ partial,
                                                                int32_t* __restrict__ partial_idx, int nblk) {
    __shared__ f32x4 rs[256];
    __shared__ f32x4 rm[256];
    __shared__ int ri[256][4];
    const int n = blockIdx.y, blk = blockIdx.x;
    const int CQ = C / 4;
    const int R = 256 / CQ;
    const int cq = threadIdx.x % CQ, prow = threadIdx.x / CQ;
    const int c = cq * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int mi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if (prow < R) {
        const float* xn = x + (size_t)n * HW * x_cs;
        const int p0 = blk * POOL_PPB;
        const int p1 = adh_min_i(p0 + POOL_PPB, HW);
        // eight independent 16-byte loads in flight per lane (HBM-bound: one load per iteration left the pass at 2.4 TB/s)
        for (int p = p0 + prow; p < p1; p += 8 * R) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = p + u * R < p1 ? p + u * R : p;
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xn + (size_t)q * x_cs + c));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = p + u * R;
                if (q < p1) {
                    s += v[u];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (v[u][j] > m[j]) {     // strictly greater, ascending q: the first index wins ties
                            m[j] = v[u][j];
                            mi[j] = q;
                        }
                }
            }
        }
    }
    rs[threadIdx.x] = s;
    rm[threadIdx.x] = m;
#pragma unroll
    for (int j = 0; j < 4; ++j) ri[threadIdx.x][j] = mi[j];
    __syncthreads();
    if (threadIdx.x < CQ) {
        f32x4 S = {0.f, 0.f, 0.f, 0.f};
        f32x4 M = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int MI[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
        for (int r = 0; r < R; ++r) {
            const int t = r * CQ + threadIdx.x;
            S += rs[t];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = rm[t][j];
                const int vi = ri[t][j];
                if (v > M[j] || (v == M[j] && vi < MI[j])) {
                    M[j] = v;
                    MI[j] = vi;
                }
            }
        }
        const size_t base = ((size_t)n * nblk + blk);
        *reinterpret_cast<f32x4*>(partial + (base * 2 + 0) * C + c) = S;
        *reinterpret_cast<f32x4*>(partial + (base * 2 + 1) * C + c) = M;
#pragma unroll
        for (int j = 0; j < 4; ++j) partial_idx[base * C + c + j] = MI[j];
    }
}

// 256 threads per (image, 32 channels): 8 partial streams per channel, combined through LDS (the serial walk over
// nblk = HW / 1024 partials by one thread per channel was latency-bound: 0.25 ms at HW = 524 288)
__global__ __launch_bounds__(256) void cbam_pool_final_kernel(const float* __restrict__ partial,
                                                              const int32_t* __restrict__ partial_idx, int nblk, int HW, int C,
                                                              float* __restrict__ pooled, int32_t* __restrict__ amax_idx) {
    __shared__ double rs[8][32];
    __shared__ float rm[8][32];
    __shared__ int ri[8][32];
    const int n = blockIdx.y;
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double S = 0.0;
    float M = -INFINITY;
    int MI = 0x7fffffff;
    if (c < C)
        for (int b = grp; b < nblk; b += 8) {
            const size_t base = (size_t)n * nblk + b;
            S += (double)partial[(base * 2 + 0) * C + c];
            const float v = partial[(base * 2 + 1) * C + c];
            const int vi = partial_idx[base * C + c];
            if (v > M || (v == M && vi < MI)) {
                M = v;
                MI = vi;
            }
        }
    rs[grp][cl] = S;
    rm[grp][cl] = M;
    ri[grp][cl] = MI;
    __syncthreads();
    if (grp == 0 && c < C) {
        double T = 0.0;
        float Mx = -INFINITY;
        int Ix = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            T += rs[r][cl];
            const float v = rm[r][cl];
            const int vi = ri[r][cl];
            if (v > Mx || (v == Mx && vi < Ix)) {
                Mx = v;
                Ix = vi;
            }
        }
        pooled[((size_t)n * 2 + 0) * C + c] = (float)(T / (double)HW);
        pooled[((size_t)n * 2 + 1) * C + c] = Mx;
        amax_idx[(size_t)n * C + c] = Ix;
    }
}

extern "C" int adh_cbam_pool(void* stream, const float* x, int x_cs, int N, int HW, int C, float* partial,
                             int32_t* partial_idx, int nblk, float* pooled, int32_t* amax_idx) {
    if (!x || !partial || !partial_idx || !pooled || !amax_idx || N < 1 || HW < 1 || C < 4 || (C & 3) || C > 1024)
        return ADH_E_ARG;
    if (nblk != adh_cbam_pool_num_blocks(HW)) return ADH_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cbam_pool_partial_kernel, dim3(nblk, N), dim3(256), 0, s, x, x_cs, HW, C, partial, partial_idx,
                       nblk);
    hipLaunchKernelGGL(cbam_pool_final_kernel, dim3(adh_ceil_div(C, 32), N), dim3(256), 0, s, partial, partial_idx, nblk,
                       HW, C, pooled, amax_idx);
    return adh_check_launch();
}

// ca = sigmoid(W2 relu(W1 avg) + W2 relu(W1 max)); hidden[n][2][Ch] saved (post-relu)
__global__ __launch_bounds__(256) void cbam_mlp_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                       const float* __restrict__ w2, int C, int Ch,
                                                       float* __restrict__ ca, float* __restrict__ hidden) {
    extern __shared__ float sm[];  // [2][C] pooled, [2][Ch] hidden
    float* pv = sm;
    float* hv = sm + 2 * C;
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) pv[i] = pooled[(size_t)n * 2 * C + i];
    __syncthreads();
    // hidden: one wave per (which, j) pair, strided
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int o = wave; o < 2 * Ch; o += (blockDim.x >> 6)) {
        const int which = o / Ch, j = o - which * Ch;
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += w1[(size_t)j * C + c] * pv[which * C + c];
        acc = wave_sum(acc);
        if (lane == 0) {
            const float hval = fmaxf(acc, 0.f);
            hv[o] = hval;
            hidden[(size_t)n * 2 * Ch + o] = hval;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int j = 0; j < Ch; ++j) acc += w2[(size_t)c * Ch + j] * hv[j];
        float acc2 = 0.f;
        for (int j = 0; j < Ch; ++j) acc2 += w2[(size_t)c * Ch + j] * hv[Ch + j];
        ca[(size_t)n * C + c] = 1.0f / (1.0f + expf(-(acc + acc2)));
    }
}

extern "C" int adh_cbam_mlp(void* stream, const float* pooled, const float* w1, const float* w2, int N, int C, int Ch,
                            float* ca, float* hidden) {
    if (!pooled || !w1 || !w2 || !ca || !hidden || N < 1 || C < 1 || Ch < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(cbam_mlp_kernel, dim3(N), dim3(256), (2 * C + 2 * Ch) * sizeof(float), (hipStream_t)stream, pooled,
                       w1, w2, C, Ch, ca, hidden);
    return adh_check_launch();
}

// per pixel: mean_c / max_c / first argmax_c of x*ca ; 8 lanes per pixel
__global__ __launch_bounds__(256) void cbam_spatial_stats_kernel(const float* __restrict__ x, int x_cs,
                                                                 const float* __restrict__ ca, int HW, int C,
                                                                 float* __restrict__ smap, int32_t* __restrict__ cidx) {
    const int n = blockIdx.y;
    const int sub = threadIdx.x & 7;
    const int CQ = C / 4;
    const float* xn = x + (size_t)n * HW * x_cs;
    const float* can = ca + (size_t)n * C;
    const float invC = 1.0f / (float)C;
    for (int p = blockIdx.x * 32 + (threadIdx.x >> 3); p < HW; p += gridDim.x * 32) {
        float s = 0.f, m = -INFINITY;
        int mi = 0x7fffffff;
        for (int q = sub; q < CQ; q += 8) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xn + (size_t)p * x_cs + q * 4)) *
                            *reinterpret_cast<const f32x4*>(can + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s += v[j];
                if (v[j] > m) {
                    m = v[j];
                    mi = q * 4 + j;
                }
            }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            const float om = __shfl_xor(m, o, 64);
            const int oi = __shfl_xor(mi, o, 64);
            if (om > m || (om == m && oi < mi)) {
                m = om;
                mi = oi;
            }
        }
        if (sub == 0) {
            smap[((size_t)n * HW + p) * 2 + 0] = s * invC;
            smap[((size_t)n * HW + p) * 2 + 1] = m;
            cidx[(size_t)n * HW + p] = mi;
        }
    }
}

extern "C" int adh_cbam_spatial_stats(void* stream, const float* x, int x_cs, const float* ca, int N, int HW, int C,
                                      float* smap, int32_t* cidx) {
    if (!x || !ca || !smap || !cidx || N < 1 || HW < 1 || C < 4 || (C & 3)) return ADH_E_ARG;
    const int blocks = adh_min_i(adh_ceil_div(HW, 32), CBAM_ROW_MAXBLK);
    hipLaunchKernelGGL(cbam_spatial_stats_kernel, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, x, x_cs, ca, HW, C,
                       smap, cidx);
    return adh_check_launch();
}

// sa = sigmoid(conv7x7(smap)); out = x*ca*sa.  Block = 64 consecutive pixels of one row.
// pass 3a: sa = sigmoid(conv7x7([mean_C, max_C])) -- one thread per pixel, the 2-channel map tile and its halo in LDS
#define SA_TW 32
#define SA_TH 8
__global__ __launch_bounds__(256) void cbam_sa_kernel(const float* __restrict__ smap, const float* __restrict__ wsp, int H,
                                                      int W, float* __restrict__ sa) {
    __shared__ float tile[SA_TH + 6][SA_TW + 6][2];
    __shared__ float s_w[98];
    const int n = blockIdx.z, y0 = blockIdx.y * SA_TH, x0 = blockIdx.x * SA_TW;
    if (threadIdx.x < 98) s_w[threadIdx.x] = wsp[threadIdx.x];
    const float* sm = smap + (size_t)n * H * W * 2;
    for (int i = threadIdx.x; i < (SA_TH + 6) * (SA_TW + 6); i += 256) {
        const int r = i / (SA_TW + 6), cc = i - r * (SA_TW + 6);
        const int yy = y0 + r - 3, xx = x0 + cc - 3;
        float a = 0.f, b = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const float* q = sm + ((size_t)yy * W + xx) * 2;
            a = q[0];
            b = q[1];
        }
        tile[r][cc][0] = a;
        tile[r][cc][1] = b;
    }
    __syncthreads();
    const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
    const int y = y0 + ly, x = x0 + lx;
    if (y < H && x < W) {
        float v = 0.f;
        // same summation order as the reference's conv2d row walk (ky, kx ascending; channel 0 then 1 per tap)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
                v += tile[ly + ky][lx + kx][0] * s_w[ky * 7 + kx] + tile[ly + ky][lx + kx][1] * s_w[49 + ky * 7 + kx];
        sa[((size_t)n * H + y) * W + x] = 1.0f / (1.0f + expf(-v));
    }
}

// pass 3b: out = x * ca[c] * sa[p] -- streaming, a thread keeps one channel quad (cbam_scale layout = bn_act.hip's):
// the launch has T threads, T a multiple of CQ; thread t walks pixels t / CQ, + T / CQ, ... eight at a time
__global__ __launch_bounds__(256) void cbam_scale_kernel(const float* __restrict__ x, int x_cs, const float* __restrict__ ca,
                                                         const float* __restrict__ sa, int64_t HW, int CQ,
                                                         float* __restrict__ out, int out_cs) {
    const int n = blockIdx.y;
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t pstep = (int64_t)gridDim.x * 256 / CQ;
    int64_t p = t / CQ;
    const int c = (int)(t - p * CQ) * 4;
    const f32x4 cav = *reinterpret_cast<const f32x4*>(ca + (size_t)n * CQ * 4 + c);
    const float* xn = x + (size_t)n * HW * x_cs;
    float* on = out + (size_t)n * HW * out_cs;
    const float* san = sa + (size_t)n * HW;
    for (; p < HW; p += 8 * pstep) {
        f32x4 v[8];
        float sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t q = p + u * pstep < HW ? p + u * pstep : p;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xn + q * x_cs + c));
            sv[u] = san[q];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t q = p + u * pstep;
            if (q < HW) __builtin_nontemporal_store(v[u] * cav * sv[u], reinterpret_cast<f32x4*>(on + q * out_cs + c));
        }
    }
}

static int cbam_scale_blocks(int64_t HW, int CQ) {
    int g = CQ, r = 256;   // gcd(CQ, 256)
    while (r) { const int t = g % r; g = r; r = t; }
    const int mult = CQ / g;
    int64_t want = (HW * CQ + 256 * 8 - 1) / (256 * 8);
    if (want > CBAM_SCALE_MAXBLK) want = CBAM_SCALE_MAXBLK;
    int64_t blocks = (want + mult - 1) / mult * mult;
    if (blocks < mult) blocks = mult;
    return (int)blocks;
}

extern "C" int adh_cbam_apply(void* stream, const float* x, int x_cs, const float* ca, const float* smap,
                              const float* wsp, int N, int H, int W, int C, float* sa, float* out, int out_cs) {
    if (!x || !ca || !smap || !wsp || !sa || !out || N < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || (x_cs & 3) || (out_cs & 3) ||
        N > 65535)
        return ADH_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cbam_sa_kernel, dim3(adh_ceil_div(W, SA_TW), adh_ceil_div(H, SA_TH), N), dim3(256), 0, s, smap, wsp, H, W,
                       sa);
    hipLaunchKernelGGL(cbam_scale_kernel, dim3(cbam_scale_blocks((int64_t)H * W, C / 4), N), dim3(256), 0, s, x, x_cs, ca, sa,
                       (int64_t)H * W, C / 4, out, out_cs);
    return adh_check_launch();
}

// ------------------------------------------- backward ---------------------------------------------
// A: gsa_pre[n][hw] = (sum_c g*x*ca) * sa*(1-sa)
__global__ __launch_bounds__(256) void cbam_bwd_a_kernel(const float* __restrict__ g, int g_cs,
                                                         const float* __restrict__ x, int x_cs,
                                                         const float* __restrict__ ca, const float* __restrict__ sa,
                                                         int HW, int C, float* __restrict__ gsa_pre) {
    const int n = blockIdx.y;
    const int sub = threadIdx.x & 7;
    const int CQ = C / 4;
    const float* xn = x + (size_t)n * HW * x_cs;
    const float* gn = g + (size_t)n * HW * g_cs;
    const float* can = ca + (size_t)n * C;
    for (int p = blockIdx.x * 32 + (threadIdx.x >> 3); p < HW; p += gridDim.x * 32) {
        float s = 0.f;
        for (int q = sub; q < CQ; q += 8) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xn + (size_t)p * x_cs + q * 4)) *
                            *reinterpret_cast<const f32x4*>(can + q * 4) *
                            __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gn + (size_t)p * g_cs + q * 4));
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (sub == 0) {
            const float a = sa[(size_t)n * HW + p];
            gsa_pre[(size_t)n * HW + p] = s * a * (1.f - a);
        }
    }
}

extern "C" int adh_cbam_bwd_a(void* stream, const float* g, int g_cs, const float* x, int x_cs, const float* ca,
                              const float* sa, int N, int HW, int C, float* gsa_pre) {
    if (!g || !x || !ca || !sa || !gsa_pre || N < 1 || HW < 1 || C < 4 || (C & 3)) return ADH_E_ARG;
    const int blocks = adh_min_i(adh_ceil_div(HW, 32), CBAM_ROW_MAXBLK);
    hipLaunchKernelGGL(cbam_bwd_a_kernel, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, g, g_cs, x, x_cs, ca, sa, HW,
                       C, gsa_pre);
    return adh_check_launch();
}

// B: gsmap = conv7x7^T(gsa_pre) ; dwsp partial sums.  One thread per pixel, 256 pixels per block.
extern "C" int adh_cbam_bwd_b_num_blocks(int N, int H, int W) { return adh_ceil_div((int64_t)N * H * W, 256); }

__global__ __launch_bounds__(256) void cbam_bwd_b_kernel(const float* __restrict__ gsa_pre,
                                                         const float* __restrict__ smap, const float* __restrict__ wsp,
                                                         int N, int H, int W, float* __restrict__ gsmap,
                                                         float* __restrict__ dwsp_partial) {
    __shared__ float s_w[98];
    __shared__ float s_acc[4][98];
    if (threadIdx.x < 98) s_w[threadIdx.x] = wsp[threadIdx.x];
    __syncthreads();
    const int64_t total = (int64_t)N * H * W;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < total;
    int n = 0, y = 0, xx = 0;
    if (valid) {
        xx = (int)(idx % W);
        const int64_t r = idx / W;
        y = (int)(r % H);
        n = (int)(r / H);
    }
    const float* gp = gsa_pre + (size_t)n * H * W;
    const float* sm = smap + (size_t)n * H * W * 2;
    const float gs_here = valid ? gp[(size_t)y * W + xx] : 0.f;
    float g0 = 0.f, g1 = 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int ky = 0; ky < 7; ++ky) {
        for (int kx = 0; kx < 7; ++kx) {
            // transposed conv: this input pixel (y,xx) contributed to output (y-ky+3, xx-kx+3) with weight [ky][kx]
            const int oy = y - ky + 3, ox = xx - kx + 3;
            if (valid && oy >= 0 && oy < H && ox >= 0 && ox < W) {
                const float go = gp[(size_t)oy * W + ox];
                g0 += go * s_w[ky * 7 + kx];
                g1 += go * s_w[49 + ky * 7 + kx];
            }
            // weight gradient: output pixel (y,xx) read smap at (y+ky-3, xx+kx-3)
            const int iy = y + ky - 3, ix = xx + kx - 3;
            float a0 = 0.f, a1 = 0.f;
            if (valid && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const float* q = sm + ((size_t)iy * W + ix) * 2;
                a0 = q[0] * gs_here;
                a1 = q[1] * gs_here;
            }
            a0 = wave_sum(a0);
            a1 = wave_sum(a1);
            if (lane == 0) {
                s_acc[wave][ky * 7 + kx] = a0;
                s_acc[wave][49 + ky * 7 + kx] = a1;
            }
        }
    }
    if (valid) {
        gsmap[idx * 2 + 0] = g0;
        gsmap[idx * 2 + 1] = g1;
    }
    __syncthreads();
    if (threadIdx.x < 98)
        dwsp_partial[(size_t)blockIdx.x * 98 + threadIdx.x] =
            (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + (s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void cbam_dwsp_final_kernel(const float* __restrict__ partial, int nblk, float* dwsp,
                                                              int accumulate) {
    __shared__ double red[256];
    const int o = blockIdx.x;  // 0..97
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partial[(size_t)b * 98 + o];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) dwsp[o] = accumulate ? dwsp[o] + (float)red[0] : (float)red[0];
}

extern "C" int adh_cbam_bwd_b(void* stream, const float* gsa_pre, const float* smap, const float* wsp, int N, int H,
                              int W, float* gsmap, float* dwsp_partial, int nblk, float* dwsp, int accumulate) {
    if (!gsa_pre || !smap || !wsp || !gsmap || !dwsp_partial || !dwsp) return ADH_E_ARG;
    if (nblk != adh_cbam_bwd_b_num_blocks(N, H, W)) return ADH_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cbam_bwd_b_kernel, dim3(nblk), dim3(256), 0, s, gsa_pre, smap, wsp, N, H, W, gsmap, dwsp_partial);
    hipLaunchKernelGGL(cbam_dwsp_final_kernel, dim3(98), dim3(256), 0, s, dwsp_partial, nblk, dwsp, accumulate);
    return adh_check_launch();
}

// gx1 for a channel quad of one pixel
__device__ __forceinline__ f32x4 cbam_gx1(const f32x4 g, float sa, float gmean_over_c, float gmax, int cidx, int c) {
    f32x4 r = g * sa + gmean_over_c;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c + j == cidx) r[j] += gmax;
    return r;
}

// C: gca_partial[n][blk][C] = sum_hw gx1 * x
__global__ __launch_bounds__(256) void cbam_bwd_c_kernel(const float* __restrict__ g, int g_cs,
                                                         const float* __restrict__ x, int x_cs,
                                                         const float* __restrict__ sa, const float* __restrict__ gsmap,
                                                         const int32_t* __restrict__ cidx, int HW, int C,
                                                         float* __restrict__ gca_partial, int nblk) {
    __shared__ f32x4 rs[256];
    const int n = blockIdx.y, blk = blockIdx.x;
    const int CQ = C / 4;
    const int R = 256 / CQ;
    const int cq = threadIdx.x % CQ, prow = threadIdx.x / CQ;
    const int c = cq * 4;
    const float invC = 1.0f / (float)C;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (prow < R) {
        const float* xn = x + (size_t)n * HW * x_cs;
        const float* gn = g + (size_t)n * HW * g_cs;
        const int p0 = blk * POOL_PPB;
        const int p1 = adh_min_i(p0 + POOL_PPB, HW);
        // four pixels per trip, every load of the trip issued before the first use (one pixel per trip left this pass at
        // 4.8 TB/s: a single 16-byte load pair in flight per thread)
        for (int p = p0 + prow; p < p1; p += 4 * R) {
            f32x4 gv[4], xv[4];
            float sv[4], gm[4], gx[4];
            int ci[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = p + u * R < p1 ? p + u * R : p;
                const size_t pp = (size_t)n * HW + q;
                gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gn + (size_t)q * g_cs + c));
                xv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xn + (size_t)q * x_cs + c));
                sv[u] = sa[pp];
                gm[u] = gsmap[pp * 2 + 0];
                gx[u] = gsmap[pp * 2 + 1];
                ci[u] = cidx[pp];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p + u * R < p1) s += cbam_gx1(gv[u], sv[u], gm[u] * invC, gx[u], ci[u], c) * xv[u];
        }
    }
    rs[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < CQ) {
        f32x4 S = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) S += rs[r * CQ + threadIdx.x];
        *reinterpret_cast<f32x4*>(gca_partial + ((size_t)n * nblk + blk) * C + c) = S;
    }
}

extern "C" int adh_cbam_bwd_c(void* stream, const float* g, int g_cs, const float* x, int x_cs, const float* sa,
                              const float* gsmap, const int32_t* cidx, int N, int HW, int C, float* gca_partial,
                              int nblk) {
    if (!g || !x || !sa || !gsmap || !cidx || !gca_partial || C < 4 || (C & 3) || C > 1024) return ADH_E_ARG;
    if (nblk != adh_cbam_pool_num_blocks(HW)) return ADH_E_ARG;
    hipLaunchKernelGGL(cbam_bwd_c_kernel, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, g, g_cs, x, x_cs, sa, gsmap,
                       cidx, HW, C, gca_partial, nblk);
    return adh_check_launch();
}

// D: tiny; single block, images processed in order (deterministic weight-gradient accumulation)
// pass D in three small launches instead of one serial workgroup (which walked N images x nblk partials per thread:
// 0.6 ms of pure latency at HW = 524 288):
//   d1: gpre[n][c] = (sum_blk gca_partial) * ca (1 - ca)      grid (C/32, N), 8 partial streams per channel in flight
//   d2: per image: hidden gradients through W2 and the ReLU masks, gpool through W1        grid N
//   d3: dW1, dW2 = sum over images in a fixed order (deterministic)                        grid ceil(2 C Ch / 256)
// scratch: gpre [N][C] | ghid [N][2][Ch]
__global__ __launch_bounds__(256) void cbam_bwd_d1_kernel(const float* __restrict__ gca_partial, int nblk,
                                                          const float* __restrict__ ca, int C, float* __restrict__ gpre) {
    __shared__ double red[8][32];
    const int n = blockIdx.y;
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0;
    if (c < C)
        for (int b = grp; b < nblk; b += 8) s += (double)gca_partial[((size_t)n * nblk + b) * C + c];
    red[grp][cl] = s;
    __syncthreads();
    if (grp == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][cl];
        const float a = ca[(size_t)n * C + c];
        gpre[(size_t)n * C + c] = (float)t * a * (1.f - a);
    }
}

__global__ __launch_bounds__(256) void cbam_bwd_d2_kernel(const float* __restrict__ gpre_all, const float* __restrict__ hidden,
                                                          const float* __restrict__ w1, const float* __restrict__ w2, int C,
                                                          int Ch, float* __restrict__ ghid_all, float* __restrict__ gpool) {
    extern __shared__ float sm[];
    float* gpre = sm;            // [C]
    float* ghid = sm + C;        // [2][Ch]
    const int n = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int c = threadIdx.x; c < C; c += blockDim.x) gpre[c] = gpre_all[(size_t)n * C + c];
    __syncthreads();
    for (int o = wave; o < 2 * Ch; o += nw) {
        const int which = o / Ch, j = o - which * Ch;
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += w2[(size_t)c * Ch + j] * gpre[c];
        acc = wave_sum(acc);
        if (lane == 0) {
            const float v = hidden[(size_t)n * 2 * Ch + o] > 0.f ? acc : 0.f;
            ghid[o] = v;
            ghid_all[(size_t)n * 2 * Ch + o] = v;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float ga = 0.f, gm = 0.f;
        for (int j = 0; j < Ch; ++j) {
            const float w = w1[(size_t)j * C + c];
            ga += w * ghid[j];
            gm += w * ghid[Ch + j];
        }
        gpool[((size_t)n * 2 + 0) * C + c] = ga;
        gpool[((size_t)n * 2 + 1) * C + c] = gm;
    }
}

__global__ __launch_bounds__(256) void cbam_bwd_d3_kernel(const float* __restrict__ gpre, const float* __restrict__ ghid,
                                                          const float* __restrict__ pooled, const float* __restrict__ hidden,
                                                          int N, int C, int Ch, float* dw1, float* dw2, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = C * Ch;
    if (i < total) {            // dw2 [C][Ch]
        const int c = i / Ch, j = i - c * Ch;
        float v = 0.f;
        for (int n = 0; n < N; ++n)
            v += gpre[(size_t)n * C + c] * (hidden[(size_t)n * 2 * Ch + j] + hidden[(size_t)n * 2 * Ch + Ch + j]);
        dw2[i] = accumulate ? dw2[i] + v : v;
    } else if (i < 2 * total) {   // dw1 [Ch][C]
        const int k = i - total;
        const int j = k / C, c = k - j * C;
        float v = 0.f;
        for (int n = 0; n < N; ++n)
            v += ghid[(size_t)n * 2 * Ch + j] * pooled[((size_t)n * 2 + 0) * C + c] +
                 ghid[(size_t)n * 2 * Ch + Ch + j] * pooled[((size_t)n * 2 + 1) * C + c];
        dw1[k] = accumulate ? dw1[k] + v : v;
    }
}

extern "C" int adh_cbam_bwd_d_scratch_floats(int N, int C, int Ch) { return N * (C + 2 * Ch); }

extern "C" int adh_cbam_bwd_d(void* stream, const float* gca_partial, int nblk, const float* ca, const float* pooled,
                              const float* hidden, const float* w1, const float* w2, int N, int C, int Ch, float* gpool,
                              float* dw1, float* dw2, int accumulate, float* scratch) {
    if (!gca_partial || !ca || !pooled || !hidden || !w1 || !w2 || !gpool || !dw1 || !dw2 || !scratch || N < 1 || C < 1 ||
        Ch < 1 || nblk < 1 || N > 65535)
        return ADH_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    float* gpre = scratch;
    float* ghid = scratch + (size_t)N * C;
    hipLaunchKernelGGL(cbam_bwd_d1_kernel, dim3(adh_ceil_div(C, 32), N), dim3(256), 0, s, gca_partial, nblk, ca, C, gpre);
    hipLaunchKernelGGL(cbam_bwd_d2_kernel, dim3(N), dim3(256), (C + 2 * Ch) * sizeof(float), s, gpre, hidden, w1, w2, C, Ch,
                       ghid, gpool);
    hipLaunchKernelGGL(cbam_bwd_d3_kernel, dim3(adh_ceil_div(2 * C * Ch, 256)), dim3(256), 0, s, gpre, ghid, pooled, hidden, N,
                       C, Ch, dw1, dw2, accumulate);
    return adh_check_launch();
}

// E: gx = gx1*ca + gpool.avg/HW + gpool.max*[hw==amax_idx]
__global__ __launch_bounds__(256) void cbam_bwd_e_kernel(const float* __restrict__ g, int g_cs,
                                                         const float* __restrict__ ca, const float* __restrict__ sa,
                                                         const float* __restrict__ gsmap,
                                                         const int32_t* __restrict__ cidx,
                                                         const float* __restrict__ gpool,
                                                         const int32_t* __restrict__ amax_idx, int HW, int C,
                                                         float* __restrict__ gx, int gx_cs) {
    // thread = one channel quad for the whole launch (grid = a multiple of CQ / gcd(CQ, 256) blocks, cbam_scale_blocks): its
    // per-channel constants are loaded once, and four pixels' loads are in flight per trip
    const int n = blockIdx.y;
    const int CQ = C / 4;
    const float invC = 1.0f / (float)C, invHW = 1.0f / (float)HW;
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t pstep = (int64_t)gridDim.x * 256 / CQ;
    int64_t p = t / CQ;
    const int c = (int)(t - p * CQ) * 4;
    const float* gn = g + (size_t)n * HW * g_cs;
    float* gxn = gx + (size_t)n * HW * gx_cs;
    const f32x4 cav = *reinterpret_cast<const f32x4*>(ca + (size_t)n * C + c);
    const f32x4 gavg = *reinterpret_cast<const f32x4*>(gpool + ((size_t)n * 2 + 0) * C + c) * invHW;
    const f32x4 gmx = *reinterpret_cast<const f32x4*>(gpool + ((size_t)n * 2 + 1) * C + c);
    int am[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) am[j] = amax_idx[(size_t)n * C + c + j];
    for (; p < HW; p += 4 * pstep) {
        f32x4 gv[4];
        float sv[4], gm[4], gxm[4];
        int ci[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = p + u * pstep < HW ? p + u * pstep : p;
            const size_t pp = (size_t)n * HW + q;
            gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gn + (size_t)q * g_cs + c));
            sv[u] = sa[pp];
            gm[u] = gsmap[pp * 2 + 0];
            gxm[u] = gsmap[pp * 2 + 1];
            ci[u] = cidx[pp];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = p + u * pstep;
            if (q < HW) {
                f32x4 r = cbam_gx1(gv[u], sv[u], gm[u] * invC, gxm[u], ci[u], c);
                r = r * cav + gavg;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (am[j] == (int)q) r[j] += gmx[j];
                __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(gxn + (size_t)q * gx_cs + c));
            }
        }
    }
}

extern "C" int adh_cbam_bwd_e(void* stream, const float* g, int g_cs, const float* x, int x_cs, const float* ca,
                              const float* sa, const float* gsmap, const int32_t* cidx, const float* gpool,
                              const int32_t* amax_idx, int N, int HW, int C, float* gx, int gx_cs) {
    (void)x;
    (void)x_cs;
    if (!g || !ca || !sa || !gsmap || !cidx || !gpool || !amax_idx || !gx || C < 4 || (C & 3)) return ADH_E_ARG;
    const int blocks = cbam_scale_blocks(HW, C / 4);
    hipLaunchKernelGGL(cbam_bwd_e_kernel, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, g, g_cs, ca, sa, gsmap, cidx,
                       gpool, amax_idx, HW, C, gx, gx_cs);
    return adh_check_launch();
}
