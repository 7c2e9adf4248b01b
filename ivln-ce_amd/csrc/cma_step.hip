// Fused recurrent / attention head of one MapCMA rollout step (MapCMANet.forward after the encoders,
// ivlnce_baselines/models/map_cma_policy.py:305-353, `_attn` :266-274; habitat-lab RNNStateEncoder single step):
//
//   state = GRU1([dep_in | map_in | prev], h1 * mask)
//   text  = softmax(((W_q state + b_q) . text_k) * s - 1e8 * pad) . txt
//   dep'  = softmax(((W_tq text + b_tq) . dep_k) * s) . dep_v          (map' likewise)
//   feats = GRU2(ReLU(W_c [state | text | dep' | map' | prev] + b_c), h2 * mask)
//
// As separate ops this is a chain of 10 dependent launches (GRU, 4 skinny linears, logits, softmax-out, the two
// short attentions, GRU), ~65 us of the 4-env step.  Two things shorten it:
//
// 1. Algebra that removes two links of the chain.  Everything that depends only on the instruction is folded
//    ahead of the recurrent state (the caller does it in the instruction branch, off the critical path):
//        Mq[h][i]  = sum_c W_q[c][h] text_k[c][i]   (+ row H: sum_c b_q[c] text_k[c][i])   -> logits = state . Mq
//        TQb[c][i] = sum_c' W_tq[c][c'] txt[c'][i] + b_tq[c]                               -> q2 = sum_i a_i TQb[:, i]
//    and the query of the two short attentions never materialises: with S[i][p] = sum_c TQb[c][i] k[c][p] (a
//    [L x 16] table per row and attention, computed here in phase 1 beside GRU-1),  q2 . k[:, p] = sum_i a_i S[i][p].
//    Five dependent phases remain: GRU-1 -> text logits -> {text, dep', map'} -> compress -> GRU-2; the S tables and
//    GRU-2's hidden half (W_hh2 h2, known at step start) ride in the launches of phases 2 and 3, off the chain.  (Summation order changes at the 1e-7 level; the goldens' 2e-4 bar is untouched.)
//
// 2. One kernel per phase, each with the grid its phase wants (512 / 340 / 672 / 512 / 512 workgroups at 4 envs), behind
//    ONE C-ABI call.  A persistent single-launch form (512 resident workgroups walking the phases, a device-scope
//    counter barrier between them: release fence -> atomic add -> relaxed polling -> acquire fence, write-through
//    stores for everything a later phase reads) was built and measured on MI355X in this round: 1.093 ms per 4-env
//    step against 0.977 ms for the phase launches and 0.988 ms for the unfused chain - four barriers over 512
//    workgroups cost ~25 us each, five times a kernel boundary inside a replayed graph - and its replays under
//    hipGraph were not bit-reproducible in the test suite.  It was dropped; DESIGN.md section 3 keeps the numbers.
//    Round 3 priced that barrier on its own (tools/barrier_bench.hip, profiles/r03_barrier_bench.txt): the form used
//    here - flat counter, fences on both sides - costs 26.6 us at 512 workgroups, which is the 25 us seen; the
//    XCD-hierarchical form 6.7 us, the fence-free form with write-through payload 4.2 us at 256 workgroups.  Four of the
//    best of those (the phases need ~512 workgroups: 15 MB of weights, one output per workgroup) still cost more than
//    four launch boundaries of a replayed graph (3-4 us each, tools/chain_bench.hip), so the phase launches stay.  (The
//    replay irreproducibility was not chased once the form had lost on time; the persistent sequence GRU, csrc/gru_seq.hip,
//    where the exchange is small enough to win, zeroes its counters with a memset node per launch, reads the exchanged
//    state with sc1 loads only, and is tested for reproducibility under load.)
//
// Weights (15 MB fp32) are spread over the grid: a WORKGROUP owns one output (hidden unit / linear output), 64 / 32 /
// 16 lanes share a batch row and split K in 16-byte loads (4 / 8 / 16 rows in flight); nothing here is MFMA-shaped
// (<= 64 batch rows against 512-1536 outputs, a GEMV per row).  Bound: L2 -> CU latency per phase, not FLOPs.
// (A first version gave each WAVE an output and looped over rows in registers on a 128-workgroup grid: 50 us slower
// per step than the unfused chain - per-wave serial latency, not bandwidth, is what these phases are made of.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/ivln_hip.h"

namespace {

constexpr int CT = 256;         // threads per workgroup
constexpr int MAX_L = 512;      // longest instruction axis handled (ATT_MAX_I of the unfused kernel)

typedef ivln_cma_step_desc Desc;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Values a later phase reads are stored write-through at agent scope (`global_store ... sc1`): neighbouring outputs
// share 128-byte lines but are produced by workgroups on different XCDs, and a write-through store leaves no
// partially-updated copy of the line behind in the producing XCD's L2 (MI355X_MICROARCH.md, inter-workgroup
// visibility).  Between kernels the boundary's release / acquire would cover it too; the cost is nil.
__device__ __forceinline__ void st_pub(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LPR lanes share one batch row and split K in 16-byte pieces; 256 / LPR rows are in flight per pass.  A (row, output)
// dot product then needs one log2(LPR)-step shuffle reduction and no LDS (the layout of k_gru_step, nn_ops.hip).
template <int LPR>
__device__ __forceinline__ float lpr_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int LPR>
__device__ __forceinline__ float lpr_dot(const float* __restrict__ w, const float* __restrict__ x, int K, int l) {
    float a = 0.f;
    for (int k = l * 4; k < K; k += LPR * 4) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        a = fmaf(wv.x, xv.x, a);
        a = fmaf(wv.y, xv.y, a);
        a = fmaf(wv.z, xv.z, a);
        a = fmaf(wv.w, xv.w, a);
    }
    return a;
}

// ---- one masked GRU unit j for all rows (workgroup-level) ---------------------------------------------
// gh_pre != nullptr: hidden-side pre-activations (bias included) were computed earlier (GRU-2's phase-1 half)
template <int LPR>
__device__ __forceinline__ void gru_unit(const Desc& D, int j, const float* x, int64_t ldx, int I, const float* w_ih,
                                         const float* w_hh, const float* b_ih, const float* b_hh, const float* h_in,
                                         const float* gh_pre, float* out1, int64_t ld1, float* out2, int64_t ld2) {
    constexpr int RPB = CT / LPR;
    const int H = D.H;
    const int l = threadIdx.x % LPR, rr = threadIdx.x / LPR;
    for (int r0 = 0; r0 < D.rows; r0 += RPB) {
        const int row = r0 + rr;
        const bool ok = row < D.rows;
        const int rc = ok ? row : 0;
        const float mk = D.mask[rc] ? 1.f : 0.f;
        float gi[3], gh[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            gi[g] = lpr_dot<LPR>(w_ih + ((int64_t)g * H + j) * I, x + (int64_t)rc * ldx, I, l);
            if (!gh_pre) gh[g] = lpr_dot<LPR>(w_hh + ((int64_t)g * H + j) * H, h_in + (int64_t)rc * D.ld_h, H, l);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            gi[g] = lpr_sum<LPR>(gi[g]);
            if (!gh_pre) gh[g] = lpr_sum<LPR>(gh[g]);
        }
        if (l == 0 && ok) {
            float a[3], b[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                a[g] = gi[g] + b_ih[g * H + j];
                b[g] = gh_pre ? gh_pre[(int64_t)row * 3 * H + g * H + j] : gh[g] * mk + b_hh[g * H + j];
            }
            const float hp = h_in[(int64_t)row * D.ld_h + j] * mk;
            const float rg = sigmoidf_(a[0] + b[0]);
            const float zg = sigmoidf_(a[1] + b[1]);
            const float ng = tanhf(a[2] + rg * b[2]);
            const float hn = (1.f - zg) * ng + zg * hp;
            st_pub(out1 + (int64_t)row * ld1 + j, hn);
            if (out2) st_pub(out2 + (int64_t)row * ld2 + j, hn);
        }
    }
}

// scratch regions, each starting on a 128-byte line
__host__ __device__ __forceinline__ int64_t al32(int64_t n) { return (n + 31) & ~(int64_t)31; }
__device__ __forceinline__ float* ws_logits(const Desc& D) { return D.ws; }
__device__ __forceinline__ float* ws_S(const Desc& D) { return D.ws + al32((int64_t)D.rows * D.L); }
__device__ __forceinline__ float* ws_gh2(const Desc& D) { return ws_S(D) + al32((int64_t)D.rows * D.L * 2 * D.P); }
__device__ __forceinline__ float* ws_c2(const Desc& D) { return ws_gh2(D) + al32((int64_t)D.rows * 3 * D.H); }

// ---- GRU-1 (workgroup item = hidden unit) -----------------------------------------------------------------
template <int LPR>
__device__ void phase1(const Desc& D, int wg, int nwg) {
    const int sin_w = D.d_out + D.m_out + D.E;
    for (int j = wg; j < D.H; j += nwg)
        gru_unit<LPR>(D, j, D.state_in, sin_w, sin_w, D.w_ih1, D.w_hh1, D.b_ih1, D.b_hh1, D.h_in, nullptr, D.x2, D.x2w,
                      D.h_out, D.ld_ho);
}

// ---- hidden half of GRU-2: gh2[r][g*H + j] = W_hh2[g*H + j] . (h2[r] * mask[r]) + b_hh2 (item = unit j).  Depends on
// nothing but the incoming state, so it rides in the attention phase's launch, off the chain. ----
template <int LPR>
__device__ void side_gh2(const Desc& D, int wg, int nwg) {
    constexpr int RPB = CT / LPR;
    const int H = D.H;
    const int l = threadIdx.x % LPR, rr = threadIdx.x / LPR;
    for (int j = wg; j < H; j += nwg) {
        for (int r0 = 0; r0 < D.rows; r0 += RPB) {
            const int row = r0 + rr;
            const bool ok = row < D.rows;
            const int rc = ok ? row : 0;
            const float mk = D.mask[rc] ? 1.f : 0.f;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float v = lpr_dot<LPR>(D.w_hh2 + ((int64_t)g * H + j) * H, D.h_in + H + (int64_t)rc * D.ld_h, H, l);
                v = lpr_sum<LPR>(v);
                if (l == 0 && ok) st_pub(ws_gh2(D) + (int64_t)row * 3 * H + g * H + j, v * mk + D.b_hh2[g * H + j]);
            }
        }
    }
}

// ---- S[n][i][s*P + p] = sum_c TQb[n][c][i] * k_s[n][c][p],  s = 0: depth keys, 1: map keys.  Item = one (n, i):
// thread = (16 slices of c) x (p), 16 channels each with every load issued up front (one latency, not a 256-deep
// chain: (16 i x 16 p) items with the whole sum per thread made their launch 21 us), slices meet by two shuffles
// and LDS.  Depends on the encoders only: rides in the logits phase's launch. ----
__device__ void side_S(const Desc& D, int wg, int nwg) {
    __shared__ float sp[4][32];
    const int P = D.P;
    for (int t = wg; t < D.rows * D.L; t += nwg) {
        const int n = t / D.L, i = t - n * D.L;
        const int p = threadIdx.x & 15, cs = threadIdx.x >> 4;
        const int cper = D.Hq / 16;
        float a0 = 0.f, a1 = 0.f;
        if (p < P) {
            const float* tq = D.TQb + (int64_t)n * D.TQb_img + (int64_t)(cs * cper) * D.L + i;
            const float* dk = D.dkv + ((int64_t)n * (D.Hq + D.d_out) + cs * cper) * P + p;
            const float* mkp = D.mkv + ((int64_t)n * (D.Hq + D.m_out) + cs * cper) * P + p;
#pragma unroll 16
            for (int c = 0; c < cper; ++c) {
                const float tv = tq[(int64_t)c * D.L];
                a0 = fmaf(tv, dk[c * P], a0);
                a1 = fmaf(tv, mkp[c * P], a1);
            }
        }
        a0 += __shfl_xor(a0, 16, 64);
        a1 += __shfl_xor(a1, 16, 64);
        a0 += __shfl_xor(a0, 32, 64);
        a1 += __shfl_xor(a1, 32, 64);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) < 16) {
            sp[wave][p] = a0;
            sp[wave][16 + p] = a1;
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int q = threadIdx.x & 15, sset = threadIdx.x >> 4;
            if (q < P) {
                const float v = (sp[0][threadIdx.x] + sp[1][threadIdx.x]) + (sp[2][threadIdx.x] + sp[3][threadIdx.x]);
                st_pub(ws_S(D) + ((int64_t)n * D.L + i) * 2 * P + sset * P + q, v);
            }
        }
        __syncthreads();
    }
}

// ---- phase 2: text-attention logits = (state . Mq + bias row - 1e8 * pad) * scale --------------------
__device__ void phase2(const Desc& D, int wg, int nwg) {
    __shared__ float part[16][17];
    const int tiles = (D.L + 15) / 16;
    const int ii = threadIdx.x & 15, hp = threadIdx.x >> 4;  // position in the tile, one of 16 slices of H
    const int hs = D.H / 16;
    for (int it = wg; it < D.rows * tiles; it += nwg) {
        const int n = it / tiles, i = (it - n * tiles) * 16 + ii;
        float acc = 0.f;
        if (i < D.L) {
            const float* st = D.x2 + (int64_t)n * D.x2w + hp * hs;
            const float* m = D.Mq + (int64_t)n * D.Mq_img + (int64_t)(hp * hs) * D.L + i;
#pragma unroll 8
            for (int h = 0; h < hs; ++h) acc = fmaf(st[h], m[(int64_t)h * D.L], acc);
        }
        part[hp][ii] = acc;
        __syncthreads();
        if (hp == 0 && i < D.L) {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) s += part[u][ii];
            s += D.Mq[(int64_t)n * D.Mq_img + (int64_t)D.H * D.L + i];
            if (i >= D.lengths[n]) s = s - 1e8f;
            st_pub(ws_logits(D) + (int64_t)n * D.L + i, s * D.scale);
        }
        __syncthreads();
    }
}

// ---- phase 3: text = a . txt;  dep' / map' through the S tables -------------------------------------
// Workgroup item = (row n, 16 outputs): 16 lanes per output split the summation axis, so the longest dependent chain
// is L / 16 loads (the first version gave 4 lanes an output: L / 4 = 20-50 dependent strided loads, 13.6 us).
__device__ __forceinline__ int phase3_chunks(const Desc& D) { return (D.Ct + D.d_out + D.m_out) / 16; }

__device__ void phase3(const Desc& D, int wg, int nwg) {
    __shared__ float a[MAX_L];
    __shared__ float red[8];
    __shared__ float pl[16][17];
    __shared__ float ad[32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_txt = D.Ct / 16, n_dep = D.d_out / 16;
    const int chunks = phase3_chunks(D);
    const int P = D.P;
    for (int it = wg; it < D.rows * chunks; it += nwg) {
        const int n = it / chunks, ch = it - n * chunks;
        // softmax over the instruction axis (every item of row n repeats it: L <= 512 values)
        const float* lg = ws_logits(D) + (int64_t)n * D.L;
        float l0 = tid < D.L ? lg[tid] : -INFINITY, l1 = tid + CT < D.L ? lg[tid + CT] : -INFINITY;
        float m = wave_max(fmaxf(l0, l1));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float e0 = tid < D.L ? expf(l0 - m) : 0.f, e1 = tid + CT < D.L ? expf(l1 - m) : 0.f;
        float s = wave_sum(e0 + e1);
        if (lane == 0) red[4 + wave] = s;
        __syncthreads();
        const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
        if (tid < D.L) a[tid] = e0 * inv;
        if (tid + CT < D.L) a[tid + CT] = e1 * inv;
        __syncthreads();
        const int co = tid >> 4, part = tid & 15;  // output within the chunk, 1 of 16 lanes of that output
        float acc = 0.f;
        int dst;
        if (ch < n_txt) {
            const int c = ch * 16 + co;
            const float* tp = D.txt + ((int64_t)n * D.Ct + c) * D.L;
#pragma unroll 4
            for (int i = part; i < D.L; i += 16) acc = fmaf(a[i], tp[i], acc);
            dst = D.H + c;
        } else {
            const int sset = ch < n_txt + n_dep ? 0 : 1;
            const int cch = sset == 0 ? ch - n_txt : ch - n_txt - n_dep;
            {   // logits over the P grid positions: sum_i a_i S[i][p]   (thread = (position p, 1 of 16 slices of i))
                const int p = part, ip = co;
                float t = 0.f;
                if (p < P) {
                    const float* S = ws_S(D) + (int64_t)n * D.L * 2 * P + sset * P + p;
#pragma unroll 4
                    for (int i = ip; i < D.L; i += 16) t = fmaf(a[i], S[(int64_t)i * 2 * P], t);
                }
                pl[ip][p] = t;
            }
            __syncthreads();
            if (tid < 64) {
                float l = -INFINITY;
                if (tid < P) {
                    float t = 0.f;
#pragma unroll
                    for (int u = 0; u < 16; ++u) t += pl[u][tid];
                    l = t * D.scale;
                }
                const float mx = wave_max(l);
                const float e = tid < P ? expf(l - mx) : 0.f;
                const float sm = wave_sum(e);
                if (tid < P) ad[tid] = e * (1.f / sm);
            }
            __syncthreads();
            const int Ck = D.Hq, Cv = sset == 0 ? D.d_out : D.m_out;
            const float* kv = sset == 0 ? D.dkv : D.mkv;
            const int c = cch * 16 + co;
            if (part < P) acc = ad[part] * kv[((int64_t)n * (Ck + Cv) + Ck + c) * P + part];
            dst = D.H + D.Ct + (sset == 0 ? 0 : D.d_out) + c;
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 8, 64);
        if (part == 0) st_pub(D.x2 + (int64_t)n * D.x2w + dst, acc);
        __syncthreads();
    }
}

// ---- phase 4: c2 = ReLU(W_c x2 + b_c)  (workgroup item = output o) -------------------------------------
template <int LPR>
__device__ void phase4(const Desc& D, int wg, int nwg) {
    constexpr int RPB = CT / LPR;
    const int l = threadIdx.x % LPR, rr = threadIdx.x / LPR;
    for (int o = wg; o < D.H; o += nwg) {
        for (int r0 = 0; r0 < D.rows; r0 += RPB) {
            const int row = r0 + rr;
            const bool ok = row < D.rows;
            float v = lpr_dot<LPR>(D.w_c + (int64_t)o * D.x2w, D.x2 + (int64_t)(ok ? row : 0) * D.x2w, D.x2w, l);
            v = lpr_sum<LPR>(v);
            if (l == 0 && ok) st_pub(ws_c2(D) + (int64_t)row * D.H + o, fmaxf(v + D.b_c[o], 0.f));
        }
    }
}

// ---- phase 5: GRU-2 (input half here, hidden half from phase 1) ----------------------------------------
template <int LPR>
__device__ void phase5(const Desc& D, int wg, int nwg) {
    for (int j = wg; j < D.H; j += nwg)
        gru_unit<LPR>(D, j, ws_c2(D), D.H, D.H, D.w_ih2, D.w_hh2, D.b_ih2, D.b_hh2, D.h_in + D.H, ws_gh2(D), D.feats, D.H,
                      D.h_out + D.H, D.ld_ho);
}

// One kernel per phase.  Work that is off the chain shares a launch with the phase it can hide behind: the first
// `main` workgroups run the phase, the rest the side work.
template <int PH, int LPR>
__global__ __launch_bounds__(CT) void k_cma_phase(const Desc D, int main) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    if constexpr (PH == 1) phase1<LPR>(D, wg, nwg);
    if constexpr (PH == 2) {
        if (wg < main) phase2(D, wg, main);
        else side_S(D, wg - main, nwg - main);
    }
    if constexpr (PH == 3) {
        if (wg < main) phase3(D, wg, main);
        else side_gh2<LPR>(D, wg - main, nwg - main);
    }
    if constexpr (PH == 4) phase4<LPR>(D, wg, nwg);
    if constexpr (PH == 5) phase5<LPR>(D, wg, nwg);
}

template <int LPR>
void launch_cma(const Desc& D, int mode, hipStream_t s) {
    const int tiles = (D.L + 15) / 16;
    const int chunks = (D.Ct + D.d_out + D.m_out) / 16;
    (void)mode;
    hipLaunchKernelGGL((k_cma_phase<1, LPR>), dim3(D.H), dim3(CT), 0, s, D, D.H);
    hipLaunchKernelGGL((k_cma_phase<2, LPR>), dim3(D.rows * tiles + D.rows * D.L), dim3(CT), 0, s, D, D.rows * tiles);
    hipLaunchKernelGGL((k_cma_phase<3, LPR>), dim3(D.rows * chunks + D.H), dim3(CT), 0, s, D, D.rows * chunks);
    hipLaunchKernelGGL((k_cma_phase<4, LPR>), dim3(D.H), dim3(CT), 0, s, D, D.H);
    hipLaunchKernelGGL((k_cma_phase<5, LPR>), dim3(D.H), dim3(CT), 0, s, D, D.H);
}

}  // namespace

extern "C" {

int64_t ivln_cma_step_ws_floats(int rows, int L, int P, int H) {
    // logits (rows*L) + S (rows*L*2P) + gh2 (rows*3H) + c2 (rows*H) + 16 floats for {barrier counter, error word}
    return al32((int64_t)rows * L) + al32((int64_t)rows * L * 2 * P) + al32((int64_t)rows * 3 * H) + al32((int64_t)rows * H);
}

int ivln_cma_step_fwd(const ivln_cma_step_desc* d, int mode, void* stream) {
    if (!d || d->rows <= 0 || !d->ws || !d->x2 || !d->feats) return IVLN_E_INVALID;
    if (d->L <= 0 || d->L > MAX_L || d->P <= 0 || d->P > 16) return IVLN_E_UNSUPPORTED;
    if ((d->H & 63) || (d->Hq & 15) || (d->Ct & 15) || (d->d_out & 15) || (d->m_out & 15)) return IVLN_E_UNSUPPORTED;
    const int sin_w = d->d_out + d->m_out + d->E;
    if ((sin_w & 3) || (d->x2w & 3) || (d->ld_h & 3) || d->x2w != d->H + d->Ct + d->d_out + d->m_out + d->E)
        return IVLN_E_UNSUPPORTED;
    Desc D = *d;
    if (D.Mq_img <= 0) D.Mq_img = (int64_t)(D.H + 1) * D.L;
    if (D.TQb_img <= 0) D.TQb_img = (int64_t)D.Hq * D.L;
    hipStream_t s = (hipStream_t)stream;
    if (D.rows <= 4) launch_cma<64>(D, mode, s);
    else if (D.rows <= 8) launch_cma<32>(D, mode, s);
    else launch_cma<16>(D, mode, s);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

}  // extern "C"
