// Fused softmax(q k^T / sqrt(dh)) v for gfx950: one kernel per attention, no score matrix in HBM.
//
// Replaces the per-head `attention_scores -> softmax -> bmm` of the diffusers attention the MuseTalk UNet
// runs through (reference: musetalk/models/unet.py:36-47 loads UNet2DConditionModel with
// BasicTransformerBlock.attn1 / attn2) and the Whisper encoder's qkv_attention (reference:
// musetalk/whisper/whisper/model.py:62-93).
//
// Layout of the work (wave64, MFMA 16x16x32 bf16, fp32 accumulate):
//   * one workgroup = 4 waves = 4 x (16*QB) queries of one (batch, head); it walks the keys in tiles of KT.
//   * S^T = K Q^T: the key tile is the MFMA A operand, so a lane ends up with 4 consecutive KEYS of ONE query
//     (column l%16).  Row max / row sum are then 4 in-lane values + two cross-lane steps (xor 16, 32).
//   * the lane's probabilities are already the B operand of O^T = V^T P^T: the contraction index of that MFMA is
//     defined as (8g+j) <-> key 16*(2i + j/4) + 4g + j%4, so P never goes through LDS.
//   * V^T fragments come from the row-major V tile in LDS through ds_read_b64_tr_b16 (4 keys x 16 channels per
//     16-lane group, delivered transposed).
//   * MF_PREC_BF16X3: q, k, v and p are (hi, lo) bf16 pairs and every product is lo*hi + hi*lo + hi*hi.
//   * K/V tiles are staged global -> registers -> LDS; the loads of tile t+1 are in flight under the MFMAs of tile t.
#include "mf_nn.h"
#include <cmath>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct AttnArgs {
    const bf16_t *q_hi, *q_lo, *k_hi, *k_lo, *v_hi, *v_lo;
    bf16_t *o_hi, *o_lo;
    int64_t q_b, k_b, v_b, o_b;       // elements per batch item
    int q_row, k_row, v_row, o_row;   // elements per token
    int Tq, Tk, heads, qtiles, total;
    float scale_log2e;                // dh^-0.5 * log2(e): softmax runs on exp2
};

namespace {

__device__ __forceinline__ uint32_t f2bf_rne(float f) {
    return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);      // round to nearest even in hardware (v_cvt_pk_bf16_f32)
}
__device__ __forceinline__ float bf2f_(uint32_t h16) { return __uint_as_float(h16 << 16); }

__device__ __forceinline__ s16x4 lds_tr16(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

template <int DH, int QB, int KT, bool X3>
__global__ __launch_bounds__(256) void k_attention(const AttnArgs a) {
    constexpr int NP = X3 ? 2 : 1;
    constexpr int NK32 = (DH + 31) / 32;         // 32-deep QK^T steps; channels [DH, NK32*32) are zero on both operands
    constexpr int DHP = NK32 * 32 + 8;           // LDS row pitch (elements); rows stay 16-byte aligned
    constexpr int MB = (DH + 15) / 16;           // 16-channel blocks of O^T
    constexpr int KB = KT / 16;                  // 16-key blocks of S^T per tile
    constexpr int CPR = DH / 8;                  // 16-byte chunks per K / V row
    constexpr int CH = KT * CPR;                 // chunks per (tensor, plane) tile
    constexpr int RND = (CH + 255) / 256;
    static_assert(DH % 8 == 0, "head dim");
    static_assert(KT % 32 == 0, "key tile");

    __shared__ __attribute__((aligned(16))) bf16_t Ks[NP][KT][DHP];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[NP][KT][DHP];

    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware order: the q-tiles of one (batch, head) share its K/V through one XCD's L2
    const int bid = blockIdx.x, nt = a.total;
    const int qq = nt >> 3, rr = nt & 7, xcd = bid & 7;
    const int t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int grp = t / a.qtiles, qt = t - grp * a.qtiles;
    const int b = grp / a.heads, h = grp - b * a.heads;

    const bf16_t* qp[NP] = {a.q_hi + b * a.q_b + h * DH};
    const bf16_t* kp[NP] = {a.k_hi + b * a.k_b + h * DH};
    const bf16_t* vp[NP] = {a.v_hi + b * a.v_b + h * DH};
    if constexpr (X3) {
        qp[1] = a.q_lo + b * a.q_b + h * DH;
        kp[1] = a.k_lo + b * a.k_b + h * DH;
        vp[1] = a.v_lo + b * a.v_b + h * DH;
    }

    // the pitch columns [DH, DHP) are read by the last QK^T step / the last V^T block and never staged: zero once
    for (int i = tid; i < NP * KT * (DHP - DH); i += 256) {
        const int r = i / (DHP - DH), c = DH + i - r * (DHP - DH);
        (&Ks[0][0][0])[r * DHP + c] = 0;
        (&Vs[0][0][0])[r * DHP + c] = 0;
    }

    // ---- Q fragments (B operand: lane = query l16, 8 consecutive channels 8g..8g+7 of each 32-step) ----
    const int q0 = qt * (64 * QB) + wave * (16 * QB);
    bf16x8 qf[NP][QB][NK32];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qrow = q0 + qb * 16 + l16;
        qrow = qrow < a.Tq ? qrow : a.Tq - 1;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bf16_t* src = qp[p] + (int64_t)qrow * a.q_row;
#pragma unroll
            for (int ks = 0; ks < NK32; ++ks) {
                const int d = ks * 32 + g * 8;
                const u32x4 z = {0u, 0u, 0u, 0u};
                u32x4 raw = (ks * 32 + 32 <= DH || d < DH) ? *reinterpret_cast<const u32x4*>(src + d) : z;
                // consumed here, so no Q load is still counted in vmcnt when the key loop starts (the compiler would
                // otherwise wait on the K/V prefetch of tile t+1 in front of the first MFMA of tile t)
                asm volatile("" : "+v"(raw));
                qf[p][qb][ks] = __builtin_bit_cast(bf16x8, raw);
            }
        }
    }

    // ---- K / V staging: global -> registers (prefetch) -> LDS ----
    u32x4 pre[2 * NP][RND];
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RND; ++r) {
            const int c = tid + r * 256;
            if (CH % 256 == 0 || c < CH) {
                const int row = c / CPR, col = c - row * CPR;
                const int key = tile * KT + row;
                const bool ok = key < a.Tk;
                const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    pre[p][r] = ok ? *reinterpret_cast<const u32x4*>(kp[p] + (int64_t)key * a.k_row + col * 8) : z;
                    pre[NP + p][r] = ok ? *reinterpret_cast<const u32x4*>(vp[p] + (int64_t)key * a.v_row + col * 8) : z;
                }
            }
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RND; ++r) {
            const int c = tid + r * 256;
            if (CH % 256 == 0 || c < CH) {
                const int row = c / CPR, col = c - row * CPR;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    *reinterpret_cast<u32x4*>(&Ks[p][row][col * 8]) = pre[p][r];
                    *reinterpret_cast<u32x4*>(&Vs[p][row][col * 8]) = pre[NP + p][r];
                }
            }
        }
    };

    f32x4 oacc[MB][QB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) oacc[mb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m_run[qb] = -INFINITY; l_run[qb] = 0.f; }

    const int ntiles = (a.Tk + KT - 1) / KT;
    load_tile(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();   // every wave is done with the previous tile's LDS image
        store_tile();
        __syncthreads();
        if (tile + 1 < ntiles) load_tile(tile + 1);

        // ---- S^T = K Q^T ----
        f32x4 s[KB][QB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) s[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int krow = kb * 16 + l16;
#pragma unroll
            for (int ks = 0; ks < NK32; ++ks) {
                const bf16x8 khi = *reinterpret_cast<const bf16x8*>(&Ks[0][krow][ks * 32 + g * 8]);
                bf16x8 klo;
                if constexpr (X3) klo = *reinterpret_cast<const bf16x8*>(&Ks[NP - 1][krow][ks * 32 + g * 8]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if constexpr (X3) {
                        s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(klo, qf[0][qb][ks], s[kb][qb], 0, 0, 0);
                        s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(khi, qf[NP - 1][qb][ks], s[kb][qb], 0, 0, 0);
                    }
                    s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(khi, qf[0][qb][ks], s[kb][qb], 0, 0, 0);
                }
            }
        }

        // ---- online softmax over this tile's keys (lane: query l16, keys kb*16 + 4g + i) ----
        const bool last = (tile + 1) * KT > a.Tk;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = s[kb][qb][i] * a.scale_log2e;
                    if (last && tile * KT + kb * 16 + g * 4 + i >= a.Tk) v = -INFINITY;
                    s[kb][qb][i] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qb], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f(s[kb][qb][i] - m_new);
                    s[kb][qb][i] = p;
                    ps += p;
                }
            l_run[qb] = l_run[qb] * alpha + ps;   // per-lane partial: the 4 key groups g are summed once at the end
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i) oacc[mb][qb][i] *= alpha;
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int pi = 0; pi < KB / 2; ++pi) {
            s16x8 phi[QB], plo[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float p = s[2 * pi + (j >> 2)][qb][j & 3];
                    const uint32_t hb = f2bf_rne(p);
                    phi[qb][j] = (short)hb;
                    if constexpr (X3) plo[qb][j] = (short)f2bf_rne(p - bf2f_(hb));
                }
            // V rows 32*pi + 4g + (l16 >> 2) (+16), channels 16*mb + 4*(l16 & 3): the transposing read hands lane l16
            // channel 16*mb + l16 of the 4 keys 32*pi + 4g .. +3
            const int vrow = pi * 32 + g * 4 + (l16 >> 2), vcol = (l16 & 3) * 4;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const s16x4 a0 = lds_tr16(&Vs[0][vrow][mb * 16 + vcol]);
                const s16x4 a1 = lds_tr16(&Vs[0][vrow + 16][mb * 16 + vcol]);
                const s16x8 vhi = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                s16x8 vlo;
                if constexpr (X3) {
                    const s16x4 b0 = lds_tr16(&Vs[NP - 1][vrow][mb * 16 + vcol]);
                    const s16x4 b1 = lds_tr16(&Vs[NP - 1][vrow + 16][mb * 16 + vcol]);
                    vlo = s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if constexpr (X3) {
                        oacc[mb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vlo), __builtin_bit_cast(bf16x8, phi[qb]), oacc[mb][qb], 0, 0, 0);
                        oacc[mb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vhi), __builtin_bit_cast(bf16x8, plo[qb]), oacc[mb][qb], 0, 0, 0);
                    }
                    oacc[mb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vhi), __builtin_bit_cast(bf16x8, phi[qb]), oacc[mb][qb], 0, 0, 0);
                }
            }
        }
    }

    // ---- O = O^T / l : lane holds channels 16*mb + 4g .. +3 of query l16 ----
    bf16_t* oh = a.o_hi + b * a.o_b + h * DH;
    bf16_t* ol = X3 ? a.o_lo + b * a.o_b + h * DH : nullptr;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l = l_run[qb];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
        const int qrow = q0 + qb * 16 + l16;
        if (qrow >= a.Tq) continue;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int d = mb * 16 + g * 4;
            if (d >= DH) continue;
            float v[4];
            uint32_t hb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = oacc[mb][qb][i] * inv; hb[i] = f2bf_rne(v[i]); }
            const int64_t o = (int64_t)qrow * a.o_row + d;
            *reinterpret_cast<u32x2*>(oh + o) = u32x2{hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
            if constexpr (X3) {
                uint32_t lb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) lb[i] = f2bf_rne(v[i] - bf2f_(hb[i]));
                *reinterpret_cast<u32x2*>(ol + o) = u32x2{lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
            }
        }
    }
}

template <int DH, int QB, int KT>
int launch_prec(const AttnArgs& a, bool x3, hipStream_t s) {
    if (x3)
        hipLaunchKernelGGL((k_attention<DH, QB, KT, true>), dim3(a.total), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_attention<DH, QB, KT, false>), dim3(a.total), dim3(256), 0, s, a);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

template <int DH, int KT>
int launch_qb(AttnArgs& a, int groups, bool x3, hipStream_t s) {
    // 32 queries per wave halve the K / V^T fragment reads per MFMA; take them when the chip still gets >= 2 waves of
    // workgroups, otherwise spread the queries over more workgroups
    // (round 5 A/Bs on the UNet's 1024 x 1024 self-attention, 64 us per launch: 16 instead of 32 queries per wave -- same step time; two LDS images and one barrier
    // per key tile instead of two -- same step time.  The loop is vector-bound: softmax + the (hi, lo) split of P are ~600 instructions + 34 exponentials per wave and
    // 64-key tile against 84 MFMAs.)
    const bool wide = (int64_t)groups * ((a.Tq + 127) / 128) >= 512;
    if (DH <= 80 && wide) {
        a.qtiles = (a.Tq + 127) / 128;
        a.total = groups * a.qtiles;
        return launch_prec<DH, (DH <= 80 ? 2 : 1), KT>(a, x3, s);
    }
    a.qtiles = (a.Tq + 63) / 64;
    a.total = groups * a.qtiles;
    return launch_prec<DH, 1, KT>(a, x3, s);
}

}  // namespace

bool mf_attention_supported(int dh) { return dh == 40 || dh == 64 || dh == 80 || dh == 160; }

int mf_attention(const ActView& q, const ActView& k, const ActView& v, const ActView& out, int heads, int batch, int precision,
                 hipStream_t s, int tq, int tk) {
    const int C = q.C, dh = C / heads;
    MF_REQUIRE(heads > 0 && C % heads == 0 && k.C == C && v.C == C && out.C == C, "attention: channel mismatch");
    MF_REQUIRE(mf_attention_supported(dh), "attention: no fused kernel for head dim %d", dh);
    // tokens must be contiguous rows: halo-free buffers, or single-row sequences (their interior is one contiguous run)
    for (const ActBuf* b : {q.buf, k.buf, v.buf, out.buf})
        MF_REQUIRE(b->halo == 0 || b->H == 1, "attention: needs contiguous token buffers");
    MF_REQUIRE(q.coff % 8 == 0 && k.coff % 8 == 0 && v.coff % 8 == 0 && out.coff % 8 == 0, "attention: views must start on 8-channel groups");
    const bool x3 = precision == MF_PREC_BF16X3;
    AttnArgs a{};
    const int64_t qo = mf_interior(*q.buf) + q.coff, ko = mf_interior(*k.buf) + k.coff, vo = mf_interior(*v.buf) + v.coff,
                  oo = mf_interior(*out.buf) + out.coff;
    a.q_hi = q.buf->hi + qo; a.k_hi = k.buf->hi + ko; a.v_hi = v.buf->hi + vo; a.o_hi = out.buf->hi + oo;
    if (x3) {
        a.q_lo = q.buf->lo + qo; a.k_lo = k.buf->lo + ko; a.v_lo = v.buf->lo + vo; a.o_lo = out.buf->lo + oo;
    }
    a.q_b = q.buf->per_batch(); a.k_b = k.buf->per_batch(); a.v_b = v.buf->per_batch(); a.o_b = out.buf->per_batch();
    a.q_row = q.buf->C; a.k_row = k.buf->C; a.v_row = v.buf->C; a.o_row = out.buf->C;
    a.Tq = q.buf->H * q.buf->W; a.Tk = k.buf->H * k.buf->W; a.heads = heads;
    MF_REQUIRE(a.Tq > 0 && a.Tk > 0 && v.buf->H * v.buf->W == a.Tk && out.buf->H * out.buf->W == a.Tq, "attention: token count mismatch");
    // sequence prefixes (Whisper's last block only needs the queries whose outputs are consumed; a shortened context): base
    // pointers and batch strides stay those of the full buffers
    MF_REQUIRE(tq >= 0 && tq <= a.Tq && tk >= 0 && tk <= a.Tk, "attention: prefix (%d queries, %d keys) exceeds the sequences (%d, %d)", tq, tk, a.Tq, a.Tk);
    if (tq > 0) a.Tq = tq;
    if (tk > 0) a.Tk = tk;
    a.scale_log2e = (float)(1.4426950408889634 / std::sqrt((double)dh));
    const int groups = batch * heads;
    switch (dh) {
        case 40: return launch_qb<40, 64>(a, groups, x3, s);
        case 64: return launch_qb<64, 64>(a, groups, x3, s);
        case 80: return launch_qb<80, 64>(a, groups, x3, s);
        case 160: return launch_qb<160, 32>(a, groups, x3, s);
    }
    return MF_ERR_INVALID;
}

// ---- C ABI: the fused attention on fp32 [B][T][heads*dh] device tensors (test seam, like mf_conv2d_*) ----
namespace {
struct TokBuf {
    ActBuf b;
    ~TokBuf() {
        if (b.hi) (void)hipFree(b.hi);
        if (b.lo) (void)hipFree(b.lo);
    }
    int alloc(int C, int T, int batch, bool x3) {
        b.C = C; b.H = 1; b.W = T; b.halo = 0;
        const size_t bytes = ((size_t)batch * b.per_batch() + 64) * sizeof(bf16_t);
        MF_HIP(hipMalloc(&b.hi, bytes));
        MF_HIP(hipMemset(b.hi, 0, bytes));
        if (x3) {
            MF_HIP(hipMalloc(&b.lo, bytes));
            MF_HIP(hipMemset(b.lo, 0, bytes));
        }
        return MF_OK;
    }
};
}  // namespace

extern "C" int mf_attention_forward(const float* q, const float* k, const float* v, float* out, int batch, int tq, int tk,
                                    int heads, int head_dim, int precision, void* stream) {
    MF_REQUIRE(q && k && v && out, "attention_forward: null argument");
    MF_REQUIRE(batch > 0 && tq > 0 && tk > 0 && heads > 0, "attention_forward: batch=%d tq=%d tk=%d heads=%d", batch, tq, tk, heads);
    MF_REQUIRE(precision == MF_PREC_BF16 || precision == MF_PREC_BF16X3, "attention_forward: unknown precision %d", precision);
    MF_REQUIRE(mf_attention_supported(head_dim), "attention_forward: head_dim %d has no fused kernel (40, 64, 80, 160)", head_dim);
    hipStream_t s = (hipStream_t)stream;
    const bool x3 = precision == MF_PREC_BF16X3;
    const int C = heads * head_dim;
    TokBuf bq, bk, bv, bo;
    int rc;
    if ((rc = bq.alloc(C, tq, batch, x3)) || (rc = bk.alloc(C, tk, batch, x3)) || (rc = bv.alloc(C, tk, batch, x3)) ||
        (rc = bo.alloc(C, tq, batch, x3)))
        return rc;
    MF_HIP(hipDeviceSynchronize());
    if ((rc = mf_rows_from_f32(q, nullptr, ActView{&bq.b, 0, C}, batch, s))) return rc;
    if ((rc = mf_rows_from_f32(k, nullptr, ActView{&bk.b, 0, C}, batch, s))) return rc;
    if ((rc = mf_rows_from_f32(v, nullptr, ActView{&bv.b, 0, C}, batch, s))) return rc;
    if ((rc = mf_attention(ActView{&bq.b, 0, C}, ActView{&bk.b, 0, C}, ActView{&bv.b, 0, C}, ActView{&bo.b, 0, C}, heads, batch, precision, s)))
        return rc;
    if ((rc = mf_rows_to_f32(ActView{&bo.b, 0, C}, out, batch, s))) return rc;
    MF_HIP(hipStreamSynchronize(s));
    return MF_OK;
}
