// Per-batch glue of the MuseTalk render loop on the device (musereal.py:91-108): the cached avatar latents and the
// Whisper features stay in HBM; a batch is assembled by two gathers instead of `torch.cat` of B host-indexed tensors + `np.stack` + H2D.
//   * latents: out[i] = pool[rows[i]]            rows = mirror index of each (session, frame) into a pool of cached latents
//                                                 (musereal.py:92-97, `__mirror_index` :44-50)
//   * audio:   out[i] = feat[rows of chunk i]    Audio2Feature.get_sliced_feature (audio2feature.py:16-45): 10 consecutive 20-ms
//                                                 feature rows, clamped to the window, each (n_layer + 1) x 384 -> one (50, 384) chunk
// HBM-bound copies: 32 KB per latent, 75 KB per chunk; indices travel as kernel arguments (no host buffer to keep alive, capturable).
#include "mf_common.h"

namespace {

constexpr int MAX_ROWS = 256;
struct GatherArgs {
    const float* src; float* dst;
    int64_t row_elems;      // elements per gathered row (multiple of 4)
    int n;
    int rows[MAX_ROWS];
};

__global__ __launch_bounds__(256) void k_gather_rows(const GatherArgs a) {
    const int i = blockIdx.y;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= a.row_elems) return;
    const float4 v = *reinterpret_cast<const float4*>(a.src + (int64_t)a.rows[i] * a.row_elems + q * 4);
    *reinterpret_cast<float4*>(a.dst + (int64_t)i * a.row_elems + q * 4) = v;
}

constexpr int MAX_CHUNKS = 128;
struct ChunkArgs {
    const float* feat; float* dst;
    int T, row_elems, rows_per_chunk, n;
    int left[MAX_CHUNKS];   // first (unclamped) feature row of each chunk
};

__global__ __launch_bounds__(256) void k_feature_chunks(const ChunkArgs a) {
    const int i = blockIdx.y, j = blockIdx.z;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= a.row_elems) return;
    int r = a.left[i] + j;
    r = r < 0 ? 0 : (r > a.T - 1 ? a.T - 1 : r);          // min(length - 1, max(0, idx)), audio2feature.py:36-38
    const float4 v = *reinterpret_cast<const float4*>(a.feat + (int64_t)r * a.row_elems + q * 4);
    *reinterpret_cast<float4*>(a.dst + ((int64_t)i * a.rows_per_chunk + j) * a.row_elems + q * 4) = v;
}

}  // namespace

extern "C" int mf_gather_rows_f32(const float* pool, int n_pool_rows, int64_t row_elems, const int* rows, int n, float* out, void* stream) {
    MF_REQUIRE(pool && rows && out && n > 0 && row_elems > 0 && row_elems % 4 == 0, "gather_rows: bad argument (row_elems must be a multiple of 4)");
    for (int i = 0; i < n; ++i) MF_REQUIRE(rows[i] >= 0 && rows[i] < n_pool_rows, "gather_rows: row %d = %d out of range (%d rows)", i, rows[i], n_pool_rows);
    hipStream_t s = (hipStream_t)stream;
    for (int i0 = 0; i0 < n; i0 += MAX_ROWS) {
        GatherArgs a{};
        a.src = pool; a.dst = out + (int64_t)i0 * row_elems; a.row_elems = row_elems;
        a.n = n - i0 < MAX_ROWS ? n - i0 : MAX_ROWS;
        for (int i = 0; i < a.n; ++i) a.rows[i] = rows[i0 + i];
        const unsigned gx = (unsigned)((row_elems / 4 + 255) / 256);
        hipLaunchKernelGGL(k_gather_rows, dim3(gx, a.n), dim3(256), 0, s, a);
        MF_HIP(hipGetLastError());
    }
    return MF_OK;
}

extern "C" int mf_whisper_feature_chunks(const float* feat, int T, int row_elems, const int* left_rows, int rows_per_chunk, int n_chunks, float* out,
                                         void* stream) {
    MF_REQUIRE(feat && left_rows && out && T > 0 && n_chunks > 0 && rows_per_chunk > 0 && rows_per_chunk <= 64 && row_elems > 0 && row_elems % 4 == 0,
               "whisper_feature_chunks: bad argument");
    hipStream_t s = (hipStream_t)stream;
    for (int i0 = 0; i0 < n_chunks; i0 += MAX_CHUNKS) {
        ChunkArgs a{};
        a.feat = feat; a.dst = out + (int64_t)i0 * rows_per_chunk * row_elems; a.T = T; a.row_elems = row_elems; a.rows_per_chunk = rows_per_chunk;
        a.n = n_chunks - i0 < MAX_CHUNKS ? n_chunks - i0 : MAX_CHUNKS;
        for (int i = 0; i < a.n; ++i) a.left[i] = left_rows[i0 + i];
        hipLaunchKernelGGL(k_feature_chunks, dim3((row_elems / 4 + 255) / 256, a.n, rows_per_chunk), dim3(256), 0, s, a);
        MF_HIP(hipGetLastError());
    }
    return MF_OK;
}
