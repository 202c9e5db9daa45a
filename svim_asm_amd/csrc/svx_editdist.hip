// svx_editdist.hip — batched global unit-cost edit distance on gfx950.
//
// Arithmetic of edlib.align(a, b)["editDistance"] (default mode NW) as called by
// compute_distance (reference SVIM_COMBINE.py:50,64,76,88,100).  The pairing step only
// needs "distance <= max_edit_distance or not" (complete linkage cut), so
// the kernel is a thresholded (Ukkonen-banded) Needleman-Wunsch:
//   one workgroup per pair; the band |j - i| <= k lives in LDS as ONE array V[d]
//   (d = j - i); anti-diagonal s = i + j is updated in place — cells of one anti-diagonal
//   are independent, read only the opposite-parity neighbours V[d±1] (anti-diagonal s-1)
//   and their own previous value (s-2) — one __syncthreads per anti-diagonal.
// Result: exact distance when <= k, 0xFFFFFFFF otherwise.  k_max = 0xFFFFFFFF ("exact") is
// served by re-running unresolved pairs with a 4x wider band until the band covers the
// longer sequence.  Bytes compare exactly (the reference does not fold case of INS alleles).
#include "svx_internal.h"

#include <algorithm>

namespace {

constexpr uint32_t kInf = 0x3FFFFFFFu;
constexpr uint32_t kMaxBand = 16000;  // (2k+3) * 4 B must fit in 160 KiB of LDS

struct EdArgs {
    const uint8_t* seq;
    const uint64_t* a_off;
    const uint32_t* a_len;
    const uint64_t* b_off;
    const uint32_t* b_len;
    const uint32_t* sel;  // nullable: indices of the pairs to (re)compute
    uint32_t n;
    uint32_t k;
    uint32_t* dist;
};

__global__ __launch_bounds__(256) void k_edit_band(EdArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t V[];  // V[0 .. 2k+2], entry d+k+1
    const int tid = threadIdx.x;
    const int k = (int)p.k;
    for (uint32_t w = blockIdx.x; w < p.n; w += gridDim.x) {
        const uint32_t pi = p.sel ? p.sel[w] : w;
        const uint8_t* a = p.seq + p.a_off[pi];
        const uint8_t* b = p.seq + p.b_off[pi];
        const int la = (int)p.a_len[pi], lb = (int)p.b_len[pi];
        const int diff = lb - la;
        if (diff > k || -diff > k) {  // distance >= |la - lb| > k
            if (tid == 0) p.dist[pi] = 0xFFFFFFFFu;
            continue;
        }
        for (int i = tid; i < 2 * k + 3; i += blockDim.x) V[i] = kInf;
        __syncthreads();
        const int S = la + lb;
        for (int s = 0; s <= S; ++s) {
            // cells (i, j) with i + j = s, d = j - i ≡ s (mod 2), |d| <= k, 0<=i<=la, 0<=j<=lb
            int dlo = -k, dhi = k;
            if (dlo < -s) dlo = -s;             // j >= 0 → d >= -s ... (i <= s)
            if (dhi > s) dhi = s;               // i >= 0 → d <= s
            if (dlo < s - 2 * la) dlo = s - 2 * la;  // i <= la → d >= s - 2 la
            if (dhi > 2 * lb - s) dhi = 2 * lb - s;  // j <= lb → d <= 2 lb - s
            if (((dlo - s) & 1) != 0) ++dlo;    // parity of d must equal parity of s
            for (int d = dlo + 2 * tid; d <= dhi; d += 2 * (int)blockDim.x) {
                const int i = (s - d) >> 1, j = (s + d) >> 1;
                uint32_t v;
                if (i == 0) v = (uint32_t)j;
                else if (j == 0) v = (uint32_t)i;
                else {
                    const uint32_t diag = V[d + k + 1] + (a[i - 1] != b[j - 1] ? 1u : 0u);
                    const uint32_t up = V[d + k + 2] + 1u;    // (i-1, j): diagonal d+1
                    const uint32_t left = V[d + k] + 1u;      // (i, j-1): diagonal d-1
                    v = diag < up ? diag : up;
                    v = v < left ? v : left;
                    if (v > kInf) v = kInf;
                }
                V[d + k + 1] = v;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const uint32_t v = V[diff + k + 1];
            p.dist[pi] = (v <= (uint32_t)k) ? v : 0xFFFFFFFFu;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int svx_edit_distance_batch(svx_ctx* ctx, const uint8_t* seq, uint64_t seq_bytes,
                                       const uint64_t* a_off, const uint32_t* a_len,
                                       const uint64_t* b_off, const uint32_t* b_len, uint32_t n_pairs,
                                       uint32_t k_max, uint32_t* dist) {
    if (!ctx) return SVX_E_INVALID;
    if (n_pairs == 0) return SVX_OK;
    if (!a_off || !a_len || !b_off || !b_len || !dist || (seq_bytes && !seq)) return SVX_E_INVALID;
    const bool exact = (k_max == 0xFFFFFFFFu);
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (a_off[i] + a_len[i] > seq_bytes || b_off[i] + b_len[i] > seq_bytes) {
            SVX_SET_ERR(ctx, "pair %u reads past the sequence pool", i);
            return SVX_E_INVALID;
        }
        max_len = std::max(max_len, std::max(a_len[i], b_len[i]));
    }
    if (!exact && k_max > kMaxBand) {
        if (max_len <= kMaxBand) k_max = max_len;  // a band as wide as the longest sequence is exact
        else {
            SVX_SET_ERR(ctx, "k_max=%u exceeds the LDS band limit %u", k_max, kMaxBand);
            return SVX_E_TOO_LARGE;
        }
    }
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    size_t need = svx_take_bytes(seq_bytes ? seq_bytes : 1, 1) + 2 * svx_take_bytes(n_pairs, 8) +
                  4 * svx_take_bytes(n_pairs, 4);
    int rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    EdArgs a;
    uint8_t* d_seq = svx_stage_take<uint8_t>(ctx, seq_bytes ? seq_bytes : 1);
    uint64_t* d_ao = svx_stage_take<uint64_t>(ctx, n_pairs);
    uint64_t* d_bo = svx_stage_take<uint64_t>(ctx, n_pairs);
    uint32_t* d_al = svx_stage_take<uint32_t>(ctx, n_pairs);
    uint32_t* d_bl = svx_stage_take<uint32_t>(ctx, n_pairs);
    uint32_t* d_sel = svx_stage_take<uint32_t>(ctx, n_pairs);
    uint32_t* d_dist = svx_stage_take<uint32_t>(ctx, n_pairs);
    if (seq_bytes) SVX_HIP(ctx, hipMemcpyAsync(d_seq, seq, seq_bytes, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_ao, a_off, (size_t)n_pairs * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_bo, b_off, (size_t)n_pairs * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_al, a_len, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_bl, b_len, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
    a.seq = d_seq; a.a_off = d_ao; a.a_len = d_al; a.b_off = d_bo; a.b_len = d_bl;
    a.dist = d_dist;
    a.sel = nullptr;
    a.n = n_pairs;
    // a band wider than the longer sequence adds nothing: clip it (keeps the LDS footprint small)
    uint32_t k = exact ? std::min<uint32_t>(256u, std::max<uint32_t>(max_len, 1u))
                       : std::min<uint32_t>(k_max, std::max<uint32_t>(max_len, 1u));
    std::vector<uint32_t> sel;
    SVX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_edit_band),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (;;) {
        a.k = k;
        const size_t lds = ((size_t)2 * k + 3) * 4;
        uint32_t grid = std::min<uint32_t>(a.n, (uint32_t)ctx->n_cu * 8u);
        hipLaunchKernelGGL(k_edit_band, dim3(grid), dim3(256), lds, ctx->stream, a);
        SVX_HIP(ctx, hipGetLastError());
        SVX_HIP(ctx, hipMemcpyAsync(dist, d_dist, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!exact || k >= max_len) {
            if (!exact && k < k_max)  // band == longest sequence: every distance is exact, none exceeds k_max
                for (uint32_t i = 0; i < n_pairs; ++i)
                    if (dist[i] == 0xFFFFFFFFu) { SVX_SET_ERR(ctx, "internal: clipped band missed a distance"); return SVX_E_INVALID; }
            break;
        }
        sel.clear();
        for (uint32_t i = 0; i < n_pairs; ++i)
            if (dist[i] == 0xFFFFFFFFu) sel.push_back(i);
        if (sel.empty()) break;
        uint32_t nk = std::min<uint32_t>(std::max<uint32_t>(k * 4, 1u), max_len);
        if (nk > kMaxBand) {
            SVX_SET_ERR(ctx, "exact distance needs a band of %u > LDS limit %u", nk, kMaxBand);
            return SVX_E_TOO_LARGE;
        }
        k = nk;
        SVX_HIP(ctx, hipMemcpyAsync(d_sel, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        a.sel = d_sel;
        a.n = (uint32_t)sel.size();
    }
    return SVX_OK;
}
