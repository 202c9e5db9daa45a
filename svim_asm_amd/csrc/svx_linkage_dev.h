// svx_linkage_dev.h — device routine shared by svx_linkage.hip and svx_postpass.hip: complete linkage +
// flat cut of ONE partition by one lane, in scipy's label order (see svx_linkage.hip for the procedure
// and the reference call sites).  `mem` is svx_link_bytes(n) bytes of 8-byte aligned scratch (LDS or HBM).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__host__ __device__ constexpr size_t svx_link_bytes(uint32_t n) {
    return n < 2 ? 0
                 : 8 * ((size_t)n * (n - 1) / 2 + 2 * ((size_t)n - 1)) + 4 * (7 * (size_t)n - 3) + ((2 * (size_t)n - 1 + 7) / 8) * 8;
}
static __device__ __forceinline__ size_t cidx(uint32_t n, uint32_t i, uint32_t j) {
    if (i > j) { const uint32_t t = i; i = j; j = t; }
    return (size_t)n * i - (size_t)i * (i + 1) / 2 + (j - i - 1);
}

static __device__ void svx_linkage_cut_one(const uint32_t n, const double* __restrict__ cond, const double cutoff,
                                uint32_t* __restrict__ labels, char* mem) {
    if (n == 0) return;
    if (n == 1) { labels[0] = 1; return; }
    const size_t m = (size_t)n * (n - 1) / 2;
    double* D = reinterpret_cast<double*>(mem);
    double* zd = D + m;
    double* md = zd + (n - 1);
    int* size = reinterpret_cast<int*>(md + (n - 1));
    int* chain = size + n;
    int* zx = chain + n;
    int* zy = zx + (n - 1);
    int* parent = zy + (n - 1);
    int* stack = parent + (2 * n - 1);
    unsigned char* visited = reinterpret_cast<unsigned char*>(stack + n);
    for (size_t i = 0; i < m; ++i) D[i] = cond[i];
    for (uint32_t i = 0; i < n; ++i) size[i] = 1;
    // ---- nearest-neighbour chain
    int chain_len = 0;
    for (uint32_t k = 0; k + 1 < n; ++k) {
        int x = 0, y = 0;
        double cur = 0;
        if (chain_len == 0) {
            chain_len = 1;
            for (uint32_t i = 0; i < n; ++i)
                if (size[i] > 0) { chain[0] = (int)i; break; }
        }
        for (;;) {
            x = chain[chain_len - 1];
            if (chain_len > 1) {
                y = chain[chain_len - 2];
                cur = D[cidx(n, (uint32_t)x, (uint32_t)y)];
            } else {
                cur = __builtin_huge_val();
            }
            for (uint32_t i = 0; i < n; ++i) {
                if (size[i] == 0 || (int)i == x) continue;
                const double d = D[cidx(n, (uint32_t)x, i)];
                if (d < cur) { cur = d; y = (int)i; }
            }
            if (chain_len > 1 && y == chain[chain_len - 2]) break;
            chain[chain_len++] = y;
        }
        chain_len -= 2;
        if (x > y) { const int t = x; x = y; y = t; }
        const int nx = size[x], ny = size[y];
        zx[k] = x; zy[k] = y; zd[k] = cur;
        size[x] = 0;
        size[y] = nx + ny;
        for (uint32_t i = 0; i < n; ++i) {
            if (size[i] == 0 || (int)i == y) continue;
            const double a = D[cidx(n, i, (uint32_t)x)], b = D[cidx(n, i, (uint32_t)y)];
            D[cidx(n, i, (uint32_t)y)] = a > b ? a : b;
        }
    }
    // ---- stable sort of the merges by distance
    for (uint32_t i = 1; i + 1 < n; ++i) {
        const int tx = zx[i], ty = zy[i];
        const double td = zd[i];
        uint32_t j = i;
        while (j > 0 && zd[j - 1] > td) { zx[j] = zx[j - 1]; zy[j] = zy[j - 1]; zd[j] = zd[j - 1]; --j; }
        zx[j] = tx; zy[j] = ty; zd[j] = td;
    }
    // ---- union-find relabelling: cluster ids n, n+1, ... in sorted order
    for (uint32_t i = 0; i < 2 * n - 1; ++i) { parent[i] = (int)i; visited[i] = 0; }
    int next = (int)n;
    for (uint32_t i = 0; i + 1 < n; ++i) {
        int r0 = zx[i], r1 = zy[i];
        {
            int p = r0, root = r0;
            while (parent[root] != root) root = parent[root];
            while (parent[p] != root) { const int q = parent[p]; parent[p] = root; p = q; }
            r0 = root;
        }
        {
            int p = r1, root = r1;
            while (parent[root] != root) root = parent[root];
            while (parent[p] != root) { const int q = parent[p]; parent[p] = root; p = q; }
            r1 = root;
        }
        zx[i] = r0 < r1 ? r0 : r1;
        zy[i] = r0 < r1 ? r1 : r0;
        parent[r0] = next;
        parent[r1] = next;
        ++next;
    }
    // ---- maximum distance below every internal node (children are earlier rows)
    for (uint32_t i = 0; i + 1 < n; ++i) {
        double v = zd[i];
        if (zx[i] >= (int)n && md[zx[i] - (int)n] > v) v = md[zx[i] - (int)n];
        if (zy[i] >= (int)n && md[zy[i] - (int)n] > v) v = md[zy[i] - (int)n];
        md[i] = v;
    }
    // ---- flat clusters
    int kk = 0, n_cluster = 0, leader = -1;
    stack[0] = 2 * (int)n - 2;
    while (kk >= 0) {
        const int root = stack[kk] - (int)n;
        const int lc = zx[root], rc = zy[root];
        if (leader == -1 && md[root] <= cutoff) { leader = root; ++n_cluster; }
        if (lc >= (int)n && !visited[lc]) { visited[lc] = 1; stack[++kk] = lc; continue; }
        if (rc >= (int)n && !visited[rc]) { visited[rc] = 1; stack[++kk] = rc; continue; }
        if (lc < (int)n) { if (leader == -1) ++n_cluster; labels[lc] = (uint32_t)n_cluster; }
        if (rc < (int)n) { if (leader == -1) ++n_cluster; labels[rc] = (uint32_t)n_cluster; }
        if (leader == root) leader = -1;
        --kk;
    }
}

